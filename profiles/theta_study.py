import sys; sys.path.insert(0,'.')
import numpy as np
from nbody_amd import Engine, physics
n = 1 << 20
b = physics.plummer(n, seed=1)
pos = b.pos.astype(np.float32).astype(np.float64); mass = b.mass.astype(np.float32).astype(np.float64)
with Engine(mode="direct", precision="f64") as e:
    e.upload(pos, b.vel, mass)
    exact, _ = e.accel()
def relerr(a, r): return np.linalg.norm(a - r, axis=1) / np.linalg.norm(r, axis=1)
for pw, thetas in ((False, (0.5, 0.7, 1.0)), (True, (0.5, 0.6, 0.7, 0.8, 1.0))):
    for th in thetas:
        with Engine(mode="tree", precision="f32", theta=th, per_wave=pw) as e:
            e.upload(pos, b.vel, mass)
            acc, _ = e.accel()
            it = e.stats().interactions
            e.step(2); e.sync()
            ms = min(e.step_timed().ms_force for _ in range(4))
        er = relerr(acc, exact)
        print("per_wave" if pw else "per_body", "theta %.1f" % th, "walk %.3f ms" % ms, "int/body %.0f" % (it / n),
              "err median %.2e p99 %.2e max %.2e" % (np.median(er), np.quantile(er, 0.99), er.max()), flush=True)
