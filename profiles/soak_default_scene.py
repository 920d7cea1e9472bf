"""The reference's default scene (main.go:32-40: 1,000 random bodies and two cores, collisions on, theta 1, an hour per
step) for thousands of steps on the distributed tree behind one handle (three ranks on this GPU) and on one engine, fp64:
survivors, merges and every bit of the state compared every 500 steps.

usage: python3 profiles/soak_default_scene.py [steps=5000] [ranks=3] [precision=f64]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
precision = sys.argv[3] if len(sys.argv) > 3 else "f64"
b = physics.makebodies(1000, seed=1)
n = len(b.mass)
rad = np.full(n, 2.0); rad[:2] = 1.0
kw = dict(mode="tree", precision=precision, theta=1.0, dt=3600.0, collisions=True)
one = Engine(exact_sums=True, **kw)
dist = Engine(devices=[0] * ranks, distributed=True, **kw)
for e in (one, dist):
    e.upload(b.pos, b.vel, b.mass)
    e.set_radius(rad)
t0 = time.perf_counter()
done = 0
while done < steps:
    one.step(500); dist.step(500)
    done += 500
    a, c = one.download(), dist.download()
    sa, sc = one.stats(), dist.stats()
    same = all(np.array_equal(x, y) for x, y in zip(a, c)) and np.array_equal(one.get_radius(), dist.get_radius())
    print(f"step {done:6d}: bodies {sa.n_alive} / {sc.n_alive}, merges {sa.merges} / {sc.merges}, same bits: {same}, "
          f"{time.perf_counter() - t0:.1f} s", flush=True)
    if not same or sa.n_alive != sc.n_alive or sa.merges != sc.merges:
        sys.exit(1)
one.close(); dist.close()
