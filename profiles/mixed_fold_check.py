"""Mixed precision after the change of where fp32 partial sums join the fp64 totals (by node, not by wave visit):
bits of the two walk kernels, of a distributed run against one engine, accuracy against the fp64 direct sum, and time.
Run on the GPU box: python3 profiles/mixed_fold_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(b, steps, knob=None, **kw):
    if knob is None:
        os.environ.pop("NBODY_WALK_BODIES_PER_LANE", None)
    else:
        os.environ["NBODY_WALK_BODIES_PER_LANE"] = knob
    with Engine(mode="tree", precision="mixed", **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(steps)
        p, v, m, ids = e.download()
        acc, alive = e.accel()
    return p, v, ids, acc


for n in (129, 8193, 300000, 1 << 20):
    b = physics.plummer(n, seed=n % 5, a=2e3)
    r1 = run(b, 3, "1", theta=0.6, eps=2.0)
    r2 = run(b, 3, "2", theta=0.6, eps=2.0)
    print(n, "pair == single:", all(np.array_equal(a, c) for a, c in zip(r1, r2)),
          "max dv", np.abs(r1[1] - r2[1]).max(), flush=True)

# distributed (four GPUs' worth on one) against one engine with exact sums
b = physics.galaxy_pair(60000, seed=5)
with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, exact_sums=True) as e:
    e.upload(b.pos, b.vel, b.mass)
    e.step(4)
    want = e.download()
with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, distributed=True, devices=[0, 0, 0, 0]) as e:
    e.upload(b.pos, b.vel, b.mass)
    e.step(4)
    got = e.download()
o1, o2 = np.argsort(want[3]), np.argsort(got[3])
print("distributed == one engine:", np.array_equal(want[0][o1], got[0][o2]), np.array_equal(want[1][o1], got[1][o2]),
      "max dv", np.abs(want[1][o1] - got[1][o2]).max(), flush=True)

# accuracy against the fp64 direct sum on fp32-rounded inputs (sample of 256 bodies)
n = 200000
b = physics.plummer(n, seed=3)
pos = b.pos.astype(np.float32).astype(np.float64)
mass = b.mass.astype(np.float32).astype(np.float64)
with Engine(mode="tree", precision="mixed", theta=0.5) as e:
    e.upload(pos, b.vel, mass)
    acc, alive = e.accel()
with Engine(mode="tree", precision="f64", theta=0.5) as e:
    e.upload(pos, b.vel, mass)
    acc64, _ = e.accel()
err = np.linalg.norm(acc - acc64, axis=1) / np.linalg.norm(acc64, axis=1)
print("mixed tree vs fp64 tree: median %.3g p99.9 %.3g max %.3g" % (np.median(err), np.quantile(err, 0.999), err.max()), flush=True)

# time
for n, scene in ((1 << 20, "plummer"), (1 << 24, "galaxies")):
    b = physics.plummer(n, seed=1) if scene == "plummer" else physics.galaxy_pair(n, seed=1)
    for knob in (("1", "2") if n == 1 << 20 else (None,)):
        if knob is None:
            os.environ.pop("NBODY_WALK_BODIES_PER_LANE", None)
        else:
            os.environ["NBODY_WALK_BODIES_PER_LANE"] = knob
        with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0 if scene == "galaxies" else 0.0) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(3)
            e.sync()
            t = time.perf_counter()
            e.step(10)
            e.sync()
            ms = (time.perf_counter() - t) * 100
            print(n, scene, "bodies per lane", knob, "step %.3f ms" % ms, "force kernel %.3f ms" % e.time_force_kernel(5), flush=True)
