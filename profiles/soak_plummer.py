"""A Plummer sphere of 2^17 bodies inside the reference's bound for thousands of steps on the distributed tree behind one
handle (five ranks on this GPU) and on one engine, fp32 (or the precision named): every bit of the state compared every
250 steps.

usage: python3 profiles/soak_plummer.py [steps=2000] [ranks=5] [log2 bodies=17] [precision=f32]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 5
log2n = int(sys.argv[3]) if len(sys.argv) > 3 else 17
precision = sys.argv[4] if len(sys.argv) > 4 else "f32"
b = physics.plummer(1 << log2n, seed=3)
kw = dict(mode="tree", precision=precision, theta=0.6, dt=3600.0)
one = Engine(exact_sums=True, **kw)
dist = Engine(devices=[0] * ranks, distributed=True, **kw)
for e in (one, dist):
    e.upload(b.pos, b.vel, b.mass)
t0 = time.perf_counter()
done = 0
while done < steps:
    one.step(250); dist.step(250)
    done += 250
    a, c = one.download(), dist.download()
    same = all(np.array_equal(x, y) for x, y in zip(a, c))
    print(f"step {done:6d}: bodies {len(a[3])} / {len(c[3])}, same bits: {same}, {time.perf_counter() - t0:.1f} s", flush=True)
    if not same:
        sys.exit(1)
one.close(); dist.close()
