"""The engine's limit, 2^26 = 67,108,864 bodies (Plummer sphere, fp32, theta 0.5): three steps on one engine and on the
key-range-distributed tree with eight ranks behind one handle (all on this GPU); times per step and whether the two end
with the same bits.

usage: python3 profiles/limit_2p26.py [log2 bodies=26] [ranks=8] [steps=3]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
b = physics.plummer(1 << log2n, seed=1)
print(f"{len(b.mass)} bodies made", flush=True)
out = []
for name, kw in (("one engine", dict(exact_sums=True)), (f"distributed tree, {ranks} ranks", dict(devices=[0] * ranks, distributed=True))):
    with Engine(mode="tree", precision="f32", theta=0.5, **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(1); e.sync()
        t0 = time.perf_counter()
        e.step(steps); e.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        st = e.stats()
        p, v, m, ids = e.download()
        print(f"{name}: {ms:.1f} ms per step, {st.n_alive} bodies alive after {st.steps} steps", flush=True)
        out.append((p, v, ids))
print("same bits:", all(np.array_equal(x, y) for x, y in zip(out[0], out[1])), flush=True)
