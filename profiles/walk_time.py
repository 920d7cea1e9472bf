"""Force-kernel time of the tree walk for whichever build NBODY_AMD_LIB names (to compare builds on one box): fp32 Plummer
spheres of 1M, 2M, 4M and 16M bodies, fp64 and mixed precision at 1M.  python3 profiles/walk_time.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402

print("library:", os.environ.get("NBODY_AMD_LIB", "in-tree"), flush=True)
for m, precision in ((1, "f32"), (2, "f32"), (4, "f32"), (16, "f32"), (1, "f64"), (1, "mixed")):
    n = m << 20
    b = physics.plummer(n, seed=1)
    with Engine(mode="tree", precision=precision, theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(3)
        e.sync()
        fk = [e.time_force_kernel(1) for _ in range(7)]
        print(f"{n} {precision}: force kernel {np.median(fk):.3f} ms (min {min(fk):.3f})", flush=True)
