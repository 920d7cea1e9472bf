"""Small launches of the tree walk with 64 >> s bodies per wavefront (NBODY_WALK_SPARSE=s): force kernel and step time,
the same bits for every s.  python3 profiles/walk_sparse_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402

scenes = [("default scene", physics.makebodies(1000, seed=1), 1.0)] + \
         [(f"plummer {n}", physics.plummer(n, seed=1, a=2e3), 0.5) for n in (4096, 16384, 65536, 262144)]
for precision in ("f32", "f64"):
    for name, b, theta in scenes:
        ref = None
        row = []
        for s in range(7):
            os.environ["NBODY_WALK_SPARSE"] = str(s)
            with Engine(mode="tree", precision=precision, theta=theta) as e:
                e.upload(b.pos, b.vel, b.mass)
                e.step(5)
                e.sync()
                t = time.perf_counter()
                e.step(100)
                e.sync()
                ms = (time.perf_counter() - t) * 10
                fk = np.median([e.time_force_kernel(1) for _ in range(5)])
                state = e.download()
            same = True if ref is None else all(np.array_equal(x, y) for x, y in zip(ref, state))
            ref = ref or state
            row.append(f"s={s}: {fk * 1e3:.1f}/{ms * 1e3:.1f}{'' if same else ' BITS DIFFER'}")
        print(f"{precision} {name}: force kernel / step in us: " + "  ".join(row), flush=True)
