"""Sort and build times of the three ways to sum the cells (prefix differences in fp64, child by child, exact 128-bit
prefix differences) at 1M, 2M and 16M bodies, fp32 Plummer sphere, theta 0.5: medians of seven timed steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

for n in (1 << 20, 1 << 21, 1 << 24):
    b = physics.plummer(n, seed=1)
    for name, kw in (("prefix", {}), ("tree", dict(tree_sums=True)), ("exact", dict(exact_sums=True))):
        for prec in ("f32",) if n > (1 << 20) else ("f32", "f64"):
            if prec == "f64" and name == "prefix":
                continue
            with Engine(mode="tree", precision=prec, theta=0.5, **kw) as e:
                e.upload(b.pos, b.vel, b.mass)
                e.step(2)
                ts = [e.step_timed() for _ in range(7)]
                print(f"n {n} {prec} sums {name}: sort {np.median([t.ms_sort for t in ts]):.3f} build {np.median([t.ms_build for t in ts]):.3f} "
                      f"force {np.median([t.ms_force for t in ts]):.3f} ms", flush=True)
