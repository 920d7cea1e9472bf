"""Phases of a Barnes-Hut step (sort = keys, eight radix passes and the reordering of the state; build; force), Plummer
sphere, fp32, median of nine timed steps per size.

usage: python3 profiles/sort_phase.py [log2 of the bodies ...]   (default 20 21 22 24)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 24]
for log2n in sizes:
    b = physics.plummer(1 << log2n, seed=1)
    with Engine(mode="tree", precision="f32", theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(3)
        runs = [e.step_timed() for _ in range(9)]
        print(f"2^{log2n} bodies: "
              f"sort {np.median([r.ms_sort for r in runs]):.3f} ms, build {np.median([r.ms_build for r in runs]):.3f} ms, "
              f"force {np.median([r.ms_force for r in runs]):.3f} ms", flush=True)
