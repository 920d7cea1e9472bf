#!/usr/bin/env python3
"""Times the tree walk's kernel variants against each other (NBODY_WALK_BODIES_PER_LANE = 1 / 2) and checks that they
agree: fp32 bit for bit, mixed precision too.
usage: python3 profiles/walk_variants.py [plummer_f32_sizes_in_2^20 ...]   (then the two-galaxy scene in mixed precision)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbody_amd import Engine, physics


def run(b, knob, **kw):
    os.environ["NBODY_WALK_BODIES_PER_LANE"] = str(knob)
    with Engine(mode="tree", theta=0.5, device=0, **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(2)
        st = e.step_timed()
        e.step(1)
        p, v, _, _ = e.download()
    return st.ms_force, st.ms_sort + st.ms_build, p, v


sizes = [float(a) for a in sys.argv[1:]] or [1, 16]
for mi in sizes:
    n = int(mi * (1 << 20))
    b = physics.plummer(n, seed=1)
    f1, o1, p1, v1 = run(b, 1, precision="f32")
    f2, o2, p2, v2 = run(b, 2, precision="f32")
    print(f"f32 plummer n={n}: walk one/two bodies per lane {f1:.3f} / {f2:.3f} ms (sort+build {o1:.3f}); same bits: "
          f"{np.array_equal(p1, p2) and np.array_equal(v1, v2)}", flush=True)
for half in (1 << 19, 1 << 23):
    g = physics.galaxy_pair(half, seed=1)
    f1, o1, p1, v1 = run(g, 1, precision="mixed", eps=1.0)
    f2, o2, p2, v2 = run(g, 2, precision="mixed", eps=1.0)
    dv = np.abs(v1 - v2).max() / np.abs(v1).max()
    dp = np.abs(p1 - p2).max() / np.abs(p1).max()
    print(f"mixed galaxy pair 2x{half}: walk one/two bodies per lane {f1:.3f} / {f2:.3f} ms (sort+build {o1:.3f}); "
          f"max |dv|/max|v| {dv:.2e}, max |dx|/max|x| {dp:.2e} after 4 steps", flush=True)
