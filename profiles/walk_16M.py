#!/usr/bin/env python3
"""The two 16M-body Barnes-Hut workloads of the benchmark configurations, five steps each, for a kernel trace:
the fp32 Plummer sphere (configuration 4) and the two 8M-body galaxies in mixed precision with eps = 1 m
(configuration 5).  Both walk two bodies per lane (walk_pair_kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics

for name, b, kw in (("plummer_16M_f32", physics.plummer(1 << 24, seed=1), dict(precision="f32")),
                    ("galaxy_pair_2x8M_mixed_eps1", physics.galaxy_pair(1 << 23, seed=1), dict(precision="mixed", eps=1.0))):
    with Engine(mode="tree", theta=0.5, device=0, **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(5)
        st = e.step_timed()
        print(f"{name}: sort {st.ms_sort:.3f} build {st.ms_build:.3f} walk {st.ms_force:.3f} ms", flush=True)
