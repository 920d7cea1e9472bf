"""Benchmark configuration 5 at full size and length: two 8,388,608-body galaxies (mixed precision: fp32 pair terms, fp64
sums and state, softening 1 m, theta 0.5), 1000 steps of 3600 s, on the key-range-distributed tree with eight ranks
(all on this GPU, driven by the engine itself) and on one engine; energy and momentum every 100 steps.

usage: python3 profiles/config5_1000_steps.py [steps=1000] [ranks=8]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = physics.galaxy_pair(1 << 23, seed=1)
n = len(b)
pscale = float((b.mass * np.linalg.norm(b.vel, axis=1)).sum())
for name, kw in ((f"distributed tree, {ranks} ranks", dict(devices=[0] * ranks, distributed=True)), ("one engine", {})):
    with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        ke0, pe0, p0 = e.energy()
        e0 = ke0 + pe0
        print(f"{name}: {n} bodies, E0 = {e0:.6e} (K {ke0:.6e}, U {pe0:.6e})", flush=True)
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            k = min(100, steps - done)
            e.step(k)
            done += k
            ke, pe, p = e.energy()
            st = e.stats()
            print(f"  step {done:5d}: |dE/E0| = {abs((ke + pe - e0) / e0):.3e}   |dP| / sum m|v| = {np.linalg.norm(p - p0) / pscale:.3e}   "
                  f"bodies {st.n_alive}   wall {time.perf_counter() - t0:6.1f} s", flush=True)
