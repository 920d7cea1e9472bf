"""More bodies than one engine takes (2^26): a Plummer sphere of 2^28 = 268,435,456 bodies, fp32, theta 0.5, on the
key-range-distributed tree with eight ranks behind one handle (all on this GPU; every rank holds an eighth).  Steps,
survivors, momentum and energy drift; nothing to compare with but the invariants.

usage: python3 profiles/beyond_limit.py [log2 bodies=28] [ranks=8] [steps=3]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics                                       # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << log2n
chunk = 1 << 24
pos = np.empty((n, 3)); vel = np.empty((n, 3)); mass = np.empty(n)
for k in range(n // chunk):                                 # (a sum of Plummer spheres of one scale is a Plummer sphere)
    b = physics.plummer(chunk, seed=100 + k)
    pos[k * chunk:(k + 1) * chunk] = b.pos; vel[k * chunk:(k + 1) * chunk] = b.vel
    mass[k * chunk:(k + 1) * chunk] = b.mass / (n // chunk)
    print(f"made {(k + 1) * chunk} bodies", flush=True)
# each chunk's velocities belong to a sphere of its own (total) mass; with 1/k of the mass each, the speeds of the whole are the same
with Engine(mode="tree", precision="f32", theta=0.5, devices=[0] * ranks, distributed=True) as e:
    t0 = time.perf_counter()
    e.upload(pos, vel, mass)
    print(f"uploaded in {time.perf_counter() - t0:.1f} s", flush=True)
    ke0, pe0, p0 = e.energy()
    print(f"E0 = {ke0 + pe0:.6e} (K {ke0:.6e}, U {pe0:.6e})", flush=True)
    e.step(1); e.sync()
    t0 = time.perf_counter()
    e.step(steps); e.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    st = e.stats()
    ke, pe, p = e.energy()
    pscale = float((mass * np.linalg.norm(vel, axis=1)).sum())
    print(f"{n} bodies, {ranks} ranks: {ms:.1f} ms per step; {st.n_alive} alive after {st.steps} steps; "
          f"|dE/E0| = {abs((ke + pe - ke0 - pe0) / (ke0 + pe0)):.3e}, |dP| / sum m|v| = {np.linalg.norm(p - p0) / pscale:.3e}", flush=True)
