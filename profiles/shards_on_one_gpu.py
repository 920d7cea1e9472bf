"""Eight shards of the 16M-body sphere behind one handle on one GPU (every shard holds all bodies, sorts and builds, and
walks its eighth): time per step, for the walk's XCD dealing given by NBODY_WALK_XCD_CHUNK_LOG2 (unset: the engine's choice)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics

b = physics.plummer(1 << 24, seed=1)
with Engine(mode="tree", precision="f32", theta=0.5, devices=[0] * 8) as e:
    e.upload(b.pos, b.vel, b.mass)
    e.step(2); e.sync()
    t0 = time.perf_counter()
    e.step(5); e.sync()
    print(f"xcd chunk log2 {os.environ.get('NBODY_WALK_XCD_CHUNK_LOG2', 'auto')}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step of eight shards", flush=True)
