"""Key-range-distributed tree, eight ranks side by side on one GPU: the timed step (no drains between phases) beside the
phase table, for either walk kernel.

usage: python3 profiles/dist_timed.py [scene=sphere|galaxies] [world=8] [steps=10] [log2 of the bodies=24]
(with few bodies the step is host work and launch latencies only: what a rank spends outside its kernels)
NBODY_WALK_BODIES_PER_LANE=1|2 picks the walk kernel of every rank (unset: the engine's own choice)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                                                # noqa: E402
from nbody_amd import physics                                               # noqa: E402
from nbody_amd.disttree import DistRank, DistributedTree, LocalComm        # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sphere"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
log2n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
if scene == "sphere":
    b = physics.plummer(1 << log2n, seed=1)
    kw = dict(precision="f32", theta=0.5)
else:
    b = physics.galaxy_pair(1 << (log2n - 1), seed=1)
    kw = dict(precision="mixed", theta=0.5, eps=1.0)
n = len(b.mass)
per = (n + world - 1) // world
ranks = [DistRank(r, world, **kw) for r in range(world)]
for r in ranks:
    a, z = r.rank * per, min(n, (r.rank + 1) * per)
    r.upload(int(per * 1.25) + 4096, b.pos[a:z], b.vel[a:z], b.mass[a:z], a)
    r.use_torch_stream()
tree = DistributedTree(ranks, LocalComm(world), freeze_gc=os.environ.get("NBODY_DIST_GC") != "plain")
tree.distribute()
tree.step(2)                    # the first step moves every body to its owner
each = []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree.step(1)
    torch.cuda.synchronize()
    each.append((time.perf_counter() - t0) * 1e3)
print("steps one by one, ms:", [round(t, 2) for t in each], flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
tree.step(steps)
torch.cuda.synchronize()
timed = (time.perf_counter() - t0) * 1e3 / steps
runs = [tree.step_profiled() for _ in range(5)]
keys = sorted({k for t in runs for k in t})      # (a run in which new splitters were chosen has a phase of its own)
acc = {k: float(np.median([t.get(k, 0.0) for t in runs])) / world for k in keys}
print(f"{scene} 2^{log2n} bodies world {world} lanes {os.environ.get('NBODY_WALK_BODIES_PER_LANE', 'auto')}: timed step per rank "
      f"{timed / world:.3f} ms (all ranks {timed:.2f}); phase sums per rank {sum(acc.values()):.3f}:",
      {k: round(v, 3) for k, v in acc.items()}, flush=True)
for r in ranks:
    r.close()
