"""One rank of the key-range-distributed tree over RCCL (world size 1) on a 16M-body scene, step by step: wall time of
every step with and without a drain after it, then the phase table.  Shows where a timed step spends what the phase sums
do not account for.

usage: python3 profiles/dist_one_rank.py [scene=galaxies|sphere]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch                                                                # noqa: E402
import torch.distributed as dist                                            # noqa: E402
from nbody_amd import physics                                               # noqa: E402
from nbody_amd.disttree import DistRank, DistributedTree, TorchComm        # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "galaxies"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
if scene == "sphere":
    b = physics.plummer(1 << 24, seed=1)
    kw = dict(precision="f32", theta=0.5)
else:
    b = physics.galaxy_pair(1 << 23, seed=1)
    kw = dict(precision="mixed", theta=0.5, eps=1.0)
n = len(b.mass)
dr = DistRank(0, 1, device=0, **kw)
dr.upload(int(n * 1.25) + 4096, b.pos, b.vel, b.mass, 0)
dr.use_torch_stream()
del b
tree = DistributedTree([dr], TorchComm())
tree.distribute()
each = []
for k in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree.step(1)
    torch.cuda.synchronize()
    each.append((time.perf_counter() - t0) * 1e3)
print("steps drained one by one, ms:", [round(t, 2) for t in each], "rebalances", tree.rebalances, flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
tree.step(5)
torch.cuda.synchronize()
print("five steps back to back, ms per step:", round((time.perf_counter() - t0) * 1e3 / 5, 2), "rebalances", tree.rebalances, flush=True)
runs = [tree.step_profiled() for _ in range(5)]
for t in runs:
    print("phases:", {k: round(v, 2) for k, v in t.items()}, "sum", round(sum(t.values()), 2), flush=True)
dr.close()
dist.destroy_process_group()
