"""Steps the key-range-distributed tree with eight ranks side by side on one GPU (16,777,216-body Plummer sphere,
theta 0.5, fp32): the command profiles/collect.sh traces for profiles/<tag>_dist_kernel_stats.txt.  Prints the phase
times of two drained steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nbody_amd import physics                                               # noqa: E402
from nbody_amd.disttree import DistRank, DistributedTree, LocalComm        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = physics.plummer(n, seed=1)
per = (n + world - 1) // world
ranks = [DistRank(r, world, precision="f32", theta=0.5) for r in range(world)]
for r in ranks:
    a, z = r.rank * per, min(n, (r.rank + 1) * per)
    r.upload(int(per * 1.25) + 4096, b.pos[a:z], b.vel[a:z], b.mass[a:z], a)
    r.use_torch_stream()
tree = DistributedTree(ranks, LocalComm(world))
tree.distribute()
tree.step(2)                    # the first step moves every body to its owner
for _ in range(2):
    t = tree.step_profiled()
    print("phase ms per rank:", {k: round(v / world, 3) for k, v in t.items()}, "step", round(sum(t.values()) / world, 3))
tree.step(4)
for r in ranks:
    r.close()
