"""Worker of the world_size-2 rehearsals (started by tests/test_sharded_gloo.py and the GPU variant).

argv: out_prefix mode n nsteps engine_kind
engine_kind "oracle": a CPU stand-in shard whose force pass is the CPU oracle, so that the sharding and
gather logic of nbody_amd.sharded can be exercised without a GPU.  "gpu": real engines (all ranks share
cuda:0; gloo stages the gathers through the host).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nbody_amd import physics  # noqa: E402
from nbody_amd.sharded import ShardedStepper  # noqa: E402


class OracleShard:
    """Same methods as nbody_amd.sharded.EngineShard, state in CPU tensors, forces from the CPU oracle."""

    def __init__(self, bodies, mode, theta, dt, collisions=False):
        from oracle import oracle as O
        # collisions: the calling sequence of an engine with collisions on (advance leaves accelerations, the ranks
        # exchange those, commit updates every body); the stand-in's bodies never touch, so no merge happens
        self.collisions = collisions
        self.acc = torch.zeros(len(bodies), 4, dtype=torch.float64)
        self.O = O
        self.n = len(bodies)
        self.mode, self.theta, self.dt = mode, theta, dt
        self.cur = [torch.from_numpy(np.concatenate([bodies.pos, bodies.mass[:, None]], 1).copy()),
                    torch.from_numpy(np.concatenate([bodies.vel, np.zeros((self.n, 1))], 1).copy())]
        self.nxt = [torch.zeros_like(self.cur[0]), torch.zeros_like(self.cur[1])]
        self.first, self.count = 0, self.n

    def set_shard(self, first, count):
        self.first, self.count = first, count

    def prepare(self):
        pass

    def advance(self):
        pos = self.cur[0][:, :3].numpy(); mass = self.cur[0][:, 3].numpy(); vel = self.cur[1][:, :3].numpy()
        if self.mode == "tree":
            f, _, _ = self.O.forces(mass, pos, mode="tree", com_mode="true", theta=self.theta, drop_tail=False)
        else:
            f = self.O.direct_rows(mass, pos, nthreads=2)
        s = slice(self.first, self.first + self.count)
        if self.collisions:
            self.acc[s, :3] = torch.from_numpy(f[s] / mass[s, None])
            return
        v = vel[s] + f[s] / mass[s, None] * self.dt
        p = pos[s] + v * self.dt
        self.nxt[0][s, :3] = torch.from_numpy(p); self.nxt[0][s, 3] = self.cur[0][s, 3]
        self.nxt[1][s, :3] = torch.from_numpy(v)

    def advance_range(self, first, count):
        keep = (self.first, self.count)
        self.first, self.count = first, count
        self.advance()
        self.first, self.count = keep

    def next_state(self):
        return self.nxt

    def accelerations(self):
        return self.acc

    def commit(self):
        if self.collisions:
            v = self.cur[1][:, :3] + self.acc[:, :3] * self.dt
            self.nxt[1][:, :3] = v
            self.nxt[0][:, :3] = self.cur[0][:, :3] + v * self.dt
            self.nxt[0][:, 3] = self.cur[0][:, 3]
        self.cur, self.nxt = self.nxt, self.cur


def main():
    out_prefix, mode, n, nsteps, kind = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if kind == "comm":
        # the host-side collectives of the distributed tree, on the CPU: ragged arrays, a message longer than the first capacity
        from nbody_amd.disttree import TorchComm
        comm = TorchComm()
        rng = np.random.default_rng(100 + rank)
        a = rng.integers(0, 1 << 60, 5 + 3 * rank, dtype=np.int64)
        b2 = rng.normal(size=(rank + 1, 6))
        big = rng.integers(0, 255, 70000 * (rank + 1), dtype=np.uint8)       # beyond the 64 KiB the exchange starts with
        first = comm.allgather_many([[a, b2]])[0]
        second = comm.allgather_many([[a, b2, big]])[0]
        third = comm.allgather([b2])[0]
        out = {}
        for r in range(world):
            out[f"a{r}"], out[f"b{r}"] = first[r]
            out[f"c{r}"], out[f"d{r}"], out[f"big{r}"] = second[r]
            out[f"e{r}"] = third[r]
        np.savez(f"{out_prefix}.rank{rank}.npz", cap=comm._cap, **out)
        dist.barrier()
        dist.destroy_process_group()
        return
    bodies = physics.makebodies(n - 2, seed=11)
    if kind == "dist":
        # key-range-distributed tree: this rank starts with its slice of the upload order, all ranks share cuda:0
        from nbody_amd.disttree import DistRank, DistributedTree, TorchComm
        prec = sys.argv[6]
        per = (n + world - 1) // world
        a, z = min(n, rank * per), min(n, (rank + 1) * per)
        extra = {}
        rad = None
        if "collide" in sys.argv[7:]:
            # a crowd that merges: the same generator calls in every process, so every rank sees the same scene
            rng = np.random.default_rng(5)
            bodies = physics.Bodies(rng.uniform(-150.0, 150.0, (n, 3)), rng.normal(0, 1e-3, (n, 3)), np.abs(rng.normal(1e3, 500, n)) + 1)
            rad = np.random.default_rng(6).uniform(0.5, 3.0, n)
            extra = dict(collisions=True, dt=50.0)
        if "line" in sys.argv[7:]:
            # bodies on a line with random velocities: after one step most of them belong to another rank (the ranks must
            # agree to take their leavers back and choose new splitters)
            rng = np.random.default_rng(77)
            pos = np.zeros((n, 3)); pos[:, 0] = rng.uniform(-3e4, 3e4, n)
            bodies = physics.Bodies(pos, rng.normal(0, 1e-2, (n, 3)), np.abs(rng.normal(1e4, 5e3, n)) + 1.0)
            extra = dict(dt=200.0)
        r = DistRank(rank, world, precision=prec, theta=0.5, **extra)
        r.upload(int(per * 2.5) + 1024, bodies.pos[a:z], bodies.vel[a:z], bodies.mass[a:z], a)
        if rad is not None:
            r.set_radius(rad[a:z], 8.0)
        r.use_torch_stream()
        dt = DistributedTree([r], TorchComm())
        dt.distribute()
        dt.step(nsteps)
        idx, pos, vel, mass = dt.download()
        np.savez(f"{out_prefix}.rank{rank}.npz", pos=pos, vel=vel, mass=mass, ids=idx, rebalances=dt.rebalances)
        r.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if kind == "oracle":
        shard = OracleShard(bodies, mode, 0.5, 3600.0, collisions="collide" in sys.argv[6:])
        # the CPU stand-in has no wavefronts to align the ranges with; "split" as a seventh argument rehearses the
        # two-ranges-per-rank exchange
        stepper = ShardedStepper(shard, align=1, split="split" in sys.argv[6:])
        stepper.step(nsteps)
        pos = shard.cur[0][:, :3].numpy(); vel = shard.cur[1][:, :3].numpy()
        more = {}
    else:
        from nbody_amd import Engine
        from nbody_amd.sharded import EngineShard
        prec = sys.argv[6]
        extra = {}
        if "collide" in sys.argv[7:]:
            # a crowd that merges (the scene of tests/test_gpu_collide.py)
            rng = np.random.default_rng(5)
            bodies = physics.Bodies(rng.uniform(-60.0, 60.0, (n, 3)), rng.normal(0, 1e-3, (n, 3)), np.abs(rng.normal(1e3, 500, n)) + 1)
            extra = dict(collisions=True, dt=100.0)
        eng = Engine(mode=mode, precision=prec, theta=0.5, **extra)
        eng.upload(bodies.pos, bodies.vel, bodies.mass)
        if extra:
            eng.set_radius(np.random.default_rng(6).uniform(0.5, 3.0, n))
        stepper = ShardedStepper(EngineShard(eng), split="split" in sys.argv[7:])
        stepper.step(nsteps)
        pos, vel, mass, ids = eng.download()
        more = dict(mass=mass, ids=ids)
        if extra:
            more["radius"] = eng.get_radius()
            more["merges"] = eng.stats().merges
    np.savez(f"{out_prefix}.rank{rank}.npz", pos=pos, vel=vel, first=stepper.first, count=stepper.count, **more)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
