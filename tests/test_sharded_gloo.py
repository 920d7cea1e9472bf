"""world_size-2 rehearsal of the multi-GPU stepper on the CPU (gloo): target ranges, the in-place and the
padded all-gather, and that two sharded ranks reproduce the single-process result exactly.  The force pass
of the stand-in shard is the CPU oracle; the real engines are rehearsed the same way in the GPU tests."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from nbody_amd import physics
from nbody_amd.sharded import shard_range
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(tmp_path, world, args, timeout=300):
    port = free_port()
    prefix = str(tmp_path / "out")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), prefix] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r]}"
    return [np.load(f"{prefix}.rank{r}.npz") for r in range(world)]


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 8, 1002, 65536):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c, per = shard_range(n, r, world, align=1)
                assert per == (n + world - 1) // world and 0 <= c <= per
                seen += list(range(f, f + c))
            assert seen == list(range(n))
            seen = []
            for r in range(world):                      # default: ranges start at multiples of 256 bodies
                f, c, per = shard_range(n, r, world)
                assert per % 256 == 0 and f % 256 == 0 or c == 0
                seen += list(range(f, f + c))
            assert seen == list(range(n))


@pytest.mark.parametrize("mode,n,split", [("direct", 64, False), ("direct", 63, False), ("tree", 64, False), ("tree", 61, False),
                                         ("direct", 64, True), ("tree", 64, True), ("tree", 61, "collide")])
def test_two_ranks_reproduce_single_process(tmp_path, mode, n, split):
    """split: every rank owns a slice of each half of the body order and exchanges the halves one after the other (the
    arrangement that lets the first exchange run beside the second force pass on RCCL).  "collide": the calling
    sequence of an engine with collisions on, where the ranks exchange accelerations and every rank updates all bodies."""
    res = run_ranks(tmp_path, 2, [mode, str(n), "3", "oracle", "-"] + ([split if isinstance(split, str) else "split"] if split else []))
    split = split is True
    assert np.array_equal(res[0]["pos"], res[1]["pos"]) and np.array_equal(res[0]["vel"], res[1]["vel"])
    if split:
        assert int(res[0]["first"]) == 0 and int(res[0]["count"]) == n // 4 and int(res[1]["first"]) == n // 4
    else:
        assert int(res[0]["first"]) == 0 and int(res[1]["first"]) == int(res[0]["count"])
    b = physics.makebodies(n - 2, seed=11)
    pos, vel = b.pos.copy(), b.vel.copy()
    for _ in range(3):
        if mode == "tree":
            f, _, _ = O.forces(b.mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False)
        else:
            f = O.direct_rows(b.mass, pos, nthreads=2)
        vel = vel + f / b.mass[:, None] * 3600.0
        pos = pos + vel * 3600.0
    assert np.array_equal(res[0]["pos"], pos) and np.array_equal(res[0]["vel"], vel)


def test_ragged_allgather_of_the_distributed_tree(tmp_path):
    """TorchComm.allgather_many over gloo, three ranks: arrays of different lengths per rank come back whole on every rank,
    in one collective while they fit the agreed capacity, and the capacity grows when a message does not."""
    world = 3
    res = run_ranks(tmp_path, world, ["-", "0", "0", "comm"])
    for r in range(world):
        rng = np.random.default_rng(100 + r)
        a = rng.integers(0, 1 << 60, 5 + 3 * r, dtype=np.int64)
        b = rng.normal(size=(r + 1, 6))
        big = rng.integers(0, 255, 70000 * (r + 1), dtype=np.uint8)
        for got in res:
            assert np.array_equal(got[f"a{r}"], a) and np.array_equal(got[f"c{r}"], a)
            assert np.array_equal(got[f"b{r}"], b) and np.array_equal(got[f"d{r}"], b) and np.array_equal(got[f"e{r}"], b)
            assert np.array_equal(got[f"big{r}"], big)
    assert all(int(got["cap"]) >= 210000 for got in res)
