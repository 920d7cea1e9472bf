"""The key-range-distributed tree (nbe_dist_*, nbody_amd/disttree.py): every body on one rank only, local trees,
pruned fragments exchanged.  Checked against ONE engine stepping all bodies (whose own parity with the CPU restatement
is the business of tests/test_gpu_tree.py): every body must visit the nodes it visits in the one octree, in the same
order, so fp32 and fp64 runs agree bit for bit when the one engine also sums its cells child by child
(exact_sums=True), and so do mixed-precision runs: a body's fp32 partial sums join its fp64 totals at nodes picked by
their records, not by what its wavefront has visited."""
import numpy as np
import pytest

from nbody_amd import Engine, EngineError, physics
from nbody_amd.disttree import DistRank, DistributedTree, LocalComm, choose_splitters
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run_distributed(b, world, nsteps, capacity=None, per_step=None, rebalance_at=1.10, **kw):
    n = len(b)
    per = (n + world - 1) // world
    ranks = [DistRank(r, world, **kw) for r in range(world)]
    try:
        for r in ranks:
            a, z = min(n, r.rank * per), min(n, (r.rank + 1) * per)
            r.upload(capacity or int(per * 1.5) + 1024, b.pos[a:z], b.vel[a:z], b.mass[a:z], a)
            r.use_torch_stream()
        dt = DistributedTree(ranks, LocalComm(world), rebalance_at=rebalance_at)
        dt.distribute()
        for s in range(nsteps):
            dt.step(1)
            if per_step:
                per_step(s, dt)
        return dt.download()
    finally:
        for r in ranks:
            r.close()


def run_one(b, nsteps, **kw):
    with Engine(mode="tree", exact_sums=True, **kw) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(nsteps)
        p, v, m, ids = e.download()
    return ids, p, v, m


@pytest.mark.parametrize("how", ["tree_sums", "exact_sums"])
def test_cells_summed_child_by_child_equal_the_prefix_sums_to_rounding(gpu, how):
    """NBE_SUMS_TREE and NBE_SUMS_EXACT on fp32 node records: the same cells, masses and centres as the default build to
    fp32 rounding, and the centres of the pointer octree of the CPU restatement."""
    b = physics.plummer(30000, seed=3, a=2e3)
    tabs = []
    for kw in ({}, {how: True}):
        with Engine(mode="tree", precision="f32", theta=0.5, **kw) as e:
            e.upload(b.pos, b.vel, b.mass)
            tabs.append(e.tree_nodes())
    a, t = tabs
    assert a.shape == t.shape and np.array_equal(a[:, 4:], t[:, 4:])
    assert np.abs(a[:, 3] - t[:, 3]).max() <= 2e-7 * a[:, 3].max()
    assert np.abs(a[:, :3] - t[:, :3]).max() <= 2 ** -23 * 32768
    dump, _ = O.tree_dump(b.mass.astype(np.float32).astype(np.float64), b.pos.astype(np.float32).astype(np.float64),
                          com_mode="true", max_depth=21)
    keep = dump[(dump["kind"] == 0) | (dump["nchild"] >= 2)]
    assert len(keep) == len(t)
    assert np.abs(t[:, :3] - np.stack([keep["com"][:, k] for k in range(3)], 1)).max() <= 2 ** -23 * 32768


@pytest.mark.parametrize("n,world,precision,nsteps", [(5003, 2, "f32", 5), (5003, 3, "f32", 5), (20001, 4, "f64", 4),
                                                      (300, 5, "f32", 4), (100000, 8, "f32", 3), (1001, 3, "f64", 6),
                                                      (7, 3, "f32", 2), (2, 3, "f32", 3), (1, 2, "f64", 2)])
def test_ranks_on_one_gpu_equal_one_engine_bit_for_bit(gpu, n, world, precision, nsteps):
    """2 to 8 ranks (uneven n, ranks with a handful of bodies) against one engine: identical survivors, positions and
    velocities after several steps, in which bodies change owner."""
    b = physics.plummer(n, seed=8, a=2e3)
    moved = []
    got = run_distributed(b, world, nsteps, precision=precision, theta=0.6, per_step=lambda s, dt: moved.append(sum(dt.last_migrated)))
    want = run_one(b, nsteps, precision=precision, theta=0.6)
    assert len(got[0]) == n and np.array_equal(got[0], want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a, w)
    if n >= 5000:
        assert all(m > 0 for m in moved[1:])          # bodies did cross splitters after the first redistribution


def test_softened_galaxies_and_mixed_precision(gpu):
    """Two galaxies on approach with softening (the scene of the colliding-galaxies configuration, small): bit for bit
    in fp32 and in mixed precision, the precision of that configuration."""
    b = physics.galaxy_pair(6000, seed=5)
    for precision in ("f32", "mixed"):
        got = run_distributed(b, 4, 4, precision=precision, theta=0.5, eps=1.0)
        want = run_one(b, 4, precision=precision, theta=0.5, eps=1.0)
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])


@pytest.mark.parametrize("seed,n,world,precision", [(700, 400, 3, "f32"), (701, 3000, 5, "f64"), (702, 3000, 2, "f32"),
                                                    (703, 50, 4, "f64")])
def test_coincident_bodies_are_summed_in_one_order(gpu, seed, n, world, precision):
    """Bodies with the same 63-bit key leave the stable sort in an order that depends on the path they took (previous
    sorts on one engine, arrival on a rank); the reorder puts runs of up to 64 of them in body-index order, so a scene
    in which most bodies share their position with others is bit-equal too."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(-2e4, 2e4, (n // 4 + 1, 3))
    pos = base[rng.integers(len(base), size=n)]
    pos = pos + (rng.random((n, 1)) < 0.3) * rng.normal(0, 1e-3, (n, 3))
    b = physics.Bodies(pos, rng.normal(0, 1e-2, (n, 3)), np.abs(rng.normal(1e4, 5e3, n)) + 1)
    got = run_distributed(b, world, 4, precision=precision, theta=0.6, dt=200.0)
    want = run_one(b, 4, precision=precision, theta=0.6, dt=200.0)
    assert np.array_equal(got[0], want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a, w)


def test_bodies_that_leave_the_bound_are_dropped_on_their_rank(gpu):
    """tree.go:290-292 on a distributed run: a body that flies out is removed by its owner's next build, as one engine
    removes it; the others carry on identically."""
    b = physics.plummer(4000, seed=2, a=2e3)
    pos, vel = b.pos.copy(), b.vel.copy()
    pos[11] = (32760.0, 5.0, 5.0); vel[11] = (3.0, 0.0, 0.0)          # out through the +x face in the second step
    pos[12] = (-32767.5, -7.0, 1.0); vel[12] = (-1.0, 0.0, 0.0)       # out through the -x face in the first step
    bb = physics.Bodies(pos, vel, b.mass)
    got = run_distributed(bb, 3, 5, precision="f32", theta=0.6, dt=3.0)
    want = run_one(bb, 5, precision="f32", theta=0.6, dt=3.0)
    assert 11 not in want[0] and 12 not in want[0] and len(want[0]) == 3998
    assert np.array_equal(got[0], want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a, w)


def test_call_order_and_unsupported_configurations(gpu):
    b = physics.plummer(500, seed=1, a=2e3)
    for kw in (dict(per_wave=True), dict(ref_com=True)):
        r = DistRank(0, 2, **kw)
        with pytest.raises(EngineError, match="distributed tree needs"):
            r.upload(1000, b.pos, b.vel, b.mass, 0)
        r.close()
    r = DistRank(0, 2, precision="f32")
    try:
        with pytest.raises(EngineError, match="nbe_dist_upload"):
            r.classify()
        with pytest.raises(EngineError, match="capacity"):
            r.upload(100, b.pos, b.vel, b.mass, 0)
        r.upload(1000, b.pos, b.vel, b.mass, 0)
        with pytest.raises(EngineError, match="distributed engine"):
            r.engine.step(1)
        with pytest.raises(EngineError, match="before nbe_dist_assemble"):
            r.advance()
        with pytest.raises(EngineError, match="before nbe_dist_classify"):
            r.receive(0)
        with pytest.raises(EngineError, match="must not decrease"):
            r.set_splitters(np.array([0, (1 << 63) + 5, 1 << 63], dtype=np.uint64))
    finally:
        r.close()


def test_a_rank_that_runs_out_of_slots_says_so(gpu):
    b = physics.plummer(4000, seed=2, a=2e3)
    with pytest.raises(EngineError, match="slots are free|too small|do not hold the bodies"):
        # all bodies start on rank 0 and rank 1 has room for 1000
        ranks = [DistRank(r, 2, precision="f32") for r in range(2)]
        try:
            ranks[0].upload(5000, b.pos, b.vel, b.mass, 0)
            ranks[1].upload(1000, b.pos[:0], b.vel[:0], b.mass[:0], 4000)
            for r in ranks:
                r.use_torch_stream()
            dt = DistributedTree(ranks, LocalComm(2))
            dt.distribute()
            dt.step(1)
        finally:
            for r in ranks:
                r.close()


@pytest.mark.parametrize("world,precision", [(2, "f32"), (3, "f64"), (8, "f32")])
def test_one_process_per_rank_over_torch_distributed(gpu, tmp_path, world, precision):
    """The same stepper with one rank per process and the exchanges as torch.distributed collectives (gloo here, all
    ranks on cuda:0; "nccl" = RCCL takes the same path with device buffers): together the ranks hold what one engine
    reaches, bit for bit."""
    from tests.test_sharded_gloo import run_ranks
    n = 6001
    res = run_ranks(tmp_path, world, ["tree", str(n), "4", "dist", precision], timeout=240)
    b = physics.makebodies(n - 2, seed=11)
    want = run_one(b, 4, precision=precision, theta=0.5)
    ids = np.concatenate([r["ids"] for r in res])
    order = np.argsort(ids)
    assert np.array_equal(ids[order], want[0])
    assert all(len(r["ids"]) > 0 for r in res)
    for name, w in (("pos", want[1]), ("vel", want[2]), ("mass", want[3])):
        assert np.array_equal(np.concatenate([r[name] for r in res])[order], w), name


def test_one_process_per_rank_with_collisions(gpu, tmp_path):
    """Collisions on, one rank per process (gloo, all on cuda:0): the survivors and their states equal one engine's."""
    from tests.test_sharded_gloo import run_ranks
    n, nsteps = 6000, 12
    res = run_ranks(tmp_path, 3, ["tree", str(n), str(nsteps), "dist", "f64", "collide"], timeout=240)
    rng = np.random.default_rng(5)
    pos, vel, mass = rng.uniform(-150.0, 150.0, (n, 3)), rng.normal(0, 1e-3, (n, 3)), np.abs(rng.normal(1e3, 500, n)) + 1
    rad = np.random.default_rng(6).uniform(0.5, 3.0, n)
    with Engine(mode="tree", precision="f64", theta=0.5, dt=50.0, collisions=True, exact_sums=True) as e:
        e.upload(pos, vel, mass); e.set_radius(rad)
        e.step(nsteps)
        wp, wv, wm, wid = e.download()
        merges = e.stats().merges
    assert merges > 50
    idx = np.concatenate([r["ids"] for r in res]); order = np.argsort(idx)
    assert np.array_equal(idx[order], wid)
    for key, want in (("pos", wp), ("vel", wv), ("mass", wm)):
        assert np.array_equal(np.concatenate([r[key] for r in res])[order], want)


def test_a_step_that_sends_most_bodies_elsewhere_gets_new_splitters_first(gpu):
    """Bodies on a line (y = z = 0 exactly) with random velocities: all keys lie in octants 6 and 7 at first, and after one
    step the bodies fill all eight octants, so seven eighths of them are bound for rank 0 of two: more than its spare slots
    hold.  Every rank sees the counts and the room (nbe_dist_room), the leavers go back to their slots
    (nbe_dist_unclassify), new splitters are chosen and the step goes on: the same bits as one engine, from the Python
    host and from the engine's own driver."""
    rng = np.random.default_rng(77)
    n = 40000
    pos = np.zeros((n, 3)); pos[:, 0] = rng.uniform(-3e4, 3e4, n)
    vel = rng.normal(0, 1e-2, (n, 3)); mass = np.abs(rng.normal(1e4, 5e3, n)) + 1.0
    kw = dict(precision="f32", theta=0.6, dt=200.0)
    with Engine(mode="tree", exact_sums=True, **kw) as one:
        one.upload(pos, vel, mass)
        one.step(3)
        want = one.download()
    with Engine(mode="tree", devices=[0, 0], distributed=True, **kw) as e:
        e.upload(pos, vel, mass)
        e.step(3)
        got = e.download()
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
    world, per = 2, n // 2
    ranks = [DistRank(r, world, **kw) for r in range(world)]
    try:
        for r in ranks:
            a = r.rank * per
            r.upload(int(per * 1.25) + 4096, pos[a:a + per], vel[a:a + per], mass[a:a + per], a)
            r.use_torch_stream()
        dt = DistributedTree(ranks, LocalComm(world), rebalance_at=None)
        dt.distribute()
        dt.step(3)
        assert dt.rebalances == 1                          # (rebalance_at=None: only the overflow could have asked for it)
        idx, p, v, m = dt.download()
        # undone by hand: classify, take it back, classify again gives the same counts and loses nobody
        first = [r.classify() for r in ranks]
        for r in ranks:
            r.unclassify()
        again = [r.classify() for r in ranks]
        assert all(np.array_equal(a, b) for a, b in zip(first, again)) and sum(int(c.sum()) for c in first) > 0
        for r in ranks:
            r.unclassify()
        idx2, p2, v2, m2 = dt.download()
    finally:
        for r in ranks:
            r.close()
    assert np.array_equal(idx, want[3]) and np.array_equal(p, want[0]) and np.array_equal(v, want[1]) and np.array_equal(m, want[2])
    assert np.array_equal(idx2, idx) and np.array_equal(p2, p) and np.array_equal(v2, v) and np.array_equal(m2, m)


def test_overflowing_step_over_torch_distributed(gpu, tmp_path):
    """The scene of the test above with one rank per process (gloo, both on cuda:0): both processes read the same counts
    and rooms out of one gather and turn back together; same bits as one engine."""
    from tests.test_sharded_gloo import run_ranks
    n = 40000
    res = run_ranks(tmp_path, 2, ["tree", str(n), "3", "dist", "f32", "line"], timeout=240)
    rng = np.random.default_rng(77)
    pos = np.zeros((n, 3)); pos[:, 0] = rng.uniform(-3e4, 3e4, n)
    vel = rng.normal(0, 1e-2, (n, 3)); mass = np.abs(rng.normal(1e4, 5e3, n)) + 1.0
    with Engine(mode="tree", precision="f32", theta=0.5, dt=200.0, exact_sums=True) as e:
        e.upload(pos, vel, mass)
        e.step(3)
        wp, wv, wm, wid = e.download()
    assert all(int(r["rebalances"]) >= 1 for r in res)
    idx = np.concatenate([r["ids"] for r in res]); order = np.argsort(idx)
    assert np.array_equal(idx[order], wid)
    for key, want in (("pos", wp), ("vel", wv), ("mass", wm)):
        assert np.array_equal(np.concatenate([r[key] for r in res])[order], want)


def test_a_crowd_in_few_alignment_cells_needs_more_room_per_gpu(gpu, monkeypatch):
    """With collisions on, splitters sit on the borders of cells at least four radius caps wide (here 4096 m for radii up
    to 40 m).  20,000 bodies inside one such cell cannot be shared out over three shards:
    refused with a message that names the way out, NBODY_DIST_HEADROOM, with which the run equals one engine's."""
    rng = np.random.default_rng(21)
    n = 20000
    pos = 2048.0 + rng.uniform(-900.0, 900.0, (n, 3)); vel = rng.normal(0, 1e-2, (n, 3))     # inside the cell [0, 4096)^3
    mass = np.abs(rng.normal(1e4, 5e3, n)) + 1.0; rad = rng.uniform(0.5, 40.0, n)
    kw = dict(mode="tree", precision="f64", theta=0.6, dt=200.0, collisions=True)
    monkeypatch.delenv("NBODY_DIST_HEADROOM", raising=False)
    with Engine(devices=[0, 0, 0], distributed=True, **kw) as e:
        e.upload(pos, vel, mass); e.set_radius(rad)
        with pytest.raises(EngineError, match="NBODY_DIST_HEADROOM"):
            e.step(1)
    with Engine(exact_sums=True, **kw) as one:
        one.upload(pos, vel, mass); one.set_radius(rad)
        one.step(2)
        want, wst = one.download(), one.stats()
    monkeypatch.setenv("NBODY_DIST_HEADROOM", "3")
    with Engine(devices=[0, 0, 0], distributed=True, **kw) as e:
        e.upload(pos, vel, mass); e.set_radius(rad)
        e.step(2)
        got, gst = e.download(), e.stats()
    assert gst.merges == wst.merges > 0 and gst.n_alive == wst.n_alive
    for a, w in zip(got, want):
        assert np.array_equal(a, w)


def test_new_splitters_during_a_run_change_nothing_but_the_owners(gpu):
    """Splitters are chosen anew whenever a rank holds too many bodies (forced here before every step): bodies change
    owner wholesale, the trajectories stay those of one engine."""
    b = physics.plummer(20000, seed=6, a=2e3)
    seen = []
    got = run_distributed(b, 4, 5, precision="f32", theta=0.6, rebalance_at=0.0,
                          per_step=lambda s, dt: seen.append((dt.rebalances, list(dt.lives))))
    want = run_one(b, 5, precision="f32", theta=0.6)
    assert seen[-1][0] == 4 and all(sum(l) == 20000 for _, l in seen)
    assert np.array_equal(got[0], want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a, w)


def test_energy_of_a_distributed_run(gpu):
    """The ranks' shares of the kinetic energy, the momentum and the tree potential add up to what one engine reports for
    the same state (above 262,144 bodies one engine also sums the potential over the nodes its walk accepts)."""
    n, world = 300000, 4
    b = physics.plummer(n, seed=9, a=2e3)
    per = (n + world - 1) // world
    ranks = [DistRank(r, world, precision="f32", theta=0.5, eps=1.0) for r in range(world)]
    try:
        for r in ranks:
            a, z = r.rank * per, min(n, (r.rank + 1) * per)
            r.upload(int(per * 1.5), b.pos[a:z], b.vel[a:z], b.mass[a:z], a)
            r.use_torch_stream()
        dt = DistributedTree(ranks, LocalComm(world))
        dt.distribute()
        dt.step(2)
        ke, pe, mom = dt.energy()
        dt.step(1)                                     # the prepared state is used, not rebuilt
        got = dt.download()
    finally:
        for r in ranks:
            r.close()
    with Engine(mode="tree", precision="f32", theta=0.5, eps=1.0, exact_sums=True) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(2)
        ke1, pe1, mom1 = e.energy()
        e.step(1)
        p, v, m, ids = e.download()
    assert abs(ke - ke1) <= 1e-12 * ke1 and abs(pe - pe1) <= 1e-10 * abs(pe1)
    assert np.abs(mom - mom1).max() <= 1e-12 * (b.mass * np.linalg.norm(b.vel, axis=1)).sum()
    assert np.array_equal(got[0], ids) and np.array_equal(got[1], p) and np.array_equal(got[2], v)


def test_long_run_of_colliding_galaxies_with_owners_changing(gpu):
    """150 steps of two galaxies on approach on four ranks: thousands of bodies change owner, the splitters are chosen
    anew whenever the fullest rank is 2 % over the mean, and the state still equals one engine's bit for bit; energy and
    momentum of the distributed run stay within the drift of the integrator."""
    b = physics.galaxy_pair(20000, seed=3)
    moved, rebal = [], []
    e0 = []

    def watch(s, dt):
        moved.append(sum(dt.last_migrated))
        rebal.append(dt.rebalances)
        if s in (0, 149):
            e0.append(dt.energy())
    got = run_distributed(b, 4, 150, precision="f32", theta=0.6, eps=1.0, rebalance_at=1.02, per_step=watch)
    want = run_one(b, 150, precision="f32", theta=0.6, eps=1.0)
    assert np.array_equal(got[0], want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a, w)
    assert sum(moved[1:]) > 1000 and rebal[-1] >= 1
    (ke0, pe0, p0), (ke1, pe1, p1) = e0
    assert abs((ke1 + pe1) - (ke0 + pe0)) <= 2e-2 * abs(ke0 + pe0)
    scale = (b.mass * np.linalg.norm(b.vel, axis=1)).sum()
    assert np.abs(p1 - p0).max() <= 1e-3 * scale


def test_one_rank_over_rccl(gpu):
    """The collectives of TorchComm on the "nccl" backend (RCCL) with device buffers, on a one-rank group: the rank owns
    every key, its one fragment is the whole tree, and the run equals one engine's."""
    import os
    import torch
    import torch.distributed as dist
    from nbody_amd.disttree import TorchComm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        except Exception as ex:                      # no rendezvous on this box: nothing to do with the engine
            pytest.skip(f"cannot create a one-rank RCCL group here: {ex}")
    try:
        b = physics.plummer(30000, seed=4, a=2e3)
        r = DistRank(0, 1, precision="f32", theta=0.6)
        try:
            r.upload(40000, b.pos, b.vel, b.mass, 0)
            r.use_torch_stream()
            comm = TorchComm()
            assert comm.backend == "nccl"
            dt = DistributedTree([r], comm)
            dt.distribute()
            dt.step(4)
            got = dt.download()
        finally:
            r.close()
        want = run_one(b, 4, precision="f32", theta=0.6)
        for a, w in zip(got, want):
            assert np.array_equal(a, w)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("precision,n,devices", [("f32", 20001, [0, 0, 0]), ("f64", 5003, [0, 0]), ("f32", 100000, [0] * 8)])
def test_distributed_tree_behind_one_engine_handle(gpu, precision, n, devices):
    """nbe_config.distributed with num_devices > 1: the engine itself runs the distributed tree on its GPUs (one host
    thread each, peer copies for the exchanges; here several shards on one GPU).  Upload, steps, download, statistics and
    energies through the ordinary calls; the state equals one engine's bit for bit."""
    b = physics.plummer(n, seed=8, a=2e3)
    with Engine(mode="tree", precision=precision, theta=0.6, devices=devices, distributed=True) as e:
        e.upload(b.pos, b.vel, b.mass, ids=np.arange(n, dtype=np.uint64) + 1000)
        e.step(2)
        ke, pe, mom = e.energy()
        e.step(2)
        p, v, m, ids = e.download()
        st = e.stats()
        acc, alive = e.accel()
        for call in (e.tree_nodes, e.step_timed, lambda: e.time_force_kernel(1)):
            with pytest.raises(EngineError, match="distributed = 1|num_devices"):
                call()
    assert st.n == n and st.n_alive == n and st.steps == 4
    want = run_one(b, 4, precision=precision, theta=0.6)
    assert np.array_equal(ids, want[0] + 1000)
    assert np.array_equal(p, want[1]) and np.array_equal(v, want[2]) and np.array_equal(m, want[3])
    with Engine(mode="tree", precision=precision, theta=0.6, exact_sums=True) as one:
        one.upload(b.pos, b.vel, b.mass)
        one.step(2)
        ke1, pe1, mom1 = one.energy()              # (the exact pair sum at these sizes; the distributed run sums over accepted nodes)
        one.step(2)
        acc1, alive1 = one.accel()
    assert abs(ke - ke1) <= 1e-6 * ke1 and abs(pe - pe1) <= 5e-3 * abs(pe1)
    # nbe_accel: every GPU's accelerations of its own bodies, in upload order
    assert np.array_equal(alive, alive1) and alive.all()
    assert np.array_equal(acc, acc1)


@pytest.mark.parametrize("precision,devices", [("f32", [0, 0, 0]), ("mixed", [0, 0])])
def test_frame_stream_of_a_distributed_engine(gpu, precision, devices):
    """main.go:90-101 on the distributed tree: no GPU holds all bodies, every GPU sends its own into one frame.  The
    frames carry the bodies as each step is about to read them: the same rows (by id) as one engine streams, and frame k
    is the state that a download after k * every steps returns, narrowed to fp32."""
    n, every, nsteps = 6001, 2, 7
    b = physics.plummer(n, seed=4, a=2e3)
    ids = np.arange(n, dtype=np.uint64) * 3 + 7
    got = {}
    with Engine(mode="tree", precision=precision, theta=0.6, devices=devices, distributed=True) as e:
        e.upload(b.pos, b.vel, b.mass, ids=ids)
        e.frame_stream_begin(lambda f, xyzm, i, r: got.__setitem__(f, (xyzm.copy(), i.copy(), r)), every=every, slots=2, first_frame=5)
        with pytest.raises(EngineError, match="already on"):
            e.frame_stream_begin(lambda *a: None)
        e.step(nsteps)
        e.frame_stream_end()
        e.step(1)                                 # streaming is over: no further frame
        e.frame_stream_end()                      # ending twice is harmless
    assert sorted(got) == [5, 6, 7, 8]            # steps 0, 2, 4, 6
    want = {}
    with Engine(mode="tree", precision=precision, theta=0.6, exact_sums=True) as one:
        one.upload(b.pos, b.vel, b.mass, ids=ids)
        one.frame_stream_begin(lambda f, xyzm, i, r: want.__setitem__(f, (xyzm.copy(), i.copy())), every=every, slots=2, first_frame=5)
        one.step(nsteps)
        one.frame_stream_end()
    for f in sorted(want):
        xyzm, i, r = got[f]
        assert r is None and len(i) == n
        o, ow = np.argsort(i), np.argsort(want[f][1])
        assert np.array_equal(i[o], want[f][1][ow])
        if precision == "f32":
            assert np.array_equal(xyzm[o], want[f][0][ow])
        else:
            assert np.abs(xyzm[o] - want[f][0][ow]).max() <= 1e-6 * np.abs(want[f][0]).max()
    # the first frame is the upload itself
    o = np.argsort(got[5][1])
    assert np.array_equal(got[5][0][o][:, :3], b.pos.astype(np.float32))


def test_distributed_needs_several_devices_and_tree_mode(gpu):
    with pytest.raises(EngineError, match="num_devices > 1"):
        Engine(mode="tree", distributed=True)
    with pytest.raises(EngineError, match="needs tree mode"):
        Engine(mode="direct", devices=[0, 0], distributed=True)
    with pytest.raises(EngineError, match="needs tree mode"):
        Engine(mode="tree", devices=[0, 0], distributed=True, per_wave=True)


def test_a_rank_can_be_loaded_again(gpu):
    """nbe_dist_upload on an engine that has run before starts a new run (other capacity, other bodies)."""
    b1, b2 = physics.plummer(3000, seed=1, a=2e3), physics.plummer(5000, seed=2, a=2e3)
    ranks = [DistRank(r, 2, precision="f32", theta=0.6) for r in range(2)]
    try:
        for b in (b1, b2):
            n = len(b)
            for r in ranks:
                a, z = r.rank * (n // 2), (r.rank + 1) * (n // 2)
                r.upload(n, b.pos[a:z], b.vel[a:z], b.mass[a:z], a)
                r.use_torch_stream()
            dt = DistributedTree(ranks, LocalComm(2))
            dt.distribute()
            dt.step(3)
            got = dt.download()
            want = run_one(b, 3, precision="f32", theta=0.6)
            for a, w in zip(got, want):
                assert np.array_equal(a, w)
    finally:
        for r in ranks:
            r.close()


@pytest.mark.parametrize("precision,world", [("f64", 3), ("f32", 4)])
def test_collisions_on_the_distributed_tree_equal_one_engine(gpu, precision, world):
    """The reference's default run has collisions on (main.go:24, 151-167; tree.go:245-281).  A crowded scene over 20
    steps on several ranks against one engine: every step's queue length, the merges so far and the bodies left are
    equal, and so are the survivors' identities, masses, positions and velocities at the end, bit for bit (both sum their
    cells exactly, so the forces are the same bits, and every rank merges its own pairs in the reference's queue order).
    The splitters are aligned to cells of at least four times the radius cap, here 64 m cells."""
    rng = np.random.default_rng(5)
    n = 6000
    pos = rng.uniform(-150.0, 150.0, (n, 3))
    vel = rng.normal(0, 1e-3, (n, 3))
    mass = np.abs(rng.normal(1e3, 500, n)) + 1
    rad = rng.uniform(0.5, 3.0, n)
    nsteps, cap = 20, 8.0
    kw = dict(precision=precision, theta=0.7, dt=50.0)
    want_steps = []
    with Engine(mode="tree", collisions=True, exact_sums=True, **kw) as e:
        e.upload(pos, vel, mass)
        e.set_radius(rad)
        for _ in range(nsteps):
            e.step(1)
            st = e.stats()
            want_steps.append((int(st.collision_pairs), int(st.merges), int(st.n_alive)))
        wp, wv, wm, wid = e.download()
    assert want_steps[-1][1] > 100                                  # the scene does collide
    per = (n + world - 1) // world
    ranks = [DistRank(r, world, collisions=True, **kw) for r in range(world)]
    got_steps = []
    try:
        for r in ranks:
            a, z = min(n, r.rank * per), min(n, (r.rank + 1) * per)
            r.upload(int(per * 2.5) + 1024, pos[a:z], vel[a:z], mass[a:z], a)
            r.set_radius(rad[a:z], cap)
            r.use_torch_stream()
        dt = DistributedTree(ranks, LocalComm(world))
        dt.distribute()
        for _ in range(nsteps):
            dt.step(1)
            sts = [r.engine.stats() for r in ranks]
            got_steps.append((sum(int(s.collision_pairs) for s in sts), sum(int(s.merges) for s in sts), sum(int(s.n_alive) for s in sts)))
        idx, p, v, m = dt.download()
    finally:
        for r in ranks:
            r.close()
    assert got_steps == want_steps
    assert np.array_equal(idx, wid) and np.array_equal(m, wm)
    assert np.array_equal(p, wp) and np.array_equal(v, wv)


def test_collisions_behind_one_handle_and_a_radius_above_the_cap(gpu):
    """Engine(devices=[0, 0, 0], distributed=True, collisions=True): the engine drives the ranks itself; the same merges
    and survivors as one engine.  Radii above the cap, or given after the run has started, are refused."""
    rng = np.random.default_rng(9)
    n = 4000
    pos = rng.uniform(-100.0, 100.0, (n, 3)); vel = rng.normal(0, 1e-3, (n, 3))
    mass = np.abs(rng.normal(1e3, 500, n)) + 1; rad = rng.uniform(0.5, 3.0, n)
    kw = dict(mode="tree", precision="f64", theta=0.7, dt=50.0, collisions=True)
    frames_one, frames_dist = [], []
    with Engine(exact_sums=True, **kw) as one:
        one.upload(pos, vel, mass); one.set_radius(rad)
        one.frame_stream_begin(lambda f, xyzm, ids, radius: frames_one.append((f, np.sort(ids), radius[np.argsort(ids)])), every=2, slots=3)
        one.step(10)
        one.frame_stream_end()
        want, wst, wrad = one.download(), one.stats(), one.get_radius()
    with Engine(devices=[0, 0, 0], distributed=True, **kw) as e:
        e.upload(pos, vel, mass); e.set_radius(rad)
        e.frame_stream_begin(lambda f, xyzm, ids, radius: frames_dist.append((f, np.sort(ids), radius[np.argsort(ids)])), every=2, slots=3)
        e.step(10)
        e.frame_stream_end()
        got, gst, grad = e.download(), e.stats(), e.get_radius()
        with pytest.raises(EngineError, match="right after"):
            e.set_radius(rad)
    assert wst.merges > 10 and gst.merges == wst.merges and gst.n_alive == wst.n_alive
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
    # the frames hold the bodies that are left, absorbed ones gone at once (main.go:91-95)
    assert [f[0] for f in frames_dist] == [f[0] for f in frames_one] == [0, 1, 2, 3, 4]
    for (_, a, ra), (_, w, rw) in zip(frames_dist, frames_one):
        assert np.array_equal(a, w) and np.array_equal(ra, rw)
    assert len(frames_one[-1][1]) < n
    # merged bodies have grown (physics.go:139): the radii of the survivors, and those the frames carry
    assert np.array_equal(grad, wrad) and grad.max() > rad.max()
    r = DistRank(0, 1, precision="f32", collisions=True)
    try:
        r.upload(2000, pos[:1000], vel[:1000], mass[:1000], 0)
        big = rad[:1000].copy(); big[7] = 5.0
        with pytest.raises(EngineError, match="above the cap"):
            r.set_radius(big, 4.0)
        r.set_radius(big, 6.0)
        r.use_torch_stream()
        dt = DistributedTree([r], LocalComm(1))
        dt.distribute()
        dt.step(1)
        with pytest.raises(EngineError, match="right after"):
            r.set_radius(big, 6.0)
    finally:
        r.close()


@pytest.mark.parametrize("scene", ["mass_ratio", "tiny_masses", "huge_masses", "offset_bound", "massless"])
def test_exact_cell_sums_hold_for_extreme_scenes(gpu, scene):
    """NBE_SUMS_EXACT quantises every body's mass and moments to 2^-126 of (total mass x the bound's reach): the cells of
    fp64 node records must come out as the child-by-child fp64 sums do, to a few ulps, whatever the spread of the masses,
    their scale, or where the bound sits; and a permutation of the upload order must not change one bit of them."""
    rng = np.random.default_rng(42)
    n = 20000
    kw = dict(mode="tree", precision="f64", theta=0.5)
    pos = rng.normal(0, 3e3, (n, 3)).clip(-3e4, 3e4)
    mass = np.abs(rng.normal(1e3, 300, n)) + 1
    if scene == "mass_ratio":
        mass[::1000] = 1e15                      # 20 bodies outweigh the rest by twelve orders of magnitude
    elif scene == "tiny_masses":
        mass *= 1e-25
    elif scene == "huge_masses":
        mass *= 1e22
    elif scene == "offset_bound":
        pos = pos * 0.01 + np.array([5e5, -2e5, 7e5])
        kw.update(bound_center=[5e5, -2e5, 7e5], bound_width=3000.0)
    elif scene == "massless":
        mass[rng.random(n) < 0.3] = 0.0
    tabs = []
    for extra in (dict(tree_sums=True), dict(exact_sums=True)):
        with Engine(**kw, **extra) as e:
            e.upload(pos, np.zeros_like(pos), mass)
            tabs.append(e.tree_nodes())
    t, x = tabs
    assert t.shape == x.shape and np.array_equal(t[:, 4:], x[:, 4:])
    cells = t[:, 4] > 0
    assert np.abs(x[cells, 3] - t[cells, 3]).max() <= 1e-13 * t[cells, 3].max()
    heavy = cells & (t[:, 3] > 0)
    scale = np.abs(pos).max()
    assert np.abs(x[heavy, :3] - t[heavy, :3]).max() <= 1e-12 * scale
    perm = rng.permutation(n)
    with Engine(**kw, exact_sums=True) as e:
        e.upload(pos[perm], np.zeros_like(pos), mass[perm])
        y = e.tree_nodes()
    assert np.array_equal(x[cells], y[cells])
