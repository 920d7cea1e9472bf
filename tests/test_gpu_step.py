"""Whole-step behaviour on the GPU: energy and momentum drift, sharded stepping, timing/stat plumbing."""
import os
import sys

import numpy as np
import pytest

from nbody_amd import Engine, EngineError, physics
from oracle import oracle as O
from tests.test_sharded_gloo import run_ranks

pytestmark = pytest.mark.gpu


def test_energy_matches_oracle(gpu):
    b = physics.plummer(3000, seed=2)
    for precision in ("f32", "f64"):
        with Engine(mode="direct", precision=precision, eps=3.0) as e:
            e.upload(b.pos, b.vel, b.mass)
            ke, pe, mom = e.energy()
        oke, ope, omom = O.energy(b.mass, b.pos, b.vel, eps=3.0)
        tol = 1e-12 if precision == "f64" else 1e-5
        assert abs(ke - oke) <= tol * oke and abs(pe - ope) <= tol * abs(ope)
        assert np.abs(mom - omom).max() <= tol * (b.mass * np.linalg.norm(b.vel, axis=1)).sum()


@pytest.mark.parametrize("mode,precision", [("direct", "f64"), ("direct", "mixed"), ("tree", "mixed"), ("tree", "f64")])
def test_energy_and_momentum_drift_two_galaxies(gpu, mode, precision):
    """Scaled-down benchmark configuration 5 (two Plummer spheres on the reference cores' approach course): kick-drift
    Euler with eps = 20 m and dt = 3600 s (dynamical time of each sphere ~ 2e5 s, so dt ~ t_dyn/55).
    Bounds over 300 steps: |dE/E0| <= 2e-3; |dP| / sum m|v| <= 1e-10 for the direct sum (pairwise forces
    cancel up to rounding) and <= 2e-3 for Barnes-Hut at theta = 0.5 (cell forces are not pairwise)."""
    b = physics.galaxy_pair(4096, a=1.5e3, total_mass=1e9, seed=5)
    with Engine(mode=mode, precision=precision, theta=0.5, eps=20.0) as e:
        e.upload(b.pos, b.vel, b.mass)
        ke0, pe0, p0 = e.energy()
        e.step(300)
        ke1, pe1, p1 = e.energy()
        st = e.stats()
    assert st.n_alive == len(b) and st.steps == 300
    e0, e1 = ke0 + pe0, ke1 + pe1
    assert abs((e1 - e0) / e0) <= 2e-3
    pscale = (b.mass * np.linalg.norm(b.vel, axis=1)).sum()
    assert np.linalg.norm(p1 - p0) / pscale <= (1e-10 if mode == "direct" and precision == "f64" else
                                                 1e-7 if mode == "direct" else 2e-3)


def test_step_timed_reports_phases(gpu):
    b = physics.plummer(1 << 16, seed=1)
    with Engine(mode="tree", precision="f32", theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(1)
        st = e.step_timed()
        assert st.ms_sort > 0 and st.ms_build > 0 and st.ms_force > 0
        assert abs(st.ms_total - (st.ms_sort + st.ms_build + st.ms_force)) < 0.05 * st.ms_total + 0.01
        assert st.interactions > 0 and st.steps == 2
        ms = e.time_force_kernel(3)
        assert 0 < ms < 100
    with Engine(mode="direct", precision="f32") as e:
        e.upload(b.pos, b.vel, b.mass)
        st = e.step_timed()
        assert st.ms_force > 0 and st.interactions == (1 << 16) * ((1 << 16) - 1)


def test_call_order_errors(gpu):
    from nbody_amd import EngineError
    b = physics.uniform_cube(100, seed=1)
    with Engine(mode="tree") as e:
        e.upload(b.pos, b.vel, b.mass)
        with pytest.raises(EngineError):
            e.shard_advance()                 # not prepared
        e.set_shard(0, 50)
        with pytest.raises(EngineError):
            e.step(1)                         # partial shard must use the shard calls
        with pytest.raises(EngineError):
            e.set_shard(90, 20)               # outside the body range
        e.shard_prepare(); e.shard_advance()
        with pytest.raises(EngineError):
            e.download()                      # between advance and commit
        e.shard_commit()
        with pytest.raises(EngineError):
            e.shard_commit()


@pytest.mark.parametrize("mode,precision,n", [("tree", "f32", 4096), ("tree", "f64", 1001), ("direct", "f32", 4096),
                                              ("direct", "mixed", 999), ("tree", "mixed", 3001), ("tree", "mixed", 4096)])
def test_two_ranks_on_one_gpu_match_single_engine(gpu, tmp_path, mode, precision, n):
    """world_size 2 over gloo, both ranks on cuda:0, real engines: each advances half the targets and the
    halves are all-gathered; the result must equal one engine stepping everything.  The tree walk gives each
    body its private result whatever wave it sits in, so tree mode must agree bit for bit; the direct sum
    may split its source range differently for a half-size launch (summation order), hence a tolerance."""
    # 4096 bodies divide into two 256-aligned ranges per rank: that case also runs the stepper's split exchange
    res = run_ranks(tmp_path, 2, [mode, str(n), "3", "gpu", precision] + (["split"] if n % 1024 == 0 else []), timeout=240)
    assert np.array_equal(res[0]["pos"], res[1]["pos"]) and np.array_equal(res[0]["vel"], res[1]["vel"])
    b = physics.makebodies(n - 2, seed=11)
    with Engine(mode=mode, precision=precision, theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(3)
        p, v, m, ids = e.download()
    if mode == "tree":
        assert np.array_equal(res[0]["pos"], p) and np.array_equal(res[0]["vel"], v)
    else:
        assert np.abs(res[0]["pos"] - p).max() <= 1e-6 * np.abs(p).max()
        assert np.abs(res[0]["vel"] - v).max() <= 1e-5 * np.abs(v).max()


@pytest.mark.parametrize("mode,precision,n", [("tree", "f32", 8192), ("tree", "mixed", 30001), ("direct", "f32", 20000)])
def test_eight_ranks_on_one_gpu_match_single_engine(gpu, tmp_path, mode, precision, n):
    """The same with world_size 8, the number of ranks of the full node (8,192 bodies split into the two ranges per rank of
    the stepper's overlapped exchange)."""
    world = 8
    res = run_ranks(tmp_path, world, [mode, str(n), "3", "gpu", precision] + (["split"] if n % (512 * world) == 0 else []), timeout=300)
    for r in res[1:]:
        assert np.array_equal(res[0]["pos"], r["pos"]) and np.array_equal(res[0]["vel"], r["vel"])
    b = physics.makebodies(n - 2, seed=11)
    with Engine(mode=mode, precision=precision, theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(3)
        p, v, m, ids = e.download()
    if mode == "tree":
        assert np.array_equal(res[0]["pos"], p) and np.array_equal(res[0]["vel"], v)
    else:
        assert np.abs(res[0]["pos"] - p).max() <= 1e-6 * np.abs(p).max()


@pytest.mark.parametrize("mode,precision", [("direct", "f32"), ("tree", "f32"), ("tree", "mixed"), ("direct", "f64")])
def test_frame_stream_delivers_the_state_before_each_step(gpu, mode, precision):
    """main.go:90-101: a copy of the live bodies goes to the output workers before every iteration.  The streamed
    frames (side stream + pinned ring + delivery thread) must equal what a synchronous download would have
    returned at the same point, and arrive in order, also when only a few ring slots exist."""
    b = physics.makebodies(3000, seed=4)
    nsteps, every = 12, 3
    want = {}
    with Engine(mode=mode, precision=precision, theta=0.7) as e:
        e.upload(b.pos, b.vel, b.mass, b.ids + 100)
        for s in range(nsteps):
            if s % every == 0:
                p, v, m, ids = e.download()
                want[7 + s // every] = (p, m, ids)
            e.step(1)
    got = []
    with Engine(mode=mode, precision=precision, theta=0.7) as e:
        e.upload(b.pos, b.vel, b.mass, b.ids + 100)
        e.frame_stream_begin(lambda f, xyzm, ids, radius: got.append((f, xyzm, ids)), every=every, slots=2, first_frame=7)
        e.step(nsteps)
        e.frame_stream_end()
        e.step(2)                      # streaming is off again
    assert [f for f, _, _ in got] == sorted(want)
    for f, xyzm, ids in got:
        p, m, want_ids = want[f]
        order = np.argsort(ids)
        assert np.array_equal(ids[order], want_ids)
        assert np.array_equal(xyzm[order, :3], p.astype(np.float32))
        assert np.array_equal(xyzm[order, 3], m.astype(np.float32))


@pytest.mark.parametrize("mode", ["direct", "tree"])
def test_streamed_frames_equal_the_oracles_states(gpu, mode):
    """Frame k of the stream is the state before step k*every (main.go:90-101): it must equal the CPU oracle's state
    after k*every steps of main.go:104-178 (fp64 engine; frames carry fp32 x, y, z, mass, so positions agree to
    fp32 rounding of the oracle's values plus the fp64 engine's 1e-10 trajectory tolerance)."""
    b = physics.makebodies(598, seed=6)
    every, nframes = 3, 4
    got = {}
    with Engine(mode=mode, precision="f64", theta=0.7) as e:
        e.upload(b.pos, b.vel, b.mass, b.ids + 40)
        e.frame_stream_begin(lambda f, xyzm, ids, radius: got.__setitem__(f, (xyzm, ids)), every=every, slots=3, first_frame=2)
        e.step(every * (nframes - 1) + 1)
        e.frame_stream_end()
    assert sorted(got) == [2, 3, 4, 5]
    kw = dict(mode="tree", com_mode="true", theta=0.7, drop_tail=False, max_depth=21) if mode == "tree" else dict(mode="direct")
    for k in range(nframes):
        pos, vel, alive, _ = O.step(b.mass, b.pos, b.vel, nsteps=k * every, **kw)
        xyzm, ids = got[2 + k]
        order = np.argsort(ids)
        assert alive.all() and np.array_equal(ids[order], b.ids + 40)
        assert np.array_equal(xyzm[order, 3], b.mass.astype(np.float32))
        want = pos.astype(np.float32)
        err = np.abs(xyzm[order, :3].astype(np.float64) - pos).max()
        assert err <= 2e-7 * np.abs(pos).max(), err
        assert (xyzm[order, :3] == want).mean() > 0.99          # all but a few last-bit roundings are identical


def test_tree_potential_at_scale_matches_pair_sum(gpu):
    """Above 262,144 bodies nbe_energy takes the potential from the tree walk (same acceptance as the forces).
    On a 300k-body Plummer sphere it must agree with the exact pair sum of the direct engine to 2e-3, and the
    virial ratio 2K/|W| of the sampled sphere must be near 1."""
    b = physics.plummer(300000, seed=9)
    with Engine(mode="tree", precision="f32", theta=0.5, eps=1.0) as e:
        e.upload(b.pos, b.vel, b.mass)
        ke_t, pe_t, mom_t = e.energy()
        e.step(1)                           # the walk state left by energy() must not disturb stepping
        assert e.stats().n_alive == len(b)
    with Engine(mode="direct", precision="f32", eps=1.0) as e:
        e.upload(b.pos, b.vel, b.mass)
        ke_d, pe_d, mom_d = e.energy()
    assert abs(ke_t - ke_d) <= 1e-9 * ke_d
    assert abs(pe_t - pe_d) <= 2e-3 * abs(pe_d)
    assert abs(2 * ke_d / abs(pe_d) - 1.0) < 0.02


def test_upload_rejects_bad_masses_and_leaves_an_empty_engine(gpu):
    """A failed upload must not leave half a state behind: the engine is empty and usable afterwards."""
    rng = np.random.default_rng(0)
    pos = rng.uniform(-100, 100, (50, 3)); vel = np.zeros((50, 3)); mass = np.full(50, 10.0)
    with Engine(mode="tree") as e:
        e.upload(pos, vel, mass)
        bad = mass.copy(); bad[7] = -1.0
        with pytest.raises(EngineError, match=r"mass\[7\]"):
            e.upload(pos, vel, bad)
        bad[7] = np.nan
        with pytest.raises(EngineError, match="negative or NaN"):
            e.upload(pos, vel, bad)
        e.upload(pos, vel, mass)                 # the previous state is untouched by a rejected upload
        e.step(2)
        assert len(e.download()[0]) == 50


@pytest.mark.parametrize("mode", ["direct", "tree"])
def test_steps_can_be_captured_in_a_hip_graph(gpu, mode):
    """A step (collisions off) is kernels and memsets on the engine's stream, no allocation and no host read-back,
    so a host may capture it.  Two steps return every double buffer to where it was: replaying that graph k times
    equals 2k plain steps bit for bit."""
    import torch
    b = physics.makebodies(3000, seed=3)
    with Engine(mode=mode, precision="f32", theta=0.8) as plain:
        plain.upload(b.pos, b.vel, b.mass)
        plain.step(2 + 2 * 5)
        want = plain.download()
    s = torch.cuda.Stream()
    with Engine(mode=mode, precision="f32", theta=0.8) as e, torch.cuda.stream(s):
        e.upload(b.pos, b.vel, b.mass)
        e.set_stream(s.cuda_stream)
        e.step(2)                                   # allocates nothing afterwards
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            e.step(2)
        for _ in range(4):                          # the capture itself does not execute; 1 + 4 replays = 10 steps
            g.replay()
        g.replay()
        s.synchronize()
        got = e.download()
        e.set_stream(None)
    for a, w in zip(got, want):
        assert np.array_equal(a, w)


@pytest.mark.parametrize("mode,precision,extra", [("tree", "f32", {}), ("tree", "mixed", {}), ("direct", "f32", {}),
                                                  ("tree", "f32", {"per_wave": True}),
                                                  ("tree", "f64", {"ref_com": True, "drop_tail": True})])
@pytest.mark.parametrize("n", [20000, 20480])
def test_single_rank_rccl_stepper_with_overlapped_velocity_exchange(gpu, mode, precision, extra, n):
    """The RCCL path of ShardedStepper on one GPU (one-rank "nccl" group, collectives forced): positions are
    exchanged first, the velocity exchange runs beside the next step's key sort (nbe_shard_sort), and the result
    equals plain nbe_step bit for bit; also with the per-wave walk and with the reference's own arithmetic (whose
    build borrows a buffer of the sort that runs ahead)."""
    import torch
    import torch.distributed as dist
    from nbody_amd.sharded import EngineShard, ShardedStepper
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        except Exception as ex:                      # no rendezvous on this box: nothing to do with the engine
            pytest.skip(f"cannot create a one-rank RCCL group here: {ex}")
    try:
        # 20,480 bodies divide into two 256-aligned halves: the stepper then advances two ranges per rank and sends
        # the first one's positions while the second is computed
        b = physics.plummer(n, seed=8, a=2e3)
        with Engine(mode=mode, precision=precision, theta=0.6, **extra) as plain:
            plain.upload(b.pos, b.vel, b.mass)
            plain.step(6)
            want = plain.download()
        with Engine(mode=mode, precision=precision, theta=0.6, **extra) as e:
            e.upload(b.pos, b.vel, b.mass)
            st = ShardedStepper(EngineShard(e, device_index=0), rank=0, world=1, always_gather=True, split=n % 512 == 0)
            assert st.backend == "nccl" and st.split == (n % 512 == 0)
            st.step(6)
            torch.cuda.synchronize()
            got = e.download()
            e.set_stream(None)
        for a, w in zip(got, want):
            assert np.array_equal(a, w)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("n", [3000, 20000])
def test_sort_running_ahead_does_not_change_who_is_alive(gpu, n):
    """nbe_shard_sort sorts the keys of the next step while the state is still in the previous order.  A body that
    has just left the bound is found by that sort; until the state is reordered, downloads, energies and frames must
    keep describing the previous order: same survivors as an engine that has not sorted ahead."""
    b = physics.plummer(n, seed=4, a=2e3)                                # truncated at 24 km: all inside the bound
    pos, vel = b.pos.copy(), b.vel.copy()
    pos[17] = (32767.0, 10.0, -5.0); vel[17] = (1.0, 0.0, 0.0)          # leaves through the +x face in the first step
    with Engine(mode="tree", precision="f32", theta=0.6, eps=1.0) as plain, Engine(mode="tree", precision="f32", theta=0.6, eps=1.0) as e:
        for eng in (plain, e):
            eng.upload(pos, vel, b.mass)
        plain.step(1)
        e.shard_prepare(); e.shard_advance(); e.shard_commit()
        e.shard_sort()                                                   # finds n - 1 live bodies, state still unsorted
        want, got = plain.download(), e.download()
        assert len(want[0]) == n
        for a, w in zip(got, want):
            assert np.array_equal(a, w)
        ew, eg = plain.energy(), e.energy()
        # (block sums are added with atomics: the last bits depend on their order)
        assert abs(ew[0] - eg[0]) <= 1e-12 * ew[0] and abs(ew[1] - eg[1]) <= 1e-12 * abs(ew[1])
        assert np.abs(ew[2] - eg[2]).max() <= 1e-12 * (b.mass * np.linalg.norm(vel, axis=1)).sum()
        plain.step(1)
        e.shard_prepare(); e.shard_advance(); e.shard_commit()
        want, got = plain.download(), e.download()
        assert len(want[0]) == n - 1 and 17 not in want[3]
        for a, w in zip(got, want):
            assert np.array_equal(a, w)


@pytest.mark.parametrize("mode,precision,extra", [("tree", "f32", {}), ("tree", "mixed", {}), ("tree", "f32", {"per_wave": True}),
                                                  ("tree", "f64", {"ref_com": True, "drop_tail": True}), ("direct", "f32", {})])
def test_engine_driving_several_shards_itself(gpu, mode, precision, extra):
    """Engine(devices=[0, 0, 0]): one handle, three shards on this GPU, ranges exchanged by the engine's own peer
    copies (no torch.distributed).  Tree mode equals a one-shard engine bit for bit, with frames, energies and
    accelerations served by the first shard; the direct sum agrees to the rounding of its source-range split."""
    b = physics.plummer(20001, seed=8, a=2e3)
    frames = []
    with Engine(mode=mode, precision=precision, theta=0.6, **extra) as one:
        one.upload(b.pos, b.vel, b.mass)
        acc1, _ = one.accel()
        one.step(5)
        want = one.download()
        en1 = one.energy()
    with Engine(mode=mode, precision=precision, theta=0.6, devices=[0, 0, 0], **extra) as e:
        e.upload(b.pos, b.vel, b.mass)
        acc3, _ = e.accel()
        e.frame_stream_begin(lambda f, xyzm, ids, radius: frames.append((f, xyzm, ids)), every=2, slots=2)
        e.step(5)
        e.frame_stream_end()
        got = e.download()
        en3 = e.energy()
        assert e.stats().steps == 5
        with pytest.raises(EngineError, match="num_devices"):
            e.step_timed()
    assert [f for f, _, _ in frames] == [0, 1, 2] and all(len(ids) == len(b) for _, _, ids in frames)
    if mode == "tree":
        assert np.array_equal(acc1, acc3)
        for a, w in zip(got, want):
            assert np.array_equal(a, w)
        assert abs(en1[0] - en3[0]) <= 1e-12 * en1[0] and abs(en1[1] - en3[1]) <= 1e-12 * abs(en1[1])
    else:
        assert np.abs(got[0] - want[0]).max() <= 1e-6 * np.abs(want[0]).max()
        assert np.abs(got[1] - want[1]).max() <= 1e-5 * np.abs(want[1]).max()


@pytest.mark.parametrize("distributed", [False, True])
def test_upload_right_after_a_step_of_a_several_shard_engine(gpu, distributed):
    """nbe_step returns without waiting; an upload that follows at once frees and reallocates every shard's buffers
    while the other shards' copies may still read them, so the upload has to wait for all shards first.  Step, upload
    other bodies, step, with no synchronisation in between: the second scene must come out as from a fresh engine."""
    b1 = physics.plummer(30001, seed=21, a=2e3)
    b2 = physics.plummer(20001, seed=22, a=2e3)
    kw = dict(exact_sums=True) if distributed else {}
    with Engine(mode="tree", precision="f32", theta=0.6, **kw) as one:
        one.upload(b2.pos, b2.vel, b2.mass)
        one.step(3)
        want = one.download()
    with Engine(mode="tree", precision="f32", theta=0.6, devices=[0, 0, 0], distributed=distributed) as e:
        e.upload(b1.pos, b1.vel, b1.mass)
        e.step(3)
        e.upload(b2.pos, b2.vel, b2.mass)
        e.step(3)
        got = e.download()
    for a, w in zip(got, want):
        assert np.array_equal(a, w)


def test_two_engines_in_one_process_do_not_share_state(gpu):
    """Engines own their streams and workspaces: two of them stepped in turn give what each gives alone."""
    b1 = physics.makebodies(1500, seed=1); b2 = physics.plummer(2500, seed=2)
    alone = []
    for b, mode in ((b1, "tree"), (b2, "direct")):
        with Engine(mode=mode, precision="f32") as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(6)
            alone.append(e.download())
    with Engine(mode="tree", precision="f32") as e1, Engine(mode="direct", precision="f32") as e2:
        e1.upload(b1.pos, b1.vel, b1.mass); e2.upload(b2.pos, b2.vel, b2.mass)
        for _ in range(3):
            e1.step(2); e2.step(2)
        got = [e1.download(), e2.download()]
    for g, a in zip(got, alone):
        for x, y in zip(g, a):
            assert np.array_equal(x, y)


def test_tree_body_limit():
    """Tree mode addresses node records with 32-bit byte offsets: more than 2^26 bodies are refused at upload,
    before any array is read (the zero pages below are never touched)."""
    n = (1 << 26) + 1
    z3 = np.zeros((n, 3)); z1 = np.zeros(n)
    with Engine(mode="tree", precision="f32") as e:
        with pytest.raises(EngineError, match="2\\^26"):
            e.upload(z3, z3, z1)
        assert e.stats().n == 0


@pytest.mark.parametrize("precision", ["f32", "mixed"])
def test_shards_walking_two_bodies_per_lane_equal_one_engine(gpu, monkeypatch, precision):
    """The walk with two bodies per lane (large scenes; forced here) on three shards behind one handle: the one-engine
    run bit for bit, also in mixed precision."""
    monkeypatch.setenv("NBODY_WALK_BODIES_PER_LANE", "2")
    b = physics.plummer(30011, seed=4, a=2e3)
    with Engine(mode="tree", precision=precision, theta=0.6, eps=1.0) as one:
        one.upload(b.pos, b.vel, b.mass)
        one.step(4)
        want = one.download()
    with Engine(mode="tree", precision=precision, theta=0.6, eps=1.0, devices=[0, 0, 0]) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(4)
        got = e.download()
    for a, w in zip(got, want):
        assert np.array_equal(a, w)


@pytest.mark.parametrize("precision", ["mixed", "f32", "f64"])
@pytest.mark.parametrize("knob", ["1", "2"])
def test_ranges_that_begin_anywhere_equal_one_engine(gpu, monkeypatch, precision, knob):
    """nbe_shard_advance_range over ranges that begin at odd places (7, 1000, 4099 ...), so that every wavefront holds
    other bodies than in the one-engine run: the same bits in every precision, with either walk kernel.  (Mixed
    precision cuts a body's fp32 partial sums at nodes picked by their records, not by the wavefront's visits.)"""
    monkeypatch.setenv("NBODY_WALK_BODIES_PER_LANE", knob)
    n = 9001
    b = physics.plummer(n, seed=6, a=2e3)
    with Engine(mode="tree", precision=precision, theta=0.6, eps=0.5) as one:
        one.upload(b.pos, b.vel, b.mass)
        one.step(3)
        want = one.download()
    cuts = [0, 7, 1000, 4099, 4100, 8193, n]
    with Engine(mode="tree", precision=precision, theta=0.6, eps=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        for _ in range(3):
            e.shard_prepare()
            for a, z in zip(cuts[:-1], cuts[1:]):
                e.shard_advance_range(a, z - a)
            e.shard_commit()
        got = e.download()
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
