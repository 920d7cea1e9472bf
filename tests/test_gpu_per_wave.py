"""The per-wave opening criterion (NBE_WALK_PER_WAVE): the 128 Morton-consecutive bodies of a wavefront (two per
lane) open a cell together as soon as one of them would open it (tree.go:216-217 evaluated for each, decided for all).

It is not the reference's interaction set, so it is checked against (a) a replay of exactly that rule in numpy
over the engine's own node table (fp64 engine: accelerations <= 1e-10, interaction counts equal), (b) the fp64
direct sum, where it must be at least as accurate as the per-body criterion, and (c) the per-body walk, whose
interactions it must contain in number."""
import numpy as np
import pytest

from nbody_amd import Engine, physics
from oracle import oracle as O
from tests.util import f32, relerr

pytestmark = pytest.mark.gpu
G_CONST = 6.67408e-11


def replay_per_wave(nodes, theta):
    """Accelerations (without G) and force evaluations of the per-wave rule over a pre-order node table
    [x, y, z, m, edge, skip]; leaves (edge 0) are the bodies in sorted order."""
    com, mass, edge, skip = nodes[:, :3], nodes[:, 3], nodes[:, 4], nodes[:, 5].astype(np.int64)
    leaves = np.nonzero(edge == 0)[0]
    total = len(nodes)
    acc = np.zeros((len(leaves), 3))
    count = 0
    for w0 in range(0, len(leaves), 128):
        p = com[leaves[w0:w0 + 128]]
        a = np.zeros_like(p)
        cur = 0
        while cur < total:
            d = com[cur] - p
            r2 = (d * d).sum(1)
            with np.errstate(divide="ignore", invalid="ignore"):
                opens = edge[cur] / np.sqrt(r2) > theta
            if edge[cur] > 0 and opens.any():
                cur += 1
                continue
            nz = r2 > 0
            with np.errstate(divide="ignore", invalid="ignore"):
                wgt = np.where(nz, mass[cur] / (r2 * np.sqrt(r2)), 0.0)
            a += d * wgt[:, None]
            count += int(nz.sum())
            cur = skip[cur]
        acc[w0:w0 + 128] = a
    return acc, count


@pytest.mark.parametrize("n,theta", [(2, 0.5), (64, 0.5), (65, 1.0), (129, 0.5), (1000, 0.5), (4097, 0.7)])
def test_fp64_engine_equals_a_replay_of_the_rule(gpu, n, theta):
    rng = np.random.default_rng(n)
    pos = rng.normal(0, 3e3, (n, 3)); mass = np.abs(rng.normal(1e3, 500, n)) + 1
    with Engine(mode="tree", precision="f64", theta=theta, per_wave=True) as e:
        e.upload(pos, np.zeros((n, 3)), mass)
        acc, alive = e.accel()
        st = e.stats()
        nodes = np.asarray(e.tree_nodes()).reshape(-1, 6)
        p_sorted = nodes[nodes[:, 4] == 0][:, :3]
    ref_sorted, count = replay_per_wave(nodes, theta)
    # the replay is in sorted order: match bodies by position (distinct by construction)
    order = np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0])); order_s = np.lexsort((p_sorted[:, 2], p_sorted[:, 1], p_sorted[:, 0]))
    ref = np.zeros_like(acc); ref[order] = ref_sorted[order_s] * G_CONST
    assert alive.all()
    assert relerr(acc, ref).max() <= 1e-10
    assert st.interactions == count


@pytest.mark.parametrize("precision", ["f32", "mixed", "f64"])
def test_at_least_as_accurate_as_the_per_body_walk(gpu, precision):
    b = physics.plummer(20000, seed=5)
    pos, mass = f32(b.pos), f32(b.mass)
    exact = O.direct_rows(mass, pos, nthreads=8) / mass[:, None]
    err, inter = {}, {}
    for per_wave in (False, True):
        with Engine(mode="tree", precision=precision, theta=0.5, per_wave=per_wave) as e:
            e.upload(pos, b.vel, mass)
            acc, _ = e.accel()
            err[per_wave] = relerr(acc, exact)
            inter[per_wave] = e.stats().interactions
    assert inter[True] > inter[False]                      # everybody opens at least what it would open alone
    assert np.median(err[True]) <= np.median(err[False]) and np.quantile(err[True], 0.99) <= np.quantile(err[False], 0.99)
    assert np.median(err[True]) <= 1e-2


def test_steps_and_drift_with_the_per_wave_walk(gpu):
    b = physics.plummer(8192, seed=2)
    with Engine(mode="tree", precision="mixed", theta=0.5, eps=20.0, per_wave=True) as e:
        e.upload(b.pos, b.vel, b.mass)
        k0, p0, _ = e.energy()
        e.step(100)
        k1, p1, _ = e.energy()
        assert len(e.download()[3]) == 8192
    assert abs((k1 + p1) - (k0 + p0)) <= 2e-3 * abs(k0 + p0)


def test_per_wave_needs_plain_tree_mode(gpu):
    from nbody_amd.engine import EngineError
    for kw in (dict(mode="direct"), dict(mode="tree", ref_com=True), dict(mode="tree", drop_tail=True)):
        with pytest.raises(EngineError):
            Engine(per_wave=True, **kw)


def test_shards_form_the_same_groups_as_one_engine(gpu):
    """Groups are the aligned 128-body chunks of the sorted order, so uneven shards (boundaries inside a chunk)
    advance their bodies to the same bits as an engine that advances all of them."""
    import torch
    from nbody_amd.sharded import EngineShard
    b = physics.plummer(3000, seed=9)

    def advanced(first, count):
        with Engine(mode="tree", precision="f32", theta=0.5, per_wave=True) as e:
            e.upload(b.pos, b.vel, b.mass)
            sh = EngineShard(e)
            sh.set_shard(first, count)
            sh.prepare()
            sh.advance()
            pos, vel = sh.next_state()
            torch.cuda.synchronize()
            out = pos[first:first + count].cpu().numpy().copy(), vel[first:first + count].cpu().numpy().copy()
            sh.commit()
            return out

    full_pos, full_vel = advanced(0, 3000)
    for first, count in ((0, 1000), (1000, 1111), (2111, 889)):
        p, v = advanced(first, count)
        assert np.array_equal(p, full_pos[first:first + count]) and np.array_equal(v, full_vel[first:first + count])


def test_collisions_run_behind_the_per_wave_walk(gpu):
    """The collision search, queue and merge do not depend on who opened which cell for the forces: the first step
    of a dense cloud queues and merges exactly what it does behind the per-body walk."""
    b = physics.plummer(6000, seed=4, a=60.0)
    merges = {}
    for per_wave in (False, True):
        with Engine(mode="tree", precision="f64", theta=0.5, collisions=True, per_wave=per_wave) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(1)
            st = e.stats()
            merges[per_wave] = (st.merges, st.collision_pairs, st.n_alive)
    assert merges[True] == merges[False] and merges[True][0] > 0


def test_potential_summed_by_the_per_wave_walk(gpu):
    """Above 262,144 bodies nbe_energy sums the potential over the nodes the walk accepts: the per-wave walk's sum
    (finer nodes) agrees with the per-body walk's to the accuracy of either."""
    b = physics.plummer(300000, seed=6)
    pe = {}
    for per_wave in (False, True):
        with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, per_wave=per_wave) as e:
            e.upload(b.pos, b.vel, b.mass)
            pe[per_wave] = e.energy()[1]
    assert pe[True] < 0 and abs(pe[True] - pe[False]) <= 1e-3 * abs(pe[False])


@pytest.mark.parametrize("seed", range(12))
def test_fp64_engine_equals_the_oracles_group_rule(gpu, seed):
    """The C oracle restates the rule over its own pointer octree (oracle.forces_group: groups of 128 bodies in
    depth-first leaf order); it shares nothing with the engine.  Random scenes: sizes around the group size,
    clustered and uniform clouds, bodies outside the bound, theta 0.3 .. 1."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([1, 2, 127, 128, 129, 255, 700, 2000, 5000]))
    if rng.random() < 0.5:
        pos = rng.uniform(-3.4e4, 3.4e4, (n, 3))                 # a few outside the 2^16 bound
    else:
        c = rng.uniform(-2e4, 2e4, (n // 100 + 1, 3))
        pos = c[rng.integers(len(c), size=n)] + rng.normal(0, rng.choice([5.0, 300.0]), (n, 3))
    mass = np.abs(rng.normal(1e4, 5e3, n)) + 1.0
    theta = float(rng.choice([0.3, 0.5, 0.7, 1.0]))
    with Engine(mode="tree", precision="f64", theta=theta, per_wave=True) as e:
        e.upload(pos, np.zeros((n, 3)), mass)
        acc, alive = e.accel()
        st = e.stats()
    f, oalive, ost = O.forces_group(mass, pos, group=128, theta=theta)
    assert np.array_equal(alive.astype(bool), oalive.astype(bool))
    live = oalive.astype(bool)
    assert st.interactions == ost.interactions
    if live.sum() > 1:
        ref = f[live] / mass[live, None]
        assert relerr(acc[live], ref).max() <= 1e-10
