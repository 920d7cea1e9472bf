"""Benchmark configuration 5 at its full size on one GPU: two 8,388,608-body galaxies on the approach course of the
reference's cores (main.go:35-36), Barnes-Hut theta = 0.5, fp32 pair terms with fp64 sums and an fp64 kick-drift
integrator (physics.go:101-115), 100 fused steps of dt = 3600 s with softening eps = 1 m."""
import numpy as np
import pytest

from nbody_amd import Engine, physics
from oracle import oracle as O
from tests.util import relerr

pytestmark = pytest.mark.gpu


def test_two_8m_body_galaxies_mixed_precision_100_steps(gpu):
    """Size-independent checks at 16,777,216 bodies:
    (a) nothing is lost: every body survives the 100 sorts, IDs and masses come back intact and in upload order;
    (b) a 48-body sample of the accelerations agrees with the fp64 direct sum over all 16.7M sources (the oracle's
        row loop) within what theta = 0.5 gives: median <= 1e-2, max <= 8e-2;
    (c) drift over 100 steps, bounds asserted: |dE/E0| <= 2e-3 (the potential is summed over the nodes the walk
        accepts, itself good to ~1e-3) and |dP| / sum m|v| <= 1e-5.  Measured: 6.3e-4 and 1.4e-6."""
    eps, nsteps = 1.0, 100
    b = physics.galaxy_pair(1 << 23, seed=1)
    n = len(b)
    ids = np.arange(n, dtype=np.uint64) * 3 + 7
    with Engine(mode="tree", precision="mixed", theta=0.5, eps=eps) as e:
        e.upload(b.pos, b.vel, b.mass, ids)
        acc, alive = e.accel()
        st0 = e.stats()
        ke0, pe0, p0 = e.energy()
        e.step(nsteps)
        ke1, pe1, p1 = e.energy()
        st1 = e.stats()
        p, v, m, got_ids = e.download()
    assert alive.all() and st0.n_alive == n and st1.n_alive == n and st1.steps == nsteps
    assert st0.tree_nodes == n + st0.tree_cells and 0 < st0.tree_cells < n
    assert np.array_equal(got_ids, ids) and np.array_equal(m, b.mass)
    assert np.isfinite(p).all() and np.isfinite(v).all()
    assert np.abs(p - b.pos).max() < 100 * 3600.0 * 0.05            # nobody moves faster than 0.05 m/s here
    sample = np.arange(4321, n, n // 48)[:48]
    ref = np.concatenate([O.direct_rows(b.mass, b.pos, t0=int(t), t1=int(t) + 1, eps=eps) for t in sample]) / b.mass[sample][:, None]
    er = relerr(acc[sample], ref)
    assert np.median(er) <= 1e-2 and er.max() <= 8e-2, (np.median(er), er.max())
    e0, e1 = ke0 + pe0, ke1 + pe1
    # two virialised Plummer spheres (K = 3 pi G M^2 / (64 a) each) plus the bulk motion of the approach course
    M, a = 1e9, 1.5e3
    c0, c1 = physics.default_cores()
    bulk = 0.5 * M * sum(c.Vx ** 2 + c.Vy ** 2 + c.Vz ** 2 for c in (c0, c1))
    assert abs(ke0 / (2 * 3 * np.pi * physics.G * M * M / (64 * a) + bulk) - 1.0) < 0.05
    assert abs((e1 - e0) / e0) <= 2e-3, abs((e1 - e0) / e0)
    pscale = (b.mass * np.linalg.norm(b.vel, axis=1)).sum()
    assert np.linalg.norm(p1 - p0) / pscale <= 1e-5, np.linalg.norm(p1 - p0) / pscale


def test_two_8m_body_galaxies_on_the_distributed_tree_equal_one_engine_bit_for_bit(gpu):
    """The same scene and precision with every body on one of eight ranks only (eight GPUs' worth behind one handle, on
    this GPU): after three steps the state equals that of one engine that sums its cells exactly, bit for bit, all
    16,777,216 bodies.  Mixed precision cuts a body's fp32 partial sums at nodes picked by their records, so the bodies
    that happen to share its wavefront on a rank do not change its result."""
    b = physics.galaxy_pair(1 << 23, seed=1)
    n = len(b)
    out = []
    for kw in (dict(exact_sums=True), dict(devices=[0] * 8, distributed=True)):
        with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, **kw) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(3)
            p, v, m, ids = e.download()
            assert e.stats().n_alive == n
        o = np.argsort(ids, kind="stable")
        out.append((ids[o], p[o], v[o], m[o]))
    for a, w in zip(out[1], out[0]):
        assert np.array_equal(a, w)
