"""Direct-sum path on the GPU against the CPU oracle, through the C ABI (nbody_amd.Engine is a thin ctypes
binding).  Tolerances are relative errors of the per-body acceleration vector, |a - a_ref| / |a_ref|:
  fp64 engine   vs fp64 oracle on the same inputs:               max    <= 1e-12
  fp32 engine   vs fp64 oracle on the fp32-rounded inputs:       median <= 1e-6, p99.9 <= 1e-4
  mixed engine  (fp32 pair terms, fp64 sums) on rounded inputs:  median <= 1e-6, p99.9 <= 1e-4
The fp32 tail is made of bodies whose net force is a small residue of large cancelling pair terms, so the
error is also checked against the sum of pair magnitudes (<= 2e-6 everywhere)."""
import numpy as np
import pytest

from nbody_amd import Engine, physics
from oracle import oracle as O
from tests.util import f32, golden_cases, load_golden, relerr

pytestmark = pytest.mark.gpu
G = load_golden()
CASES = golden_cases(G)


def oracle_acc(mass, pos, eps=0.0, t0=0, t1=None):
    f = O.direct_rows(mass, pos, t0=t0, t1=t1, eps=eps, nthreads=8)
    t1 = len(mass) if t1 is None else t1
    return f / mass[t0:t1, None]


def engine_acc(pos, vel, mass, precision, **kw):
    with Engine(mode="direct", precision=precision, **kw) as e:
        e.upload(pos, vel, mass)
        acc, alive = e.accel()
        assert alive.all()
        return acc


def check(acc, ref, precision):
    er = relerr(acc, ref)
    if precision == "f64":
        assert er.max() <= 1e-12, er.max()
    else:
        assert np.median(er) <= 1e-6 and np.quantile(er, 0.999) <= 1e-4, (np.median(er), np.quantile(er, 0.999))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("precision", ["f32", "mixed", "f64"])
def test_accel_matches_golden_direct_forces(gpu, name, precision):
    mass, pos, vel = G[f"{name}/mass"], G[f"{name}/pos"], G[f"{name}/vel"]
    acc = engine_acc(pos, vel, mass, precision)
    if precision == "f64":
        ref = G[f"{name}/direct_force"] / mass[:, None]      # pair-loop order of main.go:107-128
        assert relerr(acc, ref).max() <= 1e-12
    else:
        check(acc, oracle_acc(f32(mass), f32(pos)), precision)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 1023, 1025, 4097])
@pytest.mark.parametrize("precision", ["f32", "mixed", "f64"])
def test_ragged_sizes(gpu, n, precision):
    rng = np.random.default_rng(n)
    pos = rng.uniform(-1e4, 1e4, (n, 3)); mass = np.abs(rng.normal(1e3, 500, n)) + 1; vel = np.zeros((n, 3))
    acc = engine_acc(pos, vel, mass, precision)
    if n == 1:
        assert np.all(acc == 0)
        return
    if precision == "f64":
        check(acc, oracle_acc(mass, pos), precision)
    else:
        check(acc, oracle_acc(f32(mass), f32(pos)), precision)


def test_empty_body_set(gpu):
    with Engine(mode="direct") as e:
        e.upload(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0))
        e.step(3)
        pos, vel, mass, ids = e.download()
        assert pos.shape == (0, 3) and ids.shape == (0,)
        assert e.accel()[0].shape == (0, 3)


def test_fp32_error_normalised_by_pair_magnitudes(gpu):
    rng = np.random.default_rng(5)
    n = 2048
    pos = f32(rng.uniform(-1e4, 1e4, (n, 3))); mass = f32(np.abs(rng.normal(1e3, 500, n)) + 1)
    acc = engine_acc(pos, np.zeros((n, 3)), mass, "f32")
    ref = oracle_acc(mass, pos)
    d = pos[None, :, :] - pos[:, None, :]
    r2 = (d * d).sum(-1); np.fill_diagonal(r2, np.inf)
    pair_sum = (O.G * mass[None, :] / r2).sum(1)          # sum_j |a_ij|
    assert (np.linalg.norm(acc - ref, axis=1) / pair_sum).max() <= 2e-6


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_softening(gpu, precision):
    rng = np.random.default_rng(9)
    n = 700
    pos = f32(rng.normal(0, 50, (n, 3))); mass = f32(np.full(n, 1e4))
    acc = engine_acc(pos, np.zeros((n, 3)), mass, precision, eps=5.0)
    check(acc, oracle_acc(mass, pos, eps=5.0), precision)


def test_coincident_bodies_give_zero_mutual_force_not_nan(gpu):
    """physics.go:136 divides by r*r = 0 for a coincident pair (NaN); the engine's stated deviation is that
    such a pair exerts no force on each other."""
    pos = np.array([[10.0, 10, 10], [10.0, 10, 10], [500.0, 10, 10]]); mass = np.array([1e3, 1e3, 1e3])
    for precision in ("f32", "f64"):
        acc = engine_acc(pos, np.zeros((3, 3)), mass, precision)
        assert np.isfinite(acc).all()
        want = O.G * 1e3 / 490.0**2
        assert abs(acc[0, 0] - want) <= 1e-6 * want and abs(acc[2, 0] + 2 * want) <= 2e-6 * want


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("precision,tol,median_tol", [("f64", 1e-11, 1e-15), ("mixed", 5e-5, 1e-7), ("f32", 1e-4, 2e-6)])
def test_ten_steps_match_golden(gpu, name, precision, tol, median_tol):
    """Ten iterations of main.go:104-178 (direct).  Error metric per body: max_k |x_k - x_ref_k| / max|x_ref|,
    likewise for velocities; bounds on the worst body and on the median body.  fp64 differs from the oracle
    by summation order only; in the fp32 and mixed modes the worst body is one that passes within metres of a
    1e10 kg core (unsoftened, physics.go:136), where fp32 pair terms lose digits."""
    mass, pos, vel = G[f"{name}/mass"], G[f"{name}/pos"], G[f"{name}/vel"]
    with Engine(mode="direct", precision=precision) as e:
        e.upload(pos, vel, mass)
        e.step(10)
        p, v, m, ids = e.download()
    assert list(ids) == list(range(len(mass)))
    pr, vr = G[f"{name}/step10_direct_pos"], G[f"{name}/step10_direct_vel"]
    ep = np.abs(p - pr).max(1) / np.abs(pr).max()
    ev = np.abs(v - vr).max(1) / np.abs(vr).max()
    assert ep.max() <= tol and ev.max() <= 10 * tol
    assert np.median(ep) <= median_tol and np.median(ev) <= 10 * median_tol
    assert np.array_equal(m, mass if precision != "f32" else mass.astype(np.float32).astype(np.float64))


def test_64k_uniform_cloud_fp32_against_oracle_sample(gpu):
    """Benchmark configuration 2: 65,536 bodies uniform in [-1e4,1e4)^3, m = 1e3, fp32 direct sum; the fp64 oracle
    is evaluated on a 2,048-target sample against all sources."""
    b = physics.uniform_cube(65536, seed=1)
    pos, mass = f32(b.pos), f32(b.mass)
    acc = engine_acc(pos, b.vel, mass, "f32")
    t0, t1 = 30000, 32048
    check(acc[t0:t1], oracle_acc(mass, pos, t0=t0, t1=t1), "f32")


def test_1m_bodies_newton_third_law_and_shard_consistency(gpu):
    """Size-independent properties at the benchmark size: sum_i m_i a_i = 0 for pairwise forces, and a
    target shard computes the same accelerations as the full pass."""
    b = physics.plummer(1 << 20, seed=3)
    with Engine(mode="direct", precision="f32") as e:
        e.upload(b.pos, b.vel, b.mass)
        acc, _ = e.accel()
    net = (b.mass[:, None] * acc).sum(0)
    scale = (b.mass * np.linalg.norm(acc, axis=1)).sum()
    assert np.linalg.norm(net) <= 1e-5 * scale
    # oracle on a small sample of targets
    pos, mass = f32(b.pos), f32(b.mass)
    t0, t1 = 500000, 500128
    check(acc[t0:t1], oracle_acc(mass, pos, t0=t0, t1=t1), "f32")


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 1023, 4097, 70001])
@pytest.mark.parametrize("precision", ["f32", "mixed"])
def test_equal_mass_scenes_take_the_massless_pair_loop(gpu, n, precision, monkeypatch):
    """When every body has the same mass the fp32 pair loop leaves the mass out (w = rinv^3, the mass folded into
    G) and pads its last tile with sources 1e18 m away instead of massless ones.  Same tolerances as the general
    kernel against the oracle, and the two kernels agree with each other to fp32 rounding of the sums; sizes cover
    partial tiles and the split source ranges of small scenes.  A mass that is not representable in fp32 is
    compared as the fp32 value the kernels hold."""
    rng = np.random.default_rng(n)
    pos = f32(rng.uniform(-1e4, 1e4, (n, 3))); vel = np.zeros((n, 3))
    mass = np.full(n, 1e9 / 1048576 * 1.000001)
    acc = engine_acc(pos, vel, mass, precision)
    if n == 1:
        assert np.all(acc == 0)
        return
    ref = oracle_acc(f32(mass), pos)
    check(acc, ref, precision)
    monkeypatch.setenv("NBODY_NO_EQUAL_MASS", "1")          # read at upload: the general kernel on the same scene
    general = engine_acc(pos, vel, mass, precision)
    monkeypatch.delenv("NBODY_NO_EQUAL_MASS")
    er = relerr(acc, general)
    assert np.median(er) <= 5e-7 and np.quantile(er, 0.999) <= 1e-4, (np.median(er), er.max())


def test_equal_mass_steps_match_general_kernel_steps(gpu, monkeypatch):
    """Ten fused steps of an equal-mass cloud with either pair loop: positions agree to fp32 rounding."""
    b = physics.plummer(5000, seed=3)
    out = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("NBODY_NO_EQUAL_MASS", "1")
        with Engine(mode="direct", precision="mixed", eps=1.0) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(10)
            out.append(e.download()[0])
    monkeypatch.delenv("NBODY_NO_EQUAL_MASS")
    assert np.abs(out[0] - out[1]).max() <= 1e-6 * np.abs(out[1]).max()
