"""Hand-derived known answers (the same ones tests/test_oracle_kat.py pins on the CPU restatements) on the GPU
engine, through the C ABI, plus run-time parameters the reference hard-codes (theta, dt, G, bound)."""
import math

import numpy as np
import pytest

from nbody_amd import Engine
from oracle import oracle as O
from tests.util import relerr

pytestmark = pytest.mark.gpu
G = 6.67408e-11


@pytest.mark.parametrize("mode", ["direct", "tree"])
@pytest.mark.parametrize("precision", ["f32", "mixed", "f64"])
def test_two_body_force_and_update_order(gpu, mode, precision):
    """physics.go:135-145 and 101-115: a = G m_other / r^2 along the unit vector; v += a*dt, then x += v_new*dt."""
    mass = np.array([3.0e6, 5.0e4])
    pos = np.array([[0.0, 0.0, 0.0], [3.0, 4.0, 12.0]])          # r = 13
    vel = np.array([[0.5, 0.0, 0.0], [0.0, -0.25, 0.0]])
    tol = 1e-14 if precision == "f64" else 1e-6   # fp32 pair arithmetic: a few ulp of 6e-8
    with Engine(mode=mode, precision=precision, dt=10.0) as e:
        e.upload(pos, vel, mass)
        acc, alive = e.accel()
        want0 = G * 5.0e4 / 169.0 * np.array([3.0, 4.0, 12.0]) / 13.0
        want1 = -G * 3.0e6 / 169.0 * np.array([3.0, 4.0, 12.0]) / 13.0
        assert np.abs(acc[0] - want0).max() <= tol * np.abs(want0).max()
        assert np.abs(acc[1] - want1).max() <= tol * np.abs(want1).max()
        e.step(1)
        p, v, m, ids = e.download()
    v_want = vel + np.stack([want0, want1]) * 10.0
    p_want = pos + v_want * 10.0
    assert np.abs(v - v_want).max() <= 10 * tol * np.abs(v_want).max()
    assert np.abs(p - p_want).max() <= 10 * tol * np.abs(p_want).max()


def test_cell_membership_is_half_open_and_midpoints_go_high(gpu):
    """tree.go:65-72 (lo <= p < hi) and tree.go:128-132 (a point on a cell midpoint belongs to the high octant):
    checked through the tree the engine builds.  Bodies sit exactly on cell boundaries of the 2^16 cube."""
    pos = np.array([[0.0, 0.0, 0.0],            # centre of the root: octant 7 (HHH)
                    [-1.0, -1.0, -1.0],         # octant 0
                    [16384.0, 16384.0, 16384.0],  # centre of octant 7: its own HHH child
                    [16383.0, 16383.0, 16383.0],  # octant 7, child 0
                    [-32768.0, 0.0, 0.0],       # lower bound is inside
                    [32768.0, 0.0, 0.0],        # upper bound is outside
                    [math.nextafter(32768.0, 0.0), 0.0, 0.0]])
    mass = np.full(len(pos), 1e5)
    with Engine(mode="tree", precision="f64", theta=0.5) as e:
        e.upload(pos, np.zeros_like(pos), mass)
        acc, alive = e.accel()
        nodes = e.tree_nodes()
    assert list(alive) == [1, 1, 1, 1, 1, 0, 1]
    f, oalive, st = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, max_depth=21)
    assert np.array_equal(alive, oalive)
    live = alive == 1
    assert relerr(acc[live], (f / mass[:, None])[live]).max() <= 1e-12
    # pre-order of the leaves = Morton order with octant bit0 = X, bit1 = Y, bit2 = Z (tree.go:28-39):
    # (-1,-1,-1) [octant 0], then x >= 0 / y,z < 0 ..., bodies 0, 3, 2 inside octant 7 in child order 0 < 7
    leaves = nodes[nodes[:, 4] == 0][:, :3]
    order = [list(p) for p in leaves]
    assert order.index([-1.0, -1.0, -1.0]) == 0
    assert order.index([0.0, 0.0, 0.0]) < order.index([16383.0, 16383.0, 16383.0]) < order.index([16384.0, 16384.0, 16384.0])


def test_negative_zero_is_treated_like_zero(gpu):
    """Stated deviation: octantBits (tree.go:128-132) sends -0.0 to the low octant, whose contains() then rejects it,
    so the reference silently drops the body from the mass sums; the engine treats -0.0 as +0.0."""
    pos = np.array([[-0.0, 5.0, 5.0], [100.0, 5.0, 5.0], [-100.0, 5.0, 5.0]])
    mass = np.array([1e6, 1e6, 1e6])
    with Engine(mode="tree", precision="f64", theta=0.5) as e:
        e.upload(pos, np.zeros_like(pos), mass)
        acc, alive = e.accel()
    assert alive.all()
    ref = O.direct_rows(mass, np.abs(pos) * np.sign(pos + 1e-300)) / mass[:, None]
    # the middle body feels two equal and opposite pulls: compare with the size of one of them, not with their sum
    assert np.abs(acc - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("theta", [0.0, 0.3, 0.5, 1.0, 2.0, 1e9])
def test_theta_is_a_runtime_parameter(gpu, theta):
    """main.go:73 hard-codes theta = 1.  theta = 0 opens every cell (exact pair sum, N*(N-1) evaluations); a huge
    theta accepts the root for every body (one evaluation each, the body's own mass included, tree.go:203-208)."""
    rng = np.random.default_rng(12)
    n = 400
    pos = rng.uniform(-2e4, 2e4, (n, 3)); mass = np.abs(rng.normal(1e4, 3e3, n)) + 1
    with Engine(mode="tree", precision="f64", theta=theta) as e:
        e.upload(pos, np.zeros_like(pos), mass)
        acc, alive = e.accel()
        st = e.stats()
    f, _, ost = O.forces(mass, pos, mode="tree", com_mode="true", theta=theta, drop_tail=False, max_depth=21)
    assert st.interactions == ost.interactions
    assert relerr(acc, f / mass[:, None]).max() <= 1e-9
    if theta == 0.0:
        assert st.interactions == n * (n - 1)
        assert relerr(acc, O.direct_rows(mass, pos) / mass[:, None]).max() <= 1e-12
    if theta == 1e9:
        assert st.interactions == n


def test_custom_bound_g_and_dt(gpu):
    """main.go:74-76 fixes the bound at centre 0, width 2^16 and physics.go:132 fixes G; both are run-time values
    here.  A shifted power-of-two bound must give the same cells as the oracle built with that bound."""
    rng = np.random.default_rng(3)
    n = 1000
    centre = (1000.0, -2000.0, 512.0)
    pos = rng.uniform(-500, 500, (n, 3)) + np.array(centre); mass = np.full(n, 1e3); vel = rng.normal(0, 1e-3, (n, 3))
    pos[0] = [centre[0] + 600.0, centre[1], centre[2]]       # outside a 1024-wide bound
    with Engine(mode="tree", precision="f64", theta=0.7, G=2.0 * G, dt=100.0, bound_center=centre, bound_width=1024.0) as e:
        e.upload(pos, vel, mass)
        acc, alive = e.accel()
        e.step(3)
        p, v, m, ids = e.download()
    f, oalive, _ = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.7, drop_tail=False, max_depth=21,
                            bound_center=centre, bound_width=1024.0)
    assert np.array_equal(alive, oalive) and alive[0] == 0 and alive[1:].all()
    assert relerr(acc[1:], 2.0 * (f / mass[:, None])[1:]).max() <= 1e-10
    assert len(p) == n - 1 and list(ids[:3]) == [1, 2, 3]


def test_all_bodies_are_walked_not_only_multiples_of_four(gpu):
    """main.go:136-147 leaves the last N mod 4 bodies without force; the engine walks everyone (stated deviation)."""
    rng = np.random.default_rng(5)
    n = 10
    pos = rng.uniform(-1000, 1000, (n, 3)); mass = np.full(n, 1e5)
    with Engine(mode="tree", precision="f64", theta=1.0) as e:
        e.upload(pos, np.zeros_like(pos), mass)
        acc, _ = e.accel()
    assert (np.abs(acc).sum(1) > 0).all()
    f_drop, _, _ = O.forces(mass, pos, mode="tree", com_mode="true", theta=1.0, drop_tail=True)
    assert np.all(f_drop[8:] == 0)
    assert relerr(acc[:8], (f_drop / mass[:, None])[:8]).max() <= 1e-12


def test_huge_masses_need_fp64(gpu):
    """main.go:251-305 (solarsystem, SI units): planetary masses and distances overflow the fp32 pair arithmetic;
    the fp64 engine reproduces the oracle's year-long direct integration."""
    mass = np.array([1.9885e30, 5.97237e24, 1.8982e27])
    pos = np.array([[0.0, 0, 0], [1.495975e11, 0, 0], [7.7836e11, 0, 0]])
    vel = np.array([[0.0, 0, 0], [0, 29.78e3, 0], [0, 13.07e3, 0]])
    with Engine(mode="direct", precision="f64") as e:
        e.upload(pos, vel, mass)
        e.step(2000)
        p, v, m, ids = e.download()
    pr, vr, _, _ = O.step(mass, pos, vel, nsteps=2000, mode="direct")
    assert np.abs(p - pr).max() <= 1e-9 * np.abs(pr).max()
