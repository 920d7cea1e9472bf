"""Known answers derived by hand from the reference source, checked on both CPU restatements.

Each test names the reference lines it pins.  These are the only reference-anchored pins available: the
reference has no tests of its own.
"""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_py as P

G = 6.67408e-11


def test_constants_match_reference():
    assert O.G == G == P.G                      # physics.go:132
    assert O.DT == 3600.0                       # main.go:28
    assert O.THETA == 1.0                       # main.go:73
    assert O.BOUND_WIDTH == float(0x10000)      # main.go:76 (0x1p16)
    assert O.GROUPS == 4                        # main.go:134


def test_two_body_force_equal_and_opposite():
    """physics.go:135-145: F = G m1 m2 / r^2 along the unit vector."""
    mass = np.array([3.0e6, 5.0e4])
    pos = np.array([[0.0, 0.0, 0.0], [3.0, 4.0, 12.0]])      # r = 13
    f, _, _ = O.forces(mass, pos, mode="direct")
    mag = G * 3.0e6 * 5.0e4 / 169.0
    want = mag * np.array([3.0, 4.0, 12.0]) / 13.0
    assert np.allclose(f[0], want, rtol=1e-15, atol=0)
    assert np.array_equal(f[0], -f[1])
    a, b = P.Body(0, 3.0e6, 1, 0, 0, 0), P.Body(1, 5.0e4, 1, 3, 4, 12)
    P.gravity(P.dist(a, b), a, b)
    assert [a.fx, a.fy, a.fz] == list(f[0])


def test_update_is_kick_then_drift():
    """physics.go:101-115: v += f/m*dt first, then x += v*dt with the NEW v; f is cleared."""
    b = P.Body(0, 2.0, 1, 1.0, 0.0, 0.0, 0.5, 0.0, 0.0)
    b.fx = 4.0
    b.update(10.0)
    assert b.Vx == 0.5 + 4.0 / 2.0 * 10.0 == 20.5
    assert b.X == 1.0 + 20.5 * 10.0
    assert (b.fx, b.fy, b.fz) == (0.0, 0.0, 0.0)
    # the C restatement: two bodies, one step
    mass = np.array([2.0, 1e9]); pos = np.array([[1.0, 0, 0], [101.0, 0, 0]]); vel = np.array([[0.5, 0, 0], [0, 0, 0]])
    p, v, _, _ = O.step(mass, pos, vel, nsteps=1, mode="direct", dt=10.0)
    fx = G * 2.0 * 1e9 / 1e4
    v0 = 0.5 + fx / 2.0 * 10.0
    assert v[0, 0] == v0 and p[0, 0] == 1.0 + v0 * 10.0


def test_octant_bits_midpoint_goes_high_and_negative_zero_goes_low():
    """tree.go:128-132: the bit is the complement of the sign bit of p - mid."""
    assert P.octant_bits((0, 0, 0), (0.0, 0.0, 0.0)) == 0b111
    assert P.octant_bits((0, 0, 0), (-0.0, 1.0, -1.0)) == 0b010
    assert P.octant_bits((1.0, 1.0, 1.0), (1.0, 0.5, 2.0)) == 0b101       # bit0 = X, bit1 = Y, bit2 = Z
    assert P.octant_bits((0, 0, 0), (math.nan, 0.0, 0.0)) in (0b110, 0b111)  # NaN sign is unspecified


def test_contains_is_half_open():
    """tree.go:65-72: lo <= p < hi on every axis."""
    b = P.Bound((0.0, 0.0, 0.0), (8.0, 8.0, 8.0))
    assert b.contains((-4.0, -4.0, -4.0))
    assert not b.contains((4.0, 0.0, 0.0))
    assert b.contains((math.nextafter(4.0, 0.0), 0.0, 0.0))
    assert not b.contains((math.nan, 0.0, 0.0))


def test_child_cell_geometry():
    """tree.go:101-110: child centre = parent centre +- width/4 per axis, child width = width/2."""
    parent = P.Bound((0.0, 0.0, 0.0), (65536.0,) * 3)
    c = P.octant_bound(parent, 0b101)
    assert c.center == (16384.0, -16384.0, 16384.0) and c.width == (32768.0,) * 3
    c = P.octant_bound(parent, 0b000)
    assert c.center == (-16384.0,) * 3


def test_out_of_bounds_and_nan_bodies_are_removed_for_good():
    """tree.go:148-150, 290-292: a body outside the bound (or NaN) is dropped by maketree and gets no force."""
    mass = np.array([1e6, 1e6, 1e6, 1e6, 1e6])
    pos = np.array([[0.0, 0, 0], [100.0, 0, 0], [32768.0, 0, 0], [-32768.0, 0, 0], [math.nan, 0, 0]])
    f, alive, _ = O.forces(mass, pos, mode="tree", com_mode="ref", theta=1.0, drop_tail=False)
    assert list(alive) == [1, 1, 0, 1, 0]          # +32768 is outside (upper bound exclusive), -32768 inside
    assert np.all(f[2] == 0) and np.all(f[4] == 0)
    assert f[0, 0] != 0


def test_opening_test_uses_full_width_and_distance_to_centre_of_mass():
    """tree.go:212-217: open iff bounds.max()/r > theta, r measured to the cell's centre of mass."""
    # bodies 1 and 2 share the depth-1 cell of edge 32768 centred (16384,16384,16384); body 0 probes it.
    m = np.array([1.0, 1e6, 1e6])
    theta = 1.0
    for com_mode in ("true", "ref"):
        for x0 in (-16000.0, -20000.0, 16384.0 - 32768.0 - 116.0 + 1e-6, 16384.0 - 32768.0 - 116.0 - 1e-6):
            pos = np.array([[x0, 16384.0, 16384.0], [16000.0, 16384.0, 16384.0], [17000.0, 16384.0, 16384.0]])
            if com_mode == "true":
                cell_com = (pos[1] + pos[2]) / 2
                root_com = (m[:, None] * pos).sum(0) / m.sum()
            else:   # the reference's running update, tree.go:189-193
                cell_com = pos[1] + pos[2] * (m[2] / (m[1] + m[2]))
                root_com = pos[0] + pos[1] * (m[1] / (m[0] + m[1])) + pos[2] * (m[2] / m.sum())
            r_root = np.linalg.norm(root_com - pos[0])
            r_cell = np.linalg.norm(cell_com - pos[0])
            if 65536.0 / r_root > theta:
                mine = 2 if 32768.0 / r_cell > theta else 1
            else:
                mine = 1
            _, _, st = O.forces(m, pos, mode="tree", com_mode=com_mode, theta=theta, drop_tail=False)
            # bodies 1 and 2 are deep inside the root and their own cell: they open both and see 2 leaves each
            assert st.interactions == mine + 4, (com_mode, x0)
    # the two probes straddling r = 32768 in true-COM mode really differ
    pos = np.array([[0.0, 16384.0, 16384.0], [16000.0, 16384.0, 16384.0], [17000.0, 16384.0, 16384.0]])
    counts = []
    for dx in (1e-6, -1e-6):
        pos[0, 0] = 16500.0 - 32768.0 + dx
        counts.append(O.forces(m, pos, mode="tree", com_mode="true", theta=theta, drop_tail=False)[2].interactions)
    assert counts == [6, 5]


def test_reference_centre_of_mass_update_formula():
    """tree.go:156-160, 189-193: com = p1 + sum_{k>=2} p_k m_k / M_k (not a mass-weighted mean)."""
    mass = np.array([2.0, 6.0, 8.0])
    pos = np.array([[100.0, 10.0, 1.0], [300.0, 20.0, 2.0], [-500.0, 40.0, 3.0]])
    _, _, st = O.forces(mass, pos, mode="tree", com_mode="ref", drop_tail=False)
    want = pos[0] + pos[1] * (6.0 / 8.0) + pos[2] * (8.0 / 16.0)
    assert list(st.root_com) == list(want) and st.root_mass == 16.0
    _, _, st = O.forces(mass, pos, mode="tree", com_mode="true", drop_tail=False)
    true = (mass[:, None] * pos).sum(0) / mass.sum()
    assert np.allclose(st.root_com, true, rtol=1e-15)


def test_internal_cell_containing_the_body_can_be_accepted_with_its_own_mass():
    """tree.go:203-208: only the body's own leaf is skipped; a cell holding the body is tested like any other."""
    mass = np.array([1e3, 1e3])
    pos = np.array([[1.0, 1.0, 1.0], [1.5, 1.0, 1.0]])
    # with a huge theta nothing is opened: each body takes the root cell (both masses, itself included)
    f, _, st = O.forces(mass, pos, mode="tree", com_mode="true", theta=1e9, drop_tail=False)
    assert st.interactions == 2
    com = np.array([1.25, 1.0, 1.0])
    r = 0.25
    assert np.allclose(f[0], G * 2e3 * 1e3 / r**2 * (com - pos[0]) / r, rtol=1e-14)


def test_tree_walk_leaves_out_the_last_n_mod_4_bodies():
    """main.go:136-147: groupsize = len/4; the slices cover only 4*groupsize bodies."""
    rng = np.random.default_rng(3)
    n = 10
    mass = np.full(n, 1e5); pos = rng.uniform(-1000, 1000, (n, 3))
    f, _, _ = O.forces(mass, pos, mode="tree", com_mode="ref", drop_tail=True)
    assert np.all(f[8:] == 0) and np.all(np.abs(f[:8]).sum(1) > 0)
    f, _, _ = O.forces(mass, pos, mode="tree", com_mode="ref", drop_tail=False)
    assert np.all(np.abs(f).sum(1) > 0)


def test_frames_count_runs_one_extra_step():
    """main.go:30,80: frames = int(years*365*24) and the loop is `frame <= frames`, i.e. frames+1 steps."""
    years = 1.0
    frames = int(years * 365 * 24)
    assert frames == 8760 and len(range(0, frames + 1)) == 8761


def test_solar_system_earth_returns_after_one_year_of_direct_steps():
    """main.go:251-305 (SI units): after 8766 one-hour direct steps Earth is back near its start."""
    mass = np.array([1.9885e30, 3.3011e23, 4.8675e24, 5.97237e24, 6.4171e23, 1.8982e27, 5.6834e26, 8.681e25, 1.02413e26])
    x = np.array([0.0, (69816900e3 + 46001200e3) / 2, -(108939000e3 + 107477000e3) / 2, (152100000e3 + 147095000e3) / 2,
                  -(249200000e3 + 206700000e3) / 2, (816.2e9 + 740.52e9) / 2, -(1514.5e9 + 1352.55e9) / 2,
                  (3.008e12 + 2.742e12) / 2, -(4.54e12 + 4.46e12) / 2])
    vy = np.array([0.0, 47.362e3, -35.02e3, 29.78e3, -24.007e3, 13.07e3, -9.68e3, 6.8e3, -5.43e3])
    pos = np.zeros((9, 3)); pos[:, 0] = x
    vel = np.zeros((9, 3)); vel[:, 1] = vy
    p, v, _, _ = O.step(mass, pos, vel, nsteps=8766, mode="direct")
    earth0, earth1 = pos[3] - pos[0], p[3] - p[0]
    au = np.linalg.norm(earth0)
    assert np.linalg.norm(earth1 - earth0) < 0.05 * au
    assert abs(np.linalg.norm(earth1) - au) < 0.02 * au


def test_coincident_bodies_and_the_depth_cap_extension():
    """Identical coordinates make push (tree.go:170-189) split again and again until the child centre stops
    moving in fp64 and contains() (tree.go:148) rejects the body: the recursion ends after about 60 levels
    and the later body silently drops out of the tree (push still returns true, tree.go:196).  The
    restatement's max_depth extension (not in the reference) instead keeps every body of a depth-21 cell as
    a one-body leaf, which is the structure the GPU build produces; tree.go:234-238 applied to two distinct
    leaves at distance 0 then yields NaN for that pair."""
    mass = np.array([1e3, 1e3, 1e3])
    pos = np.array([[10.0, 10.0, 10.0], [10.0, 10.0, 10.0], [500.0, 10.0, 10.0]])
    want = G * 1e3 * 2e3 / 490.0**2
    f, alive, st = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, max_depth=0)
    assert st.overflow == 0 and alive.all() and 40 < st.nodes < 200
    assert np.isfinite(f).all() and st.interactions == 3          # body 1 is in no leaf: nobody sees it
    assert abs(-f[2, 0] - want) < 1e-12 * want                    # ...but its mass still sits in the cell sums
    f, alive, st = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, max_depth=21)
    assert st.overflow == 0 and alive.all() and st.nodes < 40
    assert abs(-f[2, 0] - want) < 1e-12 * want
    assert np.isnan(f[0]).all() and np.isnan(f[1]).all()


def test_group_rule_with_groups_of_one_is_the_per_body_walk():
    """oracle.forces_group restates the engine's optional per-wave opening rule (not a reference behaviour) over the
    pointer octree.  With one body per group it must be the reference's own walk, bit for bit; with larger groups
    every body gets at least as many force evaluations."""
    rng = np.random.default_rng(3)
    n = 700
    pos = rng.normal(0, 4e3, (n, 3)); mass = np.abs(rng.normal(1e3, 500, n)) + 1
    f1, _, s1 = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, max_depth=21)
    g1, _, t1 = O.forces_group(mass, pos, group=1, theta=0.5)
    assert np.array_equal(f1, g1) and s1.interactions == t1.interactions
    g128, _, t128 = O.forces_group(mass, pos, group=128, theta=0.5)
    assert t128.interactions > t1.interactions
    exact = O.direct_rows(mass, pos)
    err = lambda f: np.median(np.linalg.norm(f - exact, axis=1) / np.linalg.norm(exact, axis=1))
    assert err(g128) <= err(g1)
