"""The collision restatements (oracle/nbody_oracle.c against oracle/oracle_py.py) and hand-derived answers for the
reference's collision rules: combine (physics.go:153-162), containsSphere / intersectSphere (tree.go:75-86),
checkCollision (tree.go:245-281), the queue and merge loop (main.go:151-167) and the direct loop (main.go:107-128).
The reference has no tests or fixtures for any of this: parity is unpinned, these pin the two restatements
to each other and to the arithmetic of the Go statements."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_py as P


def crowd(n, half, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-half, half, (n, 3)), rng.normal(0, 1e-3, (n, 3)), np.abs(rng.normal(1e3, 500, n)) + 1,
            rng.uniform(0.5, 3.0, n))


@pytest.mark.parametrize("mode", ["tree", "direct"])
@pytest.mark.parametrize("seed", [0, 1])
def test_c_and_python_restatements_agree(mode, seed):
    pos, vel, mass, rad = crowd(250, 30.0, seed)
    nsteps = 3
    m2, r2, p2, v2, alive, st = O.step_collide(mass, rad, pos, vel, nsteps=nsteps, mode=mode, com_mode="ref",
                                               drop_tail=True, dt=100.0)
    bodies = [P.Body(i, mass[i], rad[i], *pos[i], *vel[i]) for i in range(len(mass))]
    P.simulate(bodies, nsteps, tree=(mode == "tree"), dt=100.0, collide=True)
    assert st.merges > 0
    assert [b is not None for b in bodies] == [bool(a) for a in alive]
    for i, b in enumerate(bodies):
        if b is None:
            continue
        assert b.Mass == m2[i]
        assert abs(b.Radius - r2[i]) <= 1e-15 * r2[i]          # cbrt: libm against a Newton-polished power (a few ulp)
        assert (b.X, b.Y, b.Z) == pytest.approx(tuple(p2[i]), rel=1e-15, abs=0)
        assert (b.Vx, b.Vy, b.Vz) == pytest.approx(tuple(v2[i]), rel=1e-13, abs=1e-20)


def test_queue_order_is_body_order_then_child_order():
    """main.go:154-159 + tree.go:273-278: bodies search in slice order, children are visited 0..7."""
    pos, vel, mass, rad = crowd(250, 30.0, 4)
    q, alive = O.collision_queue(mass, rad, pos, com_mode="ref")
    bodies = [P.Body(i, mass[i], rad[i], *pos[i]) for i in range(len(mass))]
    root = P.maketree(bodies, P.Bound((0.0, 0.0, 0.0), (65536.0,) * 3))
    queue = []
    for i in range(len(bodies)):
        P.check_collision(root, bodies, i, queue)
    assert len(queue) > 10 and [tuple(x) for x in q] == queue
    assert all(q[k][0] <= q[k + 1][0] for k in range(len(q) - 1))


def test_combine_known_answer():
    """physics.go:154-158 with ma=3, mb=5, va=1, vb=-2: Mass becomes 8 first, so v = (8*1 + 5*-2) / (8+5)."""
    a, b = P.Body(0, 3.0, 1.0, Vx=1.0), P.Body(1, 5.0, 2.0, Vx=-2.0)
    a.fx, b.fx = 0.25, 0.5
    P.combine(a, b)
    assert a.Mass == 8.0 and a.Vx == (8.0 * 1.0 + 5.0 * -2.0) / 13.0 and a.fx == 0.75
    assert abs(a.Radius - 9.0 ** (1.0 / 3.0)) < 1e-15
    m, r, p, v, alive, st = O.step_collide([3.0, 5.0], [1.0, 2.0], [[0.25, 0.25, 0.25], [0.75, 0.25, 0.25]],
                                           [[1.0, 0, 0], [-2.0, 0, 0]], nsteps=1, mode="direct", dt=0.0)
    assert list(alive) == [1, 0] and m[0] == 8.0 and st.merges == 1
    assert abs(v[0, 0] - (-2.0 / 13.0)) < 1e-12           # gravity of one step with dt = 0 adds nothing


def test_sphere_tests_use_the_full_width():
    """tree.go:75-86: both compare |p - centre| with the FULL width -/+ r; intersectSphere passes on ANY axis."""
    b = P.Bound((0.0, 0.0, 0.0), (8.0, 8.0, 8.0))
    assert P.contains_sphere(b, (6.9, 0.0, 0.0), 1.0)          # 6.9 <= 8 - 1 although the cell ends at 4
    assert not P.contains_sphere(b, (7.1, 0.0, 0.0), 1.0)
    assert P.intersect_sphere(b, (100.0, 100.0, 9.0), 1.0)     # one axis within 8 + 1 is enough
    assert not P.intersect_sphere(b, (100.0, 100.0, 9.1), 1.0)


def test_neighbours_across_a_cell_boundary_are_missed():
    """tree.go:265-270: two touching bodies on either side of the root's x = 0 plane never meet; the direct loop
    (main.go:122-126) merges them."""
    mass, rad = [10.0, 10.0, 10.0], [1.0, 1.0, 1.0]
    pos = [[-0.25, 500.3, 500.3], [0.25, 500.3, 500.3], [9000.0, 9000.0, 9000.0]]
    q, _ = O.collision_queue(mass, rad, pos)
    assert len(q) == 0
    m, r, p, v, alive, st = O.step_collide(mass, rad, pos, np.zeros((3, 3)), nsteps=1, mode="direct", dt=1.0)
    assert list(alive) == [1, 0, 1] and st.merges == 1
    shifted = (np.array(pos) + [0.5, 0, 0]).tolist()
    q, _ = O.collision_queue(mass, rad, shifted)
    assert [tuple(x) for x in q] == [(0, 1), (1, 0)]


def test_merged_pairs_later_in_the_queue_are_dropped():
    """main.go:161-163: four mutually touching bodies queue 12 pairs; body 0 absorbs the other three."""
    base = np.array([2000.0, 2000.0, 2000.0]) + 0.01
    pos = base + np.array([[0.1, 0.1, 0.1], [0.6, 0.1, 0.1], [0.1, 0.6, 0.1], [0.6, 0.6, 0.6]])
    m, r, p, v, alive, st = O.step_collide([1.0, 2.0, 3.0, 4.0], [0.5] * 4, pos, np.zeros((4, 3)), nsteps=1, mode="tree", dt=1.0)
    assert st.queued == 12 and st.merges == 3 and list(alive) == [1, 0, 0, 0] and m[0] == 10.0
    assert abs(r[0] - math.pow(4 * 0.125, 1.0 / 3.0)) < 1e-15


def test_collisions_off_is_the_old_step():
    pos, vel, mass, rad = crowd(200, 30.0, 2)
    p1, v1, a1, _ = O.step(mass, pos, vel, nsteps=2, mode="tree", dt=50.0)
    far = np.full(len(mass), 1e-9)                              # radii too small to touch anything
    m2, r2, p2, v2, a2, st = O.step_collide(mass, far, pos, vel, nsteps=2, mode="tree", dt=50.0)
    assert st.merges == 0 and np.array_equal(p1, p2) and np.array_equal(v1, v2) and np.array_equal(a1, a2)


GOLDEN_COLLIDE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "golden_collide.npz")
STEP_KW = {"direct": dict(mode="direct"),
           "tree_ref": dict(mode="tree", com_mode="ref", theta=1.0, drop_tail=True),
           "tree_true": dict(mode="tree", com_mode="true", theta=0.7, drop_tail=False, max_depth=21)}


@pytest.mark.parametrize("name", ["crowd120", "crowd400"])
@pytest.mark.parametrize("tag", ["direct", "tree_ref", "tree_true"])
def test_collision_golden_vectors(name, tag):
    """tests/golden/golden_collide.npz (made by tests/golden/make_golden.py collide): four steps with merges."""
    g = np.load(GOLDEN_COLLIDE, allow_pickle=False)
    mass, pos, vel, rad = (g[f"{name}/{k}"] for k in ("mass", "pos", "vel", "radius"))
    m, r, p, v, alive, st = O.step_collide(mass, rad, pos, vel, nsteps=4, dt=100.0, **STEP_KW[tag])
    key = f"{name}/step4_{tag}"
    assert np.array_equal(alive, g[key + "_alive"]) and [st.merges, st.queued] == list(g[key + "_meta"])
    live = alive == 1
    assert np.array_equal(m[live], g[key + "_mass"][live]) and np.array_equal(p[live], g[key + "_pos"][live])
    assert np.array_equal(v[live], g[key + "_vel"][live]) and np.array_equal(r[live], g[key + "_radius"][live])
