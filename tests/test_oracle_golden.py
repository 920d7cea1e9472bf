"""The CPU restatements against the committed golden vectors (tests/golden/make_golden.py).

The reference ships no tests or fixtures (its .gitignore even hides *_test.go), so these vectors are the
restatements' own outputs: they pin both restatements against regressions and against each other, bit for bit.
"""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_py as P
from tests.util import golden_cases, load_golden

G = load_golden()
CASES = golden_cases(G)
TREE_VARIANTS = [("ref", 1.0, True), ("ref", 1.0, False), ("ref", 0.5, True), ("ref", 0.5, False),
                 ("true", 1.0, False), ("true", 0.5, False)]


def _inputs(name):
    return G[f"{name}/mass"], G[f"{name}/pos"], G[f"{name}/vel"]


@pytest.mark.parametrize("name", CASES)
def test_c_direct_matches_golden(name):
    mass, pos, _ = _inputs(name)
    f, alive, st = O.forces(mass, pos, mode="direct")
    assert np.array_equal(f, G[f"{name}/direct_force"])
    assert alive.all()
    assert st.interactions == len(mass) * (len(mass) - 1)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("com,theta,drop", TREE_VARIANTS)
def test_c_tree_matches_golden(name, com, theta, drop):
    mass, pos, _ = _inputs(name)
    key = f"{name}/tree_{com}_th{theta}_{'drop' if drop else 'all'}"
    f, alive, st = O.forces(mass, pos, mode="tree", com_mode=com, theta=theta, drop_tail=drop)
    assert np.array_equal(f, G[key + "_force"])
    meta = G[key + "_meta"]
    assert st.interactions == int(meta[0]) and st.nodes == int(meta[1])
    assert st.root_mass == meta[2] and list(st.root_com) == list(meta[3:6])


@pytest.mark.parametrize("name", [c for c in CASES if len(G[f"{c}/mass"]) <= 64])
def test_python_restatement_matches_golden(name):
    mass, pos, vel = _inputs(name)
    n = len(mass)
    bs = [P.Body(i, float(mass[i]), 2.0, *map(float, pos[i])) for i in range(n)]
    P.direct_pass(bs)
    assert np.array_equal(np.array([[b.fx, b.fy, b.fz] for b in bs]), G[f"{name}/direct_force"])
    for com, theta, drop in TREE_VARIANTS:
        bs = [P.Body(i, float(mass[i]), 2.0, *map(float, pos[i])) for i in range(n)]
        root, cnt = P.tree_pass(bs, theta=theta, true_com=(com == "true"), drop_tail=drop)
        key = f"{name}/tree_{com}_th{theta}_{'drop' if drop else 'all'}"
        assert np.array_equal(np.array([[b.fx, b.fy, b.fz] for b in bs]), G[key + "_force"]), key
        meta = G[key + "_meta"]
        assert cnt == int(meta[0]) and P.count_nodes(root) == int(meta[1])
        assert root.totalMass == meta[2] and list(root.centerOfMass) == list(meta[3:6])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag,kw", [("direct", dict(mode="direct")),
                                    ("tree_ref", dict(mode="tree", com_mode="ref", theta=1.0, drop_tail=True)),
                                    ("tree_true", dict(mode="tree", com_mode="true", theta=0.5, drop_tail=False))])
def test_c_ten_steps_match_golden(name, tag, kw):
    mass, pos, vel = _inputs(name)
    p, v, alive, _ = O.step(mass, pos, vel, nsteps=10, **kw)
    assert np.array_equal(p, G[f"{name}/step10_{tag}_pos"])
    assert np.array_equal(v, G[f"{name}/step10_{tag}_vel"])


@pytest.mark.parametrize("name", [c for c in CASES if len(G[f"{c}/mass"]) <= 8])
def test_python_ten_steps_match_golden(name):
    mass, pos, vel = _inputs(name)
    bs = [P.Body(i, float(mass[i]), 2.0, *map(float, pos[i]), *map(float, vel[i])) for i in range(len(mass))]
    P.simulate(bs, 10, tree=True, theta=1.0, true_com=False, drop_tail=True)
    assert np.array_equal(np.array([[b.X, b.Y, b.Z] for b in bs]), G[f"{name}/step10_tree_ref_pos"])
    assert np.array_equal(np.array([[b.Vx, b.Vy, b.Vz] for b in bs]), G[f"{name}/step10_tree_ref_vel"])


def test_walk_thread_count_does_not_change_forces():
    mass, pos, _ = _inputs("uniform1024")
    f4, _, _ = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, groups=4)
    f1, _, _ = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, groups=1)
    f7, _, _ = O.forces(mass, pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, groups=7)
    assert np.array_equal(f4, f1) and np.array_equal(f4, f7)


def test_direct_rows_equals_pair_loop_to_rounding():
    mass, pos, _ = _inputs("galaxy256")
    fr = O.direct_rows(mass, pos)
    fp = G["galaxy256/direct_force"]
    assert np.abs(fr - fp).max() <= 1e-12 * np.abs(fp).max()
