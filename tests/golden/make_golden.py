"""Generates tests/golden/golden_small.npz: seeded inputs and the outputs of the CPU restatements.

The reference (Go) cannot be built or run in this image and holds no fixtures of its own, so these vectors
come from this repository's two independent restatements of the reference's arithmetic (oracle/nbody_oracle.c
and oracle/oracle_py.py), which must agree bit for bit before anything is written.  PARITY UNPINNED BY THE
REFERENCE: the vectors pin the restatements against regressions and pin the GPU against them, not against a
run of the Go program.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nbody_amd import physics  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import oracle_py as P  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")


def py_bodies(mass, pos, vel):
    return [P.Body(i, float(mass[i]), 2.0, *map(float, pos[i]), *map(float, vel[i])) for i in range(len(mass))]


def py_forces(mass, pos, vel, tree, **kw):
    bs = py_bodies(mass, pos, vel)
    if tree:
        _, cnt = P.tree_pass(bs, **kw)
    else:
        P.direct_pass(bs)
        cnt = len(bs) * (len(bs) - 1)
    f = np.array([[b.fx, b.fy, b.fz] if b is not None else [0.0, 0.0, 0.0] for b in bs])
    return f, cnt


def py_sim(mass, pos, vel, nsteps, tree, **kw):
    bs = P.simulate(py_bodies(mass, pos, vel), nsteps, tree=tree, **kw)
    p = np.array([[b.X, b.Y, b.Z] for b in bs])
    v = np.array([[b.Vx, b.Vy, b.Vz] for b in bs])
    return p, v


def main():
    out = {}
    cases = [("galaxy", 0), ("galaxy", 1), ("galaxy", 6), ("uniform", 64), ("galaxy", 254), ("uniform", 1024)]
    for kind, n in cases:
        b = physics.makebodies(n, seed=7 + n) if kind == "galaxy" else physics.uniform_cube(n, seed=7 + n)
        if kind == "uniform":
            rng = np.random.default_rng(100 + n)
            b.mass[:] = np.abs(rng.normal(1e3, 500, n)) + 1.0
            b.vel[:] = rng.normal(0, 1e-3, (n, 3))
        N = len(b)
        name = f"{kind}{N}"
        mass, pos, vel = b.mass, b.pos, b.vel
        out[f"{name}/mass"], out[f"{name}/pos"], out[f"{name}/vel"] = mass, pos, vel
        check_py = N <= 256

        f, _, st = O.forces(mass, pos, mode="direct")
        out[f"{name}/direct_force"] = f
        if check_py:
            fp, _ = py_forces(mass, pos, vel, tree=False)
            assert np.array_equal(f, fp), name
        for com in ("ref", "true"):
            for theta in (1.0, 0.5):
                for drop in (True, False):
                    if drop and com == "true":
                        continue
                    f, alive, st = O.forces(mass, pos, mode="tree", com_mode=com, theta=theta, drop_tail=drop)
                    key = f"{name}/tree_{com}_th{theta}_{'drop' if drop else 'all'}"
                    out[key + "_force"] = f
                    out[key + "_meta"] = np.array([st.interactions, st.nodes, st.root_mass, *st.root_com])
                    if check_py or (N <= 1024 and com == "true" and theta == 0.5):
                        fp, cnt = py_forces(mass, pos, vel, tree=True, theta=theta, true_com=(com == "true"), drop_tail=drop)
                        assert np.array_equal(f, fp), key
                        assert cnt == st.interactions, key
        # ten steps of the three step flavours the tests use
        for tag, kw, pykw in (("direct", dict(mode="direct"), dict(tree=False)),
                              ("tree_ref", dict(mode="tree", com_mode="ref", theta=1.0, drop_tail=True),
                               dict(tree=True, theta=1.0, true_com=False, drop_tail=True)),
                              ("tree_true", dict(mode="tree", com_mode="true", theta=0.5, drop_tail=False),
                               dict(tree=True, theta=0.5, true_com=True, drop_tail=False))):
            p, v, alive, _ = O.step(mass, pos, vel, nsteps=10, **kw)
            out[f"{name}/step10_{tag}_pos"], out[f"{name}/step10_{tag}_vel"] = p, v
            if N <= 64:
                pp, pv = py_sim(mass, pos, vel, 10, **pykw)
                assert np.array_equal(p, pp) and np.array_equal(v, pv), (name, tag)
        print(f"{name}: ok")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


OUT_COLLIDE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_collide.npz")


def main_collide():
    """golden_collide.npz: crowded scenes stepped with collisions on (tree: tree.go:245-281 + main.go:151-167;
    direct: main.go:107-128), outputs of the C restatement, checked against the Python one before writing
    (radii to a few ulp: the two use different cube roots)."""
    out = {}
    for name, n, half, seed in (("crowd120", 120, 14.0, 31), ("crowd400", 400, 24.0, 32)):
        rng = np.random.default_rng(seed)
        pos = rng.uniform(-half, half, (n, 3)); vel = rng.normal(0, 1e-3, (n, 3))
        mass = np.abs(rng.normal(1e3, 500, n)) + 1; rad = rng.uniform(0.5, 3.0, n)
        out[f"{name}/mass"], out[f"{name}/pos"], out[f"{name}/vel"], out[f"{name}/radius"] = mass, pos, vel, rad
        for tag, kw, pykw in (("direct", dict(mode="direct"), dict(tree=False)),
                              ("tree_ref", dict(mode="tree", com_mode="ref", theta=1.0, drop_tail=True),
                               dict(tree=True, theta=1.0, true_com=False, drop_tail=True)),
                              ("tree_true", dict(mode="tree", com_mode="true", theta=0.7, drop_tail=False, max_depth=21),
                               dict(tree=True, theta=0.7, true_com=True, drop_tail=False))):
            m, r, p, v, alive, st = O.step_collide(mass, rad, pos, vel, nsteps=4, dt=100.0, **kw)
            assert st.merges > 0, (name, tag)
            bodies = [P.Body(i, float(mass[i]), float(rad[i]), *map(float, pos[i]), *map(float, vel[i])) for i in range(n)]
            P.simulate(bodies, 4, dt=100.0, collide=True, **pykw)
            assert [bb is not None for bb in bodies] == [bool(a) for a in alive], (name, tag)
            for i, bb in enumerate(bodies):
                if bb is not None:
                    assert bb.Mass == m[i] and abs(bb.Radius - r[i]) <= 1e-15 * r[i], (name, tag, i)
                    assert max(abs(bb.X - p[i, 0]), abs(bb.Y - p[i, 1]), abs(bb.Z - p[i, 2])) <= 1e-13 * half, (name, tag, i)
            key = f"{name}/step4_{tag}"
            out[key + "_mass"], out[key + "_radius"], out[key + "_pos"], out[key + "_vel"] = m, r, p, v
            out[key + "_alive"] = alive
            out[key + "_meta"] = np.array([st.merges, st.queued])
        print(f"{name}: ok")
    np.savez_compressed(OUT_COLLIDE, **out)
    print("wrote", OUT_COLLIDE, os.path.getsize(OUT_COLLIDE), "bytes")


if __name__ == "__main__":
    if "collide" in sys.argv[1:]:
        main_collide()
    else:
        main()
