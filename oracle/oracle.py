"""ctypes binding of oracle/liboracle.so (the C restatement in nbody_oracle.c).

Test infrastructure only.  PARITY UNPINNED BY THE REFERENCE (it has no tests or fixtures and cannot be
built here); see the header of nbody_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

G = 6.67408e-11          # physics.go:132
DT = 3600.0              # main.go:28
THETA = 1.0              # main.go:73
BOUND_WIDTH = 65536.0    # main.go:74-76
GROUPS = 4               # main.go:134


class Params(C.Structure):
    _fields_ = [("mode", C.c_int), ("com_mode", C.c_int), ("drop_tail", C.c_int), ("groups", C.c_int),
                ("max_depth", C.c_int), ("theta", C.c_double), ("dt", C.c_double),
                ("bound_center", C.c_double * 3), ("bound_width", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("interactions", C.c_long), ("nodes", C.c_long), ("root_mass", C.c_double),
                ("root_com", C.c_double * 3), ("build_seconds", C.c_double), ("walk_seconds", C.c_double),
                ("update_seconds", C.c_double), ("overflow", C.c_int), ("merges", C.c_long), ("queued", C.c_long)]


class NodeInfo(C.Structure):
    _fields_ = [("depth", C.c_int), ("kind", C.c_int), ("particle", C.c_long), ("nchild", C.c_int),
                ("mass", C.c_double), ("com", C.c_double * 3), ("center", C.c_double * 3), ("width", C.c_double)]


def build(force=False):
    """Compile liboracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        up = C.POINTER(C.c_ubyte)
        _lib.orc_forces.argtypes = [C.c_long, dp, dp, up, C.POINTER(Params), dp, C.POINTER(Stats)]
        _lib.orc_forces_group.argtypes = [C.c_long, dp, dp, up, C.POINTER(Params), C.c_long, dp, C.POINTER(Stats)]
        _lib.orc_step.argtypes = [C.c_long, dp, dp, dp, up, C.POINTER(Params), C.c_int, C.POINTER(Stats)]
        lp = C.POINTER(C.c_long)
        _lib.orc_step2.argtypes = [C.c_long, dp, dp, dp, dp, up, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(Stats)]
        _lib.orc_collision_queue.argtypes = [C.c_long, dp, dp, dp, up, C.POINTER(Params), lp, C.c_long]
        _lib.orc_collision_queue.restype = C.c_long
        _lib.orc_tree_dump.argtypes = [C.c_long, dp, dp, up, C.POINTER(Params), C.POINTER(NodeInfo), C.c_long]
        _lib.orc_tree_dump.restype = C.c_long
        _lib.orc_energy.argtypes = [C.c_long, dp, dp, dp, up, C.c_double, dp, dp, dp]
        _lib.orc_direct_rows.argtypes = [C.c_long, dp, dp, up, C.c_double, C.c_long, C.c_long, C.c_int, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte)) if a is not None else None


def make_params(mode="tree", com_mode="ref", drop_tail=True, groups=GROUPS, max_depth=0, theta=THETA, dt=DT,
                bound_center=(0.0, 0.0, 0.0), bound_width=BOUND_WIDTH):
    p = Params()
    p.mode = {"direct": 0, "tree": 1}[mode]
    p.com_mode = {"ref": 0, "true": 1}[com_mode]
    p.drop_tail = int(bool(drop_tail))
    p.groups = int(groups)
    p.max_depth = int(max_depth)
    p.theta = float(theta)
    p.dt = float(dt)
    p.bound_center[:] = [float(c) for c in bound_center]
    p.bound_width = float(bound_width)
    return p


def _prep(mass, pos, vel=None, alive=None):
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    n = mass.shape[0]
    assert pos.shape[0] == n
    if vel is not None:
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        assert vel.shape[0] == n
    alive = np.ones(n, dtype=np.uint8) if alive is None else np.ascontiguousarray(alive, dtype=np.uint8).copy()
    return n, mass, pos, vel, alive


def forces(mass, pos, alive=None, **kw):
    """Forces (N, not accelerations) on every body.  Returns (force[n,3], alive[n], Stats)."""
    n, mass, pos, _, alive = _prep(mass, pos, None, alive)
    p = make_params(**kw)
    out = np.zeros((n, 3), dtype=np.float64)
    st = Stats()
    lib().orc_forces(n, _dp(mass), _dp(pos), _up(alive), C.byref(p), _dp(out), C.byref(st))
    return out, alive, st


def forces_group(mass, pos, group=128, alive=None, **kw):
    """Forces with the opening test decided per `group` bodies that follow each other in Morton order (the
    engine's optional per-wave walk; not a reference behaviour).  Returns (force[n,3], alive[n], Stats)."""
    n, mass, pos, _, alive = _prep(mass, pos, None, alive)
    kw.setdefault("com_mode", "true"); kw.setdefault("drop_tail", False); kw.setdefault("max_depth", 21)
    p = make_params(mode="tree", **kw)
    out = np.zeros((n, 3), dtype=np.float64)
    st = Stats()
    lib().orc_forces_group(n, _dp(mass), _dp(pos), _up(alive), C.byref(p), int(group), _dp(out), C.byref(st))
    return out, alive, st


def step(mass, pos, vel, nsteps=1, alive=None, **kw):
    """Advance nsteps iterations.  Returns (pos, vel, alive, Stats); inputs are not modified."""
    n, mass, pos, vel, alive = _prep(mass, pos, vel, alive)
    pos = pos.copy()
    vel = vel.copy()
    p = make_params(**kw)
    st = Stats()
    lib().orc_step(n, _dp(mass), _dp(pos), _dp(vel), _up(alive), C.byref(p), int(nsteps), C.byref(st))
    return pos, vel, alive, st


def step_collide(mass, radius, pos, vel, nsteps=1, alive=None, **kw):
    """Advance nsteps iterations with collisions on (main.go:122-126 direct, :151-167 tree).
    Returns (mass, radius, pos, vel, alive, Stats); inputs are not modified."""
    n, mass, pos, vel, alive = _prep(mass, pos, vel, alive)
    mass, pos, vel = mass.copy(), pos.copy(), vel.copy()
    radius = np.full(n, 2.0) if radius is None else np.ascontiguousarray(radius, dtype=np.float64).copy()
    p = make_params(**kw)
    st = Stats()
    lib().orc_step2(n, _dp(mass), _dp(radius), _dp(pos), _dp(vel), _up(alive), C.byref(p), 1, int(nsteps), C.byref(st))
    return mass, radius, pos, vel, alive, st


def collision_queue(mass, radius, pos, alive=None, **kw):
    """The (searcher, resident) pairs checkCollision queues for this state, in queue order (main.go:153-159)."""
    n, mass, pos, _, alive = _prep(mass, pos, None, alive)
    radius = np.full(n, 2.0) if radius is None else np.ascontiguousarray(radius, dtype=np.float64)
    p = make_params(mode="tree", **kw)
    cap = 64
    while True:
        buf = np.zeros((cap, 2), dtype=np.int64)
        a = alive.copy()
        k = lib().orc_collision_queue(n, _dp(mass), _dp(radius), _dp(pos), _up(a), C.byref(p),
                                      buf.ctypes.data_as(C.POINTER(C.c_long)), cap)
        if k <= cap:
            return buf[:k].copy(), a
        cap = k


def tree_dump(mass, pos, alive=None, **kw):
    """Pre-order dump of the octree built from the bodies; returns a numpy structured array."""
    n, mass, pos, _, alive = _prep(mass, pos, None, alive)
    p = make_params(mode="tree", **kw)
    cap = 8 * n + 64
    while True:
        buf = (NodeInfo * cap)()
        k = lib().orc_tree_dump(n, _dp(mass), _dp(pos), _up(alive), C.byref(p), buf, cap)
        if k <= cap:
            break
        cap = k
    arr = np.ctypeslib.as_array(buf)[:k] if k else np.zeros(0, dtype=np.dtype(NodeInfo))
    return np.array(arr, copy=True), alive


def energy(mass, pos, vel, alive=None, eps=0.0):
    n, mass, pos, vel, alive = _prep(mass, pos, vel, alive)
    ke, pe = C.c_double(), C.c_double()
    mom = np.zeros(3)
    lib().orc_energy(n, _dp(mass), _dp(pos), _dp(vel), _up(alive), float(eps), C.byref(ke), C.byref(pe), _dp(mom))
    return ke.value, pe.value, mom


def direct_rows(mass, pos, t0=0, t1=None, alive=None, eps=0.0, nthreads=8):
    """fp64 direct-sum forces on targets [t0, t1) from all bodies (row order, threaded over targets)."""
    n, mass, pos, _, alive_arr = _prep(mass, pos, None, alive)
    t1 = n if t1 is None else t1
    out = np.zeros((t1 - t0, 3), dtype=np.float64)
    lib().orc_direct_rows(n, _dp(mass), _dp(pos), _up(alive_arr) if alive is not None else None, float(eps),
                          int(t0), int(t1), int(nthreads), _dp(out))
    return out
