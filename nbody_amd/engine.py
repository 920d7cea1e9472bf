"""ctypes binding of libnbody_engine.so (C ABI in include/nbody_engine.h).

There is no CPU fallback: loading fails loudly when the HIP library has not been built, and Engine()
fails when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MODE_DIRECT, MODE_TREE = 0, 1
PREC_F32, PREC_MIXED, PREC_F64 = 0, 1, 2
_MODES = {"direct": MODE_DIRECT, "tree": MODE_TREE}
_PRECS = {"f32": PREC_F32, "mixed": PREC_MIXED, "f64": PREC_F64}

# Every symbol include/nbody_engine.h declares.
EXPORTS = ("nbe_default_config", "nbe_create", "nbe_destroy", "nbe_last_error", "nbe_upload", "nbe_download",
           "nbe_accel", "nbe_step", "nbe_step_timed", "nbe_sync", "nbe_stats", "nbe_energy", "nbe_set_shard",
           "nbe_shard_prepare", "nbe_shard_advance", "nbe_shard_commit", "nbe_device_state", "nbe_set_stream",
           "nbe_time_force_kernel", "nbe_tree_nodes", "nbe_frame_stream_begin", "nbe_frame_stream_end",
           "nbe_set_radius", "nbe_get_radius", "nbe_shard_sort", "nbe_accel_direct_noalloc", "nbe_shard_advance_range")

FRAME_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float),
                       C.POINTER(C.c_uint64))


QUIRK_REF_COM, QUIRK_DROP_TAIL = 1, 2
WALK_PER_BODY, WALK_PER_WAVE = 0, 1


class Config(C.Structure):
    _fields_ = [("mode", C.c_int), ("precision", C.c_int), ("G", C.c_double), ("theta", C.c_double),
                ("dt", C.c_double), ("eps", C.c_double), ("bound_center", C.c_double * 3),
                ("bound_width", C.c_double), ("device", C.c_int), ("collisions", C.c_int),
                ("reference_quirks", C.c_int), ("walk_criterion", C.c_int), ("num_devices", C.c_int),
                ("cell_sums", C.c_int), ("device_ids", C.POINTER(C.c_int)), ("distributed", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_alive", C.c_uint64), ("steps", C.c_uint64), ("interactions", C.c_uint64),
                ("tree_nodes", C.c_uint64), ("tree_cells", C.c_uint64), ("merges", C.c_uint64),
                ("collision_pairs", C.c_uint64), ("ms_sort", C.c_float),
                ("ms_build", C.c_float), ("ms_force", C.c_float), ("ms_total", C.c_float)]


class DeviceState(C.Structure):
    _fields_ = [("pos_mass", C.c_void_p), ("vel", C.c_void_p), ("pos_mass_next", C.c_void_p),
                ("vel_next", C.c_void_p), ("n", C.c_uint64), ("elem_bytes", C.c_uint32), ("reserved", C.c_uint32),
                ("acc", C.c_void_p)]


class EngineError(RuntimeError):
    """A C-ABI call returned a non-zero code (the reference panics on every error, main.go:213-238)."""


def library_path():
    """The in-tree library; NBODY_AMD_LIB overrides it (used to compare builds of the same sources)."""
    return os.environ.get("NBODY_AMD_LIB") or os.path.join(_HERE, "libnbody_engine.so")


_lib = None


def load_library():
    """dlopen the engine and declare every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EngineError(f"{path} is missing: build it with `make -C nbody_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    dp, u64p, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    lib.nbe_default_config.argtypes = [C.POINTER(Config)]
    lib.nbe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.nbe_destroy.argtypes = [vp]
    lib.nbe_destroy.restype = None
    lib.nbe_last_error.argtypes = []
    lib.nbe_last_error.restype = C.c_char_p
    lib.nbe_upload.argtypes = [vp, C.c_size_t, dp, dp, dp, u64p]
    lib.nbe_download.argtypes = [vp, dp, dp, dp, u64p, C.POINTER(C.c_size_t)]
    lib.nbe_accel.argtypes = [vp, dp, u8p]
    lib.nbe_step.argtypes = [vp, C.c_int]
    lib.nbe_step_timed.argtypes = [vp, C.POINTER(Stats)]
    lib.nbe_sync.argtypes = [vp]
    lib.nbe_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.nbe_energy.argtypes = [vp, dp, dp, dp]
    lib.nbe_set_shard.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.nbe_shard_sort.argtypes = [vp]
    lib.nbe_shard_prepare.argtypes = [vp]
    lib.nbe_shard_advance.argtypes = [vp]
    lib.nbe_shard_commit.argtypes = [vp]
    lib.nbe_shard_advance_range.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.nbe_device_state.argtypes = [vp, C.POINTER(DeviceState)]
    lib.nbe_set_stream.argtypes = [vp, vp]
    lib.nbe_time_force_kernel.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    lib.nbe_tree_nodes.argtypes = [vp, dp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.nbe_frame_stream_begin.argtypes = [vp, C.c_int, C.c_int, C.c_int, FRAME_CB, vp]
    lib.nbe_frame_stream_end.argtypes = [vp]
    lib.nbe_set_radius.argtypes = [vp, dp]
    lib.nbe_get_radius.argtypes = [vp, dp, C.POINTER(C.c_size_t)]
    lib.nbe_accel_direct_noalloc.argtypes = [C.c_int, vp, C.c_size_t, vp, vp, C.c_double, C.c_double]
    for name in EXPORTS:
        if name not in ("nbe_destroy", "nbe_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Engine:
    """One engine per GPU.  Mirrors the reference's step seam (main.go:104-178): upload the live bodies,
    step(), download the copy a frame would take (main.go:90-101)."""

    def __init__(self, mode="direct", precision="f32", theta=None, dt=None, eps=None, G=None,
                 bound_center=None, bound_width=None, device=0, collisions=False, ref_com=False,
                 drop_tail=False, per_wave=False, devices=None, tree_sums=False, exact_sums=False, distributed=False):
        """devices: a list of HIP device ordinals; with more than one the engine itself drives all of them from this
        process (targets split into contiguous ranges, ranges copied between the GPUs every step); the same ordinal
        may appear more than once.
        tree_sums: fp32 node records take their cells' sums child by child in octant order (NBE_SUMS_TREE) instead of
        from prefix sums.
        exact_sums: cells take their sums from 128-bit integer prefix sums (NBE_SUMS_EXACT): bits that do not depend on
        how the bodies are partitioned, which is what the key-range-distributed build uses and equals bit for bit."""
        self._lib = load_library()
        cfg = Config()
        self._check(self._lib.nbe_default_config(C.byref(cfg)))
        cfg.mode = _MODES[mode] if isinstance(mode, str) else int(mode)
        cfg.precision = _PRECS[precision] if isinstance(precision, str) else int(precision)
        if theta is not None:
            cfg.theta = float(theta)
        if dt is not None:
            cfg.dt = float(dt)
        if eps is not None:
            cfg.eps = float(eps)
        if G is not None:
            cfg.G = float(G)
        if bound_center is not None:
            cfg.bound_center[:] = [float(c) for c in bound_center]
        if bound_width is not None:
            cfg.bound_width = float(bound_width)
        cfg.device = int(device)
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            cfg.num_devices = len(devices)
            cfg.device_ids = C.cast(ids, C.POINTER(C.c_int))
        cfg.cell_sums = 2 if exact_sums else (1 if tree_sums else 0)
        # several devices: every body on one GPU only, pruned subtrees exchanged (instead of all bodies on every GPU)
        cfg.distributed = 1 if distributed else 0
        cfg.collisions = int(bool(collisions))      # False = the reference's -nocollision run (main.go:24)
        # the reference's running "centre of mass" (tree.go:189-193) and its unwalked last n mod 4 bodies
        # (main.go:134-148), for number-for-number comparisons with the Go program
        cfg.reference_quirks = (QUIRK_REF_COM if ref_com else 0) | (QUIRK_DROP_TAIL if drop_tail else 0)
        # per_wave: the 128 bodies of a wavefront open cells together (faster, finer than the reference's per-body
        # test, not its interaction set)
        cfg.walk_criterion = WALK_PER_WAVE if per_wave else WALK_PER_BODY
        self.cfg = cfg
        self._h = C.c_void_p()
        self._check(self._lib.nbe_create(C.byref(cfg), C.byref(self._h)))
        cfg.device_ids = None                      # nbe_create copied the list
        self.n = 0

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"nbody_engine error {rc}: {self._lib.nbe_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nbe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- state ----
    def upload(self, pos, vel, mass, ids=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        mass = np.ascontiguousarray(mass, dtype=np.float64).reshape(-1)
        n = mass.shape[0]
        if pos.shape[0] != n or vel.shape[0] != n:
            raise ValueError("pos, vel and mass disagree on the number of bodies")
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            if ids.shape[0] != n:
                raise ValueError("ids has the wrong length")
            idp = ids.ctypes.data_as(C.POINTER(C.c_uint64))
        self._check(self._lib.nbe_upload(self._h, n, _dp(pos), _dp(vel), _dp(mass), idp))
        self.n = n

    def download(self):
        """Returns (pos[k,3], vel[k,3], mass[k], ids[k]) of the k live bodies in upload order."""
        n = self.n
        pos = np.zeros((n, 3)); vel = np.zeros((n, 3)); mass = np.zeros(n); ids = np.zeros(n, dtype=np.uint64)
        k = C.c_size_t(0)
        self._check(self._lib.nbe_download(self._h, _dp(pos), _dp(vel), _dp(mass),
                                           ids.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(k)))
        k = k.value
        return pos[:k], vel[:k], mass[:k], ids[:k]

    def set_radius(self, radius):
        """Body radii (physics.go:94) in upload order; collisions only.  Bodies start with radius 2 (physics.go:20)."""
        radius = np.ascontiguousarray(radius, dtype=np.float64).reshape(-1)
        if radius.shape[0] != self.n:
            raise ValueError("radius has the wrong length")
        self._check(self._lib.nbe_set_radius(self._h, _dp(radius)))

    def get_radius(self):
        """Radii of the live bodies, in the order of download()."""
        radius = np.zeros(self.n)
        k = C.c_size_t(0)
        self._check(self._lib.nbe_get_radius(self._h, _dp(radius), C.byref(k)))
        return radius[:k.value]

    # ---- force / step ----
    def accel(self):
        """Accelerations f/m of all uploaded bodies (upload order) and their alive flags."""
        acc = np.zeros((self.n, 3))
        alive = np.ones(self.n, dtype=np.uint8)
        self._check(self._lib.nbe_accel(self._h, _dp(acc), alive.ctypes.data_as(C.POINTER(C.c_uint8))))
        return acc, alive

    def step(self, nsteps=1):
        self._check(self._lib.nbe_step(self._h, int(nsteps)))

    def step_timed(self):
        st = Stats()
        self._check(self._lib.nbe_step_timed(self._h, C.byref(st)))
        return st

    def sync(self):
        self._check(self._lib.nbe_sync(self._h))

    def stats(self):
        st = Stats()
        self._check(self._lib.nbe_stats(self._h, C.byref(st)))
        return st

    def energy(self):
        ke, pe = C.c_double(), C.c_double()
        mom = np.zeros(3)
        self._check(self._lib.nbe_energy(self._h, C.byref(ke), C.byref(pe), _dp(mom)))
        return ke.value, pe.value, mom

    def time_force_kernel(self, reps=5):
        ms = C.c_float()
        self._check(self._lib.nbe_time_force_kernel(self._h, int(reps), C.byref(ms)))
        return ms.value

    def tree_nodes(self):
        """Pre-order node table [nodes, 6]: com x,y,z, mass, cell edge (0 = leaf), skip index."""
        cnt = C.c_size_t(0)
        self._check(self._lib.nbe_tree_nodes(self._h, None, 0, C.byref(cnt)))
        out = np.zeros((cnt.value, 6))
        if cnt.value:
            self._check(self._lib.nbe_tree_nodes(self._h, _dp(out), cnt.value, C.byref(cnt)))
        return out

    # ---- frame streaming ----
    def frame_stream_begin(self, callback, every=1, slots=32, first_frame=0):
        """callback(frame, xyzm[n_alive,4] float32, ids[n_alive] uint64, radius[n_alive] float32 or None) runs on a
        library thread for the state before every `every`-th step; the arrays are copies the callback may keep.
        radius is None unless collisions are on."""
        def tramp(_user, frame, n, xyzm, radius, ids):
            a = np.ctypeslib.as_array(xyzm, shape=(n, 4)).copy() if n else np.zeros((0, 4), np.float32)
            i = np.ctypeslib.as_array(ids, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
            r = None
            if radius:
                r = np.ctypeslib.as_array(radius, shape=(n,)).copy() if n else np.zeros(0, np.float32)
            callback(frame, a, i, r)
        thunk = FRAME_CB(tramp)
        self._check(self._lib.nbe_frame_stream_begin(self._h, int(every), int(slots), int(first_frame), thunk, None))
        self._frame_cb = thunk                    # keep the thunk alive while streaming (a refused call leaves the old one in place)

    def frame_stream_end(self):
        self._check(self._lib.nbe_frame_stream_end(self._h))
        self._frame_cb = None

    # ---- sharded stepping ----
    def set_shard(self, first, count):
        self._check(self._lib.nbe_set_shard(self._h, int(first), int(count)))

    def shard_sort(self):
        self._check(self._lib.nbe_shard_sort(self._h))

    def shard_prepare(self):
        self._check(self._lib.nbe_shard_prepare(self._h))

    def shard_advance(self):
        self._check(self._lib.nbe_shard_advance(self._h))

    def shard_commit(self):
        self._check(self._lib.nbe_shard_commit(self._h))

    def shard_advance_range(self, first, count):
        self._check(self._lib.nbe_shard_advance_range(self._h, int(first), int(count)))

    def device_state(self):
        ds = DeviceState()
        self._check(self._lib.nbe_device_state(self._h, C.byref(ds)))
        return ds

    def set_stream(self, hip_stream):
        self._check(self._lib.nbe_set_stream(self._h, C.c_void_p(hip_stream)))


def accel_direct_noalloc(device, hip_stream, n, d_pos_mass, d_acc, G=6.67408e-11, eps=0.0):
    """nbe_accel_direct_noalloc: fp32 direct-sum accelerations of n bodies from device pointer d_pos_mass
    (n x 4 fp32: x, y, z, mass) into device pointer d_acc (n x 4 fp32), queued on hip_stream; allocates nothing."""
    lib = load_library()
    rc = lib.nbe_accel_direct_noalloc(int(device), C.c_void_p(hip_stream), int(n), C.c_void_p(d_pos_mass),
                                      C.c_void_p(d_acc), float(G), float(eps))
    if rc != 0:
        raise EngineError(f"nbody_engine error {rc}: {lib.nbe_last_error().decode()}")
