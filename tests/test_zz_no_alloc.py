"""nbe_accel_direct_noalloc, the force pass that allocates nothing, against nbe_accel of an engine and against the
CPU restatement of the reference's direct loop (main.go:105-128; parity unpinned: the reference holds no fixtures).

Named so that pytest collects it after every other test file: a failure here must not hide the others behind -x."""
import ctypes as C

import numpy as np
import pytest

from nbody_amd import Engine, engine as E, physics
from oracle import oracle as O
from tests.util import f32, relerr

pytestmark = pytest.mark.gpu


def _engine_accel(b, eps):
    with Engine(mode="direct", precision="f32", eps=eps) as e:
        e.upload(b.pos, b.vel, b.mass)
        acc, alive = e.accel()
    assert alive.all()
    return acc.astype(np.float32)


def test_no_alloc_force_pass_equals_nbe_accel_on_two_calls_in_a_row(gpu):
    """Two calls in a row through the C ABI on one stream, the second with fewer bodies (it finds the first call's
    partial sums in the device array): each must equal the engine's nbe_accel for the same bodies.  Both sizes spread
    the source range over several slices, i.e. both use the fixed array; the masses differ from body to body, so
    both entry points run the same kernels and the accelerations must agree bit for bit."""
    import torch
    lib = E.load_library()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for n, eps in ((50000, 0.0), (3001, 2.0)):
        b = physics.makebodies(n - 2, seed=n)
        assert len(np.unique(b.mass)) > 1
        want = _engine_accel(b, eps)
        posm = torch.from_numpy(np.concatenate([b.pos, b.mass[:, None]], axis=1).astype(np.float32)).to(dev)
        acc = torch.full((n, 4), float("nan"), dtype=torch.float32, device=dev)
        rc = lib.nbe_accel_direct_noalloc(0, C.c_void_p(stream), n, C.c_void_p(posm.data_ptr()), C.c_void_p(acc.data_ptr()),
                                          6.67408e-11, eps)
        assert rc == 0, lib.nbe_last_error().decode()
        got = acc.cpu().numpy()
        assert np.isfinite(got).all() and not got[:, 3].any()
        assert np.array_equal(got[:, :3], want), np.abs(got[:, :3] - want).max()
        # and against the CPU direct sum on the fp32-rounded inputs (the smaller scene in full, a sample of the larger)
        pos32, m32 = f32(b.pos), f32(b.mass)
        rows = np.arange(n) if n <= 4096 else np.arange(0, n, n // 512)[:512]
        ref = np.concatenate([O.direct_rows(m32, pos32, t0=int(t), t1=int(t) + 1, eps=eps, nthreads=1) for t in rows])
        ref = ref / m32[rows][:, None]
        er = relerr(got[rows, :3].astype(np.float64), ref)
        assert np.median(er) <= 1e-6 and np.quantile(er, 0.999) <= 1e-4, (np.median(er), er.max())


def test_no_alloc_force_pass_leaves_the_callers_device_current(gpu):
    """The entry point makes `device` current for its launches and must put the caller's device back."""
    import torch
    lib = E.load_library()
    hip = C.CDLL("libamdhip64.so")
    before = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(before)) == 0
    t = torch.ones(64, 4, dtype=torch.float32, device="cuda:0")
    out = torch.empty_like(t)
    ndev = torch.cuda.device_count()
    if ndev > 1:                                  # make another device current: the call must switch back to it
        assert hip.hipSetDevice(1) == 0
        before = C.c_int(1)
    assert lib.nbe_accel_direct_noalloc(0, None, 64, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), 1.0, 1.0) == 0
    after = C.c_int(-2)
    assert hip.hipGetDevice(C.byref(after)) == 0
    assert after.value == before.value
    if ndev > 1:
        assert hip.hipSetDevice(0) == 0
    torch.cuda.synchronize()


def test_no_alloc_force_pass_rejects_what_does_not_fit(gpu):
    import torch
    lib = E.load_library()
    t = torch.zeros(8, 4, dtype=torch.float32, device="cuda")
    p = C.c_void_p(t.data_ptr())
    assert lib.nbe_accel_direct_noalloc(0, None, 262145, p, p, 1.0, 0.0) == -4
    assert b"262144" in lib.nbe_last_error()
    assert lib.nbe_accel_direct_noalloc(0, None, 8, None, p, 1.0, 0.0) == -1
    assert lib.nbe_accel_direct_noalloc(0, None, 8, p, p, 1.0, -1.0) == -1
    assert lib.nbe_accel_direct_noalloc(99, None, 8, p, p, 1.0, 0.0) == -1
    assert lib.nbe_accel_direct_noalloc(0, None, 0, None, None, 1.0, 0.0) == 0
