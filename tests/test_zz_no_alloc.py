"""nbe_accel_direct_noalloc, the force pass that allocates nothing, against nbe_accel of an engine.

Named so that pytest collects it after every other test file: a failure here must not hide the others behind -x."""
import ctypes as C

import numpy as np
import pytest

from nbody_amd import Engine, engine as E, physics

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
