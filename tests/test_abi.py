"""The C-ABI library: it loads, exports every symbol the header declares, and validates arguments.
No compute calls here (no GPU needed); on a machine without a GPU engine creation must fail loudly."""
import ctypes as C
import os
import re

import pytest

from nbody_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbe_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in nbody_engine.h but not exported"
    from nbody_amd import disttree
    assert sorted(list(E.EXPORTS) + disttree.DIST_EXPORTS) == syms


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(E.Config) == 2 * 4 + 4 * 8 + 3 * 8 + 8 + 4 * 4 + 4 + 4 + 8 + 8      # ... num_devices, cell_sums, device_ids, distributed + padding
    from nbody_amd import disttree
    assert C.sizeof(disttree.DistBuffers) == 4 * 8 + 3 * 8 + 2 * 4 and disttree.FRAGMENT.itemsize == 8 + 4 * 4 + 8 * 8
    assert C.sizeof(E.Stats) == 8 * 8 + 4 * 4
    assert C.sizeof(E.DeviceState) == 4 * 8 + 8 + 4 + 4 + 8


def test_default_config_holds_reference_constants():
    lib = E.load_library()
    cfg = E.Config()
    assert lib.nbe_default_config(C.byref(cfg)) == 0
    assert cfg.G == 6.67408e-11 and cfg.theta == 1.0 and cfg.dt == 3600.0 and cfg.eps == 0.0
    assert cfg.bound_width == 65536.0 and list(cfg.bound_center) == [0.0, 0.0, 0.0]
    assert cfg.mode == E.MODE_DIRECT and cfg.precision == E.PREC_F32 and cfg.collisions == 0
    assert cfg.reference_quirks == 0
    assert lib.nbe_default_config(None) == -1
    assert b"NULL" in lib.nbe_last_error()


def test_create_rejects_bad_arguments():
    lib = E.load_library()
    cfg = E.Config()
    lib.nbe_default_config(C.byref(cfg))
    h = C.c_void_p()
    cfg.mode = 7
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    cfg.mode = 0
    cfg.collisions = 2
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    cfg.collisions = 0
    cfg.reference_quirks = E.QUIRK_REF_COM           # tree mode only
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    cfg.mode = 1
    cfg.reference_quirks = 4                         # unknown bit
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    cfg.reference_quirks = 0
    cfg.precision = 9
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1
    cfg.precision = 0
    cfg.bound_width = 0.0
    assert lib.nbe_create(C.byref(cfg), C.byref(h)) == -1
    assert lib.nbe_create(None, C.byref(h)) == -1
    assert lib.nbe_step(None, 1) == -1 and lib.nbe_upload(None, 0, None, None, None, None) == -1


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.EngineError) as ei:
        E.Engine()
    assert "no HIP device" in str(ei.value) or "device" in str(ei.value)


def _build_c_client(name="abi_smoke"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "c", name)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                           "-o", exe, os.path.join(root, "tests", "c", name + ".c"), "-L" + os.path.join(root, "nbody_amd"),
                           "-lnbody_engine", "-Wl,-rpath," + os.path.join(root, "nbody_amd")])
    return exe


def test_header_is_plain_c99_and_a_c_client_links():
    """include/nbody_engine.h must serve a cgo binding: strict C99, no C++; the client links against the library
    and, without a GPU, sees nbe_create fail loudly."""
    import subprocess
    import torch
    exe = _build_c_client()
    if not torch.cuda.is_available():
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
        assert r.returncode == 0 and b"no HIP device" in r.stdout


@pytest.mark.gpu
def test_c_client_steps_a_scene(gpu):
    import subprocess
    r = subprocess.run([_build_c_client(), "run"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and b"abi smoke ok" in r.stdout, r.stdout.decode()


def test_multi_shard_c_client_builds():
    _build_c_client("abi_multi")


@pytest.mark.gpu
def test_c_client_with_several_shards_equals_one_shard(gpu):
    """One engine handle with num_devices = 2, 3, 4 (all on device 0) against one shard: tree mode bit for bit in all
    three precisions, direct sum to rounding; 5,003 bodies (uneven ranges), one of which leaves the bound."""
    import subprocess
    r = subprocess.run([_build_c_client("abi_multi")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"abi multi ok" in r.stdout, r.stdout.decode()
