"""The C++ host (host/nbody_main.cpp): the reference's driver loop (main.go) above the C ABI."""
import os
import struct
import subprocess

import numpy as np
import pytest

from nbody_amd import physics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "nbody")


def run(args, cwd):
    return subprocess.run([BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)


def read_frame(path):
    raw = open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 0)[0]
    rec = np.frombuffer(raw, dtype=np.dtype([("id", "<u4"), ("xyzmr", "<f4", 5)]), count=n, offset=4)
    assert len(raw) == 4 + 24 * n
    return rec


def test_host_builds_and_rejects_unknown_flags(tmp_path):
    assert os.path.exists(BIN), "build it with `make -C host` (or __graft_entry__.build())"
    r = run(["-bogus"], tmp_path)
    assert r.returncode == 2 and b"flag provided but not defined" in r.stdout


def test_host_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run(["-n", "10", "-y", "0.001", "-nocollision", "-norender"], tmp_path)
    assert r.returncode == 2 and b"no HIP device" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tree", [False, True])
def test_host_frame_loop_save_and_resume(gpu, tmp_path, tree):
    """frames = int(y*365*24) and the loop runs frames+1 iterations (main.go:30,80); every iteration sends the
    state BEFORE its step to the workers (main.go:90-101); -save / -state continue the frame numbering
    (main.go:42-47, 199-201)."""
    mode = ["-tree"] if tree else []
    r = run(["-n", "500", "-y", "0.004", "-nocollision", "-save", "-seed", "5"] + mode, tmp_path)   # frames = 35
    assert r.returncode == 0, r.stdout.decode()
    out = r.stdout.decode()
    assert "bodies: 502" in out and "frames: 35" in out and ("tree: true" if tree else "tree: false") in out
    files = sorted(os.listdir(tmp_path / "frames"))
    assert files == [f"{k:010d}.frame" for k in range(36)]
    f0, f35 = read_frame(tmp_path / "frames" / files[0]), read_frame(tmp_path / "frames" / files[35])
    assert len(f0) == 502 and set(f0["id"]) == set(range(502))
    ids = np.argsort(f0["id"])
    assert list(f0["xyzmr"][ids][0][:4]) == [-9000.0, -100.0, -2000.0, 1e10] and f0["xyzmr"][ids][0][4] == 1.0
    assert np.isfinite(f35["xyzmr"]).all() and not np.array_equal(np.sort(f0["xyzmr"][:, 0]), np.sort(f35["xyzmr"][:, 0]))
    state = tmp_path / f"{36:010d}.data"
    assert state.exists()
    # resume: 0.001 y = 8 more frames, numbered from 36
    r = run(["-n", "500", "-y", "0.001", "-nocollision", "-state", str(state), "-frames-dir", "more"] + mode, tmp_path)
    assert r.returncode == 0, r.stdout.decode()
    assert sorted(os.listdir(tmp_path / "more")) == [f"{k:010d}.frame" for k in range(36, 36 + 9)]
    first = read_frame(tmp_path / "more" / f"{36:010d}.frame")
    assert len(first) >= 495
    # -norender runs the same loop without frames
    r = run(["-n", "500", "-y", "0.002", "-nocollision", "-norender"] + mode, tmp_path)
    assert r.returncode == 0 and b"Done. Took" in r.stdout


@pytest.mark.gpu
def test_host_tree_run_with_collisions(gpu, tmp_path):
    """The reference's default is collisions on (main.go:24); in tree mode the host runs them on the GPU.  A dense
    seed scene loses bodies, the frames shrink with it and carry the merged radii."""
    r = run(["-n", "3000", "-y", "0.003", "-tree", "-seed", "3", "-save"], tmp_path)        # frames = 26
    assert r.returncode == 0, r.stdout.decode()
    assert "collisions: true" in r.stdout.decode()
    files = sorted(os.listdir(tmp_path / "frames"))
    assert files == [f"{k:010d}.frame" for k in range(27)]
    first, last = read_frame(tmp_path / "frames" / files[0]), read_frame(tmp_path / "frames" / files[-1])
    assert len(first) == 3002 and len(last) <= len(first)
    assert abs(float(first["xyzmr"][:, 3].sum()) - float(last["xyzmr"][:, 3].sum())) <= 1e-3 * float(first["xyzmr"][:, 3].sum())
    if len(last) < len(first):
        assert last["xyzmr"][:, 4].max() > 2.0            # cbrt(2^3 + 2^3) > 2


@pytest.mark.gpu
def test_host_default_invocation(gpu, tmp_path):
    """`nbody` with the reference's own defaults (main.go:17-24): 10 bodies + 2 cores, direct sum, collisions on,
    frames rendered; a short run here (-y)."""
    r = run(["-y", "0.002"], tmp_path)                      # frames = 17
    assert r.returncode == 0, r.stdout.decode()
    out = r.stdout.decode()
    assert "collisions: true" in out and "tree: false" in out and "bodies: 12" in out
    files = sorted(os.listdir(tmp_path / "frames"))
    assert files == [f"{k:010d}.frame" for k in range(18)]
    assert len(read_frame(tmp_path / "frames" / files[0])) == 12


def test_state_files_are_the_same_gob_stream_in_both_hosts(tmp_path):
    """-save / -state (main.go:199-248): the C++ host (host/gob.hpp) and the Python host (nbody_amd/gobfile.py) write
    the same bytes for the same frameJob and read each other's files; -recode needs no GPU."""
    from nbody_amd import gobfile as G
    rng = np.random.default_rng(5)
    n = 40
    ids = np.arange(n, dtype=np.uint64); ids[3] = 2 ** 40 + 1
    mass = np.abs(rng.normal(1e3, 500, n)); radius = np.full(n, 2.0); radius[0] = 1.0
    pos = rng.normal(0, 1e3, (n, 3)); vel = rng.normal(0, 1e-2, (n, 3))
    vel[5] = 0.0; pos[6, 1] = 0.0; mass[7] = 0.0                      # zero fields are left out of the stream
    src = tmp_path / "in"; src.mkdir()
    name = G.export_frame_data(36, ids, mass, radius, pos, vel, directory=str(src))
    r = run(["-recode", name], tmp_path)
    assert r.returncode == 0, r.stdout.decode()
    out = tmp_path / "0000000036.data"
    assert out.read_bytes() == open(name, "rb").read()
    f, i2, m2, r2, p2, v2 = G.import_frame_data(str(out))
    assert f == 36 and np.array_equal(i2, ids) and np.array_equal(m2, mass) and np.array_equal(p2, pos) and np.array_equal(v2, vel)
    bad = tmp_path / "bad.data"; bad.write_bytes(open(name, "rb").read()[:-5])
    r = run(["-recode", str(bad)], tmp_path)
    assert r.returncode == 2 and b"gob" in r.stdout


@pytest.mark.gpu
def test_host_asgo_run_equals_the_reference_configuration_of_the_oracle(gpu, tmp_path):
    """-asgo: the tree path with the Go program's own arithmetic (running cell centres tree.go:189-193, last n mod 4
    bodies unwalked main.go:134-148).  A state file goes in (-state), 0.001 y = 8 frames = 9 iterations run
    (main.go:80), and the final state this host saves as frame 9 is compared with 9 steps of the oracle as the
    reference runs: fp64, 1e-10 of the largest coordinate."""
    from nbody_amd import gobfile as G
    from oracle import oracle as O
    b = physics.makebodies(203, seed=4)                          # 205 bodies: body 204 is never walked
    n = len(b.mass)
    src = tmp_path / "in"; src.mkdir()
    name = G.export_frame_data(0, np.arange(n, dtype=np.uint64), b.mass, np.full(n, 2.0), b.pos, b.vel, directory=str(src))
    r = run(["-state", name, "-y", "0.001", "-tree", "-nocollision", "-asgo", "-precision", "f64", "-save"], tmp_path)
    assert r.returncode == 0, r.stdout.decode()
    f, ids, m2, r2, p2, v2 = G.import_frame_data(str(tmp_path / f"{9:010d}.data"))
    rp, rv, ralive, _ = O.step(b.mass, b.pos, b.vel, nsteps=9, mode="tree", com_mode="ref", theta=1.0, drop_tail=True)
    k = ids.astype(np.int64)
    assert f == 9 and np.array_equal(k, np.nonzero(ralive)[0])
    assert np.abs(p2 - rp[k]).max() <= 1e-10 * np.abs(rp[k]).max()
    assert np.abs(v2 - rv[k]).max() <= 1e-10 * np.abs(rv[k]).max()
    assert np.allclose(p2[k == n - 1][0], b.pos[n - 1] + 9 * 3600.0 * b.vel[n - 1], rtol=1e-14, atol=0)


@pytest.mark.gpu
def test_host_distributed_tree_streams_the_frames_of_one_gpu(gpu, tmp_path):
    """-distributed (several shards on this one GPU through -devices): no shard holds all bodies and every frame is
    gathered from all of them; the frame files hold the rows of a one-GPU run (by id, fp32 tree: the same bits when the
    one GPU sums its cells child by child, to rounding against the default prefix sums)."""
    common = ["-n", "4000", "-y", "0.002", "-nocollision", "-tree", "-seed", "9", "-theta", "0.6"]      # frames = 17
    r = run(common + ["-devices", "0,0,0", "-distributed", "-frames-dir", "dist"], tmp_path)
    assert r.returncode == 0, r.stdout.decode()
    r = run(common + ["-frames-dir", "one"], tmp_path)
    assert r.returncode == 0, r.stdout.decode()
    files = sorted(os.listdir(tmp_path / "dist"))
    assert files == sorted(os.listdir(tmp_path / "one")) == [f"{k:010d}.frame" for k in range(18)]
    for name in (files[0], files[9], files[17]):
        a, b = read_frame(tmp_path / "dist" / name), read_frame(tmp_path / "one" / name)
        a, b = a[np.argsort(a["id"])], b[np.argsort(b["id"])]
        assert np.array_equal(a["id"], b["id"]) and 3900 < len(a) <= 4002        # (a few bodies leave the bound, in both runs alike)
        assert np.abs(a["xyzmr"] - b["xyzmr"]).max() <= 1e-5 * np.abs(b["xyzmr"][:, :3]).max()
    first = read_frame(tmp_path / "dist" / files[0])
    one = read_frame(tmp_path / "one" / files[0])
    assert np.array_equal(first[np.argsort(first["id"])]["xyzmr"], one[np.argsort(one["id"])]["xyzmr"])


@pytest.mark.gpu
def test_host_distributed_tree_with_collisions_equals_one_gpu(gpu, tmp_path):
    """The reference's default run has collisions on (main.go:24).  -distributed with them, fp64 (both runs then sum their
    cells exactly): the frame files, radii included, and the state saved at the end hold the rows of a one-GPU run bit
    for bit."""
    from nbody_amd import gobfile as G
    common = ["-n", "3000", "-y", "0.002", "-tree", "-seed", "3", "-theta", "0.7", "-precision", "f64", "-save"]
    for name, extra in (("dist", ["-devices", "0,0,0", "-distributed"]), ("one", [])):
        d = tmp_path / name; d.mkdir()
        r = run(common + extra + ["-frames-dir", "frames"], d)
        assert r.returncode == 0, r.stdout.decode()
    files = sorted(os.listdir(tmp_path / "dist" / "frames"))
    assert files == sorted(os.listdir(tmp_path / "one" / "frames")) and len(files) == 18
    for name in (files[0], files[8], files[17]):
        a, b = read_frame(tmp_path / "dist" / "frames" / name), read_frame(tmp_path / "one" / "frames" / name)
        a, b = a[np.argsort(a["id"])], b[np.argsort(b["id"])]
        assert np.array_equal(a["id"], b["id"]) and np.array_equal(a["xyzmr"], b["xyzmr"])
    sa = G.import_frame_data(str(tmp_path / "dist" / f"{18:010d}.data"))
    sb = G.import_frame_data(str(tmp_path / "one" / f"{18:010d}.data"))
    assert sa[0] == sb[0] == 18
    for x, y in zip(sa[1:], sb[1:]):
        assert np.array_equal(x, y)
