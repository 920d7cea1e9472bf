"""Random call orders, most of them wrong (steps before any upload, a commit without an advance, receive without classify,
radii of the wrong length, frame streams with zero slots, downloads from a distributed engine that holds nothing), on
Engine and on DistRank: every call either works or raises EngineError / ValueError, and the process survives.  8,000
sequences in a one-off run (NBODY_FUZZ_SEEDS=4000), 160 here.  The first such run found two crashes: nbe_download on an
engine with distributed = 1 before any upload, and nbe_shard_commit with collisions on an engine that holds no body.
(Collected late, before the last file only: a crash here would take the rest of the run with it.)"""
import numpy as np
import pytest

from nbody_amd import Engine
from nbody_amd.engine import EngineError
from nbody_amd.disttree import DistRank

pytestmark = pytest.mark.gpu


def test_engine_misuse(gpu):
    for s in range(int(__import__("os").environ.get("NBODY_FUZZ_SEEDS", "80"))):
        rng = np.random.default_rng(50000 + s)
        mode = str(rng.choice(["direct", "tree"]))
        kw = dict(mode=mode, precision=str(rng.choice(["f32", "mixed", "f64"])), collisions=bool(rng.random() < 0.5))
        if rng.random() < 0.4:
            kw["devices"] = [0] * int(rng.integers(2, 4))
            if mode == "tree" and rng.random() < 0.5:
                kw["distributed"] = True
        try:
            e = Engine(**kw)
        except EngineError:
            continue
        log = [str(kw)]
        n = 0
        try:
            for k in range(int(rng.integers(3, 30))):
                op = "upload" if (k == 0 and rng.random() < 0.7) else str(rng.choice(["upload", "step", "step", "download", "stats", "radius_set", "radius_get", "accel", "energy", "nodes",
                                     "frames_on", "frames_off", "shard", "prepare", "advance", "commit", "sort", "timed", "kernel", "sync", "devstate"]))
                log.append(op)
                try:
                    if op == "upload":
                        n = int(rng.choice([0, 1, 5, 300, 9000]))
                        e.upload(rng.uniform(-3e4, 3e4, (n, 3)), rng.normal(0, 1e-2, (n, 3)), np.abs(rng.normal(1e4, 5e3, n)) + 1)
                    elif op == "step": e.step(int(rng.integers(0, 3)))
                    elif op == "download": e.download()
                    elif op == "stats": e.stats()
                    elif op == "radius_set": e.set_radius(rng.uniform(0.5, 3, n if rng.random() < 0.8 else n + 1))
                    elif op == "radius_get": e.get_radius()
                    elif op == "accel": e.accel()
                    elif op == "energy": e.energy()
                    elif op == "nodes": e.tree_nodes()
                    elif op == "frames_on": e.frame_stream_begin(lambda *a: None, every=int(rng.integers(0, 3)), slots=int(rng.integers(0, 4)))
                    elif op == "frames_off": e.frame_stream_end()
                    elif op == "shard": e.set_shard(int(rng.integers(0, 3)) * 256, int(rng.integers(0, max(n, 1) + 5)))
                    elif op == "prepare": e.shard_prepare()
                    elif op == "advance": e.shard_advance()
                    elif op == "commit": e.shard_commit()
                    elif op == "sort": e.shard_sort()
                    elif op == "timed": e.step_timed()
                    elif op == "kernel": e.time_force_kernel(1)
                    elif op == "sync": e.sync()
                    elif op == "devstate": e.device_state()
                except (EngineError, ValueError):
                    pass
        finally:
            e.close()


def test_dist_rank_misuse(gpu):
    for s in range(int(__import__("os").environ.get("NBODY_FUZZ_SEEDS", "80"))):
        rng = np.random.default_rng(60000 + s)
        world = int(rng.integers(1, 4))
        r = DistRank(int(rng.integers(0, world)), world, precision=str(rng.choice(["f32", "mixed", "f64"])), collisions=bool(rng.random() < 0.4))
        log = []
        n = 0
        try:
            for k in range(int(rng.integers(3, 30))):
                op = ["upload", "mass", "sample", "split"][k] if (k < 4 and world == 1) else str(rng.choice(["upload", "mass", "radius", "sample", "split", "classify", "room", "unclassify", "receive", "build",
                                     "export", "assemble", "advance", "energy", "download", "get_radius", "accel"]))
                log.append(op)
                try:
                    if op == "upload":
                        n = int(rng.choice([0, 1, 5, 300, 9000]))
                        r.upload(n + int(rng.integers(0, 3)) * 500, rng.uniform(-3e4, 3e4, (n, 3)), rng.normal(0, 1e-2, (n, 3)), np.abs(rng.normal(1e4, 5e3, n)) + 1, 0)
                    elif op == "mass": r.set_total_mass(float(rng.choice([1e9, 0.0, -1.0])))
                    elif op == "radius": r.set_radius(rng.uniform(0.5, 3, n), float(rng.choice([8.0, 1.0])))
                    elif op == "sample": r.sample_keys(int(rng.choice([0, 1, 64, 4096])))
                    elif op == "split":
                        sp = np.sort(rng.integers(0, 1 << 62, world + 1).astype(np.uint64)); sp[0] = 0; sp[-1] = np.uint64(1) << np.uint64(63)
                        if rng.random() < 0.2: sp[-1] = 5
                        r.set_splitters(sp)
                    elif op == "classify": r.classify()
                    elif op == "room": r.room()
                    elif op == "unclassify": r.unclassify()
                    elif op == "receive": r.receive(int(rng.choice([0, 0, 0, 3])))
                    elif op == "build": r.build()
                    elif op == "advance": r.advance()
                    elif op == "energy": r.energy()
                    elif op == "download": r.download()
                    elif op == "get_radius": r.get_radius()
                    elif op == "accel": r.accel()
                    elif op == "export" and world == 1:
                        fr, bx, live = r.build()
                        r.export(fr, np.array([len(fr)], dtype=np.int32), [bx])
                    elif op == "assemble" and world == 1:
                        r.assemble(np.zeros(1, np.int32), np.zeros(1, np.uint64))
                except (EngineError, ValueError, AssertionError):
                    pass
        finally:
            r.close()


def test_null_arguments_are_refused_not_dereferenced(gpu):
    """Every exported entry point with a NULL engine, and with a live engine and NULL / zero for everything else: an
    error code or a harmless success, never a crash (what a C or Go host would meet with a forgotten argument)."""
    import ctypes as C
    from nbody_amd import engine as E
    from nbody_amd import disttree
    lib = E.load_library()
    skip = {"nbe_last_error", "nbe_destroy", "nbe_create", "nbe_default_config", "nbe_accel_direct_noalloc"}
    names = [n for n in list(E.EXPORTS) + disttree.DIST_EXPORTS if n not in skip]
    proto = C.CFUNCTYPE(C.c_int, *([C.c_void_p] * 10))
    zeros = [None] * 9
    pos = np.random.default_rng(1).uniform(-1e4, 1e4, (500, 3))
    for name in names:
        fn = proto(C.cast(getattr(lib, name), C.c_void_p).value)      # (a pointer of its own: the library's stays typed)
        print("null sweep:", name, flush=True)             # (shown with -s, or in the captured output of a failure)
        fn(None, *zeros)
        for kw in (dict(mode="tree"), dict(mode="tree", collisions=True), dict(mode="direct"),
                   dict(mode="tree", devices=[0, 0]), dict(mode="tree", devices=[0, 0], distributed=True, collisions=True)):
            with Engine(**kw) as e:
                fn(e._h, *zeros)
                e.upload(pos, np.zeros((500, 3)), np.full(500, 1e6))
                fn(e._h, *zeros)
                try:
                    e.step(1)
                except EngineError:                        # (an empty shard, say, which nbe_step then refuses)
                    pass
                fn(e._h, *zeros)
        r = DistRank(0, 1)
        try:
            fn(r._h, *zeros)
            r.upload(600, pos, np.zeros((500, 3)), np.full(500, 1e6), 0)
            fn(r._h, *zeros)
            try:
                r.set_total_mass(5e8); r.set_splitters(np.array([0, 1 << 63], dtype=np.uint64)); r.classify()
            except EngineError:
                pass
            fn(r._h, *zeros)
        finally:
            r.close()
    cfg = E.Config()
    assert lib.nbe_create(None, None) != 0 and lib.nbe_default_config(None) != 0
    h = C.c_void_p()
    lib.nbe_create(C.byref(cfg), C.byref(h))               # (an all-zero config may or may not be valid; it must not crash)
    if h.value:
        lib.nbe_destroy(h)
    lib.nbe_destroy(None)
