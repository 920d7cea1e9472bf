"""The sqlite sink (database.go) above the frame stream."""
import os
import sqlite3

import numpy as np
import pytest

from nbody_amd import physics
from nbody_amd.sinks import SqliteSink, go_round


def test_go_round_is_half_away_from_zero():
    """math.Round (database.go:105-109), unlike numpy's round-half-even."""
    assert list(go_round([0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, -2.5001, 0.0, 1e10 + 0.5])) == \
        [1.0, 2.0, 3.0, -1.0, -2.0, 2.0, -3.0, 0.0, 1e10 + 1.0]


def test_schema_rows_and_indices(tmp_path):
    name = str(tmp_path / "bodies.sqlite")
    with SqliteSink(name) as db:
        ids = np.array([5, 2, 9], dtype=np.uint64)
        xyzm = np.array([[1.5, -2.5, 3.49, 1000.4], [0.0, 10.6, -0.4, 1e10], [7.0, 8.0, 9.0, 2.5]], dtype=np.float32)
        db.write(0, ids, xyzm, radius=np.array([2.0, 1.0, 2.52]))
        db.write(1, ids[:2], xyzm[:2])                       # default radius 2 (physics.go:20)
        db.create_indices()
        rows = db.query_frame(0)                             # ORDER BY id ASC, database.go:46
        assert rows == [(0, 2, 0.0, 11.0, -0.0, 1e10, 1.0), (0, 5, 2.0, -3.0, 3.0, 1000.0, 2.0),
                        (0, 9, 7.0, 8.0, 9.0, 3.0, 3.0)]
        assert db.query_frame(1) == [(1, 2, 0.0, 11.0, -0.0, 1e10, 2.0), (1, 5, 2.0, -3.0, 3.0, 1000.0, 2.0)]
    con = sqlite3.connect(name)
    cols = [(r[1], r[2]) for r in con.execute("PRAGMA table_info(bodies)")]
    assert cols == [("frame", "INTEGER"), ("id", "INTEGER"), ("x", "REAL"), ("y", "REAL"), ("z", "REAL"),
                    ("mass", "REAL"), ("radius", "REAL")]                   # database.go:28-37
    idx = sorted(r[1] for r in con.execute("PRAGMA index_list(bodies)"))
    assert idx == ["idx_frame", "idx_id", "idx_mass"]                      # database.go:39-43
    assert con.execute("SELECT COUNT(*) FROM bodies").fetchone()[0] == 5
    con.close()


def test_existing_file_is_refused(tmp_path):
    """database.go:50-54 exits when the file exists."""
    name = tmp_path / "bodies.sqlite"
    name.write_bytes(b"")
    with pytest.raises(FileExistsError, match="exists"):
        SqliteSink(str(name))


@pytest.mark.gpu
@pytest.mark.parametrize("collide", [False, True])
def test_simulation_streams_frames_into_sqlite(gpu, tmp_path, collide):
    """main.go:80-101 with frameToSqlite as the worker: frames+1 frames, each the live bodies before its step."""
    b = physics.makebodies(300, seed=2)
    sim = physics.Simulation(b, tree=True, nocollision=not collide)
    name = str(tmp_path / "bodies.sqlite")
    frames = 12
    with SqliteSink(name) as db:
        sim.run(frames, db)
        db.create_indices()
        assert db.frames_written == frames + 1
        f0 = db.query_frame(0)
        assert [r[1] for r in f0] == list(range(302))
        want = go_round(np.column_stack([b.pos.astype(np.float32), b.mass.astype(np.float32), b.radius]))
        assert np.array_equal(np.array([r[2:] for r in f0]), want)
        last = db.query_frame(frames)
        assert 0 < len(last) <= 302 and abs(sum(r[5] for r in last) - sum(r[5] for r in f0)) <= 302
    final = sim.frame()
    sim.close()
    assert len(final) <= 302 and len(final.radius) == len(final)


def test_chunk_store_files_and_round_trip(tmp_path):
    """gob.go: frames collect per chunk, a full chunk becomes `%010d.chunk` named by its HIGH frame index, zlib
    around a gob of map[frame]map[id]renderbody; the last chunk may hold fewer frames (gob.go:108-111)."""
    from nbody_amd import sinks
    rng = np.random.default_rng(0)
    last_frame, per_chunk, n = 6, 3, 5
    sink = sinks.ChunkSink(last_frame, per_chunk, compression_lvl=2, output_dir=str(tmp_path / "chunks"))
    assert sink.expected == [3, 3, 1]
    sent = {}
    for frame in [1, 0, 2, 3, 6, 5, 4]:                       # workers deliver out of order
        ids = np.arange(n, dtype=np.uint64) + 10
        xyzm = rng.normal(0, 1e3, (n, 4)).astype(np.float32)
        radius = np.full(n, 2.0, np.float32); radius[0] = 2.5198421
        sent[frame] = (ids, xyzm, radius)
        sink.write(frame, ids, xyzm, radius)
    names = sorted(os.path.basename(f) for f in sink.files_written)
    assert names == ["0000000002.chunk", "0000000005.chunk", "0000000008.chunk"]
    assert sinks.determine_chunk_size_from_folder(str(tmp_path / "chunks")) == 3
    chunk, low, high = sinks.read_chunk_from_frame(str(tmp_path / "chunks"), 4)
    assert (low, high) == (3, 5) and sorted(chunk) == [3, 4, 5]
    ids, xyzm, radius = sent[4]
    for k, i in enumerate(ids):
        b = chunk[4][int(i)]
        assert (np.float32(b["X"]), np.float32(b["Y"]), np.float32(b["Z"]), np.float32(b["Mass"])) == tuple(xyzm[k])
        assert np.float32(b["Radius"]) == radius[k]
    assert sorted(sinks.read_chunk_from_filename(str(tmp_path / "chunks" / "0000000008.chunk"))) == [6]
    assert sinks.ideal_chunk_size(50002) == 28 and sinks.frames_for_chunk(2, 3) == (6, 8)


@pytest.mark.gpu
def test_save_and_resume_through_gob(gpu, tmp_path):
    """-save / -state (main.go:42-47, 199-201): a resumed run continues exactly where the saved one stood."""
    b = physics.makebodies(200, seed=4)
    sim = physics.Simulation(b, tree=True, precision="f64")
    sim.step(5)
    name = sim.save(str(tmp_path))
    assert os.path.basename(name) == "0000000005.data"
    sim.step(5)
    want = sim.frame()
    sim.close()
    again = physics.Simulation.resume(name, tree=True, precision="f64")
    assert again.frame_index == 5
    again.step(5)
    got = again.frame()
    again.close()
    assert np.array_equal(got.ids, want.ids) and np.array_equal(got.pos, want.pos) and np.array_equal(got.vel, want.vel)
