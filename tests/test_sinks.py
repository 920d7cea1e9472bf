"""The sqlite sink (database.go) above the frame stream."""
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
