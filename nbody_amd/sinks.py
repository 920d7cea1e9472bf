"""Frame sinks of the reference above the engine's frame stream.

`SqliteSink` writes the database of database.go: same table, same column order, same rounding, one transaction
per frame, indices created at the end.  The reference's other sinks are a PNG renderer (render.go) and a gob chunk
store (gob.go, a Go-only serialisation); neither is rebuilt here.
"""
import os
import sqlite3
import threading

import numpy as np

# database.go:28-37
SCHEMA = """
CREATE TABLE bodies (
	frame 	INTEGER,
	id 		INTEGER, -- body id
	x 		REAL,
	y 		REAL,
	z 		REAL,
	mass 	REAL,
	radius 	REAL);
"""

# database.go:39-43
INDICES = """
CREATE INDEX idx_frame ON bodies (frame, id);
CREATE INDEX idx_id ON bodies (id);
CREATE INDEX idx_mass ON bodies (mass);
"""

INSERT = "INSERT INTO bodies VALUES (?, ?, ?, ?, ?, ?, ?);"                    # database.go:45
QUERY_FRAME = "SELECT * FROM bodies WHERE frame = ? ORDER BY id ASC;"          # database.go:46


def go_round(a):
    """math.Round: nearest integer, halves away from zero (numpy's round goes to even)."""
    a = np.asarray(a, dtype=np.float64)
    return np.copysign(np.floor(np.abs(a) + 0.5), a)


class SqliteSink:
    """opendb + frameToSqlite + createIndices (database.go:49-123).

    The reference refuses to open an existing file (database.go:50-54 prints "<name> exists" and exits);
    here that is a FileExistsError.  Journal and synchronous writes are off as in the reference's DSN
    (database.go:57).  `write` may be called from the engine's delivery thread; calls are serialised, as the
    reference notes that only one sqlite writer is useful (database.go:25)."""

    def __init__(self, filename):
        if os.path.exists(filename):
            raise FileExistsError(f"{filename} exists")
        self.filename = filename
        self._db = sqlite3.connect(filename, check_same_thread=False, isolation_level=None)
        self._db.execute("PRAGMA journal_mode=OFF")
        self._db.execute("PRAGMA synchronous=OFF")
        self._db.executescript(SCHEMA)                      # createTables, database.go:66-71
        self._lock = threading.Lock()
        self.frames_written = 0

    def write(self, frame, ids, xyzm, radius=None, default_radius=2.0):
        """One frameJob (render.go:33-36): a transaction with one row per body, values rounded as
        database.go:105-109 does.  radius None = every body has the default (physics.go:20)."""
        ids = np.asarray(ids)
        xyzm = np.asarray(xyzm, dtype=np.float64).reshape(-1, 4)
        r = np.full(len(ids), float(default_radius)) if radius is None else np.asarray(radius, dtype=np.float64)
        vals = go_round(np.column_stack([xyzm, r]))
        rows = [(int(frame), int(i), *map(float, v)) for i, v in zip(ids, vals)]
        with self._lock:
            self._db.execute("BEGIN")
            try:
                self._db.executemany(INSERT, rows)
            except Exception:
                self._db.execute("ROLLBACK")               # database.go:115-117
                raise
            self._db.execute("COMMIT")
            self.frames_written += 1

    def create_indices(self):
        """createIndices, database.go:74-79 (the reference calls it once after the run, main.go:194)."""
        with self._lock:
            self._db.executescript(INDICES)

    def query_frame(self, frame):
        with self._lock:
            return self._db.execute(QUERY_FRAME, (int(frame),)).fetchall()

    def close(self):
        with self._lock:
            self._db.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
