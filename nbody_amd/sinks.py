"""Frame sinks of the reference above the engine's frame stream.

`SqliteSink` writes the database of database.go: same table, same column order, same rounding, one transaction
per frame, indices created at the end.  `ChunkSink` is the chunked, zlib-compressed gob store of gob.go (written
from the encoding/gob specification; no Go exists here to read the files back, so that format is unpinned).
The reference's default sink, the PNG renderer (render.go), is in render.py.
"""
import glob
import os
import sqlite3
import threading
import zlib

import numpy as np

from . import gobfile

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


CHUNK_FILENAME_FORMAT = "%010d.chunk"                                          # gob.go:51


def frames_for_chunk(chunk_number, chunk_size):
    """gob.go:172-175: inclusive frame indices of a chunk."""
    return chunk_number * chunk_size, (chunk_number + 1) * chunk_size - 1


def ideal_chunk_size(num_bodies):
    """gob.go:177-186."""
    size_per_body, ideal, min_bucket = 24, 32 * 1 << 20, 2
    frames_per_chunk = ideal // (size_per_body * num_bodies)
    return frames_per_chunk + (min_bucket - frames_per_chunk % min_bucket)


class ChunkSink:
    """newChunkRenderer + frameToMemory + dumper (gob.go:92-212): frames collect in memory per chunk of
    `frames_per_chunk` frames; a chunk that has all its frames is gob-encoded, zlib-compressed (the reference
    settles on level 2, gob.go:47) and written to `<dir>/%010d.chunk`, named by the chunk's HIGH frame index
    (gob.go:161)."""

    def __init__(self, last_frame, frames_per_chunk, compression_lvl=2, output_dir="chunks"):
        self.output_dir, self.compression_lvl, self.chunk_size = output_dir, int(compression_lvl), int(frames_per_chunk)
        nchunks = last_frame // self.chunk_size + 1
        self.expected = [0] * nchunks                                           # gob.go:108-111
        for frame in range(last_frame + 1):
            self.expected[frame // self.chunk_size] += 1
        self.current = [0] * nchunks
        self.chunks = [None] * nchunks
        self._lock = threading.Lock()
        self.files_written = []

    def write(self, frame, ids, xyzm, radius=None, default_radius=2.0):
        """frameToMemory for one frameJob (gob.go:193-207), then finishedFrame (gob.go:122-139)."""
        radius = np.full(len(ids), default_radius, np.float32) if radius is None else np.asarray(radius, np.float32)
        record = (np.asarray(ids).copy(), np.asarray(xyzm, np.float32).reshape(-1, 4).copy(), radius.copy())
        with self._lock:
            c = int(frame) // self.chunk_size
            self.current[c] += 1
            full = self.current[c] == self.expected[c]
            if self.chunks[c] is None:
                self.chunks[c] = {}
            self.chunks[c][int(frame)] = record
            data = self.chunks[c] if full else None
            if full:
                self.chunks[c] = None
        if full:
            self._dump(c, data)

    def _dump(self, chunk_num, data):
        _, high = frames_for_chunk(chunk_num, self.chunk_size)
        os.makedirs(self.output_dir, exist_ok=True)                             # gob.go:146
        raw = zlib.compress(gobfile.encode_render_index(data), self.compression_lvl)
        name = os.path.join(self.output_dir, CHUNK_FILENAME_FORMAT % high)
        with open(name, "wb") as f:
            f.write(raw)
        with self._lock:
            self.files_written.append(name)


def read_chunk_from_filename(filename):
    """readChunkFromFilename (gob.go:226-246): {frame: {id: {"X","Y","Z","Mass","Radius"}}}."""
    with open(filename, "rb") as f:
        vals = gobfile.decode_stream(zlib.decompress(f.read()))
    if len(vals) != 1 or not isinstance(vals[0], dict):
        raise ValueError("not a renderindex chunk")
    return vals[0]


def determine_chunk_size_from_folder(directory):
    """determineChunkSizeFromFolder (gob.go:251-274)."""
    matches = sorted(glob.glob(os.path.join(directory, "*.chunk")))
    if not matches:
        raise FileNotFoundError(f"found 0 chunk files in '{directory}'")
    frames = [int(os.path.splitext(os.path.basename(m))[0]) for m in matches[:2]]
    return frames[0] + 1 if len(frames) < 2 else frames[1] - frames[0]


def read_chunk_from_frame(directory, frame_number):
    """readChunkFromFrame (gob.go:215-224) -> (chunk, low, high)."""
    size = determine_chunk_size_from_folder(directory)
    low, high = frames_for_chunk(frame_number // size, size)
    return read_chunk_from_filename(os.path.join(directory, CHUNK_FILENAME_FORMAT % high)), low, high
