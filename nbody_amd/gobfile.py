"""Save / resume files of the reference (main.go:199-248): one encoding/gob stream holding a `frameJob`
(render.go:33-36: Frame int, Bodies []body) with the exported fields of `body` (physics.go:91-96: ID uint64 and
eight float64; the force accumulators are unexported and gob leaves them out).

No Go toolchain exists where this was written, so a file written here has never been read by the reference and
vice versa: PARITY UNPINNED for this format.  The wire rules are those of the encoding/gob package documentation
(unsigned / signed / float encodings, delta-encoded struct fields with omitted zero values, type descriptors sent
as wireType values under negative ids); its worked example `Point{22, 33}` is the known answer the tests pin the
primitives with.  The reader parses type descriptors generically, so it does not depend on the ids a Go
encoder happens to assign.
"""
import os
import struct

import numpy as np

# type ids fixed by the gob package
T_BOOL, T_INT, T_UINT, T_FLOAT, T_BYTES, T_STRING = 1, 2, 3, 4, 5, 6
FIRST_USER_ID = 65

BODY_FIELDS = ("ID", "Mass", "Radius", "X", "Y", "Z", "Vx", "Vy", "Vz")


# ---- primitives ------------------------------------------------------------------------------------------------
def enc_uint(u):
    if u < 128:
        return bytes([u])
    raw = u.to_bytes((u.bit_length() + 7) // 8, "big")
    return bytes([256 - len(raw)]) + raw


def enc_int(i):
    return enc_uint((~i << 1) | 1 if i < 0 else i << 1)


def enc_float(f):
    return enc_uint(int.from_bytes(struct.pack("<d", f), "big"))    # Float64bits, byte-reversed


def enc_string(s):
    b = s.encode()
    return enc_uint(len(b)) + b


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def eof(self):
        return self.p >= len(self.d)

    def uint(self):
        b = self.d[self.p]
        self.p += 1
        if b < 128:
            return b
        n = 256 - b
        if n > 8 or self.p + n > len(self.d):
            raise ValueError("gob: bad unsigned integer")
        v = int.from_bytes(self.d[self.p:self.p + n], "big")
        self.p += n
        return v

    def int(self):
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def float(self):
        return struct.unpack("<d", self.uint().to_bytes(8, "big"))[0]

    def bytes(self):
        n = self.uint()
        v = self.d[self.p:self.p + n]
        if len(v) != n:
            raise ValueError("gob: truncated")
        self.p += n
        return v


# ---- writer ----------------------------------------------------------------------------------------------------
def _common(name, tid):
    return b"\x01" + enc_string(name) + b"\x01" + enc_int(tid) + b"\x00"           # CommonType{Name, Id}


def _struct_def(name, tid, fields):
    body = b"\x01" + _common(name, tid) + b"\x01" + enc_uint(len(fields))          # structType{CommonType, Field}
    for fname, ftid in fields:
        body += b"\x01" + enc_string(fname) + b"\x01" + enc_int(ftid) + b"\x00"    # fieldType{Name, Id}
    return b"\x03" + body + b"\x00" + b"\x00"                                       # wireType.StructT is field 2


def _slice_def(name, tid, elem):
    body = b"\x01" + _common(name, tid) + b"\x01" + enc_int(elem) + b"\x00"        # sliceType{CommonType, Elem}
    return b"\x02" + body + b"\x00"                                                 # wireType.SliceT is field 1


def _map_def(name, tid, key, elem):
    body = b"\x01" + _common(name, tid) + b"\x01" + enc_int(key) + b"\x01" + enc_int(elem) + b"\x00"   # mapType
    return b"\x04" + body + b"\x00"                                                 # wireType.MapT is field 3


def _message(tid, payload):
    m = enc_int(tid) + payload
    return enc_uint(len(m)) + m


def _struct_value(values, encoders):
    out, last = b"", -1
    for k, (v, enc) in enumerate(zip(values, encoders)):
        if v == 0:                       # zero values are not sent
            continue
        out += enc_uint(k - last) + enc(v)
        last = k
    return out + b"\x00"


def encode_frame_job(frame, ids, mass, radius, pos, vel):
    """The byte stream of gob.NewEncoder(w).Encode(frameJob{...}) (main.go:218-219).  Ids follow the order in
    which the gob package registers the types: frameJob 65, body 66, []body 67; descriptors go out outermost first."""
    t_job, t_body, t_slice = FIRST_USER_ID, FIRST_USER_ID + 1, FIRST_USER_ID + 2
    out = _message(-t_job, _struct_def("frameJob", t_job, [("Frame", T_INT), ("Bodies", t_slice)]))
    out += _message(-t_slice, _slice_def("[]main.body", t_slice, t_body))
    out += _message(-t_body, _struct_def("body", t_body, [("ID", T_UINT)] + [(f, T_FLOAT) for f in BODY_FIELDS[1:]]))
    n = len(ids)
    bodies = b""
    encs = [enc_uint] + [enc_float] * 8
    for k in range(n):
        vals = [int(ids[k]), float(mass[k]), float(radius[k]), *map(float, pos[k]), *map(float, vel[k])]
        bodies += _struct_value(vals, encs)
    value = b""
    last = -1
    if frame != 0:
        value += enc_uint(1) + enc_int(int(frame))
        last = 0
    if n > 0:
        value += enc_uint(1 - last) + enc_uint(n) + bodies
    return out + _message(t_job, value + b"\x00")


def encode_render_index(frames):
    """gob of a `renderindex` (gob.go:54-61): map[uint32]framedata, framedata = map[uint32]renderbody,
    renderbody = {X, Y, Z, Mass, Radius float32} (sent as float64 like every gob float).
    `frames` maps frame number -> (ids, xyzm[n,4], radius[n]).  Type ids in the gob package's registration order:
    renderbody 65, framedata 66, renderindex 67; descriptors outermost first; a top-level map is preceded by a zero
    field delta."""
    t_body, t_fd, t_ri = FIRST_USER_ID, FIRST_USER_ID + 1, FIRST_USER_ID + 2
    out = _message(-t_ri, _map_def("renderindex", t_ri, T_UINT, t_fd))
    out += _message(-t_fd, _map_def("framedata", t_fd, T_UINT, t_body))
    out += _message(-t_body, _struct_def("renderbody", t_body, [(f, T_FLOAT) for f in ("X", "Y", "Z", "Mass", "Radius")]))
    val = enc_uint(len(frames))
    encs = [enc_float] * 5
    for frame in sorted(frames):
        ids, xyzm, radius = frames[frame]
        xyzm = np.asarray(xyzm, dtype=np.float32).reshape(-1, 4)
        radius = np.asarray(radius, dtype=np.float32)
        val += enc_uint(int(frame)) + enc_uint(len(ids))
        for k in range(len(ids)):
            val += enc_uint(int(ids[k]) & 0xffffffff)                                # uint32(b.ID), gob.go:198
            val += _struct_value([float(xyzm[k, 0]), float(xyzm[k, 1]), float(xyzm[k, 2]), float(xyzm[k, 3]),
                                  float(radius[k])], encs)
    return out + _message(t_ri, b"\x00" + val)


# ---- reader ----------------------------------------------------------------------------------------------------
def _read_wire_type(r):
    """A wireType value -> ("struct", [(name, id)...]) | ("slice", elem) | ("array", elem, len) | ("map", key, elem)."""
    def common():
        fld = -1
        while True:
            d = r.uint()
            if d == 0:
                return
            fld += d
            if fld == 0:
                r.bytes()
            elif fld == 1:
                r.int()
            else:
                raise ValueError("gob: unknown CommonType field")

    kind = None
    fld = -1
    while True:
        d = r.uint()
        if d == 0:
            break
        fld += d
        inner = -1
        if fld == 2:                                 # StructT
            fields = []
            while True:
                d2 = r.uint()
                if d2 == 0:
                    break
                inner += d2
                if inner == 0:
                    common()
                elif inner == 1:
                    for _ in range(r.uint()):
                        name, tid, f = "", 0, -1
                        while True:
                            d3 = r.uint()
                            if d3 == 0:
                                break
                            f += d3
                            if f == 0:
                                name = r.bytes().decode()
                            elif f == 1:
                                tid = r.int()
                        fields.append((name, tid))
            kind = ("struct", fields)
        elif fld in (0, 1, 3):                       # ArrayT, SliceT, MapT
            vals = {}
            while True:
                d2 = r.uint()
                if d2 == 0:
                    break
                inner += d2
                if inner == 0:
                    common()
                else:
                    vals[inner] = r.int()
            kind = {0: ("array", vals.get(1, 0), vals.get(2, 0)), 1: ("slice", vals.get(1, 0)),
                    3: ("map", vals.get(1, 0), vals.get(2, 0))}[fld]
        else:
            raise ValueError("gob: unsupported wireType field %d" % fld)
    if kind is None:
        raise ValueError("gob: empty type descriptor")
    return kind


def _read_value(r, tid, types):
    if tid == T_BOOL:
        return r.uint() != 0
    if tid == T_INT:
        return r.int()
    if tid == T_UINT:
        return r.uint()
    if tid == T_FLOAT:
        return r.float()
    if tid == T_BYTES:
        return r.bytes()
    if tid == T_STRING:
        return r.bytes().decode()
    if tid not in types:
        raise ValueError("gob: value of undefined type %d" % tid)
    t = types[tid]
    if t[0] == "struct":
        out = {name: _zero(ftid, types) for name, ftid in t[1]}
        fld = -1
        while True:
            d = r.uint()
            if d == 0:
                return out
            fld += d
            if fld >= len(t[1]):
                raise ValueError("gob: field number out of range")
            name, ftid = t[1][fld]
            out[name] = _read_value(r, ftid, types)
    if t[0] in ("slice", "array"):
        return [_read_value(r, t[1], types) for _ in range(r.uint())]
    if t[0] == "map":
        return {_read_value(r, t[1], types): _read_value(r, t[2], types) for _ in range(r.uint())}
    raise ValueError("gob: unsupported kind " + t[0])


def _zero(tid, types):
    if tid in (T_INT, T_UINT):
        return 0
    if tid == T_FLOAT:
        return 0.0
    if tid == T_BOOL:
        return False
    if tid == T_STRING:
        return ""
    if tid == T_BYTES:
        return b""
    t = types.get(tid)
    if t and t[0] == "struct":
        return {name: _zero(ftid, types) for name, ftid in t[1]}
    if t and t[0] == "map":
        return {}
    return []


def decode_stream(data):
    """All top-level values of a gob stream, as Python dicts / lists / numbers."""
    r, types, values = _Reader(data), {}, []
    while not r.eof():
        n = r.uint()
        m = _Reader(r.d[r.p:r.p + n])
        if len(m.d) != n:
            raise ValueError("gob: truncated message")
        r.p += n
        tid = m.int()
        if tid < 0:
            types[-tid] = _read_wire_type(m)
            continue
        if tid in types and types[tid][0] == "struct":
            values.append(_read_value(m, tid, types))
        else:
            if m.uint() != 0:                        # a non-struct top-level value is preceded by a zero delta
                raise ValueError("gob: missing singleton marker")
            values.append(_read_value(m, tid, types))
    return values


def decode_frame_job(data):
    """-> (frame, ids[u64], mass, radius, pos[n,3], vel[n,3]) from the bytes of a `%010d.data` file."""
    vals = decode_stream(data)
    if len(vals) != 1 or not isinstance(vals[0], dict) or "Bodies" not in vals[0]:
        raise ValueError("gob: not a frameJob")
    job = vals[0]
    bodies = job.get("Bodies") or []
    n = len(bodies)
    ids = np.array([b.get("ID", 0) for b in bodies], dtype=np.uint64)
    col = lambda f: np.array([b.get(f, 0.0) for b in bodies], dtype=np.float64)
    pos = np.stack([col("X"), col("Y"), col("Z")], axis=1) if n else np.zeros((0, 3))
    vel = np.stack([col("Vx"), col("Vy"), col("Vz")], axis=1) if n else np.zeros((0, 3))
    return int(job.get("Frame", 0)), ids, col("Mass"), col("Radius"), pos, vel


# ---- the two helpers of main.go ---------------------------------------------------------------------------------
def export_frame_data(frame, ids, mass, radius, pos, vel, directory="."):
    """exportFrameData (main.go:210-224): writes `%010d.data` and returns its path."""
    fname = os.path.join(directory, "%010d.data" % frame)
    try:
        with open(fname, "wb") as f:
            f.write(encode_frame_job(frame, ids, mass, radius, pos, vel))
    except Exception:
        if os.path.exists(fname):
            os.remove(fname)                          # main.go:221
        raise
    return fname


def import_frame_data(filename):
    """importFrameData (main.go:226-241)."""
    with open(filename, "rb") as f:
        return decode_frame_job(f.read())
