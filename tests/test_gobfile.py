"""Save / resume files (main.go:199-248): the encoding/gob stream of a frameJob.  Parity unpinned (no Go here to
read or write one); the primitives and the framing are pinned by the worked example of the encoding/gob package
documentation, `type Point struct{ X, Y int }` with the value Point{22, 33}."""
import numpy as np
import pytest

from nbody_amd import gobfile as G

DOC_POINT = bytes.fromhex(
    "1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00"
    "07 ff 82 01 2c 01 42 00".replace(" ", ""))


def test_primitives_match_the_package_documentation():
    assert G.enc_uint(0) == b"\x00" and G.enc_uint(7) == b"\x07" and G.enc_uint(256) == bytes.fromhex("fe0100")
    assert G.enc_int(22) == b"\x2c" and G.enc_int(33) == b"\x42" and G.enc_int(65) == bytes.fromhex("ff82")
    assert G.enc_int(-65) == bytes.fromhex("ff81") and G.enc_int(-129) == bytes.fromhex("fe0101")
    assert G.enc_float(17.0) == bytes.fromhex("fe3140")           # 0x4031000000000000 byte-reversed
    assert G.enc_float(0.0) == b"\x00"
    r = G._Reader(bytes.fromhex("fe3140ff81fe0100"))
    assert r.float() == 17.0 and r.int() == -65 and r.uint() == 256 and r.eof()


def test_documented_point_example_both_ways():
    made = G._message(-65, G._struct_def("Point", 65, [("X", G.T_INT), ("Y", G.T_INT)])) + \
        G._message(65, G._struct_value([22, 33], [G.enc_int, G.enc_int]))
    assert made == DOC_POINT
    assert G.decode_stream(DOC_POINT) == [{"X": 22, "Y": 33}]


def test_frame_job_round_trip_with_zero_fields():
    """Zero values are not sent (Frame 0, ID 0, a velocity component of 0) and come back as zeros."""
    ids = np.array([0, 7, 2 ** 40 + 5], dtype=np.uint64)
    mass = np.array([1e10, 0.0, 1234.5]); radius = np.array([1.0, 2.0, 2.5198420997897464])
    pos = np.array([[-9000.0, -100.0, -2000.0], [0.0, 0.0, 0.0], [1e-300, -2.5, 3.0e11]])
    vel = np.array([[0.004, 0.0, -0.001], [0.0, 0.0, 0.0], [1.0, -1.0, 0.0]])
    for frame in (0, 36, 8760):
        data = G.encode_frame_job(frame, ids, mass, radius, pos, vel)
        f, i2, m2, r2, p2, v2 = G.decode_frame_job(data)
        assert f == frame and np.array_equal(i2, ids) and np.array_equal(m2, mass) and np.array_equal(r2, radius)
        assert np.array_equal(p2, pos) and np.array_equal(v2, vel)
    empty = G.encode_frame_job(3, ids[:0], mass[:0], radius[:0], pos[:0], vel[:0])
    f, i2, m2, r2, p2, v2 = G.decode_frame_job(empty)
    assert f == 3 and len(i2) == 0 and p2.shape == (0, 3)


def test_descriptors_are_what_a_go_encoder_sends_first():
    """frameJob (render.go:33-36) then []body then body (physics.go:91-96, exported fields only), ids 65 / 67 / 66."""
    data = G.encode_frame_job(1, np.array([1], np.uint64), [2.0], [3.0], [[4.0, 5.0, 6.0]], [[7.0, 8.0, 9.0]])
    r = G._Reader(data)
    seen = []
    while not r.eof():
        n = r.uint(); m = G._Reader(r.d[r.p:r.p + n]); r.p += n
        tid = m.int()
        seen.append((tid, G._read_wire_type(m) if tid < 0 else None))
    assert [t for t, _ in seen] == [-65, -67, -66, 65]
    assert seen[0][1] == ("struct", [("Frame", 2), ("Bodies", 67)])
    assert seen[1][1] == ("slice", 66)
    assert seen[2][1] == ("struct", [("ID", 3)] + [(f, 4) for f in ("Mass", "Radius", "X", "Y", "Z", "Vx", "Vy", "Vz")])


def test_reader_accepts_other_ids_and_orders(tmp_path):
    """A Go encoder may number the types differently; the reader goes by the descriptors."""
    t_job, t_body, t_slice = 70, 72, 71
    data = G._message(-t_body, G._struct_def("body", t_body, [("ID", G.T_UINT), ("Mass", G.T_FLOAT), ("X", G.T_FLOAT)]))
    data += G._message(-t_slice, G._slice_def("[]main.body", t_slice, t_body))
    data += G._message(-t_job, G._struct_def("frameJob", t_job, [("Frame", G.T_INT), ("Bodies", t_slice)]))
    body = G._struct_value([9, 5.0, -1.5], [G.enc_uint, G.enc_float, G.enc_float])
    data += G._message(t_job, b"\x01" + G.enc_int(12) + b"\x01" + G.enc_uint(1) + body + b"\x00")
    f, ids, mass, radius, pos, vel = G.decode_frame_job(data)
    assert f == 12 and list(ids) == [9] and mass[0] == 5.0 and pos[0, 0] == -1.5 and radius[0] == 0.0
    with pytest.raises(ValueError):
        G.decode_frame_job(data[:-3])
    name = G.export_frame_data(36, ids, mass, radius, pos, vel, directory=str(tmp_path))
    assert name.endswith("0000000036.data")                      # main.go:211
    assert G.import_frame_data(name)[0] == 36
