"""The PNG renderer (nbody_amd/render.py, after render.go) against the scalar restatement of the Go source in
oracle/render_py.py, pixel for pixel, plus properties that do not need the restatement.  No image rendered by the
reference exists (parity unpinned)."""
import math
import os

import numpy as np
import pytest

from nbody_amd import physics, render
from oracle import render_py as R


def scene(n, seed, spread=3000.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 4))
    b[:, :3] = rng.normal(0.0, spread, (n, 3))
    b[:, 3] = np.abs(rng.normal(1e3, 500.0, n))
    b[0] = (-9000.0, -100.0, -2000.0, 1e10)      # the two cores of main.go:35-36
    b[1] = (9000.0, 100.0, 2000.0, 1e10)
    return b


def oracle_film(frame, b, w, h):
    pix, cam = R.render_frame(frame, [tuple(r) for r in b], w, h)
    return np.array(pix, dtype=np.uint8), cam


@pytest.mark.parametrize("frame,seed,size", [(0, 1, (320, 180)), (37, 2, (320, 180)), (1000, 3, (160, 90)), (200, 4, (1920, 1080))])
def test_film_equals_scalar_restatement(frame, seed, size):
    w, h = size
    b = scene(300 if w < 1920 else 120, seed)
    film = render.render_frame(frame, b, w, h, labels=False)
    ref, cam = oracle_film(frame, b, w, h)
    assert film.shape == (h, w, 4)
    assert np.array_equal(film, ref)
    assert (film[:, :, :3].reshape(-1, 3).any(axis=1)).sum() > 50      # something was drawn


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_film_equals_restatement_with_bodies_everywhere(seed):
    """Bodies far outside the bound: behind the camera (w <= 0), left and right of the image (the depth index then
    names a cell of the neighbouring row), above and below it, and coincident ones."""
    b = scene(600, seed, spread=40000.0)
    b[10] = b[11]                                # coincident bodies: equal depth, the later one stays
    b[12, :3] = b[0, :3]                         # on top of a light
    frame = 100 * seed
    film = render.render_frame(frame, b, 96, 54, labels=False)
    ref, _ = oracle_film(frame, b, 96, 54)
    assert np.array_equal(film, ref)


def test_camera_matches_restatement_bitwise():
    b = scene(10, 5)
    for frame in (0, 1, 359, 1440, 8760):
        cam = render.Camera(frame, b[0, :3], b[1, :3])
        _, ref = R.render_frame(frame, [tuple(r) for r in b[:2]], 64, 36)
        c2 = render.Camera(frame, b[0, :3], b[1, :3], 64, 36)
        assert list(c2.pos) == ref["cam_pos"] and c2.fov == ref["fov"] and c2.angle == ref["angle"]
        assert [c2.vp[r, c] for c in range(4) for r in range(4)] == ref["vp"]
        # the orbit keeps its radius and height (rotation about y)
        assert abs(np.linalg.norm(cam.pos) - 32768.0) < 1e-6
        assert abs(cam.pos[1] - 32768.0 / math.sqrt(27.0)) < 1e-6


def test_aim_point_lands_in_the_image_centre():
    b = scene(2, 6)
    cam = render.Camera(12, b[0, :3], b[1, :3])
    front, nx, ny, nz = render._project(cam.vp, cam.middle[None, :])
    assert front[0] and abs(nx[0]) < 1e-9 and abs(ny[0]) < 1e-9
    # the two cores span two thirds of the image height (fov is 1.5 x their separation, render.go:207) when the
    # camera looks at them square on; from the side they can only be closer together
    f2, x2, y2, _ = render._project(cam.vp, b[:2, :3])
    assert f2.all() and np.hypot((x2[0] - x2[1]) * 16 / 9, y2[0] - y2[1]) <= 2 / 1.5 * (1 + 1e-9)


def test_depth_test_keeps_the_nearest_body_and_the_last_of_equals():
    b = scene(2, 7)
    cam = render.Camera(0, b[0, :3], b[1, :3], 64, 36)
    toward = cam.pos - cam.middle
    toward /= np.linalg.norm(toward)
    # three bodies on the line of sight through the aim point: the nearest must stay whatever the order
    pts = np.array([cam.middle + toward * d for d in (-4000.0, 3000.0, 0.0)])
    colors = np.array([[10, 0, 0], [0, 20, 0], [0, 0, 30]], dtype=np.uint8)
    for perm in ([0, 1, 2], [2, 1, 0], [1, 0, 2]):
        film = np.zeros((36, 64, 4), dtype=np.uint8)
        z = np.full((36, 64), np.inf)
        render.splat_bodies(film, z, cam.vp, pts[perm], colors[perm])
        assert tuple(film[18, 32, :3]) == (0, 20, 0)
        assert np.isfinite(z).sum() == 1
    # equal depth: the later one overwrites (render.go:441 tests with <=)
    film = np.zeros((36, 64, 4), dtype=np.uint8)
    z = np.full((36, 64), np.inf)
    render.splat_bodies(film, z, cam.vp, pts[[2, 2]], colors[[0, 2]])
    assert tuple(film[18, 32, :3]) == (0, 0, 30)


def test_depth_index_runs_into_the_next_row_like_the_reference():
    """A point left of the image still lowers the depth of the cell its index names (render.go:439, 576-578): a
    farther body at that pixel of the row above is then hidden although nothing was drawn there."""
    w, h = 64, 36
    vp = np.eye(4)                                               # NDC in, NDC out
    film = np.zeros((h, w, 4), dtype=np.uint8)
    z = np.full((h, w), np.inf)
    def ndc(px, py):                                             # centre of pixel (px, py)
        return (px + 0.5) * 2 / w - 1, 1 - (py + 0.5) * 2 / h
    x_out, y_row = ndc(-4, 10)                                   # -3.5 truncates to -3: index = 10*w - 3 = pixel (w-3, 9)
    x_hid, y_hid = ndc(w - 3, 9)
    pts = np.array([[x_out, y_row, 0.2], [x_hid, y_hid, 0.5]])
    colors = np.array([[200, 0, 0], [0, 200, 0]], dtype=np.uint8)
    render.splat_bodies(film, z, vp, pts, colors)
    assert not film[:, :, :3].any()                              # the first is outside, the second is "behind" it
    assert z[9, w - 3] == 0.2
    # the scalar restatement agrees
    f = R.Film(w, h)
    ident = [1.0 if i % 5 == 0 else 0.0 for i in range(16)]
    R.plot_point_3d(f, (200, 0, 0, 255), ident, list(pts[0]))
    R.plot_point_3d(f, (0, 200, 0, 255), ident, list(pts[1]))
    assert f.z[9 * w + w - 3] == 0.2 and f.pix[9][w - 3] == (0, 0, 0, 255)


def test_body_at_a_light_is_black_like_the_reference():
    """The cores sit on their own lights: the light direction normalises a zero vector, the colour is NaN and
    converts to 0 (render.go:57-59, 82-89)."""
    b = scene(5, 8)
    cam = render.Camera(3, b[0, :3], b[1, :3])
    c = render.body_colors(cam.pos, b[0, :3], b[1, :3], b[:, :3])
    assert tuple(c[0]) == (0, 0, 0) and tuple(c[1]) == (0, 0, 0)
    assert c[2:].any()


def test_png_round_trip_and_sink(tmp_path):
    b = scene(200, 9)
    film = render.render_frame(5, b, 320, 180)
    data = render.encode_png(film)
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    assert np.array_equal(render.decode_png(data), film)
    labelled = film[16:80, :120, :3].any()
    assert labelled                                              # the four labels top left
    with render.ImageSink(str(tmp_path / "img"), workers=2, width=320, height=180) as sink:
        for frame in range(4):
            sink.write(frame, np.arange(len(b)), b.astype(np.float32))
    names = sorted(os.listdir(tmp_path / "img"))
    assert names == ["%010d.png" % k for k in range(4)]
    back = render.decode_png(open(tmp_path / "img" / names[3], "rb").read())
    assert np.array_equal(back, render.render_frame(3, b.astype(np.float32), 320, 180))
    quirk = render.ImageSink(str(tmp_path / "img2"), workers=1, reference_names=True)
    assert os.path.basename(quirk.filename(5)) == "%010d.png5"   # render.go:249 uses Sprint, not Sprintf
    quirk.close()


def test_needs_two_bodies():
    with pytest.raises(ValueError):
        render.render_frame(0, np.zeros((1, 4)))


def test_default_scene_renders():
    g = physics.makebodies(1000, seed=1)
    xyzm = np.concatenate([g.pos, g.mass[:, None]], axis=1)
    film = render.render_frame(0, xyzm, 480, 270)
    lit = film[:, :, :3].reshape(-1, 3).any(axis=1).sum()
    assert lit > 300


def test_cli_renders_host_frame_files(tmp_path):
    b = scene(50, 11).astype(np.float32)
    frames = tmp_path / "frames"
    frames.mkdir()
    for k in (0, 7):
        with open(frames / ("%010d.frame" % k), "wb") as f:
            f.write(np.uint32(len(b)).tobytes())
            for i, r in enumerate(b):
                f.write(np.uint32(i).tobytes() + np.float32([r[0], r[1], r[2], r[3], 2.0]).tobytes())
    ids, xyzm, radius = render.read_frame_file(str(frames / "0000000007.frame"))
    assert np.array_equal(ids, np.arange(50)) and np.array_equal(xyzm, b) and (radius == 2.0).all()
    assert render.main([str(frames), str(tmp_path / "img"), "--workers", "1"]) == 0
    assert sorted(os.listdir(tmp_path / "img")) == ["0000000000.png", "0000000007.png"]
    film = render.decode_png(open(tmp_path / "img" / "0000000007.png", "rb").read())
    assert np.array_equal(film, render.render_frame(7, b))


@pytest.mark.gpu
def test_simulation_streams_frames_into_the_renderer(tmp_path):
    """Engine frame stream -> ImageSink: the image of frame k is the rendering of the state before step k."""
    g = physics.makebodies(300, seed=3)
    sim = physics.Simulation(g, tree=True, nocollision=True)
    states = {}

    class Tee:
        def __init__(self, inner): self.inner = inner
        def write(self, frame, ids, xyzm, radius):
            states[frame] = np.array(xyzm, copy=True)
            self.inner.write(frame, ids, xyzm, radius)
    with render.ImageSink(str(tmp_path / "img"), workers=2, width=480, height=270) as sink:
        sim.run(3, Tee(sink))
    sim.close()
    assert sorted(os.listdir(tmp_path / "img")) == ["%010d.png" % k for k in range(4)]
    for k in range(4):
        film = render.decode_png(open(tmp_path / "img" / ("%010d.png" % k), "rb").read())
        assert np.array_equal(film, render.render_frame(k, states[k], 480, 270))
    assert not np.array_equal(states[0], states[3])
