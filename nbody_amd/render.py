"""The reference's default frame sink, the PNG renderer of render.go, behind the engine's frame stream.

`ImageSink.write(frame, ids, xyzm, radius)` has the signature of the other sinks (sinks.py) and does what
frameToImages (render.go:97-259) does with a frameJob: sort the bodies by mass, aim an orbiting perspective camera
at the midpoint of the two heaviest, draw the axes and the simulation bound as z-tested lines, light every body
with six lights and splat it as one z-tested pixel, encode the film as PNG.  The per-body work (projection,
lighting, depth test) is vectorised with numpy and reproduces the outcome of the reference's sequential loop,
including its depth-buffer index that runs into the neighbouring row for points left or right of the image
(render.go:439, 576-578).  The reference runs four such workers on one channel (main.go:51-61); so does this.

Deviations: the four text labels (render.go:231-234) use the 8x16 Inconsolata face of golang.org/x/image, which does
not exist here; they are drawn with a small built-in 5x7 face instead (labels=False leaves them out).  The reference
names its files with fmt.Sprint instead of fmt.Sprintf (render.go:249) and so writes "img/%010d.png<frame>";
`reference_names=True` repeats that, the default is the "%010d.png" its constant intends.  The output folder is
created if missing (the reference panics).  Positions arrive as the fp32 of the frame stream.
No rendered image of the reference exists to compare with (parity unpinned); tests check this module pixel for
pixel against a scalar restatement of the Go source.
"""
import math
import os
import queue
import struct
import threading
import zlib

import numpy as np

WIDTH, HEIGHT = 1920, 1080                 # render.go:99-100
CAM_RADIUS = float(1 << 15)                # render.go:103
AXIS_LENGTH = CAM_RADIUS / 10.0            # render.go:104
SIM_WIDTH = float(1 << 16)                 # render.go:105
LIGHT_INTENSITY = 1e6                      # render.go:110
FILENAME_FORMAT = "%010d.png"              # render.go:31

WHITE = (255, 255, 255, 255)
GRAY = (128, 128, 128, 255)
RED = (255, 0, 0, 255)
GREEN = (0, 255, 0, 255)
BLUE = (0, 0, 255, 255)
CORNER_ORDER = ((0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 4), (1, 5), (2, 6), (3, 7))


# ---- camera (mathgl conventions: column vectors, GL clip space) -------------------------------------------------
def _normalize(v):
    with np.errstate(divide="ignore", invalid="ignore"):
        return v * (1.0 / math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))) if np.any(v) else v * np.inf


def _quat_rotate(angle, axis, v):
    w, s = math.cos(angle / 2), math.sin(angle / 2)
    qv = axis * s
    cross = np.cross(qv, v)
    return v + cross * (2 * w) + np.cross(qv * 2, cross)


def look_at(eye, center, up):
    f = _normalize(center - eye)
    s = _normalize(np.cross(f, _normalize(up)))
    u = np.cross(s, f)
    m = np.array([[s[0], s[1], s[2], 0.0], [u[0], u[1], u[2], 0.0], [-f[0], -f[1], -f[2], 0.0], [0.0, 0.0, 0.0, 1.0]])
    t = np.eye(4)
    t[:3, 3] = -eye
    return _matmul4(m, t)


def perspective(fovy, aspect, near, far):
    nmf = near - far
    f = 1.0 / math.tan(fovy / 2.0)
    return np.array([[f / aspect, 0, 0, 0], [0, f, 0, 0], [0, 0, (near + far) / nmf, (2.0 * far * near) / nmf], [0, 0, -1, 0]])


def _matmul4(a, b):
    """4x4 product with the terms of every element added in index order (what a plain loop does; BLAS may not)."""
    out = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            acc = 0.0
            for k in range(4):
                acc += float(a[r, k]) * float(b[k, c])
            out[r, c] = acc
    return out


class Camera:
    """Camera of one frame (render.go:200-208): orbits the origin by a quarter degree per frame and frames the two
    heaviest bodies."""

    def __init__(self, frame, g0, g1, width=WIDTH, height=HEIGHT):
        g0 = np.asarray(g0, dtype=np.float64)
        g1 = np.asarray(g1, dtype=np.float64)
        self.middle = (g0 + g1) * 0.5
        self.angle = (float(frame) * math.pi / 180.0) / 4.0
        start = _normalize(np.array([1.0, 1.0, 5.0])) * CAM_RADIUS
        self.pos = _quat_rotate(-self.angle, np.array([0.0, 1.0, 0.0]), start)
        d = g0 - g1
        c = self.pos - self.middle
        dl = math.sqrt(float(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        cl = math.sqrt(float(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]))
        self.fov = 2 * math.atan(0.5 * (dl * 1.5) / cl)                 # fovFromRadiusBase, render.go:93
        self.vp = _matmul4(perspective(self.fov, float(width) / float(height), 0.1, 100),
                           look_at(self.pos, self.middle, np.array([0.0, 1.0, 0.0])))


def _go_int(a):
    """Go's int(float64): truncation toward zero; NaN and out-of-range become the smallest int64 (amd64)."""
    a = np.asarray(a, dtype=np.float64)
    bad = ~(np.abs(a) < 9.2e18)
    return np.where(bad, np.int64(-(1 << 63)), np.where(bad, 0.0, a).astype(np.int64))


def _project(vp, pts):
    """pts (n,3) -> (w > 0 mask, screen x, screen y, ndc z) as plotpoint3d does (render.go:430-438)."""
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    t = [vp[r, 0] * x + vp[r, 1] * y + vp[r, 2] * z + vp[r, 3] * 1.0 for r in range(4)]
    front = t[3] > 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = 1 / t[3]
        nx, ny, nz = t[0] * inv, t[1] * inv, t[2] * inv
    return front, nx, ny, nz


# ---- lighting (render.go:42-91) -----------------------------------------------------------------------------
def body_colors(cam_pos, g0, g1, pts):
    """sumLo for every body, white surface: (n, 3) uint8."""
    lights = [(np.array([AXIS_LENGTH, 0, 0]), np.array([1.0, 0.001, 0.001]), LIGHT_INTENSITY),
              (np.array([0, AXIS_LENGTH, 0]), np.array([0.001, 1.0, 0.001]), LIGHT_INTENSITY),
              (np.array([0, 0, AXIS_LENGTH]), np.array([0.001, 0.001, 1.0]), LIGHT_INTENSITY),
              (np.asarray(g0, dtype=np.float64), np.array([0.9, 0.9, 0.5]), 0.075 * LIGHT_INTENSITY),
              (np.asarray(g1, dtype=np.float64), np.array([0.9, 0.9, 0.5]), 0.075 * LIGHT_INTENSITY)]
    n = len(pts)
    total = [np.zeros(n), np.zeros(n), np.zeros(n)]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        tc = [cam_pos[k] - pts[:, k] for k in range(3)]
        inv = 1.0 / np.sqrt(tc[0] * tc[0] + tc[1] * tc[1] + tc[2] * tc[2])
        dc = [tc[k] * inv for k in range(3)]
        for pos, color, intensity in lights:
            lp = [pos[k] - pts[:, k] for k in range(3)]
            lensqr = lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]
            linv = 1.0 / np.sqrt(lensqr)
            ln = [lp[k] * linv for k in range(3)]
            frac = ((dc[0] * ln[0] + dc[1] * ln[1] + dc[2] * ln[2]) + 1.0) / 2.0
            k_ = math.pi * intensity * frac / lensqr
            for ch in range(3):
                total[ch] = total[ch] + color[ch] * k_
        dl = _normalize(np.asarray(cam_pos, dtype=np.float64))          # the directional light from the camera
        frac = ((dc[0] * dl[0] + dc[1] * dl[1] + dc[2] * dl[2]) + 1.0) / 2.0
        for ch in range(3):
            total[ch] = total[ch] + 0.1 * frac
        out = np.zeros((n, 3), dtype=np.uint8)
        for ch in range(3):
            v = total[ch] * 255.0
            v = np.where(v < 0.0, 0.0, np.where(v > 255.0, 255.0, v))   # mgl64.Clamp lets NaN through
            out[:, ch] = np.where(np.isnan(v), 0.0, v).astype(np.int64).astype(np.uint8)
    return out


# ---- lines (render.go:449-574): a few thousand pixels per frame, drawn one by one ------------------------------
_PLANES = ((np.array([1.0, 0, 0]), np.array([-1.0, 0, 0])), (np.array([-1.0, 0, 0]), np.array([1.0, 0, 0])),
           (np.array([0, 1.0, 0]), np.array([0, -1.0, 0])), (np.array([0, -1.0, 0]), np.array([0, 1.0, 0])))


def _dot3(a, b):
    return float(a[0]) * float(b[0]) + float(a[1]) * float(b[1]) + float(a[2]) * float(b[2])


def _clip(p1, p2):
    for n, p in _PLANES:
        b1 = _dot3(p1 - p, n) < 0
        b2 = _dot3(p2 - p, n) < 0
        if b1 and b2:
            return True
        if b1:
            dline, dplane = p2 - p1, p - p1
            p1[:] = p1 + dline * (_dot3(dplane, n) / _dot3(dline, n))
        elif b2:
            dline, dplane = p1 - p2, p - p2
            p2[:] = p2 + dline * (_dot3(dplane, n) / _dot3(dline, n))
    return False


def _plot_line_3d(film, zbuf, color, vp, p1, p2):
    h, w = film.shape[:2]

    def xf(p):
        return np.array([float(vp[r, 0]) * p[0] + float(vp[r, 1]) * p[1] + float(vp[r, 2]) * p[2] + float(vp[r, 3]) * 1.0
                         for r in range(4)])
    t1, t2 = xf(p1), xf(p2)
    if t1[3] <= 0 and t2[3] <= 0:
        return
    for behind, front in ((t1, t2), (t2, t1)):
        if behind[3] <= 0:
            t = (0 - behind[3]) / (front[3] - behind[3])
            behind[:3] = behind[:3] + t * (front[:3] - behind[:3])
            behind[3] = 1
            break
    a = t1[:3] * (1 / t1[3])
    b = t2[:3] * (1 / t2[3])
    if _clip(a, b):
        return
    x0, y0 = int(_go_int((a[0] + 1) * float(w) / 2)), int(_go_int((1 - a[1]) * float(h) / 2))
    x1, y1 = int(_go_int((b[0] + 1) * float(w) / 2)), int(_go_int((1 - b[1]) * float(h) / 2))
    z0, z1 = float(a[2]), float(b[2])
    xs, ys, zs = x0, y0, z0
    full = math.sqrt(float(x1 - x0) ** 2 + float(y1 - y0) ** 2)
    dx, sx = abs(x1 - x0), (1 if x0 < x1 else -1)
    dy, sy = -abs(y1 - y0), (1 if y0 < y1 else -1)
    err = dx + dy
    nz = zbuf.size
    flat = zbuf.reshape(-1)
    while True:
        i = x0 + y0 * w
        if 0 <= i < nz and z0 <= flat[i]:
            if 0 <= x0 < w and 0 <= y0 < h:
                film[y0, x0] = color
            flat[i] = z0
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy
        cur = math.sqrt(float(x0 - xs) ** 2 + float(y0 - ys) ** 2)
        z0 = zs + (z1 - zs) * (cur / full)


def draw_background(film, zbuf, vp):
    """drawBG (render.go:149-175): black film, the three axes, the twelve edges of the simulation bound."""
    film[:, :, :3] = 0
    film[:, :, 3] = 255
    zero = np.zeros(3)
    _plot_line_3d(film, zbuf, RED, vp, zero, np.array([AXIS_LENGTH, 0, 0]))
    _plot_line_3d(film, zbuf, GREEN, vp, zero, np.array([0, AXIS_LENGTH, 0]))
    _plot_line_3d(film, zbuf, BLUE, vp, zero, np.array([0, 0, AXIS_LENGTH]))
    hw = 0.5 * SIM_WIDTH
    corners = [np.array([hw * ((i & 1) * 2 - 1.0), hw * (((i & 2) >> 1) * 2 - 1.0), hw * (((i & 4) >> 2) * 2 - 1.0)]) for i in range(8)]
    for a, b in CORNER_ORDER:
        _plot_line_3d(film, zbuf, GRAY, vp, corners[a], corners[b])


# ---- bodies: the sequential depth test of render.go:236-247, all at once ---------------------------------------
def splat_bodies(film, zbuf, vp, pts, colors):
    """Outcome of calling plotpoint3d for the bodies in order.  A body is drawn iff its depth is <= everything
    tested at the same buffer index before it (the lines included), and the last body drawn at a pixel stays."""
    h, w = film.shape[:2]
    front, nx, ny, nz = _project(vp, pts)
    sx = _go_int((nx + 1) * float(w) / 2)
    sy = _go_int((1 - ny) * float(h) / 2)
    with np.errstate(over="ignore"):
        idx = sx + sy * np.int64(w)
    sentinel = np.int64(-(1 << 63))                              # a NaN or overflowing coordinate is not drawn
    ok = front & (sx != sentinel) & (sy != sentinel) & (idx >= 0) & (idx < w * h) & ~np.isnan(nz)   # NaN depth fails `<=`
    order = np.nonzero(ok)[0]
    if order.size == 0:
        return
    idx, z = idx[order], nz[order]
    flat = zbuf.reshape(-1)
    # the depth a buffer cell holds before the bodies goes first in its group
    touched = np.unique(idx)
    gi = np.concatenate([touched, idx])
    gz = np.concatenate([flat[touched], z])
    gseq = np.concatenate([np.full(touched.size, -1, dtype=np.int64), np.arange(order.size, dtype=np.int64)])
    srt = np.lexsort((gseq, gi))
    gi, gz, gseq = gi[srt], gz[srt], gseq[srt]
    # exact comparisons through dense ranks; a running minimum that restarts at every group
    rank = np.unique(gz, return_inverse=True)[1].astype(np.int64)
    group = np.cumsum(np.concatenate([[0], (gi[1:] != gi[:-1]).astype(np.int64)]))
    key = rank - group * np.int64(rank.size + 1)
    passes = (np.minimum.accumulate(key) == key) & (gseq >= 0)
    if not passes.any():
        return
    # new depth of every touched cell: the smallest depth that passed (passing depths only decrease)
    pi, pz, pseq = gi[passes], gz[passes], gseq[passes]
    last = np.concatenate([pi[1:] != pi[:-1], [True]])
    flat[pi[last]] = pz[last]
    # pixels: only bodies inside the image leave colour; the last one drawn at a pixel stays
    body = order[pseq]
    vis = (sx[body] >= 0) & (sx[body] < w) & (sy[body] >= 0) & (sy[body] < h)
    vi, vb = pi[vis], body[vis]
    if vi.size == 0:
        return
    lastv = np.concatenate([vi[1:] != vi[:-1], [True]])
    rgb = film.reshape(-1, 4)
    rgb[vi[lastv], :3] = colors[vb[lastv]]
    rgb[vi[lastv], 3] = 255


# ---- labels --------------------------------------------------------------------------------------------------
_FONT = {  # 5x7 glyphs, seven rows of five cells each, '#' = set
    "0": ".###.#...##..###.#.###..##...#.###.",
    "1": "..#...##....#....#....#....#...###.",
    "2": ".###.#...#....#...#...#...#...#####",
    "3": ".###.#...#....#..##.....##...#.###.",
    "4": "...#...##..#.#.#..#.#####...#....#.",
    "5": "######....####.....#....##...#.###.",
    "6": "..##..#...#....####.#...##...#.###.",
    "7": "#####....#...#...#...#....#....#...",
    "8": ".###.#...##...#.###.#...##...#.###.",
    "9": ".###.#...##...#.####....#...#..##..",
    ".": "..........................##...##..",
    ":": "......##...##........##...##.......",
    " ": "...................................",
    "-": "...............#####...............",
    "a": "...........###.....#.#####...#.####",
    "b": "#....#....#.##.##..##...##...#####.",
    "d": "....#....#.##.##..###...##...#.####",
    "e": "...........###.#...#######.....###.",
    "f": "..##..#..#.#...###...#....#....#...",
    "i": "..#........##....#....#....#...###.",
    "m": "..........##.#.#.#.##.#.##...##...#",
    "o": "...........###.#...##...##...#.###.",
    "r": "..........#.##.##..##....#....#....",
    "s": "...........###.#.....###.....#####.",
    "t": ".#....#...###...#....#....#..#..##.",
    "v": "..........#...##...##...#.#.#...#..",
    "°": ".##..#..#.#..#..##.................",
}


def plot_text(film, x, y, color, label):
    """Top-left anchored label (render.go:336-349 anchors an 8x16 face the same way; cells here are 8 wide too)."""
    h, w = film.shape[:2]
    for n, ch in enumerate(label):
        g = _FONT.get(ch, _FONT[" "])
        for r in range(7):
            for c in range(5):
                if g[r * 5 + c] == "#":
                    px, py = x + n * 8 + 1 + c, y + 5 + r
                    if 0 <= px < w and 0 <= py < h:
                        film[py, px] = color


# ---- PNG -----------------------------------------------------------------------------------------------------
def encode_png(film, level=6):
    """8-bit RGBA, no interlace, filter 0 on every row."""
    h, w = film.shape[:2]
    raw = np.empty((h, 1 + 4 * w), dtype=np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = film.reshape(h, 4 * w)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b""))


def decode_png(data):
    """Reads back what encode_png writes (8-bit RGBA, filter 0): (h, w, 4) uint8."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        if struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + body) & 0xffffffff):
            raise ValueError("bad chunk checksum")
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            if (depth, ctype) != (8, 6):
                raise ValueError("only 8-bit RGBA")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 4 * w)
    if raw[:, 0].any():
        raise ValueError("only filter 0")
    return raw[:, 1:].reshape(h, w, 4).copy()


# ---- one frame -------------------------------------------------------------------------------------------------
def render_frame(frame, xyzm, width=WIDTH, height=HEIGHT, labels=True):
    """The film of frameToImages for one frame: (height, width, 4) uint8.  xyzm: (n, 4) positions and masses, n >= 2
    (the reference indexes Bodies[0] and Bodies[1], render.go:200-201)."""
    b = np.asarray(xyzm, dtype=np.float64).reshape(-1, 4)
    if len(b) < 2:
        raise ValueError("the renderer needs at least two bodies (render.go:200-201)")
    b = b[np.argsort(-b[:, 3], kind="stable")]                  # render.go:195-197, heaviest first
    pts = np.ascontiguousarray(b[:, :3])
    g0, g1 = pts[0].copy(), pts[1].copy()
    cam = Camera(frame, g0, g1, width, height)
    film = np.zeros((height, width, 4), dtype=np.uint8)
    zbuf = np.full((height, width), np.inf)                     # render.go:580-584
    draw_background(film, zbuf, cam.vp)
    if labels:                                                  # render.go:231-234
        plot_text(film, 0, 16, WHITE, "frame:  %d" % frame)
        plot_text(film, 0, 32, WHITE, "bodies: %d" % len(b))
        plot_text(film, 0, 48, WHITE, "rot:    %.1f°" % math.degrees(cam.angle))
        plot_text(film, 0, 64, WHITE, "fov:    %.1f°" % math.degrees(cam.fov))
    splat_bodies(film, zbuf, cam.vp, pts, body_colors(cam.pos, g0, g1, pts))
    return film


class ImageSink:
    """frameToImages as a sink: `workers` threads (main.go:51) take frames from a queue of `backlog` entries and
    write one PNG each into `img_dir`; write() blocks while the queue is full, like the channel send."""

    def __init__(self, img_dir="img", workers=4, backlog=32, width=WIDTH, height=HEIGHT, labels=True,
                 reference_names=False, compression=6):
        self.img_dir = img_dir
        self.width, self.height, self.labels = width, height, labels
        self.reference_names = reference_names
        self.compression = compression
        os.makedirs(img_dir, exist_ok=True)
        self._q = queue.Queue(maxsize=backlog)
        self._error = None
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, workers))]
        for t in self._threads:
            t.start()

    def filename(self, frame):
        name = (FILENAME_FORMAT + str(frame)) if self.reference_names else (FILENAME_FORMAT % frame)
        return os.path.join(self.img_dir, name)

    def _work(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            try:
                frame, xyzm = job
                film = render_frame(frame, xyzm, self.width, self.height, self.labels)
                with open(self.filename(frame), "wb") as f:
                    f.write(encode_png(film, self.compression))
            except Exception as ex:          # reported by the next write() or by close()
                self._error = ex

    def write(self, frame, ids, xyzm, radius=None, default_radius=2.0):
        if self._error is not None:
            raise self._error
        self._q.put((int(frame), np.array(xyzm, dtype=np.float32, copy=True)))

    def close(self):
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        self._threads = []
        if self._error is not None:
            raise self._error

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def read_frame_file(path):
    """One frame as the C++ host writes it (host/nbody_main.cpp, frameToFiles): u32 n, then n records of
    u32 id + fp32 x, y, z, mass, radius.  Returns (ids, xyzm, radius)."""
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:4].view("<u4")[0])
    rec = raw[4:4 + 24 * n].view(np.dtype([("id", "<u4"), ("v", "<f4", (5,))]))
    return rec["id"].copy(), rec["v"][:, :4].copy(), rec["v"][:, 4].copy()


def main(argv=None):
    """python -m nbody_amd.render FRAMES_DIR [IMG_DIR]: renders every NNNNNNNNNN.frame of FRAMES_DIR."""
    import argparse
    import glob
    ap = argparse.ArgumentParser(description="render the frame files of the nbody host into PNG images")
    ap.add_argument("frames_dir")
    ap.add_argument("img_dir", nargs="?", default="img")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--reference-names", action="store_true", help='name files "%%010d.png<frame>" as the reference does')
    args = ap.parse_args(argv)
    files = sorted(glob.glob(os.path.join(args.frames_dir, "*.frame")))
    with ImageSink(args.img_dir, workers=args.workers, reference_names=args.reference_names) as sink:
        for path in files:
            ids, xyzm, radius = read_frame_file(path)
            sink.write(int(os.path.splitext(os.path.basename(path))[0]), ids, xyzm, radius)
    print(f"{len(files)} frames -> {args.img_dir}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
