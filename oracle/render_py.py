"""TEST INFRASTRUCTURE ONLY: a scalar, line-by-line restatement of the reference's PNG renderer (render.go) in
pure Python, used by tests/ to check nbody_amd/render.py pixel for pixel on small scenes.  Nothing in the product
imports it.

Parity unpinned: the reference holds no rendered image and no Go toolchain exists here, so this follows the Go
source and the published definitions of the go-gl/mathgl v1.0.0 functions it calls (go.mod), not an output of the
reference.  Text labels (render.go:231-234, an 8x16 Inconsolata face from golang.org/x/image) are left out.
"""
import math

WHITE = (255, 255, 255, 255)
GRAY = (128, 128, 128, 255)
RED = (255, 0, 0, 255)
GREEN = (0, 255, 0, 255)
BLUE = (0, 0, 255, 255)


# ---- the mathgl pieces the renderer uses ---------------------------------------------------------------------
def v_sub(a, b): return [a[0] - b[0], a[1] - b[1], a[2] - b[2]]
def v_add(a, b): return [a[0] + b[0], a[1] + b[1], a[2] + b[2]]
def v_mul(a, s): return [a[0] * s, a[1] * s, a[2] * s]
def v_dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def v_cross(a, b): return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
def v_len(a): return math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def v_normalize(a):
    """mgl64 Vec3.Normalize: multiply by 1/len (a zero vector gives inf * 0 = NaN, as in Go)."""
    n = v_len(a)
    l = math.inf if n == 0.0 else 1.0 / n
    return [_mul(a[0], l), _mul(a[1], l), _mul(a[2], l)]


def _mul(x, y):
    return x * y               # 0 * inf is nan in Python as in Go


def quat_rotate(angle, axis, v):
    """mgl64.QuatRotate(angle, axis).Rotate(v)."""
    w = math.cos(angle / 2)
    s = math.sin(angle / 2)
    qv = v_mul(axis, s)
    cross = v_cross(qv, v)
    return v_add(v_add(v, v_mul(cross, 2 * w)), v_cross(v_mul(qv, 2), cross))


def mat_mul(a, b):
    """4x4 product of column-major matrices (mgl64 Mat4.Mul4)."""
    out = [0.0] * 16
    for c in range(4):
        for r in range(4):
            out[c * 4 + r] = sum(a[k * 4 + r] * b[c * 4 + k] for k in range(4))
    return out


def mat_vec(m, v):
    return [m[0] * v[0] + m[4] * v[1] + m[8] * v[2] + m[12] * v[3],
            m[1] * v[0] + m[5] * v[1] + m[9] * v[2] + m[13] * v[3],
            m[2] * v[0] + m[6] * v[1] + m[10] * v[2] + m[14] * v[3],
            m[3] * v[0] + m[7] * v[1] + m[11] * v[2] + m[15] * v[3]]


def look_at(eye, center, up):
    f = v_normalize(v_sub(center, eye))
    s = v_normalize(v_cross(f, v_normalize(up)))
    u = v_cross(s, f)
    m = [s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0, 0, 0, 0, 1]
    t = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -eye[0], -eye[1], -eye[2], 1]
    return mat_mul(m, t)


def perspective(fovy, aspect, near, far):
    nmf = near - far
    f = 1.0 / math.tan(fovy / 2.0)
    return [f / aspect, 0, 0, 0, 0, f, 0, 0, 0, 0, (near + far) / nmf, -1, 0, 0, (2.0 * far * near) / nmf, 0]


def go_int(x):
    """Go's int(float64) for the values met here: truncation toward zero; NaN and out-of-range give the minimum
    int64 on amd64."""
    if x != x or abs(x) >= 9.2e18:
        return -(1 << 63)
    return int(x)


def gl_to_screen(x, y, w, h):
    return go_int((x + 1) * float(w) / 2), go_int((1 - y) * float(h) / 2)


def go_uint8(x):
    """uint8(float64) after the clamp of render.go:82-84; NaN passes the clamp and converts to 0 on amd64."""
    if x != x:
        return 0
    return int(x) & 0xff


# ---- render.go -----------------------------------------------------------------------------------------------
def fractilum(dir_to_cam, dir_to_light):                      # render.go:67-69
    return (v_dot(dir_to_cam, dir_to_light) + 1.0) / 2.0


def light_lo(light, p, dir_to_cam):                           # render.go:52-61
    if light["directional"]:
        frac = fractilum(dir_to_cam, light["pos"])
        return v_mul(light["color"], frac)
    lp = v_sub(light["pos"], p)
    frac = fractilum(dir_to_cam, v_normalize(lp))
    lensqr = v_dot(lp, lp)
    k = math.pi * light["intensity"] * frac
    k = (math.nan if (k != k or k == 0.0) else math.copysign(math.inf, k)) if lensqr == 0.0 else k / lensqr
    return [_mul(c, k) for c in light["color"]]


def clamp(a, lo, hi):                                         # mgl64.Clamp
    if a < lo:
        return lo
    if a > hi:
        return hi
    return a


def sum_lo(lights, cam, p, c):                                # render.go:74-91
    total = [0.0, 0.0, 0.0]
    dir_to_cam = v_normalize(v_sub(cam, p))
    for l in lights:
        total = v_add(total, light_lo(l, p, dir_to_cam))
    return (go_uint8(clamp(total[0] * float(c[0]), 0.0, 255.0)),
            go_uint8(clamp(total[1] * float(c[1]), 0.0, 255.0)),
            go_uint8(clamp(total[2] * float(c[2]), 0.0, 255.0)), 255)


class Film:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.pix = [[(0, 0, 0, 255)] * w for _ in range(h)]
        self.z = [math.inf] * (w * h)

    def set(self, x, y, c):                                   # image.RGBA.Set ignores points outside the image
        if 0 <= x < self.w and 0 <= y < self.h:
            self.pix[y][x] = c


def plot_point_3d(film, c, vp, p):                            # render.go:430-446
    t = mat_vec(vp, [p[0], p[1], p[2], 1.0])
    if t[3] <= 0:
        return
    inv = 1 / t[3]
    t = [t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv]
    x, y = gl_to_screen(t[0], t[1], film.w, film.h)
    i = x + y * film.w                                        # render.go:576-578: may name a pixel of another row
    if 0 <= i < len(film.z) and t[2] <= film.z[i]:
        film.set(x, y, c)
        film.z[i] = t[2]


def inv_lerp(x, lo, hi):
    return (x - lo) / (hi - lo)


def clip_w_to_0(behind, front):                               # render.go:529-536
    t = inv_lerp(0, behind[3], front[3])
    behind[0] += t * (front[0] - behind[0])
    behind[1] += t * (front[1] - behind[1])
    behind[2] += t * (front[2] - behind[2])
    behind[3] = 1


PLANES = [([1, 0, 0], [-1, 0, 0]), ([-1, 0, 0], [1, 0, 0]), ([0, 1, 0], [0, -1, 0]), ([0, -1, 0], [0, 1, 0])]


def clip(p1, p2):                                             # render.go:550-574; p1, p2 are modified in place
    for n, p in PLANES:
        p1behind = v_dot(v_sub(p1, p), n) < 0
        p2behind = v_dot(v_sub(p2, p), n) < 0
        if p1behind and p2behind:
            return True
        if p1behind:
            dline = v_sub(p2, p1)
            dplane = v_sub(p, p1)
            p1[:] = v_add(p1, v_mul(dline, v_dot(dplane, n) / v_dot(dline, n)))
        elif p2behind:
            dline = v_sub(p1, p2)
            dplane = v_sub(p, p2)
            p2[:] = v_add(p2, v_mul(dline, v_dot(dplane, n) / v_dot(dline, n)))
    return False


def dist_int(x0, y0, x1, y1):
    dx = float(x1 - x0)
    dy = float(y1 - y0)
    return math.sqrt(dx * dx + dy * dy)


def plot_line_zbuff(film, c, x0, y0, x1, y1, z0, z1):         # render.go:480-521
    x, y, z = x0, y0, z0
    full = dist_int(x0, y0, x1, y1)
    dx = abs(x1 - x0)
    sx = 1 if x0 < x1 else -1
    dy = -abs(y1 - y0)
    sy = 1 if y0 < y1 else -1
    err = dx + dy
    while True:
        i = x0 + y0 * film.w
        if 0 <= i < len(film.z) and z0 <= film.z[i]:
            film.set(x0, y0, c)
            film.z[i] = z0
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy
        cur = dist_int(x, y, x0, y0)
        t = cur / full
        z0 = z + (z1 - z) * t


def plot_line_3d(film, c, vp, p1, p2):                        # render.go:449-478
    t1 = mat_vec(vp, [p1[0], p1[1], p1[2], 1.0])
    t2 = mat_vec(vp, [p2[0], p2[1], p2[2], 1.0])
    if t1[3] <= 0 and t2[3] <= 0:
        return
    if t1[3] <= 0:
        clip_w_to_0(t1, t2)
    elif t2[3] <= 0:
        clip_w_to_0(t2, t1)
    a = [t1[0] * (1 / t1[3]), t1[1] * (1 / t1[3]), t1[2] * (1 / t1[3])]
    b = [t2[0] * (1 / t2[3]), t2[1] * (1 / t2[3]), t2[2] * (1 / t2[3])]
    if clip(a, b):
        return
    x1, y1 = gl_to_screen(a[0], a[1], film.w, film.h)
    x2, y2 = gl_to_screen(b[0], b[1], film.w, film.h)
    plot_line_zbuff(film, c, x1, y1, x2, y2, a[2], b[2])


def bound_corners(width):                                     # tree.go:46-57 for a cube centred on the origin
    h = 0.5 * width
    return [[h * (((i & 1) >> 0) * 2 - 1.0), h * (((i & 2) >> 1) * 2 - 1.0), h * (((i & 4) >> 2) * 2 - 1.0)] for i in range(8)]


CORNER_ORDER = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 4), (1, 5), (2, 6), (3, 7)]


def render_frame(frame, bodies, width=1920, height=1080):
    """frameToImages (render.go:97-259) for one frame.  bodies: list of (x, y, z, mass), at least two.
    Returns (rows of RGBA tuples, camera dict)."""
    cam_radius = float(1 << 15)
    axis_length = cam_radius / 10.0
    sim_width = float(1 << 16)
    intensity = 1e6
    lights = [
        {"directional": False, "pos": [axis_length, 0, 0], "color": [1.0, 0.001, 0.001], "intensity": intensity},
        {"directional": False, "pos": [0, axis_length, 0], "color": [0.001, 1.0, 0.001], "intensity": intensity},
        {"directional": False, "pos": [0, 0, axis_length], "color": [0.001, 0.001, 1.0], "intensity": intensity},
    ]
    start_cam = v_mul(v_normalize([1.0, 1.0, 5.0]), cam_radius)
    # render.go:195-197 sorts by mass, heaviest first (Go's sort is not stable; a stable sort is one of its outcomes)
    bodies = sorted(([float(v) for v in b] for b in bodies), key=lambda b: -b[3])
    film = Film(width, height)
    g0 = list(bodies[0][:3])
    g1 = list(bodies[1][:3])
    middle = v_mul(v_add(g0, g1), 0.5)
    angle = (float(frame) * math.pi / 180.0) / 4.0
    cam_pos = quat_rotate(-angle, [0.0, 1.0, 0.0], start_cam)
    fov = 2 * math.atan(0.5 * (v_len(v_sub(g0, g1)) * 1.5) / v_len(v_sub(cam_pos, middle)))
    vp = mat_mul(perspective(fov, float(width) / float(height), 0.1, 100), look_at(cam_pos, middle, [0.0, 1.0, 0.0]))
    lights.append({"directional": False, "pos": g0, "color": [0.9, 0.9, 0.5], "intensity": 0.075 * intensity})
    lights.append({"directional": False, "pos": g1, "color": [0.9, 0.9, 0.5], "intensity": 0.075 * intensity})
    lights.append({"directional": True, "pos": v_normalize(cam_pos), "color": [0.1, 0.1, 0.1], "intensity": 0.0})
    # drawBG (render.go:149-175)
    zero = [0.0, 0.0, 0.0]
    plot_line_3d(film, RED, vp, zero, [axis_length, 0, 0])
    plot_line_3d(film, GREEN, vp, zero, [0, axis_length, 0])
    plot_line_3d(film, BLUE, vp, zero, [0, 0, axis_length])
    corners = bound_corners(sim_width)
    for a, b in CORNER_ORDER:
        plot_line_3d(film, GRAY, vp, corners[a], corners[b])
    for b in bodies:
        world = [b[0], b[1], b[2]]
        plot_point_3d(film, sum_lo(lights, cam_pos, world, WHITE), vp, world)
    return film.pix, {"cam_pos": cam_pos, "fov": fov, "angle": angle, "vp": vp, "middle": middle}
