"""Host-side mirror of the reference's body/physics interface (physics.go) and seeded initial conditions.

The names follow the reference: `body`, `makebodies`, `cross`, `dist`, `gravity`, `update`, `G`.  The
per-body helpers are scalar conveniences for hosts and tests; the simulation step itself always runs on
the GPU through `Simulation` / `nbody_amd.Engine` and never through these functions.

The reference draws from Go's unseeded global math/rand stream (main.go:16 leaves the seed commented out),
which cannot be reproduced without Go; every generator here takes a numpy seed instead and reproduces the
reference's distributions, not its exact stream.
"""
import math
from dataclasses import dataclass

import numpy as np

from .engine import Engine

G = 6.67408e-11          # physics.go:132
DT = 3600.0              # main.go:28
THETA = 1.0              # main.go:73
BOUND_WIDTH = 65536.0    # main.go:74-76 (0x1p16)


@dataclass
class body:
    """physics.go:91-98"""
    ID: int = 0
    Mass: float = 0.0
    Radius: float = 0.0
    X: float = 0.0
    Y: float = 0.0
    Z: float = 0.0
    Vx: float = 0.0
    Vy: float = 0.0
    Vz: float = 0.0
    fx: float = 0.0
    fy: float = 0.0
    fz: float = 0.0

    def update(self, dt):
        """physics.go:101-115"""
        self.Vx += self.fx / self.Mass * dt
        self.Vy += self.fy / self.Mass * dt
        self.Vz += self.fz / self.Mass * dt
        self.X += self.Vx * dt
        self.Y += self.Vy * dt
        self.Z += self.Vz * dt
        self.fx, self.fy, self.fz = 0.0, 0.0, 0.0


def cross(x1, y1, z1, x2, y2, z2):
    """physics.go:76-81"""
    return (y1 * z2 - z1 * y2, z1 * x2 - x1 * z2, x1 * y2 - y1 * x2)


def dist(a, b):
    """physics.go:123-128"""
    dx, dy, dz = b.X - a.X, b.Y - a.Y, b.Z - a.Z
    return math.sqrt(dx * dx + dy * dy + dz * dz)


def gravity(r, a, b):
    """physics.go:135-145: adds to a the force on a from b."""
    f = G * (a.Mass * b.Mass) / (r * r)
    a.fx += (b.X - a.X) / r * f
    a.fy += (b.Y - a.Y) / r * f
    a.fz += (b.Z - a.Z) / r * f


def default_cores():
    """The two galaxy cores of main.go:35-36; fx,fy,fz carry the rotation axis as in the reference."""
    return [body(Mass=1e10, Radius=1.0, X=-9000, Y=-100, Z=-2000, fy=1, Vx=0.004, Vz=-0.001),
            body(Mass=1e10, Radius=1.0, X=9000, Y=100, Z=2000, fz=-1, Vx=-0.003, Vz=0.002)]


class Bodies:
    """Structure-of-arrays body set: the flat form that crosses the C ABI."""

    def __init__(self, pos, vel, mass, ids=None, radius=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        self.vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        self.mass = np.ascontiguousarray(mass, dtype=np.float64).reshape(-1)
        n = self.mass.shape[0]
        self.ids = np.arange(n, dtype=np.uint64) if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
        self.radius = np.full(n, 2.0) if radius is None else np.ascontiguousarray(radius, dtype=np.float64)

    def __len__(self):
        return self.mass.shape[0]

    def to_list(self):
        return [body(int(self.ids[i]), float(self.mass[i]), float(self.radius[i]), *map(float, self.pos[i]),
                     *map(float, self.vel[i])) for i in range(len(self))]


def makebodies(n, cores=None, seed=1):
    """physics.go:16-73: n bodies scattered in Gaussian discs around the cores on circular orbits, with
    the cores placed at the front.  Returns a `Bodies`."""
    orbital_v_dampening = 1.0
    mean_mass = 1e3
    default_radius = 2.0
    radial_sd, axial_sd = 1e3, 1e1
    cores = default_cores() if cores is None else cores
    nc = len(cores)
    rng = np.random.default_rng(seed)
    tot = n + nc
    pos = np.zeros((tot, 3)); vel = np.zeros((tot, 3)); mass = np.zeros(tot); radius = np.full(tot, default_radius)
    for i in range(nc, tot):
        mass[i] = abs(rng.standard_normal() * 500 + mean_mass)
        core = body()
        if nc > 0:
            core = cores[int(rng.integers(nc))]
        axis = (core.fx, core.fy, core.fz)
        centre = (core.X, core.Y, core.Z)
        for k in range(3):
            sd = radial_sd * (1 - abs(axis[k])) + axial_sd * abs(axis[k])
            pos[i, k] = rng.standard_normal() * sd + centre[k]
        if nc > 0:
            dx, dy, dz = centre[0] - pos[i, 0], centre[1] - pos[i, 1], centre[2] - pos[i, 2]
            d = math.sqrt(dx * dx + dy * dy + dz * dz)
            if d == 0:
                d = 1
            dx, dy, dz = dx / d, dy / d, dz / d
            dx, dy, dz = cross(dx, dy, dz, *axis)
            v = math.sqrt(G * core.Mass / d)
            vel[i] = (dx * v * orbital_v_dampening + core.Vx, dy * v * orbital_v_dampening + core.Vy,
                      dz * v * orbital_v_dampening + core.Vz)
    for i, c in enumerate(cores):
        pos[i] = (c.X, c.Y, c.Z); vel[i] = (c.Vx, c.Vy, c.Vz); mass[i] = c.Mass; radius[i] = c.Radius
    return Bodies(pos, vel, mass, radius=radius)


def uniform_cube(n, half=1e4, mass=1e3, seed=1):
    """n equal-mass bodies at rest, uniform in [-half, half)^3."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-half, half, size=(n, 3))
    return Bodies(pos, np.zeros((n, 3)), np.full(n, mass))


def plummer(n, a=2e3, total_mass=1e9, rmax_over_a=12.0, seed=1, centre=(0.0, 0.0, 0.0), bulk_velocity=(0.0, 0.0, 0.0)):
    """Plummer sphere by Aarseth-Henon-Wielen sampling: scale radius a (m), equal masses summing to
    total_mass (kg), radii truncated at rmax_over_a * a, isotropic velocities from the distribution function."""
    rng = np.random.default_rng(seed)
    r = np.empty(0)
    while r.shape[0] < n:
        x = rng.uniform(1e-12, 1.0, size=n)
        cand = a / np.sqrt(x ** (-2.0 / 3.0) - 1.0)
        r = np.concatenate([r, cand[cand < rmax_over_a * a]])
    r = r[:n]

    def iso(m):
        c = rng.uniform(-1.0, 1.0, size=m)
        phi = rng.uniform(0.0, 2 * np.pi, size=m)
        s = np.sqrt(1 - c * c)
        return np.stack([s * np.cos(phi), s * np.sin(phi), c], axis=1)

    pos = iso(n) * r[:, None]
    q = np.empty(0)
    while q.shape[0] < n:
        x = rng.uniform(0.0, 1.0, size=n)
        y = rng.uniform(0.0, 0.1, size=n)
        q = np.concatenate([q, x[y < x * x * (1 - x * x) ** 3.5]])
    q = q[:n]
    vesc = np.sqrt(2 * G * total_mass) * (r * r + a * a) ** (-0.25)
    vel = iso(n) * (q * vesc)[:, None]
    pos += np.asarray(centre, dtype=np.float64)
    vel += np.asarray(bulk_velocity, dtype=np.float64)
    return Bodies(pos, vel, np.full(n, total_mass / n))


def galaxy_pair(n_each, a=1.5e3, total_mass=1e9, seed=1):
    """Two Plummer spheres at the positions and with the approach velocities of the reference's cores
    (main.go:35-36)."""
    c0, c1 = default_cores()
    g0 = plummer(n_each, a, total_mass, seed=seed, centre=(c0.X, c0.Y, c0.Z), bulk_velocity=(c0.Vx, c0.Vy, c0.Vz))
    g1 = plummer(n_each, a, total_mass, seed=seed + 1, centre=(c1.X, c1.Y, c1.Z), bulk_velocity=(c1.Vx, c1.Vy, c1.Vz))
    return Bodies(np.concatenate([g0.pos, g1.pos]), np.concatenate([g0.vel, g1.vel]), np.concatenate([g0.mass, g1.mass]))


class Simulation:
    """The reference's frame loop seam (main.go:80-178) on the GPU engine: each `step` is one iteration of
    main.go:104-178 (collisions off); `frame()` is the copy of live bodies taken at main.go:90-101."""

    def __init__(self, bodies, tree=False, theta=THETA, dt=DT, eps=0.0, precision="f32", device=0,
                 bound_width=BOUND_WIDTH, nocollision=True, asgo=False, per_wave=False):
        """asgo: the Go program's own tree arithmetic (running cell centres tree.go:189-193, last n mod 4 bodies
        unwalked main.go:134-148); per_wave: the bodies of a wavefront open cells together (faster, finer than the
        reference's test, not its interaction set).  Both apply to tree mode only."""
        self.bodies = bodies
        self.collisions = not nocollision                     # main.go:24; tree mode only
        self.engine = Engine(mode="tree" if tree else "direct", precision=precision, theta=theta, dt=dt, eps=eps,
                             bound_width=bound_width, device=device, collisions=self.collisions,
                             ref_com=bool(asgo and tree), drop_tail=bool(asgo and tree),
                             per_wave=bool(per_wave and tree and not asgo))
        self.engine.upload(bodies.pos, bodies.vel, bodies.mass, bodies.ids)
        if self.collisions:
            self.engine.set_radius(bodies.radius)
        self.frame_index = 0

    def step(self, nsteps=1):
        self.engine.step(nsteps)
        self.frame_index += nsteps

    def frame(self):
        pos, vel, mass, ids = self.engine.download()
        if self.collisions:
            return Bodies(pos, vel, mass, ids, self.engine.get_radius())
        radius_of = dict(zip(self.bodies.ids.tolist(), self.bodies.radius.tolist()))
        return Bodies(pos, vel, mass, ids, [radius_of[int(i)] for i in ids])

    def run(self, frames, sink, iter_per_frame=1, start_frame=0, slots=32):
        """The frame loop of main.go:80-190 with a sink behind the frame channel: frames+1 iterations, each
        sending the state BEFORE its step (main.go:90-101) to `sink.write(frame, ids, xyzm, radius)` from the
        engine's delivery thread, then stepping iter_per_frame times (main.go:104)."""
        radius_of = dict(zip(self.bodies.ids.tolist(), self.bodies.radius.tolist()))

        def on_frame(frame, xyzm, ids, radius):
            if radius is None:
                radius = np.array([radius_of[int(i)] for i in ids])
            sink.write(frame, ids, xyzm, radius)

        self.engine.frame_stream_begin(on_frame, every=iter_per_frame, slots=slots, first_frame=start_frame)
        try:
            for _ in range(start_frame, frames + 1):
                self.step(iter_per_frame)
        finally:
            self.engine.frame_stream_end()

    def save(self, directory="."):
        """exportFrameData (main.go:199-201, 210-224): the live bodies as a gob `frameJob` in `%010d.data`."""
        from . import gobfile
        f = self.frame()
        return gobfile.export_frame_data(self.frame_index, f.ids, f.mass, f.radius, f.pos, f.vel, directory)

    @classmethod
    def resume(cls, filename, **kw):
        """importFrameData + overwriteBodies (main.go:42-47, 226-248): a simulation that continues at the saved
        frame number."""
        from . import gobfile
        frame, ids, mass, radius, pos, vel = gobfile.import_frame_data(filename)
        sim = cls(Bodies(pos, vel, mass, ids, radius), **kw)
        sim.frame_index = frame
        return sim

    def close(self):
        self.engine.close()
