"""Pure-Python restatement of the reference's force + integrator step, written independently of
nbody_oracle.c straight from the Go source, for cross-checking the C restatement on small inputs.

Test infrastructure only.  PARITY UNPINNED BY THE REFERENCE.  Python floats are IEEE binary64 and
CPython never fuses a*b+c, so with the same operation order the two restatements agree bit for bit.
Citations are file:line in the reference tree.
"""
import math
import struct

G = 6.67408e-11  # physics.go:132


class Body:
    """physics.go:91-98"""
    __slots__ = ("ID", "Mass", "Radius", "X", "Y", "Z", "Vx", "Vy", "Vz", "fx", "fy", "fz")

    def __init__(self, ID=0, Mass=0.0, Radius=0.0, X=0.0, Y=0.0, Z=0.0, Vx=0.0, Vy=0.0, Vz=0.0):
        self.ID, self.Mass, self.Radius = ID, Mass, Radius
        self.X, self.Y, self.Z = X, Y, Z
        self.Vx, self.Vy, self.Vz = Vx, Vy, Vz
        self.fx = self.fy = self.fz = 0.0

    def update(self, dt):
        """physics.go:101-115"""
        self.Vx += self.fx / self.Mass * dt
        self.Vy += self.fy / self.Mass * dt
        self.Vz += self.fz / self.Mass * dt
        self.X += self.Vx * dt
        self.Y += self.Vy * dt
        self.Z += self.Vz * dt
        self.fx, self.fy, self.fz = 0.0, 0.0, 0.0


def _div(a, b):
    """Go float division: x/0 is +-Inf or NaN, never an exception."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -math.inf if neg else math.inf


def dist(a, b):
    """physics.go:123-128"""
    dx = b.X - a.X
    dy = b.Y - a.Y
    dz = b.Z - a.Z
    return math.sqrt(dx * dx + dy * dy + dz * dz)


def gravity(r, a, b):
    """physics.go:135-145"""
    f = _div(G * (a.Mass * b.Mass), (r * r))
    dfx = _div(b.X - a.X, r) * f
    dfy = _div(b.Y - a.Y, r) * f
    dfz = _div(b.Z - a.Z, r) * f
    a.fx += dfx
    a.fy += dfy
    a.fz += dfz


def inelastic_collision(ma, va, mb, vb):
    """physics.go:148-150"""
    return (ma * va + mb * vb) / (ma + mb)


def _cbrt(x):
    """math.Cbrt for x >= 0, correctly rounded from a Newton-polished power (C's cbrt agrees to the last bit on
    the values the tests use; the tests compare radii with a one-ulp tolerance)."""
    if x == 0.0:
        return 0.0
    y = x ** (1.0 / 3.0)
    y = y - (y * y * y - x) / (3.0 * y * y)
    return y


def combine(a, b):
    """physics.go:153-162: a.Mass is overwritten first and then used as `ma`."""
    a.Mass = a.Mass + b.Mass
    a.Radius = _cbrt(a.Radius * a.Radius * a.Radius + b.Radius * b.Radius * b.Radius)
    a.Vx = inelastic_collision(a.Mass, a.Vx, b.Mass, b.Vx)
    a.Vy = inelastic_collision(a.Mass, a.Vy, b.Mass, b.Vy)
    a.Vz = inelastic_collision(a.Mass, a.Vz, b.Mass, b.Vz)
    a.fx = a.fx + b.fx
    a.fy = a.fy + b.fy
    a.fz = a.fz + b.fz


def direct_pass_collide(bodies):
    """main.go:107-128 with collisions on; returns the number of merges."""
    n = len(bodies)
    merges = 0
    for i in range(n - 1):
        if bodies[i] is None:
            continue
        for j in range(i + 1, n):
            if bodies[j] is None:
                continue
            r = dist(bodies[i], bodies[j])
            gravity(r, bodies[i], bodies[j])
            gravity(r, bodies[j], bodies[i])
            if r <= bodies[i].Radius + bodies[j].Radius:
                combine(bodies[i], bodies[j])
                bodies[j] = None
                merges += 1
    return merges


def direct_pass(bodies):
    """main.go:107-128 with -nocollision"""
    n = len(bodies)
    for i in range(n - 1):
        if bodies[i] is None:
            continue
        for j in range(i + 1, n):
            if bodies[j] is None:
                continue
            r = dist(bodies[i], bodies[j])
            gravity(r, bodies[i], bodies[j])
            gravity(r, bodies[j], bodies[i])


def _signbit_clear(d):
    return (~struct.unpack("<Q", struct.pack("<d", d))[0] >> 63) & 1


def octant_bits(mid, p):
    """tree.go:128-132"""
    return _signbit_clear(p[0] - mid[0]) | (_signbit_clear(p[1] - mid[1]) << 1) | (_signbit_clear(p[2] - mid[2]) << 2)


class Bound:
    """tree.go:41-43"""
    __slots__ = ("center", "width")

    def __init__(self, center, width):
        self.center, self.width = tuple(center), tuple(width)

    def max(self):
        """tree.go:60-62"""
        return max(self.width[0], max(self.width[1], self.width[2]))

    def contains(self, p):
        """tree.go:65-72"""
        c, w = self.center, self.width
        h = (w[0] * 0.5, w[1] * 0.5, w[2] * 0.5)
        return ((c[0] - h[0] <= p[0] and p[0] < c[0] + h[0]) and
                (c[1] - h[1] <= p[1] and p[1] < c[1] + h[1]) and
                (c[2] - h[2] <= p[2] and p[2] < c[2] + h[2]))


def contains_sphere(bound, p, r):
    """tree.go:75-79"""
    c, w = bound.center, bound.width
    return (abs(p[0] - c[0]) <= w[0] - r) and (abs(p[1] - c[1]) <= w[1] - r) and (abs(p[2] - c[2]) <= w[2] - r)


def intersect_sphere(bound, p, r):
    """tree.go:82-86"""
    c, w = bound.center, bound.width
    return (abs(p[0] - c[0]) <= w[0] + r) or (abs(p[1] - c[1]) <= w[1] + r) or (abs(p[2] - c[2]) <= w[2] + r)


def octant_bound(parent, oct_):
    """tree.go:101-110"""
    w = parent.width
    tx = (w[0] * 0.25 * (float((oct_ & 1) * 2) - 1.0),
          w[1] * 0.25 * (float(((oct_ & 2) >> 1) * 2) - 1.0),
          w[2] * 0.25 * (float(((oct_ & 4) >> 2) * 2) - 1.0))
    nw = (w[0] * 0.5, w[1] * 0.5, w[2] * 0.5)
    c = parent.center
    return Bound((c[0] + tx[0], c[1] + tx[1], c[2] + tx[2]), nw)


EXTERNAL, INTERNAL = 0, 1


class Node:
    """tree.go:134-141; `particle` is an index into the body list (the Go code holds a **body)."""
    __slots__ = ("kind", "children", "particle", "totalMass", "centerOfMass", "bounds")

    def __init__(self, bounds):
        self.kind = EXTERNAL
        self.children = None
        self.particle = None
        self.totalMass = 0.0
        self.centerOfMass = (0.0, 0.0, 0.0)
        self.bounds = bounds

    def push(self, bodies, bi, true_com=False):
        """tree.go:145-197"""
        b = bodies[bi]
        point = (b.X, b.Y, b.Z)
        if not self.bounds.contains(point):
            return False
        if self.kind == EXTERNAL:
            if self.particle is None:
                self.particle = bi
                self.centerOfMass = point
                self.totalMass = b.Mass
                return True
            self.children = [None] * 8
            oct_ = octant_bits(self.bounds.center, self.centerOfMass)
            if self.children[oct_] is None:
                self.children[oct_] = Node(octant_bound(self.bounds, oct_))
            self.children[oct_].push(bodies, self.particle, true_com)
            self.particle = None
            self.kind = INTERNAL
        oct_ = octant_bits(self.bounds.center, point)
        if self.children[oct_] is None:
            self.children[oct_] = Node(octant_bound(self.bounds, oct_))
        if self.children[oct_].push(bodies, bi, true_com):
            if not true_com:
                self.totalMass += b.Mass
                s = b.Mass / self.totalMass
                c = self.centerOfMass
                self.centerOfMass = (c[0] + point[0] * s, c[1] + point[1] * s, c[2] + point[2] * s)
            else:
                mold = self.totalMass
                mnew = mold + b.Mass
                c = self.centerOfMass
                self.centerOfMass = ((c[0] * mold + point[0] * b.Mass) / mnew,
                                     (c[1] * mold + point[1] * b.Mass) / mnew,
                                     (c[2] * mold + point[2] * b.Mass) / mnew)
                self.totalMass = mnew
        return True

    def gravity(self, bodies, bi, theta):
        """tree.go:202-240; returns the number of force evaluations."""
        if self.totalMass == 0:
            return 0
        if self.particle is not None and self.particle == bi:
            return 0
        b = bodies[bi]
        c = self.centerOfMass
        d = (c[0] - b.X, c[1] - b.Y, c[2] - b.Z)
        r = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        if self.kind == INTERNAL:
            criterion = _div(self.bounds.max(), r)
            if criterion > theta:
                cnt = 0
                for i in range(8):
                    if self.children[i] is not None:
                        cnt += self.children[i].gravity(bodies, bi, theta)
                return cnt
        f = _div(G * (self.totalMass * b.Mass), (r * r))
        b.fx += _div(c[0] - b.X, r) * f
        b.fy += _div(c[1] - b.Y, r) * f
        b.fz += _div(c[2] - b.Z, r) * f
        return 1


def check_collision(node, bodies, bi, queue):
    """tree.go:245-281"""
    if node.particle is not None and node.particle == bi:
        return
    b = bodies[bi]
    p = (b.X, b.Y, b.Z)
    if not intersect_sphere(node.bounds, p, b.Radius):
        return
    if node.kind == EXTERNAL:
        if node.particle is not None and bodies[node.particle] is not None:
            o = bodies[node.particle]
            if dist(o, b) <= o.Radius + b.Radius:
                queue.append((bi, node.particle))
        return
    oct_ = octant_bits(node.bounds.center, p)
    ch = node.children[oct_]
    if ch is not None and contains_sphere(ch.bounds, p, b.Radius):
        check_collision(ch, bodies, bi, queue)
    else:
        for c in node.children:
            if c is not None:
                check_collision(c, bodies, bi, queue)


def tree_collisions(root, bodies):
    """main.go:152-167; returns (queue, merges)."""
    queue = []
    for i in range(len(bodies)):
        if bodies[i] is None:
            continue
        check_collision(root, bodies, i, queue)
    merges = 0
    for a, b in queue:
        if bodies[a] is None or bodies[b] is None:
            continue
        combine(bodies[a], bodies[b])
        bodies[b] = None
        merges += 1
    return queue, merges


def maketree(bodies, simbound, true_com=False):
    """tree.go:284-295"""
    root = Node(simbound)
    for i in range(len(bodies)):
        if bodies[i] is None:
            continue
        if not root.push(bodies, i, true_com):
            bodies[i] = None
    return root


def tree_pass(bodies, theta=1.0, bound_center=(0.0, 0.0, 0.0), bound_width=65536.0, true_com=False,
              groups=4, drop_tail=True):
    """main.go:133-149: build, then walk the quarters (serially here; the groups write disjoint bodies)."""
    root = maketree(bodies, Bound(bound_center, (bound_width,) * 3), true_com)
    n = len(bodies)
    groupsize = n // groups
    count = 0
    for g in range(groups):
        lo, hi = g * groupsize, (g + 1) * groupsize
        if not drop_tail and g == groups - 1:
            hi = n
        for i in range(lo, hi):
            if bodies[i] is None:
                continue
            count += root.gravity(bodies, i, theta)
    return root, count


def simulate(bodies, nsteps, tree=True, dt=3600.0, collide=False, **kw):
    """main.go:104-178 repeated nsteps times; collide=False is the -nocollision run."""
    for _ in range(nsteps):
        if tree:
            root, _ = tree_pass(bodies, **kw)
            if collide:
                tree_collisions(root, bodies)
        elif collide:
            direct_pass_collide(bodies)
        else:
            direct_pass(bodies)
        for b in bodies:
            if b is not None:
                b.update(dt)
    return bodies


def count_nodes(node):
    """Number of allocated nodes below and including `node` (to compare with the C arena count)."""
    if node is None:
        return 0
    k = 1
    if node.children is not None:
        for c in node.children:
            k += count_nodes(c)
    return k
