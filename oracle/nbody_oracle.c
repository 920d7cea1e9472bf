/*
 * nbody_oracle.c -- CPU restatement (C, IEEE fp64, no FMA contraction) of the force-evaluation +
 * integrator step of the reference n-body program.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped engine links, loads or calls this file; it is the
 * checker that the parity tests, the smoke test and the benchmark's CPU-baseline leg compare against.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no tests, golden vectors or fixtures, and no
 * Go toolchain exists where this was written, so this restatement has never been compared with output
 * of the real program.  It follows the Go source statement by statement (citations below, relative to
 * the reference tree) and is cross-checked bit for bit against an independently written pure-Python
 * restatement (oracle/oracle_py.py) plus hand-derived known answers.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread (Go on amd64 emits no fused multiply-adds).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* physics.go:132 */
#define ORC_G 6.67408e-11

/* physics.go:91-98.  The reference keeps []*body with nil = deleted; here `alive` plays that part. */
typedef struct {
    uint64_t ID;
    double Mass, Radius;
    double X, Y, Z;
    double Vx, Vy, Vz;
    double fx, fy, fz;
    int alive;
} orc_body;

/* physics.go:101-115 */
static void body_update(orc_body *b, double dt)
{
    b->Vx += b->fx / b->Mass * dt;
    b->Vy += b->fy / b->Mass * dt;
    b->Vz += b->fz / b->Mass * dt;
    b->X += b->Vx * dt;
    b->Y += b->Vy * dt;
    b->Z += b->Vz * dt;
    b->fx = 0; b->fy = 0; b->fz = 0;
}

/* physics.go:123-128 */
static double body_dist(const orc_body *a, const orc_body *b)
{
    double dx = b->X - a->X;
    double dy = b->Y - a->Y;
    double dz = b->Z - a->Z;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

/* physics.go:135-145: adds to a the force on a from b. */
static void body_gravity(double r, orc_body *a, const orc_body *b)
{
    double f = ORC_G * (a->Mass * b->Mass) / (r * r);
    double dfx = (b->X - a->X) / r * f;
    double dfy = (b->Y - a->Y) / r * f;
    double dfz = (b->Z - a->Z) / r * f;
    a->fx += dfx;
    a->fy += dfy;
    a->fz += dfz;
}

/* main.go:107-128 with collisions off: serial i<j loop, both directions per pair. */
static void direct_forces(orc_body *bodies, long n)
{
    for (long i = 0; i < n - 1; i++) {
        if (!bodies[i].alive) continue;
        for (long j = i + 1; j < n; j++) {
            if (!bodies[j].alive) continue;
            double r = body_dist(&bodies[i], &bodies[j]);
            body_gravity(r, &bodies[i], &bodies[j]);
            body_gravity(r, &bodies[j], &bodies[i]);
        }
    }
}

/* physics.go:148-150 */
static double inelastic_collision(double ma, double va, double mb, double vb)
{
    return (ma * va + mb * vb) / (ma + mb);
}

/* physics.go:153-162: b merges into a.  a->Mass is already the sum when the velocities are mixed, exactly
 * as the reference writes it. */
static void body_combine(orc_body *a, const orc_body *b)
{
    a->Mass = a->Mass + b->Mass;
    a->Radius = cbrt(a->Radius * a->Radius * a->Radius + b->Radius * b->Radius * b->Radius);
    a->Vx = inelastic_collision(a->Mass, a->Vx, b->Mass, b->Vx);
    a->Vy = inelastic_collision(a->Mass, a->Vy, b->Mass, b->Vy);
    a->Vz = inelastic_collision(a->Mass, a->Vz, b->Mass, b->Vz);
    a->fx = a->fx + b->fx;
    a->fy = a->fy + b->fy;
    a->fz = a->fz + b->fz;
}

/* main.go:107-128 with collisions on: a pair that touches merges on the spot and the absorbed body is
 * skipped from then on.  Returns the number of merges. */
static long direct_forces_collide(orc_body *bodies, long n)
{
    long merges = 0;
    for (long i = 0; i < n - 1; i++) {
        if (!bodies[i].alive) continue;
        for (long j = i + 1; j < n; j++) {
            if (!bodies[j].alive) continue;
            double r = body_dist(&bodies[i], &bodies[j]);
            body_gravity(r, &bodies[i], &bodies[j]);
            body_gravity(r, &bodies[j], &bodies[i]);
            if (r <= bodies[i].Radius + bodies[j].Radius) {
                body_combine(&bodies[i], &bodies[j]);
                bodies[j].alive = 0;
                merges++;
            }
        }
    }
    return merges;
}

/* ---------------------------------------------------------------- octree (tree.go) */

typedef struct { double center[3], width[3]; } nodebound;   /* tree.go:41-43 */

/* tree.go:60-62 */
static double bound_max(const nodebound *n)
{
    return fmax(n->width[0], fmax(n->width[1], n->width[2]));
}

/* tree.go:65-72 */
static int bound_contains(const nodebound *n, const double p[3])
{
    double hw0 = n->width[0] * 0.5, hw1 = n->width[1] * 0.5, hw2 = n->width[2] * 0.5;
    return (n->center[0] - hw0 <= p[0] && p[0] < n->center[0] + hw0) &&
           (n->center[1] - hw1 <= p[1] && p[1] < n->center[1] + hw1) &&
           (n->center[2] - hw2 <= p[2] && p[2] < n->center[2] + hw2);
}

/* tree.go:75-79: the full width (not half) minus r is compared with the offset from the centre. */
static int bound_contains_sphere(const nodebound *n, const double p[3], double r)
{
    double w0 = n->width[0] - r, w1 = n->width[1] - r, w2 = n->width[2] - r;
    double d0 = p[0] - n->center[0], d1 = p[1] - n->center[1], d2 = p[2] - n->center[2];
    return fabs(d0) <= w0 && fabs(d1) <= w1 && fabs(d2) <= w2;
}

/* tree.go:82-86: full width plus r, and one axis within reach is enough (the reference writes ||). */
static int bound_intersect_sphere(const nodebound *n, const double p[3], double r)
{
    double w0 = n->width[0] + r, w1 = n->width[1] + r, w2 = n->width[2] + r;
    double d0 = p[0] - n->center[0], d1 = p[1] - n->center[1], d2 = p[2] - n->center[2];
    return fabs(d0) <= w0 || fabs(d1) <= w1 || fabs(d2) <= w2;
}

/* tree.go:101-110 (with scale :89-92 and translate :95-98) */
static nodebound octant_bound(const nodebound *parent, unsigned oct)
{
    nodebound r;
    double tx0 = parent->width[0] * 0.25 * ((double)((oct & 1u) * 2u) - 1.0);
    double tx1 = parent->width[1] * 0.25 * ((double)(((oct & 2u) >> 1) * 2u) - 1.0);
    double tx2 = parent->width[2] * 0.25 * ((double)(((oct & 4u) >> 2) * 2u) - 1.0);
    r.width[0] = parent->width[0] * 0.5;
    r.width[1] = parent->width[1] * 0.5;
    r.width[2] = parent->width[2] * 0.5;
    r.center[0] = parent->center[0] + tx0;
    r.center[1] = parent->center[1] + tx1;
    r.center[2] = parent->center[2] + tx2;
    return r;
}

static uint64_t f64bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* tree.go:128-132: bit k set iff the sign bit of point[k]-mid[k] is clear. */
static unsigned octant_bits(const double mid[3], const double p[3])
{
    return (unsigned)((~f64bits(p[0] - mid[0]) >> 63) |
                      ((~f64bits(p[1] - mid[1]) >> 63) << 1) |
                      ((~f64bits(p[2] - mid[2]) >> 63) << 2));
}

enum { KIND_EXTERNAL = 0, KIND_INTERNAL = 1 };

/* tree.go:134-141.  `particle` is an index into the body array (-1 = nil).
 * Extension beyond the reference (used only when max_depth > 0): a node at the depth cap that would
 * have to split again becomes a bucket whose children are one-body leaves kept in `bucket`. */
typedef struct orc_node {
    int kind;
    int has_children;                 /* children slice allocated (tree.go:170) */
    struct orc_node *children[8];
    long particle;
    double totalMass;
    double centerOfMass[3];
    nodebound bounds;
    int depth;
    struct orc_node **bucket; long nbucket, capbucket;
} orc_node;

typedef struct arena_block { struct arena_block *next; long used, cap; orc_node nodes[]; } arena_block;

typedef struct {
    arena_block *blocks;
    long nnodes;
    int com_mode;       /* 0 = as the reference computes it (tree.go:189-193), 1 = mass-weighted mean */
    int max_depth;      /* 0 = unbounded recursion like the reference; >0 = bucket at that depth */
    int overflow;       /* set when recursion exceeded the guard (coincident bodies, max_depth == 0) */
    orc_body *bodies;
} orc_tree;

#define ORC_DEPTH_GUARD 1200

static orc_node *new_node(orc_tree *t, const nodebound *b, int depth)
{
    arena_block *blk = t->blocks;
    if (!blk || blk->used == blk->cap) {
        long cap = blk ? blk->cap * 2 : 4096;
        arena_block *nb = (arena_block *)malloc(sizeof(arena_block) + (size_t)cap * sizeof(orc_node));
        nb->next = blk; nb->used = 0; nb->cap = cap;
        t->blocks = blk = nb;
    }
    orc_node *n = &blk->nodes[blk->used++];
    memset(n, 0, sizeof(*n));
    n->particle = -1;
    n->bounds = *b;
    n->depth = depth;
    t->nnodes++;
    return n;
}

static void tree_free(orc_tree *t)
{
    for (arena_block *b = t->blocks; b;) {
        for (long i = 0; i < b->used; i++) free(b->nodes[i].bucket);
        arena_block *nx = b->next; free(b); b = nx;
    }
    t->blocks = NULL;
}

static void add_mass(orc_tree *t, orc_node *n, const orc_body *b, const double point[3])
{
    if (t->com_mode == 0) {
        /* tree.go:191-192 */
        n->totalMass += b->Mass;
        double s = b->Mass / n->totalMass;
        n->centerOfMass[0] = n->centerOfMass[0] + point[0] * s;
        n->centerOfMass[1] = n->centerOfMass[1] + point[1] * s;
        n->centerOfMass[2] = n->centerOfMass[2] + point[2] * s;
    } else {
        double mold = n->totalMass;
        double mnew = mold + b->Mass;
        n->centerOfMass[0] = (n->centerOfMass[0] * mold + point[0] * b->Mass) / mnew;
        n->centerOfMass[1] = (n->centerOfMass[1] * mold + point[1] * b->Mass) / mnew;
        n->centerOfMass[2] = (n->centerOfMass[2] * mold + point[2] * b->Mass) / mnew;
        n->totalMass = mnew;
    }
}

/* tree.go:145-197 */
static int node_push(orc_tree *t, orc_node *n, long bi)
{
    const orc_body *b = &t->bodies[bi];
    double point[3] = { b->X, b->Y, b->Z };
    if (!bound_contains(&n->bounds, point)) return 0;
    if (n->depth > ORC_DEPTH_GUARD) { t->overflow = 1; return 1; }

    if (t->max_depth > 0 && n->depth == t->max_depth && (n->bucket || n->particle >= 0)) {
        /* depth-cap extension: keep every body of this cell as a one-body leaf child */
        if (!n->bucket) {
            n->capbucket = 4;
            n->bucket = (orc_node **)malloc(sizeof(orc_node *) * (size_t)n->capbucket);
            orc_node *c = new_node(t, &n->bounds, n->depth + 1);
            c->particle = n->particle;
            c->centerOfMass[0] = n->centerOfMass[0]; c->centerOfMass[1] = n->centerOfMass[1];
            c->centerOfMass[2] = n->centerOfMass[2];
            c->totalMass = n->totalMass;
            n->bucket[n->nbucket++] = c;
            n->particle = -1;
            n->kind = KIND_INTERNAL;
        }
        if (n->nbucket == n->capbucket) {
            n->capbucket *= 2;
            n->bucket = (orc_node **)realloc(n->bucket, sizeof(orc_node *) * (size_t)n->capbucket);
        }
        orc_node *c = new_node(t, &n->bounds, n->depth + 1);
        c->particle = bi;
        c->centerOfMass[0] = point[0]; c->centerOfMass[1] = point[1]; c->centerOfMass[2] = point[2];
        c->totalMass = b->Mass;
        n->bucket[n->nbucket++] = c;
        add_mass(t, n, b, point);
        return 1;
    }

    if (n->kind == KIND_EXTERNAL) {
        if (n->particle < 0) {                       /* tree.go:156-161 */
            n->particle = bi;
            n->centerOfMass[0] = point[0]; n->centerOfMass[1] = point[1]; n->centerOfMass[2] = point[2];
            n->totalMass = b->Mass;
            return 1;
        }
        /* tree.go:170-177: split, push the resident body down by its stored centre of mass */
        n->has_children = 1;
        unsigned oct = octant_bits(n->bounds.center, n->centerOfMass);
        if (!n->children[oct]) {
            nodebound cb = octant_bound(&n->bounds, oct);
            n->children[oct] = new_node(t, &cb, n->depth + 1);
        }
        node_push(t, n->children[oct], n->particle);
        n->particle = -1;
        n->kind = KIND_INTERNAL;
        /* fallthrough, tree.go:181 */
    }
    {   /* tree.go:183-193 */
        unsigned oct = octant_bits(n->bounds.center, point);
        if (!n->children[oct]) {
            nodebound cb = octant_bound(&n->bounds, oct);
            n->children[oct] = new_node(t, &cb, n->depth + 1);
        }
        if (node_push(t, n->children[oct], bi)) add_mass(t, n, b, point);
    }
    return 1;                                        /* tree.go:196 */
}

/* tree.go:284-295 */
static orc_node *maketree(orc_tree *t, long n, const nodebound *simbound)
{
    orc_node *root = new_node(t, simbound, 0);
    for (long i = 0; i < n; i++) {
        if (!t->bodies[i].alive) continue;
        if (!node_push(t, root, i)) t->bodies[i].alive = 0;   /* tree.go:290-292 */
    }
    return root;
}

/* tree.go:202-240; returns the number of force evaluations applied to the body. */
static long node_gravity(const orc_node *n, orc_body *bodies, long bi, double theta)
{
    if (n->totalMass == 0) return 0;
    if (n->particle == bi) return 0;
    orc_body *b = &bodies[bi];
    double d0 = n->centerOfMass[0] - b->X, d1 = n->centerOfMass[1] - b->Y, d2 = n->centerOfMass[2] - b->Z;
    double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);               /* Vec3.Sub(p).Len() */
    if (n->kind == KIND_INTERNAL) {
        double criterion = bound_max(&n->bounds) / r;
        if (criterion > theta) {
            long cnt = 0;
            if (n->bucket) {
                for (long i = 0; i < n->nbucket; i++) cnt += node_gravity(n->bucket[i], bodies, bi, theta);
            } else {
                for (int i = 0; i < 8; i++)
                    if (n->children[i]) cnt += node_gravity(n->children[i], bodies, bi, theta);
            }
            return cnt;
        }
    }
    double f = ORC_G * (n->totalMass * b->Mass) / (r * r);
    b->fx += (n->centerOfMass[0] - b->X) / r * f;
    b->fy += (n->centerOfMass[1] - b->Y) / r * f;
    b->fz += (n->centerOfMass[2] - b->Z) / r * f;
    return 1;
}

/* The queue of main.go:153: pairs (searching body, resident body of the leaf it touched). */
typedef struct { long (*pair)[2]; long len, cap; } pairlist;

static void pairlist_push(pairlist *l, long a, long b)
{
    if (l->len == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 64;
        l->pair = (long (*)[2])realloc(l->pair, sizeof(long[2]) * (size_t)l->cap);
    }
    l->pair[l->len][0] = a; l->pair[l->len][1] = b; l->len++;
}

/* tree.go:245-281.  A bucket of the depth-cap extension checks all of its leaves. */
static void node_check_collision(const orc_node *n, const orc_body *bodies, long bi, pairlist *out)
{
    if (n->particle == bi) return;                                   /* tree.go:246-248 */
    const orc_body *b = &bodies[bi];
    double p[3] = { b->X, b->Y, b->Z };
    if (!bound_intersect_sphere(&n->bounds, p, b->Radius)) return;
    if (n->kind == KIND_EXTERNAL) {
        if (n->particle >= 0 && bodies[n->particle].alive) {
            double r = body_dist(&bodies[n->particle], b);
            if (r <= bodies[n->particle].Radius + b->Radius) pairlist_push(out, bi, n->particle);
        }
        return;
    }
    if (n->bucket) {
        for (long i = 0; i < n->nbucket; i++) node_check_collision(n->bucket[i], bodies, bi, out);
        return;
    }
    unsigned oct = octant_bits(n->bounds.center, p);
    if (n->children[oct] && bound_contains_sphere(&n->children[oct]->bounds, p, b->Radius)) {
        node_check_collision(n->children[oct], bodies, bi, out);      /* tree.go:266-270 */
    } else {
        for (int i = 0; i < 8; i++)
            if (n->children[i]) node_check_collision(n->children[i], bodies, bi, out);
    }
}

/* main.go:152-167: every live body searches the tree in slice order, then the queue is worked off in order;
 * a pair with a body that an earlier merge removed is dropped.  Returns the number of merges. */
static long tree_collisions(const orc_node *root, orc_body *bodies, long n, long *queued)
{
    pairlist q = { NULL, 0, 0 };
    for (long i = 0; i < n; i++) {
        if (!bodies[i].alive) continue;
        node_check_collision(root, bodies, i, &q);
    }
    long merges = 0;
    for (long k = 0; k < q.len; k++) {
        orc_body *a = &bodies[q.pair[k][0]], *b = &bodies[q.pair[k][1]];
        if (!a->alive || !b->alive) continue;
        body_combine(a, b);
        b->alive = 0;
        merges++;
    }
    if (queued) *queued += q.len;
    free(q.pair);
    return merges;
}

typedef struct { const orc_node *root; orc_body *bodies; long lo, hi; double theta; long count; } walk_job;

static void *walk_thread(void *arg)
{
    walk_job *j = (walk_job *)arg;
    long c = 0;
    for (long i = j->lo; i < j->hi; i++) {
        if (!j->bodies[i].alive) continue;
        c += node_gravity(j->root, j->bodies, i, j->theta);
    }
    j->count = c;
    return NULL;
}

/* main.go:133-149.  groups = 4 in the reference; groupsize = n/groups, so with drop_tail the last
 * n % groups bodies get no force, exactly as the slicing at main.go:147 leaves them out. */
static long tree_forces(orc_tree *t, long n, const nodebound *simbound, double theta, int groups, int drop_tail,
                        double *build_seconds, double *walk_seconds, int collide, long *merges, long *queued);

#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static long tree_forces(orc_tree *t, long n, const nodebound *simbound, double theta, int groups, int drop_tail,
                        double *build_seconds, double *walk_seconds, int collide, long *merges, long *queued)
{
    double t0 = now_s();
    orc_node *root = maketree(t, n, simbound);
    double t1 = now_s();
    if (groups < 1) groups = 1;
    if (groups > 256) groups = 256;
    walk_job jobs[256]; pthread_t th[256];
    long groupsize = n / groups;
    for (int g = 0; g < groups; g++) {
        jobs[g].root = root; jobs[g].bodies = t->bodies; jobs[g].theta = theta; jobs[g].count = 0;
        jobs[g].lo = g * groupsize;
        jobs[g].hi = (g + 1) * groupsize;
        if (!drop_tail && g == groups - 1) jobs[g].hi = n;
    }
    if (groups == 1) walk_thread(&jobs[0]);
    else {
        for (int g = 0; g < groups; g++) pthread_create(&th[g], NULL, walk_thread, &jobs[g]);
        for (int g = 0; g < groups; g++) pthread_join(th[g], NULL);
    }
    long total = 0;
    for (int g = 0; g < groups; g++) total += jobs[g].count;
    double t2 = now_s();
    if (collide) *merges += tree_collisions(root, t->bodies, n, queued);
    if (build_seconds) *build_seconds += t1 - t0;
    if (walk_seconds) *walk_seconds += t2 - t1;
    return total;
}

/* ---------------------------------------------------------------- exported C interface (ctypes) */

typedef struct {
    int mode;            /* 0 direct (main.go:105-128), 1 tree (main.go:129-149) */
    int com_mode;        /* tree: 0 = reference's running update, 1 = mass-weighted mean */
    int drop_tail;       /* tree: 1 = leave out the last n % groups bodies like main.go:136-147 */
    int groups;          /* tree: walk threads; the reference uses 4 (main.go:134) */
    int max_depth;       /* tree: 0 = unbounded like the reference */
    double theta;        /* main.go:73 uses 1 */
    double dt;           /* main.go:28 uses 3600 */
    double bound_center[3];
    double bound_width;  /* main.go:74-76 uses centre 0, width 2^16 */
} orc_params;

typedef struct {
    long interactions;   /* force evaluations applied (tree: accepted node/leaf; direct: ordered pairs) */
    long nodes;          /* tree nodes allocated in the last build */
    double root_mass, root_com[3];
    double build_seconds, walk_seconds, update_seconds;
    int overflow;
    long merges;         /* bodies absorbed by collisions (physics.go:153-162) */
    long queued;         /* tree mode: pairs queued by checkCollision (main.go:153-159) */
} orc_stats;

static void load_bodies(orc_body *bs, long n, const double *mass, const double *pos, const double *vel,
                        const unsigned char *alive);
static void load_radius(orc_body *bs, long n, const double *radius)
{
    for (long i = 0; i < n; i++) bs[i].Radius = radius ? radius[i] : 2.0;   /* physics.go:20 defaultRadius */
}
static void load_bodies(orc_body *bs, long n, const double *mass, const double *pos, const double *vel,
                        const unsigned char *alive)
{
    for (long i = 0; i < n; i++) {
        memset(&bs[i], 0, sizeof(orc_body));
        bs[i].ID = (uint64_t)i;
        bs[i].Mass = mass[i];
        bs[i].X = pos[3 * i]; bs[i].Y = pos[3 * i + 1]; bs[i].Z = pos[3 * i + 2];
        if (vel) { bs[i].Vx = vel[3 * i]; bs[i].Vy = vel[3 * i + 1]; bs[i].Vz = vel[3 * i + 2]; }
        bs[i].alive = alive ? alive[i] != 0 : 1;
    }
}

static void forces_once(orc_body *bs, long n, const orc_params *p, orc_stats *st, int collide)
{
    if (p->mode == 0) {
        double t0 = now_s();
        if (collide) st->merges += direct_forces_collide(bs, n);
        else direct_forces(bs, n);
        long live = 0;
        for (long i = 0; i < n; i++) live += bs[i].alive;
        st->interactions += live * (live - 1);
        st->walk_seconds += now_s() - t0;
    } else {
        orc_tree t; memset(&t, 0, sizeof(t));
        t.com_mode = p->com_mode; t.max_depth = p->max_depth; t.bodies = bs;
        nodebound sb;
        for (int k = 0; k < 3; k++) { sb.center[k] = p->bound_center[k]; sb.width[k] = p->bound_width; }
        st->interactions += tree_forces(&t, n, &sb, p->theta, p->groups, p->drop_tail,
                                        &st->build_seconds, &st->walk_seconds, collide, &st->merges, &st->queued);
        st->nodes = t.nnodes;
        /* the root is the first node of the oldest arena block */
        arena_block *b = t.blocks; while (b->next) b = b->next;
        st->root_mass = b->nodes[0].totalMass;
        for (int k = 0; k < 3; k++) st->root_com[k] = b->nodes[0].centerOfMass[k];
        st->overflow |= t.overflow;
        tree_free(&t);
    }
}

/* Forces (not accelerations) on every body for the given state; `alive` is updated in tree mode
 * (out-of-bounds bodies are removed, tree.go:290-292). */
int orc_forces(long n, const double *mass, const double *pos, unsigned char *alive, const orc_params *p,
               double *force_out, orc_stats *st)
{
    orc_body *bs = (orc_body *)malloc(sizeof(orc_body) * (size_t)(n > 0 ? n : 1));
    orc_stats local; memset(&local, 0, sizeof(local));
    load_bodies(bs, n, mass, pos, NULL, alive);
    if (n > 0) forces_once(bs, n, p, &local, 0);
    for (long i = 0; i < n; i++) {
        force_out[3 * i] = bs[i].fx; force_out[3 * i + 1] = bs[i].fy; force_out[3 * i + 2] = bs[i].fz;
        if (alive) alive[i] = (unsigned char)bs[i].alive;
    }
    free(bs);
    if (st) *st = local;
    return 0;
}

/* ---- not in the reference: the opening test of tree.go:216-217 decided per group of bodies -------------------
 * The engine's optional per-wave walk lets `group` bodies that follow each other in the tree's depth-first leaf order
 * (children 0..7, tree.go:221-225: the order of their Morton keys) decide together: a cell is opened for all of
 * them as soon as one of them would open it.  This restates that rule over the pointer octree so that the engine can be
 * checked against something that shares no code with it.  A body skips its own leaf (tree.go:206-208) and, like the
 * engine, a body at distance 0 (the reference would produce NaN there). */
static void collect_leaves(const orc_node *n, long *order, long *k)
{
    if (n->totalMass == 0 && n->particle < 0 && n->kind == KIND_EXTERNAL) return;
    if (n->kind == KIND_EXTERNAL) { if (n->particle >= 0) order[(*k)++] = n->particle; return; }
    if (n->bucket) { for (long i = 0; i < n->nbucket; i++) collect_leaves(n->bucket[i], order, k); return; }
    for (int i = 0; i < 8; i++) if (n->children[i]) collect_leaves(n->children[i], order, k);
}

static long group_gravity(const orc_node *n, orc_body *bodies, const long *members, long cnt, double theta)
{
    if (n->totalMass == 0) return 0;
    if (n->kind == KIND_INTERNAL) {
        int any_open = 0;
        for (long m = 0; m < cnt && !any_open; m++) {
            const orc_body *b = &bodies[members[m]];
            double d0 = n->centerOfMass[0] - b->X, d1 = n->centerOfMass[1] - b->Y, d2 = n->centerOfMass[2] - b->Z;
            double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (bound_max(&n->bounds) / r > theta) any_open = 1;      /* tree.go:216-217 for this body */
        }
        if (any_open) {
            long c = 0;
            if (n->bucket) { for (long i = 0; i < n->nbucket; i++) c += group_gravity(n->bucket[i], bodies, members, cnt, theta); }
            else { for (int i = 0; i < 8; i++) if (n->children[i]) c += group_gravity(n->children[i], bodies, members, cnt, theta); }
            return c;
        }
    }
    long c = 0;
    for (long m = 0; m < cnt; m++) {
        if (n->particle == members[m]) continue;
        orc_body *b = &bodies[members[m]];
        double d0 = n->centerOfMass[0] - b->X, d1 = n->centerOfMass[1] - b->Y, d2 = n->centerOfMass[2] - b->Z;
        double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if (r == 0) continue;
        double f = ORC_G * (n->totalMass * b->Mass) / (r * r);        /* tree.go:234-238 */
        b->fx += d0 / r * f; b->fy += d1 / r * f; b->fz += d2 / r * f;
        c++;
    }
    return c;
}

int orc_forces_group(long n, const double *mass, const double *pos, unsigned char *alive, const orc_params *p, long group,
                     double *force_out, orc_stats *st)
{
    orc_body *bs = (orc_body *)malloc(sizeof(orc_body) * (size_t)(n > 0 ? n : 1));
    long *order = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    orc_stats local; memset(&local, 0, sizeof(local));
    load_bodies(bs, n, mass, pos, NULL, alive);
    for (long i = 0; i < n; i++) { bs[i].fx = 0; bs[i].fy = 0; bs[i].fz = 0; }
    if (n > 0 && group > 0) {
        orc_tree t; memset(&t, 0, sizeof(t));
        t.com_mode = p->com_mode; t.max_depth = p->max_depth; t.bodies = bs;
        nodebound sb;
        for (int k = 0; k < 3; k++) { sb.center[k] = p->bound_center[k]; sb.width[k] = p->bound_width; }
        orc_node *root = maketree(&t, n, &sb);
        long k = 0;
        collect_leaves(root, order, &k);
        for (long g0 = 0; g0 < k; g0 += group)
            local.interactions += group_gravity(root, bs, order + g0, (k - g0 < group ? k - g0 : group), p->theta);
        local.nodes = t.nnodes;
        tree_free(&t);
    }
    for (long i = 0; i < n; i++) {
        force_out[3 * i] = bs[i].fx; force_out[3 * i + 1] = bs[i].fy; force_out[3 * i + 2] = bs[i].fz;
        if (alive) alive[i] = (unsigned char)bs[i].alive;
    }
    free(order); free(bs);
    if (st) *st = local;
    return 0;
}

/* Advance the state by nsteps iterations of main.go:104-178.  collide = 0 is the `-nocollision` run; with
 * collide != 0 masses, radii and the live set change too (radius NULL = every body 2, physics.go:20). */
int orc_step2(long n, double *mass, double *radius, double *pos, double *vel, unsigned char *alive,
              const orc_params *p, int collide, int nsteps, orc_stats *st)
{
    orc_body *bs = (orc_body *)malloc(sizeof(orc_body) * (size_t)(n > 0 ? n : 1));
    orc_stats local; memset(&local, 0, sizeof(local));
    load_bodies(bs, n, mass, pos, vel, alive);
    load_radius(bs, n, radius);
    for (int s = 0; s < nsteps && n > 0; s++) {
        forces_once(bs, n, p, &local, collide);
        double t0 = now_s();
        for (long i = 0; i < n; i++) {                 /* main.go:171-177 */
            if (!bs[i].alive) continue;
            body_update(&bs[i], p->dt);
        }
        local.update_seconds += now_s() - t0;
    }
    for (long i = 0; i < n; i++) {
        pos[3 * i] = bs[i].X; pos[3 * i + 1] = bs[i].Y; pos[3 * i + 2] = bs[i].Z;
        vel[3 * i] = bs[i].Vx; vel[3 * i + 1] = bs[i].Vy; vel[3 * i + 2] = bs[i].Vz;
        if (collide) { mass[i] = bs[i].Mass; if (radius) radius[i] = bs[i].Radius; }
        if (alive) alive[i] = (unsigned char)bs[i].alive;
    }
    free(bs);
    if (st) *st = local;
    return 0;
}

int orc_step(long n, const double *mass, double *pos, double *vel, unsigned char *alive, const orc_params *p,
             int nsteps, orc_stats *st)
{
    return orc_step2(n, (double *)mass, NULL, pos, vel, alive, p, 0, nsteps, st);
}

/* One collision search over the tree of the given state (no forces, no merging): the queue of
 * main.go:153-159 as index pairs (searcher, resident).  Returns the queue length; at most cap pairs are written. */
long orc_collision_queue(long n, const double *mass, const double *radius, const double *pos, unsigned char *alive,
                         const orc_params *p, long *pairs_out, long cap)
{
    orc_body *bs = (orc_body *)malloc(sizeof(orc_body) * (size_t)(n > 0 ? n : 1));
    load_bodies(bs, n, mass, pos, NULL, alive);
    load_radius(bs, n, radius);
    orc_tree t; memset(&t, 0, sizeof(t));
    t.com_mode = p->com_mode; t.max_depth = p->max_depth; t.bodies = bs;
    nodebound sb;
    for (int k = 0; k < 3; k++) { sb.center[k] = p->bound_center[k]; sb.width[k] = p->bound_width; }
    orc_node *root = maketree(&t, n, &sb);
    pairlist q = { NULL, 0, 0 };
    for (long i = 0; i < n; i++) {
        if (!bs[i].alive) continue;
        node_check_collision(root, bs, i, &q);
    }
    for (long k = 0; k < q.len && k < cap; k++) { pairs_out[2 * k] = q.pair[k][0]; pairs_out[2 * k + 1] = q.pair[k][1]; }
    long len = q.len;
    if (alive) for (long i = 0; i < n; i++) alive[i] = (unsigned char)bs[i].alive;
    free(q.pair);
    tree_free(&t);
    free(bs);
    return len;
}

/* Tree inspection for structure parity: pre-order dump (children in octant order 0..7) of every node
 * that holds mass.  Per node: depth, kind, particle index (-1 if none), number of non-empty children,
 * mass, centre of mass, cell centre and width.  Returns the number of nodes written (<= cap). */
typedef struct { int depth, kind; long particle; int nchild; double mass, com[3], center[3], width; } orc_node_info;

static long dump_rec(const orc_node *n, orc_node_info *out, long cap, long k)
{
    if (n->totalMass == 0 && n->particle < 0 && n->kind == KIND_EXTERNAL) return k;
    long self = k;
    if (k < cap) {
        out[k].depth = n->depth; out[k].kind = n->kind; out[k].particle = n->particle; out[k].nchild = 0;
        out[k].mass = n->totalMass;
        for (int i = 0; i < 3; i++) { out[k].com[i] = n->centerOfMass[i]; out[k].center[i] = n->bounds.center[i]; }
        out[k].width = bound_max(&n->bounds);
    }
    k++;
    int nc = 0;
    if (n->bucket) {
        for (long i = 0; i < n->nbucket; i++) { long k2 = dump_rec(n->bucket[i], out, cap, k); nc += k2 > k; k = k2; }
    } else if (n->kind == KIND_INTERNAL) {
        for (int i = 0; i < 8; i++)
            if (n->children[i]) { long k2 = dump_rec(n->children[i], out, cap, k); nc += k2 > k; k = k2; }
    }
    if (self < cap) out[self].nchild = nc;
    return k;
}

long orc_tree_dump(long n, const double *mass, const double *pos, unsigned char *alive, const orc_params *p,
                   orc_node_info *out, long cap)
{
    orc_body *bs = (orc_body *)malloc(sizeof(orc_body) * (size_t)(n > 0 ? n : 1));
    load_bodies(bs, n, mass, pos, NULL, alive);
    orc_tree t; memset(&t, 0, sizeof(t));
    t.com_mode = p->com_mode; t.max_depth = p->max_depth; t.bodies = bs;
    nodebound sb;
    for (int k = 0; k < 3; k++) { sb.center[k] = p->bound_center[k]; sb.width[k] = p->bound_width; }
    orc_node *root = maketree(&t, n, &sb);
    long k = dump_rec(root, out, cap, 0);
    if (alive) for (long i = 0; i < n; i++) alive[i] = (unsigned char)bs[i].alive;
    tree_free(&t);
    free(bs);
    return k;
}

/* Total kinetic and potential energy and linear momentum (fp64, plain pair sum; eps softens r). */
int orc_energy(long n, const double *mass, const double *pos, const double *vel, const unsigned char *alive,
               double eps, double *kinetic, double *potential, double momentum[3])
{
    double ke = 0, pe = 0, px = 0, py = 0, pz = 0;
    for (long i = 0; i < n; i++) {
        if (alive && !alive[i]) continue;
        double vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
        ke += 0.5 * mass[i] * (vx * vx + vy * vy + vz * vz);
        px += mass[i] * vx; py += mass[i] * vy; pz += mass[i] * vz;
        for (long j = i + 1; j < n; j++) {
            if (alive && !alive[j]) continue;
            double dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1], dz = pos[3 * j + 2] - pos[3 * i + 2];
            pe -= ORC_G * mass[i] * mass[j] / sqrt(dx * dx + dy * dy + dz * dz + eps * eps);
        }
    }
    *kinetic = ke; *potential = pe; momentum[0] = px; momentum[1] = py; momentum[2] = pz;
    return 0;
}

/* fp64 direct sum with Plummer softening, row order (target i, sources j ascending, j != i); threaded
 * over targets.  This is the physical ground truth the tree results are also compared with; with
 * eps = 0 it is the same arithmetic as body_gravity accumulated in the order of main.go:107-128. */
typedef struct { long n, lo, hi; const double *mass, *pos; const unsigned char *alive; double eps2; double *out; } ds_job;
static void *ds_thread(void *arg)
{
    ds_job *j = (ds_job *)arg;
    for (long i = j->lo; i < j->hi; i++) {
        double fx = 0, fy = 0, fz = 0;
        if (!j->alive || j->alive[i]) {
            double xi = j->pos[3 * i], yi = j->pos[3 * i + 1], zi = j->pos[3 * i + 2], mi = j->mass[i];
            for (long k = 0; k < j->n; k++) {
                if (k == i || (j->alive && !j->alive[k])) continue;
                double dx = j->pos[3 * k] - xi, dy = j->pos[3 * k + 1] - yi, dz = j->pos[3 * k + 2] - zi;
                double r = sqrt(dx * dx + dy * dy + dz * dz + j->eps2);
                double f = ORC_G * (mi * j->mass[k]) / (r * r);
                fx += dx / r * f; fy += dy / r * f; fz += dz / r * f;
            }
        }
        j->out[3 * i] = fx; j->out[3 * i + 1] = fy; j->out[3 * i + 2] = fz;
    }
    return NULL;
}

/* Forces on targets [t0, t1) from all n sources. */
int orc_direct_rows(long n, const double *mass, const double *pos, const unsigned char *alive, double eps,
                    long t0, long t1, int nthreads, double *force_out)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    ds_job jobs[256]; pthread_t th[256];
    long span = t1 - t0, chunk = (span + nthreads - 1) / nthreads;
    int used = 0;
    for (int t = 0; t < nthreads; t++) {
        long lo = t0 + t * chunk, hi = lo + chunk; if (hi > t1) hi = t1;
        if (lo >= hi) break;
        jobs[used] = (ds_job){ n, lo, hi, mass, pos, alive, eps * eps, force_out - 3 * t0 };
        used++;
    }
    for (int t = 0; t < used; t++) pthread_create(&th[t], NULL, ds_thread, &jobs[t]);
    for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    return 0;
}
