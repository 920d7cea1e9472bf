/*
 * nbody_engine.h -- C ABI of the MI355X gravity engine (libnbody_engine.so).
 *
 * The reference program has no plugin or FFI interface; its seam is the body of the per-frame iteration
 * in main.go:104-178: "given the live body set, advance it by dt" (direct O(N^2) loop main.go:105-128 or
 * octree build + walk main.go:129-149, then body.update main.go:171-177).  Every entry point below
 * replaces a piece of that iteration and says which.  A Go host binds them through cgo with flat
 * []float64 / []uint64 slices (no Go pointers are retained after a call returns).
 *
 * Conventions: every function returns 0 on success or a negative NBE_E_* code; nbe_last_error() gives the
 * message of the calling thread's last failure.  The reference panics on every error (main.go:213-238),
 * so a host that wants identical behaviour turns a non-zero return into a panic.  The caller owns all host
 * arrays; the library copies during the call.  One engine is driven by one thread at a time.  There is no
 * CPU fallback: creation fails when no HIP device is usable.
 */
#ifndef NBODY_ENGINE_H
#define NBODY_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbe_engine nbe_engine;

enum { NBE_MODE_DIRECT = 0, NBE_MODE_TREE = 1 };
/* NBE_PREC_F32: fp32 state and arithmetic.  NBE_PREC_MIXED: fp64 state and integrator, fp32 pair terms
 * summed in fp32 per source tile and across tiles in fp64.  NBE_PREC_F64: fp64 throughout. */
enum { NBE_PREC_F32 = 0, NBE_PREC_MIXED = 1, NBE_PREC_F64 = 2 };

enum {
    NBE_OK = 0,
    NBE_E_INVALID = -1,     /* bad argument or call order */
    NBE_E_DEVICE = -2,      /* HIP runtime error, or no device */
    NBE_E_NOMEM = -3,
    NBE_E_UNSUPPORTED = -4
};

/* Run-time form of the compile-time constants of main.go:28 (dt), :73 (theta), :74-76 (simbound) and
 * physics.go:132 (G).  eps is Plummer softening added under the square root; 0 keeps the reference's
 * unsoftened force (physics.go:136, tree.go:234). */
typedef struct {
    int mode;               /* NBE_MODE_*: the `-tree` flag, main.go:22 */
    int precision;          /* NBE_PREC_* */
    double G, theta, dt, eps;
    double bound_center[3];
    double bound_width;     /* edge of the cubic simulation bound; bodies outside are removed in tree mode */
    int device;             /* HIP device ordinal */
    int collisions;         /* 0 = the `-nocollision` run (main.go:24); 1 = bodies that touch merge (tree mode:
                               tree.go:245-281 + main.go:151-167; direct sum: main.go:122-126; physics.go:153-162) */
    int reference_quirks;   /* tree mode: bit set of NBE_QUIRK_*; 0 = physically meant behaviour */
    int walk_criterion;     /* tree mode: NBE_WALK_PER_BODY (the reference's test, default) or NBE_WALK_PER_WAVE */
    int num_devices;        /* 0 or 1: one GPU, `device`.  G > 1: the engine drives G GPUs from this process, see below */
    int cell_sums;          /* tree mode: NBE_SUMS_PREFIX (default), NBE_SUMS_TREE or NBE_SUMS_EXACT, see below */
    const int *device_ids;  /* num_devices HIP device ordinals (copied by nbe_create); NULL = devices 0 .. G-1 */
    int distributed;        /* with num_devices > 1 in tree mode: 1 = every body lives on one GPU only and the GPUs exchange
                               pruned subtrees (the key-range-distributed tree, nbe_dist_* below, driven by the engine itself) */
} nbe_config;

/* Several GPUs behind one engine (num_devices = G > 1).  The reference's data-parallel strategy is a fan-out of
 * contiguous body ranges over workers that share one read-only tree (main.go:134-149); here a worker is a GPU.  Every
 * GPU holds all n bodies as sources and, in tree mode, sorts and builds the same tree; GPU g computes the forces on
 * and integrates the g-th contiguous range of ceil(n/G) bodies (device order) and the engine copies every range to
 * the other GPUs over the peer links (hipMemcpyPeerAsync on the receiving GPU's stream, ordered by events; no host
 * synchronisation inside nbe_step).  Results are those of one GPU: bit for bit in tree mode, to the rounding of a
 * differently split source range in the direct sum.  nbe_upload, nbe_step, nbe_download, nbe_accel, nbe_energy,
 * nbe_stats, nbe_sync, nbe_tree_nodes, nbe_time_force_kernel and the frame stream work as with one GPU (reads come
 * from the first GPU; nbe_set_radius reaches every GPU).  Collisions work in tree mode (the GPUs share out the force pass and
 * each runs the search, the merges and the update of all bodies), not in the direct sum; nbe_step_timed, nbe_set_stream,
 * nbe_device_state and the nbe_shard_* calls are refused with NBE_E_UNSUPPORTED.  The same ordinal may be listed more than once (several shards on one GPU).
 * With cfg.distributed = 1 (tree mode, per-body criterion, no reference_quirks; collisions allowed) the GPUs do not replicate the
 * bodies: the engine runs the key-range-distributed tree described at nbe_dist_upload below on its GPUs, one host thread per
 * GPU, with peer copies for the exchanges.  nbe_upload, nbe_step, nbe_download, nbe_stats, nbe_energy (tree potential),
 * nbe_set_radius (right after nbe_upload), nbe_get_radius, nbe_accel, nbe_sync and the frame stream, radii included, work (no GPU
 * holds all bodies: every GPU snapshots its own before the step and sends
 * them into its stretch of the frame's pinned buffer on a stream of its own; the rows of a frame are in key order);
 * results equal those of one GPU with cell_sums = NBE_SUMS_EXACT bit for bit in fp32 and fp64; the other calls are refused. */

/* How a cell's mass and centre of mass are summed (fp64 either way).
 *   NBE_SUMS_PREFIX   differences of prefix sums over the Morton-sorted bodies: three launches, the fastest, and the
 *                     rounding of a cell's sums depends on everything sorted before it.  fp32 node records only
 *                     (the error stays below their rounding); fp64 node records use NBE_SUMS_EXACT instead.
 *   NBE_SUMS_TREE     every cell adds up its children in octant order, bodies as they stand (fp64): a cell's sums depend
 *                     on nothing outside it; one launch per tree depth.
 *   NBE_SUMS_EXACT    every body's (m, m*x, m*y, m*z) is cut to a multiple of a quantum fixed for the run (2^-126 of the
 *                     total mass times the bound's reach) and summed as 128-bit integers; cells take differences of the
 *                     prefix sums (three launches, like NBE_SUMS_PREFIX).  Integer sums do not depend on the order of
 *                     addition, so trees built from any partition of the bodies carry the same bits: this is what the
 *                     key-range-distributed build (nbe_dist_*) uses, and one engine with this setting is what its results
 *                     equal bit for bit.  Works for fp32 and fp64 node records. */
enum { NBE_SUMS_PREFIX = 0, NBE_SUMS_TREE = 1, NBE_SUMS_EXACT = 2 };

/* Who decides whether a cell is opened (tree.go:216-217, edge / distance > theta).
 *   NBE_WALK_PER_BODY   every body for itself, as the reference's recursion does: each body receives exactly the
 *                       interactions of tree.go:202-240.
 *   NBE_WALK_PER_WAVE   the 128 Morton-consecutive bodies that share a wavefront (two per lane; the aligned 128-body
 *                       chunks of the sorted order) decide together: a cell is opened for all of them when one of them
 *                       would open it.  No body ever gets a coarser interaction than the reference's test allows, most
 *                       get finer ones (about twice the force evaluations at theta = 0.5), and the walk is 1.3-1.4x
 *                       faster because the per-body bookkeeping is gone and the arithmetic packs.  Not the reference's
 *                       interaction set: for production runs, not for parity checks. */
enum { NBE_WALK_PER_BODY = 0, NBE_WALK_PER_WAVE = 1 };

/* Two things the reference's tree path does that a gravity solver would not, reproduced on request so that a run
 * can be compared with the Go program number for number (both are off by default):
 *   NBE_QUIRK_REF_COM    a cell's "centre of mass" is the running value of tree.go:189-193,
 *                        com += p * (m / totalMass) in insertion (upload) order, which never rescales the old value:
 *                        com = p_1 + sum_{k>=2} p_k * m_k / (m_1 + ... + m_k).  Computed serially per cell, so
 *                        the build is slow for large scenes (the cell of all n bodies is one chain of n steps).
 *   NBE_QUIRK_DROP_TAIL  the walk covers 4 * (n / 4) bodies (main.go:134-148): the last n mod 4 bodies of the
 *                        upload order feel no force but are still integrated. */
enum { NBE_QUIRK_REF_COM = 1, NBE_QUIRK_DROP_TAIL = 2 };

typedef struct {
    uint64_t n, n_alive;
    uint64_t steps;             /* steps taken since upload */
    uint64_t interactions;      /* force evaluations of one force pass: direct = targets*(sources-1); tree = accepted
                                   body-node and body-leaf evaluations (tree.go:232-238), counted by the last
                                   nbe_accel / nbe_step_timed / nbe_time_force_kernel (plain steps do not count) */
    uint64_t tree_nodes;        /* leaves + branching cells of the last build */
    uint64_t tree_cells;        /* branching cells of the last build */
    uint64_t merges;            /* bodies absorbed by collisions since upload (physics.go:153-162) */
    uint64_t collision_pairs;   /* tree: pairs queued by the collision search of the last step (main.go:153-159);
                                   direct sum: merges of the last step */
    float ms_sort, ms_build, ms_force, ms_total;   /* device time of the last timed step (nbe_step_timed) */
} nbe_stats_t;

/* Raw device pointers of the engine's state for a host that exchanges shards itself (for example with
 * RCCL through torch.distributed).  pos_mass and vel hold `n` elements of 4 reals (x,y,z,mass / vx,vy,vz,0);
 * `elem_bytes` is 16 (fp32 state) or 32 (fp64 state).  *_next are the buffers nbe_shard_advance writes. */
typedef struct {
    void *pos_mass, *vel;
    void *pos_mass_next, *vel_next;
    uint64_t n;
    uint32_t elem_bytes;
    uint32_t reserved;
    void *acc;              /* n elements of 4 reals: the accelerations nbe_shard_advance writes with collisions on */
} nbe_device_state_t;

/* Fills cfg with the reference's constants: tree off, fp32, G = 6.67408e-11 (physics.go:132),
 * theta = 1 (main.go:73), dt = 3600 (main.go:28), eps = 0, bound centre 0 width 2^16 (main.go:74-76);
 * collisions off (the reference's default is on, main.go:24: set cfg.collisions = 1 for that). */
int nbe_default_config(nbe_config *cfg);

int nbe_create(const nbe_config *cfg, nbe_engine **out);
void nbe_destroy(nbe_engine *e);

/* Thread-local message of the last failing call. */
const char *nbe_last_error(void);

/* Replaces the body slice built by makebodies (physics.go:16-73) / overwriteBodies (main.go:243-248):
 * n bodies as host structure-of-arrays, positions and velocities as xyzxyz...  `id` may be NULL
 * (then id = index, as physics.go:37,69 assign).  All bodies start alive.
 * Limits: 2^31-1 bodies in direct mode, 2^26 in tree mode (NBE_E_UNSUPPORTED beyond; with distributed = 1 that is the limit
 * of one GPU's share, not of the run: 2^28 bodies have run as eight shares); masses must be >= 0.
 * With distributed = 1 every GPU gets room for a quarter more than an even share of the bodies; NBODY_DIST_HEADROOM in the
 * environment (1.25 .. the number of GPUs, read here) gives more, for scenes that cannot be shared out evenly. */
int nbe_upload(nbe_engine *e, size_t n, const double *pos_xyz, const double *vel_xyz, const double *mass,
               const uint64_t *id);

/* Body radii (physics.go:94), used only by collisions.  nbe_upload gives every body the reference's
 * defaultRadius 2 (physics.go:20); nbe_set_radius replaces them (n values in upload order) and needs an engine
 * created with collisions on.  nbe_get_radius returns the radii of the live bodies in the order of nbe_download. */
int nbe_set_radius(nbe_engine *e, const double *radius);
int nbe_get_radius(nbe_engine *e, double *radius, size_t *n_alive);

/* The copy of live bodies taken per frame at main.go:90-101.  Bodies removed by the tree bound check
 * (tree.go:290-292) are left out, as the nil skip at main.go:91-95 does; survivors keep upload order.
 * Any output pointer may be NULL.  *n_alive receives the number of bodies written. */
int nbe_download(nbe_engine *e, double *pos_xyz, double *vel_xyz, double *mass, uint64_t *id, size_t *n_alive);

/* Force pass only (main.go:105-149 without the update loop): accelerations f/m of all n uploaded bodies
 * in upload order, xyzxyz...; removed bodies get 0.  `alive_out` (n bytes) may be NULL. */
int nbe_accel(nbe_engine *e, double *acc_xyz, uint8_t *alive_out);

/* nsteps iterations of main.go:104-178: force pass, then v += f/m*dt; x += v*dt (physics.go:101-115)
 * fused into the force kernel's epilogue.  Asynchronous on the engine's stream.
 * With collisions on (tree mode) a step is: force pass, collision search of every live body through the tree in
 * upload order (tree.go:245-281, reproducing its pruning rules as written), the queued pairs merged one after
 * the other in queue order (main.go:160-166, physics.go:153-162), then the update; an absorbed body leaves the
 * simulation like one that left the bound.  Such a step reads the queue length back and so synchronises.
 * With collisions on in the direct sum, every step first tests all pairs for contact; a step with a touching pair
 * is replayed as the reference's i < j sweep with merges on the spot (main.go:107-128), the others run fused. */
int nbe_step(nbe_engine *e, int nsteps);

/* One step bracketed by device events; fills stats->ms_* (synchronises). */
int nbe_step_timed(nbe_engine *e, nbe_stats_t *stats);

int nbe_sync(nbe_engine *e);
int nbe_stats(nbe_engine *e, nbe_stats_t *stats);

/* Total kinetic energy, potential energy (softened by eps) and linear momentum of the live bodies; used for
 * drift checks.  Up to 262,144 bodies the potential is the exact fp64 pair sum; above that, tree mode sums it over
 * the nodes its walk accepts (same theta as the forces) and direct mode accepts up to 4,194,304 bodies.  In tree mode
 * "live" means inside the bound, as the coming step will see them (the call builds the tree, which that step reuses).
 * In NBE_PREC_MIXED the tree's potential is summed, like its forces, over the fp32 copies of the positions: two bodies
 * that coincide there (closer than fp32 resolves, millimetres at the reference's scale) add nothing, where the pair sum
 * reads the fp64 positions and counts them.
 * No counterpart in the reference (its only validation is the rendered frames). */
int nbe_energy(nbe_engine *e, double *kinetic, double *potential, double momentum[3]);

/* ---- sharded stepping (one engine per GPU, every engine holds all n bodies as sources) ----
 * The reference's only data-parallel strategy is to give each of 4 goroutines a contiguous quarter of the
 * bodies against a shared read-only tree (main.go:134-149); a shard is that quarter.
 *   nbe_set_shard(first, count)   targets [first, first+count) in device order belong to this engine.  Tree mode with
 *                                 the per-body test matches a one-engine run bit for bit in every precision wherever
 *                                 the ranges begin (NBE_PREC_MIXED cuts a body's fp32 partial sums at nodes picked by
 *                                 their records, not by its wavefront); NBE_WALK_PER_WAVE decides by aligned groups
 *                                 of 128 bodies, so there choose `first` as a multiple of 128
 *   nbe_shard_sort                optional, tree mode: the part of the sort that reads positions only (keys, radix
 *                                 passes), so that it can run while the velocities of the other engines still arrive
 *   nbe_shard_prepare             tree mode: sort by Morton key, reorder state, build the tree (replicated)
 *   nbe_shard_advance             forces on the shard + fused update, written into *_next at the same indices
 *   nbe_shard_advance_range       the same for an explicit range [first, first+count), any number of times per step:
 *                                 a host that owns several ranges can exchange one while the next is computed
 *   (caller all-gathers pos_mass_next and vel_next across engines)
 *   nbe_shard_commit              *_next become current
 * nbe_step is prepare + advance over all bodies + commit.
 * With collisions on (tree mode only; NBE_E_UNSUPPORTED in the direct sum) nbe_shard_advance computes the accelerations of
 * the range into `acc` (nbe_device_state) instead of updating, the caller all-gathers `acc`, and nbe_shard_commit runs
 * the collision search, the merges and the update of all bodies on every engine alike (tree.go:245-281, main.go:151-167,
 * physics.go:101-115, 153-162); radii must have been given to every engine. */
int nbe_set_shard(nbe_engine *e, uint64_t first, uint64_t count);
int nbe_shard_sort(nbe_engine *e);
int nbe_shard_prepare(nbe_engine *e);
int nbe_shard_advance(nbe_engine *e);
int nbe_shard_advance_range(nbe_engine *e, uint64_t first, uint64_t count);
int nbe_shard_commit(nbe_engine *e);
int nbe_device_state(nbe_engine *e, nbe_device_state_t *out);

/* ---- key-range-distributed tree (one engine per GPU, every body resident on ONE engine) ----
 * The sharded stepping above replicates the bodies and the tree build on every engine; this path distributes both, for
 * runs where the replicated sort and build would cap the scaling.  It still computes what main.go:130-149 computes: every
 * body walks the one octree of all bodies (tree.go:202-240); the octree is only never assembled in one place.
 *   * Rank r owns the bodies whose Morton key lies in [split[r], split[r+1]) and holds nothing else.  After every update
 *     the bodies that crossed a splitter move to their new owner (nbe_dist_classify / the caller's all-to-all of the
 *     migration records / nbe_dist_receive).
 *   * Every rank sorts its own bodies and builds the octree over them (nbe_dist_build).  A subtree whose cell lies
 *     wholly inside the rank's key range is exactly that subtree of the global octree (a "fragment"); the few cells
 *     above the fragments, which span ranks, are rebuilt on every rank from the fragments' roots (key, depth, exact
 *     sums), all-gathered by the caller.  Cell sums are NBE_SUMS_EXACT, so they carry the bits of a one-engine build.
 *   * For every other rank q, nbe_dist_export prunes each fragment to the nodes some body of q could visit (cells whose
 *     opening test no point of q's bounding box can pass are cut below), in pre-order with skip pointers; the caller
 *     moves these records (all-to-all), and nbe_dist_assemble lays out the rank's own fragments, the imported ones and
 *     the spanning cells as one pre-order node array.  The walk kernel then runs on it unchanged (nbe_dist_advance):
 *     every body visits exactly the nodes it visits in the one-engine tree, in the same order, so fp32 and fp64 runs
 *     equal a one-engine run with cell_sums = NBE_SUMS_EXACT bit for bit, and so do NBE_PREC_MIXED runs: a body's fp32
 *     partial sums join its fp64 totals after accepted nodes that are picked by their records alone.
 *     (Bodies that share all 63 key bits, i.e. lie in the same 3 cm cell of the reference's bound, are ordered by their
 *     upload index on one engine and by their run-wide index here, in runs of up to 64 such bodies, so they too are
 *     summed in one order everywhere; a longer run keeps the order in which its bodies reached the rank, and results
 *     then agree to rounding instead of bit for bit.)
 * Tree mode, NBE_WALK_PER_BODY, no reference_quirks; collisions as described at nbe_dist_set_radius.  nbe_step and the
 * nbe_shard_* calls are refused on such an engine. */
typedef struct {
    uint64_t key;           /* Morton key of the fragment's first body */
    int32_t level;          /* depth of the fragment's root cell; 255: the fragment is one body */
    int32_t start;          /* pre-order index of the root in the owner's node array */
    int32_t nodes;          /* nodes of the subtree */
    int32_t reserved;
    uint64_t sum[8];        /* exact sums of the fragment's bodies (NBE_SUMS_EXACT): mass, m*x, m*y, m*z as 128-bit two's
                               complement integers, low word first, in the run's quanta */
} nbe_fragment_t;

typedef struct {
    void *outbox, *inbox;   /* migration records, `slots` of them each; a record is record_bytes long (position+mass,
                               velocity+index, and with collisions on the radius) */
    void *export_nodes;     /* what nbe_dist_export wrote: node records for the other ranks, destination after destination */
    void *import_nodes;     /* where the caller puts the records exported to this rank, source rank after source rank */
    uint64_t slots, export_cap, import_cap;   /* capacities in records / nodes */
    uint32_t record_bytes, node_bytes;
} nbe_dist_buffers_t;

#define NBE_DIST_MAX_FRAGMENTS 2048
#define NBE_DIST_MAX_BOXES 8

/* Replaces nbe_upload: the `n` bodies this rank starts with (any subset of the run's bodies; they are redistributed by
 * the first nbe_dist_classify), in an engine with room for `capacity` >= n bodies.  `first_index` numbers them:
 * body i gets the run-wide index first_index + i, which travels with it and comes back from nbe_dist_download. */
int nbe_dist_upload(nbe_engine *e, int rank, int world, size_t capacity, size_t n, const double *pos_xyz,
                    const double *vel_xyz, const double *mass, uint64_t first_index);
/* Up to max_samples evenly spaced keys of the rank's bodies in sorted order (for choosing splitters); *n_live = bodies
 * inside the bound. */
int nbe_dist_sample_keys(nbe_engine *e, size_t max_samples, uint64_t *keys_out, size_t *count, size_t *n_live);
/* The total mass of the run's bodies (all ranks), the same value on every rank: it fixes the quantum of the exact cell sums
 * (NBE_SUMS_EXACT).  Needed once, before the first nbe_dist_build; an upper bound will do. */
int nbe_dist_set_total_mass(nbe_engine *e, double total_mass);
/* Collisions on (cfg.collisions = 1; tree.go:245-281, main.go:151-167, physics.go:153-162): radii of the rank's uploaded
 * bodies in upload order (NULL keeps the reference's defaultRadius 2, physics.go:20) and the largest radius any body of the
 * run may reach, the same value on every rank (default 32 m).  A collision search stays inside the cells around its body
 * that are at most four radii wide; nbe_dist_set_splitters moves every splitter down to the border of a cell at least four
 * caps wide, so such cells, the pairs a search can queue and the pairs that share a body all lie on one rank, and every
 * rank searches, merges (its own pairs in the reference's queue order) and updates for itself after the force pass.  Queue
 * lengths, merges and survivors equal those of one engine.  A body whose radius grows beyond the cap, or more than 262,144
 * bodies of one rank queueing pairs in one step, makes nbe_dist_advance fail with NBE_E_UNSUPPORTED.  Right after
 * nbe_dist_upload, before nbe_dist_set_splitters. */
int nbe_dist_set_radius(nbe_engine *e, const double *radius, double radius_cap);
/* split[0] = 0 <= split[1] <= ... <= split[world] = 2^63: rank r owns keys in [split[r], split[r+1]). */
int nbe_dist_set_splitters(nbe_engine *e, const uint64_t *split);
int nbe_dist_buffers(nbe_engine *e, nbe_dist_buffers_t *out);
/* Packs the bodies that belong to other ranks into the outbox (those for rank d start at record
 * send_counts[0] + ... + send_counts[d-1]) and removes them here.  Synchronises. */
int nbe_dist_classify(nbe_engine *e, uint32_t *send_counts);
/* After nbe_dist_classify: the number of records this rank can take in the nbe_dist_receive that follows (its free
 * slots, those of its leavers included).  A host adds up what every rank is about to be sent; where that exceeds a rank's
 * room (the bodies' distribution over key space changed more in one step than the spare slots allow for), every rank calls
 * nbe_dist_unclassify, new splitters are chosen from the bodies as they now are, and the step starts again with
 * nbe_dist_classify (nbody_amd/disttree.py and the engine's own driver do this). */
int nbe_dist_room(nbe_engine *e, uint64_t *room);
/* Undoes the last nbe_dist_classify (before any nbe_dist_receive): the leavers return from the outbox to their slots. */
int nbe_dist_unclassify(nbe_engine *e);
/* Takes over the first k records of the inbox. */
int nbe_dist_receive(nbe_engine *e, size_t k);
/* Sorts the rank's bodies, builds its tree and lists its fragments in key order (at most `max`).  boxes: up to
 * NBE_DIST_MAX_BOXES bounding boxes (min x,y,z, max x,y,z) that together hold the rank's bodies: a key range is a union of
 * cells that can lie far apart, so the fragments are grouped into runs that share a small cell and every run gets the box of
 * its bodies.  Synchronises. */
int nbe_dist_build(nbe_engine *e, nbe_fragment_t *frags, int max, int *count, double *boxes, int *nboxes, uint64_t *n_live);
/* all_frags: the fragments of rank 0, then rank 1, ... (frag_counts[r] each); boxes: world x NBE_DIST_MAX_BOXES x 6, of
 * which the first box_counts[r] of rank r count.  Writes the pruned fragments for every other rank into export_nodes;
 * export_offsets[d] .. export_offsets[d+1] are the nodes for rank d, export_sizes[d * frag_counts[rank] + f] the nodes of
 * this rank's fragment f in there.  Synchronises. */
int nbe_dist_export(nbe_engine *e, const nbe_fragment_t *all_frags, const int *frag_counts, const double *boxes,
                    const int *box_counts, int32_t *export_sizes, uint64_t *export_offsets);
/* import_sizes: for every rank r in turn, the sizes its frag_counts[r] fragments have in its export to this rank (this
 * rank's own entries are ignored); import_offsets[r]: where rank r's export starts in import_nodes, in nodes. */
int nbe_dist_assemble(nbe_engine *e, const int32_t *import_sizes, const uint64_t *import_offsets);
/* The host-side half of nbe_dist_assemble on its own (needs no GPU and no engine; for tests of the stitching): given the
 * fragments of all ranks in key order and the number of nodes each occupies, lays out the global pre-order.  tops receives
 * the cells that span fragments (first index, index after the subtree, depth, exact sums of the fragments below), frag_dst the first index of every fragment, *total the number of nodes. */
typedef struct {
    int32_t dst, skip, level, reserved;
    uint64_t sum[8];        /* as nbe_fragment_t.sum: the sums of the fragments below, added as integers */
} nbe_dist_top_cell_t;
int nbe_dist_layout(const nbe_fragment_t *all_frags, int nfrags, const int32_t *sizes, nbe_dist_top_cell_t *tops, int max_tops,
                    int *ntops, int32_t *frag_dst, int64_t *total);
/* Force pass over the assembled node array + fused update of the rank's bodies (main.go:130-149, 171-177). */
int nbe_dist_advance(nbe_engine *e);
/* This rank's share of nbe_energy, between nbe_dist_assemble and nbe_dist_advance (the state the next advance starts
 * from): kinetic energy and momentum of its bodies, and its bodies' share of the potential energy, -G/2 sum m_i phi_i with
 * phi_i summed over the nodes body i's walk accepts in the assembled array (all bodies of the run, softened by eps).  The
 * run's totals are the sums over the ranks.  Synchronises. */
int nbe_dist_energy(nbe_engine *e, double *kinetic, double *potential, double momentum[3]);
/* This rank's share of nbe_accel, between nbe_dist_assemble and nbe_dist_advance: the accelerations f/m of its bodies
 * (tree.go:202-240 over the assembled array) with their run-wide indices (index may be NULL).  Synchronises. */
int nbe_dist_accel(nbe_engine *e, uint64_t *index, double *acc_xyz, size_t *count);
/* The rank's live bodies: run-wide index, position, velocity, mass (any pointer may be NULL). */
int nbe_dist_download(nbe_engine *e, uint64_t *index, double *pos_xyz, double *vel_xyz, double *mass, size_t *count);
/* Collisions on: the radii of the rank's live bodies (physics.go:94; merged bodies have grown, physics.go:139) with their
 * run-wide indices, in the order of nbe_dist_download (index may be NULL). */
int nbe_dist_get_radius(nbe_engine *e, uint64_t *index, double *radius, size_t *count);

/* Run the engine's work on a caller-owned hipStream_t instead of the engine's own non-blocking stream, so
 * that it is ordered with the caller's other work (copies, collectives).  NULL selects HIP's default (null)
 * stream.  The previous stream is drained first. */
int nbe_set_stream(nbe_engine *e, void *hip_stream);

/* ---- frame streaming ----
 * The reference hands a copy of the live bodies to its output workers before every step through a channel of
 * capacity 32 (main.go:50-61, 90-101).  Here the copy leaves the GPU on a side stream into a ring of pinned host
 * buffers while the next step computes, and a library thread delivers it:
 *   cb(user, frame, n_alive, xyzm, radius, ids)
 *                                         xyzm = n_alive x (x, y, z, mass) as fp32 and radius = n_alive x fp32 (the
 *                                         record the reference's own frame store keeps, gob.go:54-61; radius is NULL
 *                                         when collisions are off: radii are then whatever the host uploaded),
 *                                         ids = body IDs; all valid only during the call; bodies come in device order.
 * `every` = steps between frames (the reference's iterPerFrame, main.go:29); frame k is the state before step
 * k*every since the call, numbered from `first_frame` (startFrame, main.go:41-47).  `slots` buffers are in flight at
 * most (the channel capacity); nbe_step blocks when all are waiting for the callback, as the channel send would.
 * nbe_frame_stream_end delivers what is pending and stops the thread; nbe_destroy and nbe_upload imply it. */
typedef void (*nbe_frame_cb)(void *user, int frame, size_t n_alive, const float *xyzm, const float *radius,
                             const uint64_t *ids);
int nbe_frame_stream_begin(nbe_engine *e, int every, int slots, int first_frame, nbe_frame_cb cb, void *user);
int nbe_frame_stream_end(nbe_engine *e);

/* ---- tree inspection (tests) ----
 * Builds the tree of the current state if needed and copies its nodes to the host in depth-first pre-order
 * (children in octant order 0..7, the order of tree.go:221-225): 6 doubles per node = centre of mass x,y,z,
 * mass, cell edge (0 for a one-body leaf; tree.go:60-62 for a cell) and the pre-order index of the first
 * node after its subtree.  One-child chains of the reference's pointer tree are not materialised.
 * *count receives the number of nodes; at most `cap` are written. */
int nbe_tree_nodes(nbe_engine *e, double *out, size_t cap, size_t *count);

/* ---- dominant-kernel timing for benchmarks ----
 * Runs `reps` back-to-back launches of the force kernel of the current mode on the current state without
 * integrating, brackets them with HIP events on the engine's stream and returns the mean milliseconds
 * per launch. */
int nbe_time_force_kernel(nbe_engine *e, int reps, float *ms_per_launch);

/* ---- force pass that allocates nothing ----
 * No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It is the force pass of nbe_accel for the fp32 direct sum
 * (main.go:105-128 without the update loop) on arrays the caller owns: d_pos_mass holds n x (x, y, z, mass) as fp32 in
 * device memory, d_acc receives n x (ax, ay, az, 0) as fp32 in device memory (a = f/m), both on HIP device `device`;
 * the kernels are queued on `hip_stream` (a hipStream_t; NULL = the default stream) and the call returns without
 * waiting.  eps is the softening of nbe_config (0 = the reference's force), G the gravitational constant.
 * The partial sums that nbe_accel keeps in an allocated buffer live in a fixed 20 MiB array in device memory that
 * each GPU holds from the library's first use on it until the process ends; hence at most 262,144 bodies
 * (NBE_E_UNSUPPORTED beyond), and two calls that overlap in time on one GPU (different streams or threads) race on
 * that array: issue them on one stream, or one after the other.  No hipMalloc, hipFree or other allocating call is
 * made, on the device or the host; the caller's current HIP device is the same after the call as before it.  The first
 * call on a device looks the array's address up (hipGetSymbolAddress, which may load the code object): make that
 * call outside graph capture; every later call on that device only queues kernels.  Results equal nbe_accel's for the same bodies bit for bit when the masses are
 * not all equal (nbe_accel switches to a kernel without the mass factor for equal masses: same values to rounding). */
int nbe_accel_direct_noalloc(int device, void *hip_stream, size_t n, const float *d_pos_mass, float *d_acc, double G,
                             double eps);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_ENGINE_H */
