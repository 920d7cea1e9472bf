// Key-range-distributed Barnes-Hut tree: every body is resident on one engine (rank) only.
//
// The reference walks one octree of all bodies with a fan-out over workers (main.go:130-149, tree.go:202-240).  Here
// the octree is never assembled in one place: rank r owns the bodies whose Morton key lies in [split[r], split[r+1]),
// builds the octree of those, and the ranks exchange just enough of their trees for every body to find, on its own
// rank, every node its walk of the global octree would visit:
//   * a subtree whose cell lies wholly inside the rank's key range is that very subtree of the global octree (all
//     bodies of the cell are here): a "fragment".  Fragments are found from the geometry records of the nodes;
//   * the cells above the fragments span ranks.  There are few of them (the ancestors of the splitters); every rank
//     rebuilds them from the all-gathered fragment roots (plan_layout, on the host), adding the children's fp64 sums in
//     octant order exactly as cell_mass_kernel does, so these cells carry the bits of a one-engine build;
//   * for another rank q a fragment is pruned to what q can need: a cell whose opening test (tree.go:216-217) no point
//     of q's bounding box can pass is accepted by every body of q, so nothing below it is ever visited from there.
//     The pruned fragments keep pre-order and get new skip pointers (export_* kernels);
//   * own fragments, imported ones and the spanning cells are laid out as one pre-order node array (assemble_kernel),
//     on which the unchanged walk kernel runs.
// Bodies that cross a splitter are handed to their new owner between steps (classify / pack / append kernels).
// The host moves the bytes between ranks (torch.distributed all-to-all / all-gather in nbody_amd/disttree.py); this file
// holds everything else.
#include "engine_internal.h"
#include "kernels_common.h"
#include "tree_keys.h"
#include "tree_sums.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

struct Segment {          // one fragment in the assembled node array
    int dst;              // first index there
    int count;            // nodes
    int src;              // first index in the source array (own nodes / import buffer)
    int rebase;           // added to every skip pointer: dst - (the index the source's skips are relative to)
    int remote;           // 0: own node array, 1: import buffer
};

struct TopCell {          // a cell that spans ranks
    int dst, skip, level, pad;
    Q4 sum;               // exact sums of its bodies (tree_sums.h)
};

#define BOX_PARTS 128     // workgroups that share the bodies of one fragment (32: the few large fragments then take 0.07 ms longer)
#define EXPORT_BATCH 8    // destinations whose exports are made by the same launches
#define DIST_MAX_WORLD 1024   // ranks (per-destination counters of a workgroup live in LDS)

struct DistState {
    int rank = 0, world = 1;
    int rec_elems = 2;                  // elements of a migration record: position+mass, velocity+index, and with collisions on the radius
    size_t cap = 0;                     // body slots of this engine (== e->n)
    size_t live_hi = 0;                 // slots [0, live_hi) may hold bodies, everything behind is free
    std::vector<uint64_t> split;        // world + 1
    uint64_t* d_split = nullptr;
    uint32_t* d_counts = nullptr;       // the counters of a migration, see counts_starts / counts_cursors
    uint32_t* h_counts = nullptr;       // pinned copy
    void* outbox = nullptr;
    void* inbox = nullptr;
    uint32_t* free_slots = nullptr;     // slots vacated by the last classify
    uint16_t* dest_of = nullptr;        // per slot, left by the last classify: 0 = stays (or no body), d + 1 = moves to rank d
    uint32_t n_free = 0;
    int frag_hint = 0;                  // fragments of the last build (sizes the grid of the next build's box kernel)
    bool classified = false, received = false, built = false, exported = false, assembled = false, have_total = false;
    // fragments of the last build
    nbe_fragment_t* d_frags = nullptr;
    int* d_nfrag = nullptr;             // [0] fragments of the last build, [2..5] a copy of the build's counters (one read-back for both)
    nbe_fragment_t* h_frags = nullptr;  // pinned
    int* h_small = nullptr;             // pinned scratch: counters and sizes
    double* d_aabb = nullptr;           // [MAX_FRAGMENTS][6] boxes of the fragments' bodies, then [MAX_FRAGMENTS][BOX_PARTS][6] partial ones
    double* h_aabb = nullptr;           // pinned [MAX_FRAGMENTS][6]
    std::vector<nbe_fragment_t> my_frags;          // sorted by key
    std::vector<nbe_fragment_t> all_frags;         // rank after rank
    std::vector<int> frag_counts;
    // export
    uint32_t* x_diff = nullptr;         // export work arrays, see export_t
    uint32_t* x_cover = nullptr;
    uint32_t* x_flag = nullptr;
    uint32_t* x_index = nullptr;
    void* export_nodes = nullptr;
    size_t export_cap = 0;
    void* d_xout = nullptr;             // one block: the three arrays below
    void* h_xout = nullptr;             // pinned copy
    size_t xout_bytes = 0;
    unsigned long long* d_export_base = nullptr;   // [0, world] first exported node per destination and the end; then two batch bases
    int* d_export_overflow = nullptr;   // set when the export buffer is too small
    int* d_export_sizes = nullptr;      // [world][fragments of this rank], room for world x NBE_DIST_MAX_FRAGMENTS
    void* d_xin = nullptr;              // one block: d_boxes, d_box_counts, d_frags_sorted
    void* h_xin = nullptr;              // pinned staging of the same
    size_t xin_bytes = 0;
    nbe_fragment_t* d_frags_sorted = nullptr;      // this rank's fragments in key order (what the export's kernels read)
    int* d_frag_bodies = nullptr;       // [MAX_FRAGMENTS][2]: first body and one past the last body of every fragment
    double* d_boxes = nullptr;          // world x (MAX_BOXES + 1) x 6: a rank's boxes, then the box around them all
    int* d_box_counts = nullptr;        // world
    // import + assembled array
    void* import_nodes = nullptr;
    size_t import_cap = 0;
    void* let_nodes = nullptr;
    size_t let_cap = 0;
    int* let_counters = nullptr;        // [0] live bodies of this rank, [1] nodes of the assembled array: the head of d_ain
    void* d_ain = nullptr;              // nbe_dist_assemble's input block on the device
    void* h_ain = nullptr;              // pinned staging of the same
    size_t ain_bytes = 0;
    size_t seg_cap = 0, top_cap = 0;
};

// Layout of the counters of a migration (32-word stretches, so that the words the pack pass hammers with atomics share no
// cache line with the ones it only reads): [0, world) the leavers per destination as classify_kernel counted them;
// [A, A + world) where each destination's stretch of the outbox starts; [B, B + world] running cursors within the
// stretches, then the freed slots.
__host__ __device__ static inline int counts_starts(int world) { return (world + 31) / 32 * 32; }
__host__ __device__ static inline int counts_cursors(int world) { return 2 * ((world + 31) / 32 * 32); }
static inline size_t counts_words(int world) { return (size_t)counts_cursors(world) + (size_t)world + 1; }

static int dfree(void** p)
{
    if (*p) { NBE_HIP(hipFree(*p)); *p = nullptr; }
    return NBE_OK;
}

void dist_destroy(nbe_engine* e)
{
    DistState* d = e->dist;
    if (!d) return;
    void** dev[] = {(void**)&d->d_split, (void**)&d->d_counts, &d->outbox, &d->inbox, (void**)&d->free_slots, (void**)&d->dest_of, (void**)&d->d_frags,
                    (void**)&d->d_nfrag, (void**)&d->d_aabb, (void**)&d->x_diff, (void**)&d->x_cover, (void**)&d->x_flag,
                    (void**)&d->x_index, &d->export_nodes, &d->d_xout, &d->d_xin,
                    (void**)&d->d_frag_bodies, &d->import_nodes, &d->let_nodes, &d->d_ain};
    for (void** p : dev) (void)dfree(p);
    void* host[] = {d->h_counts, d->h_frags, d->h_small, d->h_aabb, d->h_ain, d->h_xout, d->h_xin};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    delete d;
    e->dist = nullptr;
}

static int check_dist(nbe_engine* e, const char* name)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (!e->dist) { nbe_set_error("%s needs an engine set up with nbe_dist_upload", name); return NBE_E_INVALID; }
    NBE_HIP(hipSetDevice(e->cfg.device));
    return NBE_OK;
}

static void bound_lo(const nbe_engine* e, double lo[3], double* W)
{
    *W = e->cfg.bound_width;
    for (int c = 0; c < 3; c++) lo[c] = e->cfg.bound_center[c] - 0.5 * *W;
}

// ---- setting up ------------------------------------------------------------------------------------------------
// Run-wide body indices: what `orig` holds on a distributed engine (the plain engine keeps upload indices there).
__global__ __launch_bounds__(256) void add_index_kernel(uint32_t* __restrict__ orig0, uint32_t* __restrict__ orig1, int n, uint32_t first)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    orig0[i] += first;
    orig1[i] += first;
}

extern "C" int nbe_dist_upload(nbe_engine* e, int rank, int world, size_t capacity, size_t n, const double* pos, const double* vel,
                               const double* mass, uint64_t first_index)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (e->multi) { nbe_set_error("nbe_dist_upload on an engine with num_devices > 1"); return NBE_E_UNSUPPORTED; }
    if (e->cfg.mode != NBE_MODE_TREE || e->cfg.reference_quirks || e->cfg.walk_criterion != NBE_WALK_PER_BODY) {
        nbe_set_error("the distributed tree needs tree mode, the per-body criterion and no reference_quirks");
        return NBE_E_UNSUPPORTED;
    }
    if (world < 1 || world > DIST_MAX_WORLD || rank < 0 || rank >= world) { nbe_set_error("bad rank %d of %d", rank, world); return NBE_E_INVALID; }
    if (capacity < n || capacity == 0) { nbe_set_error("capacity %zu < n %zu (or 0)", capacity, n); return NBE_E_INVALID; }
    if (capacity > (1ull << 26) || first_index + n > 0xffffffffull) { nbe_set_error("at most 2^26 bodies per rank and 2^32 in all"); return NBE_E_UNSUPPORTED; }
    if (n > 0 && (!pos || !vel || !mass)) { nbe_set_error("nbe_dist_upload: NULL array"); return NBE_E_INVALID; }
    for (size_t i = 0; i < n; i++)
        if (!(mass[i] >= 0.0)) { nbe_set_error("mass[%zu] is negative or NaN", i); return NBE_E_INVALID; }
    NBE_HIP(hipSetDevice(e->cfg.device));
    dist_destroy(e);
    DistState* d = new (std::nothrow) DistState();
    if (!d) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
    e->dist = d;
    e->cfg.cell_sums = NBE_SUMS_EXACT;                // sums that do not depend on the partition of the bodies
    e->tw.cap = 0;                                   // the tree workspace is laid out anew (sums by tree, scans over nodes)
    d->rank = rank; d->world = world; d->cap = capacity; d->live_hi = n;
    // free slots hold massless bodies parked outside the bound: the sort sends them behind the live ones
    double lo[3], W;
    bound_lo(e, lo, &W);
    int rc = NBE_OK;
    try {
        std::vector<double> hp(3 * capacity), hv(3 * capacity, 0.0), hm(capacity, 0.0);
        if (n) { memcpy(hp.data(), pos, 3 * n * sizeof(double)); memcpy(hv.data(), vel, 3 * n * sizeof(double)); memcpy(hm.data(), mass, n * sizeof(double)); }
        for (size_t i = n; i < capacity; i++)
            for (int c = 0; c < 3; c++) hp[3 * i + c] = lo[c] + 2.0 * W;
        rc = engine_upload_state(e, capacity, hp.data(), hv.data(), hm.data(), nullptr);
    } catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); rc = NBE_E_NOMEM; }
    if (rc != NBE_OK) return rc;
    e->small_max = 0;                                // one code path for every size
    hipLaunchKernelGGL(add_index_kernel, dim3(cdiv(capacity, 256)), dim3(256), 0, e->stream, e->orig[0], e->orig[1], (int)capacity,
                       (uint32_t)first_index);
    NBE_HIP(hipGetLastError());
    NBE_TRY(collide_workspace_reserve(e));           // the geometry records of the nodes live in the collision workspace
    const size_t W2 = (size_t)world;
    const size_t nodes2 = 2 * capacity + 2;
    const size_t node_bytes = e->force64 ? sizeof(NodeD) : sizeof(NodeF);
    NBE_HIP(hipMalloc((void**)&d->d_split, (W2 + 1) * 8));
    NBE_HIP(hipMalloc((void**)&d->d_counts, counts_words(world) * 4));
    NBE_HIP(hipHostMalloc((void**)&d->h_counts, (2 * W2 + 1) * 4, hipHostMallocDefault));
    d->rec_elems = e->cfg.collisions ? 3 : 2;
    NBE_HIP(hipMalloc(&d->outbox, capacity * (size_t)d->rec_elems * e->elem));
    NBE_HIP(hipMalloc(&d->inbox, capacity * (size_t)d->rec_elems * e->elem));
    NBE_HIP(hipMalloc((void**)&d->free_slots, capacity * 4));
    NBE_HIP(hipMalloc((void**)&d->dest_of, capacity * sizeof(uint16_t)));
    NBE_HIP(hipMalloc((void**)&d->d_frags, NBE_DIST_MAX_FRAGMENTS * sizeof(nbe_fragment_t)));
    NBE_HIP(hipMalloc((void**)&d->d_nfrag, 8 * sizeof(int)));
    NBE_HIP(hipHostMalloc((void**)&d->h_frags, NBE_DIST_MAX_FRAGMENTS * sizeof(nbe_fragment_t), hipHostMallocDefault));
    NBE_HIP(hipHostMalloc((void**)&d->h_small, (16 + (W2 + 1) * 2 + W2 * NBE_DIST_MAX_FRAGMENTS) * sizeof(int), hipHostMallocDefault));
    NBE_HIP(hipMalloc((void**)&d->d_frag_bodies, (size_t)NBE_DIST_MAX_FRAGMENTS * 2 * sizeof(int)));
    NBE_HIP(hipMalloc((void**)&d->d_aabb, (size_t)NBE_DIST_MAX_FRAGMENTS * (BOX_PARTS + 1) * 6 * sizeof(double)));
    NBE_HIP(hipHostMalloc((void**)&d->h_aabb, (size_t)NBE_DIST_MAX_FRAGMENTS * 6 * sizeof(double), hipHostMallocDefault));
    // export work arrays (see export_t): two packed words per node and their scan, a byte and a 64-bit word per node,
    // and the per-block counts of eight destinations with their scan
    NBE_HIP(hipMalloc((void**)&d->x_diff, 2 * nodes2 * 4));
    NBE_HIP(hipMalloc((void**)&d->x_cover, 2 * nodes2 * 4));
    NBE_HIP(hipMalloc((void**)&d->x_flag, (nodes2 / 4 + 2 * ((size_t)EXPORT_BATCH * (nodes2 / 256 + 1) + 2) + 8) * 4));
    NBE_HIP(hipMalloc((void**)&d->x_index, nodes2 * 8));
    // With few bodies per rank every other rank needs most of a rank's tree; with many, a small part of it.
    // (measured at 2.1M bodies per rank of a 16.7M-body run: 1.7 to 3.4M imported nodes, a third to two thirds of a rank's
    // own; the share falls as ranks grow, and a buffer that is too small is reported, not overrun)
    size_t others = W2 > 5 ? 4 : (W2 > 1 ? W2 - 1 : 1);
    if (capacity >= (32u << 20)) others = 1;
    else if (capacity >= (8u << 20) && others > 2) others = 2;
    d->export_cap = nodes2 * others;
    d->import_cap = nodes2 * others;
    d->let_cap = nodes2 + d->import_cap + (size_t)NBE_DIST_MAX_FRAGMENTS * W2;
    if (d->let_cap * node_bytes >= (1ull << 32) && !e->force64) d->let_cap = ((1ull << 32) - 64) / node_bytes;   // the walk's 32-bit byte offsets
    NBE_HIP(hipMalloc(&d->export_nodes, d->export_cap * node_bytes));
    NBE_HIP(hipMalloc(&d->import_nodes, d->import_cap * node_bytes));
    NBE_HIP(hipMalloc(&d->let_nodes, (d->let_cap + 2) * node_bytes));
    NBE_HIP(hipMemsetAsync(d->let_nodes, 0, (d->let_cap + 2) * node_bytes, e->stream));
    // The export's small results live in one block (one memset, one copy back): [first node per destination and the end,
    // two batch bases | overflow flag | sizes of the fragments per destination]; its small inputs in another (one copy
    // in): [the ranks' boxes | how many of them count | this rank's fragments in key order].
    d->xout_bytes = (W2 + 3) * sizeof(unsigned long long) + 2 * sizeof(int) + W2 * NBE_DIST_MAX_FRAGMENTS * sizeof(int);
    NBE_HIP(hipMalloc(&d->d_xout, d->xout_bytes));
    NBE_HIP(hipHostMalloc(&d->h_xout, d->xout_bytes, hipHostMallocDefault));
    d->d_export_base = (unsigned long long*)d->d_xout;
    d->d_export_overflow = (int*)(d->d_export_base + (W2 + 3));
    d->d_export_sizes = d->d_export_overflow + 2;
    const size_t boxes_bytes = W2 * (NBE_DIST_MAX_BOXES + 1) * 6 * sizeof(double), counts_bytes = (W2 * sizeof(int) + 7) / 8 * 8;
    d->xin_bytes = boxes_bytes + counts_bytes + (size_t)NBE_DIST_MAX_FRAGMENTS * sizeof(nbe_fragment_t);
    NBE_HIP(hipMalloc(&d->d_xin, d->xin_bytes));
    NBE_HIP(hipHostMalloc(&d->h_xin, d->xin_bytes, hipHostMallocDefault));
    d->d_boxes = (double*)d->d_xin;
    d->d_box_counts = (int*)((char*)d->d_xin + boxes_bytes);
    d->d_frags_sorted = (nbe_fragment_t*)((char*)d->d_xin + boxes_bytes + counts_bytes);
    d->seg_cap = W2 * NBE_DIST_MAX_FRAGMENTS;
    d->top_cap = W2 * NBE_DIST_MAX_FRAGMENTS;
    // what nbe_dist_assemble hands the device, in one block: the walk's two counters, then segments, first work items, cells
    d->ain_bytes = 64 + d->seg_cap * (sizeof(Segment) + sizeof(int)) + d->top_cap * sizeof(TopCell) + 64;
    NBE_HIP(hipMalloc(&d->d_ain, d->ain_bytes));
    NBE_HIP(hipMemsetAsync(d->d_ain, 0, 64, e->stream));
    NBE_HIP(hipHostMalloc(&d->h_ain, d->ain_bytes, hipHostMallocDefault));
    d->let_counters = (int*)d->d_ain;
    d->split.assign(W2 + 1, 0);
    d->split[W2] = 1ull << 63;
    for (size_t r = 1; r < W2; r++) d->split[r] = rank == 0 ? (1ull << 63) : 0;   // until splitters are set: rank 0 would own everything
    NBE_HIP(hipStreamSynchronize(e->stream));
    e->stats.n = capacity;
    e->stats.n_alive = n;
    return NBE_OK;
}

extern "C" int nbe_dist_buffers(nbe_engine* e, nbe_dist_buffers_t* out)
{
    NBE_TRY(check_dist(e, "nbe_dist_buffers"));
    if (!out) { nbe_set_error("out is NULL"); return NBE_E_INVALID; }
    DistState* d = e->dist;
    out->outbox = d->outbox; out->inbox = d->inbox;
    out->export_nodes = d->export_nodes; out->import_nodes = d->import_nodes;
    out->slots = d->cap; out->export_cap = d->export_cap; out->import_cap = d->import_cap;
    out->record_bytes = (uint32_t)((size_t)d->rec_elems * e->elem);
    out->node_bytes = (uint32_t)(e->force64 ? sizeof(NodeD) : sizeof(NodeF));
    return NBE_OK;
}

// Positions the keys are made from: the force kernels' copy (fp32 for NBE_PREC_MIXED), as the sort uses.
static const void* key_positions(const nbe_engine* e)
{
    return e->cfg.precision == NBE_PREC_MIXED ? (const void*)e->mirror[e->cur] : (const void*)e->posm[e->cur];
}

__global__ __launch_bounds__(256) void sample_keys_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ counters, int want,
                                                          uint64_t* __restrict__ out, int* __restrict__ got)
{
    const int na = counters[COUNTER_SORTED_ALIVE];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int cnt = na < want ? na : want;
    if (k == 0) { got[0] = cnt; got[1] = na; }
    if (k >= cnt) return;
    // evenly spaced, centred in their strides
    const long long idx = ((2ll * k + 1) * na) / (2ll * cnt);
    out[k] = keys[idx];
}

extern "C" int nbe_dist_sample_keys(nbe_engine* e, size_t max_samples, uint64_t* keys_out, size_t* count, size_t* n_live)
{
    NBE_TRY(check_dist(e, "nbe_dist_sample_keys"));
    DistState* d = e->dist;
    if (!keys_out || !count) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    e->n = d->live_hi;                                // only the slots that can hold bodies (see nbe_dist_build)
    if (max_samples > e->n) max_samples = e->n;
    if (max_samples == 0) { *count = 0; if (n_live) *n_live = 0; return NBE_OK; }
    e->keys_sorted = false;
    NBE_TRY(tree_sort_keys(e));                       // keys of the current state, sorted, in tw.keys[0]
    e->keys_sorted = false;                           // (the state changes before the build: it sorts again)
    uint64_t* tmp = (uint64_t*)d->x_diff;             // 2 cap + 2 words of 4 bytes: room for cap keys
    hipLaunchKernelGGL(sample_keys_kernel, dim3(cdiv(max_samples ? max_samples : 1, 256)), dim3(256), 0, e->stream, e->tw.keys[0],
                       e->tw.counters, (int)max_samples, tmp, d->d_nfrag);
    NBE_HIP(hipGetLastError());
    NBE_HIP(hipMemcpyAsync(d->h_small, d->d_nfrag, 2 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    const size_t got = (size_t)d->h_small[0];
    if (got) NBE_HIP(hipMemcpy(keys_out, tmp, got * 8, hipMemcpyDeviceToHost));
    *count = got;
    if (n_live) *n_live = (size_t)d->h_small[1];
    return NBE_OK;
}

extern "C" int nbe_dist_set_splitters(nbe_engine* e, const uint64_t* split)
{
    NBE_TRY(check_dist(e, "nbe_dist_set_splitters"));
    DistState* d = e->dist;
    if (!split) { nbe_set_error("split is NULL"); return NBE_E_INVALID; }
    if (split[0] != 0 || split[d->world] != (1ull << 63)) { nbe_set_error("split[0] must be 0 and split[world] 2^63"); return NBE_E_INVALID; }
    for (int r = 0; r < d->world; r++)
        if (split[r] > split[r + 1]) { nbe_set_error("splitters must not decrease"); return NBE_E_INVALID; }
    d->split.assign(split, split + d->world + 1);
    if (e->cfg.collisions) {
        // A collision search (tree.go:245-281) stays inside the cells around its body that are at most four radii wide.
        // Splitters are moved down to the border of a cell at least that wide (for the largest radius allowed), so that
        // every such cell lies on one rank and a rank's own tree shows a body all it can touch.  Every rank rounds alike.
        int depth = 0;
        while (depth < 21 && ldexp(e->cfg.bound_width, -(depth + 1)) >= 4.000001 * e->cw.radius_cap) depth++;
        const uint64_t mask = depth >= 21 ? ~0ull : ~((1ull << (3 * (21 - depth))) - 1ull);
        for (int r = 1; r < d->world; r++) d->split[(size_t)r] &= mask;
    }
    NBE_HIP(hipMemcpyAsync(d->d_split, d->split.data(), (d->world + 1) * 8, hipMemcpyHostToDevice, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    return NBE_OK;
}

extern "C" int nbe_dist_set_radius(nbe_engine* e, const double* radius, double radius_cap)
{
    NBE_TRY(check_dist(e, "nbe_dist_set_radius"));
    DistState* d = e->dist;
    if (!e->cfg.collisions) { nbe_set_error("nbe_dist_set_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
    if (e->stats.steps != 0 || d->classified || d->built) { nbe_set_error("nbe_dist_set_radius comes right after nbe_dist_upload"); return NBE_E_INVALID; }
    if (!(radius_cap > 0.0) || !std::isfinite(radius_cap)) { nbe_set_error("radius_cap must be positive"); return NBE_E_INVALID; }
    const size_t n = d->live_hi;                      // the bodies of the upload, still in upload order
    if (radius) {
        for (size_t i = 0; i < n; i++)
            if (!(radius[i] >= 0.0) || radius[i] > radius_cap) { nbe_set_error("radius[%zu] is negative, NaN or above the cap", i); return NBE_E_INVALID; }
        if (n) NBE_HIP(hipMemcpyAsync(e->radius[e->cur], radius, n * sizeof(double), hipMemcpyHostToDevice, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
    }
    e->cw.radius_cap = radius_cap;
    return NBE_OK;
}

extern "C" int nbe_dist_set_total_mass(nbe_engine* e, double total_mass)
{
    NBE_TRY(check_dist(e, "nbe_dist_set_total_mass"));
    if (!(total_mass >= 0.0) || !std::isfinite(total_mass)) { nbe_set_error("total_mass must be finite and >= 0"); return NBE_E_INVALID; }
    // (fp32 state holds rounded masses: a little headroom covers their sum)
    tree_set_total_mass(e, total_mass * (1.0 + 0x1p-20));
    e->dist->have_total = true;
    return NBE_OK;
}

// ---- migration ---------------------------------------------------------------------------------------------------
// Owner of a key: the number of splitters split[1 .. world-1] that are <= key.
__device__ __forceinline__ int owner_of(const uint64_t* __restrict__ split, int world, uint64_t key)
{
    int lo = 0, hi = world - 1;                       // answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (split[mid] <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Thousands of bodies leave a rank per step, for a handful of destinations: counted one by one in device memory they
// queue up behind a few addresses (~10 ns each: 0.1 ms per step, 7 ms when every body moves).  A workgroup therefore
// counts in LDS and adds its totals once.
template <typename R4>
__global__ __launch_bounds__(256) void classify_kernel(const R4* __restrict__ posk, int slots, double lox, double loy, double loz, double W,
                                                       const uint64_t* __restrict__ split, int world, uint64_t own_lo, uint64_t own_hi,
                                                       uint32_t* __restrict__ counts, uint16_t* __restrict__ dest_of)
{
    __shared__ uint32_t cnt[DIST_MAX_WORLD];
    for (int t = threadIdx.x; t < world; t += 256) cnt[t] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < slots) {
        const R4 p = posk[i];
        const uint64_t key = body_key((double)p.x, (double)p.y, (double)p.z, lox, loy, loz, W);
        // (nearly every body stays: no search for those)
        uint16_t to = 0;
        if (key != KEY_DEAD && !(key >= own_lo && key < own_hi)) {
            const int dest = owner_of(split, world, key);
            atomicAdd(&cnt[dest], 1u);
            to = (uint16_t)(dest + 1);
        }
        dest_of[i] = to;                              // the pack pass reads this instead of making the key again
    }
    __syncthreads();
    for (int t = threadIdx.x; t < world; t += 256)
        if (cnt[t]) atomicAdd(&counts[t], cnt[t]);
}

// A record is the body's position+mass element followed by its velocity element, whose unused fourth component carries
// the run-wide index (as bits in fp32 state, as a value in fp64 state).
__device__ __forceinline__ void put_index(float4& v, uint32_t idx) { v.w = __uint_as_float(idx); }
__device__ __forceinline__ void put_index(double4& v, uint32_t idx) { v.w = (double)idx; }
__device__ __forceinline__ uint32_t get_index(const float4& v) { return __float_as_uint(v.w); }
__device__ __forceinline__ uint32_t get_index(const double4& v) { return (uint32_t)v.w; }
// (the radius, a double, rides in the third element of a record: as two words of an fp32 element)
__device__ __forceinline__ void put_radius(float4& r, double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    r.x = __uint_as_float((unsigned)b); r.y = __uint_as_float((unsigned)(b >> 32));
}
__device__ __forceinline__ void put_radius(double4& r, double v) { r.x = v; }
__device__ __forceinline__ double get_radius(const float4& r)
{
    return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(r.y) << 32) | __float_as_uint(r.x)));
}
__device__ __forceinline__ double get_radius(const double4& r) { return r.x; }

// (layout of `counts`: counts_starts / counts_cursors near the top of the file)

__global__ __launch_bounds__(64) void pack_starts_kernel(uint32_t* __restrict__ counts, int world)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0;
    for (int q = 0; q < world; q++) { counts[counts_starts(world) + q] = run; run += counts[q]; }
}

// A workgroup packs the leavers among 1024 slots (four per thread): places are handed out per workgroup, ranks within
// the workgroup in LDS, one reservation per destination in memory.
#define PACK_SLOTS 4
template <typename S4>
__global__ __launch_bounds__(256) void pack_kernel(const uint16_t* __restrict__ dest_of, S4* __restrict__ posm, S4* __restrict__ vel,
                                                   float4* __restrict__ mirror, const uint32_t* __restrict__ orig, int slots, double lox,
                                                   double loy, double loz, double W, int world, uint32_t* __restrict__ counts,
                                                   S4* __restrict__ outbox, uint32_t* __restrict__ free_slots,
                                                   const double* __restrict__ radius, int rec)
{
    __shared__ uint32_t cnt[DIST_MAX_WORLD + 1], base[DIST_MAX_WORLD + 1];      // [world]: all destinations together
    int dest[PACK_SLOTS];
    bool any = false;
#pragma unroll
    for (int k = 0; k < PACK_SLOTS; k++) {
        const int i = (blockIdx.x * PACK_SLOTS + k) * 256 + threadIdx.x;
        dest[k] = i < slots ? (int)dest_of[i] - 1 : -1;
        any = any || dest[k] >= 0;
    }
    if (__syncthreads_or(any) == 0) return;           // nobody of these slots leaves
    for (int t = threadIdx.x; t <= world; t += 256) cnt[t] = 0;
    __syncthreads();
    uint32_t mine[PACK_SLOTS], mine_all[PACK_SLOTS];
#pragma unroll
    for (int k = 0; k < PACK_SLOTS; k++)
        if (dest[k] >= 0) {
            mine[k] = atomicAdd(&cnt[dest[k]], 1u);
            mine_all[k] = atomicAdd(&cnt[world], 1u);
        }
    __syncthreads();
    const int S = counts_starts(world), C = counts_cursors(world);
    for (int t = threadIdx.x; t <= world; t += 256)
        if (cnt[t]) base[t] = (t < world ? counts[S + t] : 0u) + atomicAdd(&counts[C + t], cnt[t]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PACK_SLOTS; k++) {
        if (dest[k] < 0) continue;
        const int i = (blockIdx.x * PACK_SLOTS + k) * 256 + threadIdx.x;
        const uint32_t at = base[dest[k]] + mine[k];
        S4 p = posm[i], v = vel[i];
        put_index(v, orig[i]);
        outbox[(size_t)rec * at] = p;
        outbox[(size_t)rec * at + 1] = v;
        if (radius) { S4 r; r.x = 0; r.y = 0; r.z = 0; r.w = 0; put_radius(r, radius[i]); outbox[(size_t)rec * at + 2] = r; }
        free_slots[base[world] + mine_all[k]] = (uint32_t)i;
        // the slot now holds a massless body parked outside the bound (and, where the velocity element carries an index
        // in a record, the place of its record in the outbox: nbe_dist_unclassify brings the body back from there)
        p.x = lox + 2.0 * W; p.y = loy + 2.0 * W; p.z = loz + 2.0 * W; p.w = 0;
        v.x = 0; v.y = 0; v.z = 0; put_index(v, at);
        posm[i] = p;
        vel[i] = v;
        if (mirror) mirror[i] = make_float4((float)p.x, (float)p.y, (float)p.z, 0.0f);
    }
}

// Undoes pack_kernel: every slot vacated by the last classify takes its body back from the outbox.
template <typename S4>
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ free_slots, int n_free, const S4* __restrict__ outbox,
                                                     S4* __restrict__ posm, S4* __restrict__ vel, float4* __restrict__ mirror, int rec)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_free) return;
    const uint32_t slot = free_slots[j];
    const uint32_t at = get_index(vel[slot]);
    const S4 p = outbox[(size_t)rec * at];
    S4 v = outbox[(size_t)rec * at + 1];
    v.w = 0;
    posm[slot] = p;
    vel[slot] = v;
    if (mirror) mirror[slot] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}

template <typename R4, typename S4>
static int classify_t(nbe_engine* e, uint32_t* send_counts)
{
    DistState* d = e->dist;
    const int W = d->world;
    double lo[3], width;
    bound_lo(e, lo, &width);
    const int slots = (int)d->live_hi;
    const R4* posk = (const R4*)key_positions(e);
    NBE_HIP(hipMemsetAsync(d->d_counts, 0, counts_words(W) * 4, e->stream));
    if (slots > 0) {
        hipLaunchKernelGGL((classify_kernel<R4>), dim3(cdiv(slots, 256)), dim3(256), 0, e->stream, posk, slots, lo[0], lo[1], lo[2], width,
                           d->d_split, W, d->split[d->rank], d->split[d->rank + 1], d->d_counts, d->dest_of);
        hipLaunchKernelGGL(pack_starts_kernel, dim3(1), dim3(64), 0, e->stream, d->d_counts, W);
        // the pack pass needs nothing from the host (a workgroup without leavers ends at once), so it is queued before
        // the counts are waited for
        NBE_TRY(frame_stream_guard_write(e, e->cur));
        hipLaunchKernelGGL((pack_kernel<S4>), dim3(cdiv(slots, 256 * PACK_SLOTS)), dim3(256), 0, e->stream, d->dest_of, (S4*)e->posm[e->cur],
                           (S4*)e->vel[e->cur], e->cfg.precision == NBE_PREC_MIXED ? e->mirror[e->cur] : (float4*)nullptr,
                           e->orig[e->orig_cur], slots, lo[0], lo[1], lo[2], width, W, d->d_counts, (S4*)d->outbox, d->free_slots,
                           (const double*)e->radius[e->cur], d->rec_elems);
        NBE_HIP(hipGetLastError());
    }
    NBE_HIP(hipMemcpyAsync(d->h_counts, d->d_counts, (size_t)W * 4, hipMemcpyDeviceToHost, e->stream));
    // (this wait also covers a caller that reads the outbox on streams the engine knows nothing of)
    NBE_HIP(hipStreamSynchronize(e->stream));
    uint32_t off = 0;
    for (int r = 0; r < W; r++) { send_counts[r] = d->h_counts[r]; off += d->h_counts[r]; }
    d->n_free = off;
    if ((size_t)off > d->cap) { nbe_set_error("internal: %u leavers of %zu slots", off, d->cap); return NBE_E_INVALID; }
    return NBE_OK;
}

extern "C" int nbe_dist_classify(nbe_engine* e, uint32_t* send_counts)
{
    NBE_TRY(check_dist(e, "nbe_dist_classify"));
    if (!send_counts) { nbe_set_error("send_counts is NULL"); return NBE_E_INVALID; }
    DistState* d = e->dist;
    int rc = NBE_E_INVALID;
    switch (e->cfg.precision) {
    case NBE_PREC_F32: rc = classify_t<float4, float4>(e, send_counts); break;
    case NBE_PREC_MIXED: rc = classify_t<float4, double4>(e, send_counts); break;
    case NBE_PREC_F64: rc = classify_t<double4, double4>(e, send_counts); break;
    }
    if (rc == NBE_OK) { d->classified = true; d->received = false; d->built = false; d->exported = false; d->assembled = false; e->prepared = false; e->keys_sorted = false; }
    return rc;
}

extern "C" int nbe_dist_room(nbe_engine* e, uint64_t* room)
{
    NBE_TRY(check_dist(e, "nbe_dist_room"));
    if (!room) { nbe_set_error("room is NULL"); return NBE_E_INVALID; }
    const DistState* d = e->dist;
    *room = (uint64_t)(d->cap - d->live_hi) + d->n_free;
    return NBE_OK;
}

extern "C" int nbe_dist_unclassify(nbe_engine* e)
{
    NBE_TRY(check_dist(e, "nbe_dist_unclassify"));
    DistState* d = e->dist;
    if (!d->classified || d->received) { nbe_set_error("nbe_dist_unclassify comes between nbe_dist_classify and nbe_dist_receive"); return NBE_E_INVALID; }
    if (d->n_free > 0) {
        float4* mirror = e->cfg.precision == NBE_PREC_MIXED ? e->mirror[e->cur] : nullptr;
        if (e->state64)
            hipLaunchKernelGGL((unpack_kernel<double4>), dim3(cdiv(d->n_free, 256)), dim3(256), 0, e->stream, d->free_slots, (int)d->n_free,
                               (const double4*)d->outbox, (double4*)e->posm[e->cur], (double4*)e->vel[e->cur], mirror, d->rec_elems);
        else
            hipLaunchKernelGGL((unpack_kernel<float4>), dim3(cdiv(d->n_free, 256)), dim3(256), 0, e->stream, d->free_slots, (int)d->n_free,
                               (const float4*)d->outbox, (float4*)e->posm[e->cur], (float4*)e->vel[e->cur], mirror, d->rec_elems);
        NBE_HIP(hipGetLastError());
    }
    d->n_free = 0;
    d->classified = false;
    return NBE_OK;
}

template <typename S4>
__global__ __launch_bounds__(256) void append_kernel(const S4* __restrict__ inbox, int k, const uint32_t* __restrict__ free_slots,
                                                     int n_free, int tail, S4* __restrict__ posm, S4* __restrict__ vel,
                                                     float4* __restrict__ mirror, uint32_t* __restrict__ orig,
                                                     double* __restrict__ radius, int rec)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    const int slot = j < n_free ? (int)free_slots[j] : tail + (j - n_free);
    const S4 p = inbox[(size_t)rec * j];
    S4 v = inbox[(size_t)rec * j + 1];
    if (radius) radius[slot] = get_radius(inbox[(size_t)rec * j + 2]);
    orig[slot] = get_index(v);
    v.w = 0;
    posm[slot] = p;
    vel[slot] = v;
    if (mirror) mirror[slot] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}

extern "C" int nbe_dist_receive(nbe_engine* e, size_t k)
{
    NBE_TRY(check_dist(e, "nbe_dist_receive"));
    DistState* d = e->dist;
    if (!d->classified) { nbe_set_error("nbe_dist_receive before nbe_dist_classify"); return NBE_E_INVALID; }
    const size_t extra = k > d->n_free ? k - d->n_free : 0;
    if (d->live_hi + extra > d->cap) {
        nbe_set_error("rank %d: %zu bodies arrive but only %zu slots are free (capacity %zu)", d->rank, k, d->cap - d->live_hi + d->n_free, d->cap);
        return NBE_E_NOMEM;
    }
    if (k > 0) {
        float4* mirror = e->cfg.precision == NBE_PREC_MIXED ? e->mirror[e->cur] : nullptr;
        if (e->state64)
            hipLaunchKernelGGL((append_kernel<double4>), dim3(cdiv(k, 256)), dim3(256), 0, e->stream, (const double4*)d->inbox, (int)k,
                               d->free_slots, (int)d->n_free, (int)d->live_hi, (double4*)e->posm[e->cur], (double4*)e->vel[e->cur], mirror,
                               e->orig[e->orig_cur], e->radius[e->cur], d->rec_elems);
        else
            hipLaunchKernelGGL((append_kernel<float4>), dim3(cdiv(k, 256)), dim3(256), 0, e->stream, (const float4*)d->inbox, (int)k,
                               d->free_slots, (int)d->n_free, (int)d->live_hi, (float4*)e->posm[e->cur], (float4*)e->vel[e->cur], mirror,
                               e->orig[e->orig_cur], e->radius[e->cur], d->rec_elems);
        NBE_HIP(hipGetLastError());
    }
    d->live_hi += extra;
    d->n_free = d->n_free > k ? d->n_free - (uint32_t)k : 0;     // (slots still vacant are parked bodies: the sort drops them)
    d->received = true;
    return NBE_OK;
}

size_t dist_live_slots(const nbe_engine* e) { return e->dist ? e->dist->live_hi : 0; }

// ---- build: local tree, fragments, bounding box ---------------------------------------------------------------------
// Is the depth-`level` cell around `key` wholly inside [lo, hi)?  (A body, level 255, always is: it is here because its
// key is.)
__device__ __forceinline__ bool cell_inside(uint64_t key, int level, uint64_t lo, uint64_t hi)
{
    if (level >= 21) return true;
    const uint64_t size = 1ull << (3 * (21 - level));
    const uint64_t cmin = key & ~(size - 1ull);
    return cmin >= lo && cmin + (size - 1ull) < hi;
}

// A node roots a fragment iff its cell is inside the rank's key range and the cell of the node above is not.
__global__ __launch_bounds__(256) void fragments_kernel(const CNode* __restrict__ cn, const int* __restrict__ counters,
                                                        const Q4* __restrict__ prefixq, uint64_t lo, uint64_t hi,
                                                        nbe_fragment_t* __restrict__ frags, int* __restrict__ nfrag,
                                                        int* __restrict__ frag_bodies)
{
    const int total = counters[1];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const CNode c = cn[p];
    const bool leaf = c.body >= 0;
    if (!leaf && !cell_inside(c.key, c.level, lo, hi)) return;
    if (c.plevel != 255 && cell_inside(c.key, c.plevel, lo, hi)) return;
    const int at = atomicAdd(nfrag, 1);
    if (at >= NBE_DIST_MAX_FRAGMENTS) return;
    nbe_fragment_t f;
    f.key = c.key;
    f.level = leaf ? 255 : (int)c.level;
    f.start = p;
    f.nodes = c.skip - p;
    f.reserved = at;                                  // its place in this (unordered) list: where its box will be
    // its bodies: in pre-order the first children of the cells below follow the root one by one down to the first leaf,
    // and the last node of the subtree is its last leaf
    int first = p;
    while (cn[first].body < 0) first++;
    const int b0 = cn[first].body, b1 = cn[c.skip - 1].body + 1;
    frag_bodies[2 * at] = b0;
    frag_bodies[2 * at + 1] = b1;
    // its exact sums: a difference of the prefix sums over its (consecutive) bodies
    const Q4 q = q4_sub(prefixq[b1], prefixq[b0]);
    f.sum[0] = q.m.lo; f.sum[1] = q.m.hi; f.sum[2] = q.x.lo; f.sum[3] = q.x.hi;
    f.sum[4] = q.y.lo; f.sum[5] = q.y.hi; f.sum[6] = q.z.lo; f.sum[7] = q.z.hi;
    frags[at] = f;
}

// Box of the bodies of every fragment: BOX_PARTS workgroups share a fragment's bodies, then one thread per fragment and
// coordinate reduces.  The grid covers the largest possible number of fragments.
template <typename R4>
__global__ __launch_bounds__(256) void box_partial_kernel(const R4* __restrict__ pos, const int* __restrict__ frag_bodies,
                                                          const int* __restrict__ nfrag, double* __restrict__ part)
{
    const int g = blockIdx.y;
    if (g >= *nfrag || g >= NBE_DIST_MAX_FRAGMENTS) return;
    const int first = frag_bodies[2 * g], last = frag_bodies[2 * g + 1];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = first + blockIdx.x * 256 + threadIdx.x; i < last; i += BOX_PARTS * 256) {
        const R4 p = pos[i];
        const double v[3] = {(double)p.x, (double)p.y, (double)p.z};
        for (int c = 0; c < 3; c++) { mn[c] = fmin(mn[c], v[c]); mx[c] = fmax(mx[c], v[c]); }
    }
    __shared__ double s[6][256];
    for (int c = 0; c < 3; c++) { s[c][threadIdx.x] = mn[c]; s[3 + c][threadIdx.x] = mx[c]; }
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int c = 0; c < 3; c++) {
                s[c][threadIdx.x] = fmin(s[c][threadIdx.x], s[c][threadIdx.x + w]);
                s[3 + c][threadIdx.x] = fmax(s[3 + c][threadIdx.x], s[3 + c][threadIdx.x + w]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[((size_t)g * BOX_PARTS + blockIdx.x) * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void box_final_kernel(const double* __restrict__ part, int* __restrict__ nfrag, double* __restrict__ out,
                                                        const int* __restrict__ counters)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 4) nfrag[2 + t] = counters[t];            // the build's counters ride along with the fragment count
    const int g = t / 6, c = t % 6;
    if (g >= *nfrag || g >= NBE_DIST_MAX_FRAGMENTS) return;
    double v = c < 3 ? INFINITY : -INFINITY;
    for (int b = 0; b < BOX_PARTS; b++) {
        const double x = part[((size_t)g * BOX_PARTS + b) * 6 + c];
        v = c < 3 ? fmin(v, x) : fmax(v, x);
    }
    out[g * 6 + c] = v;
}

// Depth of the smallest cell that holds both keys (63-bit keys, 3 bits per level).
static int common_level(uint64_t a, uint64_t b)
{
    if (a == b) return 21;
    const int d = __builtin_clzll(a ^ b) - 1;
    const int l = d / 3;
    return l > 21 ? 21 : l;
}

// The rank's key range is a union of cells that may lie far apart (a range that crosses the middle of the bound holds
// the end of one octant and the start of the next), so one box around all its bodies can span the bound.  The fragments,
// in key order, are merged into at most NBE_DIST_MAX_BOXES runs: always the two neighbours that share the deepest cell.
static void group_fragments(const std::vector<nbe_fragment_t>& fr, std::vector<int>* groups)
{
    struct Run { uint64_t kmin, kmax; int first, last; };
    std::vector<Run> runs;
    for (size_t i = 0; i < fr.size(); i++) {
        const nbe_fragment_t& f = fr[i];
        if (f.level >= 21) { runs.push_back(Run{f.key, f.key, (int)i, (int)i}); continue; }
        const uint64_t size = 1ull << (3 * (21 - f.level));
        const uint64_t cmin = f.key & ~(size - 1ull);
        runs.push_back(Run{cmin, cmin + (size - 1ull), (int)i, (int)i});
    }
    while (runs.size() > (size_t)NBE_DIST_MAX_BOXES) {
        size_t best = 0;
        int best_level = -1;
        for (size_t i = 0; i + 1 < runs.size(); i++) {
            const int l = common_level(runs[i].kmin, runs[i + 1].kmax);
            if (l > best_level) { best_level = l; best = i; }
        }
        runs[best].kmax = runs[best + 1].kmax;
        runs[best].last = runs[best + 1].last;
        runs.erase(runs.begin() + (long)best + 1);
    }
    groups->clear();
    for (const Run& r : runs) { groups->push_back(r.first); groups->push_back(r.last); }      // ranges of fragments, inclusive
}

// Boxes of the first `rows` fragments of the device list, into d->h_aabb (queued; the caller waits).
static int fragment_boxes(nbe_engine* e, int rows)
{
    DistState* d = e->dist;
    double* part = d->d_aabb + (size_t)NBE_DIST_MAX_FRAGMENTS * 6;
    if (e->cfg.precision == NBE_PREC_F64)
        hipLaunchKernelGGL((box_partial_kernel<double4>), dim3(BOX_PARTS, rows), dim3(256), 0, e->stream, (const double4*)key_positions(e),
                           d->d_frag_bodies, d->d_nfrag, part);
    else
        hipLaunchKernelGGL((box_partial_kernel<float4>), dim3(BOX_PARTS, rows), dim3(256), 0, e->stream, (const float4*)key_positions(e),
                           d->d_frag_bodies, d->d_nfrag, part);
    hipLaunchKernelGGL(box_final_kernel, dim3(cdiv((size_t)rows * 6, 256)), dim3(256), 0, e->stream, part, d->d_nfrag, d->d_aabb, e->tw.counters);
    NBE_HIP(hipGetLastError());
    NBE_HIP(hipMemcpyAsync(d->h_aabb, d->d_aabb, (size_t)rows * 6 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    return NBE_OK;
}

extern "C" int nbe_dist_build(nbe_engine* e, nbe_fragment_t* frags, int max, int* count, double* boxes, int* nboxes, uint64_t* n_live)
{
    NBE_TRY(check_dist(e, "nbe_dist_build"));
    DistState* d = e->dist;
    if (!frags || !count || !boxes || !nboxes) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    if (!d->have_total) { nbe_set_error("nbe_dist_build before nbe_dist_set_total_mass"); return NBE_E_INVALID; }
    e->keys_sorted = false;
    // Only the slots that can hold bodies take part: the engine's body count is the number of those (its buffers keep
    // their capacity), so the sort, the build and the walk do not pay for the free slots behind them.
    e->n = d->live_hi;
    // the box kernel's grid has a row per fragment; their number is known only after the read-back, so the grid is sized
    // from the last build's (fragments come and go a few at a time) and the rare build that outgrows it runs it again
    const int box_rows = std::min(NBE_DIST_MAX_FRAGMENTS, 2 * d->frag_hint + 64);
    NBE_HIP(hipMemsetAsync(d->d_nfrag, 0, 8 * sizeof(int), e->stream));
    if (e->n == 0) {
        // no body here: an empty tree (the count of bodies absorbed since the upload, further up in the counters, stays)
        NBE_HIP(hipMemsetAsync(e->tw.counters, 0, 4 * sizeof(int), e->stream));
        NBE_HIP(hipMemsetAsync(e->tw.counters + COUNTER_SORTED_ALIVE, 0, sizeof(int), e->stream));
    } else {
        NBE_TRY(tree_sort_and_build(e));              // (also leaves the geometry records of the nodes: key, depth, parent depth)
        hipLaunchKernelGGL(fragments_kernel, dim3(cdiv(2 * e->n, 256)), dim3(256), 0, e->stream, e->cw.cnodes, e->tw.counters,
                           (const Q4*)e->tw.prefixq, d->split[d->rank], d->split[d->rank + 1], d->d_frags, d->d_nfrag, d->d_frag_bodies);
        NBE_TRY(fragment_boxes(e, box_rows));
    }
    NBE_HIP(hipGetLastError());
    NBE_HIP(hipMemcpyAsync(d->h_small, d->d_nfrag, 8 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    // (as many fragments as the box kernel's grid has rows; the rare build with more fetches the rest below)
    NBE_HIP(hipMemcpyAsync(d->h_frags, d->d_frags, (size_t)box_rows * sizeof(nbe_fragment_t), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    const int nf = d->h_small[0];
    const int na = d->h_small[2];
    if (nf > NBE_DIST_MAX_FRAGMENTS || nf > max) { nbe_set_error("rank %d has %d fragments (limit %d)", d->rank, nf, std::min(max, NBE_DIST_MAX_FRAGMENTS)); return NBE_E_UNSUPPORTED; }
    d->frag_hint = nf;
    if (nf > box_rows) {
        NBE_TRY(fragment_boxes(e, nf));
        NBE_HIP(hipMemcpyAsync(d->h_frags, d->d_frags, (size_t)nf * sizeof(nbe_fragment_t), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
    }
    d->my_frags.assign(d->h_frags, d->h_frags + nf);
    std::sort(d->my_frags.begin(), d->my_frags.end(), [](const nbe_fragment_t& a, const nbe_fragment_t& b) { return a.key < b.key; });
    if (nf) memcpy(frags, d->my_frags.data(), (size_t)nf * sizeof(nbe_fragment_t));
    *count = nf;
    // the boxes of the rank's bodies: one per run of fragments, the union of the fragments' boxes
    std::vector<int> groups;
    group_fragments(d->my_frags, &groups);
    const int nb = (int)(groups.size() / 2);
    for (int g = 0; g < nb; g++) {
        double* bx = boxes + 6 * g;
        for (int c = 0; c < 3; c++) { bx[c] = INFINITY; bx[3 + c] = -INFINITY; }
        for (int i = groups[2 * (size_t)g]; i <= groups[2 * (size_t)g + 1]; i++) {
            const double* fb = d->h_aabb + 6 * (size_t)d->my_frags[(size_t)i].reserved;
            for (int c = 0; c < 3; c++) { bx[c] = fmin(bx[c], fb[c]); bx[3 + c] = fmax(bx[3 + c], fb[3 + c]); }
        }
    }
    *nboxes = nb;
    if (n_live) *n_live = (uint64_t)na;
    d->live_hi = (size_t)na;          // the sort has packed the live bodies into the first slots
    d->n_free = 0;
    d->built = true; d->classified = false; d->exported = false; d->assembled = false;
    e->prepared = true;
    e->stats.n_alive = (uint64_t)na;
    e->stats.tree_nodes = (uint64_t)d->h_small[3];
    e->stats.tree_cells = (uint64_t)d->h_small[4];
    e->host_nalive = (size_t)na;
    return NBE_OK;
}

// ---- export: fragments pruned for another rank -------------------------------------------------------------------------
// Can any point of the box [mn, mx] pass the opening test of this cell?  The test a body applies (tree_walk.hip) is
// r^2 < edge^2/theta^2 on fp32 records (both sides carry eps^2 with softening) and edge / r > theta on fp64 ones, with r
// measured from the cell's centre of mass.  The distance from that centre to the box bounds r from below; `slack` (absolute,
// in metres) and `rel` cover the rounding of the body's own arithmetic, so a cell judged closed here is closed for every
// body in the box whatever its roundings.
__device__ __forceinline__ double box_distance2(double x, double y, double z, const double* __restrict__ b)
{
    const double dx = fmax(fmax(b[0] - x, x - b[3]), 0.0), dy = fmax(fmax(b[1] - y, y - b[4]), 0.0), dz = fmax(fmax(b[2] - z, z - b[5]), 0.0);
    return dx * dx + dy * dy + dz * dz;
}
// Square of the distance from the centre of mass within which a body may open the cell (0: never, a body's record).
__device__ __forceinline__ double open_reach2(const NodeF& nd, double theta, double slack, double rel)
{
    if (!(nd.w2 > 0.0f)) return 0.0;
    // opens iff r^2 < w2 (eps^2 on both sides with softening; left out on the left here: errs towards "may open")
    const double t = sqrt((double)nd.w2 * (1.0 + rel) / (1.0 - rel)) + slack;
    return t * t;
}
__device__ __forceinline__ double open_reach2(const NodeD& nd, double theta, double slack, double rel)
{
    if (!(nd.w > 0.0)) return 0.0;
    if (!(theta > 0.0)) return INFINITY;
    const double t = nd.w * sqrt((1.0 + rel) / (1.0 - rel)) / theta + slack;      // opens iff w / r > theta
    return t * t;
}
// (boxes: the rank's nboxes boxes, and at index NBE_DIST_MAX_BOXES the box around all of them: most cells are far from a
// rank and one test against that box settles them)
template <typename NodeT>
__device__ __forceinline__ bool may_open(const NodeT& nd, double reach2, const double* __restrict__ boxes, int nboxes)
{
    if (nboxes == 0 || !(box_distance2((double)nd.x, (double)nd.y, (double)nd.z, boxes + 6 * NBE_DIST_MAX_BOXES) < reach2)) return false;
    for (int b = 0; b < nboxes; b++)
        if (box_distance2((double)nd.x, (double)nd.y, (double)nd.z, boxes + 6 * b) < reach2) return true;
    return false;
}

// The destinations are handled EXPORT_BATCH at a time, by the same launches (L = 2n + 2 >= nodes + 1 below).
struct DestList { int q[EXPORT_BATCH]; int n; };

// Cells inside the rank's key range that a destination's boxes cannot open hide everything below them: +1 on the node
// after, -1 on the node after their subtree; the running sum is then the number of such cells above a node (at most 21).
// Four destinations share a 32-bit word, eight bits each: sums and differences are taken modulo 2^32, every true running
// sum has all its fields in 0..21, so the scan of the packed words is exact whatever borrows its partial sums see.
// diff[h * L + p]: word h (destinations 4h .. 4h+3) of node p.
template <typename NodeT>
__global__ __launch_bounds__(256) void export_mark_kernel(const NodeT* __restrict__ nodes, const CNode* __restrict__ cn,
                                                          const int* __restrict__ counters, uint64_t lo, uint64_t hi,
                                                          const double* __restrict__ boxes, const int* __restrict__ nboxes,
                                                          DestList dests, size_t L, double theta, double eps2, double slack,
                                                          double rel, uint32_t* __restrict__ diff)
{
    const int total = counters[1];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const CNode c = cn[p];
    if (c.body >= 0 || !cell_inside(c.key, c.level, lo, hi)) return;
    const NodeT nd = nodes[p];
    const double reach2 = open_reach2(nd, theta, slack, rel);
    uint32_t add[2] = {0u, 0u};
    for (int j = 0; j < dests.n; j++) {
        const int q = dests.q[j];
        if (!may_open(nd, reach2, boxes + 6 * (size_t)(NBE_DIST_MAX_BOXES + 1) * q, nboxes[q])) add[j >> 2] += 1u << (8 * (j & 3));
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
        if (add[h]) {
            atomicAdd(&diff[h * L + p + 1], add[h]);
            atomicAdd(&diff[h * L + c.skip], 0u - add[h]);
        }
}

// cover = exclusive scan of diff (both words' stretches sum to zero, so one scan over 2L serves both): field j of
// cover[.. + p + 1] is the number of closed cells above node p for destination j.  A node goes to destination j iff it
// belongs to a fragment and that number is 0.  Per node: inc = the destinations it goes to (bit j), rank8 = for every
// destination its rank among such nodes of its 256-node block (one byte each); per block and destination: the count.
__global__ __launch_bounds__(256) void export_count_kernel(const CNode* __restrict__ cn, const int* __restrict__ counters, uint64_t lo,
                                                           uint64_t hi, const uint32_t* __restrict__ cover, size_t L, int ndest,
                                                           uint8_t* __restrict__ inc, unsigned long long* __restrict__ rank8,
                                                           uint32_t* __restrict__ blockcnt, int nblk)
{
    const int total = counters[1];
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned mask = 0;
    if (p < (size_t)total) {
        const CNode c = cn[p];
        if (c.body >= 0 || cell_inside(c.key, c.level, lo, hi)) {
            const uint32_t c0 = cover[p + 1], c1 = cover[L + p + 1];
            for (int j = 0; j < ndest; j++)
                if ((((j < 4 ? c0 : c1) >> (8 * (j & 3))) & 0xffu) == 0u) mask |= 1u << j;
        }
    }
    __shared__ uint32_t wcnt[EXPORT_BATCH][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long ranks = 0;
#pragma unroll
    for (int j = 0; j < EXPORT_BATCH; j++) {
        const unsigned long long b = __builtin_amdgcn_ballot_w64((mask >> j) & 1u);
        ranks |= (unsigned long long)__popcll(b & ((1ull << lane) - 1ull)) << (8 * j);
        if (lane == 0) wcnt[j][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EXPORT_BATCH; j++) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wcnt[j][w];
        ranks += (unsigned long long)before << (8 * j);          // a rank is at most 255: no carry between the bytes
    }
    if (p < L) { inc[p] = (uint8_t)mask; rank8[p] = ranks; }
    if (threadIdx.x < EXPORT_BATCH) {
        const int j = threadIdx.x;
        blockcnt[(size_t)j * nblk + blockIdx.x] = wcnt[j][0] + wcnt[j][1] + wcnt[j][2] + wcnt[j][3];
        if (blockIdx.x == 0 && j == 0) blockcnt[(size_t)EXPORT_BATCH * nblk] = 0;      // the scan's last element: the batch's total
    }
}

// blockoff = exclusive scan of blockcnt over all destinations of the batch: the place of node x in destination j's
// export, counted from the batch's base, is blockoff[j][x / 256] + byte j of rank8[x].
__device__ __forceinline__ uint32_t export_place(const uint32_t* __restrict__ blockoff, const unsigned long long* __restrict__ rank8,
                                                 int nblk, int j, size_t x)
{
    return blockoff[(size_t)j * nblk + (x >> 8)] + (uint32_t)((rank8[x] >> (8 * j)) & 0xffull);
}

// Leaves the destinations' first nodes in base[q], the sizes of the fragments in there, and the next batch's base.
__global__ __launch_bounds__(256) void export_sizes_kernel(const uint32_t* __restrict__ blockoff, const unsigned long long* __restrict__ rank8,
                                                           int nblk, const nbe_fragment_t* __restrict__ frags, int nfrag, DestList dests,
                                                           unsigned long long* __restrict__ base, int world,
                                                           const unsigned long long* __restrict__ batch_base,
                                                           unsigned long long* __restrict__ next_base, unsigned long long cap,
                                                           int* __restrict__ sizes, int* __restrict__ overflow)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    const int q = dests.q[j];
    if (f == 0) {
        base[q] = *batch_base + blockoff[(size_t)j * nblk];
        if (j == dests.n - 1) {
            const unsigned long long end = *batch_base + blockoff[(size_t)EXPORT_BATCH * nblk];
            *next_base = end;
            base[world] = end;
            if (end > cap) *overflow = 1;
        }
    }
    if (f < nfrag)
        sizes[(size_t)q * nfrag + f] = (int)(export_place(blockoff, rank8, nblk, j, (size_t)frags[f].start + frags[f].nodes) -
                                                              export_place(blockoff, rank8, nblk, j, (size_t)frags[f].start));
}

template <typename NodeT>
__global__ __launch_bounds__(256) void export_write_kernel(const NodeT* __restrict__ nodes, const int* __restrict__ counters,
                                                           const uint8_t* __restrict__ inc, const unsigned long long* __restrict__ rank8,
                                                           const uint32_t* __restrict__ blockoff, int nblk, int ndest,
                                                           const unsigned long long* __restrict__ batch_base,
                                                           const unsigned long long* __restrict__ next_base, unsigned long long cap,
                                                           NodeT* __restrict__ out)
{
    const int total = counters[1];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const unsigned mask = inc[p];
    if (!mask || *next_base > cap) return;            // (overflow: reported by the host, nothing written)
    NodeT nd = nodes[p];
    const int skip = nd.skip;
    const unsigned long long mine = rank8[p], behind = rank8[skip];
    for (int j = 0; j < ndest; j++) {
        if (!((mask >> j) & 1u)) continue;
        const uint32_t first = blockoff[(size_t)j * nblk];                      // the destination's first node
        const uint32_t at = blockoff[(size_t)j * nblk + ((size_t)p >> 8)] + (uint32_t)((mine >> (8 * j)) & 0xffull);
        const uint32_t to = blockoff[(size_t)j * nblk + ((size_t)skip >> 8)] + (uint32_t)((behind >> (8 * j)) & 0xffull);
        nd.skip = (int)(to - first);                  // relative to the first node of this destination's export
        out[*batch_base + at] = nd;
    }
}

template <typename NodeT>
static int export_t(nbe_engine* e, int32_t* export_sizes, uint64_t* export_offsets)
{
    DistState* d = e->dist;
    const int W = d->world;
    const int nf = (int)d->my_frags.size();
    const size_t n = e->n;
    const size_t L = 2 * n + 2;
    const int nblk = (int)cdiv(L, 256);
    const uint64_t lo = d->split[d->rank], hi = d->split[d->rank + 1];
    const double eps2 = e->cfg.eps * e->cfg.eps;
    // rounding of a body's own distance: its coordinates and the node's are within the bound (|x| <= |centre| + width),
    // each difference is off by at most an ulp of that, in the precision of the walk
    double reach = e->cfg.bound_width;
    for (int c = 0; c < 3; c++) reach = fmax(reach, fabs(e->cfg.bound_center[c]) + e->cfg.bound_width);
    const double ulp = e->force64 ? 2.3e-16 : 1.2e-7;
    const double slack = 4.0 * ulp * reach, rel = 16.0 * ulp;
    // work arrays, carved from the four buffers of (8 L + 1) words each
    uint32_t* diff = d->x_diff;                                    // [2][L]
    uint32_t* cover = d->x_cover;                                  // [2][L]
    unsigned long long* rank8 = (unsigned long long*)d->x_index;   // [L]
    uint8_t* inc = (uint8_t*)d->x_flag;                            // [L]
    uint32_t* blockcnt = d->x_flag + (L + 3) / 4 + 1;              // [8][nblk] + 1
    uint32_t* blockoff = blockcnt + (size_t)EXPORT_BATCH * nblk + 2;
    unsigned long long* batch_base = d->d_export_base + (W + 1);      // two device scalars used in turn
    // (bases, overflow flag and sizes: one block, cleared at once; the sorted fragments came in with the boxes)
    const size_t out_bytes = (size_t)(W + 3) * sizeof(unsigned long long) + 2 * sizeof(int) + (size_t)W * (size_t)std::max(nf, 1) * sizeof(int);
    NBE_HIP(hipMemsetAsync(d->d_xout, 0, out_bytes, e->stream));
    std::vector<int> others;
    for (int q = 0; q < W; q++)
        if (q != d->rank) others.push_back(q);
    int batch = 0;
    for (size_t at = 0; nf > 0 && at < others.size(); at += EXPORT_BATCH, batch++) {
        DestList dl;
        dl.n = (int)std::min<size_t>(EXPORT_BATCH, others.size() - at);
        for (int j = 0; j < EXPORT_BATCH; j++) dl.q[j] = j < dl.n ? others[at + (size_t)j] : 0;
        const unsigned long long* cur_base = batch_base + (batch & 1);
        unsigned long long* nxt_base = batch_base + ((batch + 1) & 1);
        NBE_HIP(hipMemsetAsync(diff, 0, 2 * L * 4, e->stream));
        hipLaunchKernelGGL((export_mark_kernel<NodeT>), dim3(cdiv(2 * n, 256)), dim3(256), 0, e->stream, (const NodeT*)e->tw.nodes,
                           e->cw.cnodes, e->tw.counters, lo, hi, d->d_boxes, d->d_box_counts, dl, L, e->cfg.theta, eps2, slack, rel, diff);
        NBE_TRY(device_scan_u32(e, diff, cover, 2 * L));
        hipLaunchKernelGGL(export_count_kernel, dim3(nblk), dim3(256), 0, e->stream, e->cw.cnodes, e->tw.counters, lo, hi, cover, L, dl.n,
                           inc, rank8, blockcnt, nblk);
        NBE_TRY(device_scan_u32(e, blockcnt, blockoff, (size_t)EXPORT_BATCH * nblk + 1));
        hipLaunchKernelGGL(export_sizes_kernel, dim3(cdiv(nf, 256), dl.n), dim3(256), 0, e->stream, blockoff, rank8, nblk, d->d_frags_sorted, nf, dl,
                           d->d_export_base, W, cur_base, nxt_base, (unsigned long long)d->export_cap, d->d_export_sizes, d->d_export_overflow);
        hipLaunchKernelGGL((export_write_kernel<NodeT>), dim3(cdiv(2 * n, 256)), dim3(256), 0, e->stream, (const NodeT*)e->tw.nodes,
                           e->tw.counters, inc, rank8, blockoff, nblk, dl.n, cur_base, nxt_base, (unsigned long long)d->export_cap,
                           (NodeT*)d->export_nodes);
        NBE_HIP(hipGetLastError());
    }
    // read back: offsets, overflow flag, sizes (one copy; the kernel packs the sizes [destination][fragment] without gaps)
    NBE_HIP(hipMemcpyAsync(d->h_xout, d->d_xout, out_bytes, hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    const unsigned long long* h_base = (const unsigned long long*)d->h_xout;
    const int* h_over = (const int*)(h_base + (W + 3));
    const int* h_sizes = h_over + 2;
    if (h_over[0]) { nbe_set_error("rank %d: the export buffer (%zu nodes) is too small", d->rank, d->export_cap); return NBE_E_NOMEM; }
    // destinations were written in rank order; a rank that gets nothing (this one) starts where the next one starts
    export_offsets[W] = (uint64_t)h_base[W];
    for (int q = W - 1; q >= 0; q--) export_offsets[q] = (q == d->rank || nf == 0) ? export_offsets[q + 1] : (uint64_t)h_base[q];
    for (int q = 0; q < W; q++)
        for (int f = 0; f < nf; f++) export_sizes[(size_t)q * nf + f] = q == d->rank ? 0 : h_sizes[(size_t)q * nf + f];
    return NBE_OK;
}

extern "C" int nbe_dist_export(nbe_engine* e, const nbe_fragment_t* all_frags, const int* frag_counts, const double* boxes,
                               const int* box_counts, int32_t* export_sizes, uint64_t* export_offsets)
{
    NBE_TRY(check_dist(e, "nbe_dist_export"));
    DistState* d = e->dist;
    if (!frag_counts || !boxes || !box_counts || !export_sizes || !export_offsets) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    for (int r = 0; r < d->world; r++)
        if (box_counts[r] < 0 || box_counts[r] > NBE_DIST_MAX_BOXES) { nbe_set_error("bad box count of rank %d", r); return NBE_E_INVALID; }
    if (!d->built) { nbe_set_error("nbe_dist_export before nbe_dist_build"); return NBE_E_INVALID; }
    size_t total = 0;
    for (int r = 0; r < d->world; r++) {
        if (frag_counts[r] < 0 || frag_counts[r] > NBE_DIST_MAX_FRAGMENTS) { nbe_set_error("bad fragment count of rank %d", r); return NBE_E_INVALID; }
        total += (size_t)frag_counts[r];
    }
    if (frag_counts[d->rank] != (int)d->my_frags.size()) { nbe_set_error("frag_counts[%d] is not what nbe_dist_build returned", d->rank); return NBE_E_INVALID; }
    if (total > 0 && !all_frags) { nbe_set_error("all_frags is NULL"); return NBE_E_INVALID; }
    d->all_frags.assign(all_frags, all_frags + total);
    d->frag_counts.assign(frag_counts, frag_counts + d->world);
    // boxes, box counts and this rank's fragments (key order) go to the device in one copy from pinned staging
    // (the previous export's copy was waited for before that call returned)
    const size_t boxes_bytes = (size_t)d->world * (NBE_DIST_MAX_BOXES + 1) * 6 * sizeof(double);
    const size_t counts_bytes = ((size_t)d->world * sizeof(int) + 7) / 8 * 8;
    double* hb = (double*)d->h_xin;
    memset(hb, 0, boxes_bytes);
    for (int r = 0; r < d->world; r++) {
        const double* src = boxes + (size_t)r * NBE_DIST_MAX_BOXES * 6;
        double* dst = hb + (size_t)r * (NBE_DIST_MAX_BOXES + 1) * 6;
        double* all = dst + (size_t)NBE_DIST_MAX_BOXES * 6;
        for (int c = 0; c < 3; c++) { all[c] = INFINITY; all[3 + c] = -INFINITY; }
        for (int b = 0; b < box_counts[r]; b++)
            for (int c = 0; c < 3; c++) {
                dst[6 * b + c] = src[6 * b + c];
                dst[6 * b + 3 + c] = src[6 * b + 3 + c];
                all[c] = fmin(all[c], src[6 * b + c]);
                all[3 + c] = fmax(all[3 + c], src[6 * b + 3 + c]);
            }
    }
    memcpy((char*)d->h_xin + boxes_bytes, box_counts, (size_t)d->world * sizeof(int));
    const size_t nf = d->my_frags.size();
    if (nf) memcpy((char*)d->h_xin + boxes_bytes + counts_bytes, d->my_frags.data(), nf * sizeof(nbe_fragment_t));
    NBE_HIP(hipMemcpyAsync(d->d_xin, d->h_xin, boxes_bytes + counts_bytes + nf * sizeof(nbe_fragment_t), hipMemcpyHostToDevice, e->stream));
    int rc = e->force64 ? export_t<NodeD>(e, export_sizes, export_offsets) : export_t<NodeF>(e, export_sizes, export_offsets);
    if (rc == NBE_OK) d->exported = true;
    return rc;
}

// ---- the cells above the fragments, and the layout of the assembled array ---------------------------------------------------
struct FragRef { uint64_t key; int rank, idx, size; const uint64_t* sum; };

struct Layout {
    std::vector<Segment> segs;
    std::vector<TopCell> tops;
    int total = 0;
};

// Lays out fragments [i, j), which share a cell, in pre-order from index `at`; returns the number of nodes and leaves the
// exact sums of what it laid out in `sum` (128-bit integers: the order of addition does not matter).
static Q4 q4_from_words(const uint64_t* w) { return Q4{{w[0], w[1]}, {w[2], w[3]}, {w[4], w[5]}, {w[6], w[7]}}; }

static int layout_range(const std::vector<FragRef>& fr, int i, int j, int at, const std::vector<Segment>& seg_of, Layout* out, Q4* sum)
{
    if (j - i == 1) {
        Segment s = seg_of[(size_t)i];
        s.dst = at;
        s.rebase += at;
        out->segs.push_back(s);
        *sum = q4_from_words(fr[(size_t)i].sum);
        return s.count;
    }
    const int L = common_level(fr[(size_t)i].key, fr[(size_t)j - 1].key);
    const size_t me = out->tops.size();
    out->tops.push_back(TopCell{at, 0, L, 0, q4_zero()});
    int n = 1;
    Q4 acc = q4_zero();
    int a = i;
    while (a < j) {
        const unsigned digit = L >= 21 ? 0u : (unsigned)(fr[(size_t)a].key >> (3 * (20 - L))) & 7u;
        int b = a + 1;
        while (b < j && L < 21 && ((unsigned)(fr[(size_t)b].key >> (3 * (20 - L))) & 7u) == digit) b++;
        Q4 cs;
        n += layout_range(fr, a, b, at + n, seg_of, out, &cs);
        acc = q4_add(acc, cs);
        a = b;
    }
    TopCell& t = out->tops[me];
    t.skip = at + n;
    t.sum = acc;
    *sum = acc;
    return n;
}

extern "C" int nbe_dist_layout(const nbe_fragment_t* all_frags, int nfrags, const int32_t* sizes, nbe_dist_top_cell_t* tops, int max_tops,
                               int* ntops, int32_t* frag_dst, int64_t* total)
{
    if (nfrags < 0 || (nfrags > 0 && (!all_frags || !sizes)) || !ntops || !total) { nbe_set_error("nbe_dist_layout: bad argument"); return NBE_E_INVALID; }
    std::vector<FragRef> fr;
    std::vector<Segment> seg_of;
    for (int g = 0; g < nfrags; g++) {
        if (g > 0 && !(all_frags[g - 1].key < all_frags[g].key)) { nbe_set_error("fragments are not in key order (at %d)", g); return NBE_E_INVALID; }
        if (sizes[g] < 1) { nbe_set_error("fragment %d has no nodes", g); return NBE_E_INVALID; }
        fr.push_back(FragRef{all_frags[g].key, 0, g, sizes[g], all_frags[g].sum});
        seg_of.push_back(Segment{0, sizes[g], g, 0, 0});          // src carries the fragment's number
    }
    Layout lay;
    Q4 sum;
    if (!fr.empty()) lay.total = layout_range(fr, 0, nfrags, 0, seg_of, &lay, &sum);
    *total = lay.total;
    *ntops = (int)lay.tops.size();
    if ((int)lay.tops.size() > max_tops && tops) { nbe_set_error("%zu spanning cells, room for %d", lay.tops.size(), max_tops); return NBE_E_INVALID; }
    for (size_t t = 0; tops && t < lay.tops.size(); t++) {
        tops[t].dst = lay.tops[t].dst; tops[t].skip = lay.tops[t].skip; tops[t].level = lay.tops[t].level; tops[t].reserved = 0;
        const Q4& q = lay.tops[t].sum;
        const uint64_t w[8] = {q.m.lo, q.m.hi, q.x.lo, q.x.hi, q.y.lo, q.y.hi, q.z.lo, q.z.hi};
        for (int k = 0; k < 8; k++) tops[t].sum[k] = w[k];
    }
    if (frag_dst)
        for (const Segment& s : lay.segs) frag_dst[s.src] = s.dst;
    return NBE_OK;
}

template <typename NodeT>
__global__ __launch_bounds__(256) void top_cells_kernel(const TopCell* __restrict__ tops, int ntops, double bound_width, double theta,
                                                        double eps2, int total, int shift_m, int shift_x, NodeT* __restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        // the record behind the last node, as the build leaves one: massless, never opens, skips to itself
        NodeT snt;
        snt.x = 0; snt.y = 0; snt.z = 0; snt.m = 0;
        set_edge<NodeT>(snt, 0.0, theta, eps2);
        snt.skip = total;
        out[total] = snt;
    }
    if (t >= ntops) return;
    const TopCell c = tops[t];
    NodeT nd;
    exact_cell(c.sum, shift_m, shift_x, nd);          // the expressions of the build's own cells
    set_edge<NodeT>(nd, bound_width * (1.0 / (double)(1u << c.level)), theta, eps2);
    nd.skip = c.skip;
    out[c.dst] = nd;
}

// One workgroup per 256 nodes of a segment would need a work list; instead every thread finds its segment by bisection
// over the segments' first indices (a few thousand at most, cached).
template <typename NodeT>
__global__ __launch_bounds__(256) void assemble_kernel(const Segment* __restrict__ segs, int nsegs, const int* __restrict__ seg_first,
                                                       const NodeT* __restrict__ own, const NodeT* __restrict__ imported,
                                                       long long work, NodeT* __restrict__ out)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= work) return;
    int lo = 0, hi = nsegs - 1;                       // last segment whose first work item is <= w
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((long long)seg_first[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const Segment s = segs[lo];
    const int k = (int)(w - seg_first[lo]);
    NodeT nd = s.remote ? imported[s.src + k] : own[s.src + k];
    nd.skip += s.rebase;
    out[s.dst + k] = nd;
}

extern "C" int nbe_dist_assemble(nbe_engine* e, const int32_t* import_sizes, const uint64_t* import_offsets)
{
    NBE_TRY(check_dist(e, "nbe_dist_assemble"));
    DistState* d = e->dist;
    if (!d->exported) { nbe_set_error("nbe_dist_assemble before nbe_dist_export"); return NBE_E_INVALID; }
    if (!import_sizes || !import_offsets) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    const int W = d->world;
    std::vector<FragRef> fr;
    std::vector<Segment> seg_of;
    size_t g = 0;
    for (int r = 0; r < W; r++) {
        int within = 0;                               // nodes of rank r's export to this rank before the fragment
        for (int f = 0; f < d->frag_counts[(size_t)r]; f++, g++) {
            const nbe_fragment_t& fg = d->all_frags[g];
            Segment s;
            s.dst = 0;
            if (r == d->rank) {
                // own fragment: the subtree as the build left it; its skips are indices of the own node array
                s.count = fg.nodes; s.src = fg.start; s.rebase = -fg.start; s.remote = 0;
            } else {
                // imported: skips are relative to the first node of r's export to this rank
                s.count = import_sizes[g]; s.src = (int)import_offsets[r] + within; s.rebase = -within; s.remote = 1;
                if (s.count < 1 || (size_t)s.src + (size_t)s.count > d->import_cap) { nbe_set_error("bad import size of fragment %d of rank %d", f, r); return NBE_E_INVALID; }
                within += s.count;
            }
            if (!fr.empty() && !(fr.back().key < fg.key)) { nbe_set_error("fragments are not in key order (rank %d, fragment %d)", r, f); return NBE_E_INVALID; }
            fr.push_back(FragRef{fg.key, r, f, s.count, fg.sum});
            seg_of.push_back(s);
        }
    }
    Layout lay;
    Q4 sum;
    if (!fr.empty()) lay.total = layout_range(fr, 0, (int)fr.size(), 0, seg_of, &lay, &sum);
    if ((size_t)lay.total > d->let_cap) { nbe_set_error("rank %d: the assembled tree (%d nodes) exceeds the buffer (%zu)", d->rank, lay.total, d->let_cap); return NBE_E_NOMEM; }
    if (lay.segs.size() > d->seg_cap || lay.tops.size() > d->top_cap) { nbe_set_error("too many fragments"); return NBE_E_UNSUPPORTED; }
    // work items: the nodes of all segments, numbered segment after segment.  The two counters the walk reads, the
    // segments, their first work items and the spanning cells are packed into one pinned block and go up in one copy
    // (the previous step's copy was consumed long ago: every step waits for the device several times).
    const size_t nseg = lay.segs.size(), ntop = lay.tops.size();
    const size_t off_first = 64, off_segs = off_first + (nseg * sizeof(int) + 15) / 16 * 16;
    const size_t off_tops = off_segs + (nseg * sizeof(Segment) + 15) / 16 * 16;
    const size_t in_bytes = off_tops + ntop * sizeof(TopCell);
    if (in_bytes > d->ain_bytes) { nbe_set_error("internal: layout block too small"); return NBE_E_INVALID; }
    char* hin = (char*)d->h_ain;
    int* h_first = (int*)(hin + off_first);
    Segment* h_segs = (Segment*)(hin + off_segs);
    TopCell* h_tops = (TopCell*)(hin + off_tops);
    long long work = 0;
    for (size_t k = 0; k < nseg; k++) { h_first[k] = (int)work; h_segs[k] = lay.segs[k]; work += lay.segs[k].count; }
    for (size_t t = 0; t < ntop; t++) h_tops[t] = lay.tops[t];
    ((int*)hin)[0] = (int)d->live_hi;
    ((int*)hin)[1] = lay.total;
    NBE_HIP(hipMemcpyAsync(d->d_ain, d->h_ain, in_bytes, hipMemcpyHostToDevice, e->stream));
    const int* d_seg_first = (const int*)((char*)d->d_ain + off_first);
    const Segment* d_segments = (const Segment*)((char*)d->d_ain + off_segs);
    const TopCell* d_top = (const TopCell*)((char*)d->d_ain + off_tops);
    const double eps2 = e->cfg.eps * e->cfg.eps;
    const int ntops = (int)lay.tops.size();
    if (e->force64) {
        hipLaunchKernelGGL((top_cells_kernel<NodeD>), dim3(cdiv(ntops + 1, 256)), dim3(256), 0, e->stream, d_top, ntops, e->cfg.bound_width,
                           e->cfg.theta, eps2, lay.total, e->tw.q_shift_m, e->tw.q_shift_x, (NodeD*)d->let_nodes);
        if (work > 0)
            hipLaunchKernelGGL((assemble_kernel<NodeD>), dim3(cdiv((size_t)work, 256)), dim3(256), 0, e->stream, d_segments,
                               (int)lay.segs.size(), d_seg_first, (const NodeD*)e->tw.nodes, (const NodeD*)d->import_nodes, work,
                               (NodeD*)d->let_nodes);
    } else {
        hipLaunchKernelGGL((top_cells_kernel<NodeF>), dim3(cdiv(ntops + 1, 256)), dim3(256), 0, e->stream, d_top, ntops, e->cfg.bound_width,
                           e->cfg.theta, eps2, lay.total, e->tw.q_shift_m, e->tw.q_shift_x, (NodeF*)d->let_nodes);
        if (work > 0)
            hipLaunchKernelGGL((assemble_kernel<NodeF>), dim3(cdiv((size_t)work, 256)), dim3(256), 0, e->stream, d_segments,
                               (int)lay.segs.size(), d_seg_first, (const NodeF*)e->tw.nodes, (const NodeF*)d->import_nodes, work,
                               (NodeF*)d->let_nodes);
    }
    NBE_HIP(hipGetLastError());
    d->assembled = true;
    e->stats.tree_nodes = (uint64_t)lay.total;
    return NBE_OK;
}

// ---- force pass + update ---------------------------------------------------------------------------------------------------
extern "C" int nbe_dist_advance(nbe_engine* e)
{
    NBE_TRY(check_dist(e, "nbe_dist_advance"));
    DistState* d = e->dist;
    if (!d->assembled) { nbe_set_error("nbe_dist_advance before nbe_dist_assemble"); return NBE_E_INVALID; }
    NBE_TRY(frame_stream_guard_write(e, e->cur ^ 1));
    // the walk reads the node array and the two counters through the engine's tree workspace
    void* nodes = e->tw.nodes;
    int* counters = e->tw.counters;
    e->tw.nodes = d->let_nodes;
    e->tw.counters = d->let_counters;
    const bool collide = e->cfg.collisions != 0;
    const int rc = tree_forces(e, e->cur, 0, e->n, !collide, false, e->cur ^ 1);
    e->tw.nodes = nodes;
    e->tw.counters = counters;
    NBE_TRY(rc);
    // Collisions (tree.go:245-281, main.go:151-167): the accelerations wait in the engine's buffer, and search, merges and
    // update run over the rank's own tree: whatever a body can touch lies in a cell of at most four radii, which the aligned
    // splitters keep on one rank (nbe_dist_set_splitters), and pairs that share a body share that cell.
    if (collide) NBE_TRY(collide_and_update(e));
    e->cur ^= 1;
    e->prepared = false;
    d->assembled = false; d->built = false; d->exported = false;
    e->stats.steps++;
    return NBE_OK;
}

extern "C" int nbe_dist_energy(nbe_engine* e, double* ke, double* pe, double mom[3])
{
    NBE_TRY(check_dist(e, "nbe_dist_energy"));
    DistState* d = e->dist;
    if (!d->assembled) { nbe_set_error("nbe_dist_energy needs the assembled tree (between nbe_dist_assemble and nbe_dist_advance)"); return NBE_E_INVALID; }
    // instrumented walk: leaves every body's potential in the fourth component of its acceleration record
    void* nodes = e->tw.nodes;
    int* counters = e->tw.counters;
    e->tw.nodes = d->let_nodes;
    e->tw.counters = d->let_counters;
    const int rc = tree_forces(e, e->cur, 0, e->n, false, true, e->cur ^ 1);
    e->tw.nodes = nodes;
    e->tw.counters = counters;
    NBE_TRY(rc);
    double h[5] = {0, 0, 0, 0, 0};
    NBE_TRY(energy_local_sums(e, (int)d->live_hi, true, h));
    if (ke) *ke = h[0];
    if (mom) { mom[0] = h[1]; mom[1] = h[2]; mom[2] = h[3]; }
    if (pe) *pe = -0.5 * e->cfg.G * h[4];
    return NBE_OK;
}

// This rank's share of nbe_accel: the accelerations of its bodies from the assembled tree (the instrumented pass that
// nbe_dist_energy runs too; it also counts the rank's force evaluations), with their run-wide indices.
static int accel_t(nbe_engine* e, uint64_t* index, double* acc, size_t* count)
{
    DistState* d = e->dist;
    *count = 0;
    const size_t k = d->live_hi;
    if (k == 0) return NBE_OK;
    void* nodes = e->tw.nodes;
    int* counters = e->tw.counters;
    e->tw.nodes = d->let_nodes;
    e->tw.counters = d->let_counters;
    const int rc = tree_forces(e, e->cur, 0, e->n, false, true, e->cur ^ 1);
    e->tw.nodes = nodes;
    e->tw.counters = counters;
    NBE_TRY(rc);
    std::vector<uint32_t> ho(k);
    std::vector<double> ha;
    std::vector<float> fa;
    NBE_HIP(hipMemcpyAsync(ho.data(), e->orig[e->orig_cur], k * 4, hipMemcpyDeviceToHost, e->stream));
    if (e->state64) { ha.resize(4 * k); NBE_HIP(hipMemcpyAsync(ha.data(), e->acc, k * 32, hipMemcpyDeviceToHost, e->stream)); }
    else { fa.resize(4 * k); NBE_HIP(hipMemcpyAsync(fa.data(), e->acc, k * 16, hipMemcpyDeviceToHost, e->stream)); }
    NBE_HIP(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < k; i++) {
        for (int c = 0; c < 3; c++) acc[3 * i + c] = e->state64 ? ha[4 * i + c] : (double)fa[4 * i + c];
        if (index) index[i] = ho[i];
    }
    *count = k;
    return NBE_OK;
}

extern "C" int nbe_dist_accel(nbe_engine* e, uint64_t* index, double* acc_xyz, size_t* count)
{
    NBE_TRY(check_dist(e, "nbe_dist_accel"));
    if (!acc_xyz || !count) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    if (!e->dist->assembled) { nbe_set_error("nbe_dist_accel needs the assembled tree (between nbe_dist_assemble and nbe_dist_advance)"); return NBE_E_INVALID; }
    try { return accel_t(e, index, acc_xyz, count); }
    catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
}

static int download_t(nbe_engine* e, uint64_t* index, double* pos, double* vel, double* mass, double* radius, size_t* count)
{
    DistState* d = e->dist;
    if (!count) { nbe_set_error("count is NULL"); return NBE_E_INVALID; }
    const size_t k = d->live_hi;
    *count = 0;
    if (k == 0) return NBE_OK;
    std::vector<uint32_t> ho(k);
    std::vector<double> hp, hv, hr;
    std::vector<float> fp, fv;
    if (radius) {
        hr.resize(k);
        NBE_HIP(hipMemcpyAsync(hr.data(), e->radius[e->cur], k * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    }
    NBE_HIP(hipMemcpyAsync(ho.data(), e->orig[e->orig_cur], k * 4, hipMemcpyDeviceToHost, e->stream));
    if (e->state64) {
        hp.resize(4 * k); hv.resize(4 * k);
        NBE_HIP(hipMemcpyAsync(hp.data(), e->posm[e->cur], k * 32, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipMemcpyAsync(hv.data(), e->vel[e->cur], k * 32, hipMemcpyDeviceToHost, e->stream));
    } else {
        fp.resize(4 * k); fv.resize(4 * k);
        NBE_HIP(hipMemcpyAsync(fp.data(), e->posm[e->cur], k * 16, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipMemcpyAsync(fv.data(), e->vel[e->cur], k * 16, hipMemcpyDeviceToHost, e->stream));
    }
    NBE_HIP(hipStreamSynchronize(e->stream));
    // a slot vacated by the last classify holds a massless body outside the bound: not a body of the run
    double lo[3], W;
    bound_lo(e, lo, &W);
    size_t out = 0;
    for (size_t i = 0; i < k; i++) {
        const double x = e->state64 ? hp[4 * i] : (double)fp[4 * i], m = e->state64 ? hp[4 * i + 3] : (double)fp[4 * i + 3];
        if (m == 0.0 && x == (e->state64 ? lo[0] + 2.0 * W : (double)(float)(lo[0] + 2.0 * W))) continue;
        for (int c = 0; c < 3; c++) {
            if (pos) pos[3 * out + c] = e->state64 ? hp[4 * i + c] : (double)fp[4 * i + c];
            if (vel) vel[3 * out + c] = e->state64 ? hv[4 * i + c] : (double)fv[4 * i + c];
        }
        if (mass) mass[out] = m;
        if (radius) radius[out] = hr[i];
        if (index) index[out] = ho[i];
        out++;
    }
    *count = out;
    return NBE_OK;
}

extern "C" int nbe_dist_download(nbe_engine* e, uint64_t* index, double* pos, double* vel, double* mass, size_t* count)
{
    NBE_TRY(check_dist(e, "nbe_dist_download"));
    try { return download_t(e, index, pos, vel, mass, nullptr, count); }
    catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
}

extern "C" int nbe_dist_get_radius(nbe_engine* e, uint64_t* index, double* radius, size_t* count)
{
    NBE_TRY(check_dist(e, "nbe_dist_get_radius"));
    if (!e->cfg.collisions) { nbe_set_error("nbe_dist_get_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
    if (!radius) { nbe_set_error("radius is NULL"); return NBE_E_INVALID; }
    try { return download_t(e, index, nullptr, nullptr, nullptr, radius, count); }
    catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
}
