// Internal declarations shared by the engine's translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "nbody_engine.h"

// ---- error plumbing -------------------------------------------------------------------------------
void nbe_set_error(const char* fmt, ...);

#define NBE_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t err__ = (call);                                                             \
        if (err__ != hipSuccess) {                                                             \
            nbe_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return NBE_E_DEVICE;                                                               \
        }                                                                                      \
    } while (0)

#define NBE_TRY(call)                     \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != NBE_OK) return rc__;  \
    } while (0)

// ---- tree node records read by the walk kernel -----------------------------------------------------
// Nodes (one-body leaves and branching cells) are stored in depth-first pre-order with children in octant
// order 0..7, the order tree.go:221-225 visits them.  `skip` is the index of the first node after this
// node's subtree, so "descend" is index+1 and "accept or pass over" is `skip`.
// A leaf has w2 == 0 (w == 0): it can never satisfy the opening test and is always evaluated.
struct NodeF {
    float x, y, z, m;   // centre of mass and mass
    float w2;           // (cell edge / theta)^2; tree.go:216 uses the full edge, bounds.max().  0 for a leaf
    int skip;
    int pad0, pad1;
};
static_assert(sizeof(NodeF) == 32, "NodeF must be 32 bytes");

struct NodeD {
    double x, y, z, m;
    double w;           // cell edge
    int skip;
    int pad0;
    double w2;          // (edge / theta)^2: the walk's first look at the opening test is r^2 < w2, without a product
    double pad1;
};
static_assert(sizeof(NodeD) == 64, "NodeD must be 64 bytes");

// ---- engine object ---------------------------------------------------------------------------------
struct TreeWork {
    size_t cap = 0;                 // bodies the buffers are sized for
    uint64_t* keys[2] = {nullptr, nullptr};
    uint32_t* sidx[2] = {nullptr, nullptr};
    uint32_t* block_hist = nullptr; // radix sort: [256][nblocks] digit counts
    size_t block_hist_len = 0;
    uint32_t* scan_tmp = nullptr;   // block sums of the device-wide scans
    size_t scan_tmp_len = 0;
    // binary radix tree over the sorted keys (one entry per internal node i in [0, n_alive-2])
    int* bt_first = nullptr;
    int* bt_last = nullptr;
    int* bt_parent = nullptr;
    uint8_t* bt_level = nullptr;
    uint8_t* bt_cell = nullptr;     // per binary node: 1 iff it is an octree branching cell (depth differs from its parent's)
    uint32_t* lvlmask = nullptr;    // per sorted leaf: bit L set iff a branching cell of depth L starts here
    uint32_t* cellcnt = nullptr;    // exclusive scan of popcount(lvlmask), n+1 entries
    void* nodes = nullptr;          // NodeF[2n] or NodeD[2n]
    uint8_t* node_level = nullptr;  // per pre-order node: depth of a branching cell, 255 for a leaf
    double* node_sum = nullptr;     // per pre-order node: fp64 (mass, m*x, m*y, m*z)  (fp64 node records only)
    double* prefix = nullptr;       // fp32 node records: fp64 prefix sums of (m, m*x, m*y, m*z) over sorted bodies
    double* prefix_tiles = nullptr;
    void* prefixq = nullptr;        // NBE_SUMS_EXACT: 128-bit fixed-point prefix sums (64 bytes per body) and their tile totals
    void* prefixq_tiles = nullptr;
    int q_shift_m = 0, q_shift_x = 0;   // their quanta: 2^-q_shift_m (mass) and 2^-q_shift_x (mass x coordinate)
    double total_mass = 0.0;        // what the quanta were chosen from
    int* counters = nullptr;        // device scalars: [0] n_alive, [1] total nodes, [2] branching cells
    unsigned long long* d_interactions = nullptr;
};

// ---- collisions (tree mode) ------------------------------------------------------------------------------
// Geometry of the pre-order nodes for the collision search, one record per node of the walk's node array
// (same index).  The cell of a node is the depth-`level` cell that contains the Morton key `key`; a one-body
// leaf lives in the depth plevel+1 cell (the reference's leaf is the direct child of the branching cell above it).
struct CNode {
    uint64_t key;       // Morton key of the node's first body
    int body;           // sorted index of a leaf's body, -1 for a branching cell
    int skip;           // first pre-order index after the subtree
    uint8_t level;      // depth of the node's cell (leaf: plevel + 1; 22 for a leaf of a depth-21 bucket)
    uint8_t plevel;     // depth of the branching cell above, 255 for the root
    uint8_t pmask;      // which octants of that cell hold bodies (bit o = child o exists, tree.go:266)
    uint8_t pad0;
    int parent;         // pre-order index of that cell, -1 for the root
};
static_assert(sizeof(CNode) == 24, "CNode must be 24 bytes");

struct CollWork {
    size_t cap = 0;                 // bodies the buffers are sized for
    CNode* cnodes = nullptr;        // 2*cap + 2 records
    uint32_t* count = nullptr;      // per upload index: pairs queued by that body
    uint32_t* offset = nullptr;     // exclusive scan of count, cap + 1 entries
    int2* pairs = nullptr;          // queue: (searching body, resident body) as sorted indices
    size_t pairs_cap = 0;
    uint8_t* dead = nullptr;        // per device slot: absorbed since the last sort
    uint32_t* total_host = nullptr; // pinned, 4 words: [1] the number of queued pairs, read back between the counting and the filling
                                    // search; on a rank of the distributed tree also [0] the bodies that queue pairs and [2] error bits
    uint4* rank_list = nullptr;     // distributed tree: the searching bodies of this rank, unordered and then in queue order
    uint32_t* rank_meta = nullptr;  // ... and their bookkeeping (collide.hip, RankQueue)
    double radius_cap = 32.0;       // ... and the largest radius its splitters' alignment allows
    bool begun = false;             // collide_begin has queued the counting search of this step
};

// Ring of pinned host buffers that carries frames from the GPU to a delivery thread.
struct FrameSlot {
    float* xyzm = nullptr;        // pinned, n x 4 floats
    uint32_t* orig = nullptr;     // pinned, n x u32 (device order -> upload index)
    float* radius = nullptr;      // pinned, n floats (collisions on)
    uint8_t* dead = nullptr;      // pinned, n flags (direct mode with collisions on)
    int* n_alive = nullptr;       // pinned scalar
    int* d_n_alive = nullptr;     // device snapshot of the live count, taken on the compute stream
    hipEvent_t done = nullptr;    // copy finished (side stream)
    int frame = 0;
    bool busy = false;
};

struct FrameStream {
    bool active = false;
    int every = 1, first_frame = 0;
    uint64_t step_base = 0;       // stats.steps when streaming began
    uint64_t last_k = 0;          // step (counted from step_base) whose frame went out last
    bool have_last = false;
    nbe_frame_cb cb = nullptr;
    void* user = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t state_ready = nullptr;
    hipEvent_t copy_done[2] = {nullptr, nullptr};   // per state buffer: the last copy that read it has finished
    bool copy_pending[2] = {false, false};
    float4* staging = nullptr;    // device: fp32 copy of an fp64 state
    float* radius_staging = nullptr;   // device: fp32 copy of the radii (collisions on)
    std::vector<FrameSlot> slots;
    std::deque<int> queue;        // slots waiting for delivery, in frame order
    std::mutex mu;
    std::condition_variable cv_work, cv_free;
    std::thread worker;
    bool stop = false;
    std::vector<uint64_t> ids_scratch;
};

// tw.counters slot where the sort's last pass leaves the number of live bodies, until the state has been reordered
// (permute_kernel then publishes it in slot 0)
#define COUNTER_SORTED_ALIVE 8

struct MultiState;
struct DistState;

struct nbe_engine {
    nbe_config cfg{};
    MultiState* multi = nullptr;    // num_devices > 1: this object only coordinates one engine per GPU (multi.hip)
    DistState* dist = nullptr;      // key-range-distributed tree (dist.hip): this engine holds one rank's bodies
    size_t n = 0;
    double equal_mass = 0.0;        // direct sum without collisions: the common mass if every body has the same, else 0
    bool state64 = false;           // fp64 state (MIXED and F64)
    bool force64 = false;           // fp64 pair arithmetic (F64)
    size_t elem = 16;               // bytes per 4-real element of the state
    void* posm[2] = {nullptr, nullptr};
    void* vel[2] = {nullptr, nullptr};
    float4* mirror[2] = {nullptr, nullptr};   // MIXED: fp32 copy of posm used by the force kernels
    uint32_t* orig[2] = {nullptr, nullptr};   // device order -> upload index; flips only when the state is sorted
    double* radius[2] = {nullptr, nullptr};   // collisions on: body radii, indexed like posm (follows `cur`)
    CollWork cw;
    // direct mode with collisions on (direct_collide.hip); radii and absorbed flags are not reordered there, so
    // radius[0] and cw.dead are the only copies
    double4* dc_force = nullptr;    // per body: force accumulator of the replayed sweep
    int* dc_counters = nullptr;     // [0] some pair touches, [1] bodies absorbed since upload
    size_t dc_cap = 0;
    size_t direct_absorbed = 0;     // host copy of dc_counters[1]
    int orig_cur = 0;
    int cur = 0;
    int small_max = 0;              // tree mode: up to this many bodies take the single-workgroup sort and scans
    int walk_bodies_per_lane = 0;   // fp32 tree walk: 1 or 2 forces the kernel (test knob), 0 chooses by the number of targets
    int walk_xcd_shift = -1;        // NBODY_WALK_XCD_CHUNK_LOG2: -1 the engine's choice, 0 contiguous eighths, s chunks of 2^s groups of 256 bodies
    int walk_sparse = -1;           // NBODY_WALK_SPARSE=s: 64 >> s bodies per wavefront of the one-body walk (-1: the engine's choice)
    bool walk_private = false;      // NBODY_WALK_PRIVATE=1: force passes without update take the one-body-one-cursor yardstick kernel (fp32)
    bool timing = false;            // record phase events during the next step
    bool have_counts = false;       // tw.counters hold the result of a build of this upload
    uint64_t* ids = nullptr;        // host copy of body IDs in upload order
    void* acc = nullptr;            // n elements of 4 reals (force pass output)
    void* partial = nullptr;        // direct sum with split source ranges: [jsplit][n] x 4 reals
    size_t partial_len = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t shard_first = 0, shard_count = 0;
    bool shard_set = false;
    bool keys_sorted = false;       // tree mode: keys of the current state are sorted already (nbe_shard_sort ran)
    bool prepared = false;          // tree of the current state is valid
    bool advance_pending = false;   // *_next hold an advanced shard waiting for nbe_shard_commit
    int advance_parts = 0;          // advances since the last commit (nbe_shard_advance_range counts double)
    TreeWork tw;
    nbe_stats_t stats{};
    size_t host_nalive = 0;
    double* d_energy = nullptr;     // device scratch for nbe_energy
    FrameStream fs;
};

// ---- functions implemented across the translation units ----------------------------------------------
// multi.hip: one engine per GPU behind one handle
int multi_create(const nbe_config* cfg, nbe_engine** out);
void multi_destroy(nbe_engine* e);
int multi_upload(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id);
int multi_step(nbe_engine* e, int nsteps);
int multi_sync(nbe_engine* e);
int multi_set_radius(nbe_engine* e, const double* radius);
nbe_engine* multi_first(nbe_engine* e);          // the engine of the first GPU (reads are served from it)
bool multi_is_distributed(const nbe_engine* e);  // cfg.distributed: no GPU holds all bodies, reads are gathered
int multi_dist_download(nbe_engine* e, double* pos, double* vel, double* mass, uint64_t* id, size_t* n_alive_out);
int multi_dist_energy(nbe_engine* e, double* ke, double* pe, double mom[3]);
int multi_dist_get_radius(nbe_engine* e, double* radius, size_t* n_alive_out);
int multi_dist_accel(nbe_engine* e, double* acc_xyz, uint8_t* alive_out);
int multi_dist_stats(nbe_engine* e, nbe_stats_t* out);
int multi_dist_frames_begin(nbe_engine* e, int every, int slots, int first_frame, nbe_frame_cb cb, void* user);
int multi_dist_frames_end(nbe_engine* e);
size_t dist_live_slots(const nbe_engine* e);    // a rank's slots [0, this) may hold bodies; all of them do between a build and the next classify
// dist.hip
void dist_destroy(nbe_engine* e);
// engine.hip
int engine_upload_state(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id);
// direct.hip
int direct_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, int dst_buf);
// tree_build.hip
int tree_workspace_reserve(nbe_engine* e);
int tree_small_limit(void);
int tree_sum_mode(const nbe_engine* e);         // NBE_SUMS_* in effect (fp64 node records: never the fp64 prefix differences)
void tree_set_total_mass(nbe_engine* e, double total);   // quanta of NBE_SUMS_EXACT from the run's total mass
int tree_sort_keys(nbe_engine* e);              // optional first half of tree_sort_and_build (reads positions only)
int tree_sort_and_build(nbe_engine* e);         // sorts cur -> cur^1, flips e->cur, builds the tree
int device_scan_u32(nbe_engine* e, const uint32_t* in, uint32_t* out, size_t len);
// tree_walk.hip
int tree_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, bool count, int dst_buf);
// collide.hip
int collide_workspace_reserve(nbe_engine* e);
int collide_workspace_free(nbe_engine* e);
int collide_fill_radius(nbe_engine* e, double* dst, double value);
int collide_emit_nodes(nbe_engine* e);          // after the build: geometry records for the collision search
int collide_begin(nbe_engine* e);               // first half of collide_and_update: queues the counting search, does not wait
int collide_and_update(nbe_engine* e);          // search + merge + update of the state in e->cur into e->cur^1
// direct_collide.hip
int direct_collide_reserve(nbe_engine* e);
int direct_collide_free(nbe_engine* e);
int direct_collide_step(nbe_engine* e, bool* replayed);
// energy.hip
int energy_compute(nbe_engine* e, double* ke, double* pe, double mom[3]);
int energy_local_sums(nbe_engine* e, int n_live, bool tree_potential, double h[5]);
// frames.hip
int frame_stream_before_step(nbe_engine* e);    // called at the top of every step
int frame_stream_guard_write(nbe_engine* e, int buf);
int frame_stream_stop(nbe_engine* e);

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
