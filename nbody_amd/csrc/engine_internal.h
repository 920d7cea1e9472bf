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
    double pad1[2];
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
    uint32_t* lvlmask = nullptr;    // per sorted leaf: bit L set iff a branching cell of depth L starts here
    uint32_t* cellcnt = nullptr;    // exclusive scan of popcount(lvlmask), n+1 entries
    void* nodes = nullptr;          // NodeF[2n] or NodeD[2n]
    uint8_t* node_level = nullptr;  // per pre-order node: depth of a branching cell, 255 for a leaf
    double* node_sum = nullptr;     // per pre-order node: fp64 (mass, m*x, m*y, m*z)  (fp64 node records only)
    double* prefix = nullptr;       // fp32 node records: fp64 prefix sums of (m, m*x, m*y, m*z) over sorted bodies
    double* prefix_tiles = nullptr;
    int* counters = nullptr;        // device scalars: [0] n_alive, [1] total nodes, [2] branching cells
    unsigned long long* d_interactions = nullptr;
};

// Ring of pinned host buffers that carries frames from the GPU to a delivery thread.
struct FrameSlot {
    float* xyzm = nullptr;        // pinned, n x 4 floats
    uint32_t* orig = nullptr;     // pinned, n x u32 (device order -> upload index)
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
    nbe_frame_cb cb = nullptr;
    void* user = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t state_ready = nullptr;
    hipEvent_t copy_done[2] = {nullptr, nullptr};   // per state buffer: the last copy that read it has finished
    bool copy_pending[2] = {false, false};
    float4* staging = nullptr;    // device: fp32 copy of an fp64 state
    std::vector<FrameSlot> slots;
    std::deque<int> queue;        // slots waiting for delivery, in frame order
    std::mutex mu;
    std::condition_variable cv_work, cv_free;
    std::thread worker;
    bool stop = false;
    std::vector<uint64_t> ids_scratch;
};

struct nbe_engine {
    nbe_config cfg{};
    size_t n = 0;
    bool state64 = false;           // fp64 state (MIXED and F64)
    bool force64 = false;           // fp64 pair arithmetic (F64)
    size_t elem = 16;               // bytes per 4-real element of the state
    void* posm[2] = {nullptr, nullptr};
    void* vel[2] = {nullptr, nullptr};
    float4* mirror[2] = {nullptr, nullptr};   // MIXED: fp32 copy of posm used by the force kernels
    uint32_t* orig[2] = {nullptr, nullptr};   // device order -> upload index; flips only when the state is sorted
    int orig_cur = 0;
    int cur = 0;
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
    bool prepared = false;          // tree of the current state is valid
    bool advance_pending = false;   // *_next hold an advanced shard waiting for nbe_shard_commit
    TreeWork tw;
    nbe_stats_t stats{};
    size_t host_nalive = 0;
    double* d_energy = nullptr;     // device scratch for nbe_energy
    FrameStream fs;
};

// ---- functions implemented across the translation units ----------------------------------------------
// direct.hip
int direct_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, int dst_buf);
// tree_build.hip
int tree_workspace_reserve(nbe_engine* e);
int tree_sort_and_build(nbe_engine* e);         // sorts cur -> cur^1, flips e->cur, builds the tree
// tree_walk.hip
int tree_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, bool count, int dst_buf);
// energy.hip
int energy_compute(nbe_engine* e, double* ke, double* pe, double mom[3]);
// frames.hip
int frame_stream_before_step(nbe_engine* e);    // called at the top of every step
int frame_stream_guard_write(nbe_engine* e, int buf);
int frame_stream_stop(nbe_engine* e);

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
