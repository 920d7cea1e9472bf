// Collisions in tree mode: the search of tree.go:245-281 for every live body, the queue of main.go:153-159 in
// the reference's order, the serial merge of main.go:160-166 with combine (physics.go:153-162), and the update
// (physics.go:101-115) of the survivors.
//
// The search reproduces the reference's pruning rules as they are written, not as they were probably meant:
//   * intersectSphere (tree.go:82-86) compares the offset from the cell centre with the FULL cell edge plus
//     the radius and passes if ANY axis is within reach;
//   * at an internal cell, if the octant the body falls into exists and its doubled box holds the sphere
//     (containsSphere, tree.go:75-79, also with the full edge), ONLY that child is searched (tree.go:265-270),
//     so a body usually meets just the residents of the small cell around itself and misses a neighbour across
//     a cell boundary;
//   * a pair is queued when dist <= radius sum (tree.go:258-261), once per searching body that finds it.
// The engine's tree has no one-child chains.  That changes nothing here: a chain cell hands the search to its
// only child whichever branch it takes, and its own intersectSphere test is implied by the test of the cell
// below it (the child's centre is a quarter edge away and its edge is half: |d| <= w/2 + r on an axis of the
// child gives |d| <= 3w/4 + r on that axis of the parent).
#include "engine_internal.h"
#include "kernels_common.h"
#include "tree_keys.h"


struct Cell { double cx, cy, cz, w; };

#define RANK_QUEUE_CAP 262144    // bodies of one rank of the distributed tree that may queue pairs in one step

// Centre and edge of the depth-`level` cell that contains `key`: exact for a power-of-two bound, and equal to
// what octantBound (tree.go:101-110) reaches by repeated quarter-edge translations.
__device__ __forceinline__ Cell cell_of(uint64_t key, int level, double lox, double loy, double loz, double W)
{
    const uint64_t k = key >> (3 * (21 - level));
    Cell c;
    c.w = ldexp(W, -level);
    c.cx = lox + ((double)compact3(k) + 0.5) * c.w;
    c.cy = loy + ((double)compact3(k >> 1) + 0.5) * c.w;
    c.cz = loz + ((double)compact3(k >> 2) + 0.5) * c.w;
    return c;
}

// One record per pre-order node (same indices as the walk's node array): thread t < n handles sorted leaf t,
// thread n + i handles binary node i, exactly as the build's node emission does.
__global__ __launch_bounds__(256) void emit_cnodes_kernel(int n, const int* __restrict__ counters, const uint64_t* __restrict__ keys,
                                                          const int* __restrict__ bt_first, const int* __restrict__ bt_last,
                                                          const int* __restrict__ bt_parent, const uint8_t* __restrict__ bt_level,
                                                          const uint32_t* __restrict__ lvlmask, const uint32_t* __restrict__ cellcnt,
                                                          CNode* __restrict__ cn, uint8_t* __restrict__ dead,
                                                          uint32_t* __restrict__ count)
{
    const int na = counters[0];
    const int t = blockIdx.x * 256 + threadIdx.x;
    // A new tree: nobody is absorbed yet and no pair is counted (what two fills of their own did before: at a thousand
    // bodies every launch is 4 us of a 100 us step).
    if (t < n) dead[t] = 0;
    if (t <= n) count[t] = 0;
    CNode c;
    c.pmask = 0; c.pad0 = 0; c.parent = -1;
    if (t < n) {
        if (t >= na) return;
        const int pos = t + (int)cellcnt[t + 1];
        const uint64_t ki = keys[t];
        const int dl = key_delta(keys, na, t, ki, t - 1), dr = key_delta(keys, na, t, ki, t + 1);
        const int d = dl > dr ? dl : dr;
        c.key = ki; c.body = t; c.skip = pos + 1;
        if (d < 0) { c.level = 0; c.plevel = 255; }          // a single body: the root is its leaf
        else {
            const int pl = delta_level(d);
            c.plevel = (uint8_t)pl;
            c.level = (uint8_t)(pl + 1);
        }
        cn[pos] = c;
        return;
    }
    const int i = t - n;
    if (i >= na - 1) return;
    const int par = bt_parent[i];
    const int lvl = bt_level[i];
    if (!(par < 0 || bt_level[par] != lvl)) return;
    const int a = bt_first[i], b = bt_last[i];
    const int pos = a + (int)cellcnt[a] + __popc(lvlmask[a] & ((1u << lvl) - 1u));
    c.key = keys[a]; c.body = -1; c.skip = (b + 1) + (int)cellcnt[b + 1];
    c.level = (uint8_t)lvl;
    c.plevel = par < 0 ? (uint8_t)255 : bt_level[par];
    cn[pos] = c;
}

// Every branching cell tells its children where it is and which of its octants are occupied.
__global__ __launch_bounds__(256) void cnode_parents_kernel(const int* __restrict__ counters, CNode* __restrict__ cn)
{
    const int total = counters[1];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    if (cn[p].body >= 0) return;
    const int level = cn[p].level, end = cn[p].skip;
    unsigned mask = 0;
    if (level >= 21) mask = 0xff;                             // bucket of equal keys: no octants below
    else
        for (int c = p + 1; c < end; c = cn[c].skip) mask |= 1u << ((unsigned)(cn[c].key >> (3 * (20 - level))) & 7u);
    for (int c = p + 1; c < end; c = cn[c].skip) { cn[c].pmask = (uint8_t)mask; cn[c].parent = p; }
}

// Is the search of a body at (px,py,pz) with radius R confined to one child of the cell `pc`?  tree.go:265-270:
// the octant the body falls into must exist (`mask`) and its doubled box must hold the sphere.  *oct_out is that octant.
__device__ __forceinline__ bool confined(const Cell& pc, unsigned mask, double px, double py, double pz, double R, int* oct_out)
{
    const int oct = (px - pc.cx >= 0.0 ? 1 : 0) | (py - pc.cy >= 0.0 ? 2 : 0) | (pz - pc.cz >= 0.0 ? 4 : 0);
    *oct_out = oct;
    if (!((mask >> oct) & 1u)) return false;
    const double q = 0.25 * pc.w, cw = 0.5 * pc.w - R;          // the child's bound: half the edge
    const double ccx = pc.cx + ((oct & 1) ? q : -q), ccy = pc.cy + ((oct & 2) ? q : -q), ccz = pc.cz + ((oct & 4) ? q : -q);
    return fabs(px - ccx) <= cw && fabs(py - ccy) <= cw && fabs(pz - ccz) <= cw;
}

// On a rank of the key-range-distributed tree the upload (run-wide) indices of the bodies do not number a local array,
// so the counts and offsets are kept per sorted body there, and the bodies that queue anything are listed (unordered,
// through one atomic each) for collide_rank_kernel to put into the reference's queue order.  list == nullptr: one engine.
struct RankQueue {
    uint4* list;            // (run-wide index, sorted index, pairs queued, 0) of every body that queues pairs
    uint32_t* meta;         // [0] entries of the list, [1] pairs in all, [2] bit 0: a radius above rcap, bit 1: list overflow
    int cap;
    double rcap;            // searches of bodies with larger radii could reach beyond the rank's own cells
};

// tree.go:245-281 for sorted body t.  FILL = false counts the pairs the body queues (into count[upload index]);
// FILL = true writes them at offset[upload index] in the order the recursion meets them (children 0..7).
//
// From the root down, the recursion stays confined to the body's own octant for as long as containsSphere holds
// for the child it is in, which it does at every cell with edge >= 4r (the body is inside the child, at most half
// a child edge from its centre, and the test allows a full child edge minus r).  Cells on the way down always
// pass intersectSphere, because the body is inside them.  So the search proper starts at the SHALLOWEST cell above
// the body where the confinement fails: found by climbing from the body's leaf through the cells with edge < 4r.
// If there is none, the recursion ends at the body's own leaf (tree.go:246-248) with nothing queued.  Below that
// cell every node is judged as the recursion would judge it, by the decision its parent takes for it.
template <typename R4, bool FILL>
__global__ __launch_bounds__(256) void collide_search_kernel(const CNode* __restrict__ cn, const int* __restrict__ counters,
                                                             const uint64_t* __restrict__ keys, const R4* __restrict__ posm,
                                                             const double* __restrict__ radius, const uint32_t* __restrict__ orig,
                                                             const uint32_t* __restrict__ cellcnt,
                                                             double lox, double loy, double loz, double W,
                                                             uint32_t* __restrict__ count, const uint32_t* __restrict__ offset,
                                                             int2* __restrict__ pairs, RankQueue rq)
{
    const int na = counters[0];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= na) return;
    const double R = radius[t];
    if (!FILL && rq.list && R > rq.rcap) atomicOr(&rq.meta[2], 1u);
    if (na < 2) return;
    const double reach = 4.000001 * R;                 // cells at least this wide confine the search for certain
    {
        // the deepest cell above the body is the one it shares with its nearer neighbour in key order
        const uint64_t ki = keys[t];
        const int dl = key_delta(keys, na, t, ki, t - 1), dr = key_delta(keys, na, t, ki, t + 1);
        const int lp = delta_level(dl > dr ? dl : dr);
        if (lp < 21 && ldexp(W, -lp) > reach) return;
    }
    const R4 p4 = posm[t];
    const double px = (double)p4.x, py = (double)p4.y, pz = (double)p4.z;
    int top = -1;                                      // shallowest cell above the body that does not confine it
    for (int c = cn[t + (int)cellcnt[t + 1]].parent; c >= 0;) {
        const CNode nd = cn[c];
        if (nd.level >= 21) top = c;                   // bucket of equal keys: all of its leaves are searched
        else {
            const Cell pc = cell_of(nd.key, nd.level, lox, loy, loz, W);
            if (pc.w > reach) break;
            // octants present under this cell: the mask its children carry
            int oct;
            if (!confined(pc, cn[c + 1].pmask, px, py, pz, R, &oct)) top = c;
        }
        c = nd.parent;
    }
    if (top < 0) return;
    const uint32_t slot = rq.list ? (uint32_t)t : orig[t];       // where this body's count and offset live
    const uint32_t base = FILL ? offset[slot] : 0u;
    uint32_t found = 0;
    const int end = cn[top].skip;
    int cur = top + 1;
    while (cur < end) {
        const CNode nd = cn[cur];
        bool visit = true;
        if (nd.plevel < 21) {
            // the decision the cell above took for this child (tree.go:265-279)
            int oct;
            if (confined(cell_of(nd.key, nd.plevel, lox, loy, loz, W), nd.pmask, px, py, pz, R, &oct))
                visit = (int)((unsigned)(nd.key >> (3 * (20 - nd.plevel))) & 7u) == oct;
        }
        if (!visit || nd.body == t) { cur = nd.skip; continue; }              // tree.go:246-248
        const Cell c = cell_of(nd.key, nd.level < 21 ? nd.level : 21, lox, loy, loz, W);
        const double wr = c.w + R;
        if (!(fabs(px - c.cx) <= wr || fabs(py - c.cy) <= wr || fabs(pz - c.cz) <= wr)) { cur = nd.skip; continue; }
        if (nd.body >= 0) {
            const R4 q4 = posm[nd.body];
            const double dx = px - (double)q4.x, dy = py - (double)q4.y, dz = pz - (double)q4.z;
            // physics.go:123-128, without fused multiply-adds
            const double r = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
            if (r <= radius[nd.body] + R) {
                if (FILL) pairs[base + found] = make_int2(t, nd.body);
                found++;
            }
            cur = nd.skip;
        } else {
            cur++;
        }
    }
    if (!FILL) {
        count[slot] = found;
        if (rq.list && found) {
            const uint32_t k = atomicAdd(&rq.meta[0], 1u);
            if (k < (uint32_t)rq.cap) rq.list[k] = make_uint4(orig[t], (uint32_t)t, found, 0u);
            else atomicOr(&rq.meta[2], 2u);
        }
    }
}

// The reference queues the pairs of the searching bodies in slice order (main.go:153-159), i.e. by upload index; pairs
// that share a body always sit on one rank, so a rank that works its own pairs off in that order does what the one
// queue does.  Entry i goes to the place given by the number of entries with a smaller run-wide index (they are few).
__global__ __launch_bounds__(256) void collide_rank_kernel(const uint4* __restrict__ list, const uint32_t* __restrict__ meta, int cap,
                                                           uint4* __restrict__ sorted)
{
    // (all pairs of entries: the indices pass through LDS 256 at a time and every thread reads them as broadcasts; the
    // full list of 262,144 entries, far beyond any step seen, takes about two milliseconds)
    __shared__ uint32_t idx[256];
    const int m = (int)min(meta[0], (uint32_t)cap);
    if ((int)blockIdx.x * 256 >= m) return;            // (the whole workgroup alike)
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint4 me = i < m ? list[i] : make_uint4(0u, 0u, 0u, 0u);
    int before = 0;
    for (int base = 0; base < m; base += 256) {
        __syncthreads();
        if (base + (int)threadIdx.x < m) idx[threadIdx.x] = list[base + threadIdx.x].x;
        __syncthreads();
        const int lim = min(256, m - base);
        for (int j = 0; j < lim; j++) before += idx[j] < me.x ? 1 : 0;
    }
    if (i < m) sorted[before] = me;
}

// One wave: running sum of the entries' pair counts in queue order; every listed body learns where its pairs go.
__global__ __launch_bounds__(64) void collide_offsets_kernel(const uint4* __restrict__ sorted, uint32_t* __restrict__ meta, int cap,
                                                             uint32_t* __restrict__ offset)
{
    const int m = (int)min(meta[0], (uint32_t)cap);
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (int base = 0; base < m; base += 64) {
        const int k = base + lane;
        const uint4 en = k < m ? sorted[k] : make_uint4(0, 0, 0, 0);
        uint32_t inc = en.z;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (k < m) offset[en.y] = carry + inc - en.z;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) meta[1] = carry;
}

// combine (physics.go:153-162) for one queued pair, unless an earlier merge removed one of the two (main.go:161-163).
// Accelerations stand in for the reference's force accumulators: f = a*m before the merge, a = (fa + fb) / M
// after it.  a.Mass is the sum already when the velocities are mixed, as the reference writes it.
template <typename A4>
__device__ __forceinline__ int merge_pair(int a, int b, A4* __restrict__ posm, A4* __restrict__ vel, A4* __restrict__ acc,
                                          double* __restrict__ radius, uint8_t* __restrict__ dead, float4* __restrict__ mirror)
{
    if (dead[a] || dead[b]) return 0;
    A4 pa = posm[a], va = vel[a], fa = acc[a];
    const A4 pb = posm[b], vb = vel[b], fb = acc[b];
    const auto ma = pa.w, mb = pb.w;
    const auto M = ma + mb;
    const double ra = radius[a], rb = radius[b];
    radius[a] = cbrt(ra * ra * ra + rb * rb * rb);
    va.x = (M * va.x + mb * vb.x) / (M + mb);
    va.y = (M * va.y + mb * vb.y) / (M + mb);
    va.z = (M * va.z + mb * vb.z) / (M + mb);
    fa.x = (fa.x * ma + fb.x * mb) / M;
    fa.y = (fa.y * ma + fb.y * mb) / M;
    fa.z = (fa.z * ma + fb.z * mb) / M;
    pa.w = M;
    posm[a] = pa; vel[a] = va; acc[a] = fa;
    if (mirror) mirror[a].w = (float)M;
    dead[b] = 1;
    return 1;
}

// main.go:160-166: the queue is worked off in order, and the outcome depends on that order wherever pairs share
// a body.  One wave takes 64 queue entries at a time: entries that share no body with an earlier entry of the
// same 64 cannot be affected by them and are merged side by side; the others follow one at a time in queue order.
// Entries of earlier 64s are finished (and their stores fenced) before the next 64 are read.
template <typename A4>
__global__ __launch_bounds__(64) void collide_merge_kernel(const int2* __restrict__ pairs, const uint32_t* __restrict__ total_ptr,
                                                           A4* __restrict__ posm, A4* __restrict__ vel, A4* __restrict__ acc,
                                                           double* __restrict__ radius, uint8_t* __restrict__ dead,
                                                           float4* __restrict__ mirror, int* __restrict__ counters)
{
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    const uint32_t total = *total_ptr;
    int merged = 0;
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
        const bool have = q0 + lane < total;
        int a = -1 - lane, b = -65 - lane;            // lanes without an entry match nobody
        if (have) { const int2 pr = pairs[q0 + lane]; a = pr.x; b = pr.y; }
        bool shares = false;
        for (int j = 0; j < 63; j++) {
            const int aj = __shfl(a, j, 64), bj = __shfl(b, j, 64);
            if (j < lane && (aj == a || aj == b || bj == a || bj == b)) shares = true;
        }
        if (have && !shares) merged += merge_pair(a, b, posm, vel, acc, radius, dead, mirror);
        __threadfence_block();
        unsigned long long rest = __ballot(have && shares);
        while (rest) {
            const int k = __ffsll((long long)rest) - 1;
            if (lane == k) merged += merge_pair(a, b, posm, vel, acc, radius, dead, mirror);
            __threadfence_block();
            rest &= rest - 1;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) merged += __shfl_down(merged, d, 64);
    if (lane == 0) {
        counters[3] += merged;      // absorbed since the last sort
        counters[6] += merged;      // absorbed since upload
    }
}

// physics.go:101-115 for the survivors, from the acceleration buffer.  An absorbed body is parked outside the
// bound with no mass, so the next build drops it the way it drops a body that flew out (tree.go:290-292).
template <typename A4, typename A>
__global__ __launch_bounds__(256) void collide_update_kernel(const int* __restrict__ counters, int n, const A4* __restrict__ pos_in,
                                                             const A4* __restrict__ vel_in, const A4* __restrict__ acc,
                                                             const uint8_t* __restrict__ dead, const double* __restrict__ rad_in,
                                                             A dt, A parkx, A parky, A parkz, A4* __restrict__ pos_out,
                                                             A4* __restrict__ vel_out, float4* __restrict__ mirror_out,
                                                             double* __restrict__ rad_out)
{
    const int na = counters[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    A4 p = pos_in[i], v = vel_in[i];
    if (i < na) {
        if (dead[i]) {
            p.x = parkx; p.y = parky; p.z = parkz; p.w = 0;
            v.x = 0; v.y = 0; v.z = 0;
        } else {
            const A4 a = acc[i];
            v.x += a.x * dt; v.y += a.y * dt; v.z += a.z * dt;
            p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
        }
    }
    pos_out[i] = p;
    vel_out[i] = v;
    if (mirror_out) mirror_out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
    rad_out[i] = rad_in[i];
}

__global__ __launch_bounds__(256) void fill_double_kernel(double* __restrict__ out, size_t n, double v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

// ---- host side ---------------------------------------------------------------------------------------------
static int free_ptr(void** p)
{
    if (*p) { NBE_HIP(hipFree(*p)); *p = nullptr; }
    return NBE_OK;
}

int collide_workspace_free(nbe_engine* e)
{
    CollWork& w = e->cw;
    void** ptrs[] = {(void**)&w.cnodes, (void**)&w.count, (void**)&w.offset, (void**)&w.pairs, (void**)&w.dead, (void**)&w.rank_list,
                     (void**)&w.rank_meta};
    for (void** p : ptrs) NBE_TRY(free_ptr(p));
    if (w.total_host) { NBE_HIP(hipHostFree(w.total_host)); w.total_host = nullptr; }
    w.cap = 0; w.pairs_cap = 0; w.begun = false;
    return NBE_OK;
}

int collide_workspace_reserve(nbe_engine* e)
{
    CollWork& w = e->cw;
    const size_t cap = e->n > 0 ? e->n : 1;
    if (e->dist && e->cfg.collisions && !w.rank_list) {
        NBE_HIP(hipMalloc((void**)&w.rank_list, 2 * (size_t)RANK_QUEUE_CAP * sizeof(uint4)));
        NBE_HIP(hipMalloc((void**)&w.rank_meta, 4 * sizeof(uint32_t)));
    }
    if (w.cap >= cap) return NBE_OK;
    {   // (the list of a distributed rank does not depend on the body count: it survives the reallocation below)
        uint4* keep_list = w.rank_list; uint32_t* keep_meta = w.rank_meta;
        w.rank_list = nullptr; w.rank_meta = nullptr;
        NBE_TRY(collide_workspace_free(e));
        w.rank_list = keep_list; w.rank_meta = keep_meta;
    }
    NBE_HIP(hipMalloc((void**)&w.cnodes, (2 * cap + 2) * sizeof(CNode)));
    NBE_HIP(hipMalloc((void**)&w.count, (cap + 1) * 4));
    NBE_HIP(hipMalloc((void**)&w.offset, (cap + 1) * 4));
    NBE_HIP(hipMalloc((void**)&w.dead, cap));
    NBE_HIP(hipMemsetAsync(w.dead, 0, cap, e->stream));
    w.pairs_cap = 4096;
    NBE_HIP(hipMalloc((void**)&w.pairs, w.pairs_cap * sizeof(int2)));
    NBE_HIP(hipHostMalloc((void**)&w.total_host, 4 * sizeof(uint32_t), hipHostMallocDefault));
    w.cap = cap;
    return NBE_OK;
}

int collide_fill_radius(nbe_engine* e, double* dst, double value)
{
    if (e->n == 0) return NBE_OK;
    hipLaunchKernelGGL(fill_double_kernel, dim3(cdiv(e->n, 256)), dim3(256), 0, e->stream, dst, e->n, value);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// Called at the end of the build (tree of e->cur is final): geometry records, and nobody is absorbed yet.
int collide_emit_nodes(nbe_engine* e)
{
    TreeWork& t = e->tw;
    CollWork& w = e->cw;
    const int n = (int)e->n;
    hipLaunchKernelGGL(emit_cnodes_kernel, dim3(cdiv(2 * (size_t)n, 256)), dim3(256), 0, e->stream, n, t.counters, t.keys[0],
                       t.bt_first, t.bt_last, t.bt_parent, t.bt_level, t.lvlmask, t.cellcnt, w.cnodes, w.dead, w.count);
    hipLaunchKernelGGL(cnode_parents_kernel, dim3(cdiv(2 * (size_t)n, 256)), dim3(256), 0, e->stream, t.counters, w.cnodes);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// A rank of the key-range-distributed tree keeps the list of its searching bodies behind the pair buffer's bookkeeping.
static RankQueue rank_queue(nbe_engine* e)
{
    RankQueue rq;
    rq.list = e->dist ? e->cw.rank_list : nullptr;
    rq.meta = e->cw.rank_meta;
    rq.cap = RANK_QUEUE_CAP;
    rq.rcap = e->cw.radius_cap;
    return rq;
}

// The counting search, the scan of the counts and the copy of their total to the host, all queued and not waited for:
// a caller that drives several engines queues this on each of them before it finishes any.
template <typename R4>
static int collide_begin_t(nbe_engine* e, const void* force_pos)
{
    TreeWork& t = e->tw;
    CollWork& w = e->cw;
    const int n = (int)e->n;
    const int cur = e->cur;
    const double W = e->cfg.bound_width;
    const double lox = e->cfg.bound_center[0] - 0.5 * W, loy = e->cfg.bound_center[1] - 0.5 * W,
                 loz = e->cfg.bound_center[2] - 0.5 * W;
    if (n == 0) {                                     // (a rank of the distributed tree may hold no body at all)
        if (w.total_host)                             // (no workspace yet: nothing was ever uploaded)
            for (int k = 0; k < 4; k++) w.total_host[k] = 0;
        w.begun = true;
        return NBE_OK;
    }
    // (the pair counts were cleared when the tree was made: emit_cnodes_kernel)
    const RankQueue rq = rank_queue(e);
    if (rq.list) NBE_HIP(hipMemsetAsync(rq.meta, 0, 4 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL((collide_search_kernel<R4, false>), dim3(cdiv(n, 256)), dim3(256), 0, e->stream, w.cnodes, t.counters,
                       t.keys[0], (const R4*)force_pos, e->radius[cur], e->orig[e->orig_cur], t.cellcnt, lox, loy, loz, W, w.count,
                       w.offset, w.pairs, rq);
    NBE_HIP(hipGetLastError());
    if (rq.list) {
        hipLaunchKernelGGL(collide_rank_kernel, dim3(cdiv((size_t)rq.cap, 256)), dim3(256), 0, e->stream, rq.list, rq.meta, rq.cap,
                           rq.list + rq.cap);
        hipLaunchKernelGGL(collide_offsets_kernel, dim3(1), dim3(64), 0, e->stream, rq.list + rq.cap, rq.meta, rq.cap, w.offset);
        NBE_HIP(hipGetLastError());
        NBE_HIP(hipMemcpyAsync(w.total_host, rq.meta, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    } else {
        NBE_TRY(device_scan_u32(e, w.count, w.offset, (size_t)n + 1));
        NBE_HIP(hipMemcpyAsync(w.total_host + 1, w.offset + n, 4, hipMemcpyDeviceToHost, e->stream));
    }
    w.begun = true;
    return NBE_OK;
}

int collide_begin(nbe_engine* e)
{
    if (e->cw.begun) return NBE_OK;
    switch (e->cfg.precision) {
    case NBE_PREC_F32: return collide_begin_t<float4>(e, e->posm[e->cur]);
    case NBE_PREC_MIXED: return collide_begin_t<float4>(e, e->mirror[e->cur]);
    case NBE_PREC_F64: return collide_begin_t<double4>(e, e->posm[e->cur]);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}

template <typename R4, typename A4, typename A>
static int collide_and_update_t(nbe_engine* e, const void* force_pos)
{
    TreeWork& t = e->tw;
    CollWork& w = e->cw;
    const int n = (int)e->n;
    const int cur = e->cur, nxt = cur ^ 1;
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    const double W = e->cfg.bound_width;
    const double lox = e->cfg.bound_center[0] - 0.5 * W, loy = e->cfg.bound_center[1] - 0.5 * W,
                 loz = e->cfg.bound_center[2] - 0.5 * W;
    const dim3 grid(cdiv(n, 256)), block(256);

    NBE_TRY(collide_begin(e));
    w.begun = false;
    if (n == 0) { e->stats.collision_pairs = 0; return NBE_OK; }
    NBE_HIP(hipStreamSynchronize(e->stream));
    const RankQueue rq = rank_queue(e);
    const uint32_t total = w.total_host[1];
    // (a step that is refused below leaves pair counts behind that only a new tree clears: the next one builds it again)
    if (rq.list && (w.total_host[2] & 3u)) e->prepared = false;
    if (rq.list && (w.total_host[2] & 1u)) {
        nbe_set_error("a body's radius exceeds the cap of %g m given for collisions on the distributed tree (its search could reach another rank's bodies)", rq.rcap);
        return NBE_E_UNSUPPORTED;
    }
    if (rq.list && (w.total_host[2] & 2u)) {
        nbe_set_error("more than %d bodies of one rank queue collision pairs in one step", RANK_QUEUE_CAP);
        return NBE_E_UNSUPPORTED;
    }
    e->stats.collision_pairs = total;
    if (total > 0) {
        if ((size_t)total > w.pairs_cap) {
            NBE_TRY(free_ptr((void**)&w.pairs));
            w.pairs_cap = (size_t)total + (size_t)total / 2;
            NBE_HIP(hipMalloc((void**)&w.pairs, w.pairs_cap * sizeof(int2)));
        }
        hipLaunchKernelGGL((collide_search_kernel<R4, true>), grid, block, 0, e->stream, w.cnodes, t.counters, t.keys[0],
                           (const R4*)force_pos, e->radius[cur], e->orig[e->orig_cur], t.cellcnt, lox, loy, loz, W, w.count,
                           w.offset, w.pairs, rq);
        hipLaunchKernelGGL((collide_merge_kernel<A4>), dim3(1), dim3(64), 0, e->stream, w.pairs, rq.list ? rq.meta + 1 : w.offset + n, (A4*)e->posm[cur],
                           (A4*)e->vel[cur], (A4*)e->acc, e->radius[cur], w.dead, mixed ? e->mirror[cur] : (float4*)nullptr,
                           t.counters);
        NBE_HIP(hipGetLastError());
    }
    NBE_TRY(frame_stream_guard_write(e, nxt));
    hipLaunchKernelGGL((collide_update_kernel<A4, A>), grid, block, 0, e->stream, t.counters, n, (const A4*)e->posm[cur],
                       (const A4*)e->vel[cur], (const A4*)e->acc, w.dead, e->radius[cur], (A)e->cfg.dt, (A)(lox + 2.0 * W),
                       (A)(loy + 2.0 * W), (A)(loz + 2.0 * W), (A4*)e->posm[nxt], (A4*)e->vel[nxt],
                       mixed ? e->mirror[nxt] : (float4*)nullptr, e->radius[nxt]);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// The accelerations of the live bodies are in e->acc (a force pass over the tree of e->cur has just run).
int collide_and_update(nbe_engine* e)
{
    switch (e->cfg.precision) {
    case NBE_PREC_F32: return collide_and_update_t<float4, float4, float>(e, e->posm[e->cur]);
    case NBE_PREC_MIXED: return collide_and_update_t<float4, double4, double>(e, e->mirror[e->cur]);
    case NBE_PREC_F64: return collide_and_update_t<double4, double4, double>(e, e->posm[e->cur]);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}
