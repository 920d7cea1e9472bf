// GPU Barnes-Hut build: Morton keys -> LDS radix sort -> binary radix tree -> octree cells in depth-first
// pre-order -> bottom-up mass / centre-of-mass reduction.
//
// Replaces maketree/push (tree.go:145-197, 284-295).  The reference inserts bodies one by one into a
// pointer octree whose cells halve a fixed cube (octantBound tree.go:101-110) and picks the child by the
// sign of p - centre per axis (octantBits tree.go:128-132: bit0 = X, bit1 = Y, bit2 = Z, ">= centre" is
// the high half).  For a cube whose edge is a power of two that path is the sequence of 3-bit digits of
// the body's 21-bit-per-axis fixed-point coordinates, so sorting bodies by the interleaved key puts them
// in the order a depth-first walk with children 0..7 meets them, and every cell that holds two or more
// bodies in different children ("branching cell") is a maximal run of keys sharing a 3L-bit prefix.
// Chains of one-child cells, which the pointer tree keeps, are not materialised: they carry the same
// mass and centre of mass as the branching cell below them and a larger edge, so the opening test
// (tree.go:216-217) accepts the chain's top exactly when it accepts that cell, with the same force.
// Deviations: depth is capped at 21 (cell edge = bound/2^21); bodies with identical keys become leaves of
// one depth-21 cell (the reference would recurse without bound).  The centre of mass is the mass-weighted
// mean, not the running update of tree.go:189-193.
#include "engine_internal.h"
#include "tree_keys.h"
#include "tree_sums.h"

#include <cmath>
#include <cstdlib>

// tw.counters[COUNTER_SORTED_ALIVE]: live bodies found by the last key sort; becomes counters[0] when the state is
// reordered (permute_kernel)

__device__ __forceinline__ unsigned xcd_contiguous_block(void);

#define SORT_KPT 8                     // keys per thread
#define SORT_TILE (256 * SORT_KPT)     // keys per workgroup of the smallest kind (sizes the table of digit counts)
#define SCAN_TILE 2048                 // elements per workgroup in the device-wide scan (256 threads x 8)

// ---- device-wide exclusive scan of uint32 -------------------------------------------------------------
template <bool POPC>
__device__ __forceinline__ uint32_t scan_load(const uint32_t* in, size_t i, size_t len)
{
    if (i >= len) return 0;
    uint32_t v = in[i];
    return POPC ? (uint32_t)__popc(v) : v;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of a 256-thread block's per-thread totals; returns this thread's offset, *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// The same for the 256 digit owners (threads 0..255) of a workgroup of T >= 256 threads: the other threads pass 0, join
// the barriers and get nothing of use back.
template <int T>
__device__ __forceinline__ uint32_t block_excl_scan_digits(uint32_t v, uint32_t* total)
{
    if constexpr (T == 256) return block_excl_scan(v, total);
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63 && wave < 4) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <bool POPC>
__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t* __restrict__ in, size_t len,
                                                          uint32_t* __restrict__ block_sums)
{
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += scan_load<POPC>(in, base + k, len);
    uint32_t tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// One workgroup scans the block sums in place (exclusive).
__global__ __launch_bounds__(256) void scan_sums_kernel(uint32_t* __restrict__ sums, int nsums)
{
    uint32_t carry = 0;
    for (int base = 0; base < nsums; base += 256 * 8) {
        uint32_t v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int i = base + threadIdx.x * 8 + k;
            v[k] = i < nsums ? sums[i] : 0;
            s += v[k];
        }
        uint32_t tot;
        uint32_t off = carry + block_excl_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int i = base + threadIdx.x * 8 + k;
            if (i < nsums) sums[i] = off;
            off += v[k];
        }
        carry += tot;
    }
}

template <bool POPC>
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* __restrict__ in, size_t len,
                                                         const uint32_t* __restrict__ block_offsets,
                                                         uint32_t* __restrict__ out)
{
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = scan_load<POPC>(in, base + k, len); s += v[k]; }
    uint32_t tot;
    uint32_t off = block_offsets[blockIdx.x] + block_excl_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < len) out[base + k] = off;
        off += v[k];
    }
}

// Short inputs: one workgroup walks the tiles with a running carry (one launch instead of three).
template <bool POPC>
__global__ __launch_bounds__(256) void scan_single_kernel(const uint32_t* __restrict__ in, size_t len, uint32_t* __restrict__ out)
{
    uint32_t carry = 0;
    for (size_t base = 0; base < len; base += SCAN_TILE) {
        const size_t b = base + (size_t)threadIdx.x * 8;
        uint32_t v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = scan_load<POPC>(in, b + k, len); s += v[k]; }
        uint32_t tot;
        uint32_t off = carry + block_excl_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (b + k < len) out[b + k] = off;
            off += v[k];
        }
        carry += tot;
    }
}

// Radix sort offsets: the per-tile digit counts are stored digit-major, [digit][tile].  One workgroup per digit
// turns its row into exclusive prefixes over the tiles and leaves the digit's total in totals[digit]; the scatter
// kernel adds the digits below (a 256-entry scan it does itself).  One launch instead of the three of a scan
// over the whole table.
__global__ __launch_bounds__(256) void scan_rows_kernel(uint32_t* __restrict__ hist, int nblocks, uint32_t* __restrict__ totals)
{
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += SCAN_TILE) {
        const int b = base + threadIdx.x * 8;
        uint32_t v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = b + k < nblocks ? row[b + k] : 0u; s += v[k]; }
        uint32_t tot;
        uint32_t off = carry + block_excl_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (b + k < nblocks) row[b + k] = off;
            off += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

#define SCAN_SINGLE_MAX (8 * SCAN_TILE)

template <bool POPC>
static int device_exclusive_scan(nbe_engine* e, const uint32_t* in, uint32_t* out, size_t len)
{
    if (len == 0) return NBE_OK;
    if (len <= SCAN_SINGLE_MAX && e->small_max > 0) {
        hipLaunchKernelGGL((scan_single_kernel<POPC>), dim3(1), dim3(256), 0, e->stream, in, len, out);
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    }
    const unsigned nblk = cdiv(len, SCAN_TILE);
    if (nblk > e->tw.scan_tmp_len) { nbe_set_error("scan workspace too small"); return NBE_E_INVALID; }
    hipLaunchKernelGGL((scan_reduce_kernel<POPC>), dim3(nblk), dim3(256), 0, e->stream, in, len, e->tw.scan_tmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, e->stream, e->tw.scan_tmp, (int)nblk);
    hipLaunchKernelGGL((scan_apply_kernel<POPC>), dim3(nblk), dim3(256), 0, e->stream, in, len, e->tw.scan_tmp, out);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// ---- Morton keys (cell_index and body_key: tree_keys.h) --------------------------------------------------------
// One key per body.  A body is kept iff lo <= p < hi on every axis (contains, tree.go:65-72; NaN fails);
// others get the all-ones key, sort to the end and stay out of the tree for good (tree.go:290-292).
// A workgroup handles one tile of the radix sort and also counts the tile's lowest digits (the histogram of the first
// pass), while the keys are still in registers; the first scatter pass takes a body's index from its position, so no
// index array is written here.  The number of live bodies falls out of the sort's last pass (radix_scatter_kernel).
template <typename R4, int T>
__global__ __launch_bounds__(T) void keys_kernel(const R4* __restrict__ posm, int n, double lox, double loy, double loz,
                                                 double width, uint64_t* __restrict__ keys, int nblocks,
                                                 uint32_t* __restrict__ block_hist, int by_xcd)
{
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const int tile = by_xcd ? (int)xcd_contiguous_block() : (int)blockIdx.x;
    const int base = tile * (T * SORT_KPT);
#pragma unroll
    for (int k = 0; k < SORT_KPT; k++) {
        const int i = base + k * T + threadIdx.x;
        if (i < n) {
            const R4 p = posm[i];
            const uint64_t key = body_key((double)p.x, (double)p.y, (double)p.z, lox, loy, loz, width);
            keys[i] = key;
            atomicAdd(&h[(uint32_t)key & 255u], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) block_hist[(size_t)threadIdx.x * nblocks + tile] = h[threadIdx.x];
}

// ---- LSD radix sort, 8-bit digits, (key, index) pairs ---------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void radix_hist_kernel(const uint64_t* __restrict__ keys, int n, int shift, int nblocks,
                                                       uint32_t* __restrict__ block_hist, int by_xcd)
{
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const int tile = by_xcd ? (int)xcd_contiguous_block() : (int)blockIdx.x;     // the XCD that will scatter this tile also counts it
    const int base = tile * (T * SORT_KPT);
#pragma unroll
    for (int k = 0; k < SORT_KPT; k++) {
        int i = base + k * T + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) block_hist[(size_t)threadIdx.x * nblocks + tile] = h[threadIdx.x];
}

// Stable scatter.  Each wave owns a contiguous share of the tile (512 keys) and walks it 64 keys at a time; within a
// round, lanes holding the same digit find each other with 8 ballots, and a per-wave digit counter in LDS
// carries the running count across rounds.  The tile is then reordered in LDS and written out in runs.
template <int T>
__global__ __launch_bounds__(T) void radix_scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                          uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ totals, int n, int shift, int nblocks,
                                                          int* __restrict__ alive_out, int by_xcd)
{
    constexpr int WAVES = T / 64, TILE = T * SORT_KPT;
    __shared__ uint32_t wcount[WAVES][256];
    __shared__ uint32_t wbase[WAVES][256];
    __shared__ uint32_t dbase[256], gdelta[256];
    __shared__ uint64_t skey[TILE];
    __shared__ uint32_t sval[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Neighbouring tiles write neighbouring slots of every digit's run: with the workgroups of one XCD on a contiguous
    // eighth of the tiles, the halves of a line meet in one L2 instead of being written back from two (16M bodies:
    // 159 -> 140 us per pass, and the histogram pass over the same tiles 53 -> 39 us).  Sorts of a few hundred tiles,
    // which fit the caches anyway, ran 5 % slower that way and keep the plain numbering.
    const int tile = by_xcd ? (int)xcd_contiguous_block() : (int)blockIdx.x;
    for (int w = tid; w < WAVES * 256; w += T) (&wcount[0][0])[w] = 0;
    __syncthreads();

    volatile uint32_t* wc = wcount[wave];
    const int base = tile * TILE + wave * (TILE / WAVES);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint64_t k[SORT_KPT];
    uint32_t v[SORT_KPT], rank[SORT_KPT];
#pragma unroll
    for (int r = 0; r < SORT_KPT; r++) {
        const int i = base + r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? kin[i] : KEY_DEAD;
        v[r] = valid ? (vin ? vin[i] : (uint32_t)i) : 0u;       // first pass: the body's index is its position
        const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t prior = 0;
        if (valid) prior = wc[digit];
        rank[r] = prior + (uint32_t)__popcll(peers & lt_mask);
        if (valid && (peers & lt_mask) == 0) wc[digit] = prior + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // Per digit: offsets of the four waves within the digit, the digit's count in this tile, and (after the
    // scan) the digit's first slot in the tile's digit-sorted order.
    // (threads 0..255 own one digit each; in a wider workgroup the others only keep the barriers company)
    const int dt = tid & 255;
    uint32_t dcount = 0;
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) { wbase[w][tid] = run; run += wcount[w][tid]; }
        dcount = run;
    }
    {
        uint32_t tot;
        const uint32_t dstart = block_excl_scan_digits<T>(dcount, &tot);     // contains the barriers that publish wbase
        // keys with a smaller digit anywhere + keys with this digit in earlier tiles = where this digit's run goes
        // in the output; minus its first slot in the tile: global = slot + gdelta
        const uint32_t below = block_excl_scan_digits<T>(tid < 256 ? totals[dt] : 0u, &tot);
        if (tid < 256) {
            dbase[tid] = dstart;
            gdelta[tid] = below + offsets[(size_t)tid * nblocks + tile] - dstart;
            // Last pass (key bits 56..63): every live key (63 bits) has a digit below 255, the all-ones key of a body
            // outside the bound has 255, so the keys below digit 255 are the live bodies.
            if (alive_out && tile == 0 && tid == 255) *alive_out = (int)below;
        }
    }
    __syncthreads();
    // Stage the tile in digit-sorted order in LDS, then write it out with consecutive threads on consecutive
    // slots: slots of one digit are consecutive in the output too, so the stores coalesce into runs instead
    // of 8- and 4-byte scattered writes.
#pragma unroll
    for (int r = 0; r < SORT_KPT; r++) {
        const int i = base + r * 64 + lane;
        if (i < n) {
            const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
            const uint32_t slot = dbase[digit] + wbase[wave][digit] + rank[r];
            skey[slot] = k[r];
            sval[slot] = v[r];
        }
    }
    __syncthreads();
    const int tile_n = min(TILE, n - tile * TILE);
#pragma unroll
    for (int r = 0; r < SORT_KPT; r++) {
        const int slot = r * T + tid;
        if (slot < tile_n) {
            const uint64_t key = skey[slot];
            const uint32_t pos = (uint32_t)slot + gdelta[(uint32_t)(key >> shift) & 255u];
            kout[pos] = key;
            vout[pos] = sval[slot];
        }
    }
}

// ---- few bodies: keys and the whole sort in one workgroup ------------------------------------------------------------
// Up to SMALL_SORT_MAX bodies: 16 waves hold 8 keys per lane in registers and run the eight 8-bit passes through
// LDS with the same stable ballot ranking as above; one launch replaces the 41 of the tiled sort, which at this
// size are nothing but launch latency.  Also counts the live bodies.
#define SMALL_SORT_MAX 8192
#define SMALL_WAVES 16
#define SMALL_KPT 8      // SMALL_WAVES * 64 * SMALL_KPT == SMALL_SORT_MAX

template <typename R4>
__global__ __launch_bounds__(1024) void small_sort_kernel(const R4* __restrict__ posm, int n, double lox, double loy, double loz,
                                                          double width, uint64_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ idx_out, int* __restrict__ counters)
{
    __shared__ uint64_t skey[SMALL_SORT_MAX];
    __shared__ uint32_t sval[SMALL_SORT_MAX];
    __shared__ uint32_t wcount[SMALL_WAVES][256];   // per wave and digit: count, then offset within the digit
    __shared__ uint32_t dbase[256];
    __shared__ uint32_t wsum[SMALL_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each wave owns a contiguous run of 64 * rounds slots, rounds = keys per lane actually needed for n bodies
    const int rounds = (n + 64 * SMALL_WAVES - 1) / (64 * SMALL_WAVES);
    const int chunk = wave * (64 * rounds);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint64_t k[SMALL_KPT];
    uint32_t v[SMALL_KPT];
    uint32_t live = 0;
#pragma unroll
    for (int r = 0; r < SMALL_KPT; r++) {
        if (r >= rounds) break;
        const int i = chunk + r * 64 + lane;
        k[r] = KEY_DEAD;
        v[r] = (uint32_t)i;
        if (i < n) {
            const R4 p = posm[i];
            k[r] = body_key((double)p.x, (double)p.y, (double)p.z, lox, loy, loz, width);
            live += k[r] != KEY_DEAD;
        }
    }
    for (int pass = 0; pass < 8; pass++) {
        const int shift = pass * 8;
#pragma unroll
        for (int w = 0; w < SMALL_WAVES / 4; w++) wcount[w * 4 + (tid >> 8)][tid & 255] = 0;
        __syncthreads();
        volatile uint32_t* wc = wcount[wave];
        uint32_t rank[SMALL_KPT];
#pragma unroll
        for (int r = 0; r < SMALL_KPT; r++) {
            if (r >= rounds) break;
            const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
            unsigned long long peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t prior = wc[digit];
            rank[r] = prior + (uint32_t)__popcll(peers & lt_mask);
            if ((peers & lt_mask) == 0) wc[digit] = prior + (uint32_t)__popcll(peers);
        }
        __syncthreads();
        // threads 0..255 own one digit each: offsets of the waves within the digit, then the digit's first slot
        uint32_t dcount = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; w++) { const uint32_t c = wcount[w][tid]; wcount[w][tid] = dcount; dcount += c; }
        }
        {
            uint32_t inc = wave_incl_scan(dcount);
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (int w = 0; w < wave; w++) base += wsum[w];
            if (tid < 256) dbase[tid] = base + inc - dcount;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMALL_KPT; r++) {
            if (r >= rounds) break;
            const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
            const uint32_t slot = dbase[digit] + wcount[wave][digit] + rank[r];
            skey[slot] = k[r];
            sval[slot] = v[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMALL_KPT; r++) {
            if (r >= rounds) break;
            k[r] = skey[chunk + r * 64 + lane];
            v[r] = sval[chunk + r * 64 + lane];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < SMALL_KPT; r++) {
        if (r >= rounds) break;
        const int i = chunk + r * 64 + lane;
        if (i < n) { keys_out[i] = k[r]; idx_out[i] = v[r]; }     // padding slots sorted behind every real body
    }
    // live bodies
    for (int d = 32; d >= 1; d >>= 1) live += __shfl_down(live, d, 64);
    if (lane == 0) wsum[wave] = live;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < SMALL_WAVES; w++) tot += wsum[w];
        counters[COUNTER_SORTED_ALIVE] = (int)tot;
    }
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  emit_nodes_kernel, whose neighbouring
// workgroups read the same lines of the prefix sums and cell counts, runs the workgroups of one XCD over one
// contiguous eighth of the index range instead, so that a line is fetched into one L2, not into several (16M
// bodies: 874 -> 759 us; the state gather and the radix tree kernel did not gain from it: 519 -> 526, 433 -> 454 us).
// Bijective for any grid size; a placement assumption that only affects speed.
__device__ __forceinline__ unsigned xcd_contiguous_block(void)
{
    const unsigned nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
}

// Reorders the state into sorted order (gather): dst[i] = src[idx[i]].
// Bodies with the same key (coincident, or in the same depth-21 cell) come out of the stable sort in the order of the
// previous sort, which depends on the run's history and, on the distributed tree, on when a body reached its rank.  They
// are put in the order of their upload (run-wide) indices instead, so that their order, and with it the order in which
// their forces are summed, is a property of the scene: a member of a run of up to TIE_MAX equal keys counts the members
// with smaller indices and goes to that place of the run (every member sees the whole run and they all decide alike;
// longer runs keep the sort's order).
#define TIE_MAX 64
template <typename S4>
__global__ __launch_bounds__(256) void permute_kernel(const uint32_t* __restrict__ idx, const uint64_t* __restrict__ keys, int n,
                                                      const S4* __restrict__ pos_in,
                                                      const S4* __restrict__ vel_in, const uint32_t* __restrict__ orig_in,
                                                      const float4* __restrict__ mirror_in, const double* __restrict__ radius_in,
                                                      S4* __restrict__ pos_out, S4* __restrict__ vel_out,
                                                      uint32_t* __restrict__ orig_out, float4* __restrict__ mirror_out,
                                                      double* __restrict__ radius_out, int* __restrict__ counters,
                                                      uint32_t* __restrict__ lvlmask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    // the cell marks of the coming build start from zero (n + 1 entries; radix_tree_kernel sets bits with atomics)
    if (i <= n) lvlmask[i] = 0;
    if (i == 0) {
        // The sort left its live count aside (it may have run ahead, while the state was still in the previous
        // order and counters[0..3] still described that order); from here on the state is in the new order.
        counters[0] = counters[COUNTER_SORTED_ALIVE];
        counters[1] = 0; counters[2] = 0; counters[3] = 0;
    }
    if (i >= n) return;
    // (four bodies per thread with all gathers in flight together: 590 against 514 us at 16M bodies)
    const uint32_t j = idx[i];
    int to = i;
    const uint64_t key = keys[i];
    if (key != KEY_DEAD && ((i > 0 && keys[i - 1] == key) || (i + 1 < n && keys[i + 1] == key))) {
        int a = i, b = i;                             // the run [a, b] of this key, as far as TIE_MAX steps show it
        while (a > 0 && i - a < TIE_MAX && keys[a - 1] == key) a--;
        while (b + 1 < n && b - i < TIE_MAX && keys[b + 1] == key) b++;
        const bool whole = (a == 0 || keys[a - 1] != key) && (b + 1 >= n || keys[b + 1] != key);
        if (whole && b - a < TIE_MAX) {
            const uint32_t mine = orig_in[j];
            int before = 0;
            for (int t = a; t <= b; t++) before += orig_in[idx[t]] < mine ? 1 : 0;
            to = a + before;
        }
    }
    pos_out[to] = pos_in[j];
    vel_out[to] = vel_in[j];
    orig_out[to] = orig_in[j];
    if (mirror_in) mirror_out[to] = mirror_in[j];
    if (radius_in) radius_out[to] = radius_in[j];
}

// ---- binary radix tree over the sorted keys ----------------------------------------------------------------
// Also decides which binary nodes are octree cells: a node is a branching cell iff its depth (common prefix / 3)
// differs from its parent's.  The parent decides that for its two children, whose depths follow from the keys at the
// ends of their ranges, writes the verdict to bt_cell[child] and sets bit `depth` in lvlmask[first leaf of child]
// (lvlmask was zeroed by permute_kernel); the root is always a cell.
// The searches of a node with a short range (most nodes) probe keys near its own index: the workgroup keeps the keys
// of its 256 nodes and RT_HALO neighbours on either side in LDS and only longer probes go to memory.
#define RT_HALO 128
struct KeyWindow {
    const uint64_t* __restrict__ keys;
    const uint64_t* lds;
    int lo, hi, na;                       // lds[j - lo] == keys[j] for lo <= j < hi
    __device__ __forceinline__ uint64_t at(int j) const { return (j >= lo && j < hi) ? lds[j - lo] : keys[j]; }
    // delta(i, j) as key_delta: common prefix length, index as tie-break for equal keys, -1 outside [0, na)
    __device__ __forceinline__ int delta(int i, uint64_t ki, int j) const
    {
        if (j < 0 || j >= na) return -1;
        const uint64_t kj = at(j);
        if (ki != kj) return __clzll((long long)(ki ^ kj)) - 1;
        return 63 + __clz(i ^ j);
    }
};

__global__ __launch_bounds__(256) void radix_tree_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ counters,
                                                         int* __restrict__ bt_first, int* __restrict__ bt_last,
                                                         int* __restrict__ bt_parent, uint8_t* __restrict__ bt_level,
                                                         uint8_t* __restrict__ bt_cell, uint32_t* __restrict__ lvlmask)
{
    __shared__ uint64_t skeys[256 + 2 * RT_HALO];
    const int na = counters[0];
    const int base = blockIdx.x * 256;
    if (base >= na - 1) return;
    KeyWindow kw;
    kw.keys = keys; kw.lds = skeys; kw.na = na;
    kw.lo = base - RT_HALO < 0 ? 0 : base - RT_HALO;
    kw.hi = base + 256 + RT_HALO > na ? na : base + 256 + RT_HALO;
    for (int j = kw.lo + threadIdx.x; j < kw.hi; j += 256) skeys[j - kw.lo] = keys[j];
    __syncthreads();
    const int i = base + threadIdx.x;
    if (i >= na - 1) return;
    const uint64_t ki = kw.at(i);
    const int dl = kw.delta(i, ki, i - 1), dr = kw.delta(i, ki, i + 1);
    const int d = dr > dl ? 1 : -1;
    const int dmin = dr > dl ? dl : dr;
    int lmax = 2;
    while (kw.delta(i, ki, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (kw.delta(i, ki, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = kw.delta(i, ki, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (kw.delta(i, ki, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    bt_first[i] = lo;
    bt_last[i] = hi;
    const int lvl = delta_level(dnode);
    bt_level[i] = (uint8_t)lvl;
    if (i == 0) {
        bt_parent[0] = -1;
        bt_cell[0] = 1;
        atomicOr(&lvlmask[0], 1u << lvl);
    }
    if (lo != gamma) {                               // left child is internal node gamma, covering [lo, gamma]
        bt_parent[gamma] = i;
        const int cl = delta_level(kw.delta(lo, kw.at(lo), gamma));
        bt_cell[gamma] = cl != lvl;
        if (cl != lvl) atomicOr(&lvlmask[lo], 1u << cl);
    }
    if (hi != gamma + 1) {                           // right child is internal node gamma+1, covering [gamma+1, hi]
        bt_parent[gamma + 1] = i;
        const int cl = delta_level(kw.delta(gamma + 1, kw.at(gamma + 1), hi));
        bt_cell[gamma + 1] = cl != lvl;
        if (cl != lvl) atomicOr(&lvlmask[gamma + 1], 1u << cl);
    }
}

// ---- fp64 prefix sums of (m, m*x, m*y, m*z) over the sorted bodies ---------------------------------------------
// A cell covers a contiguous range [a, b] of sorted bodies, so its mass and first moments are differences of
// these prefix sums: P[b+1] - P[a].  Used for fp32 node records: the cancellation error of the difference
// (~1e-16 * |P|) stays below the fp32 rounding of the stored centre of mass.  fp64 node records are reduced
// level by level instead (cell_mass_kernel) to keep full fp64 accuracy.
#define PSUM_TILE 2048   // bodies per workgroup (256 threads x 8)

struct D4 { double m, x, y, z; };
__device__ __forceinline__ D4 d4_add(D4 a, D4 b) { return D4{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D4 d4_shfl_up(D4 v, int d)
{
    return D4{__shfl_up(v.m, d, 64), __shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64), __shfl_up(v.z, d, 64)};
}

template <typename R4> __device__ __forceinline__ D4 body_moments(const R4* __restrict__ posm, int i, int n)
{
    if (i >= n) return D4{0, 0, 0, 0};
    const R4 p = posm[i];
    const double m = (double)p.w;
    return D4{m, m * (double)p.x, m * (double)p.y, m * (double)p.z};
}

// Exclusive scan across the 256 threads of a workgroup; *total = workgroup sum.
__device__ __forceinline__ D4 block_excl_scan_d4(D4 v, D4* total)
{
    __shared__ D4 wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    D4 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        D4 t = d4_shfl_up(inc, d);
        if (lane >= d) inc = d4_add(inc, t);
    }
    if (lane == 63) wsum[wave] = inc;
    // exclusive value = inclusive value of the lane below (never "inclusive - own": that would round twice
    // and turn a non-finite dead body into NaN for everyone)
    D4 excl = d4_shfl_up(inc, 1);
    if (lane == 0) excl = D4{0, 0, 0, 0};
    __syncthreads();
    D4 base{0, 0, 0, 0}, tot{0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) base = d4_add(base, wsum[w]);
        tot = d4_add(tot, wsum[w]);
    }
    __syncthreads();
    *total = tot;
    return d4_add(base, excl);
}

// A thread sums eight consecutive bodies (that order is part of the deterministic result), but loading them
// straight from memory makes the lanes of a wave read 128 bytes apart: at 16M bodies the caches then fetched 3.4
// times the bytes of the array.  The tile is loaded with consecutive lanes on consecutive bodies into LDS and
// read back in the summation order; rows of eight elements are padded by one against bank conflicts.
template <typename R4> struct PsumTile {
    R4 e[PSUM_TILE + PSUM_TILE / 8];
    __device__ __forceinline__ void load(const R4* __restrict__ posm, int tile_base, int n)
    {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = k * 256 + threadIdx.x, i = tile_base + j;
            R4 v; v.x = 0; v.y = 0; v.z = 0; v.w = 0;
            if (i < n) v = posm[i];
            e[j + (j >> 3)] = v;
        }
        __syncthreads();
    }
    // moments of body tile_base + threadIdx.x * 8 + k (zero past the end: a padded body has mass 0)
    __device__ __forceinline__ D4 moments(int k) const
    {
        const R4 p = e[threadIdx.x * 9 + k];
        const double m = (double)p.w;
        return D4{m, m * (double)p.x, m * (double)p.y, m * (double)p.z};
    }
};

template <typename R4>
__global__ __launch_bounds__(256) void psum_reduce_kernel(const R4* __restrict__ posm, int n, D4* __restrict__ tile_sums)
{
    __shared__ PsumTile<R4> tile;
    tile.load(posm, blockIdx.x * PSUM_TILE, n);
    D4 s{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) s = d4_add(s, tile.moments(k));
    D4 tot;
    block_excl_scan_d4(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// One workgroup turns the tile sums into exclusive tile offsets, in place.
__global__ __launch_bounds__(256) void psum_scan_sums_kernel(D4* __restrict__ sums, int nsums)
{
    D4 carry{0, 0, 0, 0};
    for (int base = 0; base < nsums; base += 256 * 4) {
        D4 v[4], s{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + threadIdx.x * 4 + k;
            v[k] = i < nsums ? sums[i] : D4{0, 0, 0, 0};
            s = d4_add(s, v[k]);
        }
        D4 tot;
        D4 off = d4_add(carry, block_excl_scan_d4(s, &tot));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + threadIdx.x * 4 + k;
            if (i < nsums) sums[i] = off;
            off = d4_add(off, v[k]);
        }
        carry = d4_add(carry, tot);
    }
}

// prefix[i] = sum over bodies k < i, for i in [0, n]; summation order is fixed (thread-serial, then wave, then
// workgroup, then tile offsets), so every engine computes the same bits.
template <typename R4>
__global__ __launch_bounds__(256) void psum_apply_kernel(const R4* __restrict__ posm, int n, const D4* __restrict__ tile_offsets,
                                                         D4* __restrict__ prefix)
{
    __shared__ PsumTile<R4> tile;
    // A thread owns eight consecutive 32-byte prefix records; stored straight from registers, the lanes of a wave
    // write 256 bytes apart and the memory system sees partial lines (84 MB written for a 33.5 MB array at 1M
    // bodies).  The records go through LDS instead, four per thread at a time (rows of four padded by one against
    // bank conflicts), and leave with consecutive lanes on consecutive records.
    __shared__ D4 stage[256 * 5];
    tile.load(posm, blockIdx.x * PSUM_TILE, n);
    D4 v[8], s{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = tile.moments(k); s = d4_add(s, v[k]); }
    D4 tot;
    D4 off = d4_add(tile_offsets[blockIdx.x], block_excl_scan_d4(s, &tot));
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            stage[threadIdx.x * 5 + k] = off;
            off = d4_add(off, v[half * 4 + k]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = q * 256 + threadIdx.x, owner = r >> 2, kk = r & 3;
            const int i = blockIdx.x * PSUM_TILE + owner * 8 + half * 4 + kk;
            if (i <= n) prefix[i] = stage[owner * 5 + kk];
        }
    }
}

// Few bodies: the three steps above in one workgroup with a running carry.  The summation order differs from the
// tiled version only in where the carries are added, which every engine does alike for a given n.
template <typename R4>
__global__ __launch_bounds__(256) void psum_single_kernel(const R4* __restrict__ posm, int n, D4* __restrict__ prefix)
{
    D4 carry{0, 0, 0, 0};
    for (int base0 = 0; base0 <= n; base0 += PSUM_TILE) {
        const int base = base0 + threadIdx.x * 8;
        D4 v[8], s{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = body_moments(posm, base + k, n); s = d4_add(s, v[k]); }
        D4 tot;
        D4 off = d4_add(carry, block_excl_scan_d4(s, &tot));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (base + k <= n) prefix[base + k] = off;
            off = d4_add(off, v[k]);
        }
        carry = d4_add(carry, tot);
    }
}

// ---- exact prefix sums (NBE_SUMS_EXACT; the 128-bit arithmetic is in tree_sums.h) ----------------------------------------
template <typename R4> __device__ __forceinline__ Q4 body_moments_q(const R4& p, int shift_m, int shift_x)
{
    const double m = (double)p.w;
    return Q4{q_from_double(m, shift_m), q_from_double(m * (double)p.x, shift_x), q_from_double(m * (double)p.y, shift_x),
              q_from_double(m * (double)p.z, shift_x)};
}

// Exclusive scan across the 256 threads of a workgroup; *total = workgroup sum.
__device__ __forceinline__ Q4 block_excl_scan_q4(const Q4& v, Q4* total)
{
    __shared__ Q4 wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Q4 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Q4 t = q4_shfl_up(inc, d);
        if (lane >= d) inc = q4_add(inc, t);
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    Q4 base = q4_zero(), tot = q4_zero();
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) base = q4_add(base, wsum[w]);
        tot = q4_add(tot, wsum[w]);
    }
    __syncthreads();
    *total = tot;
    return q4_add(base, q4_sub(inc, v));
}

// Tile totals.  The order of addition is free, so consecutive lanes read consecutive bodies.
template <typename R4>
__global__ __launch_bounds__(256) void psumq_reduce_kernel(const R4* __restrict__ posm, const int* __restrict__ counters, int shift_m,
                                                           int shift_x, Q4* __restrict__ tile_sums)
{
    const int na = counters[0];
    Q4 s = q4_zero();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int i = blockIdx.x * PSUM_TILE + k * 256 + threadIdx.x;
        if (i < na) s = q4_add(s, body_moments_q(posm[i], shift_m, shift_x));
    }
    Q4 tot;
    block_excl_scan_q4(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void psumq_scan_sums_kernel(Q4* __restrict__ sums, int nsums)
{
    Q4 carry = q4_zero();
    for (int base = 0; base < nsums; base += 256 * 4) {
        Q4 v[4], s = q4_zero();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + threadIdx.x * 4 + k;
            v[k] = i < nsums ? sums[i] : q4_zero();
            s = q4_add(s, v[k]);
        }
        Q4 tot;
        Q4 off = q4_add(carry, block_excl_scan_q4(s, &tot));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + threadIdx.x * 4 + k;
            if (i < nsums) sums[i] = off;
            off = q4_add(off, v[k]);
        }
        carry = q4_add(carry, tot);
    }
}

// prefix[i] = sum over live bodies k < i, for i in [0, n] (bodies outside the bound, sorted behind the live ones, count
// as nothing).  A thread owns eight consecutive bodies; the positions come in through LDS with consecutive lanes on
// consecutive bodies, and the 64-byte prefix records leave the same way, two per thread at a time, through the same
// LDS (rows of two records padded by 16 bytes against bank conflicts).
template <typename R4>
__global__ __launch_bounds__(256) void psumq_apply_kernel(const R4* __restrict__ posm, int n, const int* __restrict__ counters,
                                                          int shift_m, int shift_x, const Q4* __restrict__ tile_offsets,
                                                          Q4* __restrict__ prefix)
{
    constexpr int STAGE_ROW = 9;                      // 16-byte words per row: two records of four words, one word of padding
    // (fp64 positions, the reference's own arithmetic and not the fast path, are read straight from memory: their tile
    // would not fit the workgroup's LDS)
    constexpr bool TILED = sizeof(R4) == 16;
    union Shared {
        PsumTile<float4> tile;
        uint4 stage[256 * STAGE_ROW];
    };
    __shared__ Shared sh;
    const int na = counters[0];
    const int tile_base = blockIdx.x * PSUM_TILE;
    if constexpr (TILED) sh.tile.load((const float4*)posm, tile_base, n);
    Q4 v[8], s = q4_zero();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int i = tile_base + threadIdx.x * 8 + k;
        v[k] = q4_zero();
        if constexpr (TILED) {
            if (i < na) v[k] = body_moments_q(sh.tile.e[threadIdx.x * 9 + k], shift_m, shift_x);
        } else {
            if (i < na) v[k] = body_moments_q(posm[i], shift_m, shift_x);
        }
        s = q4_add(s, v[k]);
    }
    Q4 tot;
    Q4 off = q4_add(tile_offsets[blockIdx.x], block_excl_scan_q4(s, &tot));     // (its barriers end the reads of the tile)
    uint4* out = (uint4*)prefix;
#pragma unroll
    for (int part = 0; part < 4; part++) {
        if (part) __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint4* row = sh.stage + threadIdx.x * STAGE_ROW + 4 * k;
            row[0] = make_uint4((unsigned)off.m.lo, (unsigned)(off.m.lo >> 32), (unsigned)off.m.hi, (unsigned)(off.m.hi >> 32));
            row[1] = make_uint4((unsigned)off.x.lo, (unsigned)(off.x.lo >> 32), (unsigned)off.x.hi, (unsigned)(off.x.hi >> 32));
            row[2] = make_uint4((unsigned)off.y.lo, (unsigned)(off.y.lo >> 32), (unsigned)off.y.hi, (unsigned)(off.y.hi >> 32));
            row[3] = make_uint4((unsigned)off.z.lo, (unsigned)(off.z.lo >> 32), (unsigned)off.z.hi, (unsigned)(off.z.hi >> 32));
            off = q4_add(off, v[part * 2 + k]);
        }
        __syncthreads();
        // 512 records of four 16-byte words leave as 2048 words, consecutive lanes on consecutive words
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int w = q * 256 + threadIdx.x;      // word of this part
            const int rec = w >> 2, word = w & 3;
            const int owner = rec >> 1, kk = rec & 1;
            const long long i = (long long)tile_base + owner * 8 + part * 2 + kk;
            if (i <= n) out[4 * i + word] = sh.stage[owner * STAGE_ROW + 4 * kk + word];
        }
    }
}

// Few bodies: one workgroup with a running carry.
template <typename R4>
__global__ __launch_bounds__(256) void psumq_single_kernel(const R4* __restrict__ posm, int n, const int* __restrict__ counters,
                                                           int shift_m, int shift_x, Q4* __restrict__ prefix)
{
    const int na = counters[0];
    Q4 carry = q4_zero();
    for (int base0 = 0; base0 <= n; base0 += PSUM_TILE) {
        const int base = base0 + threadIdx.x * 8;
        Q4 v[8], s = q4_zero();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v[k] = base + k < na ? body_moments_q(posm[base + k], shift_m, shift_x) : q4_zero();
            s = q4_add(s, v[k]);
        }
        Q4 tot;
        Q4 off = q4_add(carry, block_excl_scan_q4(s, &tot));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (base + k <= n) prefix[base + k] = off;
            off = q4_add(off, v[k]);
        }
        carry = q4_add(carry, tot);
    }
}

// Writes the pre-order node array: thread t < n handles sorted leaf t, thread n + i handles binary node i.
// cellcnt[a] = number of branching cells whose first leaf is < a (exclusive scan of popcount(lvlmask)).
// NBE_SUMS_PREFIX and NBE_SUMS_EXACT: cells get their mass and centre of mass here, from the prefix sums;
// NBE_SUMS_TREE: leaves seed node_sum and cell_mass_kernel fills the cells afterwards.
template <typename R4, typename NodeT, int SUMS>
__global__ __launch_bounds__(256) void emit_nodes_kernel(int n, int* __restrict__ counters, const R4* __restrict__ posm,
                                                         const int* __restrict__ bt_first, const int* __restrict__ bt_last,
                                                         const uint8_t* __restrict__ bt_cell, const uint8_t* __restrict__ bt_level,
                                                         const uint32_t* __restrict__ lvlmask, const uint32_t* __restrict__ cellcnt,
                                                         double bound_width, double theta, double eps2, NodeT* __restrict__ nodes,
                                                         uint8_t* __restrict__ node_level, double* __restrict__ node_sum,
                                                         const D4* __restrict__ prefix, const Q4* __restrict__ prefixq,
                                                         int shift_m, int shift_x, const uint64_t* __restrict__ keys,
                                                         const int* __restrict__ bt_parent, CNode* __restrict__ geom)
{
    // geom != nullptr (a rank of the distributed tree without collisions): the key, depth and parent depth of every node
    // are written here too, as emit_cnodes_kernel (collide.hip) would in a pass of its own over the same arrays
    const int na = counters[0];
    // Even workgroups (in the XCD-contiguous numbering) emit 256 leaves, odd ones the cells among the 256 binary nodes of
    // the same index range: the two kinds read the same lines of the cell counts, and every XCD gets its share of both.
    const unsigned vb = xcd_contiguous_block();
    const int local = (int)((vb >> 1) * 256 + threadIdx.x);
    const int t = (vb & 1u) ? n + local : (local < n ? local : 2 * n);      // >= 2n: nothing to do
    if (t == 0) {
        const int cells = na > 0 ? (int)cellcnt[na] : 0;
        counters[1] = na + cells;
        counters[2] = cells;
        // sentinel behind the last node: massless, never opens, skips to itself (a finished walk parks here)
        NodeT snt;
        snt.x = 0; snt.y = 0; snt.z = 0; snt.m = 0;
        set_edge<NodeT>(snt, 0.0, theta, eps2);
        snt.skip = na + cells;
        nodes[na + cells] = snt;
    }
    if (t < n) {
        if (t >= na) return;
        const int pos = t + (int)cellcnt[t + 1];
        const R4 p = posm[t];
        NodeT nd;
        nd.x = p.x; nd.y = p.y; nd.z = p.z; nd.m = p.w;
        set_edge<NodeT>(nd, 0.0, theta, eps2);
        nd.skip = pos + 1;
        nodes[pos] = nd;
        if (geom) {
            CNode c;
            c.pmask = 0; c.pad0 = 0; c.parent = -1;
            const uint64_t ki = keys[t];
            const int dl = key_delta(keys, na, t, ki, t - 1), dr = key_delta(keys, na, t, ki, t + 1);
            const int d = dl > dr ? dl : dr;
            c.key = ki; c.body = t; c.skip = pos + 1;
            if (d < 0) { c.level = 0; c.plevel = 255; }          // a single body: the root is its leaf
            else { const int pl = delta_level(d); c.plevel = (uint8_t)pl; c.level = (uint8_t)(pl + 1); }
            geom[pos] = c;
        }
        if (SUMS == NBE_SUMS_TREE) {
            node_level[pos] = 255;
            const double m = (double)p.w;
            node_sum[4 * (size_t)pos + 0] = m;
            node_sum[4 * (size_t)pos + 1] = m * (double)p.x;
            node_sum[4 * (size_t)pos + 2] = m * (double)p.y;
            node_sum[4 * (size_t)pos + 3] = m * (double)p.z;
        }
        return;
    }
    const int i = t - n;
    if (i >= na - 1) return;
    if (!bt_cell[i]) return;
    const int lvl = bt_level[i];
    const int a = bt_first[i], b = bt_last[i];
    const int pos = a + (int)cellcnt[a] + __popc(lvlmask[a] & ((1u << lvl) - 1u));
    NodeT nd;
    nd.x = 0; nd.y = 0; nd.z = 0; nd.m = 0;
    if (SUMS == NBE_SUMS_PREFIX) {
        const D4 lo = prefix[a], hi = prefix[b + 1];
        const double m = hi.m - lo.m;
        const double inv = m != 0.0 ? 1.0 / m : 0.0;
        nd.x = (hi.x - lo.x) * inv; nd.y = (hi.y - lo.y) * inv; nd.z = (hi.z - lo.z) * inv; nd.m = m;
    }
    if (SUMS == NBE_SUMS_EXACT) exact_cell(q4_sub(prefixq[b + 1], prefixq[a]), shift_m, shift_x, nd);
    set_edge<NodeT>(nd, bound_width * (1.0 / (double)(1u << lvl)), theta, eps2);   // edge: exact halving of a power of two
    nd.skip = (b + 1) + (int)cellcnt[b + 1];
    nodes[pos] = nd;
    if (geom) {
        CNode c;
        c.pmask = 0; c.pad0 = 0; c.parent = -1;
        const int par = bt_parent[i];
        c.key = keys[a]; c.body = -1; c.skip = nd.skip;
        c.level = (uint8_t)lvl;
        c.plevel = par < 0 ? (uint8_t)255 : bt_level[par];
        geom[pos] = c;
    }
    if (SUMS == NBE_SUMS_TREE) node_level[pos] = (uint8_t)lvl;
}

// One depth per launch, deepest first: a cell sums its children (already final) in pre-order, which is the
// octant order.  Totals are kept in fp64 beside the node array; the node gets mass and sum/mass.
template <typename NodeT>
__global__ __launch_bounds__(256) void cell_mass_kernel(const int* __restrict__ counters, int level, NodeT* __restrict__ nodes,
                                                        const uint8_t* __restrict__ node_level, double* __restrict__ node_sum)
{
    const int total = counters[1];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    if (node_level[p] != level) return;
    const int end = nodes[p].skip;
    double m = 0, sx = 0, sy = 0, sz = 0;
    for (int c = p + 1; c < end; c = nodes[c].skip) {
        m += node_sum[4 * (size_t)c + 0];
        sx += node_sum[4 * (size_t)c + 1];
        sy += node_sum[4 * (size_t)c + 2];
        sz += node_sum[4 * (size_t)c + 3];
    }
    node_sum[4 * (size_t)p + 0] = m;
    node_sum[4 * (size_t)p + 1] = sx;
    node_sum[4 * (size_t)p + 2] = sy;
    node_sum[4 * (size_t)p + 3] = sz;
    const double inv = m != 0.0 ? 1.0 / m : 0.0;
    nodes[p].x = sx * inv;
    nodes[p].y = sy * inv;
    nodes[p].z = sz * inv;
    nodes[p].m = m;
}


// ---- the reference's running "centre of mass" (NBE_QUIRK_REF_COM) -------------------------------------------
// tree.go:156-160 gives a new leaf its body's position and mass; tree.go:189-193 then updates a cell for every
// later body that lands in it with  totalMass += m;  com = com + p * (m / totalMass)  and never rescales the old
// value.  Bodies are pushed in slice order (tree.go:287-293), so a cell's value is that recurrence over its
// bodies in upload order, whatever its depth (a cell that starts as a leaf and splits keeps its values,
// tree.go:170-177).  One-child chains, which this tree does not materialise, hold the same bodies in the same
// order as the branching cell below them and therefore the same value.  The arithmetic is kept literally (one
// rounding per operation, no contraction) so that fp64 node records equal the Go program's bit for bit.
struct RefCom { double M, x, y, z; bool any; };

// Value of lane `src` (wave-uniform index) for every lane, through scalar registers: v_readlane_b32 instead of the
// LDS crossbar of __shfl, whose latency would be paid once per body in the serial loops below.
__device__ __forceinline__ double lane_value(double v, int src)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Applies the recurrence to the bodies held by the lanes of `mask`, in ascending lane order.  Only the additions
// are serial (running mass, then running centre); the division and the products of all lanes run side by side in
// between, which leaves every intermediate value exactly as the one-body-at-a-time loop rounds it.
// The bodies are first packed into lanes 0 .. cnt-1 (one crossbar move), so that the serial loops can be unrolled
// with constant lane numbers: a step is then two v_readlane, one v_add_f64 and a two-instruction latch instead
// of a bit scan, a mask update, a compare and a branch (one wave runs this alone: instruction issue is its time).
template <int K>
__device__ __forceinline__ double lane_const(double v)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, K);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), K);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int K, int END> struct RefSerial {
    // running mass: after the call lane k (first <= k < cnt) holds m_first' + ... + m_k in `total`
    static __device__ __forceinline__ void mass(double& M, double pm, double& total, int first, int cnt, int lane)
    {
#pragma clang fp contract(off)
        if (K >= cnt) return;
        if (K >= first) {
            M = M + lane_const<K>(pm);
            if (lane == K) total = M;
        }
        RefSerial<K + 1, END>::mass(M, pm, total, first, cnt, lane);
    }
    static __device__ __forceinline__ void centre(double& x, double& y, double& z, double tx, double ty, double tz, int first, int cnt)
    {
#pragma clang fp contract(off)
        if (K >= cnt) return;
        if (K >= first) { x = x + lane_const<K>(tx); y = y + lane_const<K>(ty); z = z + lane_const<K>(tz); }
        RefSerial<K + 1, END>::centre(x, y, z, tx, ty, tz, first, cnt);
    }
};
template <int END> struct RefSerial<END, END> {
    static __device__ __forceinline__ void mass(double&, double, double&, int, int, int) {}
    static __device__ __forceinline__ void centre(double&, double&, double&, double, double, double, int, int) {}
};

__device__ __forceinline__ void ref_com_push_lanes(RefCom& c, unsigned long long mask, double pm, double px, double py, double pz)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    if (mask == 0) return;
    const int cnt = __popcll(mask);
    if (mask != (cnt == 64 ? ~0ull : (1ull << cnt) - 1ull)) {
        // lane r fetches the r-th hit (ascending lane order is kept)
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        const bool hit = (mask >> lane) & 1ull;
        int from = 0;                                   // pull: the lane number of the r-th set bit
        {
            // every hit lane announces itself to lane `rank` through the crossbar (push), then all pull
            const int dst = hit ? rank : 63;            // non-hits aim at a lane nobody reads when cnt < 64
            from = __builtin_amdgcn_ds_permute(dst << 2, hit ? lane : 0);
        }
        pm = __shfl(pm, from, 64); px = __shfl(px, from, 64); py = __shfl(py, from, 64); pz = __shfl(pz, from, 64);
    }
    int first = 0;
    if (!c.any) {                                    // tree.go:156-160: the first body's position and mass
        c.any = true;
        c.M = lane_const<0>(pm); c.x = lane_const<0>(px); c.y = lane_const<0>(py); c.z = lane_const<0>(pz);
        first = 1;
    }
    double my_total = 1.0;
    RefSerial<0, 64>::mass(c.M, pm, my_total, first, cnt, lane);              // tree.go:191
    const double s = pm / my_total;                                           // tree.go:192, b.Mass / n.totalMass
    const double tx = px * s, ty = py * s, tz = pz * s;
    RefSerial<0, 64>::centre(c.x, c.y, c.z, tx, ty, tz, first, cnt);
}

// inv[upload index] = sorted index of a live body; 0xffffffff for bodies outside the bound.
__global__ __launch_bounds__(256) void ref_inv_kernel(const int* __restrict__ counters, const uint32_t* __restrict__ orig,
                                                      uint32_t* __restrict__ inv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < counters[0]) inv[orig[i]] = (uint32_t)i;
}

// One wave per binary radix-tree node; those that are octree cells overwrite the node's mass and centre.
// A cell of at most 64 bodies ranks its members by upload index inside the wave; a larger one scans the upload
// order 64 indices at a time (inv tells whether an index lies in the cell's sorted range) and takes the hits in
// ascending order.  Either way the recurrence itself is serial: the cell of all n bodies is a chain of n steps.
template <typename R4, typename NodeT>
__global__ __launch_bounds__(256) void ref_com_kernel(int n, const int* __restrict__ counters, const R4* __restrict__ posm,
                                                      const uint32_t* __restrict__ orig, const uint32_t* __restrict__ inv,
                                                      const int* __restrict__ bt_first, const int* __restrict__ bt_last,
                                                      const uint8_t* __restrict__ bt_cell, const uint8_t* __restrict__ bt_level,
                                                      const uint32_t* __restrict__ lvlmask, const uint32_t* __restrict__ cellcnt,
                                                      NodeT* __restrict__ nodes)
{
    const int na = counters[0];
    const int i = (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= na - 1) return;
    if (!bt_cell[i]) return;
    const int lvl = bt_level[i];
    const int a = bt_first[i], b = bt_last[i];
    const int c = b - a + 1;
    const int pos = a + (int)cellcnt[a] + __popc(lvlmask[a] & ((1u << lvl) - 1u));
    RefCom acc{0, 0, 0, 0, false};
    if (c <= 64) {
        const bool have = lane < c;
        const uint32_t o = have ? orig[a + lane] : 0xffffffffu;
        double pm = 0, px = 0, py = 0, pz = 0;
        if (have) { const R4 p = posm[a + lane]; pm = (double)p.w; px = (double)p.x; py = (double)p.y; pz = (double)p.z; }
        int rank = 0;
        for (int k = 0; k < c; k++) rank += (__shfl(o, k, 64) < o) ? 1 : 0;
        // lane r takes the member of rank r, so lane order is upload order
        int from = 0;
        for (int k = 0; k < c; k++) { const int rk = __shfl(rank, k, 64); if (rk == lane) from = k; }
        pm = __shfl(pm, from, 64); px = __shfl(px, from, 64); py = __shfl(py, from, 64); pz = __shfl(pz, from, 64);
        ref_com_push_lanes(acc, c == 64 ? ~0ull : (1ull << c) - 1ull, pm, px, py, pz);
    } else {
        // REF_UNROLL chunks of 64 upload indices per round, so that their loads are in flight together (one round
        // is two dependent memory latencies: the index, then the hits' bodies)
        constexpr int REF_UNROLL = 8;
        for (int u0 = 0; u0 < n; u0 += 64 * REF_UNROLL) {
            uint32_t s[REF_UNROLL];
#pragma unroll
            for (int k = 0; k < REF_UNROLL; k++) {
                const int u = u0 + 64 * k + lane;
                s[k] = u < n ? inv[u] : 0xffffffffu;
            }
            bool hit[REF_UNROLL];
            double pm[REF_UNROLL], px[REF_UNROLL], py[REF_UNROLL], pz[REF_UNROLL];
#pragma unroll
            for (int k = 0; k < REF_UNROLL; k++) {
                hit[k] = s[k] >= (uint32_t)a && s[k] <= (uint32_t)b;
                // unconditional (lanes without a hit read the cell's first body and ignore it): a load under a
                // branch is waited for inside the branch, and the eight would run one after the other
                const R4 p = posm[hit[k] ? s[k] : (uint32_t)a];
                pm[k] = (double)p.w; px[k] = (double)p.x; py[k] = (double)p.y; pz[k] = (double)p.z;
            }
#pragma unroll
            for (int k = 0; k < REF_UNROLL; k++)
                ref_com_push_lanes(acc, (unsigned long long)__ballot(hit[k]), pm[k], px[k], py[k], pz[k]);
        }
    }
    if (lane == 0) {
        nodes[pos].x = acc.x; nodes[pos].y = acc.y; nodes[pos].z = acc.z; nodes[pos].m = acc.M;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
static int free_ptr(void** p)
{
    if (*p) { NBE_HIP(hipFree(*p)); *p = nullptr; }
    return NBE_OK;
}

// Up to 2n - 1 nodes plus the sentinel; the walk addresses fp32 node records with a 32-bit byte offset (32 B each).
#define TREE_MAX_BODIES (1ull << 26)

int tree_small_limit(void) { return SMALL_SORT_MAX; }

// How the cells of this engine are summed: fp64 node records have no fp64 prefix-difference mode (its cancellation error
// would show in them); their default is the exact integer sums, whose quantum lies far below fp64 rounding (1M bodies:
// build 0.17 ms against 0.29 ms child by child, one launch per depth).
int tree_sum_mode(const nbe_engine* e)
{
    if (e->cfg.cell_sums == NBE_SUMS_EXACT) return NBE_SUMS_EXACT;
    if (e->cfg.cell_sums == NBE_SUMS_TREE) return NBE_SUMS_TREE;
    return e->force64 ? NBE_SUMS_EXACT : NBE_SUMS_PREFIX;
}

// Quanta of the exact sums (NBE_SUMS_EXACT) from the run's total mass and the bound: every |sum of m| and |sum of m*x|
// stays below 2^126 quanta.
void tree_set_total_mass(nbe_engine* e, double total)
{
    double reach = 0.0;
    for (int c = 0; c < 3; c++) reach = fmax(reach, fabs(e->cfg.bound_center[c]) + 0.5 * e->cfg.bound_width);
    if (!(total > 0.0) || !std::isfinite(total)) total = 1.0;
    if (!(reach > 0.0) || !std::isfinite(reach)) reach = 1.0;
    int em = 0, ex = 0;
    (void)frexp(total, &em);                         // total < 2^em
    (void)frexp(total * reach, &ex);
    if (!std::isfinite(total * reach)) { int er = 0; (void)frexp(reach, &er); ex = em + er; }
    e->tw.q_shift_m = 126 - em;
    e->tw.q_shift_x = 126 - ex;
    e->tw.total_mass = total;
}

int tree_workspace_reserve(nbe_engine* e)
{
    TreeWork& w = e->tw;
    const size_t n = e->n;
    if (w.cap >= n && w.cap > 0) return NBE_OK;
    void** ptrs[] = {(void**)&w.keys[0], (void**)&w.keys[1], (void**)&w.sidx[0], (void**)&w.sidx[1], (void**)&w.block_hist,
                     (void**)&w.scan_tmp, (void**)&w.bt_first, (void**)&w.bt_last, (void**)&w.bt_parent, (void**)&w.bt_level,
                     (void**)&w.bt_cell, (void**)&w.lvlmask, (void**)&w.cellcnt, (void**)&w.nodes, (void**)&w.node_level, (void**)&w.node_sum,
                     (void**)&w.counters, (void**)&w.d_interactions, (void**)&w.prefix, (void**)&w.prefix_tiles,
                     (void**)&w.prefixq, (void**)&w.prefixq_tiles};
    for (void** p : ptrs) NBE_TRY(free_ptr(p));
    const size_t cap = n > 0 ? n : 1;
    const size_t nblocks = cdiv(cap, SORT_TILE);
    w.block_hist_len = 256 * nblocks;
    size_t scan_len = w.block_hist_len > cap + 1 ? w.block_hist_len : cap + 1;
    if (e->dist && scan_len < 2 * (2 * cap + 2) + 1) scan_len = 2 * (2 * cap + 2) + 1;       // dist.hip scans two words per node
    w.scan_tmp_len = cdiv(scan_len, SCAN_TILE) + 1;
    if (w.scan_tmp_len < 256) w.scan_tmp_len = 256;      // also holds the 256 digit totals of a radix pass
    const size_t node_bytes = e->force64 ? sizeof(NodeD) : sizeof(NodeF);
    NBE_HIP(hipMalloc((void**)&w.keys[0], cap * 8));
    NBE_HIP(hipMalloc((void**)&w.keys[1], cap * 8));
    NBE_HIP(hipMalloc((void**)&w.sidx[0], cap * 4));
    NBE_HIP(hipMalloc((void**)&w.sidx[1], cap * 4));
    NBE_HIP(hipMalloc((void**)&w.block_hist, w.block_hist_len * 4));
    NBE_HIP(hipMalloc((void**)&w.scan_tmp, w.scan_tmp_len * 4));
    NBE_HIP(hipMalloc((void**)&w.bt_first, cap * 4));
    NBE_HIP(hipMalloc((void**)&w.bt_last, cap * 4));
    NBE_HIP(hipMalloc((void**)&w.bt_parent, cap * 4));
    NBE_HIP(hipMalloc((void**)&w.bt_level, cap));
    NBE_HIP(hipMalloc((void**)&w.bt_cell, cap));
    NBE_HIP(hipMalloc((void**)&w.lvlmask, (cap + 1) * 4));
    NBE_HIP(hipMalloc((void**)&w.cellcnt, (cap + 1) * 4));
    NBE_HIP(hipMalloc((void**)&w.nodes, (2 * cap + 2) * node_bytes));   // the walk prefetches one record past the end
    NBE_HIP(hipMemsetAsync(w.nodes, 0, (2 * cap + 2) * node_bytes, e->stream));
    const int sums = tree_sum_mode(e);
    if (sums == NBE_SUMS_EXACT) {
        NBE_HIP(hipMalloc((void**)&w.prefixq, (cap + 1) * sizeof(Q4)));
        NBE_HIP(hipMalloc((void**)&w.prefixq_tiles, ((size_t)cdiv(cap + 1, PSUM_TILE) + 1) * sizeof(Q4)));
    } else if (sums == NBE_SUMS_TREE) {
        NBE_HIP(hipMalloc((void**)&w.node_level, 2 * cap));
        NBE_HIP(hipMalloc((void**)&w.node_sum, 2 * cap * 4 * sizeof(double)));
    } else {
        NBE_HIP(hipMalloc((void**)&w.prefix, (cap + 1) * 4 * sizeof(double)));
        NBE_HIP(hipMalloc((void**)&w.prefix_tiles, ((size_t)cdiv(cap + 1, PSUM_TILE) + 1) * 4 * sizeof(double)));
    }
    NBE_HIP(hipMalloc((void**)&w.counters, 16 * sizeof(int)));
    NBE_HIP(hipMalloc((void**)&w.d_interactions, 8 * sizeof(unsigned long long)));
    NBE_HIP(hipMemsetAsync(w.counters, 0, 16 * sizeof(int), e->stream));
    NBE_HIP(hipMemsetAsync(w.d_interactions, 0, sizeof(unsigned long long), e->stream));
    w.cap = cap;
    return NBE_OK;
}

// Key generation and the eight radix passes with workgroups of T threads (tiles of 8 T keys).
template <typename R4, int T>
static int radix_passes_t(nbe_engine* e, const R4* force_pos, int n, double lox, double loy, double loz, double W)
{
    TreeWork& w = e->tw;
    const int nblocks = (int)cdiv(n, T * SORT_KPT);
    const int by_xcd = nblocks >= 2048 * 256 / T;
    hipLaunchKernelGGL((keys_kernel<R4, T>), dim3(nblocks), dim3(T), 0, e->stream, force_pos, n, lox, loy, loz, W, w.keys[0],
                       nblocks, w.block_hist, by_xcd);
    NBE_HIP(hipGetLastError());
    int in = 0;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = pass * 8;
        if (pass > 0)       // the key kernel has counted the digits of the first pass
            hipLaunchKernelGGL((radix_hist_kernel<T>), dim3(nblocks), dim3(T), 0, e->stream, w.keys[in], n, shift, nblocks,
                               w.block_hist, by_xcd);
        hipLaunchKernelGGL(scan_rows_kernel, dim3(256), dim3(256), 0, e->stream, w.block_hist, nblocks, w.scan_tmp);
        hipLaunchKernelGGL((radix_scatter_kernel<T>), dim3(nblocks), dim3(T), 0, e->stream, w.keys[in],
                           pass > 0 ? w.sidx[in] : (const uint32_t*)nullptr, w.keys[in ^ 1], w.sidx[in ^ 1], w.block_hist,
                           w.scan_tmp, n, shift, nblocks, pass == 7 ? w.counters + COUNTER_SORTED_ALIVE : (int*)nullptr, by_xcd);
        NBE_HIP(hipGetLastError());
        in ^= 1;
    }
    return NBE_OK;
}

// Keys of the current state and their sorted order (w.keys[0], w.sidx[0], live count): everything of the sort that
// reads positions only.  The state itself is reordered afterwards, by sort_and_build_t.
template <typename R4>
static int sort_keys_t(nbe_engine* e)
{
    TreeWork& w = e->tw;
    const int n = (int)e->n;
    const int cur = e->cur;
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    const R4* force_pos = mixed ? (const R4*)e->mirror[cur] : (const R4*)e->posm[cur];
    const double W = e->cfg.bound_width;
    const double lox = e->cfg.bound_center[0] - 0.5 * W, loy = e->cfg.bound_center[1] - 0.5 * W,
                 loz = e->cfg.bound_center[2] - 0.5 * W;
    if (n <= e->small_max) {
        hipLaunchKernelGGL((small_sort_kernel<R4>), dim3(1), dim3(1024), 0, e->stream, force_pos, n, lox, loy, loz, W, w.keys[0],
                           w.sidx[0], w.counters);
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    }
    // 63-bit keys plus the all-ones dead key: 8 passes of 8 bits, ending back in buffer 0.
    // Workgroups of 512 threads, tiles of 4,096 keys: a digit's run in a tile is 16 keys on average instead of 8, half as many
    // rows for the scan.  Sort phase at 1M / 2M / 4M / 16M bodies: 0.187 / 0.262 / 0.413 / 1.76 ms against 0.196 / 0.290 /
    // 0.420 / 1.88 with 256 threads and 0.229 / 0.263 / 0.426 / 1.74 with 1,024 (one workgroup per CU then).
    // NBODY_SORT_THREADS = 256, 512 or 1024 picks one, for measurements.
    static const int threads_env = getenv("NBODY_SORT_THREADS") ? atoi(getenv("NBODY_SORT_THREADS")) : 0;
    if (threads_env == 256) return radix_passes_t<R4, 256>(e, force_pos, n, lox, loy, loz, W);
    if (threads_env == 1024) return radix_passes_t<R4, 1024>(e, force_pos, n, lox, loy, loz, W);
    return radix_passes_t<R4, 512>(e, force_pos, n, lox, loy, loz, W);
}

template <typename R4, typename S4, typename NodeT, int SUMS>
static int sort_and_build_t(nbe_engine* e)
{
    TreeWork& w = e->tw;
    const int n = (int)e->n;
    const int cur = e->cur, nxt = cur ^ 1;
    const unsigned nb = cdiv(n, 256);
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    const double W = e->cfg.bound_width;

    if (e->timing) NBE_HIP(hipEventRecord(e->ev[0], e->stream));
    if (!e->keys_sorted) NBE_TRY(sort_keys_t<R4>(e));        // unless tree_sort_keys ran ahead of this call
    e->keys_sorted = false;
    NBE_TRY(frame_stream_guard_write(e, nxt));
    hipLaunchKernelGGL((permute_kernel<S4>), dim3(cdiv((size_t)n + 1, 256)), dim3(256), 0, e->stream, w.sidx[0], w.keys[0], n, (const S4*)e->posm[cur],
                       (const S4*)e->vel[cur], e->orig[e->orig_cur], mixed ? e->mirror[cur] : (const float4*)nullptr,
                       (const double*)e->radius[cur], (S4*)e->posm[nxt], (S4*)e->vel[nxt], e->orig[e->orig_cur ^ 1],
                       mixed ? e->mirror[nxt] : (float4*)nullptr, e->radius[nxt], w.counters, w.lvlmask);
    NBE_HIP(hipGetLastError());
    e->cur = nxt;
    e->orig_cur ^= 1;
    e->have_counts = true;
    if (e->timing) NBE_HIP(hipEventRecord(e->ev[1], e->stream));

    const R4* sorted_pos = mixed ? (const R4*)e->mirror[nxt] : (const R4*)e->posm[nxt];
    hipLaunchKernelGGL(radix_tree_kernel, dim3(nb), dim3(256), 0, e->stream, w.keys[0], w.counters, w.bt_first, w.bt_last,
                       w.bt_parent, w.bt_level, w.bt_cell, w.lvlmask);
    NBE_HIP(hipGetLastError());
    NBE_TRY((device_exclusive_scan<true>(e, w.lvlmask, w.cellcnt, (size_t)n + 1)));
    if (SUMS == NBE_SUMS_EXACT) {
        const unsigned tiles = cdiv((size_t)n + 1, PSUM_TILE);
        if (n <= e->small_max) {
            hipLaunchKernelGGL((psumq_single_kernel<R4>), dim3(1), dim3(256), 0, e->stream, sorted_pos, n, w.counters, w.q_shift_m,
                               w.q_shift_x, (Q4*)w.prefixq);
        } else {
            hipLaunchKernelGGL((psumq_reduce_kernel<R4>), dim3(tiles), dim3(256), 0, e->stream, sorted_pos, w.counters, w.q_shift_m,
                               w.q_shift_x, (Q4*)w.prefixq_tiles);
            hipLaunchKernelGGL(psumq_scan_sums_kernel, dim3(1), dim3(256), 0, e->stream, (Q4*)w.prefixq_tiles, (int)tiles);
            hipLaunchKernelGGL((psumq_apply_kernel<R4>), dim3(tiles), dim3(256), 0, e->stream, sorted_pos, n, w.counters, w.q_shift_m,
                               w.q_shift_x, (const Q4*)w.prefixq_tiles, (Q4*)w.prefixq);
        }
        NBE_HIP(hipGetLastError());
    }
    if (SUMS == NBE_SUMS_PREFIX) {
        const unsigned tiles = cdiv((size_t)n + 1, PSUM_TILE);
        if (n <= e->small_max) {
            hipLaunchKernelGGL((psum_single_kernel<R4>), dim3(1), dim3(256), 0, e->stream, sorted_pos, n, (D4*)w.prefix);
        } else {
            hipLaunchKernelGGL((psum_reduce_kernel<R4>), dim3(tiles), dim3(256), 0, e->stream, sorted_pos, n, (D4*)w.prefix_tiles);
            hipLaunchKernelGGL(psum_scan_sums_kernel, dim3(1), dim3(256), 0, e->stream, (D4*)w.prefix_tiles, (int)tiles);
            hipLaunchKernelGGL((psum_apply_kernel<R4>), dim3(tiles), dim3(256), 0, e->stream, sorted_pos, n,
                               (const D4*)w.prefix_tiles, (D4*)w.prefix);
        }
        NBE_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((emit_nodes_kernel<R4, NodeT, SUMS>), dim3(2 * cdiv((size_t)n, 256)), dim3(256), 0, e->stream, n,
                       w.counters, sorted_pos, w.bt_first, w.bt_last, w.bt_cell, w.bt_level, w.lvlmask, w.cellcnt, W,
                       e->cfg.theta, e->cfg.eps * e->cfg.eps, (NodeT*)w.nodes, w.node_level, w.node_sum, (const D4*)w.prefix,
                       (const Q4*)w.prefixq, w.q_shift_m, w.q_shift_x, w.keys[0], w.bt_parent,
                       (e->dist && !e->cfg.collisions) ? e->cw.cnodes : (CNode*)nullptr);
    NBE_HIP(hipGetLastError());
    if (SUMS == NBE_SUMS_TREE) {
        for (int level = 21; level >= 0; level--) {
            hipLaunchKernelGGL((cell_mass_kernel<NodeT>), dim3(cdiv(2 * (size_t)n, 256)), dim3(256), 0, e->stream, w.counters,
                               level, (NodeT*)w.nodes, w.node_level, w.node_sum);
        }
        NBE_HIP(hipGetLastError());
    }
    if (e->cfg.reference_quirks & NBE_QUIRK_REF_COM) {
        // the sort's second index buffer is free between sorts: it holds the inverse of the sorted order here
        uint32_t* inv = w.sidx[1];
        NBE_HIP(hipMemsetAsync(inv, 0xff, (size_t)n * 4, e->stream));
        hipLaunchKernelGGL(ref_inv_kernel, dim3(nb), dim3(256), 0, e->stream, w.counters, e->orig[e->orig_cur], inv);
        hipLaunchKernelGGL((ref_com_kernel<R4, NodeT>), dim3(cdiv((size_t)n * 64, 256)), dim3(256), 0, e->stream, n, w.counters,
                           sorted_pos, e->orig[e->orig_cur], inv, w.bt_first, w.bt_last, w.bt_cell, w.bt_level, w.lvlmask,
                           w.cellcnt, (NodeT*)w.nodes);
        NBE_HIP(hipGetLastError());
    }
    if (e->cfg.collisions) NBE_TRY(collide_emit_nodes(e));
    if (e->timing) NBE_HIP(hipEventRecord(e->ev[2], e->stream));
    return NBE_OK;
}

int device_scan_u32(nbe_engine* e, const uint32_t* in, uint32_t* out, size_t len)
{
    return device_exclusive_scan<false>(e, in, out, len);
}

// The part of the sort that needs positions only, ahead of tree_sort_and_build: a host that still waits for other
// state arrays to arrive (velocities from other GPUs) can let this run meanwhile.
int tree_sort_keys(nbe_engine* e)
{
    NBE_TRY(tree_workspace_reserve(e));
    if (e->n == 0 || e->keys_sorted) return NBE_OK;
    if (e->n > TREE_MAX_BODIES) { nbe_set_error("tree mode supports at most 2^26 bodies"); return NBE_E_UNSUPPORTED; }
    int rc = e->cfg.precision == NBE_PREC_F64 ? sort_keys_t<double4>(e) : sort_keys_t<float4>(e);
    if (rc == NBE_OK) e->keys_sorted = true;
    return rc;
}

// Sorts the current state by Morton key into the other buffer (which becomes current) and builds the tree
// over it.  Everything is queued on the engine's stream; nothing is read back.
int tree_sort_and_build(nbe_engine* e)
{
    NBE_TRY(tree_workspace_reserve(e));
    if (e->n == 0) return NBE_OK;
    if (e->n > TREE_MAX_BODIES) { nbe_set_error("tree mode supports at most 2^26 bodies"); return NBE_E_UNSUPPORTED; }
    const int sums = tree_sum_mode(e);
    switch (e->cfg.precision) {
    case NBE_PREC_F32:
        return sums == NBE_SUMS_EXACT  ? sort_and_build_t<float4, float4, NodeF, NBE_SUMS_EXACT>(e)
               : sums == NBE_SUMS_TREE ? sort_and_build_t<float4, float4, NodeF, NBE_SUMS_TREE>(e)
                                       : sort_and_build_t<float4, float4, NodeF, NBE_SUMS_PREFIX>(e);
    case NBE_PREC_MIXED:
        return sums == NBE_SUMS_EXACT  ? sort_and_build_t<float4, double4, NodeF, NBE_SUMS_EXACT>(e)
               : sums == NBE_SUMS_TREE ? sort_and_build_t<float4, double4, NodeF, NBE_SUMS_TREE>(e)
                                       : sort_and_build_t<float4, double4, NodeF, NBE_SUMS_PREFIX>(e);
    case NBE_PREC_F64:
        return sums == NBE_SUMS_EXACT ? sort_and_build_t<double4, double4, NodeD, NBE_SUMS_EXACT>(e)
                                      : sort_and_build_t<double4, double4, NodeD, NBE_SUMS_TREE>(e);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}
