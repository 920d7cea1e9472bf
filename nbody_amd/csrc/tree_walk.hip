// Wavefront-coherent Barnes-Hut traversal with the kick-drift integrator fused into the epilogue.
//
// Replaces (*node).gravity (tree.go:202-240) and its fan-out over four goroutines (main.go:134-149).
// One lane owns one body; the 64 bodies of a wave are consecutive in Morton order, so they are close in
// space and want nearly the same nodes.  The wave walks the pre-order node array with ONE shared cursor:
//   * every lane applies the reference's own test to the node under the cursor: a branching cell is
//     opened iff edge / |com - p| > theta (tree.go:216-217; evaluated as r^2 < edge^2 / theta^2 in fp32);
//   * a lane that does not open it takes its force (tree.go:232-238) and remembers `skip`, the index after
//     the node's subtree; while the cursor is below that index the lane sits out (it has already accounted
//     for everything in there), exactly as its own recursion would have returned;
//   * the cursor descends (index+1) if any lane opened the node, else jumps to `skip`.
// So each body receives precisely the interactions its private depth-first walk would give it, in the
// same order (children 0..7, tree.go:221-225), while node records are fetched once per wave through a
// wave-uniform address.  Differences from the reference: the body's own leaf and bodies at distance 0
// contribute nothing (the reference skips only the body itself and yields NaN for a coincident pair);
// all bodies are walked (main.go:136-147 leaves out the last N mod 4).
#include "engine_internal.h"
#include "kernels_common.h"

// The node under the cursor is wave-uniform and should stay in scalar registers.  hipcc selects between two
// uniform *float* values with vector instructions (v_mov + v_cndmask per field), so the records are handled as
// integer bit patterns (scalar s_cselect) and turned into floats only where a lane consumes them.
struct RawF { unsigned x, y, z, m, w2; int skip; };
struct RawD { unsigned long long x, y, z, m, w, w2; int skip; };

__device__ __forceinline__ RawF load_raw(const NodeF* __restrict__ nodes, int i)
{
    // 32-bit byte offset from the array base: the scalar load takes it as an SGPR offset (no 64-bit adds)
    // one 32-byte, 32-byte-aligned request (s_load_dwordx8): the scalar cache sustains few misses in flight,
    // so requests per visit matter more than bytes
    typedef unsigned u8v __attribute__((ext_vector_type(8)));
    const char* base = reinterpret_cast<const char*>(nodes) + (size_t)((unsigned)i * 32u);
    const u8v a = *reinterpret_cast<const u8v*>(__builtin_assume_aligned(base, 32));
    return RawF{a[0], a[1], a[2], a[3], a[4], (int)a[5]};
}
__device__ __forceinline__ RawD load_raw(const NodeD* __restrict__ nodes, int i)
{
    // the whole 64-byte record, one cache line, in one request (s_load_dwordx16): requests per visit matter more than bytes
    typedef unsigned long long u8l __attribute__((ext_vector_type(8)));
    const u8l a = *reinterpret_cast<const u8l*>(__builtin_assume_aligned(nodes + i, 64));
    return RawD{a[0], a[1], a[2], a[3], a[4], a[6], (int)(unsigned)a[5]};
}
// The same request split in two: `request_raw` issues the load and returns at once, `arrived` waits for it.  The
// compiler schedules a plain load next to its first use, which in the walk is one whole visit too late (a plain
// load held in place with a scheduling barrier works too, but costs register copies: 2.05 ms against 1.97 ms).
// Between the two calls the destination registers are in flight and the compiler does not know it: the build
// checks the device assembly for any instruction that touches them in that window (check_pending_sgprs.py).
struct PendingF { unsigned v __attribute__((ext_vector_type(8))); };
struct PendingD { unsigned long long v __attribute__((ext_vector_type(8))); };     // the 64-byte fp64 record in flight
// `after` is passed through a second statement that the compiler may not move ahead of the request: whatever is
// computed from it stays behind the request in program order.
__device__ __forceinline__ PendingF request_raw(const NodeF* __restrict__ nodes, int i, float& after)
{
    PendingF p;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(p.v) : "s"(nodes), "s"((unsigned)i * 32u));
    asm volatile("" : "+v"(after));
    return p;
}
__device__ __forceinline__ RawF arrived(PendingF& p)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v));
    return RawF{p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], (int)p.v[5]};
}
__device__ __forceinline__ void settle(PendingF& p, PendingF& q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v), "+s"(q.v)); }
__device__ __forceinline__ void settle(PendingD& p, PendingD& q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v), "+s"(q.v)); }
__device__ __forceinline__ void settle(PendingF& p) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v)); }
__device__ __forceinline__ void settle(PendingD& p) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v)); }
// A second form of the wait, for the kernel in which the register allocator answered the first with copies ahead of the
// wait (two bodies per lane in mixed precision: it gave the value in flight and the value arrived different registers,
// and a tied operand is satisfied by a copy).  Here the record is a plain input that is never redefined, and the wait is
// tied to the bodies' coordinates instead: everything that consumes the record combines it with the coordinates or is
// decided by such a value, so it stays behind the wait.  (Used everywhere it measured 1-3 % slower in the fp32 kernels.)
template <typename P>
__device__ __forceinline__ RawF arrived_behind(PendingF& p, P& x, P& y, P& z)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(z) : "s"(p.v));
    return RawF{p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], (int)p.v[5]};
}
__device__ __forceinline__ void settle_plain(PendingF& p) { asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(p.v)); }
// (fp64 records: a plain load was waited for at the end of every second visit, where the compiler copied the next
// record's centre to vector registers for the first of its two unrolled visits; 64-bit address, because 2^27 records of
// 64 bytes do not fit a 32-bit byte offset)
__device__ __forceinline__ PendingD request_raw(const NodeD* __restrict__ nodes, int i, double& after)
{
    PendingD p;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(p.v) : "s"(nodes + i));
    asm volatile("" : "+v"(after));
    return p;
}
__device__ __forceinline__ RawD arrived(PendingD& p)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p.v));
    return RawD{p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], p.v[6], (int)(unsigned)p.v[5]};
}

// v[lane] = value for a wave-uniform value and lane number, both in scalar registers produced by scalar instructions (a
// lane number written by a vector instruction would need four wait states in front of this one).
__device__ __forceinline__ void write_lane(int& v, int value, int lane)
{
    // (one scalar register per vector instruction on this chip: the lane number goes through M0, which the compiler
    // reserves and sets itself in front of any instruction of its own that reads it; none in this file does, and the
    // build's check of the device assembly fails if that changes)
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(lane));
}

__device__ __forceinline__ RawF pick(bool c, const RawF& a, const RawF& b)
{
    return RawF{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.m : b.m, c ? a.w2 : b.w2, c ? a.skip : b.skip};
}
__device__ __forceinline__ RawD pick(bool c, const RawD& a, const RawD& b)
{
    return RawD{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.m : b.m, c ? a.w : b.w, c ? a.w2 : b.w2, c ? a.skip : b.skip};
}
__device__ __forceinline__ float as_real(unsigned v) { return __uint_as_float(v); }
__device__ __forceinline__ double as_real(unsigned long long v) { return __longlong_as_double((long long)v); }

// Opening test of tree.go:216-217, "edge / r > theta".  fp32 nodes store edge^2 / theta^2, so the test is one
// compare, r^2 < edge^2/theta^2; fp64 nodes store the edge and repeat the reference's expression literally.
// A leaf stores 0 and can never open.
__device__ __forceinline__ bool node_opens(const RawF& nd, float r2, float) { return r2 < as_real(nd.w2); }
// The same with the wave's vote (the mask of the lanes that open the node) beside the lane's own verdict.
__device__ __forceinline__ bool node_opens(const RawF& nd, float r2, float, unsigned long long& opening)
{
    const bool open = r2 < as_real(nd.w2);
    opening = __builtin_amdgcn_ballot_w64(open);
    return open;
}
// Whether two non-negative doubles may be within rounding of each other, erring on the side of yes: their upper words
// (sign, exponent, top 20 mantissa bits) differ by at most one, i.e. they agree to about 1e-6 or better.  Three 32-bit
// instructions; the exact distance of the bit patterns would take five, and the wider net costs nothing: about one
// wave-visit in ten thousand has a lane in it.
__device__ __forceinline__ bool close_products(double a, double b)
{
    const unsigned ha = (unsigned)((unsigned long long)__double_as_longlong(a) >> 32);
    const unsigned hb = (unsigned)((unsigned long long)__double_as_longlong(b) >> 32);
    return ha - hb + 1u <= 2u;
}

// The fp64 expression costs a square root and a division (about half of the visit's instructions), and its verdict
// is that of  (edge / theta)^2 > r^2  unless the two sides agree to within the rounding of either form.  So the
// squares decide (the left one is stored in the record), and only when some lane of the wave finds them close
// (close_products: a net thousands of times wider than the combined rounding error of both forms, and still rarely
// hit) does the wave evaluate the reference's expression.  A lane that sits out carries r^2 = 1e300: far from any finite
// (edge / theta)^2, so the verdict "closed" is certain; r^2 = 0 against a leaf's edge 0 is close
// and goes the reference's way (0 / 0 > theta is false).
__device__ __forceinline__ bool node_opens(const RawD& nd, double r2, double theta)
{
    const double w = as_real(nd.w);
    const double a = as_real(nd.w2), b = r2;
    bool open = a > b;
    const bool unsure = close_products(a, b);
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) open = w / sqrt(r2) > theta;
    return open;
}
// (the vote is taken from each compare where it stands: the ballot of the verdict after the two ways have joined goes
// through a vector register and back)
__device__ __forceinline__ bool node_opens(const RawD& nd, double r2, double theta, unsigned long long& opening)
{
    const double w = as_real(nd.w);
    const double a = as_real(nd.w2), b = r2;
    bool open = a > b;
    opening = __builtin_amdgcn_ballot_w64(a > b);
    const bool unsure = close_products(a, b);
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
        const double q = w / sqrt(r2);
        open = q > theta;
        opening = __builtin_amdgcn_ballot_w64(q > theta);
    }
    return open;
}

// Mixed precision: where a body's fp32 partial sums join its fp64 totals.  After the term of every node that the body
// takes and whose record hashes to 0 in six bits: one node in 64, picked by what the node is (its centre), not by
// where it lies in an array or by what the wavefront has visited so far.  A body's sums are then a function of its own
// list of accepted nodes alone, so its result has the same bits whichever bodies share its wavefront: one or two bodies
// per lane, any shard boundary, a rank of the distributed tree.  (A count of the wavefront's visits, the earlier rule,
// cut the sums where the neighbours' walks happened to stand.)  Scalar: the record is in scalar registers.
__device__ __forceinline__ bool folds_here(const RawF& nd) { return (((nd.x ^ nd.y ^ nd.z) * 0x9E3779B1u) >> 26) == 0u; }
__device__ __forceinline__ bool folds_here(const RawD&) { return false; }     // (fp64 records are never walked in mixed precision)

template <bool V> struct Flag { static constexpr bool value = V; };
template <int V> struct Size { static constexpr int value = V; };

#ifdef WALK_TIMELINE
// Measurement build (make timeline; profiles/walk_timeline.py): every wavefront of walk_kernel and walk_pair_kernel
// leaves when it began and ended (constant 100 MHz clock) and where it ran, for a picture of how a launch fills and
// drains the GPU.  Not part of the product build.
#define WALK_TIMELINE_WAVES (1 << 18)
__device__ unsigned long long walk_timeline[WALK_TIMELINE_WAVES][4];
extern "C" int nbe_debug_walk_timeline(unsigned long long* out, size_t waves)
{
    if (waves > WALK_TIMELINE_WAVES) waves = WALK_TIMELINE_WAVES;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(walk_timeline), waves * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
__device__ __forceinline__ void timeline_note(unsigned wave_index, unsigned long long began)
{
    if ((threadIdx.x & 63) == 0 && wave_index < WALK_TIMELINE_WAVES) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        walk_timeline[wave_index][0] = began;
        walk_timeline[wave_index][1] = wall_clock64();
        walk_timeline[wave_index][2] = ((unsigned long long)xcc << 32) | hw;
        walk_timeline[wave_index][3] = 0;
    }
}
#endif

#ifndef WALK_BLOCK
#define WALK_BLOCK 256
#endif

// Which workgroup's bodies this workgroup walks.  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB
// L2.  xcd_shift = 0: the workgroups of one XCD own one contiguous eighth of the launch's Morton-ordered bodies, so their
// walks share the same near-field nodes in that L2 instead of every XCD streaming the whole tree (bijective for any grid
// size).  That is right when the eighths hold alike work, as those of a whole scene do by symmetry; the bodies of one rank
// of a distributed tree run from the outskirts of an octant to the dense centre, and the XCD with the last eighth would
// finish long after the others.  xcd_shift = s > 0 deals chunks of 2^s consecutive workgroups to the XCDs in turn instead
// (the grid is a multiple of 8 * 2^s then; workgroups past the bodies find nothing to do).  Only speed depends on it.
// (Handing the chunks out alternately from the front and the back of the range, so that neither end's long walks come
// last, was slower: 3.88 against 3.62 ms for a rank's 2.1M targets; chunks that run together want to be neighbours.)
__device__ __forceinline__ unsigned walk_workgroup(unsigned xcd_shift)
{
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8, k = blockIdx.x / 8;
    if (xcd_shift == 0) {
        const unsigned q8 = nwg / 8, r8 = nwg % 8;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    return ((k >> xcd_shift) << (xcd_shift + 3)) + (xcd << xcd_shift) + (k & ((1u << xcd_shift) - 1u));
}


// COUNT: instrumented pass: also counts the force evaluations (tree.go:232-238 executions) of the wave and sums
// each body's potential (sum of mass / softened distance over the same accepted nodes, returned in the w
// component of the acceleration record, without the factor -G); the production step runs without it.
// SOFT (fp64 node records only): Plummer softening is on, so the opening test must undo the eps^2 folded into r^2 (with eps = 0 the
// floor is 1e-12 m^2, far below the resolution of any cell edge, and the test uses r^2 as is).
// NS: independent traversals (groups of 64 bodies) interleaved in one wave, each with its own cursor, node
// registers and sit-out index; a group that finishes first parks on the sentinel record behind the last node
// (mass 0, skip = itself).  Measured at 1M / 4M bodies: NS = 1: 2.21 / 8.53 ms, NS = 2: 2.31 / 8.21 ms,
// NS = 3: 3.0 / 9.5 ms, so NS = 1 ships.
//
// What bounds the kernel (rocprofv3 counters and an instrumented build; 1M-body Plummer sphere, theta = 0.5):
// 6.2e7 node visits per pass (3,780 per wave, standard deviation 14 %), each 17 vector + 7 scalar instructions and
// one 32-byte scalar load.  Most vector instructions read the node from scalar registers, which halves their
// rate: about 58 issue cycles per visit against 78 cycles measured per visit and SIMD, so the vector ALU is busy
// three quarters of the time; the scalar cache serves only 29 % of the node loads (L2 serves 97 % of the rest
// after the XCD remap), and eight waves per SIMD hide most of that latency.  Consequently neither the workgroup
// size (64 .. 1024 threads: 2.16 .. 2.31 ms then; with today's shorter visit workgroups of one wavefront are worth 3 %
// to launches that fill the GPU more than once, see launch_walk) nor a second traversal per wave nor prefetching both
// successors of a node helps, and every scalar instruction removed from the loop does (five fewer: -4 %).
// The visits are not equally useful: in 84 % nobody opens the node, in 5 % nobody takes it, and in 27 % at most 8
// of the 64 lanes are awake (deep in the near field of a few bodies while the rest of the wave sits out).
// Tried: handing subtrees that at most 16 lanes open to those lanes for private walks with vector loads after
// the shared walk.  It removes 31 % of the shared visits, but a private step (two vector loads with up to 64
// different addresses) costs three shared visits and the remaining shared visits are the cache-unfriendly ones:
// 2.04 ms against 2.07 ms, not worth its complexity.  A scalar branch that spares leaves (which cannot open) the
// test, the vote and the sit-out update saves six vector instructions on paper; the compiler pays them back in
// register copies and mask bookkeeping at the join (2.21 ms with both arms written out, 2.49 ms with a shared tail).
// Keeping the sit-out state as wave masks in scalar registers (sleeping lanes' masks on a small stack, pushed when
// a cell is opened by some lanes and accepted by others) takes the vector instructions per visit from 17 to 13 and
// adds nine scalar ones: 2.15 ms, no change (2.8 ms when combined with the early request described below: its
// pushes and pops are branches inside the visit).  What ships now is the cheapest form of that idea, see "sleeping
// lanes" below: 12 vector instructions per visit instead of 17.5, 1.78 / 3.49 / 31.7 ms at 1M / 2M / 16M bodies
// against 1.87 / 3.67 / 32.6; it paid only after the loop's scalar instructions had been cut from 16.5 to 11.5 per
// visit (the wait for the last request behind the loop: no gain by itself either; with 17 scalar instructions the
// 13-instruction visit took 1.90 ms): vector and scalar instructions have to fall together.
// Two bodies per lane behind one shared cursor (128 bodies per wave,
// 0.58 times the visits, twice the arithmetic per visit): 2.57 ms.  Fetching the node with a vector load of eight
// lanes and v_readlane instead of the scalar load: 2.9 ms.  The time scales with the waves resident per SIMD
// (1, 2, 4, 8 waves: 10.2, 5.7, 3.2, 2.16 ms; the kernel as it is now: 1, 2, 4, 6, 8 waves: 8.06, 4.44, 2.61, 2.07,
// 1.77 ms), i.e. one visit is a dependent chain of about 390 (now 306) cycles (scalar load, arithmetic, vote, branch)
// and eight waves, the hardware's limit, overlap it to about 80 (now 67) cycles per visit and SIMD.
// Shortening that chain is what helps: requesting the next node right after the vote, so that the force arithmetic
// runs under the load, gives 1.97 ms (2.15 ms with the load at the top of the visit).  After that the visit is
// bound by instruction issue, where a vector instruction costs twice as much when it reads a scalar register:
// moving the loop's constant to a vector register and not letting the compiler pair the two products (which
// needs the mass copied to a vector register) gives 1.84 ms.  Testing for the end once per two visits lets the
// compiler interleave the two and wait too early (1.87 ms); taking the vote from the plain compare and masking
// sleeping lanes with scalar logic instead of the select makes it rebuild the masks through vector registers.
// A launch ends with a drain of about one wavefront's lifetime in which the GPU empties (~0.5 ms lost per launch).
// Dispatching every XCD's workgroups longest-first by the time they took in the previous step, so that the quick ones
// drain, was measured (the timing itself is free): 2.44 ms against 2.20 ms per step at 1M bodies, 4.62 against 4.14 at
// 2M, 45.1 against 36.0 at 16M.  Workgroups that are neighbours in Morton order and run at the same time share their
// near-field nodes in the XCD's L2 and the scalar cache; any other order costs more than the drain.
// (Measured again with every wavefront's begin and end on record, `make timeline`: at 1M bodies the last wavefront begins
// after 63 % of the launch, 78 % of the wave slots are occupied over all of it, and list scheduling of the recorded
// lifetimes promises 13 % for longest first; the launch itself took 1.93 ms longest first against 1.74 ms, 28.8 against
// 25.3 at 16M.  The model has no caches.)
// Cells with nothing but bodies below them hold a quarter of the visits, and a body's record can never open: taking such
// a cell's bodies in a loop of their own (12 vector instructions each instead of 17: no test, no vote, no sit-out update;
// the cursor passes over the subtree, the descend target stored in the record so that the request for the next node
// stays one select away) gave the same bits at 2.14 ms against 1.89 ms (2.43 ms with the compiler's own loads, which it
// turns into vector loads): the loop's records are waited for where the visit's are requested a whole visit ahead.
template <typename R, typename A, typename NodeT, int OUT, bool COUNT, bool SOFT, int WB>
__global__ __launch_bounds__(WB) void walk_kernel(
    const NodeT* __restrict__ nodes, const int* __restrict__ counters, const typename V4<R>::type* __restrict__ posm,
    int t0, int t1, R theta, R floor2, A G, A dt, typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out, unsigned long long* __restrict__ interactions,
    const uint32_t* __restrict__ tail_orig, unsigned tail_limit, unsigned xcd_shift, unsigned sparse)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    constexpr bool BLOCKED = !__is_same(R, A);
    const int na = counters[0];
    const int total = counters[1];
    const int tend = t1 < na ? t1 : na;
    const unsigned wg = walk_workgroup(xcd_shift);
#ifdef WALK_TIMELINE
    const unsigned long long began = wall_clock64();
#endif
    // sparse = s > 0 (few targets, see launch_walk): only the first 64 >> s lanes of a wavefront hold a body, so that the
    // targets spread over more wavefronts with shorter walks; the other lanes sit out like lanes past the last body
    const int per_wave = 64 >> sparse, lane_in_wave = (int)(threadIdx.x & 63);
    const int i = lane_in_wave < per_wave ? t0 + (int)((wg * (WB / 64) + (threadIdx.x >> 6)) * (unsigned)per_wave) + lane_in_wave : 0x7fffffff;
    if (OUT == OUT_FUSED) {
        // bodies past n_alive (removed by the bound check) move to the next buffer unchanged
        if (i >= tend && i < t1) {
            pos_out[i] = pos_in[i];
            vel_out[i] = vel_in[i];
            if (mirror_out) { const A4 p = pos_in[i]; mirror_out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w); }
        }
    }
    const bool valid = i < tend;
    if (__ballot(valid) == 0 || total <= 0) return;

    R px = 0, py = 0, pz = 0, fx = 0, fy = 0, fz = 0;
    if (valid) { const R4 p = posm[i]; px = p.x; py = p.y; pz = p.z; }
    A ax = 0, ay = 0, az = 0, phi = 0;
    __shared__ A tot[BLOCKED ? 3 : 1][BLOCKED ? WB : 1];
    if constexpr (BLOCKED) { tot[0][threadIdx.x] = 0; tot[1][threadIdx.x] = 0; tot[2][threadIdx.x] = 0; }
    // tail_orig (NBE_QUIRK_DROP_TAIL): bodies whose upload index is tail_limit or more are not walked, like the
    // last n mod 4 bodies of main.go:134-148; they sit out the whole traversal and are integrated with zero force
    const bool walked = valid && !(tail_orig && tail_orig[i] >= tail_limit);
    int my_skip = walked ? 0 : 0x7fffffff;        // the lane sleeps while cursor < my_skip
    // a vector instruction that reads a scalar register issues at half rate: keep the loop's constant in a vector one
    R floor2v = floor2;
    asm volatile("" : "+v"(floor2v));
    // Sleeping lanes.  A lane that sits out carries +infinity (fp64: 1e300, see FAR) where the others carry the floor under
    // r^2: its distance to every node is infinite, so it opens nothing and receives nothing (rsq(inf) = 0), without a
    // test of its own in the visit.  Lanes fall asleep only where the cursor descends into a cell that they accept, and wake where the
    // cursor leaves that cell; both are events of the whole wave, decided on scalar values, and rare (about a sixth of
    // the visits each).  The ends of the cells in which lanes sleep are nested: the innermost in next_wake, the ones
    // around it on a stack kept in the lanes of one vector register (at most one entry per depth of the tree).
    // Mixed precision keeps the per-lane test of the sit-out index in every visit (17 vector instructions): there the
    // events made the kernel slower at 16M bodies (two galaxies: 37.6 against 36.0 ms), and without the explicit mask on
    // the weight the compiler copies record registers while a request is in flight.
    constexpr bool EVENTS = !BLOCKED;
    // "Infinitely far": what a lane that takes nothing carries as r^2.  fp32: +infinity (v_rsq_f32 makes 0 of it).  fp64:
    // 1e300, because the Newton step behind v_rsq_f64 makes NaN of infinity; the weight of 1e300, 1e-450 times the mass,
    // underflows to exactly 0, and no threshold (edge / theta)^2 comes near it (theta = 0 stores infinity, but then no lane
    // ever accepts a cell, so none sleeps).  Anything at or beyond FAR_TEST counts as far.
    constexpr R FAR = __is_same(R, double) ? (R)1e300 : (R)INFINITY;
    constexpr R FAR_TEST = __is_same(R, double) ? (R)1e299 : (R)INFINITY;
    R seed = (walked || !EVENTS) ? floor2v : FAR;
    int next_wake = 0x7fffffff, sp = 0, ends = 0;
    unsigned count = 0;
    [[maybe_unused]] int it = 0;                  // (visits of the wave: read by the WALK_VISIT_STATS build only)
    // The cursor and the node under it live in scalar registers: one scalar load per visit.  A visit is a chain of
    // dependent steps (load, distance, test, vote, next cursor), so the next node is requested as soon as the vote
    // is in and the force arithmetic of the current node runs while that load is in flight; the record behind the
    // last node (mass 0, skip = itself) makes the request past the end harmless.
    // (Fetching both possible successors ahead of the vote was slower: the scalar unit is shared by the CU's
    // four SIMDs, and the extra loads and selects cost more than the latency they hid.)
    int cur = 0;
    // One visit: consumes the record that `here` delivers, requests the successor into `there`.
    auto visit = [&](auto& here, auto& there, auto look_for_end) -> bool {
        const auto nd = arrived(here);
        const int skip = nd.skip;
        // (the compare must not move up in front of the wait for the record: on the way out of the loop it would share
        // its registers with a request that is still in flight)
        asm volatile("" : "+s"(cur));
        // (waking the lanes for the next node at the end of the visit, under the request in flight, was slower: 1.82
        // against 1.78 ms at 1M bodies; so was a cursor that counts bytes, which takes the shift out of the chain from
        // the vote to the request: 1.80 / 3.61 / 32.3 ms against 1.79 / 3.48 / 31.8 at 1M / 2M / 16M)
        if (EVENTS && cur == next_wake) {         // the cursor leaves the innermost cell in which lanes sleep
            if (my_skip <= cur) seed = floor2v;
            --sp;
            next_wake = __builtin_amdgcn_readlane(ends, sp);
        }
        const R dx = as_real(nd.x) - px, dy = as_real(nd.y) - py, dz = as_real(nd.z) - pz;
        // The floor under r^2 (1e-12 m^2 unless softening is on) keeps rsq finite for the body's own leaf
        // and for bodies at distance 0, which then contribute exactly 0 because dx = dy = dz = 0.
        R r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, seed)));
        // sleeping lanes have an infinite distance, so one compare is both the opening test and the vote
        const bool active = EVENTS || cur >= my_skip;
        const R rtest = active ? (SOFT ? r2 - floor2v : r2) : FAR;
        unsigned long long opening;
        const bool open = node_opens(nd, rtest, theta, opening);
        int next;
        if constexpr (__is_same(NodeT, NodeD)) {
            // (written out: on the way from the fp64 vote to the request the compiler turns the verdict into a mask and back)
            asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(next) : "s"(opening), "s"(cur + 1), "s"(skip) : "scc");
        } else {
            next = opening != 0 ? cur + 1 : skip;
        }
        there = request_raw(nodes, next, r2);      // the force arithmetic below runs while this is in flight
        const R mass = as_real(nd.m);
        bool counted = false;
        // (cur < total: mixed precision may visit the record behind the last node, which is no interaction)
        if (COUNT) counted = active && !open && r2 < FAR_TEST && (dx != (R)0 || dy != (R)0 || dz != (R)0) && (!BLOCKED || cur < total);
        const bool take = active && !open;
        if (!EVENTS && take) my_skip = skip;
        if (EVENTS && opening != 0) {
            // the cursor descends: the lanes that accept the cell sleep until the cursor reaches its end
            // (fp32: "!=" and not "<", so that the compiler forms one compare for the vote and for the select, not two)
            const bool fin = __is_same(R, float) ? r2 != FAR : r2 < FAR_TEST;
            const bool rest = !open && fin;
            // (the vote on "somebody rests" from the masks of the plain compares: the ballot of a combined condition goes
            // through a vector register and back)
            if ((__builtin_amdgcn_ballot_w64(fin) & ~opening) != 0) {
                if (rest) { my_skip = skip; seed = FAR; }
                // push the end of the cell around this one (a last child ends where its parent ends: one wake for both,
                // nothing pushed; the write below is then overwritten by the next push)
                write_lane(ends, next_wake, sp);
                sp += skip != next_wake;
                next_wake = skip;
            }
            // a lane that opens the cell takes nothing from it: an infinite distance gives it the weight 0
            if (open) r2 = FAR;
        }
        // (fp64 works with 2/r, rsq2_dev: one instruction less, the sums carry a factor of exactly 8 that G takes back)
        R rinv;
        if constexpr (__is_same(R, double)) rinv = rsq2_dev(r2); else rinv = rsq_dev(r2);
        R rinv2 = rinv * rinv;
        // (keeps the compiler from pairing the two products in one packed multiply, for which it copies the mass
        // to a vector register: one more instruction that reads a scalar register)
        if (__is_same(R, float)) asm volatile("" : "+v"(rinv2));
        const bool none = !EVENTS && !take;      // (with events a lane that takes nothing has an infinite distance)
        const R w = none ? (R)0 : rinv2 * (rinv * mass);
        fx = fma(dx, w, fx);
        fy = fma(dy, w, fy);
        fz = fma(dz, w, fz);
        if (COUNT) {
            const bool use = counted;
            count += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(use));
            if (use) phi += (A)(rinv * mass);
        }
        ++it;
        cur = next;
        if constexpr (!BLOCKED) return next < total; else {
        // Mixed precision.
        if (__builtin_expect(folds_here(nd), 0)) {
            // The fp32 partial sums of the bodies that take this node join their fp64 totals (folds_here).  The totals
            // live in LDS, not in registers: as registers the compiler gave them new ones in this rarely taken branch
            // and copied them back on every visit (six 64-bit moves, +18 % time).
            // (selects, not a branch on `take`: the compiler joins a per-lane condition with the node's into one mask
            // and then sets and restores the execution mask in every visit, +19 % time)
            tot[0][threadIdx.x] += take ? (A)fx : (A)0; tot[1][threadIdx.x] += take ? (A)fy : (A)0; tot[2][threadIdx.x] += take ? (A)fz : (A)0;
            fx = take ? (R)0 : fx; fy = take ? (R)0 : fy; fz = take ? (R)0 : fz;
        }
        // The end of the walk is looked for in every second visit only (the fold test took the place of a test of the
        // visit count that the compiler had put into every second visit).  The visit in between may then fall on the
        // record behind the last node: massless, never opened, its successor is itself; folding there changes no total.
        // (the wait on the way out of the visit: with the wait behind the loop the compiler copies the record registers
        // around the fold while a request is in flight, which the build's check of the assembly rejects)
        if constexpr (decltype(look_for_end)::value) {
            if (next >= total) { settle(there); return false; }
        }
        return true;
        }
    };
    R none = 0;
    auto slot_a = request_raw(nodes, 0, none);
    decltype(slot_a) slot_b;
    for (;;) {                                    // two record slots used in turn, so nothing is copied
        if (!visit(slot_a, slot_b, Flag<!BLOCKED>{})) break;
        if (!visit(slot_b, slot_a, Flag<true>{})) break;
    }
    // The record requested last is never read.  It still has to land before its registers may serve anything else:
    // waiting here, with both slots as operands, keeps them reserved on the way out of the loop.
    if constexpr (!BLOCKED) settle(slot_a, slot_b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing is in flight here; states it on every path out of the loop

#ifdef WALK_TIMELINE
    // (the visit count is not noted: kept alive past the loop it makes the compiler copy a record that is in flight)
    timeline_note(wg * (WB / 64) + (threadIdx.x >> 6), began);
#endif
    if (COUNT && (threadIdx.x & 63) == 0 && interactions) {
        atomicAdd(interactions, (unsigned long long)count);
#ifdef WALK_VISIT_STATS
        atomicAdd(interactions + 1, (unsigned long long)it);
        atomicMax(interactions + 2, (unsigned long long)it);
        atomicAdd(interactions + 3, (unsigned long long)it * (unsigned long long)it);
        atomicAdd(interactions + 4 + (it >= 3000) + (it >= 4000) + (it >= 6000), 1ull);
#endif
    }
    if (!valid) return;
    if constexpr (BLOCKED) { ax = tot[0][threadIdx.x]; ay = tot[1][threadIdx.x]; az = tot[2][threadIdx.x]; }
    const A sx = ax + (A)fx, sy = ay + (A)fy, sz = az + (A)fz;
    const A Gs = __is_same(R, double) ? G * (A)0.125 : G;         // (fp64 sums: 8 times the plain ones, see the visit)
    if (OUT == OUT_ACC) {
        A4 o; o.x = Gs * sx; o.y = Gs * sy; o.z = Gs * sz; o.w = __is_same(R, double) ? phi * (A)0.5 : phi;
        acc_out[i] = o;
    } else {
        kick_drift<A>(i, Gs * sx, Gs * sy, Gs * sz, dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
    }
}

// ---- per-wave criterion, two bodies per lane ----------------------------------------------------------------
// NBE_WALK_PER_WAVE.  The 128 Morton-consecutive bodies of a wavefront (lane l owns bodies base + l and
// base + 64 + l) decide together: a cell is opened for all of them as soon as one of them would open it
// (tree.go:216-217 evaluated for each, decided for all), and accepted only when all accept it.  Every body then
// opens at least the cells its own walk would open, nobody ever sits out, and the selects, compares and the copy
// that keep the per-body bookkeeping of walk_kernel are gone: a visit is 3 packed subtractions, 3 packed fma and two
// compares, plus, when the node is accepted, 2 rsq, 3 packed products and 3 packed fma for the two bodies.  Two bodies
// per lane halve the node records fetched per body and give every fetch twice the arithmetic to hide under.
// Not the reference's interaction set (finer), so it is off by default.
template <typename R, typename A, typename NodeT, int OUT, bool COUNT, bool SOFT>
__global__ __launch_bounds__(WALK_BLOCK) void walk_group_kernel(
    const NodeT* __restrict__ nodes, const int* __restrict__ counters, const typename V4<R>::type* __restrict__ posm,
    int t0, int t1, R theta, R floor2, A G, A dt, typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out, unsigned long long* __restrict__ interactions)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    typedef R R2 __attribute__((ext_vector_type(2)));
    constexpr bool BLOCKED = !__is_same(R, A);
    const int na = counters[0];
    const int total = counters[1];
    const int tend = t1 < na ? t1 : na;
    // same XCD-aware placement as walk_kernel: the workgroups of one XCD own a contiguous eighth of the bodies
    const unsigned nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    const unsigned wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx[2];
    // groups are the aligned 128-body chunks of the sorted order whatever range [t0, t1) this launch covers, so a
    // sharded run forms the same groups as one engine
    idx[0] = (t0 & ~127) + (int)wg * (2 * WALK_BLOCK) + wave * 128 + lane;
    idx[1] = idx[0] + 64;
    bool valid[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = idx[k];
        if (OUT == OUT_FUSED && i >= t0 && i >= tend && i < t1) {     // bodies removed by the bound check move on unchanged
            pos_out[i] = pos_in[i];
            vel_out[i] = vel_in[i];
            if (mirror_out) { const A4 p = pos_in[i]; mirror_out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w); }
        }
        valid[k] = i >= t0 && i < tend;
    }
    if (__ballot(valid[0] || valid[1]) == 0 || total <= 0) return;

    // a lane without a body stands infinitely far away: it never asks to open a cell, and what it sums is dropped
    R2 px = {(R)INFINITY, (R)INFINITY}, py = {0, 0}, pz = {0, 0};
    // every live body of the chunk takes part in the decisions, also one that belongs to another shard: the groups
    // and their votes are then the same however the bodies are sharded; only the own bodies are written
    if (idx[0] < na) { const R4 p = posm[idx[0]]; px.x = p.x; py.x = p.y; pz.x = p.z; }
    if (idx[1] < na) { const R4 p = posm[idx[1]]; px.y = p.x; py.y = p.y; pz.y = p.z; }
    R2 fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
    A ax[2] = {0, 0}, ay[2] = {0, 0}, az[2] = {0, 0}, phi[2] = {0, 0};
    __shared__ A tot[BLOCKED ? 6 : 1][BLOCKED ? WALK_BLOCK : 1];
    if constexpr (BLOCKED) {
#pragma unroll
        for (int k = 0; k < 6; k++) tot[k][threadIdx.x] = 0;
    }
    R2 floor2v = {floor2, floor2};
    asm volatile("" : "+v"(floor2v));
    // Everything the epilogue needs from the kernel arguments is fetched here, ahead of the loop: a scalar load that
    // the compiler places at the loop's exit could be given the registers of the node request that is still in flight
    // there (the build's check of the device assembly rejects exactly that).
    asm volatile("" :: "s"(G), "s"(dt), "s"(acc_out), "s"(interactions));
    unsigned count = 0;
    int it = 0;
    int cur = 0;
    auto visit = [&](auto& here, auto& there) -> bool {
        const auto nd = arrived(here);
        const int skip = nd.skip;
        const R nx = as_real(nd.x), ny = as_real(nd.y), nz = as_real(nd.z);
        const R2 dx = R2{nx, nx} - px, dy = R2{ny, ny} - py, dz = R2{nz, nz} - pz;
        R2 r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, floor2v)));
        const R2 rt = SOFT ? r2 - floor2v : r2;
        const bool open = node_opens(nd, rt.x, theta) || node_opens(nd, rt.y, theta);
        const bool any_open = __builtin_amdgcn_ballot_w64(open) != 0;
        const int next = any_open ? cur + 1 : skip;
        R keep = r2.x;
        there = request_raw(nodes, next, keep);
        r2.x = keep;
        if (!any_open) {                               // wave-uniform: all 128 bodies take the node
            const R mass = as_real(nd.m);
            const R2 rinv = {rsq_dev(r2.x), rsq_dev(r2.y)};
            const R2 w = (rinv * rinv) * (rinv * R2{mass, mass});
            fx = __builtin_elementwise_fma(dx, w, fx);
            fy = __builtin_elementwise_fma(dy, w, fy);
            fz = __builtin_elementwise_fma(dz, w, fz);
            if (COUNT) {
                const bool u0 = valid[0] && (dx.x != (R)0 || dy.x != (R)0 || dz.x != (R)0);
                const bool u1 = valid[1] && (dx.y != (R)0 || dy.y != (R)0 || dz.y != (R)0);
                count += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(u0)) + (unsigned)__popcll(__builtin_amdgcn_ballot_w64(u1));
                if (u0) phi[0] += (A)(rinv.x * mass);
                if (u1) phi[1] += (A)(rinv.y * mass);
            }
        }
        ++it;
        if constexpr (BLOCKED) if ((it & 63) == 0) {  // totals in LDS, as in walk_kernel
            tot[0][threadIdx.x] += (A)fx.x; tot[1][threadIdx.x] += (A)fy.x; tot[2][threadIdx.x] += (A)fz.x;
            tot[3][threadIdx.x] += (A)fx.y; tot[4][threadIdx.x] += (A)fy.y; tot[5][threadIdx.x] += (A)fz.y;
            fx = R2{0, 0}; fy = R2{0, 0}; fz = R2{0, 0};
        }
        cur = next;
        if (next < total) return true;
        settle(there);       // the record requested last is never read, but has to land before its registers are reused
        return false;
    };
    R none = 0;
    auto slot_a = request_raw(nodes, 0, none);
    decltype(slot_a) slot_b;
    for (;;) {
        if (!visit(slot_a, slot_b)) break;
        if (!visit(slot_b, slot_a)) break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (COUNT && lane == 0 && interactions) atomicAdd(interactions, (unsigned long long)count);
    const R fxs[2] = {fx.x, fx.y}, fys[2] = {fy.x, fy.y}, fzs[2] = {fz.x, fz.y};
    if constexpr (BLOCKED) {
        ax[0] = tot[0][threadIdx.x]; ay[0] = tot[1][threadIdx.x]; az[0] = tot[2][threadIdx.x];
        ax[1] = tot[3][threadIdx.x]; ay[1] = tot[4][threadIdx.x]; az[1] = tot[5][threadIdx.x];
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!valid[k]) continue;
        const int i = idx[k];
        const A sx = ax[k] + (A)fxs[k], sy = ay[k] + (A)fys[k], sz = az[k] + (A)fzs[k];
        if (OUT == OUT_ACC) {
            A4 o; o.x = G * sx; o.y = G * sy; o.z = G * sz; o.w = phi[k];
            acc_out[i] = o;
        } else {
            kick_drift<A>(i, G * sx, G * sy, G * sz, dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
        }
    }
}

// ---- per-body criterion, two bodies per lane (fp32) -----------------------------------------------------------
// walk_kernel's rule and arithmetic, body for body (the same interactions in the same order, the same bits), with the
// 128 Morton-consecutive bodies base + l and base + 64 + l in lane l behind ONE cursor.  The wave visits the union of
// what its 128 bodies visit, 1.16 times the visits of a 64-body wave instead of twice, and the arithmetic of the two
// bodies is packed (v_pk_add/fma/mul_f32): 16 vector instructions per visit for two bodies against 12 for one, and
// half the node records, scalar instructions and waits per body.  It rests on the sleeping-lane scheme of walk_kernel
// (a body that sits out carries +infinity as its floor under r^2, so nothing in the visit is per body but the
// arithmetic itself); with the per-body tests and selects of the earlier visit, two bodies per lane were slower
// (2.57 against 2.08 ms at 1M bodies).
// A = double (mixed precision): the fp32 partial sums of a body join its fp64 totals where walk_kernel's do (folds_here);
// the totals live in LDS, and positions are read from the fp32 mirror while the fp64 state is integrated.
template <typename A, int OUT, bool COUNT, int WB>
__global__ __launch_bounds__(WB) void walk_pair_kernel(
    const NodeF* __restrict__ nodes, const int* __restrict__ counters, const float4* __restrict__ posm, int t0, int t1,
    float floor2, A G, A dt, typename V4<A>::type* __restrict__ acc_out, const typename V4<A>::type* __restrict__ pos_in,
    const typename V4<A>::type* __restrict__ vel_in, typename V4<A>::type* __restrict__ pos_out,
    typename V4<A>::type* __restrict__ vel_out, float4* __restrict__ mirror_out,
    unsigned long long* __restrict__ interactions, const uint32_t* __restrict__ tail_orig, unsigned tail_limit,
    unsigned xcd_shift)
{
    typedef float R2 __attribute__((ext_vector_type(2)));
    using A4 = typename V4<A>::type;
    constexpr bool BLOCKED = !__is_same(A, float);
    const int na = counters[0];
    const int total = counters[1];
    const int tend = t1 < na ? t1 : na;
    const unsigned wg = walk_workgroup(xcd_shift);     // same XCD-aware placement as walk_kernel
#ifdef WALK_TIMELINE
    const unsigned long long began = wall_clock64();
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx[2];
    idx[0] = t0 + (int)wg * (2 * WB) + wave * 128 + lane;
    idx[1] = idx[0] + 64;
    bool valid[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = idx[k];
        if (OUT == OUT_FUSED && i >= tend && i < t1) {     // bodies removed by the bound check move on unchanged
            pos_out[i] = pos_in[i];
            vel_out[i] = vel_in[i];
            if (mirror_out) { const A4 p = pos_in[i]; mirror_out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w); }
        }
        valid[k] = i < tend;
    }
    if (__ballot(valid[0] || valid[1]) == 0 || total <= 0) return;

    R2 px = {0, 0}, py = {0, 0}, pz = {0, 0};
    if (valid[0]) { const float4 p = posm[idx[0]]; px.x = p.x; py.x = p.y; pz.x = p.z; }
    if (valid[1]) { const float4 p = posm[idx[1]]; px.y = p.x; py.y = p.y; pz.y = p.z; }
    // (NBE_QUIRK_DROP_TAIL: bodies past tail_limit in upload order are not walked, as in walk_kernel)
    const bool walked0 = valid[0] && !(tail_orig && tail_orig[idx[0]] >= tail_limit);
    const bool walked1 = valid[1] && !(tail_orig && tail_orig[idx[1]] >= tail_limit);
    int my_skip0 = walked0 ? 0 : 0x7fffffff, my_skip1 = walked1 ? 0 : 0x7fffffff;
    float floor2v = floor2;
    asm volatile("" : "+v"(floor2v));
    R2 seed = {walked0 ? floor2v : (float)INFINITY, walked1 ? floor2v : (float)INFINITY};
    int next_wake = 0x7fffffff, sp = 0, ends = 0;
    R2 fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
    A phi[2] = {0, 0};
    __shared__ A tot[BLOCKED ? 6 : 1][BLOCKED ? WB : 1];
    if constexpr (BLOCKED) {
#pragma unroll
        for (int k = 0; k < 6; k++) tot[k][threadIdx.x] = 0;
    }
    // Everything the epilogue needs from the kernel arguments is fetched here, ahead of the loop: a scalar load that
    // the compiler places at the loop's exit could be given the registers of the node request that is still in flight
    asm volatile("" :: "s"(G), "s"(dt), "s"(acc_out), "s"(interactions));
    unsigned count = 0;
    int cur = 0;
    auto visit = [&](auto& here, auto& there, auto look_for_end) -> bool {
        RawF nd;
        if constexpr (BLOCKED) nd = arrived_behind(here, px, py, pz); else nd = arrived(here);
        const int skip = nd.skip;
        asm volatile("" : "+s"(cur));             // (keeps the compare behind the wait, as in walk_kernel)
        if (cur == next_wake) {                   // the cursor leaves the innermost cell in which bodies sleep
            if (my_skip0 <= cur) seed.x = floor2v;
            if (my_skip1 <= cur) seed.y = floor2v;
            --sp;
            next_wake = __builtin_amdgcn_readlane(ends, sp);
        }
        const float nx = as_real(nd.x), ny = as_real(nd.y), nz = as_real(nd.z);
        const R2 dx = R2{nx, nx} - px, dy = R2{ny, ny} - py, dz = R2{nz, nz} - pz;
        R2 r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, seed)));
        const float w2 = as_real(nd.w2);
        const bool open0 = r2.x < w2, open1 = r2.y < w2;
        // (two ballots joined by a scalar OR: the ballot of "open0 || open1" goes through a vector register)
        // (the OR of the two votes sets the scalar condition code that picks the next node: written out, because the
        // compiler compares the mask with zero again and turns the verdict into a mask and back on the way to the request)
        const unsigned long long votes0 = __builtin_amdgcn_ballot_w64(open0), votes1 = __builtin_amdgcn_ballot_w64(open1);
        unsigned long long opening;
        int next;
        asm("s_or_b64 %0, %2, %3\n\ts_cselect_b32 %1, %4, %5"
            : "=&s"(opening), "=s"(next) : "s"(votes0), "s"(votes1), "s"(cur + 1), "s"(skip) : "scc");
        float keep = r2.x;
        there = request_raw(nodes, next, keep);    // the force arithmetic below runs while this is in flight
        r2.x = keep;
        const float mass = as_real(nd.m);
        bool u0 = false, u1 = false;
        if (COUNT) {
            u0 = !open0 && r2.x < (float)INFINITY && (dx.x != 0.0f || dy.x != 0.0f || dz.x != 0.0f) && (!BLOCKED || cur < total);
            u1 = !open1 && r2.y < (float)INFINITY && (dx.y != 0.0f || dy.y != 0.0f || dz.y != 0.0f) && (!BLOCKED || cur < total);
        }
        if (opening != 0) {
            // the cursor descends: the bodies that accept the cell sleep until the cursor reaches its end
            // ("!=" and not "<": the compiler then forms one compare for the vote and for the selects, not two)
            const bool fin0 = r2.x != (float)INFINITY, fin1 = r2.y != (float)INFINITY;
            const bool rest0 = !open0 && fin0, rest1 = !open1 && fin1;
            // (the vote from the masks of the plain compares: the ballot of a combined condition goes through a vector
            // register and back)
            const unsigned long long resting = (__builtin_amdgcn_ballot_w64(fin0) & ~votes0) | (__builtin_amdgcn_ballot_w64(fin1) & ~votes1);
            if (resting != 0) {
                if (rest0) { my_skip0 = skip; seed.x = (float)INFINITY; }
                if (rest1) { my_skip1 = skip; seed.y = (float)INFINITY; }
                // push the end of the cell around this one (a last child ends where its parent ends: one wake for both)
                if (skip != next_wake) {
                    // (v_writelane_b32, as in walk_kernel, does not compile here: "illegal VGPR to SGPR copy")
                    if (lane == sp) ends = next_wake;
                    ++sp;
                    next_wake = skip;
                }
            }
            // a body that opens the cell takes nothing from it: an infinite distance gives it the weight 0
            if (open0) r2.x = (float)INFINITY;
            if (open1) r2.y = (float)INFINITY;
        }
        const R2 rinv = {rsq_dev(r2.x), rsq_dev(r2.y)};
        const R2 w = (rinv * rinv) * (rinv * R2{mass, mass});
        fx = __builtin_elementwise_fma(dx, w, fx);
        fy = __builtin_elementwise_fma(dy, w, fy);
        fz = __builtin_elementwise_fma(dz, w, fz);
        if (COUNT) {
            count += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(u0)) + (unsigned)__popcll(__builtin_amdgcn_ballot_w64(u1));
            const R2 pm = rinv * R2{mass, mass};
            phi[0] += u0 ? (A)pm.x : (A)0;
            phi[1] += u1 ? (A)pm.y : (A)0;
        }
        cur = next;
        if constexpr (!BLOCKED) return next < total; else {
        // mixed precision: the fold and the end of the walk as in walk_kernel
        if (__builtin_expect(folds_here(nd), 0)) {
            // (a body takes the node iff its distance is still finite here: asleep or opening, it carries +infinity;
            // selects, as in walk_kernel)
            const bool take0 = r2.x != (float)INFINITY, take1 = r2.y != (float)INFINITY;
            tot[0][threadIdx.x] += take0 ? (A)fx.x : (A)0; tot[1][threadIdx.x] += take0 ? (A)fy.x : (A)0; tot[2][threadIdx.x] += take0 ? (A)fz.x : (A)0;
            tot[3][threadIdx.x] += take1 ? (A)fx.y : (A)0; tot[4][threadIdx.x] += take1 ? (A)fy.y : (A)0; tot[5][threadIdx.x] += take1 ? (A)fz.y : (A)0;
            fx = R2{take0 ? 0.0f : fx.x, take1 ? 0.0f : fx.y};
            fy = R2{take0 ? 0.0f : fy.x, take1 ? 0.0f : fy.y};
            fz = R2{take0 ? 0.0f : fz.x, take1 ? 0.0f : fz.y};
        }
        // (the wait on the way out of the visit, as in walk_kernel: with the wait behind the loop the two ways out have
        // to agree on the registers of both slots, and the compiler copies one while it is in flight)
        if constexpr (decltype(look_for_end)::value) {
            if (next >= total) { settle_plain(there); return false; }
        }
        return true;
        }
    };
    float none = 0;
    auto slot_a = request_raw(nodes, 0, none);
    decltype(slot_a) slot_b;
    for (;;) {
        if (!visit(slot_a, slot_b, Flag<!BLOCKED>{})) break;
        if (!visit(slot_b, slot_a, Flag<true>{})) break;
    }
    // The record requested last is never read, but has to land before its registers are reused.
    if constexpr (!BLOCKED) settle(slot_a, slot_b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#ifdef WALK_TIMELINE
    timeline_note(wg * (WB / 64) + wave, began);
#endif
    if (COUNT && lane == 0 && interactions) atomicAdd(interactions, (unsigned long long)count);
    const float fxs[2] = {fx.x, fx.y}, fys[2] = {fy.x, fy.y}, fzs[2] = {fz.x, fz.y};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!valid[k]) continue;
        const int i = idx[k];
        A sx = (A)fxs[k], sy = (A)fys[k], sz = (A)fzs[k];
        if constexpr (BLOCKED) { sx += tot[3 * k][threadIdx.x]; sy += tot[3 * k + 1][threadIdx.x]; sz += tot[3 * k + 2][threadIdx.x]; }
        if (OUT == OUT_ACC) {
            A4 o; o.x = G * sx; o.y = G * sy; o.z = G * sz; o.w = phi[k];
            acc_out[i] = o;
        } else {
            kick_drift<A>(i, G * sx, G * sy, G * sz, dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
        }
    }
}

#define WALK_LAUNCH(...)                                                                                     \
    hipLaunchKernelGGL((soft ? walk_kernel<R, A, NodeT, __VA_ARGS__, true, WB> : walk_kernel<R, A, NodeT, __VA_ARGS__, false, WB>),

// ---- yardstick: every body walks alone ----------------------------------------------------------------------------------
// The classic traversal: one lane, one body, one cursor per lane, node records through vector loads at up to 64
// addresses per instruction.  Every lane is busy in every step (no lane waits for another's cell), which is what a
// walk that regroups bodies by the subtrees they open would offer its lanes, so its cost per step prices that idea;
// it is also what the shared cursor is measured against.  Same arithmetic and order as walk_kernel: the same bits.
// fp32, accelerations only; reached through NBODY_WALK_PRIVATE=1 (nbe_accel, nbe_time_force_kernel), never by a step.
template <bool COUNT>
__global__ __launch_bounds__(WALK_BLOCK) void walk_private_kernel(const NodeF* __restrict__ nodes, const int* __restrict__ counters,
                                                                  const float4* __restrict__ posm, int t0, int t1, float floor2, float G,
                                                                  float4* __restrict__ acc_out, unsigned long long* __restrict__ interactions)
{
    const int na = counters[0];
    const int total = counters[1];
    const int tend = t1 < na ? t1 : na;
    const unsigned nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    const unsigned wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    const int i = t0 + (int)wg * WALK_BLOCK + threadIdx.x;
    if (i >= tend) return;
    const float4 p = posm[i];
    float fx = 0, fy = 0, fz = 0, phi = 0;
    unsigned count = 0, steps = 0;
    const float4* rec = reinterpret_cast<const float4*>(nodes);
    int cur = 0;
    while (cur < total) {
        const float4 a = rec[2 * (size_t)cur];                 // x, y, z, mass
        const float4 b = rec[2 * (size_t)cur + 1];             // (edge / theta)^2 (+ eps^2), skip
        const float dx = a.x - p.x, dy = a.y - p.y, dz = a.z - p.z;
        const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, floor2)));
        const bool open = r2 < b.x;
        if (!open) {
            const float rinv = rsq_dev(r2);
            const float rinv2 = rinv * rinv;
            const float w = rinv2 * (rinv * a.w);
            fx = fmaf(dx, w, fx); fy = fmaf(dy, w, fy); fz = fmaf(dz, w, fz);
            if (COUNT && (dx != 0.0f || dy != 0.0f || dz != 0.0f)) { count++; phi += rinv * a.w; }
        }
        cur = open ? cur + 1 : __float_as_int(b.y);
        if (COUNT) steps++;
    }
    acc_out[i] = make_float4(G * fx, G * fy, G * fz, phi);
    if (COUNT && interactions) {
        atomicAdd(interactions, (unsigned long long)count);
        atomicAdd(interactions + 1, (unsigned long long)steps);       // node visits of all lanes
    }
}

static inline unsigned xcd_grid(unsigned nwg, unsigned xcd_shift)
{
    if (xcd_shift == 0) return nwg;
    const unsigned g = 8u << xcd_shift;
    return (nwg + g - 1) / g * g;
}

template <typename R, typename A, typename NodeT>
static int launch_walk(nbe_engine* e, const void* src, uint64_t t0, uint64_t t1, bool fused, bool count, int src_buf, int dst_buf)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    const size_t nt = (size_t)(t1 - t0);
    if (nt == 0) return NBE_OK;
    const double eps = e->cfg.eps;
    const double tiny = __is_same(R, float) ? 1e-12 : 1e-60;
    const R floor2 = (R)(eps * eps > tiny ? eps * eps : tiny);
    // fp32 node records carry eps^2 in their opening threshold (set_edge in tree_build.hip); only the fp64 walk, which
    // repeats the reference's expression on the bare distance, takes the softening out of r^2 again
    const bool soft = eps * eps > tiny && !__is_same(NodeT, NodeF);
    const bool group = e->cfg.walk_criterion == NBE_WALK_PER_WAVE;
    const A G = (A)e->cfg.G, dt = (A)e->cfg.dt;
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    const A4* pos_in = (const A4*)e->posm[src_buf];
    const A4* vel_in = (const A4*)e->vel[src_buf];
    A4* pos_out = (A4*)e->posm[dst_buf];
    A4* vel_out = (A4*)e->vel[dst_buf];
    float4* mirror_out = mixed ? e->mirror[dst_buf] : nullptr;
    const uint32_t* tail_orig = (e->cfg.reference_quirks & NBE_QUIRK_DROP_TAIL) ? e->orig[e->orig_cur] : nullptr;
    const unsigned tail_limit = (unsigned)(4 * (e->n / 4));
    // chunks of workgroups dealt to the XCDs in turn, or one contiguous eighth each (walk_workgroup)
    // (measured: a whole Plummer sphere walks fastest with contiguous eighths, 1M / 2M / 4M / 16M bodies 1.78 / 3.35 / 6.31 /
    // 25.6 ms against 1.97 / 3.49 / 6.60 / 25.6 with chunks of 64; a rank's 2.1M bodies of a 16M-body tree 3.77 ms against
    // 3.62 with chunks of 16 to 64; so chunks serve launches that cover a part of a scene: ranks and shards)
    const bool part = e->dist || nt < (size_t)e->n;
    if (group) {
        dim3 ggrid(cdiv((size_t)(t1 - (t0 & ~127ull)), 2 * WALK_BLOCK));
#define GROUP_LAUNCH(...)                                                                                    \
    hipLaunchKernelGGL((soft ? walk_group_kernel<R, A, NodeT, __VA_ARGS__, true> : walk_group_kernel<R, A, NodeT, __VA_ARGS__, false>), \
                       ggrid, dim3(WALK_BLOCK), 0, e->stream, (const NodeT*)e->tw.nodes, e->tw.counters, (const R4*)src,  \
                       (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc, pos_in, vel_in, pos_out, vel_out,   \
                       mirror_out, e->tw.d_interactions)
        if (fused) {
            GROUP_LAUNCH(OUT_FUSED, false);
        } else if (count) {
            NBE_HIP(hipMemsetAsync(e->tw.d_interactions, 0, 8 * sizeof(unsigned long long), e->stream));
            GROUP_LAUNCH(OUT_ACC, true);
        } else {
            GROUP_LAUNCH(OUT_ACC, false);
        }
#undef GROUP_LAUNCH
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    }
    if constexpr (__is_same(R, float) && __is_same(A, float)) {
        if (e->walk_private && !fused && !(e->cfg.reference_quirks & NBE_QUIRK_DROP_TAIL)) {
            dim3 grid(cdiv(nt, WALK_BLOCK));
            if (count) {
                NBE_HIP(hipMemsetAsync(e->tw.d_interactions, 0, 8 * sizeof(unsigned long long), e->stream));
                hipLaunchKernelGGL((walk_private_kernel<true>), grid, dim3(WALK_BLOCK), 0, e->stream, (const NodeF*)e->tw.nodes, e->tw.counters,
                                   (const float4*)src, (int)t0, (int)t1, floor2, G, (float4*)e->acc, e->tw.d_interactions);
            } else {
                hipLaunchKernelGGL((walk_private_kernel<false>), grid, dim3(WALK_BLOCK), 0, e->stream, (const NodeF*)e->tw.nodes, e->tw.counters,
                                   (const float4*)src, (int)t0, (int)t1, floor2, G, (float4*)e->acc, e->tw.d_interactions);
            }
            NBE_HIP(hipGetLastError());
            return NBE_OK;
        }
    }
    // fp32 visits: two bodies per lane where there are enough bodies to fill the GPU several times over with 128-body waves
    // (measured, pair against one body per lane: 1M bodies 1.83 / 1.78 ms, 2M 3.46 / 3.49, 3M 4.89 / 5.16, 4M 6.53 /
    // 7.35, 8M 13.1 / 15.5, 16M 26.3 / 31.8: one million bodies are exactly one fill, and the slowest wave then sets
    // the time).  The sums are the same bits either way, also in mixed precision (folds_here), so the launch's own size
    // decides.  NBODY_WALK_BODIES_PER_LANE=1 or 2 forces the choice (tests).
    // (the instrumented pass in mixed precision stays with one body per lane: the compiler rejects that variant of
    // the pair kernel, "illegal VGPR to SGPR copy")
    // (mixed precision changes over at 1M bodies: its one-body kernel still tests every lane's sit-out index in every
    // visit, 17 vector instructions against 12, so two bodies per lane win earlier: two 512k-body galaxies 2.07 against
    // 2.13 ms)
    // (fp32 changes over at 2M bodies of a launch: level with one body per lane there, ahead from then on.  A rank of the
    // key-range-distributed tree walks the tree of the whole run, longer walks per body than a scene of its own size has:
    // 2.1M targets of a 16M-body tree take 3.78 ms with two bodies per lane against 4.27 ms with one, so it changes over
    // at 1.5M.)
    bool pair = false;
    if constexpr (__is_same(R, float)) {
        const int forced = e->walk_bodies_per_lane;
        const size_t least = __is_same(A, float) ? (e->dist ? 3u * (1u << 19) : 2u * (1u << 20)) : (1u << 20);
        pair = (forced == 2 || (forced != 1 && nt >= least)) && !(count && !__is_same(A, float));
    }
    // Workgroups of one wavefront for launches that fill the GPU's 8,192 wave slots more than once.  Slots are handed out
    // by the workgroup, and the wavefronts of a workgroup of four end at different times (lifetimes vary by 12-25 %), so
    // slots wait for their siblings: 7,500 of 8,192 were busy in the steady state of a 16M-body launch, 8,030 are with
    // workgroups of one (profiles/walk_timeline.py).  Measured on one box with 64 / 128 / 256 / 512 threads, fp32: 1M bodies
    // 1.695 / 1.730 / 1.726 / 1.740 ms, 4M 6.14 / 6.30 / 6.28 / 6.40, 16M 24.90 / 25.61 / 25.69 / 26.44; mixed precision, 64
    // against 256: one body per lane 1M 1.88 / 1.94, 2M 3.54 / 3.63; two per lane 2M level, two 8M-body galaxies 52.7 / 53.5,
    // but 1M, which is one fill exactly, 1.85 / 1.80: nothing is refilled there, so such launches keep 256.  fp64 is level
    // (3.20 ms at 1M all round) and keeps 256.
    // Few targets: a wavefront's walk is a chain of dependent visits (~130 ns each); with a wavefront or two per SIMD
    // nothing hides it, and most of the 1,024 SIMDs have nothing to do.  Fewer bodies per wavefront (64 >> sparse) make
    // more and shorter walks, the same bits for every body.  Measured, force kernel with 64 / 16 / 4 / 1 bodies per
    // wavefront: the reference's default scene (1,002 bodies, theta 1) 34 / 21 / 19 / 14 us (step 98 -> 60 us), 4,096 bodies
    // 238 / 190 / 173 / 163, 16,384 bodies 372 / 303 / 287 / 562, 65,536 bodies 488 / 427 / 745 / 2,080: best where the
    // launch has 2,048 to 4,096 wavefronts (fp64, whose wavefronts are heavier, where it has 2,048), never beyond.
    unsigned sparse = 0;
    if (!pair) {
        const size_t most = __is_same(R, double) ? 2048 : 4096;
        while (sparse < 6 && cdiv(nt, 64u >> (sparse + 1)) <= most) sparse++;
        if (e->walk_sparse >= 0) sparse = (unsigned)e->walk_sparse;             // NBODY_WALK_SPARSE
    }
    const size_t wavefronts = pair ? cdiv(nt, 128) : cdiv(nt, 64u >> sparse);
    auto launch = [&](auto wb) -> int {
        constexpr int WB = decltype(wb)::value;
        // chunks of workgroups dealt to the XCDs in turn are counted in groups of 256 bodies whatever the workgroup's size
        const unsigned wb_shift = WB == 64 ? 2u : 0u;
        const unsigned xcd_shift = e->walk_xcd_shift >= 0 ? (e->walk_xcd_shift > 0 ? (unsigned)e->walk_xcd_shift + wb_shift : 0u)
                                                          : (part ? 5u + wb_shift : 0u);
        if constexpr (__is_same(R, float)) {
            if (pair) {
                dim3 pgrid(xcd_grid(cdiv(nt, 2 * WB), xcd_shift));
#define PAIR_LAUNCH(...)                                                                                                       \
    hipLaunchKernelGGL((walk_pair_kernel<A, __VA_ARGS__, WB>), pgrid, dim3(WB), 0, e->stream, (const NodeF*)e->tw.nodes,       \
                       e->tw.counters, (const float4*)src, (int)t0, (int)t1, floor2, G, dt, (A4*)e->acc, pos_in, vel_in,       \
                       pos_out, vel_out, mirror_out, e->tw.d_interactions, tail_orig, tail_limit, xcd_shift)
                if (fused) {
                    PAIR_LAUNCH(OUT_FUSED, false);
                } else if (count) {
                    if constexpr (__is_same(A, float)) {
                        NBE_HIP(hipMemsetAsync(e->tw.d_interactions, 0, 8 * sizeof(unsigned long long), e->stream));
                        PAIR_LAUNCH(OUT_ACC, true);
                    }
                } else {
                    PAIR_LAUNCH(OUT_ACC, false);
                }
#undef PAIR_LAUNCH
                NBE_HIP(hipGetLastError());
                return NBE_OK;
            }
        }
        dim3 grid(xcd_grid(cdiv(cdiv(nt, 64u >> sparse), WB / 64), xcd_shift));
        if (fused) {
            WALK_LAUNCH(OUT_FUSED, false) grid, dim3(WB), 0, e->stream, (const NodeT*)e->tw.nodes,
                               e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                               pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions, tail_orig, tail_limit, xcd_shift, sparse);
        } else if (count) {
            NBE_HIP(hipMemsetAsync(e->tw.d_interactions, 0, 8 * sizeof(unsigned long long), e->stream));
            WALK_LAUNCH(OUT_ACC, true) grid, dim3(WB), 0, e->stream, (const NodeT*)e->tw.nodes,
                               e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                               pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions, tail_orig, tail_limit, xcd_shift, sparse);
        } else {
            WALK_LAUNCH(OUT_ACC, false) grid, dim3(WB), 0, e->stream, (const NodeT*)e->tw.nodes,
                               e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                               pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions, tail_orig, tail_limit, xcd_shift, sparse);
        }
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    };
    if constexpr (__is_same(R, float)) {
        if (wavefronts > 8192) return launch(Size<64>{});
    }
    return launch(Size<WALK_BLOCK>{});
}

// Forces on sorted targets [t0, t1) of buffer `src_buf` from the tree built over that buffer.  `count` (only
// without `fused`) also counts the force evaluations into e->tw.d_interactions.
int tree_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, bool count, int dst_buf)
{
    switch (e->cfg.precision) {
    case NBE_PREC_F32: return launch_walk<float, float, NodeF>(e, e->posm[src_buf], t0, t1, fused, count, src_buf, dst_buf);
    case NBE_PREC_MIXED: return launch_walk<float, double, NodeF>(e, e->mirror[src_buf], t0, t1, fused, count, src_buf, dst_buf);
    case NBE_PREC_F64: return launch_walk<double, double, NodeD>(e, e->posm[src_buf], t0, t1, fused, count, src_buf, dst_buf);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}
