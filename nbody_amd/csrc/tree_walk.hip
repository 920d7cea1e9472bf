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
struct RawD { unsigned long long x, y, z, m, w; int skip; };

__device__ __forceinline__ RawF load_raw(const NodeF* __restrict__ nodes, int i)
{
    // 32-bit byte offset from the array base: the scalar load takes it as an SGPR offset (no 64-bit adds)
    const char* base = reinterpret_cast<const char*>(nodes) + (size_t)((unsigned)i * 32u);
    const uint4 a = *reinterpret_cast<const uint4*>(base);
    const uint2 b = *reinterpret_cast<const uint2*>(base + 16);
    return RawF{a.x, a.y, a.z, a.w, b.x, (int)b.y};
}
__device__ __forceinline__ RawD load_raw(const NodeD* __restrict__ nodes, int i)
{
    const ulonglong4 a = *reinterpret_cast<const ulonglong4*>(nodes + i);
    const ulonglong1 w = *reinterpret_cast<const ulonglong1*>(reinterpret_cast<const char*>(nodes + i) + 32);
    const int sk = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(nodes + i) + 40);
    return RawD{a.x, a.y, a.z, a.w, w.x, sk};
}
__device__ __forceinline__ RawF pick(bool c, const RawF& a, const RawF& b)
{
    return RawF{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.m : b.m, c ? a.w2 : b.w2, c ? a.skip : b.skip};
}
__device__ __forceinline__ RawD pick(bool c, const RawD& a, const RawD& b)
{
    return RawD{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.m : b.m, c ? a.w : b.w, c ? a.skip : b.skip};
}
__device__ __forceinline__ float as_real(unsigned v) { return __uint_as_float(v); }
__device__ __forceinline__ double as_real(unsigned long long v) { return __longlong_as_double((long long)v); }

// Opening test of tree.go:216-217, "edge / r > theta".  fp32 nodes store edge^2 / theta^2, so the test is one
// compare, r^2 < edge^2/theta^2; fp64 nodes store the edge and repeat the reference's expression literally.
// A leaf stores 0 and can never open.
__device__ __forceinline__ bool node_opens(const RawF& nd, float r2, float) { return r2 < as_real(nd.w2); }
__device__ __forceinline__ bool node_opens(const RawD& nd, double r2, double theta) { return as_real(nd.w) / sqrt(r2) > theta; }

// COUNT: instrumented pass: also counts the force evaluations (tree.go:232-238 executions) of the wave and sums
// each body's potential (sum of mass / softened distance over the same accepted nodes, returned in the w
// component of the acceleration record, without the factor -G); the production step runs without it.
// SOFT: Plummer softening is on, so the opening test must undo the eps^2 folded into r^2 (with eps = 0 the
// floor is 1e-12 m^2, far below the resolution of any cell edge, and the test uses r^2 as is).
template <typename R, typename A, typename NodeT, int OUT, bool COUNT, bool SOFT>
__global__ __launch_bounds__(256) void walk_kernel(
    const NodeT* __restrict__ nodes, const int* __restrict__ counters, const typename V4<R>::type* __restrict__ posm,
    int t0, int t1, R theta, R floor2, A G, A dt, typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out, unsigned long long* __restrict__ interactions)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    constexpr bool BLOCKED = !__is_same(R, A);
    const int na = counters[0];
    const int total = counters[1];
    const int tend = t1 < na ? t1 : na;
    const int i = t0 + blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < tend;
    if (__ballot(valid) == 0) return;            // whole wave past the end

    R px = 0, py = 0, pz = 0;
    if (valid) { R4 p = posm[i]; px = p.x; py = p.y; pz = p.z; }
    R fx = 0, fy = 0, fz = 0;
    A ax = 0, ay = 0, az = 0;
    int my_skip = valid ? 0 : 0x7fffffff;        // the lane sits out while cursor < my_skip
    unsigned count = 0;
    A phi = 0;
    int it = 0;
    // The cursor and the node under it live in scalar registers: one scalar load per visit.  Measured
    // alternatives that fetch both possible successors ahead of the vote were slower: the scalar unit is
    // shared by the CU's four SIMDs, so the extra loads and selects cost more than the latency they hide
    // (8 waves per SIMD already cover it).
    int cur = 0;
    while (cur < total) {
        const auto nd = load_raw(nodes, cur);
        const int skip = nd.skip;
        const R dx = as_real(nd.x) - px, dy = as_real(nd.y) - py, dz = as_real(nd.z) - pz;
        // The floor under r^2 (1e-12 m^2 unless softening is on) keeps rsq finite for the body's own leaf and
        // for bodies at distance 0, which then contribute exactly 0 because dx = dy = dz = 0.
        const R r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, floor2)));
        const bool active = cur >= my_skip;
        // lanes sitting out get an infinite distance, so one compare is both the opening test and the vote
        const R rtest = active ? (SOFT ? r2 - floor2 : r2) : (R)INFINITY;
        const bool open = node_opens(nd, rtest, theta);
        const bool take = active && !open;
        if (take) my_skip = skip;
        const R rinv = rsq_dev(r2);
        const R w = take ? (rinv * rinv) * (rinv * as_real(nd.m)) : (R)0;
        fx = fma(dx, w, fx);
        fy = fma(dy, w, fy);
        fz = fma(dz, w, fz);
        if (COUNT) {
            const bool use = take && (dx != (R)0 || dy != (R)0 || dz != (R)0);
            count += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(use));
            if (use) phi += (A)(rinv * as_real(nd.m));
        }
        cur = __builtin_amdgcn_ballot_w64(open) != 0 ? cur + 1 : skip;
        if (BLOCKED && ((++it) & 63) == 0) {
            ax += (A)fx; ay += (A)fy; az += (A)fz;
            fx = 0; fy = 0; fz = 0;
        }
    }
    ax += (A)fx; ay += (A)fy; az += (A)fz;

    if (COUNT && (threadIdx.x & 63) == 0 && interactions) atomicAdd(interactions, (unsigned long long)count);
    if (!valid) return;
    if (OUT == OUT_ACC) {
        A4 o; o.x = G * ax; o.y = G * ay; o.z = G * az; o.w = phi;
        acc_out[i] = o;
    } else {
        kick_drift<A>(i, G * ax, G * ay, G * az, dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
    }
}

// Bodies past n_alive (removed by the bound check) are carried along unchanged when the state moves to the
// next buffer.
template <typename S4>
__global__ __launch_bounds__(256) void carry_dead_kernel(const int* __restrict__ counters, int t0, int t1,
                                                         const S4* __restrict__ pos_in, const S4* __restrict__ vel_in,
                                                         const float4* __restrict__ mirror_in, S4* __restrict__ pos_out,
                                                         S4* __restrict__ vel_out, float4* __restrict__ mirror_out)
{
    const int na = counters[0];
    const int i = t0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= t1 || i < na) return;
    pos_out[i] = pos_in[i];
    vel_out[i] = vel_in[i];
    if (mirror_in) mirror_out[i] = mirror_in[i];
}

#define WALK_LAUNCH(...)                                                                                     \
    hipLaunchKernelGGL((soft ? walk_kernel<R, A, NodeT, __VA_ARGS__, true> : walk_kernel<R, A, NodeT, __VA_ARGS__, false>),

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
    const bool soft = eps * eps > tiny;
    const A G = (A)e->cfg.G, dt = (A)e->cfg.dt;
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    const A4* pos_in = (const A4*)e->posm[src_buf];
    const A4* vel_in = (const A4*)e->vel[src_buf];
    A4* pos_out = (A4*)e->posm[dst_buf];
    A4* vel_out = (A4*)e->vel[dst_buf];
    float4* mirror_out = mixed ? e->mirror[dst_buf] : nullptr;
    dim3 grid(cdiv(nt, 256));
    if (fused) {
        WALK_LAUNCH(OUT_FUSED, false) grid, dim3(256), 0, e->stream, (const NodeT*)e->tw.nodes,
                           e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                           pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions);
        hipLaunchKernelGGL((carry_dead_kernel<A4>), grid, dim3(256), 0, e->stream, e->tw.counters, (int)t0, (int)t1, pos_in,
                           vel_in, mixed ? e->mirror[src_buf] : (const float4*)nullptr, pos_out, vel_out, mirror_out);
    } else if (count) {
        NBE_HIP(hipMemsetAsync(e->tw.d_interactions, 0, 8 * sizeof(unsigned long long), e->stream));
        WALK_LAUNCH(OUT_ACC, true) grid, dim3(256), 0, e->stream, (const NodeT*)e->tw.nodes,
                           e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                           pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions);
    } else {
        WALK_LAUNCH(OUT_ACC, false) grid, dim3(256), 0, e->stream, (const NodeT*)e->tw.nodes,
                           e->tw.counters, (const R4*)src, (int)t0, (int)t1, (R)e->cfg.theta, floor2, G, dt, (A4*)e->acc,
                           pos_in, vel_in, pos_out, vel_out, mirror_out, e->tw.d_interactions);
    }
    NBE_HIP(hipGetLastError());
    return NBE_OK;
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
