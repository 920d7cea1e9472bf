// Tiled O(N^2) direct-sum force kernels with the kick-drift integrator fused into the epilogue.
//
// Replaces the serial pair loop of main.go:105-128 (dist physics.go:123-128, gravity physics.go:135-145)
// and the update loop main.go:171-177 (physics.go:101-115).  The reference accumulates force = G*mi*mj/r^2
// along (xj-xi)/r and divides by mi in update; here the acceleration a_i = G * sum_j mj*(xj-xi)/r^3 is
// accumulated directly, which is the same quantity up to rounding.
//
// Layout: each 256-thread workgroup owns 256*ITILE targets held in registers and streams the sources
// through LDS in tiles of 256 (one 16- or 32-byte element per thread per tile, read back as broadcast
// ds_read_b128).  Per source the inner loop spends 3 sub, 3 fma (r^2), 1 rsq, 3 mul, 3 fma per target.
// The self term and padded sources contribute exactly 0: r^2 carries a floor (eps^2 or a tiny constant)
// so rsq stays finite and is multiplied by dx = 0 or m = 0.
#include "engine_internal.h"
#include "kernels_common.h"

// R: type of the pair arithmetic; A: type of the running sums and of the state (A == R, or R = float with
// A = double).  Each 256-source tile is summed in R and the tile sums are accumulated in A.
template <typename R, typename A, int ITILE, int OUT>
__global__ __launch_bounds__(256) void direct_kernel(
    const typename V4<R>::type* __restrict__ src, int n_src, int t0, int t1, int jchunk, R floor2, A G, A dt,
    typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    __shared__ R4 sh[256];
    const int tid = threadIdx.x;
    const int ibase = t0 + blockIdx.x * (256 * ITILE);

    R xi[ITILE], yi[ITILE], zi[ITILE];
    A ax[ITILE], ay[ITILE], az[ITILE];
#pragma unroll
    for (int k = 0; k < ITILE; k++) {
        int i = ibase + k * 256 + tid;
        if (i > t1 - 1) i = t1 - 1;
        R4 p = src[i];
        xi[k] = p.x; yi[k] = p.y; zi[k] = p.z;
        ax[k] = 0; ay[k] = 0; az[k] = 0;
    }

    const int jb = blockIdx.y * jchunk;
    const int je = (jb + jchunk < n_src) ? jb + jchunk : n_src;
    R4 pad; pad.x = 0; pad.y = 0; pad.z = 0; pad.w = 0;
    R4 nxt = pad;
    if (jb + tid < je) nxt = src[jb + tid];

    for (int jt = jb; jt < je; jt += 256) {
        __syncthreads();
        sh[tid] = nxt;
        __syncthreads();
        nxt = pad;
        if (jt + 256 + tid < je) nxt = src[jt + 256 + tid];

        // Sums are blocked: one partial per 256-source tile, added to the running total afterwards.  This
        // keeps fp32 rounding growth at ~sqrt(256) + sqrt(N/256) ulps instead of ~sqrt(N).
        R fx[ITILE], fy[ITILE], fz[ITILE];
#pragma unroll
        for (int k = 0; k < ITILE; k++) { fx[k] = 0; fy[k] = 0; fz[k] = 0; }
#pragma unroll 8
        for (int jj = 0; jj < 256; jj++) {
            const R4 s = sh[jj];
#pragma unroll
            for (int k = 0; k < ITILE; k++) {
                R dx = s.x - xi[k], dy = s.y - yi[k], dz = s.z - zi[k];
                R r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, floor2)));
                R rinv = rsq_dev(r2);
                R w = (rinv * rinv) * (rinv * s.w);
                fx[k] = fma(dx, w, fx[k]);
                fy[k] = fma(dy, w, fy[k]);
                fz[k] = fma(dz, w, fz[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < ITILE; k++) { ax[k] += (A)fx[k]; ay[k] += (A)fy[k]; az[k] += (A)fz[k]; }
    }

#pragma unroll
    for (int k = 0; k < ITILE; k++) {
        const int i = ibase + k * 256 + tid;
        if (i >= t1) continue;
        if (OUT == OUT_PARTIAL) {
            A4 o; o.x = ax[k]; o.y = ay[k]; o.z = az[k]; o.w = 0;
            acc_out[(size_t)blockIdx.y * (size_t)(t1 - t0) + (size_t)(i - t0)] = o;
        } else if (OUT == OUT_ACC) {
            A4 o; o.x = G * ax[k]; o.y = G * ay[k]; o.z = G * az[k]; o.w = 0;
            acc_out[i] = o;
        } else {
            kick_drift<A>(i, G * ax[k], G * ay[k], G * az[k], dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
        }
    }
}

// Sums the per-source-range partial accelerations in a fixed order and finishes as above.
template <typename A, bool FUSED>
__global__ __launch_bounds__(256) void direct_finish_kernel(
    const typename V4<A>::type* __restrict__ partial, int nsplit, int t0, int t1, A G, A dt,
    typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out)
{
    using A4 = typename V4<A>::type;
    const int i = t0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= t1) return;
    const size_t nt = (size_t)(t1 - t0);
    A ax = 0, ay = 0, az = 0;
    for (int s = 0; s < nsplit; s++) {
        A4 p = partial[(size_t)s * nt + (size_t)(i - t0)];
        ax += p.x; ay += p.y; az += p.z;
    }
    if (FUSED) {
        kick_drift<A>(i, G * ax, G * ay, G * az, dt, pos_in, vel_in, pos_out, vel_out, mirror_out);
    } else {
        A4 o; o.x = G * ax; o.y = G * ay; o.z = G * az; o.w = 0;
        acc_out[i] = o;
    }
}

template <typename R, typename A, int ITILE>
static int launch_direct(nbe_engine* e, const void* src, uint64_t t0, uint64_t t1, bool fused, int src_buf, int dst_buf)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    const size_t n = e->n, nt = (size_t)(t1 - t0);
    if (nt == 0) return NBE_OK;
    if (n > 0x7fffffffull) { nbe_set_error("direct sum supports at most 2^31-1 bodies"); return NBE_E_UNSUPPORTED; }
    const unsigned iblocks = cdiv(nt, 256 * ITILE);
    // Aim for >= 4 workgroups per CU; split the source range across blockIdx.y when targets alone are too few.
    unsigned jsplit = 1;
    if (iblocks < 1024) jsplit = cdiv(1024, iblocks);
    const unsigned jtiles = cdiv(n, 256);
    if (jsplit > jtiles) jsplit = jtiles;
    if (jsplit > 64) jsplit = 64;
    size_t jchunk = (size_t)cdiv(cdiv(n, jsplit), 256) * 256;
    jsplit = cdiv(n, jchunk);

    const double eps = e->cfg.eps;
    const double tiny = __is_same(R, float) ? 1e-12 : 1e-60;
    const R floor2 = (R)(eps * eps > tiny ? eps * eps : tiny);
    const A G = (A)e->cfg.G, dt = (A)e->cfg.dt;
    const A4* pos_in = (const A4*)e->posm[src_buf];
    const A4* vel_in = (const A4*)e->vel[src_buf];
    A4* pos_out = (A4*)e->posm[dst_buf];
    A4* vel_out = (A4*)e->vel[dst_buf];
    float4* mirror_out = (e->cfg.precision == NBE_PREC_MIXED) ? e->mirror[dst_buf] : nullptr;
    A4* acc = (A4*)e->acc;

    if (jsplit == 1) {
        dim3 grid(iblocks, 1);
        if (fused)
            hipLaunchKernelGGL((direct_kernel<R, A, ITILE, OUT_FUSED>), grid, dim3(256), 0, e->stream, (const R4*)src,
                               (int)n, (int)t0, (int)t1, (int)jchunk, floor2, G, dt, acc, pos_in, vel_in, pos_out,
                               vel_out, mirror_out);
        else
            hipLaunchKernelGGL((direct_kernel<R, A, ITILE, OUT_ACC>), grid, dim3(256), 0, e->stream, (const R4*)src,
                               (int)n, (int)t0, (int)t1, (int)jchunk, floor2, G, dt, acc, pos_in, vel_in, pos_out,
                               vel_out, mirror_out);
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    }

    const size_t need = (size_t)jsplit * nt * sizeof(A4);
    if (e->partial_len < need) {
        if (e->partial) NBE_HIP(hipFree(e->partial));
        e->partial = nullptr; e->partial_len = 0;
        NBE_HIP(hipMalloc(&e->partial, need));
        e->partial_len = need;
    }
    dim3 grid(iblocks, jsplit);
    hipLaunchKernelGGL((direct_kernel<R, A, ITILE, OUT_PARTIAL>), grid, dim3(256), 0, e->stream, (const R4*)src, (int)n,
                       (int)t0, (int)t1, (int)jchunk, floor2, G, dt, (A4*)e->partial, pos_in, vel_in, pos_out, vel_out,
                       mirror_out);
    NBE_HIP(hipGetLastError());
    dim3 fgrid(cdiv(nt, 256));
    if (fused)
        hipLaunchKernelGGL((direct_finish_kernel<A, true>), fgrid, dim3(256), 0, e->stream, (const A4*)e->partial,
                           (int)jsplit, (int)t0, (int)t1, G, dt, acc, pos_in, vel_in, pos_out, vel_out, mirror_out);
    else
        hipLaunchKernelGGL((direct_finish_kernel<A, false>), fgrid, dim3(256), 0, e->stream, (const A4*)e->partial,
                           (int)jsplit, (int)t0, (int)t1, G, dt, acc, pos_in, vel_in, pos_out, vel_out, mirror_out);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// Forces on targets [t0, t1) of buffer `src_buf` from all n bodies of that buffer.  fused: integrate into
// buffer `dst_buf`; otherwise accelerations go to e->acc at the target's index.
int direct_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, int dst_buf)
{
    e->stats.interactions = (uint64_t)(t1 - t0) * (uint64_t)(e->n ? e->n - 1 : 0);
    switch (e->cfg.precision) {
    case NBE_PREC_F32: return launch_direct<float, float, 4>(e, e->posm[src_buf], t0, t1, fused, src_buf, dst_buf);
    case NBE_PREC_MIXED: return launch_direct<float, double, 4>(e, e->mirror[src_buf], t0, t1, fused, src_buf, dst_buf);
    case NBE_PREC_F64: return launch_direct<double, double, 2>(e, e->posm[src_buf], t0, t1, fused, src_buf, dst_buf);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}
