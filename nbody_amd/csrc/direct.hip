// Tiled O(N^2) direct-sum force kernels with the kick-drift integrator fused into the epilogue.
//
// Replaces the serial pair loop of main.go:105-128 (dist physics.go:123-128, gravity physics.go:135-145)
// and the update loop main.go:171-177 (physics.go:101-115).  The reference accumulates force = G*mi*mj/r^2
// along (xj-xi)/r and divides by mi in update; here the acceleration a_i = G * sum_j mj*(xj-xi)/r^3 is
// accumulated directly, which is the same quantity up to rounding.
//
// Layout: each workgroup owns BLOCK*ITILE targets held in registers and streams the sources through LDS
// in tiles of 256 (16- or 32-byte elements, read back as broadcast ds_read_b128).  Per source and target
// the inner loop spends 3 sub, 3 fma (r^2), 1 rsq, 3 mul, 3 fma.
// The self term and padded sources contribute exactly 0: r^2 carries a floor (eps^2 or a tiny constant)
// so rsq stays finite and is multiplied by dx = 0 or m = 0.
#include "engine_internal.h"
#include "kernels_common.h"
#include <cmath>

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

#define DIRECT_TILE 256   // sources per LDS tile
#ifndef DIRECT_ITILE
#define DIRECT_ITILE 4    // targets per lane of the fp32 kernels
#endif

// Per-lane target state and the tile loop of the generic (scalar arithmetic) kernel; fp64 uses it.
template <typename R, int ITILE> struct Targets;

template <int ITILE> struct Targets<double, ITILE> {
    double px[ITILE], py[ITILE], pz[ITILE], gx[ITILE], gy[ITILE], gz[ITILE];
    __device__ __forceinline__ void set(int k, double x, double y, double z) { px[k] = x; py[k] = y; pz[k] = z; }
    __device__ __forceinline__ void tile(const double4* sh, double floor2)
    {
#pragma unroll
        for (int k = 0; k < ITILE; k++) { gx[k] = 0; gy[k] = 0; gz[k] = 0; }
#pragma unroll 4
        for (int jj = 0; jj < DIRECT_TILE; jj++) {
            const double4 s = sh[jj];
#pragma unroll
            for (int k = 0; k < ITILE; k++) {
                const double dx = s.x - px[k], dy = s.y - py[k], dz = s.z - pz[k];
                const double r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, floor2)));
                const double rinv = rsq2_dev(r2);             // twice 1/r: the sums carry a factor 8, see direct_kernel's G
                const double w = (rinv * rinv) * (rinv * s.w);
                gx[k] = fma(dx, w, gx[k]); gy[k] = fma(dy, w, gy[k]); gz[k] = fma(dz, w, gz[k]);
            }
        }
    }
    __device__ __forceinline__ double sum_x(int k) const { return gx[k]; }
    __device__ __forceinline__ double sum_y(int k) const { return gy[k]; }
    __device__ __forceinline__ double sum_z(int k) const { return gz[k]; }
};

// R: type of the pair arithmetic; A: type of the running sums and of the state (A == R, or R = float with
// A = double).  Each 256-source tile is summed in R and the tile sums are accumulated in A, which keeps
// fp32 rounding growth at ~sqrt(256) + sqrt(N/256) ulps instead of ~sqrt(N).
// BLOCK threads own BLOCK*ITILE targets (lane t: targets ibase + k*BLOCK + t).  fp32 runs one wave per
// workgroup: the waves of a CU then drift apart instead of meeting at a barrier every tile, which measured
// 5 % faster than 256-thread workgroups.
template <typename R, typename A, int ITILE, int BLOCK, int OUT>
__global__ __launch_bounds__(BLOCK) void direct_kernel(
    const typename V4<R>::type* __restrict__ src, int n_src, int t0, int t1, int jchunk, R floor2, A G, A dt,
    typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    __shared__ R4 sh[DIRECT_TILE];
    const int tid = threadIdx.x;
    const int ibase = t0 + blockIdx.x * (BLOCK * ITILE);

    Targets<R, ITILE> tg;
    A ax[ITILE], ay[ITILE], az[ITILE];
#pragma unroll
    for (int k = 0; k < ITILE; k++) {
        int i = ibase + k * BLOCK + tid;
        if (i > t1 - 1) i = t1 - 1;
        const R4 p = src[i];
        tg.set(k, p.x, p.y, p.z);
        ax[k] = 0; ay[k] = 0; az[k] = 0;
    }

    // (fp64 pair arithmetic works with 2/r, rsq2_dev: its sums are exactly 8 times the plain ones.  G takes the factor
    // back here; partial sums are stored as they are and the caller hands the finishing kernel G / 8)
    if (__is_same(R, double)) G = G * (A)0.125;
    const int jb = blockIdx.y * jchunk;
    const int je = (jb + jchunk < n_src) ? jb + jchunk : n_src;
    for (int jt = jb; jt < je; jt += DIRECT_TILE) {
        __syncthreads();
        if (jt + DIRECT_TILE <= je) {
#pragma unroll
            for (int q = 0; q < DIRECT_TILE / BLOCK; q++) sh[q * BLOCK + tid] = src[jt + q * BLOCK + tid];
        } else {                                         // last, partial tile: pad with massless sources
#pragma unroll
            for (int q = 0; q < DIRECT_TILE / BLOCK; q++) {
                const int j = jt + q * BLOCK + tid;
                R4 v = src[j < je ? j : je - 1];
                if (j >= je) v.w = 0;
                sh[q * BLOCK + tid] = v;
            }
        }
        __syncthreads();
        tg.tile(sh, floor2);
#pragma unroll
        for (int k = 0; k < ITILE; k++) { ax[k] += (A)tg.sum_x(k); ay[k] += (A)tg.sum_y(k); az[k] += (A)tg.sum_z(k); }
    }

#pragma unroll
    for (int k = 0; k < ITILE; k++) {
        const int i = ibase + k * BLOCK + tid;
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

// fp32 pair arithmetic.  Targets are held two to a register pair so that the whole body is packed math
// (v_pk_add_f32, v_pk_fma_f32, v_pk_mul_f32, the source broadcast through op_sel); only the two v_rsq_f32
// per pair are scalar.  Measured on MI355X: a packed op issues in 4 cycles per wave and v_rsq_f32 in 8, so a
// pair of interactions costs 12*4 + 2*8 = 64 cycles; at the sustained clock (2.24-2.30 GHz measured in the
// kernel) that bounds one GPU at 4.6-4.7e12 interactions/s, and this kernel reaches ~94 % of it.
// One wave per workgroup (BLOCK = 64): the waves of a CU drift apart instead of meeting at a barrier every
// tile (+5 % over 256-thread workgroups).  This file is compiled with the max-ILP scheduler strategy (see the
// Makefile): it interleaves several sources so that no wait states are needed behind v_rsq_f32 (+8 %).
// EQM: every body has the same mass (the uniform and Plummer clouds of the benchmarks, any equal-mass scene).  The
// mass then leaves the pair loop, w = rinv^3 instead of rinv^2 * (rinv * m): 11 packed operations per pair instead
// of 12, and the common mass is folded into G by the caller.  Sources that pad the last tile cannot be massless
// here; they sit 1e18 m away instead, where rinv^3 underflows to exactly 0.
template <typename A, int ITILE, int BLOCK, int OUT, bool EQM>
__global__ __launch_bounds__(BLOCK) void direct_kernel_f32(
    const float4* __restrict__ src, int n_src, int t0, int t1, int jchunk, float floor2, A G, A dt,
    typename V4<A>::type* __restrict__ acc_out,
    const typename V4<A>::type* __restrict__ pos_in, const typename V4<A>::type* __restrict__ vel_in,
    typename V4<A>::type* __restrict__ pos_out, typename V4<A>::type* __restrict__ vel_out,
    float4* __restrict__ mirror_out)
{
    using A4 = typename V4<A>::type;
    static_assert(ITILE % 2 == 0, "fp32 targets are processed in pairs");
    __shared__ float4 sh[DIRECT_TILE];
    const int tid = threadIdx.x;
    const int ibase = t0 + blockIdx.x * (BLOCK * ITILE);

    f2 xi[ITILE / 2], yi[ITILE / 2], zi[ITILE / 2];
    A ax[ITILE], ay[ITILE], az[ITILE];
#pragma unroll
    for (int p = 0; p < ITILE / 2; p++) {
        int i0 = ibase + (2 * p) * BLOCK + tid, i1 = i0 + BLOCK;
        if (i0 > t1 - 1) i0 = t1 - 1;
        if (i1 > t1 - 1) i1 = t1 - 1;
        const float4 a = src[i0], b = src[i1];
        xi[p] = f2{a.x, b.x}; yi[p] = f2{a.y, b.y}; zi[p] = f2{a.z, b.z};
    }
#pragma unroll
    for (int k = 0; k < ITILE; k++) { ax[k] = 0; ay[k] = 0; az[k] = 0; }
    const f2 fl = {floor2, floor2};

    const int jb = blockIdx.y * jchunk;
    const int je = (jb + jchunk < n_src) ? jb + jchunk : n_src;
    for (int jt = jb; jt < je; jt += DIRECT_TILE) {
        __syncthreads();
        if (jt + DIRECT_TILE <= je) {
#pragma unroll
            for (int q = 0; q < DIRECT_TILE / BLOCK; q++) sh[q * BLOCK + tid] = src[jt + q * BLOCK + tid];
        } else {                                         // last, partial tile: pad with massless sources
#pragma unroll
            for (int q = 0; q < DIRECT_TILE / BLOCK; q++) {
                const int j = jt + q * BLOCK + tid;
                float4 v = src[j < je ? j : je - 1];
                if (j >= je) { v.w = 0; if (EQM) v.x = 1e18f; }
                sh[q * BLOCK + tid] = v;
            }
        }
        __syncthreads();
        f2 fx[ITILE / 2], fy[ITILE / 2], fz[ITILE / 2];
#pragma unroll
        for (int p = 0; p < ITILE / 2; p++) { fx[p] = 0; fy[p] = 0; fz[p] = 0; }
#pragma unroll 8
        for (int jj = 0; jj < DIRECT_TILE; jj++) {
            const float4 s = sh[jj];
            const f2 sx = {s.x, s.x}, sy = {s.y, s.y}, sz = {s.z, s.z}, sm = {s.w, s.w};
#pragma unroll
            for (int p = 0; p < ITILE / 2; p++) {
                f2 dx = sx - xi[p], dy = sy - yi[p], dz = sz - zi[p];
                f2 r2 = fma2(dx, dx, fma2(dy, dy, fma2(dz, dz, fl)));
                f2 rinv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                f2 w = EQM ? (rinv * rinv) * rinv : (rinv * rinv) * (rinv * sm);
                fx[p] = fma2(dx, w, fx[p]); fy[p] = fma2(dy, w, fy[p]); fz[p] = fma2(dz, w, fz[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < ITILE / 2; p++) {
            ax[2 * p] += (A)fx[p].x; ay[2 * p] += (A)fy[p].x; az[2 * p] += (A)fz[p].x;
            ax[2 * p + 1] += (A)fx[p].y; ay[2 * p + 1] += (A)fy[p].y; az[2 * p + 1] += (A)fz[p].y;
        }
    }

#pragma unroll
    for (int k = 0; k < ITILE; k++) {
        const int i = ibase + k * BLOCK + tid;
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

// Picks the packed fp32 kernel or the generic one.
template <typename R, typename A, int ITILE, int BLOCK, int OUT> struct DirectKernel {
    static constexpr auto fn(bool) { return &direct_kernel<R, A, ITILE, BLOCK, OUT>; }
};
template <typename A, int ITILE, int BLOCK, int OUT> struct DirectKernel<float, A, ITILE, BLOCK, OUT> {
    static constexpr auto fn(bool eqm)
    {
        return eqm ? &direct_kernel_f32<A, ITILE, BLOCK, OUT, true> : &direct_kernel_f32<A, ITILE, BLOCK, OUT, false>;
    }
};

// How a launch over nt targets and n sources splits the source range across blockIdx.y: aim for >= 16 waves per CU
// when the targets alone are too few.  *jchunk sources (a multiple of the tile) go to each of *jsplit slices.
static void direct_split(size_t n, size_t nt, int block, int itile, unsigned* jsplit_out, size_t* jchunk_out)
{
    const unsigned iblocks = cdiv(nt, (size_t)block * itile);
    const unsigned want_blocks = 4096 * 64 / block;
    unsigned jsplit = 1;
    if (iblocks < want_blocks) jsplit = cdiv(want_blocks, iblocks);
    const unsigned jtiles = cdiv(n, DIRECT_TILE);
    if (jsplit > jtiles) jsplit = jtiles;
    if (jsplit > 64) jsplit = 64;
    if (jsplit < 1) jsplit = 1;
    const size_t jchunk = (size_t)cdiv(cdiv(n, jsplit), DIRECT_TILE) * DIRECT_TILE;
    *jsplit_out = cdiv(n, jchunk);
    *jchunk_out = jchunk;
}

template <typename R, typename A, int ITILE, int BLOCK>
static int launch_direct(nbe_engine* e, const void* src, uint64_t t0, uint64_t t1, bool fused, int src_buf, int dst_buf)
{
    using R4 = typename V4<R>::type;
    using A4 = typename V4<A>::type;
    const size_t n = e->n, nt = (size_t)(t1 - t0);
    if (nt == 0) return NBE_OK;
    if (n > 0x7fffffffull) { nbe_set_error("direct sum supports at most 2^31-1 bodies"); return NBE_E_UNSUPPORTED; }
    const unsigned iblocks = cdiv(nt, BLOCK * ITILE);
    unsigned jsplit;
    size_t jchunk;
    direct_split(n, nt, BLOCK, ITILE, &jsplit, &jchunk);

    const double eps = e->cfg.eps;
    const double tiny = __is_same(R, float) ? 1e-12 : 1e-60;
    const R floor2 = (R)(eps * eps > tiny ? eps * eps : tiny);
    // equal masses (noted at upload): the fp32 pair loop leaves the mass out and G carries it
    const bool eqm = __is_same(R, float) && e->equal_mass > 0.0;
    const A G = (A)(eqm ? e->cfg.G * e->equal_mass : e->cfg.G), dt = (A)e->cfg.dt;
    const A4* pos_in = (const A4*)e->posm[src_buf];
    const A4* vel_in = (const A4*)e->vel[src_buf];
    A4* pos_out = (A4*)e->posm[dst_buf];
    A4* vel_out = (A4*)e->vel[dst_buf];
    float4* mirror_out = (e->cfg.precision == NBE_PREC_MIXED) ? e->mirror[dst_buf] : nullptr;
    A4* acc = (A4*)e->acc;

    if (jsplit == 1) {
        dim3 grid(iblocks, 1);
        if (fused)
            hipLaunchKernelGGL((DirectKernel<R, A, ITILE, BLOCK, OUT_FUSED>::fn(eqm)), grid, dim3(BLOCK), 0, e->stream, (const R4*)src,
                               (int)n, (int)t0, (int)t1, (int)jchunk, floor2, G, dt, acc, pos_in, vel_in, pos_out,
                               vel_out, mirror_out);
        else
            hipLaunchKernelGGL((DirectKernel<R, A, ITILE, BLOCK, OUT_ACC>::fn(eqm)), grid, dim3(BLOCK), 0, e->stream, (const R4*)src,
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
    hipLaunchKernelGGL((DirectKernel<R, A, ITILE, BLOCK, OUT_PARTIAL>::fn(eqm)), grid, dim3(BLOCK), 0, e->stream, (const R4*)src, (int)n,
                       (int)t0, (int)t1, (int)jchunk, floor2, G, dt, (A4*)e->partial, pos_in, vel_in, pos_out, vel_out,
                       mirror_out);
    NBE_HIP(hipGetLastError());
    dim3 fgrid(cdiv(nt, 256));
    const A Gfin = __is_same(R, double) ? G * (A)0.125 : G;      // (the fp64 kernel's partial sums carry a factor 8)
    if (fused)
        hipLaunchKernelGGL((direct_finish_kernel<A, true>), fgrid, dim3(256), 0, e->stream, (const A4*)e->partial,
                           (int)jsplit, (int)t0, (int)t1, Gfin, dt, acc, pos_in, vel_in, pos_out, vel_out, mirror_out);
    else
        hipLaunchKernelGGL((direct_finish_kernel<A, false>), fgrid, dim3(256), 0, e->stream, (const A4*)e->partial,
                           (int)jsplit, (int)t0, (int)t1, Gfin, dt, acc, pos_in, vel_in, pos_out, vel_out, mirror_out);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}

// Forces on targets [t0, t1) of buffer `src_buf` from all n bodies of that buffer.  fused: integrate into
// buffer `dst_buf`; otherwise accelerations go to e->acc at the target's index.
int direct_forces(nbe_engine* e, int src_buf, uint64_t t0, uint64_t t1, bool fused, int dst_buf)
{
    e->stats.interactions = (uint64_t)(t1 - t0) * (uint64_t)(e->n ? e->n - 1 : 0);
    switch (e->cfg.precision) {
    case NBE_PREC_F32: return launch_direct<float, float, DIRECT_ITILE, 64>(e, e->posm[src_buf], t0, t1, fused, src_buf, dst_buf);
    case NBE_PREC_MIXED: return launch_direct<float, double, DIRECT_ITILE, 64>(e, e->mirror[src_buf], t0, t1, fused, src_buf, dst_buf);
    case NBE_PREC_F64: return launch_direct<double, double, 2, 256>(e, e->posm[src_buf], t0, t1, fused, src_buf, dst_buf);
    }
    nbe_set_error("unknown precision %d", e->cfg.precision);
    return NBE_E_INVALID;
}

// ---- force pass for callers that cannot allocate ------------------------------------------------------------
// nbe_accel_direct_noalloc: the fp32 direct-sum force pass of nbe_accel on caller-owned device arrays, for callers
// that must not allocate (HIP graph capture, serving loops).  The only memory the pass needs beyond its input and
// output is the array of partial sums that a launch with few targets spreads over slices of the source range
// (direct_split); it lives in a fixed array in device memory instead of a hipMalloc'd buffer.  Every GPU the library
// is used on holds its own copy of the array until the process ends.
#define NOALLOC_MAX_BODIES 262144u
// slices * targets <= 4096 * 256 + targets for any body count (direct_split), hence this many partial sums at most
#define NOALLOC_SCRATCH_ELEMS ((1u << 20) + NOALLOC_MAX_BODIES)
static __device__ __align__(256) float4 scratch[NOALLOC_SCRATCH_ELEMS];

// Device address of `scratch` on each GPU, looked up once per device with that device current.  In host code the
// array's name is only a handle for hipGetSymbolAddress: it must never be handed to a kernel or a memset.
#define NOALLOC_MAX_DEVICES 64
static float4* g_scratch_ptr[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_for_device(int device, float4** out)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        NBE_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { nbe_set_error("hipGetSymbolAddress returned no address for the scratch array"); return NBE_E_DEVICE; }
        g_scratch_ptr[device] = (float4*)p;
    }
    *out = g_scratch_ptr[device];
    return NBE_OK;
}

extern "C" int nbe_accel_direct_noalloc(int device, void* hip_stream, size_t n, const float* d_pos_mass, float* d_acc,
                                        double G, double eps)
{
    if (n == 0) return NBE_OK;
    if (!d_pos_mass || !d_acc) { nbe_set_error("nbe_accel_direct_noalloc: NULL device array"); return NBE_E_INVALID; }
    if (n > NOALLOC_MAX_BODIES) {
        nbe_set_error("nbe_accel_direct_noalloc takes at most %u bodies (got %zu): its partial sums live in a fixed %zu-byte "
                      "device array; use an engine (nbe_create, nbe_accel) for more",
                      NOALLOC_MAX_BODIES, n, sizeof(float4) * (size_t)NOALLOC_SCRATCH_ELEMS);
        return NBE_E_UNSUPPORTED;
    }
    if (!(eps >= 0.0) || !std::isfinite(G)) { nbe_set_error("nbe_accel_direct_noalloc: eps must be >= 0 and G finite"); return NBE_E_INVALID; }
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) { nbe_set_error("device %d out of range", device); return NBE_E_INVALID; }
    // the caller's current device is put back on every way out
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    int prev = -1;
    NBE_HIP(hipGetDevice(&prev));
    if (prev != device) {
        NBE_HIP(hipSetDevice(device));
        guard.prev = prev;
    }
    float4* part = nullptr;
    NBE_TRY(scratch_for_device(device, &part));
    constexpr int ITILE = DIRECT_ITILE, BLOCK = 64;
    unsigned jsplit;
    size_t jchunk;
    direct_split(n, n, BLOCK, ITILE, &jsplit, &jchunk);
    if ((size_t)jsplit * n > NOALLOC_SCRATCH_ELEMS) { nbe_set_error("internal: %u slices of %zu targets exceed the scratch array", jsplit, n); return NBE_E_INVALID; }
    const float floor2 = (float)(eps * eps > 1e-12 ? eps * eps : 1e-12);
    const float Gf = (float)G;
    hipStream_t stream = (hipStream_t)hip_stream;
    const float4* src = (const float4*)d_pos_mass;
    float4* acc = (float4*)d_acc;
    const unsigned iblocks = cdiv(n, (size_t)BLOCK * ITILE);
    if (jsplit == 1) {
        hipLaunchKernelGGL((direct_kernel_f32<float, ITILE, BLOCK, OUT_ACC, false>), dim3(iblocks, 1), dim3(BLOCK), 0, stream, src,
                           (int)n, 0, (int)n, (int)jchunk, floor2, Gf, 0.0f, acc, (const float4*)nullptr, (const float4*)nullptr,
                           (float4*)nullptr, (float4*)nullptr, (float4*)nullptr);
        NBE_HIP(hipGetLastError());
        return NBE_OK;
    }
    // every partial sum [slice][target] is written by the first launch before the second reads it, so what an
    // earlier call left in the array never shows
    hipLaunchKernelGGL((direct_kernel_f32<float, ITILE, BLOCK, OUT_PARTIAL, false>), dim3(iblocks, jsplit), dim3(BLOCK), 0, stream,
                       src, (int)n, 0, (int)n, (int)jchunk, floor2, Gf, 0.0f, part, (const float4*)nullptr, (const float4*)nullptr,
                       (float4*)nullptr, (float4*)nullptr, (float4*)nullptr);
    NBE_HIP(hipGetLastError());
    hipLaunchKernelGGL((direct_finish_kernel<float, false>), dim3(cdiv(n, 256)), dim3(256), 0, stream, (const float4*)part,
                       (int)jsplit, 0, (int)n, Gf, 0.0f, acc, (const float4*)nullptr, (const float4*)nullptr, (float4*)nullptr,
                       (float4*)nullptr, (float4*)nullptr);
    NBE_HIP(hipGetLastError());
    return NBE_OK;
}
