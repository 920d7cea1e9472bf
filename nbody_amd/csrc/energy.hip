// Energy and momentum diagnostics (fp64): kinetic energy, linear momentum and the pairwise potential
// -G * sum_{i<j} mi*mj / sqrt(r^2 + eps^2) of the live bodies.  The reference has no such diagnostic; these
// sums exist to bound the integrator's drift.
#include "engine_internal.h"
#include "kernels_common.h"

__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double ws[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    return r;
}

// out[0] = sum 0.5 m v^2, out[1..3] = sum m v
template <typename S4>
__global__ __launch_bounds__(256) void kinetic_kernel(const S4* __restrict__ posm, const S4* __restrict__ vel, int n_live,
                                                      double* __restrict__ out)
{
    double ke = 0, px = 0, py = 0, pz = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_live; i += gridDim.x * 256) {
        const double m = (double)posm[i].w;
        const double vx = (double)vel[i].x, vy = (double)vel[i].y, vz = (double)vel[i].z;
        ke += 0.5 * m * (vx * vx + vy * vy + vz * vz);
        px += m * vx; py += m * vy; pz += m * vz;
    }
    ke = block_sum(ke); px = block_sum(px); py = block_sum(py); pz = block_sum(pz);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], ke); atomicAdd(&out[1], px); atomicAdd(&out[2], py); atomicAdd(&out[3], pz);
    }
}

// out[4] += sum_i m_i * sum_{j != i} m_j / sqrt(r_ij^2 + eps^2)   (twice the pair sum)
template <typename S4>
__global__ __launch_bounds__(256) void potential_kernel(const S4* __restrict__ posm, int n_live, double eps2,
                                                        double* __restrict__ out)
{
    __shared__ double4 sh[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = i < n_live ? i : n_live - 1;
    const S4 pi = posm[ic];
    const double xi = (double)pi.x, yi = (double)pi.y, zi = (double)pi.z;
    double phi = 0;
    for (int jt = 0; jt < n_live; jt += 256) {
        __syncthreads();
        {
            const int j = jt + threadIdx.x;
            double4 s; s.x = 0; s.y = 0; s.z = 0; s.w = 0;
            if (j < n_live) { S4 q = posm[j]; s.x = (double)q.x; s.y = (double)q.y; s.z = (double)q.z; s.w = (double)q.w; }
            sh[threadIdx.x] = s;
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < 256; jj++) {
            const double4 s = sh[jj];
            const double dx = s.x - xi, dy = s.y - yi, dz = s.z - zi;
            const double r2 = dx * dx + dy * dy + dz * dz + eps2;
            const bool other = (jt + jj != ic) && r2 > 0.0;
            phi += other ? s.w * rsq_dev(r2) : 0.0;
        }
    }
    double v = i < n_live ? (double)pi.w * phi : 0.0;
    v = block_sum(v);
    if (threadIdx.x == 0) atomicAdd(&out[4], v);
}

// out[4] += sum_i m_i * phi_i with phi_i taken from the w component of the acceleration records that an
// instrumented tree walk left in e->acc (device order).
template <typename S4>
__global__ __launch_bounds__(256) void tree_potential_sum_kernel(const S4* __restrict__ posm, const S4* __restrict__ acc, int n_live,
                                                                 double* __restrict__ out)
{
    double v = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_live; i += gridDim.x * 256) v += (double)posm[i].w * (double)acc[i].w;
    v = block_sum(v);
    if (threadIdx.x == 0) atomicAdd(&out[4], v);
}

// Bodies up to which the potential is the exact fp64 pair sum; above it tree mode sums the potential over the
// nodes the walk accepts (same theta as the forces), and direct mode refuses.
#define ENERGY_PAIR_SUM_MAX 262144

template <typename S4>
static int energy_t(nbe_engine* e, int n_live, bool tree_potential, double* host5)
{
    if (!e->d_energy) NBE_HIP(hipMalloc((void**)&e->d_energy, 8 * sizeof(double)));
    NBE_HIP(hipMemsetAsync(e->d_energy, 0, 8 * sizeof(double), e->stream));
    if (n_live > 0) {
        unsigned kb = cdiv(n_live, 256);
        if (kb > 1024) kb = 1024;
        hipLaunchKernelGGL((kinetic_kernel<S4>), dim3(kb), dim3(256), 0, e->stream, (const S4*)e->posm[e->cur],
                           (const S4*)e->vel[e->cur], n_live, e->d_energy);
        if (tree_potential)
            hipLaunchKernelGGL((tree_potential_sum_kernel<S4>), dim3(kb), dim3(256), 0, e->stream, (const S4*)e->posm[e->cur],
                               (const S4*)e->acc, n_live, e->d_energy);
        else
            hipLaunchKernelGGL((potential_kernel<S4>), dim3(cdiv(n_live, 256)), dim3(256), 0, e->stream,
                               (const S4*)e->posm[e->cur], n_live, e->cfg.eps * e->cfg.eps, e->d_energy);
        NBE_HIP(hipGetLastError());
    }
    NBE_HIP(hipMemcpyAsync(host5, e->d_energy, 5 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    return NBE_OK;
}

// h[0] = kinetic energy, h[1..3] = momentum, h[4] = sum m_i phi_i of the first n_live bodies of the current state, phi
// from e->acc (tree_potential) or from the fp64 pair sum over those bodies.
int energy_local_sums(nbe_engine* e, int n_live, bool tree_potential, double h[5])
{
    return e->state64 ? energy_t<double4>(e, n_live, tree_potential, h) : energy_t<float4>(e, n_live, tree_potential, h);
}

int energy_compute(nbe_engine* e, double* ke, double* pe, double mom[3])
{
    const bool tree_potential = e->cfg.mode == NBE_MODE_TREE && e->n > ENERGY_PAIR_SUM_MAX;
    if (!tree_potential && e->n > ENERGY_PAIR_SUM_MAX * 16) {
        nbe_set_error("nbe_energy in direct mode sums all pairs in fp64 and is limited to %d bodies", ENERGY_PAIR_SUM_MAX * 16);
        return NBE_E_UNSUPPORTED;
    }
    // Tree mode: the sums are over the bodies the coming step will move, i.e. those inside the bound (tree.go:290-292); the
    // build finds them (and the step reuses it).
    if (e->cfg.mode == NBE_MODE_TREE && !e->prepared && e->n > 0) { NBE_TRY(tree_sort_and_build(e)); e->prepared = true; }
    if (tree_potential) {
        // instrumented walk over the current state: leaves each body's potential in e->acc[i].w
        NBE_TRY(tree_forces(e, e->cur, 0, e->n, false, true, e->cur ^ 1));
    }
    int n_live = (int)e->n;
    if (e->have_counts) {
        int c[4];
        NBE_HIP(hipMemcpyAsync(c, e->tw.counters, sizeof(c), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
        n_live = c[0];
    }
    double h[5] = {0, 0, 0, 0, 0};
    if (e->state64) NBE_TRY((energy_t<double4>(e, n_live, tree_potential, h)));
    else NBE_TRY((energy_t<float4>(e, n_live, tree_potential, h)));
    if (ke) *ke = h[0];
    if (mom) { mom[0] = h[1]; mom[1] = h[2]; mom[2] = h[3]; }
    if (pe) *pe = -0.5 * e->cfg.G * h[4];
    return NBE_OK;
}
