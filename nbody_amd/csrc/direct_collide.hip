// Collisions inside the direct pair sweep (main.go:107-128 with collisions on).
//
// In the reference a pair that touches merges on the spot, in the middle of the i < j sweep: from then on body i
// carries the summed mass and the grown radius, and the absorbed body j is skipped by every later pair.  Forces
// are accumulated as forces (G*mi*mj/r^2) and divided by the mass only in update(), so a body that absorbs another
// halfway through its row keeps the smaller force terms of the pairs visited before.  None of that can be folded
// into the tiled direct-sum kernel, and it does not have to be: a step either has a touching pair or it has none.
//   * `touch_detect_kernel` tests every pair i < j of live bodies for r <= Ri + Rj (fp64, all pairs, tiled through
//     LDS) and raises one flag;
//   * no flag: the step is the ordinary fused direct-sum step (nothing in it can differ from the reference);
//   * flag: `direct_collide_step_kernel` replays the reference's sweep for this one step with its exact order of
//     events.  One workgroup walks the rows i in order; within a row all threads share the work: find the first
//     live j that touches i (block minimum), apply gravity both ways to the pairs up to and including that j with
//     the masses as they are at that moment (physics.go:135-145), combine (physics.go:153-162), go on behind j
//     with the new mass and radius.  Then update (physics.go:101-115).  Everything in fp64.
// An absorbed body stays in the arrays with mass 0 and radius -infinity, parked far away at a position of its own,
// so the fast kernels need no alive test; it is flagged in `dead` and left out of downloads and frames.
#include "engine_internal.h"

#define DC_THREADS 1024
#define DC_PARK 1.0e9        // parked bodies sit at (DC_PARK + 128*slot, DC_PARK, DC_PARK)

template <typename S4>
__global__ __launch_bounds__(256) void touch_detect_kernel(const S4* __restrict__ posm, const double* __restrict__ radius, int n,
                                                           int* __restrict__ flag)
{
    __shared__ double4 sh[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double xi = 0, yi = 0, zi = 0, ri = -INFINITY;
    if (i < n) { const S4 p = posm[i]; xi = (double)p.x; yi = (double)p.y; zi = (double)p.z; ri = radius[i]; }
    bool hit = false;
    for (int jt = blockIdx.x * 256; jt < n; jt += 256) {        // pairs i < j only: tiles from this block's own on
        __syncthreads();
        {
            const int j = jt + threadIdx.x;
            double4 s; s.x = 0; s.y = 0; s.z = 0; s.w = -INFINITY;
            if (j < n) { const S4 q = posm[j]; s.x = (double)q.x; s.y = (double)q.y; s.z = (double)q.z; s.w = radius[j]; }
            sh[threadIdx.x] = s;
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < 256; jj++) {
            const double4 s = sh[jj];
            const double rs = ri + s.w;                          // -infinity as soon as one of the two is absorbed
            const double dx = s.x - xi, dy = s.y - yi, dz = s.z - zi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            // a hair generous: the replay decides with the reference's own arithmetic
            hit = hit || (jt + jj > i && rs >= 0.0 && r2 <= rs * rs * (1.0 + 1e-12) + 1e-300);
        }
    }
    if (__syncthreads_or(hit) && threadIdx.x == 0) atomicOr(flag, 1);
}

__device__ __forceinline__ int block_min_int(int v, int* scratch)
{
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_down(v, d, 64); v = o < v ? o : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    int m = scratch[0];
#pragma unroll
    for (int w = 1; w < DC_THREADS / 64; w++) m = scratch[w] < m ? scratch[w] : m;
    return m;
}

__device__ __forceinline__ void block_sum3(double& x, double& y, double& z, double* scratch)
{
    for (int d = 32; d >= 1; d >>= 1) { x += __shfl_down(x, d, 64); y += __shfl_down(y, d, 64); z += __shfl_down(z, d, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        scratch[3 * (threadIdx.x >> 6)] = x; scratch[3 * (threadIdx.x >> 6) + 1] = y; scratch[3 * (threadIdx.x >> 6) + 2] = z;
    }
    __syncthreads();
    x = 0; y = 0; z = 0;
#pragma unroll
    for (int w = 0; w < DC_THREADS / 64; w++) { x += scratch[3 * w]; y += scratch[3 * w + 1]; z += scratch[3 * w + 2]; }
}

// dist (physics.go:123-128) without fused multiply-adds.
__device__ __forceinline__ double dist_rn(double dx, double dy, double dz)
{
    return __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

// One step of main.go:107-128 + 171-177 with collisions on, from the state in (posm, vel, radius, dead) into
// (posm_out, vel_out); radius and dead are updated in place.  counters[1] += merges.
template <typename S4, typename S>
__global__ __launch_bounds__(DC_THREADS) void direct_collide_step_kernel(S4* __restrict__ posm, S4* __restrict__ vel,
                                                                         double* __restrict__ radius, uint8_t* __restrict__ dead,
                                                                         double4* __restrict__ force, int n, double G, double dt,
                                                                         S4* __restrict__ posm_out, S4* __restrict__ vel_out,
                                                                         float4* __restrict__ mirror_out, int* __restrict__ counters)
{
    __shared__ int iscratch[DC_THREADS / 64];
    __shared__ double dscratch[3 * DC_THREADS / 64];
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += DC_THREADS) force[j] = make_double4(0, 0, 0, 0);
    __syncthreads();
    int merges = 0;
    for (int i = 0; i < n - 1; i++) {
        if (dead[i]) continue;                                   // same flag for every thread
        const S4 pi = posm[i], vi4 = vel[i];
        const double xi = (double)pi.x, yi = (double)pi.y, zi = (double)pi.z;
        double mi = (double)pi.w, ri = radius[i];
        double vx = (double)vi4.x, vy = (double)vi4.y, vz = (double)vi4.z;
        const double4 f0 = force[i];                             // what earlier rows put on body i
        double fx = f0.x, fy = f0.y, fz = f0.z;
        bool changed = false;
        int jstart = i + 1;
        for (;;) {
            // the first live body behind jstart that touches i as it is now (main.go:123)
            int jc = n;
            for (int j = jstart + tid; j < n; j += DC_THREADS) {
                if (dead[j]) continue;
                const S4 q = posm[j];
                const double r = dist_rn((double)q.x - xi, (double)q.y - yi, (double)q.z - zi);
                if (r <= ri + radius[j]) { jc = j; break; }
            }
            jc = block_min_int(jc, iscratch);
            // gravity both ways for the pairs (i, j), jstart <= j <= jc, with the masses of this moment (main.go:118-120)
            const int jend = jc < n ? jc : n - 1;
            double px = 0, py = 0, pz = 0;
            for (int j = jstart + tid; j <= jend; j += DC_THREADS) {
                if (dead[j]) continue;
                const S4 q = posm[j];
                const double dx = (double)q.x - xi, dy = (double)q.y - yi, dz = (double)q.z - zi;
                const double r = dist_rn(dx, dy, dz);
                if (r == 0.0) continue;                          // coincident bodies exert no force (the reference: NaN)
                const double f = G * (mi * (double)q.w) / (r * r);   // physics.go:136
                const double ux = dx / r * f, uy = dy / r * f, uz = dz / r * f;
                px += ux; py += uy; pz += uz;
                double4 fj = force[j];
                fj.x -= ux; fj.y -= uy; fj.z -= uz;
                force[j] = fj;
            }
            block_sum3(px, py, pz, dscratch);
            fx += px; fy += py; fz += pz;
            if (jc >= n) break;
            // combine(i, jc), physics.go:153-162: the mass is summed first and then used as `ma`
            const S4 qj = posm[jc], vj = vel[jc];
            const double mj = (double)qj.w, rj = radius[jc];
            const double4 fj = force[jc];
            mi = mi + mj;
            ri = cbrt(ri * ri * ri + rj * rj * rj);
            vx = (mi * vx + mj * (double)vj.x) / (mi + mj);
            vy = (mi * vy + mj * (double)vj.y) / (mi + mj);
            vz = (mi * vz + mj * (double)vj.z) / (mi + mj);
            fx += fj.x; fy += fj.y; fz += fj.z;
            changed = true;
            merges++;
            __syncthreads();                                     // everyone has read body jc
            if (tid == 0) dead[jc] = 1;                          // main.go:125
            __syncthreads();
            jstart = jc + 1;
        }
        if (tid == 0) {
            force[i] = make_double4(fx, fy, fz, 0);
            if (changed) {
                S4 p = pi; p.w = (S)mi; posm[i] = p;
                S4 v = vi4; v.x = (S)vx; v.y = (S)vy; v.z = (S)vz; vel[i] = v;
                radius[i] = ri;
            }
        }
        __syncthreads();
    }
    // update (main.go:171-177, physics.go:101-115); absorbed bodies are parked
    for (int j = tid; j < n; j += DC_THREADS) {
        S4 p = posm[j], v = vel[j];
        if (dead[j]) {
            p.x = (S)(DC_PARK + 128.0 * j); p.y = (S)DC_PARK; p.z = (S)DC_PARK; p.w = 0;
            v.x = 0; v.y = 0; v.z = 0;
            radius[j] = -INFINITY;
        } else {
            const double4 f = force[j];
            const double m = (double)p.w;
            const double nvx = (double)v.x + f.x / m * dt, nvy = (double)v.y + f.y / m * dt, nvz = (double)v.z + f.z / m * dt;
            v.x = (S)nvx; v.y = (S)nvy; v.z = (S)nvz;
            p.x = (S)((double)p.x + nvx * dt); p.y = (S)((double)p.y + nvy * dt); p.z = (S)((double)p.z + nvz * dt);
        }
        posm_out[j] = p;
        vel_out[j] = v;
        if (mirror_out) mirror_out[j] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
    }
    if (tid == 0) counters[1] += merges;
}

// ---- host side ---------------------------------------------------------------------------------------------
int direct_collide_reserve(nbe_engine* e)
{
    const size_t cap = e->n > 0 ? e->n : 1;
    if (e->dc_cap >= cap) return NBE_OK;
    if (e->dc_force) { NBE_HIP(hipFree(e->dc_force)); e->dc_force = nullptr; }
    if (e->dc_counters) { NBE_HIP(hipFree(e->dc_counters)); e->dc_counters = nullptr; }
    NBE_HIP(hipMalloc((void**)&e->dc_force, cap * sizeof(double4)));
    NBE_HIP(hipMalloc((void**)&e->dc_counters, 8 * sizeof(int)));
    NBE_HIP(hipMemsetAsync(e->dc_counters, 0, 8 * sizeof(int), e->stream));
    e->dc_cap = cap;
    return NBE_OK;
}

int direct_collide_free(nbe_engine* e)
{
    if (e->dc_force) { NBE_HIP(hipFree(e->dc_force)); e->dc_force = nullptr; }
    if (e->dc_counters) { NBE_HIP(hipFree(e->dc_counters)); e->dc_counters = nullptr; }
    e->dc_cap = 0;
    return NBE_OK;
}

template <typename S4, typename S>
static int direct_collide_step_t(nbe_engine* e, bool* replayed)
{
    const int n = (int)e->n;
    const int cur = e->cur, nxt = cur ^ 1;
    const bool mixed = e->cfg.precision == NBE_PREC_MIXED;
    *replayed = false;
    NBE_HIP(hipMemsetAsync(e->dc_counters, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL((touch_detect_kernel<S4>), dim3(cdiv(n, 256)), dim3(256), 0, e->stream, (const S4*)e->posm[cur],
                       e->radius[0], n, e->dc_counters);
    NBE_HIP(hipGetLastError());
    int c[2] = {0, 0};
    NBE_HIP(hipMemcpyAsync(c, e->dc_counters, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    e->stats.collision_pairs = 0;
    if (!c[0]) return NBE_OK;
    NBE_TRY(frame_stream_guard_write(e, cur));     // the replay merges in place before it writes the next buffer
    NBE_TRY(frame_stream_guard_write(e, nxt));
    hipLaunchKernelGGL((direct_collide_step_kernel<S4, S>), dim3(1), dim3(DC_THREADS), 0, e->stream, (S4*)e->posm[cur],
                       (S4*)e->vel[cur], e->radius[0], e->cw.dead, e->dc_force, n, e->cfg.G, e->cfg.dt, (S4*)e->posm[nxt],
                       (S4*)e->vel[nxt], mixed ? e->mirror[nxt] : (float4*)nullptr, e->dc_counters);
    NBE_HIP(hipGetLastError());
    NBE_HIP(hipMemcpyAsync(c, e->dc_counters, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    e->stats.collision_pairs = (uint64_t)(c[1] - (int)e->direct_absorbed);   // merges of this step
    e->direct_absorbed = (size_t)c[1];
    *replayed = true;
    return NBE_OK;
}

// Direct mode with collisions on: looks for a touching pair; if there is one, replays the reference's sweep for
// this step (state of e->cur into e->cur^1, *replayed = true); otherwise leaves the step to the fused kernel.
int direct_collide_step(nbe_engine* e, bool* replayed)
{
    if (e->state64) return direct_collide_step_t<double4, double>(e, replayed);
    return direct_collide_step_t<float4, float>(e, replayed);
}
