// Device helpers shared by the force kernels.
#pragma once
#include <hip/hip_runtime.h>

template <typename R> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

__device__ __forceinline__ float rsq_dev(float x) { return __builtin_amdgcn_rsqf(x); }
// v_rsq_f64 is an approximation good to about half the mantissa; one Newton step brings it to fp64 rounding level
// (the fp64 parity tests hold at 1e-12 with one step as they did with two; the direct sum gains 14 %).
__device__ __forceinline__ double rsq_dev(double x)
{
    // written with an explicit fma in the same shape as rsq2_dev below, so that "exactly half of rsq2_dev" does not
    // depend on what the compiler contracts
    const double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    return y * fma(-(0.5 * t), y, 1.5);
}

// 2 / sqrt(x): the same Newton step written as y (3 - x y^2), which leaves out the halvings of y (1.5 - 0.5 x y^2):
// three fp64 instructions behind v_rsq_f64 instead of four.  Scaling by two is exact at every step, so the result is
// exactly twice rsq_dev's; a kernel that sums dx * m * rinv^3 with it carries a factor of exactly 8 and multiplies
// its sum by G / 8 at the end: the same bits as with rsq_dev, one instruction less per interaction.
__device__ __forceinline__ double rsq2_dev(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    return y * fma(-t, y, 3.0);
}

// What a force kernel does with a target's acceleration sum.
enum { OUT_ACC = 0,       // store G*sum to the acceleration buffer
       OUT_FUSED = 1,     // integrate the body (kick_drift) into the next-state buffers
       OUT_PARTIAL = 2 }; // store the raw sum of one source range (direct sum only)

// physics.go:101-115 for one body with a = f/m already formed: v += a*dt; x += v*dt.
template <typename A>
__device__ __forceinline__ void kick_drift(int i, A ax, A ay, A az, A dt,
                                           const typename V4<A>::type* __restrict__ pos_in,
                                           const typename V4<A>::type* __restrict__ vel_in,
                                           typename V4<A>::type* __restrict__ pos_out,
                                           typename V4<A>::type* __restrict__ vel_out,
                                           float4* __restrict__ mirror_out)
{
    typename V4<A>::type p = pos_in[i], v = vel_in[i];
    v.x += ax * dt; v.y += ay * dt; v.z += az * dt;
    p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
    pos_out[i] = p;
    vel_out[i] = v;
    if (mirror_out) mirror_out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}
