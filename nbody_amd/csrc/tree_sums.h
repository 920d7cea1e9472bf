// 128-bit fixed-point sums of (m, m*x, m*y, m*z): the arithmetic behind NBE_SUMS_EXACT, shared by the tree build and the
// key-range-distributed tree.
#pragma once
#include <hip/hip_runtime.h>

// NBE_SUMS_EXACT.  Every body's (m, m*x, m*y, m*z) is truncated to a multiple of a quantum that is fixed for the run
// (2^-shift_m for the mass, 2^-shift_x for the moments; chosen from the total mass and the bound so that no sum can
// reach 2^126 quanta) and summed as 128-bit integers.  Integer addition is associative, so a cell's sums are the same
// bits whatever the order of addition and whatever else is sorted before the cell: prefix-sum differences (three
// launches) then carry exactly the sums of any partition of the bodies, which is what the key-range-distributed build
// needs of them.  A body's quantum is 2^-126 of (total mass x bound): far below the rounding of its own fp64 product.
struct U128 { unsigned long long lo, hi; };
struct Q4 { U128 m, x, y, z; };
static_assert(sizeof(Q4) == 64, "Q4 must be 64 bytes");

__host__ __device__ __forceinline__ U128 u128_add(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}
__host__ __device__ __forceinline__ U128 u128_sub(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1ull : 0ull);
    return r;
}
__host__ __device__ __forceinline__ U128 u128_neg(U128 a)
{
    U128 r;
    r.lo = ~a.lo + 1ull;
    r.hi = ~a.hi + (r.lo == 0ull ? 1ull : 0ull);
    return r;
}
__host__ __device__ __forceinline__ Q4 q4_zero() { return Q4{{0, 0}, {0, 0}, {0, 0}, {0, 0}}; }
__host__ __device__ __forceinline__ Q4 q4_add(const Q4& a, const Q4& b) { return Q4{u128_add(a.m, b.m), u128_add(a.x, b.x), u128_add(a.y, b.y), u128_add(a.z, b.z)}; }
__host__ __device__ __forceinline__ Q4 q4_sub(const Q4& a, const Q4& b) { return Q4{u128_sub(a.m, b.m), u128_sub(a.x, b.x), u128_sub(a.y, b.y), u128_sub(a.z, b.z)}; }
__device__ __forceinline__ U128 u128_shfl_up(U128 v, int d) { return U128{__shfl_up(v.lo, d, 64), __shfl_up(v.hi, d, 64)}; }
__device__ __forceinline__ Q4 q4_shfl_up(const Q4& v, int d) { return Q4{u128_shfl_up(v.m, d), u128_shfl_up(v.x, d), u128_shfl_up(v.y, d), u128_shfl_up(v.z, d)}; }

// trunc(|v| * 2^shift) with the sign put back (two's complement); needs |v| * 2^shift < 2^127.  Every step is exact: the
// scaling, the truncation, the split into words (the low word is the low part of a 53-bit integer).
__device__ __forceinline__ U128 q_from_double(double v, int shift)
{
    const double a = trunc(ldexp(fabs(v), shift));
    const double hi_d = floor(a * 0x1p-64);
    const double lo_d = a - hi_d * 0x1p64;
    U128 r;
    r.hi = (unsigned long long)hi_d;
    r.lo = (unsigned long long)lo_d;
    return v < 0.0 ? u128_neg(r) : r;
}
// The nearest-or-so double of q * 2^-shift (two conversions and one fused rounding; every engine converts alike).
__device__ __forceinline__ double q_to_double(U128 q, int shift)
{
    const bool neg = (long long)q.hi < 0;
    if (neg) q = u128_neg(q);
    const double d = fma((double)q.hi, 0x1p64, (double)q.lo);
    const double r = ldexp(d, -shift);
    return neg ? -r : r;
}

// Mass and centre of mass of a cell from its exact sums (the same expressions wherever a cell is made: here, and for the
// cells that span the ranks of a distributed tree).
template <typename NodeT> __device__ __forceinline__ void exact_cell(const Q4& q, int shift_m, int shift_x, NodeT& nd)
{
    const double m = q_to_double(q.m, shift_m);
    const double inv = m != 0.0 ? 1.0 / m : 0.0;
    nd.x = q_to_double(q.x, shift_x) * inv;
    nd.y = q_to_double(q.y, shift_x) * inv;
    nd.z = q_to_double(q.z, shift_x) * inv;
    nd.m = m;
}

