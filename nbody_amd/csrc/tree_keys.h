// Morton-key helpers shared by the tree build and the collision search.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "engine_internal.h"

// A key interleaves the 21-bit fixed-point cell coordinates of a body: bit 3k = X bit k, 3k+1 = Y, 3k+2 = Z, the
// child order of octantBits (tree.go:128-132).  The 3-bit digit of depth L (1..21) is (key >> 3*(21-L)) & 7.
__device__ __forceinline__ uint64_t spread3(uint32_t v)   // 21 bits -> every third bit of 63
{
    uint64_t x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint32_t compact3(uint64_t x)  // every third bit of 63 -> 21 bits
{
    x &= 0x1249249249249249ull;
    x = (x | x >> 2) & 0x10c30c30c30c30c3ull;
    x = (x | x >> 4) & 0x100f00f00f00f00full;
    x = (x | x >> 8) & 0x1f0000ff0000ffull;
    x = (x | x >> 16) & 0x1f00000000ffffull;
    x = (x | x >> 32) & 0x1fffffull;
    return (uint32_t)x;
}

#define KEY_DEAD 0xffffffffffffffffull

// delta(i, j): length of the common prefix of keys i and j (keys occupy the low 63 bits), with the index as
// tie-break for equal keys; -1 outside [0, na).
__device__ __forceinline__ int key_delta(const uint64_t* __restrict__ keys, int na, int i, uint64_t ki, int j)
{
    if (j < 0 || j >= na) return -1;
    const uint64_t kj = keys[j];
    if (ki != kj) return __clzll((long long)(ki ^ kj)) - 1;
    return 63 + __clz(i ^ j);
}

// Depth of the cell two keys with a common prefix of d bits share.
__device__ __forceinline__ int delta_level(int d) { int l = d / 3; return l > 21 ? 21 : l; }

// Index q in [0, 2^21) of the depth-21 cell with lo + q*cell <= x < lo + (q+1)*cell.  x - lo may round
// across a cell boundary, so the estimate is corrected against the boundaries themselves, which are exact
// for a power-of-two bound: the result then equals the chain of sign tests of octantBits (tree.go:128-132).
__device__ __forceinline__ uint32_t cell_index(double x, double lo, double cell)
{
    double q = floor((x - lo) / cell);
    q = fmin(fmax(q, 0.0), 2097151.0);
    if (x < lo + q * cell && q > 0.0) q -= 1.0;
    else if (x >= lo + (q + 1.0) * cell && q < 2097151.0) q += 1.0;
    return (uint32_t)q;
}

// Key of one body: KEY_DEAD unless lo <= p < hi on every axis.
__device__ __forceinline__ uint64_t body_key(double x, double y, double z, double lox, double loy, double loz, double width)
{
    const bool in = (lox <= x && x < lox + width) && (loy <= y && y < loy + width) && (loz <= z && z < loz + width);
    if (!in) return KEY_DEAD;
    const double cell = width * (1.0 / 2097152.0);    // 2^21 cells per axis at depth 21
    return spread3(cell_index(x, lox, cell)) | (spread3(cell_index(y, loy, cell)) << 1) | (spread3(cell_index(z, loz, cell)) << 2);
}


// fp32 nodes store (edge / theta)^2 for the walk's one-compare opening test; theta = 0 ("always open") maps to
// the largest float.  fp64 nodes store the edge itself.
// With softening the walk's r^2 carries eps^2 (it is the r^2 of the force); the stored threshold of a cell carries it too,
// so that the test still compares the bare distance with the edge and costs no subtraction per visit.
template <typename NodeT> __device__ __forceinline__ void set_edge(NodeT& nd, double w, double theta, double eps2);
template <> __device__ __forceinline__ void set_edge<NodeF>(NodeF& nd, double w, double theta, double eps2)
{
    const double q = theta > 0.0 ? (w / theta) * (w / theta) : (w > 0.0 ? 3.0e38 : 0.0);
    nd.w2 = (float)fmin(w > 0.0 ? q + eps2 : q, 3.0e38);
}
template <> __device__ __forceinline__ void set_edge<NodeD>(NodeD& nd, double w, double theta, double)
{
    nd.w = w;
    // (edge / theta)^2, what the walk compares r^2 with before it consults the reference's own expression; theta = 0
    // ("always open") maps to infinity for a cell and to 0 for a leaf, which never opens
    nd.w2 = theta > 0.0 ? (w / theta) * (w / theta) : (w > 0.0 ? (double)INFINITY : 0.0);
}

