// Morton-key helpers shared by the tree build and the collision search.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

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
