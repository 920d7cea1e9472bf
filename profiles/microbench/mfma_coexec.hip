// Microbenchmark 3 (gfx950): does an f32 MFMA run beside packed f32 VALU work, and what does it cost the VALU stream?
//   part 1  one instruction stream per wave: per loop step ONE mfma plus K independent v_pk_fma_f32 (K = 0..16),
//           for the three f32 shapes; cycles per step against the K packed operations alone
//   part 2  two kinds of waves on one SIMD: in a 512-thread workgroup (two waves per SIMD) waves 0-3 issue only
//           MFMAs, waves 4-7 only packed VALU (the loop of the direct-sum kernel is all packed VALU + v_rsq_f32);
//           each role's own time (s_memtime) against the same role with the partner idle
// Build: hipcc -O3 --offload-arch=gfx950 mfma_coexec.hip -o mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define PK1(r) "v_pk_fma_f32 %" #r ", %" #r ", %8, %" #r "\n"
#define PKARGS : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(pm)

template <int K> __device__ __forceinline__ void valu_block(f2& p0, f2& p1, f2& p2, f2& p3, f2& p4, f2& p5, f2& p6, f2& p7, f2 pm)
{
    if constexpr (K >= 8) { asm volatile(PK1(0) PK1(1) PK1(2) PK1(3) PK1(4) PK1(5) PK1(6) PK1(7) PKARGS); valu_block<K - 8>(p0, p1, p2, p3, p4, p5, p6, p7, pm); }
    else if constexpr (K >= 4) { asm volatile(PK1(0) PK1(1) PK1(2) PK1(3) PKARGS); valu_block<K - 4>(p4, p5, p6, p7, p0, p1, p2, p3, pm); }
    else if constexpr (K >= 2) { asm volatile(PK1(0) PK1(1) PKARGS); valu_block<K - 2>(p2, p3, p0, p1, p6, p7, p4, p5, pm); }
    else if constexpr (K >= 1) { asm volatile(PK1(0) PKARGS); }
}

// SHAPE 0: none, 1: 4x4x1 (16 blocks), 2: 16x16x4, 3: 32x32x2
template <int SHAPE, int K>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters, float seed)
{
    const float a0 = seed + threadIdx.x;
    f2 p0 = {a0, a0 + 1}, p1 = {a0 + 2, a0 + 3}, p2 = {a0 + 4, a0 + 5}, p3 = {a0 + 6, a0 + 7}, p4 = p1, p5 = p0, p6 = p3, p7 = p2;
    const f2 pm = {0.999f, 0.999f};
    f4 c4a = {0, 0, 0, 0}, c4b = c4a;
    f16v c16a = {0}, c16b = {0};
    float ma = a0 * 1e-3f, mb = 1.0f - ma;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // two accumulators used in turn, so that a step never waits for the previous step's result
            if (SHAPE == 1) { if (u & 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c4a) : "v"(ma), "v"(mb)); else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c4b) : "v"(ma), "v"(mb)); }
            if (SHAPE == 2) { if (u & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4a) : "v"(ma), "v"(mb)); else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4b) : "v"(ma), "v"(mb)); }
            if (SHAPE == 3) { if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16a) : "v"(ma), "v"(mb)); else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16b) : "v"(ma), "v"(mb)); }
            valu_block<K>(p0, p1, p2, p3, p4, p5, p6, p7, pm);
        }
    }
    float r = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + c4a[0] + c4b[1] + c16a[3] + c16b[5];
    if (r == 12345.678f) out[0] = r;
}

template <int SHAPE, int K> static void run_same(const char* name, int waves_per_simd)
{
    float* out; CK(hipMalloc(&out, 4));
    const int iters = 2000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    same_wave<SHAPE, K><<<blocks, 256>>>(out, 10, 1.0f); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0)); same_wave<SHAPE, K><<<blocks, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double steps = (double)iters * 4 * waves_per_simd;     // steps per SIMD
    printf("%-14s K=%2d waves/SIMD=%d  %.3f ms  %.1f cyc/step/SIMD (@2.4GHz)\n", name, K, waves_per_simd, best, best * 1e-3 * 2.4e9 / steps);
    CK(hipFree(out));
}

// part 2: roles by wave number inside a 512-thread workgroup; role_mask bit0: waves 0-3 run MFMAs, bit1: waves 4-7 run VALU
template <int SHAPE>
__global__ __launch_bounds__(512) void two_roles(unsigned long long* cycles, int iters, float seed, int role_mask)
{
    const int wave = threadIdx.x >> 6;
    const float a0 = seed + threadIdx.x;
    f2 p0 = {a0, a0 + 1}, p1 = {a0 + 2, a0 + 3}, p2 = {a0 + 4, a0 + 5}, p3 = {a0 + 6, a0 + 7}, p4 = p1, p5 = p0, p6 = p3, p7 = p2;
    const f2 pm = {0.999f, 0.999f};
    f4 c4a = {0, 0, 0, 0}, c4b = c4a;
    f16v c16a = {0}, c16b = {0};
    float ma = a0 * 1e-3f, mb = 1.0f - ma;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (role_mask & 1)
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (SHAPE == 2) { if (u & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4a) : "v"(ma), "v"(mb)); else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4b) : "v"(ma), "v"(mb)); }
                    if (SHAPE == 3) { if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16a) : "v"(ma), "v"(mb)); else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16b) : "v"(ma), "v"(mb)); }
                }
            }
    } else if (role_mask & 2) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) valu_block<8>(p0, p1, p2, p3, p4, p5, p6, p7, pm);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + c4a[0] + c4b[1] + c16a[3] + c16b[5];
    if (r == 12345.678f) cycles[0] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cycles[1 + wave] = t1 - t0;
}

template <int SHAPE> static void run_roles(const char* name)
{
    unsigned long long* d; CK(hipMalloc(&d, 16 * 8));
    const int iters = 2000;
    for (int mask = 1; mask <= 3; mask++) {
        CK(hipMemset(d, 0, 16 * 8));
        two_roles<SHAPE><<<256, 512>>>(d, iters, 1.0f, mask); CK(hipDeviceSynchronize());
        two_roles<SHAPE><<<256, 512>>>(d, iters, 1.0f, mask); CK(hipDeviceSynchronize());
        unsigned long long h[16]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        const double mf = (double)h[1] / (iters * 4.0), va = (double)h[1 + 4] / (iters * 4.0 * 8.0);
        printf("%-12s roles=%s%s  mfma wave: %.1f cyc/mfma   valu wave: %.2f cyc/v_pk_fma_f32\n", name, mask & 1 ? "M" : "-", mask & 2 ? "V" : "-",
               mask & 1 ? mf : 0.0, mask & 2 ? va : 0.0);
    }
    CK(hipFree(d));
}

int main()
{
    for (int w : {1, 2, 4}) {
        run_same<0, 8>("valu only", w); run_same<0, 16>("valu only", w);
        run_same<1, 0>("4x4x1", w); run_same<1, 2>("4x4x1", w); run_same<1, 4>("4x4x1", w); run_same<1, 8>("4x4x1", w);
        run_same<2, 0>("16x16x4", w); run_same<2, 2>("16x16x4", w); run_same<2, 4>("16x16x4", w); run_same<2, 6>("16x16x4", w);
        run_same<2, 8>("16x16x4", w); run_same<2, 12>("16x16x4", w); run_same<2, 16>("16x16x4", w);
        run_same<3, 0>("32x32x2", w); run_same<3, 8>("32x32x2", w); run_same<3, 12>("32x32x2", w); run_same<3, 16>("32x32x2", w);
    }
    run_roles<2>("16x16x4");
    run_roles<3>("32x32x2");
    return 0;
}
