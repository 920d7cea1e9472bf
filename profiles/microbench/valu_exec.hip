// Microbenchmark: does a VALU instruction cost fewer cycles when part of the wave is masked off in EXEC?
// 8 independent v_fma_f32 (or v_rsq_f32) per iteration under different EXEC masks, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long mask) {
  float a0=seed+threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  float m = 0.999f;
  asm volatile("s_mov_b64 exec, %0" :: "s"(mask));
  for (int i=0;i<iters;i++) {
#pragma unroll
    for (int u=0;u<8;u++) {
      if (MODE==0)
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(m));
      else
        asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                     "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
    }
  }
  asm volatile("s_mov_b64 exec, -1");
  float r = a0+a1+a2+a3+a4+a5+a6+a7;
  if (r == 12345.678f) out[0] = r;
}

template<int MODE> void run(const char* name, unsigned long long mask) {
  float* out; CK(hipMalloc(&out, 4));
  const int iters = 4000, wps = 4, blocks = 256 * wps;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks,256>>>(out, 10, 1.0f, mask);
  CK(hipDeviceSynchronize());
  float best=1e30f;
  for (int rep=0; rep<5; rep++) {
    CK(hipEventRecord(e0));
    k<MODE><<<blocks,256>>>(out, iters, 1.0f, mask);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if (ms<best) best=ms;
  }
  const double insts_per_wave = (double)iters*64;
  printf("%-10s exec=%016llx  %.3f ms  %.2f cyc/inst/SIMD (@2.4GHz)\n", name, mask, best, (best*1e-3*2.4e9)/(insts_per_wave*wps));
  CK(hipFree(out));
}
int main() {
  const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0xffffull, 0x1ull, 0x5555555555555555ull, 0x00000000ffff0000ull, 0x0000ffff0000ffffull};
  for (unsigned long long m : masks) run<0>("v_fma_f32", m);
  for (unsigned long long m : masks) run<1>("v_rsq_f32", m);
  return 0;
}
