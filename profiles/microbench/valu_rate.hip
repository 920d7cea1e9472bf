// Microbenchmark: issue rate of v_fma_f32, v_pk_fma_f32, v_rsq_f32, v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef float float2_ __attribute__((ext_vector_type(2)));

template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0=seed+threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  float2_ p0={a0,a1}, p1={a2,a3}, p2={a4,a5}, p3={a6,a7}, p4={a1,a0}, p5={a3,a2}, p6={a5,a4}, p7={a7,a6};
  double d0=a0,d1=a1,d2=a2,d3=a3,d4=a4,d5=a5,d6=a6,d7=a7;
  float m = 0.999f; float2_ pm={0.999f,0.999f}; double dm=0.999;
  for (int i=0;i<iters;i++) {
#pragma unroll
    for (int u=0;u<8;u++) {
      if (MODE==0) { // 8 independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(m));
      } else if (MODE==1) { // 8 independent v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                     "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(pm));
      } else if (MODE==2) { // 8 independent v_rsq_f32
        asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                     "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
      } else if (MODE==3) { // 8 independent v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n"
                     "v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7\n"
                     : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3),"+v"(d4),"+v"(d5),"+v"(d6),"+v"(d7) : "v"(dm));
      } else if (MODE==4) { // mix: 6 fma + 2 rsq (does the transcendental co-issue?)
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_rsq_f32 %6, %6\n"
                     "v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_rsq_f32 %7, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(m));
      } else if (MODE==5) { // 8 v_mul_f32 with an SGPR operand
        asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                     "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "s"(seed));
      } else if (MODE==6) { // 8 v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(pm));
      } else if (MODE==7) { // 8 v_rsq_f64
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
                     : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3),"+v"(d4),"+v"(d5),"+v"(d6),"+v"(d7));
      }
    }
  }
  float r = a0+a1+a2+a3+a4+a5+a6+a7 + p0.x+p1.x+p2.x+p3.x+p4.y+p5.y+p6.y+p7.y + (float)(d0+d1+d2+d3+d4+d5+d6+d7);
  if (r == 12345.678f) out[0] = r;
}

template<int MODE> void run(const char* name, int waves_per_simd, double flop_per_inst_lane) {
  float* out; CK(hipMalloc(&out, 4));
  int iters = 4000;
  int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 wave/SIMD per block
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks,256>>>(out, 10, 1.0f);
  CK(hipDeviceSynchronize());
  float best=1e30f;
  for (int rep=0; rep<5; rep++) {
    CK(hipEventRecord(e0));
    k<MODE><<<blocks,256>>>(out, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if (ms<best) best=ms;
  }
  double insts_per_wave = (double)iters*64;
  double total_wave_insts = insts_per_wave * blocks * 4;
  double cyc_per_inst_per_simd = (best*1e-3*2.4e9) / (insts_per_wave * waves_per_simd);
  double tflops = total_wave_insts*64*flop_per_inst_lane / (best*1e-3) / 1e12;
  printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD (@2.4GHz)  %.1f TFLOP/s-equiv\n", name, waves_per_simd, best, cyc_per_inst_per_simd, tflops);
  CK(hipFree(out));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1,2,4,8}) {
    run<0>("v_fma_f32", w, 2); run<1>("v_pk_fma_f32", w, 4); run<2>("v_rsq_f32", w, 1);
    run<3>("v_fma_f64", w, 2); run<4>("6fma+2rsq", w, 1.75); run<5>("v_mul_f32 sgpr", w, 1); run<6>("v_pk_mul_f32", w, 2); run<7>("v_rsq_f64", w, 1);
  }
  return 0;
}
