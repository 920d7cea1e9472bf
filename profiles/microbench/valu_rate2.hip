// Microbenchmark 2: operand-kind effects on VALU issue rate (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef float float2_ __attribute__((ext_vector_type(2)));
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0=seed+threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  float2_ p0={a0,a1}, p1={a2,a3}, p2={a4,a5}, p3={a6,a7}, p4={a1,a0}, p5={a3,a2}, p6={a5,a4}, p7={a7,a6};
  float m = 0.999f; float2_ pm={0.999f,0.999f}; float2_ spair={seed, seed*0.5f};
  for (int i=0;i<iters;i++) {
#pragma unroll
    for (int u=0;u<8;u++) {
#define ARGS8 : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)
#define PARGS8 : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7)
      if (MODE==0) { asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n" ARGS8 : "v"(m)); }
      else if (MODE==1) { asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n" ARGS8 : "v"(m)); }
      else if (MODE==2) { asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n" ARGS8 : "s"(seed)); }
      else if (MODE==3) { asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n" ARGS8 : "s"(seed)); }
      else if (MODE==4) { asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n" ARGS8 : "v"(m)); }
      else if (MODE==5) { asm volatile("v_pk_add_f32 %0, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n" PARGS8 : "v"(pm)); }
      else if (MODE==6) { asm volatile("v_pk_fma_f32 %0, %0, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %4, %8, %4 op_sel:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %5 op_sel:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %6 op_sel:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %7 op_sel:[0,1,0]\n" PARGS8 : "v"(pm)); }
      else if (MODE==7) { asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n" ARGS8 ); }
      else if (MODE==8) { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n" ARGS8 ); }
      else if (MODE==9) { asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" ARGS8 : "v"(m) : "vcc"); }
      else if (MODE==10) { asm volatile("v_pk_add_f32 %0, %8, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %8, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %8, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %8, %3 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %4, %8, %4 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, %8, %5 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6, %8, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7, %8, %7 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" PARGS8 : "s"(spair)); }
      else if (MODE==11) { asm volatile("v_pk_mul_f32 %0, %8, %0 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %8, %1 op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %8, %2 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %8, %3 op_sel_hi:[0,1]\n v_pk_mul_f32 %4, %8, %4 op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %8, %5 op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %8, %6 op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %8, %7 op_sel_hi:[0,1]\n" PARGS8 : "s"(spair)); }
      else if (MODE==12) { asm volatile("v_cmp_gt_f32 s[20:21], %8, %0\n v_cmp_gt_f32 s[22:23], %8, %1\n v_cmp_gt_f32 s[20:21], %8, %2\n v_cmp_gt_f32 s[22:23], %8, %3\n v_cmp_gt_f32 s[20:21], %8, %4\n v_cmp_gt_f32 s[22:23], %8, %5\n v_cmp_gt_f32 s[20:21], %8, %6\n v_cmp_gt_f32 s[22:23], %8, %7\n" ARGS8 : "s"(seed) : "s20","s21","s22","s23"); }
      else if (MODE==13) { asm volatile("v_min_f32 %0, %1, %0\n v_min_f32 %1, %2, %1\n v_min_f32 %2, %3, %2\n v_min_f32 %3, %4, %3\n v_min_f32 %4, %5, %4\n v_min_f32 %5, %6, %5\n v_min_f32 %6, %7, %6\n v_min_f32 %7, %0, %7\n" ARGS8 ); }
      else if (MODE==14) { asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cmp_gt_f32 vcc, %8, %1\n v_cmp_gt_f32 vcc, %8, %2\n v_cmp_gt_f32 vcc, %8, %3\n v_cmp_gt_f32 vcc, %8, %4\n v_cmp_gt_f32 vcc, %8, %5\n v_cmp_gt_f32 vcc, %8, %6\n v_cmp_gt_f32 vcc, %8, %7\n" ARGS8 : "v"(m) : "vcc"); }
      else if (MODE==15) { asm volatile("v_rsq_f32 %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_rsq_f32 %2, %2\n v_pk_mul_f32 %3, %3, %3\n v_rsq_f32 %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_rsq_f32 %6, %6\n v_pk_mul_f32 %7, %7, %7\n" : "+v"(a0),"+v"(p1),"+v"(a2),"+v"(p3),"+v"(a4),"+v"(p5),"+v"(a6),"+v"(p7)); }
    }
  }
  float r = a0+a1+a2+a3+a4+a5+a6+a7 + p0.x+p1.x+p2.x+p3.x+p4.y+p5.y+p6.y+p7.y;
  if (r == 12345.678f) out[0] = r;
}
template<int MODE> void run(const char* name, int waves_per_simd) {
  float* out; CK(hipMalloc(&out, 4));
  int iters = 4000; int blocks = 256 * waves_per_simd;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks,256>>>(out, 10, 1.0f); CK(hipDeviceSynchronize());
  float best=1e30f;
  for (int rep=0; rep<5; rep++) { CK(hipEventRecord(e0)); k<MODE><<<blocks,256>>>(out, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if (ms<best) best=ms; }
  double insts_per_wave = (double)iters*64;
  printf("%-26s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD (@2.4GHz)\n", name, waves_per_simd, best, (best*1e-3*2.4e9)/(insts_per_wave*waves_per_simd));
  CK(hipFree(out));
}
int main() {
  for (int w : {2,8}) {
    run<0>("v_mul_f32 vgpr (VOP2)", w); run<1>("v_add_f32 vgpr", w); run<2>("v_sub_f32 sgpr", w); run<3>("v_fma_f32 sgpr src1", w);
    run<4>("v_fmac_f32 vgpr", w); run<5>("v_pk_add_f32 neg", w); run<6>("v_pk_fma_f32 op_sel bcast", w); run<7>("v_mul_f32 a*a", w); run<8>("v_fma_f32 a*a+a", w); run<9>("v_cmp+v_cndmask pairs", w);
    run<10>("v_pk_add_f32 sgpr pair", w); run<11>("v_pk_mul_f32 sgpr pair", w); run<12>("v_cmp_gt_f32 sgpr->sgpr", w); run<13>("v_min_f32 vgpr", w); run<14>("v_cmp_gt_f32 vgpr->vcc", w); run<15>("v_rsq_f32 + v_pk_mul alt", w);
  }
  return 0;
}
