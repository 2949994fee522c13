// Second calibration pass: per-instruction issue cost of the VALU opcodes a
// multi-precision multiply is built from (plain add, carry add, 64-bit add,
// select, integer mad), at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); return 1;}}while(0)
#define ITERS 8192

#define KERN32(NAME, ASMSTR) \
__global__ void NAME(u32* out, u32 seed){ \
  u32 b = seed*3 + blockIdx.x; u32 acc[8]; \
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x; \
  for(int it=0; it<ITERS; ++it){ \
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile(ASMSTR : "+v"(acc[i]) : "v"(b) : "vcc"); \
  } \
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }

KERN32(k_add_u32,   "v_add_u32 %0, %0, %1")
KERN32(k_fma_f32,   "v_fma_f32 %0, %0, %1, %0")
KERN32(k_add3_u32,  "v_add3_u32 %0, %0, %1, %1")
KERN32(k_add_co,    "v_add_co_u32 %0, vcc, %0, %1")
KERN32(k_addc_co,   "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERN32(k_mov,       "v_mov_b32 %0, %1")
KERN32(k_cndmask,   "v_cndmask_b32 %0, %0, %1, vcc")
KERN32(k_and,       "v_and_b32 %0, %0, %1")
KERN32(k_alignbit,  "v_alignbit_b32 %0, %0, %1, 7")
KERN32(k_mul_u24,   "v_mul_u32_u24 %0, %0, %1")
KERN32(k_mul_lo,    "v_mul_lo_u32 %0, %0, %1")
KERN32(k_subrev_co, "v_subb_co_u32 %0, vcc, %0, %1, vcc")
KERN32(k_add_co_sgpr, "v_add_co_u32 %0, s[10:11], %0, %1")

__global__ void k_mad64_vcc(u32* out, u32 seed){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x; u64 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
// mad with separate dst (no accumulate dependency): d = a*b + c
__global__ void k_mad64_nodep(u32* out, u32 seed){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x; u64 acc[8]; u64 c=seed;
  for(int i=0;i<8;i++) acc[i]=i+seed;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(acc[i]) : "v"(a), "v"(b), "v"(c) : "vcc");
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
__global__ void k_lshl_add_u64(u32* out, u32 seed){
  u64 b = seed*3 + blockIdx.x; u64 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s;
}
// realistic row: 12 mads (carry to distinct sgpr pairs) then 12 addc consuming them
__global__ void k_row(u32* out, u32 seed){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 w[6]; u32 c[6];
  for(int i=0;i<6;i++){w[i]=i+seed; c[i]=0;}
  for(int it=0; it<ITERS; ++it){
    asm volatile(
      "v_mad_u64_u32 %0, s[8:9], %12, %13, %0\n\t"
      "v_mad_u64_u32 %1, s[10:11], %12, %13, %1\n\t"
      "v_mad_u64_u32 %2, s[12:13], %12, %13, %2\n\t"
      "v_mad_u64_u32 %3, s[14:15], %12, %13, %3\n\t"
      "v_mad_u64_u32 %4, s[16:17], %12, %13, %4\n\t"
      "v_mad_u64_u32 %5, s[18:19], %12, %13, %5\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, s[8:9]\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, s[10:11]\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, s[12:13]\n\t"
      "v_addc_co_u32 %9, vcc, 0, %9, s[14:15]\n\t"
      "v_addc_co_u32 %10, vcc, 0, %10, s[16:17]\n\t"
      "v_addc_co_u32 %11, vcc, 0, %11, s[18:19]\n\t"
      : "+v"(w[0]),"+v"(w[1]),"+v"(w[2]),"+v"(w[3]),"+v"(w[4]),"+v"(w[5]),
        "+v"(c[0]),"+v"(c[1]),"+v"(c[2]),"+v"(c[3]),"+v"(c[4]),"+v"(c[5])
      : "v"(a), "v"(b) : "vcc","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19");
  }
  u64 s=0; for(int i=0;i<6;i++) s+=w[i]+c[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}

typedef void (*kern_t)(u32*,u32);
struct K { const char* name; kern_t k; double ops_per_iter; };
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  int cus=prop.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, (size_t)cus*8*256*4*4));
  K ks[] = {
    {"v_add_u32", k_add_u32, 8}, {"v_fma_f32", k_fma_f32, 8}, {"v_add3_u32", k_add3_u32, 8},
    {"v_add_co_u32 vcc", k_add_co, 8}, {"v_add_co_u32 sgpr", k_add_co_sgpr, 8}, {"v_addc_co_u32 vcc", k_addc_co, 8}, {"v_subb_co_u32", k_subrev_co, 8},
    {"v_mov_b32", k_mov, 8}, {"v_cndmask_b32", k_cndmask, 8}, {"v_and_b32", k_and, 8}, {"v_alignbit_b32", k_alignbit, 8},
    {"v_mul_u32_u24", k_mul_u24, 8}, {"v_mul_lo_u32", k_mul_lo, 8},
    {"v_mad_u64_u32 vcc", k_mad64_vcc, 8}, {"v_mad_u64_u32 nodep", k_mad64_nodep, 8}, {"v_lshl_add_u64", k_lshl_add_u64, 8},
    {"row 6mad+6addc", k_row, 12},
  };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // warm the clocks
  for(int r=0;r<20;r++) hipLaunchKernelGGL(k_mad64_vcc, dim3(cus*8), dim3(256), 0, 0, out, 1u);
  CK(hipDeviceSynchronize());
  for(auto& k: ks){
    for(int w: {1,2,4,8}){
      int blocks=cus*w;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for(int r=0;r<3;r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u+r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=3;
      double lane_ops = (double)blocks*256*ITERS*k.ops_per_iter;
      double wi = (double)w*ITERS*k.ops_per_iter;
      printf("%-22s w=%d %8.3f ms %7.2f Tlane-op/s  ns/wave-instr=%.3f\n", k.name, w, ms, lane_ops/ms/1e9, ms*1e6/wi);
    }
  }
  return 0;
}
