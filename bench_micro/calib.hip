// Calibration microbenchmarks for the gfx950 integer-multiply pipeline.
// Measures sustained throughput of v_mad_u64_u32 and the companion carry
// instructions at 1/2/4/8 waves per SIMD, so that the pairing kernels'
// roofline denominator is a measured number, not a datasheet guess.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); return 1;}}while(0)

#define ITERS 2048

// 8 independent 64-bit accumulators per lane, 8 mads per iteration.
__global__ void k_mad64(u32* out, u32 seed, u64* clk){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed;
  u64 t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, s[8:9], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s8","s9");
  }
  u64 t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime();
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
  if(threadIdx.x==0 && blockIdx.x==0){clk[0]=t1-t0; clk[1]=r1-r0;}
}
// mad + addc carry-count pair (the 2-instruction MAC)
__global__ void k_mad_addc(u32* out, u32 seed, u64* clk){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc[8]; u32 cnt[8];
  for(int i=0;i<8;i++){acc[i]=i+seed; cnt[i]=0;}
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(cnt[i]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]+cnt[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
// same, but the 8 mads first then the 8 addc (carry in separate SGPR pairs)
__global__ void k_mad_addc_split(u32* out, u32 seed, u64* clk){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc[4]; u32 cnt[4];
  for(int i=0;i<4;i++){acc[i]=i+seed; cnt[i]=0;}
  for(int it=0; it<ITERS*2; ++it){
    asm volatile(
      "v_mad_u64_u32 %0, s[8:9], %8, %9, %0\n\t"
      "v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n\t"
      "v_mad_u64_u32 %2, s[12:13], %8, %9, %2\n\t"
      "v_mad_u64_u32 %3, s[14:15], %8, %9, %3\n\t"
      "v_addc_co_u32 %4, s[8:9], 0, %4, s[8:9]\n\t"
      "v_addc_co_u32 %5, s[10:11], 0, %5, s[10:11]\n\t"
      "v_addc_co_u32 %6, s[12:13], 0, %6, s[12:13]\n\t"
      "v_addc_co_u32 %7, s[14:15], 0, %7, s[14:15]\n\t"
      : "+v"(acc[0]),"+v"(acc[1]),"+v"(acc[2]),"+v"(acc[3]),"+v"(cnt[0]),"+v"(cnt[1]),"+v"(cnt[2]),"+v"(cnt[3])
      : "v"(a), "v"(b) : "s8","s9","s10","s11","s12","s13","s14","s15");
  }
  u64 s=0; for(int i=0;i<4;i++) s+=acc[i]+cnt[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
__global__ void k_mullo(u32* out, u32 seed, u64* clk){
  u32 b = seed*3 + blockIdx.x;
  u32 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_mulhi(u32* out, u32 seed, u64* clk){
  u32 b = seed*3 + blockIdx.x;
  u32 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_mad24(u32* out, u32 seed, u64* clk){
  u32 b = seed*3 + blockIdx.x;
  u32 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(b));
  }
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_add32(u32* out, u32 seed, u64* clk){
  u32 b = seed*3 + blockIdx.x;
  u32 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
  }
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_add64(u32* out, u32 seed, u64* clk){
  u64 b = seed*3 + blockIdx.x;
  u64 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s;
}
// dependent chain of mads: latency
__global__ void k_mad64_dep(u32* out, u32 seed, u64* clk){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc=seed;
  for(int it=0; it<ITERS; ++it){
    #pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, s[8:9], %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "s8","s9");
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)acc^(u32)(acc>>32);
}

typedef void (*kern_t)(u32*,u32,u64*);
struct K { const char* name; kern_t k; double ops_per_iter; };

int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  int cus=prop.multiProcessorCount;
  printf("device %s CUs %d clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  u32* out; u64* clk; CK(hipMalloc(&out, (size_t)cus*8*256*4*4)); CK(hipMalloc(&clk,16));
  K ks[] = {
    {"mad_u64_u32 x8 indep", k_mad64, 8},
    {"mad_u64_u32 dependent", k_mad64_dep, 8},
    {"mad+addc pair", k_mad_addc, 8},
    {"mad+addc split(4)", k_mad_addc_split, 8},
    {"mul_lo_u32", k_mullo, 8},
    {"mul_hi_u32", k_mulhi, 8},
    {"mad_u32_u24", k_mad24, 8},
    {"add_co+addc pair", k_add32, 8},
    {"lshl_add_u64", k_add64, 8},
  };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(auto& k: ks){
    for(int w: {1,2,4,8}){
      int blocks=cus*w; // 256 threads = 4 waves = one per SIMD
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, clk);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for(int r=0;r<5;r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u+r, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=5;
      double lane_ops = (double)blocks*256*ITERS*k.ops_per_iter;
      double wave_instr_per_simd = (double)w*ITERS*k.ops_per_iter; // per SIMD
      u64 h[2]={0,0}; CK(hipMemcpy(h,clk,16,hipMemcpyDeviceToHost));
      double mhz = h[1]? (double)h[0]/(double)h[1]*100.0 : 0;
      printf("%-24s w/SIMD=%d  %8.3f ms  %8.3f Tlane-op/s  ns/wave-instr/SIMD=%.3f  (cyc@2.4GHz=%.2f) clk~%.0fMHz\n", k.name, w, ms, lane_ops/ms/1e9, ms*1e6/wave_instr_per_simd, ms*1e6/wave_instr_per_simd*2.4, mhz);
    }
  }
  return 0;
}
