// Third calibration pass: dependent multiply-accumulate chains, signed mad, 64-bit shifts,
// xor-add and LDS 128-bit reads, at 1/2/4 waves per SIMD.  ns/wave-instr x 2.4 = cycles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); return 1;}}while(0)
#define ITERS 8192

// NCH independent accumulators, 8 mads per iteration in total
template <int NCH>
__global__ void k_mad_chain(u32* out, u32 seed){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x; u64 acc[NCH];
  for(int i=0;i<NCH;i++) acc[i]=i+seed;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i%NCH]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s=0; for(int i=0;i<NCH;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
__global__ void k_mad_i64(u32* out, u32 seed){
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x; u64 acc[8];
  for(int i=0;i<8;i++) acc[i]=i+seed;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32);
}
#define KERN64(NAME, ASMSTR) \
__global__ void NAME(u32* out, u32 seed){ \
  u64 acc[8]; for(int i=0;i<8;i++) acc[i]=((u64)(i+seed+threadIdx.x)<<33)|blockIdx.x; \
  for(int it=0; it<ITERS; ++it){ \
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile(ASMSTR : "+v"(acc[i])); \
  } \
  u64 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)s^(u32)(s>>32); }
KERN64(k_lshr64, "v_lshrrev_b64 %0, 1, %0")
KERN64(k_ashr64, "v_ashrrev_i64 %0, 1, %0")
#define KERN32(NAME, ASMSTR) \
__global__ void NAME(u32* out, u32 seed){ \
  u32 b = seed*3 + blockIdx.x; u32 acc[8]; \
  for(int i=0;i<8;i++) acc[i]=i+seed+threadIdx.x; \
  for(int it=0; it<ITERS; ++it){ \
    _Pragma("unroll") for(int i=0;i<8;i++) asm volatile(ASMSTR : "+v"(acc[i]) : "v"(b) : "vcc"); \
  } \
  u32 s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
KERN32(k_xad,      "v_xad_u32 %0, %0, %1, %1")
KERN32(k_xor,      "v_xor_b32 %0, %0, %1")
KERN32(k_ashr32,   "v_ashrrev_i32 %0, 3, %0")
KERN32(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1")
KERN32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
KERN32(k_bfe,      "v_bfe_u32 %0, %0, 3, 28")
KERN32(k_and_or,   "v_and_or_b32 %0, %0, %1, %1")
KERN32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")

// LDS 128-bit reads: 8 per iteration, conflict-free (consecutive uint4 per lane)
__global__ void k_ds_read128(u32* out, u32 seed){
  __shared__ uint4 buf[256*4];
  for(int i=threadIdx.x;i<1024;i+=256) buf[i]=make_uint4(i,seed,i,seed);
  __syncthreads();
  uint4 s = make_uint4(0,0,0,0);
  u32 idx = threadIdx.x;
  for(int it=0; it<ITERS; ++it){
    _Pragma("unroll") for(int i=0;i<8;i++){ uint4 v = buf[(idx + i*64) & 1023]; s.x+=v.x; s.y^=v.y; s.z+=v.z; s.w^=v.w; }
    idx += s.x & 1;
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s.x+s.y+s.z+s.w;
}

typedef void (*kern_t)(u32*,u32);
struct K { const char* name; kern_t k; double ops_per_iter; };
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  int cus=prop.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, (size_t)cus*8*256*4*4));
  K ks[] = {
    {"mad_u64 8 chains", k_mad_chain<8>, 8}, {"mad_u64 4 chains", k_mad_chain<4>, 8}, {"mad_u64 2 chains", k_mad_chain<2>, 8},
    {"mad_u64 1 chain", k_mad_chain<1>, 8}, {"v_mad_i64_i32", k_mad_i64, 8},
    {"v_lshrrev_b64", k_lshr64, 8}, {"v_ashrrev_i64", k_ashr64, 8},
    {"v_xad_u32", k_xad, 8}, {"v_xor_b32", k_xor, 8}, {"v_ashrrev_i32", k_ashr32, 8}, {"v_mul_hi_u32", k_mul_hi, 8},
    {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"v_bfe_u32", k_bfe, 8}, {"v_and_or_b32", k_and_or, 8}, {"v_lshl_add_u32", k_lshl_add, 8},
    {"ds_read_b128 (+4 valu)", k_ds_read128, 8},
  };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int r=0;r<20;r++) hipLaunchKernelGGL(k_mad_chain<8>, dim3(cus*8), dim3(256), 0, 0, out, 1u);
  CK(hipDeviceSynchronize());
  for(auto& k: ks){
    for(int w: {1,2,4}){
      int blocks=cus*w;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for(int r=0;r<3;r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 12345u+r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=3;
      double wi = (double)w*ITERS*k.ops_per_iter;
      printf("%-24s w=%d %8.3f ms  ns/wave-instr=%.3f cycles=%.2f\n", k.name, w, ms, ms*1e6/wi, ms*1e6/wi*2.4);
    }
  }
  return 0;
}
