// Latency of a global store as a wave sees it (issue -> vmcnt reaches 0), alone and with the machine full of
// waves doing the same; beside it the latency of a load that hits L2.
// build: hipcc --offload-arch=gfx950 -O3 -o bench_micro/_obj/store_latency bench_micro/store_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32;
__global__ void k(uint4* buf, unsigned long long* out, int iters, int mode, int work) {
    const size_t gw = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    uint4* p = buf + gw * 64 * 4 + (threadIdx.x & 63);  // 4 KB a wave: four full-line stores
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    unsigned long long total = 0;
    const long long whole0 = clock64();
    float acc = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        long long t0 = clock64();
        if (mode == 0) {
            p[0] = v; p[64] = v; p[128] = v; p[192] = v;
            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) only (expcnt 7, lgkmcnt 15 left alone)
        } else if (mode == 2) {  // stores nobody waits for
            p[0] = v; p[64] = v; p[128] = v; p[192] = v;
        } else if (mode == 3) {  // nothing: the arithmetic alone
        } else {
            uint4 a = p[0];
            asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
            v.x += a.x;
        }
        long long t1 = clock64();
        total += (unsigned long long)(t1 - t0);
        for (int w = 0; w < work; w++) acc = acc * 1.0001f + 0.5f;  // arithmetic between the stores
        v.y++;
    }
    if (mode >= 2) total = (unsigned long long)(clock64() - whole0);  // the whole loop
    if ((threadIdx.x & 63) == 0) out[gw] = total + (acc == 12345.f);
}
int main() {
    const int WG = 256, waves_full = 256 * 8;  // 2 waves a SIMD
    uint4* buf; unsigned long long* out;
    hipMalloc(&buf, (size_t)waves_full * 4096); hipMalloc(&out, waves_full * 8);
    hipMemset(buf, 0, (size_t)waves_full * 4096);
    std::vector<unsigned long long> h(waves_full);
    const int iters = 2000;
    for (int mode = 0; mode < 4; mode++)
        for (int work : {0, 2000})
            for (int waves : {1, waves_full}) {
                int blocks = waves == 1 ? 1 : waves / (WG / 64);
                int threads = waves == 1 ? 64 : WG;
                k<<<blocks, threads>>>(buf, out, iters, mode, work);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), out, waves * 8, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.begin() + waves);
                printf("%s work=%d waves=%d: median %.0f max %.0f ticks per iteration (clock64 ticks; 100 MHz -> x10 ns)\n",
                       mode == 0 ? "store+wait" : mode == 1 ? "load" : mode == 2 ? "loop with stores" : "loop alone", work, waves, (double)h[waves / 2] / iters, (double)h[waves - 1] / iters);
            }
    return 0;
}
