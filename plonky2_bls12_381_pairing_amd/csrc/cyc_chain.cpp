// Host side of the squaring-chain kernel (cyc_chain.h).
#include "cyc_chain.h"

#include "cyc_chain_host.h"

namespace bls381 {

hipError_t launch_cyc_chain(hipStream_t stream, const void* gin, void* gout, int n, int npad) {
    if (n <= 0) return hipSuccess;
    ChainArgs a;
    a.gin = (const uint4*)gin; a.gout = (uint4*)gout; a.n = n; a.npad = npad;
    if (n <= CHAIN_ONE_TRIO_UP_TO)
        hipLaunchKernelGGL(cyc_chain_kernel<1>, dim3((unsigned)((n + 63) / 64)), dim3(192), 0, stream, a);
    else
        hipLaunchKernelGGL(cyc_chain_kernel<CHAIN_TRIOS_LARGE>, dim3((unsigned)((n + 64 * CHAIN_TRIOS_LARGE - 1) / (64 * CHAIN_TRIOS_LARGE))),
                           dim3(192 * CHAIN_TRIOS_LARGE), 0, stream, a);
    return hipGetLastError();
}

}  // namespace bls381
