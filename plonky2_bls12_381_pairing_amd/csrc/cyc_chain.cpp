// Host side of the squaring-chain kernel (cyc_chain.h).
#include "cyc_chain.h"

#include "cyc_chain_host.h"

namespace bls381 {

hipError_t launch_cyc_chain(hipStream_t stream, const void* gin, void* gout, int n, int npad) {
    if (n <= 0) return hipSuccess;
    ChainArgs a;
    a.gin = (const uint4*)gin; a.gout = (uint4*)gout; a.n = n; a.npad = npad;
    hipLaunchKernelGGL(cyc_chain_kernel, dim3((unsigned)((n + CHAIN_ELEMS - 1) / CHAIN_ELEMS)), dim3(CHAIN_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace bls381
