// Launcher of the squaring-chain kernel; see cyc_chain.h for what it computes.
#pragma once
#include <hip/hip_runtime.h>

namespace bls381 {

enum { CHAIN_KEPT = 6 };  // values a chain keeps: after 16, 48, 57, 60, 62 and 63 squarings
// gin: slot array of 12 slots (one Fq12 per element, ark flat order, canonical words with Montgomery factor 2^392),
// gout: slot array of CHAIN_KEPT x 12 slots; both with element stride npad.  Enqueues on `stream`.
hipError_t launch_cyc_chain(hipStream_t stream, const void* gin, void* gout, int n, int npad);

}  // namespace bls381
