// Witness-trace generation for the single-pair pairing circuit (placeholder interface; see witness.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace bls381 {

struct WitnessProgram {
    size_t n_rows = 0;
    std::vector<uint8_t> kinds;  // per row: 0 add, 1 sub, 2 mul, 3 neg, 4 inv, 5 generator output
};
const WitnessProgram& witness_program();

struct WitnessDevice {
    int run(hipStream_t stream, const uint4* consts, const uint64_t* g1, const uint64_t* g2, size_t n, uint32_t* trace,
            std::string& err);
    void release();
};

}  // namespace bls381
