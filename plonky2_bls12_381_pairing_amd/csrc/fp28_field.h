// One-lane field arithmetic on the interpreter's number format (14 digits of 28 bits, Montgomery factor 2^392), for
// the long product chains of the point decoders and witness generators: the square-root exponentiation x^((p+1)/4)
// is 378 squarings and about a hundred products with nothing in between, and a product costs 517 instructions in
// this form against 746 on 12 x 32-bit limbs (fp_mul, where every multiply-accumulate is followed by an
// add-with-carry).  Values come from and go back to the 12-limb boundary form (Montgomery factor 2^384, canonical)
// through one multiplication by a constant each way.
// Measured on MI355X, 65,536 points: G1 decoding 1.31 -> 1.20 ms, G2 2.66 -> 2.36 ms (with the subgroup tests 5.43 ->
// 5.30 and 7.46 -> 7.15 ms).  The point arithmetic of the subgroup tests stays on 12 limbs: moved to this form as
// well (sums as digit-wise additions plus a carry sweep) it had a quarter fewer instructions and ran a fifth slower
// (G1 decoding with the test 5.4 -> 6.2 ms).
#pragma once
#include <hip/hip_runtime.h>

#include "fp28_dev.h"

namespace bls381 {
namespace g28 {

struct F { u32 v[D28]; };

__device__ __forceinline__ F mul(const F& a, const F& b) { F r; fp28_mul(a.v, b.v, r.v); return r; }  // in (-p, 2p)
__device__ __forceinline__ F sqr(const F& a) { return mul(a, a); }

// boundary limbs (x * 2^384, canonical) -> x * 2^392: times the integer 2^400 mod p
__device__ __forceinline__ F from_boundary(const u32* m) {
    const F k = {{0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u, 0xa4e6fe9u, 0x3e8a053u,
                  0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau}};
    F r;
    unpack28(m, r.v);
    return mul(r, k);
}
// a product's result (in (-p, 2p), digits normalised) -> boundary limbs: times the integer 2^384 mod p, then the
// representative in [0, p)
__device__ __forceinline__ void to_boundary(const F& a, u32* m) {
    const F k = {{0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u, 0x7052574u, 0x7ce5853u,
                  0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u}};
    F r = mul(a, k);
    canon28(r.v);
    pack28(r.v, m);
}

}  // namespace g28
}  // namespace bls381
