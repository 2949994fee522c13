// One-lane field arithmetic on the interpreter's number format, for the point decoders and subgroup tests: an Fq
// value is 14 digits of 28 bits (digits 0..12 normalised, signed top digit), Montgomery factor 2^392, lazily
// reduced.  A product costs about 470 instructions against about 630 on 12 x 32-bit limbs, where every
// multiply-accumulate is followed by an add-with-carry that waits on VCC (fp_mul).
//
// Bounds: products (mul, dot2) return values in (-p, 2p); add / sub are digit-wise followed by a carry sweep, so
// values grow with every sum.  The multiplier needs |a| |b| (summed over the products of one reduction) below
// 2500 p^2; the callers' formulas stay far inside that as long as what they multiply is at most four sums away
// from a product, and `times12` (the curve constant 3b = 12) ends with a reduction.  Comparisons reduce first.
// Boundary values (12 x u32, Montgomery factor 2^384, canonical) come in and go out through one multiplication
// by a constant each, as in the interpreter's programs.
#pragma once
#include <hip/hip_runtime.h>

#include "fp28_dev.h"
#include "vm_kernels.h"

namespace bls381 {
namespace g28 {

struct F { u32 v[D28]; };
struct F2 { F c0, c1; };

// constants: values times 2^392 mod p unless noted (rows of the interpreter's constant table)
__device__ __forceinline__ F one() {
    return {{0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu, 0xda0f73eu, 0x9b09b42u,
             0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au}};
}
__device__ __forceinline__ F k_2_400() {  // the integer 2^400 mod p: x * 2^384 -> x * 2^392
    return {{0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u, 0xa4e6fe9u, 0x3e8a053u,
             0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau}};
}
__device__ __forceinline__ F k_2_384() {  // the integer 2^384 mod p: x * 2^392 -> x * 2^384
    return {{0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u, 0x7052574u, 0x7ce5853u,
             0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u}};
}
__device__ __forceinline__ F k_2_784() {  // the integer 2^784 mod p: canonical x -> x * 2^392
    return {{0x10370edu, 0x6d1c345u, 0xe243d62u, 0xec45c53u, 0x3b1d65au, 0x093317du, 0xb4f36a0u, 0x5d74088u, 0xc10ea72u,
             0x865d118u, 0x7320a75u, 0xfd5cd50u, 0xcc8a759u, 0x000c8d4u}};
}

__device__ __forceinline__ F zero() { F r; for (int k = 0; k < D28; k++) r.v[k] = 0; return r; }
__device__ __forceinline__ F mul(const F& a, const F& b) { F r; fp28_mul(a.v, b.v, r.v); return r; }
__device__ __forceinline__ F sqr(const F& a) { return mul(a, a); }
// a * b + c * d, one reduction
__device__ __forceinline__ F dot2(const F& a, const F& b, const F& c, const F& d) { F r; fp28_dot2(a.v, b.v, c.v, d.v, r.v); return r; }
__device__ __forceinline__ F add(const F& a, const F& b) {
    F r;
#pragma unroll
    for (int k = 0; k < D28; k++) r.v[k] = a.v[k] + b.v[k];
    normalise28(r.v);
    return r;
}
__device__ __forceinline__ F sub(const F& a, const F& b) {
    F r;
#pragma unroll
    for (int k = 0; k < D28; k++) r.v[k] = a.v[k] - b.v[k];
    normalise28(r.v);
    return r;
}
__device__ __forceinline__ F neg(const F& a) { return sub(zero(), a); }
// into (-0.2 p, 1.2 p)
__device__ __forceinline__ F reduced(const F& a) { F r = a; reduce28(r.v); return r; }
// the representative in [0, p)
__device__ __forceinline__ F canonical(const F& a) { F r = a; reduce28(r.v); canon28(r.v); return r; }
__device__ __forceinline__ bool is_zero(const F& a) {
    const F c = canonical(a);
    u32 nz = 0;
#pragma unroll
    for (int k = 0; k < D28; k++) nz |= c.v[k];
    return nz == 0;
}
__device__ __forceinline__ bool eq(const F& a, const F& b) { return is_zero(sub(a, b)); }

// 12 x u32 in and out
__device__ __forceinline__ F from_limbs_raw(const u32* l) { F r; unpack28(l, r.v); return r; }
__device__ __forceinline__ void to_limbs_canonical(const F& a, u32* l) {  // a in (-p, 2p), digits normalised
    F c = a;
    canon28(c.v);
    pack28(c.v, l);
}
__device__ __forceinline__ F from_canonical(const u32* x) { return mul(from_limbs_raw(x), k_2_784()); }  // integer below p -> Montgomery
__device__ __forceinline__ void to_canonical(const F& a, u32* x) {                                      // Montgomery -> integer below p
    F o = zero(); o.v[0] = 1;
    to_limbs_canonical(mul(reduced(a), o), x);
}
__device__ __forceinline__ F from_boundary(const u32* m) { return mul(from_limbs_raw(m), k_2_400()); }  // factor 2^384 -> 2^392
__device__ __forceinline__ void to_boundary(const F& a, u32* m) { to_limbs_canonical(mul(reduced(a), k_2_384()), m); }

// 1 / a (0 -> 0), through the division-step inversion on boundary limbs
__device__ __forceinline__ F inv(const F& a) {
    u32 m[12], r[12];
    to_boundary(a, m);
    fp_invert_boundary(m, r);
    return from_boundary(r);
}

__device__ __forceinline__ F2 add(const F2& a, const F2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
__device__ __forceinline__ F2 sub(const F2& a, const F2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
__device__ __forceinline__ bool is_zero(const F2& a) { return is_zero(a.c0) && is_zero(a.c1); }

}  // namespace g28
}  // namespace bls381
