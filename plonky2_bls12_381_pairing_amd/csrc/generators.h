// Values of the reference's other witness generators (square roots and inverses in Fq, Fq2, Fq6), one
// lane per element, on canonical 12 x u32 limbs in and out as the generators read and write them
// (from_biguint_to_fq / set_biguint_target).  Arithmetic: 12 x 32-bit Montgomery limbs (fp_mul), the
// division-step inversion, x^((p+1)/4) for square roots (p = 3 mod 4; that chain on 28-bit digits).  Not throughput-critical: a
// circuit holds a handful of these per proof.
#pragma once
#include <hip/hip_runtime.h>

#include "fp28_field.h"
#include "fp_asm_gen.h"
#include "fp_dev.h"
#include "vm_kernels.h"

namespace bls381 {
namespace gen {

struct F { u32 v[12]; };
struct F2 { F c0, c1; };

__device__ __forceinline__ F mul(const F& a, const F& b) { F r; fp_mul(a.v, b.v, r.v); return r; }
__device__ __forceinline__ F sqr(const F& a) { return mul(a, a); }
__device__ __forceinline__ F add(const F& a, const F& b) {
    F r;
    carry_t c;
    r.v[0] = add_co(a.v[0], b.v[0], c);
#pragma unroll
    for (int k = 1; k < 12; k++) r.v[k] = addc_co(a.v[k], b.v[k], c);
    fp_cond_sub_p(r.v);
    return r;
}
__device__ __forceinline__ F sub(const F& a, const F& b) {
    const u32 P[12] = BLS381_P_LIMBS;
    F r;
    carry_t c;
    r.v[0] = sub_co(a.v[0], b.v[0], c);
#pragma unroll
    for (int k = 1; k < 12; k++) r.v[k] = subb_co(a.v[k], b.v[k], c);
    const u32 m = carry_mask(c);  // borrow: add p back
    carry_t c2;
    r.v[0] = add_co(r.v[0], P[0] & m, c2);
#pragma unroll
    for (int k = 1; k < 12; k++) r.v[k] = addc_co(r.v[k], P[k] & m, c2);
    return r;
}
__device__ __forceinline__ F zero() { F r; for (int k = 0; k < 12; k++) r.v[k] = 0; return r; }
__device__ __forceinline__ F neg(const F& a) { return sub(zero(), a); }
__device__ __forceinline__ bool is_zero(const F& a) { u32 nz = 0; for (int k = 0; k < 12; k++) nz |= a.v[k]; return nz == 0; }
__device__ __forceinline__ bool eq(const F& a, const F& b) { u32 d = 0; for (int k = 0; k < 12; k++) d |= a.v[k] ^ b.v[k]; return d == 0; }
__device__ __forceinline__ F inv(const F& a) { F r; fp_invert_boundary(a.v, r.v); return r; }
__device__ __forceinline__ bool below_p(const u32* x) {
    const u32 P[12] = BLS381_P_LIMBS;
    for (int k = 11; k >= 0; k--) { if (x[k] < P[k]) return true; if (x[k] > P[k]) return false; }
    return false;
}
// canonical limbs <-> Montgomery form
__device__ __forceinline__ F to_mont(const u32* x) {
    const F R2 = {{0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                   0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u}};
    F a; for (int k = 0; k < 12; k++) a.v[k] = x[k];
    return mul(a, R2);
}
__device__ __forceinline__ F from_mont(const F& a) { F one = zero(); one.v[0] = 1; return mul(a, one); }
__device__ __forceinline__ void store(u32* dst, const F& a) { for (int k = 0; k < 12; k++) dst[k] = a.v[k]; }

// a^((p+1)/4): the square root of a when a is a square (p = 3 mod 4).  T: any of the one-lane field types
// (their mul / sqr are found through the argument's namespace).
template <class T> __device__ __noinline__ T sqrt_chain(const T& a) {
    // (p + 1) / 4, most significant limb first; top set bit of the top limb is bit 26
    const u32 E[12] = {0x0680447au, 0x8e5ff9a6u, 0x92c6e9edu, 0x90d2eb35u, 0xd91dd2e1u, 0x3ce144afu,
                       0xd9cc34a8u, 0x3dac3d89u, 0x07aaffffu, 0xac54ffffu, 0xee7fbfffu, 0xffffeaabu};
    // four exponent bits at a time: 378 squarings and 92 + 14 products, against 228 products bit by bit
    T t[16];
    t[1] = a;
    for (int k = 2; k < 16; k++) t[k] = mul(t[k - 1], a);
    T r = a;
    bool started = false;
    for (int w = 0; w < 12; w++)
        for (int s = 28; s >= 0; s -= 4) {
            const u32 nib = (E[w] >> s) & 15u;
            if (!started) {  // the leading non-zero digit
                if (nib) { r = t[nib]; started = true; }
                continue;
            }
#pragma unroll 1
            for (int q = 0; q < 4; q++) r = sqr(r);
            if (nib) r = mul(r, t[nib]);
        }
    return r;
}
// (the chain runs on the 28-bit digit form, see fp28_field.h; a is in the 12-limb Montgomery form and so is the result)
__device__ __forceinline__ F sqrt_candidate(const F& a) {
    F r;
    g28::to_boundary(sqrt_chain(g28::from_boundary(a.v)), r.v);
    return r;
}
// sqrt in Fq: ok = false when a is not a square
__device__ __forceinline__ F sqrt_fq(const F& a, bool& ok) {
    F r = sqrt_candidate(a);
    ok = eq(sqr(r), a);
    return r;
}

__device__ __forceinline__ F2 mul(const F2& a, const F2& b) {
    return {sub(mul(a.c0, b.c0), mul(a.c1, b.c1)), add(mul(a.c0, b.c1), mul(a.c1, b.c0))};
}
__device__ __forceinline__ F2 sqr(const F2& a) { return mul(a, a); }
__device__ __forceinline__ F2 add(const F2& a, const F2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
__device__ __forceinline__ F2 sub(const F2& a, const F2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
__device__ __forceinline__ F2 mul_xi(const F2& a) { return {sub(a.c0, a.c1), add(a.c0, a.c1)}; }  // (1 + u) a
__device__ __forceinline__ bool is_zero(const F2& a) { return is_zero(a.c0) && is_zero(a.c1); }
__device__ __forceinline__ F2 inv(const F2& a) {  // conj(a) / norm(a)
    F n = inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, n), neg(mul(a.c1, n))};
}
__device__ __forceinline__ bool sgn0(const F& canonical) { return canonical.v[0] & 1u; }

// FqSqrtGenerator::run_once, /root/reference/src/fields/fq_target.rs:296-323
__global__ __launch_bounds__(64) void fq_sqrt_kernel(const u32* x, const uint8_t* sgn, int n, u32* out, uint8_t* err) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* xi = x + (size_t)i * 12;
    bool is_square;
    F r = from_mont(sqrt_fq(to_mont(xi), is_square));
    bool ok = is_square && below_p(xi);
    const bool want = sgn[i] & 1u;
    if (sgn0(r) != want) r = neg(r);
    ok = ok && sgn0(r) == want;  // root 0 cannot have its parity bit set
    store(out + (size_t)i * 12, ok ? r : zero());
    err[i] = ok ? 0 : 1;
}

// Fq2InverseGenerator::run_once, /root/reference/src/fields/fq2_target.rs:347-369 (zero maps to zero)
__global__ __launch_bounds__(64) void fq2_inverse_kernel(const u32* x, int n, u32* out, uint8_t* err) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* xi = x + (size_t)i * 24;
    const bool ok = below_p(xi) && below_p(xi + 12);
    F2 a{to_mont(xi), to_mont(xi + 12)};
    F2 r = inv(a);  // the inversion maps 0 to 0
    store(out + (size_t)i * 24, ok ? from_mont(r.c0) : zero());
    store(out + (size_t)i * 24 + 12, ok ? from_mont(r.c1) : zero());
    err[i] = ok ? 0 : 1;
}

// A square root of a in Fq2 = Fq[u]/(u^2 + 1); ok = false when a is not a square
__device__ __noinline__ F2 sqrt_fq2(const F2& a, bool& ok) {
    F2 r;
    if (is_zero(a.c1)) {
        bool sq;
        const F s = sqrt_fq(a.c0, sq);  // s^2 = +-a0 (-1 is a non-residue); with the minus sign (s u)^2 = a0
        r = sq ? F2{s, zero()} : F2{zero(), s};
    } else {
        bool sq;
        const F g = sqrt_fq(add(sqr(a.c0), sqr(a.c1)), sq);  // the norm must be a square in Fq
        const u32 HALF[12] = {0xffffd556u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                              0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};  // (p + 1) / 2
        const F half = to_mont(HALF);
        // x0 = ((a0 + g) / 2)^((p+1)/4) squares to +-(a0 + g) / 2.  With the plus sign the root is (x0, a1 / (2 x0)).
        // With the minus sign (a0 - g) / 2 is the square, and since ((a0 + g) / 2)((a0 - g) / 2) = -a1^2 / 4 its root
        // is a1 / (2 x0): the same two numbers in the other order, with no second exponentiation.
        bool sq2;
        const F x0 = sqrt_fq(mul(add(a.c0, g), half), sq2);
        const F y = mul(a.c1, inv(add(x0, x0)));
        r = sq2 ? F2{x0, y} : F2{y, x0};
    }
    const F2 chk = sqr(r);
    ok = eq(chk.c0, a.c0) && eq(chk.c1, a.c1);
    return r;
}

// Fq2SqrtGenerator::run_once, /root/reference/src/fields/fq2_target.rs:401-428
__global__ __launch_bounds__(64) void fq2_sqrt_kernel(const u32* x, const uint8_t* sgn, int n, u32* out, uint8_t* err) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* xi = x + (size_t)i * 24;
    bool ok = below_p(xi) && below_p(xi + 12);
    const F2 a{to_mont(xi), to_mont(xi + 12)};
    bool is_square;
    const F2 r = sqrt_fq2(a, is_square);
    ok = ok && is_square;
    F r0 = from_mont(r.c0), r1 = from_mont(r.c1);
    const bool want = sgn[i] & 1u;
    auto sgn0_2 = [&](const F& c0, const F& c1) { return sgn0(c0) || (is_zero(c0) && sgn0(c1)); };
    if (sgn0_2(r0, r1) != want) { r0 = neg(r0); r1 = neg(r1); }
    ok = ok && sgn0_2(r0, r1) == want;
    store(out + (size_t)i * 24, ok ? r0 : zero());
    store(out + (size_t)i * 24 + 12, ok ? r1 : zero());
    err[i] = ok ? 0 : 1;
}

// Fq6InverseGenerator::run_once, /root/reference/src/fields/fq6_target.rs:350-371; coefficients in
// MyFq6 order [c0.c0, c1.c0, c2.c0, c0.c1, c1.c1, c2.c1] (/root/reference/src/utils/my_fq6.rs:12-44)
__global__ __launch_bounds__(64) void fq6_inverse_kernel(const u32* x, int n, u32* out, uint8_t* err) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* xi = x + (size_t)i * 72;
    bool ok = true;
    for (int k = 0; k < 6; k++) ok = ok && below_p(xi + 12 * k);
    const F2 c0{to_mont(xi), to_mont(xi + 36)}, c1{to_mont(xi + 12), to_mont(xi + 48)}, c2{to_mont(xi + 24), to_mont(xi + 60)};
    ok = ok && !(is_zero(c0) && is_zero(c1) && is_zero(c2));  // inverse().unwrap() panics on zero
    const F2 t0 = sub(sqr(c0), mul_xi(mul(c1, c2)));
    const F2 t1 = sub(mul_xi(sqr(c2)), mul(c0, c1));
    const F2 t2 = sub(sqr(c1), mul(c0, c2));
    const F2 ninv = inv(add(mul(c0, t0), mul_xi(add(mul(c2, t1), mul(c1, t2)))));
    const F2 r0 = mul(t0, ninv), r1 = mul(t1, ninv), r2 = mul(t2, ninv);
    u32* o = out + (size_t)i * 72;
    const F z = zero();
    store(o, ok ? from_mont(r0.c0) : z); store(o + 12, ok ? from_mont(r1.c0) : z); store(o + 24, ok ? from_mont(r2.c0) : z);
    store(o + 36, ok ? from_mont(r0.c1) : z); store(o + 48, ok ? from_mont(r1.c1) : z); store(o + 60, ok ? from_mont(r2.c1) : z);
    err[i] = ok ? 0 : 1;
}

}  // namespace gen
}  // namespace bls381
