// Device-side BLS12-381 base-field primitives: 12 x 32-bit limbs, Montgomery form with
// R = 2^384 (the in-memory form of ark_ff's 6 x u64 Fq, /root/reference/src/native/miller_loop.rs:1),
// built on 32x32->64 multiply-accumulates (v_mad_u64_u32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bls381 {

typedef uint32_t u32;
typedef uint64_t u64;

// modulus limbs as immediates: they become literal operands, not loads
#define BLS381_P_LIMBS                                                                               \
    { 0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu,      \
      0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau }
#define BLS381_N0_DEV 0xfffcfffdu

struct Fp {
    u32 l[12];
};

// Portable carry helpers (used by the plain-C reference forms only).
__device__ __forceinline__ u32 addc(u32 a, u32 b, u32& carry) {
    u64 s = (u64)a + b + carry;
    carry = (u32)(s >> 32);
    return (u32)s;
}
__device__ __forceinline__ u32 subb(u32 a, u32 b, u32& borrow) {
    u64 d = (u64)a - b - borrow;
    borrow = (u32)(d >> 63);
    return (u32)d;
}

// Carry chains on the hardware carry bit.  The carry lives in an SGPR pair the compiler allocates
// (one bit per lane), so consecutive links are single VALU instructions and the scheduler is free
// to interleave independent chains.
typedef u64 carry_t;
__device__ __forceinline__ u32 add_co(u32 a, u32 b, carry_t& c) {
    u32 r;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 addc_co(u32 a, u32 b, carry_t& c) {
    u32 r;
    asm("v_addc_co_u32 %0, %1, %2, %3, %1" : "=v"(r), "+s"(c) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 sub_co(u32 a, u32 b, carry_t& c) {
    u32 r;
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 subb_co(u32 a, u32 b, carry_t& c) {
    u32 r;
    asm("v_subb_co_u32 %0, %1, %2, %3, %1" : "=v"(r), "+s"(c) : "v"(a), "v"(b));
    return r;
}
// all-ones where the chain ended with a borrow / carry, else zero
__device__ __forceinline__ u32 carry_mask(carry_t& c) {
    u32 r;
    asm("v_subb_co_u32 %0, %1, 0, 0, %1" : "=v"(r), "+s"(c));
    return r;
}
// (mask & a) | (~mask & b)
__device__ __forceinline__ u32 bitselect(u32 mask, u32 a, u32 b) {
    u32 r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

// n-limb a -= q when a >= q (one round of the reduction ladder); q >= 2^(32(n-1)) is not required
template <int N>
__device__ __forceinline__ void cond_sub(u32* a, const u32* q) {
    u32 t[N];
    carry_t c;
    t[0] = sub_co(a[0], q[0], c);
#pragma unroll
    for (int i = 1; i < N; i++) t[i] = subb_co(a[i], q[i], c);
    const u32 keep = carry_mask(c);  // borrow: a < q, keep a
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = bitselect(keep, a[i], t[i]);
}

// r = a - p if a >= p else a, for a < 2p
__device__ __forceinline__ void fp_cond_sub_p(u32* a) {
    const u32 P[12] = BLS381_P_LIMBS;
    cond_sub<12>(a, P);
}

// 24-limb product t = a * b (operand scanning)
__device__ __forceinline__ void mul_wide(const u32* a, const u32* b, u32* t) {
#pragma unroll
    for (int k = 0; k < 24; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            u64 x = (u64)a[i] * b[j] + t[i + j] + c;
            t[i + j] = (u32)x;
            c = x >> 32;
        }
        t[i + 12] = (u32)c;
    }
}

// Montgomery reduction of a 24-limb t < p * 2^384: r = t / 2^384 mod p, canonical
__device__ __forceinline__ void mont_redc(u32* t, u32* r) {
    const u32 P[12] = BLS381_P_LIMBS;
    u32 carry = 0;  // carry into limb i + 12 left over from the previous round
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u32 m = t[i] * BLS381_N0_DEV;
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            u64 x = (u64)m * P[j] + t[i + j] + c;
            t[i + j] = (u32)x;
            c = x >> 32;
        }
        u64 s = (u64)t[i + 12] + c + carry;
        t[i + 12] = (u32)s;
        carry = (u32)(s >> 32);
    }
    // t < p * 2^384 keeps the total below 2p * 2^384 < 2^768, so no carry leaves limb 23
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = t[12 + i];
    fp_cond_sub_p(r);
}

// Reference form: separate product and reduction in plain C (kept for cross-checking the
// hand-scheduled multiplier below on the GPU).
__device__ __forceinline__ void fp_mul_plain(const u32* a, const u32* b, u32* r) {
    u32 t[24];
    mul_wide(a, b, t);
    mont_redc(t, r);
}

// fp_mul (the hand-scheduled Montgomery multiplier) and the literal-operand reduction blocks are
// generated: fp_asm_gen.h.
__device__ __forceinline__ void fp_mul(const u32* a, const u32* b, u32* r);

}  // namespace bls381
