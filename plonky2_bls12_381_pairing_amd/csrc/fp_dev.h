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

// r = a - p if a >= p else a, for a < 2p
__device__ __forceinline__ void fp_cond_sub_p(u32* a) {
    const u32 P[12] = BLS381_P_LIMBS;
    u32 t[12], bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = subb(a[i], P[i], bw);
    u32 keep = 0u - bw;  // all ones when a < p
#pragma unroll
    for (int i = 0; i < 12; i++) a[i] = (a[i] & keep) | (t[i] & ~keep);
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

__device__ __forceinline__ void fp_mul(const u32* a, const u32* b, u32* r) {
    u32 t[24];
    mul_wide(a, b, t);
    mont_redc(t, r);
}

}  // namespace bls381
