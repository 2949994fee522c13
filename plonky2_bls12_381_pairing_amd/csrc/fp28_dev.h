// Base-field arithmetic of the lane-group interpreter: 14 signed digits of 28 bits per value,
// Montgomery form with R' = 2^392, lazily reduced.
//
// Why not the boundary's 12 x 32-bit limbs inside the interpreter: with full 32-bit limbs every
// v_mad_u64_u32 needs a second instruction to catch its carry, and a sum needs a carry chain plus
// conditional subtractions of p.  With 28-bit digits a 64-bit accumulator absorbs a whole column of
// a two-product sum plus its reduction (42 products of magnitude below 2^58) with no carry
// handling: one v_mad per multiply-accumulate.  Sums are 14 independent 32-bit additions with no
// carries at all: digits are signed and may grow to 2^31 before a sweep renormalises them, and the
// 11 spare bits (2^392 / p > 2500) make modular reduction of sums a rare, explicit operation.
// A value is   v = sum d[i] * 2^(28 i),  d[i] signed 32-bit,  possibly negative as a whole.
// Values enter and leave in the boundary form (12 x u32, Montgomery 2^384, canonical): CK_IN repacks,
// the program multiplies by 2^400 resp. 2^384 to change the Montgomery factor, CK_OUT brings the
// result into [0, p) and repacks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp_dev.h"

namespace bls381 {

enum { D28 = 14 };
#define BLS381_MASK28 0x0fffffffu
#define BLS381_N28_DEV 0xffcfffdu
#define BLS381_P28_DIGITS                                                                             \
    { 0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u, 0xf38512bu, \
      0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u }

// 12 x u32 little-endian limbs -> 14 digits of 28 bits
__device__ __forceinline__ void unpack28(const u32* l, u32* d) {
#pragma unroll
    for (int i = 0; i < D28; i++) {
        const int bit = 28 * i, w = bit >> 5, s = bit & 31;
        u32 v = l[w] >> s;
        if (s > 4 && w + 1 < 12) v |= l[w + 1] << (32 - s);
        d[i] = v & BLS381_MASK28;
    }
}
// 14 normalised digits (value below 2^384) -> 12 x u32
__device__ __forceinline__ void pack28(const u32* d, u32* l) {
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / 28, s = bit - 28 * i;  // limb w starts inside digit i at bit s
        u32 v = d[i] >> s;
        v |= d[i + 1] << (28 - s);
        if (28 - s + 28 < 32 && i + 2 < D28) v |= d[i + 2] << (56 - s);
        l[w] = v;
    }
}

// r = (a * b + c * d) / 2^392 + m * p / 2^392, an integer in (-p, 2p) congruent to
// (a * b + c * d) / 2^392 when |a * b| + |c * d| < 2500 p^2.  Digits of the operands are signed with
// |a[i]| * |b[j]| + |c[i]| * |d[j]| <= 8 * 2^56 (so a column of 28 products, 14 reduction products
// and the carry stays inside the signed 64-bit accumulator).  r has digits 0..12 in [0, 2^28) and a
// signed top digit.  Finely integrated product scanning: one accumulator, shifted down 28 bits per
// column.
template <bool TWO>
__device__ __forceinline__ void fp28_mulacc(const u32* a, const u32* b, const u32* c, const u32* d, u32* r) {
    const u32 P[D28] = BLS381_P28_DIGITS;
    u32 m[D28];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < D28; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)(int)a[i] * (int)b[k - i];
            if (TWO) acc += (int64_t)(int)c[i] * (int)d[k - i];
        }
#pragma unroll
        for (int j = 0; j < k; j++) acc += (int64_t)((u64)m[j] * P[k - j]);
        m[k] = ((u32)acc * BLS381_N28_DEV) & BLS381_MASK28;
        acc += (int64_t)((u64)m[k] * P[0]);
        acc >>= 28;
    }
#pragma unroll
    for (int k = D28; k < 2 * D28 - 1; k++) {
#pragma unroll
        for (int i = k - D28 + 1; i < D28; i++) {
            acc += (int64_t)(int)a[i] * (int)b[k - i];
            if (TWO) acc += (int64_t)(int)c[i] * (int)d[k - i];
            acc += (int64_t)((u64)m[i] * P[k - i]);
        }
        r[k - D28] = (u32)acc & BLS381_MASK28;
        acc >>= 28;
    }
    r[D28 - 1] = (u32)acc;
}
__device__ __forceinline__ void fp28_mul(const u32* a, const u32* b, u32* r) { fp28_mulacc<false>(a, b, a, b, r); }
__device__ __forceinline__ void fp28_dot2(const u32* a, const u32* b, const u32* c, const u32* d, u32* r) {
    fp28_mulacc<true>(a, b, c, d, r);
}

// Signed carry sweep: leaves digits 0..12 in [0, 2^28) and the rest, sign included, in digit 13.
__device__ __forceinline__ void normalise28(u32* d) {
#pragma unroll
    for (int i = 0; i < D28 - 1; i++) {
        const int carry = (int)d[i] >> 28;
        d[i] &= BLS381_MASK28;
        d[i + 1] += (u32)carry;
    }
}

// acc[i] += x[i] ^ m for the 14 digits: one v_xad_u32 each (the compiler emits xor + add).
__device__ __forceinline__ void xor_add28(u32* acc, const u32* x, u32 m) {
    asm("v_xad_u32 %0, %14, %28, %0\n\t"
        "v_xad_u32 %1, %15, %28, %1\n\t"
        "v_xad_u32 %2, %16, %28, %2\n\t"
        "v_xad_u32 %3, %17, %28, %3\n\t"
        "v_xad_u32 %4, %18, %28, %4\n\t"
        "v_xad_u32 %5, %19, %28, %5\n\t"
        "v_xad_u32 %6, %20, %28, %6\n\t"
        "v_xad_u32 %7, %21, %28, %7\n\t"
        "v_xad_u32 %8, %22, %28, %8\n\t"
        "v_xad_u32 %9, %23, %28, %9\n\t"
        "v_xad_u32 %10, %24, %28, %10\n\t"
        "v_xad_u32 %11, %25, %28, %11\n\t"
        "v_xad_u32 %12, %26, %28, %12\n\t"
        "v_xad_u32 %13, %27, %28, %13"
        : "+&v"(acc[0]), "+&v"(acc[1]), "+&v"(acc[2]), "+&v"(acc[3]), "+&v"(acc[4]), "+&v"(acc[5]), "+&v"(acc[6]),
          "+&v"(acc[7]), "+&v"(acc[8]), "+&v"(acc[9]), "+&v"(acc[10]), "+&v"(acc[11]), "+&v"(acc[12]), "+&v"(acc[13])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
          "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(m));
}

// acc[i] = (x[i] ^ m) + c: the first term of a sum, c = the sum's count of negated terms.
__device__ __forceinline__ void xor_add28_first(u32* acc, const u32* x, u32 m, u32 c) {
    asm("v_xad_u32 %0, %14, %28, %29\n\t"
        "v_xad_u32 %1, %15, %28, %29\n\t"
        "v_xad_u32 %2, %16, %28, %29\n\t"
        "v_xad_u32 %3, %17, %28, %29\n\t"
        "v_xad_u32 %4, %18, %28, %29\n\t"
        "v_xad_u32 %5, %19, %28, %29\n\t"
        "v_xad_u32 %6, %20, %28, %29\n\t"
        "v_xad_u32 %7, %21, %28, %29\n\t"
        "v_xad_u32 %8, %22, %28, %29\n\t"
        "v_xad_u32 %9, %23, %28, %29\n\t"
        "v_xad_u32 %10, %24, %28, %29\n\t"
        "v_xad_u32 %11, %25, %28, %29\n\t"
        "v_xad_u32 %12, %26, %28, %29\n\t"
        "v_xad_u32 %13, %27, %28, %29"
        : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]), "=&v"(acc[6]),
          "=&v"(acc[7]), "=&v"(acc[8]), "=&v"(acc[9]), "=&v"(acc[10]), "=&v"(acc[11]), "=&v"(acc[12]), "=&v"(acc[13])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
          "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(m), "v"(c));
}

// acc[i] = x[i] * k (first term of a sum) and acc[i] += x[i] * k: one v_mad_u64_u32 per digit.  Only the low
// halves of the accumulators are digits (the product of a digit and a small signed coefficient taken
// mod 2^32, exact while the sum stays inside 32 bits); the high halves are never read.
__device__ __forceinline__ void coef_mul28(u64* acc, const u32* x, u32 k, u64 z0, u64 zm, u64 zt) {
    // acc[i] = x[i] * k + z: z0 for digit 0, zm for digits 1..12, zt for the top digit
    asm("v_mad_u64_u32 %0, vcc, %14, %28, %29\n\t"
        "v_mad_u64_u32 %1, vcc, %15, %28, %30\n\t"
        "v_mad_u64_u32 %2, vcc, %16, %28, %30\n\t"
        "v_mad_u64_u32 %3, vcc, %17, %28, %30\n\t"
        "v_mad_u64_u32 %4, vcc, %18, %28, %30\n\t"
        "v_mad_u64_u32 %5, vcc, %19, %28, %30\n\t"
        "v_mad_u64_u32 %6, vcc, %20, %28, %30\n\t"
        "v_mad_u64_u32 %7, vcc, %21, %28, %30\n\t"
        "v_mad_u64_u32 %8, vcc, %22, %28, %30\n\t"
        "v_mad_u64_u32 %9, vcc, %23, %28, %30\n\t"
        "v_mad_u64_u32 %10, vcc, %24, %28, %30\n\t"
        "v_mad_u64_u32 %11, vcc, %25, %28, %30\n\t"
        "v_mad_u64_u32 %12, vcc, %26, %28, %30\n\t"
        "v_mad_u64_u32 %13, vcc, %27, %28, %31"
        : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]), "=&v"(acc[6]),
          "=&v"(acc[7]), "=&v"(acc[8]), "=&v"(acc[9]), "=&v"(acc[10]), "=&v"(acc[11]), "=&v"(acc[12]), "=&v"(acc[13])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
          "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(k), "v"(z0), "v"(zm), "v"(zt)
        : "vcc");
}
__device__ __forceinline__ void coef_mad28(u64* acc, const u32* x, u32 k) {
    asm("v_mad_u64_u32 %0, vcc, %14, %28, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %15, %28, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %16, %28, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %17, %28, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %18, %28, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %19, %28, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %20, %28, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %21, %28, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %22, %28, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %23, %28, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %24, %28, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %25, %28, %11\n\t"
        "v_mad_u64_u32 %12, vcc, %26, %28, %12\n\t"
        "v_mad_u64_u32 %13, vcc, %27, %28, %13"
        : "+&v"(acc[0]), "+&v"(acc[1]), "+&v"(acc[2]), "+&v"(acc[3]), "+&v"(acc[4]), "+&v"(acc[5]), "+&v"(acc[6]),
          "+&v"(acc[7]), "+&v"(acc[8]), "+&v"(acc[9]), "+&v"(acc[10]), "+&v"(acc[11]), "+&v"(acc[12]), "+&v"(acc[13])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
          "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(k)
        : "vcc");
}

// Digits of magnitude up to 7 * 2^28, |value| below 2500 p  ->  normalised, value in (-0.2 p, 1.2 p):
// subtracts q * p, q = an estimate of value / p from the top digit (the carries still owed to it
// move it by at most 8, 1e-4 of p's top digit), then sweeps.
__device__ __forceinline__ void reduce28(u32* d) {
    const u32 P[D28] = BLS381_P28_DIGITS;
    const int q = __mulhi((int)d[D28 - 1], 40323);  // floor(top * floor(2^32 / (top digit of p + 1)) / 2^32)
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < D28; i++) {
        c += (int64_t)q * (int)P[i];
        d[i] -= (i < D28 - 1) ? ((u32)c & BLS381_MASK28) : (u32)c;  // the top digit takes what is left, sign included
        c >>= 28;
    }
    normalise28(d);
}

// Normalised value in (-p, 2p) -> [0, p)
__device__ __forceinline__ void canon28(u32* d) {
    const u32 P[D28] = BLS381_P28_DIGITS;
    // s = +p when the value is negative, -p otherwise; keep value + s when that lands in [0, p)
    const u32 neg = (u32)((int)d[D28 - 1] >> 31);
    u32 t[D28];
    int carry = 0;
#pragma unroll
    for (int i = 0; i < D28; i++) {
        const int pi = (int)((P[i] ^ ~neg) + (~neg & 1u));  // +P[i] or -P[i]
        int v = (int)d[i] + pi + carry;
        if (i < D28 - 1) { carry = v >> 28; t[i] = (u32)v & BLS381_MASK28; }
        else t[i] = (u32)v;
    }
    // negative input: always take t (= v + p).  Otherwise take t (= v - p) unless it went negative.
    const u32 take = neg | ~(u32)((int)t[D28 - 1] >> 31);
#pragma unroll
    for (int i = 0; i < D28; i++) d[i] = (t[i] & take) | (d[i] & ~take);
}

}  // namespace bls381
