// The squaring chains of the hard part as a kernel of their own.
//
// hard_part_native (/root/reference/src/native/final_exponentiation.rs:44-72) raises to |x| five times through
// pow_native (/root/reference/src/native/fin_exp_helpers.rs:60-82): 63 squarings and 5 multiplications each.
// a^|x| is also the product of a^(2^k) over the set bits k of |x| = 0xd201000000010000, so one uninterrupted run of
// 63 squarings that keeps the six values a^(2^16), a^(2^48), a^(2^57), a^(2^60), a^(2^62), a^(2^63) does the same
// work with every multiplication moved behind it; field elements are canonical, so the product is pow_native's
// value bit for bit.  After the easy part every value lies in the cyclotomic subgroup, where a squaring is three
// Fq4 squarings (Granger-Scott).  This kernel runs such a chain with the element in registers:
//
//   a workgroup of three waves owns 64 elements; wave w holds Fq4 pair w of each (as Fq2 coefficients of the ark
//   tower: pair 0 = (c0.c0, c1.c1), pair 1 = (c1.c0, c0.c2), pair 2 = (c0.c1, c1.c2)), one element per lane,
//   four Fp values = 56 digits in registers;
//   a round is, per lane, one Fq4 squaring (four two-product multiplications with the operands in registers) and
//   four sums; the squares of pairs 1 and 2 feed each other's sums, so waves 1 and 2 swap their four products
//   through LDS, twice a barrier per round; pair 0 feeds itself.
//   The arithmetic of a round is generated assembly with every register assigned by the generator
//   (cyc_chain_asm.h, tools/gen_chain_asm.py): the round has to fit the 168 registers that three waves per SIMD
//   leave a lane, and compiled from C++ it takes 230-244 (or spills hundreds when told to fit).  What is left to the
//   compiler is memory: loads, the LDS exchange, the stores of the kept values.
//
// Number format: the interpreter's (fp28_dev.h): 14 signed digits of 28 bits, Montgomery factor 2^392.  Values enter
// and leave as canonical 12 x u32 words that keep that factor (slot arrays, as the interpreter's programs read and
// write them with their raw loads and stores): no multiplication is spent on either side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cyc_chain_asm.h"
#include "fp28_dev.h"

namespace bls381 {

// A workgroup is TRIOS independent trios of 64 elements each.  Four trios (twelve waves) for large batches: workgroups
// of one trio leave a CU's fourth SIMD idle (65,536 elements then take 1.9 ms instead of 1.2).  One trio for batches
// that do not give every CU a four-trio workgroup: 16,384 elements are 64 workgroups of four trios on 64 of the 256
// CUs (1.1 ms) or 256 of one trio, a wave on three SIMDs of every CU (0.46 ms).
enum { CHAIN_SNAPSHOTS = 6, CHAIN_ROUNDS = 63, CHAIN_TRIOS_LARGE = 4, CHAIN_ONE_TRIO_UP_TO = 40960 };
// bit k set: the value after k squarings is kept (the set bits of |x|)
#define BLS381_CHAIN_KEEP 0xd201000000010000ull

struct ChainArgs {
    const uint4* gin;  // 12 slots: the element (ark flat order), slot array with element stride npad
    uint4* gout;       // 6 x 12 slots: the kept values, in the order they are produced
    int n;
    int npad;
};

typedef u32 chain_v16 __attribute__((ext_vector_type(16)));  // one Fp value: 14 digits and two spare registers

// clobbers of the assembly blocks: their scratch registers
#define BLS381_CHAIN_CLOBBERS                                                                                             \
    "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", \
        "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156",     \
        "v157", "v158", "v159", "vcc"
// the eight values in the registers the generator assumes (all of them in and out: the spare registers behind each
// value are scratch), and the constants
#define BLS381_CHAIN_OPERANDS                                                                                       \
    "+{v[0:15]}"(x0), "+{v[16:31]}"(x1), "+{v[32:47]}"(y0), "+{v[48:63]}"(y1), "+{v[64:79]}"(T0), "+{v[80:95]}"(T1), \
        "+{v[96:111]}"(XY0), "+{v[112:127]}"(XY1)

template <int CHAIN_TRIOS>
__global__ __launch_bounds__(192 * CHAIN_TRIOS) __attribute__((amdgpu_waves_per_eu(3, 3))) void cyc_chain_kernel(ChainArgs A) {
    // what waves 1 and 2 hand each other: [wave - 1][value 0..3][16-byte piece 0..3][lane]
    __shared__ uint4 xbuf_all[CHAIN_TRIOS * 2 * 4 * 4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // wave w of the workgroup: trio w % 4, pair w / 4 (the three waves of a trio then sit on one SIMD's neighbours'
    // turns: consecutive waves of a workgroup go to consecutive SIMDs, so every SIMD gets one wave of each pair)
    const int trio = wave_all % CHAIN_TRIOS, wave = wave_all / CHAIN_TRIOS;
    uint4* const xbuf = xbuf_all + (size_t)trio * 2 * 4 * 4 * 64;
    const long e_raw = ((long)blockIdx.x * CHAIN_TRIOS + trio) * 64 + lane;
    const bool live = e_raw < A.n;
    const size_t e = live ? (size_t)e_raw : 0;  // idle lanes compute on the first element and store nothing
    // Fq2 coefficients (ark flat order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) of this wave's pair
    const int xq = wave == 0 ? 0 : (wave == 1 ? 3 : 1);
    const int yq = wave == 0 ? 4 : (wave == 1 ? 2 : 5);

    auto load = [&](int slot) {
        const uint4* p = A.gin + ((size_t)slot * 3) * A.npad + e;
        const uint4 a = p[0], b = p[A.npad], c = p[2 * (size_t)A.npad];
        const u32 l[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        u32 d[D28];
        unpack28(l, d);
        chain_v16 v = {d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13], 0u, 0u};
        return v;
    };
    chain_v16 x0 = load(2 * xq), x1 = load(2 * xq + 1), y0 = load(2 * yq), y1 = load(2 * yq + 1);
    chain_v16 T0 = {}, T1 = {}, XY0 = {}, XY1 = {};
    // s40..s53 the digits of p, s54 = -1/p mod 2^28, s55 the quotient estimate's multiplier
    const u32 PD[D28] = BLS381_P28_DIGITS;
    const chain_v16 K = {PD[0], PD[1], PD[2], PD[3], PD[4], PD[5], PD[6], PD[7], PD[8], PD[9], PD[10], PD[11], PD[12], PD[13], BLS381_N28_DEV, 40323u};

    // A round, from the four products T = (x - y)(x - xi y) and XY = x y of the pair that feeds this one
    // (Granger-Scott: even = T + XY + xi XY, odd = 2 XY):
    //   pairs 0 and 2:  x' = 3 even - 2 x,     y' = 3 odd + 2 y
    //   pair 1:         x' = 3 xi odd + 2 x,   y' = 3 even - 2 y
    int snap = 0;
#pragma unroll 1
    for (int round = 1; round <= CHAIN_ROUNDS; round++) {
        asm volatile(BLS381_CHAIN_ASM_SQUARE : BLS381_CHAIN_OPERANDS : "{s[40:55]}"(K) : BLS381_CHAIN_CLOBBERS);
        // waves 1 and 2 swap their products
        if (wave != 0) {
            uint4* w = xbuf + (size_t)(wave - 1) * 4 * 4 * 64 + lane;
            const chain_v16* v[4] = {&T0, &T1, &XY0, &XY1};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const chain_v16 t = *v[k];
                w[(k * 4 + 0) * 64] = make_uint4(t[0], t[1], t[2], t[3]);
                w[(k * 4 + 1) * 64] = make_uint4(t[4], t[5], t[6], t[7]);
                w[(k * 4 + 2) * 64] = make_uint4(t[8], t[9], t[10], t[11]);
                w[(k * 4 + 3) * 64] = make_uint4(t[12], t[13], t[14], t[15]);  // (the spare registers ride along: no moves)
            }
        }
        __syncthreads();
        if (wave != 0) {
            const uint4* r = xbuf + (size_t)(2 - wave) * 4 * 4 * 64 + lane;
            chain_v16* v[4] = {&T0, &T1, &XY0, &XY1};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 a = r[(k * 4 + 0) * 64], b = r[(k * 4 + 1) * 64], c = r[(k * 4 + 2) * 64], d = r[(k * 4 + 3) * 64];
                const chain_v16 t = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
                *v[k] = t;
            }
        }
        __syncthreads();  // (the next round's products may not land before every wave has read this round's)
        if (wave == 1) {  // (the same for every lane of the wave)
            asm volatile(BLS381_CHAIN_ASM_SUMS_1 : BLS381_CHAIN_OPERANDS : "{s[40:55]}"(K) : BLS381_CHAIN_CLOBBERS);
        } else {
            asm volatile(BLS381_CHAIN_ASM_SUMS_02 : BLS381_CHAIN_OPERANDS : "{s[40:55]}"(K) : BLS381_CHAIN_CLOBBERS);
        }

        if ((BLS381_CHAIN_KEEP >> round) & 1ull) {  // (the same for every lane)
            auto store = [&](int slot, const chain_v16& v) {
                u32 c[D28], l[12];
#pragma unroll
                for (int i = 0; i < D28; i++) c[i] = v[i];
                canon28(c);
                pack28(c, l);
                if (live) {
                    uint4* p = A.gout + ((size_t)(snap * 12 + slot) * 3) * A.npad + e;
                    p[0] = make_uint4(l[0], l[1], l[2], l[3]);
                    p[A.npad] = make_uint4(l[4], l[5], l[6], l[7]);
                    p[2 * (size_t)A.npad] = make_uint4(l[8], l[9], l[10], l[11]);
                }
            };
            store(2 * xq, x0); store(2 * xq + 1, x1); store(2 * yq, y0); store(2 * yq + 1, y1);
            snap++;
        }
    }
}

}  // namespace bls381
