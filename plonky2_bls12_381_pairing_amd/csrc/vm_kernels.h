// gfx950 kernels: the lane-group interpreter, layout conversion, Fp inversion.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fp_dev.h"
#include "fp_asm_gen.h"
#include "fp28_dev.h"
#include "vm_format.h"

namespace bls381 {

// Global slot arrays (gin, gout: boundary form, 12 x u32 per value) are structure-of-arrays:
// slot s, 16-byte chunk c (3 per Fp), element e:
//   uint4 index = (s * 3 + c) * npad + e          -> a wave reads 64 consecutive uint4
// The spill array holds interpreter values (14 digits): the same three chunks plus an 8-byte tail,
//   uint2 index = s * npad + e   in a second region.
// LDS slots use the spill array's shape with the workgroup's group count in place of npad.
struct VmArgs {
    const uint4* blocks;  // packed descriptors, 64 per block (see PackedDesc)
    int nblocks;
    int nsteps;
    int lds_slots;
    const uint4* consts;  // C_COUNT rows of 4 uint4: 14 digits, zero padded
    const uint4* gin;
    uint4* gout;
    uint4* gtmp;
    uint2* gtmp_tail;
    int n;     // number of groups (pairings) with work
    const int* n_dev;  // when set: a count in device memory that further limits n (workgroups beyond it exit at once)
    int npad;  // element stride of the global slot arrays
    unsigned long long* prof;  // TAG_PROFILE only: [0..16) cycles per step kind, [16..32) step counts
    uint4* trash;  // 64 x 4 uint4: where the lanes of a copy step that have no copy of their own store
};

enum { VM_THREADS = 256 };
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef const u32x2_t __attribute__((address_space(1))) * global_u32x2_ptr;

__device__ __forceinline__ void unpack3(const uint4& a, const uint4& b, const uint4& c, u32* x) {
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
}

// LDS bytes a workgroup of the interpreter needs
__host__ __device__ inline size_t vm_lds_bytes(int lds_slots, int K) {
    const int gpw = 4 * (64 / K);
    return (size_t)lds_slots * gpw * 56;
}

// TAG only gives the Miller-loop, final-exponentiation and auxiliary launches distinct kernel
// names in profiler output; the code is identical.  TAG_PROFILE additionally has the first lane of
// the grid add the shader-clock cycles of every step to A.prof, by step kind: 0 product, 1 sum,
// 2 sum + carry sweep, 3 two-product, 4 + CopyKind copies, 9 sum + reduction, 10 multiplier step with summed operands,
// 11 n-product step.
enum { TAG_MILLER = 0, TAG_FINAL_EXP = 1, TAG_AUX = 2, TAG_PROFILE = 3 };

template <int K, int TAG>
__global__ __launch_bounds__(VM_THREADS) void vm_kernel(VmArgs A) {
    constexpr int G = 64 / K;      // groups per wave
    constexpr int GPW = 4 * G;     // groups per workgroup
    extern __shared__ uint4 lds[];
    uint2* lds_tail = reinterpret_cast<uint2*>(lds + (size_t)A.lds_slots * 3 * GPW);

    const int n_groups = A.n_dev ? min(A.n, *A.n_dev) : A.n;
    if ((long)blockIdx.x * GPW >= n_groups) return;  // uniform over the workgroup
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int role_raw = lane / G;
    const int gw = lane % G;
    const int g = wave * G + gw;
    const int role = role_raw < K ? role_raw : 0;
    const long e = (long)blockIdx.x * GPW + g;
    const bool lane_on = (role_raw < K) && (e < n_groups);

    // the ZERO slot (second to last) of every group
    for (int i = tid; i < 3 * GPW; i += VM_THREADS) lds[(size_t)(A.lds_slots - 2) * 3 * GPW + i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < GPW; i += VM_THREADS) lds_tail[(size_t)(A.lds_slots - 2) * GPW + i] = make_uint2(0, 0);
    __syncthreads();

    auto lds_load = [&](int slot, u32* x) {
        const uint4* p = lds + (size_t)slot * 3 * GPW + g;
        uint4 a = p[0], b = p[GPW], c = p[2 * GPW];
        uint2 t = lds_tail[(size_t)slot * GPW + g];
        unpack3(a, b, c, x);
        x[12] = t.x; x[13] = t.y;
    };
    auto lds_store = [&](int slot, const u32* x) {
        uint4* p = lds + (size_t)slot * 3 * GPW + g;
        p[0] = make_uint4(x[0], x[1], x[2], x[3]);
        p[GPW] = make_uint4(x[4], x[5], x[6], x[7]);
        p[2 * GPW] = make_uint4(x[8], x[9], x[10], x[11]);
        lds_tail[(size_t)slot * GPW + g] = make_uint2(x[12], x[13]);
    };
    // Two 16-byte loads per lane fetch the descriptors of a block of SPB steps (words 0..3 and words 4..7 of
    // each); two blocks are kept ahead of the one being executed, so no step waits on descriptor memory.
    constexpr int SPB = 64 / K;
    const uint4* bptr = A.blocks + lane;
    // (loads past the last block are clamped to it rather than skipped: a conditional load would
    // become a flat load, which LDS waits cannot tell apart from LDS traffic)
    const int last_block = A.nblocks - 1;
    uint4 cur = bptr[0], cur_hi = bptr[64];
    uint4 nxt = bptr[(size_t)min(1, last_block) * 128], nxt_hi = bptr[(size_t)min(1, last_block) * 128 + 64];
    long long t_prev = 0;
    int k_prev = -1;
    int s = 0;
    // this lane's descriptor of step j of a block lives in lane j * K + role; the one for the next
    // step is fetched while the current step executes
    u32 nx = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)cur.x);
    u32 ny = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)cur.y);
    u32 nz = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)cur.z);
    for (int b = 0; b < A.nblocks; b++) {
      const uint4 ahead = bptr[(size_t)min(b + 2, last_block) * 128];
      const uint4 ahead_hi = bptr[(size_t)min(b + 2, last_block) * 128 + 64];
      const int nj = min((int)SPB, A.nsteps - s);  // steps of this block
      for (int j = 0; j < nj; j++, s++) {
        const u32 dx = nx, dy = ny, dz = nz;
        if (j + 1 < SPB) {  // a branch, not a select: the next block's registers are touched once per block only
            const int from = ((j + 1) * K + role) * 4;
            nx = (u32)__builtin_amdgcn_ds_bpermute(from, (int)cur.x);
            ny = (u32)__builtin_amdgcn_ds_bpermute(from, (int)cur.y);
            nz = (u32)__builtin_amdgcn_ds_bpermute(from, (int)cur.z);
        } else {
            asm volatile("" ::: "memory");
            nx = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)nxt.x);
            ny = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)nxt.y);
            nz = (u32)__builtin_amdgcn_ds_bpermute(role * 4, (int)nxt.z);
        }
        const int self = (j * K + role) * 4;
        const u32 hx = (u32)__builtin_amdgcn_readfirstlane((int)dx);  // step-wide fields
        const u32 type = (hx >> 2) & 7u, max_terms = (hx >> 5) & 7u, flags = (hx >> 8) & 3u;
        if (TAG == TAG_PROFILE && tid == 0 && blockIdx.x == 0) {
            const long long now = clock64();
            if (k_prev >= 0) {
                atomicAdd(A.prof + k_prev, (unsigned long long)(now - t_prev));
                atomicAdd(A.prof + 16 + k_prev, 1ull);
            }
            t_prev = now;
            k_prev = type == ST_COPY ? 4 + (int)max_terms
                                     : (type == ST_LIN ? (flags & 2u ? 9 : (flags ? 2 : 1)) : (type == ST_DOTN ? 11 : (max_terms > 1 ? 10 : (int)type)));
        }
        const bool on = lane_on && (dx & 1u);
        const u32 dst = type == ST_COPY ? dz & 0xffffu : (dx >> 16) & 0x7fu;

        // Copies carried by a multiplier or sum step.  Fill: the loads are issued now and land in LDS
        // after the step's results, their latency behind the step's arithmetic.  Spill: LDS slot ->
        // spill array, now, after the fills' loads (no step fills what it spills; waiting for the
        // loads later then never waits for these stores).
        u32 pf[D28];  // deliberately not initialised: a merge with constants would make the compiler copy
                      // the loaded registers, and so wait for the loads, right where they are issued
        u32 dw = 0;   // the descriptor word of the carried copies, fetched only by steps that carry any
        const bool carries = hx & (3u << 23);  // (one test for the steps that carry nothing: most of them)
        // The operands' third terms / a sum's terms 4 and 5 are fetched by every step (one instruction, against a test
        // and a branch to skip it), and here, before the step issues any load from global memory: the block's
        // registers came from global loads themselves, and a wait for them placed after the carried fills' loads
        // would wait for those too.  (A sum's seventh term is fetched by the sums that have one.)
        const u32 d3 = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur.w);
        u32 second_w = 0;  // a second step's sum riding in this step's descriptors: its header word
        if (hx & (1u << 10)) second_w = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.x);
        // (the spilled value's registers stay reserved until the step ends, see there: gfx9 has one counter for
        // loads and stores, and reusing a register a store still reads from means waiting for that store, which,
        // the counter being in order, means waiting for the fills' loads issued before it)
        u32 sv[D28];
        if (carries) {
            dw = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.y);
            asm volatile("" : "+v"(dw));  // nothing derived from dw is computed (and waited for) by steps that carry nothing
            if (hx & (1u << 23)) {
                // a fill may read what a spill of an earlier step wrote: drain the wave's stores first
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane_on && (dw & (1u << 19))) {
                    const u32 fk = (dw >> 7) & 3u, src = (dw >> 9) & 0x3ffu;
                    if (fk == CK_CONST) {
                        const uint4* p = A.consts + (size_t)src * 4;
                        uint4 v0 = p[0], v1 = p[1], v2 = p[2];
                        // 8 bytes, not 16: registers a load writes but nothing reads get reused at once, and
                        // the reuse then has to wait for the load
                        // (a pointer known to be global memory: laundered as a generic pointer the load would be a flat
                        // load, which counts as LDS traffic too and makes every later wait a wait for everything)
                        global_u32x2_ptr pt = (global_u32x2_ptr)(p + 3);
                        asm volatile("" : "+v"(pt));  // keeps the compiler from widening the load again
                        const u32x2_t v3 = *pt;
                        unpack3(v0, v1, v2, pf);
                        pf[12] = v3.x; pf[13] = v3.y;
                    } else {
                        const uint4* p = (fk == CK_IN ? A.gin : A.gtmp) + ((size_t)src * 3) * A.npad + e;
                        uint4 v0 = p[0], v1 = p[A.npad], v2 = p[2 * (size_t)A.npad];
                        unpack3(v0, v1, v2, pf);
                        if (fk == CK_FILL) {
                            uint2 t = A.gtmp_tail[(size_t)src * A.npad + e];
                            pf[12] = t.x; pf[13] = t.y;
                        }
                    }
                }
            }
            if (hx & (1u << 24)) {
                if (lane_on && (dw & (1u << 27))) {
                    lds_load((dw >> 20) & 0x7fu, sv);
                    const u32 t = dx >> 25;
                    uint4* p = A.gtmp + ((size_t)t * 3) * A.npad + e;
                    p[0] = make_uint4(sv[0], sv[1], sv[2], sv[3]);
                    p[A.npad] = make_uint4(sv[4], sv[5], sv[6], sv[7]);
                    p[2 * (size_t)A.npad] = make_uint4(sv[8], sv[9], sv[10], sv[11]);
                    A.gtmp_tail[(size_t)t * A.npad + e] = make_uint2(sv[12], sv[13]);
                }
            }
        }
        // Multiplier and sum steps are branch-free: idle lanes read the ZERO slot and store to TRASH.
        // A sum: dst = sum of k * x over the op's terms, k a signed 8-bit coefficient: one multiply-add per digit
        // (64-bit accumulators whose low halves are the digits; the high halves mean nothing).  The loads
        // of up to four terms are issued together so their latencies overlap.  Sums shorter than the
        // step's longest are padded with 0 * ZERO.  t01, t23, t45: the terms' descriptor words (a seventh term
        // is fetched here); B: the shifted zero the sum starts from (B * 2^28 on digit 0, B * 2^28 - B on digits
        // 1..12, -B on top); LONG = false leaves out the code for more than four terms.
        auto run_sum = [&](auto long_tag, u32 t01, u32 t23, u32 t45, int B, u32 nterms, u32 fl, u32 to) {
            constexpr bool LONG = decltype(long_tag)::value;
            auto COEF = [](u32 w) -> u32 { return (u32)((int)(w << 17) >> 24); };  // bits 7..14, sign-extended
            u64 acc[D28];
            const u64 z0 = (u32)(B << 28), zm = (u32)((B << 28) - B), zt = (u32)(-B);
            u32 t6 = 0;  // the seventh term
            auto TERM = [&](int i) -> u32 { return ((i < 2 ? t01 : (i < 4 ? t23 : (i < 6 ? t45 : t6))) >> ((i & 1) * 16)) & 0x7fffu; };
            auto terms = [&](auto first_tag, auto count_tag) {
                constexpr int T0 = decltype(first_tag)::value, N = decltype(count_tag)::value;
                u32 x[N][D28], k[N];
#pragma unroll
                for (int t = 0; t < N; t++) {
                    const u32 w = TERM(T0 + t);
                    k[t] = COEF(w);
                    lds_load(w & 0x7fu, x[t]);
                }
#pragma unroll
                for (int t = 0; t < N; t++) {
                    if (T0 + t == 0) coef_mul28(acc, x[t], k[t], z0, zm, zt);
                    else coef_mad28(acc, x[t], k[t]);
                }
            };
#define BLS381_IC(n) std::integral_constant<int, n>()
            // (uniform; written as nested two-way branches: a switch comes out of the compiler as a chain of
            // scalar flag updates)
            if (!LONG || nterms <= 4) {
                if (nterms <= 2) {
                    if (nterms <= 1) terms(BLS381_IC(0), BLS381_IC(1));
                    else terms(BLS381_IC(0), BLS381_IC(2));
                } else {
                    if (nterms == 3) terms(BLS381_IC(0), BLS381_IC(3));
                    else terms(BLS381_IC(0), BLS381_IC(4));
                }
            } else {
                if (nterms > 6) t6 = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.x);
                terms(BLS381_IC(0), BLS381_IC(4));
                if (nterms == 5) terms(BLS381_IC(4), BLS381_IC(1));
                else if (nterms == 6) terms(BLS381_IC(4), BLS381_IC(2));
                else terms(BLS381_IC(4), BLS381_IC(3));
            }
#undef BLS381_IC
            u32 r[D28];
#pragma unroll
            for (int i = 0; i < D28; i++) r[i] = (u32)acc[i];
            if (fl & 2u) reduce28(r);  // sweeps as well
            else if (fl) normalise28(r);
            lds_store(to, r);
        };
        if (type == ST_LIN) {  // most frequent first
            run_sum(std::true_type(), dy, dz, d3, (int)(dx << 16) >> 28 /* bits 12..15, signed */, max_terms, flags, dst);
        } else if (K == 6 && type == ST_DOTN) {  // (compiled into the 6-lane interpreter only)
            // dst = (sum over t of a_t * (+-b_t)) / 2^392 with one reduction.  Operand scanning: every product of
            // two digits goes to its column's 64-bit accumulator; the operands of product t + 1 are loaded while
            // product t is multiplied.  Ops with fewer products than the step's largest count are padded with
            // ZERO * ZERO.  The columns hold up to 8 x 14 products of normalised digits plus the reduction's.
            const int npairs = (int)max_terms + 1;  // uniform, 3..8
            u32 d4 = 0;
            if (npairs > 6) d4 = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.x);
            auto PAIR = [&](int t) -> u32 { return ((t < 2 ? dy : (t < 4 ? dz : (t < 6 ? d3 : d4))) >> ((t & 1) * 16)) & 0xffffu; };
            int64_t col[2 * D28];
#pragma unroll
            for (int i = 0; i < 2 * D28; i++) col[i] = 0;
            u32 pa[2][D28], pb[2][D28];
            lds_load(PAIR(0) & 0x7fu, pa[0]);
            lds_load((PAIR(0) >> 8) & 0x7fu, pb[0]);
            // (written out product by product: a loop over t would index the operand registers dynamically)
#define BLS381_PRODUCT(t)                                                                              \
    {                                                                                                  \
        if ((t) + 1 < DOTN_MAX_PAIRS && ((t) < 2 || (t) + 1 < npairs)) {                               \
            lds_load(PAIR((t) + 1) & 0x7fu, pa[((t) + 1) & 1]);                                        \
            lds_load((PAIR((t) + 1) >> 8) & 0x7fu, pb[((t) + 1) & 1]);                                 \
        }                                                                                              \
        const u32 m = (u32)((int)(PAIR(t) << 16) >> 31); /* bit 15: the product is subtracted */      \
        u32 nb[D28];                                                                                   \
        xor_add28_first(nb, pb[(t) & 1], m, m & 1u);                                                   \
        _Pragma("unroll") for (int i = 0; i < D28; i++)                                                \
            _Pragma("unroll") for (int j = 0; j < D28; j++) col[i + j] += (int64_t)(int)pa[(t) & 1][i] * (int)nb[j]; \
    }
            BLS381_PRODUCT(0) BLS381_PRODUCT(1) BLS381_PRODUCT(2)
            if (npairs > 3) { BLS381_PRODUCT(3)
                if (npairs > 4) { BLS381_PRODUCT(4)
                    if (npairs > 5) { BLS381_PRODUCT(5)
                        if (npairs > 6) { BLS381_PRODUCT(6)
                            if (npairs > 7) { BLS381_PRODUCT(7) } } } } }
#undef BLS381_PRODUCT
            // Montgomery reduction column by column: q = -col / p mod 2^28 clears the column's low 28 bits, the
            // rest moves up
            const u32 P28[D28] = BLS381_P28_DIGITS;
            u32 r[D28];
#pragma unroll
            for (int k = 0; k < D28; k++) {
                const u32 q = ((u32)col[k] * BLS381_N28_DEV) & BLS381_MASK28;
#pragma unroll
                for (int j = 0; j < D28; j++) col[k + j] += (int64_t)((u64)q * P28[j]);
                col[k + 1] += col[k] >> 28;
            }
#pragma unroll
            for (int k = D28; k < 2 * D28 - 1; k++) {
                r[k - D28] = (u32)col[k] & BLS381_MASK28;
                col[k + 1] += col[k] >> 28;
            }
            r[D28 - 1] = (u32)col[2 * D28 - 1];
            lds_store(dst, r);
        } else if (type == ST_DOT2 || type == ST_FMUL) {
            // operands: a, b (, c, d), each a signed sum of up to max_terms slots added up as they are loaded
            auto BYTE = [&](u32 word, int jop) -> u32 { return (word >> (jop * 8)) & 0xffu; };
            auto operand2 = [&](int jop, u32* v) {
                const u32 e0 = BYTE(dy, jop), e1 = BYTE(dz, jop);
                const u32 m0 = (u32)((int)(e0 << 24) >> 31), m1 = (u32)((int)(e1 << 24) >> 31);
                u32 x0[D28], x1[D28];
                lds_load(e0 & 0x7fu, x0);
                lds_load(e1 & 0x7fu, x1);
                // (x ^ m) is one short of -x for a negated term: the first term starts from their count
                xor_add28_first(v, x0, m0, 0u - (m0 + m1));
                xor_add28(v, x1, m1);
            };
            auto operand3 = [&](int jop, u32* v) {
                const u32 e0 = BYTE(dy, jop), e1 = BYTE(dz, jop), e2 = BYTE(d3, jop);
                const u32 m0 = (u32)((int)(e0 << 24) >> 31), m1 = (u32)((int)(e1 << 24) >> 31), m2 = (u32)((int)(e2 << 24) >> 31);
                u32 x0[D28], x1[D28], x2[D28];
                lds_load(e0 & 0x7fu, x0);
                lds_load(e1 & 0x7fu, x1);
                lds_load(e2 & 0x7fu, x2);
                xor_add28_first(v, x0, m0, 0u - (m0 + m1 + m2));
                xor_add28(v, x1, m1);
                xor_add28(v, x2, m2);
            };
            u32 a[D28], b[D28], r[D28];
            if (type == ST_DOT2) {
                u32 c[D28], d[D28];
                if (max_terms <= 1) {  // uniform: plain operands, only d may be negated
                    lds_load(BYTE(dy, 0) & 0x7fu, a);
                    lds_load(BYTE(dy, 1) & 0x7fu, b);
                    lds_load(BYTE(dy, 2) & 0x7fu, c);
                    lds_load(BYTE(dy, 3) & 0x7fu, d);
                    // a*b - c*d: negate d digit by digit, -d = (d ^ -1) + 1
                    const u32 m = (u32)((int)dy >> 31);  // the sign bit of d's byte
                    u32 nd[D28];
                    xor_add28_first(nd, d, m, m & 1u);
#pragma unroll
                    for (int i = 0; i < D28; i++) d[i] = nd[i];
                } else if (max_terms == 2) {
                    operand2(0, a); operand2(1, b); operand2(2, c); operand2(3, d);
                } else {
                    operand3(0, a); operand3(1, b); operand3(2, c); operand3(3, d);
                }
                fp28_dot2(a, b, c, d, r);
            } else {
                if (max_terms <= 1) {
                    lds_load(BYTE(dy, 0) & 0x7fu, a);
                    lds_load(BYTE(dy, 1) & 0x7fu, b);
                } else if (max_terms == 2) {
                    operand2(0, a); operand2(1, b);
                } else {
                    operand3(0, a); operand3(1, b);
                }
                fp28_mul(a, b, r);
            }
            lds_store(dst, r);
        } else {  // ST_COPY: every lane reads before any lane writes
            // No lane-dependent branch in here: the copy kind is the step's (uniform), and a lane without a copy of
            // its own runs the same loads and stores on harmless addresses (the constant table, the ZERO and TRASH
            // slots, the trash array).  A divergent branch anywhere in the step dispatch would have the compiler
            // linearise the whole dispatch into a chain of flag tests that every step then walks.
            const u32 kind = max_terms;  // (copy steps keep their CopyKind in the term-count field)
            const int trash_slot = A.lds_slots - 1, zero_slot = A.lds_slots - 2;
            // Element offsets into the slot arrays (below 2^32 uint4: at most 1,024 slots of 3 x 65,536), computed for
            // every lane and passed through an empty asm so that choosing between them stays a select (the compiler
            // turns a select whose operands take work to compute back into a branch).
            const u32 src = dy & 0xffffu;
            u32 off_src = src * 3u * (u32)A.npad + (u32)e, off_src_tail = src * (u32)A.npad + (u32)e;
            u32 off_dst = dst * 3u * (u32)A.npad + (u32)e, off_dst_tail = dst * (u32)A.npad + (u32)e;
            u32 off_trash = (u32)lane * 4u, off_const = src * 4u;
            asm volatile("" : "+v"(off_src), "+v"(off_src_tail), "+v"(off_dst), "+v"(off_dst_tail), "+v"(off_trash), "+v"(off_const));
            // A fill may read what an earlier spill of this wave wrote (spills themselves are not
            // waited for): drain the wave's stores first.
            if (kind == CK_FILL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            u32 v[D28];
            if (kind == CK_IN) {  // (idle lanes read the first element's slot 0: always there)
                const uint4* p = A.gin + (on ? off_src : 0u);
                uint4 v0 = p[0], v1 = p[A.npad], v2 = p[2 * (size_t)A.npad];
                u32 l[12];
                unpack3(v0, v1, v2, l);
                unpack28(l, v);
            } else if (kind == CK_FILL) {
                const uint4* p = A.gtmp + (on ? off_src : 0u);
                const uint2* pt = A.gtmp_tail + (on ? off_src_tail : 0u);
                uint4 v0 = p[0], v1 = p[A.npad], v2 = p[2 * (size_t)A.npad];
                uint2 t = *pt;
                unpack3(v0, v1, v2, v);
                v[12] = t.x; v[13] = t.y;
            } else if (kind == CK_CONST) {
                const uint4* p = A.consts + (on ? off_const : 0u);
                uint4 v0 = p[0], v1 = p[1], v2 = p[2];
                global_u32x2_ptr pt = (global_u32x2_ptr)(p + 3);
                asm volatile("" : "+v"(pt));
                const u32x2_t v3 = *pt;
                unpack3(v0, v1, v2, v);
                v[12] = v3.x; v[13] = v3.y;
            } else {
                lds_load(on ? (int)src : zero_slot, v);
            }
            // Every path consumes the loaded registers here.  Otherwise the compiler must assume a
            // load may still be in flight when other steps reuse those registers and guards them
            // with waits that also stall on unrelated stores and descriptor prefetches.
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                              "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]));
            if (kind <= CK_CONST) {
                lds_store(on ? (int)dst : trash_slot, v);
            } else {
                // idle lanes store four consecutive uint4 of the trash array, the others three slot-array chunks
                // (and, spilling, the tail)
                u32 st = on ? (u32)A.npad : 1u;
                if (kind == CK_SPILL) {
                    uint4* p = (on ? A.gtmp : A.trash) + (on ? off_dst : off_trash);
                    uint2* pt = on ? A.gtmp_tail + off_dst_tail : reinterpret_cast<uint2*>(A.trash + off_trash + 3);
                    p[0] = make_uint4(v[0], v[1], v[2], v[3]);
                    p[st] = make_uint4(v[4], v[5], v[6], v[7]);
                    p[2 * (size_t)st] = make_uint4(v[8], v[9], v[10], v[11]);
                    *pt = make_uint2(v[12], v[13]);
                } else {  // CK_OUT: below 2p -> canonical, 12 x u32
                    canon28(v);
                    u32 l[12];
                    pack28(v, l);
                    uint4* p = (on ? A.gout : A.trash) + (on ? off_dst : off_trash);
                    p[0] = make_uint4(l[0], l[1], l[2], l[3]);
                    p[st] = make_uint4(l[4], l[5], l[6], l[7]);
                    p[2 * (size_t)st] = make_uint4(l[8], l[9], l[10], l[11]);
                }
            }
        }
        if (carries) {
        if (hx & (1u << 24))
            asm volatile("" : : "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]),
                                "v"(sv[8]), "v"(sv[9]), "v"(sv[10]), "v"(sv[11]), "v"(sv[12]), "v"(sv[13]));
        if (hx & (1u << 23)) {
            // every path consumes the loaded registers here (see the copy step)
            asm volatile("" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(pf[4]), "+v"(pf[5]), "+v"(pf[6]),
                              "+v"(pf[7]), "+v"(pf[8]), "+v"(pf[9]), "+v"(pf[10]), "+v"(pf[11]), "+v"(pf[12]), "+v"(pf[13]));
            if (lane_on && (dw & (1u << 19))) {
                if (((dw >> 7) & 3u) == CK_IN) {
                    u32 l[12];
#pragma unroll
                    for (int i = 0; i < 12; i++) l[i] = pf[i];
                    unpack28(l, pf);
                }
                lds_store(dw & 0x7fu, pf);
            }
        }
        }
        // The sum of the step that followed, riding in this step's descriptors (see pack_second_sum): it runs after
        // this step's own operation has stored its result and the step's carried fill has landed (fills ride on the
        // step before their first use), as it would in a step of its own.
        if (hx & (1u << 10)) {
            const u32 t01 = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.z), t23 = (u32)__builtin_amdgcn_ds_bpermute(self, (int)cur_hi.w);
            const u32 sh = (u32)__builtin_amdgcn_readfirstlane((int)second_w);
            run_sum(std::false_type(), t01, t23, 0u, (int)(second_w << 20) >> 28 /* bits 8..11, signed */, (sh >> 12) & 7u, (sh >> 15) & 3u,
                    second_w & 0x7fu);
        }
      }
      cur = nxt; cur_hi = nxt_hi;
      nxt = ahead; nxt_hi = ahead_hi;
    }
    if (TAG == TAG_PROFILE && tid == 0 && blockIdx.x == 0 && k_prev >= 0) {
        atomicAdd(A.prof + k_prev, (unsigned long long)(clock64() - t_prev));
        atomicAdd(A.prof + 16 + k_prev, 1ull);
    }
}

// ---- layout conversion: array-of-structures user buffers <-> slot arrays
// aos: n elements of `fp_per_elem` Fp each (12 u32 per Fp); soa slots [slot0, slot0 + fp_per_elem)
// `not_below_p` (may be null): set to 1 when some value read is not a canonical residue (limbs >= p): the programs
// assume values below p at their inputs, and arkworks never holds anything else, but a C caller can.
__global__ void pack_kernel(const u32* __restrict__ aos, uint4* __restrict__ soa, int n, int npad, int fp_per_elem,
                            int slot0, unsigned* not_below_p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (element, fp)
    long total = (long)n * fp_per_elem;
    if (i >= total) return;
    long e = i / fp_per_elem;
    int f = (int)(i % fp_per_elem);
    const uint4* src = reinterpret_cast<const uint4*>(aos + i * 12);
    uint4* dst = soa + ((size_t)(slot0 + f) * 3) * npad + e;
    const uint4 v0 = src[0], v1 = src[1], v2 = src[2];
    dst[0] = v0; dst[npad] = v1; dst[2 * (size_t)npad] = v2;
    if (not_below_p) {
        const u32 P[12] = BLS381_P_LIMBS;
        u32 x[12];
        unpack3(v0, v1, v2, x);
        // x - p without keeping the difference: no borrow out of the top limb means x >= p
        u32 borrow = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) (void)subb(x[k], P[k], borrow);
        if (!borrow) *not_below_p = 1u;
    }
}
__global__ void unpack_kernel(const uint4* __restrict__ soa, u32* __restrict__ aos, int n, int npad, int fp_per_elem,
                              int slot0) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n * fp_per_elem;
    if (i >= total) return;
    long e = i / fp_per_elem;
    int f = (int)(i % fp_per_elem);
    uint4* dst = reinterpret_cast<uint4*>(aos + i * 12);
    const uint4* src = soa + ((size_t)(slot0 + f) * 3) * npad + e;
    dst[0] = src[0]; dst[1] = src[npad]; dst[2] = src[2 * (size_t)npad];
}

// Overwrites the Fq12 (slots 0..11) of every element whose flag byte is non-zero with one.
__global__ void set_one_where_flag(uint4* soa, const uint8_t* flags, const uint4* consts, int n, int npad) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || flags[e] == 0) return;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int f = 0; f < 12; f++) {
        uint4* dst = soa + ((size_t)f * 3) * npad + e;
        if (f == 0) { dst[0] = consts[C_ONE * 3]; dst[npad] = consts[C_ONE * 3 + 1]; dst[2 * (size_t)npad] = consts[C_ONE * 3 + 2]; }
        else { dst[0] = z; dst[npad] = z; dst[2 * (size_t)npad] = z; }
    }
}

// Two-pair items with a pair at infinity: bits 0..1 of flags[e] skip the first pair, bits 2..3 the second.
// Items that skip exactly one pair: the other pair's inputs (P: 2 slots, Q: 4 slots of the 12-slot input array) are
// gathered into a compact 6-slot array in list order; *count ends up as their number.
__global__ void compact_one_pair_items(const uint4* soa, const uint8_t* flags, int n, int npad, uint4* out, int* list,
                                       int* count) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const bool skip_a = flags[e] & 3u, skip_b = flags[e] & 12u;
    if (skip_a == skip_b) return;
    const int k = atomicAdd(count, 1);
    list[k] = (int)e;
    const int base = skip_a ? 6 : 0;
    for (int f = 0; f < 6; f++)
        for (int c = 0; c < 3; c++) out[((size_t)f * 3 + c) * npad + k] = soa[((size_t)(base + f) * 3 + c) * npad + e];
}
// Puts the single-pair Miller values (`only`, in list order) in their items' places and one where both pairs are skipped.
__global__ void scatter_one_pair_items(uint4* soa, const uint4* only, const uint8_t* flags, const int* list, const int* count,
                                       const uint4* consts, int n, int npad) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if ((flags[e] & 3u) && (flags[e] & 12u))
        for (int f = 0; f < 12; f++)
            for (int c = 0; c < 3; c++) soa[((size_t)f * 3 + c) * npad + e] = f == 0 ? consts[C_ONE * 3 + c] : z;
    if (e < *count) {
        const int item = list[e];
        for (int f = 0; f < 12; f++)
            for (int c = 0; c < 3; c++) soa[((size_t)f * 3 + c) * npad + item] = only[((size_t)f * 3 + c) * npad + e];
    }
}
// flags[e] = 1 when the Fq12 in slots 0..11 equals one
__global__ void is_one_kernel(const uint4* soa, const uint4* consts, int n, int npad, uint8_t* flags) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    u32 diff = 0;
    for (int f = 0; f < 12; f++)
        for (int k = 0; k < 3; k++) {
            const uint4 v = soa[((size_t)f * 3 + k) * npad + e];
            const uint4 w = f == 0 ? consts[C_ONE * 3 + k] : make_uint4(0, 0, 0, 0);
            diff |= (v.x ^ w.x) | (v.y ^ w.y) | (v.z ^ w.z) | (v.w ^ w.w);
        }
    flags[e] = diff == 0 ? 1 : 0;
}

// dst (24 slots, stride npad_dst) element i = (src[i], i + h < m ? src[i + h] : one): one level of a product tree
__global__ void tree_pack_kernel(const uint4* src, uint4* dst, const uint4* consts, int m, int h, int npad_src,
                                 int npad_dst) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * 12) return;
    long e = i / 12;
    int f = (int)(i % 12);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int c = 0; c < 3; c++) {
        dst[((size_t)f * 3 + c) * npad_dst + e] = src[((size_t)f * 3 + c) * npad_src + e];
        uint4 v;
        if (e + h < m) v = src[((size_t)f * 3 + c) * npad_src + e + h];
        else v = (f == 0) ? consts[C_ONE * 3 + c] : z;
        dst[((size_t)(12 + f) * 3 + c) * npad_dst + e] = v;
    }
}

// In-place modular inversion of slot `slot` of a slot array (boundary form: x R -> x^-1 R, R = 2^384),
// one lane per element; 0 maps to 0 and is reported through is_zero.
//
// Bernstein-Yang division steps ("safegcd", delta variant) in the arrangement of libsecp256k1's
// modinv32, sized for the 381-bit modulus: f, g, d, e are 13 signed limbs of 30 bits; each round runs
// 30 division steps on the low 32 bits of f and g, collecting a 2x2 transition matrix with 31-bit
// entries, then applies it to (f, g) exactly and to (d, e) modulo p.  37 rounds = 1,110 steps cover
// the proven bound floor((49 * 381 + 57) / 17) = 1,101, after which g = 0, f = +-1 and d = +-1/x.
// About 33,000 instructions per element, against 400,000 for the Fermat chain x^(p-2).
namespace inv30 {
enum { NL = 13, ROUNDS = 37 };
#define BLS381_M30 0x3fffffff
#define BLS381_P30_LIMBS                                                                                          \
    { 0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673, 0x12e13ce1, 0x2cd76477, \
      0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111 }
#define BLS381_PINV30 0x00030003u  // p^-1 mod 2^30

// [a; b] <- ([[u, v], [q, r]] [a; b] + [ma; mb] p) / 2^30, exact; WITH_P = false drops the p terms
template <bool WITH_P>
__device__ __forceinline__ void apply(int* a, int* b, int u, int v, int q, int r, int ma, int mb) {
    const int P[NL] = BLS381_P30_LIMBS;
    int64_t ca = (int64_t)u * a[0] + (int64_t)v * b[0];
    int64_t cb = (int64_t)q * a[0] + (int64_t)r * b[0];
    if (WITH_P) {
        // choose the multiples of p that clear the low 30 bits
        ma -= (int)((BLS381_PINV30 * (u32)ca + (u32)ma) & BLS381_M30);
        mb -= (int)((BLS381_PINV30 * (u32)cb + (u32)mb) & BLS381_M30);
        ca += (int64_t)P[0] * ma;
        cb += (int64_t)P[0] * mb;
    }
    ca >>= 30;
    cb >>= 30;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        ca += (int64_t)u * a[i] + (int64_t)v * b[i];
        cb += (int64_t)q * a[i] + (int64_t)r * b[i];
        if (WITH_P) {
            ca += (int64_t)P[i] * ma;
            cb += (int64_t)P[i] * mb;
        }
        a[i - 1] = (int)((u32)ca & BLS381_M30);
        b[i - 1] = (int)((u32)cb & BLS381_M30);
        ca >>= 30;
        cb >>= 30;
    }
    a[NL - 1] = (int)ca;
    b[NL - 1] = (int)cb;
}
}  // namespace inv30

// x R -> x^-1 R on 12 x u32 (0 -> 0); the caller tests for zero
__device__ __noinline__ void fp_invert_boundary(const u32* a, u32* res) {
    using namespace inv30;
    int f[NL] = BLS381_P30_LIMBS, g[NL], d[NL], e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        u32 v = w < 12 ? a[w] >> s : 0u;
        if (s > 2 && w + 1 < 12) v |= a[w + 1] << (32 - s);
        g[i] = (int)(v & BLS381_M30);
        d[i] = 0;
        e[i] = i == 0 ? 1 : 0;
    }
    int delta = 1;
#pragma unroll 1
    for (int round = 0; round < ROUNDS; round++) {
        // 30 division steps on the low 32 bits: 2^30 [f'; g'] = [[u, v], [q, r]] [f; g]
        u32 f0 = (u32)f[0] | ((u32)f[1] << 30), g0 = (u32)g[0] | ((u32)g[1] << 30);
        int u = 1, v = 0, q = 0, r = 1;
#pragma unroll
        for (int i = 0; i < 30; i++) {
            // delta > 0 and g odd: (delta, f, g) <- (-delta, g, -f), rows swapped likewise
            const u32 c1 = (u32)((-delta) >> 31) & (0u - (g0 & 1u));
            const u32 tf = f0, tg = g0;
            f0 = (tg & c1) | (tf & ~c1);
            g0 = ((0u - tf) & c1) | (tg & ~c1);
            const int tu = u, tv = v;
            u = (q & (int)c1) | (tu & (int)~c1);
            v = (r & (int)c1) | (tv & (int)~c1);
            q = ((-tu) & (int)c1) | (q & (int)~c1);
            r = ((-tv) & (int)c1) | (r & (int)~c1);
            delta = (delta ^ (int)c1) - (int)c1;
            // g odd: g += f, second row += first
            const u32 c2 = 0u - (g0 & 1u);
            g0 += f0 & c2;
            q += u & (int)c2;
            r += v & (int)c2;
            delta += 1;
            g0 >>= 1;
            u <<= 1;
            v <<= 1;
        }
        // (d, e) stay in (-2p, p): a negative input counts p once more
        const int sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
        apply<true>(d, e, u, v, q, r, (u & sd) + (v & se), (q & sd) + (r & se));
        apply<false>(f, g, u, v, q, r, 0, 0);
    }
    // d * sign(f) into [0, p)
    const int P[NL] = BLS381_P30_LIMBS;
    const int sign = f[NL - 1] >> 31;
    int add = d[NL - 1] >> 31;  // (-2p, p) -> (-p, p)
#pragma unroll
    for (int i = 0; i < NL; i++) d[i] = ((d[i] + (P[i] & add)) ^ sign) - sign;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { d[i + 1] += d[i] >> 30; d[i] &= BLS381_M30; }
    add = d[NL - 1] >> 31;  // (-p, p) -> [0, p)
#pragma unroll
    for (int i = 0; i < NL; i++) d[i] += P[i] & add;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { d[i + 1] += d[i] >> 30; d[i] &= BLS381_M30; }
    // 13 x 30 bits -> 12 x 32, then (x R)^-1 * R^3 / R = x^-1 R
    u32 x[12];
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit - 30 * i;
        u32 v = (u32)d[i] >> s;
        v |= (u32)d[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < NL) v |= (u32)d[i + 2] << (60 - s);
        x[w] = v;
    }
    const u32 R3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                        0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};  // R^3 mod p
    fp_mul(x, R3, res);
}

__global__ __launch_bounds__(64) void fp_inverse_kernel(uint4* soa, int slot, int n, int npad, const uint4* consts,
                                                        uint8_t* is_zero) {
    (void)consts;
    long el = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= n) return;
    uint4* p = soa + ((size_t)slot * 3) * npad + el;
    u32 a[12], res[12];
    {
        uint4 v0 = p[0], v1 = p[npad], v2 = p[2 * (size_t)npad];
        unpack3(v0, v1, v2, a);
    }
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) nz |= a[i];
    if (is_zero) is_zero[el] = (nz == 0);
    fp_invert_boundary(a, res);
    p[0] = make_uint4(res[0], res[1], res[2], res[3]);
    p[npad] = make_uint4(res[4], res[5], res[6], res[7]);
    p[2 * (size_t)npad] = make_uint4(res[8], res[9], res[10], res[11]);
}

}  // namespace bls381
