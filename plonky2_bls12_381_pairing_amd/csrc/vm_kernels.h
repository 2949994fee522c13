// gfx950 kernels: the lane-group interpreter, layout conversion, Fp inversion.
#pragma once
#include <hip/hip_runtime.h>

#include "fp_dev.h"
#include "fp_asm_gen.h"
#include "vm_format.h"

namespace bls381 {

// Global slot arrays are structure-of-arrays: slot s, 16-byte chunk c (3 per Fp), element e:
//   uint4 index = (s * 3 + c) * npad + e          -> a wave reads 64 consecutive uint4
// LDS slots use the same shape with the workgroup's group count in place of npad.
struct VmArgs {
    const StepHdr* hdrs;
    const OpDesc* descs;
    int nsteps;
    int lds_slots;
    const uint4* consts;  // C_COUNT * 3 uint4
    const uint4* kp;      // 17 rows of 4 uint4: k * p as 13 limbs (zero padded), k = 0..16
    const uint4* gin;
    uint4* gout;
    uint4* gtmp;
    int n;     // number of groups (pairings) with work
    int npad;  // element stride of the global slot arrays
};

enum { VM_THREADS = 256, KP_ROWS = 17 };

__device__ __forceinline__ void unpack3(const uint4& a, const uint4& b, const uint4& c, u32* x) {
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
}

// TAG only gives the Miller-loop, final-exponentiation and auxiliary launches distinct kernel
// names in profiler output; the code is identical.
enum { TAG_MILLER = 0, TAG_FINAL_EXP = 1, TAG_AUX = 2 };

template <int K, int TAG>
__global__ __launch_bounds__(VM_THREADS) void vm_kernel(VmArgs A) {
    constexpr int G = 64 / K;      // groups per wave
    constexpr int GPW = 4 * G;     // groups per workgroup
    extern __shared__ uint4 lds[];
    uint4* kp_lds = lds + (size_t)A.lds_slots * 3 * GPW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int role_raw = lane / G;
    const int gw = lane % G;
    const int g = wave * G + gw;
    const int role = role_raw < K ? role_raw : 0;
    const long e = (long)blockIdx.x * GPW + g;
    const bool lane_on = (role_raw < K) && (e < A.n);

    // stage the k*p table
    for (int i = tid; i < KP_ROWS * 4; i += VM_THREADS) kp_lds[i] = A.kp[i];
    // the ZERO slot (second to last) of every group
    for (int i = tid; i < 3 * GPW; i += VM_THREADS) lds[(size_t)(A.lds_slots - 2) * 3 * GPW + i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    auto lds_load = [&](int slot, u32* x) {
        const uint4* p = lds + (size_t)slot * 3 * GPW + g;
        uint4 a = p[0], b = p[GPW], c = p[2 * GPW];
        unpack3(a, b, c, x);
    };
    auto lds_store = [&](int slot, const u32* x) {
        uint4* p = lds + (size_t)slot * 3 * GPW + g;
        p[0] = make_uint4(x[0], x[1], x[2], x[3]);
        p[GPW] = make_uint4(x[4], x[5], x[6], x[7]);
        p[2 * GPW] = make_uint4(x[8], x[9], x[10], x[11]);
    };

    const uint4* dptr = reinterpret_cast<const uint4*>(A.descs) + (size_t)role * 2;
    uint4 nd0 = dptr[0], nd1 = dptr[1];

    StepHdr nh = A.hdrs[0];
    for (int s = 0; s < A.nsteps; s++) {
        const StepHdr h = nh;
        const uint4 d0 = nd0, d1 = nd1;
        if (s + 1 < A.nsteps) {  // next step's header and descriptor travel while this step computes
            nh = A.hdrs[s + 1];
            const uint4* np = dptr + (size_t)(s + 1) * K * 2;
            nd0 = np[0]; nd1 = np[1];
        }
        const bool on = lane_on && (d0.x & 1u);
        // descriptor words: w[i] lives in dword 1 + i/2 of the 8-dword descriptor
        const u32 dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        auto W = [&](int i) -> u32 { return (dw[1 + (i >> 1)] >> ((i & 1) * 16)) & 0xffffu; };

        // Multiplier and sum steps are branch-free: idle lanes read the ZERO slot and store to TRASH.
        if (h.type == ST_FMUL) {
            u32 a[12], b[12], r[12];
            lds_load(W(1), a);
            lds_load(W(2), b);
            fp_mul(a, b, r);
            lds_store(W(0), r);
        } else if (h.type == ST_DOT2) {
            u32 a[12], b[12], c[12], d[12], r[12];
            lds_load(W(1), a);
            lds_load(W(2), b);
            lds_load(W(3), c);
            lds_load(W(4), d);
            if (d0.x & 4u) {  // a*b - c*d: d <- k*p - d (k*p >= d by the scheduler's bound)
                const uint4* row = kp_lds + ((d0.x >> 16) & 0xffu) * 4;
                uint4 r0 = row[0], r1 = row[1], r2 = row[2];
                u32 q[12];
                unpack3(r0, r1, r2, q);
                carry_t cy;
                d[0] = sub_co(q[0], d[0], cy);
#pragma unroll
                for (int i = 1; i < 12; i++) d[i] = subb_co(q[i], d[i], cy);
            }
            fp_dot2(a, b, c, d, r);
            lds_store(W(0), r);
        } else if (h.type == ST_LIN) {
            // acc = bias + sum(+/- terms) in 13-limb two's complement; bias = k*p makes it >= 0
            u32 acc[13];
            const u32 kb = (d0.x >> 16) & 0xffu;
            {
                const uint4* row = kp_lds + kb * 4;
                uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
                unpack3(r0, r1, r2, acc);
                acc[12] = r3.x;
            }
#pragma unroll
            for (int t = 0; t < LIN_MAX_TERMS; t++) {
                if (t < (int)(h.max_terms & 0xffu)) {  // uniform; short sums are padded with +ZERO
                    const u32 w = W(1 + t);
                    const u32 m = (w & 0x8000u) ? 0xffffffffu : 0u;
                    u32 x[12], y[12];
                    lds_load(w & 0x7fffu, x);
#pragma unroll
                    for (int i = 0; i < 12; i++) y[i] = x[i] ^ m;
                    lin_accumulate(acc, y, m);  // acc += x, or acc += ~x + 1
                }
            }
            // conditional subtraction of 2^j * p, j = rounds-1 .. 0
            const u32 rounds = (h.max_terms >> 8) & 0xffu;
            if (rounds > 4) cond_sub_kp<4>(acc);  // 16p
            if (rounds > 3) cond_sub_kp<3>(acc);
            if (rounds > 2) cond_sub_kp<2>(acc);
            if (rounds > 1) cond_sub_kp<1>(acc);
            if (rounds > 0) cond_sub_kp<0>(acc);
            lds_store(W(0), acc);
        } else {  // ST_COPY: every lane reads before any lane writes
            const u32 kind = (d0.x >> 8) & 0xffu;
            const u32 dst = W(0), src = W(1);
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0;
            if (on) {
                if (kind == CK_IN || kind == CK_FILL) {
                    const uint4* p = (kind == CK_IN ? A.gin : A.gtmp) + ((size_t)src * 3) * A.npad + e;
                    v0 = p[0]; v1 = p[A.npad]; v2 = p[2 * (size_t)A.npad];
                } else if (kind == CK_CONST) {
                    const uint4* p = A.consts + (size_t)src * 3;
                    v0 = p[0]; v1 = p[1]; v2 = p[2];
                } else {
                    const uint4* p = lds + (size_t)src * 3 * GPW + g;
                    v0 = p[0]; v1 = p[GPW]; v2 = p[2 * GPW];
                }
            }
            if (on) {
                if (kind <= CK_CONST) {
                    uint4* p = lds + (size_t)dst * 3 * GPW + g;
                    p[0] = v0; p[GPW] = v1; p[2 * GPW] = v2;
                } else {
                    uint4* p = (kind == CK_SPILL ? A.gtmp : A.gout) + ((size_t)dst * 3) * A.npad + e;
                    p[0] = v0; p[A.npad] = v1; p[2 * (size_t)A.npad] = v2;
                }
            }
            // spilled values may be filled by another lane of this wave in a later step
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
    }
}

// ---- layout conversion: array-of-structures user buffers <-> slot arrays
// aos: n elements of `fp_per_elem` Fp each (12 u32 per Fp); soa slots [slot0, slot0 + fp_per_elem)
__global__ void pack_kernel(const u32* __restrict__ aos, uint4* __restrict__ soa, int n, int npad, int fp_per_elem,
                            int slot0) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (element, fp)
    long total = (long)n * fp_per_elem;
    if (i >= total) return;
    long e = i / fp_per_elem;
    int f = (int)(i % fp_per_elem);
    const uint4* src = reinterpret_cast<const uint4*>(aos + i * 12);
    uint4* dst = soa + ((size_t)(slot0 + f) * 3) * npad + e;
    dst[0] = src[0]; dst[npad] = src[1]; dst[2 * (size_t)npad] = src[2];
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

// In-place x <- x^(p-2) on slot `slot` of a slot array, one lane per element (Fermat inversion by
// left-to-right square-and-multiply; 0 maps to 0 and is reported through is_zero).
__global__ __launch_bounds__(64) void fp_inverse_kernel(uint4* soa, int slot, int n, int npad, const uint4* consts,
                                                        uint8_t* is_zero) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint4* p = soa + ((size_t)slot * 3) * npad + e;
    u32 a[12], r[12];
    {
        uint4 v0 = p[0], v1 = p[npad], v2 = p[2 * (size_t)npad];
        unpack3(v0, v1, v2, a);
    }
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) nz |= a[i];
    if (is_zero) is_zero[e] = (nz == 0);
    {
        uint4 v0 = consts[C_ONE * 3], v1 = consts[C_ONE * 3 + 1], v2 = consts[C_ONE * 3 + 2];
        unpack3(v0, v1, v2, r);
    }
    const u32 E[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                       0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};  // p - 2
    for (int w = 11; w >= 0; w--) {
        const u32 ew = E[w];
        for (int b = 31; b >= 0; b--) {
            u32 t[12];
            fp_mul(r, r, t);
            if ((ew >> b) & 1u) fp_mul(t, a, r);
            else {
#pragma unroll
                for (int i = 0; i < 12; i++) r[i] = t[i];
            }
        }
    }
    p[0] = make_uint4(r[0], r[1], r[2], r[3]);
    p[npad] = make_uint4(r[4], r[5], r[6], r[7]);
    p[2 * (size_t)npad] = make_uint4(r[8], r[9], r[10], r[11]);
}

}  // namespace bls381
