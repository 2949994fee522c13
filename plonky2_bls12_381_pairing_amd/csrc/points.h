// Compressed point wire format and point aggregation: what a BLS signature verifier does with its
// bytes before it reaches the pairing.  The reference takes ark-ec points; its callers get them from
// ark-bls12-381 0.4.0 (/root/reference/Cargo.toml:13) G1Affine / G2Affine deserialize_compressed and
// write them with serialize_compressed, which implement the ZCash BLS12-381 encoding:
//   G1 48 bytes big-endian x; G2 96 bytes big-endian x.c1 then x.c0; top bits of byte 0:
//   0x80 compressed, 0x40 infinity, 0x20 y is the lexicographically larger root.
// One lane per point, 12 x 32-bit Montgomery limbs (the arithmetic of generators.h).  Points leave in
// the pairing's own input layout (affine, Montgomery limbs), so decoded keys feed it without a copy.
#pragma once
#include <hip/hip_runtime.h>

#include "generators.h"

namespace bls381 {
namespace pts {

using gen::F;
using gen::F2;

enum Status : uint8_t { ST_OK = 0, ST_INFINITY = 1, ST_BAD_ENCODING = 2, ST_NOT_ON_CURVE = 3, ST_NOT_IN_SUBGROUP = 4 };

// ---- field helpers, out of line: these kernels are long chains of products and code size matters
__device__ __noinline__ F M(const F& a, const F& b) { return gen::mul(a, b); }
// (a0 + a1 u)(b0 + b1 u): each component is a sum of two products with one reduction (fp_dot2), the minus sign of
// a1 b1 goes into the operand
__device__ __noinline__ F2 M(const F2& a, const F2& b) {
    const F nb1 = gen::sub(gen::zero(), b.c1);
    F2 r;
    fp_dot2(a.c0.v, b.c0.v, a.c1.v, nb1.v, r.c0.v);
    fp_dot2(a.c0.v, b.c1.v, a.c1.v, b.c0.v, r.c1.v);
    return r;
}
__device__ __noinline__ F A(const F& a, const F& b) { return gen::add(a, b); }
__device__ __noinline__ F S(const F& a, const F& b) { return gen::sub(a, b); }
__device__ __forceinline__ F2 A(const F2& a, const F2& b) { return {A(a.c0, b.c0), A(a.c1, b.c1)}; }
__device__ __forceinline__ F2 S(const F2& a, const F2& b) { return {S(a.c0, b.c0), S(a.c1, b.c1)}; }
__device__ __forceinline__ F ngt(const F& a) { return S(gen::zero(), a); }
__device__ __forceinline__ F2 ngt(const F2& a) { return {ngt(a.c0), ngt(a.c1)}; }
__device__ __forceinline__ bool is0(const F& a) { return gen::is_zero(a); }
__device__ __forceinline__ bool is0(const F2& a) { return gen::is_zero(a.c0) && gen::is_zero(a.c1); }
__device__ __forceinline__ bool same(const F& a, const F& b) { return gen::eq(a, b); }
__device__ __forceinline__ bool same(const F2& a, const F2& b) { return gen::eq(a.c0, b.c0) && gen::eq(a.c1, b.c1); }
__device__ __forceinline__ F one_mont() {
    return {{0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
             0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u}};
}
template <class T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ F zero_of<F>() { return gen::zero(); }
template <> __device__ __forceinline__ F2 zero_of<F2>() { return {gen::zero(), gen::zero()}; }
template <class T> __device__ __forceinline__ T one_of();
template <> __device__ __forceinline__ F one_of<F>() { return one_mont(); }
template <> __device__ __forceinline__ F2 one_of<F2>() { return {one_mont(), gen::zero()}; }
// 3b times a: b = 4 on G1, 4(1 + u) on the twist
__device__ __forceinline__ F times12(const F& a) {
    const F a2 = A(a, a), a4 = A(a2, a2), a8 = A(a4, a4);
    return A(a8, a4);
}
__device__ __forceinline__ F mul_b3(const F& a) { return times12(a); }
__device__ __forceinline__ F2 mul_b3(const F2& a) { return {times12(S(a.c0, a.c1)), times12(A(a.c0, a.c1))}; }
// the curve constant b itself, in Montgomery form
__device__ __forceinline__ F four_mont() { const F o = one_mont(), t = A(o, o); return A(t, t); }
__device__ __forceinline__ void curve_b(F& b) { b = four_mont(); }
__device__ __forceinline__ void curve_b(F2& b) { b = {four_mont(), four_mont()}; }

// ---- homogeneous projective points; infinity = (0 : 1 : 0)
template <class T> struct Proj { T X, Y, Z; };
template <class T> __device__ __forceinline__ Proj<T> identity() { return {zero_of<T>(), one_of<T>(), zero_of<T>()}; }
template <class T> __device__ __forceinline__ Proj<T> from_affine(const T& x, const T& y, bool inf) {
    if (inf) return identity<T>();
    return {x, y, one_of<T>()};
}
// Complete addition on y^2 = x^3 + b (Renes, Costello, Batina 2016, algorithm 7): no special cases for
// doubling, opposite points or infinity.  12 products.
template <class T> __device__ __noinline__ Proj<T> padd(const Proj<T>& p, const Proj<T>& q) {
    T t0 = M(p.X, q.X), t1 = M(p.Y, q.Y), t2 = M(p.Z, q.Z);
    T t3 = M(A(p.X, p.Y), A(q.X, q.Y));
    t3 = S(t3, A(t0, t1));
    T t4 = M(A(p.Y, p.Z), A(q.Y, q.Z));
    t4 = S(t4, A(t1, t2));
    T y3 = M(A(p.X, p.Z), A(q.X, q.Z));
    y3 = S(y3, A(t0, t2));
    T x3 = A(t0, t0);
    t0 = A(x3, t0);
    t2 = mul_b3(t2);
    T z3 = A(t1, t2);
    t1 = S(t1, t2);
    y3 = mul_b3(y3);
    x3 = M(t4, y3);
    t2 = M(t3, t1);
    x3 = S(t2, x3);
    y3 = M(y3, t0);
    t1 = M(t1, z3);
    y3 = A(t1, y3);
    t0 = M(t0, t3);
    z3 = M(z3, t4);
    z3 = A(z3, t0);
    return {x3, y3, z3};
}
// Doubling on the same curve (algorithm 9 of the same paper): 8 products; infinity doubles to infinity.
template <class T> __device__ __noinline__ Proj<T> pdbl(const Proj<T>& p) {
    T t0 = M(p.Y, p.Y);
    T z3 = A(t0, t0);
    z3 = A(z3, z3);
    z3 = A(z3, z3);
    T t1 = M(p.Y, p.Z);
    T t2 = mul_b3(M(p.Z, p.Z));
    T x3 = M(t2, z3);
    T y3 = A(t0, t2);
    z3 = M(t1, z3);
    t1 = A(t2, t2);
    t2 = A(t1, t2);
    t0 = S(t0, t2);
    y3 = A(x3, M(t0, y3));
    x3 = M(t0, M(p.X, p.Y));
    x3 = A(x3, x3);
    return {x3, y3, z3};
}
template <class T> __device__ __forceinline__ Proj<T> pneg(const Proj<T>& p) { return {p.X, ngt(p.Y), p.Z}; }
// equality of the points two valid triples stand for
template <class T> __device__ __forceinline__ bool peq(const Proj<T>& a, const Proj<T>& b) {
    return same(M(a.X, b.Z), M(b.X, a.Z)) && same(M(a.Y, b.Z), M(b.Y, a.Z));
}
// [|x|] p for the curve parameter |x| = 0xd201000000010000 (/root/reference/src/utils/constants.rs:1-2)
template <class T> __device__ __noinline__ Proj<T> pmul_x(const Proj<T>& p) {
    Proj<T> r = p;
    for (int i = 62; i >= 0; i--) {
        r = pdbl(r);
        if ((0xd201000000010000ull >> i) & 1ull) r = padd(r, p);
    }
    return r;
}
// Membership in the order-r subgroup by endomorphism (Scott, "A note on group membership tests for G1,
// G2 and GT on BLS pairing-friendly curves"; the tests ark-bls12-381 0.4.0 runs on decoded points), with
// z = -|x|:  G1: (beta x, y) = [-z^2] P;   G2: psi(P) = [z] P with psi the untwist-Frobenius-twist map.
__device__ __forceinline__ bool in_subgroup(const F& x, const F& y) {
    const u32 BETA[12] = {0xfffefffeu, 0x2e01ffffu, 0x620a0002u, 0xde17d813u, 0xe6f89688u, 0xddb3a93bu,
                          0x6a0f77eau, 0xba69c607u, 0xdf76ce51u, 0x5f19672fu, 0x00000000u, 0x00000000u};
    const Proj<F> p = from_affine(x, y, false);
    const Proj<F> lhs = {M(gen::to_mont(BETA), x), y, one_mont()};
    return peq(lhs, pneg(pmul_x(pmul_x(p))));
}
__device__ __forceinline__ bool in_subgroup(const F2& x, const F2& y) {
    const u32 PSI_X_C1[12] = {0x0000aaadu, 0x8bfd0000u, 0x4f49fffdu, 0x409427ebu, 0x0fb85f9bu, 0x897d2965u,
                              0x89759ad4u, 0xaa0d857du, 0x63d4de85u, 0xec024086u, 0x397fe699u, 0x1a0111eau};
    const u32 PSI_Y_C0[12] = {0x121bdea2u, 0xf1ee7b04u, 0x3e67fa0au, 0x304466cfu, 0xf61eb45eu, 0xef396489u,
                              0x30b1cf60u, 0x1c3dedd9u, 0xd77a2cd9u, 0xe2e9c448u, 0x0180a68eu, 0x135203e6u};
    const u32 PSI_Y_C1[12] = {0xede3cc09u, 0xc81084fbu, 0x72ec05f4u, 0xee67992fu, 0x009241c5u, 0x77f76e17u,
                              0xc2d3435eu, 0x48395dabu, 0x6bd17ffeu, 0x6831e36du, 0x37ff400bu, 0x06af0e04u};
    const F2 cx{gen::zero(), gen::to_mont(PSI_X_C1)}, cy{gen::to_mont(PSI_Y_C0), gen::to_mont(PSI_Y_C1)};
    const F2 xc{x.c0, ngt(x.c1)}, yc{y.c0, ngt(y.c1)};
    const Proj<F2> lhs = {M(cx, xc), M(cy, yc), one_of<F2>()};
    return peq(lhs, pneg(pmul_x(from_affine(x, y, false))));
}

// ---- bytes <-> limbs
// 48 big-endian bytes -> 12 little-endian u32 limbs, flag bits of byte 0 removed when `strip`
__device__ __forceinline__ void be48_to_limbs(const uint8_t* b, u32* l, bool strip) {
    for (int k = 0; k < 12; k++) {
        const uint8_t* q = b + 44 - 4 * k;
        l[k] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    if (strip) l[11] &= 0x1fffffffu;
}
__device__ __forceinline__ void limbs_to_be48(const u32* l, uint8_t* b) {
    for (int k = 0; k < 12; k++) {
        uint8_t* q = b + 44 - 4 * k;
        q[0] = (uint8_t)(l[k] >> 24); q[1] = (uint8_t)(l[k] >> 16); q[2] = (uint8_t)(l[k] >> 8); q[3] = (uint8_t)l[k];
    }
}
__device__ __forceinline__ bool limbs_zero(const u32* l) { u32 nz = 0; for (int k = 0; k < 12; k++) nz |= l[k]; return nz == 0; }
// canonical y > (p - 1) / 2
__device__ __forceinline__ bool is_larger(const F& canonical) {
    const u32 H[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                       0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    for (int k = 11; k >= 0; k--) { if (canonical.v[k] > H[k]) return true; if (canonical.v[k] < H[k]) return false; }
    return false;
}
__device__ __forceinline__ bool is_larger_mont(const F& y) { return is_larger(gen::from_mont(y)); }
__device__ __forceinline__ bool is_larger_mont(const F2& y) {
    return is0(y.c1) ? is_larger(gen::from_mont(y.c0)) : is_larger(gen::from_mont(y.c1));
}
__device__ __forceinline__ F sqrt_of(const F& a, bool& ok) { return gen::sqrt_fq(a, ok); }
__device__ __forceinline__ F2 sqrt_of(const F2& a, bool& ok) { return gen::sqrt_fq2(a, ok); }

// y for a decoded x: the root of x^3 + b that the sort flag names
template <class T> __device__ __forceinline__ Status solve_y(const T& x, bool sort, bool check_subgroup, T& y) {
    T b; curve_b(b);
    bool ok;
    y = sqrt_of(A(M(M(x, x), x), b), ok);
    if (!ok) return ST_NOT_ON_CURVE;
    if (is_larger_mont(y) != sort) y = ngt(y);
    if (check_subgroup && !in_subgroup(x, y)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}

__device__ __forceinline__ void store_f(u32* dst, const F& a) { for (int k = 0; k < 12; k++) dst[k] = a.v[k]; }
__device__ __forceinline__ F load_f(const u32* src) { F a; for (int k = 0; k < 12; k++) a.v[k] = src[k]; return a; }

// in: n x 48 bytes; out: n x {x, y} Montgomery limbs (zero unless status is ST_OK); status: n bytes
__global__ __launch_bounds__(64) void g1_decompress_kernel(const uint8_t* in, int n, int check_subgroup, u32* out,
                                                           uint8_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + (size_t)i * 48;
    const bool comp = b[0] & 0x80u, inf = b[0] & 0x40u, sort = b[0] & 0x20u;
    u32 xl[12];
    be48_to_limbs(b, xl, true);
    Status st;
    F x = gen::zero(), y = gen::zero();
    if (!comp) st = ST_BAD_ENCODING;
    else if (inf) st = (!sort && limbs_zero(xl)) ? ST_INFINITY : ST_BAD_ENCODING;
    else if (!gen::below_p(xl)) st = ST_BAD_ENCODING;
    else {
        x = gen::to_mont(xl);
        st = solve_y(x, sort, check_subgroup != 0, y);
    }
    const bool good = st == ST_OK;
    store_f(out + (size_t)i * 24, good ? x : gen::zero());
    store_f(out + (size_t)i * 24 + 12, good ? y : gen::zero());
    status[i] = st;
}
// in: n x 96 bytes (x.c1 then x.c0); out: n x {x.c0, x.c1, y.c0, y.c1}
__global__ __launch_bounds__(64) void g2_decompress_kernel(const uint8_t* in, int n, int check_subgroup, u32* out,
                                                           uint8_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + (size_t)i * 96;
    const bool comp = b[0] & 0x80u, inf = b[0] & 0x40u, sort = b[0] & 0x20u;
    u32 x1[12], x0[12];
    be48_to_limbs(b, x1, true);
    be48_to_limbs(b + 48, x0, false);
    Status st;
    F2 x = zero_of<F2>(), y = zero_of<F2>();
    if (!comp) st = ST_BAD_ENCODING;
    else if (inf) st = (!sort && limbs_zero(x0) && limbs_zero(x1)) ? ST_INFINITY : ST_BAD_ENCODING;
    else if (!gen::below_p(x0) || !gen::below_p(x1)) st = ST_BAD_ENCODING;
    else {
        x = {gen::to_mont(x0), gen::to_mont(x1)};
        st = solve_y(x, sort, check_subgroup != 0, y);
    }
    const bool good = st == ST_OK;
    const F z = gen::zero();
    u32* o = out + (size_t)i * 48;
    store_f(o, good ? x.c0 : z); store_f(o + 12, good ? x.c1 : z);
    store_f(o + 24, good ? y.c0 : z); store_f(o + 36, good ? y.c1 : z);
    status[i] = st;
}
// What a subgroup-test program (build_g1_subgroup / build_g2_subgroup) returned for n decoded points: diff_words words
// each, all zero for a point of the subgroup.  A point the decoder accepted that fails becomes ST_NOT_IN_SUBGROUP and,
// like every rejected point, zero (pt_words words).
__global__ __launch_bounds__(64) void subgroup_status_kernel(const u32* diff, int diff_words, int n, uint8_t* status, u32* pts,
                                                             int pt_words) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || status[i] != ST_OK) return;
    u32 nz = 0;
    for (int k = 0; k < diff_words; k++) nz |= diff[(size_t)i * diff_words + k];
    if (nz == 0) return;
    status[i] = ST_NOT_IN_SUBGROUP;
    for (int k = 0; k < pt_words; k++) pts[(size_t)i * pt_words + k] = 0;
}
// points: n x {x, y} Montgomery limbs, inf: n bytes (may be null) -> n x 48 bytes
__global__ __launch_bounds__(64) void g1_compress_kernel(const u32* pts, const uint8_t* inf, int n, uint8_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* b = out + (size_t)i * 48;
    if (inf && inf[i]) {
        for (int k = 0; k < 48; k++) b[k] = k == 0 ? 0xc0 : 0;
        return;
    }
    const F x = gen::from_mont(load_f(pts + (size_t)i * 24)), y = load_f(pts + (size_t)i * 24 + 12);
    limbs_to_be48(x.v, b);
    b[0] |= 0x80u | (is_larger_mont(y) ? 0x20u : 0u);
}
__global__ __launch_bounds__(64) void g2_compress_kernel(const u32* pts, const uint8_t* inf, int n, uint8_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* b = out + (size_t)i * 96;
    if (inf && inf[i]) {
        for (int k = 0; k < 96; k++) b[k] = k == 0 ? 0xc0 : 0;
        return;
    }
    const u32* p = pts + (size_t)i * 48;
    const F x0 = gen::from_mont(load_f(p)), x1 = gen::from_mont(load_f(p + 12));
    const F2 y{load_f(p + 24), load_f(p + 36)};
    limbs_to_be48(x1.v, b);
    limbs_to_be48(x0.v, b + 48);
    b[0] |= 0x80u | (is_larger_mont(y) ? 0x20u : 0u);
}

// ---- aggregation: the sum of each group of `count` points, by halving passes over projective triples.
// Triples are stored as 3 * FQ consecutive Fq (FQ = 1 for G1, 2 for G2).
template <class T> struct Width;
template <> struct Width<F> { static constexpr int FQ = 1; };
template <> struct Width<F2> { static constexpr int FQ = 2; };
__device__ __forceinline__ void load_t(const u32* s, F& a) { a = load_f(s); }
__device__ __forceinline__ void load_t(const u32* s, F2& a) { a = {load_f(s), load_f(s + 12)}; }
__device__ __forceinline__ void store_t(u32* d, const F& a) { store_f(d, a); }
__device__ __forceinline__ void store_t(u32* d, const F2& a) { store_f(d, a.c0); store_f(d + 12, a.c1); }
template <class T> __device__ __forceinline__ Proj<T> load_proj(const u32* s) {
    constexpr int W = Width<T>::FQ * 12;
    Proj<T> p;
    load_t(s, p.X); load_t(s + W, p.Y); load_t(s + 2 * W, p.Z);
    return p;
}
template <class T> __device__ __forceinline__ void store_proj(u32* d, const Proj<T>& p) {
    constexpr int W = Width<T>::FQ * 12;
    store_t(d, p.X); store_t(d + W, p.Y); store_t(d + 2 * W, p.Z);
}
// first pass: out[g][j] = P[g][2j] + P[g][2j+1] from affine inputs; half = ceil(count / 2)
template <class T>
__global__ __launch_bounds__(64) void agg_first_kernel(const u32* pts, const uint8_t* inf, int n_groups, int count, int half,
                                                       u32* out) {
    constexpr int W = Width<T>::FQ * 12;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_groups * half) return;
    const long g = t / half, j = t % half;
    auto get = [&](long k) {
        if (k >= count) return identity<T>();
        const long idx = g * count + k;
        T x, y;
        load_t(pts + idx * 2 * W, x); load_t(pts + idx * 2 * W + W, y);
        return from_affine(x, y, inf && inf[idx]);
    };
    store_proj(out + t * 3 * W, padd(get(2 * j), get(2 * j + 1)));
}
// later passes: in holds `count` triples per group, out gets half = ceil(count / 2)
template <class T>
__global__ __launch_bounds__(64) void agg_pass_kernel(const u32* in, int n_groups, int count, int half, u32* out) {
    constexpr int W = Width<T>::FQ * 12;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_groups * half) return;
    const long g = t / half, j = t % half;
    const Proj<T> a = load_proj<T>(in + (g * count + 2 * j) * 3 * W);
    const Proj<T> b = 2 * j + 1 < count ? load_proj<T>(in + (g * count + 2 * j + 1) * 3 * W) : identity<T>();
    store_proj(out + t * 3 * W, padd(a, b));
}
__device__ __forceinline__ F inv_t(const F& a) { return gen::inv(a); }
__device__ __forceinline__ F2 inv_t(const F2& a) { return gen::inv(a); }
// one triple per group -> affine point + infinity flag
template <class T>
__global__ __launch_bounds__(64) void agg_finish_kernel(const u32* in, int n_groups, u32* out, uint8_t* out_inf) {
    constexpr int W = Width<T>::FQ * 12;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const Proj<T> p = load_proj<T>(in + (size_t)g * 3 * W);
    const bool inf = is0(p.Z);
    const T zi = inv_t(p.Z);  // the inversion maps 0 to 0
    store_t(out + (size_t)g * 2 * W, M(p.X, zi));
    store_t(out + (size_t)g * 2 * W + W, inf ? zero_of<T>() : M(p.Y, zi));
    out_inf[g] = inf ? 1 : 0;
}

// ---- signature checks from bytes: glue between the decoders and the two-pair pairing
// -G1 (the negated standard generator) for every item, Montgomery limbs
__global__ void fill_neg_g1_generator(u32* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 X[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                       0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    const u32 Y[12] = {0xf318883au, 0xff526c2au, 0x383b0270u, 0x92899ce4u, 0x9fa9d055u, 0x89d7738du,
                       0xa344c12au, 0x12caf35bu, 0x964b5317u, 0x3cff1b76u, 0xe9774430u, 0x0e44d2edu};
    for (int k = 0; k < 12; k++) { out[(size_t)i * 24 + k] = X[k]; out[(size_t)i * 24 + 12 + k] = Y[k]; }
}
// Decoder statuses of (public key, message point, signature) -> pairing flags and a validity byte.  A key at
// infinity is rejected (KeyValidate), and so is a sum of keys that comes out as infinity (FastAggregateVerify runs
// KeyValidate on the aggregate: {pk, -pk} must not verify anything) and a message point at infinity (no message
// hashes to it; with a signature at infinity it would make the product empty, i.e. one, under any key).  A signature
// at infinity only leaves its pair out of the product, which then cannot be one.
__global__ void verify_flags_kernel(const uint8_t* st_pk, const uint8_t* st_msg, const uint8_t* st_sig, int n,
                                    uint8_t* pair_flags, uint8_t* valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool ok = st_pk[i] == ST_OK && st_msg[i] == ST_OK && st_sig[i] <= ST_INFINITY;
    valid[i] = ok ? 1 : 0;
    pair_flags[i] = !ok ? 15 : (uint8_t)(st_sig[i] == ST_INFINITY ? 8 : 0);
}
// Status of each group's key sum: the worst status among its keys when any key was not a finite valid point,
// else infinity or fine as the sum turned out.
__global__ void group_status_kernel(const uint8_t* st_keys, const uint8_t* sum_inf, int n_groups, int group_size,
                                    uint8_t* st_group) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    uint8_t worst = ST_OK;
    for (int k = 0; k < group_size; k++) {
        const uint8_t s = st_keys[(size_t)g * group_size + k];
        if (s != ST_OK) worst = s == ST_INFINITY ? (worst > ST_BAD_ENCODING ? worst : (uint8_t)ST_BAD_ENCODING) : (s > worst ? s : worst);
    }
    st_group[g] = worst != ST_OK ? worst : (sum_inf[g] ? (uint8_t)ST_INFINITY : (uint8_t)ST_OK);
}
__global__ void verify_verdict_kernel(const uint8_t* is_one, const uint8_t* valid, int n, uint8_t* verdict) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) verdict[i] = (is_one[i] && valid[i]) ? 1 : 0;
}

}  // namespace pts
}  // namespace bls381
