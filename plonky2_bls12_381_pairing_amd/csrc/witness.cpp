#include "witness.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>

#include "bls381_constants.h"
#include "fp_dev.h"
#include "fp_asm_gen.h"
#include "fp28_dev.h"

namespace bls381 {

// ================================================================= host: symbolic replay
namespace {

struct WB {  // builder: every method appends one row
    std::vector<uint8_t> kinds;
    std::vector<uint32_t> ra, rb;
    uint32_t emit(WitnessKind k, uint32_t a, uint32_t b) {
        kinds.push_back(k); ra.push_back(a); rb.push_back(b);
        return (uint32_t)kinds.size() - 1;  // WREF_ROW space
    }
    // FqTarget ops, /root/reference/src/fields/fq_target.rs:144-187
    uint32_t add(uint32_t a, uint32_t b) { return emit(WK_ADD, a, b); }
    uint32_t sub(uint32_t a, uint32_t b) { return emit(WK_SUB, a, b); }
    uint32_t mul(uint32_t a, uint32_t b) { return emit(WK_MUL, a, b); }
    uint32_t neg(uint32_t a) { return emit(WK_NEG, a, a); }
};
// rows of a level that the lanes of one proof take together (BLS381_WIT_GROUP: 0 or 1 = every row at its earliest level)
inline int env_group() {
    const char* e = getenv("BLS381_WIT_GROUP");
    const int g = e ? atoi(e) : 16;
    return g < 0 || g > 64 ? 16 : g;
}
inline uint32_t cst(int id) { return WREF_CONST | (uint32_t)id; }
inline uint32_t inp(int id) { return WREF_INPUT | (uint32_t)id; }

struct F2 { uint32_t c[2]; };
struct F6 { uint32_t c[6]; };    // MyFq6 order  [c00,c10,c20,c01,c11,c21]
struct F12 { uint32_t c[12]; };  // MyFq12 order

// ---- Fq2Target, /root/reference/src/fields/fq2_target.rs
F2 add(WB& w, F2 a, F2 b) { return {{w.add(a.c[0], b.c[0]), w.add(a.c[1], b.c[1])}}; }      // :94-104
F2 sub(WB& w, F2 a, F2 b) { return {{w.sub(a.c[0], b.c[0]), w.sub(a.c[1], b.c[1])}}; }      // :117-127
F2 neg(WB& w, F2 a) { return {{w.neg(a.c[0]), w.neg(a.c[1])}}; }                            // :106-115
F2 mul(WB& w, F2 a, F2 b) {                                                                 // :145-163
    uint32_t a0b0 = w.mul(a.c[0], b.c[0]);
    uint32_t a1b1 = w.mul(a.c[1], b.c[1]);
    uint32_t c0 = w.sub(a0b0, a1b1);
    uint32_t a0b1 = w.mul(a.c[0], b.c[1]);
    uint32_t a1b0 = w.mul(a.c[1], b.c[0]);
    uint32_t c1 = w.add(a0b1, a1b0);
    return {{c0, c1}};
}
F2 simple_square(WB& w, F2 a) { return mul(w, a, a); }                                      // :218-220
F2 dbl(WB& w, F2 a) { return add(w, a, a); }                                                // :222-224
F2 mul_assign_by_fp(WB& w, F2 a, uint32_t s) { return {{w.mul(a.c[0], s), w.mul(a.c[1], s)}}; }  // :226-234
F2 mul_by_nonresidue(WB& w, F2 a) { return {{w.sub(a.c[0], a.c[1]), w.add(a.c[0], a.c[1])}}; }   // :236-248
F2 conjugate(WB& w, F2 a) { return {{a.c[0], w.neg(a.c[1])}}; }                             // :250-256

// ---- Fq6Target, /root/reference/src/fields/fq6_target.rs
F6 add(WB& w, F6 a, F6 b) { F6 r; for (int i = 0; i < 6; i++) r.c[i] = w.add(a.c[i], b.c[i]); return r; }  // :96-106
F6 sub(WB& w, F6 a, F6 b) { F6 r; for (int i = 0; i < 6; i++) r.c[i] = w.sub(a.c[i], b.c[i]); return r; }  // :119-129
inline void split(F6 s, F2& s0, F2& s1, F2& s2) { s0 = {{s.c[0], s.c[3]}}; s1 = {{s.c[1], s.c[4]}}; s2 = {{s.c[2], s.c[5]}}; }
inline F6 join(F2 t1, F2 t2, F2 t3) { return {{t1.c[0], t2.c[0], t3.c[0], t1.c[1], t2.c[1], t3.c[1]}}; }
F6 mul_by_01(WB& w, F6 s, F2 c0, F2 c1) {                                                   // :198-239
    F2 s0, s1, s2; split(s, s0, s1, s2);
    F2 a_a = mul(w, s0, c0);
    F2 b_b = mul(w, s1, c1);
    F2 t1 = mul(w, s2, c1);
    t1 = mul_by_nonresidue(w, t1);
    t1 = add(w, t1, a_a);
    F2 c0c1 = add(w, c0, c1);
    F2 s0s1 = add(w, s0, s1);
    F2 t2 = mul(w, c0c1, s0s1);
    t2 = sub(w, t2, a_a);
    t2 = sub(w, t2, b_b);
    F2 t3 = mul(w, s2, c0);
    t3 = add(w, t3, b_b);
    return join(t1, t2, t3);
}
F6 mul_by_1(WB& w, F6 s, F2 c1) {                                                           // :241-279
    F2 s0, s1, s2; split(s, s0, s1, s2);
    F2 b_b = mul(w, s1, c1);
    F2 tmp = add(w, s1, s2);
    F2 t1 = mul(w, c1, tmp);
    t1 = sub(w, t1, b_b);
    t1 = mul_by_nonresidue(w, t1);
    tmp = add(w, s0, s1);
    F2 t2 = mul(w, c1, tmp);
    t2 = sub(w, t2, b_b);
    return join(t1, t2, b_b);
}
F6 mul_by_nonresidue(WB& w, F6 s) {                                                         // :282-311
    F2 s0, s1, s2; split(s, s0, s1, s2);
    F2 c0 = mul_by_nonresidue(w, s2);
    return join(c0, s0, s1);
}

// ---- Fq12Target, /root/reference/src/fields/fq12_target.rs
F12 mul(WB& w, const F12& a, const F12& b) {                                                // :118-188
    std::vector<uint32_t> a0b0, a0b1, a1b0, a1b1;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            uint32_t c00 = w.mul(a.c[i], b.c[j]);
            uint32_t c01 = w.mul(a.c[i], b.c[j + 6]);
            uint32_t c10 = w.mul(a.c[i + 6], b.c[j]);
            uint32_t c11 = w.mul(a.c[i + 6], b.c[j + 6]);
            if ((size_t)(i + j) < a0b0.size()) {
                a0b0[i + j] = w.add(a0b0[i + j], c00);
                a0b1[i + j] = w.add(a0b1[i + j], c01);
                a1b0[i + j] = w.add(a1b0[i + j], c10);
                a1b1[i + j] = w.add(a1b1[i + j], c11);
            } else {
                a0b0.push_back(c00); a0b1.push_back(c01); a1b0.push_back(c10); a1b1.push_back(c11);
            }
        }
    uint32_t m[11], pl[11];
    for (int i = 0; i < 11; i++) { m[i] = w.sub(a0b0[i], a1b1[i]); pl[i] = w.add(a0b1[i], a1b0[i]); }
    F12 o;
    for (int i = 0; i < 6; i++) {
        if (i < 5) {
            uint32_t term1 = w.mul(m[i + 6], cst(WC_ONE));
            uint32_t term2 = w.neg(pl[i + 6]);
            uint32_t s = w.add(m[i], term1);
            o.c[i] = w.add(s, term2);
        } else o.c[i] = m[i];
    }
    for (int i = 0; i < 6; i++) {
        if (i < 5) {
            uint32_t term2 = w.mul(pl[i + 6], cst(WC_ONE));
            uint32_t s = w.add(pl[i], m[i + 6]);
            o.c[i + 6] = w.add(s, term2);
        } else o.c[i + 6] = pl[i];
    }
    return o;
}
inline void to_fq6(const F12& a, F6& c0, F6& c1) {                                          // :190-239
    c0 = {{a.c[0], a.c[2], a.c[4], a.c[6], a.c[8], a.c[10]}};
    c1 = {{a.c[1], a.c[3], a.c[5], a.c[7], a.c[9], a.c[11]}};
}
inline F12 from_fq6(F6 c0, F6 c1) {                                                         // :241-256
    return {{c0.c[0], c1.c[0], c0.c[1], c1.c[1], c0.c[2], c1.c[2], c0.c[3], c1.c[3], c0.c[4], c1.c[4], c0.c[5], c1.c[5]}};
}
F12 mul_by_014(WB& w, const F12& f, F2 c0, F2 c1, F2 c4) {                                  // :258-295
    F6 s0, s1; to_fq6(f, s0, s1);
    F6 aa = mul_by_01(w, s0, c0, c1);
    F6 bb = mul_by_1(w, s1, c4);
    F2 o = add(w, c1, c4);
    s1 = add(w, s1, s0);
    s1 = mul_by_01(w, s1, c0, o);
    s1 = sub(w, s1, aa);
    s1 = sub(w, s1, bb);
    s0 = mul_by_nonresidue(w, bb);
    s0 = add(w, s0, aa);
    return from_fq6(s0, s1);
}
F12 conjugate(WB& w, F12 a) {                                                               // :314-323
    for (int k : {1, 3, 5, 7, 9, 11}) a.c[k] = w.neg(a.c[k]);
    return a;
}

// ---- curves/g2.rs
struct Proj { F2 x, y, z; };
struct Coeff { F2 c0, c1, c2; };
Coeff double_in_place(WB& w, Proj& r, uint32_t two_inv) {                                   // :59-98
    F2 a = mul(w, r.x, r.y);
    a = mul_assign_by_fp(w, a, two_inv);
    F2 b = simple_square(w, r.y);
    F2 c = simple_square(w, r.z);
    F2 coeff_b = {{cst(WC_FOUR), cst(WC_FOUR)}};
    F2 c_double = dbl(w, c);
    c_double = add(w, c_double, c);
    F2 e = mul(w, coeff_b, c_double);
    F2 e_double = dbl(w, e);
    F2 f = add(w, e_double, e);
    F2 g = add(w, b, f);
    g = mul_assign_by_fp(w, g, two_inv);
    F2 y_add_z = add(w, r.y, r.z);
    F2 y_z_squared = simple_square(w, y_add_z);
    F2 b_add_c = add(w, b, c);
    F2 h = sub(w, y_z_squared, b_add_c);
    F2 i = sub(w, e, b);
    F2 j = simple_square(w, r.x);
    F2 e_square = simple_square(w, e);
    F2 b_sub_f = sub(w, b, f);
    F2 nx = mul(w, a, b_sub_f);
    F2 g_square = simple_square(w, g);
    F2 e_sq_double = dbl(w, e_square);
    e_sq_double = add(w, e_sq_double, e_square);
    F2 ny = sub(w, g_square, e_sq_double);
    F2 nz = mul(w, b, h);
    F2 j_double = dbl(w, j);
    F2 j_triple = add(w, j_double, j);
    r = Proj{nx, ny, nz};
    return Coeff{i, j_triple, neg(w, h)};
}
Coeff add_in_place(WB& w, Proj& r, F2 qx, F2 qy) {                                          // :100-129
    F2 qy_z = mul(w, qy, r.z);
    F2 theta = sub(w, r.y, qy_z);
    F2 qx_z = mul(w, qx, r.z);
    F2 lambda = sub(w, r.x, qx_z);
    F2 c = simple_square(w, theta);
    F2 d = simple_square(w, lambda);
    F2 e = mul(w, lambda, d);
    F2 f = mul(w, r.z, c);
    F2 g = mul(w, r.x, d);
    F2 g_double = dbl(w, g);
    F2 e_add_f = add(w, e, f);
    F2 h = sub(w, e_add_f, g_double);
    F2 nx = mul(w, lambda, h);
    F2 g_sub_h = sub(w, g, h);
    F2 e_mul_y = mul(w, e, r.y);
    F2 theta_gh = mul(w, theta, g_sub_h);
    F2 ny = sub(w, theta_gh, e_mul_y);
    F2 nz = mul(w, r.z, e);
    F2 lambda_qy = mul(w, lambda, qy);
    F2 theta_qx = mul(w, theta, qx);
    F2 j = sub(w, theta_qx, lambda_qy);
    r = Proj{nx, ny, nz};
    return Coeff{j, neg(w, theta), lambda};
}

const uint64_t BLSX = BLS381_X;

// ---- final_exponentiation_helpers.rs
F12 frobenius_map(WB& w, const F12& a, int power) {                                         // :14-57
    F2 out[6];
    for (int i = 0; i < 6; i++) {
        F2 x = {{a.c[i], a.c[i + 6]}};
        if (power % 2) x = conjugate(w, x);
        if (i == 0) out[i] = x;  // frob_coeff == one
        else {
            int base = (power == 1 ? WC_FROB1_BASE : WC_FROB2_BASE) + 2 * (i - 1);
            out[i] = mul(w, x, F2{{cst(base), cst(base + 1)}});
        }
    }
    F12 o;
    for (int i = 0; i < 6; i++) { o.c[i] = out[i].c[0]; o.c[i + 6] = out[i].c[1]; }
    return o;
}
F12 pow_target(WB& w, const F12& a) {                                                       // :59-85
    F12 res = a;
    for (int i = 62; i >= 0; i--) {
        res = mul(w, res, res);
        if ((BLSX >> i) & 1) res = mul(w, res, a);
    }
    return res;
}

WitnessProgram build_program(int group) {
    WB w;
    WitnessProgram P;
    // G2PreparedTarget::from, curves/g2.rs:148-178
    uint32_t two = w.add(cst(WC_ONE), cst(WC_ONE));
    uint32_t two_inv = w.emit(WK_INV, two, two);
    F2 qx = {{inp(2), inp(3)}}, qy = {{inp(4), inp(5)}};
    Proj r{qx, qy, F2{{cst(WC_ONE), cst(WC_ZERO)}}};
    std::vector<Coeff> coeffs;
    for (int i = 62; i >= 0; i--) {
        coeffs.push_back(double_in_place(w, r, two_inv));
        if ((BLSX >> i) & 1) coeffs.push_back(add_in_place(w, r, qx, qy));
    }
    // multi_miller_loop, miller_loop.rs:19-61 (one pair; multiply_elements is the identity)
    F12 f;
    for (int k = 0; k < 12; k++) f.c[k] = cst(k == 0 ? WC_ONE : WC_ZERO);
    size_t k = 0;
    uint32_t px = inp(0), py = inp(1);
    auto ell_target = [&](const Coeff& c) {                                                 // :63-79
        F2 c2 = mul_assign_by_fp(w, c.c2, py);
        F2 c1 = mul_assign_by_fp(w, c.c1, px);
        f = mul_by_014(w, f, c.c0, c1, c2);
    };
    for (int i = 62; i >= 0; i--) {
        f = mul(w, f, f);
        ell_target(coeffs[k++]);
        if ((BLSX >> i) & 1) ell_target(coeffs[k++]);
    }
    f = conjugate(w, f);
    // final_exponentiation.rs:12-21 easy_part
    F12 a = f;
    F12 f1 = conjugate(w, a);
    // Fq12Target::inv (fq12_target.rs:302-312): generator output, then x * inv connected to one
    P.gen_start = w.kinds.size();
    for (int i = 0; i < 12; i++) P.inv_input_refs[i] = a.c[i];
    F12 inv;
    for (int i = 0; i < 12; i++) inv.c[i] = w.emit(WK_GEN, 0, 0);
    (void)mul(w, a, inv);
    F12 f2 = mul(w, f1, inv);                                                               // div, :297-300
    F12 f3 = frobenius_map(w, f2, 2);
    F12 rr = mul(w, f3, f2);
    // hard_part_target, final_exponentiation.rs:23-55
    F12 y0 = mul(w, rr, rr);
    F12 y1 = pow_target(w, rr);
    y1 = conjugate(w, y1);
    F12 y2 = conjugate(w, rr);
    y1 = mul(w, y1, y2);
    y2 = pow_target(w, y1);
    y2 = conjugate(w, y2);
    y1 = conjugate(w, y1);
    y1 = mul(w, y1, y2);
    y2 = pow_target(w, y1);
    y2 = conjugate(w, y2);
    y1 = frobenius_map(w, y1, 1);
    y1 = mul(w, y1, y2);
    F12 r2 = mul(w, rr, y0);
    y0 = pow_target(w, y1);
    y0 = conjugate(w, y0);
    y2 = pow_target(w, y0);
    y2 = conjugate(w, y2);
    y0 = frobenius_map(w, y1, 2);
    y1 = conjugate(w, y1);
    y1 = mul(w, y1, y2);
    y1 = mul(w, y1, y0);
    (void)mul(w, r2, y1);

    P.group = group;
    {   // BLS381_WIT_SCHED=levels: dependency levels with a barrier after each; default: rounds in row order
        const char* e = getenv("BLS381_WIT_SCHED");
        P.rounds = !(e && std::string(e) == "levels") && group > 1;
    }
    P.n_rows = w.kinds.size();
    P.kinds = w.kinds;
    for (uint8_t kd : w.kinds) P.counts[kd]++;
    P.mul_index.assign(P.n_rows, 0xffffffffu);
    {
        uint32_t k = 0;
        for (size_t r = 0; r < P.n_rows; r++) if (P.kinds[r] == WK_MUL) P.mul_index[r] = k++;
    }
    const size_t phase_lo[2] = {0, P.gen_start + 12}, phase_hi[2] = {P.gen_start, P.n_rows};
    for (int phase = 0; phase < 2; phase++)
        for (size_t i = phase_lo[phase]; i < phase_hi[phase]; i++)
            P.circuit_ops[phase].push_back(WitnessOp{(uint32_t)i | ((uint32_t)w.kinds[i] << 28), w.ra[i], w.rb[i]});

    // ---- what the kernel runs (WitnessProgram::ops): products get a left operand times 2^392
    struct XRow { uint8_t kind; uint32_t id, a, b; };
    std::vector<XRow> xrows[2];
    {
        auto is_row = [](uint32_t ref) { return (ref >> 30) == 0; };
        auto times_r = [](uint32_t ref) {  // a constant or an input in its form times 2^392
            return (ref >> 30) == 1 ? cst((int)(ref & WREF_MASK) + WC_TIMES_R) : inp((int)(ref & WREF_MASK) + WIN_COUNT);
        };
        // A product of the two factors of an earlier product, in either order, is a copy of that row: a squaring
        // Fq12Target::mul(x, x) asks for x_i * x_j and x_j * x_i separately (66 of its 144 products), and most
        // Fq12 products of the circuit are the squarings of pow_target and of the Miller loop.
        const uint32_t NONE = 0xffffffffu;
        std::vector<uint32_t> copy_of(P.n_rows, NONE);
        {
            std::map<std::pair<uint32_t, uint32_t>, uint32_t> first;
            const size_t copy_distance = getenv("BLS381_WIT_COPY_DISTANCE") ? (size_t)atol(getenv("BLS381_WIT_COPY_DISTANCE")) : 1u << 30;
            for (size_t i = 0; i < P.n_rows; i++)
                if (w.kinds[i] == WK_MUL) {
                    const std::pair<uint32_t, uint32_t> key{std::min(w.ra[i], w.rb[i]), std::max(w.ra[i], w.rb[i])};
                    auto it = first.find(key);
                    if (it == first.end()) first[key] = (uint32_t)i;
                    else if (i - it->second <= copy_distance) copy_of[i] = it->second;
                    else it->second = (uint32_t)i;  // too far back to still be in the row cache: multiply again, and later copies take this row
                }
        }
        // in how many products between two rows a row is a factor: the factor used more often gets the shadow
        std::vector<uint32_t> uses(P.n_rows, 0);
        for (size_t i = 0; i < P.n_rows; i++)
            if (w.kinds[i] == WK_MUL && copy_of[i] == NONE && is_row(w.ra[i]) && is_row(w.rb[i])) {
                uses[w.ra[i] & WREF_MASK]++;
                if (w.rb[i] != w.ra[i]) uses[w.rb[i] & WREF_MASK]++;
            }
        std::vector<uint32_t> shadow_of(P.n_rows, NONE);
        uint32_t next_shadow = (uint32_t)P.n_rows;
        for (int phase = 0; phase < 2; phase++) {
            const uint32_t first = next_shadow;
            for (size_t i = phase_lo[phase]; i < phase_hi[phase]; i++) {
                const uint32_t a = w.ra[i], b = w.rb[i];
                if (w.kinds[i] != WK_MUL) { xrows[phase].push_back(XRow{w.kinds[i], (uint32_t)i, a, b}); continue; }
                if (copy_of[i] != NONE) {
                    xrows[phase].push_back(XRow{WK_ADD, (uint32_t)i, WREF_ROW | copy_of[i], cst(WC_ZERO)});
                    continue;
                }
                if (a == cst(WC_ONE) || b == cst(WC_ONE)) {  // x * 1: a copy
                    xrows[phase].push_back(XRow{WK_ADD, (uint32_t)i, a == cst(WC_ONE) ? b : a, cst(WC_ZERO)});
                    continue;
                }
                if (!is_row(a)) { xrows[phase].push_back(XRow{WK_MUL, (uint32_t)i, times_r(a), b}); continue; }
                if (!is_row(b)) { xrows[phase].push_back(XRow{WK_MUL, (uint32_t)i, times_r(b), a}); continue; }
                const uint32_t ia = a & WREF_MASK, ib = b & WREF_MASK;
                uint32_t left = NONE, right = 0;
                if (shadow_of[ia] != NONE) { left = shadow_of[ia]; right = b; }
                else if (shadow_of[ib] != NONE) { left = shadow_of[ib]; right = a; }
                else {
                    const bool take_a = uses[ia] >= uses[ib];
                    const uint32_t x = take_a ? ia : ib;
                    shadow_of[x] = next_shadow++;
                    xrows[phase].push_back(XRow{WK_MUL, shadow_of[x], cst(WC_R2), WREF_ROW | x});
                    left = shadow_of[x]; right = take_a ? b : a;
                }
                xrows[phase].push_back(XRow{WK_MUL, (uint32_t)i, WREF_ROW | left, right});
            }
            P.n_shadow[phase] = next_shadow - first;
        }
        P.n_ext = next_shadow;
    }
    // row references share their word with the row-cache fields: 18 bits of row number
    if (P.n_ext > (size_t)WROW_MASK + 1) { P.error = "witness trace has more rows than an operand reference can name"; return P; }

    // level scheduling per phase
    std::vector<uint32_t> level(P.n_ext, 0);
    for (int phase = 0; phase < 2; phase++) {
        const std::vector<XRow>& X = xrows[phase];
        const size_t nx = X.size();
        std::vector<int32_t> local(P.n_ext, -1);  // position in X of the rows this phase makes
        for (size_t q = 0; q < nx; q++) local[X[q].id] = (int32_t)q;
        // rows of the other phase (and generator rows) are complete before this phase starts
        auto in_phase = [&](uint32_t ref) { return (ref >> 30) == 0 && local[ref & WREF_MASK] >= 0; };
        uint32_t maxl = 0;
        for (size_t q = 0; q < nx; q++) {  // (X is in an order in which every row comes after the rows it reads)
            const uint32_t la = in_phase(X[q].a) ? level[X[q].a & WREF_MASK] + 1 : 0;
            const uint32_t lb = in_phase(X[q].b) ? level[X[q].b & WREF_MASK] + 1 : 0;
            level[X[q].id] = std::max(la, lb);
            maxl = std::max(maxl, level[X[q].id]);
        }
        // Which level a row runs in.  The lanes of a proof take a level's rows `group` at a time, and a group that
        // holds one product costs as much as a full group of products, so rows that can wait are held back until
        // they fill a group: every row has a latest level (the one before its first consumer's latest level), a row
        // at its latest level must run, and the others ready by then fill that group up, or run when a whole group
        // of them is ready.  Products first in every level, then the rest, so that few groups hold both.
        std::vector<std::vector<uint32_t>> by(maxl + 1);  // positions in X
        if (P.rounds) {
            // Rounds in row order.  X is in row order (a shadow row right before the first product that needs it), so
            // the position in X is the priority: a round takes the ready operations of one kind with the lowest
            // positions.  The other operations go first whenever fewer than `group` products are ready or their first
            // is earlier in the trace than the first ready product: they are cheap (a round costs what its dearest
            // operation costs), they make products ready, and the sums that follow four products in the trace then
            // run right behind them, so that the two halves of a 128-byte line of the trace are written a round or two
            // apart, while the line is still in the L2 cache (in dependency levels the sums of an Fq12 product come up to
            // six levels after its 144 products, the half-written lines leave the cache in between, and the trace's
            // bytes reach memory at half the rate a sequential writer gets).
            by.clear();
            std::vector<uint32_t> npred(nx, 0);
            std::vector<std::vector<uint32_t>> users(nx);
            for (size_t q = 0; q < nx; q++) {
                const uint32_t refs[2] = {X[q].a, X[q].b};
                for (int k = 0; k < (refs[0] == refs[1] ? 1 : 2); k++)
                    if (in_phase(refs[k])) { users[local[refs[k] & WREF_MASK]].push_back((uint32_t)q); npred[q]++; }
            }
            typedef std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> MinQ;
            MinQ ready[2];  // products, others
            for (size_t q = 0; q < nx; q++) if (!npred[q]) ready[X[q].kind == WK_MUL ? 0 : 1].push((uint32_t)q);
            size_t done = 0;
            while (!ready[0].empty() || !ready[1].empty()) {
                int k = 0;
                if (ready[0].empty()) k = 1;
                else if (!ready[1].empty() && (ready[0].size() < (size_t)group || ready[1].top() < ready[0].top())) k = 1;
                std::vector<uint32_t> round;
                while (round.size() < (size_t)group && !ready[k].empty()) { round.push_back(ready[k].top()); ready[k].pop(); }
                for (uint32_t q : round)
                    for (uint32_t u : users[q])
                        if (--npred[u] == 0) ready[X[u].kind == WK_MUL ? 0 : 1].push(u);
                done += round.size();
                by.push_back(std::move(round));
            }
            if (done != nx) { P.error = "witness round scheduling left rows over"; return P; }
        } else if (group <= 1) {
            for (size_t q = 0; q < nx; q++) by[level[X[q].id]].push_back((uint32_t)q);
        } else {
            std::vector<uint32_t> late(nx, maxl), npred(nx, 0), ready_at(nx, 0);
            std::vector<std::vector<uint32_t>> users(nx);
            for (size_t q = 0; q < nx; q++) {
                const uint32_t refs[2] = {X[q].a, X[q].b};
                for (int k = 0; k < (refs[0] == refs[1] ? 1 : 2); k++)
                    if (in_phase(refs[k])) { users[local[refs[k] & WREF_MASK]].push_back((uint32_t)q); npred[q]++; }
            }
            for (size_t q = nx; q-- > 0;)
                for (uint32_t u : users[q]) late[q] = std::min(late[q], late[u] - 1);
            std::vector<std::vector<uint32_t>> arrives(maxl + 2);
            for (size_t q = 0; q < nx; q++) if (!npred[q]) arrives[0].push_back((uint32_t)q);
            std::vector<uint32_t> pool[2];  // waiting rows: products, others
            for (uint32_t l = 0; l <= maxl; l++) {
                for (uint32_t q : arrives[l]) pool[X[q].kind == WK_MUL ? 0 : 1].push_back(q);
                for (int k = 0; k < 2; k++) {
                    auto& pl = pool[k];
                    std::stable_sort(pl.begin(), pl.end(), [&](uint32_t x, uint32_t y) { return late[x] < late[y]; });
                    size_t must = 0;
                    while (must < pl.size() && late[pl[must]] <= l) must++;
                    size_t take = must ? (must + group - 1) / group * group : pl.size() / group * group;
                    take = std::min(take, pl.size());
                    for (size_t t = 0; t < take; t++) {
                        const uint32_t q = pl[t];
                        by[l].push_back(q);
                        for (uint32_t u : users[q]) {
                            ready_at[u] = std::max(ready_at[u], l + 1);
                            if (--npred[u] == 0) arrives[ready_at[u]].push_back(u);
                        }
                    }
                    pl.erase(pl.begin(), pl.begin() + take);
                }
            }
            if (!pool[0].empty() || !pool[1].empty()) { P.error = "witness level assignment left rows over"; return P; }
        }
        P.level_start[phase].push_back(0);
        for (auto& rows : by) {
            for (uint32_t q : rows) P.ops[phase].push_back(WitnessOp{X[q].id | ((uint32_t)X[q].kind << 28), X[q].a, X[q].b});
            P.level_start[phase].push_back((uint32_t)P.ops[phase].size());
        }
        if (getenv("BLS381_WIT_STATS")) {  // dependency-level widths
            size_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (auto& rows : by) {
                size_t wd = rows.size();
                hist[wd <= 4 ? 0 : wd <= 8 ? 1 : wd <= 16 ? 2 : wd <= 32 ? 3 : wd <= 64 ? 4 : wd <= 128 ? 5 : wd <= 256 ? 6 : 7]++;
            }
            fprintf(stderr, "witness phase %d: %zu operations (%zu of them shadow rows) in %zu levels; widths <=4:%zu <=8:%zu <=16:%zu <=32:%zu <=64:%zu <=128:%zu <=256:%zu more:%zu\n",
                    phase, nx, P.n_shadow[phase], by.size(), hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
            size_t iters = 0, with_mul = 0, with_lin = 0, mixed = 0, n_mul = 0;
            for (auto& rows : by)
                for (size_t j = 0; j < rows.size(); j += 16) {
                    bool m = false, l = false;
                    for (size_t t = j; t < std::min(rows.size(), j + (size_t)16); t++) { (X[rows[t]].kind == WK_MUL ? m : l) = true; n_mul += X[rows[t]].kind == WK_MUL; }
                    iters++; with_mul += m; with_lin += l; mixed += m && l;
                }
            fprintf(stderr, "  sixteen rows at a time: %zu groups, %zu with a product (%zu products), %zu with other rows, %zu with both\n", iters, with_mul, n_mul, with_lin, mixed);
        }
    }
    return P;
}

}  // namespace

// Row cache plan.  The kernel gives every proof `cache_rows` rows of LDS; which finished rows go there, and which
// operand reads find their row there, is decided here, once, because the program is the same for every proof:
// the lanes of a proof take the operations of a level `lpp` at a time (one iteration: all its reads, then all its
// writes), and walking the iterations in order with every row's later reads known, a finished row replaces the
// resident row whose next read lies furthest ahead (never one placed in the same iteration: two lanes would
// write one entry).  Three operand reads in four then come from LDS at 32 rows, where a cache addressed by row
// number holds one in three.  The operations come back with the cache fields of WitnessOp filled in.
static std::vector<WitnessOp> plan_in_order(const WitnessProgram& P, int phase, std::vector<WitnessOp> ops, int lpp, int cache_rows,
                                            size_t* reads_out, size_t* hits_out) {
    const auto& ls = P.level_start[phase];
    size_t reads = 0, hits = 0;
    auto n_operands = [](const WitnessOp& o) { const uint32_t k = (o.dst >> 28) & WKIND_MASK; return k == WK_NEG || k == WK_INV ? 1 : 2; };
    // iteration number of every operation
    std::vector<uint32_t> iter_of(ops.size());
    uint32_t it = 0;
    for (size_t l = 0; l + 1 < ls.size(); l++)
        for (uint32_t i = ls[l]; i < ls[l + 1]; i += (uint32_t)lpp, it++)
            for (uint32_t j = i; j < std::min(i + (uint32_t)lpp, ls[l + 1]); j++) iter_of[j] = it;
    std::vector<std::vector<uint32_t>> uses(P.n_ext);  // iterations that read a row, ascending
    for (size_t j = 0; j < ops.size(); j++) {
        uint32_t* refs[2] = {&ops[j].a, &ops[j].b};
        for (int k = 0; k < n_operands(ops[j]); k++)
            if ((*refs[k] >> 30) == 0) { uses[*refs[k] & WROW_MASK].push_back(iter_of[j]); reads++; }
    }
    // round in which this phase makes a row (-1: before the phase), and the last round before the current one in which
    // the wave loads from global memory
    std::vector<int64_t> made_at(P.n_ext, -1);
    for (size_t j = 0; j < ops.size(); j++) made_at[ops[j].dst & WROW_MASK] = iter_of[j];
    int64_t last_wait = -1;
    size_t drains = 0;
    std::vector<uint8_t> read_from_memory(P.n_ext, 0);  // some read of the row goes to global memory
    {
        std::vector<size_t> cursor(P.n_ext, 0);
        auto next_use = [&](uint32_t row, uint32_t now) -> uint32_t {
            size_t& c = cursor[row];
            while (c < uses[row].size() && uses[row][c] <= now) c++;
            return c < uses[row].size() ? uses[row][c] : 0xffffffffu;
        };
        std::vector<int> slot_of(P.n_ext, -1);
        std::vector<int64_t> row_in(cache_rows, -1), placed_at(cache_rows, -1);
        for (size_t j = 0; j < ops.size();) {
            size_t e = j;
            while (e < ops.size() && iter_of[e] == iter_of[j]) e++;
            const uint32_t now = iter_of[j];
            bool reads_memory = false;
            for (size_t q = j; q < e; q++) {
                uint32_t* refs[2] = {&ops[q].a, &ops[q].b};
                for (int k = 0; k < n_operands(ops[q]); k++) {
                    if ((*refs[k] >> 30) != 0) { reads_memory |= !wref_is_immediate(*refs[k]); continue; }
                    const uint32_t row = *refs[k] & WROW_MASK;
                    const int sl = slot_of[row];
                    if (sl >= 0) { *refs[k] |= WCACHE_FLAG | ((uint32_t)sl << WCACHE_SLOT_SHIFT); hits++; continue; }
                    reads_memory = true;
                    read_from_memory[row] = 1;
                    // Without a barrier between rounds, a row comes from global memory only if the wave has waited for
                    // its memory operations since the row was stored: it waits whenever it loads from memory (its
                    // memory operations finish in the order it issued them), so a load issued in a later round than
                    // such a wait, itself later than the store's round, finds the row.  Where the plan cannot show that
                    // wait, the operation asks for one before its reads.
                    if (P.rounds && made_at[row] >= 0 && !(made_at[row] < last_wait)) { ops[q].dst |= WDRAIN_FLAG; drains++; }
                }
            }
            if (reads_memory) last_wait = (int64_t)now;
            for (size_t q = j; q < e && cache_rows > 0; q++) {
                const uint32_t row = ops[q].dst & WROW_MASK;
                const uint32_t nu = next_use(row, now);
                if (nu == 0xffffffffu) continue;
                int victim = -1;
                uint32_t furthest = 0;
                for (int sl = 0; sl < cache_rows; sl++) {
                    if (placed_at[sl] == (int64_t)now) continue;
                    const uint32_t u = row_in[sl] < 0 ? 0xffffffffu : next_use((uint32_t)row_in[sl], now);
                    if (victim < 0 || u > furthest) { victim = sl; furthest = u; }
                    if (u == 0xffffffffu) break;
                }
                if (victim < 0 || furthest <= nu) continue;
                if (row_in[victim] >= 0) slot_of[row_in[victim]] = -1;
                row_in[victim] = row;
                placed_at[victim] = now;
                slot_of[row] = victim;
                ops[q].dst |= WCACHE_FLAG | ((uint32_t)victim << WCACHE_SLOT_SHIFT);
            }
            j = e;
        }
    }
    // A shadow row that every reader finds in the cache need not exist in global memory (the trace's own rows all do:
    // they are the output).  Rows the other phase reads are read from memory there.
    if (cache_rows > 0) {
        for (const WitnessOp& o : P.ops[1 - phase]) {
            if ((o.a >> 30) == 0) read_from_memory[o.a & WROW_MASK] = 1;
            if (n_operands(o) == 2 && (o.b >> 30) == 0) read_from_memory[o.b & WROW_MASK] = 1;
        }
        for (WitnessOp& o : ops) {
            const uint32_t row = o.dst & WROW_MASK;
            if (row >= P.n_rows && !read_from_memory[row]) o.dst |= WNOSTORE_FLAG;
        }
    }
    if (reads_out) *reads_out = reads;
    if (hits_out) *hits_out = hits;
    if (getenv("BLS381_WIT_STATS")) fprintf(stderr, "witness phase %d plan: %zu operations wait for earlier stores before their reads\n", phase, drains);
    return ops;
}

// The plan for the operations in the order the program holds them, then twice more with the operations of every
// level reordered by what the plan before found: a wave waits for memory as soon as one of its 64 lanes reads a row
// from the trace in global memory, so within a level (products first, as before) the operations with such a read
// come together, and the groups of `lpp` after them read LDS only.  (The order within a level is free: its rows do
// not depend on one another.)  BLS381_WIT_GATHER_MISSES=0 keeps the program's order.
std::vector<WitnessOp> plan_row_cache(const WitnessProgram& P, int phase, int lpp, int cache_rows, size_t* reads_out,
                                      size_t* hits_out) {
    std::vector<WitnessOp> planned = plan_in_order(P, phase, P.ops[phase], lpp, cache_rows, reads_out, hits_out);
    const auto& ls = P.level_start[phase];
    auto stats = [&](const char* what) {  // groups of lpp operations in which no lane reads the trace in global memory
        if (!getenv("BLS381_WIT_STATS")) return;
        size_t groups = 0, quiet = 0;
        for (size_t l = 0; l + 1 < ls.size(); l++)
            for (uint32_t i = ls[l]; i < ls[l + 1]; i += (uint32_t)lpp) {
                bool miss = false;
                for (uint32_t j = i; j < std::min(i + (uint32_t)lpp, ls[l + 1]); j++) {
                    const WitnessOp& o = planned[j];
                    const uint32_t k = (o.dst >> 28) & WKIND_MASK;
                    miss |= (o.a >> 30) == 0 && !(o.a & WCACHE_FLAG);
                    miss |= !(k == WK_NEG || k == WK_INV) && (o.b >> 30) == 0 && !(o.b & WCACHE_FLAG);
                }
                groups++; quiet += !miss;
            }
        fprintf(stderr, "witness phase %d plan (%s): %zu of %zu groups of %d read LDS only\n", phase, what, quiet, groups, lpp);
    };
    stats("program order");
    const char* e = getenv("BLS381_WIT_GATHER_MISSES");
    if (cache_rows == 0 || (e && atoi(e) == 0)) return planned;
    const uint32_t plan_bits = WCACHE_FLAG | (WCACHE_SLOT_MASK << WCACHE_SLOT_SHIFT);
    for (int pass = 0; pass < 2; pass++) {
        // products first, then the other kinds one after the other (lanes of a group that hold different kinds run
        // one kind after the other); within a kind, the operations that read the trace first
        auto class_of = [&](const WitnessOp& o) {
            const uint32_t k = (o.dst >> 28) & WKIND_MASK;
            const bool two = !(k == WK_NEG || k == WK_INV);
            const bool miss = ((o.a >> 30) == 0 && !(o.a & WCACHE_FLAG)) || (two && (o.b >> 30) == 0 && !(o.b & WCACHE_FLAG));
            const int rank = k == WK_MUL ? 0 : k == WK_ADD ? 1 : k == WK_SUB ? 2 : k == WK_NEG ? 3 : 4;
            return rank * 2 + (miss ? 0 : 1);
        };
        std::vector<WitnessOp> order;
        order.reserve(planned.size());
        for (size_t l = 0; l + 1 < ls.size(); l++)
            for (int c = 0; c < 10; c++) {
                const size_t first = order.size();
                for (uint32_t j = ls[l]; j < ls[l + 1]; j++)
                    if (class_of(planned[j]) == c) {
                        WitnessOp o = planned[j];
                        o.dst &= ~(plan_bits | WDRAIN_FLAG | WNOSTORE_FLAG);
                        if ((o.a >> 30) == 0) o.a &= ~plan_bits;
                        if ((o.b >> 30) == 0) o.b &= ~plan_bits;
                        order.push_back(o);
                    }
                // by row number within the class: neighbouring lanes then store neighbouring rows of the trace
                std::sort(order.begin() + first, order.end(), [](const WitnessOp& x, const WitnessOp& y) { return (x.dst & WROW_MASK) < (y.dst & WROW_MASK); });
            }
        planned = plan_in_order(P, phase, std::move(order), lpp, cache_rows, reads_out, hits_out);
        stats("reads of the trace gathered");
    }
    return planned;
}

std::vector<WitnessOp> pad_rounds(const WitnessProgram& P, int phase, const std::vector<WitnessOp>& planned, int lpp) {
    const auto& ls = P.level_start[phase];
    std::vector<WitnessOp> out;
    out.reserve(planned.size() + ls.size() * (size_t)lpp / 2);
    const WitnessOp nop{(uint32_t)WK_NOP << 28, 0, 0};
    for (size_t l = 0; l + 1 < ls.size(); l++)
        for (uint32_t i = ls[l]; i < ls[l + 1]; i += (uint32_t)lpp) {
            const uint32_t e = std::min(i + (uint32_t)lpp, ls[l + 1]);
            const bool level_ends = !P.rounds && e == ls[l + 1];
            for (uint32_t j = i; j < i + (uint32_t)lpp; j++) {
                WitnessOp o = j < e ? planned[j] : nop;
                if (level_ends) o.dst |= WBARRIER_FLAG;
                out.push_back(o);
            }
        }
    // whole blocks of WOPS_PER_BLOCK operations (the kernel fetches its operations a block at a time)
    if (WOPS_PER_BLOCK % lpp == 0)
        while (out.size() % WOPS_PER_BLOCK) out.push_back(nop);
    return out;
}

const WitnessProgram& witness_program() {
    // one program per level assignment asked for (BLS381_WIT_GROUP), built on first use and kept
    static std::mutex mu;
    static std::map<int, std::unique_ptr<WitnessProgram>> built;
    const int group = env_group();
    const char* sched = getenv("BLS381_WIT_SCHED");
    std::lock_guard<std::mutex> g(mu);
    auto& p = built[group * 2 + (sched && std::string(sched) == "levels" ? 1 : 0)];
    if (!p) p.reset(new WitnessProgram(build_program(group)));
    return *p;
}

// ================================================================= device
// a row of the circuit's own operand spaces (the auxiliary-value kernel's loads)
__device__ __forceinline__ void w_load(u32 ref, const u32* T, const u32* consts, const u32* inputs, u32* x) {
    const u32* p;
    u32 sp = ref >> 30, idx = ref & WROW_MASK;
    if (sp == 0) p = T + (size_t)idx * 12;
    else if (sp == 1) p = consts + (size_t)idx * 12;
    else p = inputs + (size_t)idx * 12;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
}

// lpp lanes per proof: a workgroup of blockDim.x lanes serves blockDim.x / lpp proofs side by side.
// Nine operand reads in ten name a row written at most 256 rows earlier, six in ten one at most 64 rows earlier,
// but with thousands of proofs in flight those rows have left every hardware cache by then.  So each proof keeps
// `cache_rows` rows in LDS as well, managed by the plan in the operations (plan_row_cache): a reference with
// WCACHE_FLAG reads its row from the named entry, a result with the flag is written there besides the trace in
// global memory, which always holds every finished row.  The lanes of a proof sit in one wave (lpp divides 64) and
// a wave executes its LDS instructions in order, all lanes together: every lane's reads of an iteration precede
// every lane's writes of it, as the plan assumes.
// Rows from `rows` on are the shadow rows (WitnessProgram::ops): `shadow` holds n_shadow of them per proof.
// (Compiled for five waves per SIMD, 96 registers, so that LDS and not the register file decides how many workgroups a
// CU holds.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void witness_kernel(const WitnessOp* ops, int n_rounds,
                                                      u32* trace, size_t rows, u32* shadow, size_t n_shadow, const u32* consts,
                                                      const u32* inputs, int lpp, int nproofs, int cache_rows, u32* trash) {
    extern __shared__ uint4 wcache[];
    const int proof = blockIdx.x * (blockDim.x / lpp) + threadIdx.x / lpp;
    const int sub = threadIdx.x % lpp;
    const bool live = proof < nproofs;
    u32* T = trace + (size_t)(live ? proof : 0) * rows * 12;
    u32* S = shadow + (size_t)(live ? proof : 0) * n_shadow * 12;
    auto row_at = [&](u32 idx) { return idx < rows ? T + (size_t)idx * 12 : S + (size_t)(idx - (u32)rows) * 12; };
    // where the lanes without a row in a round put their three stores: one 128-byte line per wave (the same 48 bytes
    // from every such lane, round after round: the line stays in the L2 cache)
    u32* const my_trash = trash + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    const u32* in = inputs + (size_t)(live ? proof : 0) * WIN_TOTAL * 12;
    const u32 P[12] = BLS381_P_LIMBS;
    uint4* cdata = wcache + (size_t)(threadIdx.x / lpp) * cache_rows * 3;  // this proof's entries: 3 uint4 each
    auto load = [&](u32 ref, u32* x) {
        const u32 sp = ref >> 30, idx = ref & WROW_MASK;
        // Two separate paths with the address space in the pointer's type and a plain vector as the loaded type: copies
        // of uint4 structures go through generic pointers, the two paths then merge into one generic load (address
        // chosen by a select), and generic loads count against both memory counters, so that every later wait is a
        // wait for everything, and take the long way to LDS.
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        v4u a, b, c;
        if (ref == (WREF_CONST | (u32)WC_ZERO)) {
            a = b = c = v4u{0u, 0u, 0u, 0u};
        } else if (ref == (WREF_CONST | (u32)WC_R2)) {
            const u32 K[12] = BLS381_POW784_LIMBS;
            a = v4u{K[0], K[1], K[2], K[3]}; b = v4u{K[4], K[5], K[6], K[7]}; c = v4u{K[8], K[9], K[10], K[11]};
        } else if (sp == 0 && (ref & WCACHE_FLAG)) {
            const __attribute__((address_space(3))) v4u* q =
                (const __attribute__((address_space(3))) v4u*)(cdata + (size_t)((ref >> WCACHE_SLOT_SHIFT) & WCACHE_SLOT_MASK) * 3);
            a = q[0]; b = q[1]; c = q[2];
        } else {
            const u32* g = sp == 0 ? row_at(idx) : (sp == 1 ? consts : in) + (size_t)idx * 12;
            const __attribute__((address_space(1))) v4u* q = (const __attribute__((address_space(1))) v4u*)(uintptr_t)g;
            a = q[0]; b = q[1]; c = q[2];
            // (the values are "used" inside this branch: the wait for the loads then sits here, where the loads are;
            // placed behind the join of the branches it would be executed by every round, and a wait for a load is
            // a wait for every store before it)
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        }
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
    };
    // A wave's memory operations finish in the order it issued them and one counter covers loads and stores, so
    // whatever waits for a load also waits for every store issued before it, and a store takes microseconds when the
    // memory system is busy.  A round that loads nothing from global memory waits for no store: the stores of many
    // rounds are then under way while the wave computes (with a load per round for the round's operations the
    // compiler put a wait for all but the two youngest memory operations between a round's stores: every round waited
    // for the stores of the round before, 16,384 proofs took 49 ms where the arithmetic alone takes 31).  Hence:
    //  - the operations come from LDS: rounds are padded to exactly lpp operations and whole blocks of 128 (pad_rounds);
    //    a wave keeps two blocks in LDS, and once per block (128 / lpp rounds) its lanes fetch the block after next
    //    from global memory, two operations each, which they put into LDS a block later;
    //  - the constant zero and 2^784 are immediates (copies and shadow rows, a third of all operations, name them);
    //  - operands come from the LDS row cache (98 % of them); a round in which some lane reads a row, a constant or an
    //    input from global memory does wait;
    //  - every lane stores in every round (lanes without a row write one 128-byte line per wave that stays in L2), so
    //    the stores are straight-line code.
    // (LDS address space in the pointer's type: through generic pointers these would be generic loads and stores, which
    // count against the memory counter as well)
    typedef __attribute__((address_space(3))) u32 lds_u32;
    lds_u32* const ring = (lds_u32*)(reinterpret_cast<u32*>(wcache + (size_t)(blockDim.x / lpp) * cache_rows * 3) + (size_t)(threadIdx.x >> 6) * (WOPS_RING_BYTES / 4));
    const u32 lane = threadIdx.x & 63u;
    const int rounds_per_block = WOPS_PER_BLOCK / lpp, n_blocks = n_rounds / rounds_per_block;  // (lpp divides 64)
    // a lane's share of a block: WOPS_PER_BLOCK / 64 operations
    struct Share { WitnessOp o[WOPS_PER_BLOCK / 64]; };
    auto fetch = [&](int blk, Share& f) {
        const __attribute__((address_space(1))) u32* src = (const __attribute__((address_space(1))) u32*)(uintptr_t)(ops + (size_t)blk * WOPS_PER_BLOCK);
#pragma unroll
        for (int i = 0; i < WOPS_PER_BLOCK / 64; i++)
            f.o[i] = WitnessOp{src[(lane + 64 * i) * 3], src[(lane + 64 * i) * 3 + 1], src[(lane + 64 * i) * 3 + 2]};
    };
    auto stash = [&](int blk, const Share& f) {
        lds_u32* half = ring + (size_t)(blk & 1) * WOPS_PER_BLOCK * 3;
#pragma unroll
        for (int i = 0; i < WOPS_PER_BLOCK / 64; i++) {
            half[(lane + 64 * i) * 3 + 0] = f.o[i].dst; half[(lane + 64 * i) * 3 + 1] = f.o[i].a; half[(lane + 64 * i) * 3 + 2] = f.o[i].b;
        }
    };
    auto op_at = [&](int blk, int r) {
        const lds_u32* q = ring + ((size_t)(blk & 1) * WOPS_PER_BLOCK + (size_t)r * lpp + sub) * 3;
        return WitnessOp{q[0], q[1], q[2]};
    };
    if (n_blocks > 0) { Share f; fetch(0, f); stash(0, f); }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    WitnessOp nxt = n_blocks > 0 ? op_at(0, 0) : WitnessOp{(u32)WK_NOP << 28, 0, 0};
    for (int blk = 0; blk < n_blocks; blk++)
    for (int rd = 0; rd < rounds_per_block; rd++) {
        {
            const WitnessOp op = nxt;
            if (rd + 1 < rounds_per_block) nxt = op_at(blk, rd + 1);
            const bool has = live && ((op.dst >> 28) & WKIND_MASK) != WK_NOP;
            const u32 kind = (op.dst >> 28) & WKIND_MASK, row = op.dst & WROW_MASK;
            // (rounds without barriers: a read the plan could not place behind an earlier wait of this wave)
            if (__any(has && (op.dst & WDRAIN_FLAG))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            u32 a[12], b[12], r[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (has) {
            load(op.a, a);
            if (kind == WK_INV) {
                // the only inv_nonnative on the path inverts the constant 2 (curves/g2.rs:149-151)
                load(WREF_CONST | (u32)WC_TWO_INV, r);
            } else if (kind == WK_MUL) {
                // a = (the left factor) * 2^392, b canonical: one Montgomery product on 28-bit digits is the canonical
                // product (every multiply-accumulate a single instruction; the Barrett product on 12 x 32-bit limbs
                // that this replaces took 336 multiply-accumulates with an add-with-carry behind each)
                load(op.b, b);
                u32 da[D28], db[D28], dr[D28];
                unpack28(a, da);
                unpack28(b, db);
                fp28_mul(da, db, dr);
                canon28(dr);
                pack28(dr, r);
            } else if (kind == WK_ADD) {
                load(op.b, b);
                carry_t c;
                r[0] = add_co(a[0], b[0], c);
#pragma unroll
                for (int k = 1; k < 12; k++) r[k] = addc_co(a[k], b[k], c);
                fp_cond_sub_p(r);
            } else if (kind == WK_SUB) {
                load(op.b, b);
                carry_t c;
                r[0] = sub_co(a[0], b[0], c);
#pragma unroll
                for (int k = 1; k < 12; k++) r[k] = subb_co(a[k], b[k], c);
                const u32 m = carry_mask(c);  // borrow: add p back
                carry_t c2;
                r[0] = add_co(r[0], P[0] & m, c2);
#pragma unroll
                for (int k = 1; k < 12; k++) r[k] = addc_co(r[k], P[k] & m, c2);
            } else {  // WK_NEG: p - a, and 0 stays 0
                u32 nz = 0;
#pragma unroll
                for (int k = 0; k < 12; k++) nz |= a[k];
                const u32 m = nz ? 0xffffffffu : 0u;
                carry_t c;
                r[0] = sub_co(P[0] & m, a[0], c);
#pragma unroll
                for (int k = 1; k < 12; k++) r[k] = subb_co(P[k] & m, a[k], c);
            }
            }
            {
                uint4* d = reinterpret_cast<uint4*>(has && !(op.dst & WNOSTORE_FLAG) ? row_at(row) : my_trash);
                d[0] = make_uint4(r[0], r[1], r[2], r[3]);
                d[1] = make_uint4(r[4], r[5], r[6], r[7]);
                d[2] = make_uint4(r[8], r[9], r[10], r[11]);
            }
            // (the compiler must not move this store above the reads of this iteration, nor the next iteration's
            // reads above it: other lanes of the wave read what this lane replaces)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (has && (op.dst & WCACHE_FLAG)) {
                uint4* q = cdata + (size_t)((op.dst >> WCACHE_SLOT_SHIFT) & WCACHE_SLOT_MASK) * 3;
                q[0] = make_uint4(r[0], r[1], r[2], r[3]);
                q[1] = make_uint4(r[4], r[5], r[6], r[7]);
                q[2] = make_uint4(r[8], r[9], r[10], r[11]);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // a dependency level ends (programs arranged in levels; every operation of the round says so)
            if (op.dst & WBARRIER_FLAG) __syncthreads();
            if (rd + 1 == rounds_per_block && blk + 1 < n_blocks) {
                // the next block goes into the half of the ring that the block before this one has left, the fetch of
                // the block after it starts
                Share f;
                fetch(blk + 1, f);
                stash(blk + 1, f);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                nxt = op_at(blk + 1, 0);
            }
        }
    }
}

// inputs: Montgomery {px,py} and {qx0,qx1,qy0,qy1} -> per proof 6 canonical values, then the same times 2^392
__global__ void witness_inputs_kernel(const u32* g1, const u32* g2, const u32* consts, u32* inputs, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * WIN_TOTAL) return;
    int e = i / WIN_TOTAL, f = i % WIN_TOTAL % WIN_COUNT;
    const bool times_r = i % WIN_TOTAL >= WIN_COUNT;
    const u32* src = f < 2 ? g1 + ((size_t)e * 2 + f) * 12 : g2 + ((size_t)e * 4 + (f - 2)) * 12;
    // x * 2^384 times 1, or times 2^392, divided by 2^384
    const u32* factor = consts + (size_t)(times_r ? WC_TIMES_R + WC_ONE : WC_ONE) * 12;
    u32 a[12], k[12], r[12];
    for (int l = 0; l < 12; l++) { a[l] = src[l]; k[l] = factor[l]; }
    fp_mul(a, k, r);
    for (int l = 0; l < 12; l++) inputs[(size_t)i * 12 + l] = r[l];
}
// x[e][k] = trace row refs[k] of proof e   /   trace generator rows = inv[e][k]
__global__ void witness_gather_kernel(const u32* trace, size_t rows, const u32* refs, u32* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 12) return;
    int e = i / 12, k = i % 12;
    const u32* src = trace + ((size_t)e * rows + (refs[k] & WREF_MASK)) * 12;
    for (int l = 0; l < 12; l++) x[(size_t)i * 12 + l] = src[l];
}
__global__ void witness_scatter_kernel(u32* trace, size_t rows, size_t gen_start, const u32* inv, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 12) return;
    int e = i / 12, k = i % 12;
    u32* dst = trace + ((size_t)e * rows + gen_start + k) * 12;
    for (int l = 0; l < 12; l++) dst[l] = inv[(size_t)i * 12 + l];
}

// Auxiliary values of the non-native gadgets, one thread per (proof, operation), from the finished trace:
//   mul rows: q = (a * b - r) / p (exact), 24 limbs   add rows: 1 when a + b >= p (p was subtracted)
//   sub rows: 1 when a < b (p was added)   neg rows: 1 when a != 0 (computed as 0 - a)   inv row: (a * r - 1) / p
// a, b = the row's operands, r = the row's value.
__global__ void witness_aux_kernel(const WitnessOp* ops, int nops, const u32* mul_index, const u32* trace, size_t rows,
                                   const u32* consts, const u32* inputs, int nproofs, size_t n_mul, u32* mul_quot, uint8_t* carry,
                                   u32* inv_quot) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nproofs * nops) return;
    const int proof = (int)(t / nops);
    const WitnessOp op = ops[t % nops];
    const u32 kind = (op.dst >> 28) & WKIND_MASK, row = op.dst & WROW_MASK;
    const u32* T = trace + (size_t)proof * rows * 12;
    const u32* in = inputs + (size_t)proof * WIN_TOTAL * 12;
    const u32 P[12] = BLS381_P_LIMBS;
    u32 a[12], b[12], r[12];
    w_load(op.a, T, consts, in, a);
    w_load(WREF_ROW | row, T, consts, in, r);
    if (kind == WK_MUL || kind == WK_INV) {
        // (the output's base pointer is what may be null, not an address derived from it)
        u32* const qbase = kind == WK_MUL ? mul_quot : inv_quot;
        if (!qbase) {
            if (carry) carry[(size_t)proof * rows + row] = 0;
            return;
        }
        if (kind == WK_MUL) w_load(op.b, T, consts, in, b);
        else for (int k = 0; k < 12; k++) b[k] = r[k];  // a * a^-1
        u32 w[24];
        for (int k = 0; k < 24; k++) w[k] = 0;
        for (int i = 0; i < 12; i++) {
            u64 c = 0;
            for (int j = 0; j < 12; j++) { c += (u64)a[i] * b[j] + w[i + j]; w[i + j] = (u32)c; c >>= 32; }
            w[i + 12] = (u32)c;
        }
        // minus the remainder (r for a product, 1 for the inverse): the difference is an exact multiple of p
        u32 borrow = 0;
        for (int k = 0; k < 24; k++) {
            const u32 sub = kind == WK_MUL ? (k < 12 ? r[k] : 0u) : (k == 0 ? 1u : 0u);
            w[k] = subb(w[k], sub, borrow);
        }
        // exact division by the odd p, limb by limb from the bottom: q_i = w_i / p mod 2^32, then w -= q_i p 2^(32 i)
        u32 q[24];
        for (int k = 0; k < 24; k++) q[k] = 0;
        for (int i = 0; i < 12; i++) {
            const u32 qi = w[i] * 0x00030003u;  // p^-1 mod 2^32
            q[i] = qi;
            u64 c = 0;
            u32 bw = 0;
            for (int j = 0; j < 12; j++) {
                c += (u64)qi * P[j];
                w[i + j] = subb(w[i + j], (u32)c, bw);
                c >>= 32;
            }
            for (int k = i + 12; k < 24; k++) { w[k] = subb(w[k], (u32)c, bw); c = 0; }
        }
        u32* dst = qbase + (kind == WK_MUL ? ((size_t)proof * n_mul + mul_index[row]) * 24 : (size_t)proof * 24);
        for (int k = 0; k < 24; k++) dst[k] = q[k];
        if (carry) carry[(size_t)proof * rows + row] = 0;
        return;
    }
    uint8_t flag = 0;
    if (kind == WK_ADD) {
        w_load(op.b, T, consts, in, b);
        u32 c = 0, differs = 0;
        for (int k = 0; k < 12; k++) differs |= addc(a[k], b[k], c) ^ r[k];
        flag = (c || differs) ? 1 : 0;
    } else if (kind == WK_SUB) {
        w_load(op.b, T, consts, in, b);
        u32 bw = 0;
        for (int k = 0; k < 12; k++) (void)subb(a[k], b[k], bw);
        flag = (uint8_t)bw;
    } else if (kind == WK_NEG) {
        u32 nz = 0;
        for (int k = 0; k < 12; k++) nz |= a[k];
        flag = nz ? 1 : 0;
    }
    if (carry) carry[(size_t)proof * rows + row] = flag;
}

static void fill_wconsts(std::vector<uint32_t>& c) {
    static_assert(sizeof(BLS381_WITNESS_CONST) == (size_t)WC_TOTAL * 48, "constant table and WitnessConst disagree");
    c.assign(&BLS381_WITNESS_CONST[0][0], &BLS381_WITNESS_CONST[0][0] + (size_t)WC_TOTAL * 12);
}

#define WCHK(x)                                                                  \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                \
            return -3;                                                           \
        }                                                                        \
    } while (0)

static hipError_t ensure(void*& p, size_t& have, size_t need) {
    if (need <= have) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; have = 0; }
    hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess) have = need;
    return e;
}

// Device memory to the caller's pageable memory through two pinned buffers: the DMA engine fills one (57 GB/s on
// MI355X hosts measured here, against 17.6 GB/s for a copy straight into pageable memory, which the runtime stages
// itself in small pieces) while host threads copy the other out (31 GB/s per thread; three threads keep up with the
// DMA and take the first-touch page faults of a fresh destination in parallel).  Everything enqueued on `stream`
// before the call is finished when it returns.
int WitnessDevice::copy_out_staged(hipStream_t stream, void* dst_host, const void* src_dev, size_t bytes, std::string& err) {
    const size_t STAGE = (size_t)64 << 20;
    if (bytes < STAGE / 4) {  // small: not worth the pipeline
        WCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        WCHK(hipStreamSynchronize(stream));
        return 0;
    }
    for (int k = 0; k < 2; k++) {
        if (!h_stage_[k]) WCHK(hipHostMalloc(&h_stage_[k], STAGE, hipHostMallocDefault));
        if (!stage_ev_[k]) WCHK(hipEventCreateWithFlags(&stage_ev_[k], hipEventDisableTiming));
    }
    const size_t nchunks = (bytes + STAGE - 1) / STAGE;
    auto len = [&](size_t i) { return std::min(STAGE, bytes - i * STAGE); };
    auto issue = [&](size_t i) -> hipError_t {
        hipError_t e = hipMemcpyAsync(h_stage_[i & 1], (const char*)src_dev + i * STAGE, len(i), hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipEventRecord(stage_ev_[i & 1], stream);
    };
    WCHK(issue(0));
    for (size_t i = 0; i < nchunks; i++) {
        WCHK(hipEventSynchronize(stage_ev_[i & 1]));
        if (i + 1 < nchunks) WCHK(issue(i + 1));  // (its buffer was copied out in the iteration before)
        char* const d = (char*)dst_host + i * STAGE;
        const char* const h = (const char*)h_stage_[i & 1];
        const size_t n = len(i);
        const int T = 3;
        std::thread helpers[T - 1];
        for (int t = 1; t < T; t++)
            helpers[t - 1] = std::thread([=] { std::memcpy(d + n * t / T, h + n * t / T, n * (t + 1) / T - n * t / T); });
        std::memcpy(d, h, n / T);
        for (int t = 1; t < T; t++) helpers[t - 1].join();
    }
    return 0;
}

int WitnessDevice::run(hipStream_t stream, void* ctx, inverse_fn inv, const uint64_t* g1, const uint64_t* g2, size_t n,
                       uint32_t* trace, std::string& err, bool on_device, const WitnessAux* aux) {
    const WitnessProgram& P = witness_program();
    if (!P.error.empty()) { err = P.error; return -1; }
    if (uploaded_ && prog_group_ != P.group * 2 + (P.rounds ? 1 : 0)) {  // another level assignment than the one uploaded
        WCHK(hipStreamSynchronize(stream));
        release();
    }
    if (!uploaded_) {
        release();  // an earlier upload that failed half-way leaves nothing behind
        for (int ph = 0; ph < 2; ph++) {
            WCHK(hipMalloc((void**)&d_ops_[ph], P.circuit_ops[ph].size() * sizeof(WitnessOp)));
            WCHK(hipMemcpy(d_ops_[ph], P.circuit_ops[ph].data(), P.circuit_ops[ph].size() * sizeof(WitnessOp), hipMemcpyHostToDevice));
        }
        std::vector<uint32_t> wc;
        fill_wconsts(wc);
        WCHK(hipMalloc((void**)&d_consts_, wc.size() * 4));
        WCHK(hipMemcpy(d_consts_, wc.data(), wc.size() * 4, hipMemcpyHostToDevice));
        WCHK(hipMalloc((void**)&d_inv_refs_, 48));
        WCHK(hipMemcpy(d_inv_refs_, P.inv_input_refs, 48, hipMemcpyHostToDevice));
        WCHK(hipMalloc((void**)&d_mul_index_, P.n_rows * 4));
        WCHK(hipMemcpy(d_mul_index_, P.mul_index.data(), P.n_rows * 4, hipMemcpyHostToDevice));
        uploaded_ = true;
        prog_group_ = P.group * 2 + (P.rounds ? 1 : 0);
    }
    // Sixteen lanes per proof, four proofs per one-wave workgroup.  History of the geometry (proofs/s, device form): in
    // dependency levels about 35 rows wide, wider groups mostly idled at their barriers and the kernel hid its latency
    // only by the number of proofs in flight: 256 lanes per proof x 512 proofs per pass 29 k; 64 lanes x 16,384 104 k;
    // 32 lanes 145 k; 16 lanes 167 k; 8 lanes 162 k; a Barrett product in place of two Montgomery products 190 k; an LDS
    // row cache of 48-53 rows per proof 242 k; rows held back to fill groups of 16 lanes 268 k.  In rounds without
    // barriers (the default now), with one product instead of the Barrett product, the kernel moves 9-10 MB per proof
    // through a memory system that 8,192-16,384 interleaved streams of 48-byte rows hold to about 3.5 TB/s, and what
    // pays is fewer bytes: fewer proofs in flight with more cache rows each (8,192 x 104 rows: eight workgroups of
    // 20 KB on a CU; 330-347 k against 314-340 k at 16,384 x 48, by box).
    // The staging buffer of the host-pointer form (6.8 MB per proof) keeps that form to 1,024 per pass;
    // its copy back to the host is what bounds it anyway.
    size_t CH = on_device ? (P.rounds ? 8192 : 16384) : 1024;
    int block = 64, lpp = 16;
    // tuning overrides; out-of-range values are errors (a stride of 0 would never advance the loop below)
    if (const char* e = getenv("BLS381_WIT_CH")) {
        const long v = atol(e);
        if (v < 1 || v > 32768) { err = "BLS381_WIT_CH must be in 1..32768"; return -1; }
        CH = (size_t)v;
    }
    if (const char* e = getenv("BLS381_WIT_BLOCK")) block = atoi(e);
    if (const char* e = getenv("BLS381_WIT_LPP")) lpp = atoi(e);
    // rows per proof kept in LDS, see witness_kernel and plan_row_cache.  In dependency levels (measured at 16,384
    // proofs in flight): none 190 k, 16 rows 218 k, 32 rows 234 k, 48 rows 242 k, 64 rows 205 k (sixteen one-wave
    // workgroups of four proofs per CU hold 144 KiB of LDS at 48 rows; more costs occupancy); 53 rows are the most that
    // leave a CU its sixteen workgroups.  In rounds: 104 rows with 8,192 proofs in flight (eight workgroups per CU).
    int cache_rows = P.rounds ? 88 : 37;
    if (const char* e = getenv("BLS381_WIT_CACHE")) cache_rows = atoi(e);
    if (block < 1 || block > 256 || block % 64) { err = "BLS381_WIT_BLOCK must be 64, 128, 192 or 256"; return -1; }
    if (lpp < 1 || lpp > 64 || 64 % lpp) { err = "BLS381_WIT_LPP must divide 64"; return -1; }
    if (cache_rows < 0 || cache_rows > (int)WCACHE_MAX_ROWS) { err = "BLS381_WIT_CACHE must be in 0..128"; return -1; }
    const size_t cache_bytes = (size_t)(block / lpp) * cache_rows * 48 + (size_t)(block / 64) * WOPS_RING_BYTES;
    if (cache_bytes > 64 * 1024) { err = "BLS381_WIT_CACHE: more than 64 KiB of LDS per workgroup"; return -1; }
    if (plan_lpp_ != lpp || plan_rows_ != cache_rows) {
        for (int ph = 0; ph < 2; ph++) {
            const std::vector<WitnessOp> planned = pad_rounds(P, ph, plan_row_cache(P, ph, lpp, cache_rows), lpp);
            WCHK(hipStreamSynchronize(stream));  // an earlier pass may still read the plan being replaced
            if (d_plan_[ph]) { WCHK(hipFree(d_plan_[ph])); d_plan_[ph] = nullptr; }  // (its size depends on lpp)
            WCHK(hipMalloc((void**)&d_plan_[ph], planned.size() * sizeof(WitnessOp)));
            WCHK(hipMemcpy(d_plan_[ph], planned.data(), planned.size() * sizeof(WitnessOp), hipMemcpyHostToDevice));
            n_rounds_[ph] = planned.size() / (size_t)lpp;
        }
        plan_lpp_ = lpp;
        plan_rows_ = cache_rows;
    }
    const int per_block = block / lpp;
    for (size_t off = 0; off < n; off += CH) {
        int m = (int)std::min(CH, n - off);
        size_t tbytes = (size_t)m * P.n_rows * 48;
        if (!on_device) WCHK(ensure(d_trace_, trace_bytes_, tbytes));
        u32* const d_trace_m = on_device ? trace + off * P.n_rows * 12 : (u32*)d_trace_;
        WCHK(ensure(d_inputs_, inputs_bytes_, (size_t)m * WIN_TOTAL * 48));
        const size_t n_shadow = P.n_shadow[0] + P.n_shadow[1];
        WCHK(ensure(d_shadow_, shadow_bytes_, (size_t)m * n_shadow * 48 + 48));
        const unsigned grid = (unsigned)((m + per_block - 1) / per_block);
        WCHK(ensure(d_trash_, trash_bytes_, (size_t)grid * (block / 64) * 128));  // one line per wave, see witness_kernel
        WCHK(ensure(d_x_, x_bytes_, (size_t)m * 576 * 2));
        WCHK(ensure(d_g_, g_bytes_, (size_t)m * 288));
        u32* d_g1 = (u32*)d_g_;
        u32* d_g2 = d_g1 + (size_t)m * 24;
        const hipMemcpyKind in_kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        WCHK(hipMemcpyAsync(d_g1, g1 + off * 12, (size_t)m * 96, in_kind, stream));
        WCHK(hipMemcpyAsync(d_g2, g2 + off * 24, (size_t)m * 192, in_kind, stream));
        hipLaunchKernelGGL(witness_inputs_kernel, dim3((m * WIN_TOTAL + 63) / 64), dim3(64), 0, stream, d_g1, d_g2, (const u32*)d_consts_, (u32*)d_inputs_, m);
        WCHK(hipGetLastError());
        hipLaunchKernelGGL(witness_kernel, dim3(grid), dim3(block), cache_bytes, stream, d_plan_[0], (int)n_rounds_[0],
                           d_trace_m, P.n_rows, (u32*)d_shadow_, n_shadow, d_consts_, (const u32*)d_inputs_, lpp, m, cache_rows, (u32*)d_trash_);
        WCHK(hipGetLastError());
        u32* d_x = (u32*)d_x_;
        u32* d_inv = d_x + (size_t)m * 144;
        hipLaunchKernelGGL(witness_gather_kernel, dim3((m * 12 + 63) / 64), dim3(64), 0, stream, (const u32*)d_trace_m,
                           P.n_rows, d_inv_refs_, d_x, m);
        WCHK(hipGetLastError());
        int rc = inv(ctx, d_x, (size_t)m, d_inv);
        if (rc) { err = "Fq12 inversion inside the witness trace failed (zero Miller value?)"; return rc; }
        hipLaunchKernelGGL(witness_scatter_kernel, dim3((m * 12 + 63) / 64), dim3(64), 0, stream, d_trace_m, P.n_rows,
                           P.gen_start, d_inv, m);
        WCHK(hipGetLastError());
        hipLaunchKernelGGL(witness_kernel, dim3(grid), dim3(block), cache_bytes, stream, d_plan_[1], (int)n_rounds_[1],
                           d_trace_m, P.n_rows, (u32*)d_shadow_, n_shadow, d_consts_, (const u32*)d_inputs_, lpp, m, cache_rows, (u32*)d_trash_);
        WCHK(hipGetLastError());
        if (aux && (aux->mul_quot || aux->carry || aux->inv_quot)) {
            const size_t n_mul = P.counts[WK_MUL];
            // staged on the device per pass (host pointers) or written in place (device pointers)
            u32* d_mq = nullptr; uint8_t* d_cy = nullptr; u32* d_iq = nullptr;
            if (on_device) {
                d_mq = aux->mul_quot ? aux->mul_quot + off * n_mul * 24 : nullptr;
                d_cy = aux->carry ? aux->carry + off * P.n_rows : nullptr;
                d_iq = aux->inv_quot ? aux->inv_quot + off * 24 : nullptr;
            } else {
                const size_t b_mq = aux->mul_quot ? (size_t)m * n_mul * 96 : 0, b_cy = aux->carry ? (size_t)m * P.n_rows : 0,
                             b_iq = aux->inv_quot ? (size_t)m * 96 : 0;
                WCHK(ensure(d_aux_, aux_bytes_, b_mq + b_cy + b_iq + 256));
                char* base = (char*)d_aux_;
                if (b_mq) d_mq = (u32*)base;
                if (b_iq) d_iq = (u32*)(base + b_mq);
                if (b_cy) d_cy = (uint8_t*)(base + b_mq + b_iq);
            }
            if (d_cy) WCHK(hipMemsetAsync(d_cy, 0, (size_t)m * P.n_rows, stream));  // generator rows carry nothing
            for (int ph = 0; ph < 2; ph++) {
                const long threads = (long)m * (long)P.circuit_ops[ph].size();
                hipLaunchKernelGGL(witness_aux_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_ops_[ph],
                                   (int)P.circuit_ops[ph].size(), (const u32*)d_mul_index_, (const u32*)d_trace_m, P.n_rows, d_consts_,
                                   (const u32*)d_inputs_, m, n_mul, d_mq, d_cy, d_iq);
                WCHK(hipGetLastError());
            }
            if (!on_device) {
                if (d_mq) { int rc2 = copy_out_staged(stream, aux->mul_quot + off * n_mul * 24, d_mq, (size_t)m * n_mul * 96, err); if (rc2) return rc2; }
                if (d_cy) WCHK(hipMemcpyAsync(aux->carry + off * P.n_rows, d_cy, (size_t)m * P.n_rows, hipMemcpyDeviceToHost, stream));
                if (d_iq) WCHK(hipMemcpyAsync(aux->inv_quot + off * 24, d_iq, (size_t)m * 96, hipMemcpyDeviceToHost, stream));
            }
        }
        if (!on_device) { int rc2 = copy_out_staged(stream, trace + off * P.n_rows * 12, d_trace_, tbytes, err); if (rc2) return rc2; }
        WCHK(hipStreamSynchronize(stream));
    }
    return 0;
}

void WitnessDevice::release() {
    for (int ph = 0; ph < 2; ph++) {
        for (void* p : {(void*)d_ops_[ph], (void*)d_plan_[ph]}) if (p) (void)hipFree(p);
        d_ops_[ph] = d_plan_[ph] = nullptr;
        n_rounds_[ph] = 0;
    }
    plan_lpp_ = plan_rows_ = -1;
    for (int k = 0; k < 2; k++) {
        if (h_stage_[k]) (void)hipHostFree(h_stage_[k]);
        if (stage_ev_[k]) (void)hipEventDestroy(stage_ev_[k]);
        h_stage_[k] = nullptr; stage_ev_[k] = nullptr;
    }
    for (void* p : {(void*)d_consts_, (void*)d_inv_refs_, (void*)d_mul_index_, d_trace_, d_inputs_, d_x_, d_g_, d_aux_, d_shadow_, d_trash_}) if (p) (void)hipFree(p);
    d_consts_ = nullptr; d_inv_refs_ = nullptr; d_mul_index_ = nullptr; d_trace_ = d_inputs_ = d_x_ = d_g_ = d_aux_ = d_shadow_ = d_trash_ = nullptr;
    trace_bytes_ = inputs_bytes_ = x_bytes_ = g_bytes_ = aux_bytes_ = shadow_bytes_ = trash_bytes_ = 0;
    uploaded_ = false;
}

}  // namespace bls381
