// Symbolic tower arithmetic and the program builders.  Formulas follow the reference where the
// VALUE of an intermediate is observable (projective G2 steps, line coefficients: the Miller-loop
// output before the final exponentiation must match arkworks bit for bit) and are free elsewhere
// (Karatsuba, cyclotomic squaring), because field elements have one canonical representation.
#include "tower_sym.h"

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include "bls381_constants.h"

namespace bls381 {

// Programs for 6 lanes per pairing build the Fq6 products and the sparse line product as schoolbook sums of
// products under one reduction per component (n-product operations) instead of Karatsuba over two-product
// operations; BLS381_LAZY_PRODUCTS=0 turns that off.  The operations come in sixes and twelves: they fill the lanes
// of a 6-lane group and leave a quarter of an 8-lane group idle.  Measured on MI355X (same box, ms per kernel):
// 61,440 pairs at 6 lanes: Miller loop 10.7 (12.8 in the Karatsuba form), final exponentiation 12.3 (12.8);
// 65,536 pairs at 8 lanes, Karatsuba form: 11.8 and 12.9 (n-product form at 8 lanes: 13.3 and 13.7).  Per pairing
// the 6-lane programs are 1 % ahead of the 8-lane default when the batch is whole rounds of the GPU at ten pairings
// per wave (multiples of 20,480) and behind it otherwise, so 8 lanes stay the default width.  Only the 6-lane
// interpreter contains the n-product path (in the others its code and registers cost 2 %).
static thread_local bool g_lazy_products = false;
void set_lazy_products_for(int K) {
    const char* e = getenv("BLS381_LAZY_PRODUCTS");
    g_lazy_products = K == 6 && !(e && atoi(e) == 0);
}
Tower::Tower(Dag& d) : d_(d), lazy_(g_lazy_products) {}

// ------------------------------------------------------------------ unreduced products
UFp Tower::uadd(const UFp& a, const UFp& b, int sign_b) {
    UFp r = a;
    for (auto t : b.t) { t.coef *= sign_b; r.t.push_back(t); }
    return r;
}

UFq2 Tower::umul(const SFq2& a, const SFq2& b) {
    UFq2 r;
    auto put = [&](UFp& dst, SFp x, SFp y, int coef) {
        if (x.is_zero() || y.is_zero()) return;
        dst.t.push_back(UTerm{mat(x).id, mat(y).id, coef});
    };
    put(r.c0, a.c0, b.c0, 1); put(r.c0, a.c1, b.c1, -1);
    put(r.c1, a.c0, b.c1, 1); put(r.c1, a.c1, b.c0, 1);
    return r;
}

UFq6 Tower::umul(const SFq6& a, const SFq6& b) {
    UFq2 p00 = umul(a.c0, b.c0), p01 = umul(a.c0, b.c1), p02 = umul(a.c0, b.c2);
    UFq2 p10 = umul(a.c1, b.c0), p11 = umul(a.c1, b.c1), p12 = umul(a.c1, b.c2);
    UFq2 p20 = umul(a.c2, b.c0), p21 = umul(a.c2, b.c1), p22 = umul(a.c2, b.c2);
    return {uadd(p00, umul_xi(uadd(p12, p21))), uadd(uadd(p01, p10), umul_xi(p22)), uadd(uadd(p02, p11), p20)};
}

// One value from a list of signed products.  Products with the same first factor are merged into one product
// with the sum of their second factors (the sums are shared between the lists that ask for the same one: the
// real and the imaginary part of xi * x * y need x * (y0 - y1) and x * (y0 + y1)), so an Fq2 product under xi
// costs two multiplications per component like any other.
SFp Tower::reduce(const UFp& u) {
    // equal products add up
    std::vector<UTerm> terms;
    for (auto& t : u.t) {
        bool found = false;
        for (auto& o : terms) if (o.a == t.a && o.b == t.b) { o.coef += t.coef; found = true; break; }
        if (!found) terms.push_back(t);
    }
    terms.erase(std::remove_if(terms.begin(), terms.end(), [](const UTerm& t) { return t.coef == 0; }), terms.end());
    if (terms.empty()) return SZERO;
    // group by first factor, in order of first appearance
    std::vector<int> firsts;
    for (auto& t : terms) if (std::find(firsts.begin(), firsts.end(), t.a) == firsts.end()) firsts.push_back(t.a);
    std::vector<Product> prods;
    for (int a : firsts) {
        std::vector<std::pair<int, int>> bs;  // (node, coef)
        for (auto& t : terms) if (t.a == a) bs.push_back({t.b, t.coef});
        if (bs.size() == 1) {
            const int k = std::abs(bs[0].second);
            if (k > 2) {  // a multiple: through a sum as well
                int id = d_.lin({Term{bs[0].first, k}});
                prods.push_back(Product{a, id, bs[0].second > 0 ? 1 : -1});
            } else {
                for (int r = 0; r < k; r++) prods.push_back(Product{a, bs[0].first, bs[0].second > 0 ? 1 : -1});
            }
            continue;
        }
        std::sort(bs.begin(), bs.end());
        int sign = 1;
        if (bs[0].second < 0) { sign = -1; for (auto& b : bs) b.second = -b.second; }
        auto it = shared_sums_.find(bs);
        int id;
        if (it != shared_sums_.end()) id = it->second;
        else {
            std::vector<Term> lt;
            for (auto& b : bs) lt.push_back(Term{b.first, b.second});
            id = d_.lin(lt);
            shared_sums_[bs] = id;
        }
        prods.push_back(Product{a, id, sign});
    }
    // products to be added first: the kernel's order is the list's
    std::stable_sort(prods.begin(), prods.end(), [](const Product& x, const Product& y) { return x.sign > y.sign; });
    auto emit = [&](const std::vector<Product>& ps) -> int {
        if (ps.size() == 1) {
            int id = d_.fmul(ps[0].a, ps[0].b);
            return ps[0].sign > 0 ? id : d_.lin({Term{id, -1}});
        }
        if (ps.size() == 2) return d_.dot2s(ps[0].a, ps[0].b, ps[0].sign, ps[1].a, ps[1].b, ps[1].sign);
        return d_.dotn(ps);
    };
    if ((int)prods.size() <= DOTN_MAX_PAIRS) return SFp{emit(prods)};
    std::vector<Term> parts;
    for (size_t i = 0; i < prods.size(); i += DOTN_MAX_PAIRS)
        parts.push_back(Term{emit(std::vector<Product>(prods.begin() + i, prods.begin() + std::min(prods.size(), i + DOTN_MAX_PAIRS))), 1});
    return SFp{d_.lin(parts)};
}

// ------------------------------------------------------------------ Fp
SFp Tower::mat(SFp a) {
    if (a.is_one()) return cst(C_ONE);
    return a;
}

SFp Tower::lin(const std::vector<std::pair<SFp, int>>& terms) {
    std::vector<Term> t;
    for (auto& p : terms) {
        if (p.first.is_zero()) continue;
        SFp m = mat(p.first);
        t.push_back(Term{m.id, p.second});
    }
    if (t.empty()) return SZERO;
    if (t.size() == 1 && t[0].coef == 1) return SFp{t[0].node};
    return SFp{d_.lin(t)};
}
SFp Tower::lin(std::initializer_list<std::pair<SFp, int>> terms) {
    return lin(std::vector<std::pair<SFp, int>>(terms));
}

SFp Tower::mul(SFp a, SFp b) {
    if (a.is_zero() || b.is_zero()) return SZERO;
    if (a.is_one()) return b;
    if (b.is_one()) return a;
    return SFp{d_.fmul(a.id, b.id)};
}

static int materialise_for_store(Dag& d, Tower& T, SFp a) {
    if (a.is_zero()) {
        // 0 = 1 - 1 keeps the store path uniform
        SFp one = T.cst(C_ONE);
        return d.lin({Term{one.id, 1}, Term{one.id, -1}});
    }
    return a.is_one() ? T.cst(C_ONE).id : a.id;
}
void Tower::out(SFp a, int slot) {
    d_.store_out(d_.fmul(materialise_for_store(d_, *this, a), d_.load_const(C_OUT)), slot);
}
void Tower::out_raw(SFp a, int slot) {
    d_.store_out(d_.fmul(materialise_for_store(d_, *this, a), d_.load_const(C_ONE)), slot);
}
void Tower::out_canon(SFp a, int slot) {
    d_.store_out(d_.fmul(materialise_for_store(d_, *this, a), d_.load_const(C_OUT_CANON)), slot);
}

// ------------------------------------------------------------------ Fq2
SFq2 Tower::lin2(const std::vector<std::pair<SFq2, int>>& terms) {
    std::vector<std::pair<SFp, int>> t0, t1;
    for (auto& p : terms) {
        t0.push_back({p.first.c0, p.second});
        t1.push_back({p.first.c1, p.second});
    }
    return {lin(t0), lin(t1)};
}

SFq2 Tower::mul(const SFq2& a, const SFq2& b) {
    bool sparse = a.c0.is_zero() || a.c1.is_zero() || b.c0.is_zero() || b.c1.is_zero() ||
                  a.c0.is_one() || b.c0.is_one();
    if (sparse) {
        SFp t00 = mul(a.c0, b.c0), t11 = mul(a.c1, b.c1);
        SFp t01 = mul(a.c0, b.c1), t10 = mul(a.c1, b.c0);
        return {sub(t00, t11), add(t01, t10)};
    }
    // two sums of two products, one reduction each: c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0
    SFp a0 = mat(a.c0), a1 = mat(a.c1), b0 = mat(b.c0), b1 = mat(b.c1);
    SFp c0{d_.dot2(a0.id, b0.id, a1.id, b1.id, /*neg_d=*/true)};
    SFp c1{d_.dot2(a0.id, b1.id, a1.id, b0.id)};
    return {c0, c1};
}

SFq2 Tower::sqr(const SFq2& a) {
    if (a.c1.is_zero()) return {mul(a.c0, a.c0), SZERO};
    if (a.c0.is_zero()) return {neg(mul(a.c1, a.c1)), SZERO};
    // two-product forms, no sums around them: c0 = a0 a0 - a1 a1, c1 = a0 a1 + a0 a1
    // (as two single products (a0 + a1)(a0 - a1) and a0 (a1 + a1) the squarings would multiply less, but single
    // products ride in two-product steps at a two-product's cost unless eight of them are ready at once, which the
    // Miller loop's twenty per iteration rarely are: 11 product-only steps of 1,315, and 500 more sums)
    SFp a0 = mat(a.c0), a1 = mat(a.c1);
    SFp c0{d_.dot2(a0.id, a0.id, a1.id, a1.id, /*neg_d=*/true)};
    SFp c1{d_.dot2(a0.id, a1.id, a0.id, a1.id)};
    return {c0, c1};
}

// ------------------------------------------------------------------ Fq6
SFq6 Tower::mul(const SFq6& a, const SFq6& b) {
    if (lazy_) return reduce(umul(a, b));
    const int zeros = is_zero(a.c0) + is_zero(a.c1) + is_zero(a.c2) + is_zero(b.c0) + is_zero(b.c1) + is_zero(b.c2);
    // (one zero component in all: Karatsuba below still saves a product, five against the schoolbook's six)
    if (zeros > 1) {
        // schoolbook with zero propagation
        SFq2 p00 = mul(a.c0, b.c0), p01 = mul(a.c0, b.c1), p02 = mul(a.c0, b.c2);
        SFq2 p10 = mul(a.c1, b.c0), p11 = mul(a.c1, b.c1), p12 = mul(a.c1, b.c2);
        SFq2 p20 = mul(a.c2, b.c0), p21 = mul(a.c2, b.c1), p22 = mul(a.c2, b.c2);
        SFq2 c0 = add(p00, mul_xi(add(p12, p21)));
        SFq2 c1 = add(add(p01, p10), mul_xi(p22));
        SFq2 c2 = add(add(p02, p11), p20);
        return {c0, c1, c2};
    }
    // Karatsuba: 6 Fq2 products
    SFq2 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
    SFq2 t12 = mul(add(a.c1, a.c2), add(b.c1, b.c2));
    SFq2 t01 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    SFq2 t02 = mul(add(a.c0, a.c2), add(b.c0, b.c2));
    SFq2 x = lin2({{t12, 1}, {v1, -1}, {v2, -1}});
    SFq2 c0 = add(v0, mul_xi(x));
    SFq2 c1 = add(lin2({{t01, 1}, {v0, -1}, {v1, -1}}), mul_xi(v2));
    SFq2 c2 = lin2({{t02, 1}, {v0, -1}, {v2, -1}, {v1, 1}});
    return {c0, c1, c2};
}

// Sparse products as restated at /root/reference/src/fields/fq6_target.rs:198-279.
SFq6 Tower::mul_by_01(const SFq6& s, const SFq2& c0, const SFq2& c1) {
    SFq2 a_a = mul(s.c0, c0);
    SFq2 b_b = mul(s.c1, c1);
    SFq2 t1 = add(mul_xi(mul(s.c2, c1)), a_a);
    SFq2 t2 = lin2({{mul(add(c0, c1), add(s.c0, s.c1)), 1}, {a_a, -1}, {b_b, -1}});
    SFq2 t3 = add(mul(s.c2, c0), b_b);
    return {t1, t2, t3};
}
SFq6 Tower::mul_by_1(const SFq6& s, const SFq2& c1) {
    // (s0 + s1 v + s2 v^2) * c1 v = xi s2 c1 + s0 c1 v + s1 c1 v^2
    return {mul_xi(mul(s.c2, c1)), mul(s.c0, c1), mul(s.c1, c1)};
}

// ------------------------------------------------------------------ Fq12
SFq12 Tower::mul(const SFq12& a, const SFq12& b) {
    SFq6 m0 = mul(a.c0, b.c0);
    SFq6 m1 = mul(a.c1, b.c1);
    SFq6 m01 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    SFq6 c1 = sub(sub(m01, m0), m1);
    SFq6 c0 = add(m0, mul_v(m1));
    return {c0, c1};
}

SFq12 Tower::sqr(const SFq12& a) {
    // complex squaring: c0' = (c0 + c1)(c0 + v c1) - ab - v ab, c1' = 2ab, ab = c0 c1
    SFq6 ab = mul(a.c0, a.c1);
    SFq6 t = mul(add(a.c0, a.c1), add(a.c0, mul_v(a.c1)));
    SFq6 c0 = sub(sub(t, ab), mul_v(ab));
    SFq6 c1 = add(ab, ab);
    return {c0, c1};
}

// /root/reference/src/fields/fq12_target.rs:258-295
SFq12 Tower::mul_by_014(const SFq12& f, const SFq2& c0, const SFq2& c1, const SFq2& c4) {
    if (lazy_) {
        // f * (l0 + l1 w) with l0 = (c0, c1, 0), l1 = (0, c4, 0): schoolbook, every component of the result is one
        // operation of at most six products
        const SFq6 l0{c0, c1, zero2()}, l1{zero2(), c4, zero2()};
        SFq6 r0 = reduce(uadd(umul(f.c0, l0), umul_v(umul(f.c1, l1))));
        SFq6 r1 = reduce(uadd(umul(f.c0, l1), umul(f.c1, l0)));
        return {r0, r1};
    }
    SFq6 aa = mul_by_01(f.c0, c0, c1);
    SFq6 bb = mul_by_1(f.c1, c4);
    SFq2 o = add(c1, c4);
    SFq6 s1 = mul_by_01(add(f.c1, f.c0), c0, o);
    s1 = sub(sub(s1, aa), bb);
    SFq6 s0 = add(mul_v(bb), aa);
    return {s0, s1};
}

// Granger-Scott squaring in the cyclotomic subgroup (three Fq4 squarings).
SFq12 Tower::cyclotomic_sqr(const SFq12& a) {
    const SFq2 &r0 = a.c0.c0, &r4 = a.c0.c1, &r3 = a.c0.c2, &r2 = a.c1.c0, &r1 = a.c1.c1, &r5 = a.c1.c2;
    auto fq4_sqr = [&](const SFq2& x, const SFq2& y, SFq2& t_even, SFq2& t_odd) {
        // (x + y s)^2 with s^2 = xi:  t_even = x^2 + xi y^2 = (x - y)(x - xi y) + (1 + xi) xy,  t_odd = 2xy.
        // The differences (rather than (x + y)(x + xi y) - (1 + xi) xy) keep the digits of the factors within
        // [-2, 2] * 2^28, so the multiplier can take them as sums of slots with no carry sweep in between.
        SFq2 xy = mul(x, y);
        SFq2 t = mul(sub(x, y), sub(x, mul_xi(y)));
        t_even = add(add(t, xy), mul_xi(xy));
        t_odd = add(xy, xy);
    };
    SFq2 t0, t1, t2, t3, t4, t5;
    fq4_sqr(r0, r1, t0, t1);
    fq4_sqr(r2, r3, t2, t3);
    fq4_sqr(r4, r5, t4, t5);
    // z0 = 3 t0 - 2 r0 ; z1 = 3 t1 + 2 r1
    SFq2 z0 = lin2({{t0, 1}, {t0, 1}, {t0, 1}, {r0, -1}, {r0, -1}});
    SFq2 z1 = lin2({{t1, 1}, {t1, 1}, {t1, 1}, {r1, 1}, {r1, 1}});
    // z2 = 3 xi t5 + 2 r2 ; z3 = 3 t4 - 2 r3
    SFq2 x5 = mul_xi(t5);
    SFq2 z2 = lin2({{x5, 1}, {x5, 1}, {x5, 1}, {r2, 1}, {r2, 1}});
    SFq2 z3 = lin2({{t4, 1}, {t4, 1}, {t4, 1}, {r3, -1}, {r3, -1}});
    // z4 = 3 t2 - 2 r4 ; z5 = 3 t3 + 2 r5
    SFq2 z4 = lin2({{t2, 1}, {t2, 1}, {t2, 1}, {r4, -1}, {r4, -1}});
    SFq2 z5 = lin2({{t3, 1}, {t3, 1}, {t3, 1}, {r5, 1}, {r5, 1}});
    return {{z0, z4, z3}, {z2, z1, z5}};
}

// Coefficient i of w^i is conjugated (odd power) and multiplied by gamma_power^i.
// /root/reference/src/native/fin_exp_helpers.rs:10-47; the constants are precomputed.
SFq12 Tower::frobenius(const SFq12& a, int power) {
    power %= 12;
    if (power != 1 && power != 2) {
        // The map is the power-th iterate of the Frobenius automorphism (gamma_k^i = xi^(i (p^k - 1) / 6) is what
        // k applications multiply coefficient i by), so any power is composed from the two whose constants
        // are tabulated; field elements are canonical, so the value is the reference's.
        SFq12 r = a;
        for (int k = 0; k < power / 2; k++) r = frobenius(r, 2);
        if (power & 1) r = frobenius(r, 1);
        return r;
    }
    // w-basis: A0 = c0.c0, A1 = c1.c0, A2 = c0.c1, A3 = c1.c1, A4 = c0.c2, A5 = c1.c2
    const SFq2* A[6] = {&a.c0.c0, &a.c1.c0, &a.c0.c1, &a.c1.c1, &a.c0.c2, &a.c1.c2};
    SFq2 B[6];
    for (int i = 0; i < 6; i++) {
        SFq2 x = *A[i];
        if (power == 1) x = conj(x);
        if (i == 0) {
            B[i] = x;
        } else if (power == 2) {
            B[i] = mul_fp(x, cst(C_FROB2_BASE + (i - 1)));
        } else {
            SFq2 g{cst(C_FROB1_BASE + 2 * (i - 1)), cst(C_FROB1_BASE + 2 * (i - 1) + 1)};
            B[i] = mul(x, g);
        }
    }
    return {{B[0], B[2], B[4]}, {B[1], B[3], B[5]}};
}

// /root/reference/src/native/fin_exp_helpers.rs:60-82: plain MSB-first square-and-multiply by the
// bits of |x| (its get_naf never emits a negative digit).
SFq12 Tower::pow_x(const SFq12& a, bool cyclotomic) {
    SFq12 res = a;
    for (int i = 62; i >= 0; i--) {
        res = cyclotomic ? cyclotomic_sqr(res) : sqr(res);
        if ((BLS381_X >> i) & 1) res = mul(res, a);
    }
    return res;
}

// pow_native for any exponent (little-endian 64-bit words, as the reference's Vec<u64>): its get_naf
// never carries and both digit signs multiply by a, so it is MSB-first binary exponentiation; an
// exponent of zero returns a itself (the loop never starts), /root/reference/src/native/fin_exp_helpers.rs:60-82.
SFq12 Tower::pow(const SFq12& a, const std::vector<uint64_t>& exp) {
    int top = -1;
    for (int i = (int)exp.size() * 64 - 1; i >= 0 && top < 0; i--)
        if ((exp[i / 64] >> (i % 64)) & 1) top = i;
    SFq12 res = a;
    for (int i = top - 1; i >= 0; i--) {
        res = sqr(res);
        if ((exp[i / 64] >> (i % 64)) & 1) res = mul(res, a);
    }
    return res;
}

// /root/reference/src/native/final_exponentiation.rs:44-72
SFq12 Tower::hard_part(const SFq12& r_in, bool cyclotomic) {
    SFq12 r = r_in;
    SFq12 y0 = cyclotomic ? cyclotomic_sqr(r) : sqr(r);
    SFq12 y1 = conj(pow_x(r, cyclotomic));
    SFq12 y2 = conj(r);
    y1 = mul(y1, y2);
    y2 = conj(pow_x(y1, cyclotomic));
    y1 = conj(y1);
    y1 = mul(y1, y2);
    y2 = conj(pow_x(y1, cyclotomic));
    y1 = frobenius(y1, 1);
    y1 = mul(y1, y2);
    r = mul(r, y0);
    y0 = conj(pow_x(y1, cyclotomic));
    y2 = conj(pow_x(y0, cyclotomic));
    y0 = frobenius(y1, 2);
    y1 = conj(y1);
    y1 = mul(y1, y2);
    y1 = mul(y1, y0);
    return mul(r, y1);
}

// a^-1 = conj-like: (c0 - c1 w) / (c0^2 - v c1^2); Fq6 and Fq2 inverses by norms.
Tower::InvCarry Tower::inv_part1(const SFq12& a) {
    SFq6 t = sub(mul(a.c0, a.c0), mul_v(mul(a.c1, a.c1)));
    SFq2 s0 = sub(sqr(t.c0), mul_xi(mul(t.c1, t.c2)));
    SFq2 s1 = sub(mul_xi(sqr(t.c2)), mul(t.c0, t.c1));
    SFq2 s2 = sub(sqr(t.c1), mul(t.c0, t.c2));
    SFq2 m = add(mul(t.c0, s0), mul_xi(add(mul(t.c2, s1), mul(t.c1, s2))));
    SFp n = add(mul(m.c0, m.c0), mul(m.c1, m.c1));
    return InvCarry{n, m, SFq6{s0, s1, s2}};
}
SFq12 Tower::inv_part2(const SFq12& a, const InvCarry& c, SFp norm_inv) {
    SFq2 minv{mul(c.m.c0, norm_inv), neg(mul(c.m.c1, norm_inv))};
    SFq6 tinv = mul_fq2(c.s, minv);
    return {mul(a.c0, tinv), neg(mul(a.c1, tinv))};
}

// ------------------------------------------------------------------ pairing pieces
// /root/reference/src/native/miller_loop.rs:50-69
Tower::EllCoeff Tower::g2_double(G2Proj& r, SFp two_inv) {
    SFq2 a = mul_fp(mul(r.x, r.y), two_inv);
    SFq2 b = sqr(r.y);
    SFq2 c = sqr(r.z);
    // e = COEFF_B * 3c with COEFF_B = 4(1+u)
    SFq2 c3 = lin2({{c, 1}, {c, 1}, {c, 1}});
    SFq2 c12 = lin2({{c3, 1}, {c3, 1}, {c3, 1}, {c3, 1}});
    SFq2 e = mul_xi(c12);
    SFq2 f = lin2({{e, 1}, {e, 1}, {e, 1}});
    SFq2 g = mul_fp(add(b, f), two_inv);
    SFq2 h = sub(sqr(add(r.y, r.z)), add(b, c));
    SFq2 i = sub(e, b);
    SFq2 j = sqr(r.x);
    SFq2 e_square = sqr(e);
    SFq2 nx = mul(a, sub(b, f));
    SFq2 ny = lin2({{sqr(g), 1}, {e_square, -1}, {e_square, -1}, {e_square, -1}});
    SFq2 nz = mul(b, h);
    r = G2Proj{nx, ny, nz};
    return EllCoeff{i, lin2({{j, 1}, {j, 1}, {j, 1}}), neg(h)};
}

// g2_double with (x, y, z) -> 4 * (x', y', z') of the reference's step: a = x y (twice the reference's), g = b + f
// (twice), so 4 x' = 2 a (b - f), 4 y' = g^2 - 12 e^2, 4 z' = 4 b h.  With R = lambda * (the reference's R), lambda
// in Fq, every line coefficient (e - b, 3 x^2, -h) is lambda^2 times the reference's.
Tower::EllCoeff Tower::g2_double_scaled(G2Proj& r) {
    SFq2 a = mul(r.x, r.y);
    SFq2 b = sqr(r.y);
    SFq2 c = sqr(r.z);
    SFq2 c12 = lin2({{c, 12}});
    SFq2 e = mul_xi(c12);
    SFq2 f = lin2({{e, 3}});
    SFq2 g = add(b, f);
    SFq2 h = sub(sqr(add(r.y, r.z)), add(b, c));
    SFq2 i = sub(e, b);
    SFq2 j = sqr(r.x);
    SFq2 e_square = sqr(e);
    SFq2 nx = lin2({{mul(a, sub(b, f)), 2}});
    SFq2 ny = lin2({{sqr(g), 1}, {e_square, -12}});
    SFq2 nz = lin2({{mul(b, h), 4}});
    r = G2Proj{nx, ny, nz};
    return EllCoeff{i, lin2({{j, 3}}), neg(h)};
}

// /root/reference/src/native/miller_loop.rs:71-87
Tower::EllCoeff Tower::g2_add(G2Proj& r, const SFq2& qx, const SFq2& qy) {
    SFq2 theta = sub(r.y, mul(qy, r.z));
    SFq2 lambda = sub(r.x, mul(qx, r.z));
    SFq2 c = sqr(theta);
    SFq2 d = sqr(lambda);
    SFq2 e = mul(lambda, d);
    SFq2 f = mul(r.z, c);
    SFq2 g = mul(r.x, d);
    SFq2 h = lin2({{e, 1}, {f, 1}, {g, -1}, {g, -1}});
    SFq2 nx = mul(lambda, h);
    SFq2 ny = sub(mul(theta, sub(g, h)), mul(e, r.y));
    SFq2 nz = mul(r.z, e);
    SFq2 j = sub(mul(theta, qx), mul(lambda, qy));
    r = G2Proj{nx, ny, nz};
    return EllCoeff{j, neg(theta), lambda};
}

// /root/reference/src/native/miller_loop.rs:185-194
SFq12 Tower::ell(const SFq12& f, const EllCoeff& c, SFp px, SFp py) {
    SFq2 c2 = mul_fp(c.c2, py);
    SFq2 c1 = mul_fp(c.c1, px);
    return mul_by_014(f, c.c0, c1, c2);
}

SFq12 Tower::in12(int base) {
    SFp v[12];
    for (int i = 0; i < 12; i++) v[i] = in(base + i);
    return {{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}}, {{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
}
void Tower::out12(const SFq12& a, int base) {
    const SFp v[12] = {a.c0.c0.c0, a.c0.c0.c1, a.c0.c1.c0, a.c0.c1.c1, a.c0.c2.c0, a.c0.c2.c1,
                       a.c1.c0.c0, a.c1.c0.c1, a.c1.c1.c0, a.c1.c1.c1, a.c1.c2.c0, a.c1.c2.c1};
    for (int i = 0; i < 12; i++) out(v[i], base + i);
}

SFq12 Tower::in12_raw(int base) {
    SFp v[12];
    for (int i = 0; i < 12; i++) v[i] = in_raw(base + i);
    return {{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}}, {{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
}
void Tower::out12_raw(const SFq12& a, int base) {
    const SFp v[12] = {a.c0.c0.c0, a.c0.c0.c1, a.c0.c1.c0, a.c0.c1.c1, a.c0.c2.c0, a.c0.c2.c1,
                       a.c1.c0.c0, a.c1.c0.c1, a.c1.c1.c0, a.c1.c1.c1, a.c1.c2.c0, a.c1.c2.c1};
    for (int i = 0; i < 12; i++) out_raw(v[i], base + i);
}

// ------------------------------------------------------------------ builders
// The Miller loop walks the bits of |x| below the top one, MSB first
// (BitIteratorBE::without_leading_zeros([BLS_X]).skip(1)): 63 doublings, 5 additions.
static void miller_body(Tower& T, SFq12& f, bool emit_coeffs, int q_base, int p_base, bool scaled = false) {
    Dag& d = T.dag();
    (void)d;
    // each input is converted once (a multiplication) and then lives like any other value
    const SFq2 qx{T.in(q_base + 0), T.in(q_base + 1)}, qy{T.in(q_base + 2), T.in(q_base + 3)};
    SFp px = SZERO, py = SZERO;
    if (!emit_coeffs) { px = T.in(p_base + 0); py = T.in(p_base + 1); }
    Tower::G2Proj r{qx, qy, T.one2()};
    int k = 0;
    for (int i = 62; i >= 0; i--) {
        Tower::EllCoeff c = scaled ? T.g2_double_scaled(r) : T.g2_double(r, T.cst(C_TWO_INV));
        if (emit_coeffs) {
            T.out(c.c0.c0, 6 * k + 0); T.out(c.c0.c1, 6 * k + 1);
            T.out(c.c1.c0, 6 * k + 2); T.out(c.c1.c1, 6 * k + 3);
            T.out(c.c2.c0, 6 * k + 4); T.out(c.c2.c1, 6 * k + 5);
        } else {
            f = T.sqr(f);
            f = T.ell(f, c, px, py);
        }
        k++;
        if ((BLS381_X >> i) & 1) {
            c = T.g2_add(r, qx, qy);
            if (emit_coeffs) {
                T.out(c.c0.c0, 6 * k + 0); T.out(c.c0.c1, 6 * k + 1);
                T.out(c.c1.c0, 6 * k + 2); T.out(c.c1.c1, 6 * k + 3);
                T.out(c.c2.c0, 6 * k + 4); T.out(c.c2.c1, 6 * k + 5);
            } else {
                f = T.ell(f, c, px, py);
            }
            k++;
        }
    }
    assert(k == 68);
}

Dag build_miller() {
    Dag d;
    d.n_in = 6; d.n_out = 12;
    Tower T(d);
    SFq12 f = T.one12();
    miller_body(T, f, false, /*q_base=*/2, /*p_base=*/0);
    f = T.conj(f);  // BLS_X_IS_NEGATIVE, /root/reference/src/native/miller_loop.rs:157-159
    T.out12(f, 0);
    d.optimise();
    return d;
}

Dag build_miller_scaled() {
    Dag d;
    d.n_in = 6; d.n_out = 12;
    Tower T(d);
    SFq12 f = T.one12();
    miller_body(T, f, false, /*q_base=*/2, /*p_base=*/0, /*scaled=*/true);
    f = T.conj(f);
    T.out12(f, 0);
    d.optimise();
    return d;
}

// Two pairs sharing one accumulator, as multi_miller_loop_native runs a chunk of pairs
// (/root/reference/src/native/miller_loop.rs:139-156): f is squared once per bit, then multiplied by
// each pair's line.  Equal, as a field element, to the product of the two single-pair values.
// gin: P1 (2), Q1 (4), P2 (2), Q2 (4).
Dag build_miller_two_pairs(bool scaled) {
    Dag d;
    d.n_in = 12; d.n_out = 12;
    Tower T(d);
    SFq12 f = T.one12();
    SFp px[2], py[2];
    SFq2 qx[2], qy[2];
    Tower::G2Proj r[2];
    for (int j = 0; j < 2; j++) {
        const int b = 6 * j;
        px[j] = T.in(b); py[j] = T.in(b + 1);
        qx[j] = SFq2{T.in(b + 2), T.in(b + 3)}; qy[j] = SFq2{T.in(b + 4), T.in(b + 5)};
        r[j] = Tower::G2Proj{qx[j], qy[j], T.one2()};
    }
    // The two lines of a step are multiplied with each other first (two elements with three of six coefficients: six
    // Fq2 products by Karatsuba) and f by their product (five of six coefficients: 17), 23 Fq2 products where two sparse
    // products of f cost 26.  The value of f is the same field element either way.
    auto both = [&](const Tower::EllCoeff& a, const Tower::EllCoeff& b) {
        const SFq2 c0 = a.c0, c1 = T.mul_fp(a.c1, px[0]), c4 = T.mul_fp(a.c2, py[0]);
        const SFq2 d0 = b.c0, d1 = T.mul_fp(b.c1, px[1]), d4 = T.mul_fp(b.c2, py[1]);
        const SFq2 p00 = T.mul(c0, d0), p11 = T.mul(c1, d1), p44 = T.mul(c4, d4);
        const SFq2 p01 = T.mul(T.add(c0, c1), T.add(d0, d1)), p04 = T.mul(T.add(c0, c4), T.add(d0, d4));
        const SFq2 p14 = T.mul(T.add(c1, c4), T.add(d1, d4));
        // (c0 + c1 v + c4 v w)(d0 + d1 v + d4 v w), w^2 = v, v^3 = xi
        const SFq12 lines{{T.add(p00, T.mul_xi(p44)), T.lin2({{p01, 1}, {p00, -1}, {p11, -1}}), p11},
                          {T.zero2(), T.lin2({{p04, 1}, {p00, -1}, {p44, -1}}), T.lin2({{p14, 1}, {p11, -1}, {p44, -1}})}};
        f = T.mul(f, lines);
    };
    for (int i = 62; i >= 0; i--) {
        f = T.sqr(f);
        {
            const Tower::EllCoeff a = scaled ? T.g2_double_scaled(r[0]) : T.g2_double(r[0], T.cst(C_TWO_INV));
            const Tower::EllCoeff b = scaled ? T.g2_double_scaled(r[1]) : T.g2_double(r[1], T.cst(C_TWO_INV));
            both(a, b);
        }
        if ((BLS381_X >> i) & 1) {
            const Tower::EllCoeff a = T.g2_add(r[0], qx[0], qy[0]), b = T.g2_add(r[1], qx[1], qy[1]);
            both(a, b);
        }
    }
    f = T.conj(f);
    T.out12(f, 0);
    d.optimise();
    return d;
}

// Miller loop against line coefficients prepared once (G2Prepared::ell_coeffs reused for many P,
// /root/reference/src/native/miller_loop.rs:13-16, 139-156): the 68 coefficients are constants
// C_COUNT + 6 k + {c0.c0, c0.c1, c1.c0, c1.c1, c2.c0, c2.c1}, only f is computed.  gin: G1 (2).
Dag build_miller_fixed_q() {
    Dag d;
    d.n_in = 2; d.n_out = 12;
    Tower T(d);
    SFq12 f = T.one12();
    const SFp px = T.in(0), py = T.in(1);
    int k = 0;
    auto coeff = [&]() {
        const int b = C_COUNT + 6 * k++;
        return Tower::EllCoeff{{T.cst(b), T.cst(b + 1)}, {T.cst(b + 2), T.cst(b + 3)}, {T.cst(b + 4), T.cst(b + 5)}};
    };
    for (int i = 62; i >= 0; i--) {
        f = T.sqr(f);
        f = T.ell(f, coeff(), px, py);
        if ((BLS381_X >> i) & 1) f = T.ell(f, coeff(), px, py);
    }
    assert(k == 68);
    f = T.conj(f);
    T.out12(f, 0);
    d.optimise();
    return d;
}

Dag build_g2_prepare() {
    Dag d;
    d.n_in = 4; d.n_out = 68 * 6;
    Tower T(d);
    SFq12 f = T.one12();
    miller_body(T, f, true, 0, 0);
    d.optimise();
    return d;
}

// Membership of a point of the twist in the order-r subgroup: psi(Q) = [z]Q with z = -|x| (Scott, "A note on group
// membership tests for G1, G2 and GT on BLS pairing-friendly curves"; the test ark-bls12-381 0.4.0 runs on a decoded
// G2 point, /root/reference/Cargo.toml:13).  R = [|x|]Q by complete projective doublings and additions (Renes,
// Costello, Batina 2016, algorithms 7 and 9 for a = 0; the twist has odd order, so they have no exceptional inputs),
// psi(x, y) = (conj(x) / xi^((p-1)/3), conj(y) / xi^((p-1)/2)).  psi(Q) = -R in projective coordinates, cleared of the
// two divisions:  conj(x) Z = X xi^((p-1)/3)  and  conj(y) Z = -Y xi^((p-1)/2).
// gin: Q affine (4); gout: the two differences (4): all zero iff Q is in the subgroup (Q itself not infinity).
Dag build_g2_subgroup() {
    Dag d;
    d.n_in = 4; d.n_out = 4;
    Tower T(d);
    const SFq2 qx{T.in(0), T.in(1)}, qy{T.in(2), T.in(3)};
    auto b3 = [&](const SFq2& a) {  // 3 b a with b = 4 (1 + u)  (12 a first: a sum's digits must stay within 32 bits)
        return T.mul_xi(T.lin2({{a, 12}}));
    };
    struct Pt { SFq2 X, Y, Z; };
    // (with s = 3 xi Z^2 = t2 / 4 of the paper's algorithm: the sums between a doubling's two levels of products are two
    // deep instead of three, and every coefficient sum stays within the digits' 32 bits)
    auto dbl = [&](const Pt& p) {
        SFq2 t0 = T.sqr(p.Y);
        SFq2 z8 = T.lin2({{t0, 8}});
        SFq2 t1 = T.mul(p.Y, p.Z);
        SFq2 zz = T.sqr(p.Z);
        SFq2 s{T.lin({{zz.c0, 3}, {zz.c1, -3}}), T.lin({{zz.c0, 3}, {zz.c1, 3}})};
        SFq2 xy = T.mul(p.X, p.Y);
        SFq2 x3 = T.mul(T.lin2({{s, 4}}), z8);
        SFq2 z3 = T.mul(t1, z8);
        SFq2 t0m = T.lin2({{t0, 1}, {s, -12}});
        SFq2 y3 = T.add(x3, T.mul(t0m, T.lin2({{t0, 1}, {s, 4}})));
        SFq2 xo = T.lin2({{T.mul(t0m, xy), 2}});
        return Pt{xo, y3, z3};
    };
    auto add_q = [&](const Pt& p) {  // p + (qx, qy, 1)
        SFq2 t0 = T.mul(p.X, qx), t1 = T.mul(p.Y, qy), t2 = p.Z;
        SFq2 t3 = T.sub(T.mul(T.add(p.X, p.Y), T.add(qx, qy)), T.add(t0, t1));
        SFq2 t4 = T.add(T.mul(qy, p.Z), p.Y);   // (Y1 + Z1)(qy + 1) - t1 - t2
        SFq2 y3 = T.add(T.mul(qx, p.Z), p.X);   // (X1 + Z1)(qx + 1) - t0 - t2
        SFq2 x3 = T.lin2({{t0, 3}});
        t2 = b3(t2);
        SFq2 z3 = T.add(t1, t2);
        t1 = T.sub(t1, t2);
        y3 = b3(y3);
        SFq2 xo = T.sub(T.mul(t3, t1), T.mul(t4, y3));
        SFq2 yo = T.add(T.mul(t1, z3), T.mul(y3, x3));
        SFq2 zo = T.add(T.mul(z3, t4), T.mul(x3, t3));
        return Pt{xo, yo, zo};
    };
    Pt r{qx, qy, T.one2()};
    for (int i = 62; i >= 0; i--) {
        r = dbl(r);
        if ((0xd201000000010000ull >> i) & 1ull) r = add_q(r);
    }
    const SFq2 g2{T.cst(C_FROB1_BASE + 2), T.cst(C_FROB1_BASE + 3)}, g3{T.cst(C_FROB1_BASE + 4), T.cst(C_FROB1_BASE + 5)};
    const SFq2 e1 = T.sub(T.mul(T.conj(qx), r.Z), T.mul(r.X, g2));
    const SFq2 e2 = T.add(T.mul(T.conj(qy), r.Z), T.mul(r.Y, g3));
    T.out(e1.c0, 0); T.out(e1.c1, 1); T.out(e2.c0, 2); T.out(e2.c1, 3);
    d.optimise();
    return d;
}

// The same for a decoded G1 point: (beta x, y) = [-z^2]P with beta the cube root of unity xi^((p^2-1)/3) (the test
// of the same note and library).  [z^2]P as [|x|]([|x|]P), the second pass adding a projective point; in projective
// coordinates beta x Z = X and y Z = -Y.  gin: P affine (2); gout: the two differences (2).
Dag build_g1_subgroup() {
    Dag d;
    d.n_in = 2; d.n_out = 2;
    Tower T(d);
    const SFp px = T.in(0), py = T.in(1);
    struct Pt { SFp X, Y, Z; };
    auto dbl = [&](const Pt& p) {  // (3 b = 12; s = 3 Z^2 as above)
        SFp t0 = T.mul(p.Y, p.Y);
        SFp z8 = T.lin({{t0, 8}});
        SFp t1 = T.mul(p.Y, p.Z);
        SFp s = T.lin({{T.mul(p.Z, p.Z), 3}});
        SFp xy = T.mul(p.X, p.Y);
        SFp z3 = T.mul(t1, z8);
        SFp t0m = T.lin({{t0, 1}, {s, -12}});
        SFp y3{T.dag().dot2(T.mat(T.lin({{s, 4}})).id, T.mat(z8).id, T.mat(t0m).id, T.mat(T.lin({{t0, 1}, {s, 4}})).id)};
        SFp xo = T.lin({{T.mul(t0m, xy), 2}});
        return Pt{xo, y3, z3};
    };
    auto add = [&](const Pt& p, const Pt& q) {
        SFp t0 = T.mul(p.X, q.X), t1 = T.mul(p.Y, q.Y), t2 = T.mul(p.Z, q.Z);
        SFp t3 = T.sub(T.mul(T.add(p.X, p.Y), T.add(q.X, q.Y)), T.add(t0, t1));
        SFp t4 = T.sub(T.mul(T.add(p.Y, p.Z), T.add(q.Y, q.Z)), T.add(t1, t2));
        SFp y3 = T.sub(T.mul(T.add(p.X, p.Z), T.add(q.X, q.Z)), T.add(t0, t2));
        SFp x3 = T.lin({{t0, 3}});
        t2 = T.lin({{t2, 12}});
        SFp z3 = T.add(t1, t2);
        t1 = T.sub(t1, t2);
        y3 = T.lin({{y3, 12}});
        SFp xo{T.dag().dot2(T.mat(t3).id, T.mat(t1).id, T.mat(t4).id, T.mat(y3).id, /*neg_d=*/true)};
        SFp yo{T.dag().dot2(T.mat(t1).id, T.mat(z3).id, T.mat(y3).id, T.mat(x3).id)};
        SFp zo{T.dag().dot2(T.mat(z3).id, T.mat(t4).id, T.mat(x3).id, T.mat(t3).id)};
        return Pt{xo, yo, zo};
    };
    auto mul_x = [&](const Pt& base) {
        Pt r = base;
        for (int i = 62; i >= 0; i--) {
            r = dbl(r);
            if ((0xd201000000010000ull >> i) & 1ull) r = add(r, base);
        }
        return r;
    };
    const Pt r = mul_x(mul_x(Pt{px, py, SONE}));
    const SFp beta = T.cst(C_FROB2_BASE + 1);
    const SFp e1 = T.sub(T.mul(T.mul(beta, px), r.Z), r.X);
    const SFp e2{T.dag().dot2(T.mat(py).id, T.mat(r.Z).id, T.mat(r.Y).id, T.cst(C_ONE).id)};
    T.out(e1, 0); T.out(e2, 1);
    d.optimise();
    return d;
}

Dag build_ell() {
    Dag d;
    d.n_in = 20; d.n_out = 12;
    Tower T(d);
    SFq12 f = T.in12(0);
    Tower::EllCoeff c{{T.in(12), T.in(13)}, {T.in(14), T.in(15)}, {T.in(16), T.in(17)}};
    f = T.ell(f, c, T.in(18), T.in(19));
    T.out12(f, 0);
    d.optimise();
    return d;
}

// Carry layout between the two halves: [0] norm / its inverse, [1..13) the input a,
// [13..15) m, [15..21) s.
static void store_carry(Tower& T, const SFq12& a, const Tower::InvCarry& c) {
    T.out(c.norm, 0);
    T.out12(a, 1);
    T.out(c.m.c0, 13); T.out(c.m.c1, 14);
    T.out(c.s.c0.c0, 15); T.out(c.s.c0.c1, 16);
    T.out(c.s.c1.c0, 17); T.out(c.s.c1.c1, 18);
    T.out(c.s.c2.c0, 19); T.out(c.s.c2.c1, 20);
}
static SFq12 load_carry_inverse(Tower& T, SFq12& a) {
    SFp ninv = T.in(0);
    a = T.in12(1);
    Tower::InvCarry c;
    c.norm = SZERO;
    c.m = SFq2{T.in(13), T.in(14)};
    c.s = SFq6{{T.in(15), T.in(16)}, {T.in(17), T.in(18)}, {T.in(19), T.in(20)}};
    return T.inv_part2(a, c, ninv);
}

Dag build_final_exp_part1() {
    Dag d;
    d.n_in = 12; d.n_out = FE_CARRY;
    Tower T(d);
    SFq12 a = T.in12(0);
    Tower::InvCarry c = T.inv_part1(a);
    store_carry(T, a, c);
    d.optimise();
    return d;
}

// /root/reference/src/native/final_exponentiation.rs:8-19 then :44-72
static SFq12 easy_part_tail(Tower& T) {
    SFq12 a;
    SFq12 ainv = load_carry_inverse(T, a);
    SFq12 f2 = T.mul(T.conj(a), ainv);
    SFq12 f3 = T.frobenius(f2, 2);
    return T.mul(f3, f2);
}

Dag build_final_exp_part2(bool cyclotomic_hard_part) {
    Dag d;
    d.n_in = FE_CARRY; d.n_out = 12;
    Tower T(d);
    SFq12 f = easy_part_tail(T);
    f = T.hard_part(f, cyclotomic_hard_part);
    T.out12(f, 0);
    d.optimise();
    return d;
}

// The final exponentiation's second part in six programs around five runs of the squaring-chain kernel
// (cyc_chain.h).  hard_part_native (/root/reference/src/native/final_exponentiation.rs:44-72) raises to |x| five times;
// each of those is one chain run (63 squarings that keep a^(2^k) for the six set bits k of |x|) followed by the
// product of the six kept values, which the next program forms: the same 63 squarings and 5 multiplications as
// pow_native (/root/reference/src/native/fin_exp_helpers.rs:60-82), in another order, and the same field element.
// All programs but the first read one slot array F and write into it (canonical words, factor 2^392):
//   F[0..72) the six kept values of the latest chain run, F[72..) twelve-slot areas A, B, C, D, E, G (FE2_AREA_*).
static SFq12 chain_product(Tower& T) {
    SFq12 s[6];
    for (int k = 0; k < 6; k++) s[k] = T.in12_raw(12 * k);
    // one after the other: as a tree ((s0 s1)(s2 s3))(s4 s5) the independent products interleave, more values are alive
    // than a group's LDS slots hold and the programs spill more (six programs 3.9 ms against 3.6 per 65,536; at 16 lanes
    // per item, with twice the slots, 4.0)
    SFq12 acc = T.mul(s[0], s[1]);
    for (int k = 2; k < 6; k++) acc = T.mul(acc, s[k]);
    return acc;
}
Dag build_final_exp_chain_part(int part) {
    Dag d;
    Tower T(d);
    switch (part) {
        case 0: {  // carry of part 1 -> r = easy part (A), y0 = r^2 (B); the first chain runs on A
            d.n_in = FE_CARRY; d.n_out = 24;
            SFq12 r = easy_part_tail(T);
            T.out12_raw(r, 0);
            T.out12_raw(T.cyclotomic_sqr(r), 12);
            break;
        }
        case 1: {  // y1 = conj(r^|x|) conj(r) -> C; the second chain runs on C        (:46-48)
            d.n_in = FE2_AREA_A + 12; d.n_out = 12;
            SFq12 r = T.in12_raw(FE2_AREA_A);
            T.out12_raw(T.mul(T.conj(chain_product(T)), T.conj(r)), 0);
            break;
        }
        case 2: {  // y1 = conj(y1) conj(y1^|x|) -> D; the third chain runs on D       (:49-52)
            d.n_in = FE2_AREA_C + 12; d.n_out = 12;
            SFq12 y1 = T.in12_raw(FE2_AREA_C);
            T.out12_raw(T.mul(T.conj(y1), T.conj(chain_product(T))), 0);
            break;
        }
        case 3: {  // y1 = frobenius(y1) conj(y1^|x|) -> E, r = r y0 -> G; the fourth chain runs on E   (:53-57)
            d.n_in = FE2_AREA_D + 12; d.n_out = 24;
            SFq12 r = T.in12_raw(FE2_AREA_A), y0 = T.in12_raw(FE2_AREA_B), y1 = T.in12_raw(FE2_AREA_D);
            T.out12_raw(T.mul(T.frobenius(y1, 1), T.conj(chain_product(T))), 0);
            T.out12_raw(T.mul(r, y0), 12);
            break;
        }
        case 4: {  // y0 = conj(y1^|x|) -> C; the fifth chain runs on C                 (:58-59)
            d.n_in = 72; d.n_out = 12;
            T.out12_raw(T.conj(chain_product(T)), 0);
            break;
        }
        default: {  // y2 = conj(y0^|x|); result = r (conj(y1) y2 frobenius^2(y1))      (:60-71), boundary form
            d.n_in = FE2_AREA_G + 12; d.n_out = 12;
            SFq12 y1 = T.in12_raw(FE2_AREA_E), r = T.in12_raw(FE2_AREA_G);
            SFq12 y2 = T.conj(chain_product(T));
            SFq12 y0 = T.frobenius(y1, 2);
            SFq12 m = T.mul(T.mul(T.conj(y1), y2), y0);
            T.out12(T.mul(r, m), 0);
            break;
        }
    }
    d.optimise();
    return d;
}

Dag build_easy_part2() {
    Dag d;
    d.n_in = FE_CARRY; d.n_out = 12;
    Tower T(d);
    T.out12(easy_part_tail(T), 0);
    d.optimise();
    return d;
}

Dag build_hard_part_generic() {
    Dag d;
    d.n_in = 12; d.n_out = 12;
    Tower T(d);
    T.out12(T.hard_part(T.in12(0), false), 0);
    d.optimise();
    return d;
}

// MyFq12 index k <-> ark flat index (/root/reference/src/utils/helpers.rs:39-41).
static const int MY_TO_ARK[12] = {0, 6, 2, 8, 4, 10, 1, 7, 3, 9, 5, 11};

Dag build_fq12_inv_limbs_part1() {
    Dag d;
    d.n_in = 12; d.n_out = FE_CARRY;
    Tower T(d);
    SFp v[12];
    for (int k = 0; k < 12; k++) v[MY_TO_ARK[k]] = T.in_canon(k);
    SFq12 a{{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}}, {{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
    Tower::InvCarry c = T.inv_part1(a);
    store_carry(T, a, c);
    d.optimise();
    return d;
}
Dag build_fq12_inv_limbs_part2() {
    Dag d;
    d.n_in = FE_CARRY; d.n_out = 12;
    Tower T(d);
    SFq12 a;
    SFq12 r = load_carry_inverse(T, a);
    const SFp v[12] = {r.c0.c0.c0, r.c0.c0.c1, r.c0.c1.c0, r.c0.c1.c1, r.c0.c2.c0, r.c0.c2.c1,
                       r.c1.c0.c0, r.c1.c0.c1, r.c1.c1.c0, r.c1.c1.c1, r.c1.c2.c0, r.c1.c2.c1};
    for (int k = 0; k < 12; k++) T.out_canon(v[MY_TO_ARK[k]], k);
    d.optimise();
    return d;
}

Dag build_fq12_mul() {
    Dag d;
    d.n_in = 24; d.n_out = 12;
    Tower T(d);
    T.out12(T.mul(T.in12(0), T.in12(12)), 0);
    d.optimise();
    return d;
}

// conjugate_fp12 (/root/reference/src/native/fin_exp_helpers.rs:132-142): the odd powers of w change sign,
// i.e. c1 -> -c1 in the tower; ark order in and out.
Dag build_conjugate() {
    Dag d;
    d.n_in = 12; d.n_out = 12;
    Tower T(d);
    T.out12(T.conj(T.in12(0)), 0);
    d.optimise();
    return d;
}

Dag build_pow(const std::vector<uint64_t>& exp) {
    Dag d;
    d.n_in = 12; d.n_out = 12;
    Tower T(d);
    T.out12(T.pow(T.in12(0), exp), 0);
    d.optimise();
    return d;
}

Dag build_frobenius(int power) {
    Dag d;
    d.n_in = 12; d.n_out = 12;
    Tower T(d);
    T.out12(T.frobenius(T.in12(0), power), 0);
    d.optimise();
    return d;
}

}  // namespace bls381
