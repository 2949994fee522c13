// Symbolic Fq2 / Fq6 / Fq12 arithmetic over a Dag: every method appends FMUL / LIN
// nodes and returns handles.  Zero and one are tracked symbolically so sparse
// operands (line functions, the initial f = 1) cost nothing.
//
// Tower (ark-bls12-381): Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(1+u)), Fq12 = Fq6[w]/(w^2-v).
#pragma once
#include <cstdint>
#include <initializer_list>
#include <map>
#include <utility>
#include <vector>

#include "tape.h"

namespace bls381 {

struct SFp {
    int id;  // >= 0 node, -1 zero, -2 one
    bool is_zero() const { return id == -1; }
    bool is_one() const { return id == -2; }
};
static const SFp SZERO{-1};
static const SFp SONE{-2};

struct SFq2 { SFp c0, c1; };

// Unreduced values: a signed sum of products of Fp values, not yet multiplied out.  Sums, differences and
// multiples of xi of such values only edit the list; Tower::reduce turns a list into one n-product
// operation with a single reduction (or several and a sum when it is longer than an operation holds).
struct UTerm { int a, b, coef; };  // coef * node a * node b
struct UFp { std::vector<UTerm> t; };
struct UFq2 { UFp c0, c1; };
struct UFq6 { UFq2 c0, c1, c2; };
struct SFq6 { SFq2 c0, c1, c2; };
struct SFq12 { SFq6 c0, c1; };

class Tower {
   public:
    explicit Tower(Dag& d);
    Dag& dag() { return d_; }

    // ---- Fp
    SFp mat(SFp a);  // turns the symbolic one into a loaded constant
    SFp lin(std::initializer_list<std::pair<SFp, int>> terms);
    SFp lin(const std::vector<std::pair<SFp, int>>& terms);
    SFp add(SFp a, SFp b) { return lin({{a, 1}, {b, 1}}); }
    SFp sub(SFp a, SFp b) { return lin({{a, 1}, {b, -1}}); }
    SFp neg(SFp a) { return lin({{a, -1}}); }
    SFp mul(SFp a, SFp b);
    SFp cst(int cid) { return SFp{d_.load_const(cid)}; }
    // Boundary values are 12 x u32 with Montgomery factor 2^384 (or canonical integers for the
    // *_canon pair); inside a program values carry the factor 2^392.  One multiplication by a
    // constant changes the factor on the way in and on the way out.
    SFp in(int slot) { return SFp{d_.fmul(d_.load_in(slot), d_.load_const(C_IN))}; }
    SFp in_canon(int slot) { return SFp{d_.fmul(d_.load_in(slot), d_.load_const(C_IN_CANON))}; }
    void out_canon(SFp a, int slot);
    void out(SFp a, int slot);
    // Values handed from one program to the next keep the factor 2^392: canonical words in, no multiplication; on the
    // way out one multiplication by one brings a sum back into the range a store accepts.
    SFp in_raw(int slot) { return SFp{d_.load_in(slot)}; }
    void out_raw(SFp a, int slot);

    // ---- Fq2
    SFq2 zero2() { return {SZERO, SZERO}; }
    SFq2 one2() { return {SONE, SZERO}; }
    bool is_zero(const SFq2& a) { return a.c0.is_zero() && a.c1.is_zero(); }
    SFq2 add(const SFq2& a, const SFq2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
    SFq2 sub(const SFq2& a, const SFq2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
    SFq2 neg(const SFq2& a) { return {neg(a.c0), neg(a.c1)}; }
    SFq2 dbl(const SFq2& a) { return add(a, a); }
    SFq2 conj(const SFq2& a) { return {a.c0, neg(a.c1)}; }
    SFq2 mul_xi(const SFq2& a) { return {sub(a.c0, a.c1), add(a.c0, a.c1)}; }  // * (1+u)
    SFq2 mul(const SFq2& a, const SFq2& b);
    SFq2 sqr(const SFq2& a);
    SFq2 mul_fp(const SFq2& a, SFp s) { return {mul(a.c0, s), mul(a.c1, s)}; }
    // sum of signed Fq2 terms as one LIN per component
    SFq2 lin2(const std::vector<std::pair<SFq2, int>>& terms);

    // ---- unreduced products (see UFp)
    UFp uadd(const UFp& a, const UFp& b, int sign_b = 1);
    UFq2 uadd(const UFq2& a, const UFq2& b) { return {uadd(a.c0, b.c0), uadd(a.c1, b.c1)}; }
    UFq2 umul_xi(const UFq2& a) { return {uadd(a.c0, a.c1, -1), uadd(a.c0, a.c1)}; }  // * (1+u)
    UFq2 umul(const SFq2& a, const SFq2& b);   // the four products of the components, zero components dropped
    UFq6 umul(const SFq6& a, const SFq6& b);   // schoolbook over umul
    UFq6 uadd(const UFq6& a, const UFq6& b) { return {uadd(a.c0, b.c0), uadd(a.c1, b.c1), uadd(a.c2, b.c2)}; }
    UFq6 umul_v(const UFq6& a) { return {umul_xi(a.c2), a.c0, a.c1}; }
    SFp reduce(const UFp& u);
    SFq2 reduce(const UFq2& u) { return {reduce(u.c0), reduce(u.c1)}; }
    SFq6 reduce(const UFq6& u) { return {reduce(u.c0), reduce(u.c1), reduce(u.c2)}; }

    // ---- Fq6
    SFq6 zero6() { return {zero2(), zero2(), zero2()}; }
    SFq6 add(const SFq6& a, const SFq6& b) { return {add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
    SFq6 sub(const SFq6& a, const SFq6& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
    SFq6 neg(const SFq6& a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
    SFq6 mul_v(const SFq6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
    SFq6 mul(const SFq6& a, const SFq6& b);
    SFq6 mul_fq2(const SFq6& a, const SFq2& s) { return {mul(a.c0, s), mul(a.c1, s), mul(a.c2, s)}; }
    SFq6 mul_by_01(const SFq6& s, const SFq2& c0, const SFq2& c1);
    SFq6 mul_by_1(const SFq6& s, const SFq2& c1);

    // ---- Fq12
    SFq12 one12() { return {{one2(), zero2(), zero2()}, zero6()}; }
    SFq12 mul(const SFq12& a, const SFq12& b);
    SFq12 sqr(const SFq12& a);
    SFq12 conj(const SFq12& a) { return {a.c0, neg(a.c1)}; }
    SFq12 mul_by_014(const SFq12& f, const SFq2& c0, const SFq2& c1, const SFq2& c4);
    SFq12 cyclotomic_sqr(const SFq12& a);
    SFq12 frobenius(const SFq12& a, int power);  // any power (taken mod 12)
    SFq12 pow(const SFq12& a, const std::vector<uint64_t>& exp);  // generic squarings, any exponent
    SFq12 pow_x(const SFq12& a, bool cyclotomic);  // a^|x|, MSB-first binary

    // Fq12 inverse split around the Fp inversion.
    struct InvCarry {
        SFp norm;       // element of Fp to invert
        SFq2 m;         // Fq6 norm (an Fq2)
        SFq6 s;         // adjugate of the Fq6 norm
    };
    InvCarry inv_part1(const SFq12& a);
    SFq12 inv_part2(const SFq12& a, const InvCarry& c, SFp norm_inv);

    // ---- pairing pieces
    struct G2Proj { SFq2 x, y, z; };
    struct EllCoeff { SFq2 c0, c1, c2; };
    EllCoeff g2_double(G2Proj& r, SFp two_inv);
    // The same step with R kept at four times the reference's coordinates per doubling (no halvings): the same
    // projective point, line coefficients scaled by an element of Fq, which the final exponentiation removes.
    EllCoeff g2_double_scaled(G2Proj& r);
    EllCoeff g2_add(G2Proj& r, const SFq2& qx, const SFq2& qy);
    SFq12 ell(const SFq12& f, const EllCoeff& c, SFp px, SFp py);
    SFq12 hard_part(const SFq12& r, bool cyclotomic);

    // load / store helpers (ark flat order)
    SFq12 in12(int base);
    void out12(const SFq12& a, int base);
    SFq12 in12_raw(int base);
    void out12_raw(const SFq12& a, int base);

   private:
    Dag& d_;
    bool lazy_;  // Fq6 products as n-product operations (schoolbook, one reduction per component)
    std::map<std::vector<std::pair<int, int>>, int> shared_sums_;  // sums made for merged products, by their terms
};

}  // namespace bls381
