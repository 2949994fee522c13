// Host-side program construction for the lane-group interpreter:
//   Dag        - expression graph over Fp values (FMUL / LIN / loads / stores)
//   tower      - symbolic Fq2/Fq6/Fq12 arithmetic and the pairing algorithms on top of Dag
//   schedule() - list-schedules a Dag into K-wide steps and assigns LDS slots
// The numeric work happens on the GPU; nothing here touches field values.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "vm_format.h"

namespace bls381 {

enum NodeKind : uint8_t { NK_FMUL, NK_LIN, NK_LOADIN, NK_LOADC, NK_STOREOUT, NK_DOT2, NK_DOTN };

struct Term {
    int node;
    int coef;  // small signed integer, never zero (operands of multiplications: +1 or -1)
};

struct Node {
    NodeKind kind;
    // FMUL: op[0] * op[1]; DOT2: op[0] * op[1] + op[2] * op[3].  Every operand is a signed sum of up to
    // MUL_MAX_TERMS values: one term as built, more once optimise() has folded an operand's sum into it.
    std::vector<Term> op[4];
    int a = -1;               // STOREOUT source
    int idx = -1;             // LOADIN: gin slot, LOADC: constant id, STOREOUT: gout slot
    std::vector<Term> terms;  // LIN: sum of coef * node
    // DOTN: sum over t of pa[t] * pb[t], subtracted where pb[t].coef = -1 (pa's coefficients are 1): up to
    // DOTN_MAX_PAIRS products of plain values under one reduction
    std::vector<Term> pa, pb;
};

struct Product {
    int a, b;
    int sign;  // +1 or -1
};

class Dag {
   public:
    std::vector<Node> nodes;
    int n_in = 0, n_out = 0;

    int load_in(int slot);
    int load_const(int cid);
    int fmul(int a, int b);
    int dot2(int a, int b, int c, int d, bool neg_d = false);  // a*b + c*d, or a*b - c*d
    int dot2s(int a, int b, int sign_ab, int c, int d, int sign_cd);  // +-a*b +- c*d
    int dotn(const std::vector<Product>& products);              // sum of signed products, one reduction
    int lin(std::vector<Term> terms);
    void store_out(int node, int slot);

    // Inlines single-use sums into the sums that consume them (coefficients multiply, equal terms merge),
    // folds short sums that only feed multiplications into those multiplications' operands, and drops
    // nodes nothing depends on.
    void optimise();
};

struct Program {
    std::string name;
    int K = 0;
    int n_lds_slots = 0;   // Fp slots of LDS each group needs
    int n_tmp_slots = 0;   // Fp slots of the global spill array each group needs
    int n_in = 0, n_out = 0;
    std::vector<StepHdr> hdrs;
    std::vector<OpDesc> descs;  // hdrs.size() * K
    // statistics
    int n_fmul = 0, n_lin = 0, n_lin_terms = 0, n_copy = 0, n_dot2 = 0;
    int n_fmul_steps = 0, n_lin_steps = 0, n_copy_steps = 0, n_dot2_steps = 0;
    int n_piggy_fill = 0, n_piggy_spill = 0;  // copies riding on compute steps
    int n_reduce = 0, n_norm = 0;  // sums followed by a reduction / by a carry sweep only
    int n_wide_mul_steps = 0;      // multiplier steps with more than one term in some operand
    int n_dotn = 0, n_dotn_products = 0, n_dotn_steps = 0;  // n-product ops, their products, their steps
};

// Schedules `dag` for groups of K lanes with at most `lds_slots` LDS slots per group.
Program schedule(const Dag& dag, int K, int lds_slots, const std::string& name);

// Chooses the form of the Fq6 products for the programs this thread builds next: n-product operations for K = 6
// (unless BLS381_LAZY_PRODUCTS=0), Karatsuba over two-product operations otherwise (tower_sym.cpp).
void set_lazy_products_for(int K);

// ---- program builders (symbolic versions of the reference's native functions) ----
// gin/gout slot numbering: G1 = {x, y}; G2 = {x.c0, x.c1, y.c0, y.c1};
// Fq12 = ark tower order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1.

// multi_miller_loop_native for one pair, conjugated.      /root/reference/src/native/miller_loop.rs:121-162
// gin: G1 (2) + G2 (4); gout: Fq12 (12).
Dag build_miller();
// The same loop for callers that go on to the final exponentiation: G2 doublings without the halvings
// (Tower::g2_double_scaled).  Its value differs from the reference's Miller value by a factor in Fq.
Dag build_miller_scaled();
// G2Prepared::from: the 68 line coefficients.                /root/reference/src/native/miller_loop.rs:90-119
// gin: G2 (4); gout: 68 * 6 (coefficient k, Fq2 j, component c at 6*k + 2*j + c).
Dag build_g2_prepare();
// The Miller loop with the 68 line coefficients taken from constants C_COUNT .. C_COUNT + 408
// (a prepared G2 point shared by the whole batch).  gin: G1 (2); gout: Fq12 (12).
Dag build_miller_fixed_q();
Dag build_miller_two_pairs(bool scaled = false);  // scaled: G2 doublings without the halvings, as build_miller_scaled
// psi(Q) = [z]Q on a decoded G2 point (see tower_sym.cpp).  gin: G2 (4); gout: 4 values, all zero iff Q is in G2.
Dag build_g2_subgroup();
// (beta x, y) = [-z^2]P on a decoded G1 point.  gin: G1 (2); gout: 2 values, both zero iff P is in G1.
Dag build_g1_subgroup();
// ell(f, coeffs, p).                                         /root/reference/src/native/miller_loop.rs:185-194
// gin: f (12) + coeff (6) + G1 (2); gout: f (12).
Dag build_ell();
// Final exponentiation, split around the one Fp inversion.   /root/reference/src/native/final_exponentiation.rs:8-78
// part 1: gin f (12)           -> gout: [0] = norm in Fp, [1..] = values part 2 needs (FE_CARRY total)
// part 2: gin = part 1's gout with slot 0 replaced by 1/norm -> gout: Fq12 (12)
enum { FE_CARRY = 21 };
Dag build_final_exp_part1();
Dag build_final_exp_part2(bool cyclotomic_hard_part);
// Part 2 as six programs with the squaring-chain kernel between them (see tower_sym.cpp): slot areas of the array
// they share, and the slots that array has.
enum { FE2_AREA_A = 72, FE2_AREA_B = 84, FE2_AREA_C = 96, FE2_AREA_D = 108, FE2_AREA_E = 120, FE2_AREA_G = 132, FE2_CHAIN_SLOTS = 144 };
Dag build_final_exp_chain_part(int part);  // part 0..5
// easy_part_native alone (part 2 of it; part 1 is build_final_exp_part1).
Dag build_easy_part2();
// hard_part_native on an arbitrary Fq12 (generic squarings).  gin 12 -> gout 12.
Dag build_hard_part_generic();
// Fq12 inverse of a canonical (non-Montgomery) value in MyFq12 order, as Fq12InverseGenerator
// computes it.                                               /root/reference/src/fields/fq12_target.rs:374-397
// part 1: gin x (12, canonical, MyFq12 order) -> gout [0] = norm, [1..FE_CARRY)
// part 2: -> gout x^-1 (12, canonical, MyFq12 order)
Dag build_fq12_inv_limbs_part1();
Dag build_fq12_inv_limbs_part2();
// Fq12 product (ark order in and out): gin a (12) + b (12) -> gout 12.
Dag build_fq12_mul();
// frobenius_map_native(a, power) for any power (taken mod 12), ark order in/out.
Dag build_frobenius(int power);
// conjugate_fp12 and pow_native(a, exp) (exp: little-endian 64-bit words), ark order in/out.
Dag build_conjugate();
Dag build_pow(const std::vector<uint64_t>& exp);

}  // namespace bls381
