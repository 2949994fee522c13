// Dag bookkeeping, the LIN-inlining pass, and the K-wide list scheduler with LDS slot
// assignment (furthest-next-use eviction to the global spill array).
#include "tape.h"

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <stdexcept>

namespace bls381 {

static int env_int(const char* nm, int dflt) { const char* v = getenv(nm); return v ? atoi(v) : dflt; }

int Dag::load_in(int slot) {
    Node n; n.kind = NK_LOADIN; n.idx = slot;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::load_const(int cid) {
    Node n; n.kind = NK_LOADC; n.idx = cid;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::fmul(int a, int b) {
    assert(a >= 0 && b >= 0);
    Node n; n.kind = NK_FMUL; n.op[0] = {Term{a, 1}}; n.op[1] = {Term{b, 1}};
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::dot2(int a, int b, int c, int d, bool neg_d) {
    assert(a >= 0 && b >= 0 && c >= 0 && d >= 0);
    Node n; n.kind = NK_DOT2;
    n.op[0] = {Term{a, 1}}; n.op[1] = {Term{b, 1}}; n.op[2] = {Term{c, 1}}; n.op[3] = {Term{d, neg_d ? -1 : 1}};
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::dot2s(int a, int b, int sign_ab, int c, int d, int sign_cd) {
    assert(a >= 0 && b >= 0 && c >= 0 && d >= 0);
    Node n; n.kind = NK_DOT2;
    n.op[0] = {Term{a, 1}}; n.op[1] = {Term{b, sign_ab < 0 ? -1 : 1}}; n.op[2] = {Term{c, 1}}; n.op[3] = {Term{d, sign_cd < 0 ? -1 : 1}};
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::dotn(const std::vector<Product>& products) {
    assert(!products.empty() && (int)products.size() <= DOTN_MAX_PAIRS);
    Node n; n.kind = NK_DOTN;
    for (auto& pr : products) {
        assert(pr.a >= 0 && pr.b >= 0 && (pr.sign == 1 || pr.sign == -1));
        n.pa.push_back(Term{pr.a, 1});
        n.pb.push_back(Term{pr.b, pr.sign});
    }
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
// equal nodes merge into one term, terms that cancel vanish
static std::vector<Term> merge_terms(const std::vector<Term>& in) {
    std::vector<Term> out;
    for (auto& t : in) {
        bool found = false;
        for (auto& o : out) if (o.node == t.node) { o.coef += t.coef; found = true; break; }
        if (!found) out.push_back(t);
    }
    std::vector<Term> nz;
    for (auto& o : out) if (o.coef != 0) nz.push_back(o);
    return nz;
}
// Digits are signed 32-bit numbers with some room kept for the reduction's arithmetic: a sum of normalised values
// (digits in [0, 2^28)) is fine while its positive and its negative coefficients each add up to at most 7, or,
// computed around a shifted zero (see the scheduler's bound analysis), while both together add up to at most 14.
static bool coefficients_fit(const std::vector<Term>& terms) {
    int pos = 0, neg = 0;
    for (auto& t : terms) {
        if (std::abs(t.coef) > LIN_MAX_COEF) return false;
        (t.coef > 0 ? pos : neg) += std::abs(t.coef);
    }
    static const bool use_bias = env_int("BLS381_LIN_BIAS", 1) != 0;
    return use_bias ? pos + neg <= 14 : (pos <= 7 && neg <= 7);
}
int Dag::lin(std::vector<Term> terms) {
    assert(!terms.empty());
    std::vector<Term> merged = merge_terms(terms);
    // (x - x stays as written: a zero value is still a value; so does a sum whose merged coefficients would not
    // fit a digit: the scheduler then sweeps its sources)
    if (!merged.empty() && coefficients_fit(merged)) terms = merged;
    Node n; n.kind = NK_LIN; n.terms = std::move(terms);
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
void Dag::store_out(int node, int slot) {
    assert(node >= 0);
    Node n; n.kind = NK_STOREOUT; n.a = node; n.idx = slot;
    nodes.push_back(n);
}

static inline int n_operands(const Node& n) { return n.kind == NK_DOT2 ? 4 : (n.kind == NK_FMUL ? 2 : 0); }
static inline bool is_mul(const Node& n) { return n.kind == NK_FMUL || n.kind == NK_DOT2; }
// anything that ends in a Montgomery reduction: its result has normalised digits and a value bound of its own
static inline bool is_product(const Node& n) { return is_mul(n) || n.kind == NK_DOTN; }

static void for_each_src(const Node& n, const std::function<void(int)>& f) {
    switch (n.kind) {
        case NK_FMUL:
        case NK_DOT2:
            for (int j = 0; j < n_operands(n); j++) for (auto& t : n.op[j]) f(t.node);
            break;
        case NK_LIN: for (auto& t : n.terms) f(t.node); break;
        case NK_DOTN:
            for (auto& t : n.pa) f(t.node);
            for (auto& t : n.pb) f(t.node);
            break;
        case NK_STOREOUT: f(n.a); break;
        default: break;
    }
}

static void drop_dead_nodes(std::vector<Node>& nodes) {
    const int n = (int)nodes.size();
    std::vector<char> live(n, 0);
    for (int i = n - 1; i >= 0; i--) {
        if (nodes[i].kind == NK_STOREOUT) live[i] = 1;
        if (live[i]) for_each_src(nodes[i], [&](int s) { live[s] = 1; });
    }
    std::vector<int> remap(n, -1);
    std::vector<Node> out;
    for (int i = 0; i < n; i++) {
        if (!live[i]) continue;
        Node nd = nodes[i];
        for (int j = 0; j < 4; j++) for (auto& t : nd.op[j]) t.node = remap[t.node];
        for (auto& t : nd.terms) t.node = remap[t.node];
        for (auto& t : nd.pa) t.node = remap[t.node];
        for (auto& t : nd.pb) t.node = remap[t.node];
        if (nd.kind == NK_STOREOUT) nd.a = remap[nd.a];
        remap[i] = (int)out.size();
        out.push_back(nd);
    }
    nodes.swap(out);
}

void Dag::optimise() {
    // ---- 1. single-use sums move into the sums that consume them
    {
        const int n = (int)nodes.size();
        std::vector<int> uses(n, 0);
        for (auto& nd : nodes) for_each_src(nd, [&](int s) { uses[s]++; });
        for (int i = 0; i < n; i++) {
            Node& nd = nodes[i];
            if (nd.kind != NK_LIN) continue;
            bool changed = true;
            while (changed) {
                changed = false;
                for (size_t k = 0; k < nd.terms.size(); k++) {
                    const Term t = nd.terms[k];
                    const Node& in = nodes[t.node];
                    if (in.kind != NK_LIN || uses[t.node] != 1) continue;
                    std::vector<Term> nt;
                    for (size_t j = 0; j < nd.terms.size(); j++) {
                        if (j != k) { nt.push_back(nd.terms[j]); continue; }
                        for (auto& it : in.terms) nt.push_back(Term{it.node, it.coef * t.coef});
                    }
                    nt = merge_terms(nt);
                    if (nt.empty() || nt.size() > (size_t)LIN_MAX_TERMS || !coefficients_fit(nt)) continue;
                    uses[t.node] = 0;
                    nd.terms = nt;
                    changed = true;
                    break;
                }
            }
        }
        drop_dead_nodes(nodes);
    }
    // ---- 2. short sums that only feed multiplications become part of those multiplications' operands:
    // the multiplier adds them up in registers as it loads them.  A sum step costs every lane of the group
    // some hundred instructions however short its sums are; a term added on the way into a multiplier
    // costs about twenty.
    const int max_terms = std::min((int)MUL_MAX_TERMS, env_int("BLS381_ABSORB_TERMS", MUL_MAX_TERMS));
    const int max_uses2 = env_int("BLS381_ABSORB_USES2", 4), max_uses3 = env_int("BLS381_ABSORB_USES3", 2);
    if (max_terms >= 1) {
        const int n = (int)nodes.size();
        std::vector<int> mul_uses(n, 0), other_uses(n, 0);
        for (auto& nd : nodes) {
            if (is_mul(nd)) for_each_src(nd, [&](int s) { mul_uses[s]++; });
            else for_each_src(nd, [&](int s) { other_uses[s]++; });
        }
        // with every source normalised (digits in [0, 2^28)) an operand's digits stay below
        // max(positive terms, negative terms) * 2^28: what the column accumulator has to hold
        auto mag_of = [](const std::vector<Term>& op) {
            int pos = 0, neg = 0;
            for (auto& t : op) (t.coef > 0 ? pos : neg) += std::abs(t.coef);
            return std::max(pos, neg);
        };
        std::vector<char> absorb(n, 0);
        for (int i = 0; i < n; i++) {
            const Node& L = nodes[i];
            if (L.kind != NK_LIN || other_uses[i] != 0 || mul_uses[i] == 0) continue;
            const int nt = (int)L.terms.size();
            if (nt > max_terms) continue;
            bool unit = true;
            for (auto& t : L.terms) unit = unit && std::abs(t.coef) == 1;
            if (!unit) continue;
            if ((nt == 2 && mul_uses[i] > max_uses2) || (nt == 3 && mul_uses[i] > max_uses3)) continue;
            absorb[i] = 1;
        }
        // a multiplication takes the sums it can hold: at most MUL_MAX_TERMS terms per operand, and the
        // products' digit magnitudes within the accumulator's budget of 8
        for (int i = 0; i < n; i++) {
            Node& nd = nodes[i];
            if (!is_mul(nd)) continue;
            std::vector<Term> cand[4];
            bool any = false;
            for (int j = 0; j < n_operands(nd); j++) {
                for (auto& t : nd.op[j]) {
                    if (absorb[t.node]) { for (auto& it : nodes[t.node].terms) cand[j].push_back(Term{it.node, it.coef * t.coef}); any = true; }
                    else cand[j].push_back(t);
                }
                cand[j] = merge_terms(cand[j]);
            }
            if (!any) continue;
            bool ok = true;
            int sd = 0;
            for (int j = 0; j < n_operands(nd); j++) {
                ok = ok && !cand[j].empty() && (int)cand[j].size() <= MUL_MAX_TERMS;
                for (auto& t : cand[j]) ok = ok && std::abs(t.coef) == 1;
            }
            if (ok) for (int j = 0; j < n_operands(nd); j += 2) sd += mag_of(cand[j]) * mag_of(cand[j + 1]);
            if (!ok || sd > 8) {
                // this consumer keeps its operands as they are, so the sums must exist after all
                for (int j = 0; j < n_operands(nd); j++) for (auto& t : nd.op[j]) absorb[t.node] = 0;
            }
        }
        for (int i = 0; i < n; i++) {
            Node& nd = nodes[i];
            if (!is_mul(nd)) continue;
            for (int j = 0; j < n_operands(nd); j++) {
                std::vector<Term> nt;
                for (auto& t : nd.op[j]) {
                    if (absorb[t.node]) for (auto& it : nodes[t.node].terms) nt.push_back(Term{it.node, it.coef * t.coef});
                    else nt.push_back(t);
                }
                nd.op[j] = merge_terms(nt);
            }
        }
        drop_dead_nodes(nodes);
    }
}

// ------------------------------------------------------------------------------- scheduler
namespace {

enum OpClass { OC_FMUL = 0, OC_LIN = 1, OC_COPY = 2, OC_DOTN = 3 };
inline OpClass class_of(NodeKind k) {
    if (k == NK_FMUL || k == NK_DOT2) return OC_FMUL;  // one multiplier class; see ST_DOT2
    if (k == NK_DOTN) return OC_DOTN;
    if (k == NK_LIN) return OC_LIN;
    return OC_COPY;
}

struct RawStep {
    OpClass cls;
    std::vector<int> ops;  // node ids
};

struct CopyOp {
    CopyKind kind;
    int dst, src;
};

}  // namespace

Program schedule(const Dag& dag, int K, int lds_slots, const std::string& name) {
    const std::vector<Node>& N = dag.nodes;
    const int n = (int)N.size();
    Program P;
    P.name = name; P.K = K; P.n_in = dag.n_in; P.n_out = dag.n_out;

    // Inputs and constants are never scheduled: they count as values that exist from the start
    // outside the LDS and are filled in (by a carried copy) right before each use.
    auto is_load = [&](int i) { return N[i].kind == NK_LOADIN || N[i].kind == NK_LOADC; };
    std::vector<std::vector<int>> succ(n);
    std::vector<int> npred(n, 0);
    for (int i = 0; i < n; i++) {
        std::set<int> srcs;
        for_each_src(N[i], [&](int s) { srcs.insert(s); });
        for (int s : srcs) { succ[s].push_back(i); if (!is_load(s)) npred[i]++; }
    }
    // priority = longest weighted path to a sink
    std::vector<int> prio(n, 0);
    for (int i = n - 1; i >= 0; i--) {
        int w = N[i].kind == NK_FMUL ? 7 : (N[i].kind == NK_DOT2 ? 9 : (N[i].kind == NK_LIN ? 2 : 1));
        if (N[i].kind == NK_DOTN) w = 5 + 2 * (int)N[i].pa.size();
        int best = 0;
        for (int s : succ[i]) best = std::max(best, prio[s]);
        prio[i] = best + w;
    }

    // ---- bound analysis.  Values are lazily reduced signed-digit numbers:
    //   |value| < vb * p                          (value bound)
    //   dlo * 2^28 <= digit <= dhi * 2^28         (digit interval of digits 0..12; the top digit follows from vb)
    //   product / two-product sum: result vb = 1 + ceil(sum of vb products / 2520)  (2^392 / p = 2520.2),
    //                              digits in [0, 1]; an operand is a signed sum of values, its digit magnitude
    //                              the larger end of its interval; needs the sum over the products of
    //                              magnitude(a) * magnitude(b) <= 8  (64-bit column accumulator)
    //   sum:                       vb adds up, the interval follows the coefficients' signs; magnitude <= 7 (32-bit digits)
    //   sum + carry sweep:         digits in [0, 1]
    //   sum + sweep + reduction:   vb = 2, digits in [0, 1]
    // A sum gets a sweep or a reduction only when a consumer's limit demands it.
    // (a product of larger operands is still exact, its result is just wider: |result| <
    // (1 + sum / 2520) p.  The hard limits are the 28-bit top digit of an operand, vb <= 2047, and the
    // same for results.)
    const long MUL_VB_LIMIT = 2520L * 2046, MUL_DB_LIMIT = 8;
    const int SUM_VB_LIMIT = 2047, SUM_HI_LIMIT = 7, SUM_LO_LIMIT = -7;  // digits of magnitude up to 7 * 2^28: what the reduction takes
    std::vector<int> vb(n, 1), dlo(n, 0), dhi(n, 1), dmag(n, 1);
    std::vector<char> lin_norm(n, 0), lin_reduce(n, 0);
    std::vector<int> lin_bias(n, 0);  // sums whose digits only fit 32 bits around a shifted zero, see below
    int n_reduce = 0, n_norm = 0;
    // picks the sum among `cands` with the largest bound that has not been treated yet
    std::vector<char> narrow(n, 0);  // products whose result must stay below 2p: their own operands get reduced
    auto widest = [&](const std::vector<int>& cands, const std::vector<int>& bnd, const std::vector<char>& done_flag) {
        int best = -1;
        for (int x : cands) {
            const bool sum_ok = N[x].kind == NK_LIN && !done_flag[x] && !lin_reduce[x];
            const bool mul_ok = is_product(N[x]) && !narrow[x] && bnd[x] > 2 && &bnd == &vb;
            if ((sum_ok || mul_ok) && (best < 0 || bnd[x] > bnd[best])) best = x;
        }
        if (best < 0) {
            std::string what;
            for (int x : cands)
                what += " [kind " + std::to_string((int)N[x].kind) + " vb " + std::to_string(vb[x]) + " digits " + std::to_string(dlo[x]) + ".." +
                        std::to_string(dhi[x]) + (lin_norm[x] ? " swept" : "") + (lin_reduce[x] ? " reduced" : "") + "]";
            throw std::runtime_error("scheduler: operand bounds cannot be met (" + name + "):" + what);
        }
        return best;
    };
    // marks the chosen node: a sum gets the flag, a wide product is made narrow instead
    auto treat = [&](int x, std::vector<char>& flag, int& counter) {
        if (N[x].kind == NK_LIN) { flag[x] = 1; counter++; }
        else narrow[x] = 1;
    };
    // Sums with several consumers are treated eagerly once their bounds pass a threshold: one sweep
    // or reduction there spares each consumer (and their consumers) its own.
    const int VB_EAGER = env_int("BLS381_VB_EAGER", 400), DB_EAGER = env_int("BLS381_DB_EAGER", 3);
    const bool use_bias = env_int("BLS381_LIN_BIAS", 1) != 0;
    const int REDUCE_SIBLINGS = env_int("BLS381_REDUCE_SIBLINGS", 12), REDUCE_SIBLING_RATIO = env_int("BLS381_REDUCE_SIBLING_RATIO", 4);
    struct SumBound { long v; int lo, hi, mag; };
    auto sum_bound = [&](const std::vector<Term>& terms) {
        SumBound b{0, 0, 0, 0};
        for (auto& t : terms) {
            b.v += (long)std::abs(t.coef) * vb[t.node];
            b.lo += t.coef > 0 ? t.coef * dlo[t.node] : t.coef * dhi[t.node];
            b.hi += t.coef > 0 ? t.coef * dhi[t.node] : t.coef * dlo[t.node];
        }
        b.mag = std::max(-b.lo, b.hi);
        return b;
    };
    // One pass in node order; when node i asks for a treatment of an operand x, everything from x on is
    // recomputed (nothing before x depends on x's flags).
    {
        for (int i = 0; i < n;) {
            const Node& nd = N[i];
            int redo = -1;  // node whose flags changed
            // A reduction is a flag of the whole step: every sum that shares the step pays for it.  So a sum that must
            // be reduced takes its neighbours in node order along (the components of one tower value, which the
            // scheduler puts into the same steps) when they have grown as well: their next reductions then fall
            // together too, instead of one or two in every other step.
            auto treated = [&](int x, std::vector<char>& flag, int& counter) {
                treat(x, flag, counter);
                redo = x;
                if (&flag == &lin_reduce && N[x].kind == NK_LIN && REDUCE_SIBLINGS > 0) {
                    for (int y = std::max(0, x - REDUCE_SIBLINGS); y < std::min(i, x + REDUCE_SIBLINGS + 1); y++) {
                        if (y == x || N[y].kind != NK_LIN || lin_reduce[y] || succ[y].size() < 2) continue;
                        if ((long)vb[y] * REDUCE_SIBLING_RATIO < vb[x]) continue;
                        lin_reduce[y] = 1; counter++;
                        redo = std::min(redo, y);
                    }
                }
            };
            switch (nd.kind) {
                case NK_LOADIN: vb[i] = 10; dlo[i] = 0; dhi[i] = 1; break;  // any 384-bit pattern is below 10p
                case NK_LOADC: vb[i] = 1; dlo[i] = 0; dhi[i] = 1; break;
                case NK_FMUL:
                case NK_DOT2: {
                    std::vector<int> ops;
                    long sv = 0;
                    int sd = 0;
                    bool operand_too_wide = false;
                    for (int j = 0; j < n_operands(nd); j += 2) {
                        const SumBound a = sum_bound(nd.op[j]), b = sum_bound(nd.op[j + 1]);
                        sv += a.v * b.v;
                        sd += a.mag * b.mag;
                        operand_too_wide |= a.hi > SUM_HI_LIMIT || a.lo < SUM_LO_LIMIT || b.hi > SUM_HI_LIMIT || b.lo < SUM_LO_LIMIT;
                    }
                    for (int j = 0; j < n_operands(nd); j++) for (auto& t : nd.op[j]) ops.push_back(t.node);
                    bool vb_fail = sv > (narrow[i] ? 2500L : MUL_VB_LIMIT);
                    for (int j = 0; j < n_operands(nd); j++) vb_fail |= sum_bound(nd.op[j]).v > SUM_VB_LIMIT;
                    if (vb_fail) treated(widest(ops, vb, lin_reduce), lin_reduce, n_reduce);
                    else if (sd > MUL_DB_LIMIT || operand_too_wide) treated(widest(ops, dmag, lin_norm), lin_norm, n_norm);
                    vb[i] = 1 + (int)((sv + 2519) / 2520); dlo[i] = 0; dhi[i] = 1;
                    break;
                }
                case NK_DOTN: {
                    std::vector<int> ops;
                    long sv = 0;
                    int sd = 0;
                    bool too_wide = false;
                    for (size_t t = 0; t < nd.pa.size(); t++) {
                        const int a = nd.pa[t].node, b = nd.pb[t].node;
                        sv += (long)vb[a] * vb[b];
                        sd += dmag[a] * dmag[b];
                        too_wide |= vb[a] > SUM_VB_LIMIT || vb[b] > SUM_VB_LIMIT;
                        ops.push_back(a); ops.push_back(b);
                    }
                    if (sv > (narrow[i] ? 2500L : MUL_VB_LIMIT) || too_wide) treated(widest(ops, vb, lin_reduce), lin_reduce, n_reduce);
                    else if (sd > MUL_DB_LIMIT) treated(widest(ops, dmag, lin_norm), lin_norm, n_norm);
                    vb[i] = 1 + (int)((sv + 2519) / 2520); dlo[i] = 0; dhi[i] = 1;
                    break;
                }
                case NK_LIN: {
                    const SumBound b = sum_bound(nd.terms);
                    std::vector<int> ops;
                    for (auto& t : nd.terms) ops.push_back(t.node);
                    // A sum whose digit interval is no wider than 32 bits hold, but lopsided, is computed around a
                    // shifted zero: the digit vector (B, B - B/2^28, ..., -B/2^28 on top) * 2^28 has the value 0, so
                    // starting the accumulators from it moves every digit by B * 2^28 and changes nothing else.
                    // Such a sum is swept right away.
                    int bias = 0;
                    bool fits = b.hi <= SUM_HI_LIMIT && b.lo >= SUM_LO_LIMIT;
                    if (!fits && use_bias && b.hi - b.lo <= SUM_HI_LIMIT - SUM_LO_LIMIT) {
                        bias = -((b.hi + b.lo) / 2);  // centred
                        if (b.hi + bias > SUM_HI_LIMIT) bias = SUM_HI_LIMIT - b.hi;
                        if (b.lo + bias < SUM_LO_LIMIT) bias = SUM_LO_LIMIT - b.lo;
                        fits = bias >= -8 && bias <= 7 && b.hi + bias <= SUM_HI_LIMIT && b.lo + bias >= SUM_LO_LIMIT;
                        if (!fits) bias = 0;
                    }
                    if (b.v > SUM_VB_LIMIT) treated(widest(ops, vb, lin_reduce), lin_reduce, n_reduce);
                    else if (!fits) treated(widest(ops, dmag, lin_norm), lin_norm, n_norm);
                    else if (bias != 0) { if (!lin_norm[i] && !lin_reduce[i]) { lin_norm[i] = 1; n_norm++; } }
                    else if (succ[i].size() >= 2) {
                        if (b.v >= VB_EAGER && !lin_reduce[i]) { lin_reduce[i] = 1; n_reduce++; }
                        else if (b.mag >= DB_EAGER && !lin_norm[i] && !lin_reduce[i]) { lin_norm[i] = 1; n_norm++; }
                    }
                    lin_bias[i] = bias;
                    vb[i] = lin_reduce[i] ? 2 : (int)b.v;
                    // (the sweep / reduction flags are per step: a sum that asks for neither may still get one from a
                    // neighbour in its step, so its interval has to cover the normalised form as well)
                    if (lin_reduce[i] || lin_norm[i]) { dlo[i] = 0; dhi[i] = 1; } else { dlo[i] = std::min(b.lo, 0); dhi[i] = std::max(b.hi, 1); }
                    break;
                }
                case NK_STOREOUT:
                    // a store handles (-p, 2p) with normalised digits: what multiplications produce
                    if (!is_product(N[nd.a]))
                        throw std::runtime_error("scheduler: stored value must come from a multiplication");
                    if (vb[nd.a] > 2 && !narrow[nd.a]) { narrow[nd.a] = 1; redo = nd.a; }
                    break;
            }
            dmag[i] = std::max(-dlo[i], dhi[i]);
            i = redo >= 0 ? redo : i + 1;
        }
    }
    if (const char* tn = getenv("BLS381_SCHED_NODE")) {
        std::vector<int> todo{atoi(tn)};
        for (int depth = 0; depth < 3; depth++) {
            std::vector<int> nxt;
            for (int x : todo) {
                if (x < 0 || x >= n) continue;
                fprintf(stderr, "node %d kind %d vb %d [%d..%d]%s%s bias %d:", x, (int)N[x].kind, vb[x], dlo[x], dhi[x], lin_norm[x] ? " swept" : "",
                        lin_reduce[x] ? " reduced" : "", lin_bias[x]);
                for (auto& t : N[x].terms) { fprintf(stderr, " %+d*n%d", t.coef, t.node); nxt.push_back(t.node); }
                for (int j = 0; j < n_operands(N[x]); j++) { fprintf(stderr, " |"); for (auto& t : N[x].op[j]) fprintf(stderr, " %+d*n%d", t.coef, t.node); }
                fprintf(stderr, "\n");
            }
            todo = nxt;
        }
    }
    if (getenv("BLS381_SCHED_VERIFY")) {
        for (int i = 0; i < n; i++) {
            const Node& nd = N[i];
            if (is_mul(nd)) {
                int sd = 0;
                for (int j = 0; j < n_operands(nd); j += 2) sd += sum_bound(nd.op[j]).mag * sum_bound(nd.op[j + 1]).mag;
                if (sd > MUL_DB_LIMIT) fprintf(stderr, "verify: node %d product magnitudes %d\n", i, sd);
            } else if (nd.kind == NK_DOTN) {
                int sd = 0;
                for (size_t t = 0; t < nd.pa.size(); t++) sd += dmag[nd.pa[t].node] * dmag[nd.pb[t].node];
                if (sd > MUL_DB_LIMIT) fprintf(stderr, "verify: node %d product magnitudes %d\n", i, sd);
            } else if (nd.kind == NK_LIN) {
                const SumBound b = sum_bound(nd.terms);
                const bool flagged = lin_norm[i] || lin_reduce[i];
                if ((flagged ? 0 : std::min(b.lo, 0)) != dlo[i] || (flagged ? 1 : std::max(b.hi, 1)) != dhi[i])
                    fprintf(stderr, "verify: sum %d interval %d..%d stored %d..%d flagged %d bias %d\n", i, b.lo, b.hi, dlo[i], dhi[i], (int)flagged, lin_bias[i]);
                if (b.lo + lin_bias[i] < SUM_LO_LIMIT || b.hi + lin_bias[i] > SUM_HI_LIMIT)
                    fprintf(stderr, "verify: sum %d interval %d..%d bias %d leaves a digit\n", i, b.lo, b.hi, lin_bias[i]);
            }
        }
    }
    n_norm = 0;
    for (int i = 0; i < n; i++) if (lin_norm[i] && lin_reduce[i]) lin_norm[i] = 0;  // a reduction sweeps anyway
    for (int i = 0; i < n; i++) n_norm += lin_norm[i];
    P.n_reduce = n_reduce; P.n_norm = n_norm;

    // how a multiplication's operands are laid out in its descriptor.  One term per operand with signs
    // that fit "a * b + c * (+-d)" (products swapped if need be) runs on the short path; anything else
    // takes the path that adds up signed terms.
    struct MulShape { int width; bool swap; };
    auto mul_shape = [&](const Node& nd) {
        MulShape sh{1, false};
        for (int j = 0; j < n_operands(nd); j++) sh.width = std::max(sh.width, (int)nd.op[j].size());
        if (sh.width > 1) return sh;
        const int s1 = nd.op[0][0].coef * nd.op[1][0].coef;
        const int s2 = nd.kind == NK_DOT2 ? nd.op[2][0].coef * nd.op[3][0].coef : 1;
        if (s1 > 0) return sh;
        if (nd.kind == NK_DOT2 && s2 > 0) { sh.swap = true; return sh; }
        sh.width = 2;  // both products negative: needs the signed path
        return sh;
    };

    // ---- pass 1: list scheduling into K-wide single-class steps
    const int WINDOW = env_int("BLS381_WINDOW", 600);
    const bool split_mul = env_int("BLS381_SPLIT_MUL", 0) != 0;
    const bool group_wide = env_int("BLS381_GROUP_WIDE", 1) != 0;
    std::vector<char> done(n, 0);
    std::vector<int> ready;  // nodes whose predecessors all sit in closed steps
    int lo = 0, n_done = 0;
    for (int i = 0; i < n; i++) {
        if (is_load(i)) { done[i] = 1; n_done++; }
        else if (npred[i] == 0) ready.push_back(i);
    }
    std::vector<RawStep> raw;
    while (n_done < n) {
        while (lo < n && done[lo]) lo++;
        std::vector<int> cand[4];
        for (int r : ready) if (r < lo + WINDOW) cand[class_of(N[r].kind)].push_back(r);
        int cls;
        // full multiplier steps first, then sums and copies (which unlock more multiplications),
        // then partial multiplier steps
        if ((int)cand[OC_DOTN].size() >= K) cls = OC_DOTN;
        else if ((int)cand[OC_FMUL].size() >= K) cls = OC_FMUL;
        else if (!cand[OC_LIN].empty()) cls = OC_LIN;
        else if (!cand[OC_COPY].empty()) cls = OC_COPY;
        else if (cand[OC_DOTN].size() >= cand[OC_FMUL].size() && !cand[OC_DOTN].empty()) cls = OC_DOTN;
        else if (!cand[OC_FMUL].empty()) cls = OC_FMUL;
        else if (!cand[OC_DOTN].empty()) cls = OC_DOTN;
        else throw std::runtime_error("scheduler: no candidate");
        auto& c = cand[cls];
        std::sort(c.begin(), c.end(), [&](int a, int b) {
            if (prio[a] != prio[b]) return prio[a] > prio[b];
            return a < b;
        });
        if (cls == OC_FMUL && split_mul) {
            // a single product riding in a two-product step costs a two-product: keep the kinds apart
            // whenever one of them fills a step on its own
            std::vector<int> singles, doubles;
            for (int x : c) (N[x].kind == NK_FMUL ? singles : doubles).push_back(x);
            if ((int)doubles.size() >= K) c.swap(doubles);
            else if ((int)singles.size() >= K) c.swap(singles);
        }
        if ((int)c.size() > K && cls == OC_FMUL && group_wide) {
            // operands of several terms make the whole step take the long load path: keep the most urgent
            // op and fill the step with ops of its kind first
            const int lead_w = mul_shape(N[c[0]]).width;
            std::stable_sort(c.begin() + 1, c.end(), [&](int a, int b) {
                return std::abs(mul_shape(N[a]).width - lead_w) < std::abs(mul_shape(N[b]).width - lead_w);
            });
        }
        if ((int)c.size() > K && cls == OC_DOTN) {
            // every lane of the step multiplies as many products as its largest op: group by product count
            const int lead_n = (int)N[c[0]].pa.size();
            std::stable_sort(c.begin() + 1, c.end(), [&](int a, int b) {
                return std::abs((int)N[a].pa.size() - lead_n) < std::abs((int)N[b].pa.size() - lead_n);
            });
        }
        if ((int)c.size() > K) {
            if (cls == OC_LIN) {
                // keep the most urgent op, fill the step with the ops closest to it in cost
                int lead = c[0];
                auto cost = [&](int x) { return (int)N[x].terms.size() * 22 + lin_norm[x] * 40 + lin_reduce[x] * 100; };
                std::stable_sort(c.begin() + 1, c.end(), [&](int a, int b) {
                    return std::abs(cost(a) - cost(lead)) < std::abs(cost(b) - cost(lead));
                });
            }
        }
        {
            // at most K ops whose distinct operands and results all fit the group's LDS slots at once
            const size_t budget = (size_t)std::max(8, lds_slots - 2 - 2);
            std::set<int> touched;
            std::vector<int> kept;
            for (int x : c) {
                if ((int)kept.size() == K) break;
                std::set<int> t2 = touched;
                for_each_src(N[x], [&](int s2) { t2.insert(s2); });
                if (N[x].kind != NK_STOREOUT) t2.insert(x);
                if (t2.size() > budget && !kept.empty()) continue;
                touched.swap(t2);
                kept.push_back(x);
            }
            c.swap(kept);
        }
        RawStep st; st.cls = (OpClass)cls; st.ops = c;
        std::set<int> chosen(c.begin(), c.end());
        std::vector<int> nr;
        for (int r : ready) if (!chosen.count(r)) nr.push_back(r);
        for (int x : c) {
            done[x] = 1; n_done++;
            for (int s : succ[x]) if (--npred[s] == 0) nr.push_back(s);
        }
        ready.swap(nr);
        raw.push_back(st);
    }

    // ---- pass 2: LDS slot assignment with spilling
    const int S = (int)raw.size();
    std::vector<std::vector<int>> use_steps(n);  // steps in which a node's value is read
    for (int s = 0; s < S; s++)
        for (int op : raw[s].ops) {
            std::set<int> srcs;
            for_each_src(N[op], [&](int x) { srcs.insert(x); });
            for (int x : srcs) use_steps[x].push_back(s);
        }
    std::vector<size_t> use_ptr(n, 0);
    auto next_use = [&](int node, int after_step) -> int {  // first use step > after_step, or INT_MAX
        auto& u = use_steps[node];
        size_t p = use_ptr[node];
        while (p < u.size() && u[p] <= after_step) p++;
        return p < u.size() ? u[p] : 0x7fffffff;
    };

    // The last two LDS slots are reserved: ZERO always holds 0 (padding operand of idle lanes and
    // of sums shorter than the step's longest), TRASH receives what idle lanes store.  With them
    // the multiplier and sum steps need no per-lane branches.
    const int ZERO_SLOT = lds_slots - 2, TRASH_SLOT = lds_slots - 1;
    lds_slots -= 2;
    std::vector<int> lds_of(n, -1), tmp_of(n, -1);
    std::vector<int> owner(lds_slots, -1);
    std::vector<int> free_lds;
    for (int i = lds_slots - 1; i >= 0; i--) free_lds.push_back(i);
    std::vector<int> free_tmp;
    int n_tmp = 0;

    // A copy step holds copies of one kind, named in its header: the kernel branches on the kind once for all lanes.
    auto emit_copy_steps = [&](std::vector<CopyOp>& all) {
        for (uint32_t kind = 0; kind <= CK_OUT; kind++) {
            std::vector<CopyOp> ops;
            for (const CopyOp& c : all) if ((uint32_t)c.kind == kind) ops.push_back(c);
            for (size_t base = 0; base < ops.size(); base += K) {
                StepHdr h{ST_COPY, kind};
                P.hdrs.push_back(h);
                for (int r = 0; r < K; r++) {
                    OpDesc d; std::memset(&d, 0, sizeof d);
                    if (base + r < ops.size()) {
                        const CopyOp& c = ops[base + r];
                        d.hdr = 1u | ((uint32_t)c.kind << 8);
                        d.w[0] = (uint16_t)c.dst; d.w[1] = (uint16_t)c.src;
                        P.n_copy++;
                    }
                    P.descs.push_back(d);
                }
                P.n_copy_steps++;
            }
        }
        all.clear();
    };

    // Copies ride on compute steps where they can ("piggyback"): a lane of a multiplier or sum step
    // may, besides its operation, spill one LDS slot (read and stored as the step starts) and fill one
    // (global loads issued as the step starts, LDS written after the step's results).  So a fill for
    // step s rides on the compute step before it, with its latency behind that step's arithmetic, and
    // an eviction rides on the step whose start still sees the victim.  What cannot ride (no carrier,
    // carrier full, victim produced by the carrier itself) becomes a copy step as before.
    bool prev_open = false;      // the last emitted step is a compute step nothing has followed yet
    size_t prev_idx = 0;         // its index in P.hdrs
    std::set<int> prev_results;  // nodes it produces
    int prev_nfill = 0, prev_nspill = 0;
    std::set<int> prev_spilled;  // spill-array slots written by spills riding on it
    const bool piggyback = getenv("BLS381_NO_PIGGYBACK") == nullptr;

    for (int s = 0; s < S; s++) {
        const RawStep& st = raw[s];
        std::set<int> srcs;
        for (int op : st.ops) for_each_src(N[op], [&](int x) { srcs.insert(x); });
        std::vector<CopyOp> spills_a, spills_b, fills;
        std::vector<std::pair<int, int>> cur_spills;  // (LDS slot, spill-array slot) riding on this step
        std::set<int> pinned(srcs.begin(), srcs.end());  // may not be evicted while forming this step
        const bool carrier_prev = piggyback && prev_open;
        const bool carrier_cur = piggyback && st.cls != OC_COPY;

        // how the slot's previous occupant left: 0 nothing to do, 1 spill rides, 2 spill is a copy step
        auto take_slot = [&](bool for_fill, int& how) -> int {
            how = 0;
            if (!free_lds.empty()) { int x = free_lds.back(); free_lds.pop_back(); return x; }
            // evict the resident value with the furthest next use; for a fill prefer victims the
            // carrier does not produce (their spill can ride on it)
            int victim = -1, far = -1;
            for (int pass = 0; pass < 2 && victim < 0; pass++)
                for (int sl = 0; sl < lds_slots; sl++) {
                    int o = owner[sl];
                    if (o < 0 || pinned.count(o)) continue;
                    if (pass == 0 && for_fill && carrier_prev && prev_results.count(o)) continue;
                    int nu = next_use(o, s - 1);
                    if (nu > far) { far = nu; victim = sl; }
                }
            if (victim < 0) throw std::runtime_error("scheduler: LDS slots exhausted (" + name + ")");
            int o = owner[victim];
            bool clean = (N[o].kind == NK_LOADIN || N[o].kind == NK_LOADC || tmp_of[o] >= 0);
            const bool from_prev = carrier_prev && prev_results.count(o);
            if (!clean && far != 0x7fffffff) {
                int t;
                if (!free_tmp.empty()) { t = free_tmp.back(); free_tmp.pop_back(); }
                else t = n_tmp++;
                tmp_of[o] = t;
                if (for_fill) {
                    if (carrier_prev && !from_prev && prev_nspill < K && t < 128) {  // 7-bit fields in the packed descriptor
                        OpDesc& d = P.descs[prev_idx * K + prev_nspill++];
                        d.w[DW_SPILL] = (uint16_t)(victim | 0x8000); d.w[DW_SPILL + 1] = (uint16_t)t;
                        prev_spilled.insert(t);
                        P.hdrs[prev_idx].max_terms |= 1u << 17;
                        P.n_piggy_spill++;
                        how = 1;
                    } else {
                        spills_a.push_back(CopyOp{CK_SPILL, t, victim});
                        how = 2;
                    }
                } else {
                    if (carrier_cur && (int)cur_spills.size() < K && t < 128) { cur_spills.push_back({victim, t}); how = 1; }
                    else { spills_b.push_back(CopyOp{CK_SPILL, t, victim}); how = 2; }
                }
            } else if (from_prev) {
                how = 2;  // the carrier writes this slot itself: nothing may be filled into it while it runs
            }
            lds_of[o] = -1; owner[victim] = -1;
            return victim;
        };

        // 1. bring non-resident sources back
        for (int x : srcs) {
            if (lds_of[x] >= 0) continue;
            int how;
            int sl = take_slot(true, how);
            lds_of[x] = sl; owner[sl] = x;
            CopyOp f;
            if (N[x].kind == NK_LOADIN) f = CopyOp{CK_IN, sl, N[x].idx};
            else if (N[x].kind == NK_LOADC) f = CopyOp{CK_CONST, sl, N[x].idx};
            else { assert(tmp_of[x] >= 0); f = CopyOp{CK_FILL, sl, tmp_of[x]}; }
            // (a step's fills are loaded before its spills are stored: it must not fill what it spills)
            const bool aliases = f.kind == CK_FILL && prev_spilled.count(f.src);
            if (carrier_prev && how != 2 && prev_nfill < K && f.src < 1024 && !aliases) {
                OpDesc& d = P.descs[prev_idx * K + prev_nfill++];
                d.w[DW_FILL] = (uint16_t)(sl | 0x8000); d.w[DW_FILL + 1] = (uint16_t)(f.src | ((uint32_t)f.kind << 12));
                P.hdrs[prev_idx].max_terms |= 1u << 16;
                P.n_piggy_fill++;
            } else {
                fills.push_back(f);
            }
        }
        // record operand slots now (before any source slot is recycled for a destination)
        std::map<int, int> src_slot;
        for (int x : srcs) src_slot[x] = lds_of[x];
        // 2. release sources that die here
        for (int x : srcs) {
            if (next_use(x, s) == 0x7fffffff) {
                int sl = lds_of[x];
                owner[sl] = -1; lds_of[x] = -1; free_lds.push_back(sl);
                if (tmp_of[x] >= 0) { free_tmp.push_back(tmp_of[x]); tmp_of[x] = -1; }
            }
        }
        // 3. destinations.  Spills made from here on may reuse a spill-array slot released in (2),
        //    which a fill of this very step still has to read: they run after the fills (those that
        //    ride on this step do: every fill for it has landed before it starts).
        std::map<int, int> dst_slot;
        for (int op : st.ops) {
            if (N[op].kind == NK_STOREOUT) continue;
            int how;
            int sl = take_slot(false, how);
            lds_of[op] = sl; owner[sl] = op; dst_slot[op] = sl;
            pinned.insert(op);
        }
        // a spill reads LDS before this step's fills and results overwrite it
        emit_copy_steps(spills_a);
        emit_copy_steps(fills);
        emit_copy_steps(spills_b);
        // 4. the step itself (a step of copies: one step per kind of copy)
        std::vector<std::vector<int>> parts;
        if (st.cls == OC_COPY) {
            for (NodeKind nk : {NK_LOADIN, NK_LOADC, NK_STOREOUT}) {
                std::vector<int> part;
                for (int op : st.ops) if (N[op].kind == nk) part.push_back(op);
                if (!part.empty()) parts.push_back(part);
            }
        } else {
            parts.push_back(st.ops);
        }
        for (const std::vector<int>& step_ops : parts) {
        StepHdr h; h.max_terms = 0;
        if (st.cls == OC_COPY)
            h.max_terms = N[step_ops[0]].kind == NK_LOADIN ? CK_IN : (N[step_ops[0]].kind == NK_LOADC ? CK_CONST : CK_OUT);
        int step_rounds = 0;
        bool any_dot2 = false;
        int mul_width = 1;
        for (int op : step_ops) {
            any_dot2 |= (N[op].kind == NK_DOT2);
            if (is_mul(N[op])) mul_width = std::max(mul_width, mul_shape(N[op]).width);
        }
        // a step holding a two-product op runs the two-product multiplier on every lane; single
        // products in it use zero for the second product
        h.type = st.cls == OC_FMUL ? (any_dot2 ? ST_DOT2 : ST_FMUL) : (st.cls == OC_LIN ? ST_LIN : (st.cls == OC_DOTN ? ST_DOTN : ST_COPY));
        if (st.cls == OC_DOTN) {
            if (K != 6) throw std::runtime_error("scheduler: n-product operations need 6 lanes per group (" + name + ")");
            h.max_terms = 3;  // the kernel multiplies at least three products
        }
        if (st.cls == OC_FMUL) h.max_terms = (uint32_t)mul_width;
        for (int r = 0; r < K; r++) {
            OpDesc d; std::memset(&d, 0, sizeof d);
            if (st.cls != OC_COPY) {  // idle lane / padding defaults
                d.w[0] = (uint16_t)TRASH_SLOT;
                for (int t = 1; t <= 12; t++) d.w[t] = (uint16_t)ZERO_SLOT;
                if (st.cls == OC_DOTN) for (int t = 0; t < 4; t++) d.w[DW_DOTN_HI + t] = (uint16_t)ZERO_SLOT;
                if (r < (int)cur_spills.size()) {
                    d.w[DW_SPILL] = (uint16_t)(cur_spills[r].first | 0x8000); d.w[DW_SPILL + 1] = (uint16_t)cur_spills[r].second;
                    P.n_piggy_spill++;
                }
            }
            if (r < (int)step_ops.size()) {
                int op = step_ops[r];
                const Node& nd = N[op];
                d.hdr = 1u;
                switch (nd.kind) {
                    case NK_FMUL:
                    case NK_DOT2: {
                        const MulShape sh = mul_shape(nd);
                        if (getenv("BLS381_SCHED_TRACE") && atoi(getenv("BLS381_SCHED_TRACE")) == (int)P.hdrs.size()) {
                            fprintf(stderr, "step %zu lane %d node %d:", P.hdrs.size(), r, op);
                            for (int j = 0; j < n_operands(nd); j++) {
                                fprintf(stderr, " op%d{", j);
                                for (auto& t : nd.op[j])
                                    fprintf(stderr, " %+d*n%d(kind %d [%d..%d]%s%s slot %d)", t.coef, t.node, (int)N[t.node].kind, dlo[t.node], dhi[t.node],
                                            lin_norm[t.node] ? " swept" : "", lin_reduce[t.node] ? " reduced" : "", src_slot[t.node]);
                                fprintf(stderr, " }");
                            }
                            fprintf(stderr, "\n");
                        }
                        if (nd.kind == NK_DOT2) { d.hdr |= 2u; P.n_dot2++; } else P.n_fmul++;
                        d.w[0] = (uint16_t)dst_slot[op];
                        for (int j = 0; j < n_operands(nd); j++) {
                            const std::vector<Term>& terms = nd.op[sh.swap ? (j ^ 2) : j];
                            // on the short path the sign of each product sits on its second factor, and only
                            // d's is looked at: a * b + c * (+-d)
                            const int prod_sign = terms[0].coef * nd.op[(sh.swap ? (j ^ 2) : j) ^ 1][0].coef;
                            for (size_t t = 0; t < terms.size(); t++) {
                                bool neg = terms[t].coef < 0;
                                if (mul_width == 1) neg = (j & 1) && prod_sign < 0;
                                d.w[1 + 3 * j + t] = (uint16_t)(src_slot[terms[t].node] | (neg ? 0x8000 : 0));
                            }
                        }
                        break;
                    }
                    case NK_DOTN: {
                        const int np = (int)nd.pa.size();
                        assert(np >= 1 && np <= DOTN_MAX_PAIRS);
                        d.w[0] = (uint16_t)dst_slot[op];
                        for (int t = 0; t < np; t++) {
                            const int ia = t < 6 ? 1 + 2 * t : DW_DOTN_HI + 2 * (t - 6);
                            d.w[ia] = (uint16_t)src_slot[nd.pa[t].node];
                            d.w[ia + 1] = (uint16_t)(src_slot[nd.pb[t].node] | (nd.pb[t].coef < 0 ? 0x8000 : 0));
                        }
                        h.max_terms = std::max<uint32_t>(h.max_terms, np);
                        P.n_dotn++; P.n_dotn_products += np;
                        break;
                    }
                    case NK_LIN: {
                        int nt = (int)nd.terms.size();
                        assert(nt <= LIN_MAX_TERMS);
                        d.hdr |= (uint32_t)nt << 8;
                        d.hdr |= ((uint32_t)lin_bias[op] & 0xfu) << 24;
                        d.w[0] = (uint16_t)dst_slot[op];
                        for (int t = 0; t < nt; t++) {
                            assert(nd.terms[t].coef != 0 && std::abs(nd.terms[t].coef) <= LIN_MAX_COEF);
                            d.w[1 + t] = (uint16_t)(src_slot[nd.terms[t].node] | (((uint32_t)nd.terms[t].coef & 0xffu) << 7));
                        }
                        h.max_terms = std::max<uint32_t>(h.max_terms, nt);
                        step_rounds |= (lin_reduce[op] ? 2 : 0) | (lin_norm[op] ? 1 : 0);
                        P.n_lin++; P.n_lin_terms += nt;
                        break;
                    }
                    case NK_LOADIN:
                        d.hdr |= (uint32_t)CK_IN << 8; d.w[0] = (uint16_t)dst_slot[op]; d.w[1] = (uint16_t)nd.idx; P.n_copy++;
                        break;
                    case NK_LOADC:
                        d.hdr |= (uint32_t)CK_CONST << 8; d.w[0] = (uint16_t)dst_slot[op]; d.w[1] = (uint16_t)nd.idx; P.n_copy++;
                        break;
                    case NK_STOREOUT:
                        d.hdr |= (uint32_t)CK_OUT << 8; d.w[0] = (uint16_t)nd.idx; d.w[1] = (uint16_t)src_slot[nd.a]; P.n_copy++;
                        break;
                }
            }
            P.descs.push_back(d);
        }
        if (st.cls == OC_LIN) h.max_terms |= (uint32_t)step_rounds << 8;  // carry sweep / reduction wanted by some op
        if (!cur_spills.empty()) h.max_terms |= 1u << 17;
        P.hdrs.push_back(h);
        if (st.cls == OC_COPY) P.n_copy_steps++;
        if (st.cls == OC_FMUL && mul_width > 1) P.n_wide_mul_steps++;
        if (st.cls == OC_FMUL && any_dot2) P.n_dot2_steps++;
        else if (st.cls == OC_FMUL) P.n_fmul_steps++;
        }
        prev_open = st.cls != OC_COPY;
        prev_idx = P.hdrs.size() - 1;
        prev_results.clear();
        for (int op : st.ops) if (N[op].kind != NK_STOREOUT) prev_results.insert(op);
        prev_nfill = 0; prev_nspill = (int)cur_spills.size();
        prev_spilled.clear();
        for (auto& cs : cur_spills) prev_spilled.insert(cs.second);
        if (st.cls == OC_LIN) P.n_lin_steps++;
        else if (st.cls == OC_DOTN) P.n_dotn_steps++;
        // values produced but never read (cannot happen after DCE) would leak a slot; guard anyway
        for (int op : st.ops)
            if (N[op].kind != NK_STOREOUT && use_steps[op].empty()) {
                int sl = lds_of[op]; owner[sl] = -1; lds_of[op] = -1; free_lds.push_back(sl);
            }
    }
    P.n_lds_slots = lds_slots + 2;
    P.n_tmp_slots = n_tmp;
    return P;
}

}  // namespace bls381
