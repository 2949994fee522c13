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
    Node n; n.kind = NK_FMUL; n.a = a; n.b = b;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::dot2(int a, int b, int c, int d, bool neg_d) {
    assert(a >= 0 && b >= 0 && c >= 0 && d >= 0);
    Node n; n.kind = NK_DOT2; n.a = a; n.b = b; n.c = c; n.d = d; n.neg_d = neg_d;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
int Dag::lin(std::vector<Term> terms) {
    assert(!terms.empty());
    Node n; n.kind = NK_LIN; n.terms = std::move(terms);
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}
void Dag::store_out(int node, int slot) {
    assert(node >= 0);
    Node n; n.kind = NK_STOREOUT; n.a = node; n.idx = slot;
    nodes.push_back(n);
}

static void for_each_src(const Node& n, const std::function<void(int)>& f) {
    switch (n.kind) {
        case NK_FMUL: f(n.a); f(n.b); break;
        case NK_DOT2: f(n.a); f(n.b); f(n.c); f(n.d); break;
        case NK_LIN: for (auto& t : n.terms) f(t.node); break;
        case NK_STOREOUT: f(n.a); break;
        default: break;
    }
}

void Dag::optimise() {
    const int n = (int)nodes.size();
    std::vector<int> uses(n, 0);
    for (auto& nd : nodes) for_each_src(nd, [&](int s) { uses[s]++; });
    // inline single-use LINs into LIN consumers
    for (int i = 0; i < n; i++) {
        Node& nd = nodes[i];
        if (nd.kind != NK_LIN) continue;
        bool changed = true;
        while (changed) {
            changed = false;
            for (size_t k = 0; k < nd.terms.size(); k++) {
                Term t = nd.terms[k];
                const Node& in = nodes[t.node];
                if (in.kind != NK_LIN || uses[t.node] != 1) continue;
                if (nd.terms.size() - 1 + in.terms.size() > (size_t)LIN_MAX_TERMS) continue;
                std::vector<Term> nt;
                for (size_t j = 0; j < nd.terms.size(); j++) {
                    if (j != k) { nt.push_back(nd.terms[j]); continue; }
                    for (auto& it : in.terms) nt.push_back(Term{it.node, it.neg != t.neg});
                }
                uses[t.node] = 0;
                nd.terms = nt;
                changed = true;
                break;
            }
        }
    }
    // dead-code elimination
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
        if (nd.kind == NK_FMUL) { nd.a = remap[nd.a]; nd.b = remap[nd.b]; }
        else if (nd.kind == NK_DOT2) { nd.a = remap[nd.a]; nd.b = remap[nd.b]; nd.c = remap[nd.c]; nd.d = remap[nd.d]; }
        else if (nd.kind == NK_STOREOUT) { nd.a = remap[nd.a]; }
        else if (nd.kind == NK_LIN) for (auto& t : nd.terms) t.node = remap[t.node];
        remap[i] = (int)out.size();
        out.push_back(nd);
    }
    nodes.swap(out);
}

// ------------------------------------------------------------------------------- scheduler
namespace {

enum OpClass { OC_FMUL = 0, OC_LIN = 1, OC_COPY = 2, OC_DOT2 = 3 };
inline OpClass class_of(NodeKind k) {
    if (k == NK_FMUL || k == NK_DOT2) return OC_FMUL;  // one multiplier class; see ST_DOT2
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
        int best = 0;
        for (int s : succ[i]) best = std::max(best, prio[s]);
        prio[i] = best + w;
    }

    // ---- bound analysis.  Values are lazily reduced signed-digit numbers:
    //   |value| < vb * p                 (value bound)
    //   |digit| <= db * 2^28             (digit bound; the top digit follows from vb)
    //   product / two-product sum: result vb = 1 + ceil(sum of vb products / 2520)  (2^392 / p = 2520.2),
    //                              db = 1; needs sum of db products <= 8  (64-bit column accumulator)
    //   sum:                       vb and db add up; db <= 7 (32-bit digits)
    //   sum + carry sweep:         db = 1
    //   sum + sweep + reduction:   vb = 2, db = 1
    // A sum gets a sweep or a reduction only when a consumer's limit demands it.
    // (a product of larger operands is still exact, its result is just wider: |result| <
    // (1 + sum / 2520) p.  The hard limits are the 28-bit top digit of an operand, vb <= 2047, and the
    // same for results.)
    const long MUL_VB_LIMIT = 2520L * 2046, MUL_DB_LIMIT = 8;
    const int SUM_VB_LIMIT = 2047, SUM_DB_LIMIT = 7;
    std::vector<int> vb(n, 1), db(n, 1);
    std::vector<char> lin_norm(n, 0), lin_reduce(n, 0);
    int n_reduce = 0, n_norm = 0;
    // picks the sum among `cands` with the largest bound that has not been treated yet
    std::vector<char> narrow(n, 0);  // products whose result must stay below 2p: their own operands get reduced
    auto widest = [&](const std::vector<int>& cands, const std::vector<int>& bnd, const std::vector<char>& done_flag) {
        int best = -1;
        for (int x : cands) {
            const bool sum_ok = N[x].kind == NK_LIN && !done_flag[x] && !lin_reduce[x];
            const bool mul_ok = (N[x].kind == NK_FMUL || N[x].kind == NK_DOT2) && !narrow[x] && bnd[x] > 2 && &bnd == &vb;
            if ((sum_ok || mul_ok) && (best < 0 || bnd[x] > bnd[best])) best = x;
        }
        if (best < 0) throw std::runtime_error("scheduler: operand bounds cannot be met (" + name + ")");
        return best;
    };
    // marks the chosen node: a sum gets the flag, a wide product is made narrow instead
    auto treat = [&](int x, std::vector<char>& flag, int& counter) {
        if (N[x].kind == NK_LIN) { flag[x] = 1; counter++; }
        else narrow[x] = 1;
    };
    // Sums with several consumers are treated eagerly once their bounds pass a threshold: one sweep
    // or reduction there spares each consumer (and their consumers) its own.
    auto env_int = [](const char* nm, int dflt) { const char* v = getenv(nm); return v ? atoi(v) : dflt; };
    const int VB_EAGER = env_int("BLS381_VB_EAGER", 400), DB_EAGER = env_int("BLS381_DB_EAGER", 3);
    // One pass in node order; when node i asks for a treatment of an operand x, everything from x on is
    // recomputed (nothing before x depends on x's flags).
    {
        for (int i = 0; i < n;) {
            const Node& nd = N[i];
            int redo = -1;  // node whose flags changed
            auto treated = [&](int x, std::vector<char>& flag, int& counter) { treat(x, flag, counter); redo = x; };
            switch (nd.kind) {
                case NK_LOADIN: vb[i] = 10; db[i] = 1; break;  // any 384-bit pattern is below 10p
                case NK_LOADC: vb[i] = 1; db[i] = 1; break;
                case NK_FMUL:
                case NK_DOT2: {
                    std::vector<int> ops{nd.a, nd.b};
                    long sv = (long)vb[nd.a] * vb[nd.b];
                    int sd = db[nd.a] * db[nd.b];
                    if (nd.kind == NK_DOT2) {
                        ops.push_back(nd.c); ops.push_back(nd.d);
                        sv += (long)vb[nd.c] * vb[nd.d]; sd += db[nd.c] * db[nd.d];
                    }
                    if (sv > (narrow[i] ? 2500L : MUL_VB_LIMIT)) treated(widest(ops, vb, lin_reduce), lin_reduce, n_reduce);
                    else if (sd > MUL_DB_LIMIT) treated(widest(ops, db, lin_norm), lin_norm, n_norm);
                    vb[i] = 1 + (int)((sv + 2519) / 2520); db[i] = 1;
                    break;
                }
                case NK_LIN: {
                    int sv = 0, sd = 0;
                    std::vector<int> ops;
                    for (auto& t : nd.terms) { sv += vb[t.node]; sd += db[t.node]; ops.push_back(t.node); }
                    if (sv > SUM_VB_LIMIT) treated(widest(ops, vb, lin_reduce), lin_reduce, n_reduce);
                    else if (sd > SUM_DB_LIMIT) treated(widest(ops, db, lin_norm), lin_norm, n_norm);
                    else if (succ[i].size() >= 2) {
                        if (sv >= VB_EAGER && !lin_reduce[i]) { lin_reduce[i] = 1; n_reduce++; }
                        else if (sd >= DB_EAGER && !lin_norm[i] && !lin_reduce[i]) { lin_norm[i] = 1; n_norm++; }
                    }
                    vb[i] = lin_reduce[i] ? 2 : sv;
                    db[i] = (lin_reduce[i] || lin_norm[i]) ? 1 : sd;
                    break;
                }
                case NK_STOREOUT:
                    // a store handles (-p, 2p) with normalised digits: what multiplications produce
                    if (N[nd.a].kind != NK_FMUL && N[nd.a].kind != NK_DOT2)
                        throw std::runtime_error("scheduler: stored value must come from a multiplication");
                    if (vb[nd.a] > 2 && !narrow[nd.a]) { narrow[nd.a] = 1; redo = nd.a; }
                    break;
            }
            i = redo >= 0 ? redo : i + 1;
        }
    }
    n_norm = 0;
    for (int i = 0; i < n; i++) if (lin_norm[i] && lin_reduce[i]) lin_norm[i] = 0;  // a reduction sweeps anyway
    for (int i = 0; i < n; i++) n_norm += lin_norm[i];
    P.n_reduce = n_reduce; P.n_norm = n_norm;

    // ---- pass 1: list scheduling into K-wide single-class steps
    auto env_int2 = [](const char* nm, int dflt) { const char* v = getenv(nm); return v ? atoi(v) : dflt; };
    const int WINDOW = env_int2("BLS381_WINDOW", 600);
    const bool split_mul = env_int2("BLS381_SPLIT_MUL", 0) != 0;
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
        if ((int)cand[OC_FMUL].size() >= K) cls = OC_FMUL;
        else if (!cand[OC_LIN].empty()) cls = OC_LIN;
        else if (!cand[OC_COPY].empty()) cls = OC_COPY;
        else if (!cand[OC_FMUL].empty()) cls = OC_FMUL;
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
        if ((int)c.size() > K) {
            if (cls == OC_LIN) {
                // keep the most urgent op, fill the step with the ops closest to it in cost
                int lead = c[0];
                auto cost = [&](int x) { return (int)N[x].terms.size() * 20 + lin_norm[x] * 40 + lin_reduce[x] * 100; };
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

    auto emit_copy_steps = [&](std::vector<CopyOp>& ops) {
        for (size_t base = 0; base < ops.size(); base += K) {
            StepHdr h{ST_COPY, 0};
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
        ops.clear();
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
                        d.w[11] = (uint16_t)(victim | 0x8000); d.w[12] = (uint16_t)t;
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
                d.w[9] = (uint16_t)(sl | 0x8000); d.w[10] = (uint16_t)(f.src | ((uint32_t)f.kind << 12));
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
        // 4. the step itself
        StepHdr h; h.max_terms = 0;
        int step_rounds = 0;
        bool any_dot2 = false;
        for (int op : st.ops) any_dot2 |= (N[op].kind == NK_DOT2);
        // a step holding a two-product op runs the two-product multiplier on every lane; single
        // products in it use zero for the second product
        h.type = st.cls == OC_FMUL ? (any_dot2 ? ST_DOT2 : ST_FMUL) : (st.cls == OC_LIN ? ST_LIN : ST_COPY);
        for (int r = 0; r < K; r++) {
            OpDesc d; std::memset(&d, 0, sizeof d);
            if (st.cls != OC_COPY) {  // idle lane / padding defaults
                d.w[0] = (uint16_t)TRASH_SLOT;
                for (int t = 1; t <= 8; t++) d.w[t] = (uint16_t)ZERO_SLOT;
                if (r < (int)cur_spills.size()) {
                    d.w[11] = (uint16_t)(cur_spills[r].first | 0x8000); d.w[12] = (uint16_t)cur_spills[r].second;
                    P.n_piggy_spill++;
                }
            }
            if (r < (int)st.ops.size()) {
                int op = st.ops[r];
                const Node& nd = N[op];
                d.hdr = 1u;
                switch (nd.kind) {
                    case NK_FMUL:
                        d.w[0] = (uint16_t)dst_slot[op]; d.w[1] = (uint16_t)src_slot[nd.a]; d.w[2] = (uint16_t)src_slot[nd.b];
                        P.n_fmul++;
                        break;
                    case NK_DOT2:
                        d.hdr |= 2u;  // second product present
                        if (nd.neg_d) d.hdr |= 4u;  // subtract it
                        d.w[0] = (uint16_t)dst_slot[op]; d.w[1] = (uint16_t)src_slot[nd.a]; d.w[2] = (uint16_t)src_slot[nd.b];
                        d.w[3] = (uint16_t)src_slot[nd.c]; d.w[4] = (uint16_t)src_slot[nd.d];
                        P.n_dot2++;
                        break;
                    case NK_LIN: {
                        int nt = (int)nd.terms.size();
                        assert(nt <= LIN_MAX_TERMS);
                        d.hdr |= (uint32_t)nt << 8;
                        d.w[0] = (uint16_t)dst_slot[op];
                        for (int t = 0; t < nt; t++)
                            d.w[1 + t] = (uint16_t)(src_slot[nd.terms[t].node] | (nd.terms[t].neg ? 0x8000 : 0));
                        int nneg = 0;
                        for (auto& t : nd.terms) nneg += t.neg;
                        d.hdr |= (uint32_t)nneg << 24;
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
        prev_open = st.cls != OC_COPY;
        prev_idx = P.hdrs.size() - 1;
        prev_results.clear();
        for (int op : st.ops) if (N[op].kind != NK_STOREOUT) prev_results.insert(op);
        prev_nfill = 0; prev_nspill = (int)cur_spills.size();
        prev_spilled.clear();
        for (auto& cs : cur_spills) prev_spilled.insert(cs.second);
        if (st.cls == OC_FMUL && any_dot2) P.n_dot2_steps++;
        else if (st.cls == OC_FMUL) P.n_fmul_steps++;
        else if (st.cls == OC_LIN) P.n_lin_steps++;
        else P.n_copy_steps++;
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
