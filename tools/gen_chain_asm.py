#!/usr/bin/env python3
"""Generates plonky2_bls12_381_pairing_amd/csrc/cyc_chain_asm.h: the arithmetic of one round of the squaring-chain
kernel (cyc_chain.h) as gfx950 assembly with every register assigned here.

Why assembly: the round is four two-product Montgomery multiplications and four sums on 56 digits of state, and it
has to fit the 168 registers that three waves per SIMD leave a lane.  Compiled from C++ the same code takes 230-244
registers (or spills 150-550 of them when told to fit) and chains no accumulator across columns (one 64-bit addition
per column more).  Here a multiplication is 588 multiply-accumulates + 69 other instructions.

The instruction list is built once and used twice: printed as assembly, and run by a small interpreter on Python
integers (`--check`), which compares a round against the Granger-Scott formulas on big integers, so register
assignment mistakes show up without a GPU.

Register file (a lane): v0..v63 the Fq4 pair (x.c0, x.c1, y.c0, y.c1: 14 digits each at 16-register strides),
v64..v127 the four products T.c0, T.c1, XY.c0, XY.c1 likewise, v128..v159 scratch; the two spare registers behind
every value are scratch too.  s40..s53 the digits of p, s54 = -1/p mod 2^28, s55 = 40323 (quotient estimate).
"""
import argparse
import os
import random
import sys

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
D = 14
MASK = (1 << 28) - 1
PD = [(P >> (28 * i)) & MASK for i in range(D)]
N28 = (-pow(P, -1, 1 << 28)) % (1 << 28)
QEST = 40323

X0, X1, Y0, Y1, T0, T1, XY0, XY1 = 0, 16, 32, 48, 64, 80, 96, 112
SP, SN, SQ = 40, 54, 55


def val(base):
    return [base + i for i in range(D)]


class Prog:
    def __init__(self):
        self.ins = []

    def emit(self, *t):
        self.ins.append(t)

    # ---- instruction constructors (dst first; 64-bit values are named by their low register, which is even)
    def mad_i64(self, acc, a, b, src=None):      # acc = (i32)a * (i32)b + (src or acc); b: 'v', ('s', n) or ('c', k)
        self.emit("mad_i64", acc, a, b, acc if src is None else src)

    def mad_u64(self, acc, a, b):                # acc = (u32)a * (u32)b + acc
        self.emit("mad_u64", acc, a, b, acc)

    def text(self):
        out = []

        def src(o):
            if isinstance(o, tuple):
                return ("s%d" % o[1]) if o[0] == "s" else str(o[1])
            return "v%d" % o

        def pair(r):
            return "0" if r == "zero" else "v[%d:%d]" % (r, r + 1)
        for t in self.ins:
            op = t[0]
            if op == "mad_i64":
                out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (pair(t[1]), src(t[2]), src(t[3]), pair(t[4])))
            elif op == "mad_u64":
                out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (pair(t[1]), src(t[2]), src(t[3]), pair(t[4])))
            elif op == "mul_lo":
                out.append("v_mul_lo_u32 v%d, %s, %s" % (t[1], src(t[2]), src(t[3])))
            elif op == "mul_hi_i32":
                out.append("v_mul_hi_i32 v%d, %s, %s" % (t[1], src(t[2]), src(t[3])))
            elif op == "and_mask":
                out.append("v_and_b32_e32 v%d, 0xfffffff, v%d" % (t[1], t[2]))
            elif op == "ashr64":
                out.append("v_ashrrev_i64 %s, 28, %s" % (pair(t[1]), pair(t[1])))
            elif op == "sub":                     # dst = a - b (a may be the constant 0)
                out.append("v_sub_u32_e32 v%d, %s, v%d" % (t[1], src(t[2]), t[3]))
            elif op == "add":
                out.append("v_add_u32_e32 v%d, v%d, v%d" % (t[1], t[2], t[3]))
            elif op == "lshl1_add":               # dst = (a << 1) + b
                out.append("v_lshl_add_u32 v%d, v%d, 1, v%d" % (t[1], t[2], t[3]))
            elif op == "mov":
                out.append("v_mov_b32_e32 v%d, v%d" % (t[1], t[2]))
            elif op == "mul_i24":                 # dst = k * a (small values)
                out.append("v_mul_i32_i24_e32 v%d, %s, v%d" % (t[1], src(t[2]), t[3]))
            elif op == "mad_i24":                 # dst = k * a + c
                out.append("v_mad_i32_i24 v%d, v%d, %s, v%d" % (t[1], t[3], src(t[2]), t[4]))
            else:
                raise ValueError(op)
        return out

    # ---- the interpreter: v: dict register -> u32, s: dict -> u32
    def run(self, v, s):
        def u(x):
            return x & 0xFFFFFFFF

        def i32(x):
            x &= 0xFFFFFFFF
            return x - (1 << 32) if x >> 31 else x

        def get(o):
            if isinstance(o, tuple):
                return u(s[o[1]]) if o[0] == "s" else u(o[1])
            return v[o]

        def get64(r):
            return 0 if r == "zero" else (v[r] | (v[r + 1] << 32))

        def put64(r, x):
            x &= (1 << 64) - 1
            v[r] = x & 0xFFFFFFFF
            v[r + 1] = x >> 32
        written = set()
        for t in self.ins:
            op = t[0]
            if op == "mad_i64":
                assert t[1] % 2 == 0
                put64(t[1], i32(get(t[2])) * i32(get(t[3])) + get64(t[4]))
            elif op == "mad_u64":
                put64(t[1], get(t[2]) * get(t[3]) + get64(t[4]))
            elif op == "mul_lo":
                v[t[1]] = u(get(t[2]) * get(t[3]))
            elif op == "mul_hi_i32":
                v[t[1]] = u((i32(get(t[2])) * i32(get(t[3]))) >> 32)
            elif op == "and_mask":
                v[t[1]] = v[t[2]] & MASK
            elif op == "ashr64":
                x = get64(t[1])
                if x >> 63:
                    x -= 1 << 64
                put64(t[1], x >> 28)
            elif op == "sub":
                v[t[1]] = u(get(t[2]) - v[t[3]])
            elif op == "add":
                v[t[1]] = u(v[t[2]] + v[t[3]])
            elif op == "lshl1_add":
                v[t[1]] = u((v[t[2]] << 1) + v[t[3]])
            elif op == "mov":
                v[t[1]] = v[t[2]]
            elif op == "mul_i24":
                v[t[1]] = u(i32(get(t[2])) * i32(v[t[3]]))
            elif op == "mad_i24":
                v[t[1]] = u(i32(get(t[2])) * i32(v[t[3]]) + i32(v[t[4]]))
            else:
                raise ValueError(op)
            written.add(t[1])
        return written


def dot2(p, a, b, c, d, r, m, acc, sp=SP, sn=SN):
    """r = (a b + c d) / 2^392 + (multiple of p) / 2^392, as fp28_mulacc<true> (fp28_dev.h): one accumulator, shifted
    down 28 bits per column.  c = d = None: the single product a b.  sp, sn: the scalar registers that hold the digits
    of p and -1/p mod 2^28."""
    first = True
    for k in range(D):
        for i in range(k + 1):
            p.mad_i64(acc, a[i], b[k - i], "zero" if first else None)
            first = False
            if c is not None:
                p.mad_i64(acc, c[i], d[k - i])
        for j in range(k):
            p.mad_u64(acc, m[j], ("s", sp + k - j))
        p.emit("mul_lo", m[k], acc, ("s", sn))
        p.emit("and_mask", m[k], m[k])
        p.mad_u64(acc, m[k], ("s", sp))
        p.emit("ashr64", acc)
    for k in range(D, 2 * D - 1):
        for i in range(k - D + 1, D):
            p.mad_i64(acc, a[i], b[k - i])
            if c is not None:
                p.mad_i64(acc, c[i], d[k - i])
            p.mad_u64(acc, m[i], ("s", sp + k - i))
        p.emit("and_mask", r[k - D], acc)
        p.emit("ashr64", acc)
    p.emit("mov", r[D - 1], acc)


# scratch of the squaring block
E0 = [128 + i for i in range(D)]
E1 = [142 + i for i in range(D)]
M = [156, 157, 158, 159, 14, 15, 30, 31, 46, 47, 62, 63, 78, 79]
ACC = 126


def square_block():
    """T = (x - y)(x - xi y), XY = x y.  x - xi y = (d.c0 + y.c1, d.c1 - y.c0) with d = x - y."""
    p = Prog()
    x0, x1, y0, y1 = val(X0), val(X1), val(Y0), val(Y1)
    d0, d1 = val(XY0), val(XY1)       # (the products XY are formed last, in the registers d leaves)
    for i in range(D):
        p.emit("sub", d0[i], x0[i], y0[i])
        p.emit("sub", d1[i], x1[i], y1[i])
        p.emit("add", E0[i], d0[i], y1[i])
        p.emit("sub", E1[i], d1[i], y0[i])
    dot2(p, d0, E1, d1, E0, val(T1), M, ACC)
    for i in range(D):
        p.emit("sub", d1[i], ("c", 0), d1[i])
    dot2(p, d0, E0, d1, E1, val(T0), M, ACC)
    nx1 = E0
    for i in range(D):
        p.emit("sub", nx1[i], ("c", 0), x1[i])
    # XY.c0 lands in d0's registers while a, b, c, d = x0, y0, -x1, y1 are read: no overlap
    dot2(p, x0, y0, nx1, y1, val(XY0), M, ACC)
    dot2(p, x0, y1, x1, y0, val(XY1), M, ACC)
    return p


def lin2(p, a, k, own, s, acc, t, q):
    """own = k a + s own - q p, carried; q from the top digits (chain_lin2 in cyc_chain.h)."""
    p.emit("mul_i24", t, ("c", k), a[D - 1])
    p.emit("mad_i24", t, ("c", s), own[D - 1], t)
    p.emit("mul_hi_i32", q, t, ("s", SQ))
    p.emit("sub", q, ("c", 0), q)          # -q
    for i in range(D):
        p.mad_i64(acc, a[i], ("c", k), "zero" if i == 0 else None)
        p.mad_i64(acc, own[i], ("c", s))
        p.mad_i64(acc, q, ("s", SP + i))
        if i < D - 1:
            p.emit("and_mask", own[i], acc)
            p.emit("ashr64", acc)
        else:
            p.emit("mov", own[i], acc)


def sums_block(pair1):
    """even = T + (1 + xi) XY in T's place; then
       pairs 0, 2: x' = 3 even - 2 x, y' = 6 XY + 2 y;   pair 1: x' = 6 xi XY + 2 x, y' = 3 even - 2 y."""
    p = Prog()
    t0, t1, a0, a1 = val(T0), val(T1), val(XY0), val(XY1)
    tmp, acc, tt, q = 128, 130, 132, 133
    for i in range(D):
        p.emit("lshl1_add", tmp, a0[i], t0[i])
        p.emit("sub", t0[i], tmp, a1[i])
        p.emit("lshl1_add", tmp, a1[i], t1[i])
        p.emit("add", t1[i], tmp, a0[i])
    if pair1:
        z0 = [136 + i for i in range(D)]   # xi XY = (XY.c0 - XY.c1, XY.c0 + XY.c1): c0 into scratch, c1 in place
        for i in range(D):
            p.emit("sub", z0[i], a0[i], a1[i])
            p.emit("add", a1[i], a0[i], a1[i])
        lin2(p, z0, 6, val(X0), 2, acc, tt, q)
        lin2(p, a1, 6, val(X1), 2, acc, tt, q)
        lin2(p, t0, 3, val(Y0), -2, acc, tt, q)
        lin2(p, t1, 3, val(Y1), -2, acc, tt, q)
    else:
        lin2(p, t0, 3, val(X0), -2, acc, tt, q)
        lin2(p, t1, 3, val(X1), -2, acc, tt, q)
        lin2(p, a0, 6, val(Y0), 2, acc, tt, q)
        lin2(p, a1, 6, val(Y1), 2, acc, tt, q)
    return p


# ------------------------------------------------------------------ check against big integers
R392 = 1 << 392


def digits(x):
    """non-negative integer below 2^392 -> 14 digits"""
    return [(x >> (28 * i)) & MASK for i in range(D)]


def value(v, base):
    tot = 0
    for i in range(D):
        d = v[base + i]
        if i == D - 1 and d >> 31:
            d -= 1 << 32
        tot += d << (28 * i)
    return tot


def fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def check(rounds=20, seed=1):
    rng = random.Random(seed)
    sq, s02, s1 = square_block(), sums_block(False), sums_block(True)
    sregs = {SP + i: PD[i] for i in range(D)}
    sregs[SN] = N28
    sregs[SQ] = QEST
    frm = lambda x: x * pow(R392, -1, P) % P     # Montgomery form -> value
    for pair1 in (False, True):
        # two lanes stand for two waves: `me` holds the pair that is updated, `other` the pair whose products feed it
        v = {r: rng.getrandbits(32) for r in range(168)}
        cur = [rng.randrange(P) for _ in range(4)]       # x0, x1, y0, y1 of the pair being squared (Montgomery form)
        own = [rng.randrange(P) for _ in range(4)]       # the pair that receives
        for r in range(rounds):
            for base, x in zip((X0, X1, Y0, Y1), cur):
                for i, d in enumerate(digits(x)):
                    v[base + i] = d
            sq.run(v, sregs)
            x = (frm(cur[0]), frm(cur[1]))
            y = (frm(cur[2]), frm(cur[3]))
            xy = fq2_mul(x, y)
            xiy = ((y[0] - y[1]) % P, (y[0] + y[1]) % P)
            t = fq2_mul(((x[0] - y[0]) % P, (x[1] - y[1]) % P), ((x[0] - xiy[0]) % P, (x[1] - xiy[1]) % P))
            got = [frm(value(v, b) % P) for b in (T0, T1, XY0, XY1)]
            assert got == [t[0], t[1], xy[0], xy[1]], ("square block", r)
            for b in (T0, T1, XY0, XY1):
                assert -P < value(v, b) < 2 * P and all(v[b + i] <= MASK for i in range(D - 1))
            # the receiving pair
            for base, xv in zip((X0, X1, Y0, Y1), own):
                for i, d in enumerate(digits(xv)):
                    v[base + i] = d
            (s1 if pair1 else s02).run(v, sregs)
            even = ((t[0] + xy[0] + xy[0] - xy[1]) % P, (t[1] + xy[1] + xy[0] + xy[1]) % P)
            odd = (2 * xy[0] % P, 2 * xy[1] % P)
            ox, oy = (frm(own[0]), frm(own[1])), (frm(own[2]), frm(own[3]))
            if pair1:
                xio = ((odd[0] - odd[1]) % P, (odd[0] + odd[1]) % P)
                want = [(3 * xio[0] + 2 * ox[0]) % P, (3 * xio[1] + 2 * ox[1]) % P, (3 * even[0] - 2 * oy[0]) % P, (3 * even[1] - 2 * oy[1]) % P]
            else:
                want = [(3 * even[0] - 2 * ox[0]) % P, (3 * even[1] - 2 * ox[1]) % P, (3 * odd[0] + 2 * oy[0]) % P, (3 * odd[1] + 2 * oy[1]) % P]
            res = [value(v, b) for b in (X0, X1, Y0, Y1)]
            assert [frm(x % P) for x in res] == want, ("sums block", pair1, r)
            for b, x in zip((X0, X1, Y0, Y1), res):
                assert -P // 100 < x < P + P // 100, ("range", x / P)
                assert all(v[b + i] <= MASK for i in range(D - 1))
            # next round: the updated pair is squared (as a lane's own pair is), a fresh pair receives
            cur = [x % P for x in res]
            # feed the unreduced representative too: the kernel squares what the sums leave
            for base, x in zip((X0, X1, Y0, Y1), res):
                pass
            own = [rng.randrange(P) for _ in range(4)]
    n_sq = len(sq.ins)
    mads = sum(1 for t in sq.ins if t[0].startswith("mad_"))
    print("check ok: squaring block %d instructions (%d multiply-accumulates), sums %d / %d" % (n_sq, mads, len(s02.ins), len(s1.ins)))


def check_unreduced(seed=2):
    """The squaring block on what the sums block leaves (values a little outside [0, p), a signed top digit)."""
    rng = random.Random(seed)
    sq, s02 = square_block(), sums_block(False)
    sregs = {SP + i: PD[i] for i in range(D)}
    sregs[SN] = N28
    sregs[SQ] = QEST
    frm = lambda x: x * pow(R392, -1, P) % P
    v = {r: rng.getrandbits(32) for r in range(168)}
    for base in (X0, X1, Y0, Y1):
        for i, d in enumerate(digits(rng.randrange(P))):
            v[base + i] = d
    for r in range(40):
        before = [value(v, b) for b in (X0, X1, Y0, Y1)]
        keep = {b + i: v[b + i] for b in (X0, X1, Y0, Y1) for i in range(D)}
        sq.run(v, sregs)
        assert all(v[k] == keep[k] for k in keep), "the squaring block changed its inputs"
        x = (frm(before[0] % P), frm(before[1] % P))
        y = (frm(before[2] % P), frm(before[3] % P))
        xy = fq2_mul(x, y)
        t = fq2_mul(((x[0] - y[0]) % P, (x[1] - y[1]) % P), ((x[0] - y[0] + y[1]) % P, (x[1] - y[0] - y[1]) % P))
        assert [frm(value(v, b) % P) for b in (T0, T1, XY0, XY1)] == [t[0], t[1], xy[0], xy[1]], r
        s02.run(v, sregs)      # pair 0: feeds itself
    print("check ok: 40 rounds of a self-feeding pair")


def c_string(lines):
    return "\n".join('    "%s\\n\\t"' % ln for ln in lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if args.check:
        check()
        check_unreduced()
        return
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "plonky2_bls12_381_pairing_amd", "csrc", "cyc_chain_asm.h")
    sq, s02, s1 = square_block(), sums_block(False), sums_block(True)
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_chain_asm.py (which also checks these instruction lists against big integers): do not edit.\n")
        f.write("// One round of the squaring-chain kernel, every register assigned by the generator:\n")
        f.write("//   v0..v63 the Fq4 pair (x.c0, x.c1, y.c0, y.c1, 14 digits each at 16-register strides), v64..v127 the products\n")
        f.write("//   T.c0, T.c1, XY.c0, XY.c1, v128..v159 and the two spare registers behind every value scratch;\n")
        f.write("//   s40..s53 the digits of p, s54 = -1/p mod 2^28, s55 = 40323.\n")
        f.write("#pragma once\n\n")
        f.write("// T = (x - y)(x - xi y), XY = x y: %d instructions, %d of them multiply-accumulates\n" % (len(sq.ins), sum(1 for t in sq.ins if t[0].startswith("mad_"))))
        f.write("#define BLS381_CHAIN_ASM_SQUARE \\\n" + c_string(sq.text()).replace("\n", " \\\n") + "\n\n")
        f.write("// pairs 0 and 2: x' = 3 (T + (1 + xi) XY) - 2 x, y' = 6 XY + 2 y, reduced and carried: %d instructions\n" % len(s02.ins))
        f.write("#define BLS381_CHAIN_ASM_SUMS_02 \\\n" + c_string(s02.text()).replace("\n", " \\\n") + "\n\n")
        f.write("// pair 1: x' = 6 xi XY + 2 x, y' = 3 (T + (1 + xi) XY) - 2 y: %d instructions\n" % len(s1.ins))
        f.write("#define BLS381_CHAIN_ASM_SUMS_1 \\\n" + c_string(s1.text()).replace("\n", " \\\n") + "\n")
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
