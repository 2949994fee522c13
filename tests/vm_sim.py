"""Big-integer interpreter for the scheduled step programs (test infrastructure).

Executes the binary program returned by bls381_debug_program exactly as the GPU kernel's
contract states: per step every lane reads its operands, then every lane writes.  Used to
check the host-side scheduler and slot allocator against the oracle without a GPU.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "py"))
import bls381_ref as R
from plonky2_bls12_381_pairing_amd import _lib

P = R.P
RBITS = 392                       # the interpreter's Montgomery factor: 14 digits of 28 bits
PINV = pow(P, -1, 1 << RBITS)
MUL_LIMIT = (1 << RBITS) // P     # 2520: operand-bound products a multiplication tolerates
ST_FMUL, ST_LIN, ST_COPY, ST_DOT2, ST_DOTN = 0, 1, 2, 3, 4
CK_IN, CK_FILL, CK_CONST, CK_SPILL, CK_OUT = range(5)
N_CONST = 21
DEBUG = bool(os.environ.get('VM_SIM_DEBUG'))
PROV = {}
TPROV = {}
DW = 12                           # 32-bit words per host-side descriptor (OpDesc: hdr + 22 half-words)


def get_program(which, K=8, lds_slots=0):
    info = (C.c_uint32 * 8)()
    _lib.check(_lib.lib.bls381_debug_program(which.encode(), K, lds_slots, None, None, 0, info))
    nsteps = info[1]
    hdrs = (C.c_uint32 * (2 * nsteps))()
    descs = (C.c_uint32 * (DW * nsteps * K))()
    _lib.check(_lib.lib.bls381_debug_program(which.encode(), K, lds_slots, hdrs, descs, nsteps, info))
    return dict(K=K, nsteps=nsteps, lds_slots=info[2], tmp_slots=info[3], n_in=info[4], n_out=info[5],
                n_fmul=info[6], n_lin=info[7], hdrs=list(hdrs), descs=list(descs))


def get_constants():
    buf = (C.c_uint32 * (16 * N_CONST))()
    _lib.check(_lib.lib.bls381_debug_constants(buf, N_CONST))
    out = []
    for i in range(N_CONST):
        row = buf[16 * i:16 * i + 16]
        assert all(d < 1 << 28 for d in row[:14]) and row[14] == 0 and row[15] == 0
        out.append(sum(d << (28 * k) for k, d in enumerate(row[:14])))
    return out


MASK = (1 << 28) - 1
P28 = [(P >> (28 * i)) & MASK for i in range(14)]
N28 = (-pow(P, -1, 1 << 28)) % (1 << 28)


def to_digits(v):
    assert 0 <= v < 1 << RBITS
    return [(v >> (28 * i)) & MASK for i in range(14)]


def value(d):
    return sum(x << (28 * i) for i, x in enumerate(d))


def s32(x):
    """What a 32-bit register holding x means as a signed number; the kernel's digits are those."""
    assert -(1 << 31) <= x < (1 << 31), "digit leaves 32 bits"
    return x


def mulacc(a, b, c=None, d=None):
    """The kernel's column-wise Montgomery product sum.  The result is determined by its value (digits
    0..12 normalised, signed top digit), so it is computed on whole numbers; the 64-bit column
    accumulator is checked through the worst column: 14 products per operand pair, 14 reduction
    products and the carry."""
    worst = 14 * max(abs(x) for x in a) * max(abs(x) for x in b) + 14 * MASK * MASK + (1 << 36)
    t = value(a) * value(b)
    if c is not None:
        worst += 14 * max(abs(x) for x in c) * max(abs(x) for x in d)
        t += value(c) * value(d)
    assert worst < 1 << 63, "column accumulator leaves 64 bits"
    m = (-t * PINV) % (1 << RBITS)
    r = (t + m * P) >> RBITS
    return [(r >> (28 * i)) & MASK for i in range(13)] + [s32(r >> (28 * 13))]


def mulacc_columns(a, b, c=None, d=None):
    """Digit-for-digit form of mulacc (what the kernel executes); the tests cross-check the two."""
    m = [0] * 14
    r = [0] * 14
    acc = 0

    def chk():
        assert -(1 << 63) <= acc < (1 << 63), "column accumulator leaves 64 bits"
    for k in range(14):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
            if c is not None:
                acc += c[i] * d[k - i]
        for j in range(k):
            acc += m[j] * P28[k - j]
        m[k] = ((acc & 0xffffffff) * N28) & MASK
        acc += m[k] * P28[0]
        chk()
        assert acc & MASK == 0
        acc >>= 28
    for k in range(14, 27):
        for i in range(k - 13, 14):
            acc += a[i] * b[k - i]
            if c is not None:
                acc += c[i] * d[k - i]
            acc += m[i] * P28[k - i]
        chk()
        r[k - 14] = acc & MASK
        acc >>= 28
    r[13] = s32(acc)
    return r


def normalise(d):
    d = list(d)
    for i in range(13):
        carry = d[i] >> 28
        d[i] &= MASK
        d[i + 1] = s32(d[i + 1] + carry)
    return d


def reduce(d):
    d = list(d)
    q = (d[13] * 40323) >> 32
    c = 0
    for i in range(14):
        c += q * P28[i]
        d[i] = s32(d[i] - (c & MASK if i < 13 else c))   # the top digit takes what is left, sign included
        c >>= 28
    return normalise(d)


def canon(d):
    v = value(d)
    assert all(0 <= x <= MASK for x in d[:13]), "stored value is not normalised"
    assert -P < v < 2 * P, "stored value outside (-p, 2p)"
    return v + P if v < 0 else (v - P if v >= P else v)


def run_program(prog, gin, strict=True, extra_consts=()):
    """gin: list of ints in boundary form (12 x u32 patterns), one per input slot; extra_consts: values
    (times 2^392 mod p) that follow the built-in constant table.  Returns gout list."""
    K = prog["K"]
    consts = [to_digits(c) for c in list(get_constants()) + list(extra_consts)]
    lds = [None] * prog["lds_slots"]
    lds[prog["lds_slots"] - 2] = [0] * 14      # the ZERO slot; the last slot is TRASH
    gtmp = {}
    gout = [None] * prog["n_out"]
    hd, ds = prog["hdrs"], prog["descs"]
    for s in range(prog["nsteps"]):
        typ, mt = hd[2 * s], hd[2 * s + 1]
        writes = []
        late = []                                   # carried fills: LDS written after the step's results
        if typ != ST_COPY:
            halves = lambda r: [h for x in ds[(s * K + r) * DW + 1:(s * K + r + 1) * DW] for h in (x & 0xffff, x >> 16)]
            carried = [halves(r)[13:17] for r in range(K)]     # w[13], w[14] fill; w[15], w[16] spill
            any_fill = any_spill = False
            stored = set()
            for w13, w14, w15, w16 in carried:               # the fills' loads first
                if w13 & 0x8000:
                    any_fill = True
                    kind, src = w14 >> 12, w14 & 0xfff
                    assert src < 1024
                    if kind == CK_IN:
                        val = to_digits(gin[src])
                    elif kind == CK_FILL:
                        val = gtmp[src]
                        stored.add(src)
                    else:
                        assert kind == CK_CONST
                        val = consts[src]
                    late.append((w13 & 0x7fff, val))
                    if DEBUG and os.environ.get('VM_SIM_DEBUG') == "fill%d" % s:
                        print("step", s, "fill to slot", w13 & 0x7fff, "kind", kind, "src", src, "min/max digit %.2f %.2f" % (min(val) / 2 ** 28, max(val) / 2 ** 28), "tmp written", TPROV.get(src) if kind == CK_FILL else None)
            for w13, w14, w15, w16 in carried:               # then the spills: LDS as the step finds it
                if w15 & 0x8000:
                    any_spill = True
                    assert lds[w15 & 0x7fff] is not None and w16 < 128
                    assert w16 not in stored, "a step fills what it spills"
                    gtmp[w16] = lds[w15 & 0x7fff]
                    TPROV[w16] = (s, "carried", w15 & 0x7fff, PROV.get(w15 & 0x7fff))
            assert bool(mt & 0x10000) == any_fill and bool(mt & 0x20000) == any_spill, "step-wide carry flags are wrong"
        zero_slot = prog["lds_slots"] - 2
        for r in range(K):
            d = ds[(s * K + r) * DW:(s * K + r + 1) * DW]
            w = []
            for x in d[1:]:
                w += [x & 0xffff, x >> 16]
            if not (d[0] & 1):
                if typ != ST_COPY:  # idle lanes run branch-free on ZERO and store to TRASH
                    assert w[0] == prog["lds_slots"] - 1 and all(x == zero_slot for x in w[1:13])
                    if typ == ST_DOTN:
                        assert all(x == zero_slot for x in w[17:21])
                continue
            if typ in (ST_FMUL, ST_DOT2):
                width = mt & 0xff
                assert 1 <= width <= 3
                nops = 4 if typ == ST_DOT2 else 2

                def operand(j):
                    """Signed sum of the operand's terms, as the multiplier forms it in registers."""
                    acc = [0] * 14
                    for t in range(3):
                        e = w[1 + 3 * j + t]
                        v = lds[e & 0x7fff]
                        assert v is not None, "multiplier reads an empty slot"
                        if t >= width:
                            assert e == zero_slot, "term beyond the step's operand width"
                            continue
                        neg = bool(e & 0x8000)
                        if width == 1 and j != 3:
                            assert not neg, "the short path only looks at d's sign"
                        acc = [s32(x - y if neg else x + y) for x, y in zip(acc, v)]
                    return acc
                ops = [operand(j) for j in range(nops)]
                if DEBUG:
                    worst = sum(max(abs(x) for x in ops[2 * q]) * max(abs(x) for x in ops[2 * q + 1]) for q in range(nops // 2))
                    if 14 * worst + 14 * MASK * MASK + (1 << 36) >= 1 << 63:
                        print("step", s, "lane", r, "width", width, "terms", [hex(x) for x in w[1:13]])
                        for j in range(nops):
                            for t in range(width):
                                e = w[1 + 3 * j + t]
                                print("  op", j, "term", t, "slot", e & 0x7f, "neg", bool(e & 0x8000), "written by", PROV.get(e & 0x7f), "max digit %.2f" % (max(abs(x) for x in lds[e & 0x7f]) / 2 ** 28))
                if typ == ST_DOT2 and not d[0] & 2:
                    assert value(ops[2]) == 0 and value(ops[3]) == 0      # single product: padded with ZERO
                assert all(abs(value(x)) < 2048 * P for x in ops), "multiplier operand too wide"
                t = value(ops[0]) * value(ops[1]) + (value(ops[2]) * value(ops[3]) if nops == 4 else 0)
                out = mulacc(*ops)
                assert (value(out) << RBITS) % P == t % P and abs(value(out)) < 2048 * P
                writes.append(("lds", w[0], out))
            elif typ == ST_DOTN:
                npairs = mt & 0xff
                assert 3 <= npairs <= 8
                idx = [1 + 2 * t if t < 6 else 17 + 2 * (t - 6) for t in range(8)]
                t_val, worst, n_real = 0, 14 * MASK * MASK + (1 << 36), 0
                for t in range(8):
                    ea, eb = w[idx[t]], w[idx[t] + 1]
                    if t >= npairs:
                        assert ea == zero_slot and eb == zero_slot, "product beyond the step's count"
                        continue
                    assert not ea & 0x8000
                    a, b = lds[ea & 0x7f], lds[eb & 0x7f]
                    assert a is not None and b is not None, "multiplier reads an empty slot"
                    assert abs(value(a)) < 2048 * P and abs(value(b)) < 2048 * P, "multiplier operand too wide"
                    worst += 14 * max(abs(x) for x in a) * max(abs(x) for x in b)
                    t_val += -value(a) * value(b) if eb & 0x8000 else value(a) * value(b)
                    n_real += (ea != zero_slot)
                assert worst < 1 << 63, "column accumulator leaves 64 bits"
                assert n_real >= 1
                m = (-t_val * PINV) % (1 << RBITS)
                rr = (t_val + m * P) >> RBITS
                assert abs(rr) < 2048 * P
                out = [(rr >> (28 * i)) & MASK for i in range(13)] + [s32(rr >> (28 * 13))]
                writes.append(("lds", w[0], out))
            elif typ == ST_LIN:
                def lin(nt, B, terms, dst):
                    B -= 16 if B >= 8 else 0
                    assert B == 0 or mt & 0x300, "a sum around a shifted zero must be swept"
                    acc = [B << 28] + [(B << 28) - B] * 12 + [-B]        # a digit vector of value 0
                    assert value(acc) == 0
                    for t in range(len(terms)):
                        e = terms[t]
                        v = lds[e & 0x7f]
                        assert v is not None, "LIN reads an empty slot"
                        coef = (e >> 7) & 0xff
                        coef -= 256 if coef >= 128 else 0
                        if t >= nt:
                            assert e == zero_slot               # padding: ZERO with coefficient 0
                            continue
                        assert coef != 0
                        acc = [s32(x + coef * y) for x, y in zip(acc, v)]
                    assert abs(value(acc)) < MUL_LIMIT * P
                    if mt & 0x200:
                        assert all(abs(x) <= 7 << 28 for x in acc[:13]), "digit too wide for the reduction"
                        acc = reduce(acc)                # sweeps as well
                        assert -P < value(acc) < 2 * P
                    elif mt & 0x100:
                        acc = normalise(acc)
                    writes.append(("lds", dst, acc))
                nt = (d[0] >> 8) & 0xff
                assert nt <= (mt & 0xff) <= 7
                lin(nt, (d[0] >> 24) & 0xf, w[1:1 + (mt & 0xff)], w[0])
            else:
                kind = (d[0] >> 8) & 0xff
                assert kind == (mt & 0xff)  # the kernel takes the kind of a copy step from the step's header
                if kind == CK_IN:
                    assert gin[w[1]] < 1 << 384
                    writes.append(("lds", w[0], to_digits(gin[w[1]])))
                elif kind == CK_FILL:
                    writes.append(("lds", w[0], gtmp[w[1]]))
                elif kind == CK_CONST:
                    writes.append(("lds", w[0], consts[w[1]]))
                elif kind == CK_SPILL:
                    assert lds[w[1]] is not None
                    writes.append(("tmp", w[0], lds[w[1]]))
                else:
                    assert lds[w[1]] is not None
                    writes.append(("out", w[0], canon(lds[w[1]])))
        for idx, v in late:
            assert all(not (w == "lds" and i == idx) for w, i, _ in writes), "carried fill lands on a result of its carrier"
            writes.append(("lds", idx, v))
        seen = set()
        for where, idx, v in writes:
            assert (where, idx) not in seen, "two lanes write one slot in one step"
            seen.add((where, idx))
            if where == "lds":
                lds[idx] = v
                if DEBUG:
                    PROV[idx] = (s, typ, hex(mt))
            elif where == "tmp":
                gtmp[idx] = v
                TPROV[idx] = (s, "copy step")
            else:
                gout[idx] = v
    return gout
