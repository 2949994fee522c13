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
RINV = pow(1 << 384, -1, P)
ST_FMUL, ST_LIN, ST_COPY, ST_DOT2 = 0, 1, 2, 3
CK_IN, CK_FILL, CK_CONST, CK_SPILL, CK_OUT = range(5)


def get_program(which, K=8, lds_slots=0):
    info = (C.c_uint32 * 8)()
    _lib.check(_lib.lib.bls381_debug_program(which.encode(), K, lds_slots, None, None, 0, info))
    nsteps = info[1]
    hdrs = (C.c_uint32 * (2 * nsteps))()
    descs = (C.c_uint32 * (8 * nsteps * K))()
    _lib.check(_lib.lib.bls381_debug_program(which.encode(), K, lds_slots, hdrs, descs, nsteps, info))
    return dict(K=K, nsteps=nsteps, lds_slots=info[2], tmp_slots=info[3], n_in=info[4], n_out=info[5],
                n_fmul=info[6], n_lin=info[7], hdrs=list(hdrs), descs=list(descs))


def get_constants():
    n = 19
    buf = (C.c_uint32 * (12 * n))()
    _lib.check(_lib.lib.bls381_debug_constants(buf, n))
    return [R.u32_limbs_to_int(buf[12 * i:12 * i + 12]) for i in range(n)]


def run_program(prog, gin, strict=True):
    """gin: list of ints (Montgomery residues, one per input slot).  Returns gout list."""
    K = prog["K"]
    consts = get_constants()
    lds = [None] * prog["lds_slots"]
    lds[prog["lds_slots"] - 2] = 0            # the ZERO slot; the last slot is TRASH
    gtmp = {}
    gout = [None] * prog["n_out"]
    hd, ds = prog["hdrs"], prog["descs"]
    for s in range(prog["nsteps"]):
        typ, mt = hd[2 * s], hd[2 * s + 1]
        writes = []
        for r in range(K):
            d = ds[(s * K + r) * 8:(s * K + r) * 8 + 8]
            if not (d[0] & 1):
                if typ != ST_COPY:  # idle lanes run branch-free on ZERO and store to TRASH
                    assert d[1] & 0xffff == prog["lds_slots"] - 1
                    assert all(x == (prog["lds_slots"] - 2) * 0x10001 for x in d[2:]) and d[1] >> 16 == prog["lds_slots"] - 2
                continue
            w = []
            for x in d[1:]:
                w += [x & 0xffff, x >> 16]
            if typ == ST_FMUL:
                a, b = lds[w[1]], lds[w[2]]
                assert a is not None and b is not None, "FMUL reads an empty slot"
                assert a * b < P << 384, "FMUL operands exceed the Montgomery bound"
                writes.append(("lds", w[0], a * b * RINV % P))
            elif typ == ST_DOT2:
                a, b = lds[w[1]], lds[w[2]]
                c2, d2 = lds[w[3]], lds[w[4]]
                assert None not in (a, b, c2, d2), "DOT2 reads an empty slot"
                if not d[0] & 2:
                    assert c2 == 0 and d2 == 0          # single product: padded with ZERO
                if d[0] & 4:
                    kd = (d[0] >> 16) & 0xff
                    assert d2 <= kd * P, "negated operand exceeds its bound"
                    d2 = kd * P - d2
                assert a * b + c2 * d2 < P << 384, "DOT2 operands exceed the Montgomery bound"
                writes.append(("lds", w[0], (a * b + c2 * d2) * RINV % P))
            elif typ == ST_LIN:
                nt = (d[0] >> 8) & 0xff
                kb = (d[0] >> 16) & 0xff
                assert nt <= (mt & 0xff) and kb <= 16
                acc = kb * P
                for t in range(mt & 0xff):
                    v = lds[w[1 + t] & 0x7fff]
                    if t >= nt:
                        assert v == 0 and not w[1 + t] & 0x8000
                    assert v is not None, "LIN reads an empty slot"
                    acc += -v if w[1 + t] & 0x8000 else v
                assert 0 <= acc < 1 << 416, "13-limb accumulator out of range"
                rounds = (mt >> 8) & 0xff
                assert rounds <= 5
                for j in range(rounds - 1, -1, -1):         # the kernel's ladder 2^(r-1) p ... p
                    if acc >= (P << j):
                        acc -= P << j
                assert acc < 1 << 384, "unreduced value does not fit 12 limbs"
                writes.append(("lds", w[0], acc))
            else:
                kind = (d[0] >> 8) & 0xff
                if kind == CK_IN:
                    writes.append(("lds", w[0], gin[w[1]]))
                elif kind == CK_FILL:
                    writes.append(("lds", w[0], gtmp[w[1]]))
                elif kind == CK_CONST:
                    writes.append(("lds", w[0], consts[w[1]]))
                elif kind == CK_SPILL:
                    assert lds[w[1]] is not None
                    writes.append(("tmp", w[0], lds[w[1]]))
                else:
                    assert lds[w[1]] is not None and lds[w[1]] < P, "stored value is not canonical"
                    writes.append(("out", w[0], lds[w[1]]))
        seen = set()
        for where, idx, v in writes:
            assert (where, idx) not in seen, "two lanes write one slot in one step"
            seen.add((where, idx))
            if where == "lds":
                lds[idx] = v
            elif where == "tmp":
                gtmp[idx] = v
            else:
                gout[idx] = v
    return gout
