"""Host logic without a GPU: the C-ABI library loads and exports every declared symbol, the
scheduler's step programs compute the oracle's values in a big-integer interpreter, and the
product path fails loudly when no GPU is present."""
import ctypes as C
import os
import re
import sys

import pytest

import bls381_ref as R
import helpers as H
import vm_sim as S
from plonky2_bls12_381_pairing_amd import _lib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
M, F = R.to_mont, R.from_mont


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bls381_hip.h")).read()
    declared = set(re.findall(r"\b(bls381_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("bls381_ctx")
    assert declared == set(_lib.EXPORTED)
    for name in declared:
        assert hasattr(_lib.lib, name), name


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_no_cpu_fallback():
    if _has_gpu():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = _lib.lib.bls381_ctx_create(C.byref(ctx), 0)
    assert rc == -2 and not ctx.value                      # BLS381_ERR_NO_DEVICE, no context
    assert b"no HIP device" in _lib.lib.bls381_last_error()
    import plonky2_bls12_381_pairing_amd as B
    with pytest.raises(B.Bls381Error):
        B.Context(0)


def test_argument_errors_need_no_device():
    """Bad arguments are refused before anything touches the GPU: BLS381_ERR_INVALID and a message."""
    L = _lib.lib
    ctx = C.c_void_p()
    assert L.bls381_ctx_create_ex(C.byref(ctx), 0, 7, 0) == -1 and not ctx.value
    assert b"lanes_per_group" in L.bls381_last_error()
    assert L.bls381_ctx_create(None, 0) == -1
    buf = (C.c_uint64 * 72)()
    assert L.bls381_pairing_batch(None, buf, buf, None, 1, buf) == -1
    assert b"ctx is NULL" in L.bls381_last_error()
    assert L.bls381_final_exp_batch(None, buf, 1, buf, None) == -1
    assert L.bls381_pairing_witness_kinds(None, 0) == -1
    info = (C.c_uint32 * 8)()
    assert L.bls381_debug_program(b"no-such-program", 8, 0, None, None, 0, info) == -1
    assert b"unknown program" in L.bls381_last_error()
    assert L.bls381_debug_program(b"fe1", 7, 0, None, None, 0, info) == -1
    L.bls381_ctx_destroy(None)                                # a no-op
    L.bls381_g2_prepared_destroy(None)


def test_constant_table():
    """The interpreter's constants carry the Montgomery factor 2^392 (14 digits of 28 bits)."""
    c = S.get_constants()
    F28 = lambda x: x * pow(1 << 392, -1, R.P) % R.P
    assert F28(c[0]) == 1 and F28(c[1]) == R.TWO_INV
    assert c[17] == pow(2, 400, R.P) and c[18] == pow(2, 384, R.P) and c[19] == pow(2, 784, R.P) and c[20] == 1
    g1, g2 = R.frob_coeffs(1), R.frob_coeffs(2)
    for i in range(1, 6):
        assert (F28(c[2 + 2 * (i - 1)]), F28(c[3 + 2 * (i - 1)])) == R.fq2_pow(g1, i)
        assert F28(c[12 + i - 1]) == R.fq2_pow(g2, i)[0]


def test_simulator_multiplier_forms_agree():
    """The whole-number form of the Montgomery product sum equals the column-wise form the kernel runs,
    signed and unnormalised digits included."""
    import random
    rng = random.Random(5)
    for _ in range(200):
        ops = [[rng.randrange(-(1 << 29), 1 << 29) for _ in range(13)] + [rng.randrange(-(1 << 22), 1 << 22)] for _ in range(4)]
        assert S.mulacc(*ops) == S.mulacc_columns(*ops)
        assert S.mulacc(ops[0], ops[1]) == S.mulacc_columns(ops[0], ops[1])


@pytest.mark.parametrize("K,slots", [(8, 0), (6, 40), (9, 0), (12, 0), (16, 0)])
def test_miller_and_final_exp_programs(K, slots):
    ps, qs = H.random_pairs(1, 100 + K)
    p, q = ps[0], qs[0]
    prog = S.get_program("miller", K, slots)
    assert prog["n_in"] == 6 and prog["n_out"] == 12
    out = S.run_program(prog, [M(p[0]), M(p[1]), M(q[0][0]), M(q[0][1]), M(q[1][0]), M(q[1][1])])
    ml = R.fq12_from_flat_ark([F(x) for x in out])
    assert ml == R.multi_miller_loop([p], [q])
    mid = S.run_program(S.get_program("fe1", K, slots), [M(x) for x in R.fq12_flatten_ark(ml)])
    mid[0] = M(R.fq_inv(F(mid[0])))                       # the Fp inversion kernel's job
    out = S.run_program(S.get_program("fe2", K, slots), mid)
    assert R.fq12_from_flat_ark([F(x) for x in out]) == R.native_pairing([p], [q])
    if K == 8:
        # the loop the pairing calls run (G2 doublings without their halvings): another Miller value, off by a factor in
        # Fq, and the same pairing
        out = S.run_program(S.get_program("millers", K, slots), [M(p[0]), M(p[1]), M(q[0][0]), M(q[0][1]), M(q[1][0]), M(q[1][1])])
        scaled = R.fq12_from_flat_ark([F(x) for x in out])
        assert scaled != ml
        mid = S.run_program(S.get_program("fe1", K, slots), [M(x) for x in R.fq12_flatten_ark(scaled)])
        mid[0] = M(R.fq_inv(F(mid[0])))
        out = S.run_program(S.get_program("fe2", K, slots), mid)
        assert R.fq12_from_flat_ark([F(x) for x in out]) == R.native_pairing([p], [q])


def test_auxiliary_programs():
    a, b = H.random_fq12(2, 17)
    fa, fb = R.fq12_flatten_ark(a), R.fq12_flatten_ark(b)
    run = lambda name, gin: [F(x) for x in S.run_program(S.get_program(name), [M(v) for v in gin])]
    assert R.fq12_from_flat_ark(run("fq12mul", fa + fb)) == R.fq12_mul(a, b)
    assert R.fq12_from_flat_ark(run("hardgen", fa)) == R.hard_part_native(a)
    for pw in (0, 1, 2, 3, 6, 7, 11):
        want = R.myfq12_to_fq12(R.frobenius_map_native(R.myfq12_from_fq12(a), pw))
        assert R.fq12_from_flat_ark(run("frob%d" % pw, fa)) == want
    assert R.fq12_from_flat_ark(run("conj", fa)) == R.fq12_conj(a)
    for exp in (0, 0xb, (5 << 64) | 3):                      # pow_native: the program name carries the exponent words
        name = "pow:" + "".join("%016x" % ((exp >> (64 * i)) & (2 ** 64 - 1)) for i in range(max(1, (exp.bit_length() + 63) // 64)))
        want = R.myfq12_to_fq12(R.pow_native(R.myfq12_from_fq12(a), exp))
        assert R.fq12_from_flat_ark(run(name, fa)) == want
    p = R.g1_mul(R.G1_GEN, 5)
    c = tuple((fb[2 * i], fb[2 * i + 1]) for i in range(3))
    assert R.fq12_from_flat_ark(run("ell", fa + fb[:6] + [p[0], p[1]])) == R.ell(a, c, p)
    q = R.g2_mul(R.G2_GEN, 77)
    got = run("g2prep", [q[0][0], q[0][1], q[1][0], q[1][1]])
    assert got == [x for cf in R.g2_prepare(q) for fq2 in cf for x in fq2]
    # the G2 subgroup test psi(Q) = [z]Q as a program: four zeros for a point of the subgroup, and only for such a point
    import bls381_points as PT
    sub = lambda pt: run("g2sub", [pt[0][0], pt[0][1], pt[1][0], pt[1][1]])
    assert sub(q) == [0, 0, 0, 0] and sub(R.g2_neg(R.G2_GEN)) == [0, 0, 0, 0]
    x0 = 1
    while R.fq2_pow(R.fq2_add(R.fq2_mul(R.fq2_square((x0, 1)), (x0, 1)), (4, 4)), (R.P * R.P - 1) // 2) != R.FQ2_ONE:
        x0 += 1
    e = bytearray((1).to_bytes(48, "big") + x0.to_bytes(48, "big")); e[0] |= 0x80
    status, outside = PT.g2_decompress(bytes(e), check_subgroup=False)
    assert status == PT.OK and PT.g2_decompress(bytes(e))[0] == PT.NOT_IN_SUBGROUP and any(sub(outside))
    # ... and the G1 test (beta x, y) = [-z^2]P
    sub1 = lambda pt: run("g1sub", [pt[0], pt[1]])
    assert sub1(p) == [0, 0] and sub1(R.G1_GEN) == [0, 0]
    x1 = 4
    while R.fq_sqrt((x1 ** 3 + 4) % R.P) is None:
        x1 += 1
    assert any(sub1((x1, R.fq_sqrt((x1 ** 3 + 4) % R.P))))
    # the schedules for four lanes per point (large batches run the two tests that way) give the same answers
    run4 = lambda name, gin: [F(x) for x in S.run_program(S.get_program(name, 4, 0), [M(v) for v in gin])]
    assert run4("g2sub", [q[0][0], q[0][1], q[1][0], q[1][1]]) == [0, 0, 0, 0] and run4("g1sub", [p[0], p[1]]) == [0, 0]
    assert run4("g2sub", [outside[0][0], outside[0][1], outside[1][0], outside[1][1]]) == sub(outside)
    # easy part through the split inverse
    mid = S.run_program(S.get_program("fe1"), [M(v) for v in fa])
    mid[0] = M(R.fq_inv(F(mid[0])))
    out = [F(x) for x in S.run_program(S.get_program("easy2"), mid)]
    assert R.fq12_from_flat_ark(out) == R.myfq12_to_fq12(R.easy_part_native(R.myfq12_from_fq12(a)))
    # Fq12InverseGenerator on canonical MyFq12-ordered values (no Montgomery form at the boundary)
    am = R.myfq12_from_fq12(a)
    mid = S.run_program(S.get_program("inv1"), am)
    mid[0] = M(R.fq_inv(F(mid[0])))
    assert S.run_program(S.get_program("inv2"), mid) == R.myfq12_from_fq12(R.fq12_inv(a))


def test_fixed_q_miller_program():
    """The Miller program that takes its 68 line coefficients from the constant table (one prepared Q,
    many P) against multi_miller_loop_native on the same pair."""
    ps, qs = H.random_pairs(1, 4711)
    p, q = ps[0], qs[0]
    coeffs = [x for cf in R.g2_prepare(q) for fq2 in cf for x in fq2]
    assert len(coeffs) == 408
    extra = [x * (1 << 392) % R.P for x in coeffs]
    prog = S.get_program("millerfq")
    assert prog["n_in"] == 2 and prog["n_out"] == 12
    out = S.run_program(prog, [M(p[0]), M(p[1])], extra_consts=extra)
    assert R.fq12_from_flat_ark([F(x) for x in out]) == R.multi_miller_loop([p], [q])
    # the prepared form drops the G2 arithmetic: about 5,192 of the 7,141 multiplications remain
    assert prog["n_fmul"] < 0.8 * S.get_program("miller")["n_fmul"]


def test_two_pair_miller_program():
    """Two pairs on one accumulator against multi_miller_loop_native over both pairs."""
    ps, qs = H.random_pairs(2, 991)
    prog = S.get_program("miller2")
    assert prog["n_in"] == 12 and prog["n_out"] == 12
    gin = []
    for p, q in zip(ps, qs):
        gin += [M(p[0]), M(p[1]), M(q[0][0]), M(q[0][1]), M(q[1][0]), M(q[1][1])]
    out = S.run_program(prog, gin)
    assert R.fq12_from_flat_ark([F(x) for x in out]) == R.multi_miller_loop(ps, qs)
    # one shared squaring chain: clearly fewer products than two single-pair loops
    single = S.get_program("miller")
    assert prog["n_fmul"] < 1.8 * single["n_fmul"]


def test_lazy_product_programs(monkeypatch):
    """The same pairing through the programs built from n-product operations (schoolbook Fq6 products, one reduction
    per component: the form of the programs for 6 lanes per pairing), in the big-integer interpreter; the 8-lane
    programs and BLS381_LAZY_PRODUCTS=0 keep the Karatsuba form."""
    ps, qs = H.random_pairs(1, 4242)
    p, q = ps[0], qs[0]
    gin = [M(p[0]), M(p[1]), M(q[0][0]), M(q[0][1]), M(q[1][0]), M(q[1][1])]
    want = R.multi_miller_loop([p], [q])
    for K in (6,):
        prog = S.get_program("miller", K, 0)
        steps = [prog["hdrs"][2 * s] for s in range(prog["nsteps"])]
        assert steps.count(S.ST_DOTN) > 100
        out = S.run_program(prog, gin)
        assert R.fq12_from_flat_ark([F(x) for x in out]) == want
    p8 = S.get_program("miller", 8, 0)
    assert S.ST_DOTN not in [p8["hdrs"][2 * s] for s in range(p8["nsteps"])]
    monkeypatch.setenv("BLS381_LAZY_PRODUCTS", "0")
    p6 = S.get_program("miller", 6, 0)
    assert S.ST_DOTN not in [p6["hdrs"][2 * s] for s in range(p6["nsteps"])]


def test_program_statistics():
    """Multiplier work, counted in the work model's unit (12-limb multiply-accumulates: 300 per product,
    444 per two-product sum with its single reduction, 144 n + 156 per n-product sum), stays within 2% of the
    reference work model (7,141 / 8,119 Fp multiplications x 300)."""
    p = S.get_program("miller", 8, 0)
    assert 0.7 * 7141 * 300 < p["n_fmul"] <= 1.02 * 7141 * 300
    assert S.get_program("fe2", 8, 0)["n_fmul"] <= 8119 * 300


def test_witness_program_rows_match_circuit_op_totals():
    """src/pairing.rs:13-20 after G2PreparedTarget::from: 141,364 Fq operations + 12 generator outputs."""
    import numpy as np
    import witness_ref as W
    import plonky2_bls12_381_pairing_amd as B
    kinds = B.witness_kinds()
    assert B.witness_rows() == len(kinds) == 141376
    cnt = np.bincount(kinds, minlength=6)
    assert dict(add=cnt[0], sub=cnt[1], mul=cnt[2], neg=cnt[3], inv=cnt[4], gen=cnt[5]) == W.EXPECTED_COUNTS
    tr, _ = W.pairing_witness(R.G1_GEN, R.G2_GEN)
    names = ["add", "sub", "mul", "neg", "inv", "gen"]
    assert all(names[kinds[i]] == tr.rows[i][0] for i in range(len(kinds)))


@pytest.mark.parametrize("group,sched", [(None, None), ("0", None), ("8", None), ("64", None), (None, "levels"), ("32", "levels")])
def test_witness_row_cache_plan_replays(group, sched, monkeypatch):
    """The witness kernel reads operand rows from a per-proof LDS cache wherever the host-made plan says the row
    is there.  Replaying the plan: every such read finds its row, no two results of one iteration share an entry,
    the operations are otherwise those of the circuit replay, and the plan serves most reads at the default size.
    The replay also checks the level assignment under the plan (rows held back to fill groups of 16 by default, of
    other sizes or not at all here): every row made exactly once, from rows of earlier levels.  By default the
    operations run in rounds without barriers between them; the replay then also checks that every row read from global
    memory was stored before an earlier wait of the wave, or that its operation asks for that wait."""
    import ctypes as C
    from plonky2_bls12_381_pairing_amd import _lib
    if group is not None:
        monkeypatch.setenv("BLS381_WIT_GROUP", group)
    if sched is not None:
        monkeypatch.setenv("BLS381_WIT_SCHED", sched)
    out = (C.c_uint64 * 4)()
    reads = None
    for lpp, rows in ((16, 32), (16, 0), (16, 1), (16, 48), (16, 53), (8, 16), (32, 64), (64, 128), (1, 5)):
        assert _lib.lib.bls381_debug_witness_cache_plan(lpp, rows, out) == 0
        assert out[2] == 0, (lpp, rows, list(out))
        assert 141364 < out[3] < 141364 * 1.08   # the circuit's operations and the rows made for the products' left operands
        reads = reads or out[0]
        assert out[0] == reads  # reads of trace rows do not depend on the plan
        if rows == 0:
            assert out[1] == 0
        if (lpp, rows) == (16, 53) and group is None:   # the defaults: rounds of 16 in row order (or dependency levels)
            assert out[1] > (0.9 if sched is None else 0.8) * out[0]
    assert _lib.lib.bls381_debug_witness_cache_plan(16, 129, out) == -1
    assert _lib.lib.bls381_debug_witness_cache_plan(5, 32, out) == -1


@pytest.mark.parametrize("sched", [None, "levels"])
def test_witness_kernel_operations_compute_the_trace(sched, monkeypatch):
    """The witness kernel multiplies with one Montgomery product on 28-bit digits, left * right / 2^392, so the
    operations it runs give every product a left operand times 2^392: a constant or circuit input kept in that form,
    or a row made for the purpose (2^784 * row / 2^392) that all products with that factor share.  The operation list,
    run here on Python integers in the kernel's order, level by level, must reproduce the circuit replay's trace row
    for row, read nothing before it is made, and spend few extra products."""
    import ctypes as C
    import numpy as np
    import witness_ref as W
    from plonky2_bls12_381_pairing_amd import _lib
    if sched is not None:
        monkeypatch.setenv("BLS381_WIT_SCHED", sched)
    P = R.P
    RR = 1 << 392
    rinv = pow(RR, -1, P)
    q_aff = R.g2_mul(R.G2_GEN, 0x1234567)
    p_aff = R.g1_mul(R.G1_GEN, 0x7654321)
    tr, _ = W.pairing_witness(p_aff, q_aff)
    want = [v for _, v in tr.rows]
    n_rows = len(want)
    frob = []
    for k in (1, 2):
        g = R.fq2_pow((1, 1), (P ** k - 1) // 6)
        for i in range(1, 6):
            gi = R.fq2_pow(g, i)
            frob += [gi[0], gi[1]]
    consts = [0, 1, 4, R.fq_inv(2)] + frob
    consts = consts + [c * RR % P for c in consts] + [pow(2, 784, P)]
    inputs = [p_aff[0], p_aff[1], q_aff[0][0], q_aff[0][1], q_aff[1][0], q_aff[1][1]]
    inputs = inputs + [v * RR % P for v in inputs]
    rows = {}
    gen_rows = [i for i, (k, _) in enumerate(tr.rows) if k == 'gen']
    sizes = (C.c_uint64 * 4)()
    extra_products = 0
    for phase in (0, 1):
        assert _lib.lib.bls381_debug_witness_ops(phase, None, 0, None, 0, sizes) == 0
        n_ops, n_lv = int(sizes[0]), int(sizes[1])
        assert int(sizes[2]) == n_rows
        ops = np.zeros((n_ops, 3), dtype=np.uint32)
        lv = np.zeros(n_lv, dtype=np.uint32)
        assert _lib.lib.bls381_debug_witness_ops(phase, ops.ctypes.data_as(_lib.u32p), n_ops, lv.ctypes.data_as(_lib.u32p), n_lv, sizes) == 0
        if phase == 1:
            for i in gen_rows:            # the generator's output lands between the phases
                rows[i] = want[i]
        def value(ref, made_before):
            sp, idx = int(ref) >> 30, int(ref) & 0x3ffff
            assert not (int(ref) >> 18) & 0xff, "the program itself carries no cache fields"
            if sp == 1: return consts[idx]
            if sp == 2: return inputs[idx]
            assert idx in made_before, "a row is read before it is made"
            return rows[idx]
        for l in range(n_lv - 1):
            made_before = set(rows.keys())
            new = {}
            for dst, a, b in ops[lv[l]:lv[l + 1]]:
                kind, row = int(dst) >> 28, int(dst) & 0x3ffff
                va = value(a, made_before)
                if kind == 0: v = (va + value(b, made_before)) % P
                elif kind == 1: v = (va - value(b, made_before)) % P
                elif kind == 2:
                    v = va * value(b, made_before) * rinv % P
                    extra_products += row >= n_rows
                elif kind == 3: v = (-va) % P
                elif kind == 4: v = R.fq_inv(va)
                else: raise AssertionError(kind)
                assert row not in rows and row not in new
                new[row] = v
            rows.update(new)
    n_shadow = int(sizes[3])
    assert extra_products == n_shadow and len(rows) == n_rows + n_shadow
    assert all(rows[i] == want[i] for i in range(n_rows))
    assert n_shadow < 0.15 * W.EXPECTED_COUNTS['mul']


def test_one_hip_runtime_in_either_import_order():
    """This package imported before torch, and torch after it, end up on one libamdhip64 (torch's copy, loaded
    first by the package's loader): two runtimes in one process are what makes torch lose its GPUs."""
    import subprocess
    code = ("import plonky2_bls12_381_pairing_amd, torch\n"
            "print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-500:]
    libs = eval(out.stdout.strip().splitlines()[-1])
    assert len(libs) == 1 and "torch" in libs[0], libs


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pairing_c_abi")
    libdir = os.path.join(root, "plonky2_bls12_381_pairing_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
           os.path.join(root, "examples", "pairing_c_abi.c"), "-L", libdir, "-lbls381_hip", "-Wl,-rpath," + libdir, "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-800:]
    return exe


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/bls381_hip.h compiles as strict C99 and as C++, and the plain-C example builds and links against the
    shared library with nothing but gcc."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "bls381_hip.h"\nint main(void) { bls381_ctx* c = 0; return bls381_ctx_create(&c, 0) == 0 ? 0 : 1; }\n')
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only"],
                ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++"]):
        out = subprocess.run(cmd + ["-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-800:]
    _build_c_example(tmp_path)


def _c_prototypes(hdr):
    """name -> (return kind, [parameter kinds]) of every function the header declares."""
    text = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)

    def kind(decl):
        d = re.sub(r"\[[^\]]*\]", "*", decl.strip())          # uint64_t out[72] is a pointer parameter
        ptr = d.count("*")
        base = re.sub(r"\b(const|struct)\b|\*", " ", d).split()
        if not base:
            return None
        t = base[0] if base[0] != "unsigned" else ("unsigned" if len(base) < 2 or base[1] not in ("int", "char") else "unsigned " + base[1])
        t = {"uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int": "i32", "unsigned": "u32", "unsigned int": "u32", "size_t": "usize",
             "char": "c_char", "float": "f32", "double": "f64", "void": "void"}.get(t, "opaque:" + t)
        return ("*" * ptr) + t
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(bls381_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        if "typedef" in ret:
            continue
        params = [] if args.strip() in ("", "void") else [kind(a) for a in args.split(",")]
        protos[name] = (kind(ret), params)
    return protos


def _rust_prototypes(src):
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    block = re.sub(r"//.*", "", block)

    def kind(t):
        t = t.strip()
        ptr = len(re.findall(r"\*(?:const|mut)", t))
        base = re.sub(r"\*(?:const|mut)", " ", t).split()[-1]
        base = {"Bls381Ctx": "opaque:bls381_ctx", "Bls381MultiCtx": "opaque:bls381_multi_ctx", "G2PreparedDev": "opaque:bls381_g2_prepared"}.get(base, base)
        return ("*" * ptr) + base
    protos = {}
    for name, args, ret in re.findall(r"pub fn (bls381_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        params = [kind(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        protos[name] = (kind(ret) if ret.strip() else "void", params)
    return protos


def test_rust_ffi_declarations_match_the_header():
    """No Rust toolchain here, so the extern block of rust/bls381-hip-sys cannot be compiled against the library:
    every function it declares must exist in include/bls381_hip.h with the same parameters (count, order, pointer
    depth, element type) and return type."""
    hdr = _c_prototypes(open(os.path.join(ROOT, "include", "bls381_hip.h")).read())
    rust = _rust_prototypes(open(os.path.join(ROOT, "rust", "bls381-hip-sys", "src", "lib.rs")).read())
    assert len(hdr) > 60 and len(rust) > 50
    for name, (ret, params) in rust.items():
        assert name in hdr, name
        c_ret, c_params = hdr[name]
        assert ret == c_ret, (name, ret, c_ret)
        assert params == c_params, (name, params, c_params)
    # the entry points of the path itself are all bound
    for name in ("bls381_pairing_batch", "bls381_multi_miller_loop", "bls381_final_exp_batch", "bls381_pairing_witness_batch",
                 "bls381_pairing_witness_aux_batch", "bls381_fq12_inverse_limbs_batch", "bls381_g2_prepare_batch", "bls381_ell_batch"):
        assert name in rust, name


def test_chained_final_exponentiation_programs():
    """Part 2 of the final exponentiation as the library runs it on large batches: six step programs with the
    squaring-chain kernel between them.  The programs run in the big-integer interpreter, the oracle's squarings stand
    in for the kernel (the GPU test test_squaring_chain_kernel checks the kernel against the same squarings): the
    result is the reference's pairing value, and equal to the single program's."""
    R392 = pow(2, 392, R.P)
    INV392 = R.fq_inv(R392)

    def chain(area):                                      # 12 raw values -> the six kept powers, 72 raw values
        x = R.fq12_from_flat_ark([v * INV392 % R.P for v in area])
        kept = []
        for k in range(1, 64):
            x = R.fq12_square(x)
            if (R.BLS_X >> k) & 1:
                kept += [c * R392 % R.P for c in R.fq12_flatten_ark(x)]
        assert len(kept) == 72
        return kept

    ps, qs = H.random_pairs(1, 977)
    p, q = ps[0], qs[0]
    ml = R.multi_miller_loop([p], [q])
    mid = S.run_program(S.get_program("fe1"), [M(x) for x in R.fq12_flatten_ark(ml)])
    mid[0] = M(R.fq_inv(F(mid[0])))
    A, Bq, Cc, Dd, E, G = 72, 84, 96, 108, 120, 132
    Fa = [0] * 144
    progs = [S.get_program("fe2s%d" % k) for k in range(6)]
    assert [pr["n_in"] for pr in progs] == [21, 84, 108, 120, 72, 144] and [pr["n_out"] for pr in progs] == [24, 12, 12, 24, 12, 12]
    out = S.run_program(progs[0], mid)
    Fa[A:A + 24] = out
    # the easy part's value sits in A, in the programs' internal form
    easy = R.myfq12_to_fq12(R.easy_part_native(R.myfq12_from_fq12(ml)))
    assert [v * INV392 % R.P for v in Fa[A:A + 12]] == R.fq12_flatten_ark(easy)
    for k, (src, dst, n_out) in enumerate([(A, Cc, 12), (Cc, Dd, 12), (Dd, E, 24), (E, Cc, 12)], start=1):
        Fa[0:72] = chain(Fa[src:src + 12])
        out = S.run_program(progs[k], Fa[:progs[k]["n_in"]])
        Fa[dst:dst + n_out] = out
    Fa[0:72] = chain(Fa[Cc:Cc + 12])
    out = S.run_program(progs[5], Fa)
    got = R.fq12_from_flat_ark([F(x) for x in out])
    assert got == R.native_pairing([p], [q])
    single = S.run_program(S.get_program("fe2"), mid)
    assert out == single
