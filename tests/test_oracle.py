"""The oracle against implementation-independent known answers, the committed golden vectors,
and the C restatement against the Python one.  CPU only."""
import json
import os
import random

import numpy as np
import pytest

import bls381_ref as R
import witness_ref as W
import helpers as H

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairing_vectors.json")


def _fq12(hexes):
    return R.fq12_from_flat_ark([int(h, 16) for h in hexes])


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def test_constants():
    z = -R.BLS_X
    assert R.R_ORDER == z ** 4 - z ** 2 + 1
    assert R.P == (z - 1) ** 2 * R.R_ORDER // 3 + z
    assert R.P % 4 == 3 and R.P % 6 == 1            # asserts at src/native/fin_exp_helpers.rs:13-14
    assert (z - 1) ** 2 * (z + R.P) * (z * z + R.P * R.P - 1) + 3 == 3 * (R.P ** 4 - R.P ** 2 + 1) // R.R_ORDER
    assert R.g1_on_curve(R.G1_GEN) and R.g2_on_curve(R.G2_GEN)
    assert R.g1_mul(R.G1_GEN, R.R_ORDER) is None and R.g2_mul(R.G2_GEN, R.R_ORDER) is None
    assert len(R.g2_prepare(R.G2_GEN)) == 68
    assert R.frob_coeffs(2)[1] == 0 and R.frob_coeffs(1)[1] != 0


def test_pairing_known_answers(golden):
    e = R.native_pairing([R.G1_GEN], [R.G2_GEN])
    # public constant: first coefficient of the Gt generator in zkcrypto/bls12_381
    assert "%096x" % R.fq12_flatten_ark(e)[0] == golden["gt_generator_c000"]
    assert e != R.FQ12_ONE and R.fq12_pow(e, R.R_ORDER) == R.FQ12_ONE
    a, b = 0x1234567, 0x89abcdef01
    e2 = R.native_pairing([R.g1_mul(R.G1_GEN, a)], [R.g2_mul(R.G2_GEN, b)])
    assert e2 == R.fq12_pow(e, a * b)                                   # bilinearity
    en = R.native_pairing([R.G1_GEN], [R.g2_neg(R.G2_GEN)])
    assert R.fq12_mul(e, en) == R.FQ12_ONE                              # e(P,-Q) e(P,Q) = 1


def test_final_exponentiation_known_answers():
    rng = random.Random(7)
    x = R.fq12_from_flat_ark([rng.randrange(R.P) for _ in range(12)])
    assert R.final_exponentiation(x) == R.fq12_pow(x, R.FINAL_EXP_EXPONENT)     # x^(3(p^12-1)/r)
    # src/native/final_exponentiation.rs:124-141 (the reference's own fixed identity)
    assert R.myfq12_to_fq12(R.easy_part_native(R.myfq12_from_fq12(x))) == R.fq12_pow(x, R.EASY_EXPONENT)
    with pytest.raises(ZeroDivisionError):
        R.final_exponentiation(R.FQ12_ZERO)


def test_frobenius_is_the_p_power_map():
    """frobenius_map_native(x, k) (src/native/fin_exp_helpers.rs:10-47) against its definition x -> x^(p^k) by plain
    square-and-multiply: pins the tabulated constants and the w-power ordering without reference to any other code."""
    x = H.random_fq12(1, 12)[0]
    m = R.myfq12_from_fq12(x)
    for k in (1, 2, 3):
        assert R.myfq12_to_fq12(R.frobenius_map_native(m, k)) == R.fq12_pow(x, R.P ** k)
    assert R.fq12_conj(x) == R.fq12_pow(x, R.P ** 6)


def test_myfq12_roundtrip_and_mul():
    # src/utils/helpers.rs:248-267
    a, b = H.random_fq12(2, 3)
    am, bm = R.myfq12_from_fq12(a), R.myfq12_from_fq12(b)
    assert R.myfq12_to_fq12(am) == a
    assert R.myfq12_to_fq12(R.myfq12_mul(am, bm)) == R.fq12_mul(a, b)
    assert R.myfq12_to_fq12([(x + y) % R.P for x, y in zip(am, bm)]) == R.fq12_add(a, b)


def test_multi_miller_loop_semantics():
    ps, qs = H.random_pairs(6, 21)
    ps[1] = None
    qs[4] = None
    full = R.multi_miller_loop(ps, qs)
    prod = R.FQ12_ONE
    for p, q in zip(ps, qs):
        prod = R.fq12_mul(prod, R.multi_miller_loop([p], [q]))      # chunks of 4 == product of singles
    assert full == prod
    assert R.multi_miller_loop([], []) == R.FQ12_ONE
    with pytest.raises(ValueError):
        R.multi_miller_loop(ps, qs[:-1])


def test_golden_vectors(golden):
    for c in golden["cases"]:
        p = (int(c["g1"][0], 16), int(c["g1"][1], 16))
        q = ((int(c["g2"][0], 16), int(c["g2"][1], 16)), (int(c["g2"][2], 16), int(c["g2"][3], 16)))
        assert p == R.g1_mul(R.G1_GEN, int(c["a"], 16)) and q == R.g2_mul(R.G2_GEN, int(c["b"], 16))
        ml = R.multi_miller_loop([p], [q])
        assert ml == _fq12(c["miller"])
        assert R.final_exponentiation(ml) == _fq12(c["pairing"])
    fe = golden["final_exp"]
    x = _fq12(fe["x"])
    assert R.final_exponentiation(x) == _fq12(fe["final_exp"])
    assert R.hard_part_native(x) == _fq12(fe["hard"])
    assert R.fq12_inv(x) == _fq12(fe["inverse"])


def test_witness_replay_totals(golden):
    c = golden["cases"][1]
    p = (int(c["g1"][0], 16), int(c["g1"][1], 16))
    q = ((int(c["g2"][0], 16), int(c["g2"][1], 16)), (int(c["g2"][2], 16), int(c["g2"][3], 16)))
    tr, out = W.pairing_witness(p, q)
    assert tr.counts == W.EXPECTED_COUNTS
    assert sum(v for k, v in tr.counts.items() if k != "gen") == 141364
    assert len(tr.rows) == golden["witness"]["rows"] == 141376
    assert R.myfq12_to_fq12(out) == _fq12(c["pairing"])
    import hashlib
    h = hashlib.sha256()
    for _, v in tr.rows:
        h.update(v.to_bytes(48, "little"))
    assert h.hexdigest() == golden["witness"]["sha256_of_rows_le48"]
    # per-gadget counts quoted for the circuit
    t = W.Trace()
    a = R.myfq12_from_fq12(H.random_fq12(1, 1)[0])
    W.fq12_mul(t, a, a)
    assert len(t.rows) == 301 and t.counts["mul"] == 154


def test_c_oracle_matches_python():
    import c_oracle as CO
    ps, qs = H.random_pairs(5, 99)
    ps[2] = None
    fl = H.inf_flags(ps, qs)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    out = CO.pairing_batch(g1, g2, fl)
    ml = CO.pairing_batch(g1, g2, fl, final_exp=False)
    for i in range(5):
        assert H.fq12_from_row(ml[i]) == R.multi_miller_loop([ps[i]], [qs[i]])
        assert H.fq12_from_row(out[i]) == R.native_pairing([ps[i]], [qs[i]])
    assert H.fq12_from_row(CO.multi_miller_loop(g1, g2, fl)) == R.multi_miller_loop(ps, qs)
    x = H.random_fq12(2, 5)
    xa = H.fq12_array(x)
    assert H.fq12_from_row(CO.final_exp_batch(xa)[1]) == R.final_exponentiation(x[1])
    assert H.fq12_from_row(CO.hard_part(xa[0])) == R.hard_part_native(x[0])
    assert H.fq12_from_row(CO.fq12_inverse(xa[0])) == R.fq12_inv(x[0])
    assert H.fq12_from_row(CO.fq12_mul(xa[0], xa[1])) == R.fq12_mul(x[0], x[1])
    cf = CO.g2_prepare(g2[0])
    ref = R.g2_prepare(qs[0])
    for k in (0, 1, 33, 67):
        for j in range(3):
            assert (H.u64_to_fq(cf[k, j, 0]), H.u64_to_fq(cf[k, j, 1])) == ref[k][j]
    with pytest.raises(ZeroDivisionError):
        CO.final_exp_batch(np.zeros((1, 72), dtype=np.uint64))


def test_divstep_inversion_model():
    """The limb-accurate model of the GPU's division-step inversion (register widths asserted inside)
    agrees with Python's modular inverse."""
    import random
    import divstep_model as D
    rng = random.Random(11)
    xs = [0, 1, 2, D.P - 1, D.P - 2, (D.P + 1) // 2, 1 << 380] + [rng.randrange(D.P) for _ in range(300)]
    for x in xs:
        assert D.modinv(x) == (pow(x, -1, D.P) if x else 0)


def test_other_generator_oracles():
    """Square-root and inverse generator restatements satisfy their defining equations."""
    rng = random.Random(21)
    for _ in range(40):
        x, s = rng.randrange(R.P), rng.randrange(2)
        try:
            r = R.fq_sqrt_generator(x, s)
            assert r * r % R.P == x and (r & 1) == s
        except ValueError:
            assert pow(x, (R.P - 1) // 2, R.P) == R.P - 1 or x == 0
        a = (rng.randrange(R.P), rng.randrange(R.P))
        sq = R.fq2_square(a)
        r = R.fq2_sqrt_generator(sq, s)
        assert R.fq2_square(r) == sq and R.sgn0_fq2(r) == bool(s)
        assert R.fq2_mul(R.fq2_inverse_generator(a), a) == (1, 0)
        c = [rng.randrange(R.P) for _ in range(6)]
        inv = R.fq6_inverse_generator(c)
        A = ((c[0], c[3]), (c[1], c[4]), (c[2], c[5]))
        Bv = ((inv[0], inv[3]), (inv[1], inv[4]), (inv[2], inv[5]))
        assert R.fq6_mul(A, Bv) == R.FQ6_ONE
    assert R.fq2_inverse_generator((0, 0)) == (0, 0)


def test_c_oracle_optimised_variant_gives_the_same_values():
    """The second CPU row of the benchmark (cyclotomic squarings instead of the reference's 144-multiplication
    products) computes bit-identical pairings."""
    import c_oracle as CO
    ps, qs = H.random_pairs(6, 909)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    ref = CO.pairing_batch(g1, g2)
    CO.set_fast(True)
    try:
        fast = CO.pairing_batch(g1, g2)
    finally:
        CO.set_fast(False)
    assert np.array_equal(ref, fast)


def test_point_format_oracle():
    """Compressed encodings: the generators' well-known byte strings, round trips, rejections, and the
    order-r test against curve points outside the subgroup."""
    import random
    import bls381_points as PT
    assert PT.g1_compress(R.G1_GEN) == PT.G1_GEN_COMPRESSED
    assert PT.g2_compress(R.G2_GEN) == PT.G2_GEN_COMPRESSED
    assert PT.g1_decompress(PT.G1_GEN_COMPRESSED) == (PT.OK, R.G1_GEN)
    assert PT.g2_decompress(PT.G2_GEN_COMPRESSED) == (PT.OK, R.G2_GEN)
    assert PT.g1_decompress(PT.g1_compress(None)) == (PT.INFINITY, None)
    assert PT.g2_decompress(PT.g2_compress(None)) == (PT.INFINITY, None)
    rng = random.Random(11)
    for _ in range(3):
        k = rng.randrange(1, R.R_ORDER)
        for p in (R.g1_mul(R.G1_GEN, k), R.g1_neg(R.g1_mul(R.G1_GEN, k))):
            assert PT.g1_decompress(PT.g1_compress(p)) == (PT.OK, p)
        for q in (R.g2_mul(R.G2_GEN, k), R.g2_neg(R.g2_mul(R.G2_GEN, k))):
            assert PT.g2_decompress(PT.g2_compress(q)) == (PT.OK, q)
    # rejections
    b = bytearray(PT.G1_GEN_COMPRESSED); b[0] &= 0x7F
    assert PT.g1_decompress(bytes(b))[0] == PT.BAD_ENCODING          # uncompressed flag
    assert PT.g1_decompress(bytes([0xE0]) + bytes(47))[0] == PT.BAD_ENCODING   # infinity with the sort flag
    assert PT.g1_decompress(bytes([0xC0]) + bytes(46) + b"\x01")[0] == PT.BAD_ENCODING
    too_big = bytearray(R.P.to_bytes(48, "big")); too_big[0] |= 0x80
    assert PT.g1_decompress(bytes(too_big))[0] == PT.BAD_ENCODING    # x = p
    # x = 1: 1 + 4 = 5 is not a square mod p
    assert R.fq_sqrt(5) is None
    one = bytearray((1).to_bytes(48, "big")); one[0] |= 0x80
    assert PT.g1_decompress(bytes(one))[0] == PT.NOT_ON_CURVE
    # a curve point outside the subgroup passes without the check and fails with it
    x = 4
    while R.fq_sqrt((x ** 3 + 4) % R.P) is None:
        x += 1
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    st, pt = PT.g1_decompress(bytes(enc), check_subgroup=False)
    assert st == PT.OK and R.g1_on_curve(pt) and not PT.g1_in_subgroup(pt)
    assert PT.g1_decompress(bytes(enc))[0] == PT.NOT_IN_SUBGROUP
    # aggregation is the group law
    pts = [R.g1_mul(R.G1_GEN, k) for k in (3, 5, 9)]
    assert PT.g1_aggregate(pts) == R.g1_mul(R.G1_GEN, 17)
    assert PT.g1_aggregate([pts[0], R.g1_neg(pts[0])]) is None


def test_point_format_golden_vectors():
    import json, os
    import bls381_points as PT
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "point_vectors.json")))
    assert bytes.fromhex(v["generator_g1"]) == PT.G1_GEN_COMPRESSED == PT.g1_compress(R.G1_GEN)
    assert bytes.fromhex(v["generator_g2"]) == PT.G2_GEN_COMPRESSED == PT.g2_compress(R.G2_GEN)
    p1, p2 = [], []
    for c in v["cases"]:
        st, p = PT.g1_decompress(bytes.fromhex(c["g1_bytes"]))
        assert st == PT.OK and [int(x, 16) for x in c["g1"]] == list(p)
        st, q = PT.g2_decompress(bytes.fromhex(c["g2_bytes"]))
        assert st == PT.OK and [int(x, 16) for x in c["g2"]] == [q[0][0], q[0][1], q[1][0], q[1][1]]
        assert p == R.g1_mul(R.G1_GEN, int(c["k"], 16))
        p1.append(p); p2.append(q)
    assert PT.g1_compress(PT.g1_aggregate(p1)).hex() == v["sum_g1"]
    assert PT.g2_compress(PT.g2_aggregate(p2)).hex() == v["sum_g2"]
    for e in v["on_curve_outside_subgroup_g1"]:
        assert PT.g1_decompress(bytes.fromhex(e))[0] == PT.NOT_IN_SUBGROUP
