"""Parity of the HIP path with the oracle, through the C ABI, on a real MI355X.
Integer work: every comparison is bit-exact."""
import json
import os

import numpy as np
import pytest

import bls381_ref as R
import helpers as H

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairing_vectors.json")


@pytest.fixture(scope="module")
def ctx():
    # PyTorch-ROCm ships its own HIP runtime; when both runtimes live in one process torch's has to
    # come up first (the other order leaves torch without a device on some hosts)
    import torch
    torch.cuda.init()
    import plonky2_bls12_381_pairing_amd as B
    c = B.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def CO():
    import c_oracle
    return c_oracle


def _fq12(hexes):
    return R.fq12_from_flat_ark([int(h, 16) for h in hexes])


def test_golden_vectors(ctx):
    g = json.load(open(GOLDEN))
    ps = [(int(c["g1"][0], 16), int(c["g1"][1], 16)) for c in g["cases"]]
    qs = [((int(c["g2"][0], 16), int(c["g2"][1], 16)), (int(c["g2"][2], 16), int(c["g2"][3], 16))) for c in g["cases"]]
    ml = ctx.miller_loop_batch(H.g1_array(ps), H.g2_array(qs))
    pr = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
    for i, c in enumerate(g["cases"]):
        assert H.fq12_from_row(ml[i]) == _fq12(c["miller"])
        assert H.fq12_from_row(pr[i]) == _fq12(c["pairing"])
    assert "%096x" % R.fq12_flatten_ark(H.fq12_from_row(pr[0]))[0] == g["gt_generator_c000"]
    fe = g["final_exp"]
    x = H.fq12_array([_fq12(fe["x"])])
    assert H.fq12_from_row(ctx.final_exponentiation_batch(x)[0]) == _fq12(fe["final_exp"])
    assert H.fq12_from_row(ctx.easy_part_batch(x)[0]) == _fq12(fe["easy"])
    assert H.fq12_from_row(ctx.hard_part_batch(x)[0]) == _fq12(fe["hard"])
    assert H.fq12_from_row(ctx.frobenius_map_batch(x, 1)[0]) == _fq12(fe["frob1"])
    assert H.fq12_from_row(ctx.frobenius_map_batch(x, 2)[0]) == _fq12(fe["frob2"])


@pytest.mark.parametrize("n", [1, 7, 33, 257])
def test_pairing_ragged_sizes_vs_python_oracle(ctx, n):
    ps, qs = H.random_pairs(min(n, 9), 1000 + n)
    ps, qs = (ps * (n // len(ps) + 1))[:n], (qs * (n // len(qs) + 1))[:n]
    qs = qs[1:] + qs[:1]                                  # pair P_i with a different Q
    out = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
    cache = {}
    for i in range(n):
        key = (ps[i], qs[i])
        if key not in cache:
            cache[key] = R.native_pairing([ps[i]], [qs[i]])
        assert H.fq12_from_row(out[i]) == cache[key]


def test_empty_batches(ctx):
    z1, z2 = np.zeros((0, 12), np.uint64), np.zeros((0, 24), np.uint64)
    assert ctx.pairing_batch(z1, z2).shape == (0, 72)
    assert ctx.miller_loop_batch(z1, z2).shape == (0, 72)
    assert ctx.final_exponentiation_batch(np.zeros((0, 72), np.uint64)).shape == (0, 72)
    out = ctx.multi_miller_loop_native(z1, z2)            # empty product
    assert H.fq12_from_row(out) == R.FQ12_ONE


def test_infinity_pairs_and_length_mismatch(ctx):
    ps, qs = H.random_pairs(6, 5)
    ps[0] = None
    qs[3] = None
    ps[5] = None
    qs[5] = None
    fl = H.inf_flags(ps, qs)
    ml = ctx.miller_loop_batch(H.g1_array(ps), H.g2_array(qs), fl)
    pr = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs), fl)
    for i in range(6):
        assert H.fq12_from_row(ml[i]) == R.multi_miller_loop([ps[i]], [qs[i]])
        assert H.fq12_from_row(pr[i]) == R.native_pairing([ps[i]], [qs[i]])
    assert H.fq12_from_row(pr[0]) == R.FQ12_ONE
    with pytest.raises(ValueError):                        # zip_eq panics in the reference
        ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs[:-1]))


def test_final_exponentiation_arbitrary_input_and_zero(ctx):
    import plonky2_bls12_381_pairing_amd as B
    xs = H.random_fq12(5, 77)                             # not in the cyclotomic subgroup
    out = ctx.final_exponentiation_batch(H.fq12_array(xs))
    for i, x in enumerate(xs):
        assert H.fq12_from_row(out[i]) == R.final_exponentiation(x)
    easy = ctx.easy_part_batch(H.fq12_array(xs[:2]))
    hard = ctx.hard_part_batch(H.fq12_array(xs[:2]))
    for i in range(2):
        assert H.fq12_from_row(easy[i]) == R.myfq12_to_fq12(R.easy_part_native(R.myfq12_from_fq12(xs[i])))
        assert H.fq12_from_row(hard[i]) == R.hard_part_native(xs[i])
    bad = H.fq12_array([xs[0], R.FQ12_ZERO])
    with pytest.raises(B.Bls381Error) as ei:               # the reference panics on division by zero
        ctx.final_exponentiation_batch(bad)
    assert ei.value.code == -4


def test_tower_helpers(ctx):
    a, b = H.random_fq12(3, 8), H.random_fq12(3, 9)
    out = ctx.fq12_mul_batch(H.fq12_array(a), H.fq12_array(b))
    for i in range(3):
        assert H.fq12_from_row(out[i]) == R.fq12_mul(a[i], b[i])
        assert R.myfq12_from_fq12(H.fq12_from_row(out[i])) == R.myfq12_mul(R.myfq12_from_fq12(a[i]), R.myfq12_from_fq12(b[i]))
    for pw in list(range(12)) + [13, 26]:                   # the reference reduces the power mod 12
        fr = ctx.frobenius_map_batch(H.fq12_array(a), pw)
        for i in range(3):
            assert H.fq12_from_row(fr[i]) == R.myfq12_to_fq12(R.frobenius_map_native(R.myfq12_from_fq12(a[i]), pw))
    cj = ctx.conjugate_batch(H.fq12_array(a))
    for i in range(3):                                        # conjugate_fp12: odd MyFq12 coefficients change sign
        m = R.myfq12_from_fq12(a[i])
        assert R.myfq12_from_fq12(H.fq12_from_row(cj[i])) == [c if k % 2 == 0 else (-c) % R.P for k, c in enumerate(m)]
        assert H.fq12_from_row(cj[i]) == R.fq12_conj(a[i])


def test_pow_native(ctx):
    """pow_native on arbitrary (non-cyclotomic) input: by BLS_X as the hard part calls it, by other exponents
    including multi-word ones, and the reference's quirk that exponent zero returns the input."""
    a = H.random_fq12(3, 81)
    px = ctx.pow_x_batch(H.fq12_array(a))
    for i in range(3):
        want = R.myfq12_to_fq12(R.pow_native(R.myfq12_from_fq12(a[i]), R.BLS_X))
        assert H.fq12_from_row(px[i]) == want == R.fq12_pow(a[i], R.BLS_X)
    for exp in (0, 1, 2, 3, 0xb, (1 << 64) + 5, (0x1234567 << 70) | 0x89abcdef):
        got = ctx.pow_native_batch(H.fq12_array(a[:2]), exp)
        for i in range(2):
            assert H.fq12_from_row(got[i]) == R.myfq12_to_fq12(R.pow_native(R.myfq12_from_fq12(a[i]), exp))
    assert np.array_equal(ctx.pow_native_batch(H.fq12_array(a), [R.BLS_X]), px)
    assert np.array_equal(ctx.pow_native_batch(H.fq12_array(a), 0), H.fq12_array(a))


def test_identities_the_reference_asserts_hold_on_the_gpu_path(ctx):
    """Known answers that depend on no implementation, checked with the library's own entry points only (the
    oracle supplies inputs, never expected values):
      * the Frobenius map is the p-power map: frobenius_map_native(x, k) == x^(p^k), with the power taken by
        pow_native(., p) applied k times (generic squarings, arbitrary x);
      * easy_part_native(x) == x^((p^6 - 1)(p^2 + 1)), the reference's own fixed assertion
        (/root/reference/src/native/final_exponentiation.rs:124-141), in the division-free form
        easy(x) * x^(p^2) * x == x^(p^8) * x^(p^6);
      * bilinearity with the exponent applied on the device: e([a]P, [b]Q) == e(P, Q)^(ab), and e(P, Q)^r == 1,
        e(P, Q) != 1."""
    xs = H.random_fq12(2, 4401)
    x = H.fq12_array(xs)
    p_words = [(R.P >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
    powers = [x]
    for k in range(1, 9):
        powers.append(ctx.pow_native_batch(powers[-1], p_words))
        if k <= 3 or k == 6:
            assert np.array_equal(ctx.frobenius_map_batch(x, k), powers[k]), k
    # x^(p^6) is the conjugate (the map of order two of Fq12 over Fq6)
    assert np.array_equal(powers[6], ctx.conjugate_batch(x))
    easy = ctx.easy_part_batch(x)
    lhs = ctx.fq12_mul_batch(ctx.fq12_mul_batch(easy, powers[2]), x)
    rhs = ctx.fq12_mul_batch(powers[8], powers[6])
    assert np.array_equal(lhs, rhs)
    # bilinearity
    import random
    rng = random.Random(4402)
    ps, qs = H.random_pairs(3, 4403)
    a = [rng.randrange(1, 1 << 64) for _ in range(3)]
    b = [rng.randrange(1, 1 << 64) for _ in range(3)]
    e_pq = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
    e_ab = ctx.pairing_batch(H.g1_array([R.g1_mul(p, k) for p, k in zip(ps, a)]), H.g2_array([R.g2_mul(q, k) for q, k in zip(qs, b)]))
    one = H.fq12_array([R.FQ12_ONE])[0]
    r_words = [(R.R_ORDER >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    for i in range(3):
        ab = a[i] * b[i]
        got = ctx.pow_native_batch(e_pq[i:i + 1], [ab & 0xFFFFFFFFFFFFFFFF, ab >> 64])
        assert np.array_equal(got[0], e_ab[i]), i
        assert not np.array_equal(e_pq[i], one)
    assert all(np.array_equal(row, one) for row in ctx.pow_native_batch(e_pq, r_words))


def test_g2_prepare_and_ell(ctx):
    ps, qs = H.random_pairs(3, 31)
    cf = ctx.g2_prepare_batch(H.g2_array(qs))
    assert cf.shape == (3, 68, 3, 2, 6)
    for i in range(3):
        ref = R.g2_prepare(qs[i])
        for k in range(68):
            for j in range(3):
                assert (H.u64_to_fq(cf[i, k, j, 0]), H.u64_to_fq(cf[i, k, j, 1])) == ref[k][j]
    f = H.random_fq12(3, 32)
    coeffs = np.ascontiguousarray(cf[:, 5].reshape(3, 36))
    out = ctx.ell_batch(H.fq12_array(f), coeffs, H.g1_array(ps))
    for i in range(3):
        assert H.fq12_from_row(out[i]) == R.ell(f[i], R.g2_prepare(qs[i])[5], ps[i])


@pytest.mark.parametrize("n", [1, 2, 5, 13])
def test_multi_miller_loop_and_multi_pairing(ctx, n):
    ps, qs = H.random_pairs(n, 200 + n)
    fl = None
    if n == 5:
        ps[2] = None
        fl = H.inf_flags(ps, qs)
    got = ctx.multi_miller_loop_native(H.g1_array(ps), H.g2_array(qs), fl)
    assert H.fq12_from_row(got) == R.multi_miller_loop(ps, qs)
    e, is_one = ctx.native_pairing(H.g1_array(ps), H.g2_array(qs), fl)
    assert H.fq12_from_row(e) == R.native_pairing(ps, qs)
    assert not is_one


def test_aggregate_signature_check_512_keys(ctx, CO):
    """sum_i e(pk_i, H_i) * e(-G1, sigma) = 1 with sigma = sum sk_i H_i (a 512-key sync committee)."""
    import random
    rng = random.Random(512)
    n = 512
    base = 8                                              # distinct keys/messages, tiled to 512
    sks = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
    hs = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
    pks = [R.g1_mul(R.G1_GEN, s) for s in sks]
    Hs = [R.g2_mul(R.G2_GEN, h) for h in hs]
    ps = [pks[i % base] for i in range(n)]
    qs = [Hs[i % base] for i in range(n)]
    sigma_scalar = sum(sks[i % base] * hs[i % base] for i in range(n)) % R.R_ORDER
    sigma = R.g2_mul(R.G2_GEN, sigma_scalar)
    ps.append(R.g1_neg(R.G1_GEN))
    qs.append(sigma)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    e, is_one = ctx.native_pairing(g1, g2)
    assert is_one and H.fq12_from_row(e) == R.FQ12_ONE
    ml = ctx.multi_miller_loop_native(g1, g2)
    assert np.array_equal(ml, CO.multi_miller_loop(g1, g2))          # bit-for-bit vs the C oracle
    # negative case: one wrong signature share
    qs[-1] = R.g2_mul(R.G2_GEN, (sigma_scalar + 1) % R.R_ORDER)
    e, is_one = ctx.native_pairing(g1, H.g2_array(qs))
    assert not is_one
    # the multi-GPU finishing step on gathered partial products
    parts = np.stack([ctx.multi_miller_loop_native(g1[i::4], g2[i::4]) for i in range(4)])
    e2, one2 = ctx.finish_multi_pairing(parts)
    assert one2 and H.fq12_from_row(e2) == R.FQ12_ONE


def test_fq12_inverse_generator_limbs(ctx):
    import plonky2_bls12_381_pairing_amd as B
    xs = H.random_fq12(4, 55)
    lim = np.zeros((4, 144), np.uint32)
    for i, x in enumerate(xs):
        lim[i] = [l for c in R.myfq12_from_fq12(x) for l in R.fq_to_u32_limbs(c)]
    inv = ctx.fq12_inverse_limbs_batch(lim)
    for i, x in enumerate(xs):
        want = R.myfq12_from_fq12(R.fq12_inv(x))
        got = [R.u32_limbs_to_int(inv[i, 12 * k:12 * k + 12]) for k in range(12)]
        assert got == want
    bad = lim.copy()
    bad[1] = 0                                            # zero: inverse().unwrap() panics
    with pytest.raises(B.Bls381Error):
        ctx.fq12_inverse_limbs_batch(bad)
    bad = lim.copy()
    bad[2, :12] = R.fq_to_u32_limbs(R.P)                  # >= p: from_biguint_to_fq panics
    with pytest.raises(B.Bls381Error):
        ctx.fq12_inverse_limbs_batch(bad)


def test_full_batch_65536_against_c_oracle_and_bilinearity(ctx, CO):
    """The full-size case: 65,536 independent pairings, every output compared with the CPU oracle
    on the distinct inputs, plus size-independent properties on the whole batch."""
    n = 65536
    base = 64
    ps, qs = H.random_pairs(base, 4242)
    g1b, g2b = H.g1_array(ps), H.g2_array(qs)
    idx_p = np.arange(n) % base
    idx_q = (np.arange(n) * 7 + np.arange(n) // base) % base
    g1, g2 = g1b[idx_p], g2b[idx_q]
    out = ctx.pairing_batch(g1, g2)
    # distinct (p, q) combinations, each checked once against the C oracle
    combos = {}
    for i in range(n):
        combos.setdefault((int(idx_p[i]), int(idx_q[i])), i)
    keys = list(combos.keys())
    ref = CO.pairing_batch(g1b[[k[0] for k in keys]], g2b[[k[1] for k in keys]])
    ref_of = {k: ref[j] for j, k in enumerate(keys)}
    for i in range(n):
        assert np.array_equal(out[i], ref_of[(int(idx_p[i]), int(idx_q[i]))]), i
    # e(P, Q) * e(P, -Q) = 1 through the same kernels (multiplication on the GPU as well)
    m = 1024
    g2n = H.g2_array([R.g2_neg(q) for q in qs])[idx_q[:m]]
    neg = ctx.pairing_batch(g1[:m], g2n)
    prod = ctx.fq12_mul_batch(out[:m], neg)
    one = H.fq12_array([R.FQ12_ONE])[0]
    assert all(np.array_equal(prod[i], one) for i in range(m))


def test_chained_final_exponentiation_on_ragged_batches(ctx, CO):
    """Batches above 8,192 pairs take the hard part's squaring chains through the register-resident kernel and the
    multiplications around them through six programs; below, one program does it all.  Ragged sizes just above the
    switch (a last workgroup of the chain kernel with 1, 65 and 201 live elements) and around the size at which the
    chain kernel goes from workgroups of 64 elements to workgroups of 256: every output equals the same
    inputs run in small batches through the single program, and a sample equals the CPU oracle."""
    base = 48
    ps, qs = H.random_pairs(base, 90210)
    g1b, g2b = H.g1_array(ps), H.g2_array(qs)
    for n in (8192 + 1, 8192 + 65, 8192 + 2048 + 201, 40960, 40960 + 1, 49999):   # (the chain kernel changes its workgroup shape above 40,960)
        idx_p = (np.arange(n) * 11 + 3) % base
        idx_q = (np.arange(n) // 5 + np.arange(n)) % base
        g1, g2 = g1b[idx_p], g2b[idx_q]
        out = ctx.pairing_batch(g1, g2)
        small = np.concatenate([ctx.pairing_batch(g1[i:i + 4096], g2[i:i + 4096]) for i in range(0, n, 4096)])
        assert np.array_equal(out, small), n
        probe = np.array([0, 63, 64, 255, 256, 8191, 8192, n - 2, n - 1])
        assert np.array_equal(out[probe], CO.pairing_batch(g1[probe], g2[probe])), n


def test_batch_spanning_two_chunks(ctx, CO):
    """A batch larger than the library's 65,536-element working chunk (maximum-size edge): elements on
    both sides of the chunk boundary and the ragged tail against the CPU oracle, and the tail chunk
    equal to the same inputs run as a batch of their own."""
    n = 65536 + 777
    base = 32
    ps, qs = H.random_pairs(base, 777)
    g1b, g2b = H.g1_array(ps), H.g2_array(qs)
    idx_p = (np.arange(n) * 5 + 1) % base
    idx_q = (np.arange(n) // 3) % base
    g1, g2 = g1b[idx_p], g2b[idx_q]
    out = ctx.pairing_batch(g1, g2)
    probe = np.array([0, 1, 65535, 65536, 65537, 65536 + 64, n - 2, n - 1])
    assert np.array_equal(out[probe], CO.pairing_batch(g1[probe], g2[probe]))
    assert np.array_equal(out[65536:], ctx.pairing_batch(g1[65536:], g2[65536:]))
    ml = ctx.miller_loop_batch(g1, g2)
    fe = ctx.final_exponentiation_batch(ml[65500:65600])
    fe = fe[0] if isinstance(fe, tuple) else fe
    assert np.array_equal(fe, out[65500:65600])


def test_pairing_witness_trace(ctx):
    """Every row of the witness trace (canonical 12 x u32 limbs) against the CPU replay of the
    circuit's builder calls, for three proofs."""
    import witness_ref as W
    ps, qs = H.random_pairs(3, 808)
    trace = ctx.pairing_witness_batch(H.g1_array(ps), H.g2_array(qs))
    assert trace.shape == (3, 141376, 12)
    for e in range(3):
        tr, out = W.pairing_witness(ps[e], qs[e])
        want = np.array([R.fq_to_u32_limbs(v) for _, v in tr.rows], dtype=np.uint32)
        bad = np.nonzero((trace[e] != want).any(axis=1))[0]
        assert bad.size == 0, (e, bad[:5], [tr.rows[i][0] for i in bad[:5]])
        # the last row group is the pairing value in MyFq12 order
        assert R.myfq12_to_fq12(out) == R.native_pairing([ps[e]], [qs[e]])


def test_pairing_witness_trace_256_proofs(ctx, CO, monkeypatch):
    """The configured size of the witness workload: 256 proofs.  Every row of a strided sample of proofs against the
    CPU replay, the pairing value inside the trace against the C oracle's native pairing for all 256, and the same
    batch again in several passes (host form: three of at most 96 proofs, device form: six of at most 50), which must
    give the same bytes."""
    import torch
    import witness_ref as W
    from plonky2_bls12_381_pairing_amd import _lib
    bp, bq = H.random_pairs(16, 2560)
    ps = [bp[i % 16] for i in range(256)]
    qs = [bq[i // 16] for i in range(256)]
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    trace = ctx.pairing_witness_batch(g1, g2)
    assert trace.shape == (256, 141376, 12)
    out_rows = None
    for e in range(0, 256, 17):
        tr, out = W.pairing_witness(ps[e], qs[e])
        want = np.array([R.fq_to_u32_limbs(v) for _, v in tr.rows], dtype=np.uint32)
        bad = np.nonzero((trace[e] != want).any(axis=1))[0]
        assert bad.size == 0, (e, bad[:5], [tr.rows[i][0] for i in bad[:5]])
        # where the pairing value's twelve coefficients (MyFq12 order) sit: the same rows in every proof
        vals = [v for _, v in tr.rows]
        rows = [max(i for i in range(141300, 141376) if vals[i] == o) for o in out]
        assert out_rows is None or rows == out_rows
        out_rows = rows
    native = CO.pairing_batch(g1, g2)
    for e in range(256):
        coeffs = R.myfq12_from_fq12(H.fq12_from_row(native[e]))
        assert np.array_equal(trace[e][out_rows], np.array([R.fq_to_u32_limbs(v) for v in coeffs], dtype=np.uint32)), e
    monkeypatch.setenv("BLS381_WIT_CH", "96")
    assert np.array_equal(ctx.pairing_witness_batch(g1, g2), trace)
    monkeypatch.setenv("BLS381_WIT_CH", "50")
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    d_out = torch.zeros(trace.size, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.bls381_pairing_witness_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), 256, d_out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32).reshape(trace.shape), trace)


@pytest.mark.parametrize("cache,lpp", [("0", "16"), ("16", "16"), ("1", "16"), ("48", "8"), ("100", "32"), ("128", "64")])
def test_witness_kernel_geometries(ctx, cache, lpp, monkeypatch):
    """The witness kernel with no LDS row cache, one entry, small and large ones, at 8, 16, 32 and 64 lanes per proof
    (each with its own cache plan): the same trace bytes as the default (53 rows, 16 lanes), which
    test_pairing_witness_trace checks against the CPU replay."""
    ps, qs = H.random_pairs(5, 9191)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    want = ctx.pairing_witness_batch(g1, g2)
    monkeypatch.setenv("BLS381_WIT_CACHE", cache)
    monkeypatch.setenv("BLS381_WIT_LPP", lpp)
    assert np.array_equal(ctx.pairing_witness_batch(g1, g2), want)


@pytest.mark.parametrize("group", ["0", "8", "64"])
def test_witness_level_assignments(ctx, group, monkeypatch):
    """Rows of the trace that can wait are held back until they fill a group of the lanes of a proof (16 by default):
    the same trace bytes with every row at its earliest level and with groups of other sizes."""
    ps, qs = H.random_pairs(5, 4242)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    want = ctx.pairing_witness_batch(g1, g2)
    monkeypatch.setenv("BLS381_WIT_GROUP", group)
    assert np.array_equal(ctx.pairing_witness_batch(g1, g2), want)
    monkeypatch.delenv("BLS381_WIT_GROUP")
    assert np.array_equal(ctx.pairing_witness_batch(g1, g2), want)


def test_witness_schedules_and_geometries_agree(ctx, monkeypatch):
    """The kernel's operations run in rounds taken in row order, without barriers, reading what the rounds before made
    from the proof's LDS row cache (the default), or in dependency levels with a barrier after each
    (BLS381_WIT_SCHED=levels).  The trace bytes are the same in both, with a cache too small to hold what a round's
    successor reads (rows then come back from global memory behind an explicit wait), without a cache, and at 32 and
    8 lanes per proof; every row against the CPU replay."""
    import witness_ref as W
    ps, qs = H.random_pairs(6, 1717)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    want = ctx.pairing_witness_batch(g1, g2)
    tr, _ = W.pairing_witness(ps[3], qs[3])
    assert np.array_equal(want[3], np.array([R.fq_to_u32_limbs(v) for _, v in tr.rows], dtype=np.uint32))
    for env in ({"BLS381_WIT_SCHED": "levels"}, {"BLS381_WIT_CACHE": "3"}, {"BLS381_WIT_CACHE": "0"},
                {"BLS381_WIT_LPP": "32", "BLS381_WIT_CACHE": "96"}, {"BLS381_WIT_LPP": "8", "BLS381_WIT_CACHE": "20"},
                {"BLS381_WIT_SCHED": "levels", "BLS381_WIT_CACHE": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(ctx.pairing_witness_batch(g1, g2), want), env
        for k in env:
            monkeypatch.delenv(k)


def test_pairing_witness_auxiliary_values(ctx):
    """The gadgets' auxiliary values beside the trace (mul quotients, add/sub/neg carry bytes, the inv row's quotient)
    against the CPU replay, which forms them from Python integers: a * b == q * p + r for every one of the 70,656
    mul rows of two proofs.  Parity with the plonky2_ecdsa fork is unpinned (its source is not in the reference tree)."""
    import witness_ref as W
    ps, qs = H.random_pairs(2, 909)
    trace, mq, cy, iq = ctx.pairing_witness_aux_batch(H.g1_array(ps), H.g2_array(qs))
    assert mq.shape == (2, 70656, 24) and cy.shape == (2, 141376) and iq.shape == (2, 24)
    assert np.array_equal(trace, ctx.pairing_witness_batch(H.g1_array(ps), H.g2_array(qs)))
    for e in range(2):
        tr, _ = W.pairing_witness(ps[e], qs[e])
        want_q, want_c, want_i = tr.aux()
        limbs24 = lambda v: [(v >> (32 * k)) & 0xffffffff for k in range(24)]
        assert np.array_equal(mq[e], np.array([limbs24(q) for q in want_q], dtype=np.uint32))
        assert np.array_equal(cy[e], np.array(want_c, dtype=np.uint8))
        assert iq[e].tolist() == limbs24(want_i)
        assert 0 < cy[e].sum() < 141376 and not mq[e][:, 12:].any()      # quotients are below p


def test_witness_auxiliary_outputs_may_be_null(ctx):
    """include/bls381_hip.h lets any of the three auxiliary outputs be NULL: every subset gives the same bytes in the
    outputs it does ask for as the call that asks for all three (and writes nowhere else: the trace is compared too)."""
    ps, qs = H.random_pairs(2, 910)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    full = ctx.pairing_witness_aux_batch(g1, g2)
    for mask in range(7):
        want = (bool(mask & 1), bool(mask & 2), bool(mask & 4))
        got = ctx.pairing_witness_aux_batch(g1, g2, want)
        assert np.array_equal(got[0], full[0]), mask
        for k in range(3):
            if want[k]:
                assert np.array_equal(got[1 + k], full[1 + k]), (mask, k)
            else:
                assert got[1 + k] is None


def test_fp_mul_kernel_against_big_integers(ctx):
    """The hand-scheduled Montgomery multiplier on 2^16 random pairs plus edge values (0, 1, p-1,
    R mod p, all-ones below 2p) against Python integers and against the plain-C form."""
    import ctypes as C
    import random
    from plonky2_bls12_381_pairing_amd import _lib
    rng = random.Random(99)
    edge = [0, 1, R.P - 1, R.MONT_R, R.P - 2, (1 << 381) - 1, 2 * R.P - 1, (1 << 382) - 1, 0xffffffff, 1 << 352]
    pairs = [(x, y) for x in edge for y in edge if x * y < R.P << 384]
    pairs += [(rng.randrange(R.P), rng.randrange(R.P)) for _ in range(65536 - len(pairs))]
    n = len(pairs)
    a = np.array([R.fq_to_u32_limbs(x) for x, _ in pairs], dtype=np.uint32)
    b = np.array([R.fq_to_u32_limbs(y) for _, y in pairs], dtype=np.uint32)
    out = np.empty_like(a)
    plain = np.empty_like(a)
    p32 = lambda v: v.ctypes.data_as(_lib.u32p)
    _lib.check(_lib.lib.bls381_debug_fp_mul(ctx.handle, p32(a), p32(b), n, p32(out), p32(plain)))
    rinv = pow(1 << 384, -1, R.P)
    want = np.array([R.fq_to_u32_limbs(x * y * rinv % R.P) for x, y in pairs], dtype=np.uint32)
    assert np.array_equal(out, want)
    assert np.array_equal(plain, want)


def test_fp_inverse_kernel_against_big_integers(ctx):
    """The division-step inversion on 2^16 values: edge cases (0, R mod p = one, p-1, 2, small and
    high-bit patterns) and random residues, against Python's modular inverse; zero is flagged."""
    import ctypes as C
    import random
    from plonky2_bls12_381_pairing_amd import _lib
    rng = random.Random(1234)
    edge = [0, 1, 2, 3, R.P - 1, R.P - 2, R.MONT_R, (R.P + 1) // 2, 1 << 380, (1 << 380) - 1, 0xffffffff, 1 << 32,
            1 << 352, (1 << 381) - 1 - ((1 << 381) - 1 >= R.P) * (1 << 200)]
    vals = [v % R.P for v in edge] + [1 << k for k in range(381)] + [R.P - (1 << k) for k in range(380)]
    vals += [rng.randrange(R.P) for _ in range(65536 - len(vals))]
    n = len(vals)
    a = np.array([R.fq_to_u32_limbs(x) for x in vals], dtype=np.uint32)
    out = np.empty_like(a)
    flags = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.lib.bls381_debug_fp_inverse(ctx.handle, a.ctypes.data_as(_lib.u32p), n, out.ctypes.data_as(_lib.u32p),
                                                flags.ctypes.data_as(C.POINTER(C.c_uint8))))
    r2 = pow(1 << 384, 2, R.P)      # (x R)^-1 R^2 = x^-1 R
    want = np.array([R.fq_to_u32_limbs(pow(x, -1, R.P) * r2 % R.P if x else 0) for x in vals], dtype=np.uint32)
    assert np.array_equal(out, want)
    assert flags.tolist() == [int(x == 0) for x in vals]


def test_fixed_q_mode_against_general_path_and_oracle(ctx):
    """One prepared G2 point against many G1 points (G2Prepared reuse, miller_loop.rs:13-16): Miller
    values and pairings equal the general entry points on (P_i, Q) bit for bit, equal the Python oracle
    on a sample, honour the infinity flags, and a prepared point at infinity yields one everywhere."""
    n = 300
    ps, qs = H.random_pairs(n, 2024)
    q = qs[0]
    g1 = H.g1_array(ps)
    g2 = np.repeat(H.g2_array([q]), n, axis=0)
    flags = np.zeros(n, dtype=np.uint8)
    flags[[3, 77, 299]] = 1
    prep = ctx.g2_prepared(H.g2_array([q])[0])
    ml = ctx.miller_loop_fixed_q_batch(prep, g1, flags)
    assert np.array_equal(ml, ctx.miller_loop_batch(g1, g2, flags))
    pr = ctx.pairing_fixed_q_batch(prep, g1, flags)
    assert np.array_equal(pr, ctx.pairing_batch(g1, g2, flags))
    for i in (0, 1, 150):
        assert H.fq12_from_row(ml[i]) == R.multi_miller_loop([ps[i]], [q])
        assert H.fq12_from_row(pr[i]) == R.native_pairing([ps[i]], [q])
    one = H.fq12_array([R.FQ12_ONE])[0]
    assert np.array_equal(pr[3], one) and np.array_equal(ml[77], one)
    prep.close()
    inf = ctx.g2_prepared(None, infinity=True)
    assert all(np.array_equal(r, one) for r in ctx.pairing_fixed_q_batch(inf, g1[:5]))
    inf.close()
    assert ctx.pairing_fixed_q_batch(ctx.g2_prepared(H.g2_array([q])[0]), g1[:0]).shape == (0, 72)


def test_other_witness_generators(ctx):
    """FqSqrt / Fq2Sqrt / Fq2Inverse / Fq6Inverse generator values on canonical limbs against the Python
    oracle: random inputs, both requested signs, the edge cases (0, 1, p-1, pure-real and pure-imaginary
    Fq2 squares), and the inputs on which the reference panics (non-residues, zero in Fq6, limbs >= p)."""
    import random
    import plonky2_bls12_381_pairing_amd as B
    rng = random.Random(55)
    L = lambda v: R.fq_to_u32_limbs(v)
    # Fq square roots
    xs = [0, 1, 4, R.P - 1, 2] + [rng.randrange(R.P) ** 2 % R.P for _ in range(150)] + [rng.randrange(R.P) for _ in range(150)]
    sg = [0, 1, 0, 1, 1] + [rng.randrange(2) for _ in range(300)]
    out, err = ctx.fq_sqrt_limbs_batch(np.array([L(x) for x in xs], dtype=np.uint32), sg, return_err=True)
    n_err = 0
    for x, s, o, e in zip(xs, sg, out, err):
        try:
            want = R.fq_sqrt_generator(x, s)
            assert e == 0 and R.u32_limbs_to_int(o) == want
        except ValueError:
            assert e == 1 and not o.any()
            n_err += 1
    assert 50 < n_err < 250
    # Fq2 square roots
    pts = [(0, 0), (1, 0), (R.P - 1, 0), (0, 1), (0, R.P - 4), (9, 0)]
    pts += [R.fq2_square((rng.randrange(R.P), rng.randrange(R.P))) for _ in range(150)]
    pts += [(rng.randrange(R.P), rng.randrange(R.P)) for _ in range(100)]
    sg = [rng.randrange(2) for _ in pts]
    sg[0] = 0
    out, err = ctx.fq2_sqrt_limbs_batch(np.array([L(a) + L(b) for a, b in pts], dtype=np.uint32), sg, return_err=True)
    for a, s, o, e in zip(pts, sg, out, err):
        try:
            want = R.fq2_sqrt_generator(a, s)
            assert e == 0 and (R.u32_limbs_to_int(o[:12]), R.u32_limbs_to_int(o[12:])) == want, (a, s)
        except ValueError:
            assert e == 1 and not o.any()
    assert 20 < int(err.sum()) < 90
    # Fq2 inverses: zero maps to zero without an error
    pts = [(0, 0), (1, 0), (0, 1), (R.P - 1, R.P - 1)] + [(rng.randrange(R.P), rng.randrange(R.P)) for _ in range(200)]
    out = ctx.fq2_inverse_limbs_batch(np.array([L(a) + L(b) for a, b in pts], dtype=np.uint32))
    for a, o in zip(pts, out):
        assert (R.u32_limbs_to_int(o[:12]), R.u32_limbs_to_int(o[12:])) == R.fq2_inverse_generator(a)
    # Fq6 inverses in MyFq6 order; zero and out-of-range limbs are errors
    cs = [[1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], [0, 1, 0, 0, 0, 0]] + [[rng.randrange(R.P) for _ in range(6)] for _ in range(100)]
    out = ctx.fq6_inverse_limbs_batch(np.array([sum((L(v) for v in c), []) for c in cs], dtype=np.uint32))
    for c, o in zip(cs, out):
        assert [R.u32_limbs_to_int(o[12 * k:12 * k + 12]) for k in range(6)] == R.fq6_inverse_generator(c)
    bad = np.array([sum((L(v) for v in c), []) for c in ([0] * 6, cs[5])], dtype=np.uint32)
    bad[1, :12] = L(R.P)
    with pytest.raises(B.Bls381Error):
        ctx.fq6_inverse_limbs_batch(bad)
    out, err = ctx.fq6_inverse_limbs_batch(bad, return_err=True)
    assert err.tolist() == [1, 1] and not out.any()


def test_witness_trace_device_pointer_form(ctx):
    """bls381_pairing_witness_batch_dev writes the trace in place in device memory: same bytes as the
    host-pointer form."""
    import torch
    from plonky2_bls12_381_pairing_amd import _lib
    ps, qs = H.random_pairs(3, 606)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    want = ctx.pairing_witness_batch(g1, g2)
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.zeros(want.size, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.bls381_pairing_witness_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), 3, out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(want.shape), want)


@pytest.mark.parametrize("lanes,slots", [(6, 0), (9, 0), (12, 0), (16, 0), (8, 36)])
def test_other_lane_group_geometries(lanes, slots):
    """Every compiled lane-group width (and a reduced LDS slot count, which forces more spilling) computes
    the same pairings and fixed-Q pairings as the oracle."""
    import plonky2_bls12_381_pairing_amd as B
    ps, qs = H.random_pairs(11, 1000 + lanes)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    with B.Context(0, lanes, slots) as c:
        out = c.pairing_batch(g1, g2)
        for i in (0, 5, 10):
            assert H.fq12_from_row(out[i]) == R.native_pairing([ps[i]], [qs[i]])
        prep = c.g2_prepared(g2[2])
        assert np.array_equal(c.pairing_fixed_q_batch(prep, g1[2:3]), out[2:3])
        prep.close()


@pytest.mark.parametrize("knob,value", [("BLS381_REDUCE_SIBLINGS", "0"), ("BLS381_NO_PIGGYBACK", "1"), ("BLS381_WINDOW", "200"),
                                        ("BLS381_ABSORB_TERMS", "1"), ("BLS381_VB_EAGER", "50"), ("BLS381_NO_STEP_PAIRS", "1")])
def test_other_schedules_give_the_same_values(ctx, knob, value, monkeypatch):
    """The scheduler's policies (which sums are reduced together, copies riding on compute steps or in steps of their
    own, the look-ahead window, sums folded into multiplier operands or not, eager reductions) change the step programs,
    never the values, and neither does running every sum step on descriptors of its own (BLS381_NO_STEP_PAIRS): a context
    built under another policy returns the default context's bytes for pairings, Miller values and final exponentiations
    of arbitrary input."""
    import plonky2_bls12_381_pairing_amd as B
    ps, qs = H.random_pairs(37, 4242)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    want, want_ml = ctx.pairing_batch(g1, g2), ctx.miller_loop_batch(g1, g2)
    x = H.fq12_array(H.random_fq12(5, 77))
    want_fe = ctx.final_exponentiation_batch(x)
    monkeypatch.setenv(knob, value)
    with B.Context(0) as c:
        assert np.array_equal(c.pairing_batch(g1, g2), want)
        assert np.array_equal(c.miller_loop_batch(g1, g2), want_ml)
        assert np.array_equal(c.final_exponentiation_batch(x), want_fe)


def test_lazy_product_programs(monkeypatch):
    """The 6-lane programs built from n-product operations (schoolbook Fq6 products and line products, one reduction
    per component) give the same pairings, Miller values and Fq12 products, bit for bit, as the same width with
    BLS381_LAZY_PRODUCTS=0 (Karatsuba form)."""
    import plonky2_bls12_381_pairing_amd as B
    ps, qs = H.random_pairs(70, 7706)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    monkeypatch.setenv("BLS381_LAZY_PRODUCTS", "0")
    with B.Context(0, 6, 0) as c:
        want, want_ml = c.pairing_batch(g1, g2), c.miller_loop_batch(g1, g2)
    monkeypatch.delenv("BLS381_LAZY_PRODUCTS")
    with B.Context(0, 6, 0) as c:
        assert np.array_equal(c.pairing_batch(g1, g2), want)
        assert np.array_equal(c.miller_loop_batch(g1, g2), want_ml)
        prod = c.fq12_mul_batch(want[:35], want[35:])
        for i in (0, 17, 34):
            assert H.fq12_from_row(prod[i]) == R.fq12_mul(H.fq12_from_row(want[i]), H.fq12_from_row(want[35 + i]))
    for i in (0, 69):
        assert H.fq12_from_row(want[i]) == R.native_pairing([ps[i]], [qs[i]])


def test_three_chunk_batch_on_device_pointers(ctx, CO):
    """135,000 pairs (two chunks of 65,536 and one of 3,928, which runs the 16-lane program variants) through the
    device-pointer calls: the C oracle on a sample that touches every chunk and both sides of every boundary,
    periodicity over the whole output, and the Miller values of the same batch against a single small call."""
    import torch
    from plonky2_bls12_381_pairing_amd import _lib
    n, period = 135000, 1000
    ps, qs = H.random_pairs(40, 6161)
    a, b = H.g1_array(ps), H.g2_array(qs)
    i = np.arange(period)
    g1 = np.tile(a[i % 40], (n // period, 1))
    g2 = np.tile(b[(i // 40) % 40], (n // period, 1))
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    d_out = torch.empty((n, 72), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib.bls381_pairing_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), None, n, d_out.data_ptr()))
    out = d_out.cpu().numpy().view(np.uint64)
    probe = np.array([0, 1, 65535, 65536, 65537, 131071, 131072, 131073, n - 2, n - 1])
    assert np.array_equal(out[probe], CO.pairing_batch(g1[probe], g2[probe]))
    assert np.array_equal(out.reshape(n // period, period, 72), np.broadcast_to(out[:period], (n // period, period, 72)))
    _lib.check(_lib.lib.bls381_miller_loop_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), None, n, d_out.data_ptr()))
    ml = d_out.cpu().numpy().view(np.uint64)
    assert np.array_equal(ml[:period], ctx.miller_loop_batch(g1[:period], g2[:period]))
    assert np.array_equal(ml.reshape(n // period, period, 72), np.broadcast_to(ml[:period], (n // period, period, 72)))


def _bytes_array(encodings):
    return np.frombuffer(b"".join(encodings), dtype=np.uint8).reshape(len(encodings), -1)


def test_compressed_points_and_aggregation(ctx, monkeypatch):
    """Wire format and aggregation against the big-integer oracle: every decoded limb, status byte,
    encoded byte and aggregate is compared; the device's endomorphism subgroup tests are checked
    against [r]P = O on curve points inside and outside the subgroup."""
    import random
    import bls381_points as PT
    rng = random.Random(2024)
    ks = [rng.randrange(1, R.R_ORDER) for _ in range(6)]
    g1s = [R.G1_GEN] + [R.g1_mul(R.G1_GEN, k) for k in ks] + [R.g1_neg(R.g1_mul(R.G1_GEN, ks[0])), None]
    g2s = [R.G2_GEN] + [R.g2_mul(R.G2_GEN, k) for k in ks[:3]] + [R.g2_neg(R.g2_mul(R.G2_GEN, ks[0])), None]

    # ---- G1 decode
    enc = [PT.g1_compress(p) for p in g1s]
    assert enc[0] == PT.G1_GEN_COMPRESSED
    bad = []
    b = bytearray(PT.G1_GEN_COMPRESSED); b[0] &= 0x7F; bad.append(bytes(b))           # not the compressed form
    bad.append(bytes([0xE0]) + bytes(47))                                            # infinity + sort flag
    bad.append(bytes([0xC0]) + bytes(46) + b"\x01")                                  # infinity with x != 0
    tb = bytearray(R.P.to_bytes(48, "big")); tb[0] |= 0x80; bad.append(bytes(tb))     # x = p
    one = bytearray((1).to_bytes(48, "big")); one[0] |= 0x80; bad.append(bytes(one))  # 5 is not a square
    outside = []
    x = 4
    while len(outside) < 3:
        if R.fq_sqrt((x ** 3 + 4) % R.P) is not None:
            e = bytearray(x.to_bytes(48, "big")); e[0] |= 0x80 | (0x20 if len(outside) % 2 else 0)
            outside.append(bytes(e))
        x += 1
    allenc = enc + bad + outside
    for check, form in ((True, None), (True, "0"), (False, None)):   # form "0": the subgroup test inside the decoder
        if form is not None:
            monkeypatch.setenv("BLS381_G1_SUBGROUP_PROGRAM", form)
        want = [PT.g1_decompress(e, check_subgroup=check) for e in allenc]
        out, status = ctx.g1_decompress_batch(_bytes_array(allenc), check_subgroup=check, return_status=True)
        monkeypatch.delenv("BLS381_G1_SUBGROUP_PROGRAM", raising=False)
        assert status.tolist() == [s for s, _ in want]
        assert np.array_equal(out, H.g1_array([p for _, p in want]))
    assert [s for s, _ in [PT.g1_decompress(e) for e in outside]] == [PT.NOT_IN_SUBGROUP] * 3
    for count in (300, 8300):   # (as for G2 below: 16 and 4 lanes per point)
        many = [outside[i % 3] if i % 11 == 3 else enc[i % (len(enc) - 1)] for i in range(count)]
        out, status = ctx.g1_decompress_batch(_bytes_array(many), return_status=True)
        assert status.tolist() == [PT.NOT_IN_SUBGROUP if i % 11 == 3 else 0 for i in range(count)]
        assert all(not out[i].any() for i in range(count) if i % 11 == 3)
    with pytest.raises(Exception, match="subgroup"):
        ctx.g1_decompress_batch(_bytes_array(outside))
    good, status = ctx.g1_decompress_batch(_bytes_array(enc))
    assert np.array_equal(good, H.g1_array(g1s)) and status.tolist() == [0] * (len(g1s) - 1) + [1]

    # ---- G2 decode
    enc2 = [PT.g2_compress(q) for q in g2s]
    assert enc2[0] == PT.G2_GEN_COMPRESSED
    bad2 = []
    b = bytearray(PT.G2_GEN_COMPRESSED); b[0] &= 0x7F; bad2.append(bytes(b))
    bad2.append(bytes([0xC0]) + bytes(94) + b"\x01")
    tb = bytearray(bytes(48) + R.P.to_bytes(48, "big")); tb[0] |= 0x80; bad2.append(bytes(tb))   # x.c0 = p
    outside2, off_curve2 = [], []
    x0 = 1
    while len(outside2) < 2 or len(off_curve2) < 1:
        xx = (x0, 1)
        rhs = R.fq2_add(R.fq2_mul(R.fq2_square(xx), xx), (4, 4))
        e = bytearray((1).to_bytes(48, "big") + x0.to_bytes(48, "big")); e[0] |= 0x80
        if R.fq2_pow(rhs, (R.P * R.P - 1) // 2) == R.FQ2_ONE:
            if len(outside2) < 2:
                outside2.append(bytes(e))
        elif len(off_curve2) < 1:
            off_curve2.append(bytes(e))
        x0 += 1
    allenc2 = enc2 + bad2 + off_curve2 + outside2
    for check, form in ((True, None), (True, "0"), (False, None)):   # form "0": the subgroup test inside the decoder
        if form is not None:
            monkeypatch.setenv("BLS381_G2_SUBGROUP_PROGRAM", form)
        want = [PT.g2_decompress(e, check_subgroup=check) for e in allenc2]
        out, status = ctx.g2_decompress_batch(_bytes_array(allenc2), check_subgroup=check, return_status=True)
        monkeypatch.delenv("BLS381_G2_SUBGROUP_PROGRAM", raising=False)
        assert status.tolist() == [s for s, _ in want]
        assert np.array_equal(out, H.g2_array([q for _, q in want]))
    # batches that are not one wave, every eleventh point outside the subgroup: 300 points (16 lanes per point) and
    # 8,300 (past the size from which the test program runs at four lanes per point)
    for count in (300, 8300):
        many = [outside2[i % 2] if i % 11 == 3 else enc2[i % (len(enc2) - 1)] for i in range(count)]
        out, status = ctx.g2_decompress_batch(_bytes_array(many), return_status=True)
        assert status.tolist() == [PT.NOT_IN_SUBGROUP if i % 11 == 3 else 0 for i in range(count)]
        assert all(not out[i].any() for i in range(count) if i % 11 == 3)
        expected = H.g2_array(g2s[:-1])
        assert all(np.array_equal(out[i], expected[i % len(expected)]) for i in range(count) if i % 11 != 3)
    assert PT.g2_decompress(outside2[0])[0] == PT.NOT_IN_SUBGROUP and PT.g2_decompress(off_curve2[0])[0] == PT.NOT_ON_CURVE

    # ---- encode
    inf1 = np.array([1 if p is None else 0 for p in g1s], dtype=np.uint8)
    assert ctx.g1_compress_batch(H.g1_array(g1s), inf1).tobytes() == b"".join(enc)
    inf2 = np.array([1 if q is None else 0 for q in g2s], dtype=np.uint8)
    assert ctx.g2_compress_batch(H.g2_array(g2s), inf2).tobytes() == b"".join(enc2)

    # ---- aggregation: groups of 1, 2, 7 and 512 points, with infinity members, repeats and opposite points
    base = [R.g1_mul(R.G1_GEN, k) for k in range(1, 9)]
    for size in (1, 2, 7, 512):
        groups = []
        for g in range(3):
            members = [base[rng.randrange(8)] for _ in range(size)]
            if size >= 2 and g == 1:
                members[0] = None
                members[1] = members[-1]                      # a repeated point: the doubling case
            if size == 2 and g == 2:
                members = [base[3], R.g1_neg(base[3])]        # sums to infinity
            groups.append(members)
        flat = [p for grp in groups for p in grp]
        inf = np.array([1 if p is None else 0 for p in flat], dtype=np.uint8)
        out, out_inf = ctx.g1_aggregate_batch(H.g1_array(flat), size, inf)
        want = [PT.g1_aggregate(grp) for grp in groups]
        assert out_inf.tolist() == [1 if w is None else 0 for w in want]
        assert np.array_equal(out, H.g1_array(want))
    base2 = [R.g2_mul(R.G2_GEN, k) for k in range(1, 5)]
    for size in (1, 5, 64):
        groups = [[base2[rng.randrange(4)] for _ in range(size)] for _ in range(2)]
        if size >= 5:
            groups[1][2] = None
            groups[1][3] = R.g2_neg(groups[1][4])
        flat = [q for grp in groups for q in grp]
        inf = np.array([1 if q is None else 0 for q in flat], dtype=np.uint8)
        out, out_inf = ctx.g2_aggregate_batch(H.g2_array(flat), size, inf)
        want = [PT.g2_aggregate(grp) for grp in groups]
        assert out_inf.tolist() == [1 if w is None else 0 for w in want]
        assert np.array_equal(out, H.g2_array(want))
    # decoded keys feed the pairing: e(sum of keys, Q) against the oracle
    keys = [R.g1_mul(R.G1_GEN, k) for k in (2, 3, 5)]
    dec, _ = ctx.g1_decompress_batch(_bytes_array([PT.g1_compress(p) for p in keys]))
    agg, agg_inf = ctx.g1_aggregate_batch(dec, 3)
    assert agg_inf.tolist() == [0]
    got = ctx.pairing_batch(agg, H.g2_array([R.G2_GEN]))
    assert H.fq12_from_row(got[0]) == R.native_pairing([R.g1_mul(R.G1_GEN, 10)], [R.G2_GEN])


def test_two_pairs_per_item(ctx, CO):
    """native_pairing over two pairs per item (the signature-check shape) against the oracle's
    multi_miller_loop / native_pairing, with pairs at infinity, valid and invalid signatures."""
    import random
    rng = random.Random(77)
    n = 9
    pa, qa = H.random_pairs(n, 501)
    pb, qb = H.random_pairs(n, 502)
    flags = np.zeros(n, dtype=np.uint8)
    pa[1] = None; flags[1] = 1           # first pair skipped
    qb[2] = None; flags[2] = 8           # second pair skipped
    pa[3] = None; qb[3] = None; flags[3] = 1 | 8   # both skipped -> one
    # items 4, 5: a valid signature check  e(pk, H) * e(-G1, sk * H) = 1, item 6: a wrong signature
    for i in (4, 5, 6):
        sk, h = rng.randrange(1, R.R_ORDER), rng.randrange(1, R.R_ORDER)
        hm = R.g2_mul(R.G2_GEN, h)
        pa[i], qa[i] = R.g1_mul(R.G1_GEN, sk), hm
        pb[i], qb[i] = R.g1_neg(R.G1_GEN), R.g2_mul(hm, sk if i != 6 else sk + 1)
    arrs = (H.g1_array(pa), H.g2_array(qa), H.g1_array(pb), H.g2_array(qb))
    ml = ctx.miller_loop_two_pairs_batch(*arrs, inf_flags=flags)
    out, is_one = ctx.pairing_two_pairs_batch(*arrs, inf_flags=flags)
    for i in range(n):
        ps = [p for p, q in ((pa[i], qa[i]), (pb[i], qb[i])) if p is not None and q is not None]
        qs = [q for p, q in ((pa[i], qa[i]), (pb[i], qb[i])) if p is not None and q is not None]
        assert H.fq12_from_row(ml[i]) == R.multi_miller_loop(ps, qs), i
        want = R.native_pairing(ps, qs)
        assert H.fq12_from_row(out[i]) == want, i
        assert bool(is_one[i]) == (want == R.FQ12_ONE)
    assert is_one.tolist() == [0, 0, 0, 1, 1, 1, 0, 0, 0]
    # a larger batch against the per-pair path: product of the two single Miller values, then the same final exponentiation
    m = 2048 + 37
    g1 = np.tile(arrs[0][[0, 7, 8]], (m // 3 + 1, 1))[:m]; g2 = np.tile(arrs[1][[0, 7, 8]], (m // 3 + 1, 1))[:m]
    h1 = np.tile(arrs[2][[8, 0, 7]], (m // 3 + 1, 1))[:m]; h2 = np.tile(arrs[3][[8, 0, 7]], (m // 3 + 1, 1))[:m]
    got, _ = ctx.pairing_two_pairs_batch(g1, g2, h1, h2)
    prod = ctx.fq12_mul_batch(ctx.miller_loop_batch(g1, g2), ctx.miller_loop_batch(h1, h2))
    assert np.array_equal(got, ctx.final_exponentiation_batch(prod))


def test_point_format_golden_vectors(ctx):
    v = json.load(open(os.path.join(os.path.dirname(GOLDEN), "point_vectors.json")))
    enc1 = [bytes.fromhex(c["g1_bytes"]) for c in v["cases"]] + [bytes.fromhex(v["infinity_g1"])]
    enc2 = [bytes.fromhex(c["g2_bytes"]) for c in v["cases"]] + [bytes.fromhex(v["infinity_g2"])]
    g1, st1 = ctx.g1_decompress_batch(_bytes_array(enc1))
    g2, st2 = ctx.g2_decompress_batch(_bytes_array(enc2))
    n = len(v["cases"])
    assert st1.tolist() == [0] * n + [1] and st2.tolist() == [0] * n + [1]
    want1 = [tuple(int(x, 16) for x in c["g1"]) for c in v["cases"]] + [None]
    want2 = [((int(c["g2"][0], 16), int(c["g2"][1], 16)), (int(c["g2"][2], 16), int(c["g2"][3], 16))) for c in v["cases"]] + [None]
    assert np.array_equal(g1, H.g1_array(want1)) and np.array_equal(g2, H.g2_array(want2))
    assert ctx.g1_compress_batch(g1, st1).tobytes() == b"".join(enc1)
    assert ctx.g2_compress_batch(g2, st2).tobytes() == b"".join(enc2)
    s1, i1 = ctx.g1_aggregate_batch(g1, n + 1, st1)
    s2, i2 = ctx.g2_aggregate_batch(g2, n + 1, st2)
    assert ctx.g1_compress_batch(s1, i1).tobytes().hex() == v["sum_g1"]
    assert ctx.g2_compress_batch(s2, i2).tobytes().hex() == v["sum_g2"]
    _, st = ctx.g1_decompress_batch(_bytes_array([bytes.fromhex(e) for e in v["on_curve_outside_subgroup_g1"]]), return_status=True)
    assert st.tolist() == [4, 4]


def test_signature_checks_from_bytes(ctx, monkeypatch):
    """bls381_verify_batch against the oracle: decode with oracle/py/bls381_points.py, then
    native_pairing([pk, -G1], [msg, sig]) == 1 with the big-integer pairing."""
    import random
    import bls381_points as PT
    rng = random.Random(4242)
    pks, msgs, sigs = [], [], []
    for i in range(4):
        sk, h = rng.randrange(1, R.R_ORDER), rng.randrange(1, R.R_ORDER)
        hm = R.g2_mul(R.G2_GEN, h)
        pks.append(PT.g1_compress(R.g1_mul(R.G1_GEN, sk)))
        msgs.append(PT.g2_compress(hm))
        sigs.append(PT.g2_compress(R.g2_mul(hm, sk if i != 1 else sk + 1)))      # item 1: wrong signature
    # item 4: key at infinity with signature at infinity (the product is one, the key is still rejected)
    pks.append(PT.g1_compress(None)); msgs.append(msgs[0]); sigs.append(PT.g2_compress(None))
    # item 5: signature at infinity under a real key
    pks.append(pks[0]); msgs.append(msgs[0]); sigs.append(PT.g2_compress(None))
    # item 6: malformed key (x = p); item 7: key on the curve but outside the subgroup
    tb = bytearray(R.P.to_bytes(48, "big")); tb[0] |= 0x80
    pks.append(bytes(tb)); msgs.append(msgs[2]); sigs.append(sigs[2])
    x = 4
    while R.fq_sqrt((x ** 3 + 4) % R.P) is None:
        x += 1
    e = bytearray(x.to_bytes(48, "big")); e[0] |= 0x80
    pks.append(bytes(e)); msgs.append(msgs[2]); sigs.append(sigs[2])
    # item 8: a valid triple again, after the rejects
    pks.append(pks[3]); msgs.append(msgs[3]); sigs.append(sigs[3])
    # item 9: message point and signature both at infinity under a real key: the product would be empty (one);
    # a message point at infinity is rejected
    pks.append(pks[0]); msgs.append(PT.g2_compress(None)); sigs.append(PT.g2_compress(None))
    # items 10, 11: a message point, then a signature, on the twist but outside the subgroup
    x0 = 1
    while True:
        xx = (x0, 1)
        if R.fq2_pow(R.fq2_add(R.fq2_mul(R.fq2_square(xx), xx), (4, 4)), (R.P * R.P - 1) // 2) == R.FQ2_ONE:
            break
        x0 += 1
    e = bytearray((1).to_bytes(48, "big") + x0.to_bytes(48, "big")); e[0] |= 0x80
    assert PT.g2_decompress(bytes(e))[0] == PT.NOT_IN_SUBGROUP
    pks.append(pks[0]); msgs.append(bytes(e)); sigs.append(sigs[0])
    pks.append(pks[0]); msgs.append(msgs[0]); sigs.append(bytes(e))
    verdict, status = ctx.verify_batch(_bytes_array(pks), _bytes_array(msgs), _bytes_array(sigs))
    want_v, want_s = [], []
    neg_g1 = R.g1_neg(R.G1_GEN)
    for pk, m, s in zip(pks, msgs, sigs):
        (s1, p), (s2, hm), (s3, sg) = PT.g1_decompress(pk), PT.g2_decompress(m), PT.g2_decompress(s)
        want_s.append([s1, s2, s3])
        if s1 != PT.OK or s2 != PT.OK or s3 > PT.INFINITY:
            want_v.append(0)
            continue
        pairs = [(p, hm), (neg_g1, sg)]
        pairs = [(a, b) for a, b in pairs if a is not None and b is not None]
        want_v.append(1 if R.native_pairing([a for a, _ in pairs], [b for _, b in pairs]) == R.FQ12_ONE else 0)
    assert status.tolist() == want_s
    assert verdict.tolist() == want_v == [1, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0]
    assert status[10].tolist() == [0, PT.NOT_IN_SUBGROUP, 0] and status[11].tolist() == [0, 0, PT.NOT_IN_SUBGROUP]
    # the same with the G2 subgroup tests inside the decoders (one lane per point) instead of the step program
    monkeypatch.setenv("BLS381_G2_SUBGROUP_PROGRAM", "0")
    monkeypatch.setenv("BLS381_G1_SUBGROUP_PROGRAM", "0")
    v1, s1 = ctx.verify_batch(_bytes_array(pks), _bytes_array(msgs), _bytes_array(sigs))
    monkeypatch.delenv("BLS381_G2_SUBGROUP_PROGRAM")
    monkeypatch.delenv("BLS381_G1_SUBGROUP_PROGRAM")
    assert v1.tolist() == want_v and s1.tolist() == want_s
    # without the subgroup test the off-subgroup key decodes, and its check fails in the pairing instead
    v2, s2 = ctx.verify_batch(_bytes_array(pks), _bytes_array(msgs), _bytes_array(sigs), check_subgroup=False)
    assert s2[7].tolist() == [0, 0, 0] and s2[10].tolist() == [0, 0, 0] and v2.tolist() == want_v


def test_new_entry_points_edge_sizes(ctx, CO):
    """Empty inputs, and a two-pair batch that spans two chunks with infinity flags in the second one."""
    import bls381_points as PT
    z8 = np.zeros((0, 48), dtype=np.uint8)
    out, st = ctx.g1_decompress_batch(z8)
    assert out.shape == (0, 12) and st.shape == (0,)
    assert ctx.g2_compress_batch(np.zeros((0, 24), dtype=np.uint64)).shape == (0, 96)
    agg, inf = ctx.g1_aggregate_batch(np.zeros((0, 12), dtype=np.uint64), 4)
    assert agg.shape == (0, 12) and inf.shape == (0,)
    v, s = ctx.verify_batch(z8, np.zeros((0, 96), dtype=np.uint8), np.zeros((0, 96), dtype=np.uint8))
    assert v.shape == (0,) and s.shape == (0, 3)
    e = np.zeros((0, 12), dtype=np.uint64); f = np.zeros((0, 24), dtype=np.uint64)
    o, one = ctx.pairing_two_pairs_batch(e, f, e, f)
    assert o.shape == (0, 72) and one.shape == (0,)
    with pytest.raises(ValueError):
        ctx.g1_aggregate_batch(np.zeros((5, 12), dtype=np.uint64), 4)      # not a whole number of groups

    n = 65536 + 70
    ps, qs = H.random_pairs(5, 808)
    g1 = np.tile(H.g1_array(ps), (n // 5 + 1, 1))[:n]; g2 = np.tile(H.g2_array(qs), (n // 5 + 1, 1))[:n]
    h1 = np.roll(g1, 1, 0); h2 = np.roll(g2, 2, 0)
    flags = np.zeros(n, dtype=np.uint8)
    flags[65536 + 3] = 1          # first pair skipped
    flags[65536 + 11] = 4         # second pair skipped
    flags[65536 + 20] = 2 | 8     # both skipped
    flags[17] = 8
    got, is_one = ctx.pairing_two_pairs_batch(g1, g2, h1, h2, inf_flags=flags)
    idx = np.array([0, 17, 65535, 65536, 65536 + 3, 65536 + 11, 65536 + 20, n - 1])
    ma, mb = CO.pairing_batch(g1[idx], g2[idx], final_exp=False), CO.pairing_batch(h1[idx], h2[idx], final_exp=False)
    one = H.fq12_array([R.FQ12_ONE])[0]
    for k, i in enumerate(idx):
        if flags[i] & 3: ma[k] = one
        if flags[i] & 12: mb[k] = one
    want = ctx.final_exponentiation_batch(ctx.fq12_mul_batch(ma, mb))
    assert np.array_equal(got[idx], want)
    assert is_one[65536 + 20] == 1 and is_one.sum() == 1


def test_one_gpu_shard_of_the_8m_configuration(ctx, CO):
    """1,048,576 pairs (what one of 8 GPUs gets of 8,388,608) through the host-pointer call: a strided sample
    against the CPU oracle, and the periodicity of the synthetic input as a whole-output property."""
    n, period = 1 << 20, 4096
    ps, qs = H.random_pairs(64, 3003)
    a, b = H.g1_array(ps), H.g2_array(qs)
    i = np.arange(period)
    g1p, g2p = a[i % 64], b[(i // 64) % 64]                 # 4,096 distinct pairs
    g1, g2 = np.tile(g1p, (n // period, 1)), np.tile(g2p, (n // period, 1))
    out = ctx.pairing_batch(g1, g2)
    sample = np.arange(0, n, n // 128) + 5                    # 1/8192 of the outputs, every chunk touched
    assert np.array_equal(out[sample], CO.pairing_batch(g1[sample], g2[sample]))
    first = out[:period]
    assert np.array_equal(out.reshape(n // period, period, 72), np.broadcast_to(first, (n // period, period, 72)))
    assert np.array_equal(first[:64], CO.pairing_batch(g1p[:64], g2p[:64]))


def test_fast_aggregate_verify_from_bytes(ctx):
    """512 signers, one message: keys decoded and summed on the device, then the two-pair check; against
    signatures built with the oracle's group arithmetic."""
    import random
    import bls381_points as PT
    rng = random.Random(512)
    sks = [rng.randrange(1, R.R_ORDER) for _ in range(512)]
    # 512 scalar multiplications in Python are slow: derive the keys from a short chain instead
    base = R.g1_mul(R.G1_GEN, sks[0])
    keys, ks, acc, k = [], [], base, sks[0]
    step_k = rng.randrange(1, R.R_ORDER)
    step = R.g1_mul(R.G1_GEN, step_k)
    for _ in range(512):
        keys.append(acc); ks.append(k)
        acc = R.g1_add(acc, step); k = (k + step_k) % R.R_ORDER
    hm = R.g2_mul(R.G2_GEN, rng.randrange(1, R.R_ORDER))
    total = sum(ks) % R.R_ORDER
    sig = R.g2_mul(hm, total)
    pk = _bytes_array([PT.g1_compress(p) for p in keys])
    msg, good, bad = PT.g2_compress(hm), PT.g2_compress(sig), PT.g2_compress(R.g2_mul(hm, total + 1))
    valid, st = ctx.fast_aggregate_verify(pk, msg, good)
    assert valid and not st.any()
    assert not ctx.fast_aggregate_verify(pk, msg, bad)[0]
    assert ctx.fast_aggregate_verify(pk, msg, good, check_subgroup=False)[0]
    # a subset of 3 and of 1 signers
    for cnt in (3, 1):
        s = PT.g2_compress(R.g2_mul(hm, sum(ks[:cnt]) % R.R_ORDER))
        assert ctx.fast_aggregate_verify(pk[:cnt], msg, s)[0]
        assert not ctx.fast_aggregate_verify(pk[:cnt + 1], msg, s)[0]
    # a key at infinity or a malformed key invalidates the whole check, and is reported
    broken = pk.copy(); broken[7] = np.frombuffer(PT.g1_compress(None), dtype=np.uint8)
    valid, st = ctx.fast_aggregate_verify(broken, msg, good)
    assert not valid and st[7] == PT.INFINITY and st.sum() == 1
    broken = pk.copy(); broken[100, 0] &= 0x7F
    valid, st = ctx.fast_aggregate_verify(broken, msg, good)
    assert not valid and st[100] == PT.BAD_ENCODING
    assert ctx.fast_aggregate_verify(pk[:0], msg, good)[0] is False


def test_fast_aggregate_verify_batch(ctx):
    """Four aggregate checks of 64 signers in one call: valid, wrong signature, a malformed key, and keys that sum
    to infinity: KeyValidate on the aggregate rejects that check whatever the signature (with the signature at
    infinity the product would be empty, i.e. one: the {pk, -pk} forgery)."""
    import random
    import bls381_points as PT
    rng = random.Random(64)
    step_k = rng.randrange(1, R.R_ORDER)
    step = R.g1_mul(R.G1_GEN, step_k)
    k0 = rng.randrange(1, R.R_ORDER)
    keys, ks, acc, k = [], [], R.g1_mul(R.G1_GEN, k0), k0
    for _ in range(64):
        keys.append(acc); ks.append(k)
        acc = R.g1_add(acc, step); k = (k + step_k) % R.R_ORDER
    enc = [PT.g1_compress(p) for p in keys]
    hm = [R.g2_mul(R.G2_GEN, rng.randrange(1, R.R_ORDER)) for _ in range(4)]
    total = sum(ks) % R.R_ORDER
    opposite = [PT.g1_compress(p) for p in keys[:32]] + [PT.g1_compress(R.g1_neg(p)) for p in keys[:32]]
    pk = np.stack([_bytes_array(enc), _bytes_array(enc), _bytes_array(enc), _bytes_array(opposite)])
    pk[2, 9, 0] &= 0x7F                                              # check 2: key 9 is not in compressed form
    msg = _bytes_array([PT.g2_compress(h) for h in hm])
    sig = _bytes_array([PT.g2_compress(R.g2_mul(hm[0], total)), PT.g2_compress(R.g2_mul(hm[1], total + 1)),
                        PT.g2_compress(R.g2_mul(hm[2], total)), PT.g2_compress(None)])
    verdict, status = ctx.fast_aggregate_verify_batch(pk, msg, sig)
    assert verdict.tolist() == [1, 0, 0, 0]
    assert status[2, 9] == PT.BAD_ENCODING and status.sum() == PT.BAD_ENCODING
    # check 3 with a finite signature fails as well
    sig = sig.copy()
    sig[3] = sig[0]
    assert ctx.fast_aggregate_verify_batch(pk, msg, sig)[0].tolist() == [1, 0, 0, 0]


def test_concurrent_callers(ctx, CO):
    """Several host threads on one context (calls serialise on the context's lock) and a second context on the
    same device driven from another thread at the same time: every result equals the CPU oracle's."""
    import threading
    import plonky2_bls12_381_pairing_amd as B
    ps, qs = H.random_pairs(24, 9090)
    g1b, g2b = H.g1_array(ps), H.g2_array(qs)
    jobs = []
    for t in range(4):
        n = (300, 64, 1000, 17)[t]
        ip, iq = (np.arange(n) * (t + 1)) % 24, (np.arange(n) * 5 + t) % 24
        jobs.append((g1b[ip], g2b[iq]))
    other = B.Context(0)
    results, errors = [None] * 4, []

    def work(t):
        try:
            c = other if t == 3 else ctx
            for _ in range(3):
                results[t] = c.pairing_batch(*jobs[t])
        except Exception as e:  # surfaced below: an exception in a thread would otherwise pass silently
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    other.close()
    assert not errors, errors
    for t in range(4):
        probe = np.unique(np.concatenate([np.arange(0, len(jobs[t][0]), 7), [len(jobs[t][0]) - 1]]))
        assert np.array_equal(results[t][probe], CO.pairing_batch(jobs[t][0][probe], jobs[t][1][probe])), t


def test_multi_device_context(ctx, CO):
    """bls381_multi_ctx over a device list: the one-GPU box lists device 0 twice and three times, so the shards run
    on separate contexts and host threads as they would on separate GPUs.  Results must equal the one-context
    calls bit for bit: a 65,536 + 777 batch of independent pairings with infinity flags on both sides of a shard
    boundary, the 513-pair product (valid and forged), sizes smaller than the device count, and the witness
    trace."""
    import plonky2_bls12_381_pairing_amd as B
    n = 65536 + 777
    ps, qs = H.random_pairs(7, 2025)
    g1 = np.tile(H.g1_array(ps), (n // 7 + 1, 1))[:n]
    g2 = np.roll(np.tile(H.g2_array(qs), (n // 7 + 1, 1))[:n], 3, 0)
    flags = np.zeros(n, np.uint8)
    flags[[0, n // 2 - 1, n // 2, n - 1]] = [1, 2, 3, 1]
    want = ctx.pairing_batch(g1, g2, flags)
    with B.MultiContext([0, 0]) as m2, B.MultiContext([0, 0, 0]) as m3:
        assert m2.devices == 2 and m3.devices == 3
        assert np.array_equal(m2.pairing_batch(g1, g2, flags), want)
        k = 1000
        assert np.array_equal(m3.pairing_batch(g1[:k], g2[:k], flags[:k]), want[:k])
        assert np.array_equal(m3.miller_loop_batch(g1[:k], g2[:k], flags[:k]), ctx.miller_loop_batch(g1[:k], g2[:k], flags[:k]))
        assert np.array_equal(m2.final_exponentiation_batch(want[:50]), ctx.final_exponentiation_batch(want[:50]))
        # fewer items than devices, and none
        assert np.array_equal(m3.pairing_batch(g1[:2], g2[:2]), want[:2] if not flags[:2].any() else ctx.pairing_batch(g1[:2], g2[:2]))
        assert m3.pairing_batch(g1[:0], g2[:0]).shape == (0, 72)
        # the aggregate-signature product: 512 (pk_i, H_i) pairs and (-G1, sigma)
        import random
        rng = random.Random(77)
        sks = [rng.randrange(1, R.R_ORDER) for _ in range(8)]
        hs = [rng.randrange(1, R.R_ORDER) for _ in range(8)]
        pk = [R.g1_mul(R.G1_GEN, s) for s in sks]
        hm = [R.g2_mul(R.G2_GEN, h) for h in hs]
        idx = [(i % 8, (i // 8) % 8) for i in range(512)]
        total = sum(sks[a] * hs[b] for a, b in idx) % R.R_ORDER
        sigma = R.g2_mul(R.G2_GEN, total)
        a1 = H.g1_array([pk[a] for a, _ in idx] + [R.g1_neg(R.G1_GEN)])
        a2 = H.g2_array([hm[b] for _, b in idx] + [sigma])
        for mc in (m2, m3):
            out, one = mc.native_pairing(a1, a2)
            assert one and np.array_equal(out, ctx.native_pairing(a1, a2)[0])
            assert np.array_equal(mc.multi_miller_loop_native(a1, a2), ctx.multi_miller_loop_native(a1, a2))
        forged = a2.copy()
        forged[-1] = H.g2_array([R.g2_mul(R.G2_GEN, total + 1)])[0]
        out, one = m3.native_pairing(a1, forged)
        assert not one and np.array_equal(out, ctx.native_pairing(a1, forged)[0])
        out, one = m3.native_pairing(a1[:0], a2[:0])             # empty product on every shard
        assert one
        # witness traces, three proofs over two devices
        assert np.array_equal(m2.pairing_witness_batch(g1[:3], g2[:3]), ctx.pairing_witness_batch(g1[:3], g2[:3]))
    with pytest.raises(B.Bls381Error):
        B.MultiContext([0, 99])


def test_workspace_reservation(CO):
    """bls381_ctx_reserve grows the workspace ahead of time: the first large call after it allocates nothing (the
    device's free memory is the same before and after the call) and gives the usual values."""
    import torch
    import plonky2_bls12_381_pairing_amd as B
    from plonky2_bls12_381_pairing_amd import _lib
    ps, qs = H.random_pairs(6, 4321)
    n = 20000
    g1 = np.tile(H.g1_array(ps), (n // 6 + 1, 1))[:n]
    g2 = np.tile(H.g2_array(qs), (n // 6 + 1, 1))[:n]
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.empty((n, 72), dtype=torch.int64, device=dev)
    small = torch.empty((100, 72), dtype=torch.int64, device=dev)
    with B.Context(0) as c:
        c.reserve(n)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        _lib.check(_lib.lib.bls381_pairing_batch_dev(c.handle, d1.data_ptr(), d2.data_ptr(), None, n, out.data_ptr()))
        _lib.check(_lib.lib.bls381_pairing_batch_dev(c.handle, d1.data_ptr(), d2.data_ptr(), None, 100, small.data_ptr()))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free_before
        c.reserve(0)
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[:6], CO.pairing_batch(g1[:6], g2[:6]))
    assert np.array_equal(got[6:12], got[:6]) and np.array_equal(small.cpu().numpy().view(np.uint64), got[:100])


def test_plain_c_caller(tmp_path):
    """examples/pairing_c_abi.c, built with gcc alone, prints e(G1, G2): the same limbs as the oracle's pairing of the
    generators (whose first coefficient is the public constant of zkcrypto's Gt generator)."""
    import subprocess
    from test_host_cpu import _build_c_example
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    got = [int("".join(line.split()[1:]), 16) for line in out.stdout.strip().splitlines()]
    want = R.fq12_flatten_ark(R.native_pairing([R.G1_GEN], [R.G2_GEN]))
    assert got == [x * (1 << 384) % R.P for x in want]


def test_argument_errors_on_a_live_context(ctx):
    """NULL buffers, sizes past the documented limits and bad selectors: BLS381_ERR_INVALID, nothing launched,
    and the context keeps working afterwards."""
    import ctypes as C
    from plonky2_bls12_381_pairing_amd import _lib
    L, h = _lib.lib, ctx.handle
    buf = (C.c_uint64 * 72)()
    small = (C.c_uint8 * 96)()
    assert L.bls381_pairing_batch(h, None, buf, None, 1, buf) == -1
    assert L.bls381_pairing_batch(h, buf, buf, None, 1, None) == -1
    assert L.bls381_pairing_batch_dev(h, None, None, None, 5, None) == -1
    assert L.bls381_final_exp_batch(h, None, 3, buf, None) == -1
    assert L.bls381_frobenius_batch(h, None, 3, 1, buf) == -1
    assert L.bls381_pow_batch(h, buf, None, 1, 1, buf) == -1
    assert L.bls381_pow_batch(h, buf, buf, 7, 1, buf) == -1
    assert b"exponent" in L.bls381_last_error()
    assert L.bls381_pairing_fixed_q_batch(h, None, buf, None, 1, buf) == -1
    assert b"prepared" in L.bls381_last_error()
    one = C.c_int(0)
    assert L.bls381_multi_pairing(h, buf, buf, None, (1 << 20) + 1, buf, C.byref(one)) == -1
    assert L.bls381_finish_multi_pairing(h, buf, 0, buf, C.byref(one)) == -1
    assert L.bls381_g1_decompress_batch(h, small, (1 << 22) + 1, 1, buf, small) == -1
    assert b"2^22" in L.bls381_last_error()
    assert L.bls381_pairing_witness_batch(h, None, buf, 1, None) == -1
    ps, qs = H.random_pairs(2, 31337)
    out = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
    assert H.fq12_from_row(out[1]) == R.native_pairing([ps[1]], [qs[1]])


def test_field_elements_not_below_p_are_refused(ctx):
    """The Montgomery-form entry points take the limbs ark_ff holds, which are always below p.  A C caller can pass
    anything: a value with limbs >= p makes the whole call fail with BLS381_ERR_INVALID (the layout-conversion kernel
    notes it), whichever input array holds it, on host and device pointers, and the context keeps working."""
    import torch
    import plonky2_bls12_381_pairing_amd as B
    from plonky2_bls12_381_pairing_amd import _lib
    ps, qs = H.random_pairs(4, 515)
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    good = ctx.pairing_batch(g1, g2)
    p_limbs = np.array(R.fq_to_u64_limbs(R.P), dtype=np.uint64)
    for which, col in (("g1", 6), ("g2", 18), ("g1", 0)):
        b1, b2 = g1.copy(), g2.copy()
        (b1 if which == "g1" else b2)[2, col:col + 6] = p_limbs           # exactly p: the smallest value that is refused
        for fn in (ctx.pairing_batch, ctx.miller_loop_batch):
            with pytest.raises(B.Bls381Error) as ei:
                fn(b1, b2)
            assert ei.value.code == -1 and "not below p" in str(ei.value)
    b1 = g1.copy()
    b1[3, 0:6] = np.uint64(0xFFFFFFFFFFFFFFFF)                            # all ones
    d1 = torch.from_numpy(b1.view(np.int64)).cuda()
    d2 = torch.from_numpy(g2.view(np.int64)).cuda()
    out = torch.zeros((4, 72), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert _lib.lib.bls381_pairing_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), None, 4, out.data_ptr()) == -1
    assert b"not below p" in _lib.lib.bls381_last_error()
    x = H.fq12_array(H.random_fq12(2, 516))
    x[1, 66:72] = p_limbs
    for fn in (ctx.final_exponentiation_batch, ctx.conjugate_batch, lambda v: ctx.fq12_mul_batch(v, v)):
        with pytest.raises(B.Bls381Error):
            fn(x)
    # p - 1 is the largest value that passes, and the context is unharmed
    pm1 = np.array(R.fq_to_u64_limbs(R.P - 1), dtype=np.uint64)
    x[1, 66:72] = pm1
    assert ctx.conjugate_batch(x).shape == (2, 72)
    assert np.array_equal(ctx.pairing_batch(g1, g2), good)


def _raw392(vals):
    """Fq12 values -> (n, 144) u32: canonical words carrying the kernels' internal Montgomery factor 2^392."""
    out = np.zeros((len(vals), 144), dtype=np.uint32)
    for i, v in enumerate(vals):
        out[i] = [w for c in R.fq12_flatten_ark(v) for w in R.fq_to_u32_limbs(c * (1 << 392) % R.P)]
    return out


def test_squaring_chain_kernel(ctx):
    """The kernel that runs the hard part's 63-squaring chains in registers (three waves per 64 elements, generated
    assembly): the six powers a^(2^k), k = 16, 48, 57, 60, 62, 63, it keeps against the oracle's plain squarings, for
    batch sizes on both sides of a wave's 64 and a workgroup's 256 elements; their product is pow_native(a, BLS_X)."""
    import ctypes as C
    from plonky2_bls12_381_pairing_amd import _lib
    xs = [R.myfq12_to_fq12(R.easy_part_native(R.myfq12_from_fq12(x))) for x in H.random_fq12(3, 6301)]   # cyclotomic subgroup
    want = []
    for x in xs:
        row, cur = [], x
        for k in range(1, 64):
            cur = R.fq12_square(cur)
            if (R.BLS_X >> k) & 1:
                row.append(cur)
        assert len(row) == 6
        prod = row[0]
        for v in row[1:]:
            prod = R.fq12_mul(prod, v)
        assert prod == R.fq12_pow(x, R.BLS_X)
        want.append(_raw392(row).reshape(-1))
    want = np.array(want)
    inv392 = R.fq_inv((1 << 392) % R.P)
    for n in (1, 3, 64, 65, 200, 256, 257, 833):     # (a workgroup is four groups of 64 elements)
        a = np.ascontiguousarray(_raw392([xs[i % 3] for i in range(n)]))
        out = np.zeros((n, 6 * 144), dtype=np.uint32)
        _lib.check(_lib.lib.bls381_debug_cyc_chain(ctx.handle, a.ctypes.data_as(_lib.u32p), n, out.ctypes.data_as(_lib.u32p)))
        for i in range(n):
            assert np.array_equal(out[i], want[i % 3]), (n, i)
    assert inv392  # (the raw form decodes with this factor; kept here for readers of a failing comparison)
