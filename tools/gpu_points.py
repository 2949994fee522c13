"""Timing of the wire-format and aggregation calls (host buffers in and out, copies included)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import helpers as H
import bls381_ref as R
import bls381_points as PT
import plonky2_bls12_381_pairing_amd as B

ctx = B.Context(0)
base1 = [R.g1_mul(R.G1_GEN, k) for k in range(1, 65)]
base2 = [R.g2_mul(R.G2_GEN, k) for k in range(1, 65)]
e1 = np.frombuffer(b"".join(PT.g1_compress(p) for p in base1), dtype=np.uint8).reshape(64, 48)
e2 = np.frombuffer(b"".join(PT.g2_compress(q) for q in base2), dtype=np.uint8).reshape(64, 96)


def timed(label, fn, n, reps=3):
    fn()
    best = min((lambda t0: (fn(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(reps))
    print("%-46s n=%7d  %8.2f ms  %10.0f /s" % (label, n, best * 1e3, n / best), flush=True)


for n in (65536,):
    d1 = np.tile(e1, (n // 64, 1))
    d2 = np.tile(e2, (n // 64, 1))
    timed("G1 decode, on-curve only", lambda: ctx.g1_decompress_batch(d1, check_subgroup=False), n)
    timed("G1 decode + subgroup test", lambda: ctx.g1_decompress_batch(d1, check_subgroup=True), n)
    timed("G2 decode, on-curve only", lambda: ctx.g2_decompress_batch(d2, check_subgroup=False), n)
    timed("G2 decode + subgroup test", lambda: ctx.g2_decompress_batch(d2, check_subgroup=True), n)
    p1, _ = ctx.g1_decompress_batch(d1, check_subgroup=False)
    p2, _ = ctx.g2_decompress_batch(d2, check_subgroup=False)
    timed("G1 encode", lambda: ctx.g1_compress_batch(p1), n)
    timed("G2 encode", lambda: ctx.g2_compress_batch(p2), n)
    timed("G1 aggregate, 128 groups of 512 keys", lambda: ctx.g1_aggregate_batch(p1, 512), n)
    timed("G1 aggregate, 1 group of 512 keys", lambda: ctx.g1_aggregate_batch(p1[:512], 512), 512)
    timed("G2 aggregate, 128 groups of 512 signatures", lambda: ctx.g2_aggregate_batch(p2, 512), n)

# signature checks from bytes: 64 distinct (key, message point, signature) triples tiled
import random
rng = random.Random(7)
pk, mg, sg = [], [], []
for _ in range(64):
    sk, h = rng.randrange(1, R.R_ORDER), rng.randrange(1, R.R_ORDER)
    hm = R.g2_mul(R.G2_GEN, h)
    pk.append(PT.g1_compress(R.g1_mul(R.G1_GEN, sk))); mg.append(PT.g2_compress(hm)); sg.append(PT.g2_compress(R.g2_mul(hm, sk)))
tile = lambda enc, w: np.tile(np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(64, w), (65536 // 64, 1))
bpk, bmg, bsg = tile(pk, 48), tile(mg, 96), tile(sg, 96)
for label, chk in (("all three inputs", 7), ("keys and signatures", 5), ("none", 0)):
    v, _ = ctx.verify_batch(bpk, bmg, bsg, check_subgroup=chk)
    assert v.all()
    timed("signature checks from bytes, subgroup tests: %s" % label, lambda: ctx.verify_batch(bpk, bmg, bsg, check_subgroup=chk), 65536)

# one signature by 512 signers (keys from a chain, as in the test)
keys, acc = [], R.g1_mul(R.G1_GEN, 12345)
stepp = R.g1_mul(R.G1_GEN, 999331)
tot = 0
for i in range(512):
    keys.append(PT.g1_compress(acc)); tot += 12345 + 999331 * i
    acc = R.g1_add(acc, stepp)
hm = R.g2_mul(R.G2_GEN, 424242)
pk512 = np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(512, 48)
msgb, sigb = PT.g2_compress(hm), PT.g2_compress(R.g2_mul(hm, tot % R.R_ORDER))
for label, chk in (("all inputs", 7), ("signature only", 4), ("none", 0)):
    assert ctx.fast_aggregate_verify(pk512, msgb, sigb, check_subgroup=chk)[0]
    timed("512-signer aggregate check from bytes, subgroup tests: %s" % label, lambda: ctx.fast_aggregate_verify(pk512, msgb, sigb, check_subgroup=chk), 512)
pk_b = np.tile(pk512[None], (128, 1, 1)); msg_b = np.tile(np.frombuffer(msgb, dtype=np.uint8)[None], (128, 1))
sig_b = np.tile(np.frombuffer(sigb, dtype=np.uint8)[None], (128, 1))
for label, chk in (("all inputs", 7), ("none", 0)):
    assert ctx.fast_aggregate_verify_batch(pk_b, msg_b, sig_b, check_subgroup=chk)[0].all()
    timed("128 aggregate checks of 512 signers in one call, subgroup tests: %s" % label,
          lambda: ctx.fast_aggregate_verify_batch(pk_b, msg_b, sig_b, check_subgroup=chk), 128)
