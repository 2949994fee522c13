"""The workload behind profiles/r1_verify_kernel_stats.csv: 65,536 signature checks from compressed bytes with every
subgroup test on, three calls, then one 512-signer aggregate check.  Run under rocprofv3 --kernel-trace --stats."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import helpers as H  # noqa: F401  (puts oracle/py on the path)
import bls381_ref as R
import bls381_points as PT
import plonky2_bls12_381_pairing_amd as B

ctx = B.Context(0)
rng = random.Random(7)
pk, mg, sg = [], [], []
for _ in range(64):
    sk, h = rng.randrange(1, R.R_ORDER), rng.randrange(1, R.R_ORDER)
    hm = R.g2_mul(R.G2_GEN, h)
    pk.append(PT.g1_compress(R.g1_mul(R.G1_GEN, sk)))
    mg.append(PT.g2_compress(hm))
    sg.append(PT.g2_compress(R.g2_mul(hm, sk)))
tile = lambda enc, w: np.tile(np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(64, w), (65536 // 64, 1))
bpk, bmg, bsg = tile(pk, 48), tile(mg, 96), tile(sg, 96)
for _ in range(3):
    v, _ = ctx.verify_batch(bpk, bmg, bsg, check_subgroup=7)
    assert v.all()
one_sig = PT.g2_decompress(bytes(bsg[0]))[1]
ok, _ = ctx.fast_aggregate_verify(bpk[:1].repeat(512, 0), bytes(bmg[0]), PT.g2_compress(R.g2_mul(one_sig, 512)),
                                  check_subgroup=7)
assert ok
print("verify profile workload done", flush=True)
