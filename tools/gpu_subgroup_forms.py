"""Cross-check of the two forms of the subgroup tests on the GPU: 20,000 decoded points, half of them on the curve but
outside the subgroup, through the step programs (four lanes per point at this size) and through the one-lane tests
inside the decoders: same statuses, same points."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import bls381_ref as R
import bls381_points as PT
import plonky2_bls12_381_pairing_amd as B
ctx = B.Context(0)
rng = random.Random(99)
# G2: 24 points in the subgroup, 24 curve points outside it
ins = [PT.g2_compress(R.g2_mul(R.G2_GEN, rng.randrange(1, R.R_ORDER))) for _ in range(24)]
outs = []
x0 = 1
while len(outs) < 24:
    xx = (x0, rng.randrange(1, 1000))
    rhs = R.fq2_add(R.fq2_mul(R.fq2_square(xx), xx), (4, 4))
    if R.fq2_pow(rhs, (R.P * R.P - 1) // 2) == R.FQ2_ONE:
        e = bytearray(xx[1].to_bytes(48, "big") + xx[0].to_bytes(48, "big")); e[0] |= 0x80 | (0x20 if len(outs) % 2 else 0)
        outs.append(bytes(e))
    x0 += 1
n = 20000
pick = [rng.randrange(48) for _ in range(n)]
enc = [ins[k] if k < 24 else outs[k - 24] for k in pick]
arr = np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(n, 96)
res = {}
for form in ("1", "0"):
    os.environ["BLS381_G2_SUBGROUP_PROGRAM"] = form
    out, st = ctx.g2_decompress_batch(arr, check_subgroup=True, return_status=True)
    res[form] = (out.copy(), st.copy())
assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])
assert res["1"][1].tolist() == [0 if k < 24 else 4 for k in pick]
print("G2: program and one-lane forms agree on", n, "points,", sum(1 for k in pick if k >= 24), "outside the subgroup")
# G1
ins1 = [PT.g1_compress(R.g1_mul(R.G1_GEN, rng.randrange(1, R.R_ORDER))) for _ in range(24)]
outs1 = []
x = 4
while len(outs1) < 24:
    if R.fq_sqrt((x ** 3 + 4) % R.P) is not None:
        e = bytearray(x.to_bytes(48, "big")); e[0] |= 0x80 | (0x20 if len(outs1) % 2 else 0)
        if PT.g1_decompress(bytes(e))[0] == PT.NOT_IN_SUBGROUP: outs1.append(bytes(e))
    x += 1
enc = [ins1[k] if k < 24 else outs1[k - 24] for k in pick]
arr = np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(n, 48)
res = {}
for form in ("1", "0"):
    os.environ["BLS381_G1_SUBGROUP_PROGRAM"] = form
    out, st = ctx.g1_decompress_batch(arr, check_subgroup=True, return_status=True)
    res[form] = (out.copy(), st.copy())
assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])
assert res["1"][1].tolist() == [0 if k < 24 else 4 for k in pick]
print("G1: program and one-lane forms agree on", n, "points")
