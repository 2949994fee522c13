#!/usr/bin/env python3
"""Generates tests/golden/point_vectors.json from oracle/py/bls381_points.py (seeded scalars).

Carries the two public constants the oracle is pinned by (the compressed encodings of the standard G1 and G2
generators) and, for seeded multiples of them, the encodings, the decoded coordinates and their sums.
"""
import json
import os
import random
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import bls381_ref as R
import bls381_points as PT

def hx(v): return "%096x" % v

rng = random.Random(0x5EEDB15381 + 4)
ks = [1, 2, R.R_ORDER - 1] + [rng.randrange(1, R.R_ORDER) for _ in range(5)]
cases = []
for k in ks:
    p, q = R.g1_mul(R.G1_GEN, k), R.g2_mul(R.G2_GEN, k)
    cases.append(dict(k=hex(k), g1_bytes=PT.g1_compress(p).hex(), g2_bytes=PT.g2_compress(q).hex(),
                      g1=[hx(p[0]), hx(p[1])], g2=[hx(q[0][0]), hx(q[0][1]), hx(q[1][0]), hx(q[1][1])]))
total = sum(ks) % R.R_ORDER
s1, s2 = R.g1_mul(R.G1_GEN, total), R.g2_mul(R.G2_GEN, total)
# curve points outside the order-r subgroup (smallest x >= 4 with x^3 + 4 a square)
outside = []
x = 4
while len(outside) < 2:
    if R.fq_sqrt((x ** 3 + 4) % R.P) is not None:
        b = bytearray(x.to_bytes(48, "big")); b[0] |= 0x80
        assert PT.g1_decompress(bytes(b))[0] == PT.NOT_IN_SUBGROUP
        outside.append(bytes(b).hex())
    x += 1
out = dict(generator_g1=PT.G1_GEN_COMPRESSED.hex(), generator_g2=PT.G2_GEN_COMPRESSED.hex(),
           infinity_g1=PT.g1_compress(None).hex(), infinity_g2=PT.g2_compress(None).hex(),
           cases=cases, sum_g1=PT.g1_compress(s1).hex(), sum_g2=PT.g2_compress(s2).hex(),
           on_curve_outside_subgroup_g1=outside)
path = os.path.join(ROOT, "tests", "golden", "point_vectors.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
