"""Decode time against batch size: shows how far the one-lane-per-point decoders are from filling the chip."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import helpers as H  # noqa: F401
import bls381_ref as R
import bls381_points as PT
import plonky2_bls12_381_pairing_amd as B

ctx = B.Context(0)
e1 = np.frombuffer(b"".join(PT.g1_compress(R.g1_mul(R.G1_GEN, k)) for k in range(1, 65)), dtype=np.uint8).reshape(64, 48)
e2 = np.frombuffer(b"".join(PT.g2_compress(R.g2_mul(R.G2_GEN, k)) for k in range(1, 65)), dtype=np.uint8).reshape(64, 96)
for n in (16384, 32768, 65536, 131072, 262144):
    d1, d2 = np.tile(e1, (n // 64, 1)), np.tile(e2, (n // 64, 1))
    for label, fn in (("G1 curve", lambda: ctx.g1_decompress_batch(d1, check_subgroup=False)),
                      ("G1 curve+subgroup", lambda: ctx.g1_decompress_batch(d1, check_subgroup=True)),
                      ("G2 curve", lambda: ctx.g2_decompress_batch(d2, check_subgroup=False)),
                      ("G2 curve+subgroup", lambda: ctx.g2_decompress_batch(d2, check_subgroup=True))):
        fn()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        print("%-20s n=%7d %8.2f ms %6.1f ns/point" % (label, n, best * 1e3, best * 1e9 / n), flush=True)
