#!/usr/bin/env python3
"""Timing of the witness-trace path (141,376 rows of 12 x u32 per proof) on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
ps, qs = H.random_pairs(8, 31)
ctx = B.Context(0)
for n in (8, 64, 256, 1024):
    g1 = np.tile(H.g1_array(ps), (n // 8, 1)); g2 = np.tile(H.g2_array(qs), (n // 8, 1))
    ctx.pairing_witness_batch(g1, g2)
    t0 = time.perf_counter()
    tr = ctx.pairing_witness_batch(g1, g2)
    dt = time.perf_counter() - t0
    print("%d proofs: %.1f ms (%.2f ms per proof, %.1f MB of trace, host copy included)" % (n, dt * 1e3, dt * 1e3 / n, tr.nbytes / 1e6))
