#!/usr/bin/env python3
"""Latency of the multi-pairing product form (config: 513 pairs -> one Fq12, final exponentiation, is-one test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
ps, qs = H.random_pairs(16, 5)
geo = [int(x) for x in os.environ.get('GEO', '0:0').split(':')]
ctx = B.Context(0, geo[0], geo[1])
for n in (1, 16, 513, 4096, 16384):
    reps = (n + 15) // 16
    g1 = np.tile(H.g1_array(ps), (reps, 1))[:n]; g2 = np.tile(H.g2_array(qs), (reps, 1))[:n]
    ctx.native_pairing(g1, g2)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.native_pairing(g1, g2)
    dt = (time.perf_counter() - t0) / 3
    print("multi-pairing of %6d pairs: %.2f ms" % (n, dt * 1e3))
