#!/usr/bin/env python3
"""PCIe-inclusive rate: the host-pointer entry point on one 65,536-pair batch (tuning aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
sys.argv = [sys.argv[0]]
import bench
g1, g2, _, _ = bench.synth_inputs(np, H, bench.BATCH)
ctx = B.Context(0)
ctx.pairing_batch(g1, g2)
t0 = time.perf_counter()
for _ in range(5):
    out = ctx.pairing_batch(g1, g2)
dt = (time.perf_counter() - t0) / 5
print("host-pointer pairing_batch: %.2f ms per 65,536 pairs = %.0f pairings/s" % (dt * 1e3, bench.BATCH / dt))
g1b, g2b = np.tile(g1, (4, 1)), np.tile(g2, (4, 1))
ctx.pairing_batch(g1b, g2b)
t0 = time.perf_counter()
for _ in range(3):
    out = ctx.pairing_batch(g1b, g2b)
dt = (time.perf_counter() - t0) / 3
print("host-pointer pairing_batch: %.2f ms per 262,144 pairs = %.0f pairings/s" % (dt * 1e3, 4 * bench.BATCH / dt))
