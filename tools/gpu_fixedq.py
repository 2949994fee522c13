#!/usr/bin/env python3
"""Timing of the fixed-Q mode (one prepared G2 point, 65,536 G1 points) beside the general path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
sys.argv = [sys.argv[0]]
import bench
g1, g2, _, g2b = bench.synth_inputs(np, H, bench.BATCH)
ctx = B.Context(0)
prep = ctx.g2_prepared(g2b[0])
for name, fn in (("fixed-Q pairing", lambda: ctx.pairing_fixed_q_batch(prep, g1)), ("general pairing", lambda: ctx.pairing_batch(g1, g2))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t0) / 3
    print("%s: %.2f ms per 65,536 (host pointers); device kernels: %s ms" % (name, dt * 1e3, ["%.2f" % x for x in ctx.last_timing()]))
h1, h2 = np.roll(g1, 1, 0), np.roll(g2, 3, 0)
fn = lambda: ctx.pairing_two_pairs_batch(g1, g2, h1, h2)
fn()
t0 = time.perf_counter()
for _ in range(3):
    fn()
print("two pairs per item: %.2f ms per 65,536 items (host pointers)" % ((time.perf_counter() - t0) / 3 * 1e3))
