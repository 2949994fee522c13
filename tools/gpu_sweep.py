"""Throughput sweep over lanes-per-pairing / LDS slots (development tool, needs a GPU)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import *
import plonky2_bls12_381_pairing_amd as B
ps, qs = random_pairs(6, 3)
n = int(os.environ.get("SWEEP_N", "65536"))
g1 = np.tile(g1_array(ps), (n // 6 + 1, 1))[:n]; g2 = np.tile(g2_array(qs), (n // 6 + 1, 1))[:n]
want = [R.native_pairing([ps[i]], [qs[i]]) for i in range(6)]
for spec in sys.argv[1:]:
    K, slots = [int(x) for x in spec.split(':')]
    try:
        ctx = B.Context(0, K, slots)
    except Exception as e:
        print(spec, "ctx failed:", e, flush=True); continue
    best = None
    for rep in range(3):
        out = ctx.pairing_batch(g1, g2); ms = ctx.last_timing()
        if best is None or ms[3] < best[3]: best = ms
    ok = all(fq12_from_row(out[i]) == want[i] for i in range(6)) and fq12_from_row(out[n - 1]) == want[(n - 1) % 6]
    print("K=%2d slots=%3d ok=%s miller %.2f fe1+inv %.2f fe2 %.2f total %.2f ms -> %.0f pairings/s" % (K, slots, ok, best[0], best[1], best[2], best[3], n / (best[3] * 1e-3)), flush=True)
    ctx.close()
