#!/usr/bin/env python3
"""Device-pointer pairing call by batch size (tuning aid): ms per call and pairings/s, with the kernels' shares."""
import os, sys, time
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd._lib import lib, check
ps, qs = H.random_pairs(8, 77)
ctx = B.Context(0)
dev = torch.device("cuda", 0)
sizes = [int(x) for x in os.environ.get("SIZES", "256,1024,4096,8192,8256,12288,16384,24576,32768,40960,49152,65536,98304,131072").split(",")]
nmax = max(sizes)
g1 = torch.from_numpy(np.tile(H.g1_array(ps), (nmax // 8 + 1, 1))[:nmax].view(np.int64)).to(dev)
g2 = torch.from_numpy(np.tile(H.g2_array(qs), (nmax // 8 + 1, 1))[:nmax].view(np.int64)).to(dev)
out = torch.empty((nmax, 72), dtype=torch.int64, device=dev)
for n in sizes:
    best = None
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        check(lib.bls381_pairing_batch_dev(ctx.handle, g1.data_ptr(), g2.data_ptr(), None, n, out.data_ptr()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = dt if best is None else min(best, dt)
    ms = ctx.last_timing()
    print("n %7d: %7.3f ms  %9.0f pairings/s   last chunk: miller %.2f  fe1+inv %.2f  part 2 %.2f ms" % (n, best * 1e3, n / best, ms[0], ms[1], ms[2]), flush=True)
