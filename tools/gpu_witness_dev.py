#!/usr/bin/env python3
"""Timing of the device-pointer witness-trace call (the trace stays in HBM)."""
import os, sys, time
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd._lib import lib, check
ps, qs = H.random_pairs(8, 31)
ctx = B.Context(0)
rows = B.witness_rows()
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", "4096"))
g1 = torch.from_numpy(np.tile(H.g1_array(ps), (n // 8, 1)).view(np.int64)).to(dev)
g2 = torch.from_numpy(np.tile(H.g2_array(qs), (n // 8, 1)).view(np.int64)).to(dev)
trace = torch.empty((n, rows, 12), dtype=torch.int32, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.bls381_pairing_witness_batch_dev(ctx.handle, g1.data_ptr(), g2.data_ptr(), n, trace.data_ptr()))
    dt = time.perf_counter() - t0
print("cache %s block %s chunk %s: %d proofs %.1f ms = %.4f ms per proof, %.0f proofs/s, %.0f GB/s written" %
      (os.environ.get("BLS381_WIT_CACHE", "default"), os.environ.get("BLS381_WIT_BLOCK", "default"), os.environ.get("BLS381_WIT_CH", "default"), n, dt * 1e3, dt * 1e3 / n, n / dt,
       n * rows * 48 / dt / 1e9), flush=True)
print("checksum", int(trace[:8].to(torch.int64).sum().item()), int(trace[-8:].to(torch.int64).sum().item()))
