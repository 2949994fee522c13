#!/usr/bin/env python3
"""Timing of the squaring-chain kernel alone: 65,536 values, 63 cyclotomic squarings each (bls381_debug_cyc_chain)."""
import os, sys
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd import _lib
n = int(os.environ.get("N", "65536"))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 28, size=(n, 144), dtype=np.uint32)   # any words below p's top: timing only
out = np.zeros((n, 6 * 144), dtype=np.uint32)
ctx = B.Context(0)
for rep in range(4):
    _lib.check(_lib.lib.bls381_debug_cyc_chain(ctx.handle, a.ctypes.data_as(_lib.u32p), n, out.ctypes.data_as(_lib.u32p)))
    ms = ctx.last_timing()[2]
    # 63 squarings x 18 Fp multiplications x 300 multiply-accumulates in the reference's work units
    print("chain kernel: %d values %.3f ms  (%.2f TMAC/s in reference work units, %.3f of 39.3)" %
          (n, ms, n * 63 * 18 * 300 / ms / 1e9, n * 63 * 18 * 300 / ms / 1e9 / 39.3216), flush=True)
