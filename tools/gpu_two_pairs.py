"""Throughput of the two-pairs-per-item call (device-resident inputs) beside two single-pair passes."""
import os
import sys
import time

import torch
torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd._lib import lib, check

geo = [int(x) for x in os.environ.get('GEO', '0:0').split(':')]
ctx = B.Context(0, geo[0], geo[1])
n = 65536
ps, qs = H.random_pairs(64, 9)
g1 = np.tile(H.g1_array(ps), (n // 64, 1)); g2 = np.tile(H.g2_array(qs), (n // 64, 1))
dev = torch.device("cuda", 0)
d = [torch.from_numpy(a.view(np.int64)).to(dev) for a in (g1, g2, np.roll(g1, 1, 0), np.roll(g2, 1, 0))]
out = torch.empty((n, 72), dtype=torch.int64, device=dev)
one = torch.empty(n, dtype=torch.uint8, device=dev)
ms = (B._lib.C.c_float * 4)()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.bls381_pairing_two_pairs_batch_dev(ctx.handle, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                 None, n, out.data_ptr(), one.data_ptr()))
    dt = time.perf_counter() - t0
    lib.bls381_last_timing(ctx.handle, ms)
    print("two pairs per item: %.2f ms per %d items (%.0f items/s); miller2 %.2f ms, final exp %.2f ms" %
          (dt * 1e3, n, n / dt, ms[0], ms[1] + ms[2]), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.bls381_miller_loop_batch_dev(ctx.handle, d[0].data_ptr(), d[1].data_ptr(), None, n, out.data_ptr()))
    dt = time.perf_counter() - t0
    print("single-pair Miller loop: %.2f ms per %d" % (dt * 1e3, n), flush=True)
