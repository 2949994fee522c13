"""The multi-pairing product over several ranks, end to end on real kernels: 512 (key, message point) pairs plus
(-G1, signature) are cut into contiguous shards, every rank reduces its shard to one Fq12 on the GPU, the partial
products are all-gathered and every rank finishes (product, final exponentiation, compare with one).

Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 tools/gpu_sharded_multi_pairing.py
On a one-GPU box set BLS381_BENCH_BACKEND=gloo BLS381_BENCH_DEVICE=0 (all ranks share the card); with one rank per
GPU leave both unset (RCCL)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
import helpers as H
import bls381_ref as R
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd import parallel

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("BLS381_BENCH_BACKEND", "nccl")
device_index = int(os.environ.get("BLS381_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(device_index)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
ctx = B.Context(device_index)

rng = random.Random(512)            # the same inputs on every rank
base, n = 8, 512
sks = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
hs = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
pks = [R.g1_mul(R.G1_GEN, s) for s in sks]
Hs = [R.g2_mul(R.G2_GEN, h) for h in hs]
ps = [pks[i % base] for i in range(n)] + [R.g1_neg(R.G1_GEN)]
scalar = sum(sks[i % base] * hs[i % base] for i in range(n)) % R.R_ORDER
for label, sig_scalar, expect in (("valid", scalar, True), ("forged", (scalar + 1) % R.R_ORDER, False)):
    qs = [Hs[i % base] for i in range(n)] + [R.g2_mul(R.G2_GEN, sig_scalar)]
    g1, g2 = H.g1_array(ps), H.g2_array(qs)
    lo, hi = parallel.shard_range(n + 1, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    partial = ctx.multi_miller_loop_native(g1[lo:hi], g2[lo:hi])
    parts = parallel.all_gather_partials(partial, torch.device("cuda", device_index) if backend == "nccl" else None)
    e, is_one = ctx.finish_multi_pairing(parts)
    dt = time.perf_counter() - t0
    assert bool(is_one) == expect, (label, rank)
    whole, whole_one = ctx.native_pairing(g1, g2)           # the same check on one rank
    assert np.array_equal(e, whole) and bool(whole_one) == expect, (label, rank)
    print("rank %d/%d %s: shard [%d, %d), is_one=%s, equal to the one-rank result, %.2f ms" % (rank, world, label, lo, hi, bool(is_one), dt * 1e3), flush=True)
dist.barrier()
dist.destroy_process_group()
ctx.close()
