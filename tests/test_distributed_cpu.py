"""The N > 1 path on CPU: two gloo ranks shard a multi-pairing, all-gather the Fq12 partials and
each multiplies them out; the oracle stands in for the per-rank GPU work."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers as H
from plonky2_bls12_381_pairing_amd.parallel import all_gather_partials, shard_range

R = H.R


def test_shard_range_partitions():
    for n in (0, 1, 7, 512, 513, 65536):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps, qs = H.random_pairs(5, 321)
    lo, hi = shard_range(len(ps), rank, world)
    # per-rank partial: the Miller-loop product of the shard (conjugated, as the library returns it)
    part = R.multi_miller_loop(ps[lo:hi], qs[lo:hi])
    gathered = all_gather_partials(H.fq12_array([part])[0])
    assert gathered.shape == (world, 72)
    prod = R.FQ12_ONE
    for row in gathered:
        prod = R.fq12_mul(prod, H.fq12_from_row(row))
    ok = (prod == R.multi_miller_loop(ps, qs)) and (R.final_exponentiation(prod) == R.native_pairing(ps, qs))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_multi_pairing_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
