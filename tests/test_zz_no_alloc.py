"""The entry point that allocates nothing (bls381_pairing_batch_dev_noalloc) against the entry point it is a
variant of (bls381_pairing_batch_dev), through the C ABI: bit-exact, two calls in a row, the second smaller."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_matches_pairing_batch_dev():
    import torch
    torch.cuda.init()
    import plonky2_bls12_381_pairing_amd as B
    from plonky2_bls12_381_pairing_amd import _lib
    L = _lib.lib
    ps, qs = H.random_pairs(16, seed=4242)
    with B.Context(0) as ctx:
        for n, with_flags in ((300, True), (37, False)):       # the second call is smaller than the first
            p = (ps * (n // 16 + 1))[:n]
            q = (qs * (n // 16 + 1))[1:n + 1]
            g1, g2 = H.g1_array(p), H.g2_array(q)
            fl = np.zeros(n, np.uint8)
            if with_flags:
                fl[5] = 1
                fl[n - 1] = 2
            d1 = torch.from_numpy(g1.view(np.int64)).cuda()
            d2 = torch.from_numpy(g2.view(np.int64)).cuda()
            df = torch.from_numpy(fl).cuda()
            want = torch.zeros((n, 72), dtype=torch.int64, device="cuda")
            got = torch.full((n, 72), -1, dtype=torch.int64, device="cuda")
            fptr = df.data_ptr() if with_flags else None
            torch.cuda.synchronize()
            _lib.check(L.bls381_pairing_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), fptr, n, want.data_ptr()))
            _lib.check(L.bls381_pairing_batch_dev_noalloc(ctx.handle, d1.data_ptr(), d2.data_ptr(), fptr, n, got.data_ptr()))
            assert torch.equal(got, want)
            if with_flags:                                      # flagged pairs give one
                one = H.fq12_array([H.R.FQ12_ONE])[0]
                assert np.array_equal(got[5].cpu().numpy().view(np.uint64), one)
        # a size beyond the fixed workspace is refused before anything runs
        n = 4097
        big1 = torch.zeros((n, 12), dtype=torch.int64, device="cuda")
        big2 = torch.zeros((n, 24), dtype=torch.int64, device="cuda")
        bout = torch.zeros((n, 72), dtype=torch.int64, device="cuda")
        assert L.bls381_pairing_batch_dev_noalloc(ctx.handle, big1.data_ptr(), big2.data_ptr(), None, n, bout.data_ptr()) == -1
        assert b"at most 4096" in L.bls381_last_error()
        assert L.bls381_pairing_batch_dev_noalloc(ctx.handle, None, None, None, 0, None) == 0
    # On a context that has run nothing yet (no workspace of its own to lean on; the kernels' code is on the device
    # already, loaded by the calls above), with the driver's free-memory figure unchanged across the calls: the entry
    # point exists for callers that may not allocate.
    with B.Context(0) as fresh:
        n = 64
        g1, g2 = H.g1_array((ps * 4)[:n]), H.g2_array((qs * 4)[:n])
        d1 = torch.from_numpy(g1.view(np.int64)).cuda()
        d2 = torch.from_numpy(g2.view(np.int64)).cuda()
        got = torch.full((n, 72), -1, dtype=torch.int64, device="cuda")
        want = torch.zeros((n, 72), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(2):
            _lib.check(L.bls381_pairing_batch_dev_noalloc(fresh.handle, d1.data_ptr(), d2.data_ptr(), None, n, got.data_ptr()))
        assert torch.cuda.mem_get_info()[0] == free_before
        _lib.check(L.bls381_pairing_batch_dev(fresh.handle, d1.data_ptr(), d2.data_ptr(), None, n, want.data_ptr()))
        assert torch.equal(got, want)
        assert np.array_equal(got[:16].cpu().numpy().view(np.uint64), H.fq12_array([H.R.native_pairing([p], [q]) for p, q in zip(ps, qs)]))
