"""Host-side mirror of the reference's `native` module on top of the C ABI.

Names follow /root/reference/src/native/{pairing,miller_loop,final_exponentiation,fin_exp_helpers}.rs.
Arrays are numpy uint64 in the layouts of include/bls381_hip.h (Fq = 6 x u64 Montgomery limbs):
  g1: (n, 12)   g2: (n, 24)   fq12: (n, 72)   inf_flags: (n,) uint8 or None
Every function raises Bls381Error where the reference panics or the GPU call fails.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Bls381Error, check, lib


def _p64(a):
    return a.ctypes.data_as(_lib.u64p)


def _p32(a):
    return a.ctypes.data_as(_lib.u32p)


def _p8(a):
    return None if a is None else a.ctypes.data_as(_lib.u8p)


def _arr64(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError("expected shape (n, %d), got %r" % (cols, a.shape))
    return a


def _flags(f, n):
    if f is None:
        return None
    f = np.ascontiguousarray(f, dtype=np.uint8)
    if f.shape != (n,):
        raise ValueError("inf_flags must have shape (n,)")
    return f


class G2Prepared:
    """Device-resident G2Prepared (68 line coefficients of one G2 point, miller_loop.rs:13-16)."""

    def __init__(self, ctx, g2, infinity=False):
        self._p = C.c_void_p()
        g2 = _arr64(np.asarray(g2 if g2 is not None else np.zeros(24), dtype=np.uint64).reshape(1, 24), 24)
        check(lib.bls381_g2_prepared_create(ctx._h, _p64(g2), int(bool(infinity)), C.byref(self._p)))

    @property
    def handle(self):
        return self._p

    def close(self):
        if self._p:
            lib.bls381_g2_prepared_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


CHECK_KEYS, CHECK_MESSAGES, CHECK_SIGNATURES = 1, 2, 4


class Context:
    """One GPU.  lanes_per_group / lds_slots = 0 select the defaults."""

    def __init__(self, device_id=0, lanes_per_group=0, lds_slots=0):
        self._h = C.c_void_p()
        check(lib.bls381_ctx_create_ex(C.byref(self._h), device_id, lanes_per_group, lds_slots))

    def close(self):
        if self._h:
            lib.bls381_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def reserve(self, max_items):
        """Grows the workspace now for calls of up to max_items items (bls381_ctx_reserve)."""
        check(lib.bls381_ctx_reserve(self._h, int(max_items)))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- src/native/miller_loop.rs
    def miller_loop_batch(self, g1, g2, inf_flags=None):
        """multi_miller_loop_native([p], [q]) per pair (miller_loop.rs:121-162)."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("zip_eq: length mismatch")  # the reference panics in zip_eq (miller_loop.rs:129)
        fl = _flags(inf_flags, n)
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_miller_loop_batch(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out)))
        return out

    # ---- G2Prepared reused for many P (miller_loop.rs:13-16, 139-156)
    def g2_prepared(self, g2, infinity=False):
        """G2Prepared::from(q) kept on the device: returns a G2Prepared handle object."""
        return G2Prepared(self, g2, infinity)

    def miller_loop_fixed_q_batch(self, prepared, g1, p_is_infinity=None):
        """multi_miller_loop_native([p_i], [q]) for every p_i against one prepared q."""
        g1 = _arr64(g1, 12)
        n = g1.shape[0]
        fl = _flags(p_is_infinity, n)
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_miller_loop_fixed_q_batch(self._h, prepared.handle, _p64(g1), _p8(fl), n, _p64(out)))
        return out

    def pairing_fixed_q_batch(self, prepared, g1, p_is_infinity=None):
        """native_pairing([p_i], [q]) for every p_i against one prepared q."""
        g1 = _arr64(g1, 12)
        n = g1.shape[0]
        fl = _flags(p_is_infinity, n)
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_pairing_fixed_q_batch(self._h, prepared.handle, _p64(g1), _p8(fl), n, _p64(out)))
        return out

    def multi_miller_loop_native(self, g1, g2, inf_flags=None):
        """multi_miller_loop_native over all pairs: one Fq12 (miller_loop.rs:121-162)."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("zip_eq: length mismatch")
        fl = _flags(inf_flags, n)
        out = np.empty(72, dtype=np.uint64)
        check(lib.bls381_multi_miller_loop(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out)))
        return out

    def g2_prepare_batch(self, g2):
        """G2Prepared::from(q).ell_coeffs (miller_loop.rs:90-119): (n, 68, 3, 2, 6) u64."""
        g2 = _arr64(g2, 24)
        n = g2.shape[0]
        out = np.empty((n, 68 * 36), dtype=np.uint64)
        check(lib.bls381_g2_prepare_batch(self._h, _p64(g2), n, _p64(out)))
        return out.reshape(n, 68, 3, 2, 6)

    def ell_batch(self, f, coeffs, g1):
        """ell(f, coeffs, p) (miller_loop.rs:185-194)."""
        f, coeffs, g1 = _arr64(f, 72), _arr64(coeffs, 36), _arr64(g1, 12)
        n = f.shape[0]
        if coeffs.shape[0] != n or g1.shape[0] != n:
            raise ValueError("f, coeffs and g1 must have one row per item")
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_ell_batch(self._h, _p64(f), _p64(coeffs), _p64(g1), n, _p64(out)))
        return out

    # ---- src/native/final_exponentiation.rs, fin_exp_helpers.rs
    def final_exponentiation_batch(self, fq12):
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        z = np.zeros(n, dtype=np.uint8)
        check(lib.bls381_final_exp_batch(self._h, _p64(fq12), n, _p64(out), _p8(z)))
        return out

    def easy_part_batch(self, fq12):
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        z = np.zeros(n, dtype=np.uint8)
        check(lib.bls381_easy_part_batch(self._h, _p64(fq12), n, _p64(out), _p8(z)))
        return out

    def hard_part_batch(self, fq12):
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_hard_part_batch(self._h, _p64(fq12), n, _p64(out)))
        return out

    def frobenius_map_batch(self, fq12, power):
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_frobenius_batch(self._h, _p64(fq12), power, n, _p64(out)))
        return out

    def conjugate_batch(self, fq12):
        """conjugate_fp12 (fin_exp_helpers.rs:132-142) per element."""
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_conjugate_batch(self._h, _p64(fq12), n, _p64(out)))
        return out

    def pow_native_batch(self, fq12, exp):
        """pow_native(a, exp) (fin_exp_helpers.rs:60-82) per element; exp: a non-negative int below 2^384 or a list
        of little-endian 64-bit words (the reference's Vec<u64>)."""
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        if isinstance(exp, int):
            if exp < 0 or exp >> 384:
                raise ValueError("exponent out of range")
            words = [(exp >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(max(1, (exp.bit_length() + 63) // 64))]
        else:
            words = [int(w) for w in exp]
        e = np.array(words, dtype=np.uint64)
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_pow_batch(self._h, _p64(fq12), _p64(e), len(words), n, _p64(out)))
        return out

    def pow_x_batch(self, fq12):
        """pow_native(a, vec![BLS_X])."""
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_pow_x_batch(self._h, _p64(fq12), n, _p64(out)))
        return out

    def fq12_mul_batch(self, a, b):
        a, b = _arr64(a, 72), _arr64(b, 72)
        n = a.shape[0]
        if b.shape[0] != n:
            raise ValueError("a and b must have one row per item")
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_fq12_mul_batch(self._h, _p64(a), _p64(b), n, _p64(out)))
        return out

    # ---- src/native/pairing.rs
    def pairing_batch(self, g1, g2, inf_flags=None):
        """native_pairing([p], [q]) per pair (pairing.rs:8-13)."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("zip_eq: length mismatch")
        fl = _flags(inf_flags, n)
        out = np.empty((n, 72), dtype=np.uint64)
        check(lib.bls381_pairing_batch(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out)))
        return out

    def native_pairing(self, g1, g2, inf_flags=None):
        """native_pairing over all pairs (pairing.rs:8-13): (Fq12, is_one)."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("zip_eq: length mismatch")
        fl = _flags(inf_flags, n)
        out = np.empty(72, dtype=np.uint64)
        one = C.c_int(0)
        check(lib.bls381_multi_pairing(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out), C.byref(one)))
        return out, bool(one.value)

    def finish_multi_pairing(self, partials):
        partials = _arr64(partials, 72)
        out = np.empty(72, dtype=np.uint64)
        one = C.c_int(0)
        check(lib.bls381_finish_multi_pairing(self._h, _p64(partials), partials.shape[0], _p64(out), C.byref(one)))
        return out, bool(one.value)

    # ---- witness side (src/fields/fq12_target.rs:357-409, src/pairing.rs:13-20)
    def fq12_inverse_limbs_batch(self, x):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        if x.ndim != 2 or x.shape[1] != 144:
            raise ValueError("expected shape (n, 144)")
        n = x.shape[0]
        inv = np.empty((n, 144), dtype=np.uint32)
        err = np.zeros(n, dtype=np.uint8)
        check(lib.bls381_fq12_inverse_limbs_batch(self._h, _p32(x), n, _p32(inv), _p8(err)))
        return inv

    # ---- the other witness generators (canonical 12 x u32 limbs in and out)
    def _limbs_call(self, fn, x, cols, sgn=None):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        if x.ndim != 2 or x.shape[1] != cols:
            raise ValueError("expected shape (n, %d)" % cols)
        n = x.shape[0]
        out = np.empty((n, cols), dtype=np.uint32)
        err = np.zeros(n, dtype=np.uint8)
        if sgn is None:
            rc = fn(self._h, _p32(x), n, _p32(out), _p8(err))
        else:
            sgn = np.ascontiguousarray(sgn, dtype=np.uint8)
            if sgn.shape != (n,):
                raise ValueError("sgn must have shape (n,)")
            rc = fn(self._h, _p32(x), _p8(sgn), n, _p32(out), _p8(err))
        return rc, out, err

    def fq_sqrt_limbs_batch(self, x, sgn, return_err=False):
        """FqSqrtGenerator::run_once (fields/fq_target.rs:296-323) per element."""
        rc, out, err = self._limbs_call(lib.bls381_fq_sqrt_limbs_batch, x, 12, sgn)
        if return_err:
            return out, err
        check(rc)
        return out

    def fq2_sqrt_limbs_batch(self, x, sgn, return_err=False):
        """Fq2SqrtGenerator::run_once (fields/fq2_target.rs:401-428) per element."""
        rc, out, err = self._limbs_call(lib.bls381_fq2_sqrt_limbs_batch, x, 24, sgn)
        if return_err:
            return out, err
        check(rc)
        return out

    def fq2_inverse_limbs_batch(self, x):
        """Fq2InverseGenerator::run_once (fields/fq2_target.rs:347-369) per element."""
        rc, out, err = self._limbs_call(lib.bls381_fq2_inverse_limbs_batch, x, 24)
        check(rc)
        return out

    def fq6_inverse_limbs_batch(self, x, return_err=False):
        """Fq6InverseGenerator::run_once (fields/fq6_target.rs:350-371) per element, MyFq6 order."""
        rc, out, err = self._limbs_call(lib.bls381_fq6_inverse_limbs_batch, x, 72)
        if return_err:
            return out, err
        check(rc)
        return out

    def _two_pairs(self, g1a, g2a, g1b, g2b, inf_flags, with_final_exp):
        g1a, g2a, g1b, g2b = _arr64(g1a, 12), _arr64(g2a, 24), _arr64(g1b, 12), _arr64(g2b, 24)
        n = g1a.shape[0]
        if not (g2a.shape[0] == g1b.shape[0] == g2b.shape[0] == n):
            raise ValueError("the four point arrays must have one row per item")
        f = _flags(inf_flags, n)
        out = np.zeros((n, 72), dtype=np.uint64)
        if with_final_exp:
            is_one = np.zeros(n, dtype=np.uint8)
            check(lib.bls381_pairing_two_pairs_batch(self._h, _p64(g1a), _p64(g2a), _p64(g1b), _p64(g2b), _p8(f), n,
                                                     _p64(out), _p8(is_one)))
            return out, is_one
        check(lib.bls381_miller_loop_two_pairs_batch(self._h, _p64(g1a), _p64(g2a), _p64(g1b), _p64(g2b), _p8(f), n, _p64(out)))
        return out

    def miller_loop_two_pairs_batch(self, g1a, g2a, g1b, g2b, inf_flags=None):
        """multi_miller_loop_native([pa, pb], [qa, qb]) per item (native/miller_loop.rs:121-162)."""
        return self._two_pairs(g1a, g2a, g1b, g2b, inf_flags, False)

    def pairing_two_pairs_batch(self, g1a, g2a, g1b, g2b, inf_flags=None):
        """native_pairing([pa, pb], [qa, qb]) per item -> ((n, 72) u64, (n,) result-is-one flags)."""
        return self._two_pairs(g1a, g2a, g1b, g2b, inf_flags, True)

    # ---- compressed points and aggregation (ark-bls12-381 deserialize_compressed / serialize_compressed)
    def _decompress(self, fn, data, nbytes, cols, check_subgroup, return_status):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if data.ndim != 2 or data.shape[1] != nbytes:
            raise ValueError("expected shape (n, %d)" % nbytes)
        n = data.shape[0]
        out = np.zeros((n, cols), dtype=np.uint64)
        status = np.zeros(n, dtype=np.uint8)
        rc = fn(self._h, _p8(data), n, 1 if check_subgroup else 0, _p64(out), _p8(status))
        if return_status:
            return out, status
        check(rc)
        return out, status

    def g1_decompress_batch(self, data, check_subgroup=True, return_status=False):
        """(n, 48) bytes -> ((n, 12) u64 affine points, (n,) status); raises on a rejected point unless
        return_status."""
        return self._decompress(lib.bls381_g1_decompress_batch, data, 48, 12, check_subgroup, return_status)

    def g2_decompress_batch(self, data, check_subgroup=True, return_status=False):
        return self._decompress(lib.bls381_g2_decompress_batch, data, 96, 24, check_subgroup, return_status)

    def _compress(self, fn, pts, cols, nbytes, inf):
        pts = _arr64(pts, cols)
        n = pts.shape[0]
        out = np.zeros((n, nbytes), dtype=np.uint8)
        inf = _flags(inf, n)
        check(fn(self._h, _p64(pts), _p8(inf), n, _p8(out)))
        return out

    def g1_compress_batch(self, g1, inf=None):
        return self._compress(lib.bls381_g1_compress_batch, g1, 12, 48, inf)

    def g2_compress_batch(self, g2, inf=None):
        return self._compress(lib.bls381_g2_compress_batch, g2, 24, 96, inf)

    def _aggregate(self, fn, pts, cols, group_size, inf):
        pts = _arr64(pts, cols)
        n = pts.shape[0]
        if group_size <= 0 or n % group_size:
            raise ValueError("the number of points must be a multiple of group_size")
        groups = n // group_size
        out = np.zeros((groups, cols), dtype=np.uint64)
        out_inf = np.zeros(groups, dtype=np.uint8)
        inf = _flags(inf, n)
        check(fn(self._h, _p64(pts), _p8(inf), groups, group_size, _p64(out), _p8(out_inf)))
        return out, out_inf

    def g1_aggregate_batch(self, g1, group_size, inf=None):
        """Sum of each run of group_size points -> ((groups, 12) u64, (groups,) infinity flags)."""
        return self._aggregate(lib.bls381_g1_aggregate_batch, g1, 12, group_size, inf)

    def g2_aggregate_batch(self, g2, group_size, inf=None):
        return self._aggregate(lib.bls381_g2_aggregate_batch, g2, 24, group_size, inf)

    def verify_batch(self, pk, msg, sig, check_subgroup=True):
        """Signature checks from bytes: (n, 48) keys, (n, 96) message points, (n, 96) signatures ->
        ((n,) verdicts, (n, 3) decoder statuses).  check_subgroup: True (all three inputs), False (none) or a
        mask of CHECK_KEYS | CHECK_MESSAGES | CHECK_SIGNATURES."""
        if check_subgroup is True:
            check_subgroup = 7
        elif check_subgroup is False:
            check_subgroup = 0
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        msg = np.ascontiguousarray(msg, dtype=np.uint8)
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        n = pk.shape[0]
        if pk.shape != (n, 48) or msg.shape != (n, 96) or sig.shape != (n, 96):
            raise ValueError("expected shapes (n, 48), (n, 96), (n, 96)")
        verdict = np.zeros(n, dtype=np.uint8)
        status = np.zeros((n, 3), dtype=np.uint8)
        check(lib.bls381_verify_batch(self._h, _p8(pk), _p8(msg), _p8(sig), n, int(check_subgroup), _p8(verdict), _p8(status)))
        return verdict, status

    def fast_aggregate_verify(self, pk, msg, sig, check_subgroup=True):
        """One signature (96 bytes) by the (n, 48) keys over one message point (96 bytes) -> (valid, key statuses)."""
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        msg = np.ascontiguousarray(np.frombuffer(bytes(msg), dtype=np.uint8))
        sig = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
        n = pk.shape[0]
        if pk.shape != (n, 48) or msg.shape != (96,) or sig.shape != (96,):
            raise ValueError("expected (n, 48) keys and 96-byte message point and signature")
        if check_subgroup is True:
            check_subgroup = 7
        elif check_subgroup is False:
            check_subgroup = 0
        valid = _lib.C.c_int(0)
        status = np.zeros(n, dtype=np.uint8)
        check(lib.bls381_fast_aggregate_verify(self._h, _p8(pk), n, _p8(msg), _p8(sig), int(check_subgroup),
                                               _lib.C.byref(valid), _p8(status)))
        return bool(valid.value), status

    def fast_aggregate_verify_batch(self, pk, msg, sig, check_subgroup=True):
        """(checks, keys, 48) keys, (checks, 96) message points, (checks, 96) signatures ->
        ((checks,) verdicts, (checks, keys) key statuses)."""
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        msg = np.ascontiguousarray(msg, dtype=np.uint8)
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        if pk.ndim != 3 or pk.shape[2] != 48 or msg.shape != (pk.shape[0], 96) or sig.shape != (pk.shape[0], 96):
            raise ValueError("expected shapes (checks, keys, 48), (checks, 96), (checks, 96)")
        if check_subgroup is True:
            check_subgroup = 7
        elif check_subgroup is False:
            check_subgroup = 0
        g, k = pk.shape[0], pk.shape[1]
        verdict = np.zeros(g, dtype=np.uint8)
        status = np.zeros((g, k), dtype=np.uint8)
        check(lib.bls381_fast_aggregate_verify_batch(self._h, _p8(pk), g, k, _p8(msg), _p8(sig), int(check_subgroup),
                                                     _p8(verdict), _p8(status)))
        return verdict, status

    def pairing_witness_batch(self, g1, g2, out=None):
        """The witness trace of n proofs, (n, rows, 12) u32.  `out`: an array of that shape to write into (a caller that
        produces traces repeatedly keeps one: 6.8 MB per proof of fresh memory are page faults on every call)."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 must have one row per proof")
        rows = lib.bls381_pairing_witness_rows()
        if out is None:
            out = np.empty((n, rows, 12), dtype=np.uint32)
        elif out.shape != (n, rows, 12) or out.dtype != np.uint32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint32 array of shape (n, rows, 12)")
        check(lib.bls381_pairing_witness_batch(self._h, _p64(g1), _p64(g2), n, _p32(out)))
        return out

    def pairing_witness_aux_batch(self, g1, g2, want=(True, True, True)):
        """The trace plus the gadgets' auxiliary values: (trace, mul_quot (n, mul rows, 24), carry (n, rows),
        inv_quot (n, 24)).  `want` says which of the three auxiliary outputs to ask for; the others are passed as
        NULL, which the C ABI allows, and come back as None."""
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 must have one row per proof")
        rows, n_mul = lib.bls381_pairing_witness_rows(), lib.bls381_pairing_witness_mul_rows()
        trace = np.empty((n, rows, 12), dtype=np.uint32)
        mq = np.empty((n, n_mul, 24), dtype=np.uint32) if want[0] else None
        cy = np.empty((n, rows), dtype=np.uint8) if want[1] else None
        iq = np.empty((n, 24), dtype=np.uint32) if want[2] else None
        check(lib.bls381_pairing_witness_aux_batch(self._h, _p64(g1), _p64(g2), n, _p32(trace), _p32(mq) if want[0] else None,
                                                   _p8(cy) if want[1] else None, _p32(iq) if want[2] else None))
        return trace, mq, cy, iq

    # ---- measurement
    def last_timing(self):
        ms = (C.c_float * 4)()
        check(lib.bls381_last_timing(self._h, ms))
        return list(ms)

    def last_chain_timing(self):
        """(ms of the five squaring-chain launches, ms of the six programs around them) of the last final
        exponentiation, or None when it ran as the single program."""
        ms = (C.c_float * 2)()
        used = C.c_int(0)
        check(lib.bls381_last_chain_timing(self._h, ms, C.byref(used)))
        return (ms[0], ms[1]) if used.value else None

    def calibrate_mac_rate(self):
        v = C.c_double(0)
        check(lib.bls381_calibrate_mac_rate(self._h, C.byref(v)))
        return v.value

    @property
    def handle(self):
        return self._h


class MultiContext:
    """Several GPUs behind one handle (bls381_multi_ctx): contiguous shards, one host thread per device."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        check(lib.bls381_multi_ctx_create(C.byref(self._h), ids, len(device_ids)))

    def close(self):
        if self._h:
            lib.bls381_multi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def devices(self):
        return lib.bls381_multi_ctx_devices(self._h)

    def _pairs(self, fn, g1, g2, inf_flags):
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("zip_eq: length mismatch")
        fl = _flags(inf_flags, n)
        out = np.empty((n, 72), dtype=np.uint64)
        check(fn(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out)))
        return out

    def pairing_batch(self, g1, g2, inf_flags=None):
        return self._pairs(lib.bls381_multi_ctx_pairing_batch, g1, g2, inf_flags)

    def miller_loop_batch(self, g1, g2, inf_flags=None):
        return self._pairs(lib.bls381_multi_ctx_miller_loop_batch, g1, g2, inf_flags)

    def final_exponentiation_batch(self, fq12):
        fq12 = _arr64(fq12, 72)
        n = fq12.shape[0]
        out = np.empty((n, 72), dtype=np.uint64)
        z = np.zeros(n, dtype=np.uint8)
        check(lib.bls381_multi_ctx_final_exp_batch(self._h, _p64(fq12), n, _p64(out), _p8(z)))
        return out

    def multi_miller_loop_native(self, g1, g2, inf_flags=None):
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        fl = _flags(inf_flags, n)
        out = np.empty(72, dtype=np.uint64)
        check(lib.bls381_multi_ctx_multi_miller_loop(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out)))
        return out

    def native_pairing(self, g1, g2, inf_flags=None):
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        fl = _flags(inf_flags, n)
        out = np.empty(72, dtype=np.uint64)
        one = C.c_int(0)
        check(lib.bls381_multi_ctx_multi_pairing(self._h, _p64(g1), _p64(g2), _p8(fl), n, _p64(out), C.byref(one)))
        return out, bool(one.value)

    def pairing_witness_batch(self, g1, g2):
        g1, g2 = _arr64(g1, 12), _arr64(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 must have one row per proof")
        trace = np.empty((n, lib.bls381_pairing_witness_rows(), 12), dtype=np.uint32)
        check(lib.bls381_multi_ctx_pairing_witness_batch(self._h, _p64(g1), _p64(g2), n, _p32(trace)))
        return trace

    def verify_batch(self, pk, msg, sig, check_subgroup=7):
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        msg = np.ascontiguousarray(msg, dtype=np.uint8)
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        n = pk.shape[0]
        if pk.shape != (n, 48) or msg.shape != (n, 96) or sig.shape != (n, 96):
            raise ValueError("expected shapes (n, 48), (n, 96), (n, 96)")
        verdict = np.zeros(n, dtype=np.uint8)
        status = np.zeros((n, 3), dtype=np.uint8)
        check(lib.bls381_multi_ctx_verify_batch(self._h, _p8(pk), _p8(msg), _p8(sig), n, int(check_subgroup), _p8(verdict), _p8(status)))
        return verdict, status


def witness_rows():
    return lib.bls381_pairing_witness_rows()


def witness_kinds():
    n = lib.bls381_pairing_witness_rows()
    k = np.empty(n, dtype=np.uint8)
    check(lib.bls381_pairing_witness_kinds(_p8(k), n))
    return k
