"""ctypes binding of oracle/c/liboracle.so (test infrastructure; see bls381_oracle.c)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "c", "liboracle.so")
lib = C.CDLL(_PATH)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
lib.orc_pairing_batch.restype = C.c_int
lib.orc_pairing_batch.argtypes = [_u64p, _u64p, _u8p, C.c_size_t, _u64p, C.c_int]
lib.orc_final_exp_batch.restype = C.c_int
lib.orc_final_exp_batch.argtypes = [_u64p, C.c_size_t, _u64p]
lib.orc_multi_miller_loop.restype = None
lib.orc_multi_miller_loop.argtypes = [_u64p, _u64p, _u8p, C.c_size_t, _u64p]
for _n, _a in (("orc_easy_part", [_u64p, _u64p]), ("orc_final_exponentiation", [_u64p, _u64p]),
               ("orc_fq12_inverse", [_u64p, _u64p])):
    getattr(lib, _n).restype = C.c_int
    getattr(lib, _n).argtypes = _a
for _n, _a in (("orc_hard_part", [_u64p, _u64p]), ("orc_g2_prepare", [_u64p, _u64p]),
               ("orc_ell", [_u64p, _u64p, _u64p, _u64p]), ("orc_fq12_mul", [_u64p, _u64p, _u64p]),
               ("orc_frobenius", [_u64p, C.c_int, _u64p])):
    getattr(lib, _n).restype = None
    getattr(lib, _n).argtypes = _a


def _p(a):
    return a.ctypes.data_as(_u64p)


def pairing_batch(g1, g2, flags=None, final_exp=True):
    g1 = np.ascontiguousarray(g1, dtype=np.uint64); g2 = np.ascontiguousarray(g2, dtype=np.uint64)
    n = g1.shape[0]
    out = np.empty((n, 72), dtype=np.uint64)
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8).ctypes.data_as(_u8p)
    rc = lib.orc_pairing_batch(_p(g1), _p(g2), f, n, _p(out), 1 if final_exp else 0)
    if rc:
        raise ZeroDivisionError("final exponentiation of zero")
    return out


def multi_miller_loop(g1, g2, flags=None):
    g1 = np.ascontiguousarray(g1, dtype=np.uint64); g2 = np.ascontiguousarray(g2, dtype=np.uint64)
    out = np.empty(72, dtype=np.uint64)
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8).ctypes.data_as(_u8p)
    lib.orc_multi_miller_loop(_p(g1), _p(g2), f, g1.shape[0], _p(out))
    return out


def final_exp_batch(x):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty_like(x)
    if lib.orc_final_exp_batch(_p(x), x.shape[0], _p(out)):
        raise ZeroDivisionError("final exponentiation of zero")
    return out


def _unary(fn, x, *extra):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty(72, dtype=np.uint64)
    r = fn(_p(x), *extra, _p(out))
    if r:
        raise ZeroDivisionError
    return out


def easy_part(x): return _unary(lib.orc_easy_part, x)
def hard_part(x): return _unary(lib.orc_hard_part, x)
def fq12_inverse(x): return _unary(lib.orc_fq12_inverse, x)
def frobenius(x, power): return _unary(lib.orc_frobenius, x, power)


def fq12_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty(72, dtype=np.uint64)
    lib.orc_fq12_mul(_p(a), _p(b), _p(out))
    return out


def g2_prepare(q):
    q = np.ascontiguousarray(q, dtype=np.uint64)
    out = np.empty(68 * 36, dtype=np.uint64)
    lib.orc_g2_prepare(_p(q), _p(out))
    return out.reshape(68, 3, 2, 6)


def ell(f, coeff, p):
    f = np.ascontiguousarray(f, dtype=np.uint64); coeff = np.ascontiguousarray(coeff, dtype=np.uint64)
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = np.empty(72, dtype=np.uint64)
    lib.orc_ell(_p(f), _p(coeff), _p(p), _p(out))
    return out


lib.orc_set_threads.restype = None
lib.orc_set_threads.argtypes = [C.c_int]
lib.orc_get_max_threads.restype = C.c_int


def set_fast(on):
    """Optimised CPU variant (cyclotomic squarings, Karatsuba tower products) instead of the reference's algorithm."""
    lib.orc_set_fast.restype = None
    lib.orc_set_fast.argtypes = [C.c_int]
    lib.orc_set_fast(int(bool(on)))


def set_threads(n):
    lib.orc_set_threads(int(n))
