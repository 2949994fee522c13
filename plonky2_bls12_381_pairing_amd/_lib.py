"""ctypes loader for libbls381_hip.so (the C ABI declared in include/bls381_hip.h).

The library is built in-tree by __graft_entry__.build() / `make -C plonky2_bls12_381_pairing_amd/csrc`.
There is no CPU implementation behind these bindings: without the shared library the import fails,
and without a gfx950 device every compute call returns an error.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLS381_LIB_PATH") or os.path.join(_HERE, "libbls381_hip.so")  # (the override: A/B builds)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libbls381_hip.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950); there is no CPU fallback")

# One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so under the soname of the system's
# (/opt/rocm), and the dynamic loader hands whichever was loaded first to every later user of that soname.  Torch on
# the system's copy can fail to find its GPUs, so where torch is installed its copy is loaded first, whether or not
# torch itself has been imported yet: this library and a later `import torch` then share it in either order.
# (BLS381_NO_TORCH_RUNTIME=1 turns this off: the library then binds the runtime the dynamic loader finds.)
PRELOADED_HIP_RUNTIME = None  # path of the libamdhip64.so this module loaded ahead of the library, if it did
if "torch" not in sys.modules and os.environ.get("BLS381_NO_TORCH_RUNTIME") != "1":
    try:
        _spec = importlib.util.find_spec("torch")
        _dirs = list(_spec.submodule_search_locations) if _spec and _spec.submodule_search_locations else []
        _hip = os.path.join(_dirs[0], "lib", "libamdhip64.so") if _dirs else ""
        if _hip and os.path.exists(_hip):
            C.CDLL(_hip, mode=C.RTLD_GLOBAL)
            PRELOADED_HIP_RUNTIME = _hip
    except (ImportError, OSError, ValueError):
        pass  # no torch, or its runtime does not load here: the system's runtime serves this library alone

lib = C.CDLL(LIB_PATH)


def hip_runtime_in_use():
    """Paths of the libamdhip64 copies mapped into this process (one, when the loader above did its job) and the
    version hipRuntimeGetVersion reports for the one the library is bound to (None if it cannot be asked)."""
    paths = []
    try:
        with open("/proc/self/maps") as f:
            paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        pass
    version = None
    try:
        rt = C.CDLL(paths[0]) if paths else None
        if rt is not None:
            v = C.c_int(0)
            if rt.hipRuntimeGetVersion(C.byref(v)) == 0:
                version = v.value
    except (OSError, AttributeError):
        pass
    return {"paths": paths, "runtime_version": version, "preloaded": PRELOADED_HIP_RUNTIME}

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
ctxp = C.c_void_p

def _sig(name, restype, *argtypes):
    if os.environ.get("BLS381_LIB_PATH") and not hasattr(lib, name):
        return None  # an older build loaded for an A/B comparison may lack the newer entry points
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f

_sig("bls381_ctx_create", C.c_int, C.POINTER(ctxp), C.c_int)
_sig("bls381_ctx_create_ex", C.c_int, C.POINTER(ctxp), C.c_int, C.c_int, C.c_int)
_sig("bls381_ctx_destroy", None, ctxp)
_sig("bls381_ctx_reserve", C.c_int, ctxp, C.c_size_t)
_sig("bls381_last_error", C.c_char_p)
_sig("bls381_miller_loop_batch", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_final_exp_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p, u8p)
_sig("bls381_pairing_batch", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_multi_miller_loop", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_multi_pairing", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p, C.POINTER(C.c_int))
_sig("bls381_finish_multi_pairing", C.c_int, ctxp, u64p, C.c_size_t, u64p, C.POINTER(C.c_int))
_sig("bls381_g2_prepare_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p)
_sig("bls381_ell_batch", C.c_int, ctxp, u64p, u64p, u64p, C.c_size_t, u64p)
_sig("bls381_easy_part_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p, u8p)
_sig("bls381_hard_part_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p)
_sig("bls381_frobenius_batch", C.c_int, ctxp, u64p, C.c_uint, C.c_size_t, u64p)
_sig("bls381_conjugate_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p)
_sig("bls381_pow_batch", C.c_int, ctxp, u64p, u64p, C.c_size_t, C.c_size_t, u64p)
_sig("bls381_pow_x_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p)
_sig("bls381_fq12_mul_batch", C.c_int, ctxp, u64p, u64p, C.c_size_t, u64p)
_sig("bls381_fq12_inverse_limbs_batch", C.c_int, ctxp, u32p, C.c_size_t, u32p, u8p)
_sig("bls381_pairing_witness_rows", C.c_size_t)
_sig("bls381_pairing_witness_kinds", C.c_int, u8p, C.c_size_t)
_sig("bls381_pairing_witness_batch", C.c_int, ctxp, u64p, u64p, C.c_size_t, u32p)
_sig("bls381_pairing_witness_mul_rows", C.c_size_t)
_sig("bls381_pairing_witness_aux_batch", C.c_int, ctxp, u64p, u64p, C.c_size_t, u32p, u32p, u8p, u32p)
_sig("bls381_pairing_witness_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("bls381_pairing_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("bls381_pairing_batch_dev_noalloc", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("bls381_miller_loop_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("bls381_final_exp_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
_sig("bls381_g2_prepared_create", C.c_int, ctxp, u64p, C.c_int, C.POINTER(C.c_void_p))
_sig("bls381_g2_prepared_destroy", None, C.c_void_p)
_sig("bls381_miller_loop_fixed_q_batch", C.c_int, ctxp, C.c_void_p, u64p, C.POINTER(C.c_uint8), C.c_size_t, u64p)
_sig("bls381_pairing_fixed_q_batch", C.c_int, ctxp, C.c_void_p, u64p, C.POINTER(C.c_uint8), C.c_size_t, u64p)
_sig("bls381_pairing_fixed_q_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("bls381_fq_sqrt_limbs_batch", C.c_int, ctxp, u32p, C.POINTER(C.c_uint8), C.c_size_t, u32p, C.POINTER(C.c_uint8))
_sig("bls381_fq2_sqrt_limbs_batch", C.c_int, ctxp, u32p, C.POINTER(C.c_uint8), C.c_size_t, u32p, C.POINTER(C.c_uint8))
_sig("bls381_fq2_inverse_limbs_batch", C.c_int, ctxp, u32p, C.c_size_t, u32p, C.POINTER(C.c_uint8))
_sig("bls381_fq6_inverse_limbs_batch", C.c_int, ctxp, u32p, C.c_size_t, u32p, C.POINTER(C.c_uint8))
_sig("bls381_g1_decompress_batch", C.c_int, ctxp, u8p, C.c_size_t, C.c_int, u64p, u8p)
_sig("bls381_g2_decompress_batch", C.c_int, ctxp, u8p, C.c_size_t, C.c_int, u64p, u8p)
_sig("bls381_g1_compress_batch", C.c_int, ctxp, u64p, u8p, C.c_size_t, u8p)
_sig("bls381_g2_compress_batch", C.c_int, ctxp, u64p, u8p, C.c_size_t, u8p)
_sig("bls381_g1_aggregate_batch", C.c_int, ctxp, u64p, u8p, C.c_size_t, C.c_size_t, u64p, u8p)
_sig("bls381_g2_aggregate_batch", C.c_int, ctxp, u64p, u8p, C.c_size_t, C.c_size_t, u64p, u8p)
_sig("bls381_miller_loop_two_pairs_batch", C.c_int, ctxp, u64p, u64p, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_pairing_two_pairs_batch", C.c_int, ctxp, u64p, u64p, u64p, u64p, u8p, C.c_size_t, u64p, u8p)
_sig("bls381_pairing_two_pairs_batch_dev", C.c_int, ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
_sig("bls381_verify_batch", C.c_int, ctxp, u8p, u8p, u8p, C.c_size_t, C.c_int, u8p, u8p)
_sig("bls381_fast_aggregate_verify", C.c_int, ctxp, u8p, C.c_size_t, u8p, u8p, C.c_int, C.POINTER(C.c_int), u8p)
_sig("bls381_fast_aggregate_verify_batch", C.c_int, ctxp, u8p, C.c_size_t, C.c_size_t, u8p, u8p, C.c_int, u8p, u8p)
_sig("bls381_multi_ctx_create", C.c_int, C.POINTER(ctxp), C.POINTER(C.c_int), C.c_int)
_sig("bls381_multi_ctx_destroy", None, ctxp)
_sig("bls381_multi_ctx_devices", C.c_int, ctxp)
_sig("bls381_multi_ctx_pairing_batch", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_multi_ctx_miller_loop_batch", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_multi_ctx_final_exp_batch", C.c_int, ctxp, u64p, C.c_size_t, u64p, u8p)
_sig("bls381_multi_ctx_multi_miller_loop", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p)
_sig("bls381_multi_ctx_multi_pairing", C.c_int, ctxp, u64p, u64p, u8p, C.c_size_t, u64p, C.POINTER(C.c_int))
_sig("bls381_multi_ctx_pairing_witness_batch", C.c_int, ctxp, u64p, u64p, C.c_size_t, u32p)
_sig("bls381_multi_ctx_verify_batch", C.c_int, ctxp, u8p, u8p, u8p, C.c_size_t, C.c_int, u8p, u8p)
_sig("bls381_debug_program", C.c_int, C.c_char_p, C.c_int, C.c_int, u32p, u32p, C.c_size_t, u32p)
_sig("bls381_debug_constants", C.c_int, u32p, C.c_size_t)
_sig("bls381_debug_witness_cache_plan", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64))
_sig("bls381_debug_witness_ops", C.c_int, C.c_int, u32p, C.c_size_t, u32p, C.c_size_t, C.POINTER(C.c_uint64))
_sig("bls381_debug_profile", C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64))
_sig("bls381_debug_fp_mul", C.c_int, ctxp, u32p, u32p, C.c_size_t, u32p, u32p)
_sig("bls381_debug_fp_inverse", C.c_int, ctxp, u32p, C.c_size_t, u32p, C.POINTER(C.c_uint8))
_sig("bls381_debug_cyc_chain", C.c_int, ctxp, u32p, C.c_size_t, u32p)
_sig("bls381_last_chain_timing", C.c_int, ctxp, C.POINTER(C.c_float), C.POINTER(C.c_int))
_sig("bls381_last_timing", C.c_int, ctxp, C.POINTER(C.c_float))
_sig("bls381_calibrate_mac_rate", C.c_int, ctxp, C.POINTER(C.c_double))

# every symbol include/bls381_hip.h declares
EXPORTED = [
    "bls381_ctx_create", "bls381_ctx_create_ex", "bls381_ctx_destroy", "bls381_ctx_reserve", "bls381_last_error",
    "bls381_miller_loop_batch", "bls381_final_exp_batch", "bls381_pairing_batch", "bls381_multi_miller_loop",
    "bls381_multi_pairing", "bls381_finish_multi_pairing", "bls381_g2_prepare_batch", "bls381_ell_batch",
    "bls381_easy_part_batch", "bls381_hard_part_batch", "bls381_frobenius_batch", "bls381_conjugate_batch", "bls381_pow_batch", "bls381_pow_x_batch", "bls381_fq12_mul_batch",
    "bls381_fq12_inverse_limbs_batch", "bls381_pairing_witness_rows", "bls381_pairing_witness_kinds",
    "bls381_pairing_witness_batch", "bls381_pairing_witness_mul_rows", "bls381_pairing_witness_aux_batch", "bls381_pairing_witness_batch_dev", "bls381_pairing_batch_dev", "bls381_pairing_batch_dev_noalloc", "bls381_miller_loop_batch_dev",
    "bls381_final_exp_batch_dev", "bls381_g2_prepared_create", "bls381_g2_prepared_destroy",
    "bls381_fq_sqrt_limbs_batch", "bls381_fq2_sqrt_limbs_batch", "bls381_fq2_inverse_limbs_batch", "bls381_fq6_inverse_limbs_batch",
    "bls381_miller_loop_fixed_q_batch", "bls381_pairing_fixed_q_batch", "bls381_pairing_fixed_q_batch_dev", "bls381_debug_program", "bls381_debug_constants", "bls381_debug_witness_cache_plan", "bls381_debug_witness_ops", "bls381_debug_profile", "bls381_debug_fp_mul", "bls381_debug_fp_inverse", "bls381_debug_cyc_chain", "bls381_last_chain_timing", "bls381_last_timing",
    "bls381_calibrate_mac_rate", "bls381_g1_decompress_batch", "bls381_g2_decompress_batch", "bls381_g1_compress_batch",
    "bls381_g2_compress_batch", "bls381_g1_aggregate_batch", "bls381_g2_aggregate_batch",
    "bls381_multi_ctx_create", "bls381_multi_ctx_destroy", "bls381_multi_ctx_devices", "bls381_multi_ctx_pairing_batch",
    "bls381_multi_ctx_miller_loop_batch", "bls381_multi_ctx_final_exp_batch", "bls381_multi_ctx_multi_miller_loop",
    "bls381_multi_ctx_multi_pairing", "bls381_multi_ctx_pairing_witness_batch", "bls381_multi_ctx_verify_batch",
    "bls381_miller_loop_two_pairs_batch", "bls381_pairing_two_pairs_batch", "bls381_pairing_two_pairs_batch_dev", "bls381_verify_batch", "bls381_fast_aggregate_verify", "bls381_fast_aggregate_verify_batch",
]


class Bls381Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bls381 error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise Bls381Error(rc, lib.bls381_last_error().decode())
