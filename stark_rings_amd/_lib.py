"""ctypes binding of libstarkrings_hip.so (C ABI declared in include/stark_rings_hip.h).

The library is the product; this module only loads it.  There is no CPU fallback: if the shared
object is missing, or no HIP device is present, callers get an exception.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SR_LIB_PATH") or os.path.join(HERE, "libstarkrings_hip.so")

u64p = ctypes.POINTER(ctypes.c_uint64)
_c = ctypes

# every symbol include/stark_rings_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sr_ctx_create": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "sr_ctx_create_ex": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(_c.c_void_p)]),
    "sr_ctx_create_group": (_c.c_int, [_c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.c_int, _c.c_void_p, _c.POINTER(_c.c_void_p)]),
    "sr_shard_range": (_c.c_int, [_c.c_size_t, _c.c_int, _c.c_int, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
    "sr_ctx_reserve_scratch": (_c.c_int, [_c.c_void_p, _c.c_size_t]),
    "sr_ctx_plan_in_use": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_size_t)]),
    "sr_ctx_destroy": (_c.c_int, [_c.c_void_p]),
    "sr_ctx_degree": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_size_t)]),
    "sr_ctx_limbs": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "sr_ctx_twiddle_block": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)]),
    "sr_ctx_twiddles_updated": (_c.c_int, [_c.c_void_p]),
    "sr_ntt_fwd_batch": (_c.c_int, [_c.c_void_p, u64p, _c.c_size_t]),
    "sr_ntt_inv_batch": (_c.c_int, [_c.c_void_p, u64p, _c.c_size_t]),
    "sr_pointwise_mul_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_add_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_sub_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_ring_mul_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, u64p, _c.c_size_t]),
    "sr_neg_batch": (_c.c_int, [_c.c_void_p, u64p, _c.c_size_t]),
    "sr_scale_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_add_scalar_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_int, _c.c_size_t]),
    "sr_neg_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_scale_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, u64p, _c.c_size_t, _c.c_void_p]),
    "sr_mul_elem_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_mul_elem_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_sum_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_product_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, _c.c_size_t]),
    "sr_sum_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_product_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_add_scalar_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, u64p, _c.c_int, _c.c_size_t, _c.c_void_p]),
    "sr_reduce_batch": (_c.c_int, [_c.c_void_p, u64p, _c.c_size_t, u64p, _c.c_size_t]),
    "sr_ntt_fwd_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_ntt_inv_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_pointwise_mul_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_add_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_sub_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_matvec_ntt_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_spmv_ntt_dev": (_c.c_int, [_c.c_void_p] * 6 + [_c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_spmv_bad_index_count": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_ulonglong), _c.c_void_p]),
    "sr_matmul_ntt_dev": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_size_t] * 3 + [_c.c_void_p]),
    "sr_matvec_ntt": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_size_t] * 2),
    "sr_spmv_ntt": (_c.c_int, [_c.c_void_p] * 6 + [_c.c_size_t] * 2),
    "sr_matmul_ntt": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_size_t] * 3),
    "sr_decompose_balanced_batch_dev": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_decompose_overflow_count": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_ulonglong), _c.c_void_p]),
    "sr_recompose_batch_dev": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_decompose_balanced_batch": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_size_t, _c.c_size_t]),
    "sr_recompose_batch": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_size_t, _c.c_size_t]),
    "sr_decompose_balanced_batch_wide_dev": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_uint64, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_recompose_batch_wide_dev": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_uint64, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "sr_decompose_balanced_batch_wide": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_uint64, _c.c_size_t, _c.c_size_t]),
    "sr_recompose_batch_wide": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_uint64, _c.c_uint64, _c.c_size_t, _c.c_size_t]),
    "sr_wire_coeff_bytes": (_c.c_size_t, [_c.c_void_p]),
    "sr_serialize_batch_dev": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_size_t, _c.c_void_p]),
    "sr_deserialize_batch_dev": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_size_t, _c.c_void_p]),
    "sr_wire_invalid_count": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_ulonglong), _c.c_void_p]),
    "sr_serialize_batch": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_size_t]),
    "sr_deserialize_batch": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_size_t]),
    "sr_rot_batch_dev": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_size_t, _c.c_void_p]),
    "sr_rot_batch": (_c.c_int, [_c.c_void_p, u64p, _c.c_size_t]),
    "sr_ring_mul_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_ring_mul_ntt_rhs_batch": (_c.c_int, [_c.c_void_p, u64p, u64p, u64p, _c.c_size_t]),
    "sr_ring_mul_ntt_rhs_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_pack32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_unpack32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_ntt_fwd_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_ntt_inv_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_ring_mul_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_pointwise_mul_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_add_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_sub_packed32_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_reduce_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "sr_fill_uniform_dev": (_c.c_int, [_c.c_void_p, _c.c_uint64, _c.c_uint64, _c.c_size_t, _c.c_void_p, _c.c_void_p]),
    "sr_count_noncanonical_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint64), _c.c_void_p]),
    "sr_ctx_profile_enable": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "sr_ctx_profile_read": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64)]),
    "sr_ctx_profile_read_sampled": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    "sr_selftest_field_op": (_c.c_int, [_c.c_int, _c.c_int, u64p, u64p, u64p]),
    "sr_selftest_rep_counters": (_c.c_int, [u64p, _c.c_int]),
    "sr_last_error_string": (_c.c_char_p, []),
    "sr_version": (_c.c_char_p, []),
}



class Plan(ctypes.Structure):
    """sr_plan of include/stark_rings_hip.h: the kernel plan of a context, fixed when it is created."""
    _fields_ = [("flags", _c.c_uint32), ("log_tile", _c.c_int32), ("stark_whole_max", _c.c_int32), ("chunk_polys", _c.c_uint32),
                ("scratch_limit_bytes", _c.c_uint64), ("host_chunk_mb", _c.c_uint32), ("lanes", _c.c_uint32)]


PLAN_GENERIC_KERNELS, PLAN_GL_NO_COLS256, PLAN_RT_NO_COLS256 = 1, 2, 4
PLAN_GL_REGTILE, PLAN_STARK_NO_LAZY, PLAN_STARK_GENERIC_ON_LAZY = 8, 16, 32
PLAN_NO_HOST_PIN = 64
PLAN_GL_PLAIN_COLS = 128
PLAN_GL_SPLIT_ROWS = 256


def plan_from_env(ring=None):
    """Test / A-B helper: the SR_* switches of earlier rounds, parsed HERE (the library itself reads no environment variable).
    ring: ring id, so that SR_GOLDILOCKS_GENERIC only affects Goldilocks contexts and so on."""
    e = os.environ
    f = 0
    if (ring in (None, 0) and e.get("SR_GOLDILOCKS_GENERIC") == "1") or (ring in (None, 1) and e.get("SR_BABYBEAR_GENERIC") == "1"):
        f |= PLAN_GENERIC_KERNELS
    if e.get("SR_GL_COLS256") == "0":
        f |= PLAN_GL_NO_COLS256
    if e.get("SR_RT_COLS256") == "0":
        f |= PLAN_RT_NO_COLS256
    if e.get("SR_GOLDILOCKS_REGTILE") == "1":
        f |= PLAN_GL_REGTILE
    if e.get("SR_STARK_LAZY") == "0":
        f |= PLAN_STARK_NO_LAZY
    if e.get("SR_STARK_TUNED") == "0":
        f |= PLAN_STARK_GENERIC_ON_LAZY
    if e.get("SR_HOST_PIN") == "0":
        f |= PLAN_NO_HOST_PIN
    if e.get("SR_GL_KEEP_COLS") == "0":
        f |= PLAN_GL_PLAIN_COLS
    if e.get("SR_GL_SPLIT_ROWS") == "1":
        f |= PLAN_GL_SPLIT_ROWS
    p = Plan()
    p.flags = f
    p.log_tile = int(e.get("SR_LOG_TILE", "0") or 0)
    p.stark_whole_max = int(e.get("SR_ST_WHOLE_MAX", "0") or 0)
    p.chunk_polys = int(e.get("SR_CHUNK_POLYS", "0") or 0)
    p.scratch_limit_bytes = int(e.get("SR_SCRATCH_LIMIT_MB", "0") or 0) << 20
    p.host_chunk_mb = int(e.get("SR_HOST_CHUNK_MB", "0") or 0)
    p.lanes = int(e.get("SR_LANES", "0") or 0)
    return p


_lib = None


class BackendMissing(RuntimeError):
    pass


def load():
    """Load the HIP shared object; raises BackendMissing (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendMissing(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH
            )
        # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process
        # cannot both own the device.  Import torch first (when present) so that our DT_NEEDED
        # libamdhip64.so.7 resolves to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - plain C users run without torch
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the header and the library ever diverge
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().sr_last_error_string().decode()
