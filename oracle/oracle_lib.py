"""ORACLE (test infrastructure): ctypes loader for oracle/libsr_oracle.so + numpy helpers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Buffers are numpy uint64 arrays in the reference's in-memory layout (element-major,
coefficient-minor, N little-endian u64 limbs per coefficient, Montgomery form).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDILOCKS, BABYBEAR, STARK = 0, 1, 2
FROG = 3
FIELD_ID = {"goldilocks": GOLDILOCKS, "babybear": BABYBEAR, "stark": STARK, "frog": FROG}
LIMBS = {GOLDILOCKS: 1, BABYBEAR: 1, STARK: 4, FROG: 1}

_lib = None
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    so = os.path.join(HERE, "libsr_oracle.so")
    src = os.path.join(HERE, "sr_oracle.c")
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)
    ):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return so


PORTABLE_FLAGS = "-O3 -march=x86-64-v2"
_flags = PORTABLE_FLAGS


def use_native_build():
    """bench.py's TIMED cpu_baseline only (`-O3 -march=native`): compile the restatement on THIS machine for THIS
    machine's cores and load that build from now on; returns the flags in force (the portable ones when no compiler is here)."""
    global _lib, _flags
    so = os.path.join(HERE, "libsr_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = None
        lib(so)
        _flags = "-O3 -march=native"
    except Exception:  # no gcc / make on this machine: the portable build stays
        _lib = None
        lib()
    return _flags


def lib(path=None):
    global _lib
    if _lib is None:
        L = ctypes.CDLL(path or build())
        sz, i, u64 = ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        sigs = {
            "sro_limbs": (i, [i]),
            "sro_to_mont": (None, [i, _u64p, _u64p, sz]),
            "sro_from_mont": (None, [i, _u64p, _u64p, sz]),
            "sro_pow2_fwd": (i, [i, _u64p, i]),
            "sro_pow2_inv": (i, [i, _u64p, i]),
            "sro_pow2_pointwise": (i, [i, _u64p, _u64p, sz]),
            "sro_pow2_reduce": (i, [i, _u64p, sz, _u64p, i]),
            "sro_schoolbook": (i, [i, _u64p, _u64p, sz, _u64p]),
            "sro_rot": (None, [i, _u64p, sz, i, _u64p]),
            "sro_wire_bytes": (sz, [i]),
            "sro_serialize": (None, [i, _u64p, sz, ctypes.c_void_p]),
            "sro_deserialize": (sz, [i, ctypes.c_void_p, sz, _u64p]),
            "sro_decompose_balanced": (i, [i, _u64p, sz, sz, ctypes.c_uint64, sz, _u64p]),
            "sro_recompose": (i, [i, _u64p, sz, sz, ctypes.c_uint64, sz, _u64p]),
            "sro_pow2_ring_mul": (i, [i, _u64p, _u64p, _u64p, i]),
            "sro_pow2_fwd_batch": (i, [i, _u64p, i, sz, i]),
            "sro_pow2_inv_batch": (i, [i, _u64p, i, sz, i]),
            "sro_pow2_ring_mul_batch": (i, [i, _u64p, _u64p, _u64p, i, sz, i]),
            "sro_g24_crt": (None, [_u64p]),
            "sro_g24_icrt": (None, [_u64p]),
            "sro_g24_homogenize": (None, [_u64p]),
            "sro_g24_dehomogenize": (None, [_u64p]),
            "sro_g24_ntt_mul": (None, [_u64p, _u64p]),
            "sro_g24_reduce": (None, [_u64p, sz, _u64p]),
            "sro_bb72_crt": (None, [_u64p]),
            "sro_bb72_icrt": (None, [_u64p]),
            "sro_bb72_homogenize": (None, [_u64p]),
            "sro_bb72_dehomogenize": (None, [_u64p]),
            "sro_bb72_ntt_mul": (None, [_u64p, _u64p]),
            "sro_bb72_reduce": (None, [_u64p, sz, _u64p]),
            "sro_frog16_crt": (None, [_u64p]),
            "sro_frog16_icrt": (None, [_u64p]),
            "sro_frog16_homogenize": (None, [_u64p]),
            "sro_frog16_dehomogenize": (None, [_u64p]),
            "sro_frog16_ntt_mul": (None, [_u64p, _u64p]),
            "sro_frog16_reduce": (None, [_u64p, sz, _u64p]),
            "sro_fill_uniform": (None, [i, u64, u64, sz, _u64p]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


# ---------------------------------------------------------------- int <-> limb arrays
def ints_to_limbs(vals, limbs):
    out = np.zeros(len(vals) * limbs, dtype=np.uint64)
    mask = (1 << 64) - 1
    for k, v in enumerate(vals):
        for l in range(limbs):
            out[k * limbs + l] = (int(v) >> (64 * l)) & mask
    return out


def limbs_to_ints(arr, limbs):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, limbs)
    return [sum(int(row[l]) << (64 * l) for l in range(limbs)) for row in arr]


def to_mont(field, std_ints):
    L = LIMBS[field]
    a = ints_to_limbs(std_ints, L)
    out = np.empty_like(a)
    lib().sro_to_mont(field, ptr(a), ptr(out), len(std_ints))
    return out


def from_mont(field, arr):
    L = LIMBS[field]
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    out = np.empty_like(arr)
    lib().sro_from_mont(field, ptr(arr), ptr(out), arr.size // L)
    return limbs_to_ints(out, L)


def fill_uniform(field, seed, first_coeff, n_coeffs):
    out = np.empty(n_coeffs * LIMBS[field], dtype=np.uint64)
    lib().sro_fill_uniform(field, seed, first_coeff, n_coeffs, ptr(out))
    return out


# ---------------------------------------------------------------- pow2 ring helpers (copies in, returns new arrays)
def pow2_fwd(field, a, log2d, batch=1, threads=1):
    a = np.array(a, dtype=np.uint64, copy=True)
    assert lib().sro_pow2_fwd_batch(field, ptr(a), log2d, batch, threads) == 0
    return a


def pow2_inv(field, a, log2d, batch=1, threads=1):
    a = np.array(a, dtype=np.uint64, copy=True)
    assert lib().sro_pow2_inv_batch(field, ptr(a), log2d, batch, threads) == 0
    return a


def pow2_pointwise(field, a, b):
    a = np.array(a, dtype=np.uint64, copy=True)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    assert lib().sro_pow2_pointwise(field, ptr(a), ptr(b), a.size // LIMBS[field]) == 0
    return a


def pow2_ring_mul(field, a, b, log2d, batch=1, threads=1):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    assert lib().sro_pow2_ring_mul_batch(field, ptr(out), ptr(a), ptr(b), log2d, batch, threads) == 0
    return out


def pow2_reduce(field, c, in_len, log2d):
    c = np.ascontiguousarray(c, dtype=np.uint64)
    out = np.empty((1 << log2d) * LIMBS[field], dtype=np.uint64)
    assert lib().sro_pow2_reduce(field, ptr(c), in_len, ptr(out), log2d) == 0
    return out


def schoolbook(field, a, b, d):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty((2 * d - 1) * LIMBS[field], dtype=np.uint64)
    assert lib().sro_schoolbook(field, ptr(a), ptr(b), d, ptr(out)) == 0
    return out


def rot(field, a, d, trinomial=False):
    """Cyclotomic::rot of every ring element of the batch"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    w = d * LIMBS[field]
    for e in range(a.size // w):
        src = np.ascontiguousarray(a[e * w:(e + 1) * w])
        dst = np.empty(w, dtype=np.uint64)
        lib().sro_rot(field, ptr(src), d, 1 if trinomial else 0, ptr(dst))
        out[e * w:(e + 1) * w] = dst
    return out


def decompose_balanced(field, a, d, batch, b, k):
    """returns (digits, overflow): digits = batch * k ring elements, digit j of element e at index e * k + j"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty(batch * k * d * LIMBS[field], dtype=np.uint64)
    rc = lib().sro_decompose_balanced(field, ptr(a), d, batch, b, k, ptr(out))
    if rc < 0:
        raise ValueError("bad decomposition basis")
    return out, bool(rc)


def recompose(field, digits, d, batch_out, b, k):
    digits = np.ascontiguousarray(digits, dtype=np.uint64)
    out = np.empty(batch_out * d * LIMBS[field], dtype=np.uint64)
    assert lib().sro_recompose(field, ptr(digits), d, batch_out, b, k, ptr(out)) == 0
    return out


def wire_bytes(field):
    return int(lib().sro_wire_bytes(field))


def serialize(field, a):
    """ark-serialize bytes of the flat coefficients (uint8 array)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.size // LIMBS[field]
    out = np.zeros(max(n * wire_bytes(field), 1), dtype=np.uint8)
    lib().sro_serialize(field, ptr(a), n, out.ctypes.data_as(ctypes.c_void_p))
    return out[:n * wire_bytes(field)]


def deserialize(field, wire):
    """-> (coefficients in Montgomery form, number of coefficients >= p)"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    n = wire.size // wire_bytes(field)
    out = np.zeros(max(n * LIMBS[field], 1), dtype=np.uint64)
    src = wire if wire.size else np.zeros(8, dtype=np.uint8)
    bad = lib().sro_deserialize(field, src.ctypes.data_as(ctypes.c_void_p), n, ptr(out))
    return out[:n * LIMBS[field]], int(bad)


def small(fn_name, a, b=None):
    """Call a small-ring in-place function on a copy (batched over leading dim)."""
    a = np.array(a, dtype=np.uint64, copy=True)
    fn = getattr(lib(), fn_name)
    width = 24 if "g24" in fn_name else (16 if "frog16" in fn_name else 72)
    flat = a.reshape(-1, width)
    if b is not None:
        bf = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, width)
    for r in range(flat.shape[0]):
        row = np.ascontiguousarray(flat[r])
        if b is None:
            fn(ptr(row))
        else:
            fn(ptr(row), ptr(np.ascontiguousarray(bf[r])))
        flat[r] = row
    return a
