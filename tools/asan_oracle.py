#!/usr/bin/env python3
"""Runs the C oracle under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool):
  gcc -O1 -g -fsanitize=address,undefined -fPIC -std=gnu11 -shared -o build_tmp/libsr_oracle_asan.so oracle/sr_oracle.c -lpthread
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_oracle.py
Exercises every exported oracle function on small inputs (all fields, ragged reduce lengths, empty inputs)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_lib as O
O._lib = None
# load the sanitised build in place of the normal one
lib = ctypes.CDLL(os.path.join(ROOT, "build_tmp", "libsr_oracle_asan.so"))
import pyref as P
def bind(name, res, args):
    f = getattr(lib, name); f.restype = res; f.argtypes = args; return f
u64p = ctypes.POINTER(ctypes.c_uint64); sz = ctypes.c_size_t; i = ctypes.c_int
ptr = lambda a: a.ctypes.data_as(u64p)
fill = bind("sro_fill_uniform", None, [i, ctypes.c_uint64, ctypes.c_uint64, sz, u64p])
fwd = bind("sro_pow2_fwd_batch", i, [i, u64p, i, sz, i])
mul = bind("sro_pow2_ring_mul_batch", i, [i, u64p, u64p, u64p, i, sz, i])
dec = bind("sro_decompose_balanced", i, [i, u64p, sz, sz, ctypes.c_uint64, sz, u64p])
rec = bind("sro_recompose", i, [i, u64p, sz, sz, ctypes.c_uint64, sz, u64p])
for field, limbs in ((0,1),(1,1),(2,4)):
    for k in (0, 1, 4, 9):
        d = 1 << k; batch = 5
        a = np.zeros(batch*d*limbs, dtype=np.uint64); b = np.zeros_like(a); out = np.zeros_like(a)
        fill(field, 1, 0, batch*d, ptr(a)); fill(field, 2, 0, batch*d, ptr(b))
        assert mul(field, ptr(out), ptr(a), ptr(b), k, batch, 3) == 0
        c = a.copy(); assert fwd(field, ptr(c), k, batch, 2) == 0
        dg = np.zeros(batch*d*limbs*70 if field != 2 else batch*d*limbs*260, dtype=np.uint64)
        kk = 70 if field != 2 else 260
        assert dec(field, ptr(a), d, batch, 2, kk, ptr(dg)) == 0
        back = np.zeros_like(a); rec(field, ptr(dg), d, batch, 2, kk, ptr(back)); assert np.array_equal(back, a)
for name, w in (("g24", 24), ("bb72", 72), ("frog16", 16)):
    field = {"g24":0, "bb72":1, "frog16":3}[name]
    a = np.zeros(w, dtype=np.uint64); fill(field, 3, 0, w, ptr(a)); b = a.copy()
    for fn in ("crt", "icrt", "homogenize", "dehomogenize"):
        bind("sro_%s_%s" % (name, fn), None, [u64p])(ptr(a))
    bind("sro_%s_ntt_mul" % name, None, [u64p, u64p])(ptr(a), ptr(b))
    o = np.zeros(w, dtype=np.uint64)
    for n in (0, 1, w, 2*w - 1, 2*w):
        src = np.zeros(max(n,1), dtype=np.uint64)
        bind("sro_%s_reduce" % name, None, [u64p, sz, u64p])(ptr(src), n, ptr(o))
ser = bind("sro_serialize", None, [i, u64p, sz, ctypes.c_void_p])
des = bind("sro_deserialize", sz, [i, ctypes.c_void_p, sz, u64p])
wb = bind("sro_wire_bytes", sz, [i])
for field, limbs in ((0, 1), (1, 1), (2, 4), (3, 1)):
    n = 37
    a = np.zeros(n * limbs, dtype=np.uint64); fill(field, 5, 0, n, ptr(a))
    wire = np.zeros(n * wb(field), dtype=np.uint8)
    ser(field, ptr(a), n, wire.ctypes.data_as(ctypes.c_void_p))
    back = np.zeros_like(a)
    assert des(field, wire.ctypes.data_as(ctypes.c_void_p), n, ptr(back)) == 0 and np.array_equal(back, a)
    wire[:] = 0xFF
    assert des(field, wire.ctypes.data_as(ctypes.c_void_p), n, ptr(back)) == n
print("asan/ubsan run finished")
