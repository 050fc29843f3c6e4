#!/usr/bin/env python3
"""Throughput of the "next" rows on one GPU, JSON lines: sparse mat-vec, dense mat-mat, balanced gadget
decomposition and recomposition, the ark-serialize wire codec.  All are single streaming passes: the figure of merit is bytes moved / time against HBM."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_rings_amd import CyclotomicRing


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def emit(**kw):
    print(json.dumps(kw))


for name, k, nrows, ncols, per_row in (("goldilocks", 16, 64, 1024, 64), ("babybear", 16, 64, 1024, 64), ("stark", 12, 64, 1024, 64)):
    ring = CyclotomicRing(name, k)
    w = ring.words_per_elem
    nnz = nrows * per_row
    vals = torch.empty(nnz * w, dtype=torch.int64, device="cuda")
    v = torch.empty(ncols * w, dtype=torch.int64, device="cuda")
    y = torch.empty(nrows * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(vals, 1)
    ring.fill_uniform_dev(v, 2)
    rng = np.random.default_rng(1)
    cols = torch.from_numpy(rng.integers(0, ncols, nnz).astype(np.int32)).cuda()
    ptr = torch.arange(0, nnz + 1, per_row, dtype=torch.int64, device="cuda")
    dt = timed(lambda: ring.spmv_ntt_dev(y, vals, cols, ptr, v, nrows, ncols))
    assert ring.spmv_bad_index_count() == 0
    gb = (vals.numel() + y.numel()) * 8 / 1e9   # the stored entries are streamed once; v (ncols elements) is re-read from cache
    emit(op="spmv_ntt", ring=name, log2_degree=k, nrows=nrows, ncols=ncols, nnz=nnz, ms=dt * 1e3, gbytes_per_s=gb / dt,
         slot_macs_per_s=nnz * ring.degree / dt)
    n, m, p = 32, 64, 32
    a = vals[:n * m * w]
    b = torch.empty(m * p * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(b, 3)
    out = torch.empty(n * p * w, dtype=torch.int64, device="cuda")
    dt = timed(lambda: ring.matmul_ntt_dev(out, a, b, n, m, p))
    emit(op="matmul_ntt", ring=name, log2_degree=k, n=n, m=m, p=p, ms=dt * 1e3,
         gbytes_per_s_compulsory=(a.numel() + b.numel() + out.numel()) * 8 / 1e9 / dt, slot_macs_per_s=n * m * p * ring.degree / dt)
    for basis, pad in ((2, None), (1 << 4, None), (1 << 16, None), (10, None)):
        pbits = {"goldilocks": 64, "babybear": 31, "stark": 252}[name]
        padn = 1
        while (basis // 2) * (basis ** padn - 1) // (basis - 1) < (1 << (pbits - 1)):
            padn += 1
        padn += 1
        batch = max(1, (1 << 27) // (ring.degree * ring.limbs * padn))     # ~1 GiB of digits
        src = torch.empty(batch * w, dtype=torch.int64, device="cuda")
        ring.fill_uniform_dev(src, 5)
        dig = torch.empty(batch * padn * w, dtype=torch.int64, device="cuda")
        dt = timed(lambda: ring.gadget_decompose_dev(dig, src, basis, padn))
        assert ring.decompose_overflow_count() == 0
        emit(op="gadget_decompose", ring=name, log2_degree=k, basis=basis, padding_size=padn, batch=batch, ms=dt * 1e3,
             gbytes_per_s=(src.numel() + dig.numel()) * 8 / 1e9 / dt, coefficients_per_s=batch * ring.degree / dt)
        back = torch.empty_like(src)
        dt = timed(lambda: ring.gadget_recompose_dev(back, dig, basis, padn))
        assert torch.equal(back, src)
        emit(op="gadget_recompose", ring=name, log2_degree=k, basis=basis, padding_size=padn, batch=batch, ms=dt * 1e3,
             gbytes_per_s=(src.numel() + dig.numel()) * 8 / 1e9 / dt, coefficients_per_s=batch * ring.degree / dt)
        del src, dig, back
    # ark-serialize wire codec (8f #3): memory image -> wire bytes and back, one streaming pass each
    batch = (1 << 31) // (ring.degree * ring.limbs * 8)           # 2 GiB of ring elements
    src = torch.empty(batch * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(src, 9)
    wire = torch.empty(batch * ring.degree * ring.wire_coeff_bytes, dtype=torch.uint8, device="cuda")
    dt = timed(lambda: ring.serialize_dev(wire, src))
    emit(op="serialize", ring=name, log2_degree=k, batch=batch, wire_coeff_bytes=ring.wire_coeff_bytes, ms=dt * 1e3,
         gbytes_per_s=(src.numel() * 8 + wire.numel()) / 1e9 / dt, coefficients_per_s=batch * ring.degree / dt)
    back = torch.empty_like(src)
    dt = timed(lambda: ring.deserialize_dev(back, wire))
    assert ring.wire_invalid_count() == 0 and torch.equal(back, src)
    emit(op="deserialize", ring=name, log2_degree=k, batch=batch, wire_coeff_bytes=ring.wire_coeff_bytes, ms=dt * 1e3,
         gbytes_per_s=(src.numel() * 8 + wire.numel()) / 1e9 / dt, coefficients_per_s=batch * ring.degree / dt)
    del src, wire, back
    ring.close()
