#!/usr/bin/env python3
"""Matrix<RqNTT> * vec throughput (sr_matvec_ntt_dev): HBM-bound fused multiply-accumulate over M."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

for name, k, nrows, ncols in (("goldilocks", 16, 64, 256), ("goldilocks", 10, 1024, 4096), ("babybear", 16, 64, 256), ("stark", 12, 64, 512)):
    ring = CyclotomicRing(name, k)
    w = ring.words_per_elem
    m = torch.empty(nrows * ncols * w, dtype=torch.int64, device="cuda")
    v = torch.empty(ncols * w, dtype=torch.int64, device="cuda")
    y = torch.empty(nrows * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(m, 1)
    ring.fill_uniform_dev(v, 2)
    ring.matvec_ntt_dev(y, m, v, nrows, ncols)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ring.matvec_ntt_dev(y, m, v, nrows, ncols)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gb = (m.numel() + v.numel() + y.numel()) * 8 / 1e9
    print("%-10s D=2^%-2d %5d x %-5d  %7.3f ms  %7.1f GB/s (M read once + v + y)  %6.1f G slot-MACs/s" % (
        name, k, nrows, ncols, dt * 1e3, gb / dt, nrows * ncols * ring.degree / dt / 1e9))
    ring.close()
