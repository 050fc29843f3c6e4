#!/usr/bin/env python3
"""Matrix<RqNTT> * vec throughput (sr_matvec_ntt_dev): HBM-bound fused multiply-accumulate over M."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

# the last three are the reference's own RqNTT types (Fq3 / Fq9 / Fq4 slots, csrc/small_linalg.hpp): one slot per lane
for name, k, nrows, ncols in (("goldilocks", 16, 64, 256), ("goldilocks", 10, 1024, 4096), ("babybear", 16, 64, 256), ("stark", 12, 64, 512),
                              ("goldilocks24", 0, 2048, 4096), ("babybear72", 0, 1024, 2048), ("frog16", 0, 2048, 4096)):
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
    slots = ring.degree // {"goldilocks24": 3, "babybear72": 9, "frog16": 4}.get(name, 1)
    print("%-12s D=%-5d %5d x %-5d  %7.3f ms  %7.1f GB/s (M read once + v + y)  %6.1f G slot-MACs/s" % (
        name, ring.degree, nrows, ncols, dt * 1e3, gb / dt, nrows * ncols * slots / dt / 1e9))
    ring.close()
