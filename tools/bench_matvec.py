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

# Matrix<RqNTT> x Matrix<RqNTT> over the reference's own rings (2 x 2 output blocks, operands through LDS) and a short-and-wide
# mat-vec (rows cut into parts over several workgroups): round 3
for name, n, m, p in (("goldilocks24", 256, 512, 256), ("babybear72", 128, 512, 128), ("frog16", 256, 512, 256)):
    ring = CyclotomicRing(name, 0)
    w = ring.words_per_elem
    a = torch.empty(n * m * w, dtype=torch.int64, device="cuda")
    b = torch.empty(m * p * w, dtype=torch.int64, device="cuda")
    y = torch.empty(n * p * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(a, 3)
    ring.fill_uniform_dev(b, 4)
    ring.matmul_ntt_dev(y, a, b, n, m, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ring.matmul_ntt_dev(y, a, b, n, m, p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    slots = ring.degree // {"goldilocks24": 3, "babybear72": 9, "frog16": 4}[name]
    print("%-12s mat-mat %4d x %-4d x %-4d  %7.3f ms  %6.1f G slot-MACs/s  (operands %.0f MB, streamed %.1f x from L2/LDS at %.0f GB/s)" % (
        name, n, m, p, dt * 1e3, n * m * p * slots / dt / 1e9, (a.numel() + b.numel()) * 8 / 1e6, (n * m * p * 2 * w * 8 / 2) / ((a.numel() + b.numel()) * 8),
        n * m * p * w * 8 / dt / 1e9))
    ring.close()
for name, nrows, ncols in (("goldilocks24", 16, 1 << 18), ("babybear72", 16, 1 << 17), ("frog16", 16, 1 << 18)):
    ring = CyclotomicRing(name, 0)
    w = ring.words_per_elem
    m = torch.empty(nrows * ncols * w, dtype=torch.int64, device="cuda")
    v = torch.empty(ncols * w, dtype=torch.int64, device="cuda")
    y = torch.empty(nrows * w, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(m, 1)
    ring.fill_uniform_dev(v, 2)
    ring.matvec_ntt_dev(y, m, v, nrows, ncols)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ring.matvec_ntt_dev(y, m, v, nrows, ncols)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-12s short-and-wide mat-vec %3d x %-7d  %7.3f ms  %7.1f GB/s" % (name, nrows, ncols, dt * 1e3, (m.numel() + v.numel() + y.numel()) * 8 / dt / 1e9))
    ring.close()

