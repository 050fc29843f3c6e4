#!/usr/bin/env python3
"""Stand-alone transform throughput (elementwise_crt / elementwise_icrt / slot product) at BASELINE sizes, data resident in HBM."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

for name, k, batch in (("goldilocks", 16, 1 << 14), ("babybear", 16, 1 << 14), ("stark", 12, 1 << 12), ("goldilocks", 20, 1 << 10)):
    ring = CyclotomicRing(name, k)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(a, 1)
    ring.fill_uniform_dev(b, 2)
    one = b[:ring.words_per_elem].clone()
    for op, fn, streams in (("crt", lambda: ring.elementwise_crt_dev(a), 2), ("icrt", lambda: ring.elementwise_icrt_dev(a), 2),
                            ("slot product", lambda: ring.ntt_mul_dev(a, b), 3),
                            ("x one element", lambda: ring.mul_elem_dev(a, one), 2)):   # Matrix<R> *= &R: read + write the batch
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-10s D=2^%-2d batch %-6d %-13s %9.1f k elements/s  %7.1f GB/s (algorithmic)  %7.3f ms" % (
            name, k, batch, op, batch / dt / 1e3, streams * n * 8 / dt / 1e9, dt * 1e3))
    ring.close()
