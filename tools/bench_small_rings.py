#!/usr/bin/env python3
"""Throughput of the reference-native small rings (Goldilocks-24, BabyBear-72, Frog-16, Stark-16) on one GPU: CRT, ICRT, slot product,
fused ring product over a large batch resident in HBM.  Prints elements/s and effective HBM GB/s (algorithmic bytes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

for name, D in (("goldilocks24", 24), ("babybear72", 72), ("frog16", 16), ("stark16", 16)):
    ring = CyclotomicRing("stark", 4) if name == "stark16" else CyclotomicRing(name)   # stark_prime/mod.rs:34-68: X^16 + 1
    batch = 1 << 22
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n, dtype=torch.int64, device="cuda")
    out = torch.empty_like(a)
    ring.fill_uniform_dev(a, 1)
    ring.fill_uniform_dev(b, 2)
    ops = {
        "crt": (lambda: ring.elementwise_crt_dev(a), 2),
        "icrt": (lambda: ring.elementwise_icrt_dev(a), 2),
        "slot product": (lambda: ring.ntt_mul_dev(a, b), 3),
        "ring product": (lambda: ring.mul_dev(out, a, b), 3),
    }
    for op, (fn, streams) in ops.items():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        gb = streams * n * 8 / 1e9
        print("%-13s %-13s %8.1f M elements/s  %7.1f GB/s (algorithmic)  %6.3f ms" % (name, op, batch / dt / 1e6, gb / dt, dt * 1e3))
    ring.close()
