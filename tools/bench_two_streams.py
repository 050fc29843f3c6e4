#!/usr/bin/env python3
"""Experiment: does running the two halves of the config-2 batch on two streams (two contexts, so two operand scratches) overlap the
memory-bound column passes of one half with the VALU-bound rows kernel of the other?  Prints ms per full batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

k, batch = 16, 1 << 14
d = 1 << k
rings = [CyclotomicRing("goldilocks", k) for _ in range(4)]
a = torch.empty(batch * d, dtype=torch.int64, device="cuda")
b = torch.empty(batch * d, dtype=torch.int64, device="cuda")
rings[0].fill_uniform_dev(a, 1)
rings[0].fill_uniform_dev(b, 2)
for parts in (1, 4, 16, 32, 64, 128):
    for nstreams in (1, 2, 3, 4):
        if parts == 1 and nstreams > 1:
            continue
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        n = batch // parts
        for r in rings:
            r.reserve_scratch(n)

        def step():
            for i in range(parts):
                s = streams[i % nstreams]
                with torch.cuda.stream(s):
                    sl = slice(i * n * d, (i + 1) * n * d)
                    rings[i % nstreams].mul_dev(a[sl], a[sl], b[sl], stream=s)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        print("parts %d streams %d: %.3f ms per batch" % (parts, nstreams, (time.perf_counter() - t0) / reps * 1e3))
