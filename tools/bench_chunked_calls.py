#!/usr/bin/env python3
"""Experiment: one ring product over the whole batch against the same batch as consecutive calls on chunks (the scratch of a chunk
stays in the 256 MiB Infinity Cache when the chunk is small).  usage: bench_chunked_calls.py [ring] [log2_degree] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "babybear"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 14
plan = _lib.Plan()
plan.chunk_polys = batch  # the library itself runs each call as one set of launches
ring = CyclotomicRing(name, k, plan=plan)
w = ring.words_per_elem
a = torch.empty(batch * w, dtype=torch.int64, device="cuda")
b = torch.empty(batch * w, dtype=torch.int64, device="cuda")
ring.fill_uniform_dev(a, 1)
ring.fill_uniform_dev(b, 2)
for n in (batch, 4096, 2048, 1024, 512, 256, 128, 64):
    if n > batch:
        continue

    def step():
        for e in range(0, batch, n):
            sl = slice(e * w, (e + n) * w)
            ring.mul_dev(a[sl], a[sl], b[sl])

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    print("%s D=2^%d batch %d in calls of %5d elements: %.3f ms per batch" % (name, k, batch, n, (time.perf_counter() - t0) / reps * 1e3))
