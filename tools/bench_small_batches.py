#!/usr/bin/env python3
"""Ring products on batches below and around one lane chunk: the library's default plan against one stream with ONE set of launches and
against two lanes with half-batch chunks.  usage: bench_small_batches.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing
from stark_rings_amd._lib import Plan


def run(name, k, batch, chunk, lanes):
    p = Plan()
    p.chunk_polys = chunk
    p.lanes = lanes
    ring = CyclotomicRing(name, k, plan=p)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    out = torch.empty_like(a)
    ring.fill_uniform_dev(a, 1)
    ring.fill_uniform_dev(b, 2)
    ring.reserve_scratch(batch)
    for _ in range(3):
        ring.mul_dev(out, a, b)
    torch.cuda.synchronize()
    best = 1e9
    reps = 100 if n < (1 << 26) else 20
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            ring.mul_dev(out, a, b)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    ring.close()
    return best


print("| ring | D | batch | default | one stream, one chunk | two lanes, chunks of batch/2 | two lanes, chunks of batch/4 |\n|---|---|---|---|---|---|---|")
cases = [("goldilocks", 16, b) for b in (16, 32, 64, 128, 256, 512, 1024, 2048)] + [("goldilocks", 20, b) for b in (2, 4, 8, 16, 64)] + \
        [("babybear", 16, b) for b in (32, 128, 256, 512, 2048)] + [("goldilocks", 18, b) for b in (8, 32, 128)]
if len(sys.argv) > 3:
    cases = [(sys.argv[i], int(sys.argv[i + 1]), int(sys.argv[i + 2])) for i in range(1, len(sys.argv) - 2, 3)]
for name, k, batch in cases:
    row = [run(name, k, batch, 0, 0), run(name, k, batch, batch, 1)]
    for div in (2, 4):
        row.append(run(name, k, batch, batch // div, 2) if batch // div >= 1 else float("nan"))
    print("| %s | 2^%d | %d | %s |" % (name, k, batch, " | ".join("%.4f" % v for v in row)), flush=True)
