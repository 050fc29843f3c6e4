#!/usr/bin/env python3
"""Stark ring product and forward transform over 2^24 coefficients for D = 2^9 .. 2^13 (SR_ST_WHOLE_MAX selects one tile per
ring element or strided passes + 512-coefficient tiles): the measurements behind st::whole_max()."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from stark_rings_amd import CyclotomicRing
for k in (9, 10, 11, 12, 13):
    ring = CyclotomicRing("stark", k)
    batch = (1 << 24) >> k
    w = ring.words_per_elem
    a = torch.empty(batch * w, dtype=torch.int64, device="cuda"); b = torch.empty_like(a); o = torch.empty_like(a)
    ring.fill_uniform_dev(a, 1); ring.fill_uniform_dev(b, 2)
    def run():
        ring.mul_dev(o, a, b)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): ring.elementwise_crt_dev(a)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / 5
    print("k=%d whole_max=%s mul %.3f ms  fwd %.3f ms" % (k, os.environ.get("SR_ST_WHOLE_MAX", "12"), dt * 1e3, df * 1e3), flush=True)
    ring.close()
