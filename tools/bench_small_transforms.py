#!/usr/bin/env python3
"""Stand-alone transforms on small batches: library default against the forced plans (looking for cliffs like the one the products had)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing
from stark_rings_amd._lib import Plan


def run(name, k, batch, lanes, which):
    p = Plan()
    p.lanes = lanes
    ring = CyclotomicRing(name, k, plan=p)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(a, 1)
    ring.reserve_scratch(batch)
    fn = ring.elementwise_crt_dev if which == "crt" else ring.elementwise_icrt_dev
    for _ in range(3):
        fn(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            fn(a)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    ring.close()
    return best


print("| ring | D | batch | op | default | lanes = 1 | lanes = 2 |\n|---|---|---|---|---|---|---|")
for name, k, batches in (("goldilocks", 16, (32, 64, 128, 256, 512, 1024)), ("goldilocks", 20, (4, 16, 32, 64)), ("babybear", 16, (64, 256, 512, 1024, 2048)),
                         ("stark", 12, (8, 64, 512))):
    for batch in batches:
        for which in ("crt", "icrt"):
            print("| %s | 2^%d | %d | %s | %.4f | %.4f | %.4f |" % (name, k, batch, which, run(name, k, batch, 0, which), run(name, k, batch, 1, which),
                                                              run(name, k, batch, 2, which)), flush=True)
