#!/usr/bin/env python3
"""The device-pointer ring product captured into a HIP graph by the caller and replayed, against eager calls: where launches bound
the call (small batches) the replay wins; a full BASELINE batch is unchanged.  usage: bench_graph_replay.py [ring log2_degree batch ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stark_rings_amd import CyclotomicRing

cases = [("goldilocks", 16, 16), ("goldilocks", 16, 128), ("goldilocks", 16, 256), ("goldilocks", 16, 1024), ("goldilocks", 16, 4096),
         ("goldilocks", 16, 16384), ("goldilocks", 20, 64), ("goldilocks", 20, 1024), ("goldilocks", 10, 1), ("goldilocks", 10, 1024),
         ("babybear", 16, 1024), ("babybear", 16, 16384), ("stark", 12, 64), ("stark", 12, 4096)]
if len(sys.argv) > 3:
    cases = [(sys.argv[i], int(sys.argv[i + 1]), int(sys.argv[i + 2])) for i in range(1, len(sys.argv) - 2, 3)]
print("| ring | D | batch | eager ms | graph replay ms | replay / eager |\n|---|---|---|---|---|---|")
for name, k, batch in cases:
    ring = CyclotomicRing(name, k)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(a, 11)
    ring.fill_uniform_dev(b, 12)
    ring.reserve_scratch(batch)
    want = torch.empty_like(a)
    ring.mul_dev(want, a, b)
    out = torch.zeros_like(a)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ring.mul_dev(out, a, b, stream=s)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ring.mul_dev(out, a, b, stream=torch.cuda.current_stream())
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want), (name, k, batch)
    res = {}
    reps = 200 if n < (1 << 24) else (20 if n < (1 << 28) else 6)
    for label, fn in (("eager", lambda: ring.mul_dev(out, a, b)), ("graph", g.replay), ("eager", lambda: ring.mul_dev(out, a, b)), ("graph", g.replay)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        res[label] = min(res.get(label, ms), ms)
    print("| %s | 2^%d | %d | %.4f | %.4f | %.2f |" % (name, k, batch, res["eager"], res["graph"], res["graph"] / res["eager"]), flush=True)
    del g
    ring.close()
