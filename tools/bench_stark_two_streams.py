import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from stark_rings_amd import CyclotomicRing
k, batch = 12, 1 << 12
w = 4 << k
rings = [CyclotomicRing("stark", k) for _ in range(2)]
a = torch.empty(batch * w, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
rings[0].fill_uniform_dev(a, 1); rings[0].fill_uniform_dev(b, 2)
for parts in (1, 4, 8, 16, 32):
    for ns in (1, 2):
        if parts == 1 and ns > 1: continue
        streams = [torch.cuda.Stream() for _ in range(ns)]
        n = batch // parts
        for r in rings: r.reserve_scratch(n)
        def step():
            for i in range(parts):
                s = streams[i % ns]
                with torch.cuda.stream(s):
                    sl = slice(i * n * w, (i + 1) * n * w)
                    rings[i % ns].mul_dev(a[sl], a[sl], b[sl], stream=s)
        for _ in range(2): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print("stark parts %d streams %d: %.3f ms per batch" % (parts, ns, (time.perf_counter() - t0) / 10 * 1e3))
