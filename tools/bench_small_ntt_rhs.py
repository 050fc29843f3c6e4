import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from stark_rings_amd import CyclotomicRing
from stark_rings_amd._lib import Plan
def run(name, k, batch, lanes):
    p = Plan(); p.lanes = lanes
    ring = CyclotomicRing(name, k, plan=p)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a); out = torch.empty_like(a)
    ring.fill_uniform_dev(a, 1); ring.fill_uniform_dev(b, 2)
    ring.reserve_scratch(batch)
    for _ in range(3): ring.mul_ntt_rhs_dev(out, a, b)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100): ring.mul_ntt_rhs_dev(out, a, b)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    ring.close()
    return best
print("| ring | D | batch | default | lanes=1 | lanes=2 |")
for name, k, bs in (("goldilocks", 16, (128, 256, 384, 512, 1024)), ("goldilocks", 20, (8, 16, 32, 64))):
    for b in bs:
        print("| %s | 2^%d | %d | %.4f | %.4f | %.4f |" % (name, k, b, run(name, k, b, 0), run(name, k, b, 1), run(name, k, b, 2)), flush=True)
