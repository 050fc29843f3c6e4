import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from stark_rings_amd import CyclotomicRing
ring = CyclotomicRing("goldilocks", 16)
batch = 16384
n = batch * ring.words_per_elem
a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
ring.fill_uniform_dev(a, 1); ring.fill_uniform_dev(b, 2)
ring.reserve_scratch(batch)
for _ in range(3): ring.mul_dev(a, a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(int(os.environ.get("SR_STEPS", "10"))): ring.mul_dev(a, a, b)
torch.cuda.synchronize()
print("SR_LANES=%s: %.3f ms per batch" % (os.environ.get("SR_LANES", "default"), (time.perf_counter() - t0) * 1e3 / int(os.environ.get("SR_STEPS", "10"))))
