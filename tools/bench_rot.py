import sys, time, torch
sys.path.insert(0, "/root/repo")
from stark_rings_amd import CyclotomicRing
for name, k, batch in (("goldilocks", 16, 4096), ("babybear", 16, 4096), ("stark", 12, 4096), ("goldilocks24", 0, 1 << 22)):
    ring = CyclotomicRing(name, k)
    n = batch * ring.words_per_elem
    a = torch.empty(n, dtype=torch.int64, device="cuda"); o = torch.empty_like(a)
    ring.fill_uniform_dev(a, 1)
    ring.rot_dev(o, a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): ring.rot_dev(o, a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("rot %-12s %7.3f ms %7.1f GB/s" % (name, dt * 1e3, 2 * n * 8 / dt / 1e9))
    ring.close()
