"""A/B of the lane plans' column pass (round 4): cols256_keep_kernel (a workgroup owns a column chunk and keeps its twist factors,
three workgroups per CU) against the plain cols256_kernel (SR_PLAN_GL_PLAIN_COLS), Goldilocks D = 2^16, two lanes, default chunks.
Every measurement runs in a process of its own (the hardware queue a stream lands on depends on the streams the process created
before: two contexts in one process are not a fair A/B), alternating.  usage: python tools/bench_keep_cols.py [reps]
(Degrees 2^17 .. 2^19 were measured with the kernel forced on through a local edit of gl_launch_cols256_lane: DESIGN.md 6.)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(k, batch, flags):
    import torch

    from stark_rings_amd import CyclotomicRing
    from stark_rings_amd._lib import Plan

    d = 1 << k
    ta = torch.empty(batch * d, dtype=torch.int64, device="cuda")
    tb = torch.empty_like(ta)
    out = torch.empty_like(ta)
    p = Plan()
    p.flags, p.lanes = flags, 2
    ring = CyclotomicRing("goldilocks", k, plan=p)
    ring.reserve_scratch(batch)
    ring.fill_uniform_dev(ta, 1, 0)
    ring.fill_uniform_dev(tb, 2, 0)
    for _ in range(3):
        ring.mul_dev(out, ta, tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ring.mul_dev(out, ta, tb)
    torch.cuda.synchronize()
    print("%.4f" % ((time.perf_counter() - t0) / 10 * 1e3))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for k, batch in ((16, 16384),):
        res = {"keep": [], "plain": []}
        for _ in range(reps):
            for name, flags in (("keep", 0), ("plain", 128)):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(k), str(batch), str(flags)], stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=300)
                res[name].append(r.stdout.decode().strip().split("\n")[-1] if r.returncode == 0 else "failed")
        print("D=2^%d batch %d, two lanes: keep %s ms, plain %s ms" % (k, batch, " ".join(res["keep"]), " ".join(res["plain"])), flush=True)


if __name__ == "__main__":
    main()
