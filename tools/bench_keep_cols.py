"""A/B of the lane plans' column pass (round 4): cols256_keep_kernel (a workgroup owns a column chunk and keeps its twist factors,
three workgroups per CU) against the plain cols256_kernel (SR_PLAN_GL_PLAIN_COLS), Goldilocks 2^16 <= D <= 2^19, two lanes, default
chunks, alternating contexts on one box.  usage: python tools/bench_keep_cols.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stark_rings_amd import CyclotomicRing  # noqa: E402
from stark_rings_amd._lib import Plan  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for k, batch in ((16, 8192), (17, 4096), (18, 2048), (19, 1024)):
        d = 1 << k
        ta = torch.empty(batch * d, dtype=torch.int64, device="cuda")
        tb = torch.empty_like(ta)
        out = torch.empty_like(ta)
        rings = {}
        for name, flags in (("keep", 0), ("plain", 128)):
            p = Plan()
            p.flags, p.lanes = flags, 2
            rings[name] = CyclotomicRing("goldilocks", k, plan=p)
            rings[name].reserve_scratch(batch)
        rings["keep"].fill_uniform_dev(ta, 1, 0)
        rings["keep"].fill_uniform_dev(tb, 2, 0)
        res = {"keep": [], "plain": []}
        for r in range(reps):
            for name in ("keep", "plain"):
                ring = rings[name]
                for _ in range(2):
                    ring.mul_dev(out, ta, tb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    ring.mul_dev(out, ta, tb)
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / 8 * 1e3)
        print("D=2^%d batch %d: keep %s ms, plain %s ms" % (k, batch, " ".join("%.3f" % v for v in res["keep"]),
                                                           " ".join("%.3f" % v for v in res["plain"])), flush=True)
        for ring in rings.values():
            ring.close()
        del ta, tb, out


if __name__ == "__main__":
    main()
