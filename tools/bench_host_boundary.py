#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (what a drop-in Rust caller holding Vec<RqPoly> in host memory sees):
ring products per second through sr_ring_mul_batch with pageable numpy buffers.  Never the bench `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import oracle_lib as O
from stark_rings_amd import CyclotomicRing

from stark_rings_amd._lib import PLAN_NO_HOST_PIN, Plan

# each case twice: the caller's pageable buffers registered for the call (hipHostRegister, the default since round 3) and left pageable
for name, k, batch, pin in [(n_, k_, b_, p_) for (n_, k_, b_) in (("goldilocks", 16, 2048), ("goldilocks", 16, 8192), ("goldilocks", 10, 524288),
                                                                  ("babybear", 16, 8192), ("stark", 12, 32768)) for p_ in (True, False)]:
    F = O.FIELD_ID[name]
    plan = Plan()
    if not pin:
        plan.flags = PLAN_NO_HOST_PIN
    ring = CyclotomicRing(name, k, plan=plan)
    a = O.fill_uniform(F, 1, 0, batch << k)
    b = O.fill_uniform(F, 2, 0, batch << k)
    out = np.empty_like(a)
    ring.mul(a, b, out)  # warm-up: staging buffers get allocated
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ring.mul(a, b, out)
    dt = (time.perf_counter() - t0) / reps
    gb = 3 * a.nbytes / 1e9
    print("%-10s D=2^%-2d batch %-6d  %-22s %8.1f ms  %9.0f ring-muls/s  %5.1f GB/s over PCIe (2 in + 1 out)" % (
        name, k, batch, "caller pages pinned" if pin else "pageable (round 2)", dt * 1e3, batch / dt, gb / dt))
    ring.close()
