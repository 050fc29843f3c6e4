import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import oracle_lib as O
os.environ["SR_GL_COLS256"] = "0"
from stark_rings_amd import CyclotomicRing
k, batch = 16, 3
F = O.GOLDILOCKS
ring = CyclotomicRing("goldilocks", k, device=0)
print("created", flush=True)
a = O.fill_uniform(F, 5, 0, batch << k)
want = O.pow2_fwd(F, a, k, batch, 4)
got = ring.elementwise_crt(a.copy())
print("crt ok", np.array_equal(got, want), flush=True)
print("icrt ok", np.array_equal(ring.elementwise_icrt(got.copy()), a), flush=True)
