import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import oracle_lib as O
os.environ["SR_GL_COLS256"] = "0"
from stark_rings_amd import CyclotomicRing
k, batch = 16, 3
F = O.GOLDILOCKS
ring = CyclotomicRing("goldilocks", k, device=0)
p, n = ring.twiddle_block()
a = torch.from_numpy(O.fill_uniform(F, 5, 0, batch << k).view(np.int64)).cuda()
d = 1 << k
print("tables %x size %x  tw %x itw %x twist_f %x twist_ip %x" % (p, n, p, p + d * 8, p + 2 * d * 8, p + 3 * d * 8), flush=True)
print("data %x size %x" % (a.data_ptr(), a.numel() * 8), flush=True)
ring.elementwise_crt_dev(a)
torch.cuda.synchronize()
print("done", flush=True)
