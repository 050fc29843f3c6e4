#!/usr/bin/env python3
"""Full-size differential run (not part of pytest): EVERY product of a BASELINE-sized batch against the oracle, with
adversarial coefficient patterns mixed into the batch (all p-1, 2^32 boundaries, sparse monomials, alternating extremes).
usage: fuzz_full_parity.py [goldilocks|babybear|stark] [log2_degree] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

import oracle_lib as O
import pyref as P
from stark_rings_amd import CyclotomicRing

name = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 12
F = O.FIELD_ID[name]
p = P.PRIMES[name][0]
d = 1 << k
L = O.LIMBS[F]
ring = CyclotomicRing(name, k)
a = O.fill_uniform(F, 0xABCD01, 0, batch * d)
b = O.fill_uniform(F, 0xABCD02, 0, batch * d)
specials = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 32) % p, ((1 << 32) - 1) % p, ((1 << 32) + 1) % p,
            (1 << 63) % p, ((1 << 64) - 1) % p, (p - (1 << 32)) % p, (p - (1 << 32) + 1) % p, 0xFFFFFFFF00000000 % p]
w = d * L
e = 0
for v in specials:                       # constant polynomials of every special value, and X^(D-1) times it
    if e + 2 >= batch:
        break
    a[e * w:(e + 1) * w] = O.to_mont(F, [v] * d)
    b[e * w:(e + 1) * w] = O.to_mont(F, [specials[(e * 7 + 3) % len(specials)]] * d)
    e += 1
    mono = [0] * d
    mono[d - 1] = v
    a[e * w:(e + 1) * w] = O.to_mont(F, mono)
    e += 1
alt = [(p - 1) if i & 1 else 0 for i in range(d)]
a[e * w:(e + 1) * w] = O.to_mont(F, alt)
b[e * w:(e + 1) * w] = O.to_mont(F, alt[::-1])
t0 = time.time()
want = O.pow2_ring_mul(F, a, b, k, batch, os.cpu_count() or 8)
t1 = time.time()
ta = torch.from_numpy(a.view(np.int64)).cuda()
tb = torch.from_numpy(b.view(np.int64)).cuda()
out = torch.empty_like(ta)
ring.mul_dev(out, ta, tb)
torch.cuda.synchronize()
got = out.cpu().numpy().view(np.uint64)
bad = np.nonzero(got != want)[0]
print("%s D=2^%d batch %d: oracle %.1f s; mismatching words: %d" % (name, k, batch, t1 - t0, bad.size))
fa = O.pow2_fwd(F, a, k, batch, os.cpu_count() or 8)
tf = torch.from_numpy(a.view(np.int64)).cuda()
ring.elementwise_crt_dev(tf)
bad2 = int((tf.cpu().numpy().view(np.uint64) != fa).sum())
ring.elementwise_icrt_dev(tf)
bad3 = int((tf.cpu().numpy().view(np.uint64) != a).sum())
print("crt mismatches %d, icrt(crt) mismatches %d" % (bad2, bad3))
sys.exit(1 if (bad.size or bad2 or bad3) else 0)
