#!/usr/bin/env python3
"""Full-size differential run: EVERY word of a BASELINE-sized batch against the oracle, with adversarial coefficient
patterns mixed into the batch (all p-1, 2^32 boundaries, sparse monomials, alternating extremes).  Run by the GPU test suite
(tests/test_full_size_parity.py) and stand-alone:
usage: fuzz_full_parity.py [goldilocks|babybear|stark] [log2_degree] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def adversarial_batch(name, k, batch):
    import numpy as np

    import oracle_lib as O
    import pyref as P

    F = O.FIELD_ID[name]
    p = P.PRIMES[name][0]
    d = 1 << k
    L = O.LIMBS[F]
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
    if e < batch:
        alt = [(p - 1) if i & 1 else 0 for i in range(d)]
        a[e * w:(e + 1) * w] = O.to_mont(F, alt)
        b[e * w:(e + 1) * w] = O.to_mont(F, alt[::-1])
    return a, b


def run(name, k, batch, threads=None):
    """returns (mismatching words of a * b, of crt(a), of icrt(crt(a)), operands written?, oracle seconds, mismatching words of the
    product with b handed over in NTT form, mismatching words of the packed-u32 product (BabyBear only))"""
    import numpy as np
    import torch

    import oracle_lib as O
    from stark_rings_amd import CyclotomicRing

    threads = threads or len(os.sched_getaffinity(0))
    F = O.FIELD_ID[name]
    ring = CyclotomicRing(name, k)
    a, b = adversarial_batch(name, k, batch)
    t0 = time.time()
    want = O.pow2_ring_mul(F, a, b, k, batch, threads)
    t1 = time.time()
    ta = torch.from_numpy(a.view(np.int64)).cuda()
    tb = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.empty_like(ta)
    ring.mul_dev(out, ta, tb)
    torch.cuda.synchronize()
    bad = int((out.cpu().numpy().view(np.uint64) != want).sum())
    # the constant-operand product (b kept in NTT form: its own fused kernel and lanes path on the tuned Goldilocks plan): every word too
    tbn = tb.clone()
    ring.elementwise_crt_dev(tbn)
    out.fill_(-1)
    ring.mul_ntt_rhs_dev(out, ta, tbn)
    torch.cuda.synchronize()
    bad_rhs = int((out.cpu().numpy().view(np.uint64) != want).sum())
    del tbn
    bad_packed = 0
    if name == "babybear":  # the opt-in packed-u32 boundary: pack, multiply on packed words, unpack -- every word again
        pa = torch.empty(ta.numel(), dtype=torch.int32, device="cuda")
        pb = torch.empty_like(pa)
        ring.pack32_dev(pa, ta)
        ring.pack32_dev(pb, tb)
        po = torch.full_like(pa, -1)
        ring.mul_packed32_dev(po, pa, pb)
        ring.unpack32_dev(out, po)
        torch.cuda.synchronize()
        bad_packed = int((out.cpu().numpy().view(np.uint64) != want).sum())
        ring.unpack32_dev(out, pa)      # the packed operands must come back untouched as well
        torch.cuda.synchronize()
        bad_packed += int((out.cpu().numpy().view(np.uint64) != a).sum())
        del pa, pb, po
    del want, out
    written = int((tb.cpu().numpy().view(np.uint64) != b).sum()) + int((ta.cpu().numpy().view(np.uint64) != a).sum())
    del tb
    fa = O.pow2_fwd(F, a, k, batch, threads)
    ring.elementwise_crt_dev(ta)
    bad2 = int((ta.cpu().numpy().view(np.uint64) != fa).sum())
    del fa
    ring.elementwise_icrt_dev(ta)
    bad3 = int((ta.cpu().numpy().view(np.uint64) != a).sum())
    ring.close()
    return bad, bad2, bad3, written, t1 - t0, bad_rhs, bad_packed


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 12
    bad, bad2, bad3, written, secs, bad_rhs, bad_packed = run(name, k, batch)
    print("%s D=2^%d batch %d: oracle %.1f s; mismatching words: %d" % (name, k, batch, secs, bad))
    print("crt mismatches %d, icrt(crt) mismatches %d, operand words written %d, ntt-rhs product mismatches %d, packed-u32 mismatches %d"
          % (bad2, bad3, written, bad_rhs, bad_packed))
    sys.exit(1 if (bad or bad2 or bad3 or written or bad_rhs or bad_packed) else 0)
