"""Every word of the BASELINE-sized batches against the oracle, in the driver-run GPU suite.

  C2 Goldilocks D = 2^16, batch 2^14     C3 BabyBear D = 2^16, batch 2^14     C5 Stark D = 2^12, batch 2^12
      a * b, crt(a), icrt(crt(a)) and a * b with b handed over in NTT form (sr_ring_mul_ntt_rhs_batch_dev) and, for BabyBear, the product on the opt-in packed-u32 boundary: every output word compared (adversarial patterns mixed into the batch: constant polynomials of
      p - 1, 2^32 +- 1, (p +- 1) / 2, monomials, alternating extremes), and neither operand may be written.
  C4 Goldilocks D = 2^20, the per-GPU shard of 8192 elements (64 GiB per operand, generated on device): the fused product in
      place, 64 sampled elements against the oracle, every output canonical, icrt(crt(c)) == c on a sub-range.
The oracle needs about 20 s per 2^30-coefficient case on 16 host threads; host memory peaks near 35 GiB."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib as O


@pytest.mark.parametrize("name,k,batch", [("goldilocks", 16, 1 << 14), ("babybear", 16, 1 << 14), ("stark", 12, 1 << 12)])
def test_every_word_of_the_baseline_batch(name, k, batch):
    import fuzz_full_parity as fz

    bad, bad_crt, bad_icrt, written, _, bad_rhs, bad_packed = fz.run(name, k, batch)
    assert (bad, bad_crt, bad_icrt, written, bad_rhs, bad_packed) == (0, 0, 0, 0, 0, 0)


def test_config4_shard_sampled_against_the_oracle():
    import torch

    from stark_rings_amd import CyclotomicRing

    k, batch, n_sample = 20, 1 << 13, 64
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < (150 << 30):
        pytest.skip("needs 150 GiB of free HBM (two 64 GiB operands + 16 GiB operand scratch)")
    F = O.GOLDILOCKS
    d = 1 << k
    ring = CyclotomicRing("goldilocks", k)
    a = torch.empty(batch * d, dtype=torch.int64, device="cuda")
    b = torch.empty(batch * d, dtype=torch.int64, device="cuda")
    ring.fill_uniform_dev(a, 0x5EED0001, 0)
    ring.fill_uniform_dev(b, 0x5EED0002, 0)
    ring.mul_dev(a, a, b)          # in place: c over a (the bench's configuration); runs in chunks of the 16 GiB scratch
    torch.cuda.synchronize()
    assert ring.count_noncanonical_dev(a) == 0
    sample = sorted({(i * (batch - 1)) // (n_sample - 1) for i in range(n_sample)})
    ea = np.concatenate([O.fill_uniform(F, 0x5EED0001, e * d, d) for e in sample])
    eb = np.concatenate([O.fill_uniform(F, 0x5EED0002, e * d, d) for e in sample])
    want = O.pow2_ring_mul(F, ea, eb, k, len(sample), len(os.sched_getaffinity(0)))
    for i, e in enumerate(sample):
        got = a[e * d:(e + 1) * d].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want[i * d:(i + 1) * d]), "element %d" % e
        assert np.array_equal(b[e * d:(e + 1) * d].cpu().numpy().view(np.uint64), eb[i * d:(i + 1) * d]), "b written at %d" % e
    rt = a[:16 * d].clone()
    ring.elementwise_crt_dev(rt)
    ring.elementwise_icrt_dev(rt)
    assert torch.equal(rt, a[:16 * d])
    ring.close()
