"""stark_rings_amd -- MI355X (gfx950) backend for stark-rings' CRT/NTT ring-multiplication path.

Layout:
  csrc/                 hand-written HIP kernels + the C ABI (include/stark_rings_hip.h)
  _lib.py               ctypes loader of libstarkrings_hip.so
  rings.py              host-side mirror of the reference interface (CyclotomicConfig / CRT / ICRT /
                        Flatten at batch granularity) on top of the C ABI
  wire.py               ark-serialize framing of Vec / Matrix / SparseMatrix around the device codec
  monomial.py           the reference's monomial helpers (monomial.rs) over the ring product
  sharding.py           batch sharding across the GPUs of one node (one process per GPU)
"""
from .rings import (  # noqa: F401
    BABYBEAR_72,
    BABYBEAR_POW2,
    GOLDILOCKS_24,
    GOLDILOCKS_POW2,
    STARK_POW2,
    CyclotomicRing,
    RingError,
)
