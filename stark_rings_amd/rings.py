"""Host-side mirror of the reference's ring interface, at batch granularity, over the C ABI.

Reference surface mirrored (crates/ring/src/cyclotomic_ring/):
  CyclotomicConfig<N>                      ring_config.rs:11-35   -> CyclotomicRing (one per ring + degree)
    reduce_in_place                        ring_config.rs:23      -> CyclotomicRing.reduce
    crt_in_place / crt                     ring_config.rs:27,29   -> CyclotomicRing.elementwise_crt
    icrt_in_place / icrt                   ring_config.rs:30,34   -> CyclotomicRing.elementwise_icrt
  CRT::elementwise_crt / ICRT::elementwise_icrt   crt.rs:10-25, 34-49 (in place, same allocation)
  RqNTT * RqNTT (MulAssign)                ntt_form.rs:159-225    -> CyclotomicRing.ntt_mul
  RqPoly * RqPoly                          coeff_form.rs:250-258  -> CyclotomicRing.mul
  Flatten::flatten_to_coeffs / promote_from_coeffs   flatten.rs:10-34 -> same names

Buffers are numpy uint64 arrays (host entry points) or torch uint64/int64 CUDA tensors (device
entry points), in the reference's in-memory layout: element-major, D coefficients per element,
N little-endian u64 limbs per coefficient, Montgomery residues.

Errors: the reference panics on wrong lengths (e.g. goldilocks/ntt.rs:136); here RingError is raised.
"""
import ctypes

import numpy as np

from . import _lib

GOLDILOCKS_POW2, BABYBEAR_POW2, STARK_POW2, GOLDILOCKS_24, BABYBEAR_72, FROG_16 = 0, 1, 2, 3, 4, 5
PROF_TAGS = ("fwd_cols", "rows", "inv_cols", "pointwise", "other")

_RING_NAMES = {
    "goldilocks": GOLDILOCKS_POW2,
    "babybear": BABYBEAR_POW2,
    "stark": STARK_POW2,
    "goldilocks24": GOLDILOCKS_24,
    "babybear72": BABYBEAR_72,
    "frog16": FROG_16,          # X^16 + 1 over the frog prime, 4 x Fq4 (frog_ring/mod.rs:62-107)
}


# BaseFieldConfig moduli: goldilocks/mod.rs:20-24, babybear/mod.rs:21-25, stark_prime/mod.rs:20-24, frog_ring/mod.rs:19-25
_MODULUS = {
    GOLDILOCKS_POW2: 2**64 - 2**32 + 1, GOLDILOCKS_24: 2**64 - 2**32 + 1,
    BABYBEAR_POW2: 2013265921, BABYBEAR_72: 2013265921,
    STARK_POW2: 2**251 + 17 * 2**192 + 1,
    FROG_16: 15912092521325583641,
}


class RingError(RuntimeError):
    pass


def _np_ptr(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]):
        raise RingError("expected a C-contiguous numpy uint64 array")
    return a.ctypes.data_as(_lib.u64p)


def _basis_words(basis, decompose):
    """(lo, hi) 64-bit words of a decomposition basis (the reference takes b: u128, balanced_decomposition/mod.rs:62).  Anything
    outside [0, 2^128) is refused instead of being truncated by ctypes.  decompose_balanced_in_place casts `b as i128` (mod.rs:73),
    so a basis of 2^127 or more would be a NEGATIVE basis there; that case is refused here (and by the C ABI) rather than computed
    with unsigned semantics the reference does not have.  Recomposition (`R::from(b)`, mod.rs:105-117) has no such cast."""
    basis = int(basis)
    if not 0 <= basis < 1 << 128:
        raise RingError("basis out of the u128 range")
    if decompose and basis >= 1 << 127:
        raise RingError("basis >= 2^127: negative after the reference's `b as i128` (balanced_decomposition/mod.rs:73); not supported")
    return basis & (2**64 - 1), basis >> 64


class CyclotomicRing:
    """One ring configuration bound to one HIP device (the analogue of a `CyclotomicConfig` impl)."""

    def __init__(self, ring, log2_degree=0, device=0, plan=None):
        """plan: a _lib.Plan (sr_plan); None = the SR_* switches of the environment as parsed by _lib.plan_from_env (all unset =
        the library defaults).  The library itself never reads the environment."""
        if isinstance(ring, str):
            ring = _RING_NAMES[ring]
        self._lib = _lib.load()
        self._ctx = ctypes.c_void_p()
        if plan is None:
            plan = _lib.plan_from_env(int(ring))
        self.plan = plan
        rc = self._lib.sr_ctx_create_ex(int(ring), int(log2_degree), int(device), ctypes.byref(plan), ctypes.byref(self._ctx))
        if rc != 0:
            self._ctx = None
            raise RingError("sr_ctx_create_ex failed (%d): %s" % (rc, _lib.last_error()))
        self.ring = int(ring)
        self.device = int(device)
        d = ctypes.c_size_t()
        l = ctypes.c_int()
        self._check(self._lib.sr_ctx_degree(self._ctx, ctypes.byref(d)))
        self._check(self._lib.sr_ctx_limbs(self._ctx, ctypes.byref(l)))
        self.degree = d.value          # PolyRing::dimension()
        self.limbs = l.value           # N
        self.words_per_elem = self.degree * self.limbs
        self.modulus = _MODULUS[self.ring]

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.sr_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RingError("stark_rings_hip call failed (%d): %s" % (rc, _lib.last_error()))

    def _batch_of(self, arr_words):
        if arr_words % self.words_per_elem != 0:
            # promote_from_coeffs returns None in the reference (flatten.rs:22-24); transforms panic
            raise RingError("buffer length %d is not a multiple of D*N = %d" % (arr_words, self.words_per_elem))
        return arr_words // self.words_per_elem

    # -- Flatten (flatten.rs:10-34): zero-copy views ---------------------------------------------
    def flatten_to_coeffs(self, elems):
        """(batch, D[, N]) -> flat view of batch*D coefficients."""
        return elems.reshape(-1, self.limbs) if self.limbs > 1 else elems.reshape(-1)

    def promote_from_coeffs(self, flat):
        """flat coefficients -> (batch, D[, N]) view, or None if the length is not a multiple of D."""
        n = flat.size if isinstance(flat, np.ndarray) else flat.numel()
        if n % self.words_per_elem != 0:
            return None
        shape = (n // self.words_per_elem, self.degree) + ((self.limbs,) if self.limbs > 1 else ())
        return flat.reshape(shape)

    # -- host-buffer API (numpy) -----------------------------------------------------------
    def elementwise_crt(self, data):
        """CRT::elementwise_crt (crt.rs:10-25): in place on a numpy uint64 buffer."""
        self._check(self._lib.sr_ntt_fwd_batch(self._ctx, _np_ptr(data), self._batch_of(data.size)))
        return data

    def elementwise_icrt(self, data):
        """ICRT::elementwise_icrt (crt.rs:34-49): in place."""
        self._check(self._lib.sr_ntt_inv_batch(self._ctx, _np_ptr(data), self._batch_of(data.size)))
        return data

    def ntt_mul(self, lhs, rhs):
        """lhs *= rhs slot-wise in CRT form (ntt_form.rs:213-225), in place on lhs."""
        if lhs.size != rhs.size:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_pointwise_mul_batch(self._ctx, _np_ptr(lhs), _np_ptr(rhs), self._batch_of(lhs.size)))
        return lhs

    def add(self, lhs, rhs):
        """lhs += rhs element-wise (ntt_form.rs:227-285 / coeff_form.rs Add), in place on lhs."""
        if lhs.size != rhs.size:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_add_batch(self._ctx, _np_ptr(lhs), _np_ptr(rhs), self._batch_of(lhs.size)))
        return lhs

    def sub(self, lhs, rhs):
        """lhs -= rhs element-wise (ntt_form.rs:588-638), in place on lhs."""
        if lhs.size != rhs.size:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_sub_batch(self._ctx, _np_ptr(lhs), _np_ptr(rhs), self._batch_of(lhs.size)))
        return lhs

    def _scalar(self, scalar):
        """one base-field element as the library takes it: `limbs` u64 words, the Montgomery memory image of Fp::from(rhs)"""
        s = np.ascontiguousarray(np.asarray(scalar, dtype=np.uint64).reshape(-1))
        if s.size != self.limbs:
            raise RingError("scalar must be %d u64 limb(s)" % self.limbs)
        return s

    def neg(self, data):
        """Neg (coeff_form.rs:270-278, ntt_form.rs:191-203): data = -data word-wise, in place, either form."""
        self._check(self._lib.sr_neg_batch(self._ctx, _np_ptr(data), self._batch_of(data.size)))
        return data

    def scale(self, data, scalar):
        """Mul<Fp> / Mul<primitive> / MulAssign (coeff_form.rs:390-408, 610-650; ntt_form.rs:373-425): every coefficient (slot
        component) times one base-field scalar given as its Montgomery image; in place, either form."""
        s = self._scalar(scalar)
        self._check(self._lib.sr_scale_batch(self._ctx, _np_ptr(data), _np_ptr(s), self._batch_of(data.size)))
        return data

    def mul_elem(self, data, elem):
        """`Matrix<R> *= &R` / `SparseMatrix<R> *= &R` (linear_algebra/src/matrix.rs:207-211, sparse_matrix.rs:303-307) on the
        matrix's flat storage: every element of the batch (CRT/NTT form) times ONE ring element, slot-wise, in place."""
        if elem.size != self.words_per_elem:
            raise RingError("mul_elem: the multiplier is not one ring element")
        if data.size:
            self._check(self._lib.sr_mul_elem_batch(self._ctx, _np_ptr(data), _np_ptr(elem), self._batch_of(data.size)))
        return data

    def add_scalar(self, data, scalar, ntt_form):
        """Add<primitive> (coeff_form.rs:652-700: coefficient 0 of every element; ntt_form.rs:427-505: component 0 of every slot);
        Sub: pass the negated scalar.  In place."""
        s = self._scalar(scalar)
        self._check(self._lib.sr_add_scalar_batch(self._ctx, _np_ptr(data), _np_ptr(s), 1 if ntt_form else 0, self._batch_of(data.size)))
        return data

    def sum(self, elems):
        """impl Sum for RqPoly / RqNTT (coeff_form.rs:507-521, ntt_form.rs:640-654): `iter.fold(zero(), |acc, x| acc + x)` over a slice
        of ring elements, word-wise, either form; an empty slice gives zero().  Returns one ring element."""
        out = np.empty(self.words_per_elem, dtype=np.uint64)
        src = elems if elems.size else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_sum_batch(self._ctx, _np_ptr(out), _np_ptr(src), self._batch_of(elems.size)))
        return out

    def product(self, elems_ntt):
        """impl Product for RqNTT (ntt_form.rs:656-670): `iter.fold(one(), |acc, x| acc * x)`, slot-wise; an empty slice gives one()."""
        out = np.empty(self.words_per_elem, dtype=np.uint64)
        src = elems_ntt if elems_ntt.size else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_product_batch(self._ctx, _np_ptr(out), _np_ptr(src), self._batch_of(elems_ntt.size)))
        return out

    def product_poly(self, elems):
        """impl Product for RqPoly (coeff_form.rs:523-537: `iter.fold(one(), |acc, x| acc * x)` with the ring product) as
        icrt(product(crt(x_i))): the CRT is a ring isomorphism, so the slot-wise product of the transforms is the transform of the
        product.  The input is not modified."""
        t = self.elementwise_crt(elems.copy()) if elems.size else elems
        return self.elementwise_icrt(self.product(t))

    def mul(self, a, b, out=None):
        """Coefficient-form product a * b (coeff_form.rs:250-258) via icrt(crt(a) * crt(b))."""
        if a.size != b.size:
            raise RingError("operand lengths differ")
        if out is None:
            out = np.empty_like(a)
        self._check(self._lib.sr_ring_mul_batch(self._ctx, _np_ptr(out), _np_ptr(a), _np_ptr(b), self._batch_of(a.size)))
        return out

    def matvec_ntt(self, m, v, nrows, ncols):
        """Host buffers: Matrix<RqNTT>::checked_mul_vec (matrix.rs:168-178); RingError where the reference returns None."""
        w = self.words_per_elem
        if m.size != nrows * ncols * w or v.size != ncols * w:
            raise RingError("matvec: DifferentLengths")
        y = np.empty(max(nrows * w, 1), dtype=np.uint64)[:nrows * w]
        self._check(self._lib.sr_matvec_ntt(self._ctx, _np_ptr(y) if nrows else _np_ptr(np.zeros(1, dtype=np.uint64)),
                                            _np_ptr(m) if m.size else _np_ptr(np.zeros(1, dtype=np.uint64)),
                                            _np_ptr(v) if v.size else _np_ptr(np.zeros(1, dtype=np.uint64)), nrows, ncols))
        return y

    def spmv_ntt(self, rows, v, ncols):
        """rows: the reference's SparseMatrix.coeffs -- a list (one per row) of lists of (ring element as uint64 words, column).
        SparseMatrix<RqNTT>::checked_mul_vec (sparse_matrix.rs:201-211); an out-of-range column raises RingError (the
        reference panics on v[col])."""
        w = self.words_per_elem
        if v.size != ncols * w:
            raise RingError("spmv: DifferentLengths")
        nrows = len(rows)
        row_ptr = np.zeros(nrows + 1, dtype=np.uint64)
        for r, row in enumerate(rows):
            row_ptr[r + 1] = row_ptr[r] + len(row)
        nnz = int(row_ptr[-1])
        vals = np.zeros(max(nnz * w, 1), dtype=np.uint64)
        cols = np.zeros(max(nnz, 1), dtype=np.uint32)
        j = 0
        for row in rows:
            for elem, col in row:
                vals[j * w:(j + 1) * w] = elem
                cols[j] = col
                j += 1
        y = np.empty(max(nrows * w, 1), dtype=np.uint64)
        vv = v if v.size else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_spmv_ntt(self._ctx, _np_ptr(y), _np_ptr(vals), cols.ctypes.data_as(ctypes.c_void_p), _np_ptr(row_ptr), _np_ptr(vv),
                                          nrows, ncols))
        return y[:nrows * w]

    def matmul_ntt(self, a, b, n, m, p):
        """Host buffers: Matrix<RqNTT>::checked_mul_mat (matrix.rs:148-166)."""
        w = self.words_per_elem
        if a.size != n * m * w or b.size != m * p * w:
            raise RingError("matmul: DifferentLengths")
        y = np.empty(max(n * p * w, 1), dtype=np.uint64)
        z = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_matmul_ntt(self._ctx, _np_ptr(y), _np_ptr(a) if a.size else _np_ptr(z),
                                            _np_ptr(b) if b.size else _np_ptr(z), n, m, p))
        return y[:n * p * w]

    def rot(self, data):
        """Cyclotomic::rot (traits.rs:54-66) of every element of the batch, in place: coefficients times X modulo the ring."""
        self._check(self._lib.sr_rot_batch(self._ctx, _np_ptr(data), self._batch_of(data.size)))
        return data

    def gadget_decompose(self, a, basis, padding_size):
        """GadgetDecompose for a Vec of ring elements in coefficient form (balanced_decomposition/mod.rs:163-175): returns
        len * padding_size elements, digit j of element e at index e * padding_size + j.  RingError where the reference panics
        (basis 0, 1 or odd; more than padding_size digits needed)."""
        batch = self._batch_of(a.size)
        out = np.empty(max(batch * padding_size * self.words_per_elem, 1), dtype=np.uint64)
        src = a if a.size else np.zeros(1, dtype=np.uint64)
        lo, hi = _basis_words(basis, True)
        self._check(self._lib.sr_decompose_balanced_batch_wide(self._ctx, _np_ptr(out), _np_ptr(src), lo, hi, padding_size, batch))
        return out[:batch * padding_size * self.words_per_elem]

    def gadget_recompose(self, digits, basis, padding_size):
        """GadgetRecompose (mod.rs:177-189): len / padding_size elements, each sum_j basis^j * digits[e * padding_size + j]."""
        n = self._batch_of(digits.size)
        if padding_size == 0 or n % padding_size:
            raise RingError("recompose: length is not a multiple of padding_size")
        batch_out = n // padding_size
        out = np.empty(max(batch_out * self.words_per_elem, 1), dtype=np.uint64)
        src = digits if digits.size else np.zeros(1, dtype=np.uint64)
        lo, hi = _basis_words(basis, False)
        self._check(self._lib.sr_recompose_batch_wide(self._ctx, _np_ptr(out), _np_ptr(src), lo, hi, padding_size, batch_out))
        return out[:batch_out * self.words_per_elem]

    # -- GadgetDecompose / GadgetRecompose for Matrix<R> and SparseMatrix<R> (balanced_decomposition/mod.rs:276-352) ------------
    def matrix_gadget_decompose(self, mat, nrows, ncols, basis, padding_size):
        """Matrix<R> of nrows x ncols ring elements (row-major, coefficient form) -> nrows x (padding_size * ncols): every row
        is gadget-decomposed as a slice (mod.rs:291-296), i.e. entry (r, c) becomes entries (r, c * k .. c * k + k - 1).  Row-major
        storage makes this the batch decomposition of the flat element array.  Returns (flat words, nrows, ncols * padding_size)."""
        if mat.size != nrows * ncols * self.words_per_elem:
            raise RingError("matrix_gadget_decompose: DifferentLengths")
        return self.gadget_decompose(mat, basis, padding_size), nrows, ncols * padding_size

    def matrix_gadget_recompose(self, mat, nrows, ncols, basis, padding_size):
        """Inverse shape map (mod.rs:299-307): nrows x ncols -> nrows x (ncols / padding_size)."""
        if mat.size != nrows * ncols * self.words_per_elem or padding_size == 0 or ncols % padding_size:
            raise RingError("matrix_gadget_recompose: ncols is not a multiple of padding_size")
        return self.gadget_recompose(mat, basis, padding_size), nrows, ncols // padding_size

    def sparse_gadget_decompose(self, rows, ncols, basis, padding_size):
        """SparseMatrix<R> (rows: list of lists of (element words, column)) -> the same with ncols * padding_size columns
        (mod.rs:323-336): stored entry (e, c) becomes (digit_i(e), c * k + i) for i < k, zero digits dropped ("maintain full
        sparsity", mod.rs:208-229).  One batched device decomposition over all stored entries."""
        w = self.words_per_elem
        flat = [np.ascontiguousarray(e, dtype=np.uint64) for row in rows for e, _ in row]
        if any(e.size != w for e in flat):
            raise RingError("sparse_gadget_decompose: entry is not one ring element")
        if any(c >= ncols for row in rows for _, c in row):
            raise RingError("sparse_gadget_decompose: column out of range")
        digits = self.gadget_decompose(np.concatenate(flat) if flat else np.zeros(0, dtype=np.uint64), basis, padding_size)
        out, j = [], 0
        for row in rows:
            new = []
            for _, c in row:
                for i in range(padding_size):
                    dgt = digits[(j * padding_size + i) * w:(j * padding_size + i + 1) * w]
                    if dgt.any():               # r != R::zero(): the Montgomery image of zero is all-zero words
                        new.append((dgt.copy(), c * padding_size + i))
                j += 1
            out.append(new)
        return out, ncols * padding_size

    def sparse_gadget_recompose(self, rows, ncols, basis, padding_size):
        """mod.rs:231-266, 339-351: consecutive entries whose column / padding_size agree form one original entry; missing
        digits are zero."""
        if padding_size == 0 or ncols % padding_size:
            raise RingError("sparse_gadget_recompose: ncols is not a multiple of padding_size")
        w = self.words_per_elem
        groups, shape = [], []
        for row in rows:
            cnt, prev = 0, None
            for e, c in row:
                idx = c // padding_size
                if idx != prev:
                    groups.append((idx, np.zeros(padding_size * w, dtype=np.uint64)))
                    cnt += 1
                    prev = idx
                groups[-1][1][(c % padding_size) * w:(c % padding_size + 1) * w] = e
            shape.append(cnt)
        vals = self.gadget_recompose(np.concatenate([g[1] for g in groups]) if groups else np.zeros(0, dtype=np.uint64), basis,
                                     padding_size)
        out, j = [], 0
        for cnt in shape:
            out.append([(vals[(j + t) * w:(j + t + 1) * w].copy(), groups[j + t][0]) for t in range(cnt)])
            j += cnt
        return out, ncols // padding_size

    @property
    def wire_coeff_bytes(self):
        """Bytes per coefficient on the ark-serialize wire: 8 (Goldilocks, frog), 4 (BabyBear), 32 (Stark)."""
        return int(self._lib.sr_wire_coeff_bytes(self._ctx))

    def serialize(self, a):
        """CanonicalSerialize of every ring element of the batch (coeff_form.rs:154-189, ntt_form.rs:24): the flat coefficients as
        little-endian standard-form integers, wire_coeff_bytes each, no length prefix.  Returns a uint8 array."""
        batch = self._batch_of(a.size)
        out = np.empty(max(batch * self.degree * self.wire_coeff_bytes, 1), dtype=np.uint8)
        src = a if a.size else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_serialize_batch(self._ctx, out.ctypes.data_as(ctypes.c_void_p), _np_ptr(src), batch))
        return out[:batch * self.degree * self.wire_coeff_bytes]

    def deserialize(self, wire):
        """CanonicalDeserialize of a whole number of ring elements; RingError (ark: InvalidData) when a coefficient is >= p."""
        wire = np.ascontiguousarray(wire, dtype=np.uint8)
        per = self.degree * self.wire_coeff_bytes
        if wire.size % per:
            raise RingError("deserialize: not a whole number of ring elements")
        batch = wire.size // per
        out = np.empty(max(batch * self.words_per_elem, 1), dtype=np.uint64)
        src = wire if wire.size else np.zeros(8, dtype=np.uint8)
        self._check(self._lib.sr_deserialize_batch(self._ctx, _np_ptr(out), src.ctypes.data_as(ctypes.c_void_p), batch))
        return out[:batch * self.words_per_elem]

    def reduce(self, coeffs, in_len_per_elem, batch):
        """CyclotomicConfig::reduce_in_place (ring_config.rs:23): (batch, in_len) -> (batch, D)."""
        if coeffs.size != batch * in_len_per_elem * self.limbs:
            raise RingError("reduce: buffer length does not match batch * in_len")
        out = np.empty(batch * self.words_per_elem, dtype=np.uint64)
        src = coeffs if coeffs.size else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sr_reduce_batch(self._ctx, _np_ptr(src), in_len_per_elem, _np_ptr(out), batch))
        return out

    # -- device-resident API (torch CUDA tensors of 8-byte integers) --------------------------
    def _dev(self, t):
        if not (t.is_cuda and t.is_contiguous() and t.element_size() == 8):
            raise RingError("expected a contiguous CUDA tensor of 8-byte integers")
        if t.device.index != self.device:
            raise RingError("tensor lives on cuda:%d, this context on cuda:%d" % (t.device.index, self.device))
        return ctypes.c_void_p(t.data_ptr()), t.numel()

    def _stream(self, stream):
        import torch

        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        if s.device.index != self.device:
            raise RingError("stream belongs to cuda:%d, this context to cuda:%d" % (s.device.index, self.device))
        return ctypes.c_void_p(s.cuda_stream)

    # -- packed-u32 boundary (BabyBear power-of-two rings; include/stark_rings_hip.h "packed-u32 boundary") ------------------------
    # A packed tensor holds the LOW HALF of every reference limb: int32 / uint32 CUDA tensors of batch * D words, the uint32
    # (a * 2^64 mod p) of babybear/mod.rs:18-26's Fp64 -- same Montgomery residue, four bytes.
    def _dev32(self, t):
        if not (t.is_cuda and t.is_contiguous() and t.element_size() == 4):
            raise RingError("expected a contiguous CUDA tensor of 4-byte integers (packed-u32 image)")
        if t.device.index != self.device:
            raise RingError("tensor lives on cuda:%d, this context on cuda:%d" % (t.device.index, self.device))
        return ctypes.c_void_p(t.data_ptr()), t.numel()

    def pack32_dev(self, out32, in64, stream=None):
        """8-byte reference image -> packed image (low word of every limb; the input must be canonical)."""
        po, n = self._dev32(out32)
        pi, m = self._dev(in64)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_pack32_batch_dev(self._ctx, po, pi, self._batch_of(n), self._stream(stream)))
        return out32

    def unpack32_dev(self, out64, in32, stream=None):
        po, n = self._dev(out64)
        pi, m = self._dev32(in32)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_unpack32_batch_dev(self._ctx, po, pi, self._batch_of(n), self._stream(stream)))
        return out64

    def elementwise_crt_packed32_dev(self, t, stream=None):
        p, n = self._dev32(t)
        self._check(self._lib.sr_ntt_fwd_packed32_batch_dev(self._ctx, p, self._batch_of(n), self._stream(stream)))
        return t

    def elementwise_icrt_packed32_dev(self, t, stream=None):
        p, n = self._dev32(t)
        self._check(self._lib.sr_ntt_inv_packed32_batch_dev(self._ctx, p, self._batch_of(n), self._stream(stream)))
        return t

    def mul_packed32_dev(self, out, a, b, stream=None):
        """out = a * b on packed images (RqPoly * &RqPoly, coeff_form.rs:250-258); a and b are only read; out may be a."""
        po, n = self._dev32(out)
        pa, m = self._dev32(a)
        pb, q = self._dev32(b)
        if not (n == m == q):
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_ring_mul_packed32_batch_dev(self._ctx, po, pa, pb, self._batch_of(n), self._stream(stream)))
        return out

    def _ew32(self, fn, lhs, rhs, stream):
        pl, n = self._dev32(lhs)
        pr, m = self._dev32(rhs)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(fn(self._ctx, pl, pr, self._batch_of(n), self._stream(stream)))
        return lhs

    def ntt_mul_packed32_dev(self, lhs, rhs, stream=None):
        return self._ew32(self._lib.sr_pointwise_mul_packed32_batch_dev, lhs, rhs, stream)

    def add_packed32_dev(self, lhs, rhs, stream=None):
        return self._ew32(self._lib.sr_add_packed32_batch_dev, lhs, rhs, stream)

    def sub_packed32_dev(self, lhs, rhs, stream=None):
        return self._ew32(self._lib.sr_sub_packed32_batch_dev, lhs, rhs, stream)

    def reserve_scratch(self, batch):
        """sr_ctx_reserve_scratch: pre-size the operand scratch so that no later mul_dev of up to `batch` elements blocks.  With
        sr_plan.lanes = 0 (auto) this is also where the library times its two plans once and keeps the faster (plan_in_use)."""
        self._check(self._lib.sr_ctx_reserve_scratch(self._ctx, int(batch)))

    def plan_in_use(self):
        """sr_ctx_plan_in_use: (plan, probe) -- the context's sr_plan with `lanes` resolved to what the library settled on (0 = auto,
        not settled yet) and, when the library measured, {"two_lanes_ms", "one_stream_ms", "elems"} of its probe (else None)."""
        plan = _lib.Plan()
        ms = (ctypes.c_double * 2)()
        n = ctypes.c_size_t(0)
        self._check(self._lib.sr_ctx_plan_in_use(self._ctx, ctypes.byref(plan), ms, ctypes.byref(n)))
        probe = {"two_lanes_ms": ms[0], "one_stream_ms": ms[1], "elems": int(n.value)} if n.value else None
        return plan, probe

    def elementwise_crt_dev(self, t, stream=None):
        p, n = self._dev(t)
        self._check(self._lib.sr_ntt_fwd_batch_dev(self._ctx, p, self._batch_of(n), self._stream(stream)))
        return t

    def elementwise_icrt_dev(self, t, stream=None):
        p, n = self._dev(t)
        self._check(self._lib.sr_ntt_inv_batch_dev(self._ctx, p, self._batch_of(n), self._stream(stream)))
        return t

    def ntt_mul_dev(self, lhs, rhs, stream=None):
        pl, n = self._dev(lhs)
        pr, m = self._dev(rhs)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_pointwise_mul_batch_dev(self._ctx, pl, pr, self._batch_of(n), self._stream(stream)))
        return lhs

    def add_dev(self, lhs, rhs, stream=None):
        pl, n = self._dev(lhs)
        pr, m = self._dev(rhs)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_add_batch_dev(self._ctx, pl, pr, self._batch_of(n), self._stream(stream)))
        return lhs

    def sub_dev(self, lhs, rhs, stream=None):
        pl, n = self._dev(lhs)
        pr, m = self._dev(rhs)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_sub_batch_dev(self._ctx, pl, pr, self._batch_of(n), self._stream(stream)))
        return lhs

    def neg_dev(self, t, stream=None):
        p, n = self._dev(t)
        self._check(self._lib.sr_neg_batch_dev(self._ctx, p, self._batch_of(n), self._stream(stream)))
        return t

    def scale_dev(self, t, scalar, stream=None):
        p, n = self._dev(t)
        s = self._scalar(scalar)
        self._check(self._lib.sr_scale_batch_dev(self._ctx, p, _np_ptr(s), self._batch_of(n), self._stream(stream)))
        return t

    def mul_elem_dev(self, t, elem, stream=None):
        p, n = self._dev(t)
        pe, ne = self._dev(elem)
        if ne != self.words_per_elem:
            raise RingError("mul_elem: the multiplier is not one ring element")
        self._check(self._lib.sr_mul_elem_batch_dev(self._ctx, p, pe, self._batch_of(n), self._stream(stream)))
        return t

    def sum_dev(self, out, elems, stream=None):
        """Sum over a device-resident slice (see sum): out = one ring element, must not overlap elems."""
        if out.numel() != self.words_per_elem:
            raise RingError("sum: out is not one ring element")
        n = self._batch_of(elems.numel())
        po = self._dev(out)[0]
        self._check(self._lib.sr_sum_batch_dev(self._ctx, po, self._dev(elems)[0] if n else po, n, self._stream(stream)))
        return out

    def product_dev(self, out, elems_ntt, stream=None):
        """Product over a device-resident slice in CRT/NTT form (see product)."""
        if out.numel() != self.words_per_elem:
            raise RingError("product: out is not one ring element")
        n = self._batch_of(elems_ntt.numel())
        po = self._dev(out)[0]
        self._check(self._lib.sr_product_batch_dev(self._ctx, po, self._dev(elems_ntt)[0] if n else po, n, self._stream(stream)))
        return out

    def product_poly_dev(self, out, elems, stream=None):
        """Product of coefficient-form elements (see product_poly); `elems` is transformed IN PLACE (it holds crt(elems) afterwards)."""
        if elems.numel():
            self.elementwise_crt_dev(elems, stream=stream)
        self.product_dev(out, elems, stream=stream)
        return self.elementwise_icrt_dev(out, stream=stream)

    def add_scalar_dev(self, t, scalar, ntt_form, stream=None):
        p, n = self._dev(t)
        s = self._scalar(scalar)
        self._check(self._lib.sr_add_scalar_batch_dev(self._ctx, p, _np_ptr(s), 1 if ntt_form else 0, self._batch_of(n), self._stream(stream)))
        return t

    def matvec_ntt_dev(self, y, m, v, nrows, ncols, stream=None):
        """y = M v for M (nrows x ncols ring elements, row-major) and v (ncols elements), all in CRT/NTT form:
        Matrix<RqNTT>::checked_mul_vec (linear_algebra/src/matrix.rs:168-178); the reference returns None on a length
        mismatch, here RingError is raised."""
        py, ny = self._dev(y)
        pm, nm = self._dev(m)
        pv, nv = self._dev(v)
        if nm != nrows * ncols * self.words_per_elem or nv != ncols * self.words_per_elem or ny != nrows * self.words_per_elem:
            raise RingError("matvec: DifferentLengths")
        self._check(self._lib.sr_matvec_ntt_dev(self._ctx, py, pm, pv, nrows, ncols, self._stream(stream)))
        return y

    def spmv_ntt_dev(self, y, vals, cols, row_ptr, v, nrows, ncols, stream=None):
        """y = S v for a CSR sparse matrix of ring elements in CRT/NTT form: SparseMatrix<RqNTT>::checked_mul_vec
        (linear_algebra/src/sparse_matrix.rs:201-211).  vals: nnz ring elements (int64 words), cols: int32 column of each,
        row_ptr: int64 [nrows + 1].  Entries whose column is >= ncols are skipped and counted (spmv_bad_index_count)."""
        import torch

        py, ny = self._dev(y)
        pv, nv = self._dev(v)
        if cols.dtype != torch.int32 or row_ptr.dtype != torch.int64 or row_ptr.numel() != nrows + 1:
            raise RingError("spmv: cols must be int32 and row_ptr int64 of length nrows + 1")
        nnz = cols.numel()
        if vals.numel() != nnz * self.words_per_elem or nv != ncols * self.words_per_elem or ny != nrows * self.words_per_elem:
            raise RingError("spmv: DifferentLengths")
        pvals = self._dev(vals)[0] if nnz else 0
        self._check(self._lib.sr_spmv_ntt_dev(self._ctx, py, pvals, cols.data_ptr() if nnz else 0, row_ptr.data_ptr(), pv,
                                              nrows, ncols, self._stream(stream)))
        return y

    def spmv_bad_index_count(self, stream=None):
        import ctypes

        n = ctypes.c_ulonglong(0)
        self._check(self._lib.sr_spmv_bad_index_count(self._ctx, ctypes.byref(n), self._stream(stream)))
        return int(n.value)

    def matmul_ntt_dev(self, y, a, b, n, m, p, stream=None):
        """Y (n x p) = A (n x m) B (m x p), dense row-major, CRT/NTT form: Matrix<RqNTT>::checked_mul_mat
        (linear_algebra/src/matrix.rs:148-166)."""
        py, ny = self._dev(y)
        pa, na = self._dev(a)
        pb, nb = self._dev(b)
        w = self.words_per_elem
        if na != n * m * w or nb != m * p * w or ny != n * p * w:
            raise RingError("matmul: DifferentLengths")
        self._check(self._lib.sr_matmul_ntt_dev(self._ctx, py, pa, pb, n, m, p, self._stream(stream)))
        return y

    def rot_dev(self, out, a, stream=None):
        po, n = self._dev(out)
        pa, m = self._dev(a)
        if n != m:
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_rot_batch_dev(self._ctx, po, pa, self._batch_of(n), self._stream(stream)))
        return out

    def gadget_decompose_dev(self, out, a, basis, padding_size, stream=None):
        po, n = self._dev(out)
        pa, m = self._dev(a)
        if n != m * padding_size:
            raise RingError("decompose: out must hold len * padding_size elements")
        lo, hi = _basis_words(basis, True)
        self._check(self._lib.sr_decompose_balanced_batch_wide_dev(self._ctx, po, pa, lo, hi, padding_size,
                                                                  self._batch_of(m), self._stream(stream)))
        return out

    def decompose_overflow_count(self, stream=None):
        n = ctypes.c_ulonglong(0)
        self._check(self._lib.sr_decompose_overflow_count(self._ctx, ctypes.byref(n), self._stream(stream)))
        return int(n.value)

    def gadget_recompose_dev(self, out, digits, basis, padding_size, stream=None):
        po, n = self._dev(out)
        pd, m = self._dev(digits)
        if m != n * padding_size:
            raise RingError("recompose: digits must hold len(out) * padding_size elements")
        lo, hi = _basis_words(basis, False)
        self._check(self._lib.sr_recompose_batch_wide_dev(self._ctx, po, pd, lo, hi, padding_size,
                                                         self._batch_of(n), self._stream(stream)))
        return out

    def serialize_dev(self, wire, a, offsets=None, stream=None):
        """wire: uint8 CUDA tensor; a: ring elements; offsets: optional int64 CUDA tensor, one byte offset (multiple of 8) per
        element, for callers that interleave framing words; None = densely packed."""
        import torch

        pa, n = self._dev(a)
        batch = self._batch_of(n)
        if wire.dtype != torch.uint8 or not wire.is_cuda or not wire.is_contiguous():
            raise RingError("serialize: wire must be a contiguous uint8 CUDA tensor")
        if offsets is None:
            if wire.numel() != batch * self.degree * self.wire_coeff_bytes:
                raise RingError("serialize: wire must hold len * D * wire_coeff_bytes bytes")
            po = 0
        else:
            if offsets.dtype != torch.int64 or offsets.numel() != batch or not offsets.is_cuda:
                raise RingError("serialize: offsets must be an int64 CUDA tensor with one entry per element")
            po = offsets.data_ptr()
        self._check(self._lib.sr_serialize_batch_dev(self._ctx, wire.data_ptr(), pa, po, batch, self._stream(stream)))
        return wire

    def deserialize_dev(self, out, wire, offsets=None, stream=None):
        """Inverse of serialize_dev; coefficients >= p are stored as 0 and counted (wire_invalid_count)."""
        import torch

        po_, n = self._dev(out)
        batch = self._batch_of(n)
        if wire.dtype != torch.uint8 or not wire.is_cuda or not wire.is_contiguous():
            raise RingError("deserialize: wire must be a contiguous uint8 CUDA tensor")
        if offsets is None:
            if wire.numel() != batch * self.degree * self.wire_coeff_bytes:
                raise RingError("deserialize: wire must hold len * D * wire_coeff_bytes bytes")
            po = 0
        else:
            if offsets.dtype != torch.int64 or offsets.numel() != batch or not offsets.is_cuda:
                raise RingError("deserialize: offsets must be an int64 CUDA tensor with one entry per element")
            po = offsets.data_ptr()
        self._check(self._lib.sr_deserialize_batch_dev(self._ctx, po_, wire.data_ptr(), po, batch, self._stream(stream)))
        return out

    def wire_invalid_count(self, stream=None):
        n = ctypes.c_ulonglong(0)
        self._check(self._lib.sr_wire_invalid_count(self._ctx, ctypes.byref(n), self._stream(stream)))
        return int(n.value)

    def mul_ntt_rhs(self, a, b_ntt):
        """Host buffers: icrt(crt(a) (.) b_ntt) with b_ntt = crt(b) already in CRT/NTT form; returns a new array."""
        if a.size != b_ntt.size:
            raise RingError("operand lengths differ")
        out = np.empty_like(a)
        self._check(self._lib.sr_ring_mul_ntt_rhs_batch(self._ctx, _np_ptr(out), _np_ptr(a), _np_ptr(b_ntt), self._batch_of(a.size)))
        return out

    def mul_ntt_rhs_dev(self, out, a, b_ntt, stream=None):
        """out = icrt(crt(a) (.) b_ntt) for b_ntt = crt(b) already in CRT/NTT form (the constant-operand product); a and b_ntt
        are only read; out may be a."""
        po, n = self._dev(out)
        pa, m = self._dev(a)
        pb, q = self._dev(b_ntt)
        if not (n == m == q):
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_ring_mul_ntt_rhs_batch_dev(self._ctx, po, pa, pb, self._batch_of(n), self._stream(stream)))
        return out

    def mul_dev(self, out, a, b, stream=None):
        """out = a * b (RqPoly * &RqPoly, coeff_form.rs:250-258); a and b are only read; out may be a."""
        po, n = self._dev(out)
        pa, m = self._dev(a)
        pb, q = self._dev(b)
        if not (n == m == q):
            raise RingError("operand lengths differ")
        self._check(self._lib.sr_ring_mul_batch_dev(self._ctx, po, pa, pb, self._batch_of(n), self._stream(stream)))
        return out

    def reduce_dev(self, out, coeffs, in_len_per_elem, stream=None):
        po, n = self._dev(out)
        pi, m = self._dev(coeffs)
        batch = self._batch_of(n)
        if m != batch * in_len_per_elem * self.limbs:
            raise RingError("reduce: buffer length does not match batch * in_len")
        self._check(self._lib.sr_reduce_batch_dev(self._ctx, pi, in_len_per_elem, po, batch, self._stream(stream)))
        return out

    def fill_uniform_dev(self, t, seed, first_coeff=0, stream=None):
        p, n = self._dev(t)
        if n % self.limbs:
            raise RingError("buffer is not a whole number of coefficients")
        self._check(self._lib.sr_fill_uniform_dev(self._ctx, seed, first_coeff, n // self.limbs, p, self._stream(stream)))
        return t

    def count_noncanonical_dev(self, t, stream=None):
        p, n = self._dev(t)
        c = ctypes.c_uint64()
        self._check(self._lib.sr_count_noncanonical_dev(self._ctx, p, n // self.limbs, ctypes.byref(c), self._stream(stream)))
        return c.value

    # -- twiddle sharing across GPUs (one RCCL broadcast at start-up) -------------------------
    def twiddle_block(self):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        self._check(self._lib.sr_ctx_twiddle_block(self._ctx, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def twiddles_updated(self):
        self._check(self._lib.sr_ctx_twiddles_updated(self._ctx))

    # -- per-kernel timing --------------------------------------------------------------------
    def profile_enable(self, on=True):
        """on: False / True (every launch bracketed by HIP events) or an int N >= 2 (every N-th launch only: leaves a two-lane step as
        it runs, see sr_ctx_profile_enable)"""
        self._check(self._lib.sr_ctx_profile_enable(self._ctx, int(on)))

    def profile_read(self):
        """per tag: ms and launches of the BRACKETED launches, seen = every launch that went by since the last read"""
        ms = (ctypes.c_double * len(PROF_TAGS))()
        n = (ctypes.c_uint64 * len(PROF_TAGS))()
        seen = (ctypes.c_uint64 * len(PROF_TAGS))()
        self._check(self._lib.sr_ctx_profile_read_sampled(self._ctx, ms, n, seen))
        return {t: {"ms": ms[i], "launches": int(n[i]), "seen": int(seen[i])} for i, t in enumerate(PROF_TAGS)}
