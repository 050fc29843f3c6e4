// stark_rings.hpp -- C++17 host-side mirror of the reference's ring interface for the CRT/NTT hot path,
// header-only, on top of the C ABI (stark_rings_hip.h).  The reference is Rust; no Rust toolchain exists in
// this image, so the host layer a crates/ring maintainer would write in Rust is restated
// here in C++ with the same names, argument meaning and error behaviour:
//
//   reference (crates/ring/src/cyclotomic_ring/)                     here
//   ---------------------------------------------------------------  ------------------------------------------
//   trait CyclotomicConfig<N>            ring_config.rs:11-35        class CyclotomicConfig  (one ring + degree)
//     reduce_in_place(&mut Vec<Fp>)      ring_config.rs:23             reduce_in_place(std::vector<Fp>&)
//     crt_in_place(&mut [Fp])            ring_config.rs:27             crt_in_place(Fp*, len)
//     icrt_in_place(&mut [Fp])           ring_config.rs:34             icrt_in_place(Fp*, len)
//     crt(Vec<Fp>) -> Vec<CRTField>      ring_config.rs:29             crt(std::vector<Fp>) (same allocation)
//     icrt(Vec<CRTField>) -> Vec<Fp>     ring_config.rs:30             icrt(std::vector<Fp>)
//     CRT_FIELD_EXTENSION_DEGREE         ring_config.rs:19             crt_field_extension_degree()
//   CyclotomicPolyRingGeneral<C,N,D>     coeff_form.rs:31-33         class RqPolyVec   (a Vec<RqPoly>, flat)
//     Mul  (poly_mul + reduce)           coeff_form.rs:54-67,250-258   operator*, operator*=
//   CyclotomicPolyRingNTTGeneral<C,N,D>  ntt_form.rs:25-27           class RqNTTVec    (a Vec<RqNTT>, flat)
//     Mul / MulAssign (slot-wise)        ntt_form.rs:159-225           operator*, operator*=
//     Add / Sub                          ntt_form.rs:227-285,588-638   operator+=, operator-=  (both forms)
//   CRT::elementwise_crt                 crt.rs:10-25                RqPolyVec::elementwise_crt() &&  (in place, same allocation)
//   ICRT::elementwise_icrt               crt.rs:34-49                RqNTTVec::elementwise_icrt() &&
//   Flatten::flatten_to_coeffs           flatten.rs:11-17            flatten_to_coeffs(Vec&&) -> std::vector<uint64_t>
//   Flatten::promote_from_coeffs         flatten.rs:19-33            promote_from_coeffs(...) -> std::optional (nullopt if len % D != 0)
//   Matrix<RqNTT> (linear_algebra)       matrix.rs:14-20             class MatrixNTT        (dense, row-major, flat)
//     checked_mul_vec / try_mul_vec      matrix.rs:168-183             checked_mul_vec -> std::optional (nullopt: DifferentLengths), try_mul_vec throws
//     checked_mul_mat                    matrix.rs:148-166             checked_mul_mat -> std::optional
//     MulAssign<&R> (Matrix, Sparse)     matrix.rs:207-211,            operator*=(const RqNTTVec &one_element): every entry times one ring element
//                                        sparse_matrix.rs:303-307
//   GadgetDecompose / GadgetRecompose    balanced_decomposition/     gadget_decompose(const RqPolyVec&, b, k) / gadget_recompose(...)
//     for &[R] / Vec<R>                  mod.rs:163-206                (digit j of element e = element e * k + j; throws where it panics)
//   SparseMatrix<RqNTT>                  sparse_matrix.rs:17-22      class SparseMatrixNTT  (coeffs: rows of (element, column))
//     checked_mul_vec / try_mul_vec      sparse_matrix.rs:201-216      same contract; an out-of-range column throws (the reference panics)
//   CanonicalSerialize / Deserialize     coeff_form.rs:154-189,      serialize_compressed(vec / matrix) -> bytes, deserialize_*(cfg, bytes):
//     for ring elements, Vec, Matrix,    ntt_form.rs:24,               the element bytes come from the device codec (sr_serialize_batch), the
//     SparseMatrix                       matrix.rs:111-145,            u64 length words and (R, usize) pairs are written here; InvalidData and
//                                        sparse_matrix.rs:158-200      short input throw
//   monomial / unit_monomial / psi /     crates/ring/src/            same names; BaseRing scalars are passed as their signed representative
//     exp / exp_signed / psi_range_check monomial.rs:17-93             (Zq::center and sign, ring.rs:160-181) in an int64_t
//   decompose basis `b: u128`            balanced_decomposition/     every gadget function takes `u128` (unsigned __int128); 2^127 and above
//                                        mod.rs:62,73                  is refused for decomposition (negative after the reference's `as i128`)
//   DecomposeToVec for &[R] / Vec<R>     mod.rs:119-161              decompose_to_vec(v, b, k) -> std::vector<RqPolyVec> (one Vec<R> per element)
//   GadgetDecompose / GadgetRecompose    mod.rs:208-352              class MatrixPoly / SparseMatrixPoly: gadget_decompose (n x m -> n x k m;
//     for &[(R, usize)], Matrix<R>,                                    sparse entries (e, c) -> (digit_i, c k + i), zero digits dropped),
//     SparseMatrix<R>                                                  gadget_recompose; sparse_row_gadget_decompose / _recompose for one row
//   Cyclotomic::rot / into_rot_iter /    traits.rs:54-91             rot(RqPolyVec&) (every element times X), Rotation, into_rot_iter(v)
//     Rotation
//   (one context per GPU of a node)      SURVEY 8b / 8e              class ContextGroup: sr_ctx_create_group + sr_shard_range
//   (device-resident batches)            --                          CyclotomicConfig::*_dev(device pointers, stream), reserve_scratch,
//                                                                      plan_in_use, and the packed-u32 BabyBear forms (*_packed32_dev)
//
// What is deliberately narrowed: ring elements of degree 2^16 are 512 KiB, so the per-element `Copy` value type of
// the reference (ring.rs:13-15) is not mirrored; the drop-in seam is the batch (hard part 4).
// Errors: the reference panics (assert_eq!(len, D): goldilocks/ntt.rs:136, stark_prime/ntt.rs:122; "Wrong length":
// coeff_form.rs:39); here std::length_error / std::runtime_error are thrown.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "stark_rings_hip.h"

namespace stark_rings {

class CyclotomicConfig {
public:
    // plan: the kernel plan of the context (sr_plan), nullptr = defaults; fixed for the life of the context
    CyclotomicConfig(sr_ring ring, int log2_degree = 0, int device = 0, const sr_plan *plan = nullptr) : ring_(ring) {
        sr_ctx *c = nullptr;
        check(sr_ctx_create_ex(ring, log2_degree, device, plan, &c), "sr_ctx_create_ex");
        ctx_.reset(c, [](sr_ctx *p) { sr_ctx_destroy(p); });
        check(sr_ctx_degree(c, &degree_), "sr_ctx_degree");
        check(sr_ctx_limbs(c, &limbs_), "sr_ctx_limbs");
    }
    // adopts a context created elsewhere (ContextGroup); `owned` contexts are destroyed with the last copy
    CyclotomicConfig(sr_ring ring, sr_ctx *adopt) : ring_(ring) {
        ctx_.reset(adopt, [](sr_ctx *p) { sr_ctx_destroy(p); });
        check(sr_ctx_degree(adopt, &degree_), "sr_ctx_degree");
        check(sr_ctx_limbs(adopt, &limbs_), "sr_ctx_limbs");
    }
    sr_ring ring() const { return ring_; }
    size_t dimension() const { return degree_; }            // PolyRing::dimension()
    int limbs() const { return limbs_; }                    // N
    size_t words_per_elem() const { return degree_ * (size_t)limbs_; }
    size_t wire_coeff_bytes() const { return sr_wire_coeff_bytes(raw()); }   // bytes of one serialised coefficient
    size_t wire_elem_bytes() const { return degree_ * wire_coeff_bytes(); }
    int crt_field_extension_degree() const {
        return ring_ == SR_RING_GOLDILOCKS_24 ? 3 : (ring_ == SR_RING_BABYBEAR_72 ? 9 : (ring_ == SR_RING_FROG_16 ? 4 : 1));
    }
    sr_ctx *raw() const { return ctx_.get(); }

    // coefficients: up to 2*D coefficients of N limbs each -> reduced mod Phi, resized to D
    void reduce_in_place(std::vector<uint64_t> &coefficients) const {
        if (coefficients.size() % limbs_) throw std::length_error("reduce_in_place: not whole coefficients");
        const size_t in_len = coefficients.size() / limbs_;
        std::vector<uint64_t> out(words_per_elem());
        std::vector<uint64_t> dummy(1);
        check(sr_reduce_batch(raw(), in_len ? coefficients.data() : dummy.data(), in_len, out.data(), 1), "reduce_in_place");
        coefficients.swap(out);
    }
    void crt_in_place(uint64_t *coefficients, size_t len_words) const {
        if (len_words != words_per_elem()) throw std::length_error("crt_in_place: coefficients.len() != D");
        check(sr_ntt_fwd_batch(raw(), coefficients, 1), "crt_in_place");
    }
    void icrt_in_place(uint64_t *evaluations, size_t len_words) const {
        if (len_words != words_per_elem()) throw std::length_error("icrt_in_place: evaluations.len() != D");
        check(sr_ntt_inv_batch(raw(), evaluations, 1), "icrt_in_place");
    }
    std::vector<uint64_t> crt(std::vector<uint64_t> coefficients) const {
        crt_in_place(coefficients.data(), coefficients.size());
        return coefficients;
    }
    std::vector<uint64_t> icrt(std::vector<uint64_t> evaluations) const {
        icrt_in_place(evaluations.data(), evaluations.size());
        return evaluations;
    }
    static void check(int rc, const char *what) {
        if (rc != SR_OK) throw std::runtime_error(std::string(what) + ": " + sr_last_error_string());
    }

    // ---- device-resident batches: pointers are HIP device pointers, `stream` a hipStream_t (as void *: no HIP header needed here),
    //      every call is asynchronous on that stream (include/stark_rings_hip.h, "device-resident entry points") ----
    void reserve_scratch(size_t batch) const { check(sr_ctx_reserve_scratch(raw(), batch), "reserve_scratch"); }
    // the plan the library settled on (lanes: 0 = auto and not settled yet, 1 = one stream, 2 = two lanes); probe_ms / probe_elems optional
    sr_plan plan_in_use(double probe_ms[2] = nullptr, size_t *probe_elems = nullptr) const {
        sr_plan p{};
        check(sr_ctx_plan_in_use(raw(), &p, probe_ms, probe_elems), "plan_in_use");
        return p;
    }
    void elementwise_crt_dev(uint64_t *d, size_t batch, void *stream) const { check(sr_ntt_fwd_batch_dev(raw(), d, batch, stream), "elementwise_crt_dev"); }
    void elementwise_icrt_dev(uint64_t *d, size_t batch, void *stream) const { check(sr_ntt_inv_batch_dev(raw(), d, batch, stream), "elementwise_icrt_dev"); }
    void ntt_mul_dev(uint64_t *lhs, const uint64_t *rhs, size_t batch, void *stream) const {
        check(sr_pointwise_mul_batch_dev(raw(), lhs, rhs, batch, stream), "ntt_mul_dev");
    }
    void add_dev(uint64_t *lhs, const uint64_t *rhs, size_t batch, void *stream) const { check(sr_add_batch_dev(raw(), lhs, rhs, batch, stream), "add_dev"); }
    void sub_dev(uint64_t *lhs, const uint64_t *rhs, size_t batch, void *stream) const { check(sr_sub_batch_dev(raw(), lhs, rhs, batch, stream), "sub_dev"); }
    // Neg, Mul<scalar>, Add<scalar> (coeff_form.rs:270-278, 390-408, 610-700; ntt_form.rs:191-203, 373-505); scalar: host pointer to the
    // Montgomery image of the base-field element
    void neg_dev(uint64_t *d, size_t batch, void *stream) const { check(sr_neg_batch_dev(raw(), d, batch, stream), "neg_dev"); }
    void mul_elem_dev(uint64_t *d, const uint64_t *d_elem, size_t batch, void *stream) const { check(sr_mul_elem_batch_dev(raw(), d, d_elem, batch, stream), "mul_elem_dev"); }
    // Sum / Product over a device-resident slice (coeff_form.rs:507-537, ntt_form.rs:640-670); out = one ring element outside the slice
    void sum_dev(uint64_t *out, const uint64_t *in, size_t n, void *stream) const { check(sr_sum_batch_dev(raw(), out, in, n, stream), "sum_dev"); }
    void product_dev(uint64_t *out, const uint64_t *in_ntt, size_t n, void *stream) const { check(sr_product_batch_dev(raw(), out, in_ntt, n, stream), "product_dev"); }
    void scale_dev(uint64_t *d, const uint64_t *scalar, size_t batch, void *stream) const { check(sr_scale_batch_dev(raw(), d, scalar, batch, stream), "scale_dev"); }
    void add_scalar_dev(uint64_t *d, const uint64_t *scalar, bool ntt_form, size_t batch, void *stream) const {
        check(sr_add_scalar_batch_dev(raw(), d, scalar, ntt_form ? 1 : 0, batch, stream), "add_scalar_dev");
    }
    // out = a * b (RqPoly * &RqPoly); a, b only read; out may alias a
    void mul_dev(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, void *stream) const {
        check(sr_ring_mul_batch_dev(raw(), out, a, b, batch, stream), "mul_dev");
    }
    void mul_ntt_rhs_dev(uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch, void *stream) const {
        check(sr_ring_mul_ntt_rhs_batch_dev(raw(), out, a, b_ntt, batch, stream), "mul_ntt_rhs_dev");
    }
    void rot_dev(uint64_t *out, const uint64_t *in, size_t batch, void *stream) const { check(sr_rot_batch_dev(raw(), out, in, batch, stream), "rot_dev"); }
    void matvec_dev(uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols, void *stream) const {
        check(sr_matvec_ntt_dev(raw(), y, m, v, nrows, ncols, stream), "matvec_dev");
    }
    void matmul_dev(uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p, void *stream) const {
        check(sr_matmul_ntt_dev(raw(), y, a, b, n, m, p, stream), "matmul_dev");
    }
    void gadget_decompose_dev(uint64_t *out, const uint64_t *in, unsigned __int128 b, size_t padding_size, size_t batch, void *stream) const {
        check(sr_decompose_balanced_batch_wide_dev(raw(), out, in, (uint64_t)b, (uint64_t)(b >> 64), padding_size, batch, stream), "gadget_decompose_dev");
    }
    void gadget_recompose_dev(uint64_t *out, const uint64_t *in, unsigned __int128 b, size_t padding_size, size_t batch_out, void *stream) const {
        check(sr_recompose_batch_wide_dev(raw(), out, in, (uint64_t)b, (uint64_t)(b >> 64), padding_size, batch_out, stream), "gadget_recompose_dev");
    }
    // packed-u32 BabyBear boundary (the low half of the reference's Fp64 limb, babybear/mod.rs:18-26)
    void pack32_dev(uint32_t *out, const uint64_t *in, size_t batch, void *stream) const { check(sr_pack32_batch_dev(raw(), out, in, batch, stream), "pack32_dev"); }
    void unpack32_dev(uint64_t *out, const uint32_t *in, size_t batch, void *stream) const { check(sr_unpack32_batch_dev(raw(), out, in, batch, stream), "unpack32_dev"); }
    void elementwise_crt_packed32_dev(uint32_t *d, size_t batch, void *stream) const { check(sr_ntt_fwd_packed32_batch_dev(raw(), d, batch, stream), "crt_packed32"); }
    void elementwise_icrt_packed32_dev(uint32_t *d, size_t batch, void *stream) const { check(sr_ntt_inv_packed32_batch_dev(raw(), d, batch, stream), "icrt_packed32"); }
    void mul_packed32_dev(uint32_t *out, const uint32_t *a, const uint32_t *b, size_t batch, void *stream) const {
        check(sr_ring_mul_packed32_batch_dev(raw(), out, a, b, batch, stream), "mul_packed32_dev");
    }

private:
    sr_ring ring_;
    std::shared_ptr<sr_ctx> ctx_;
    size_t degree_ = 0;
    int limbs_ = 1;
};

class RqNTTVec;

// Vec<RqPoly>: `len` ring elements in coefficient form, flat element-major storage (flatten.rs layout)
class RqPolyVec {
public:
    RqPolyVec(CyclotomicConfig cfg, std::vector<uint64_t> words) : cfg_(std::move(cfg)), w_(std::move(words)) {
        if (w_.size() % cfg_.words_per_elem()) throw std::length_error("Wrong length");  // coeff_form.rs:39
    }
    size_t len() const { return w_.size() / cfg_.words_per_elem(); }
    const std::vector<uint64_t> &words() const { return w_; }
    const CyclotomicConfig &config() const { return cfg_; }
    uint64_t *element(size_t i) { return w_.data() + i * cfg_.words_per_elem(); }
    bool operator==(const RqPolyVec &o) const { return w_ == o.w_; }

    RqNTTVec elementwise_crt() &&;                               // crt.rs:10-25
    RqPolyVec &operator*=(const RqPolyVec &rhs) {                // coeff_form.rs:250-258 per element
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_ring_mul_batch(cfg_.raw(), w_.data(), w_.data(), rhs.w_.data(), len()), "RqPoly *");
        return *this;
    }
    // *this = icrt(crt(*this) (.) rhs_ntt words): the constant-operand product (rhs kept in NTT form); defined below RqNTTVec
    RqPolyVec &mul_assign_ntt_rhs(const std::vector<uint64_t> &rhs_ntt_words) {
        if (rhs_ntt_words.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_ring_mul_ntt_rhs_batch(cfg_.raw(), w_.data(), w_.data(), rhs_ntt_words.data(), len()),
                                "RqPoly * RqNTT");
        return *this;
    }
    friend RqPolyVec operator*(RqPolyVec lhs, const RqPolyVec &rhs) {
        lhs *= rhs;
        return lhs;
    }
    RqPolyVec &operator+=(const RqPolyVec &rhs) {                // coeff_form.rs Add<&Self>
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_add_batch(cfg_.raw(), w_.data(), rhs.w_.data(), len()), "RqPoly +=");
        return *this;
    }
    RqPolyVec &operator-=(const RqPolyVec &rhs) {
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_sub_batch(cfg_.raw(), w_.data(), rhs.w_.data(), len()), "RqPoly -=");
        return *this;
    }
    RqPolyVec operator-() && {                                   // coeff_form.rs:270-278
        CyclotomicConfig::check(sr_neg_batch(cfg_.raw(), w_.data(), len()), "-RqPoly");
        return std::move(*this);
    }
    // scalar: the Montgomery image of one base-field element (limbs() words) -- Mul<Fp>, Mul<u64> ... (coeff_form.rs:390-408, 610-650)
    RqPolyVec &operator*=(const std::vector<uint64_t> &scalar) {
        if (scalar.size() != (size_t)cfg_.limbs()) throw std::length_error("scalar: wrong number of limbs");
        CyclotomicConfig::check(sr_scale_batch(cfg_.raw(), w_.data(), scalar.data(), len()), "RqPoly *= scalar");
        return *this;
    }
    RqPolyVec &operator+=(const std::vector<uint64_t> &scalar) {  // coefficient 0 of every element (coeff_form.rs:652-700)
        if (scalar.size() != (size_t)cfg_.limbs()) throw std::length_error("scalar: wrong number of limbs");
        CyclotomicConfig::check(sr_add_scalar_batch(cfg_.raw(), w_.data(), scalar.data(), 0, len()), "RqPoly += scalar");
        return *this;
    }
    // impl Sum<&Self> (coeff_form.rs:507-521): `iter.fold(Self::zero(), |acc, x| acc + x)` -> a vector of ONE element (zero() when empty)
    RqPolyVec sum() const {
        std::vector<uint64_t> out(cfg_.words_per_elem());
        const uint64_t dummy = 0;
        CyclotomicConfig::check(sr_sum_batch(cfg_.raw(), out.data(), w_.empty() ? &dummy : w_.data(), len()), "RqPoly sum");
        return RqPolyVec(cfg_, std::move(out));
    }
    RqPolyVec product() const;  // impl Product<&Self> (coeff_form.rs:523-537); defined below RqNTTVec
    std::vector<uint64_t> into_words() && { return std::move(w_); }

private:
    CyclotomicConfig cfg_;
    std::vector<uint64_t> w_;
};

// Vec<RqNTT>: the same bytes reinterpreted as CRT slots (crt.rs:61 transmute)
class RqNTTVec {
public:
    RqNTTVec(CyclotomicConfig cfg, std::vector<uint64_t> words) : cfg_(std::move(cfg)), w_(std::move(words)) {
        if (w_.size() % cfg_.words_per_elem()) throw std::length_error("Should be of correct length");  // ntt_form.rs:703
    }
    size_t len() const { return w_.size() / cfg_.words_per_elem(); }
    const std::vector<uint64_t> &words() const { return w_; }
    const CyclotomicConfig &config() const { return cfg_; }
    bool operator==(const RqNTTVec &o) const { return w_ == o.w_; }

    RqPolyVec elementwise_icrt() && {                            // crt.rs:34-49
        CyclotomicConfig::check(sr_ntt_inv_batch(cfg_.raw(), w_.data(), len()), "elementwise_icrt");
        return RqPolyVec(cfg_, std::move(w_));
    }
    RqNTTVec &operator*=(const RqNTTVec &rhs) {                  // ntt_form.rs:213-225
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_pointwise_mul_batch(cfg_.raw(), w_.data(), rhs.w_.data(), len()), "RqNTT *=");
        return *this;
    }
    friend RqNTTVec operator*(RqNTTVec lhs, const RqNTTVec &rhs) {
        lhs *= rhs;
        return lhs;
    }
    RqNTTVec &operator+=(const RqNTTVec &rhs) {                  // ntt_form.rs:227-285
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_add_batch(cfg_.raw(), w_.data(), rhs.w_.data(), len()), "RqNTT +=");
        return *this;
    }
    RqNTTVec &operator-=(const RqNTTVec &rhs) {                  // ntt_form.rs:588-638
        if (rhs.w_.size() != w_.size()) throw std::length_error("operand lengths differ");
        CyclotomicConfig::check(sr_sub_batch(cfg_.raw(), w_.data(), rhs.w_.data(), len()), "RqNTT -=");
        return *this;
    }
    RqNTTVec operator-() && {                                    // ntt_form.rs:191-203
        CyclotomicConfig::check(sr_neg_batch(cfg_.raw(), w_.data(), len()), "-RqNTT");
        return std::move(*this);
    }
    RqNTTVec &operator*=(const std::vector<uint64_t> &scalar) {  // `*lhs *= BaseCRTField::from(rhs)` (ntt_form.rs:373-425)
        if (scalar.size() != (size_t)cfg_.limbs()) throw std::length_error("scalar: wrong number of limbs");
        CyclotomicConfig::check(sr_scale_batch(cfg_.raw(), w_.data(), scalar.data(), len()), "RqNTT *= scalar");
        return *this;
    }
    // every element of the vector times ONE ring element r (r.len() == 1), slot-wise: the loop body of `Matrix<R> *= &R`
    RqNTTVec &mul_assign_elem(const RqNTTVec &r) {
        if (r.len() != 1) throw std::length_error("mul_assign_elem: the multiplier is not one ring element");
        CyclotomicConfig::check(sr_mul_elem_batch(cfg_.raw(), w_.empty() ? dummy_word() : w_.data(), r.w_.data(), len()), "RqNTT[] *= &RqNTT");
        return *this;
    }
    RqNTTVec &operator+=(const std::vector<uint64_t> &scalar) {  // component 0 of every slot (ntt_form.rs:427-505)
        if (scalar.size() != (size_t)cfg_.limbs()) throw std::length_error("scalar: wrong number of limbs");
        CyclotomicConfig::check(sr_add_scalar_batch(cfg_.raw(), w_.data(), scalar.data(), 1, len()), "RqNTT += scalar");
        return *this;
    }
    // impl Sum<&Self> / Product<&Self> (ntt_form.rs:640-670): folds over the slice -> a vector of ONE element (zero() / one() when empty)
    RqNTTVec sum() const {
        std::vector<uint64_t> out(cfg_.words_per_elem());
        CyclotomicConfig::check(sr_sum_batch(cfg_.raw(), out.data(), w_.empty() ? dummy_word() : w_.data(), len()), "RqNTT sum");
        return RqNTTVec(cfg_, std::move(out));
    }
    RqNTTVec product() const {
        std::vector<uint64_t> out(cfg_.words_per_elem());
        CyclotomicConfig::check(sr_product_batch(cfg_.raw(), out.data(), w_.empty() ? dummy_word() : w_.data(), len()), "RqNTT product");
        return RqNTTVec(cfg_, std::move(out));
    }
    std::vector<uint64_t> into_words() && { return std::move(w_); }

private:
    static uint64_t *dummy_word() {
        static uint64_t z = 0;
        return &z;
    }
    CyclotomicConfig cfg_;
    std::vector<uint64_t> w_;
};

// `iter.fold(Self::one(), |acc, x| acc * x)` with the ring product = icrt(product(crt(x_i))): the CRT is a ring isomorphism
inline RqPolyVec RqPolyVec::product() const {
    return RqPolyVec(*this).elementwise_crt().product().elementwise_icrt();
}
inline RqNTTVec RqPolyVec::elementwise_crt() && {
    CyclotomicConfig::check(sr_ntt_fwd_batch(cfg_.raw(), w_.data(), len()), "elementwise_crt");
    return RqNTTVec(cfg_, std::move(w_));
}

// GadgetDecompose for &[R] / Vec<R> (balanced_decomposition/mod.rs:163-175, 192-198): len * padding_size elements, digit j of
// element e at index e * padding_size + j.  The reference panics on basis 0, 1 or odd and when padding_size digits do not
// suffice; std::runtime_error here.
typedef unsigned __int128 u128;  // the reference's basis type (decompose_balanced_in_place(v, b: u128, ..), mod.rs:62)
inline RqPolyVec gadget_decompose(const RqPolyVec &v, u128 b, size_t padding_size) {
    const CyclotomicConfig &cfg = v.config();
    std::vector<uint64_t> out(v.len() * padding_size * cfg.words_per_elem());
    std::vector<uint64_t> dummy(1);
    CyclotomicConfig::check(sr_decompose_balanced_batch_wide(cfg.raw(), out.empty() ? dummy.data() : out.data(),
                                                             v.words().empty() ? dummy.data() : v.words().data(), (uint64_t)b,
                                                             (uint64_t)(b >> 64), padding_size, v.len()),
                            "gadget_decompose");
    return RqPolyVec(cfg, std::move(out));
}
// GadgetRecompose (mod.rs:177-189, 200-206): len / padding_size elements
inline RqPolyVec gadget_recompose(const RqPolyVec &digits, u128 b, size_t padding_size) {
    const CyclotomicConfig &cfg = digits.config();
    if (padding_size == 0 || digits.len() % padding_size) throw std::length_error("length is not a multiple of padding_size");
    const size_t n = digits.len() / padding_size;
    std::vector<uint64_t> out(n * cfg.words_per_elem());
    std::vector<uint64_t> dummy(1);
    CyclotomicConfig::check(sr_recompose_batch_wide(cfg.raw(), out.empty() ? dummy.data() : out.data(),
                                                    digits.words().empty() ? dummy.data() : digits.words().data(), (uint64_t)b,
                                                    (uint64_t)(b >> 64), padding_size, n),
                            "gadget_recompose");
    return RqPolyVec(cfg, std::move(out));
}
// DecomposeToVec for &[R] / Vec<R> (mod.rs:119-161): one Vec<R> of padding_size digits per element of v
inline std::vector<RqPolyVec> decompose_to_vec(const RqPolyVec &v, u128 b, size_t padding_size) {
    const CyclotomicConfig &cfg = v.config();
    const RqPolyVec all = gadget_decompose(v, b, padding_size);   // digit j of element e at e * padding_size + j
    const size_t w = cfg.words_per_elem();
    std::vector<RqPolyVec> out;
    out.reserve(v.len());
    for (size_t e = 0; e < v.len(); e++)
        out.emplace_back(cfg, std::vector<uint64_t>(all.words().begin() + e * padding_size * w, all.words().begin() + (e + 1) * padding_size * w));
    return out;
}

// Cyclotomic::rot (traits.rs:54-66): every element of the batch times X modulo the ring, in place
inline void rot(RqPolyVec &v) {
    if (v.len() == 0) return;
    CyclotomicConfig::check(sr_rot_batch(v.config().raw(), v.element(0), v.len()), "rot");
}
// Rotation / into_rot_iter (traits.rs:68-91): an endless iterator x, X x, X^2 x, ... (next() returns the current value, then rotates)
class Rotation {
public:
    explicit Rotation(RqPolyVec curr) : curr_(std::move(curr)) {}
    RqPolyVec next() {
        RqPolyVec out = curr_;
        rot(curr_);
        return out;
    }

private:
    RqPolyVec curr_;
};
inline Rotation into_rot_iter(RqPolyVec v) { return Rotation(std::move(v)); }

// GadgetDecompose / GadgetRecompose for &[(R, usize)] (mod.rs:208-275): one sparse row of (element, column) pairs
using SparseRow = std::vector<std::pair<std::vector<uint64_t>, size_t>>;
inline SparseRow sparse_row_gadget_decompose(const CyclotomicConfig &cfg, const SparseRow &row, u128 b, size_t padding_size) {
    const size_t w = cfg.words_per_elem();
    std::vector<uint64_t> flat;
    for (const auto &e : row) {
        if (e.first.size() != w) throw std::length_error("Wrong length");
        flat.insert(flat.end(), e.first.begin(), e.first.end());
    }
    const RqPolyVec digits = gadget_decompose(RqPolyVec(cfg, std::move(flat)), b, padding_size);
    SparseRow out;
    for (size_t i = 0; i < row.size(); i++)
        for (size_t j = 0; j < padding_size; j++) {
            const uint64_t *d = digits.words().data() + (i * padding_size + j) * w;
            bool zero = true;
            for (size_t q = 0; q < w && zero; q++) zero = d[q] == 0;
            if (!zero) out.emplace_back(std::vector<uint64_t>(d, d + w), row[i].second * padding_size + j);  // "Maintain full sparsity"
        }
    return out;
}
inline SparseRow sparse_row_gadget_recompose(const CyclotomicConfig &cfg, const SparseRow &row, u128 b, size_t padding_size) {
    if (padding_size == 0) throw std::length_error("padding_size is zero");
    const size_t w = cfg.words_per_elem();
    // consecutive entries with the same index / padding_size form one chunk (mod.rs:241-256); missing digits are zero
    std::vector<size_t> index;
    std::vector<uint64_t> flat;
    for (size_t i = 0; i < row.size(); i++) {
        const size_t idx = row[i].second / padding_size;
        if (i == 0 || idx != row[i - 1].second / padding_size) {
            index.push_back(idx);
            flat.resize(flat.size() + padding_size * w, 0);
        }
        if (row[i].first.size() != w) throw std::length_error("Wrong length");
        std::copy(row[i].first.begin(), row[i].first.end(), flat.end() - padding_size * w + (row[i].second % padding_size) * w);
    }
    const RqPolyVec vals = gadget_recompose(RqPolyVec(cfg, std::move(flat)), b, padding_size);
    SparseRow out;
    for (size_t i = 0; i < index.size(); i++)
        out.emplace_back(std::vector<uint64_t>(vals.words().begin() + i * w, vals.words().begin() + (i + 1) * w), index[i]);
    return out;
}

// Matrix<R> in coefficient form (the decompositions work coefficient-wise): GadgetDecompose / GadgetRecompose for Matrix<R>
// (mod.rs:276-313): n x m -> n x (k m), row by row; entry (r, c) -> (r, c k .. c k + k - 1)
class MatrixPoly {
public:
    MatrixPoly(CyclotomicConfig cfg, size_t nrows, size_t ncols, std::vector<uint64_t> words)
        : cfg_(std::move(cfg)), nrows_(nrows), ncols_(ncols), w_(std::move(words)) {
        if (w_.size() != nrows_ * ncols_ * cfg_.words_per_elem()) throw std::length_error("Wrong length");
    }
    size_t nrows() const { return nrows_; }
    size_t ncols() const { return ncols_; }
    const std::vector<uint64_t> &words() const { return w_; }
    bool operator==(const MatrixPoly &o) const { return nrows_ == o.nrows_ && ncols_ == o.ncols_ && w_ == o.w_; }
    MatrixPoly gadget_decompose(u128 b, size_t padding_size) const {   // row-major storage: the flat batch IS row after row
        RqPolyVec d = stark_rings::gadget_decompose(RqPolyVec(cfg_, w_), b, padding_size);
        return MatrixPoly(cfg_, nrows_, ncols_ * padding_size, std::move(d).into_words());
    }
    MatrixPoly gadget_recompose(u128 b, size_t padding_size) const {
        if (padding_size == 0 || ncols_ % padding_size) throw std::length_error("ncols is not a multiple of padding_size");
        RqPolyVec r = stark_rings::gadget_recompose(RqPolyVec(cfg_, w_), b, padding_size);
        return MatrixPoly(cfg_, nrows_, ncols_ / padding_size, std::move(r).into_words());
    }

private:
    CyclotomicConfig cfg_;
    size_t nrows_, ncols_;
    std::vector<uint64_t> w_;
};
// SparseMatrix<R> in coefficient form: GadgetDecompose / GadgetRecompose for SparseMatrix<R> (mod.rs:315-352)
class SparseMatrixPoly {
public:
    SparseMatrixPoly(CyclotomicConfig cfg, size_t nrows, size_t ncols, std::vector<SparseRow> coeffs)
        : cfg_(std::move(cfg)), nrows_(nrows), ncols_(ncols), coeffs_(std::move(coeffs)) {
        if (coeffs_.size() != nrows_) throw std::length_error("Wrong length");
    }
    size_t nrows() const { return nrows_; }
    size_t ncols() const { return ncols_; }
    const std::vector<SparseRow> &coeffs() const { return coeffs_; }
    SparseMatrixPoly gadget_decompose(u128 b, size_t padding_size) const {
        std::vector<SparseRow> rows;
        for (const auto &r : coeffs_) rows.push_back(sparse_row_gadget_decompose(cfg_, r, b, padding_size));
        return SparseMatrixPoly(cfg_, nrows_, ncols_ * padding_size, std::move(rows));
    }
    SparseMatrixPoly gadget_recompose(u128 b, size_t padding_size) const {
        std::vector<SparseRow> rows;
        for (const auto &r : coeffs_) rows.push_back(sparse_row_gadget_recompose(cfg_, r, b, padding_size));
        return SparseMatrixPoly(cfg_, nrows_, ncols_ / padding_size, std::move(rows));
    }

private:
    CyclotomicConfig cfg_;
    size_t nrows_, ncols_;
    std::vector<SparseRow> coeffs_;
};

// One context per GPU of a node from ONE process (what a Rust host that drives all 8 GPUs binds): sr_ctx_create_group builds the
// twiddle block on device_ids[0] and peer-copies it to the others (the only inter-GPU traffic of the path); shard_range is the
// contiguous, balanced batch partition of SURVEY 8e (the first batch % n parts get one element more).
class ContextGroup {
public:
    ContextGroup(sr_ring ring, int log2_degree, const std::vector<int> &device_ids, const sr_plan *plan = nullptr) {
        std::vector<sr_ctx *> raw(device_ids.size(), nullptr);
        CyclotomicConfig::check(sr_ctx_create_group(ring, log2_degree, device_ids.data(), (int)device_ids.size(), plan, raw.data()),
                                "sr_ctx_create_group");
        for (sr_ctx *c : raw) cfgs_.emplace_back(ring, c);
    }
    size_t size() const { return cfgs_.size(); }
    const CyclotomicConfig &operator[](size_t i) const { return cfgs_.at(i); }
    // elements [first, first + count) of a batch belong to context i
    std::pair<size_t, size_t> shard_range(size_t batch, size_t i) const {
        size_t first = 0, count = 0;
        CyclotomicConfig::check(sr_shard_range(batch, (int)cfgs_.size(), (int)i, &first, &count), "sr_shard_range");
        return {first, count};
    }

private:
    std::vector<CyclotomicConfig> cfgs_;
};

// Matrix<RqNTT> (crates/linear_algebra/src/matrix.rs): nrows x ncols ring elements in CRT/NTT form, row-major, flat
class MatrixNTT {
public:
    MatrixNTT(CyclotomicConfig cfg, size_t nrows, size_t ncols, std::vector<uint64_t> words)
        : cfg_(std::move(cfg)), nrows_(nrows), ncols_(ncols), w_(std::move(words)) {
        if (w_.size() != nrows_ * ncols_ * cfg_.words_per_elem()) throw std::length_error("Wrong length");
    }
    size_t nrows() const { return nrows_; }
    size_t ncols() const { return ncols_; }
    const std::vector<uint64_t> &words() const { return w_; }
    // matrix.rs:168-178: None when ncols != v.len()
    std::optional<RqNTTVec> checked_mul_vec(const RqNTTVec &v) const {
        if (v.len() != ncols_) return std::nullopt;
        std::vector<uint64_t> y(nrows_ * cfg_.words_per_elem());
        std::vector<uint64_t> dummy(1);
        CyclotomicConfig::check(sr_matvec_ntt(cfg_.raw(), y.empty() ? dummy.data() : y.data(), w_.empty() ? dummy.data() : w_.data(),
                                              v.words().empty() ? dummy.data() : v.words().data(), nrows_, ncols_),
                                "Matrix * vec");
        return RqNTTVec(cfg_, std::move(y));
    }
    RqNTTVec try_mul_vec(const RqNTTVec &v) const {  // matrix.rs:180-183: AlgebraError::DifferentLengths
        auto r = checked_mul_vec(v);
        if (!r) throw std::length_error("DifferentLengths");
        return std::move(*r);
    }
    std::optional<MatrixNTT> checked_mul_mat(const MatrixNTT &m) const {  // matrix.rs:148-166
        if (ncols_ != m.nrows_) return std::nullopt;
        std::vector<uint64_t> y(nrows_ * m.ncols_ * cfg_.words_per_elem());
        std::vector<uint64_t> dummy(1);
        CyclotomicConfig::check(sr_matmul_ntt(cfg_.raw(), y.empty() ? dummy.data() : y.data(), w_.empty() ? dummy.data() : w_.data(),
                                              m.w_.empty() ? dummy.data() : m.w_.data(), nrows_, ncols_, m.ncols_),
                                "Matrix * Matrix");
        return MatrixNTT(cfg_, nrows_, m.ncols_, std::move(y));
    }
    MatrixNTT &operator*=(const RqNTTVec &r) {  // MulAssign<&R> for Matrix<R> (matrix.rs:207-211): every entry *= r
        if (r.len() != 1) throw std::length_error("Matrix *= &R: the multiplier is not one ring element");
        std::vector<uint64_t> dummy(1);
        CyclotomicConfig::check(sr_mul_elem_batch(cfg_.raw(), w_.empty() ? dummy.data() : w_.data(), r.words().data(), nrows_ * ncols_), "Matrix *= &R");
        return *this;
    }

private:
    CyclotomicConfig cfg_;
    size_t nrows_, ncols_;
    std::vector<uint64_t> w_;
};

// SparseMatrix<RqNTT> (crates/linear_algebra/src/sparse_matrix.rs:17-22): coeffs[r] = the stored (element, column) pairs of row r
class SparseMatrixNTT {
public:
    using Entry = std::pair<std::vector<uint64_t>, size_t>;
    SparseMatrixNTT(CyclotomicConfig cfg, size_t nrows, size_t ncols, std::vector<std::vector<Entry>> coeffs)
        : cfg_(std::move(cfg)), nrows_(nrows), ncols_(ncols) {
        if (coeffs.size() != nrows_) throw std::length_error("Wrong length");
        row_ptr_.assign(nrows_ + 1, 0);
        for (size_t r = 0; r < nrows_; r++) {
            row_ptr_[r + 1] = row_ptr_[r] + coeffs[r].size();
            for (auto &e : coeffs[r]) {
                if (e.first.size() != cfg_.words_per_elem()) throw std::length_error("Wrong length");
                vals_.insert(vals_.end(), e.first.begin(), e.first.end());
                cols_.push_back((uint32_t)e.second);
                if (e.second > 0xFFFFFFFFull) throw std::length_error("column index does not fit 32 bits");
            }
        }
    }
    size_t nrows() const { return nrows_; }
    size_t ncols() const { return ncols_; }
    std::optional<RqNTTVec> checked_mul_vec(const RqNTTVec &v) const {  // sparse_matrix.rs:201-211
        if (v.len() != ncols_) return std::nullopt;
        std::vector<uint64_t> y(nrows_ * cfg_.words_per_elem());
        std::vector<uint64_t> dummy(1);
        std::vector<uint32_t> dummy32(1);
        CyclotomicConfig::check(sr_spmv_ntt(cfg_.raw(), y.empty() ? dummy.data() : y.data(), vals_.empty() ? dummy.data() : vals_.data(),
                                            cols_.empty() ? dummy32.data() : cols_.data(), row_ptr_.data(),
                                            v.words().empty() ? dummy.data() : v.words().data(), nrows_, ncols_),
                                "SparseMatrix * vec");
        return RqNTTVec(cfg_, std::move(y));
    }
    RqNTTVec try_mul_vec(const RqNTTVec &v) const {  // sparse_matrix.rs:213-216
        auto r = checked_mul_vec(v);
        if (!r) throw std::length_error("DifferentLengths");
        return std::move(*r);
    }
    SparseMatrixNTT &operator*=(const RqNTTVec &r) {  // MulAssign<&R> for SparseMatrix<R> (sparse_matrix.rs:303-307): every stored entry *= r
        if (r.len() != 1) throw std::length_error("SparseMatrix *= &R: the multiplier is not one ring element");
        std::vector<uint64_t> dummy(1);
        CyclotomicConfig::check(sr_mul_elem_batch(cfg_.raw(), vals_.empty() ? dummy.data() : vals_.data(), r.words().data(), cols_.size()),
                                "SparseMatrix *= &R");
        return *this;
    }

private:
    CyclotomicConfig cfg_;
    size_t nrows_, ncols_;
    std::vector<uint64_t> vals_, row_ptr_;
    std::vector<uint32_t> cols_;
};

// ---- CanonicalSerialize / CanonicalDeserialize (ark-serialize; Compress and Validate make no difference for these types) ----
namespace wire_detail {
inline void put_u64(std::vector<uint8_t> &out, uint64_t v) {
    for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i)));
}
inline uint64_t get_u64(const std::vector<uint8_t> &in, size_t &pos) {
    if (pos + 8 > in.size()) throw std::runtime_error("deserialize: unexpected end of input");
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)in[pos + i] << (8 * i);
    pos += 8;
    return v;
}
// the elements alone (what [Fp; D] / RqNTT serialise to, coeff_form.rs:157-163), appended to out
inline void put_elems(const CyclotomicConfig &cfg, std::vector<uint8_t> &out, const uint64_t *words, size_t n_elems) {
    if (!n_elems) return;
    const size_t at = out.size();
    out.resize(at + n_elems * cfg.wire_elem_bytes());
    CyclotomicConfig::check(sr_serialize_batch(cfg.raw(), out.data() + at, words, n_elems), "serialize");
}
inline std::vector<uint64_t> get_elems(const CyclotomicConfig &cfg, const uint8_t *bytes, size_t n_elems) {
    std::vector<uint64_t> w(n_elems * cfg.words_per_elem());
    if (n_elems) CyclotomicConfig::check(sr_deserialize_batch(cfg.raw(), w.data(), bytes, n_elems), "deserialize");
    return w;
}
}  // namespace wire_detail

// Vec<RqPoly> / Vec<RqNTT>: u64 length, then the elements
template <class V>
inline std::vector<uint8_t> serialize_compressed(const V &v) {
    std::vector<uint8_t> out;
    wire_detail::put_u64(out, v.len());
    wire_detail::put_elems(v.config(), out, v.words().data(), v.len());
    return out;
}
template <class V>
inline V deserialize_vec(const CyclotomicConfig &cfg, const std::vector<uint8_t> &in, size_t *pos_io = nullptr) {
    size_t pos = pos_io ? *pos_io : 0;
    const uint64_t n = wire_detail::get_u64(in, pos);
    if (n > (in.size() - pos) / cfg.wire_elem_bytes()) throw std::runtime_error("deserialize: unexpected end of input");
    V v(cfg, wire_detail::get_elems(cfg, in.data() + pos, n));
    pos += n * cfg.wire_elem_bytes();
    if (pos_io) *pos_io = pos;
    return v;
}
// Matrix<RqNTT> = Vec<Vec<RqNTT>> (matrix.rs:111-124)
inline std::vector<uint8_t> serialize_compressed(const MatrixNTT &m, const CyclotomicConfig &cfg) {
    std::vector<uint8_t> out;
    wire_detail::put_u64(out, m.nrows());
    for (size_t r = 0; r < m.nrows(); r++) {
        wire_detail::put_u64(out, m.ncols());
        wire_detail::put_elems(cfg, out, m.words().data() + r * m.ncols() * cfg.words_per_elem(), m.ncols());
    }
    return out;
}
// matrix.rs:132-144: ncols is the first row's length; ragged rows (which the reference accepts and trips over later) throw
inline MatrixNTT deserialize_matrix(const CyclotomicConfig &cfg, const std::vector<uint8_t> &in) {
    size_t pos = 0;
    const uint64_t nrows = wire_detail::get_u64(in, pos);
    uint64_t ncols = 0;
    std::vector<uint64_t> words;
    for (uint64_t r = 0; r < nrows; r++) {
        RqNTTVec row = deserialize_vec<RqNTTVec>(cfg, in, &pos);
        if (r == 0) ncols = row.len();
        if (row.len() != ncols) throw std::runtime_error("deserialize: ragged matrix rows");
        words.insert(words.end(), row.words().begin(), row.words().end());
    }
    return MatrixNTT(cfg, nrows, ncols, std::move(words));
}
// SparseMatrix<RqNTT>: nrows, ncols, Vec<Vec<(RqNTT, usize)>> (sparse_matrix.rs:158-175)
inline std::vector<uint8_t> serialize_compressed(const CyclotomicConfig &cfg, size_t nrows, size_t ncols,
                                                 const std::vector<std::vector<SparseMatrixNTT::Entry>> &coeffs) {
    std::vector<uint8_t> out;
    wire_detail::put_u64(out, nrows);
    wire_detail::put_u64(out, ncols);
    wire_detail::put_u64(out, coeffs.size());
    for (const auto &row : coeffs) {
        wire_detail::put_u64(out, row.size());
        for (const auto &e : row) {
            if (e.first.size() != cfg.words_per_elem()) throw std::length_error("Wrong length");
            wire_detail::put_elems(cfg, out, e.first.data(), 1);
            wire_detail::put_u64(out, e.second);
        }
    }
    return out;
}
struct SparseParts {
    size_t nrows, ncols;
    std::vector<std::vector<SparseMatrixNTT::Entry>> coeffs;
};
inline SparseParts deserialize_sparse(const CyclotomicConfig &cfg, const std::vector<uint8_t> &in) {  // sparse_matrix.rs:183-199
    size_t pos = 0;
    SparseParts s;
    s.nrows = wire_detail::get_u64(in, pos);
    s.ncols = wire_detail::get_u64(in, pos);
    const uint64_t nlists = wire_detail::get_u64(in, pos);
    for (uint64_t r = 0; r < nlists; r++) {
        const uint64_t n = wire_detail::get_u64(in, pos);
        std::vector<SparseMatrixNTT::Entry> row;
        for (uint64_t j = 0; j < n; j++) {
            if (pos + cfg.wire_elem_bytes() > in.size()) throw std::runtime_error("deserialize: unexpected end of input");
            std::vector<uint64_t> e = wire_detail::get_elems(cfg, in.data() + pos, 1);
            pos += cfg.wire_elem_bytes();
            const uint64_t col = wire_detail::get_u64(in, pos);
            row.emplace_back(std::move(e), (size_t)col);
        }
        s.coeffs.push_back(std::move(row));
    }
    return s;
}

// ---- monomial helpers (crates/ring/src/monomial.rs:17-93), coefficient form, one element each ----------------------------
namespace monomial_detail {
// ring element from signed small integers: each coefficient's standard form is c mod p, written as wire bytes (p - |c| needs
// the modulus: a negative coefficient is produced on the device as 0 - |c|)
inline RqPolyVec from_signed(const CyclotomicConfig &cfg, const std::vector<int64_t> &coeffs) {
    const size_t d = cfg.dimension(), w = cfg.wire_coeff_bytes();
    if (coeffs.size() != d) throw std::length_error("Wrong length");
    std::vector<uint8_t> pos_b(d * w, 0), neg_b(d * w, 0);
    for (size_t i = 0; i < d; i++) {
        const uint64_t mag = coeffs[i] < 0 ? (uint64_t)0 - (uint64_t)coeffs[i] : (uint64_t)coeffs[i];
        std::vector<uint8_t> &dst = coeffs[i] < 0 ? neg_b : pos_b;
        for (size_t b = 0; b < w && b < 8; b++) dst[i * w + b] = (uint8_t)(mag >> (8 * b));
        if (w < 8 && (mag >> (8 * w))) throw std::runtime_error("coefficient magnitude does not fit the field");
    }
    RqPolyVec pos(cfg, wire_detail::get_elems(cfg, pos_b.data(), 1));
    RqPolyVec neg(cfg, wire_detail::get_elems(cfg, neg_b.data(), 1));
    pos -= neg;
    return pos;
}
}  // namespace monomial_detail

inline RqPolyVec monomial(const CyclotomicConfig &cfg, size_t i, int64_t coeff) {   // monomial.rs:17-21; index past D panics there
    if (i >= cfg.dimension()) throw std::out_of_range("monomial: index outside the ring's dimension");
    std::vector<int64_t> c(cfg.dimension(), 0);
    c[i] = coeff;
    return monomial_detail::from_signed(cfg, c);
}
inline RqPolyVec zero_monomial(const CyclotomicConfig &cfg) { return RqPolyVec(cfg, std::vector<uint64_t>(cfg.words_per_elem(), 0)); }
inline RqPolyVec unit_monomial(const CyclotomicConfig &cfg, size_t i) { return monomial(cfg, i, 1); }
inline RqPolyVec psi(const CyclotomicConfig &cfg) {                                   // monomial.rs:36-49
    const size_t d = cfg.dimension();
    std::vector<int64_t> c(d, 0);
    for (size_t i = 1; i < d / 2; i++) {
        c[i] += (int64_t)i;
        c[d - i] -= (int64_t)i;
    }
    return monomial_detail::from_signed(cfg, c);
}
// a: the signed representative of the scalar (sign(a), center(a) = |a|)
inline RqPolyVec exp(const CyclotomicConfig &cfg, int64_t a) {                        // monomial.rs:56-66
    const uint64_t mag = a < 0 ? (uint64_t)0 - (uint64_t)a : (uint64_t)a;
    if (a >= 0) return unit_monomial(cfg, mag);
    if (mag > cfg.dimension()) throw std::out_of_range("monomial: index outside the ring's dimension");
    return unit_monomial(cfg, cfg.dimension() - mag);
}
inline RqPolyVec exp_signed(const CyclotomicConfig &cfg, int64_t a) {                 // monomial.rs:72-78
    const uint64_t mag = a < 0 ? (uint64_t)0 - (uint64_t)a : (uint64_t)a;
    return monomial(cfg, mag, a < 0 ? -1 : 1);
}
// Ok(()) -> true, MonomialError::RangeCheck -> false                                  // monomial.rs:84-93
inline bool psi_range_check(const CyclotomicConfig &cfg, int64_t a) {
    RqPolyVec prod = psi(cfg) * exp(cfg, a);
    std::vector<int64_t> want(cfg.dimension(), 0);
    want[0] = a;
    RqPolyVec a_elem = monomial_detail::from_signed(cfg, want);
    for (int l = 0; l < cfg.limbs(); l++)
        if (prod.words()[l] != a_elem.words()[l]) return false;   // ct(): coefficient 0, compared in the memory image
    return true;
}

// Flatten (flatten.rs:10-34): the flat coefficient vector IS the batch's storage; both directions are moves.
template <class V>
inline std::vector<uint64_t> flatten_to_coeffs(V &&vec) {
    return std::move(vec).into_words();
}
template <class V>
inline std::optional<V> promote_from_coeffs(const CyclotomicConfig &cfg, std::vector<uint64_t> coeffs) {
    if (coeffs.size() % cfg.words_per_elem() != 0) return std::nullopt;  // flatten.rs:22-24
    return V(cfg, std::move(coeffs));
}

}  // namespace stark_rings
