/* stark_rings_hip.h -- C ABI of the MI355X (gfx950) backend for stark-rings' CRT/NTT hot path.
 *
 * Drop-in boundary: these entry points are what a Rust `extern "C"` block in crates/ring would
 * bind to replace the CPU implementations behind `CyclotomicConfig<N>`
 * (crates/ring/src/cyclotomic_ring/ring_config.rs:11-35) at BATCH granularity, i.e. at the seam
 * the reference itself exposes through `CRT::elementwise_crt` / `ICRT::elementwise_icrt`
 * (crt.rs:10-25, 34-49), `Flatten` (flatten.rs:10-34) and `into_raw_parts` (utils.rs:3-6).
 * INTEGRATION.md shows the Rust-side shim.
 *
 * Data layout (identical to the reference's, no padding, 8-byte aligned):
 *   a batch is `batch` ring elements, element-major; each element is D coefficients (or CRT
 *   slots, same bytes); each coefficient is N little-endian u64 limbs holding the ark-ff 0.4.2
 *   Montgomery residue a * 2^(64N) mod p, canonical in [0, p).  N = 1 (Goldilocks, BabyBear --
 *   BabyBear really is an Fp64 in the reference: babybear/mod.rs:25), N = 4 (Starknet prime).
 *
 * Ownership: the caller owns every buffer; transforms are in place; the library never frees or
 * reallocates caller memory.  A context owns twiddle tables, constants and staging scratch.
 *
 * Errors: every call returns 0 on success, non-zero otherwise (SR_E_*).  The reference panics on
 * a wrong length (goldilocks/ntt.rs:136, stark_prime/ntt.rs:122, coeff_form.rs:39); the Rust shim
 * turns a non-zero status into panic! to match.  `sr_last_error_string()` gives the reason.
 *
 * Threading: a context is bound to one HIP device and serialises its own calls with a mutex;
 * use one context per thread (or per stream) for concurrency.  `*_dev` calls are asynchronous
 * on the given stream.
 *
 * There is NO CPU fallback: if no HIP device is present every compute call fails with
 * SR_E_NO_DEVICE.
 */
#ifndef STARK_RINGS_HIP_H
#define STARK_RINGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ring identifiers */
enum sr_ring {
    /* generalised power-of-two configs  Fp[X]/(X^D+1), D = 2^log2_degree, fully split
     * (BaseCRTField = Fq, CRT_FIELD_EXTENSION_DEGREE = 1); algorithm and slot order of
     * stark_prime/ntt.rs:121-346 with psi = generator^((p-1)/2D).  SR_RING_STARK_POW2 with
     * log2_degree = 4 IS the reference's stark_prime ring (StarkRingConfig, stark_prime/mod.rs:34-68). */
    SR_RING_GOLDILOCKS_POW2 = 0,
    SR_RING_BABYBEAR_POW2 = 1,
    SR_RING_STARK_POW2 = 2,
    /* reference-native partially-splitting rings (log2_degree ignored) */
    SR_RING_GOLDILOCKS_24 = 3, /* X^24-X^12+1, 8 x Fq3: GoldilocksRingConfig, goldilocks/mod.rs:69-119 */
    SR_RING_BABYBEAR_72 = 4,   /* X^72-X^36+1, 8 x Fq9: BabyBearRingConfig,  babybear/mod.rs:81-131 */
    SR_RING_FROG_16 = 5        /* X^16+1 over p = 15912092521325583641, 4 x Fq4: FrogRingConfig, frog_ring/mod.rs:62-107
                                  ("next" row 4; crt/icrt frog_ring/ntt.rs:114-200, one u64 Montgomery limb per coefficient) */
};

enum sr_status {
    SR_OK = 0,
    SR_E_INVALID = 1,   /* null pointer, bad ring id, bad degree, bad length */
    SR_E_NO_DEVICE = 2, /* no HIP device / device index out of range */
    SR_E_HIP = 3,       /* a HIP runtime call failed */
    SR_E_ALLOC = 4,
    SR_E_UNSUPPORTED = 5 /* the entry point exists but this build does not provide it (sr_selftest_rep_counters outside the checking build) */
};

typedef struct sr_ctx sr_ctx;

/* ---- context -------------------------------------------------------------------------- */
/* Creates a context on HIP device `device` and builds its twiddle tables there.
 * Threading (the reference's types are Send + Sync): every entry point may be called from any thread; calls on one
 * context are serialised by the context (host-pointer calls run to completion, device-pointer calls enqueue and return).
 * Device-pointer calls on different streams of one context may overlap on the GPU; the context orders its own shared scratch
 * between streams with an event.  Contexts are independent of each other. */
int sr_ctx_create(int ring, int log2_degree, int device, sr_ctx **out);
/* The same with an explicit kernel plan, fixed for the life of the context (the library itself reads no environment
 * variable: a getenv racing a setenv in another thread of a Send + Sync host would be a data race).  plan == NULL or an
 * all-zero plan = the measured-fastest defaults, i.e. sr_ctx_create.  The alternative plans compute the same function bit
 * for bit and exist for differential tests and A/B measurements (tests/test_gpu_parity.py, README.md). */
enum sr_plan_flags {
    SR_PLAN_GENERIC_KERNELS = 1u << 0,  /* field-generic radix-2 LDS kernels instead of the tuned / register-tiled paths        */
    SR_PLAN_GL_NO_COLS256 = 1u << 1,    /* Goldilocks 2^16 <= D <= 2^20: strided register passes + 4096-point rows              */
    SR_PLAN_RT_NO_COLS256 = 1u << 2,    /* register-tiled path (BabyBear): 4-stage column passes + 12-stage rows                */
    SR_PLAN_GL_REGTILE = 1u << 3,       /* Goldilocks on the register-tiled path (cross-check of ntt_regtile.hpp)               */
    SR_PLAN_STARK_NO_LAZY = 1u << 4,    /* Stark transforms and sums on 8 x 32-bit limbs instead of nine 28-bit lazy limbs       */
    SR_PLAN_STARK_GENERIC_ON_LAZY = 1u << 5, /* Stark: generic LDS kernels on the lazy limbs instead of ntt_stark.hpp            */
    SR_PLAN_NO_HOST_PIN = 1u << 6,      /* accepted, no effect: host buffers are never registered (round 4 removed the pinning)  */
    SR_PLAN_GL_PLAIN_COLS = 1u << 7,    /* Goldilocks lane plans: the plain column pass instead of the workgroup-owns-its-columns
                                           one (cols256_keep_kernel, ntt_goldilocks.hpp)                                          */
    SR_PLAN_GL_SPLIT_ROWS = 1u << 8     /* Goldilocks D > 4096 ring products: the fused rows kernel as two launches (crt of b's tiles in
                                           the scratch, then the constant-operand product) -- A/B plan, measured slower */
};
typedef struct sr_plan {
    uint32_t flags;               /* OR of sr_plan_flags                                                                       */
    int32_t log_tile;             /* 0 = default; 8..12: LDS tile of the generic kernels                                       */
    int32_t stark_whole_max;      /* 0 = default (11); 9..12: largest log2 D the Stark kernels keep as one tile per element    */
    uint32_t chunk_polys;         /* fused ring products above one tile: ring elements per chunk of launches (0 = default: as many
                                     as the operand scratch holds; the tuned Goldilocks path: see lanes)                       */
    uint64_t scratch_limit_bytes; /* cap of the operand scratch (0 = default 16 GiB); larger batches run in chunks             */
    uint32_t host_chunk_mb;       /* chunk of the host-pointer pipeline in MiB (0 = default 128)                               */
    uint32_t lanes;               /* chunked products (tuned Goldilocks 2^16 <= D <= 2^20, register-tiled BabyBear): 0 = AUTO -- the
                                     library times both plans, warmed up and run as they will run, once on this process's real stream-to-hardware-
                                     queue mapping and keeps the faster.  The measurement runs ONLY inside sr_ctx_reserve_scratch
                                     (a blocking call anyway); a context whose host never reserves runs two lanes, unmeasured --
                                     no asynchronous _dev call ever probes, blocks on a measurement or breaks a stream capture;
                                     2 = two internal streams ("lanes"), chunks of chunk_polys or 64 MiB of coefficients each,
                                     intermediates in per-lane scratch so that they are re-read from the Infinity Cache;
                                     1 = one stream (eight large chunks, or chunk_polys).  sr_ctx_plan_in_use tells which.     */
} sr_plan;
int sr_ctx_create_ex(int ring, int log2_degree, int device, const sr_plan *plan, sr_ctx **out);
/* Pre-sizes the context's operand scratch for fused ring products of up to `batch` elements (capped by the plan's
 * scratch_limit_bytes).  Device entry points are asynchronous, with ONE exception: a ring product above one LDS tile whose
 * scratch has to grow first blocks (hipDeviceSynchronize + hipFree + hipMalloc) -- call this once after creating the context
 * (or accept that the first product of a new size blocks) and no _dev call blocks afterwards: the reservation covers ring
 * products AND stand-alone transforms of up to `batch` elements.  From then on the _dev calls of
 * that batch size can also be CAPTURED into a HIP graph by the caller (hipStreamBeginCapture on `stream`): they neither allocate
 * nor synchronise nor probe, and the internal lanes fork from and join to `stream` by events
 * (tests/test_gpu_parity.py: test_device_calls_can_be_captured_into_a_hip_graph).
 * LIMITS of capture -- a captured graph bakes in this context's scratch addresses and lane streams:
 *   (1) the scratch must not move while the graph is alive.  The library enforces its half: once a _dev call has arrived on a
 *       capturing stream, no _dev call grows a context buffer any more -- one that would have to returns SR_E_INVALID (also during the
 *       capture itself: reserve first).  Only sr_ctx_reserve_scratch still grows; calling it with a larger batch INVALIDATES graphs
 *       captured earlier on this context (destroy them first).
 *   (2) ordering between a replay and other work on the same context is the HOST's job: the event that orders users of the shared
 *       scratch is recorded at capture time, not at replay, so a replay is unordered against eager _dev calls of this context on
 *       other streams and against replays of a second graph -- do not overlap them (same stream, or an event / synchronisation of
 *       your own).  One graph per context at a time; use one context per concurrent graph.
 * Chunking of a batch (sr_plan.chunk_polys = 0): never below 64 MiB of coefficients per set of launches; products take the two
 * lanes from three and a half such chunks on, stand-alone transforms from eight, one set of launches on the caller's stream below. */
int sr_ctx_reserve_scratch(sr_ctx *ctx, size_t batch);
/* The plan the context runs: *plan = the sr_plan it was created with, with lanes resolved to 1 or 2 once the library has settled it
 * (lanes stays 0 while sr_plan.lanes = 0 and sr_ctx_reserve_scratch has not run: such a context runs two lanes).  probe_ms
 * (optional): what the probe measured, [0] two lanes, [1] one stream: milliseconds per call of probe_elems ring products, each plan
 * run as it will run (mean of four calls after four warm-up calls; 0 = not measured: the plan was given explicitly, the ring has no
 * chunked product, the batch reserved for is below eight chunks, or the probe's temporaries could not be allocated). */
int sr_ctx_plan_in_use(sr_ctx *ctx, sr_plan *plan, double probe_ms[2], size_t *probe_elems);
int sr_ctx_destroy(sr_ctx *ctx);
/* D, u64 limbs per coefficient, u64 words per ring element */
int sr_ctx_degree(const sr_ctx *ctx, size_t *degree);
int sr_ctx_limbs(const sr_ctx *ctx, int *limbs);
/* Twiddle block (forward + inverse tables, device memory) for sharing across the GPUs of a node:
 * rank 0 builds it, broadcasts the bytes (RCCL over xGMI), the other ranks adopt them. */
int sr_ctx_twiddle_block(sr_ctx *ctx, void **dev_ptr, size_t *bytes);
/* Tell the context its twiddle block was overwritten (e.g. by a broadcast). */
int sr_ctx_twiddles_updated(sr_ctx *ctx);
/* Single-process multi-GPU form of the above ("sr_ctx_create(.., device_ids[], n)"; what a Rust host that drives the
 * 8 GPUs of a node from one process binds): n contexts, one per device_ids[i], whose twiddle blocks are all copies of the block
 * device_ids[0] built (hipMemcpyPeer, i.e. xGMI between the GPUs of a node) -- the only inter-GPU traffic of the path.  out[]
 * receives n contexts (all destroyed again on failure).  sr_shard_range gives part i of a batch split contiguously and evenly
 * (the first batch % n parts get one element more), the partition of stark_rings_amd/sharding.py. */
int sr_ctx_create_group(int ring, int log2_degree, const int *device_ids, int n, const sr_plan *plan, sr_ctx **out);
int sr_shard_range(size_t batch, int n, int i, size_t *first, size_t *count);

/* ---- host-buffer entry points (stage through device memory; PCIe-inclusive) ------------- */
/* CRT::elementwise_crt  (crt.rs:10-25): coefficient form -> CRT/NTT form, in place.       */
int sr_ntt_fwd_batch(sr_ctx *ctx, uint64_t *data, size_t batch);
/* ICRT::elementwise_icrt (crt.rs:34-49): inverse, in place.                               */
int sr_ntt_inv_batch(sr_ctx *ctx, uint64_t *data, size_t batch);
/* RqNTT MulAssign<&Self> per element (ntt_form.rs:213-225; values of mul_unchecked :177-189) */
int sr_pointwise_mul_batch(sr_ctx *ctx, uint64_t *lhs_inout, const uint64_t *rhs, size_t batch);
/* RqNTT += / -= &RqNTT (ntt_form.rs:227-285, 588-638) and the same for RqPoly: coefficient-wise in either form. */
int sr_add_batch(sr_ctx *ctx, uint64_t *lhs_inout, const uint64_t *rhs, size_t batch);
int sr_sub_batch(sr_ctx *ctx, uint64_t *lhs_inout, const uint64_t *rhs, size_t batch);
/* The unary operators of the same element types (round 4), every ring id, word-wise in either form:
 *   sr_neg_batch          Neg: RqPoly `self.0.map(|x| -x)` (coeff_form.rs:270-278), RqNTT (ntt_form.rs:191-203).
 *   sr_scale_batch        every coefficient / every slot component times ONE base-field scalar: RqPoly Mul<Fp> (= poly_mul by
 *                         from_scalar, coeff_form.rs:390-408) and Mul / MulAssign<u128|u64|u32|u16|u8|bool> (`*lhs *= Fp::from(rhs)`,
 *                         coeff_form.rs:610-650); RqNTT Mul / MulAssign<primitive> (`*lhs *= BaseCRTField::from(rhs)`,
 *                         ntt_form.rs:373-425: a base-field element embedded in Fq3 / Fq9 / Fq4 scales every component).
 *   sr_add_scalar_batch   Add / AddAssign<primitive> (Sub: pass the negated scalar).  ntt_form = 0, RqPoly: coefficient 0 of every
 *                         element += scalar (coeff_form.rs:652-700); ntt_form != 0, RqNTT: component 0 of EVERY slot += scalar
 *                         (`*lhs += BaseCRTField::from(rhs)`, ntt_form.rs:427-505).
 * scalar: HOST pointer (also in the _dev forms) to the N-limb Montgomery memory image of the base-field element (what Fp::from(rhs)
 * holds); a word >= p is SR_E_INVALID. */
int sr_neg_batch(sr_ctx *ctx, uint64_t *data, size_t batch);
int sr_scale_batch(sr_ctx *ctx, uint64_t *data, const uint64_t *scalar, size_t batch);
int sr_add_scalar_batch(sr_ctx *ctx, uint64_t *data, const uint64_t *scalar, int ntt_form, size_t batch);
/* Every element of the batch (CRT/NTT form) times ONE ring element, slot-wise: `MulAssign<&R> for Matrix<R>`
 * (linear_algebra/src/matrix.rs:207-211) and `for SparseMatrix<R>` (sparse_matrix.rs:303-307) on the matrix's flat storage --
 * `row.iter_mut().for_each(|r_m| *r_m *= r)`.  Every ring id (Fq3 / Fq9 / Fq4 slot products for the reference's own rings).
 * elem: D coefficients; it must not lie inside the batch it multiplies (the _dev form: a DEVICE pointer). */
int sr_mul_elem_batch(sr_ctx *ctx, uint64_t *data_inout, const uint64_t *elem, size_t batch);
/* Batch reductions of the same element types (round 5): `Sum` and `Product` over a slice of ring elements --
 *   sr_sum_batch       impl Sum<Self> / Sum<&Self>: `iter.fold(Self::zero(), |acc, x| acc + x)` for RqPoly (coeff_form.rs:507-521)
 *                      and RqNTT (ntt_form.rs:640-654): n elements -> 1, word-wise, the same in either form, every ring id;
 *                      n = 0 gives zero().
 *   sr_product_batch   impl Product<Self> / Product<&Self> for RqNTT: `iter.fold(Self::one(), |acc, x| acc * x)`, slot-wise
 *                      (ntt_form.rs:656-670), every ring id (Fq3 / Fq9 / Fq4 slot products for the reference's own rings); n = 0
 *                      gives one() = every slot (1, 0, ..).  Product for RqPoly (coeff_form.rs:523-537, `acc * x` = the ring product)
 *                      is icrt(product(crt(x_i))): the mirrors compose it from sr_ntt_fwd_batch, this call and sr_ntt_inv_batch on
 *                      ONE element (include/stark_rings.hpp: product_poly; stark_rings_amd/rings.py: product_poly[_dev]).
 * out: one ring element; it must not overlap the n input elements.  The folds are associative and commutative and every partial
 * result is canonical, so the tree order the device uses gives the reference's left fold bit for bit.  Partial elements live in
 * context-owned temporaries (ordered between caller streams like the operand scratch; sized on first use or by
 * sr_ctx_reserve_scratch). */
int sr_sum_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *in, size_t n);
int sr_product_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *in_ntt, size_t n);
/* RqPoly * RqPoly == icrt(crt(a) * crt(b)) (coeff_form.rs:250-258; identity tested at
 * stark_prime/mod.rs:161-177).  out may alias a.                                          */
int sr_ring_mul_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch);
/* CyclotomicConfig::reduce_in_place (stark_prime/mod.rs:40-47, goldilocks/mod.rs:75-98,
 * babybear/mod.rs:87-110): each element has in_len_per_elem <= 2D coefficients -> D.      */
int sr_reduce_batch(sr_ctx *ctx, const uint64_t *in, size_t in_len_per_elem, uint64_t *out, size_t batch);

/* ---- device-resident entry points (pointers are HIP device pointers; `stream` is a
 *      hipStream_t passed as void*; asynchronous) ----------------------------------------- */
int sr_ntt_fwd_batch_dev(sr_ctx *ctx, uint64_t *d_data, size_t batch, void *stream);
int sr_ntt_inv_batch_dev(sr_ctx *ctx, uint64_t *d_data, size_t batch, void *stream);
int sr_pointwise_mul_batch_dev(sr_ctx *ctx, uint64_t *d_lhs_inout, const uint64_t *d_rhs, size_t batch, void *stream);
int sr_add_batch_dev(sr_ctx *ctx, uint64_t *d_lhs_inout, const uint64_t *d_rhs, size_t batch, void *stream);
int sr_sub_batch_dev(sr_ctx *ctx, uint64_t *d_lhs_inout, const uint64_t *d_rhs, size_t batch, void *stream);
int sr_neg_batch_dev(sr_ctx *ctx, uint64_t *d_data, size_t batch, void *stream);
int sr_scale_batch_dev(sr_ctx *ctx, uint64_t *d_data, const uint64_t *host_scalar, size_t batch, void *stream);
int sr_add_scalar_batch_dev(sr_ctx *ctx, uint64_t *d_data, const uint64_t *host_scalar, int ntt_form, size_t batch, void *stream);
int sr_mul_elem_batch_dev(sr_ctx *ctx, uint64_t *d_data_inout, const uint64_t *d_elem, size_t batch, void *stream);
int sr_sum_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, size_t n, void *stream);
int sr_product_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in_ntt, size_t n, void *stream);
/* First "next" row: y = M * v for a dense nrows x ncols matrix of ring elements in CRT/NTT form
 * (row-major, each entry one ring element) and a vector of ncols elements -- Matrix<RqNTT>::checked_mul_vec,
 * crates/linear_algebra/src/matrix.rs:168-178 -- as one fused multiply-accumulate pass over M.  Every ring id: the fully
 * split power-of-two rings (lane = slot) and the reference's own Goldilocks-24 / BabyBear-72 / Frog-16 (Fq3 / Fq9 / Fq4 slot
 * products, csrc/small_linalg.hpp); d_y must not alias d_m or d_v. */
int sr_matvec_ntt_dev(sr_ctx *ctx, uint64_t *d_y, const uint64_t *d_m, const uint64_t *d_v, size_t nrows, size_t ncols, void *stream);
/* y = S * v for a sparse nrows x ncols matrix in CSR form: d_vals[j] one ring element (CRT/NTT form), d_cols[j] its column,
 * d_row_ptr[r] .. d_row_ptr[r+1] the stored entries of row r -- SparseMatrix<RqNTT>::checked_mul_vec,
 * crates/linear_algebra/src/sparse_matrix.rs:201-211 (its Vec<Vec<(R, usize)>> rows, flattened; an empty row gives zero).
 * An entry with column >= ncols (the reference panics on v[col]) is skipped and counted; sr_spmv_bad_index_count reads
 * and clears that count (synchronises the stream).  The host-pointer form validates the indices and returns SR_E_INVALID. */
int sr_spmv_ntt_dev(sr_ctx *ctx, uint64_t *d_y, const uint64_t *d_vals, const uint32_t *d_cols, const uint64_t *d_row_ptr,
                    const uint64_t *d_v, size_t nrows, size_t ncols, void *stream);
int sr_spmv_bad_index_count(sr_ctx *ctx, unsigned long long *out, void *stream);
/* Y (n x p) = A (n x m) * B (m x p), dense row-major, CRT/NTT form -- Matrix<RqNTT>::checked_mul_mat,
 * crates/linear_algebra/src/matrix.rs:148-166.  d_y must not alias d_a or d_b. */
int sr_matmul_ntt_dev(sr_ctx *ctx, uint64_t *d_y, const uint64_t *d_a, const uint64_t *d_b, size_t n, size_t m, size_t p, void *stream);
/* Host-pointer forms of the three linear-algebra entry points (temporaries allocated per call). */
int sr_matvec_ntt(sr_ctx *ctx, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols);
int sr_spmv_ntt(sr_ctx *ctx, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr, const uint64_t *v,
                size_t nrows, size_t ncols);
int sr_matmul_ntt(sr_ctx *ctx, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p);
/* Cyclotomic::rot (crates/ring/src/traits.rs:54-66): every ring element of the batch (COEFFICIENT form) times X, modulo X^D + 1
 * (stark_prime/mod.rs:87-95, frog_ring/mod.rs:126-134 and the power-of-two rings) or X^D - X^(D/2) + 1 (goldilocks/mod.rs:138-149,
 * babybear/mod.rs:150-161).  The device form is out of place (d_out must not alias d_in); the host form works in place. */
int sr_rot_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, size_t batch, void *stream);
int sr_rot_batch(sr_ctx *ctx, uint64_t *data, size_t batch);
/* Second "next" row: balanced gadget decomposition, coefficient-wise, of `batch` ring elements in COEFFICIENT
 * form: digit j of element e is ring element e * padding_size + j of d_out (batch * padding_size elements) --
 * GadgetDecompose for &[R] (crates/ring/src/balanced_decomposition/mod.rs:163-175) over Decompose for the ring
 * (cyclotomic_ring/coeff_form.rs:587-605) over decompose_balanced_in_place (mod.rs:62-117; signed representative
 * fq_convertible.rs:21-35, stark_prime/decomposition.rs:41-53).  basis: any even value in [2, 2^64) here and the reference's whole u128 range through the
 * _wide forms below (it panics on 0, 1 and odd values: SR_E_INVALID).  Every ring id; digits are field elements in the same Montgomery
 * layout.  A coefficient that needs more than padding_size digits makes the reference panic (out[i] out of bounds): the
 * device form writes the first padding_size digits and counts it (sr_decompose_overflow_count reads and clears the count,
 * synchronising the stream); the host form returns SR_E_INVALID. */
int sr_decompose_balanced_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, uint64_t basis, size_t padding_size,
                                    size_t batch, void *stream);
int sr_decompose_overflow_count(sr_ctx *ctx, unsigned long long *out, void *stream);
/* GadgetRecompose (mod.rs:177-189, recompose :119-131): d_out[e] = sum_j basis^j * d_in[e * padding_size + j], batch_out elements */
int sr_recompose_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, uint64_t basis, size_t padding_size,
                           size_t batch_out, void *stream);
int sr_decompose_balanced_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch);
int sr_recompose_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch_out);
/* The reference's full u128 basis range: basis = basis_hi * 2^64 + basis_lo (basis_hi = 0: the calls above).  With basis >= 2^64 a
 * coefficient of a one-limb field (|x| < 2^63 < basis / 2) is its own digit 0 and every other digit is zero; for the 252-bit Stark
 * prime the digits come from a 256-by-128-bit division per digit. */
int sr_decompose_balanced_batch_wide_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, uint64_t basis_lo, uint64_t basis_hi,
                                         size_t padding_size, size_t batch, void *stream);
int sr_recompose_batch_wide_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, uint64_t basis_lo, uint64_t basis_hi,
                                size_t padding_size, size_t batch_out, void *stream);
int sr_decompose_balanced_batch_wide(sr_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t basis_lo, uint64_t basis_hi,
                                     size_t padding_size, size_t batch);
int sr_recompose_batch_wide(sr_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t basis_lo, uint64_t basis_hi, size_t padding_size,
                            size_t batch_out);
/* Third "next" row: the ark-serialize canonical wire format of a batch of ring elements --
 * CanonicalSerialize / CanonicalDeserialize for RqPoly (crates/ring/src/cyclotomic_ring/coeff_form.rs:154-189) and RqNTT
 * (ntt_form.rs:24, derived), both the flat coefficient array: each coefficient is the standard-form integer (out of
 * Montgomery form) as sr_wire_coeff_bytes() little-endian bytes (8 Goldilocks / frog, 4 BabyBear, 32 Stark; ark-ff 0.4.2
 * Fp::serialize_with_flags with EmptyFlags, third party: the byte layout is restated from the published format and is
 * PARITY UNPINNED -- the reference holds no serialised golden bytes).  An element is D * sr_wire_coeff_bytes() bytes; no
 * length prefix (the u64 length words of Vec / Matrix / SparseMatrix, matrix.rs:111-145, sparse_matrix.rs:158-200, are
 * host-side framing: include/stark_rings.hpp, stark_rings_amd/wire.py).
 * d_offsets (optional, device, one u64 per element): byte offset of the element inside the wire buffer, multiples of 8 (4 for the
 * BabyBear rings), so the
 * caller can leave room for its framing words; NULL = densely packed.  Wire and element buffers must not overlap.
 * Deserialising a coefficient >= p is ark's SerializationError::InvalidData: the device form stores 0 for it and counts it
 * (sr_wire_invalid_count reads and clears the count, synchronising the stream; misaligned offsets are counted there too and
 * their elements skipped); the host form returns SR_E_INVALID. */
size_t sr_wire_coeff_bytes(const sr_ctx *ctx);
int sr_serialize_batch_dev(sr_ctx *ctx, uint8_t *d_wire, const uint64_t *d_in, const uint64_t *d_offsets, size_t batch, void *stream);
int sr_deserialize_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint8_t *d_wire, const uint64_t *d_offsets, size_t batch,
                             void *stream);
int sr_wire_invalid_count(sr_ctx *ctx, unsigned long long *out, void *stream);
int sr_serialize_batch(sr_ctx *ctx, uint8_t *wire, const uint64_t *in, size_t batch);
int sr_deserialize_batch(sr_ctx *ctx, uint64_t *out, const uint8_t *wire, size_t batch);
/* RqPoly * &RqPoly (coeff_form.rs:250-258): d_a and d_b are only read -- the reference never mutates an operand, so one b can
 * be multiplied into many a.  d_out may alias d_a; d_b must not alias d_out.  Above one LDS tile the intermediates go
 * through the context's operand scratch (sr_ctx_reserve_scratch).  Stream semantics: one stream-ordered operation on `stream`;
 * a batch of three and a half chunks or more (64 MiB of coefficients each by default; with sr_plan.chunk_polys set: more than one
 * chunk) is forked onto the context's two internal streams and joined back onto `stream` before the call returns (sr_plan.lanes = 1, or the library's own choice when
 * lanes = 0 and its probe found one stream faster: everything on `stream` itself). */
int sr_ring_mul_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_a, const uint64_t *d_b, size_t batch, void *stream);
/* The constant-operand case: d_b_ntt already holds crt(b) (sr_ntt_fwd_batch_dev), e.g. the rows of a commitment matrix that
 * stay in NTT form (matrix.rs:168-178) while fresh coefficient-form elements arrive: d_out = icrt(crt(d_a) (.) d_b_ntt),
 * one of the three transforms saved.  Every ring id (rings without a fused kernel run forward transform, slot product and
 * inverse transform one after the other on d_out).  d_out may alias d_a; d_b_ntt is only read and must not alias d_out. */
int sr_ring_mul_ntt_rhs_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint64_t *d_a, const uint64_t *d_b_ntt, size_t batch, void *stream);
/* host-pointer form (staged through device memory like sr_ring_mul_batch) */
int sr_ring_mul_ntt_rhs_batch(sr_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch);
int sr_reduce_batch_dev(sr_ctx *ctx, const uint64_t *d_in, size_t in_len_per_elem, uint64_t *d_out, size_t batch, void *stream);
/* ---- packed-u32 boundary, SR_RING_BABYBEAR_POW2 only (opt-in; BASELINE configs[2] "packed 32-bit modmul") -------------------------
 * The reference keeps a BabyBear coefficient in an ark-ff Fp64: one u64 limb holding a * 2^64 mod p with p < 2^31
 * (crates/ring/src/cyclotomic_ring/models/babybear/mod.rs:18-26), i.e. the upper 32 bits of every word are zero.  The PACKED image
 * of a coefficient is the low half of that limb: the uint32_t (a * 2^64 mod p), canonical in [0, p) -- the same Montgomery residue
 * (R = 2^64, not 2^32) in four bytes.  A packed batch is batch * D such words, element-major like everything else.  These entry
 * points are the packed twins of sr_ntt_fwd/inv_batch_dev (CRT::elementwise_crt / ICRT::elementwise_icrt, crt.rs:10-49),
 * sr_ring_mul_batch_dev (RqPoly * &RqPoly, coeff_form.rs:250-258; operands only read, d_out may alias d_a, not d_b),
 * sr_pointwise_mul_batch_dev (ntt_form.rs:213-225) and sr_add/sub_batch_dev, value for value: unpack32(f_packed(pack32(x))) ==
 * f(x) bit for bit.  For D >= 4096 the register-tiled kernels read and write the packed words directly (half the boundary
 * bytes of the 8-byte layout); smaller degrees are widened into context-owned staging, computed and narrowed again.
 * sr_pack32_batch_dev takes the low word of every limb (the caller guarantees canonical images, whose high word is zero);
 * sr_unpack32_batch_dev zero-extends.  Neither may run in place. */
int sr_pack32_batch_dev(sr_ctx *ctx, uint32_t *d_out, const uint64_t *d_in, size_t batch, void *stream);
int sr_unpack32_batch_dev(sr_ctx *ctx, uint64_t *d_out, const uint32_t *d_in, size_t batch, void *stream);
int sr_ntt_fwd_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_data, size_t batch, void *stream);
int sr_ntt_inv_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_data, size_t batch, void *stream);
int sr_ring_mul_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_out, const uint32_t *d_a, const uint32_t *d_b, size_t batch, void *stream);
int sr_pointwise_mul_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_lhs_inout, const uint32_t *d_rhs, size_t batch, void *stream);
int sr_add_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_lhs_inout, const uint32_t *d_rhs, size_t batch, void *stream);
int sr_sub_packed32_batch_dev(sr_ctx *ctx, uint32_t *d_lhs_inout, const uint32_t *d_rhs, size_t batch, void *stream);
/* Synthetic coefficients, uniform in [0,p), counter-based (same definition as the oracle's
 * sro_fill_uniform): fills n_coeffs coefficients starting at flat coefficient index first_coeff. */
int sr_fill_uniform_dev(sr_ctx *ctx, uint64_t seed, uint64_t first_coeff, size_t n_coeffs, uint64_t *d_out, void *stream);
/* Counts coefficients that are not canonical (>= p); used by validation code. */
int sr_count_noncanonical_dev(sr_ctx *ctx, const uint64_t *d_data, size_t n_coeffs, uint64_t *host_count, void *stream);

/* ---- per-kernel timing for the bench harness: when enabled, every kernel launch is bracketed by
 *      HIP events recorded on the stream it is launched on.  sr_ctx_profile_read drains them into
 *      SR_PROF_NTAGS accumulated-milliseconds / launch-count slots and resets the accumulators. ---- */
enum sr_prof_tag {
    SR_PROF_FWD_COLS = 0,  /* strided forward stages (D > one LDS tile)                      */
    SR_PROF_ROWS = 1,      /* in-tile stages; for ring_mul the fused fwd(a),fwd(b),mul,inv kernel */
    SR_PROF_INV_COLS = 2,  /* strided inverse stages                                          */
    SR_PROF_POINTWISE = 3,
    SR_PROF_OTHER = 4,
    SR_PROF_NTAGS = 5
};
/* on = 0: off; 1: every launch bracketed; N >= 2: every N-th launch only -- two event records around EVERY launch of a two-lane plan open
 * gaps in which the other lane's kernel runs alone, so fully bracketed steps read shorter in-flight durations than undisturbed ones; a
 * sparse sample (bench.py uses 7, coprime to the 3, 4 or 6 launches of a chunk) leaves the step as it runs.
 * sr_ctx_profile_read_sampled: ms_total / launches over the bracketed launches, seen = every launch that went by, per tag. */
int sr_ctx_profile_enable(sr_ctx *ctx, int on);
int sr_ctx_profile_read(sr_ctx *ctx, double ms_total[SR_PROF_NTAGS], uint64_t launches[SR_PROF_NTAGS]);
int sr_ctx_profile_read_sampled(sr_ctx *ctx, double ms_total[SR_PROF_NTAGS], uint64_t launches[SR_PROF_NTAGS], uint64_t seen[SR_PROF_NTAGS]);

/* Host-side self-test hook for the CPU test-suite: one scalar field operation computed by the same
 * source the kernels compile (fields.hpp).  field: 0 Goldilocks, 1 BabyBear, 2 Stark.
 * op: 0 add, 1 sub, 2 in-memory (Montgomery) product, 3 twiddle product, 4 table form of a[0]; field 0 only: 5 = the tuned path's
 * compile-time shift product a[0] * 2^b[0] mod p, 1 <= b[0] <= 95.
 * Not a compute path. */
int sr_selftest_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out);

/* Test hook of the CHECKING build (libstarkrings_hip_check.so = the same sources compiled with -DSR_GL_CHECK_REPS): while the real
 * kernels of the tuned Goldilocks path run, every canonical butterfly counts a non-canonical input, every lazy butterfly a second
 * wrap or borrow, every result store a word >= p; the lazy nine-limb Stark arithmetic counts limb sums that leave int32 and
 * products that could overflow a column accumulator (csrc/fields.hpp: repcheck).  counters[0..6] as documented there, counters[7]
 * = the largest |limb| any Stark add / sub produced; reset != 0 clears them.  The counters are per-device symbols: the call visits EVERY
 * visible HIP device (synchronising each), sums [0..6] and takes the maximum of [7], so work done by contexts on any device is seen.
 * The product library returns SR_E_UNSUPPORTED: it carries no checks.
 * Not a compute path. */
int sr_selftest_rep_counters(uint64_t counters[8], int reset);

const char *sr_last_error_string(void);
const char *sr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STARK_RINGS_HIP_H */
