/* ORACLE -- test infrastructure, NOT product code.
 *
 * CPU restatement (plain C, gcc, unsigned __int128) of the reference's CRT/NTT hot path:
 * NethermindEth/stark-rings @ 2025-10-17, crates/ring/src/cyclotomic_ring/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic follows ark-ff 0.4.2 (Cargo.lock:59-62; source NOT in the reference tree):
 * Fp<MontBackend<C,N>,N> = N little-endian u64 limbs holding a * 2^(64N) mod p, canonical in [0,p).
 * All buffers below are in that in-memory (Montgomery) form unless a name says "std".
 *
 * Parity status: pinned at STANDARD-FORM level by the reference's literal KATs
 * (tests/golden/reference_kats.json).  The raw Montgomery limb image is pinned by no
 * reference test: "parity unpinned" at byte level -- it rests on ark-ff's
 * documented representation.
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SRO_GOLDILOCKS = 0, SRO_BABYBEAR = 1, SRO_STARK = 2, SRO_FROG = 3 /* frog_ring/mod.rs:19-25, only the frog16 and element-wise entry points */ };

/* limbs per field element (1, 1, 4) */
int sro_limbs(int field);

/* standard-form integers (N LE limbs each) <-> Montgomery form, n elements */
void sro_to_mont(int field, const uint64_t *std_in, uint64_t *mont_out, size_t n);
void sro_from_mont(int field, const uint64_t *mont_in, uint64_t *std_out, size_t n);

/* ark-serialize wire format (coeff_form.rs:154-189, ntt_form.rs:24; ark-ff 0.4.2 Fp::serialize_with_flags):
 * n coefficients <-> n * sro_wire_bytes(field) bytes, little-endian standard form.  sro_deserialize returns the number of
 * coefficients >= p (ark: InvalidData), each read as 0.  Byte layout restated from the published format: parity unpinned. */
size_t sro_wire_bytes(int field);
void sro_serialize(int field, const uint64_t *in, size_t n, uint8_t *out);
size_t sro_deserialize(int field, const uint8_t *in, size_t n, uint64_t *out);

/* ---- generalised power-of-two ring  Fp[X]/(X^D+1), D = 2^log2d  (SURVEY a10, Appendix A;
 *      algorithm of stark_prime/ntt.rs:121-346 with psi = g^((p-1)/2D)) ---- */
int sro_pow2_fwd(int field, uint64_t *a, int log2d);                       /* crt_in_place  */
int sro_pow2_inv(int field, uint64_t *a, int log2d);                       /* icrt_in_place */
int sro_pow2_pointwise(int field, uint64_t *lhs, const uint64_t *rhs, size_t n_slots); /* ntt_form.rs:177-189 */
int sro_pow2_reduce(int field, const uint64_t *in, size_t in_len, uint64_t *out, int log2d); /* stark_prime/mod.rs:40-47 */
/* Balanced gadget decomposition of `batch` ring elements of d coefficients, coefficient-wise, basis b (even, >= 2), k digits:
 * digit j of element e is ring element e * k + j of out (balanced_decomposition/mod.rs:62-117, 163-175; coeff_form.rs:587-605).
 * Returns 0, -1 for a bad basis, 1 if a coefficient needs more than k digits (the reference panics). */
int sro_decompose_balanced(int field, const uint64_t *in, size_t d, size_t batch, uint64_t b, size_t k, uint64_t *out);
/* gadget_recompose (mod.rs:119-131, 177-189): out[e] = sum_j b^j in[e * k + j] */
int sro_recompose(int field, const uint64_t *in, size_t d, size_t batch_out, uint64_t b, size_t k, uint64_t *out);
/* Cyclotomic::rot (traits.rs:54-66): out = X * in; in and out must not overlap */
void sro_rot(int field, const uint64_t *in, size_t d, int trinomial, uint64_t *out);
int sro_schoolbook(int field, const uint64_t *a, const uint64_t *b, size_t d, uint64_t *out_2d_minus_1); /* coeff_form.rs:54-67 */
int sro_pow2_ring_mul(int field, uint64_t *out, const uint64_t *a, const uint64_t *b, int log2d);
/* batch of independent elements; nthreads >= 1 pthreads over the batch (mirrors cfg_iter!) */
int sro_pow2_fwd_batch(int field, uint64_t *a, int log2d, size_t batch, int nthreads);
int sro_pow2_inv_batch(int field, uint64_t *a, int log2d, size_t batch, int nthreads);
int sro_pow2_ring_mul_batch(int field, uint64_t *out, const uint64_t *a, const uint64_t *b,
                            int log2d, size_t batch, int nthreads);

/* ---- reference-native small rings ---- */
void sro_g24_crt(uint64_t *a);            /* goldilocks/ntt.rs:135-228  (24 coeffs -> 8 x Fq3) */
void sro_g24_icrt(uint64_t *a);           /* goldilocks/ntt.rs:240-319 */
void sro_g24_homogenize(uint64_t *a);     /* goldilocks/ntt.rs:326-334 */
void sro_g24_dehomogenize(uint64_t *a);   /* goldilocks/ntt.rs:338-346 */
void sro_g24_ntt_mul(uint64_t *lhs, const uint64_t *rhs);   /* 8 Fq3 products, ntt_form.rs:177-189 */
void sro_g24_reduce(const uint64_t *in, size_t in_len, uint64_t *out24); /* goldilocks/mod.rs:75-98 */

void sro_bb72_crt(uint64_t *a);           /* babybear/ntt.rs:143-236 */
void sro_bb72_icrt(uint64_t *a);          /* babybear/ntt.rs:238-317 */
void sro_bb72_homogenize(uint64_t *a);    /* babybear/ntt.rs:324-333 */
void sro_bb72_dehomogenize(uint64_t *a);  /* babybear/ntt.rs:337-346 */
void sro_bb72_ntt_mul(uint64_t *lhs, const uint64_t *rhs);  /* 8 Fq9 products */
void sro_bb72_reduce(const uint64_t *in, size_t in_len, uint64_t *out72); /* babybear/mod.rs:87-110 */

/* frog ring Fq[X]/(X^16+1) -> 4 x Fq4 ("next" row 4): frog_ring/ntt.rs, in place on 16 Montgomery-form words */
void sro_frog16_crt(uint64_t *a);            /* ntt.rs:114-151 */
void sro_frog16_icrt(uint64_t *a);           /* ntt.rs:163-200 */
void sro_frog16_homogenize(uint64_t *a);     /* ntt.rs:206-211 */
void sro_frog16_dehomogenize(uint64_t *a);   /* ntt.rs:215-220 */
void sro_frog16_ntt_mul(uint64_t *lhs, const uint64_t *rhs);  /* 4 Fq4 products, mod.rs:36-60 */
void sro_frog16_reduce(const uint64_t *in, size_t in_len, uint64_t *out16); /* mod.rs:72-79 */

/* ---- synthetic inputs: counter-based PRNG shared by tests, bench and the HIP library
 *      (SplitMix64 keyed by (seed, flat coefficient index, limb, retry); rejection >= p).
 *      Produces STANDARD-form uniform residues interpreted directly as the in-memory words
 *      (i.e. the Montgomery image is uniform too). ---- */
void sro_fill_uniform(int field, uint64_t seed, uint64_t first_coeff, size_t n_coeffs, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
