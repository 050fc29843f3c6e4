/* ORACLE -- test infrastructure, NOT product code.  See sr_oracle.h for scope and parity status.
 *
 * Plain-C restatement of the reference CPU path.  File:line citations are relative to
 * /root/reference/crates/ring/src/cyclotomic_ring/ .  Field arithmetic restates ark-ff 0.4.2's
 * MontBackend (generic CIOS Montgomery multiplication with R = 2^(64N), canonical outputs).
 */
#include "sr_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ======================================================================================
 * Fp64 = Fp<MontBackend<C,1>,1>  (goldilocks/mod.rs:20-24, babybear/mod.rs:21-25)
 * ==================================================================================== */
typedef struct {
    uint64_t p;       /* modulus */
    uint64_t pinv;    /* -p^{-1} mod 2^64 */
    uint64_t r1;      /* R mod p   (Montgomery one) */
    uint64_t r2;      /* R^2 mod p */
    uint64_t gen;     /* #[generator], standard form */
} fp64_cfg;

static uint64_t neg_inv64(uint64_t p) {
    uint64_t x = p; /* Newton: x = p^{-1} mod 2^64 */
    for (int i = 0; i < 6; i++) x *= 2 - p * x;
    return (uint64_t)0 - x;
}

static inline uint64_t fp64_add(const fp64_cfg *c, uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    int carry = s < a;
    if (carry || s >= c->p) s -= c->p;
    return s;
}
static inline uint64_t fp64_sub(const fp64_cfg *c, uint64_t a, uint64_t b) {
    return a >= b ? a - b : a + (c->p - b);
}
static inline uint64_t fp64_neg(const fp64_cfg *c, uint64_t a) { return a ? c->p - a : 0; }
/* Montgomery product a*b*R^-1 mod p */
static inline uint64_t fp64_mul(const fp64_cfg *c, uint64_t a, uint64_t b) {
    u128 t = (u128)a * b;
    uint64_t m = (uint64_t)t * c->pinv;
    u128 mp = (u128)m * c->p;
    u128 s = t + mp; /* low 64 bits become zero */
    int carry = s < t;
    uint64_t hi = (uint64_t)(s >> 64);
    if (carry || hi >= c->p) hi -= c->p;
    return hi;
}
static uint64_t fp64_from_std(const fp64_cfg *c, uint64_t x) { return fp64_mul(c, x % c->p, c->r2); }
static uint64_t fp64_to_std(const fp64_cfg *c, uint64_t x) { return fp64_mul(c, x, 1); }
static uint64_t fp64_pow(const fp64_cfg *c, uint64_t base, uint64_t e) {
    uint64_t r = c->r1;
    while (e) {
        if (e & 1) r = fp64_mul(c, r, base);
        base = fp64_mul(c, base, base);
        e >>= 1;
    }
    return r;
}
static uint64_t fp64_inv(const fp64_cfg *c, uint64_t a) { return fp64_pow(c, a, c->p - 2); }

static fp64_cfg g_cfg64[4]; /* indexed by field id; [SRO_STARK] unused */
static int g_cfg64_ready;

static void fp64_init_one(fp64_cfg *c, uint64_t p, uint64_t gen) {
    c->p = p;
    c->pinv = neg_inv64(p);
    c->r1 = (uint64_t)(((u128)1 << 64) % p);
    c->r2 = (uint64_t)(((u128)c->r1 * c->r1) % p);
    c->gen = gen;
}
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void stark_init(void);
static void init_all(void) {
    fp64_init_one(&g_cfg64[SRO_GOLDILOCKS], 0xFFFFFFFF00000001ULL, 7);
    fp64_init_one(&g_cfg64[SRO_BABYBEAR], 2013265921ULL, 31);
    fp64_init_one(&g_cfg64[SRO_FROG], 15912092521325583641ULL, 3); /* frog_ring/mod.rs:19-25 */
    stark_init();
    g_cfg64_ready = 1;
}
static const fp64_cfg *cfg64(int field) {
    pthread_once(&g_once, init_all);
    return &g_cfg64[field];
}

/* ======================================================================================
 * Fp256 = Fp<MontBackend<FqConfig,4>,4>, Starknet prime (stark_prime/mod.rs:20-24)
 * ==================================================================================== */
typedef struct { uint64_t l[4]; } fe4;
static const fe4 STARK_P = {{1ULL, 0ULL, 0ULL, 0x0800000000000011ULL}};
static uint64_t STARK_PINV;
static fe4 STARK_R1, STARK_R2;

static inline int fe4_geq(const fe4 *a, const fe4 *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline uint64_t fe4_add_raw(fe4 *r, const fe4 *a, const fe4 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        r->l[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static inline uint64_t fe4_sub_raw(fe4 *r, const fe4 *a, const fe4 *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline void fe4_add(fe4 *r, const fe4 *a, const fe4 *b) {
    fe4 s;
    uint64_t c = fe4_add_raw(&s, a, b);
    if (c || fe4_geq(&s, &STARK_P)) fe4_sub_raw(&s, &s, &STARK_P);
    *r = s;
}
static inline void fe4_sub(fe4 *r, const fe4 *a, const fe4 *b) {
    fe4 s;
    if (fe4_sub_raw(&s, a, b)) fe4_add_raw(&s, &s, &STARK_P);
    *r = s;
}
/* CIOS Montgomery multiplication, R = 2^256 */
static inline void fe4_mul(fe4 *r, const fe4 *a, const fe4 *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * STARK_PINV;
        c = (u128)m * STARK_P.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * STARK_P.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe4 s = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe4_geq(&s, &STARK_P)) fe4_sub_raw(&s, &s, &STARK_P);
    *r = s;
}
static void fe4_pow(fe4 *r, const fe4 *base, const uint64_t *e, int e_limbs) {
    fe4 acc = STARK_R1, b = *base;
    for (int i = 0; i < e_limbs; i++)
        for (int bit = 0; bit < 64; bit++) {
            if ((e[i] >> bit) & 1) fe4_mul(&acc, &acc, &b);
            fe4_mul(&b, &b, &b);
        }
    *r = acc;
}
static void fe4_inv(fe4 *r, const fe4 *a) {
    fe4 e; /* p - 2: p.l[0] == 1, so the subtraction borrows through the zero limbs */
    e.l[0] = (uint64_t)-1;
    e.l[1] = (uint64_t)-1;
    e.l[2] = (uint64_t)-1;
    e.l[3] = STARK_P.l[3] - 1;
    fe4_pow(r, a, e.l, 4);
}
static void fe4_from_u64(fe4 *r, uint64_t x) {
    fe4 s = {{x, 0, 0, 0}};
    fe4_mul(r, &s, &STARK_R2);
}
static void stark_init(void) {
    STARK_PINV = neg_inv64(STARK_P.l[0]);
    /* R mod p by doubling 1 exactly 256 times; R^2 by 512 doublings */
    fe4 x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        fe4 s;
        uint64_t c = fe4_add_raw(&s, &x, &x);
        if (c || fe4_geq(&s, &STARK_P)) fe4_sub_raw(&s, &s, &STARK_P);
        x = s;
        if (i == 255) STARK_R1 = x;
    }
    STARK_R2 = x;
}

/* ====================================================================================== */
int sro_limbs(int field) { return field == SRO_STARK ? 4 : 1; }

void sro_to_mont(int field, const uint64_t *in, uint64_t *out, size_t n) {
    if (field == SRO_STARK) {
        pthread_once(&g_once, init_all);
        for (size_t i = 0; i < n; i++) {
            fe4 s;
            memcpy(&s, in + 4 * i, 32);
            while (fe4_geq(&s, &STARK_P)) fe4_sub_raw(&s, &s, &STARK_P);
            fe4_mul(&s, &s, &STARK_R2);
            memcpy(out + 4 * i, &s, 32);
        }
    } else {
        const fp64_cfg *c = cfg64(field);
        for (size_t i = 0; i < n; i++) out[i] = fp64_from_std(c, in[i]);
    }
}
void sro_from_mont(int field, const uint64_t *in, uint64_t *out, size_t n) {
    if (field == SRO_STARK) {
        pthread_once(&g_once, init_all);
        fe4 one = {{1, 0, 0, 0}};
        for (size_t i = 0; i < n; i++) {
            fe4 s;
            memcpy(&s, in + 4 * i, 32);
            fe4_mul(&s, &s, &one);
            memcpy(out + 4 * i, &s, 32);
        }
    } else {
        const fp64_cfg *c = cfg64(field);
        for (size_t i = 0; i < n; i++) out[i] = fp64_to_std(c, in[i]);
    }
}

/* ======================================================================================
 * ark-serialize wire format of field elements -- SURVEY 8f #3.
 *   RqPoly = [Fp; D] coeff_form.rs:154-189; RqNTT ntt_form.rs:24 (derived): a ring element is its flat coefficients.
 *   Per coefficient: ark-ff 0.4.2 Fp::serialize_with_flags / deserialize_with_flags with EmptyFlags (third party,
 *   Cargo.lock:59-62, not in the tree; restated from the published algorithm): the standard-form integer as
 *   ceil(MODULUS_BIT_SIZE / 8) little-endian bytes; reading rejects an integer >= p (InvalidData).
 *   PARITY UNPINNED for the byte layout: the reference holds no serialised golden bytes. */
size_t sro_wire_bytes(int field) { return field == SRO_STARK ? 32 : (field == SRO_BABYBEAR ? 4 : 8); }

void sro_serialize(int field, const uint64_t *in, size_t n, uint8_t *out) {
    const size_t w = sro_wire_bytes(field);
    const int L = sro_limbs(field);
    uint64_t std[4];
    for (size_t i = 0; i < n; i++) {
        sro_from_mont(field, in + (size_t)L * i, std, 1);
        for (size_t b = 0; b < w; b++) out[i * w + b] = (uint8_t)(std[b / 8] >> (8 * (b % 8)));
    }
}

/* returns the number of coefficients that are not below the modulus (each reads as 0) */
size_t sro_deserialize(int field, const uint8_t *in, size_t n, uint64_t *out) {
    const size_t w = sro_wire_bytes(field);
    const int L = sro_limbs(field);
    size_t bad = 0;
    if (field == SRO_STARK) pthread_once(&g_once, init_all);
    for (size_t i = 0; i < n; i++) {
        uint64_t std[4] = {0, 0, 0, 0};
        for (size_t b = 0; b < w; b++) std[b / 8] |= (uint64_t)in[i * w + b] << (8 * (b % 8));
        int ok;
        if (field == SRO_STARK) {
            fe4 s;
            memcpy(&s, std, 32);
            ok = !fe4_geq(&s, &STARK_P);
        } else {
            ok = std[0] < cfg64(field)->p;
        }
        if (!ok) {
            bad++;
            std[0] = std[1] = std[2] = std[3] = 0;
        }
        sro_to_mont(field, std, out + (size_t)L * i, 1);
    }
    return bad;
}

/* ======================================================================================
 * Balanced (gadget) decomposition -- SURVEY 8f #2.
 *   decompose_balanced_in_place  crates/ring/src/balanced_decomposition/mod.rs:62-117
 *   signed representative        fq_convertible.rs:21-35 (Fp64), stark_prime/decomposition.rs:41-53 (Fp256):
 *                                [0, (p-1)/2] stays, ](p-1)/2, p[ maps to x - p
 *   rounded_div                  crates/linear_algebra/src/ops.rs:64-80
 *   ring elements coefficient-wise  cyclotomic_ring/coeff_form.rs:587-605
 *   gadget_decompose / recompose    balanced_decomposition/mod.rs:163-175, 119-131, 177-189
 * The value is kept as sign + magnitude (the reference's i128 / BigInt): rem = curr % b has the sign of curr;
 * |rem| <= b/2 keeps the digit and truncates; otherwise the digit is rem -+ b and curr/b moves one away from zero.
 * ==================================================================================== */
typedef struct { uint64_t l[4]; } mag4; /* 256-bit magnitude, little-endian u64 limbs */
static int mag4_is_zero(const mag4 *m) { return (m->l[0] | m->l[1] | m->l[2] | m->l[3]) == 0; }
static uint64_t mag4_divrem(mag4 *m, uint64_t b) { /* m /= b, returns m % b */
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | m->l[i];
        m->l[i] = (uint64_t)(cur / b);
        rem = cur % b;
    }
    return (uint64_t)rem;
}
static void mag4_inc(mag4 *m) {
    for (int i = 0; i < 4; i++)
        if (++m->l[i]) break;
}
/* out digit j of coefficient i of element e at out[((e * k + j) * d + i) * limbs]; returns 0, -1 (bad basis), or 1 when some
 * coefficient needed more than k digits (the reference indexes out[k] there and panics) */
int sro_decompose_balanced(int field, const uint64_t *in, size_t d, size_t batch, uint64_t b, size_t k, uint64_t *out) {
    if (b < 2 || (b & 1)) return -1; /* "cannot decompose in basis 0 or 1", "decomposition basis must be even" */
    pthread_once(&g_once, init_all);
    const int L = sro_limbs(field);
    int overflow = 0;
    for (size_t e = 0; e < batch; e++)
        for (size_t i = 0; i < d; i++) {
            mag4 m = {{0, 0, 0, 0}};
            int neg = 0;
            if (field == SRO_STARK) {
                fe4 x, one = {{1, 0, 0, 0}}, half = STARK_P, t;
                memcpy(&x, in + (e * d + i) * 4, 32);
                fe4_mul(&x, &x, &one); /* standard form */
                /* (p-1)/2 */
                half.l[0] -= 1;
                for (int q = 0; q < 4; q++) half.l[q] = (half.l[q] >> 1) | (q < 3 ? half.l[q + 1] << 63 : 0);
                if (!fe4_geq(&half, &x)) { /* x > (p-1)/2 */
                    fe4_sub_raw(&t, &STARK_P, &x);
                    x = t;
                    neg = 1;
                }
                memcpy(&m, &x, 32);
            } else {
                const fp64_cfg *c = cfg64(field);
                uint64_t x = fp64_to_std(c, in[e * d + i]);
                if (x > (c->p - 1) / 2) {
                    x = c->p - x;
                    neg = 1;
                }
                m.l[0] = x;
            }
            for (size_t j = 0; j < k; j++) {
                uint64_t rem = mag4_divrem(&m, b), dig = rem;
                int dneg = neg;
                if (rem > b / 2) {
                    dig = b - rem;
                    dneg = !neg;
                    mag4_inc(&m);
                }
                uint64_t *o = out + ((e * k + j) * d + i) * L;
                if (field == SRO_STARK) {
                    fe4 v;
                    fe4_from_u64(&v, dig); /* Montgomery form of dig */
                    if (dneg && dig) {
                        fe4 t;
                        fe4_sub_raw(&t, &STARK_P, &v);
                        v = t;
                    }
                    memcpy(o, &v, 32);
                } else {
                    const fp64_cfg *c = cfg64(field);
                    uint64_t v = fp64_from_std(c, dig);
                    o[0] = dneg ? fp64_neg(c, v) : v;
                }
            }
            if (!mag4_is_zero(&m)) overflow = 1;
        }
    return overflow;
}
/* out[e][i] = sum_j b^j in[e * k + j][i]  (Horner from the top digit, mod.rs:119-131) */
int sro_recompose(int field, const uint64_t *in, size_t d, size_t batch_out, uint64_t b, size_t k, uint64_t *out) {
    pthread_once(&g_once, init_all);
    const int L = sro_limbs(field);
    for (size_t e = 0; e < batch_out; e++)
        for (size_t i = 0; i < d; i++) {
            if (field == SRO_STARK) {
                fe4 acc = {{0, 0, 0, 0}}, bb, v;
                fe4_from_u64(&bb, b);
                for (size_t j = k; j-- > 0;) {
                    fe4_mul(&acc, &acc, &bb);
                    memcpy(&v, in + ((e * k + j) * d + i) * 4, 32);
                    fe4_add(&acc, &acc, &v);
                }
                memcpy(out + (e * d + i) * 4, &acc, 32);
            } else {
                const fp64_cfg *c = cfg64(field);
                uint64_t acc = 0, bb = fp64_from_std(c, b);
                for (size_t j = k; j-- > 0;) acc = fp64_add(c, fp64_mul(c, acc, bb), in[((e * k + j) * d + i) * L]);
                out[e * d + i] = acc;
            }
        }
    return 0;
}

static unsigned brv(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

/* ======================================================================================
 * Twiddle tables for X^D+1 (cached per field/log2d).  tw[i] = psi^brv_k(i), i = 2^s + b is the
 * twiddle of stage s block b (stark_prime/ntt.rs:126,136,146,157,...: ROOTS[8]|[4],[12]|[2],[10],[6],[14]|...).
 * itw[i] = psi^-brv_k(i).  psi = generator^((p-1)/2D).
 * ==================================================================================== */
#define MAX_LOG2D 24
typedef struct {
    uint64_t *tw, *itw; /* D elements x limbs, Montgomery form */
    uint64_t dinv[4], dinv_tw1[4]; /* D^-1 and D^-1 * itw[1]  (stark_prime/ntt.rs:51-55) */
} tw_table;
static tw_table *g_tables[3][MAX_LOG2D + 1];
static pthread_mutex_t g_tab_mu = PTHREAD_MUTEX_INITIALIZER;

static const tw_table *get_table(int field, int k) {
    pthread_once(&g_once, init_all);
    pthread_mutex_lock(&g_tab_mu);
    tw_table *t = g_tables[field][k];
    if (!t) {
        size_t d = (size_t)1 << k;
        int L = sro_limbs(field);
        t = (tw_table *)calloc(1, sizeof *t);
        t->tw = (uint64_t *)malloc(d * L * 8);
        t->itw = (uint64_t *)malloc(d * L * 8);
        if (field == SRO_STARK) {
            /* e = (p-1) >> (k+1); p-1 = 2^251 + 17*2^192 */
            fe4 pm1 = STARK_P;
            pm1.l[0] = 0;
            uint64_t e[4];
            int sh = k + 1;
            for (int i = 0; i < 4; i++) {
                int wsh = sh / 64, bsh = sh % 64;
                uint64_t lo = (i + wsh < 4) ? pm1.l[i + wsh] : 0;
                uint64_t hi = (i + wsh + 1 < 4) ? pm1.l[i + wsh + 1] : 0;
                e[i] = bsh ? (lo >> bsh) | (hi << (64 - bsh)) : lo;
            }
            fe4 g, psi, psi_inv;
            fe4_from_u64(&g, 3);
            fe4_pow(&psi, &g, e, 4);
            fe4_inv(&psi_inv, &psi);
            fe4 *pw = (fe4 *)malloc(d * sizeof(fe4)), *ipw = (fe4 *)malloc(d * sizeof(fe4));
            pw[0] = STARK_R1;
            ipw[0] = STARK_R1;
            for (size_t i = 1; i < d; i++) {
                fe4_mul(&pw[i], &pw[i - 1], &psi);
                fe4_mul(&ipw[i], &ipw[i - 1], &psi_inv);
            }
            for (size_t i = 0; i < d; i++) {
                unsigned r = brv((unsigned)i, k);
                memcpy(t->tw + 4 * i, &pw[r], 32);
                memcpy(t->itw + 4 * i, &ipw[r], 32);
            }
            free(pw);
            free(ipw);
            fe4 dd, di, x;
            fe4_from_u64(&dd, (uint64_t)d);
            fe4_inv(&di, &dd);
            memcpy(t->dinv, &di, 32);
            if (d > 1) {
                memcpy(&x, t->itw + 4, 32);
                fe4_mul(&x, &x, &di);
            } else
                x = di;
            memcpy(t->dinv_tw1, &x, 32);
        } else {
            const fp64_cfg *c = cfg64(field);
            uint64_t g = fp64_from_std(c, c->gen);
            uint64_t psi = fp64_pow(c, g, (c->p - 1) >> (k + 1));
            uint64_t psi_inv = fp64_inv(c, psi);
            uint64_t *pw = (uint64_t *)malloc(d * 8), *ipw = (uint64_t *)malloc(d * 8);
            pw[0] = ipw[0] = c->r1;
            for (size_t i = 1; i < d; i++) {
                pw[i] = fp64_mul(c, pw[i - 1], psi);
                ipw[i] = fp64_mul(c, ipw[i - 1], psi_inv);
            }
            for (size_t i = 0; i < d; i++) {
                unsigned r = brv((unsigned)i, k);
                t->tw[i] = pw[r];
                t->itw[i] = ipw[r];
            }
            free(pw);
            free(ipw);
            t->dinv[0] = fp64_inv(c, fp64_from_std(c, (uint64_t)d));
            t->dinv_tw1[0] = d > 1 ? fp64_mul(c, t->dinv[0], t->itw[1]) : t->dinv[0];
        }
        g_tables[field][k] = t;
    }
    pthread_mutex_unlock(&g_tab_mu);
    return t;
}

/* ======================================================================================
 * X^D+1 forward / inverse: generalisation of stark_prime/ntt.rs:121-235 and :245-346
 * ==================================================================================== */
static void pow2_fwd64(const fp64_cfg *c, const tw_table *t, uint64_t *a, int k) {
    size_t d = (size_t)1 << k;
    for (int s = 0; s < k; s++) {
        size_t half = d >> (s + 1);
        for (size_t b = 0; b < ((size_t)1 << s); b++) {
            uint64_t w = t->tw[((size_t)1 << s) + b];
            uint64_t *lo = a + b * 2 * half, *hi = lo + half;
            for (size_t i = 0; i < half; i++) {
                uint64_t u = lo[i];
                uint64_t v = fp64_mul(c, w, hi[i]); /* ntt.rs:126-129 */
                lo[i] = fp64_add(c, u, v);
                hi[i] = fp64_sub(c, u, v);
            }
        }
    }
}
static void pow2_inv64(const fp64_cfg *c, const tw_table *t, uint64_t *a, int k) {
    size_t d = (size_t)1 << k;
    for (int s = k - 1; s >= 1; s--) {
        size_t half = d >> (s + 1);
        for (size_t b = 0; b < ((size_t)1 << s); b++) {
            uint64_t w = t->itw[((size_t)1 << s) + b];
            uint64_t *lo = a + b * 2 * half, *hi = lo + half;
            for (size_t i = 0; i < half; i++) {
                uint64_t u = lo[i], v = hi[i];
                lo[i] = fp64_add(c, u, v);                 /* ntt.rs:250-251 */
                hi[i] = fp64_mul(c, w, fp64_sub(c, u, v));
            }
        }
    }
    if (k == 0) return;
    /* "Rewind the first step": D^-1 folded in (ntt.rs:339-345) */
    size_t half = d >> 1;
    for (size_t i = 0; i < half; i++) {
        uint64_t u = a[i], v = a[half + i];
        a[i] = fp64_mul(c, t->dinv[0], fp64_add(c, u, v));
        a[half + i] = fp64_mul(c, t->dinv_tw1[0], fp64_sub(c, u, v));
    }
}
static void pow2_fwd256(const tw_table *t, fe4 *a, int k) {
    size_t d = (size_t)1 << k;
    for (int s = 0; s < k; s++) {
        size_t half = d >> (s + 1);
        for (size_t b = 0; b < ((size_t)1 << s); b++) {
            const fe4 *w = (const fe4 *)t->tw + (((size_t)1 << s) + b);
            fe4 *lo = a + b * 2 * half, *hi = lo + half;
            for (size_t i = 0; i < half; i++) {
                fe4 u = lo[i], v;
                fe4_mul(&v, w, &hi[i]);
                fe4_add(&lo[i], &u, &v);
                fe4_sub(&hi[i], &u, &v);
            }
        }
    }
}
static void pow2_inv256(const tw_table *t, fe4 *a, int k) {
    size_t d = (size_t)1 << k;
    for (int s = k - 1; s >= 1; s--) {
        size_t half = d >> (s + 1);
        for (size_t b = 0; b < ((size_t)1 << s); b++) {
            const fe4 *w = (const fe4 *)t->itw + (((size_t)1 << s) + b);
            fe4 *lo = a + b * 2 * half, *hi = lo + half;
            for (size_t i = 0; i < half; i++) {
                fe4 u = lo[i], v = hi[i], df;
                fe4_add(&lo[i], &u, &v);
                fe4_sub(&df, &u, &v);
                fe4_mul(&hi[i], w, &df);
            }
        }
    }
    if (k == 0) return;
    size_t half = d >> 1;
    for (size_t i = 0; i < half; i++) {
        fe4 u = a[i], v = a[half + i], sm, df;
        fe4_add(&sm, &u, &v);
        fe4_sub(&df, &u, &v);
        fe4_mul(&a[i], (const fe4 *)t->dinv, &sm);
        fe4_mul(&a[half + i], (const fe4 *)t->dinv_tw1, &df);
    }
}

int sro_pow2_fwd(int field, uint64_t *a, int k) {
    if (k < 0 || k > MAX_LOG2D) return 1;
    const tw_table *t = get_table(field, k);
    if (field == SRO_STARK)
        pow2_fwd256(t, (fe4 *)a, k);
    else
        pow2_fwd64(cfg64(field), t, a, k);
    return 0;
}
int sro_pow2_inv(int field, uint64_t *a, int k) {
    if (k < 0 || k > MAX_LOG2D) return 1;
    const tw_table *t = get_table(field, k);
    if (field == SRO_STARK)
        pow2_inv256(t, (fe4 *)a, k);
    else
        pow2_inv64(cfg64(field), t, a, k);
    return 0;
}
/* ntt_form.rs:177-189 (mul_unchecked; the zero shortcut of :159-175 gives the same values) */
int sro_pow2_pointwise(int field, uint64_t *lhs, const uint64_t *rhs, size_t n) {
    if (field == SRO_STARK) {
        pthread_once(&g_once, init_all);
        fe4 *l = (fe4 *)lhs;
        const fe4 *r = (const fe4 *)rhs;
        for (size_t i = 0; i < n; i++) fe4_mul(&l[i], &l[i], &r[i]);
    } else {
        const fp64_cfg *c = cfg64(field);
        for (size_t i = 0; i < n; i++) lhs[i] = fp64_mul(c, lhs[i], rhs[i]);
    }
    return 0;
}
/* stark_prime/mod.rs:40-47: lo[i] -= hi[i]; truncate to D.  in_len may be anything in [0, 2D]. */
int sro_pow2_reduce(int field, const uint64_t *in, size_t in_len, uint64_t *out, int k) {
    size_t d = (size_t)1 << k;
    int L = sro_limbs(field);
    if (in_len > 2 * d) return 1;
    for (size_t i = 0; i < d; i++) {
        if (i < in_len)
            memcpy(out + i * L, in + i * L, 8 * L);
        else
            memset(out + i * L, 0, 8 * L);
    }
    for (size_t i = d; i < in_len; i++) {
        if (field == SRO_STARK) {
            pthread_once(&g_once, init_all);
            fe4_sub((fe4 *)out + (i - d), (fe4 *)out + (i - d), (const fe4 *)in + i);
        } else {
            out[i - d] = fp64_sub(cfg64(field), out[i - d], in[i]);
        }
    }
    return 0;
}
/* coeff_form.rs:54-67 */
int sro_schoolbook(int field, const uint64_t *a, const uint64_t *b, size_t d, uint64_t *out) {
    int L = sro_limbs(field);
    memset(out, 0, (2 * d - 1) * L * 8);
    if (field == SRO_STARK) {
        pthread_once(&g_once, init_all);
        const fe4 *x = (const fe4 *)a, *y = (const fe4 *)b;
        fe4 *o = (fe4 *)out;
        for (size_t i = 0; i < d; i++)
            for (size_t j = 0; j < d; j++) {
                fe4 pr;
                fe4_mul(&pr, &x[i], &y[j]);
                fe4_add(&o[i + j], &o[i + j], &pr);
            }
    } else {
        const fp64_cfg *c = cfg64(field);
        for (size_t i = 0; i < d; i++) {
            if (!a[i]) continue;
            for (size_t j = 0; j < d; j++) out[i + j] = fp64_add(c, out[i + j], fp64_mul(c, a[i], b[j]));
        }
    }
    return 0;
}
/* icrt(crt(a) * crt(b)) == a * b  (stark_prime/mod.rs:161-177) */
int sro_pow2_ring_mul(int field, uint64_t *out, const uint64_t *a, const uint64_t *b, int k) {
    size_t d = (size_t)1 << k;
    int L = sro_limbs(field);
    uint64_t *tb = (uint64_t *)malloc(d * L * 8);
    if (!tb) return 2;
    memcpy(tb, b, d * L * 8);
    if (out != a) memcpy(out, a, d * L * 8);
    sro_pow2_fwd(field, out, k);
    sro_pow2_fwd(field, tb, k);
    sro_pow2_pointwise(field, out, tb, d);
    sro_pow2_inv(field, out, k);
    free(tb);
    return 0;
}

/* ---- batches: serial per element like crt.rs:15-22, optional pthreads over the batch ---- */
typedef struct {
    int field, k, op;
    uint64_t *out;
    const uint64_t *a, *b;
    size_t first, last;
} job_t;
static void *job_run(void *arg) {
    job_t *j = (job_t *)arg;
    size_t stride = ((size_t)1 << j->k) * sro_limbs(j->field);
    for (size_t e = j->first; e < j->last; e++) {
        if (j->op == 0)
            sro_pow2_fwd(j->field, j->out + e * stride, j->k);
        else if (j->op == 1)
            sro_pow2_inv(j->field, j->out + e * stride, j->k);
        else
            sro_pow2_ring_mul(j->field, j->out + e * stride, j->a + e * stride, j->b + e * stride, j->k);
    }
    return NULL;
}
static int run_batch(int op, int field, uint64_t *out, const uint64_t *a, const uint64_t *b, int k,
                     size_t batch, int nthreads) {
    if (k < 0 || k > MAX_LOG2D) return 1;
    get_table(field, k);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > batch) nthreads = batch ? (int)batch : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){field, k, op, out, a, b, batch * t / nthreads, batch * (t + 1) / nthreads};
        if (nthreads == 1)
            job_run(&jobs[t]);
        else
            pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}
int sro_pow2_fwd_batch(int field, uint64_t *a, int k, size_t batch, int nthreads) {
    return run_batch(0, field, a, NULL, NULL, k, batch, nthreads);
}
int sro_pow2_inv_batch(int field, uint64_t *a, int k, size_t batch, int nthreads) {
    return run_batch(1, field, a, NULL, NULL, k, batch, nthreads);
}
int sro_pow2_ring_mul_batch(int field, uint64_t *out, const uint64_t *a, const uint64_t *b, int k,
                            size_t batch, int nthreads) {
    return run_batch(2, field, out, a, b, k, batch, nthreads);
}

/* ======================================================================================
 * X^D - X^(D/2) + 1 rings (D = 24 goldilocks, D = 72 babybear)
 * ==================================================================================== */
typedef struct {
    uint64_t R[24];                 /* ROOTS_OF_UNITY_24, Montgomery form */
    uint64_t kappa, inv8, inv4;     /* goldilocks/ntt.rs:42-47, babybear/ntt.rs:136-141 */
    int ready;
} r24_consts;
static r24_consts g_r24[2];
static pthread_mutex_t g_r24_mu = PTHREAD_MUTEX_INITIALIZER;
static const r24_consts *r24(int field) {
    const fp64_cfg *c = cfg64(field);
    pthread_mutex_lock(&g_r24_mu);
    r24_consts *r = &g_r24[field];
    if (!r->ready) {
        uint64_t w = fp64_pow(c, fp64_from_std(c, c->gen), (c->p - 1) / 24);
        r->R[0] = c->r1;
        for (int i = 1; i < 24; i++) r->R[i] = fp64_mul(c, r->R[i - 1], w);
        uint64_t two_z_m1 = fp64_sub(c, fp64_add(c, r->R[4], r->R[4]), c->r1);
        r->kappa = fp64_inv(c, two_z_m1);
        r->inv8 = fp64_inv(c, fp64_from_std(c, 8));
        r->inv4 = fp64_inv(c, fp64_from_std(c, 4));
        r->ready = 1;
    }
    pthread_mutex_unlock(&g_r24_mu);
    return r;
}

/* goldilocks/ntt.rs:146-225 == babybear/ntt.rs:154-233 with D = 24 / 72 */
static void three_stage_fwd(int field, uint64_t *a, int D) {
    const fp64_cfg *c = cfg64(field);
    const r24_consts *r = r24(field);
    int h = D / 2, q = D / 4, e = D / 8;
    for (int i = 0; i < h; i++) {
        uint64_t ci = a[i], cj = a[h + i];
        uint64_t z = fp64_mul(c, r->R[4], cj);
        a[i] = fp64_add(c, ci, z);
        a[h + i] = fp64_sub(c, fp64_add(c, ci, cj), z);
    }
    static const int b1[2][2] = {{0, 2}, {2, 10}}; /* (offset in quarters, root) */
    for (int i = 0; i < q; i++)
        for (int t = 0; t < 2; t++) {
            uint64_t *lo = a + b1[t][0] * q;
            uint64_t ci = lo[i], v = fp64_mul(c, r->R[b1[t][1]], lo[q + i]);
            lo[i] = fp64_add(c, ci, v);
            lo[q + i] = fp64_sub(c, ci, v);
        }
    static const int b2[4][2] = {{0, 1}, {2, 7}, {4, 5}, {6, 11}}; /* (offset in eighths, root) */
    for (int i = 0; i < e; i++)
        for (int t = 0; t < 4; t++) {
            uint64_t *lo = a + b2[t][0] * e;
            uint64_t ci = lo[i], v = fp64_mul(c, r->R[b2[t][1]], lo[e + i]);
            lo[i] = fp64_add(c, ci, v);
            lo[e + i] = fp64_sub(c, ci, v);
        }
}
/* goldilocks/ntt.rs:250-318 == babybear/ntt.rs:249-316 */
static void three_stage_inv(int field, uint64_t *a, int D) {
    const fp64_cfg *c = cfg64(field);
    const r24_consts *r = r24(field);
    int h = D / 2, q = D / 4, e = D / 8;
    static const int b2[4][2] = {{0, 23}, {2, 17}, {4, 19}, {6, 13}};
    for (int i = 0; i < e; i++)
        for (int t = 0; t < 4; t++) {
            uint64_t *lo = a + b2[t][0] * e;
            uint64_t ci = lo[i], cj = lo[e + i];
            lo[i] = fp64_add(c, ci, cj);
            lo[e + i] = fp64_mul(c, r->R[b2[t][1]], fp64_sub(c, ci, cj));
        }
    static const int b1[2][2] = {{0, 22}, {2, 14}};
    for (int i = 0; i < q; i++)
        for (int t = 0; t < 2; t++) {
            uint64_t *lo = a + b1[t][0] * q;
            uint64_t ci = lo[i], cj = lo[q + i];
            lo[i] = fp64_add(c, ci, cj);
            lo[q + i] = fp64_mul(c, r->R[b1[t][1]], fp64_sub(c, ci, cj));
        }
    for (int i = 0; i < h; i++) {
        uint64_t ci = a[i], cj = a[h + i];
        uint64_t kd = fp64_mul(c, r->kappa, fp64_sub(c, ci, cj));
        a[i] = fp64_mul(c, r->inv8, fp64_sub(c, fp64_add(c, ci, cj), kd));
        a[h + i] = fp64_mul(c, r->inv4, kd);
    }
}

/* A homogenize map: dst[i] = +-src[perm[i]] * R[root[i]] (root -1: no multiply; -2: negate) */
typedef struct { signed char src[9]; signed char root[9]; } hmap;
static void apply_hmap(int field, uint64_t *cseg, const hmap *m, int width) {
    const fp64_cfg *c = cfg64(field);
    const r24_consts *r = r24(field);
    uint64_t old[9];
    memcpy(old, cseg, width * 8);
    for (int i = 0; i < width; i++) {
        uint64_t v = old[m->src[i]];
        if (m->root[i] == -2)
            v = fp64_neg(c, v);
        else if (m->root[i] >= 0)
            v = fp64_mul(c, v, r->R[m->root[i]]);
        cseg[i] = v;
    }
}
/* goldilocks/ntt.rs:350-437 (blocks for e = 13, 7, 19, 5, 17, 11, 23) */
static const hmap G_HOMO[7] = {
    {{0, 1, 2}, {-1, -2, -1}}, {{0, 1, 2}, {-1, 2, 4}},  {{0, 1, 2}, {-1, 6, 12}}, {{0, 2, 1}, {-1, 3, 1}},
    {{0, 2, 1}, {-1, 11, 5}},  {{0, 2, 1}, {-1, 7, 3}},  {{0, 2, 1}, {-1, 15, 7}},
};
static const hmap G_DEHOMO[7] = {
    {{0, 1, 2}, {-1, -2, -1}}, {{0, 1, 2}, {-1, 22, 20}}, {{0, 1, 2}, {-1, 18, 12}}, {{0, 2, 1}, {-1, 23, 21}},
    {{0, 2, 1}, {-1, 19, 13}}, {{0, 2, 1}, {-1, 21, 17}}, {{0, 2, 1}, {-1, 17, 9}},
};
void sro_g24_homogenize(uint64_t *a) {
    for (int b = 1; b < 8; b++) apply_hmap(SRO_GOLDILOCKS, a + 3 * b, &G_HOMO[b - 1], 3);
}
void sro_g24_dehomogenize(uint64_t *a) {
    for (int b = 1; b < 8; b++) apply_hmap(SRO_GOLDILOCKS, a + 3 * b, &G_DEHOMO[b - 1], 3);
}
void sro_g24_crt(uint64_t *a) {
    three_stage_fwd(SRO_GOLDILOCKS, a, 24);
    sro_g24_homogenize(a);
}
void sro_g24_icrt(uint64_t *a) {
    sro_g24_dehomogenize(a);
    three_stage_inv(SRO_GOLDILOCKS, a, 24);
}

/* babybear/ntt.rs:351-578; index i of src/root describes c[i] after the map, BEFORE the
 * (1,3),(2,6),(5,7) permutation (:580-588) for homogenize, AFTER it for dehomogenize. */
static const hmap B_HOMO[7] = {
    /* 13 */ {{0, 7, 5, 3, 1, 8, 6, 4, 2}, {-1, 10, 7, 4, 1, 11, 8, 5, 2}},
    /* 7  */ {{0, 4, 8, 3, 7, 2, 6, 1, 5}, {-1, 3, 6, 2, 5, 1, 4, -1, 3}},
    /* 19 */ {{0, 1, 2, 3, 4, 5, 6, 7, 8}, {-1, 2, 4, 6, 8, 10, -2, 14, 16}},
    /* 5  */ {{0, 2, 4, 6, 8, 1, 3, 5, 7}, {-1, 1, 2, 3, 4, -1, 1, 2, 3}},
    /* 17 */ {{0, 8, 7, 6, 5, 4, 3, 2, 1}, {-1, 15, 13, 11, 9, 7, 5, 3, 1}},
    /* 11 */ {{0, 5, 1, 6, 2, 7, 3, 8, 4}, {-1, 6, 1, 7, 2, 8, 3, 9, 4}},
    /* 23 */ {{0, 2, 4, 6, 8, 1, 3, 5, 7}, {-1, 5, 10, 15, 20, 2, 7, -2, 17}},
};
static const hmap B_DEHOMO[7] = {
    /* 13 */ {{0, 4, 8, 3, 7, 2, 6, 1, 5}, {-1, 23, 22, 20, 19, 17, 16, 14, 13}},
    /* 7  */ {{0, 7, 5, 3, 1, 8, 6, 4, 2}, {-1, -1, 23, 22, 21, 21, 20, 19, 18}},
    /* 19 */ {{0, 1, 2, 3, 4, 5, 6, 7, 8}, {-1, 22, 20, 18, 16, 14, -2, 10, 8}},
    /* 5  */ {{0, 5, 1, 6, 2, 7, 3, 8, 4}, {-1, -1, 23, 23, 22, 22, 21, 21, 20}},
    /* 17 */ {{0, 8, 7, 6, 5, 4, 3, 2, 1}, {-1, 23, 21, 19, 17, 15, 13, 11, 9}},
    /* 11 */ {{0, 2, 4, 6, 8, 1, 3, 5, 7}, {-1, 23, 22, 21, 20, 18, 17, 16, 15}},
    /* 23 */ {{0, 5, 1, 6, 2, 7, 3, 8, 4}, {-1, 22, 19, 17, 14, -2, 9, 7, 4}},
};
static void bb_perm(uint64_t *c) {
    static const int sw[3][2] = {{1, 3}, {2, 6}, {5, 7}};
    for (int i = 0; i < 3; i++) {
        uint64_t t = c[sw[i][0]];
        c[sw[i][0]] = c[sw[i][1]];
        c[sw[i][1]] = t;
    }
}
void sro_bb72_homogenize(uint64_t *a) {
    bb_perm(a);
    for (int b = 1; b < 8; b++) {
        apply_hmap(SRO_BABYBEAR, a + 9 * b, &B_HOMO[b - 1], 9);
        bb_perm(a + 9 * b);
    }
}
void sro_bb72_dehomogenize(uint64_t *a) {
    bb_perm(a);
    for (int b = 1; b < 8; b++) {
        bb_perm(a + 9 * b);
        apply_hmap(SRO_BABYBEAR, a + 9 * b, &B_DEHOMO[b - 1], 9);
    }
}
void sro_bb72_crt(uint64_t *a) {
    three_stage_fwd(SRO_BABYBEAR, a, 72);
    sro_bb72_homogenize(a);
}
void sro_bb72_icrt(uint64_t *a) {
    sro_bb72_dehomogenize(a);
    three_stage_inv(SRO_BABYBEAR, a, 72);
}

/* goldilocks/mod.rs:75-98, babybear/mod.rs:87-110 */
static void reduce_trinomial(int field, const uint64_t *in, size_t in_len, uint64_t *out, int D) {
    const fp64_cfg *c = cfg64(field);
#define GET(i) ((size_t)(i) < in_len ? in[i] : 0)
    for (int i = 0; i < D / 2; i++) {
        uint64_t v = fp64_sub(c, GET(i), GET(D + i));
        out[i] = fp64_sub(c, v, GET(D + D / 2 + i));
    }
    for (int i = D / 2; i < D; i++) out[i] = fp64_add(c, GET(i), GET(D / 2 + i));
#undef GET
}
void sro_g24_reduce(const uint64_t *in, size_t n, uint64_t *out) { reduce_trinomial(SRO_GOLDILOCKS, in, n, out, 24); }
void sro_bb72_reduce(const uint64_t *in, size_t n, uint64_t *out) { reduce_trinomial(SRO_BABYBEAR, in, n, out, 72); }

/* ---- slot products.  ark-ff CubicExtField::mul_assign (Karatsuba-style) computes the field
 * product with canonical coordinates; restated here as schoolbook + fold, same values. ---- */
static void fq3_mul(int field, uint64_t *x, const uint64_t *y) {
    const fp64_cfg *c = cfg64(field);
    uint64_t nr = r24(field)->R[1]; /* NONRESIDUE: goldilocks/mod.rs:42, babybear/mod.rs:40 */
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i + j] = fp64_add(c, t[i + j], fp64_mul(c, x[i], y[j]));
    x[0] = fp64_add(c, t[0], fp64_mul(c, nr, t[3]));
    x[1] = fp64_add(c, t[1], fp64_mul(c, nr, t[4]));
    x[2] = t[2];
}
void sro_g24_ntt_mul(uint64_t *lhs, const uint64_t *rhs) {
    for (int s = 0; s < 8; s++) fq3_mul(SRO_GOLDILOCKS, lhs + 3 * s, rhs + 3 * s);
}
/* Fq9 = Fq3[v]/(v^3-u): memory 3i+j <-> X^(i+3j) mod X^9 - NONRESIDUE (babybear/mod.rs:51-66, fq9.rs:18-27) */
static void fq9_mul(uint64_t *x, const uint64_t *y) {
    const fp64_cfg *c = cfg64(SRO_BABYBEAR);
    uint64_t nr = r24(SRO_BABYBEAR)->R[1];
    uint64_t px[9], py[9], t[17];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            px[i + 3 * j] = x[3 * i + j];
            py[i + 3 * j] = y[3 * i + j];
        }
    memset(t, 0, sizeof t);
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) t[i + j] = fp64_add(c, t[i + j], fp64_mul(c, px[i], py[j]));
    for (int k = 0; k < 8; k++) t[k] = fp64_add(c, t[k], fp64_mul(c, nr, t[k + 9]));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) x[3 * i + j] = t[i + 3 * j];
}
void sro_bb72_ntt_mul(uint64_t *lhs, const uint64_t *rhs) {
    for (int s = 0; s < 8; s++) fq9_mul(lhs + 9 * s, rhs + 9 * s);
}

/* ======================================================================================
 * frog ring Fq[X]/(X^16 + 1) -> 4 x Fq4: frog_ring/ntt.rs:108-267, mod.rs:36-60, 72-79
 * ==================================================================================== */
static uint64_t FROG_R[8], FROG_INV4; /* ROOTS_OF_UNITY_8, FOUR_INV in Montgomery form (ntt.rs:15-27) */
static pthread_once_t g_frog_once = PTHREAD_ONCE_INIT;
static void frog_init(void) {
    const fp64_cfg *c = cfg64(SRO_FROG);
    uint64_t w = fp64_pow(c, fp64_from_std(c, 3), (c->p - 1) / 8);
    FROG_R[0] = c->r1;
    for (int k = 1; k < 8; k++) FROG_R[k] = fp64_mul(c, FROG_R[k - 1], w);
    FROG_INV4 = fp64_inv(c, fp64_from_std(c, 4));
}
typedef struct { signed char src[4], root[4]; } frog_map; /* root -1 copy, -2 negate */
static const frog_map FROG_HOMO[4] = {   /* ntt.rs:206-211, 227-291 */
    {{0, 2, 1, 3}, {-1, -1, -1, -1}}, {{0, 2, 1, 3}, {-1, 2, 1, 3}}, {{0, 2, 3, 1}, {-1, 1, 6, -2}}, {{0, 2, 3, 1}, {-1, 3, 5, 1}}};
static const frog_map FROG_DEHOMO[4] = { /* ntt.rs:215-220 */
    {{0, 2, 1, 3}, {-1, -1, -1, -1}}, {{0, 2, 1, 3}, {-1, 7, 6, 5}}, {{0, 3, 1, 2}, {-1, -2, 7, 2}}, {{0, 3, 1, 2}, {-1, 7, 5, 3}}};
static void frog_maps(uint64_t *a, const frog_map *maps) {
    const fp64_cfg *c = cfg64(SRO_FROG);
    pthread_once(&g_frog_once, frog_init);
    for (int b = 0; b < 4; b++) {
        uint64_t seg[4];
        memcpy(seg, a + 4 * b, 32);
        for (int i = 0; i < 4; i++) {
            uint64_t v = seg[maps[b].src[i]];
            int r = maps[b].root[i];
            a[4 * b + i] = r == -1 ? v : (r == -2 ? fp64_neg(c, v) : fp64_mul(c, v, FROG_R[r]));
        }
    }
}
void sro_frog16_homogenize(uint64_t *a) { frog_maps(a, FROG_HOMO); }
void sro_frog16_dehomogenize(uint64_t *a) { frog_maps(a, FROG_DEHOMO); }
void sro_frog16_crt(uint64_t *a) { /* ntt.rs:114-151 */
    const fp64_cfg *c = cfg64(SRO_FROG);
    pthread_once(&g_frog_once, frog_init);
    for (int i = 0; i < 8; i++) {
        uint64_t x = a[i], z = fp64_mul(c, FROG_R[2], a[8 + i]);
        a[i] = fp64_add(c, x, z);
        a[8 + i] = fp64_sub(c, x, z);
    }
    for (int i = 0; i < 4; i++) {
        uint64_t x = a[i], z = fp64_mul(c, FROG_R[1], a[4 + i]);
        a[i] = fp64_add(c, x, z);
        a[4 + i] = fp64_sub(c, x, z);
        x = a[8 + i];
        z = fp64_mul(c, FROG_R[3], a[12 + i]);
        a[8 + i] = fp64_add(c, x, z);
        a[12 + i] = fp64_sub(c, x, z);
    }
    sro_frog16_homogenize(a);
}
void sro_frog16_icrt(uint64_t *a) { /* ntt.rs:163-200 */
    const fp64_cfg *c = cfg64(SRO_FROG);
    pthread_once(&g_frog_once, frog_init);
    sro_frog16_dehomogenize(a);
    for (int i = 0; i < 4; i++) {
        uint64_t x = a[i], y = a[4 + i];
        a[i] = fp64_add(c, x, y);
        a[4 + i] = fp64_mul(c, FROG_R[7], fp64_sub(c, x, y));
        x = a[8 + i];
        y = a[12 + i];
        a[8 + i] = fp64_add(c, x, y);
        a[12 + i] = fp64_mul(c, FROG_R[5], fp64_sub(c, x, y));
    }
    for (int i = 0; i < 8; i++) {
        uint64_t x = a[i], y = a[8 + i];
        a[i] = fp64_mul(c, FROG_INV4, fp64_add(c, x, y));
        a[8 + i] = fp64_mul(c, FROG_INV4, fp64_mul(c, FROG_R[6], fp64_sub(c, x, y)));
    }
}
void sro_frog16_reduce(const uint64_t *in, size_t n, uint64_t *out) { /* mod.rs:72-79 */
    const fp64_cfg *c = cfg64(SRO_FROG);
    for (size_t i = 0; i < 16; i++) {
        uint64_t lo = i < n ? in[i] : 0, hi = 16 + i < n ? in[16 + i] : 0;
        out[i] = fp64_sub(c, lo, hi);
    }
}
/* Fq4 = Fq2[v]/(v^2 - u), Fq2 = Fq[u]/(u^2 - NONRESIDUE), NONRESIDUE = ROOTS[1]; memory (c0.c0, c0.c1, c1.c0, c1.c1) */
static void frog_fq2_mul(const fp64_cfg *c, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t r0 = fp64_add(c, fp64_mul(c, a[0], b[0]), fp64_mul(c, FROG_R[1], fp64_mul(c, a[1], b[1])));
    uint64_t r1 = fp64_add(c, fp64_mul(c, a[0], b[1]), fp64_mul(c, a[1], b[0]));
    r[0] = r0;
    r[1] = r1;
}
void sro_frog16_ntt_mul(uint64_t *lhs, const uint64_t *rhs) {
    const fp64_cfg *c = cfg64(SRO_FROG);
    pthread_once(&g_frog_once, frog_init);
    for (int s = 0; s < 4; s++) {
        uint64_t *x = lhs + 4 * s;
        const uint64_t *y = rhs + 4 * s;
        uint64_t p00[2], p11[2], p01[2], p10[2];
        frog_fq2_mul(c, p00, x, y);
        frog_fq2_mul(c, p11, x + 2, y + 2);
        frog_fq2_mul(c, p01, x, y + 2);
        frog_fq2_mul(c, p10, x + 2, y);
        /* c0 = a0 b0 + u a1 b1 with u (t0 + t1 u) = NR t1 + t0 u;  c1 = a0 b1 + a1 b0 */
        x[0] = fp64_add(c, p00[0], fp64_mul(c, FROG_R[1], p11[1]));
        x[1] = fp64_add(c, p00[1], p11[0]);
        x[2] = fp64_add(c, p01[0], p10[0]);
        x[3] = fp64_add(c, p01[1], p10[1]);
    }
}

/* Cyclotomic::rot (traits.rs:54-66; stark_prime/mod.rs:87-95, goldilocks/mod.rs:138-149, babybear/mod.rs:150-161,
 * frog_ring/mod.rs:126-134): out = X * in modulo X^d + 1 (trinomial == 0) or X^d - X^(d/2) + 1 (trinomial != 0) */
void sro_rot(int field, const uint64_t *in, size_t d, int trinomial, uint64_t *out) {
    pthread_once(&g_once, init_all);
    if (field == SRO_STARK) {
        fe4 last, z = {{0, 0, 0, 0}}, t;
        memcpy(&last, in + 4 * (d - 1), 32);
        fe4_sub(&t, &z, &last);
        memcpy(out, &t, 32);
        memmove(out + 4, in, 32 * (d - 1));
    } else {
        const fp64_cfg *c = cfg64(field);
        uint64_t last = in[d - 1];
        memmove(out + 1, in, 8 * (d - 1));
        out[0] = fp64_neg(c, last);
        if (trinomial) out[d / 2] = fp64_add(c, out[d / 2], last);
    }
}

/* ======================================================================================
 * Synthetic inputs (shared definition with the HIP library's generator; see sr_oracle.h)
 * ==================================================================================== */
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t prng_word(uint64_t seed, uint64_t idx, unsigned limb, unsigned retry) {
    uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * (idx + 1));
    return mix64(x ^ (0xD1B54A32D192ED03ULL * (uint64_t)(limb + 4 * retry + 1)));
}
void sro_fill_uniform(int field, uint64_t seed, uint64_t first, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint64_t idx = first + i;
        if (field == SRO_STARK) {
            fe4 v = {{0, 0, 0, 0}};
            for (unsigned r = 0; r < 64; r++) {
                for (unsigned l = 0; l < 4; l++) v.l[l] = prng_word(seed, idx, l, r);
                v.l[3] &= 0x0FFFFFFFFFFFFFFFULL; /* 252 bits */
                if (!fe4_geq(&v, &STARK_P)) break;
                memset(&v, 0, sizeof v);
            }
            memcpy(out + 4 * i, &v, 32);
        } else {
            uint64_t p = field == SRO_GOLDILOCKS ? 0xFFFFFFFF00000001ULL : (field == SRO_FROG ? 15912092521325583641ULL : 2013265921ULL);
            uint64_t v = 0;
            for (unsigned r = 0; r < 64; r++) {
                v = prng_word(seed, idx, 0, r);
                if (field == SRO_BABYBEAR) v &= 0x7FFFFFFFULL;
                if (v < p) break;
                v = 0;
            }
            out[i] = v;
        }
    }
}
