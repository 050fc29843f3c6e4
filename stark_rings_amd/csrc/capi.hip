// C ABI of the MI355X backend (include/stark_rings_hip.h).  Host-side set-up and launch logic only;
// kernels live in ntt_generic.hpp / ntt_goldilocks.hpp / small_rings.hpp.
//
// No CPU fallback exists on purpose: without a HIP device every compute entry point fails.
#include "../../include/stark_rings_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "fields.hpp"
#include "ntt_generic.hpp"
#include "ntt_goldilocks.hpp"
#include "ntt_regtile.hpp"
#include "small_rings.hpp"
#include "small_linalg.hpp"
#include "decompose.hpp"
#include "wire.hpp"
#include "frog_ring.hpp"
#include "ntt_stark.hpp"
#include "packed32.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(SR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

enum KernelTag { K_FWD_COLS = 0, K_ROWS = 1, K_INV_COLS = 2, K_POINTWISE = 3, K_OTHER = 4, K_NTAGS = 5 };

struct Prof {
    bool on = false;
    // stride > 1: only every stride-th launch is bracketed by events (sr_ctx_profile_enable(ctx, stride)).  Two event records around
    // EVERY launch of a two-lane plan open a gap on that lane in which the other lane's kernel runs alone, so fully bracketed steps
    // report shorter in-flight durations than the timed steps have (70-82 against 90-97 us for the config-2 rows kernel, round 4);
    // a sparse sample leaves the step as it runs.  `seen` counts every launch, `launches` the bracketed ones.
    unsigned stride = 1;
    uint64_t tick = 0;
    bool open = false;  // the launch between gl_prof_begin and gl_prof_end is a bracketed one
    struct Pair {
        hipEvent_t a, b;
        int tag;
    };
    std::vector<Pair> pending;
    double ms[K_NTAGS] = {0, 0, 0, 0, 0};
    uint64_t launches[K_NTAGS] = {0, 0, 0, 0, 0};
    uint64_t seen[K_NTAGS] = {0, 0, 0, 0, 0};
    bool sample(int tag) {
        seen[tag]++;
        return tick++ % stride == 0;
    }
};

}  // namespace

struct sr_ctx {
    int ring = 0;
    int k = 0;          // log2 D (pow2 rings)
    size_t degree = 0;  // D
    int limbs = 1;      // u64 limbs per coefficient
    int device = 0;
    int log_tile = 12;
    bool fast_goldilocks = true;
    bool regtile = false;   // BabyBear (and, for cross-checks, Goldilocks with SR_GOLDILOCKS_REGTILE=1): ntt_regtile.hpp
    sr::rt::Hooks rt_hooks{};
    sr_plan plan{};         // fixed at creation (sr_ctx_create_ex); the library reads no environment variable
    void *rt_scratch[2] = {nullptr, nullptr};   // operand scratch: [0] column-stage image of b in a fused ring product (every
                                                // path); [0], [1] packed intermediates of the register-tiled path
    size_t rt_scratch_bytes[2] = {0, 0};
    // the scratch is shared by every call on this context: a call on another stream first waits for the previous user
    hipEvent_t rt_scratch_free = nullptr;
    hipStream_t rt_scratch_stream = nullptr;
    bool rt_scratch_used = false;
    void *tables = nullptr;  // [tw (D elems) | itw (D elems)] in table form
    size_t table_bytes = 0;
    // inverse stage-0 constants (table form): plain inverse, and fused ring-mul (with boundary correction)
    unsigned char inv_scale0[40], inv_scale1[40], mul_scale0[40], mul_scale1[40];
    bool stark_one_tile = false;  // the ring element is one tile of st::tile_kernel (D <= 2048 unless SR_ST_WHOLE_MAX says otherwise)
    bool stark_tuned = false;  // and k >= 4: the register-tiled kernels of ntt_stark.hpp (SR_STARK_TUNED=0: generic kernels on StarkL)
    bool stark_lazy = false;  // Stark rings: transforms run on StarkL (nine 28-bit limbs, lazy carries; stark_lazy.hpp)
    // staging for host-pointer entry points
    sr::GlLanes gl_lanes;  // tuned Goldilocks ring product on two internal streams (streams and events created on first use)
    // sr_plan.lanes = 0 (auto): the plan the context settled on -- 0 undecided, 1 one stream, 2 two lanes -- measured ONCE on this
    // process's actual stream-to-hardware-queue mapping (lanes_autoselect); lanes_probe_ms = what the two plans took ([0] two lanes,
    // [1] one stream; 0 = not measured).  probe_chunk != 0 only while the probe runs: the one-stream plan's chunk for that run.
    int lanes_choice = 0;
    double lanes_probe_ms[2] = {0, 0};
    size_t lanes_probe_elems = 0;
    size_t probe_chunk = 0;
    bool probing = false;
    static constexpr int kTmpSlots = 10;
    void *host_tmp[kTmpSlots] = {};         // device temporaries of the host-pointer linear-algebra / decomposition calls (grow-only,
    size_t host_tmp_bytes[kTmpSlots] = {};  // see DevBuf); [5], [6]: the widened operands of packed-u32 calls below D = 4096, [7]: the
                                            // row parts of a short-and-wide small-ring mat-vec, [8], [9]: the partial elements of
                                            // sr_sum_batch_dev / sr_product_batch_dev (DevBufLite)
    // A _dev call has been seen on a stream under capture: a captured graph holds the scratch pointers of this context, so they may
    // no longer move -- from then on only sr_ctx_reserve_scratch (explicit, blocking, documented to invalidate earlier graphs) grows
    // a buffer; an asynchronous call that would have to fails with SR_E_INVALID instead of freeing memory a graph still writes to.
    bool scratch_frozen = false;
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr};  // [2], [3]: second lane of the chunked host pipeline
    size_t stage_bytes[4] = {0, 0, 0, 0};
    hipStream_t stream = nullptr;
    hipStream_t out_stream = nullptr;  // device-to-host copies of the chunked host pipeline
    unsigned long long *d_counter = nullptr;
    std::mutex mu;
    Prof prof;
    sr::SmallRingConsts small{};
    sr::FrogConsts frog{};
    sr::GoldilocksFastTables gl_fast{};
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// RAII for optional per-kernel event timing on the launch stream
struct ProfScope {
    sr_ctx *c;
    hipStream_t s;
    int tag;
    hipEvent_t a = nullptr, b = nullptr;
    bool sampled = false;
    ProfScope(sr_ctx *ctx, hipStream_t st, int t) : c(ctx), s(st), tag(t) {
        sampled = c->prof.on && c->prof.sample(tag);
        if (sampled) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (sampled) {
            (void)hipEventRecord(b, s);
            c->prof.pending.push_back({a, b, tag});
        }
    }
};

// timing hooks handed to the tuned Goldilocks launcher (tags: 0 strided fwd, 1 rows, 2 strided inv)
void gl_prof_begin(void *user, int tag, hipStream_t st) {
    sr_ctx *c = (sr_ctx *)user;
    c->prof.open = false;
    if (!c->prof.on) return;
    Prof::Pair p;
    p.tag = tag == 0 ? K_FWD_COLS : (tag == 1 ? K_ROWS : K_INV_COLS);
    if (!c->prof.sample(p.tag)) return;
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    (void)hipEventRecord(p.a, st);
    c->prof.pending.push_back(p);
    c->prof.open = true;
}
void gl_prof_end(void *user, hipStream_t st) {
    sr_ctx *c = (sr_ctx *)user;
    if (!c->prof.open || c->prof.pending.empty()) return;
    (void)hipEventRecord(c->prof.pending.back().b, st);
    c->prof.open = false;
}

template <class F>
void exponent_pm1_shift(int shift, uint64_t out[4]);  // (p-1) >> shift
template <>
void exponent_pm1_shift<sr::Goldilocks>(int shift, uint64_t out[4]) {
    out[0] = shift < 64 ? (sr::Goldilocks::P - 1) >> shift : 0;
    out[1] = out[2] = out[3] = 0;
}
template <>
void exponent_pm1_shift<sr::BabyBear>(int shift, uint64_t out[4]) {
    out[0] = shift < 64 ? ((uint64_t)sr::BabyBear::P - 1) >> shift : 0;
    out[1] = out[2] = out[3] = 0;
}
template <>
void exponent_pm1_shift<sr::Stark>(int shift, uint64_t out[4]) {
    const uint64_t pm1[4] = {0, 0, 0, 0x0800000000000011ull};
    int ws = shift / 64, bs = shift % 64;
    for (int i = 0; i < 4; i++) {
        uint64_t lo = i + ws < 4 ? pm1[i + ws] : 0, hi = i + ws + 1 < 4 ? pm1[i + ws + 1] : 0;
        out[i] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
    }
}
template <>
void exponent_pm1_shift<sr::StarkL>(int shift, uint64_t out[4]) {
    exponent_pm1_shift<sr::Stark>(shift, out);
}
template <class F>
void exponent_pm2(uint64_t out[4]);  // p - 2
template <>
void exponent_pm2<sr::Goldilocks>(uint64_t out[4]) {
    out[0] = sr::Goldilocks::P - 2;
    out[1] = out[2] = out[3] = 0;
}
template <>
void exponent_pm2<sr::BabyBear>(uint64_t out[4]) {
    out[0] = (uint64_t)sr::BabyBear::P - 2;
    out[1] = out[2] = out[3] = 0;
}
template <>
void exponent_pm2<sr::Stark>(uint64_t out[4]) {
    out[0] = out[1] = out[2] = ~0ull;
    out[3] = 0x0800000000000010ull;
}
template <>
void exponent_pm2<sr::StarkL>(uint64_t out[4]) {
    exponent_pm2<sr::Stark>(out);
}
template <class F>
constexpr int kappa_bits() {  // mul_tw(x, y) = x * y * 2^-kappa_bits
    return std::is_same<F, sr::Goldilocks>::value ? 0
           : (std::is_same<F, sr::BabyBear>::value ? 32 : (std::is_same<F, sr::StarkL>::value ? sr::StarkL::kTableBits : 256));
}
template <class F>
constexpr int default_log_tile() {
    // Stark: 512 coefficients x 32 B = 16 KiB per tile (two tiles in the fused product): five workgroups per CU.  With 1024
    // coefficients (two per CU) the rows kernel took 3.23 ms at D = 2^12, batch 2^12; with 512 it takes 2.39 ms.
    return (std::is_same<F, sr::Stark>::value || std::is_same<F, sr::StarkL>::value) ? 9 : 12;
}

template <class F>
typename F::elem inv_tw(typename F::elem x) {
    uint64_t e[4];
    exponent_pm2<F>(e);
    return sr::pow_tw<F>(x, e, 4);
}
template <class F>
typename F::elem pow_small(typename F::elem x, uint64_t e) {
    uint64_t w[1] = {e};
    return sr::pow_tw<F>(x, w, 1);
}

template <class F>
int init_pow2(sr_ctx *c) {
    using E = typename F::elem;
    const int k = c->k;
    if (k > F::kTwoAdicity - 1) return fail(SR_E_INVALID, "log2_degree exceeds the field's 2-adicity");
    c->log_tile = default_log_tile<F>();
    if (c->plan.log_tile >= 8 && c->plan.log_tile <= 12) c->log_tile = c->plan.log_tile;  // tuning knob of the generic kernels
    if (k > 2 * c->log_tile && !c->stark_tuned) return fail(SR_E_INVALID, "log2_degree too large for the two-level kernels");
    const size_t d = c->degree;
    size_t extra = 0;
    if constexpr (std::is_same<F, sr::Goldilocks>::value) extra = sr::gl_fast_extra_bytes(k);
    c->table_bytes = 2 * d * sizeof(E) + extra;
    HIP_TRY(hipMalloc(&c->tables, c->table_bytes));

    // psi = g^((p-1)/2D)  (equals ROOTS_OF_UNITY_32[1] for Stark, D = 16)
    uint64_t e[4];
    exponent_pm1_shift<F>(k + 1, e);
    E psi = sr::pow_tw<F>(F::tw_from_u64(F::kGenerator), e, 4);
    E psi_inv = inv_tw<F>(psi);
    std::vector<E> pows(k + 1), ipows(k + 1);
    pows[0] = psi;
    ipows[0] = psi_inv;
    for (int j = 1; j <= k; j++) {
        pows[j] = F::mul_tw(pows[j - 1], pows[j - 1]);
        ipows[j] = F::mul_tw(ipows[j - 1], ipows[j - 1]);
    }
    E *d_pows = nullptr;
    HIP_TRY(hipMalloc(&d_pows, 2 * (k + 1) * sizeof(E)));
    HIP_TRY(hipMemcpy(d_pows, pows.data(), (k + 1) * sizeof(E), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_pows + (k + 1), ipows.data(), (k + 1) * sizeof(E), hipMemcpyHostToDevice));
    E *tw = (E *)c->tables, *itw = tw + d;
    unsigned blocks = (unsigned)((d + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sr::build_tables_kernel<F>, dim3(blocks), dim3(256), 0, c->stream, tw, itw, k, d_pows,
                       d_pows + (k + 1));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_pows));

    // inverse stage-0 constants.  D^-1, and for the fused product D^-1 * kappa / R_b so that
    // mul_tw(A~, B~) (= A B R_b^2 / kappa) comes out as (A B) R_b, the in-memory image of the product.
    E dinv = inv_tw<F>(F::tw_from_u64((uint64_t)d));
    E two = F::tw_from_u64(2);
    E kappa = pow_small<F>(two, kappa_bits<F>());
    E rb_inv = inv_tw<F>(pow_small<F>(two, F::kBoundaryBits));
    E fused = F::mul_tw(dinv, F::mul_tw(kappa, rb_inv));
    E half_turn_inv = k >= 1 ? ipows[k - 1] : F::tw_one();  // psi^-(D/2) = itw[1]
    E s0 = sr::Lazy<F>::table(dinv), s1 = sr::Lazy<F>::table(F::mul_tw(dinv, half_turn_inv));
    E m0 = sr::Lazy<F>::table(fused), m1 = sr::Lazy<F>::table(F::mul_tw(fused, half_turn_inv));
    memcpy(c->inv_scale0, &s0, sizeof(E));
    memcpy(c->inv_scale1, &s1, sizeof(E));
    memcpy(c->mul_scale0, &m0, sizeof(E));
    memcpy(c->mul_scale1, &m1, sizeof(E));
    if constexpr (std::is_same<F, sr::Goldilocks>::value) {
        // tuned-path tables live right behind [tw | itw] so one broadcast of the block ships everything
        if (sr::gl_fast_init(c->gl_fast, k, (const uint64_t *)tw, (const uint64_t *)itw, (uint64_t *)(itw + d),
                             (const uint64_t *)pows.data(), (const uint64_t *)ipows.data(), (uint64_t)dinv,
                             (uint64_t)fused, !(c->plan.flags & SR_PLAN_GL_NO_COLS256), c->plan.chunk_polys, c->stream))
            return fail(SR_E_HIP, "goldilocks fast-path table build failed");
        c->gl_fast.keep_cols = !(c->plan.flags & SR_PLAN_GL_PLAIN_COLS);
        c->gl_fast.split_rows = (c->plan.flags & SR_PLAN_GL_SPLIT_ROWS) != 0 && k > 12;
    }
    return SR_OK;
}

template <class F>
sr::NttParams<F> make_params(const sr_ctx *c, bool fused) {
    using E = typename F::elem;
    sr::NttParams<F> p;
    p.k = c->k;
    p.log_tile = c->log_tile;
    p.s_rows = c->k > c->log_tile ? c->k - c->log_tile : 0;
    p.tw = (const E *)c->tables;
    p.itw = p.tw + c->degree;
    memcpy(&p.scale0, fused ? c->mul_scale0 : c->inv_scale0, sizeof(E));
    memcpy(&p.scale1, fused ? c->mul_scale1 : c->inv_scale1, sizeof(E));
    return p;
}

template <class F>
size_t lds_bytes(const sr_ctx *c, bool two_buffers) {
    return ((size_t)F::kLdsWords * 4u << c->log_tile) * (two_buffers ? 2 : 1);
}

template <class F, int MODE>
int launch_rows(sr_ctx *c, typename F::storage *a, const typename F::storage *b, typename F::storage *out,
                size_t batch, bool fused_scale, hipStream_t st) {
    const size_t n = batch << c->k;
    const size_t tiles = (n + ((size_t)1 << c->log_tile) - 1) >> c->log_tile;
    if (tiles > 0x7FFFFFFFull) return fail(SR_E_INVALID, "batch too large for one launch");
    auto p = make_params<F>(c, fused_scale);
    ProfScope ps(c, st, K_ROWS);
    hipLaunchKernelGGL((sr::rows_kernel<F, MODE>), dim3((unsigned)tiles), dim3(sr::kThreads),
                       lds_bytes<F>(c, MODE == sr::MODE_MUL), st, a, b, out, n, p);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F, int MODE>
int launch_cols(sr_ctx *c, typename F::storage *a, const typename F::storage *src, size_t batch, bool fused_scale, hipStream_t st) {
    const size_t blocks = batch << (c->k - c->log_tile);
    if (blocks > 0x7FFFFFFFull) return fail(SR_E_INVALID, "batch too large for one launch");
    auto p = make_params<F>(c, fused_scale);
    ProfScope ps(c, st, MODE == sr::MODE_FWD ? K_FWD_COLS : K_INV_COLS);
    hipLaunchKernelGGL((sr::cols_kernel<F, MODE>), dim3((unsigned)blocks), dim3(sr::kThreads), lds_bytes<F>(c, false),
                       st, a, src, batch, p);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

template <class F>
int fwd_dev(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    auto *a = reinterpret_cast<typename F::storage *>(d);
    if (c->k == 0 || batch == 0) return SR_OK;
    if (c->k > c->log_tile) {
        int rc = launch_cols<F, sr::MODE_FWD>(c, a, a, batch, false, st);
        if (rc) return rc;
    }
    return launch_rows<F, sr::MODE_FWD>(c, a, nullptr, a, batch, false, st);
}
template <class F>
int inv_dev(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    auto *a = reinterpret_cast<typename F::storage *>(d);
    if (c->k == 0 || batch == 0) return SR_OK;
    int rc = launch_rows<F, sr::MODE_INV>(c, a, nullptr, a, batch, false, st);
    if (rc) return rc;
    if (c->k > c->log_tile) return launch_cols<F, sr::MODE_INV>(c, a, a, batch, false, st);
    return SR_OK;
}
template <class F>
int pointwise_bcast_dev(sr_ctx *c, uint64_t *lhs, const uint64_t *r, size_t n_coeffs, hipStream_t st) {
    if (n_coeffs == 0) return SR_OK;
    ProfScope ps(c, st, K_POINTWISE);
    hipLaunchKernelGGL(sr::pointwise_bcast_kernel<F>, dim3(sr::stream_blocks<F>(n_coeffs)), dim3(256), 0, st,
                       reinterpret_cast<typename F::storage *>(lhs), reinterpret_cast<const typename F::storage *>(r), n_coeffs,
                       (size_t)c->degree - 1);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int pointwise_dev(sr_ctx *c, uint64_t *lhs, const uint64_t *rhs, size_t n_coeffs, hipStream_t st) {
    if (n_coeffs == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n_coeffs);
    ProfScope ps(c, st, K_POINTWISE);
    hipLaunchKernelGGL(sr::pointwise_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<typename F::storage *>(lhs),
                       reinterpret_cast<const typename F::storage *>(rhs), n_coeffs);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int addsub_dev(sr_ctx *c, uint64_t *lhs, const uint64_t *rhs, size_t n_coeffs, bool sub, hipStream_t st) {
    if (n_coeffs == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n_coeffs);
    ProfScope ps(c, st, K_POINTWISE);
    auto *l = reinterpret_cast<typename F::storage *>(lhs);
    auto *r = reinterpret_cast<const typename F::storage *>(rhs);
    if (sub)
        hipLaunchKernelGGL((sr::addsub_kernel<F, true>), dim3((unsigned)blocks), dim3(256), 0, st, l, r, n_coeffs);
    else
        hipLaunchKernelGGL((sr::addsub_kernel<F, false>), dim3((unsigned)blocks), dim3(256), 0, st, l, r, n_coeffs);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int matvec_dev(sr_ctx *c, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols, hipStream_t st) {
    using S = typename F::storage;
    if (nrows == 0) return SR_OK;
    // RB rows share each v[c] slot a lane loads; small problems (few slots x rows) take a smaller RB so that the grid still
    // covers the chip: the largest RB that leaves at least four workgroups per CU
    const size_t chunks = (c->degree + 255) / 256;
    int rb = 4;
    while (rb > 1 && chunks * ((nrows + rb - 1) / rb) < 4096) rb >>= 1;
    const size_t blocks = chunks * ((nrows + rb - 1) / rb);
    if (blocks > 0x7FFFFFFFull) return fail(SR_E_INVALID, "matvec: too many rows for one launch");
    ProfScope ps(c, st, K_OTHER);
    S *py = reinterpret_cast<S *>(y);
    const S *pm = reinterpret_cast<const S *>(m), *pv = reinterpret_cast<const S *>(v);
    const dim3 g((unsigned)blocks), b(256);
    if (rb == 4) hipLaunchKernelGGL((sr::matvec_kernel<F, 4>), g, b, 0, st, py, pm, pv, nrows, ncols, c->k);
    else if (rb == 2) hipLaunchKernelGGL((sr::matvec_kernel<F, 2>), g, b, 0, st, py, pm, pv, nrows, ncols, c->k);
    else hipLaunchKernelGGL((sr::matvec_kernel<F, 1>), g, b, 0, st, py, pm, pv, nrows, ncols, c->k);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int spmv_dev(sr_ctx *c, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr, const uint64_t *v,
             size_t nrows, size_t ncols, hipStream_t st) {
    using S = typename F::storage;
    if (nrows == 0) return SR_OK;
    const size_t blocks = ((c->degree + 255) / 256) * nrows;
    if (blocks > 0x7FFFFFFFull) return fail(SR_E_INVALID, "spmv: too many rows for one launch");
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL((sr::spmv_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<S *>(y), reinterpret_cast<const S *>(vals), cols, row_ptr, reinterpret_cast<const S *>(v),
                       ncols, c->k, c->d_counter + 1);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int matmul_dev(sr_ctx *c, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p, hipStream_t st) {
    using S = typename F::storage;
    if (n == 0 || p == 0) return SR_OK;
    // register block of outputs per lane: RB + CB operand loads feed RB * CB multiply-adds.  BabyBear's accumulators are one
    // register each (8 x 4); Goldilocks' 96-bit sums and Stark's nine limbs cost 12 and 10 (4 x 2).
    constexpr bool bb = std::is_same<F, sr::BabyBear>::value;
    constexpr int RB = bb ? 8 : 4, CB = bb ? 4 : 2;
    const size_t blocks = ((c->degree + 255) / 256) * ((n + RB - 1) / RB) * ((p + CB - 1) / CB);
    if (blocks > 0x7FFFFFFFull) return fail(SR_E_INVALID, "matmul: too many rows or columns for one launch");
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL((sr::matmul_kernel<F, RB, CB>), dim3((unsigned)blocks),
                       dim3(256), 0, st, reinterpret_cast<S *>(y), reinterpret_cast<const S *>(a), reinterpret_cast<const S *>(b),
                       n, m, p, c->k);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
// operand scratch (defined below): grow-only device buffers owned by the context, ordered between streams by an event.  st = the
// stream of the call that needs it (the null stream is a caller's stream like any other); reserving = the call is
// sr_ctx_reserve_scratch, the one place that may grow a buffer after a capture has been seen.
int ensure_scratch(sr_ctx *c, int n_buffers, size_t bytes, hipStream_t st, bool reserving = false);
// remembers that a call arrived on a capturing stream (sr_ctx::scratch_frozen); true while `st` is being captured
bool note_capture(sr_ctx *c, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
        c->scratch_frozen = true;
        return true;
    }
    (void)hipGetLastError();
    return false;
}
int frozen_fail() {
    return fail(SR_E_INVALID, "context scratch would have to grow, but a stream capture has used it (a captured graph holds its address): "
                              "call sr_ctx_reserve_scratch for the largest batch BEFORE capturing (include/stark_rings_hip.h)");
}
int rt_scratch_acquire(sr_ctx *c, hipStream_t st);
int rt_scratch_release(sr_ctx *c, hipStream_t st);
// One use of the shared operand scratch by a call on stream st.  The release (an event recorded on st, which the next user on another
// stream waits for) happens on EVERY exit path once acquire() succeeded -- also after a failed launch, when earlier launches of the
// same call may still be using the scratch.
struct ScratchUse {
    sr_ctx *c;
    hipStream_t st;
    bool held = false;
    ScratchUse(sr_ctx *ctx, hipStream_t s) : c(ctx), st(s) {}
    int acquire() {
        note_capture(c, st);
        const int rc = rt_scratch_acquire(c, st);
        held = rc == SR_OK;
        return rc;
    }
    int release() {
        held = false;
        return rt_scratch_release(c, st);
    }
    ~ScratchUse() {
        if (held) (void)rt_scratch_release(c, st);
    }
};
// device temporary in slot `slot` of the context's grow-only set (see DevBuf below; this one is usable from the internal helpers)
struct DevBufLite {
    sr_ctx *c;
    int slot;
    void *p = nullptr;
    DevBufLite(sr_ctx *c_, int slot_) : c(c_), slot(slot_) {}
    int alloc(size_t bytes, hipStream_t st = nullptr, bool reserving = false) {
        if (bytes == 0) bytes = 8;
        const bool capturing = !reserving && note_capture(c, st);
        if (c->host_tmp_bytes[slot] < bytes) {
            if (capturing || (!reserving && c->scratch_frozen)) return frozen_fail();
            if (c->host_tmp[slot]) {
                HIP_TRY(hipDeviceSynchronize());  // a _dev call on another stream may still be using the smaller buffer
                (void)hipFree(c->host_tmp[slot]);
                c->host_tmp[slot] = nullptr;
                c->host_tmp_bytes[slot] = 0;
            }
            hipError_t e = hipMalloc(&c->host_tmp[slot], bytes);
            if (e != hipSuccess) return fail(SR_E_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
            c->host_tmp_bytes[slot] = bytes;
        }
        p = c->host_tmp[slot];
        return SR_OK;
    }
};
// ring elements of elem_bytes each the operand scratch may hold for a batch (plan cap; at least one element)
size_t scratch_polys(const sr_ctx *c, size_t batch, size_t elem_bytes) {
    const size_t cap = c->plan.scratch_limit_bytes ? (size_t)c->plan.scratch_limit_bytes : ((size_t)16 << 30);
    size_t n = cap / elem_bytes;
    if (n == 0) n = 1;
    if (c->plan.chunk_polys && n > c->plan.chunk_polys) n = c->plan.chunk_polys;
    return n < batch ? n : batch;
}
// tuned Goldilocks path: a large batch runs as eight chunks of launches (unless the plan fixes chunk_polys) -- measured on config 2:
// 18.6 ms as one set of launches, 18.27 / 18.3 / 18.5 ms in chunks of 2048 / 4096 / 1024 ring elements (the inverse column pass
// finds part of the rows kernel's output still in the Infinity Cache), and the operand scratch is an eighth of the batch
// lanes the context runs its chunked products on: the plan's explicit choice, else what the probe settled on (two until it has run)
int effective_lanes(const sr_ctx *c) { return c->plan.lanes ? (int)c->plan.lanes : (c->lanes_choice ? c->lanes_choice : 2); }
// Small batches are the other regime: a set of launches over fewer elements than one lane chunk (64 MiB of coefficients) no longer fills
// the chip, so a batch is never cut below that -- 64 Goldilocks elements of degree 2^16 as eight chunks of eight took 0.339 ms, as ONE
// set of launches 0.085 ms; 128: 0.378 against 0.147 (tools/bench_small_batches.py).
size_t lane_chunk_default(const sr_ctx *c, size_t elem_bytes) {
    (void)c;
    const size_t n = ((size_t)64 << 20) / elem_bytes;
    return n ? n : 1;
}
size_t gl_chunk_polys(const sr_ctx *c, size_t batch) {
    if (c->probe_chunk) return c->probe_chunk < batch ? c->probe_chunk : batch;  // the probe times ONE set of launches of the real chunk size
    size_t chunk = scratch_polys(c, batch, (size_t)8 << c->k);
    const size_t eighth = (batch + 7) / 8;
    if (!c->plan.chunk_polys && eighth >= lane_chunk_default(c, (size_t)8 << c->k) && chunk > eighth) chunk = eighth;
    return chunk;
}
// Two-lane plan of the tuned Goldilocks product (sr::gl_fast_ring_mul_lanes): chunk of a lane in ring elements -- 64 MiB of
// coefficients per scratch buffer by default, so that the four buffers of the two lanes are the 256 MiB of the Infinity Cache --
// capped by the plan's scratch limit (four buffers) and by sr_plan.chunk_polys when set.  Taken when the plan allows lanes
// (sr_plan.lanes != 1), the column passes are cols256 launches and the batch has enough such chunks for the lanes to pay (lanes_pay).
size_t gl_lane_chunk(const sr_ctx *c) {
    const size_t elem = (size_t)8 << c->k;
    size_t chunk = c->plan.chunk_polys ? (size_t)c->plan.chunk_polys : (((size_t)64 << 20) / elem);
    const size_t cap = c->plan.scratch_limit_bytes ? (size_t)c->plan.scratch_limit_bytes : ((size_t)16 << 30);
    if (chunk > cap / (4 * elem)) chunk = cap / (4 * elem);
    return chunk ? chunk : 1;
}
// Two lanes pay from three and a half chunks on: below, one set of launches on the caller's stream is faster -- Goldilocks degree 2^16:
// 256 elements 0.256 against 0.273 ms, 288 0.287 against 0.338, 384 0.385 against 0.413, 448 level, 512 0.522 against 0.498, 640 0.655
// against 0.633; degree 2^20: 24 elements 0.491 against 0.533, 40 level; BabyBear 2^16: 768 0.513 against 0.532, 1 280 0.887 against
// 0.858 (tools/bench_small_batches.py).  (Three chunks on two lanes are two on one and one on the other.)  A plan that fixes
// chunk_polys gets its chunks on the lanes as soon as there are two.
bool lanes_pay(const sr_ctx *c, size_t batch, size_t lane_chunk) {
    return c->plan.chunk_polys ? batch > lane_chunk : 2 * batch >= 7 * lane_chunk;
}
bool gl_use_lanes(const sr_ctx *c, size_t batch) {
    return effective_lanes(c) != 1 && c->k > 12 && c->gl_fast.cols256 && lanes_pay(c, batch, gl_lane_chunk(c));
}
// The stand-alone transforms (two launches per chunk, in place) need more chunks before the lanes pay: at four chunks one set of
// launches on the caller's stream is a quarter faster (512 Goldilocks elements of degree 2^16: 0.169 against 0.222 ms; 32 of degree
// 2^20: 0.210 against 0.268; 1 024 BabyBear: 0.251 against 0.294), at eight the two are level (tools/bench_small_transforms.py).
bool lanes_pay_transform(const sr_ctx *c, size_t batch, size_t lane_chunk) {
    return c->plan.chunk_polys ? batch > lane_chunk : batch >= 8 * lane_chunk;
}
bool gl_use_lanes_transform(const sr_ctx *c, size_t batch) {
    return effective_lanes(c) != 1 && c->k > 12 && c->gl_fast.cols256 && lanes_pay_transform(c, batch, gl_lane_chunk(c));
}
int gl_lanes_init(sr_ctx *c) {
    sr::GlLanes &L = c->gl_lanes;
    if (L.n) return SR_OK;
    HIP_TRY(hipEventCreateWithFlags(&L.fork, hipEventDisableTiming));
    // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share
    // one run one after the other -- measured inside a PyTorch process: 20.4 ms per config-2 batch with two NEW streams for the
    // lanes (the fifth stream of the process landed on a lane's queue), 16.3 ms with a queue each.  Different stream priorities
    // do separate the queues but starve the lower lane (17.9 ms).  So the lanes are the two streams the context owns anyway (the
    // host-pointer pipeline's; idle during a device call, and any earlier work on them simply runs first); a process that holds
    // more streams than hardware queues should raise GPU_MAX_HW_QUEUES (bench.py does).
    L.st[0] = c->stream;
    L.st[1] = c->out_stream;
    for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&L.join[i], hipEventDisableTiming));
    L.n = 2;
    return SR_OK;
}
template <class F>
int ring_mul_dev(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, hipStream_t st) {
    using S = typename F::storage;
    if (batch == 0) return SR_OK;
    if constexpr (!sr::Lazy<F>::value) {  // (a lazy field is never selected for D = 1)
        if (c->k == 0) {  // D = 1: the ring is Fp itself
            if (out != a) HIP_TRY(hipMemcpyAsync(out, a, batch * sizeof(S), hipMemcpyDeviceToDevice, st));
            return pointwise_dev<F>(c, out, b, batch, st);
        }
    }
    if (c->k > c->log_tile) {
        // first (strided) forward stages of both operands: a's go straight to out, b's into the operand scratch, so that
        // neither operand is written (coeff_form.rs:250-258); batches beyond the scratch cap run in chunks
        const size_t elem = sizeof(S) << c->k;
        const size_t chunk = scratch_polys(c, batch, elem);
        if (int rc = ensure_scratch(c, 1, chunk * elem, st)) return rc;
        ScratchUse su(c, st);
        if (int rc = su.acquire()) return rc;
        S *sb = reinterpret_cast<S *>(c->rt_scratch[0]);
        for (size_t e = 0; e < batch; e += chunk) {
            const size_t n = batch - e < chunk ? batch - e : chunk;
            S *o = reinterpret_cast<S *>(out) + (e << c->k);
            int rc = launch_cols<F, sr::MODE_FWD>(c, o, reinterpret_cast<const S *>(a) + (e << c->k), n, true, st);
            if (rc) return rc;
            rc = launch_cols<F, sr::MODE_FWD>(c, sb, reinterpret_cast<const S *>(b) + (e << c->k), n, true, st);
            if (rc) return rc;
            rc = launch_rows<F, sr::MODE_MUL>(c, o, sb, o, n, true, st);
            if (rc) return rc;
            rc = launch_cols<F, sr::MODE_INV>(c, o, o, n, true, st);
            if (rc) return rc;
        }
        return su.release();
    }
    return launch_rows<F, sr::MODE_MUL>(c, reinterpret_cast<S *>(const_cast<uint64_t *>(a)),
                                        reinterpret_cast<const S *>(b), reinterpret_cast<S *>(out), batch, true, st);
}
template <class F>
int reduce_dev(sr_ctx *c, const uint64_t *in, size_t in_len, uint64_t *out, size_t batch, hipStream_t st) {
    if (in_len > 2 * c->degree) return fail(SR_E_INVALID, "reduce: in_len_per_elem > 2D");
    if (batch == 0) return SR_OK;
    size_t n = batch << c->k;
    const unsigned blocks = sr::stream_blocks<F>(n);
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL(sr::reduce_pow2_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const typename F::storage *>(in), in_len,
                       reinterpret_cast<typename F::storage *>(out), c->k, batch);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int fill_dev(sr_ctx *c, uint64_t seed, uint64_t first, size_t n, uint64_t *out, hipStream_t st) {
    if (n == 0) return SR_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL(sr::fill_uniform_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, st, seed, first, n, out);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int count_dev(sr_ctx *c, const uint64_t *d, size_t n, uint64_t *host_count, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), st));
    if (n) {
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(sr::count_noncanonical_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const typename F::storage *>(d), n, c->d_counter);
        HIP_TRY(hipGetLastError());
    }
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c->d_counter, sizeof v, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *host_count = v;
    return SR_OK;
}

template <class F>
sr::rt::Params<F> make_rt_params(const sr_ctx *c, bool fused) {
    using E = typename F::elem;
    sr::rt::Params<F> p;
    p.k = c->k;
    p.c = c->k - 12;
    p.tw = (const E *)c->tables;
    p.itw = p.tw + c->degree;
    memcpy(&p.scale0, fused ? c->mul_scale0 : c->inv_scale0, sizeof(E));
    memcpy(&p.scale1, fused ? c->mul_scale1 : c->inv_scale1, sizeof(E));
    p.no_cols256 = (c->plan.flags & SR_PLAN_RT_NO_COLS256) != 0;
    return p;
}
// grow-only scratch for packed intermediates (allocated on first use; a hipMalloc here is why the very first
// call of a given size is not graph-capturable)
int ensure_scratch(sr_ctx *c, int n_buffers, size_t bytes, hipStream_t st, bool reserving) {
    const bool capturing = !reserving && note_capture(c, st);
    for (int i = 0; i < n_buffers; i++) {
        if (c->rt_scratch_bytes[i] >= bytes) continue;
        if (capturing || (!reserving && c->scratch_frozen)) return frozen_fail();
        HIP_TRY(hipDeviceSynchronize());
        if (c->rt_scratch[i]) HIP_TRY(hipFree(c->rt_scratch[i]));
        c->rt_scratch[i] = nullptr;
        c->rt_scratch_bytes[i] = 0;
        hipError_t e = hipMalloc(&c->rt_scratch[i], bytes);
        if (e != hipSuccess) return fail(SR_E_ALLOC, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
        c->rt_scratch_bytes[i] = bytes;
    }
    return SR_OK;
}
// stream ordering of the shared scratch (callers hold the context's mutex, so the bookkeeping itself is serialised)
int rt_ensure_scratch(sr_ctx *c, int n_buffers, size_t bytes, hipStream_t st) {
    if (c->k <= 12) return SR_OK;
    return ensure_scratch(c, n_buffers, bytes, st);
}
int rt_scratch_acquire(sr_ctx *c, hipStream_t st) {
    if (!c->rt_scratch_free) HIP_TRY(hipEventCreateWithFlags(&c->rt_scratch_free, hipEventDisableTiming));
    if (c->rt_scratch_used && c->rt_scratch_stream != st) HIP_TRY(hipStreamWaitEvent(st, c->rt_scratch_free, 0));
    return SR_OK;
}
int rt_scratch_release(sr_ctx *c, hipStream_t st) {
    if (!c->rt_scratch_free) HIP_TRY(hipEventCreateWithFlags(&c->rt_scratch_free, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->rt_scratch_free, st));
    c->rt_scratch_stream = st;
    c->rt_scratch_used = true;
    return SR_OK;
}
template <class F> size_t rt_lane_chunk(const sr_ctx *c);
template <class F> bool rt_use_lanes(const sr_ctx *c, size_t batch, hipStream_t st);
template <class F> bool rt_use_lanes_transform(const sr_ctx *c, size_t batch, hipStream_t st);
int gl_lanes_init(sr_ctx *c);
// stand-alone transform in chunks on the context's two streams, each lane with its own packed scratch (the two-lane plan of
// rt_ring_mul below; the scratch has the product's size, so one allocation serves both)
template <class F, int DIR, class VB = sr::rt::Boundary>
int rt_transform_lanes(sr_ctx *c, void *d, size_t batch, hipStream_t st) {
    using E = typename F::elem;
    using S = typename sr::rt::View<F, VB>::T;
    const size_t chunk = rt_lane_chunk<F>(c), words = chunk << c->k;
    if (int rc = ensure_scratch(c, 1, 4 * words * sizeof(E), st)) return rc;
    if (int rc = gl_lanes_init(c)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    sr::GlLanes &L = c->gl_lanes;
    E *base = reinterpret_cast<E *>(c->rt_scratch[0]);
    HIP_TRY(hipEventRecord(L.fork, st));  // (nothing is in flight on the lanes yet if one of these three fails)
    for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(L.st[i], L.fork, 0));
    int rc = 0;
    size_t ci = 0;
    for (size_t e = 0; e < batch && !rc; e += chunk, ci++) {
        const int i = (int)(ci & 1);
        const size_t n = batch - e < chunk ? batch - e : chunk;
        S *dc = reinterpret_cast<S *>(d) + (e << c->k);
        rc = DIR == 0 ? sr::rt::fwd<F, VB>(c->rt_hooks, dc, n, make_rt_params<F>(c, false), base + (size_t)(2 * i) * words, L.st[i])
                      : sr::rt::inv<F, VB>(c->rt_hooks, dc, n, make_rt_params<F>(c, false), base + (size_t)(2 * i) * words, L.st[i]);
    }
    for (int i = 0; i < 2; i++) {  // join even after a failed launch
        if (hipEventRecord(L.join[i], L.st[i]) != hipSuccess) rc = 1;
        if (hipStreamWaitEvent(st, L.join[i], 0) != hipSuccess) rc = 1;
    }
    if (int r2 = su.release()) return r2;
    return rc ? fail(SR_E_HIP, "register-tiled launch failed") : SR_OK;
}
// VB: the view of the caller's words -- sr::rt::Boundary (the reference's 8-byte limbs) or sr::rt::PackedStream (the packed-u32 entry points)
template <class F, class VB = sr::rt::Boundary>
int rt_fwd(sr_ctx *c, void *d, size_t batch, hipStream_t st) {
    using E = typename F::elem;
    if (rt_use_lanes_transform<F>(c, batch, st)) return rt_transform_lanes<F, 0, VB>(c, d, batch, st);
    if (int rc = rt_ensure_scratch(c, 1, (batch << c->k) * sizeof(E), st)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    if (sr::rt::fwd<F, VB>(c->rt_hooks, reinterpret_cast<typename sr::rt::View<F, VB>::T *>(d), batch, make_rt_params<F>(c, false),
                           (E *)c->rt_scratch[0], st))
        return fail(SR_E_HIP, "register-tiled launch failed");
    return su.release();
}
template <class F, class VB = sr::rt::Boundary>
int rt_inv(sr_ctx *c, void *d, size_t batch, hipStream_t st) {
    using E = typename F::elem;
    if (rt_use_lanes_transform<F>(c, batch, st)) return rt_transform_lanes<F, 1, VB>(c, d, batch, st);
    if (int rc = rt_ensure_scratch(c, 1, (batch << c->k) * sizeof(E), st)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    if (sr::rt::inv<F, VB>(c->rt_hooks, reinterpret_cast<typename sr::rt::View<F, VB>::T *>(d), batch, make_rt_params<F>(c, false),
                           (E *)c->rt_scratch[0], st))
        return fail(SR_E_HIP, "register-tiled launch failed");
    return su.release();
}
// a large batch runs as eight chunks of launches (unless the plan fixes chunk_polys), like the tuned Goldilocks path: measured on
// config 3 (BabyBear D = 2^16, batch 2^14): 11.68 ms as one set of launches, 11.38 / 11.41 / 11.56 ms in chunks of 2048 / 4096 / 1024
// ring elements, 12.4 ms and worse below 512 -- and the packed scratch is an eighth of the batch
size_t rt_chunk_polys(const sr_ctx *c, size_t batch) {
    if (c->probe_chunk) return c->probe_chunk < batch ? c->probe_chunk : batch;
    if (c->plan.chunk_polys) return c->plan.chunk_polys < batch ? c->plan.chunk_polys : batch;
    const size_t eighth = (batch + 7) / 8;   // never below one lane chunk of packed words (see gl_chunk_polys): 128 BabyBear elements of
    return eighth >= lane_chunk_default(c, (size_t)(c->ring == SR_RING_BABYBEAR_POW2 ? 4 : 8) << c->k) ? eighth : batch;   // degree 2^16 took 0.196 ms in eight chunks, 0.095 as one
}
// two-lane plan of the register-tiled product, as for the tuned Goldilocks path (gl_lane_chunk / gl_fast_ring_mul_lanes): chunks of
// 64 MiB of packed words per scratch buffer on the context's two streams, each lane with its own pair of packed buffers
template <class F>
size_t rt_lane_chunk(const sr_ctx *c) {
    const size_t elem = sizeof(typename F::elem) << c->k;
    size_t chunk = c->plan.chunk_polys ? (size_t)c->plan.chunk_polys : (((size_t)64 << 20) / elem);
    const size_t cap = c->plan.scratch_limit_bytes ? (size_t)c->plan.scratch_limit_bytes : ((size_t)16 << 30);
    if (chunk > cap / (4 * elem)) chunk = cap / (4 * elem);
    return chunk ? chunk : 1;
}
template <class F>
bool rt_use_lanes(const sr_ctx *c, size_t batch, hipStream_t st) {
    return effective_lanes(c) != 1 && c->k > 12 && lanes_pay(c, batch, rt_lane_chunk<F>(c)) && st != c->stream && st != c->out_stream;
}
template <class F>
bool rt_use_lanes_transform(const sr_ctx *c, size_t batch, hipStream_t st) {
    return effective_lanes(c) != 1 && c->k > 12 && lanes_pay_transform(c, batch, rt_lane_chunk<F>(c)) && st != c->stream && st != c->out_stream;
}
template <class F, class VB = sr::rt::Boundary>
int rt_ring_mul(sr_ctx *c, void *out, const void *a, const void *b, size_t batch, hipStream_t st) {
    using S = typename sr::rt::View<F, VB>::T;
    using E = typename F::elem;
    if (batch == 0) return SR_OK;
    if (rt_use_lanes<F>(c, batch, st)) {
        const size_t chunk = rt_lane_chunk<F>(c), words = chunk << c->k;
        if (int rc = ensure_scratch(c, 1, 4 * words * sizeof(E), st)) return rc;
        if (int rc = gl_lanes_init(c)) return rc;
        ScratchUse su(c, st);
        if (int rc = su.acquire()) return rc;
        sr::GlLanes &L = c->gl_lanes;
        E *base = reinterpret_cast<E *>(c->rt_scratch[0]);
        HIP_TRY(hipEventRecord(L.fork, st));  // (nothing is in flight on the lanes yet if one of these three fails)
        for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(L.st[i], L.fork, 0));
        int rc = 0;
        size_t ci = 0;
        for (size_t e = 0; e < batch && !rc; e += chunk, ci++) {
            const int i = (int)(ci & 1);
            const size_t n = batch - e < chunk ? batch - e : chunk;
            const size_t off = e << c->k;
            rc = sr::rt::ring_mul<F, VB>(c->rt_hooks, reinterpret_cast<S *>(out) + off, reinterpret_cast<const S *>(a) + off,
                                         reinterpret_cast<const S *>(b) + off, n, make_rt_params<F>(c, true), base + (size_t)(2 * i) * words,
                                         base + (size_t)(2 * i + 1) * words, L.st[i]);
        }
        for (int i = 0; i < 2; i++) {  // join even after a failed launch
            if (hipEventRecord(L.join[i], L.st[i]) != hipSuccess) rc = 1;
            if (hipStreamWaitEvent(st, L.join[i], 0) != hipSuccess) rc = 1;
        }
        if (int r2 = su.release()) return r2;
        return rc ? fail(SR_E_HIP, "register-tiled launch failed") : SR_OK;
    }
    const size_t chunk = c->k > 12 ? rt_chunk_polys(c, batch) : batch;
    if (int rc = rt_ensure_scratch(c, 2, (chunk << c->k) * sizeof(E), st)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    for (size_t e = 0; e < batch; e += chunk) {
        const size_t n = batch - e < chunk ? batch - e : chunk;
        const size_t off = e << c->k;
        if (sr::rt::ring_mul<F, VB>(c->rt_hooks, reinterpret_cast<S *>(out) + off, reinterpret_cast<const S *>(a) + off,
                                    reinterpret_cast<const S *>(b) + off, n, make_rt_params<F>(c, true), (E *)c->rt_scratch[0],
                                    (E *)c->rt_scratch[1], st))
            return fail(SR_E_HIP, "register-tiled launch failed");
    }
    return su.release();
}

// ---- Stark rings, k >= 4: ntt_stark.hpp ----
int st_fwd(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    if (batch == 0) return SR_OK;
    auto p = make_params<sr::StarkL>(c, false);
    auto *a = reinterpret_cast<sr::U256Storage *>(d);
    auto hook = [&](bool begin) { begin ? gl_prof_begin(c, 0, st) : gl_prof_end(c, st); };
    if (sr::st::fwd_cols(a, a, batch, p, c->stark_one_tile, st, hook)) return fail(SR_E_HIP, "stark strided launch failed");
    ProfScope ps(c, st, K_ROWS);
    if (sr::st::launch_rows<sr::MODE_FWD>(a, nullptr, a, batch, p, c->stark_one_tile, st)) return fail(SR_E_HIP, "stark rows launch failed");
    return SR_OK;
}
int st_inv(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    if (batch == 0) return SR_OK;
    auto p = make_params<sr::StarkL>(c, false);
    auto *a = reinterpret_cast<sr::U256Storage *>(d);
    {
        ProfScope ps(c, st, K_ROWS);
        if (sr::st::launch_rows<sr::MODE_INV>(a, nullptr, a, batch, p, c->stark_one_tile, st)) return fail(SR_E_HIP, "stark rows launch failed");
    }
    auto hook = [&](bool begin) { begin ? gl_prof_begin(c, 2, st) : gl_prof_end(c, st); };
    if (sr::st::inv_cols(a, batch, p, c->stark_one_tile, st, hook)) return fail(SR_E_HIP, "stark strided launch failed");
    return SR_OK;
}
int st_ring_mul(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, hipStream_t st) {
    if (batch == 0) return SR_OK;
    using S = sr::U256Storage;
    auto p = make_params<sr::StarkL>(c, true);
    if (c->stark_one_tile) {  // the ring element is one tile: one launch, a and b only read
        ProfScope ps(c, st, K_ROWS);
        if (sr::st::launch_rows<sr::MODE_MUL>(reinterpret_cast<S *>(const_cast<uint64_t *>(a)), reinterpret_cast<const S *>(b),
                                              reinterpret_cast<S *>(out), batch, p, true, st))
            return fail(SR_E_HIP, "stark rows launch failed");
        return SR_OK;
    }
    // strided stages of both operands first: a's go straight to out, b's into the operand scratch (neither is written)
    const size_t elem = sizeof(S) << c->k;
    const size_t chunk = scratch_polys(c, batch, elem);
    if (int rc = ensure_scratch(c, 1, chunk * elem, st)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    S *sb = reinterpret_cast<S *>(c->rt_scratch[0]);
    auto hook_f = [&](bool begin) { begin ? gl_prof_begin(c, 0, st) : gl_prof_end(c, st); };
    auto hook_i = [&](bool begin) { begin ? gl_prof_begin(c, 2, st) : gl_prof_end(c, st); };
    for (size_t e = 0; e < batch; e += chunk) {
        const size_t n = batch - e < chunk ? batch - e : chunk;
        S *o = reinterpret_cast<S *>(out) + (e << c->k);
        if (sr::st::fwd_cols(o, reinterpret_cast<const S *>(a) + (e << c->k), n, p, false, st, hook_f)) return fail(SR_E_HIP, "stark strided launch failed");
        if (sr::st::fwd_cols(sb, reinterpret_cast<const S *>(b) + (e << c->k), n, p, false, st, hook_f)) return fail(SR_E_HIP, "stark strided launch failed");
        {
            ProfScope ps(c, st, K_ROWS);
            if (sr::st::launch_rows<sr::MODE_MUL>(o, sb, o, n, p, false, st)) return fail(SR_E_HIP, "stark rows launch failed");
        }
        if (sr::st::inv_cols(o, n, p, false, st, hook_i)) return fail(SR_E_HIP, "stark strided launch failed");
    }
    return su.release();
}

bool is_pow2_ring(int ring) { return ring >= SR_RING_GOLDILOCKS_POW2 && ring <= SR_RING_STARK_POW2; }

// dispatch over the field of a pow2 ring
#define DISPATCH_POW2(c, CALL)                                                      \
    switch ((c)->ring) {                                                            \
        case SR_RING_GOLDILOCKS_POW2: { using F = sr::Goldilocks; return CALL; }    \
        case SR_RING_BABYBEAR_POW2: { using F = sr::BabyBear; return CALL; }        \
        case SR_RING_STARK_POW2: { using F = sr::Stark; return CALL; }              \
        default: return fail(SR_E_INVALID, "not a power-of-two ring");              \
    }

// dispatch over the coefficient field of any ring (the small rings' coefficients are Goldilocks / BabyBear elements)
#define DISPATCH_FIELD(c, CALL)                                                                          \
    switch ((c)->ring) {                                                                                 \
        case SR_RING_GOLDILOCKS_POW2: case SR_RING_GOLDILOCKS_24: { using F = sr::Goldilocks; return CALL; } \
        case SR_RING_BABYBEAR_POW2: case SR_RING_BABYBEAR_72: { using F = sr::BabyBear; return CALL; }   \
        case SR_RING_STARK_POW2: { using F = sr::Stark; return CALL; }                                   \
        case SR_RING_FROG_16: { using F = sr::Frog; return CALL; }                                       \
        default: return fail(SR_E_INVALID, "unknown ring");                                              \
    }

extern "C++" {
template <class F>
int decompose_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t b, size_t k, size_t batch, hipStream_t st) {
    using S = typename F::storage;
    const size_t n = batch * c->degree;
    if (n == 0 || k == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n);
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL((sr::dec::decompose_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<S *>(out),
                       reinterpret_cast<const S *>(in), c->degree, batch, b, sr::dec::exact_log2(b), k, c->d_counter + 2);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int recompose_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t b, size_t k, size_t batch_out, hipStream_t st) {
    using S = typename F::storage;
    const size_t n = batch_out * c->degree;
    if (n == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n);
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL((sr::dec::recompose_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<S *>(out),
                       reinterpret_cast<const S *>(in), c->degree, batch_out, b, k);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
}
extern "C++" {
template <class F>
int rot_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t batch, hipStream_t st) {
    using S = typename F::storage;
    const size_t n = batch * c->degree;
    if (n == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n);
    const size_t half = (c->ring == SR_RING_GOLDILOCKS_24 || c->ring == SR_RING_BABYBEAR_72) ? c->degree / 2 : 0;
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL((sr::rot_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<S *>(out),
                       reinterpret_cast<const S *>(in), c->degree, half, batch);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
}
extern "C++" {
template <class F>
int wire_dev(sr_ctx *c, bool ser, void *out, const void *in, const uint64_t *offsets, size_t batch, hipStream_t st) {
    using S = typename F::storage;
    const size_t n = batch * c->degree;
    if (n == 0) return SR_OK;
    const unsigned blocks = sr::stream_blocks<F>(n);
    ProfScope ps(c, st, K_OTHER);
    if (ser)
        hipLaunchKernelGGL((sr::wire::serialize_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st, (uint8_t *)out,
                           reinterpret_cast<const S *>(in), c->degree, batch, offsets, c->d_counter + 3);
    else
        hipLaunchKernelGGL((sr::wire::deserialize_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<S *>(out),
                           (const uint8_t *)in, c->degree, batch, offsets, c->d_counter + 3);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
size_t wire_width() {
    return sr::wire::Codec<F>::W;
}
}
int dev_wire(sr_ctx *c, bool ser, void *out, const void *in, const uint64_t *offsets, size_t batch, hipStream_t st) {
    DISPATCH_FIELD(c, (wire_dev<F>(c, ser, out, in, offsets, batch, st)));
}
size_t wire_coeff_bytes(const sr_ctx *c) {
    switch (c->ring) {
        case SR_RING_BABYBEAR_POW2: case SR_RING_BABYBEAR_72: return wire_width<sr::BabyBear>();
        case SR_RING_STARK_POW2: return wire_width<sr::Stark>();
        case SR_RING_FROG_16: return wire_width<sr::Frog>();
        default: return wire_width<sr::Goldilocks>();
    }
}
int dev_rot(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t batch, hipStream_t st) {
    DISPATCH_FIELD(c, (rot_dev<F>(c, out, in, batch, st)));
}
int check_basis(uint64_t b) {
    if (b < 2) return fail(SR_E_INVALID, "cannot decompose in basis 0 or 1");              // mod.rs:63-66
    if (b & 1) return fail(SR_E_INVALID, "decomposition basis must be even");              // mod.rs:69
    return SR_OK;
}
int dev_decompose(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t b, size_t k, size_t batch, hipStream_t st) {
    DISPATCH_FIELD(c, (decompose_dev<F>(c, out, in, b, k, batch, st)));
}
int dev_recompose(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t b, size_t k, size_t batch_out, hipStream_t st) {
    DISPATCH_FIELD(c, (recompose_dev<F>(c, out, in, b, k, batch_out, st)));
}
uint64_t field_modulus64(const sr_ctx *c) {
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_POW2: case SR_RING_GOLDILOCKS_24: return sr::Goldilocks::P;
        case SR_RING_BABYBEAR_POW2: case SR_RING_BABYBEAR_72: return sr::BabyBear::P;
        default: return sr::Frog::P;
    }
}
// basis = hi * 2^64 + lo with hi != 0 (the reference takes a u128).  One-limb fields: |x| <= (p - 1) / 2 < 2^63 < basis / 2, so digit 0
// is the coefficient itself and every other digit is zero -- a strided copy; Stark: 128-bit restoring division (decompose.hpp).
int dev_decompose_wide(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t lo, uint64_t hi, size_t k, size_t batch, hipStream_t st) {
    if (batch == 0 || k == 0) return SR_OK;
    const size_t w = c->degree * c->limbs * 8;
    ProfScope ps(c, st, K_OTHER);
    if (c->ring == SR_RING_STARK_POW2) {
        const size_t n = batch * c->degree;
        hipLaunchKernelGGL(sr::dec::decompose_wide_kernel, dim3(sr::stream_blocks<sr::Stark>(n)), dim3(256), 0, st,
                           reinterpret_cast<sr::U256Storage *>(out), reinterpret_cast<const sr::U256Storage *>(in), c->degree, batch,
                           sr::dec::U128{lo, hi}, k, c->d_counter + 2);
        HIP_TRY(hipGetLastError());
        return SR_OK;
    }
    HIP_TRY(hipMemsetAsync(out, 0, batch * k * w, st));
    HIP_TRY(hipMemcpy2DAsync(out, k * w, in, w, w, batch, hipMemcpyDeviceToDevice, st));
    return SR_OK;
}
int dev_recompose_wide(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t lo, uint64_t hi, size_t k, size_t batch_out, hipStream_t st) {
    if (batch_out == 0) return SR_OK;
    if (c->ring == SR_RING_STARK_POW2) {
        const size_t n = batch_out * c->degree;
        ProfScope ps(c, st, K_OTHER);
        hipLaunchKernelGGL(sr::dec::recompose_wide_kernel, dim3(sr::stream_blocks<sr::Stark>(n)), dim3(256), 0, st,
                           reinterpret_cast<sr::U256Storage *>(out), reinterpret_cast<const sr::U256Storage *>(in), c->degree, batch_out,
                           sr::dec::U128{lo, hi}, k);
        HIP_TRY(hipGetLastError());
        return SR_OK;
    }
    const unsigned __int128 b = ((unsigned __int128)hi << 64) | lo;  // R::from(b): b mod p, then the narrow Horner kernel
    return dev_recompose(c, out, in, (uint64_t)(b % field_modulus64(c)), k, batch_out, st);
}
int check_basis_wide(uint64_t lo, uint64_t hi) {
    if (hi == 0) return check_basis(lo);
    // decompose_balanced_in_place casts `b as i128` (mod.rs:73): 2^127 and above would be a negative basis there
    if (hi >> 63) return fail(SR_E_INVALID, "decomposition basis >= 2^127 (negative after the reference's cast to i128)");
    if (lo & 1) return fail(SR_E_INVALID, "decomposition basis must be even");
    return SR_OK;
}
int dev_decompose_any(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t lo, uint64_t hi, size_t k, size_t batch, hipStream_t st) {
    return hi ? dev_decompose_wide(c, out, in, lo, hi, k, batch, st) : dev_decompose(c, out, in, lo, k, batch, st);
}
int dev_recompose_any(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t lo, uint64_t hi, size_t k, size_t batch_out, hipStream_t st) {
    return hi ? dev_recompose_wide(c, out, in, lo, hi, k, batch_out, st) : dev_recompose(c, out, in, lo, k, batch_out, st);
}

int ensure_stage(sr_ctx *c, int which, size_t bytes) {
    if (c->stage_bytes[which] >= bytes) return SR_OK;
    if (c->stage[which]) HIP_TRY(hipFree(c->stage[which]));
    c->stage[which] = nullptr;
    c->stage_bytes[which] = 0;
    hipError_t e = hipMalloc(&c->stage[which], bytes);
    if (e != hipSuccess) return fail(SR_E_ALLOC, std::string("hipMalloc staging: ") + hipGetErrorString(e));
    c->stage_bytes[which] = bytes;
    return SR_OK;
}

int check(sr_ctx *c, const void *p0, const void *p1 = (const void *)1, const void *p2 = (const void *)1) {
    if (!c) return fail(SR_E_INVALID, "null context");
    if (!p0 || !p1 || !p2) return fail(SR_E_INVALID, "null buffer");
    return SR_OK;
}

// element counts are bounded so that no byte count or shift below can wrap (64 TiB of ring elements is far past any device)
int check_count(const sr_ctx *c, size_t n_elems, size_t per_elem = 1) {
    const size_t cap = ((size_t)1 << 46) / (c->degree * (size_t)c->limbs * 8);
    if (n_elems > cap || (per_elem > 1 && n_elems && per_elem > cap / n_elems)) return fail(SR_E_INVALID, "element count too large");
    return SR_OK;
}

// ---- per-ring device dispatch (pow2 rings and the reference-native small rings) --------------
int lanes_autoselect(sr_ctx *c, size_t batch);  // sr_plan.lanes = 0: measure once which plan this process's queues favour (below)
int dev_fwd(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_CRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_24) return sr::small_launch(c->small, sr::SMALL_G24_CRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_BABYBEAR_72) return sr::small_launch(c->small, sr::SMALL_B72_CRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->regtile) {
        if (c->ring == SR_RING_BABYBEAR_POW2) return rt_fwd<sr::BabyBear>(c, d, batch, st);
        return rt_fwd<sr::Goldilocks>(c, d, batch, st);
    }
    if (c->ring == SR_RING_GOLDILOCKS_POW2 && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast)) {
        if (gl_use_lanes_transform(c, batch) && st != c->stream && st != c->out_stream) {  // chunks on the two lanes, in place (no scratch)
            if (int rc = gl_lanes_init(c)) return rc;
            c->gl_lanes.chunk = gl_lane_chunk(c);
            return sr::gl_fast_transform_lanes<0>(c->gl_fast, d, c->gl_lanes, batch, st) ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
        }
        return sr::gl_fast_fwd(c->gl_fast, d, batch, st) ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
    }
    if (c->stark_tuned) return st_fwd(c, d, batch, st);
    if (c->stark_lazy) return fwd_dev<sr::StarkL>(c, d, batch, st);
    DISPATCH_POW2(c, (fwd_dev<F>(c, d, batch, st)));
}
int dev_inv(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_ICRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_24) return sr::small_launch(c->small, sr::SMALL_G24_ICRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_BABYBEAR_72) return sr::small_launch(c->small, sr::SMALL_B72_ICRT, d, nullptr, 0, d, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->regtile) {
        if (c->ring == SR_RING_BABYBEAR_POW2) return rt_inv<sr::BabyBear>(c, d, batch, st);
        return rt_inv<sr::Goldilocks>(c, d, batch, st);
    }
    if (c->ring == SR_RING_GOLDILOCKS_POW2 && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast)) {
        if (gl_use_lanes_transform(c, batch) && st != c->stream && st != c->out_stream) {
            if (int rc = gl_lanes_init(c)) return rc;
            c->gl_lanes.chunk = gl_lane_chunk(c);
            return sr::gl_fast_transform_lanes<1>(c->gl_fast, d, c->gl_lanes, batch, st) ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
        }
        return sr::gl_fast_inv(c->gl_fast, d, batch, st) ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
    }
    if (c->stark_tuned) return st_inv(c, d, batch, st);
    if (c->stark_lazy) return inv_dev<sr::StarkL>(c, d, batch, st);
    DISPATCH_POW2(c, (inv_dev<F>(c, d, batch, st)));
}
int dev_pointwise(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_MUL, l, r, 0, l, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_24) return sr::small_launch(c->small, sr::SMALL_G24_MUL, l, r, 0, l, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_BABYBEAR_72) return sr::small_launch(c->small, sr::SMALL_B72_MUL, l, r, 0, l, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    DISPATCH_POW2(c, (pointwise_dev<F>(c, l, r, batch << c->k, st)));
}
// every element of the batch *= ONE ring element r, slot-wise (CRT / NTT form): Matrix<R> *= &R, SparseMatrix<R> *= &R
int dev_mul_elem(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_MULB, l, r, 0, l, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_24 || c->ring == SR_RING_BABYBEAR_72)
        return sr::small_launch_mul_bcast(c->small, c->ring == SR_RING_BABYBEAR_72, l, r, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    DISPATCH_POW2(c, (pointwise_bcast_dev<F>(c, l, r, batch << c->k, st)));
}
int dev_addsub(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, bool sub, hipStream_t st) {
    const size_t n = batch * c->degree;
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_POW2:
        case SR_RING_GOLDILOCKS_24: return addsub_dev<sr::Goldilocks>(c, l, r, n, sub, st);
        case SR_RING_BABYBEAR_POW2:
        case SR_RING_BABYBEAR_72: return addsub_dev<sr::BabyBear>(c, l, r, n, sub, st);
        case SR_RING_FROG_16: return addsub_dev<sr::Frog>(c, l, r, n, sub, st);
        default: return addsub_dev<sr::Stark>(c, l, r, n, sub, st);
    }
}
// ---- unary word-wise operators of RqPoly / RqNTT: Neg, Mul<scalar>, Add<scalar> (every ring id, either form) -------------------
#define DISPATCH_BASE_FIELD(c, CALL)                                                       \
    switch ((c)->ring) {                                                                   \
        case SR_RING_GOLDILOCKS_POW2:                                                      \
        case SR_RING_GOLDILOCKS_24: { using F = sr::Goldilocks; return CALL; }             \
        case SR_RING_BABYBEAR_POW2:                                                        \
        case SR_RING_BABYBEAR_72: { using F = sr::BabyBear; return CALL; }                 \
        case SR_RING_FROG_16: { using F = sr::Frog; return CALL; }                         \
        default: { using F = sr::Stark; return CALL; }                                     \
    }
extern "C++" {
template <class F>
int neg_dev(sr_ctx *c, uint64_t *d, size_t n_words, hipStream_t st) {
    if (n_words == 0) return SR_OK;
    ProfScope ps(c, st, K_POINTWISE);
    hipLaunchKernelGGL(sr::neg_kernel<F>, dim3(sr::stream_blocks<F>(n_words)), dim3(256), 0, st, reinterpret_cast<typename F::storage *>(d), n_words);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int load_scalar(const uint64_t *scalar, typename F::storage *out) {
    typename F::storage s, back;
    memcpy(&s, scalar, sizeof(s));
    const typename F::elem e = F::load(&s);
    F::store(&back, e);  // a BabyBear limb with bits above the low word is no field element's image either
    if (!F::valid(e) || memcmp(&back, &s, sizeof(s)) != 0) return fail(SR_E_INVALID, "scalar is not a canonical Montgomery image (>= p)");
    *out = s;
    return SR_OK;
}
template <class F>
int scale_dev(sr_ctx *c, uint64_t *d, const uint64_t *scalar, size_t n_words, hipStream_t st) {
    typename F::storage s;
    if (int rc = load_scalar<F>(scalar, &s)) return rc;
    if (n_words == 0) return SR_OK;
    ProfScope ps(c, st, K_POINTWISE);
    hipLaunchKernelGGL(sr::scale_kernel<F>, dim3(sr::stream_blocks<F>(n_words)), dim3(256), 0, st, reinterpret_cast<typename F::storage *>(d), n_words, s);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
template <class F>
int add_scalar_dev(sr_ctx *c, uint64_t *d, const uint64_t *scalar, size_t count, size_t stride, hipStream_t st) {
    typename F::storage s;
    if (int rc = load_scalar<F>(scalar, &s)) return rc;
    if (count == 0) return SR_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 0xFFFFFFull) blocks = 0xFFFFFFull;
    ProfScope ps(c, st, K_POINTWISE);
    hipLaunchKernelGGL(sr::add_scalar_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<typename F::storage *>(d), count, stride, s);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
}  // extern "C++"
// base-field words per NTT slot: the extension degree of BaseCRTField (1 for the power-of-two rings)
size_t slot_words(const sr_ctx *c) {
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_24: return 3;
        case SR_RING_BABYBEAR_72: return 9;
        case SR_RING_FROG_16: return 4;
        default: return 1;
    }
}
int dev_neg(sr_ctx *c, uint64_t *d, size_t batch, hipStream_t st) { DISPATCH_BASE_FIELD(c, (neg_dev<F>(c, d, batch * c->degree, st))); }
int dev_scale(sr_ctx *c, uint64_t *d, const uint64_t *scalar, size_t batch, hipStream_t st) {
    DISPATCH_BASE_FIELD(c, (scale_dev<F>(c, d, scalar, batch * c->degree, st)));
}
int dev_add_scalar(sr_ctx *c, uint64_t *d, const uint64_t *scalar, bool ntt_form, size_t batch, hipStream_t st) {
    const size_t stride = ntt_form ? slot_words(c) : c->degree, count = batch * c->degree / stride;
    DISPATCH_BASE_FIELD(c, (add_scalar_dev<F>(c, d, scalar, count, stride, st)));
}
// ---- Sum / Product over a slice of ring elements (coeff_form.rs:507-537, ntt_form.rs:640-670) --------------------------------------
// partial elements one stage of the word-wise fold leaves: about 2^20 lanes in flight, at least 32 elements per lane, one element
// once 32 or fewer remain
size_t fold_parts(size_t n, size_t w) {
    if (n <= 32) return 1;
    size_t r = (n + 31) / 32, cap = ((size_t)1 << 20) / w;
    if (cap < 1) cap = 1;
    return r < cap ? r : cap;
}
size_t fold_tmp_words(size_t w) { return w > ((size_t)1 << 20) ? w : ((size_t)1 << 20); }  // r * w <= max(w, 2^20)
extern "C++" {
template <class F, bool MUL>
int fold_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, hipStream_t st) {
    using S = typename F::storage;
    const size_t w = c->degree;  // storage words (coefficients) per ring element
    DevBufLite t0(c, 8), t1(c, 9);
    ScratchUse su(c, st);
    const S *src = reinterpret_cast<const S *>(in);
    bool tmp_used = false;
    int flip = 0;
    while (true) {
        const size_t r = fold_parts(n, w);
        S *dst = reinterpret_cast<S *>(out);
        if (r > 1) {
            DevBufLite &t = flip ? t1 : t0;
            if (int rc = t.alloc(fold_tmp_words(w) * sizeof(S), st)) return rc;
            if (!tmp_used) {
                if (int rc = su.acquire()) return rc;
                tmp_used = true;
            }
            dst = reinterpret_cast<S *>(t.p);
            flip ^= 1;
        }
        const size_t lanes = r * w;
        {
            ProfScope ps(c, st, K_OTHER);
            hipLaunchKernelGGL((sr::fold_stage_kernel<F, MUL>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, dst, src, w, n, r);
            HIP_TRY(hipGetLastError());
        }
        if (r == 1) break;
        src = dst;
        n = r;
    }
    return tmp_used ? su.release() : SR_OK;
}
// the Montgomery memory image of the base field's 1 (R mod p = mul_boundary(1, R^2))
template <class F>
void mont_one_image(uint64_t out[4]) {
    typename F::storage s;
    F::store(&s, F::mul_boundary(sr::dec::Consts<F>::one(), sr::dec::Consts<F>::r2()));
    memset(out, 0, 32);
    memcpy(out, &s, sizeof(s));
}
}  // extern "C++"
// out = l (.) r slot-wise, out of place, for the three reference-native rings (their slot-product kernels take an output pointer)
int small_slot_mul_to(sr_ctx *c, uint64_t *out, const uint64_t *l, const uint64_t *r, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_MUL, l, r, 0, out, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    const int op = c->ring == SR_RING_GOLDILOCKS_24 ? sr::SMALL_G24_MUL : sr::SMALL_B72_MUL;
    return sr::small_launch(c->small, op, l, r, 0, out, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
}
// n == 0: zero() / one() (one(): component 0 of every slot is the base field's 1)
int fold_empty(sr_ctx *c, uint64_t *out, bool mul, hipStream_t st) {
    const size_t bytes = c->degree * c->limbs * 8;
    HIP_TRY(hipMemsetAsync(out, 0, bytes, st));
    if (!mul) return SR_OK;
    uint64_t one[4];
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_POW2: case SR_RING_GOLDILOCKS_24: mont_one_image<sr::Goldilocks>(one); break;
        case SR_RING_BABYBEAR_POW2: case SR_RING_BABYBEAR_72: mont_one_image<sr::BabyBear>(one); break;
        case SR_RING_FROG_16: mont_one_image<sr::Frog>(one); break;
        default: mont_one_image<sr::Stark>(one); break;
    }
    return dev_add_scalar(c, out, one, true, 1, st);
}
int dev_sum(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, hipStream_t st) {
    if (n == 0) return fold_empty(c, out, false, st);
    DISPATCH_BASE_FIELD(c, (fold_dev<F, false>(c, out, in, n, st)));
}
int dev_product(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, hipStream_t st) {
    if (n == 0) return fold_empty(c, out, true, st);
    if (is_pow2_ring(c->ring)) {  // fully split: the slot product is the base field's product on the memory images
        DISPATCH_POW2(c, (fold_dev<F, true>(c, out, in, n, st)));
    }
    // Fq3 / Fq9 / Fq4 slots: a halving tree of the rings' own slot-product kernels.  First level out of place (the input is only
    // read): tmp[i] = in[i] * in[m + i] for i < h = n / 2, m = n - h, the middle element of an odd n copied; then in place.
    const size_t w = c->degree, bytes = w * 8;
    if (n == 1) {
        HIP_TRY(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, st));
        return SR_OK;
    }
    size_t h = n / 2, m = n - h;
    DevBufLite t(c, 8);
    if (int rc = t.alloc(m * bytes, st)) return rc;
    ScratchUse su(c, st);
    if (int rc = su.acquire()) return rc;
    uint64_t *tmp = (uint64_t *)t.p;
    {
        ProfScope ps(c, st, K_POINTWISE);
        if (int rc = small_slot_mul_to(c, tmp, in, in + m * w, h, st)) return rc;
    }
    if (m > h) HIP_TRY(hipMemcpyAsync(tmp + h * w, in + h * w, bytes, hipMemcpyDeviceToDevice, st));
    n = m;
    while (n > 1) {
        h = n / 2;
        m = n - h;
        ProfScope ps(c, st, K_POINTWISE);
        if (int rc = small_slot_mul_to(c, tmp, tmp, tmp + m * w, h, st)) return rc;
        n = m;
    }
    HIP_TRY(hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToDevice, st));
    return su.release();
}
// the three reference-native rings: one slot (Fq3 / Fq9 / Fq4) per lane, small_linalg.hpp
#define DISPATCH_SLOT(c, CALL)                                                                                   \
    switch ((c)->ring) {                                                                                         \
        case SR_RING_GOLDILOCKS_24: { using SL = sr::SlotG24; const auto &K = (c)->small; return (CALL) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK; } \
        case SR_RING_BABYBEAR_72: { using SL = sr::SlotB72; const auto &K = (c)->small; return (CALL) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK; }   \
        default: { using SL = sr::SlotFrog; const auto &K = (c)->frog; return (CALL) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK; }                       \
    }
int dev_spmv(sr_ctx *c, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr, const uint64_t *v,
             size_t nrows, size_t ncols, hipStream_t st) {
    if (!is_pow2_ring(c->ring)) {
        ProfScope ps(c, st, K_OTHER);
        DISPATCH_SLOT(c, (sr::slot_spmv<SL>(K, y, vals, cols, row_ptr, v, nrows, ncols, c->d_counter + 1, st)));
    }
    if (c->stark_lazy) return spmv_dev<sr::StarkL>(c, y, vals, cols, row_ptr, v, nrows, ncols, st);  // sums of products on 28-bit lazy limbs (stark_lazy.hpp)
    DISPATCH_POW2(c, (spmv_dev<F>(c, y, vals, cols, row_ptr, v, nrows, ncols, st)));
}
int dev_matmul(sr_ctx *c, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p, hipStream_t st) {
    if (!is_pow2_ring(c->ring)) {
        ProfScope ps(c, st, K_OTHER);
        DISPATCH_SLOT(c, (sr::slot_matmul<SL>(K, y, a, b, n, m, p, st)));
    }
    if (c->stark_lazy) return matmul_dev<sr::StarkL>(c, y, a, b, n, m, p, st);  // sums of products on 28-bit lazy limbs (stark_lazy.hpp)
    DISPATCH_POW2(c, (matmul_dev<F>(c, y, a, b, n, m, p, st)));
}
int dev_matvec(sr_ctx *c, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols, hipStream_t st) {
    if (!is_pow2_ring(c->ring)) {
        ProfScope ps(c, st, K_OTHER);
        // few rows: every row is cut into parts (one workgroup each) whose sums meet in a context-owned buffer (small_linalg.hpp)
        const int groups = c->ring == SR_RING_FROG_16 ? 64 : 32;
        const unsigned nsplit = sr::slot_matvec_splits(nrows, ncols, groups);
        uint64_t *part = nullptr;
        // `part` is one buffer per context while _dev calls may come in on any stream: its users are ordered like those of the
        // operand scratch (an event recorded at release, awaited by the next user on another stream)
        ScratchUse su(c, st);
        if (nsplit > 1) {
            DevBufLite pb(c, 7);
            if (int rc = pb.alloc(nrows * nsplit * c->degree * 8, st)) return rc;
            part = (uint64_t *)pb.p;
            if (int rc = su.acquire()) return rc;
        }
        DISPATCH_SLOT(c, (sr::slot_matvec<SL>(K, y, m, v, nrows, ncols, part, nsplit, st)));
    }
    if (c->stark_lazy) return matvec_dev<sr::StarkL>(c, y, m, v, nrows, ncols, st);  // sums of products on 28-bit lazy limbs (stark_lazy.hpp)
    DISPATCH_POW2(c, (matvec_dev<F>(c, y, m, v, nrows, ncols, st)));
}
// ---- plan selection: sr_plan.lanes = 0 means "measure once, keep the winner" ------------------------------------------------
// The two-lane plans (tuned Goldilocks cols256 product, register-tiled product) only pay when the HIP runtime gives each of the
// context's two streams its own hardware queue; which queue a stream gets is decided when the stream is created (GPU_MAX_HW_QUEUES,
// the streams the process already holds), so a host that runs RCCL or a framework beside this library can end up with both lanes
// on one queue -- and then the one-stream plan is the faster one (20.4 against 17.9 ms per config-2 batch).  The library
// therefore times both plans itself, once per context, on the context's real lanes, with a private non-blocking stream standing in
// for the caller's, on device-generated uniform operands in temporary buffers.
//   What is timed (round 4; the round-3 probe timed 16 chunks once, cold, and needed a 7 % fudge): each plan exactly as it will run,
//   on n = up to 32 lane chunks -- after four warm-up calls, four calls back to back, the mean.  Two lanes: every call forks and
//   joins like a real one (its last chunk runs alone on its lane: 1 / 32 of the call here, 1 / 128 of a config-2 batch -- the probe
//   reads about 1 % against the lanes); one stream: sets of launches of the REAL batch's set size, back to back.  The faster plan
//   wins, no margin.
//   Where it runs: ONLY inside sr_ctx_reserve_scratch (a blocking call by contract: it allocates).  A context whose host never
//   reserves runs two lanes, unmeasured -- an asynchronous _dev call never probes (it may be under stream capture, and its latency
//   belongs to the caller).  sr_ctx_plan_in_use reports the outcome.
int dev_ring_mul(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, hipStream_t st);
size_t lanes_chunk_of(const sr_ctx *c) {
    if (c->regtile) return c->ring == SR_RING_BABYBEAR_POW2 ? rt_lane_chunk<sr::BabyBear>(c) : rt_lane_chunk<sr::Goldilocks>(c);
    return gl_lane_chunk(c);
}
bool lanes_candidate(const sr_ctx *c, size_t batch) {
    if (!is_pow2_ring(c->ring) || c->k <= 12) return false;
    const bool gl = c->ring == SR_RING_GOLDILOCKS_POW2 && !c->regtile && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast) && c->gl_fast.cols256;
    if (!gl && !c->regtile) return false;
    return batch > lanes_chunk_of(c);
}
int lanes_autoselect(sr_ctx *c, size_t batch) {
    if (c->plan.lanes || c->lanes_choice || c->probing || !lanes_candidate(c, batch)) return SR_OK;
    const size_t chunk = lanes_chunk_of(c);
    if (batch < 8 * chunk) return SR_OK;  // too small for a steady state: two lanes, unmeasured (effective_lanes)
    // up to 32 chunks per call (2 GiB per buffer)
    size_t nch = batch / chunk;
    if (nch > 32) nch = 32;
    const size_t n = nch * chunk;
    const size_t elem = c->degree * c->limbs * 8, bytes = n * elem;
    void *buf[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    c->probing = true;
    const bool prof_was = c->prof.on;
    c->prof.on = false;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (auto &p : buf) ok = ok && hipMalloc(&p, bytes) == hipSuccess;
    for (auto &e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    double t_plan[2] = {0, 0};  // milliseconds per call of n elements: [0] two lanes, [1] one stream
    constexpr int kWarm = 4, kCalls = 4;
    if (ok) {
        const size_t words = n * c->degree;
        auto fill = [&](void *p, uint64_t seed) {
            switch (c->ring) {
                case SR_RING_GOLDILOCKS_POW2: return fill_dev<sr::Goldilocks>(c, seed, 0, words, (uint64_t *)p, st);
                default: return fill_dev<sr::BabyBear>(c, seed, 0, words, (uint64_t *)p, st);
            }
        };
        ok = fill(buf[1], 0x9E3779B97F4A7C15ull) == SR_OK && fill(buf[2], 0xD1B54A32D192ED03ull) == SR_OK;
        // `calls` products of n elements back to back; the clock the chip holds under this load settles within the warm-up (the step
        // runs at the socket's power cap: a cold 2 ms sample reads 3-4 % fast)
        auto timed = [&](int calls, double &per_call) {
            float ms = 0;
            ok = ok && hipEventRecord(ev[0], st) == hipSuccess;
            for (int i = 0; i < calls; i++)
                ok = ok && dev_ring_mul(c, (uint64_t *)buf[0], (const uint64_t *)buf[1], (const uint64_t *)buf[2], n, st) == SR_OK;
            ok = ok && hipEventRecord(ev[1], st) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess &&
                 hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
            if (ok) per_call = ms / calls;
        };
        for (int plan = 0; plan < 2 && ok; plan++) {
            c->lanes_choice = plan == 0 ? 2 : 1;
            c->probe_chunk = 0;
            if (plan == 1) {  // one stream: sets of launches of the size the REAL batch will run in (an eighth of it by default)
                const size_t real = c->regtile ? rt_chunk_polys(c, batch) : gl_chunk_polys(c, batch);
                c->probe_chunk = real < n ? real : n;
            }
            double warm = 0;
            timed(kWarm, warm);  // allocates the plan's scratch, warms the caches and settles the clock
            timed(kCalls, t_plan[plan]);
        }
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
    }
    // every launch of the probe was joined onto st and st has been waited for: the temporaries are idle (hipFree synchronises by itself)
    if (!ok && st) (void)hipStreamSynchronize(st);
    for (auto &p : buf)
        if (p) (void)hipFree(p);
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    // the probe sized the operand scratch for ITS one-stream run: drop it, the plan that won allocates what it needs
    for (int i = 0; i < 2; i++) {
        if (c->rt_scratch[i]) (void)hipFree(c->rt_scratch[i]);
        c->rt_scratch[i] = nullptr;
        c->rt_scratch_bytes[i] = 0;
    }
    c->rt_scratch_used = false;
    c->probe_chunk = 0;
    c->probing = false;
    c->prof.on = prof_was;
    if (ok && t_plan[0] > 0 && t_plan[1] > 0) {
        c->lanes_probe_ms[0] = t_plan[0];
        c->lanes_probe_ms[1] = t_plan[1];
        c->lanes_probe_elems = n;
        c->lanes_choice = t_plan[1] < t_plan[0] ? 1 : 2;
    } else {
        // no memory for the probe's temporaries (or a launch failed: the real call will report that): keep the default, unmeasured
        (void)hipGetLastError();
        c->lanes_choice = 2;
        g_err.clear();
    }
    return SR_OK;
}

int dev_ring_mul(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) return sr::frog_launch(c->frog, sr::FROG_RINGMUL, a, b, 0, out, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_24) return sr::small_launch(c->small, sr::SMALL_G24_RINGMUL, a, b, 0, out, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->ring == SR_RING_BABYBEAR_72) return sr::small_launch(c->small, sr::SMALL_B72_RINGMUL, a, b, 0, out, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    if (c->regtile) {
        if (c->ring == SR_RING_BABYBEAR_POW2) return rt_ring_mul<sr::BabyBear>(c, out, a, b, batch, st);
        return rt_ring_mul<sr::Goldilocks>(c, out, a, b, batch, st);
    }
    if (c->ring == SR_RING_GOLDILOCKS_POW2 && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast)) {
        // chunks on two internal streams, intermediates in per-lane scratch pairs (not for the host-pointer pipeline, which
        // drives this function on the very streams the lanes are and is bound by PCIe anyway)
        if (gl_use_lanes(c, batch) && st != c->stream && st != c->out_stream) {
            const size_t chunk = gl_lane_chunk(c), words = chunk << c->k;
            if (int rc = ensure_scratch(c, 1, 4 * words * 8, st)) return rc;
            if (int rc = gl_lanes_init(c)) return rc;
            ScratchUse su(c, st);
            if (int rc = su.acquire()) return rc;
            sr::GlLanes &L = c->gl_lanes;
            uint64_t *base = reinterpret_cast<uint64_t *>(c->rt_scratch[0]);
            L.chunk = chunk;
            for (int i = 0; i < 2; i++) {
                L.sa[i] = base + (size_t)(2 * i) * words;
                L.sb[i] = base + (size_t)(2 * i + 1) * words;
            }
            const int rc = sr::gl_fast_ring_mul_lanes(c->gl_fast, out, a, b, L, batch, st);  // joins the lanes itself, also on failure
            if (int r2 = su.release()) return r2;
            return rc ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
        }
        uint64_t *scratch = nullptr;
        size_t chunk = 0;
        ScratchUse su(c, st);
        if (c->k > 12 && batch) {  // b's column stages go through the operand scratch
            chunk = gl_chunk_polys(c, batch);
            if (int rc = ensure_scratch(c, 1, chunk * ((size_t)8 << c->k), st)) return rc;
            if (int rc = su.acquire()) return rc;
            scratch = reinterpret_cast<uint64_t *>(c->rt_scratch[0]);
        }
        if (sr::gl_fast_ring_mul(c->gl_fast, out, a, b, scratch, chunk, batch, st)) return fail(SR_E_HIP, "goldilocks fast-path launch failed");
        return scratch ? su.release() : SR_OK;
    }
    if (c->stark_tuned) return st_ring_mul(c, out, a, b, batch, st);
    if (c->stark_lazy) return ring_mul_dev<sr::StarkL>(c, out, a, b, batch, st);
    DISPATCH_POW2(c, (ring_mul_dev<F>(c, out, a, b, batch, st)));
}
// out = icrt(crt(a) (.) b_ntt): forward transform, slot product and inverse transform on out, one after the other (the
// Goldilocks tuned path fuses them: gl_fast_ring_mul_rhs)
int dev_ring_mul_ntt_rhs(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch, hipStream_t st) {
    if (batch == 0) return SR_OK;
    if (c->ring == SR_RING_GOLDILOCKS_POW2 && !c->regtile && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast) && c->k >= 8) {
        if (gl_use_lanes(c, batch) && st != c->stream && st != c->out_stream) {  // as dev_ring_mul: chunks on the two lanes
            const size_t chunk = gl_lane_chunk(c), words = chunk << c->k;
            if (int rc = ensure_scratch(c, 1, 4 * words * 8, st)) return rc;
            if (int rc = gl_lanes_init(c)) return rc;
            ScratchUse su(c, st);
            if (int rc = su.acquire()) return rc;
            sr::GlLanes &L = c->gl_lanes;
            uint64_t *base = reinterpret_cast<uint64_t *>(c->rt_scratch[0]);
            L.chunk = chunk;
            for (int i = 0; i < 2; i++) {
                L.sa[i] = base + (size_t)(2 * i) * words;
                L.sb[i] = base + (size_t)(2 * i + 1) * words;
            }
            const int rc = sr::gl_fast_ring_mul_rhs_lanes(c->gl_fast, out, a, b_ntt, L, batch, st);
            if (int r2 = su.release()) return r2;
            return rc ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
        }
        return sr::gl_fast_ring_mul_rhs(c->gl_fast, out, a, b_ntt, batch, st) ? fail(SR_E_HIP, "goldilocks fast-path launch failed") : SR_OK;
    }
    if (out != a) HIP_TRY(hipMemcpyAsync(out, a, batch * c->degree * c->limbs * 8, hipMemcpyDeviceToDevice, st));
    if (int rc = dev_fwd(c, out, batch, st)) return rc;
    if (int rc = dev_pointwise(c, out, b_ntt, batch, st)) return rc;
    return dev_inv(c, out, batch, st);
}
int dev_reduce(sr_ctx *c, const uint64_t *in, size_t in_len, uint64_t *out, size_t batch, hipStream_t st) {
    if (c->ring == SR_RING_FROG_16) {
        if (in_len > 2 * c->degree) return fail(SR_E_INVALID, "reduce: in_len_per_elem > 2D");
        return sr::frog_launch(c->frog, sr::FROG_REDUCE, in, nullptr, in_len, out, batch, st) ? fail(SR_E_HIP, "frog-ring launch failed") : SR_OK;
    }
    if (c->ring == SR_RING_GOLDILOCKS_24 || c->ring == SR_RING_BABYBEAR_72) {
        if (in_len > 2 * c->degree) return fail(SR_E_INVALID, "reduce: in_len_per_elem > 2D");
        int op = c->ring == SR_RING_GOLDILOCKS_24 ? sr::SMALL_G24_REDUCE : sr::SMALL_B72_REDUCE;
        return sr::small_launch(c->small, op, in, nullptr, in_len, out, batch, st) ? fail(SR_E_HIP, "small-ring launch failed") : SR_OK;
    }
    DISPATCH_POW2(c, (reduce_dev<F>(c, in, in_len, out, batch, st)));
}

}  // namespace

// ===============================================================================================
extern "C" {

const char *sr_last_error_string(void) { return g_err.c_str(); }
const char *sr_version(void) { return "stark-rings-amd 0.1 (gfx950)"; }

int sr_ctx_create(int ring, int log2_degree, int device, sr_ctx **out) {
    return sr_ctx_create_ex(ring, log2_degree, device, nullptr, out);
}
int sr_ctx_create_ex(int ring, int log2_degree, int device, const sr_plan *plan, sr_ctx **out) {
    if (!out) return fail(SR_E_INVALID, "null out pointer");
    *out = nullptr;
    if (plan) {
        if (plan->flags >> 9) return fail(SR_E_INVALID, "sr_plan: unknown flag bits");
        if (plan->log_tile && (plan->log_tile < 8 || plan->log_tile > 12)) return fail(SR_E_INVALID, "sr_plan: log_tile must be 0 or 8..12");
        if (plan->stark_whole_max && (plan->stark_whole_max < 9 || plan->stark_whole_max > 12))
            return fail(SR_E_INVALID, "sr_plan: stark_whole_max must be 0 or 9..12");
        if (plan->lanes > 2) return fail(SR_E_INVALID, "sr_plan: lanes must be 0 (default), 1 or 2");
    }
    if (ring < SR_RING_GOLDILOCKS_POW2 || ring > SR_RING_FROG_16) return fail(SR_E_INVALID, "unknown ring id");
    if (is_pow2_ring(ring) && (log2_degree < 0 || log2_degree > 24)) return fail(SR_E_INVALID, "log2_degree out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SR_E_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SR_E_NO_DEVICE, "device index out of range");
    DeviceGuard g(device);
    if (!g.ok) return fail(SR_E_HIP, "hipSetDevice failed");
    sr_ctx *c = new sr_ctx();
    c->ring = ring;
    c->device = device;
    if (plan) c->plan = *plan;
    const uint32_t pf = c->plan.flags;
    int rc = SR_OK;
    auto bail = [&](int code) {
        sr_ctx_destroy(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(SR_E_HIP, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(SR_E_HIP, "hipStreamCreate failed"));
    // four words: [0] scratch counter of count_noncanonical, [1] out-of-range column indices seen by spmv, [2] coefficients
    // that needed more digits than padding_size in a decomposition, [3] wire coefficients >= p / misaligned wire offsets
    // ([1]..[3] sticky until read)
    if (hipMalloc(&c->d_counter, 4 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(SR_E_ALLOC, "hipMalloc counter failed"));
    if (hipMemset(c->d_counter, 0, 4 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(SR_E_HIP, "hipMemset counter failed"));
    if (is_pow2_ring(ring)) {
        c->k = log2_degree;
        c->degree = (size_t)1 << log2_degree;
        c->limbs = ring == SR_RING_STARK_POW2 ? 4 : 1;
        switch (ring) {
            case SR_RING_GOLDILOCKS_POW2: rc = init_pow2<sr::Goldilocks>(c); break;
            case SR_RING_BABYBEAR_POW2: rc = init_pow2<sr::BabyBear>(c); break;
            default: {
                // SR_PLAN_STARK_NO_LAZY keeps the transforms on the 8 x 32-bit-limb arithmetic (differential tests); D = 1 has none
                c->stark_lazy = !(pf & SR_PLAN_STARK_NO_LAZY) && log2_degree >= 1;
                c->stark_tuned = c->stark_lazy && !(pf & (SR_PLAN_STARK_GENERIC_ON_LAZY | SR_PLAN_GENERIC_KERNELS)) && sr::st::supported(log2_degree);
                c->stark_one_tile = sr::st::whole(log2_degree, sr::st::whole_max(c->plan.stark_whole_max));
                rc = c->stark_lazy ? init_pow2<sr::StarkL>(c) : init_pow2<sr::Stark>(c);
                break;
            }
        }
        if (rc) return bail(rc);
        if (ring == SR_RING_GOLDILOCKS_POW2) {
            c->fast_goldilocks = !(pf & SR_PLAN_GENERIC_KERNELS);
            c->gl_fast.prof_user = c;
            c->gl_fast.prof_begin = gl_prof_begin;
            c->gl_fast.prof_end = gl_prof_end;
            c->regtile = (pf & SR_PLAN_GL_REGTILE) && c->k >= 12 && c->k <= 24;
        }
        if (ring == SR_RING_BABYBEAR_POW2) {
            c->regtile = !(pf & SR_PLAN_GENERIC_KERNELS) && c->k >= 12 && c->k <= 24;
        }
        c->rt_hooks.user = c;
        c->rt_hooks.begin = gl_prof_begin;
        c->rt_hooks.end = gl_prof_end;
    } else if (ring == SR_RING_FROG_16) {
        c->k = 0;
        c->degree = 16;
        c->limbs = 1;
        sr::frog_init(c->frog);
    } else {
        c->k = 0;
        c->degree = ring == SR_RING_GOLDILOCKS_24 ? 24 : 72;
        c->limbs = 1;
        if (sr::small_init(c->small, ring == SR_RING_GOLDILOCKS_24)) return bail(fail(SR_E_HIP, "small-ring init failed"));
    }
    *out = c;
    return SR_OK;
}

int sr_ctx_destroy(sr_ctx *c) {
    if (!c) return SR_OK;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &p : c->prof.pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    sr::gl_fast_destroy(c->gl_fast);
    sr::small_destroy(c->small);
    if (c->tables) (void)hipFree(c->tables);
    for (int i = 0; i < 4; i++)
        if (c->stage[i]) (void)hipFree(c->stage[i]);
    for (int i = 0; i < 2; i++)
        if (c->rt_scratch[i]) (void)hipFree(c->rt_scratch[i]);
    if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
    if (c->rt_scratch_free) (void)hipEventDestroy(c->rt_scratch_free);
    for (int i = 0; i < sr_ctx::kTmpSlots; i++)
        if (c->host_tmp[i]) (void)hipFree(c->host_tmp[i]);
    if (c->gl_lanes.n) {
        for (int i = 0; i < 2; i++)
            if (c->gl_lanes.join[i]) (void)hipEventDestroy(c->gl_lanes.join[i]);  // the streams are the context's own
        if (c->gl_lanes.fork) (void)hipEventDestroy(c->gl_lanes.fork);
    }
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SR_OK;
}

int sr_ctx_degree(const sr_ctx *c, size_t *degree) {
    if (!c || !degree) return fail(SR_E_INVALID, "null argument");
    *degree = c->degree;
    return SR_OK;
}
int sr_ctx_limbs(const sr_ctx *c, int *limbs) {
    if (!c || !limbs) return fail(SR_E_INVALID, "null argument");
    *limbs = c->limbs;
    return SR_OK;
}
int sr_ctx_twiddle_block(sr_ctx *c, void **dev_ptr, size_t *bytes) {
    if (!c || !dev_ptr || !bytes) return fail(SR_E_INVALID, "null argument");
    if (!is_pow2_ring(c->ring)) return fail(SR_E_INVALID, "small rings keep their constants in kernel arguments");
    *dev_ptr = c->tables;
    *bytes = c->table_bytes;
    return SR_OK;
}
int sr_ctx_twiddles_updated(sr_ctx *c) {
    if (!c) return fail(SR_E_INVALID, "null context");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    HIP_TRY(hipDeviceSynchronize());  // every table (generic and tuned) lives in the one block: nothing to rebuild
    return SR_OK;
}

// One context per device of a node, twiddles built ONCE: device_ids[0] builds the block, every other context's block is
// overwritten with a peer copy of it (over xGMI between the GPUs of one node) -- the single-process form of the one-shot
// twiddle broadcast of SURVEY 8e (processes use RCCL: stark_rings_amd/sharding.py).  No data-path traffic afterwards.
int sr_ctx_create_group(int ring, int log2_degree, const int *device_ids, int n, const sr_plan *plan, sr_ctx **out) {
    if (!device_ids || !out || n <= 0) return fail(SR_E_INVALID, "create_group: null argument or n <= 0");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    auto undo = [&](int rc) {
        for (int i = 0; i < n; i++) {
            if (out[i]) sr_ctx_destroy(out[i]);
            out[i] = nullptr;
        }
        return rc;
    };
    for (int i = 0; i < n; i++)
        if (int rc = sr_ctx_create_ex(ring, log2_degree, device_ids[i], plan, &out[i])) return undo(rc);
    if (!is_pow2_ring(ring)) return SR_OK;  // the small rings keep their constants in kernel arguments
    for (int i = 1; i < n; i++) {
        if (out[i]->table_bytes != out[0]->table_bytes) return undo(fail(SR_E_INVALID, "create_group: table sizes differ"));
        // poison first, so that a failed copy cannot leave a locally built (and therefore plausible) table behind
        DeviceGuard g(out[i]->device);
        if (hipMemset(out[i]->tables, 0xFF, out[i]->table_bytes) != hipSuccess) return undo(fail(SR_E_HIP, "create_group: hipMemset failed"));
        if (hipMemcpyPeer(out[i]->tables, out[i]->device, out[0]->tables, out[0]->device, out[0]->table_bytes) != hipSuccess)
            return undo(fail(SR_E_HIP, "create_group: hipMemcpyPeer of the twiddle block failed"));
        if (int rc = sr_ctx_twiddles_updated(out[i])) return undo(rc);
    }
    return SR_OK;
}
// contiguous, balanced split of `batch` ring elements over n contexts: part i gets [*first, *first + *count)
int sr_shard_range(size_t batch, int n, int i, size_t *first, size_t *count) {
    if (!first || !count || n <= 0 || i < 0 || i >= n) return fail(SR_E_INVALID, "shard_range: bad argument");
    const size_t q = batch / (size_t)n, r = batch % (size_t)n;
    *first = (size_t)i * q + ((size_t)i < r ? (size_t)i : r);
    *count = q + ((size_t)i < r ? 1 : 0);
    return SR_OK;
}

// ---- device-resident entry points ----
int sr_ntt_fwd_batch_dev(sr_ctx *c, uint64_t *d, size_t batch, void *stream) {
    if (int rc = check(c, d)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_fwd(c, d, batch, (hipStream_t)stream);
}
int sr_ntt_inv_batch_dev(sr_ctx *c, uint64_t *d, size_t batch, void *stream) {
    if (int rc = check(c, d)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_inv(c, d, batch, (hipStream_t)stream);
}
int sr_pointwise_mul_batch_dev(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, void *stream) {
    if (int rc = check(c, l, r)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_pointwise(c, l, r, batch, (hipStream_t)stream);
}
int sr_mul_elem_batch_dev(sr_ctx *c, uint64_t *d, const uint64_t *elem, size_t batch, void *stream) {
    if (int rc = check(c, d, elem)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    const uintptr_t w = (uintptr_t)c->degree * c->limbs * 8, pe = (uintptr_t)elem, pd = (uintptr_t)d;
    if (pe + w > pd && pd + batch * w > pe) return fail(SR_E_INVALID, "mul_elem: the element must not lie inside the batch it multiplies");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_mul_elem(c, d, elem, batch, (hipStream_t)stream);
}
static int check_fold(sr_ctx *c, const uint64_t *out, const uint64_t *in, size_t n) {
    if (int rc = check(c, out, n ? (const void *)in : (const void *)1)) return rc;
    if (int rc = check_count(c, n)) return rc;
    const uintptr_t w = (uintptr_t)c->degree * c->limbs * 8, po = (uintptr_t)out, pi = (uintptr_t)in;
    if (n && po + w > pi && pi + n * w > po) return fail(SR_E_INVALID, "sum / product: out must not overlap the input elements");
    return SR_OK;
}
int sr_sum_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, void *stream) {
    if (int rc = check_fold(c, out, in, n)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_sum(c, out, in, n, (hipStream_t)stream);
}
int sr_product_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, void *stream) {
    if (int rc = check_fold(c, out, in, n)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_product(c, out, in, n, (hipStream_t)stream);
}
int sr_add_batch_dev(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, void *stream) {
    if (int rc = check(c, l, r)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_addsub(c, l, r, batch, false, (hipStream_t)stream);
}
int sr_sub_batch_dev(sr_ctx *c, uint64_t *l, const uint64_t *r, size_t batch, void *stream) {
    if (int rc = check(c, l, r)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_addsub(c, l, r, batch, true, (hipStream_t)stream);
}
int sr_neg_batch_dev(sr_ctx *c, uint64_t *d, size_t batch, void *stream) {
    if (int rc = check(c, d)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_neg(c, d, batch, (hipStream_t)stream);
}
int sr_scale_batch_dev(sr_ctx *c, uint64_t *d, const uint64_t *scalar, size_t batch, void *stream) {
    if (int rc = check(c, d, scalar)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_scale(c, d, scalar, batch, (hipStream_t)stream);
}
int sr_add_scalar_batch_dev(sr_ctx *c, uint64_t *d, const uint64_t *scalar, int ntt_form, size_t batch, void *stream) {
    if (int rc = check(c, d, scalar)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_add_scalar(c, d, scalar, ntt_form != 0, batch, (hipStream_t)stream);
}
int sr_matvec_ntt_dev(sr_ctx *c, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols, void *stream) {
    if (int rc = check(c, y, m, v)) return rc;
    if (check_count(c, nrows, ncols)) return SR_E_INVALID;
    if (y == m || y == v) return fail(SR_E_INVALID, "matvec: y must not alias M or v");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_matvec(c, y, m, v, nrows, ncols, (hipStream_t)stream);
}
int sr_spmv_ntt_dev(sr_ctx *c, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr, const uint64_t *v,
                    size_t nrows, size_t ncols, void *stream) {
    if (int rc = check(c, y, row_ptr, v)) return rc;
    if (check_count(c, nrows) || check_count(c, ncols)) return SR_E_INVALID;   // the dense product nrows * ncols is irrelevant here
    if (y == v || y == vals) return fail(SR_E_INVALID, "spmv: y must not alias the matrix or v");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_spmv(c, y, vals, cols, row_ptr, v, nrows, ncols, (hipStream_t)stream);
}
int sr_spmv_bad_index_count(sr_ctx *c, unsigned long long *out, void *stream) {
    if (int rc = check(c, out)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(out, c->d_counter + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(c->d_counter + 1, 0, sizeof(unsigned long long), st));
    HIP_TRY(hipStreamSynchronize(st));
    return SR_OK;
}
int sr_matmul_ntt_dev(sr_ctx *c, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p, void *stream) {
    if (int rc = check(c, y, a, b)) return rc;
    if (check_count(c, n, m) || check_count(c, m, p) || check_count(c, n, p)) return SR_E_INVALID;
    if (y == a || y == b) return fail(SR_E_INVALID, "matmul: y must not alias A or B");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_matmul(c, y, a, b, n, m, p, (hipStream_t)stream);
}

int sr_rot_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t batch, void *stream) {
    if (int rc = check(c, out, in)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if (out == in) return fail(SR_E_INVALID, "rot: out must not alias in");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_rot(c, out, in, batch, (hipStream_t)stream);
}
int sr_rot_batch(sr_ctx *c, uint64_t *data, size_t batch) {
    if (int rc = check(c, data)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t bytes = batch * c->degree * c->limbs * 8;
    if (bytes == 0) return SR_OK;
    if (int rc = ensure_stage(c, 0, bytes)) return rc;
    if (int rc = ensure_stage(c, 1, bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(c->stage[0], data, bytes, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_rot(c, (uint64_t *)c->stage[1], (const uint64_t *)c->stage[0], batch, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(data, c->stage[1], bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
size_t sr_wire_coeff_bytes(const sr_ctx *c) { return c ? wire_coeff_bytes(c) : 0; }
int sr_serialize_batch_dev(sr_ctx *c, uint8_t *wire, const uint64_t *in, const uint64_t *offsets, size_t batch, void *stream) {
    if (int rc = check(c, wire, in)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if ((const void *)wire == (const void *)in) return fail(SR_E_INVALID, "serialize: wire must not alias in");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_wire(c, true, wire, in, offsets, batch, (hipStream_t)stream);
}
int sr_deserialize_batch_dev(sr_ctx *c, uint64_t *out, const uint8_t *wire, const uint64_t *offsets, size_t batch, void *stream) {
    if (int rc = check(c, out, wire)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if ((const void *)wire == (const void *)out) return fail(SR_E_INVALID, "deserialize: out must not alias wire");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_wire(c, false, out, wire, offsets, batch, (hipStream_t)stream);
}
int sr_wire_invalid_count(sr_ctx *c, unsigned long long *out, void *stream) {
    if (int rc = check(c, out)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(out, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), st));
    HIP_TRY(hipStreamSynchronize(st));
    return SR_OK;
}
int sr_decompose_balanced_batch_wide_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis_lo, uint64_t basis_hi,
                                         size_t padding_size, size_t batch, void *stream) {
    if (int rc = check(c, out, in)) return rc;
    if (int rc = check_count(c, batch, padding_size)) return rc;
    if (int rc = check_basis_wide(basis_lo, basis_hi)) return rc;
    if (out == in) return fail(SR_E_INVALID, "decompose: out must not alias in");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_decompose_any(c, out, in, basis_lo, basis_hi, padding_size, batch, (hipStream_t)stream);
}
int sr_decompose_balanced_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch,
                                    void *stream) {
    return sr_decompose_balanced_batch_wide_dev(c, out, in, basis, 0, padding_size, batch, stream);
}
int sr_decompose_overflow_count(sr_ctx *c, unsigned long long *out, void *stream) {
    if (int rc = check(c, out)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(out, c->d_counter + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(c->d_counter + 2, 0, sizeof(unsigned long long), st));
    HIP_TRY(hipStreamSynchronize(st));
    return SR_OK;
}
int sr_recompose_batch_wide_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis_lo, uint64_t basis_hi, size_t padding_size,
                                size_t batch_out, void *stream) {
    if (int rc = check(c, out, in)) return rc;
    if (int rc = check_count(c, batch_out, padding_size)) return rc;
    if (out == in) return fail(SR_E_INVALID, "recompose: out must not alias in");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_recompose_any(c, out, in, basis_lo, basis_hi, padding_size, batch_out, (hipStream_t)stream);
}
int sr_recompose_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch_out,
                           void *stream) {
    return sr_recompose_batch_wide_dev(c, out, in, basis, 0, padding_size, batch_out, stream);
}

// ---- host-pointer variants of the linear-algebra entry points (convenience entry points for callers that hold Vec<..> on the
// host; resident data uses the _dev forms): their device temporaries live in the context and only grow
namespace {
// device temporary of a host-pointer call: slot `slot` of the context's grow-only set (the calls hold the context mutex and end with a
// stream synchronisation, so a slot is free again when the next call starts; nothing is allocated or freed per call once warm)
struct DevBuf {
    sr_ctx *c;
    int slot;
    void *p = nullptr;
    DevBuf(sr_ctx *c_, int slot_) : c(c_), slot(slot_) {}
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        if (c->host_tmp_bytes[slot] < bytes) {
            if (c->host_tmp[slot]) {
                (void)hipFree(c->host_tmp[slot]);
                c->host_tmp[slot] = nullptr;
                c->host_tmp_bytes[slot] = 0;
            }
            hipError_t e = hipMalloc(&c->host_tmp[slot], bytes);
            if (e != hipSuccess) return fail(SR_E_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
            c->host_tmp_bytes[slot] = bytes;
        }
        p = c->host_tmp[slot];
        return SR_OK;
    }
};
}  // namespace
int sr_matmul_ntt(sr_ctx *c, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p) {
    if (int rc = check(c, y, a, b)) return rc;
    if (check_count(c, n, m) || check_count(c, m, p) || check_count(c, n, p)) return SR_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = c->degree * c->limbs * 8;
    DevBuf da(c, 0), db(c, 1), dy(c, 2);
    if (int rc = da.alloc(n * m * w)) return rc;
    if (int rc = db.alloc(m * p * w)) return rc;
    if (int rc = dy.alloc(n * p * w)) return rc;
    if (n * m) HIP_TRY(hipMemcpyAsync(da.p, a, n * m * w, hipMemcpyHostToDevice, c->stream));
    if (m * p) HIP_TRY(hipMemcpyAsync(db.p, b, m * p * w, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_matmul(c, (uint64_t *)dy.p, (const uint64_t *)da.p, (const uint64_t *)db.p, n, m, p, c->stream)) return rc;
    if (n * p) HIP_TRY(hipMemcpyAsync(y, dy.p, n * p * w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_matvec_ntt(sr_ctx *c, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols) {
    if (int rc = check(c, y, m, v)) return rc;
    if (check_count(c, nrows, ncols)) return SR_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = c->degree * c->limbs * 8;
    DevBuf dm(c, 0), dv(c, 1), dy(c, 2);
    if (int rc = dm.alloc(nrows * ncols * w)) return rc;
    if (int rc = dv.alloc(ncols * w)) return rc;
    if (int rc = dy.alloc(nrows * w)) return rc;
    if (nrows * ncols) HIP_TRY(hipMemcpyAsync(dm.p, m, nrows * ncols * w, hipMemcpyHostToDevice, c->stream));
    if (ncols) HIP_TRY(hipMemcpyAsync(dv.p, v, ncols * w, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_matvec(c, (uint64_t *)dy.p, (const uint64_t *)dm.p, (const uint64_t *)dv.p, nrows, ncols, c->stream)) return rc;
    if (nrows) HIP_TRY(hipMemcpyAsync(y, dy.p, nrows * w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_spmv_ntt(sr_ctx *c, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr, const uint64_t *v,
                size_t nrows, size_t ncols) {
    if (int rc = check(c, y, row_ptr, v)) return rc;
    if (check_count(c, nrows) || check_count(c, ncols)) return SR_E_INVALID;
    const size_t nnz = (size_t)row_ptr[nrows];
    if (nnz && (!vals || !cols)) return fail(SR_E_INVALID, "null buffer");
    for (size_t r = 0; r < nrows; r++)
        if (row_ptr[r] > row_ptr[r + 1]) return fail(SR_E_INVALID, "spmv: row_ptr is not monotone");
    for (size_t j = 0; j < nnz; j++)  // the reference indexes v[col] and panics when out of range (sparse_matrix.rs:208)
        if (cols[j] >= ncols) return fail(SR_E_INVALID, "spmv: column index out of range");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = c->degree * c->limbs * 8;
    DevBuf dvals(c, 0), dcols(c, 1), dptr(c, 2), dv(c, 3), dy(c, 4);
    if (int rc = dvals.alloc(nnz * w)) return rc;
    if (int rc = dcols.alloc(nnz * 4)) return rc;
    if (int rc = dptr.alloc((nrows + 1) * 8)) return rc;
    if (int rc = dv.alloc(ncols * w)) return rc;
    if (int rc = dy.alloc(nrows * w)) return rc;
    if (nnz) {
        HIP_TRY(hipMemcpyAsync(dvals.p, vals, nnz * w, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dcols.p, cols, nnz * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(dptr.p, row_ptr, (nrows + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (ncols) HIP_TRY(hipMemcpyAsync(dv.p, v, ncols * w, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_spmv(c, (uint64_t *)dy.p, (const uint64_t *)dvals.p, (const uint32_t *)dcols.p, (const uint64_t *)dptr.p,
                          (const uint64_t *)dv.p, nrows, ncols, c->stream))
        return rc;
    if (nrows) HIP_TRY(hipMemcpyAsync(y, dy.p, nrows * w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
// Sum / Product on host buffers: the n elements go to a device temporary as a whole (like the linear-algebra calls above)
static int host_fold(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n, bool mul) {
    if (int rc = check_fold(c, out, in, n)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = (size_t)c->degree * c->limbs * 8;
    DevBuf di(c, 0), dout(c, 1);
    if (int rc = di.alloc(n * w)) return rc;
    if (int rc = dout.alloc(w)) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(di.p, in, n * w, hipMemcpyHostToDevice, c->stream));
    if (int rc = mul ? dev_product(c, (uint64_t *)dout.p, (const uint64_t *)di.p, n, c->stream) : dev_sum(c, (uint64_t *)dout.p, (const uint64_t *)di.p, n, c->stream)) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, dout.p, w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_sum_batch(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n) { return host_fold(c, out, in, n, false); }
int sr_product_batch(sr_ctx *c, uint64_t *out, const uint64_t *in, size_t n) { return host_fold(c, out, in, n, true); }
int sr_decompose_balanced_batch(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch) {
    return sr_decompose_balanced_batch_wide(c, out, in, basis, 0, padding_size, batch);
}
int sr_decompose_balanced_batch_wide(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, uint64_t basis_hi, size_t padding_size,
                                     size_t batch) {
    if (int rc = check(c, out, in)) return rc;
    if (int rc = check_count(c, batch, padding_size)) return rc;
    if (int rc = check_basis_wide(basis, basis_hi)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = c->degree * c->limbs * 8;
    if (batch == 0 || padding_size == 0) return SR_OK;
    DevBuf din(c, 0), dout(c, 1);
    if (int rc = din.alloc(batch * w)) return rc;
    if (int rc = dout.alloc(batch * padding_size * w)) return rc;
    HIP_TRY(hipMemcpyAsync(din.p, in, batch * w, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counter + 2, 0, sizeof(unsigned long long), c->stream));
    if (int rc = dev_decompose_any(c, (uint64_t *)dout.p, (const uint64_t *)din.p, basis, basis_hi, padding_size, batch, c->stream)) return rc;
    unsigned long long over = 0;
    HIP_TRY(hipMemcpyAsync(&over, c->d_counter + 2, sizeof over, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out, dout.p, batch * padding_size * w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counter + 2, 0, sizeof(unsigned long long), c->stream));
    // the reference indexes out[padding_size] and panics (mod.rs:81-91)
    if (over) return fail(SR_E_INVALID, "decompose: a coefficient needs more than padding_size digits");
    return SR_OK;
}
int sr_serialize_batch(sr_ctx *c, uint8_t *wire, const uint64_t *in, size_t batch) {
    if (int rc = check(c, wire, in)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t n = batch * c->degree;
    if (n == 0) return SR_OK;
    DevBuf din(c, 0), dw(c, 1);
    if (int rc = din.alloc(n * c->limbs * 8)) return rc;
    if (int rc = dw.alloc(n * wire_coeff_bytes(c))) return rc;
    HIP_TRY(hipMemcpyAsync(din.p, in, n * c->limbs * 8, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_wire(c, true, dw.p, din.p, nullptr, batch, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(wire, dw.p, n * wire_coeff_bytes(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_deserialize_batch(sr_ctx *c, uint64_t *out, const uint8_t *wire, size_t batch) {
    if (int rc = check(c, out, wire)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t n = batch * c->degree;
    if (n == 0) return SR_OK;
    DevBuf dout(c, 0), dw(c, 1);
    if (int rc = dout.alloc(n * c->limbs * 8)) return rc;
    if (int rc = dw.alloc(n * wire_coeff_bytes(c))) return rc;
    HIP_TRY(hipMemcpyAsync(dw.p, wire, n * wire_coeff_bytes(c), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
    if (int rc = dev_wire(c, false, dout.p, dw.p, nullptr, batch, c->stream)) return rc;
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, c->d_counter + 3, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
    // Fp::deserialize_with_flags: from_bigint(..) is None for an integer >= p -> SerializationError::InvalidData
    if (bad) return fail(SR_E_INVALID, "deserialize: a coefficient is not below the modulus (InvalidData)");
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * c->limbs * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_recompose_batch(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, size_t padding_size, size_t batch_out) {
    return sr_recompose_batch_wide(c, out, in, basis, 0, padding_size, batch_out);
}
int sr_recompose_batch_wide(sr_ctx *c, uint64_t *out, const uint64_t *in, uint64_t basis, uint64_t basis_hi, size_t padding_size,
                            size_t batch_out) {
    if (int rc = check(c, out, in)) return rc;
    if (int rc = check_count(c, batch_out, padding_size)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t w = c->degree * c->limbs * 8;
    if (batch_out == 0) return SR_OK;
    DevBuf din(c, 0), dout(c, 1);
    if (int rc = din.alloc(batch_out * padding_size * w)) return rc;
    if (int rc = dout.alloc(batch_out * w)) return rc;
    if (padding_size) HIP_TRY(hipMemcpyAsync(din.p, in, batch_out * padding_size * w, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_recompose_any(c, (uint64_t *)dout.p, (const uint64_t *)din.p, basis, basis_hi, padding_size, batch_out, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, batch_out * w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}
int sr_ring_mul_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, void *stream) {
    if (int rc = check(c, out, a, b)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if (b == out) return fail(SR_E_INVALID, "ring_mul: b must not alias out");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_ring_mul(c, out, a, b, batch, (hipStream_t)stream);
}
int sr_ring_mul_ntt_rhs_batch_dev(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch, void *stream) {
    if (int rc = check(c, out, a, b_ntt)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if (b_ntt == out) return fail(SR_E_INVALID, "ring_mul_ntt_rhs: b_ntt must not alias out");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_ring_mul_ntt_rhs(c, out, a, b_ntt, batch, (hipStream_t)stream);
}
int sr_ctx_reserve_scratch(sr_ctx *c, size_t batch) {
    if (!c) return fail(SR_E_INVALID, "null context");
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (batch == 0) return SR_OK;
    {   // the two partial-element buffers of sr_sum_batch_dev / sr_product_batch_dev (a few MiB; the small rings' product tree sizes its
        // own by the slice length on first use)
        const size_t sw = (size_t)c->limbs * 8;  // bytes per coefficient
        for (int slot = 8; slot <= 9; slot++) {
            DevBufLite t(c, slot);
            if (int rc = t.alloc(fold_tmp_words(c->degree) * sw, nullptr, true)) return rc;
        }
    }
    if (!is_pow2_ring(c->ring)) return SR_OK;  // the small rings need no operand scratch
    if (int rc = lanes_autoselect(c, batch)) return rc;      // sr_plan.lanes = 0: settle the plan now, then size the scratch for it
    const size_t elem = c->degree * c->limbs * 8;
    if (c->regtile) {
        // BOTH users of the packed scratch: the ring product (four lane chunks on two lanes, else two buffers of one chunk) and the
        // stand-alone transforms, whose lanes only start at eight chunks (lanes_pay_transform): below that they run on the caller's
        // stream through ONE buffer of the whole batch (a batch of 4 .. 8 lane chunks used to be reserved for the product
        // only, and the first transform afterwards synchronised the device and reallocated)
        const bool bb = c->ring == SR_RING_BABYBEAR_POW2;
        const size_t w = bb ? 4 : 8;
        const bool lanes = bb ? rt_use_lanes<sr::BabyBear>(c, batch, nullptr) : rt_use_lanes<sr::Goldilocks>(c, batch, nullptr);
        const bool lanes_t = bb ? rt_use_lanes_transform<sr::BabyBear>(c, batch, nullptr) : rt_use_lanes_transform<sr::Goldilocks>(c, batch, nullptr);
        const size_t chunk = bb ? rt_lane_chunk<sr::BabyBear>(c) : rt_lane_chunk<sr::Goldilocks>(c);
        const size_t whole = (batch << c->k) * w, four = 4 * (chunk << c->k) * w;
        if (lanes || lanes_t)
            if (int rc = gl_lanes_init(c)) return rc;
        if (c->k <= 12) return SR_OK;
        size_t need0 = lanes ? four : ((c->k > 12 ? rt_chunk_polys(c, batch) : batch) << c->k) * w, need1 = lanes ? 0 : need0;
        const size_t transform = lanes_t ? four : whole;
        if (transform > need0) need0 = transform;
        if (int rc = ensure_scratch(c, 1, need0, nullptr, true)) return rc;
        return need1 ? ensure_scratch(c, 2, need1, nullptr, true) : SR_OK;
    }
    const bool one_launch = c->ring == SR_RING_GOLDILOCKS_POW2 && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast)
                                ? c->k <= 12
                                : (c->stark_tuned ? c->stark_one_tile : c->k <= c->log_tile);
    if (one_launch) return SR_OK;
    const bool gl = c->ring == SR_RING_GOLDILOCKS_POW2 && c->fast_goldilocks && sr::gl_fast_supported(c->gl_fast);
    if (gl && gl_use_lanes(c, batch)) {
        if (int rc = gl_lanes_init(c)) return rc;
        return ensure_scratch(c, 1, 4 * gl_lane_chunk(c) * elem, nullptr, true);
    }
    return ensure_scratch(c, 1, (gl ? gl_chunk_polys(c, batch) : scratch_polys(c, batch, elem)) * elem, nullptr, true);
}
// ---- packed-u32 boundary (BabyBear power-of-two rings; csrc/packed32.hpp) ----------------------------------------------------
extern "C++" {
namespace {
int check_packed(sr_ctx *c) {
    if (c->ring != SR_RING_BABYBEAR_POW2) return fail(SR_E_INVALID, "packed32 entry points: BabyBear power-of-two rings only");
    return SR_OK;
}
int launch_pack32(sr_ctx *c, uint32_t *out, const uint64_t *in, size_t n, hipStream_t st) {
    if (n == 0) return SR_OK;
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL(sr::p32::pack32_kernel, dim3(sr::p32::blocks_for(n)), dim3(256), 0, st, out, in, n);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
int launch_unpack32(sr_ctx *c, uint64_t *out, const uint32_t *in, size_t n, hipStream_t st) {
    if (n == 0) return SR_OK;
    ProfScope ps(c, st, K_OTHER);
    hipLaunchKernelGGL(sr::p32::unpack32_kernel, dim3(sr::p32::blocks_for(n)), dim3(256), 0, st, out, in, n);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
// D < 4096 has no register-tiled path: such (small) batches are widened into the context's staging buffers, run through the 8-byte
// kernels and narrowed again -- same values, only the traffic advantage is lost where it does not matter
template <class Fn>
int packed_via_wide(sr_ctx *c, uint32_t *out, const uint32_t *a, const uint32_t *b, size_t batch, hipStream_t st, Fn run) {
    const size_t n = batch << c->k;
    DevBufLite wa(c, 5), wb(c, 6);
    if (int rc = wa.alloc(n * 8, st)) return rc;
    if (b)
        if (int rc = wb.alloc(n * 8, st)) return rc;
    ScratchUse su(c, st);  // the staging buffers belong to the context, the call may arrive on any stream: ordered like the operand scratch
    if (int rc = su.acquire()) return rc;
    if (int rc = launch_unpack32(c, (uint64_t *)wa.p, a, n, st)) return rc;
    if (b)
        if (int rc = launch_unpack32(c, (uint64_t *)wb.p, b, n, st)) return rc;
    if (int rc = run((uint64_t *)wa.p, (uint64_t *)wb.p)) return rc;
    return launch_pack32(c, out, (const uint64_t *)wa.p, n, st);
}
}  // namespace
}  // extern "C++"
int sr_pack32_batch_dev(sr_ctx *c, uint32_t *d_out, const uint64_t *d_in, size_t batch, void *stream) {
    if (int rc = check(c, d_out, d_in)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return launch_pack32(c, d_out, d_in, batch << c->k, (hipStream_t)stream);
}
int sr_unpack32_batch_dev(sr_ctx *c, uint64_t *d_out, const uint32_t *d_in, size_t batch, void *stream) {
    if (int rc = check(c, d_out, d_in)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return launch_unpack32(c, d_out, d_in, batch << c->k, (hipStream_t)stream);
}
int sr_ntt_fwd_packed32_batch_dev(sr_ctx *c, uint32_t *d, size_t batch, void *stream) {
    if (int rc = check(c, d)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0) return SR_OK;
    if (c->regtile) {
        return rt_fwd<sr::BabyBear, sr::rt::PackedStream>(c, d, batch, st);
    }
    return packed_via_wide(c, d, d, nullptr, batch, st, [&](uint64_t *wa, uint64_t *) { return dev_fwd(c, wa, batch, st); });
}
int sr_ntt_inv_packed32_batch_dev(sr_ctx *c, uint32_t *d, size_t batch, void *stream) {
    if (int rc = check(c, d)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0) return SR_OK;
    if (c->regtile) {
        return rt_inv<sr::BabyBear, sr::rt::PackedStream>(c, d, batch, st);
    }
    return packed_via_wide(c, d, d, nullptr, batch, st, [&](uint64_t *wa, uint64_t *) { return dev_inv(c, wa, batch, st); });
}
int sr_ring_mul_packed32_batch_dev(sr_ctx *c, uint32_t *d_out, const uint32_t *d_a, const uint32_t *d_b, size_t batch, void *stream) {
    if (int rc = check(c, d_out, d_a, d_b)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if (d_b == d_out) return fail(SR_E_INVALID, "ring_mul: b must not alias out");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0) return SR_OK;
    if (c->regtile) {
        return rt_ring_mul<sr::BabyBear, sr::rt::PackedStream>(c, d_out, d_a, d_b, batch, st);
    }
    return packed_via_wide(c, d_out, d_a, d_b, batch, st, [&](uint64_t *wa, uint64_t *wb) { return dev_ring_mul(c, wa, wa, wb, batch, st); });
}
static int packed_elementwise(sr_ctx *c, uint32_t *l, const uint32_t *r, size_t batch, void *stream, int op) {
    if (int rc = check(c, l, r)) return rc;
    if (int rc = check_packed(c)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = batch << c->k;
    if (n == 0) return SR_OK;
    ProfScope ps(c, st, K_POINTWISE);
    const dim3 gr(sr::p32::blocks_for(n)), bl(256);
    if (op == 0) hipLaunchKernelGGL(sr::p32::elementwise32_kernel<0>, gr, bl, 0, st, l, r, n);
    else if (op == 1) hipLaunchKernelGGL(sr::p32::elementwise32_kernel<1>, gr, bl, 0, st, l, r, n);
    else hipLaunchKernelGGL(sr::p32::elementwise32_kernel<2>, gr, bl, 0, st, l, r, n);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}
int sr_pointwise_mul_packed32_batch_dev(sr_ctx *c, uint32_t *l, const uint32_t *r, size_t batch, void *stream) {
    return packed_elementwise(c, l, r, batch, stream, 0);
}
int sr_add_packed32_batch_dev(sr_ctx *c, uint32_t *l, const uint32_t *r, size_t batch, void *stream) {
    return packed_elementwise(c, l, r, batch, stream, 1);
}
int sr_sub_packed32_batch_dev(sr_ctx *c, uint32_t *l, const uint32_t *r, size_t batch, void *stream) {
    return packed_elementwise(c, l, r, batch, stream, 2);
}
int sr_ctx_plan_in_use(sr_ctx *c, sr_plan *plan, double probe_ms[2], size_t *probe_elems) {
    if (!c || !plan) return fail(SR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    *plan = c->plan;
    if (!plan->lanes) plan->lanes = (uint32_t)c->lanes_choice;  // 0 = auto and not settled yet (no chunked product seen)
    if (probe_ms) {
        probe_ms[0] = c->lanes_probe_ms[0];
        probe_ms[1] = c->lanes_probe_ms[1];
    }
    if (probe_elems) *probe_elems = c->lanes_probe_elems;
    return SR_OK;
}
int sr_reduce_batch_dev(sr_ctx *c, const uint64_t *in, size_t in_len, uint64_t *out, size_t batch, void *stream) {
    if (int rc = check(c, in, out)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return dev_reduce(c, in, in_len, out, batch, (hipStream_t)stream);
}
int sr_fill_uniform_dev(sr_ctx *c, uint64_t seed, uint64_t first, size_t n, uint64_t *out, void *stream) {
    if (int rc = check(c, out)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_POW2:
        case SR_RING_GOLDILOCKS_24: return fill_dev<sr::Goldilocks>(c, seed, first, n, out, st);
        case SR_RING_BABYBEAR_POW2:
        case SR_RING_BABYBEAR_72: return fill_dev<sr::BabyBear>(c, seed, first, n, out, st);
        case SR_RING_FROG_16: return fill_dev<sr::Frog>(c, seed, first, n, out, st);
        default: return fill_dev<sr::Stark>(c, seed, first, n, out, st);
    }
}
int sr_count_noncanonical_dev(sr_ctx *c, const uint64_t *d, size_t n, uint64_t *host_count, void *stream) {
    if (int rc = check(c, d, host_count)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    switch (c->ring) {
        case SR_RING_GOLDILOCKS_POW2:
        case SR_RING_GOLDILOCKS_24: return count_dev<sr::Goldilocks>(c, d, n, host_count, st);
        case SR_RING_BABYBEAR_POW2:
        case SR_RING_BABYBEAR_72: return count_dev<sr::BabyBear>(c, d, n, host_count, st);
        case SR_RING_FROG_16: return count_dev<sr::Frog>(c, d, n, host_count, st);
        default: return count_dev<sr::Stark>(c, d, n, host_count, st);
    }
}

// ---- host-buffer entry points: stage, run, copy back ----
// Host-pointer batches: results = compute(a[, b]) element-wise over the batch, through device staging buffers.
// Small batches: copy in, compute, copy out on the context's stream.  Large batches are cut into chunks (default 128 MiB per
// operand, SR_HOST_CHUNK_MB) that alternate between two sets of staging buffers: the calling thread copies chunk i in and
// launches it while a helper thread copies chunk i-1 out on a second stream, so the two PCIe directions overlap and the device
// memory needed no longer grows with the batch.  compute(s0, s1, n, stream) works in place on s0 (n elements).
extern "C++" {
static int host_pipeline(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch,
                         const std::function<int(uint64_t *, uint64_t *, size_t, hipStream_t)> &compute) {
    const size_t elem_bytes = c->degree * c->limbs * 8;
    const size_t bytes = batch * elem_bytes;
    if (bytes == 0) return SR_OK;
    // The caller's buffers are ordinary (pageable) host memory -- a Rust Vec -- and stay that way: round 3 registered them with the
    // HIP runtime for the duration of the call (hipHostRegister), which measured inside the noise (69.8-78.3 against 66.9-73.0 GB/s:
    // the pipeline is PCIe-bound either way) and had two lifetime hazards (a read-only operand shared by two contexts on two
    // threads was unregistered by whichever call returned first while the other GPU's DMA still read it; an early error return
    // unregistered pages under an in-flight copy).  Removed in round 4; SR_PLAN_NO_HOST_PIN is accepted and has no effect.
    const size_t chunk_mb = c->plan.host_chunk_mb ? c->plan.host_chunk_mb : 128;
    size_t chunk = (chunk_mb << 20) / elem_bytes;
    if (chunk == 0) chunk = 1;
    if (batch <= 2 * chunk) {  // one shot
        if (int rc = ensure_stage(c, 0, bytes)) return rc;
        if (b)
            if (int rc = ensure_stage(c, 1, bytes)) return rc;
        HIP_TRY(hipMemcpyAsync(c->stage[0], a, bytes, hipMemcpyHostToDevice, c->stream));
        if (b) HIP_TRY(hipMemcpyAsync(c->stage[1], b, bytes, hipMemcpyHostToDevice, c->stream));
        if (int rc = compute((uint64_t *)c->stage[0], (uint64_t *)c->stage[1], batch, c->stream)) return rc;
        HIP_TRY(hipMemcpyAsync(out, c->stage[0], bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return SR_OK;
    }
    const size_t chunk_bytes = chunk * elem_bytes;
    for (int i = 0; i < 4; i++)
        if (b || (i & 1) == 0)
            if (int rc = ensure_stage(c, i, chunk_bytes)) return rc;
    hipEvent_t ready[2];
    for (auto &e : ready) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    struct Job {
        const void *src;
        uint64_t *dst;
        size_t bytes;
    } jobs[2];
    std::mutex m;
    std::condition_variable cv;
    size_t submitted = 0, done = 0;
    bool stop = false;
    hipError_t worker_err = hipSuccess;
    const int device = c->device;
    hipStream_t out_stream = c->out_stream;
    std::thread worker([&] {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            int lane;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return done < submitted || stop; });
                if (done == submitted) return;
                lane = (int)(done & 1);
                j = jobs[lane];
            }
            hipError_t e = hipStreamWaitEvent(out_stream, ready[lane], 0);
            if (e == hipSuccess) e = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, out_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(out_stream);
            {
                std::lock_guard<std::mutex> lk(m);
                if (e != hipSuccess && worker_err == hipSuccess) worker_err = e;
                done++;
            }
            cv.notify_all();
        }
    });
    int rc = SR_OK;
    hipError_t main_err = hipSuccess;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    for (size_t i = 0; i < nchunks && rc == SR_OK && main_err == hipSuccess; i++) {
        const int lane = (int)(i & 1);
        const size_t first = i * chunk, n = batch - first < chunk ? batch - first : chunk;
        {
            std::unique_lock<std::mutex> lk(m);  // the lane's buffers are free once chunk i - 2 has been copied out
            cv.wait(lk, [&] { return i < 2 || done + 1 >= i; });
        }
        uint64_t *s0 = (uint64_t *)c->stage[2 * lane], *s1 = (uint64_t *)c->stage[2 * lane + 1];
        main_err = hipMemcpyAsync(s0, a + first * (elem_bytes / 8), n * elem_bytes, hipMemcpyHostToDevice, c->stream);
        if (main_err == hipSuccess && b)
            main_err = hipMemcpyAsync(s1, b + first * (elem_bytes / 8), n * elem_bytes, hipMemcpyHostToDevice, c->stream);
        if (main_err != hipSuccess) break;
        rc = compute(s0, s1, n, c->stream);
        if (rc != SR_OK) break;
        main_err = hipEventRecord(ready[lane], c->stream);
        if (main_err != hipSuccess) break;
        {
            std::lock_guard<std::mutex> lk(m);
            jobs[lane] = Job{s0, out + first * (elem_bytes / 8), n * elem_bytes};
            submitted++;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done == submitted; });
        stop = true;
    }
    cv.notify_all();
    worker.join();
    (void)hipStreamSynchronize(c->stream);
    for (auto &e : ready) (void)hipEventDestroy(e);
    if (rc != SR_OK) return rc;
    if (main_err != hipSuccess) return fail(SR_E_HIP, std::string("host pipeline: ") + hipGetErrorString(main_err));
    if (worker_err != hipSuccess) return fail(SR_E_HIP, std::string("host pipeline (copy out): ") + hipGetErrorString(worker_err));
    return SR_OK;
}
}  // extern "C++"
static int host_inplace(sr_ctx *c, uint64_t *data, size_t batch, bool fwd) {
    if (int rc = check(c, data)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return host_pipeline(c, data, data, nullptr, batch, [&](uint64_t *s0, uint64_t *, size_t n, hipStream_t st) {
        return fwd ? dev_fwd(c, s0, n, st) : dev_inv(c, s0, n, st);
    });
}
// the unary operators on host buffers: op 0 neg, 1 scale, 2 add scalar (coefficient form), 3 add scalar (NTT form)
static int host_unary(sr_ctx *c, uint64_t *data, const uint64_t *scalar, size_t batch, int op) {
    if (int rc = check(c, data, op ? (const void *)scalar : (const void *)1)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return host_pipeline(c, data, data, nullptr, batch, [&](uint64_t *s0, uint64_t *, size_t n, hipStream_t st) {
        return op == 0 ? dev_neg(c, s0, n, st) : op == 1 ? dev_scale(c, s0, scalar, n, st) : dev_add_scalar(c, s0, scalar, op == 3, n, st);
    });
}
int sr_neg_batch(sr_ctx *c, uint64_t *data, size_t batch) { return host_unary(c, data, nullptr, batch, 0); }
int sr_scale_batch(sr_ctx *c, uint64_t *data, const uint64_t *scalar, size_t batch) { return host_unary(c, data, scalar, batch, 1); }
int sr_add_scalar_batch(sr_ctx *c, uint64_t *data, const uint64_t *scalar, int ntt_form, size_t batch) {
    return host_unary(c, data, scalar, batch, ntt_form ? 3 : 2);
}
// host buffers: the one element goes to a device temporary of its own, the batch through the staged pipeline
int sr_mul_elem_batch(sr_ctx *c, uint64_t *data, const uint64_t *elem, size_t batch) {
    if (int rc = check(c, data, elem)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t bytes = (size_t)c->degree * c->limbs * 8;
    uint64_t *dr = nullptr;
    HIP_TRY(hipMalloc((void **)&dr, bytes));
    hipError_t e = hipMemcpy(dr, elem, bytes, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? host_pipeline(c, data, data, nullptr, batch, [&](uint64_t *s0, uint64_t *, size_t n, hipStream_t st) {
        return dev_mul_elem(c, s0, dr, n, st);
    }) : fail(SR_E_HIP, std::string("mul_elem: ") + hipGetErrorString(e));
    (void)hipFree(dr);   // host_pipeline returned with its stream synchronised
    return rc;
}
int sr_ntt_fwd_batch(sr_ctx *c, uint64_t *data, size_t batch) { return host_inplace(c, data, batch, true); }
int sr_ntt_inv_batch(sr_ctx *c, uint64_t *data, size_t batch) { return host_inplace(c, data, batch, false); }

enum { HB_POINTWISE = 0, HB_RING_MUL = 1, HB_ADD = 2, HB_SUB = 3, HB_RING_MUL_NTT_RHS = 4 };
static int host_binary(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch, int op) {
    if (int rc = check(c, out, a, b)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    return host_pipeline(c, out, a, b, batch, [&](uint64_t *s0, uint64_t *s1, size_t n, hipStream_t st) {
        return op == HB_RING_MUL ? dev_ring_mul(c, s0, s0, s1, n, st)
               : op == HB_RING_MUL_NTT_RHS ? dev_ring_mul_ntt_rhs(c, s0, s0, s1, n, st)
               : op == HB_POINTWISE ? dev_pointwise(c, s0, s1, n, st)
                                    : dev_addsub(c, s0, s1, n, op == HB_SUB, st);
    });
}
int sr_pointwise_mul_batch(sr_ctx *c, uint64_t *lhs, const uint64_t *rhs, size_t batch) {
    return host_binary(c, lhs, lhs, rhs, batch, HB_POINTWISE);
}
int sr_add_batch(sr_ctx *c, uint64_t *lhs, const uint64_t *rhs, size_t batch) { return host_binary(c, lhs, lhs, rhs, batch, HB_ADD); }
int sr_sub_batch(sr_ctx *c, uint64_t *lhs, const uint64_t *rhs, size_t batch) { return host_binary(c, lhs, lhs, rhs, batch, HB_SUB); }
int sr_ring_mul_batch(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t batch) {
    return host_binary(c, out, a, b, batch, HB_RING_MUL);
}
int sr_ring_mul_ntt_rhs_batch(sr_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch) {
    return host_binary(c, out, a, b_ntt, batch, HB_RING_MUL_NTT_RHS);
}
int sr_reduce_batch(sr_ctx *c, const uint64_t *in, size_t in_len, uint64_t *out, size_t batch) {
    if (int rc = check(c, in, out)) return rc;
    if (int rc = check_count(c, batch)) return rc;
    if (in_len > 2 * c->degree) return fail(SR_E_INVALID, "reduce: in_len_per_elem > 2D");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    size_t in_bytes = batch * in_len * c->limbs * 8, out_bytes = batch * c->degree * c->limbs * 8;
    if (out_bytes == 0) return SR_OK;
    if (int rc = ensure_stage(c, 0, in_bytes ? in_bytes : 8)) return rc;
    if (int rc = ensure_stage(c, 1, out_bytes)) return rc;
    if (in_bytes) HIP_TRY(hipMemcpyAsync(c->stage[0], in, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (int rc = dev_reduce(c, (const uint64_t *)c->stage[0], in_len, (uint64_t *)c->stage[1], batch, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->stage[1], out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SR_OK;
}

// ---- host-side self-test hook: runs the SAME __host__ __device__ field routines the kernels use,
// on the host, one scalar operation per call.  Lets the CPU test-suite check fields.hpp without a
// GPU.  Not a compute path: no batch entry point routes through it.
// op: 0 add, 1 sub, 2 mul_boundary (a*b*R_b^-1 on in-memory images), 3 mul_tw, 4 tw_from_u64(a[0])
extern "C++" {
template <class F>
static int selftest_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    using S = typename F::storage;
    typename F::elem x = F::load(reinterpret_cast<const S *>(a)), y = F::load(reinterpret_cast<const S *>(b)), r;
    switch (op) {
        case 0: r = F::add(x, y); break;
        case 1: r = F::sub(x, y); break;
        case 2: r = F::mul_boundary(x, y); break;
        case 3: r = F::mul_tw(x, y); break;
        case 4: r = F::tw_from_u64(a[0]); break;
        default: return fail(SR_E_INVALID, "selftest: unknown op");
    }
    F::store(reinterpret_cast<S *>(out), r);
    return SR_OK;
}
// StarkL (stark_lazy.hpp): 0 add, 1 sub, 3 mul_tw, 4 table form, 5 a chain of six lazy additions and subtractions feeding
// mul_tw and mul_data, 6 repeated quadrupling with weak reductions, 7 / 8 signed combinations -- results leave through the
// canonicalising store
static int selftest_lazy(int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    using F = sr::StarkL;
    using S = F::storage;
    F::elem x = F::load(reinterpret_cast<const S *>(a)), y = F::load(reinterpret_cast<const S *>(b)), r;
    switch (op) {
        case 0: r = F::add(x, y); break;
        case 1: r = F::sub(x, y); break;
        case 3: r = F::mul_tw(x, y); break;
        case 4: r = F::tw_from_u64(a[0]); break;
        case 5: {
            F::elem s = x, d = x;
            for (int i = 0; i < 6; i++) {
                s = F::add(s, y);
                d = F::sub(d, y);
            }
            r = F::add(F::mul_tw(s, y), F::mul_data(d, s));
            break;
        }
        case 6: {
            r = x;
            for (int i = 0; i < 4; i++) r = F::weak_reduce(F::add(F::add(r, r), F::add(r, r)));  // 256 a, reduced weakly on the way
            break;
        }
        case 7: {  // 3 a - 5 b and, case 8, 7 a - 2 b without any carry in between: canonical() on signed lazy states
            r = F::sub(F::add(F::add(x, x), x), F::add(F::add(F::add(y, y), F::add(y, y)), y));
            break;
        }
        case 8: {
            F::elem x2 = F::add(x, x), x4 = F::add(x2, x2);
            r = F::sub(F::add(F::add(x4, x2), x), F::add(y, y));
            break;
        }
        default: return fail(SR_E_INVALID, "selftest: unknown op");
    }
    F::store(reinterpret_cast<S *>(out), r);
    return SR_OK;
}
}  // extern "C++"
// op 5, field 0: the compile-time shift product of the tuned Goldilocks path, gl::mul_pow2<E>(a[0]) with E = b[0] in [1, 95]
extern "C++" {
template <int... Es>
static uint64_t mul_pow2_any(uint64_t x, int e, std::integer_sequence<int, Es...>) {
    uint64_t r = 0;
    ((e == Es + 1 ? (r = sr::gl::mul_pow2<Es + 1>(x), 0) : 0), ...);
    return r;
}
}
int sr_selftest_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    if (!a || !b || !out) return fail(SR_E_INVALID, "null argument");
    if (field == 0 && op == 5) {
        if (b[0] < 1 || b[0] > 95) return fail(SR_E_INVALID, "selftest: shift out of range");
        out[0] = mul_pow2_any(a[0], (int)b[0], std::make_integer_sequence<int, 95>{});
        return SR_OK;
    }
    alignas(16) uint64_t ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0}, to[4] = {0, 0, 0, 0};
    int words = (field == 2 || field == 4) ? 4 : 1;
    memcpy(ta, a, words * 8);
    memcpy(tb, b, words * 8);
    int rc;
    switch (field) {
        case 0: rc = selftest_op<sr::Goldilocks>(op, ta, tb, to); break;
        case 1: rc = selftest_op<sr::BabyBear>(op, ta, tb, to); break;
        case 2: rc = selftest_op<sr::Stark>(op, ta, tb, to); break;
        case 3: rc = selftest_op<sr::Frog>(op, ta, tb, to); break;
        case 4: rc = selftest_lazy(op, ta, tb, to); break;
        default: return fail(SR_E_INVALID, "selftest: unknown field");
    }
    memcpy(out, to, words * 8);
    return rc;
}

// Representative-invariant counters of the checking build (fields.hpp: repcheck).  The product library counts nothing and says so.
int sr_selftest_rep_counters(uint64_t counters[8], int reset) {
    if (!counters) return fail(SR_E_INVALID, "null argument");
#if defined(SR_GL_CHECK_REPS)
    // g_counters is a __device__ symbol, i.e. one copy per device: visit every visible device (contexts of the sharded tests live on
    // devices other than the current one), sum the violation counts and take the high-water mark's maximum
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    for (int i = 0; i < 8; i++) counters[i] = 0;
    for (int dev = 0; dev < ndev; dev++) {
        DeviceGuard g(dev);
        if (!g.ok) return fail(SR_E_HIP, "hipSetDevice failed");
        unsigned long long h[8];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(sr::repcheck::g_counters), sizeof(h)));
        for (int i = 0; i < 7; i++) counters[i] += h[i];
        if (h[7] > counters[7]) counters[7] = h[7];
        if (reset) {
            memset(h, 0, sizeof(h));
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(sr::repcheck::g_counters), h, sizeof(h)));
        }
    }
    return SR_OK;
#else
    (void)reset;
    for (int i = 0; i < 8; i++) counters[i] = 0;
    return fail(SR_E_UNSUPPORTED, "this build does not check representatives (built without -DSR_GL_CHECK_REPS)");
#endif
}

// ---- profiling ----
int sr_ctx_profile_enable(sr_ctx *c, int on) {
    if (!c) return fail(SR_E_INVALID, "null context");
    std::lock_guard<std::mutex> lk(c->mu);
    c->prof.on = on != 0;
    c->prof.stride = on > 1 ? (unsigned)on : 1u;
    c->prof.tick = 0;
    return SR_OK;
}
int sr_ctx_profile_read(sr_ctx *c, double *ms_total, uint64_t *launches) { return sr_ctx_profile_read_sampled(c, ms_total, launches, nullptr); }
int sr_ctx_profile_read_sampled(sr_ctx *c, double *ms_total, uint64_t *launches, uint64_t *seen) {
    if (!c || !ms_total || !launches) return fail(SR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    for (auto &p : c->prof.pending) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        c->prof.ms[p.tag] += ms;
        c->prof.launches[p.tag]++;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    c->prof.pending.clear();
    for (int t = 0; t < K_NTAGS; t++) {
        ms_total[t] = c->prof.ms[t];
        launches[t] = c->prof.launches[t];
        if (seen) seen[t] = c->prof.seen[t];
        c->prof.ms[t] = 0;
        c->prof.launches[t] = 0;
        c->prof.seen[t] = 0;
    }
    return SR_OK;
}

}  // extern "C"
