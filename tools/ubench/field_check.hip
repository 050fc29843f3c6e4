// Device-vs-host check of fields.hpp (same source compiled twice): catches device-only codegen problems.
#include <cstdio>
#include <vector>
#include "../../stark_rings_amd/csrc/fields.hpp"
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
using G = sr::Goldilocks;
// round 3: the lazy decimation-in-time legs (any u64 a, canonical t) and the DFT_16 networks built from them, device against host
__global__ void klazy(const uint64_t *a, const uint64_t *t, uint64_t *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s, d;
    G::addsub_lazy(a[i], t[i], s, d);
    o[2 * i] = s;
    o[2 * i + 1] = d;
}
// per lane: 16 canonical inputs -> the DIF network (canonical), the DIT network with lazy butterflies, the lazy inverse network
__global__ void knet(const uint64_t *x, uint64_t *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t u[16], v[16], w[16];
    for (int j = 0; j < 16; j++) u[j] = v[j] = w[j] = x[16 * i + j];
    sr::gl::dft16_fwd(u);
    sr::gl::dft16_fwd_dit<sr::gl::kPhased, true>(v);
    sr::gl::dft16_inv<sr::gl::kPhased, true>(w);
    for (int j = 0; j < 16; j++) {
        o[48 * i + j] = u[j];
        o[48 * i + 16 + j] = v[j];
        o[48 * i + 32 + j] = w[j];
    }
}
template <class F>
SR_HD void fq3_mul(uint64_t *x, const uint64_t *y, uint64_t nr) {
    uint64_t t[5];
    for (int i = 0; i < 5; i++) t[i] = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i + j] = F::add(t[i + j], F::mul_boundary_pre(x[i], y[j]));
    for (int i = 0; i < 2; i++) t[i] = F::add(t[i], F::mul_tw(t[i + 3], nr));
    for (int m = 0; m < 3; m++) x[m] = F::boundary_post(t[m]);
}
__global__ void k3(const uint64_t *a, const uint64_t *b, uint64_t *o, int n, uint64_t nr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]}, y[3] = {b[3 * i], b[3 * i + 1], b[3 * i + 2]};
    fq3_mul<G>(x, y, nr);
    for (int m = 0; m < 3; m++) o[3 * i + m] = x[m];
}
__global__ void k(const uint64_t *a, const uint64_t *b, uint64_t *o, int n, uint64_t cst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i * 6 + 0] = G::add(a[i], b[i]);
    o[i * 6 + 1] = G::sub(a[i], b[i]);
    o[i * 6 + 2] = G::mul(a[i], b[i]);
    o[i * 6 + 3] = G::mul_boundary(a[i], b[i]);
    o[i * 6 + 4] = G::mul(a[i], cst);            // constant in SGPRs
    o[i * 6 + 5] = G::add(G::add(0ull, G::sub(a[i], b[i])), G::sub(G::sub(b[i], a[i]), G::add(a[i], 0ull)));  // carry-op chains
}
// round 4: everything a lazy representative can reach, on NON-canonical operands (any 64-bit word): the general product, the folds
// behind it and every compile-time shift product.  Device against the host build AND against plain 128-bit integer arithmetic.
__global__ void kraw(const uint64_t *a, const uint64_t *b, uint64_t *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long bo = __ballot((b[i] >> 63) != 0);   // a lane mask, as the borrow of a carry chain is
    o[4 * i + 0] = G::mul(a[i], b[i]);
    o[4 * i + 1] = G::mad_eps_fix(a[i], (uint32_t)b[i]);
    o[4 * i + 2] = G::fix_fold(a[i], bo, (uint32_t)b[i]);
    o[4 * i + 3] = G::reduce128(a[i], b[i]);
}
template <int... Es>
__global__ void kpow2(const uint64_t *a, uint64_t *o, int n, std::integer_sequence<int, Es...>) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ((o[95 * (size_t)i + Es] = sr::gl::mul_pow2<Es + 1>(a[i])), ...);
}
template <int... Es>
static void host_pow2(uint64_t x, uint64_t *o, std::integer_sequence<int, Es...>) {
    ((o[Es] = sr::gl::mul_pow2<Es + 1>(x)), ...);
}
static uint64_t mulmod(uint64_t x, uint64_t y) { return (uint64_t)(((unsigned __int128)(x % G::P) * (y % G::P)) % G::P); }
static uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main() {
    const int n = 1 << 16;
    std::vector<uint64_t> a(n), b(n), o(n * 6);
    const uint64_t sp[] = {0, 1, 2, G::P - 1, G::P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 1ull << 40, 1ull << 63, 0xFFFFFFFEFFFFFFFFull};
    for (int i = 0; i < n; i++) {
        a[i] = i < 121 ? sp[i / 11] : mix(i) % G::P;
        b[i] = i < 121 ? sp[i % 11] : mix(i + 0x1234567) % G::P;
    }
    uint64_t *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, n * 48);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    const uint64_t cst = 1ull << 40;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n, cst);
    hipMemcpy(o.data(), dout, n * 48, hipMemcpyDeviceToHost);
    int bad = 0;
    const char *names[] = {"add", "sub", "mul", "mul_boundary", "mul(sgpr 2^40)", "carry chains"};
    for (int i = 0; i < n; i++) {
        uint64_t w[6] = {G::add(a[i], b[i]), G::sub(a[i], b[i]), G::mul(a[i], b[i]), G::mul_boundary(a[i], b[i]), G::mul(a[i], cst), G::add(G::add(0ull, G::sub(a[i], b[i])), G::sub(G::sub(b[i], a[i]), G::add(a[i], 0ull)))};
        for (int j = 0; j < 6; j++)
            if (o[i * 6 + j] != w[j] && bad++ < 12) printf("MISMATCH %s a=%016llx b=%016llx dev=%016llx host=%016llx\n", names[j], (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)o[i * 6 + j], (unsigned long long)w[j]);
    }
    {
        const int m3 = n / 3;
        hipLaunchKernelGGL(k3, dim3((m3 + 255) / 256), dim3(256), 0, 0, da, db, dout, m3, cst);
        hipMemcpy(o.data(), dout, (size_t)m3 * 24, hipMemcpyDeviceToHost);
        for (int i = 0; i < m3; i++) {
            uint64_t x[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]};
            fq3_mul<G>(x, &b[3 * i], cst);
            for (int m = 0; m < 3; m++)
                if (o[3 * i + m] != x[m] && bad++ < 12) printf("MISMATCH fq3[%d] i=%d dev=%016llx host=%016llx\n", m, i, (unsigned long long)o[3 * i + m], (unsigned long long)x[m]);
        }
    }
    {   // lazy legs: a runs over every kind of 64-bit value (canonical or not), t over canonical values
        const uint64_t av[] = {0, 1, G::P - 1, G::P, G::P + 1, ~0ull, ~0ull - 1, 0xFFFFFFFFull, 0x100000000ull, 1ull << 63, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull};
        const uint64_t tv[] = {0, 1, 2, G::P - 1, G::P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 1ull << 63, 0xFFFFFFFEFFFFFFFFull, 0xFFFFFFFE00000002ull};
        std::vector<uint64_t> la(n), lt(n), lo(2 * (size_t)n);
        for (int i = 0; i < n; i++) {
            la[i] = i < 132 ? av[i / 11] : mix(i + 77);            // any u64
            lt[i] = i < 132 ? tv[i % 11] : mix(i + 0x777) % G::P;
        }
        hipMemcpy(da, la.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, lt.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(klazy, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        hipMemcpy(lo.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            uint64_t s, d;
            G::addsub_lazy(la[i], lt[i], s, d);
            const unsigned __int128 P = G::P;
            const bool ok_host = (unsigned __int128)s % P == ((unsigned __int128)la[i] + lt[i]) % P &&
                                 (unsigned __int128)d % P == ((unsigned __int128)la[i] % P + P - lt[i]) % P;
            if ((lo[2 * i] != s || lo[2 * i + 1] != d || !ok_host) && bad++ < 12)
                printf("MISMATCH lazy a=%016llx t=%016llx dev=(%016llx,%016llx) host=(%016llx,%016llx) host value %s\n", (unsigned long long)la[i], (unsigned long long)lt[i],
                       (unsigned long long)lo[2 * i], (unsigned long long)lo[2 * i + 1], (unsigned long long)s, (unsigned long long)d, ok_host ? "ok" : "WRONG");
        }
    }
    {   // networks: the lazy DIT forward network equals the canonical DIF one modulo p, slot 0 canonical; lazy inverse against canonical inverse
        const int m = n / 16;
        std::vector<uint64_t> xs(n), no(48 * (size_t)m);
        const uint64_t ev[] = {0, 1, G::P - 1, G::P - 2, 0xFFFFFFFFull, 0x100000000ull, 1ull << 63, 0xFFFFFFFF00000000ull};
        for (int i = 0; i < n; i++) xs[i] = i < 4096 ? ev[mix(i) & 7] : mix(i + 31) % G::P;
        uint64_t *dn;
        hipMalloc(&dn, (size_t)m * 48 * 8);
        hipMemcpy(da, xs.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(knet, dim3((m + 255) / 256), dim3(256), 0, 0, da, dn, m);
        hipMemcpy(no.data(), dn, (size_t)m * 48 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < m; i++) {
            uint64_t u[16], w[16];
            for (int j = 0; j < 16; j++) u[j] = w[j] = xs[16 * i + j];
            sr::gl::dft16_fwd(u);
            sr::gl::dft16_inv(w);
            for (int j = 0; j < 16; j++) {
                const bool f0 = no[48 * i + j] == u[j], f1 = no[48 * i + 16 + j] % G::P == u[j], f2 = no[48 * i + 32 + j] % G::P == w[j];
                const bool c0 = j != 0 || (no[48 * i + 16] < G::P && no[48 * i + 32] < G::P);
                if (!(f0 && f1 && f2 && c0) && bad++ < 12) printf("MISMATCH network i=%d slot %d dif %d dit-lazy %d inv-lazy %d slot0-canonical %d\n", i, j, f0, f1, f2, c0);
            }
        }
    }
    {   // non-canonical operands: p, p + 1, 2^64 - 1, 2^64 - 2^32 ... and uniform 64-bit words
        const uint64_t rv[] = {0, 1, G::P - 1, G::P, G::P + 1, G::P + 0xFFFFFFFEull, ~0ull, ~0ull - 1, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull,
                               0xFFFFFFFFull, 0x100000000ull, 1ull << 63, (1ull << 63) - 1, 0x8000000080000000ull, 0xFFFFFFFF80000000ull};
        const int nr = 16, m = 1 << 14;
        std::vector<uint64_t> ra(m), rb(m), ro(4 * (size_t)m), po(95 * (size_t)m);
        for (int i = 0; i < m; i++) {
            ra[i] = i < nr * nr ? rv[i / nr] : mix(i + 0xABC);   // any u64
            rb[i] = i < nr * nr ? rv[i % nr] : mix(i + 0xDEF);
        }
        uint64_t *dp;
        hipMalloc(&dp, 95 * (size_t)m * 8);
        hipMemcpy(da, ra.data(), m * 8, hipMemcpyHostToDevice); hipMemcpy(db, rb.data(), m * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kraw, dim3(m / 256), dim3(256), 0, 0, da, db, dout, m);
        hipMemcpy(ro.data(), dout, (size_t)m * 32, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(kpow2, dim3(m / 256), dim3(256), 0, 0, da, dp, m, std::make_integer_sequence<int, 95>{});
        hipMemcpy(po.data(), dp, 95 * (size_t)m * 8, hipMemcpyDeviceToHost);
        const char *rn[] = {"mul(any, any)", "mad_eps_fix", "fix_fold", "reduce128"};
        uint64_t two[96];
        two[0] = 1;
        for (int e = 1; e < 96; e++) two[e] = mulmod(two[e - 1], 2);
        for (int i = 0; i < m; i++) {
            const uint64_t x = ra[i], y = rb[i], hl = (uint32_t)y, bo = y >> 63;
            const uint64_t rfix = bo ? x + G::P : x;                                     // wraps like the masked 64-bit add
            const unsigned __int128 P = G::P;
            const uint64_t host[4] = {G::mul(x, y), G::mad_eps_fix(x, (uint32_t)hl), G::fix_fold(x, bo, (uint32_t)hl), G::reduce128(x, y)};
            const uint64_t want[4] = {mulmod(x, y), (uint64_t)(((unsigned __int128)x + (unsigned __int128)hl * G::EPS) % P),
                                      (uint64_t)(((unsigned __int128)rfix + (unsigned __int128)hl * G::EPS) % P),
                                      (uint64_t)((((unsigned __int128)y << 64) | x) % P)};
            for (int j = 0; j < 4; j++)
                if ((ro[4 * i + j] != host[j] || host[j] != want[j]) && bad++ < 12)
                    printf("MISMATCH %s x=%016llx y=%016llx dev=%016llx host=%016llx integers=%016llx\n", rn[j], (unsigned long long)x, (unsigned long long)y,
                           (unsigned long long)ro[4 * i + j], (unsigned long long)host[j], (unsigned long long)want[j]);
            uint64_t hp[95];
            host_pow2(x, hp, std::make_integer_sequence<int, 95>{});
            for (int e = 1; e <= 95; e++) {
                const uint64_t w = mulmod(x, two[e]);
                if ((po[95 * (size_t)i + e - 1] != hp[e - 1] || hp[e - 1] != w) && bad++ < 12)
                    printf("MISMATCH mul_pow2<%d> x=%016llx dev=%016llx host=%016llx integers=%016llx\n", e, (unsigned long long)x,
                           (unsigned long long)po[95 * (size_t)i + e - 1], (unsigned long long)hp[e - 1], (unsigned long long)w);
            }
        }
        printf("non-canonical operands: %d x (mul, mad_eps_fix, fix_fold, reduce128, 95 shift products) checked\n", m);
    }
    printf("field_check: %d mismatches\n", bad);
    return bad != 0;
}
