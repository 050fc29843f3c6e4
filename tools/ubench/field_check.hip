// Device-vs-host check of fields.hpp (same source compiled twice): catches device-only codegen problems.
#include <cstdio>
#include <vector>
#include "../../stark_rings_amd/csrc/fields.hpp"
using G = sr::Goldilocks;
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
    printf("field_check: %d mismatches\n", bad);
    return bad != 0;
}
