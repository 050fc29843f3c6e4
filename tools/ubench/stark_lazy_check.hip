// Device build of StarkL::mul_tw (one asm statement per column, stark_mul_cols.inc) against the host build (plain C++) of the same
// header: the two must agree limb for limb on random lazy operands (limbs up to +-2^31, twiddles normalised) and on the edges.
#include <cstdio>
#include <vector>
#include "../../stark_rings_amd/csrc/fields.hpp"
#include "../../stark_rings_amd/csrc/stark_lazy.hpp"
using L = sr::StarkL;
using E = sr::S9;
__global__ void k(const E *a, const E *b, E *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = L::mul_tw(a[i], b[i]);
}
__global__ void kd(const E *a, const E *b, E *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = L::mul_data(a[i], b[i]);
}
static uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main() {
    const int n = 1 << 16;
    std::vector<E> a(n), b(n), o(n);
    for (int i = 0; i < n; i++) {
        for (int l = 0; l < 9; l++) {
            const uint64_t r = mix(i * 32 + l), q = mix(i * 32 + 16 + l + 0x7777);
            // a: a lazy element (signed limbs of up to 31 bits in a third of the cases, 28-bit otherwise); b: a table twiddle
            a[i].l[l] = (i % 3 == 0) ? (int32_t)(r & 0x7FFFFFF0) * ((r >> 40) & 1 ? -1 : 1) : (int32_t)(r & 0xFFFFFFF);
            b[i].l[l] = (int32_t)(q & 0xFFFFFFF);
        }
        if (i % 3 == 0) { a[i].l[8] = (int32_t)(mix(i) & 0x3FFFFFF) - (1 << 25); }  // |value| < 16 p
        else a[i].l[8] &= 0x7FFFFFF;
        b[i].l[8] &= 0x7FFFFFF;
        if (i == 1) for (int l = 0; l < 9; l++) a[i].l[l] = 0;
        if (i == 2) for (int l = 0; l < 9; l++) { a[i].l[l] = 0xFFFFFFF; b[i].l[l] = 0xFFFFFFF; }
    }
    E *da, *db, *dout;
    hipMalloc(&da, n * sizeof(E)); hipMalloc(&db, n * sizeof(E)); hipMalloc(&dout, n * sizeof(E));
    hipMemcpy(da, a.data(), n * sizeof(E), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(E), hipMemcpyHostToDevice);
    int bad = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0) hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        else hipLaunchKernelGGL(kd, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        hipMemcpy(o.data(), dout, n * sizeof(E), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            const E w = pass == 0 ? L::mul_tw(a[i], b[i]) : L::mul_data(a[i], b[i]);
            for (int l = 0; l < 9; l++)
                if (w.l[l] != o[i].l[l]) { if (bad++ < 5) printf("mismatch pass %d i=%d limb %d dev %08x host %08x\n", pass, i, l, o[i].l[l], w.l[l]); break; }
        }
    }
    printf("StarkL mul_tw / mul_data device vs host: %d mismatches of %d\n", bad, 2 * n);
    return bad != 0;
}
