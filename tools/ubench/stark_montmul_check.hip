// Device build of Stark::mont_mul (FIPS with v_mad_u64_u32 carry-out, inline asm MACs) against the host build
// (textbook CIOS) of the same header, on random and edge operands.
#include <cstdio>
#include <vector>
#include "../../stark_rings_amd/csrc/fields.hpp"
using S = sr::Stark;
using E = sr::U256;
__global__ void k(const E *a, const E *b, E *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = S::mont_mul(a[i], b[i]);
}
static uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main() {
    const int n = 1 << 16;
    std::vector<E> a(n), b(n), o(n);
    E pm1 = S::modulus();
    pm1.l[0] -= 1;
    for (int i = 0; i < n; i++) {
        for (int l = 0; l < 8; l++) { a[i].l[l] = (uint32_t)mix(i * 16 + l); b[i].l[l] = (uint32_t)mix(i * 16 + 8 + l + 0x9999); }
        a[i].l[7] &= 0x07FFFFFF; b[i].l[7] &= 0x07FFFFFF;
        if (i == 0) { a[i] = pm1; b[i] = pm1; }
        if (i == 1) a[i] = S::zero();
        if (i == 2) { a[i] = pm1; b[i] = S::zero(); b[i].l[0] = 1; }
        if (i == 3) { for (int l = 0; l < 8; l++) { a[i].l[l] = 0xFFFFFFFF; b[i].l[l] = 0xFFFFFFFF; } a[i].l[7] = 0x07FFFFFF; b[i].l[7] = 0x07FFFFFF; }
    }
    E *da, *db, *dout;
    hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dout, n * 32);
    hipMemcpy(da, a.data(), n * 32, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(o.data(), dout, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        E w = S::mont_mul(a[i], b[i]);
        for (int l = 0; l < 8; l++)
            if (w.l[l] != o[i].l[l]) { if (bad++ < 5) printf("mismatch i=%d limb %d dev %08x host %08x\n", i, l, o[i].l[l], w.l[l]); break; }
    }
    printf("stark mont_mul device vs host: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
