// Streaming ceilings of this box for the element-wise kernels: a += b style (2 reads + 1 write) and copy (1 + 1), 16 bytes per lane,
// default cache policy against non-temporal accesses.  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <bool NT, bool ADD>
__global__ __launch_bounds__(256) void k(u64x2 *a, const u64x2 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64x2 y = NT ? __builtin_nontemporal_load(b + i) : b[i];
        if (ADD) {
            const u64x2 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
            y += x;
        }
        if (NT) __builtin_nontemporal_store(y, a + i); else a[i] = y;
    }
}
template <bool NT, bool ADD>
int run(u64x2 *a, u64x2 *b, size_t n, unsigned blocks, const char *name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NT, ADD>), dim3(blocks), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    printf("%-34s blocks %8u: %.3f ms  %.2f TB/s\n", name, blocks, best, (ADD ? 3.0 : 2.0) * n * 16 / best / 1e9);
    return 0;
}
int main() {
    const size_t n = (size_t)1 << 29;  // 8 GiB per buffer
    u64x2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
    for (unsigned blocks : {1u << 21, 1u << 20, 1u << 16, 16384u, 8192u, 4096u, 2048u}) {
        run<false, true>(a, b, n, blocks, "a += b, default policy");
        run<true, true>(a, b, n, blocks, "a += b, non-temporal");
        run<false, false>(a, b, n, blocks, "a = b, default policy");
        run<true, false>(a, b, n, blocks, "a = b, non-temporal");
    }
    return 0;
}
