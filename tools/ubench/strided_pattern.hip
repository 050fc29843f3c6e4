// Memory-pattern ceiling of the column passes: 256 legs (stride N2 coefficients) x COLS consecutive columns per workgroup,
// read with the pass-A mapping and written with the pass-B mapping, no arithmetic, no LDS.  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint64_t u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int COLS, int VEC>
__global__ __launch_bounds__(16 * COLS / VEC) void pattern_kernel(u64 *data, int k, const u64 *twist) {
    // VEC = coefficients per lane per access (1: dwordx2, 2: dwordx4)
    const int t = threadIdx.x;
    const int ls = k - 8;
    constexpr int LANES_PER_LEG = COLS / VEC;
    const unsigned chunks = (1u << ls) / COLS;
    const unsigned ci = blockIdx.x % chunks;
    const size_t poly = blockIdx.x / chunks;
    const int col = t % LANES_PER_LEG, rg = t / LANES_PER_LEG;
    const unsigned i = ci * COLS + col * VEC;
    u64 *pb = data + (poly << k);
    u64 x[16][VEC], tw[16][VEC];
#pragma unroll
    for (int j = 0; j < 16; j++)
#pragma unroll
        for (int v = 0; v < VEC; v++) x[j][v] = pb[((size_t)(rg + 16 * j) << ls) + i + v];
#pragma unroll
    for (int j = 0; j < 16; j++)
#pragma unroll
        for (int v = 0; v < VEC; v++) tw[j][v] = twist[((size_t)(16 * rg + j) << ls) + i + v];
#pragma unroll
    for (int j = 0; j < 16; j++)
#pragma unroll
        for (int v = 0; v < VEC; v++) pb[((size_t)(16 * rg + j) << ls) + i + v] = x[j][v] ^ tw[j][v];
}
// a wave owns 4 columns x 256 legs: lane = (col 0..3, leg group 0..15), 32-byte segments, no workgroup-wide exchange needed
__global__ __launch_bounds__(256) void pattern_wave4(u64 *data, int k, const u64 *twist) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int ls = k - 8;
    const unsigned chunks = (1u << ls) / 16;
    const unsigned ci = blockIdx.x % chunks;
    const size_t poly = blockIdx.x / chunks;
    const int col = lane & 3, rg = lane >> 2;
    const unsigned i = ci * 16 + wave * 4 + col;
    u64 *pb = data + (poly << k);
    u64 x[16], tw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = pb[((size_t)(rg + 16 * j) << ls) + i];
#pragma unroll
    for (int j = 0; j < 16; j++) tw[j] = twist[((size_t)(16 * rg + j) << ls) + i];
#pragma unroll
    for (int j = 0; j < 16; j++) pb[((size_t)(16 * rg + j) << ls) + i] = x[j] ^ tw[j];
}
__global__ void copy16(const ulonglong2 *a, ulonglong2 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <int COLS, int VEC>
int run(u64 *a, u64 *tw, size_t npoly, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)(npoly * (256 / COLS));
    float best = 1e9;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((pattern_kernel<COLS, VEC>), dim3(blocks), dim3(16 * COLS / VEC), 0, 0, a, 16, tw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    printf("cols %3d vec %d: %.3f ms  %.2f TB/s (r+w of data)\n", COLS, VEC, best, 2.0 * npoly * 65536 * 8 / best / 1e9);
    return 0;
}
int main(int argc, char **argv) {
    const size_t npoly = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    u64 *a, *b, *tw; CK(hipMalloc(&a, npoly << 19)); CK(hipMalloc(&b, npoly << 19)); CK(hipMalloc(&tw, 65536 * 8));
    CK(hipMemset(a, 1, npoly << 19)); CK(hipMemset(tw, 1, 65536 * 8));
    run<16, 1>(a, tw, npoly, 4); run<32, 1>(a, tw, npoly, 4); run<64, 1>(a, tw, npoly, 4);
    run<32, 2>(a, tw, npoly, 4); run<64, 2>(a, tw, npoly, 4); run<16, 2>(a, tw, npoly, 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern_wave4, dim3((unsigned)(npoly * 16)), dim3(256), 0, 0, a, 16, tw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) printf("wave-private 4 columns x 256 legs (32-byte segments): %.3f ms %.2f TB/s\n", ms, 2.0 * (npoly << 19) / ms / 1e9);
    }
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy16, dim3(256 * 64), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b, (npoly << 19) / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy16 a->b: %.3f ms %.2f TB/s\n", ms, 2.0 * (npoly << 19) / ms / 1e9);
    }
    return 0;
}
