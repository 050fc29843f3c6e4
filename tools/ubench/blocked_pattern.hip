// Would a tile-blocked operand scratch pay?  The column pass of D = 2^16 reads 256 legs x 16 columns (128-byte segments, 2 KiB
// apart) and today writes the same shape back.  Here the scratch side of each pass is laid out so that a workgroup's 4096 words
// are one contiguous 32 KiB block:  S(leg 16 rg + sg, column 16 ci + c) = ci * 4096 + (sg * 16 + rg) * 16 + c.
// A rows workgroup (16 legs {16 rg + sg : rg}) then reads 16 pieces of 2 KiB instead of one piece of 32 KiB.
// No arithmetic, no LDS: memory pattern only.  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint64_t u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ u64 ldn(const u64 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stn(u64 *p, u64 v) { __builtin_nontemporal_store(v, p); }

// MODE 0: natural -> natural (today), 1: natural -> blocked (forward column pass), 2: blocked -> natural (inverse column pass)
// MAP 0: blockIdx = poly * 16 + ci (neighbouring column chunks land on different XCDs); 1: the 16 chunks of a ring element go to
// ONE XCD back to back (blockIdx % 8 = XCD); 2: like 1 with pairs of ring elements interleaved
template <int MODE, int MAP = 0>
__global__ __launch_bounds__(256, 4) void cols_pattern(u64 *dst, const u64 *src, const u64 *twist) {
    const int t = threadIdx.x, c = t & 15, rg = t >> 4;
    unsigned ci = blockIdx.x & 15u;
    size_t poly = blockIdx.x >> 4;
    if (MAP == 1) {
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        ci = slot & 15u;
        poly = (size_t)(slot >> 4) * 8 + xcd;
    }
    const u64 *ps = src + (poly << 16);
    u64 *pd = dst + (poly << 16);
    const unsigned i = ci * 16 + c;
    u64 x[16], tw[16];
    if (MODE == 3) {
#pragma unroll
        for (int sg = 0; sg < 16; sg++) x[sg] = ldn(ps + ((16 * rg + sg) << 8) + i);
    } else if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = ldn(ps + ((rg + 16 * j) << 8) + i);
    } else {
#pragma unroll
        for (int sg = 0; sg < 16; sg++) x[sg] = ldn(ps + ci * 4096 + (sg * 16 + rg) * 16 + c);
    }
#pragma unroll
    for (int sg = 0; sg < 16; sg++) tw[sg] = twist[((16 * rg + sg) << 8) + i];
    if (MODE == 0) {
#pragma unroll
        for (int sg = 0; sg < 16; sg++) stn(pd + ((16 * rg + sg) << 8) + i, x[sg] ^ tw[sg]);
    } else if (MODE == 1) {
#pragma unroll
        for (int sg = 0; sg < 16; sg++) stn(pd + ci * 4096 + (sg * 16 + rg) * 16 + c, x[sg] ^ tw[sg]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) stn(pd + ((rg + 16 * j) << 8) + i, x[j] ^ tw[j]);
    }
}
// rows: lane (rho, i0) of tile sg owns leg 16 rho + sg; BLK 0: natural (tile = 16 consecutive legs), 1: blocked
template <int BLK>
__global__ __launch_bounds__(256, 4) void rows_pattern(u64 *out, const u64 *a, const u64 *b) {
    const int t = threadIdx.x, i0 = t & 15, rho = t >> 4;
    const unsigned sg = blockIdx.x & 15u;
    const size_t poly = blockIdx.x >> 4;
    const size_t base = (poly << 16) + (BLK ? (sg * 16 + rho) * 16 + i0 : sg * 4096 + rho * 256 + i0);
    constexpr int step = BLK ? 4096 : 16;
    u64 x[16], y[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = ldn(a + base + m * step);
#pragma unroll
    for (int m = 0; m < 16; m++) y[m] = ldn(b + base + m * step);
#pragma unroll
    for (int m = 0; m < 16; m++) stn(out + base + m * step, x[m] ^ y[m]);
}
template <typename L>
int timeit(const char *what, double bytes, L launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    printf("%-52s %.3f ms  %.2f TB/s\n", what, best, bytes / best / 1e9);
    return 0;
}
int main(int argc, char **argv) {
    const size_t npoly = argc > 1 ? strtoull(argv[1], 0, 10) : 2048;
    u64 *a, *b, *o, *tw;
    CK(hipMalloc(&a, npoly << 19)); CK(hipMalloc(&b, npoly << 19)); CK(hipMalloc(&o, npoly << 19)); CK(hipMalloc(&tw, 65536 * 8));
    CK(hipMemset(a, 1, npoly << 19)); CK(hipMemset(b, 2, npoly << 19)); CK(hipMemset(tw, 1, 65536 * 8));
    const unsigned blocks = (unsigned)(npoly * 16);
    const double by = (double)(npoly << 19);
    printf("npoly %zu\n", npoly);
    timeit("cols natural -> natural (today)", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<0>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("cols natural -> natural, inverse's mappings", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<3>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("cols natural -> natural, 16 chunks on one XCD", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<0, 1>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("cols inverse's mappings, 16 chunks on one XCD", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<3, 1>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("cols natural -> blocked (forward)", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<1>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("cols blocked -> natural (inverse)", 2 * by, [&] { hipLaunchKernelGGL((cols_pattern<2>), dim3(blocks), dim3(256), 0, 0, o, a, tw); });
    timeit("rows natural (2 reads + 1 write)", 3 * by, [&] { hipLaunchKernelGGL(rows_pattern<0>, dim3(blocks), dim3(256), 0, 0, o, a, b); });
    timeit("rows blocked (2 reads + 1 write)", 3 * by, [&] { hipLaunchKernelGGL(rows_pattern<1>, dim3(blocks), dim3(256), 0, 0, o, a, b); });
    return 0;
}
