// Timing-only harness for the Stark tile kernel of BASELINE config 5 (st::tile_kernel<9, MODE_MUL, false>: the last nine stages of
// D = 4096 ring products).  Junk data and junk tables: the instruction stream does not depend on the values (the 2^-25 slow path of
// StarkL::canonical aside).  Build variants with -D flags and compare; correctness is checked through the library, never here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSR_ST_XCHG=n] -o build_tmp/st_bench tools/ubench/st_bench.hip && build_tmp/st_bench [batch] [reps]
// The variants of round 5 (profiles/r05/st_bench.txt, st_bench_wave_local.txt) were one-line edits of st::Tile::exchange in a scratch copy
// of csrc/ntt_stark.hpp, never committed to the product: xchg=1 = `return;` as the first statement of exchange() (no LDS traffic at
// all: wrong results, timing only); "st_bench2" = the two __syncthreads() of the nine wave-local steps (neighbouring layouts whose
// exchanged lane bits are both below 6) replaced by wavefront-scope fences, a leading __syncthreads() on the three steps that cross waves.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../stark_rings_amd/csrc/ntt_stark.hpp"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__global__ void junk_kernel(uint32_t *p, size_t n, uint32_t seed, uint32_t mask) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        p[i] = x & mask;
    }
}

int main(int argc, char **argv) {
    using namespace sr;
    const size_t batch = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int k = 12;
    const size_t d = (size_t)1 << k, n = batch * d;
    st::S *a = nullptr, *b = nullptr, *out = nullptr;
    S9 *tw = nullptr;
    CK(hipMalloc(&a, n * sizeof(st::S)));
    CK(hipMalloc(&b, n * sizeof(st::S)));
    CK(hipMalloc(&out, n * sizeof(st::S)));
    CK(hipMalloc(&tw, 2 * d * sizeof(S9)));
    // canonical-looking images: top limb below 2^27 so that loads see "values below p"
    junk_kernel<<<4096, 256>>>((uint32_t *)a, n * 8, 1u, 0x07FFFFFFu);
    junk_kernel<<<4096, 256>>>((uint32_t *)b, n * 8, 2u, 0x07FFFFFFu);
    junk_kernel<<<256, 256>>>((uint32_t *)tw, 2 * d * 9, 3u, 0x07FFFFFFu);  // table limbs in [0, 2^27)
    CK(hipDeviceSynchronize());
    st::P p;
    p.k = k;
    p.s_rows = 0;
    p.log_tile = 9;
    p.tw = tw;
    p.itw = tw + d;
    CK(hipMemcpy(&p.scale0, tw, sizeof(S9), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&p.scale1, tw + 1, sizeof(S9), hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tiles = batch << (k - 9);
    for (int w = 0; w < 3; w++)
        if (st::launch_tile<9, MODE_MUL, false>(a, b, out, tiles, p, nullptr)) return 2;
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, nullptr));
        if (st::launch_tile<9, MODE_MUL, false>(a, b, out, tiles, p, nullptr)) return 2;
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms;
        if (ms < best) best = ms;
    }
    // back to back (what a step sees): reps launches between one pair of events
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; r++)
        if (st::launch_tile<9, MODE_MUL, false>(a, b, out, tiles, p, nullptr)) return 2;
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float b2b = 0;
    CK(hipEventElapsedTime(&b2b, e0, e1));
    uint32_t probe = 0;
    CK(hipMemcpy(&probe, out, 4, hipMemcpyDeviceToHost));
#ifndef SR_ST_XCHG
#define SR_ST_XCHG 0
#endif
    printf("tile_kernel<9, MUL> xchg=%d batch %zu (%zu tiles): mean %.4f ms, best %.4f ms, back to back %.4f ms per launch (probe %08x)\n", SR_ST_XCHG,
           batch, tiles, sum / reps, best, b2b / reps, probe);
    return 0;
}
