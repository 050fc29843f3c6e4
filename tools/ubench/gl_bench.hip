// Timing-only harness for the Goldilocks D = 2^16 kernels (junk data and tables: instruction streams and memory traffic do not
// depend on the values).  Build variants with -D flags and compare; correctness is checked through the library, not here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSR_...] -o gl_bench tools/ubench/gl_bench.hip && ./gl_bench [npoly] [reps]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <array>
#include <chrono>
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
#ifdef PERSIST
#include "cols256p_experiment.hpp"
#endif
#if defined(MIXED) || defined(PIPE)
#include "mixed256_experiment.hpp"
#endif
#ifdef KEEPPF
// round-5 experiment: the persistent column pass with the next element's loads in flight during pass B (cols_keep_prefetch.hpp);
// KEEPPF = ring-element groups per XCD and column chunk (4 = 512 workgroups at D = 2^16: two per CU)
#include "cols_keep_prefetch.hpp"
static inline unsigned keeppf_groups(size_t np) { size_t g = np / 8; if (g < 1) g = 1; if (g > KEEPPF) g = KEEPPF; return (unsigned)g; }
#define KEEP 8
#define KEEP_LAUNCH(DIRV, st_, buf, srcbuf, np, wcp, twp)                                                                                         \
    hipLaunchKernelGGL((sr::gl::cols256_keep_pf_kernel<DIRV>), dim3(8u * (1u << (k - 12)) * keeppf_groups(np)), dim3(256), 0, st_, buf, srcbuf, (u64 *)nullptr, \
                       (const u64 *)nullptr, k, wcp, twp, (unsigned)(np), keeppf_groups(np))
#elif defined(KEEP)
// the lane plans' column pass (gl::cols256_keep_kernel, in the library since round 4) against the plain one
// KEEP = ring-element groups per XCD and column chunk (8 = 1024 workgroups at D = 2^16: one round of resident workgroups)
static inline unsigned keep_groups(size_t np) { size_t g = np / 8; if (g < 1) g = 1; if (g > KEEP) g = KEEP; return (unsigned)g; }
#ifndef KEEP_DIRS
#define KEEP_DIRS 3
#endif
#define KEEP_LAUNCH(DIRV, st_, buf, srcbuf, np, wcp, twp)                                                                                         \
    do {                                                                                                                                          \
        if ((KEEP_DIRS >> DIRV) & 1)                                                                                                              \
            hipLaunchKernelGGL((cols256_keep_kernel<DIRV>), dim3(8u * (1u << (k - 12)) * keep_groups(np)), dim3(256), 0, st_, buf, srcbuf, (u64 *)nullptr, (const u64 *)nullptr, k, wcp, \
                               twp, (unsigned)(np), keep_groups(np));                                                                             \
        else                                                                                                                                      \
            hipLaunchKernelGGL((cols256_kernel<DIRV, 4>), dim3((unsigned)((np) << (k - 12))), dim3(256), 0, st_, buf, srcbuf, k, wcp, twp,         \
                               sr::xcd_grouped_tiles((np) << (k - 12), sr::gl::kColsXcdGroup));                                                   \
    } while (0)
#endif
#define ROWS_KERNEL (rows256_kernel<2>)
#ifdef COLS_W3
// the plain column pass at three workgroups per CU (168 VGPRs): separates "fewer, fatter waves" from "factors kept in registers"
using namespace sr::gl;
template <int DIR>
__global__ __launch_bounds__(256, 3) void cols256_w3_kernel(u64 *data, const u64 *src, int k, const u64 *__restrict__ wc,
                                                            const u64 *__restrict__ twist, unsigned grouped) {
    __shared__ u64 lds[ColsTile<4>::kElems];
    cols256_tile<DIR, 4>(sr::xcd_tile(blockIdx.x, kColsXcdGroup, grouped), data, src, k, wc, twist, lds);
}
#define KEEP_LAUNCH(DIRV, st_, buf, srcbuf, np, wcp, twp) hipLaunchKernelGGL((cols256_w3_kernel<DIRV>), dim3((unsigned)((np) << (k - 12))), dim3(256), 0, st_, buf, srcbuf, k, wcp, twp, sr::xcd_grouped_tiles((np) << (k - 12), sr::gl::kColsXcdGroup))
#define KEEP 1
#endif
using namespace sr::gl;
#ifdef PAIR
// forward column passes of a and b as ONE launch (one tail and one launch gap fewer per chunk)
__global__ __launch_bounds__(256, 4) void cols256_pair_kernel(u64 *da, const u64 *sa_, u64 *db, const u64 *sb_, int k, const u64 *__restrict__ wc,
                                                              const u64 *__restrict__ twist, unsigned tiles, unsigned grouped) {
    __shared__ u64 lds[ColsTile<4>::kElems];
    const bool second = blockIdx.x >= tiles;
    const unsigned bid = second ? blockIdx.x - tiles : blockIdx.x;
    cols256_tile<0, 4>(sr::xcd_tile(bid, sr::gl::kColsXcdGroup, grouped), second ? db : da, second ? sb_ : sa_, k, wc, twist, lds);
}
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void junk_kernel(u64 *p, size_t n, u64 seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x % sr::Goldilocks::P;
    }
}
int main(int argc, char **argv) {
    const size_t npoly = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int k = 16;
    const size_t n = npoly << k;
    u64 *a, *b, *tab;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    const size_t tabn = (3u << k) + 3 * 4096 + 4 * 256;
    CK(hipMalloc(&tab, tabn * 8));
    hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, a, n, 1);
    hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, b, n, 2);
    hipLaunchKernelGGL(junk_kernel, dim3(1024), dim3(256), 0, 0, tab, tabn, 3);
    CK(hipDeviceSynchronize());
    u64 *p = tab;
    Tables T{};
    T.twist_f = p; p += (1u << k); T.twist_i_plain = p; p += (1u << k); T.twist_i_mul = p; p += (1u << k);
    T.w1f = p; p += 4096; T.w1i = p; p += 4096; T.w1i_mul = p; p += 4096;
    T.w2f = p; p += 256; T.w2i = p; p += 256; T.wcf = p; p += 256; T.wci = p;
    hipEvent_t ev[5];
    for (auto &e : ev) CK(hipEventCreate(&e));
    #ifndef LCV
#define LCV 4
#endif
    const unsigned blocks = (unsigned)(npoly << (k - 12));
    const unsigned cblocks = (unsigned)(npoly << (k - 8 - LCV));
#ifdef PERSIST
    const unsigned ntiles = cblocks;
    const unsigned pgrid = PERSIST * (LCV == 6 ? 1 : LCV == 5 ? 2 : 4);  // PERSIST = number of CUs to fill
#define COLS(DIRV, buf, srcbuf, np, wcp, twp) hipLaunchKernelGGL((cols256p_kernel<DIRV, LCV>), dim3(pgrid < ntiles ? pgrid : ntiles), dim3(16 << LCV), 0, 0, buf, k, wcp, twp, ntiles)
#elif defined(KEEP)
#define COLS(DIRV, buf, srcbuf, np, wcp, twp) KEEP_LAUNCH(DIRV, 0, buf, srcbuf, np, wcp, twp)
#else
#define COLS(DIRV, buf, srcbuf, np, wcp, twp) hipLaunchKernelGGL((cols256_kernel<DIRV, LCV>), dim3((unsigned)((np) << (k - 8 - LCV))), dim3(16 << LCV), 0, 0, buf, srcbuf, k, wcp, twp, sr::xcd_grouped_tiles((np) << (k - 8 - LCV), sr::gl::kColsXcdGroup))
#endif
    double acc[4] = {0, 0, 0, 0};
#ifdef INPLACE
    for (int r = -1; r < reps; r++) {
        CK(hipEventRecord(ev[0]));
        COLS(0, a, a, npoly, T.wcf, T.twist_f);
        CK(hipEventRecord(ev[1]));
        COLS(0, b, b, npoly, T.wcf, T.twist_f);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(ROWS_KERNEL, dim3(blocks), dim3(256), 0, 0, a, b, a, T);
        CK(hipEventRecord(ev[3]));
        COLS(1, a, a, npoly, T.wci, T.twist_i_mul);
        CK(hipEventRecord(ev[4]));
        CK(hipDeviceSynchronize());
        if (r < 0) continue;
        for (int i = 0; i < 4; i++) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
    }
#else
    // the library's arrangement: chunks of launches, operands read in place, intermediates in a chunk-sized scratch pair
    const size_t ch = argc > 3 ? strtoull(argv[3], 0, 10) : (npoly >= 64 ? npoly / 8 : npoly);
    u64 *sa, *sb;
    CK(hipMalloc(&sa, ch << (k + 3))); CK(hipMalloc(&sb, ch << (k + 3)));
    const size_t nch = (npoly + ch - 1) / ch;
    std::vector<hipEvent_t> cev(nch * 5);
    for (auto &e : cev) CK(hipEventCreate(&e));
    double wall = 0;
    for (int r = -1; r < reps; r++) {
        for (size_t c = 0; c < nch; c++) {
            const size_t np = (c + 1) * ch <= npoly ? ch : npoly - c * ch;
            u64 *ac = a + ((c * ch) << k), *bc = b + ((c * ch) << k);
            CK(hipEventRecord(cev[c * 5 + 0]));
#ifdef PAIR
            CK(hipEventRecord(cev[c * 5 + 1]));
            {
                const unsigned tiles = (unsigned)(np << (k - 12));
                hipLaunchKernelGGL(cols256_pair_kernel, dim3(2 * tiles), dim3(256), 0, 0, sa, ac, sb, bc, k, T.wcf, T.twist_f, tiles,
                                   sr::xcd_grouped_tiles(tiles, sr::gl::kColsXcdGroup));
            }
#else
            COLS(0, sa, ac, np, T.wcf, T.twist_f);
            CK(hipEventRecord(cev[c * 5 + 1]));
            COLS(0, sb, bc, np, T.wcf, T.twist_f);
#endif
            CK(hipEventRecord(cev[c * 5 + 2]));
            hipLaunchKernelGGL(ROWS_KERNEL, dim3((unsigned)(np << (k - 12))), dim3(256), 0, 0, sa, sb, sa, T);
            CK(hipEventRecord(cev[c * 5 + 3]));
            COLS(1, ac, sa, np, T.wci, T.twist_i_mul);
            CK(hipEventRecord(cev[c * 5 + 4]));
        }
        CK(hipDeviceSynchronize());
        if (r < 0) continue;
        for (size_t c = 0; c < nch; c++)
            for (int i = 0; i < 4; i++) { float ms; CK(hipEventElapsedTime(&ms, cev[c * 5 + i], cev[c * 5 + i + 1])); acc[i] += ms; }
        float ms; CK(hipEventElapsedTime(&ms, cev[0], cev[nch * 5 - 1])); wall += ms;
    }
    printf("chunks of %zu: wall %.3f ms per batch; ", ch, wall / reps);
#ifdef STREAMS
    {   // the same chunks dealt round-robin to STREAMS streams, each with its own scratch pair: another stream's kernels fill a launch's tail
        hipStream_t st[STREAMS];
        u64 *ssa[STREAMS], *ssb[STREAMS];
        for (int i = 0; i < STREAMS; i++) { CK(hipStreamCreate(&st[i])); CK(hipMalloc(&ssa[i], ch << (k + 3))); CK(hipMalloc(&ssb[i], ch << (k + 3))); }
        double tot = 0;
        for (int r = -1; r < reps; r++) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            // the chunk schedule: (first element, elements, stream).  HALF_FIRST: the second lane starts with half a chunk, so that the
            // two lanes run half a chunk out of phase (one in its column passes while the other is in its rows kernel)
            std::vector<std::array<size_t, 3>> sched;
            {
                size_t pos = 0;
                int si = 0;
#ifdef HALF_FIRST
                sched.push_back({0, ch, 0});
                sched.push_back({ch, ch / 2, 1});
                pos = ch + ch / 2;
#endif
                while (pos < npoly) {
                    const size_t np = npoly - pos < ch ? npoly - pos : ch;
                    sched.push_back({pos, np, (size_t)si});
                    si = (si + 1) % STREAMS;
                    pos += np;
                }
            }
            for (const auto &it : sched) {
                const int si = (int)it[2];
                const size_t np = it[1];
                u64 *ac = a + (it[0] << k), *bc = b + (it[0] << k);
                const unsigned cb = (unsigned)(np << (k - 8 - LCV));
                const unsigned grp = sr::xcd_grouped_tiles(cb, sr::gl::kColsXcdGroup);
#ifdef LIBFLOW  // the library's flow for a *= b: a's intermediates live in a itself, only b's go through the lane's scratch
                u64 *ia = ac;
#else
                u64 *ia = ssa[si];
#endif
#ifdef PAIR
                hipLaunchKernelGGL(cols256_pair_kernel, dim3(2 * cb), dim3(256), 0, st[si], ia, ac, ssb[si], bc, k, T.wcf, T.twist_f, cb, grp);
#elif defined(KEEP)
                (void)cb; (void)grp;
                KEEP_LAUNCH(0, st[si], ia, ac, np, T.wcf, T.twist_f);
                KEEP_LAUNCH(0, st[si], ssb[si], bc, np, T.wcf, T.twist_f);
#else
                hipLaunchKernelGGL((cols256_kernel<0, LCV>), dim3(cb), dim3(16 << LCV), 0, st[si], ia, ac, k, T.wcf, T.twist_f, grp);
                hipLaunchKernelGGL((cols256_kernel<0, LCV>), dim3(cb), dim3(16 << LCV), 0, st[si], ssb[si], bc, k, T.wcf, T.twist_f, grp);
#endif
                hipLaunchKernelGGL(ROWS_KERNEL, dim3((unsigned)(np << (k - 12))), dim3(256), 0, st[si], ia, ssb[si], ia, T);
#if defined(KEEP)
                KEEP_LAUNCH(1, st[si], ac, ia, np, T.wci, T.twist_i_mul);
#else
                hipLaunchKernelGGL((cols256_kernel<1, LCV>), dim3(cb), dim3(16 << LCV), 0, st[si], ac, ia, k, T.wci, T.twist_i_mul, grp);
#endif
            }
            CK(hipDeviceSynchronize());
            if (r >= 0) tot += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        printf("[%d streams: %.3f ms per batch] ", STREAMS, tot / reps);
    }
#endif
#endif
#ifdef MIXED
    {   // the same four phases as ONE mixed-role launch (junk data: the roles work on four independent buffers)
        u64 *c2, *c3;
        CK(hipMalloc(&c2, n * 8)); CK(hipMalloc(&c3, n * 8));
        hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, c2, n, 4);
        hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, c3, n, 5);
        MixedArgs m{};
        m.fa_dst = a; m.fa_src = a; m.fb_dst = b; m.fb_src = b; m.rows_a = c2; m.rows_b = b; m.inv = c3;
        for (int r = 0; r < 4; r++) m.n_tiles[r] = blocks;
        const unsigned grid = ((4u * blocks + 255u) / 256u) * 256u;
        double tot = 0;
        for (int r = -1; r < reps; r++) {
            CK(hipEventRecord(ev[0]));
            hipLaunchKernelGGL(mixed256_kernel, dim3(grid), dim3(256), 0, 0, m, k, T, T.twist_i_mul);
            CK(hipEventRecord(ev[1]));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            if (r >= 0) tot += ms;
        }
        printf("mixed-role single launch over the same work: %.3f ms\n", tot / reps);
    }
#endif
#ifdef PIPE
    {   // chunk pipeline: launch i = fwd cols of a, b for chunk i + rows of chunk i - 1 + inverse cols of chunk i - 2, mixed roles
        u64 *sc[2];
        for (size_t N : {32, 48, 64, 96, 128, 256, 512, 1024}) {
            if (N > npoly) continue;
            CK(hipMalloc(&sc[0], N << 19)); CK(hipMalloc(&sc[1], N << 19));
            const size_t nch = (npoly + N - 1) / N;
            double tot = 0;
            for (int r = -1; r < reps; r++) {
                CK(hipEventRecord(ev[0]));
                for (size_t i = 0; i < nch + 2; i++) {
                    MixedArgs m{};
                    auto cnt = [&](size_t c) { return c < nch ? (unsigned)(((c + 1) * N <= npoly ? N : npoly - c * N) << 4) : 0u; };
                    if (i < nch) {
                        m.fa_dst = a + ((i * N) << k); m.fa_src = m.fa_dst;      // in place on a (timing only)
                        m.fb_dst = sc[i & 1]; m.fb_src = b + ((i * N) << k);
                        m.n_tiles[0] = m.n_tiles[1] = cnt(i);
                    }
                    if (i >= 1 && i - 1 < nch) { m.rows_a = a + (((i - 1) * N) << k); m.rows_b = sc[(i - 1) & 1]; m.n_tiles[2] = cnt(i - 1); }
                    if (i >= 2 && i - 2 < nch) { m.inv = a + (((i - 2) * N) << k); m.n_tiles[3] = cnt(i - 2); }
                    unsigned mx = 0;
                    for (int q = 0; q < 4; q++) mx = m.n_tiles[q] > mx ? m.n_tiles[q] : mx;
                    const unsigned grid = ((4u * mx + 255u) / 256u) * 256u;
                    hipLaunchKernelGGL(mixed256_kernel, dim3(grid), dim3(256), 0, 0, m, k, T, T.twist_i_mul);
                }
                CK(hipEventRecord(ev[1]));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
                if (r >= 0) tot += ms;
            }
            printf("pipeline, chunks of %4zu elements (%zu launches): %.3f ms per batch\n", N, nch + 2, tot / reps);
            CK(hipFree(sc[0])); CK(hipFree(sc[1]));
        }
    }
#endif
    printf("npoly %zu  cols_a %.3f  cols_b %.3f  rows %.3f  cols_inv %.3f  total %.3f ms\n", npoly, acc[0] / reps, acc[1] / reps,
           acc[2] / reps, acc[3] / reps, (acc[0] + acc[1] + acc[2] + acc[3]) / reps);
    return 0;
}
