// Round-4 experiment, included by gl_bench.hip when built with -DKEEP: a column pass whose workgroup OWNS one column
// chunk ci and walks over the ring elements of the launch, so that the 16 twist factors a lane needs (they depend on the column
// and the leg, not on the ring element) are loaded once per workgroup instead of once per tile, and the 256-entry W table sits in
// 2 KiB of LDS.  Unlike the round-2 cols256p experiment nothing is prefetched: registers hold x[16] and tw[16], as they do in pass B
// of the plain kernel already.
// Workgroup -> (XCD, ci, group): blockIdx & 7 is the XCD the hardware gives the workgroup; inside an XCD slot = blockIdx >> 3 =
// group * chunks + ci; the workgroup handles ring elements xcd + 8 (group + groups r), r = 0, 1, ...: all the column chunks of one
// ring element are in flight on ONE XCD at the same time (what xcd_tile() arranges for the plain launch).
#pragma once
namespace sr {
namespace gl {
#ifndef KEEP_P
#define KEEP_P 4
#endif
template <int DIR>
#ifndef KEEP_SB
#define KEEP_SB 1
#endif
#ifndef KEEP_WAVES
#define KEEP_WAVES 4
#endif
__global__ __launch_bounds__(256, KEEP_WAVES) void cols256_keep_kernel(u64 *data, const u64 *src, int k, const u64 *__restrict__ wc,
                                                              const u64 *__restrict__ twist, unsigned npoly, unsigned groups) {
    using CT = ColsTile<4>;
    __shared__ u64 lds[CT::kElems];
    __shared__ u64 wl[256];
    const int t = threadIdx.x;
    wl[t] = wc[t];
    const int ls = k - 8;  // log2 N2
    const unsigned chunks = 1u << (ls - 4);
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned ci = slot & (chunks - 1u), grp = slot >> (ls - 4);
    const int col = t & 15, rg = t >> 4;
    const unsigned i = ci * 16u + (unsigned)col;
    const char *tb = reinterpret_cast<const char *>(twist);
    const unsigned leg = 8u << ls;
    const unsigned offA0 = (((unsigned)rg << ls) + i) * 8u;
    const unsigned offB0 = (((unsigned)rg << (ls + 4)) + i) * 8u;
    u64 x[16], tw[16];
#pragma unroll
    for (int sg = 0; sg < 16; sg++) tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB0 + (unsigned)sg * leg));
    __syncthreads();
    for (unsigned poly = xcd + 8u * grp; poly < npoly; poly += 8u * groups) {
        char *pb = reinterpret_cast<char *>(data + ((size_t)poly << k));
        const char *ps = reinterpret_cast<const char *>(src + ((size_t)poly << k));
        // opaque per iteration: otherwise the 32 per-access offsets are hoisted out of the loop as invariants and live across it
        unsigned offA = offA0, offB = offB0;
        asm volatile("" : "+v"(offA), "+v"(offB));
        if (DIR == 0) {
#pragma unroll
            for (int jj = 0; jj < 16; jj++) x[jj] = ld_stream(reinterpret_cast<const u64 *>(ps + (offA + (unsigned)jj * 16u * leg)));
            cols_stage_fwd<0, KEEP_P>(x);
            cols_stage_fwd<1, KEEP_P>(x);
            cols_stage_fwd<2, KEEP_P>(x);
            cols_stage_fwd<3, KEEP_P>(x);
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if (KEEP_SB && (h & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keeps the 16 table reads from all being hoisted in front
            }
#pragma unroll
            for (int h = 0; h < 16; h++) lds[CT::idx(16 * h + rg, col)] = x[h];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[CT::idx(16 * rg + j, col)];
            dft16_fwd_hot<false, KEEP_P>(x);
#pragma unroll
            for (int sg = 0; sg < 16; sg++) st_scratch(reinterpret_cast<u64 *>(pb + (offB + (unsigned)sg * leg)), G::mul(x[sg], tw[sg]));
        } else {
#pragma unroll
            for (int sg = 0; sg < 16; sg++) x[sg] = ld_scratch(reinterpret_cast<const u64 *>(ps + (offB + (unsigned)sg * leg)));
#pragma unroll
            for (int sg = 0; sg < 16; sg++) {
                x[sg] = G::mul(x[sg], tw[sg]);
                if (KEEP_SB && (sg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            dft16_inv_hot<KEEP_P>(x);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[CT::idx(16 * rg + j, col)] = x[j];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = lds[CT::idx(16 * h + rg, col)];
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if (KEEP_SB && (h & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keeps the 16 table reads from all being hoisted in front
            }
            cols_stage_inv<3, KEEP_P>(x);
            cols_stage_inv<2, KEEP_P>(x);
            cols_stage_inv<1, KEEP_P>(x);
            cols_stage_inv<0, KEEP_P>(x);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) st_result(reinterpret_cast<u64 *>(pb + (offA + (unsigned)jj * 16u * leg)), x[jj]);
        }
        __syncthreads();  // every lane has read the exchange before the next ring element's writes land
    }
}
}  // namespace gl
}  // namespace sr
