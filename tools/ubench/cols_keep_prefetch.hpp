// Experiment (round 5): cols256_keep_kernel with the NEXT ring element's 16 coefficients requested while the current one is in its
// second register pass -- the loads of an iteration are issued right after the LDS writes of pass A (the registers are dead there)
// into a second register set.  32 more VGPRs than the product kernel (120 / 126): compiled for two workgroups per CU.
// Timing harness only (tools/ubench/gl_bench.hip -DKEEPPF=<groups>); same values as the product kernel by construction.
#pragma once
namespace sr {
namespace gl {

template <int DIR>
__global__ __launch_bounds__(256, 2) void cols256_keep_pf_kernel(u64 *data, const u64 *src, u64 *data2, const u64 *src2, int k,
                                                                 const u64 *__restrict__ wc, const u64 *__restrict__ twist, unsigned npoly,
                                                                 unsigned groups) {
    using CT = ColsTile<4>;
    __shared__ u64 lds[CT::kElems];
    __shared__ u64 wl[256];
    const int t = threadIdx.x;
    wl[t] = wc[t];
    const int ls = k - 8;
    const unsigned chunks = 1u << (ls - 4);
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned ci = slot & (chunks - 1u), grp = slot >> (ls - 4);
    const int col = t & 15, rg = t >> 4;
    const unsigned i = ci * 16u + (unsigned)col;
    const char *tb = reinterpret_cast<const char *>(twist);
    const unsigned leg = 8u << ls;
    const unsigned offA0 = (((unsigned)rg << ls) + i) * 8u;
    const unsigned offB0 = (((unsigned)rg << (ls + 4)) + i) * 8u;
    u64 x[16], y[16], tw[16];
#pragma unroll
    for (int sg = 0; sg < 16; sg++) tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB0 + (unsigned)sg * leg));
    __syncthreads();
    const unsigned total = data2 ? 2u * npoly : npoly;
    const unsigned step = 8u * groups;
    auto src_of = [&](unsigned it) { return reinterpret_cast<const char *>((it >= npoly ? src2 : src) + ((size_t)(it >= npoly ? it - npoly : it) << k)); };
    auto dst_of = [&](unsigned it) { return reinterpret_cast<char *>((it >= npoly ? data2 : data) + ((size_t)(it >= npoly ? it - npoly : it) << k)); };
    unsigned it = xcd + 8u * grp;
    if (it < total) {
        const char *ps = src_of(it);
#pragma unroll
        for (int j = 0; j < 16; j++)
            x[j] = DIR == 0 ? ld_stream(reinterpret_cast<const u64 *>(ps + (offA0 + (unsigned)j * 16u * leg)))
                            : ld_scratch(reinterpret_cast<const u64 *>(ps + (offB0 + (unsigned)j * leg)));
    }
    for (; it < total; it += step) {
        const bool more = it + step < total;
        char *pb = dst_of(it);
        unsigned offA = offA0, offB = offB0;
        asm volatile("" : "+v"(offA), "+v"(offB));
        if (DIR == 0) {
            cols_stage_fwd<0>(x);
            cols_stage_fwd<1>(x);
            cols_stage_fwd<2>(x);
            cols_stage_fwd<3>(x);
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if ((h & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 0; h < 16; h++) lds[CT::idx(16 * h + rg, col)] = x[h];
            if (more) {  // the next element's coefficients: in flight during the exchange, pass B and the stores
                const char *pn = src_of(it + step);
#pragma unroll
                for (int j = 0; j < 16; j++) y[j] = ld_stream(reinterpret_cast<const u64 *>(pn + (offA + (unsigned)j * 16u * leg)));
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[CT::idx(16 * rg + j, col)];
            dft16_fwd_hot(x);
#pragma unroll
            for (int sg = 0; sg < 16; sg++) st_scratch(reinterpret_cast<u64 *>(pb + (offB + (unsigned)sg * leg)), G::mul(x[sg], tw[sg]));
        } else {
#pragma unroll
            for (int sg = 0; sg < 16; sg++) {
                x[sg] = G::mul(x[sg], tw[sg]);
                if ((sg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            dft16_inv_hot(x);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[CT::idx(16 * rg + j, col)] = x[j];
            if (more) {
                const char *pn = src_of(it + step);
#pragma unroll
                for (int sg = 0; sg < 16; sg++) y[sg] = ld_scratch(reinterpret_cast<const u64 *>(pn + (offB + (unsigned)sg * leg)));
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = lds[CT::idx(16 * h + rg, col)];
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if ((h & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            cols_stage_inv<3>(x);
            cols_stage_inv<2>(x);
            cols_stage_inv<1>(x);
            cols_stage_inv<0>(x);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) st_result(reinterpret_cast<u64 *>(pb + (offA + (unsigned)jj * 16u * leg)), x[jj]);
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = y[j];
        }
    }
}

}  // namespace gl
}  // namespace sr
