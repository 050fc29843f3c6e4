// Round-2 experiment, NOT part of the library (measured slower, see DESIGN_APPENDIX.md A.2): a persistent, register-prefetching form of
// gl::cols256_kernel.  Included by gl_bench.hip when built with -DPERSIST=<CUs>.
#pragma once
namespace sr {
namespace gl {
// Persistent, software-pipelined form of cols256_kernel: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and
// requests the next tile's coefficients (and, inverse, its twist factors) while the current tile is in its second register
// pass, so that a wide tile (LC = 6: 512-byte segments, one 1024-lane workgroup per CU) keeps HBM busy across its barriers.
// Same arithmetic, same tile order inside a ring element, same results as cols256_kernel.
template <int DIR, int LC>
__global__ __launch_bounds__(16 << LC, 4) void cols256p_kernel(u64 *data, int k, const u64 *__restrict__ wc,
                                                              const u64 *__restrict__ twist, unsigned ntiles) {
    using CT = ColsTile<LC>;
    constexpr int C = CT::C;
    __shared__ u64 lds[CT::kElems];
    __shared__ u64 wl[256];
    const int t = threadIdx.x;
    if (t < 256) wl[t] = wc[t];
    __syncthreads();
    const int ls = k - 8;  // log2 N2
    const int col = t & (C - 1), rg = t >> LC;
    const char *tb = reinterpret_cast<const char *>(twist);
    const unsigned leg = 8u << ls;
    u64 x[16], nx[16];
    unsigned tile = blockIdx.x;
    if (tile >= ntiles) return;
    auto tile_base = [&](unsigned tl, unsigned &i) -> char * {
        const unsigned ci = tl & ((1u << (ls - LC)) - 1u);
        const size_t poly = tl >> (ls - LC);
        i = ci * (unsigned)C + (unsigned)col;
        return reinterpret_cast<char *>(data + (poly << k));
    };
    if (DIR == 0) {
        {
            unsigned i;
            char *pb = tile_base(tile, i);
            const unsigned offA = (((unsigned)rg << ls) + i) * 8u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) nx[jj] = *reinterpret_cast<const u64 *>(pb + (offA + (unsigned)jj * 16u * leg));
        }
        for (; tile < ntiles; tile += gridDim.x) {
            unsigned i;
            char *pb = tile_base(tile, i);
            const unsigned offB = (((unsigned)rg << (ls + 4)) + i) * 8u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) x[jj] = nx[jj];
            cols_stage_fwd<0>(x);
            cols_stage_fwd<1>(x);
            cols_stage_fwd<2>(x);
            cols_stage_fwd<3>(x);
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = G::mul(x[h], wl[h * 16 + rg]);
#pragma unroll
            for (int h = 0; h < 16; h++) lds[CT::idx(16 * h + rg, col)] = x[h];
            __syncthreads();
            const unsigned nt = tile + gridDim.x;
            if (nt < ntiles) {  // uniform: the next tile's coefficients travel while pass B runs
                unsigned ni;
                char *npb = tile_base(nt, ni);
                const unsigned noffA = (((unsigned)rg << ls) + ni) * 8u;
#pragma unroll
                for (int jj = 0; jj < 16; jj++) nx[jj] = *reinterpret_cast<const u64 *>(npb + (noffA + (unsigned)jj * 16u * leg));
            }
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[CT::idx(16 * rg + j, col)];
            __syncthreads();  // every lane has read its pass-B legs before the next tile's pass-A writes land
            u64 tw[16];
#pragma unroll
            for (int sg = 0; sg < 16; sg++) tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB + (unsigned)sg * leg));
            dft16_fwd(x);
#pragma unroll
            for (int sg = 0; sg < 16; sg++)
                *reinterpret_cast<u64 *>(pb + (offB + (unsigned)sg * leg)) = G::mul(x[sg], tw[sg]);
        }
    } else {
        u64 ntw[16];
        {
            unsigned i;
            char *pb = tile_base(tile, i);
            const unsigned offB = (((unsigned)rg << (ls + 4)) + i) * 8u;
#pragma unroll
            for (int sg = 0; sg < 16; sg++) {
                nx[sg] = *reinterpret_cast<const u64 *>(pb + (offB + (unsigned)sg * leg));
                ntw[sg] = *reinterpret_cast<const u64 *>(tb + (offB + (unsigned)sg * leg));
            }
        }
        for (; tile < ntiles; tile += gridDim.x) {
            unsigned i;
            char *pb = tile_base(tile, i);
            const unsigned offA = (((unsigned)rg << ls) + i) * 8u;
#pragma unroll
            for (int sg = 0; sg < 16; sg++) x[sg] = G::mul(nx[sg], ntw[sg]);
            dft16_inv(x);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[CT::idx(16 * rg + j, col)] = x[j];
            __syncthreads();
            const unsigned nt = tile + gridDim.x;
            if (nt < ntiles) {
                unsigned ni;
                char *npb = tile_base(nt, ni);
                const unsigned noffB = (((unsigned)rg << (ls + 4)) + ni) * 8u;
#pragma unroll
                for (int sg = 0; sg < 16; sg++) {
                    nx[sg] = *reinterpret_cast<const u64 *>(npb + (noffB + (unsigned)sg * leg));
                    ntw[sg] = *reinterpret_cast<const u64 *>(tb + (noffB + (unsigned)sg * leg));
                }
            }
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = lds[CT::idx(16 * h + rg, col)];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = G::mul(x[h], wl[h * 16 + rg]);
            cols_stage_inv<3>(x);
            cols_stage_inv<2>(x);
            cols_stage_inv<1>(x);
            cols_stage_inv<0>(x);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) *reinterpret_cast<u64 *>(pb + (offA + (unsigned)jj * 16u * leg)) = x[jj];
        }
    }
}

}  // namespace gl
}  // namespace sr
