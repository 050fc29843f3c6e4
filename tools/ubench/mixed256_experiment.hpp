// Round-2 experiment, NOT part of the library (measured: no gain, DESIGN_APPENDIX.md A.2): one launch whose workgroups take four roles
// (forward column tiles of a and b, fused rows tiles, inverse column tiles), alone and as a software pipeline over chunks of the batch.
// Included by gl_bench.hip when built with -DMIXED or -DPIPE.
#pragma once
namespace sr {
namespace gl {
// ------------------------------------------------------------------------------------------------
// mixed256: ONE launch whose workgroups take four different roles -- forward column tiles of a and of b for chunk i of the batch,
// fused rows tiles for chunk i - 1, inverse column tiles for chunk i - 2 (a software pipeline over chunks: every dependency
// points to an earlier LAUNCH, nothing synchronises inside one).  The column passes are bound by their strided HBM pattern with the
// VALU two-thirds busy, the rows kernel by VALU issue with HBM half busy; as separate launches each phase leaves one of the two
// idle on every CU.  Mixed, a CU holds tiles of all roles at once and both stay busy.
// Role of a workgroup: blocks are dealt round-robin over the 8 XCDs and, inside an XCD, over its CUs; with j = blockIdx.x >> 3 the
// role (j + (j >> 5)) & 3 walks through the four roles along the blocks one CU receives AND along consecutive j.  The tiles of a
// role keep the order of a plain launch (tile index = rank of the block among the blocks of its role).
// ------------------------------------------------------------------------------------------------
struct MixedArgs {
    u64 *fa_dst;        // role 0: forward columns of a: src -> dst (out chunk i)
    const u64 *fa_src;
    u64 *fb_dst;        // role 1: forward columns of b: src -> scratch[i & 1]
    const u64 *fb_src;
    u64 *rows_a;        // role 2: rows of chunk i - 1 (in place on out) ...
    const u64 *rows_b;  //         ... against its scratch
    u64 *inv;           // role 3: inverse columns of chunk i - 2 (in place on out)
    unsigned n_tiles[4];
};
__global__ __launch_bounds__(256, 4) void mixed256_kernel(MixedArgs m, int k, Tables T, const u64 *twist_i) {
    __shared__ u64 lds[kLdsElems];
    const unsigned j = blockIdx.x >> 3, x = blockIdx.x & 7u;
    const unsigned role = (j + (j >> 5)) & 3u;
    const unsigned tile = (((j >> 5) * 8u + ((j & 31u) >> 2)) << 3) + x;
    if (tile >= m.n_tiles[role]) return;  // uniform per workgroup
    switch (role) {
        case 0: cols256_tile<0, 4>(tile, m.fa_dst, m.fa_src, k, T.wcf, T.twist_f, lds); break;
        case 1: cols256_tile<0, 4>(tile, m.fb_dst, m.fb_src, k, T.wcf, T.twist_f, lds); break;
        case 2: rows256_tile<2>(tile, m.rows_a, m.rows_b, m.rows_a, T, lds); break;
        default: cols256_tile<1, 4>(tile, m.inv, m.inv, k, T.wci, twist_i, lds); break;
    }
}
}  // namespace gl
}  // namespace sr
