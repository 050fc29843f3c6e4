// Development aid: the 4096-point rows kernels of config 4 alone (hipcc -S in seconds): VALU / VGPR / scratch of the fused product
// against its split forms.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o build_tmp/glrows.s tools/ubench/gl_rows_isa.hip && python tools/isa_count.py build_tmp/glrows.s rows_kernel
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
using namespace sr::gl;
template __global__ void sr::gl::rows_kernel<2, 0, false>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
template __global__ void sr::gl::rows_kernel<3, 0, false>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
template __global__ void sr::gl::rows_kernel<3, 0, false, 8>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
template __global__ void sr::gl::rows_kernel<3, 0, false, 4>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
template __global__ void sr::gl::rows_kernel<0, 0, false>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
template __global__ void sr::gl::rows_kernel<1, 0, false>(u64 *, const u64 *, u64 *, Tables, const u64 *, size_t);
