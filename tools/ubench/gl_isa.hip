// Development aid: instantiates the Goldilocks hot kernels alone so that `hipcc -S` takes seconds, not minutes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o gl.s tools/ubench/gl_isa.hip && python tools/isa_count.py gl.s
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
using namespace sr::gl;
template __global__ void sr::gl::rows256_kernel<2>(u64 *, const u64 *, u64 *, Tables);
template __global__ void sr::gl::rows256_kernel<3>(u64 *, const u64 *, u64 *, Tables);
template __global__ void sr::gl::rows256_kernel<0>(u64 *, const u64 *, u64 *, Tables);
template __global__ void sr::gl::rows256_kernel<1>(u64 *, const u64 *, u64 *, Tables);
template __global__ void sr::gl::cols256_kernel<0, 4>(u64 *, const u64 *, int, const u64 *, const u64 *, unsigned);
template __global__ void sr::gl::cols256_kernel<1, 4>(u64 *, const u64 *, int, const u64 *, const u64 *, unsigned);
template __global__ void sr::gl::cols256_keep_kernel<0>(u64 *, const u64 *, u64 *, const u64 *, int, const u64 *, const u64 *, unsigned, unsigned);
template __global__ void sr::gl::cols256_keep_kernel<1>(u64 *, const u64 *, u64 *, const u64 *, int, const u64 *, const u64 *, unsigned, unsigned);
template __global__ void sr::gl::cols256_kernel<0, 5>(u64 *, const u64 *, int, const u64 *, const u64 *, unsigned);
template __global__ void sr::gl::cols256_kernel<1, 5>(u64 *, const u64 *, int, const u64 *, const u64 *, unsigned);
__global__ void probe_mul(u64 *x, const u64 *w) { x[threadIdx.x] = sr::Goldilocks::mul(x[threadIdx.x], w[threadIdx.x]); }
__global__ void probe_add(u64 *x, const u64 *w) { x[threadIdx.x] = sr::Goldilocks::add(x[threadIdx.x], w[threadIdx.x]); }
__global__ void probe_sub(u64 *x, const u64 *w) { x[threadIdx.x] = sr::Goldilocks::sub(x[threadIdx.x], w[threadIdx.x]); }
template <int E> __global__ void probe_pow2(u64 *x) { x[threadIdx.x] = mul_pow2<E>(x[threadIdx.x]); }
template __global__ void probe_pow2<12>(u64 *);
template __global__ void probe_pow2<48>(u64 *);
template __global__ void probe_pow2<72>(u64 *);
