// Tuned Goldilocks path (placeholder until the register-radix kernels land): reports unsupported so
// that capi.hip routes Goldilocks through the generic kernels.
#pragma once
#include "fields.hpp"
namespace sr {
struct GoldilocksFastTables { int k = -1; };
inline bool gl_fast_supported(const GoldilocksFastTables &) { return false; }
inline int gl_fast_init(GoldilocksFastTables &t, int k, const uint64_t *, hipStream_t) { t.k = k; return 0; }
inline void gl_fast_destroy(GoldilocksFastTables &) {}
inline int gl_fast_fwd(const GoldilocksFastTables &, uint64_t *, size_t, hipStream_t) { return 1; }
inline int gl_fast_inv(const GoldilocksFastTables &, uint64_t *, size_t, hipStream_t) { return 1; }
inline int gl_fast_ring_mul(const GoldilocksFastTables &, uint64_t *, const uint64_t *, uint64_t *, size_t, hipStream_t) { return 1; }
}  // namespace sr
