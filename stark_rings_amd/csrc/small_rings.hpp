#pragma once
#include "fields.hpp"
namespace sr {
enum { SMALL_G24_CRT, SMALL_G24_ICRT, SMALL_G24_MUL, SMALL_G24_RINGMUL, SMALL_G24_REDUCE,
       SMALL_B72_CRT, SMALL_B72_ICRT, SMALL_B72_MUL, SMALL_B72_RINGMUL, SMALL_B72_REDUCE };
struct SmallRingConsts { int dummy; };
inline int small_init(SmallRingConsts &, bool) { return 0; }
inline void small_destroy(SmallRingConsts &) {}
inline int small_launch(const SmallRingConsts &, int, const uint64_t *, const uint64_t *, size_t, uint64_t *, size_t, hipStream_t) { return 1; }
}  // namespace sr
