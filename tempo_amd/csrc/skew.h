// Time brackets of the skew AS-OF join (python/tempo/tsdf.py:164-190), shared by the join and
// by the time-sliced key shuffle so that both cut time at exactly the same places.
#pragma once

#include "common.h"

namespace tmp {

// Bracket rule of tsdf.py:178-185, same IEEE operations in the same order (the library is
// built with -ffp-contract=off).  Spark casts double -> int by truncation, saturating, NaN->0.
__device__ __forceinline__ int32_t spark_d2i(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int32_t)0x80000000;
  return (int32_t)x;
}

struct SkewParams {
  int enabled;
  int v_is_int;
  int ts_is_seconds;
  double v;
  double one_minus_fraction;
};

__device__ __forceinline__ double bracket_seconds(int64_t ts, const SkewParams &sp) {
  return sp.ts_is_seconds ? (double)ts : (double)ts / 1000000.0;
}

// ts_partition of tsdf.py:178-180 for a row whose double timestamp is d.
__device__ __forceinline__ double bracket_part(double d, const SkewParams &sp) {
  int32_t q = spark_d2i(d / sp.v);
  return sp.v_is_int ? (double)(int32_t)((uint32_t)(int32_t)sp.v * (uint32_t)q) : sp.v * (double)q;
}

// A right row is visible from the left row's bracket when it lies in that bracket, or in the
// one before it with a remainder inside the carried-over tail (tsdf.py:181-188).  The
// remainder -- one more division -- is only worked out in the second case.
__device__ __forceinline__ bool bracket_visible(int64_t ts_right, double part_left,
                                                const SkewParams &sp) {
  double d = bracket_seconds(ts_right, sp);
  double pr = bracket_part(d, sp);
  if (pr == part_left) return true;
  double nxt = sp.v_is_int ? (double)(int32_t)((uint32_t)(int32_t)pr + (uint32_t)(int32_t)sp.v)
                           : pr + sp.v;
  if (nxt != part_left) return false;
  return (d - pr) / sp.v >= sp.one_minus_fraction;
}

// Quotient of tsdf.py:179, (ts_col_double / v).cast("integer"): the index of a row's bracket.
__device__ __forceinline__ int32_t bracket_index(int64_t ts, const SkewParams &sp) {
  return spark_d2i(bracket_seconds(ts, sp) / sp.v);
}

}  // namespace tmp
