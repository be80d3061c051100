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
  // 1 / (v * 1e6) (or 1 / v for a column of seconds), or 0: no shortcut (skew_params_set fills it)
  double inv;
};

inline void skew_params_set(SkewParams *sp, int enabled, int v_is_int, int ts_is_seconds, double v,
                            double one_minus_fraction) {
  sp->enabled = enabled;
  sp->v_is_int = v_is_int;
  sp->ts_is_seconds = ts_is_seconds;
  sp->v = v;
  sp->one_minus_fraction = one_minus_fraction;
  const double c = ts_is_seconds ? v : v * 1000000.0;
  sp->inv = (enabled && c == c && c != 0.0) ? 1.0 / c : 0.0;
  if (!(sp->inv == sp->inv) || sp->inv - sp->inv != 0.0) sp->inv = 0.0;  // NaN or infinite: no shortcut
}

__device__ __forceinline__ double bracket_seconds(int64_t ts, const SkewParams &sp) {
  return sp.ts_is_seconds ? (double)ts : (double)ts / 1000000.0;
}

// Quotient of tsdf.py:179, (ts_col_double / v).cast("integer"): the index of a row's bracket.
// The two divisions (ts / 1e6, then / v) are what the reference computes, and the slow way here.
// The quick way multiplies by 1 / (1e6 v): that product and the reference's quotient start from the
// same converted timestamp and are each three roundings or fewer away from the true ratio (two
// divisions there; 1e6 v, its reciprocal and the product here), hence within 5 x 2^-53 |q| of each
// other -- the margin used is 2^-49 |q|.  When every number that close to the product truncates to
// the same integer, that integer is the reference's; only a timestamp that sits on a bracket
// boundary to within the margin (a whole multiple of v seconds does) takes the divisions.
__device__ __forceinline__ int32_t bracket_index(int64_t ts, const SkewParams &sp) {
  if (sp.inv != 0.0) {
    const double qa = (double)ts * sp.inv;
    const double m = fabs(qa) * 1.7763568394002505e-15;  // 2^-49
    const double a = qa - m, b = qa + m;
    if (a > -2147483000.0 && b < 2147483000.0) {
      const int32_t ia = (int32_t)a, ib = (int32_t)b;
      if (ia == ib) return ia;
    }
  }
  return spark_d2i(bracket_seconds(ts, sp) / sp.v);
}

// ts_partition of tsdf.py:178-180 of a row with bracket index q.
__device__ __forceinline__ double bracket_part_of(int32_t q, const SkewParams &sp) {
  return sp.v_is_int ? (double)(int32_t)((uint32_t)(int32_t)sp.v * (uint32_t)q) : sp.v * (double)q;
}
__device__ __forceinline__ double bracket_part_ts(int64_t ts, const SkewParams &sp) {
  return bracket_part_of(bracket_index(ts, sp), sp);
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
  double pr = bracket_part_ts(ts_right, sp);
  if (pr == part_left) return true;
  double nxt = sp.v_is_int ? (double)(int32_t)((uint32_t)(int32_t)pr + (uint32_t)(int32_t)sp.v)
                           : pr + sp.v;
  if (nxt != part_left) return false;
  return (bracket_seconds(ts_right, sp) - pr) / sp.v >= sp.one_minus_fraction;
}

}  // namespace tmp
