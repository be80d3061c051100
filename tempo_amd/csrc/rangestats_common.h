// Device helpers shared by the two range-statistics paths (rangestats.hip: frames inside a
// block's LDS region; rangestats_wide.hip: frames of any length).
#pragma once
#include "common.h"

namespace tmp {

template <typename T>
struct Ord;
template <>
struct Ord<double> {  // Spark ordering: NaN greatest, -0.0 == 0.0
  static __device__ __forceinline__ bool lt(double a, double b) {
    if (a != a) return false;
    if (b != b) return true;
    return a < b;
  }
  static __device__ __forceinline__ double min_id() { return __builtin_nan(""); }
  static __device__ __forceinline__ double max_id() { return -__builtin_inf(); }
  // "finite" for the fast paths: neither NaN nor infinite, and small enough that squares of differences
  // and sums of thousands of them stay finite -- a frame with anything else is evaluated by Spark's own
  // recurrence value by value, overflow to infinity included
  static __device__ __forceinline__ bool finite(double v) { return __builtin_fabs(v) < 1e150; }
};
template <>
struct Ord<int64_t> {
  static __device__ __forceinline__ bool lt(int64_t a, int64_t b) { return a < b; }
  static __device__ __forceinline__ int64_t min_id() { return INT64_MAX; }
  static __device__ __forceinline__ int64_t max_id() { return INT64_MIN; }
  static __device__ __forceinline__ bool finite(int64_t) { return true; }
};

template <typename T>
struct RsOut {
  int32_t *perm;
  double *mean;
  int64_t *count;
  T *mn;
  T *mx;
  T *sum;
  double *stddev;
  double *zscore;
  uint8_t *valid;
  int vbit;  // bit of the sort words that holds this metric's "not NULL" flag
};

template <typename T>
__device__ __forceinline__ void rs_store(const RsOut<T> &o, int64_t i, int64_t n, double mean,
                                         T mn, T mx, T sum, bool sd_ok, double sd, bool x_ok,
                                         double xd) {
  o.count[i] = n;
  bool ok = n > 0;
  o.mean[i] = ok ? mean : 0.0;
  o.mn[i] = ok ? mn : (T)0;
  o.mx[i] = ok ? mx : (T)0;
  o.sum[i] = ok ? sum : (T)0;
  o.stddev[i] = sd_ok ? sd : 0.0;
  bool z_ok = x_ok && sd_ok && sd != 0.0;  // Spark Divide: NULL on a zero divisor
  o.zscore[i] = z_ok ? (xd - mean) / sd : 0.0;
  o.valid[i] = (uint8_t)((ok ? 1 : 0) | (sd_ok ? 2 : 0) | (z_ok ? 4 : 0));
}

// Exact row-sequential evaluation of one frame: the recurrence Spark's aggregates run
// (Sum / Average: running add; CentralMomentAgg: n, avg, m2 update per value).
template <typename T>
__device__ void rs_row_serial(const uint64_t *comp, const T *xs, int64_t lo, int64_t hi, int64_t i,
                              const RsOut<T> &o) {
  int64_t n = 0;
  T mn = Ord<T>::min_id(), mx = Ord<T>::max_id();
  double dsum = 0.0, avg = 0.0, m2 = 0.0, nn = 0.0;
  uint64_t isum = 0;
  for (int64_t j = lo; j <= hi; ++j) {
    if (!((comp[j] >> o.vbit) & 1ull)) continue;
    T v = xs[j];
    if (n == 0) { mn = v; mx = v; }
    else {
      if (Ord<T>::lt(v, mn)) mn = v;
      if (Ord<T>::lt(mx, v)) mx = v;
    }
    ++n;
    nn += 1.0;
    double vd = (double)v;
    dsum += vd;
    isum += (uint64_t)v;
    double delta = vd - avg;
    double delta_n = delta / nn;
    avg += delta_n;
    m2 += delta * (delta - delta_n);
  }
  T sum;
  if constexpr (sizeof(T) == 8 && ((T)0.5 == (T)0)) sum = (T)isum; else sum = (T)dsum;
  bool x_ok = ((comp[i] >> o.vbit) & 1ull) != 0;
  rs_store<T>(o, i, n, n ? dsum / (double)n : 0.0, mn, mx, sum, n > 1,
              n > 1 ? sqrt(m2 / (nn - 1.0)) : 0.0, x_ok, (double)xs[i]);
}

__device__ __forceinline__ int64_t lower_bound_k(const uint64_t *comp, int64_t lo, int64_t hi,
                                                 uint64_t key, int low) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if ((comp[mid] >> low) < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t upper_bound_k(const uint64_t *comp, int64_t lo, int64_t hi,
                                                 uint64_t key, int low) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if ((comp[mid] >> low) <= key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// First sort word (low bits dropped) inside the frame of a row whose word is k.
__device__ __forceinline__ uint64_t frame_start_key(uint64_t k, uint64_t prim_mask, uint64_t w) {
  uint64_t rel = k & prim_mask;
  return (k - rel) + (rel > w ? rel - w : 0);
}

// 1/a and sqrt(a) from the hardware approximations plus one Newton step each: a few ulps, far
// inside the 1e-9 the statistics are held to, at a fraction of the IEEE expansions' cost.
__device__ __forceinline__ double fast_rcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  return __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
}
// sqrt(a) as fast_sqrt gives it, and 1 / sqrt(a) from the same reciprocal square root (two coupled
// Newton steps on the half-reciprocal: within a few ulps).  a <= 0: both 0.
__device__ __forceinline__ double sqrt_and_rcp(double a, double *inv) {
  *inv = 0.0;
  if (!(a > 0.0)) return 0.0;
  double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, a);  // residual a - g^2
  g = __builtin_fma(d, h, g);
  r = __builtin_fma(-h, g, 0.5);       // once more for the reciprocal, against the corrected root
  h = __builtin_fma(h, r, h);
  *inv = h + h;
  return g;
}
__device__ __forceinline__ double fast_sqrt(double a) {
  if (!(a > 0.0)) return 0.0;
  double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, a);  // residual a - g^2
  return __builtin_fma(d, h, g);
}


// Wide path (frames that leave the on-chip region): completes the rows the region kernel marked
// with kDeferred in out_valid.  rs_wide_plan sizes / carves its workspace.
constexpr uint8_t kDeferred = 0x80;

struct RsWidePlan {
  double *ps, *pq;        // per row: inclusive within-chunk prefix of (x - pivot), (x - pivot)^2
  uint32_t *pc;           // ... of counts: low 16 bits non-NULL, high 16 bits non-finite
  uint64_t *pi;           // ... of the wrapping integer sum (int64 metrics)
  uint64_t *pmn, *smn, *pmx, *smx;  // per row: prefix / suffix min and max inside the chunk
  double *tot_s, *tot_q;  // per chunk totals
  uint64_t *tot_c, *tot_i;
  double *exc_s, *exc_q;  // per chunk: exclusive prefix of the totals inside its superblock
  uint64_t *exc_c, *exc_i;
  double *pivot;          // per superblock
  double *cmean, *cm2;    // per chunk: mean of its finite values (less the superblock's pivot) and their squared deviations from it, added up around the mean itself
  uint64_t *tab_mn, *tab_mx;  // sparse tables over chunk minima / maxima, [levels][chunks]
  int64_t chunks;
  int levels;
};

void rs_wide_plan(int64_t n, bool is_i64, Workspace &ws, RsWidePlan *p);
template <typename T>
int rs_wide_run(const uint64_t *comp, const T *xs, int64_t n, const Layout &lay, uint64_t w,
                const RsOut<T> &o, RsWidePlan &p, hipStream_t st);

// withRangeStats over (key, second) cells (rangestats_cells.hip): planned only for calls with
// any_order set that can take it; rs_cells_try runs it when the table's ranges (as the ordering
// step's scan read them, seconds as the primary key) allow, and says so in *done.
struct RsCellsPlan {
  int planned;
  int64_t cap;  // cells the tables were sized for
  double *sum, *total, *acc, *lin;
  uint64_t *mn, *mx;
  uint32_t *rows, *cnt;
  void *runs[2];  // 64-byte records per cell: [0] from the block's start, [1] to the block's end
  int32_t *nonfinite;
};
void rs_cells_plan(const tmp_range_stats_args *a, Workspace &ws, RsCellsPlan *p);
int rs_cells_try(const tmp_range_stats_args *a, RsCellsPlan &p, const RangeInfo &r, hipStream_t st, bool *done);

}  // namespace tmp
