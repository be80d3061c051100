// withRangeStats: rolling RANGE-frame statistics.  Replaces WindowExec over
// rangeBetween(-W, 0) with mean/count/min/max/sum/stddev + zscore
// (python/tempo/tsdf.py:673-721; ordering :563-572).
//
// Rows are ordered by (key, floor-second); the metric's validity bit rides in bit 0 of the
// sort word.  A block owns 1024 consecutive output rows.  Its frames all lie inside one
// contiguous extent [lo(first row), hi(last row)]; when the extent fits in LDS the block
//   - loads the extent once, shifted by a block pivot (d = x - pivot),
//   - builds inclusive prefix sums of d, d*d and the valid / non-finite counts in LDS,
//   - builds a sparse table of minima / maxima over 16-row sub-blocks,
// and every row then reads its frame as differences of prefixes and one range-min/max query.
// Frames with non-finite values and blocks whose extent exceeds LDS take the exact
// row-sequential path (Spark's own Welford recurrence).
#include <string.h>

#include "common.h"

namespace tmp {

constexpr int RK_ROWS = 1024;
constexpr int RK_CAP = 3072;
constexpr int RK_THREADS = 512;
constexpr int RK_SB = 16;
constexpr int RK_NSB = RK_CAP / RK_SB;  // 192
constexpr int RK_LEVELS = 8;            // 2^7 < 192 <= 2^8

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
  static __device__ __forceinline__ bool finite(double v) { return __builtin_isfinite(v); }
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
    if (!(comp[j] & 1ull)) continue;
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
  bool x_ok = (comp[i] & 1ull) != 0;
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

template <typename T>
struct RsShared {
  static constexpr bool kIntT = ((T)0.5 == (T)0);
  T x[RK_CAP];
  double ps[RK_CAP];
  double pq[RK_CAP];
  uint64_t pi[kIntT ? RK_CAP : 1];  // exact integer prefix, int64 metrics only
  uint32_t pc[RK_CAP];   // low 16 bits: valid count, high 16 bits: non-finite count
  T mn[RK_LEVELS][RK_NSB];
  T mx[RK_LEVELS][RK_NSB];
  double wave_s[RK_THREADS / 64];
  double wave_q[RK_THREADS / 64];
  uint64_t wave_i[RK_THREADS / 64];
  uint32_t wave_c[RK_THREADS / 64];
  int64_t ext[2];
  int pivot_at;
};

template <typename T>
__global__ __launch_bounds__(RK_THREADS) void range_stats_kernel(const uint64_t *comp,
                                                                 const uint32_t *perm,
                                                                 const T *xs, int64_t n, Layout lay,
                                                                 uint64_t w, RsOut<T> o) {
  constexpr bool kInt = ((T)0.5 == (T)0);
  __shared__ RsShared<T> s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int low = lay.low_bits;
  const uint64_t pmask = lay.prim_mask();
  const int64_t t0 = (int64_t)blockIdx.x * RK_ROWS;
  const int64_t t1 = t0 + RK_ROWS < n ? t0 + RK_ROWS : n;

  // extent of the block: waves 0 and 1 search one end each (wave-uniform branches)
  if (wave < 2) {
    int64_t r;
    if (wave == 0) {
      uint64_t k = comp[t0] >> low;
      r = lower_bound_k(comp, 0, t0, frame_start_key(k, pmask, w), low);
    } else {
      uint64_t k = comp[t1 - 1] >> low;
      r = upper_bound_k(comp, t1 - 1, n, k, low) - 1;
    }
    if (lane == 0) s.ext[wave] = r;
  }
  if (tid == 0) s.pivot_at = RK_CAP;
  __syncthreads();
  const int64_t e0 = s.ext[0], e1 = s.ext[1];
  const int64_t len = e1 - e0 + 1;

  if (len > RK_CAP) {  // extent too long for LDS: exact sequential frames (block-uniform branch)
    for (int64_t i = t0 + tid; i < t1; i += RK_THREADS) {
      uint64_t k = comp[i] >> low;
      int64_t lo = lower_bound_k(comp, e0, i, frame_start_key(k, pmask, w), low);
      int64_t hi = upper_bound_k(comp, i, e1 + 1, k, low) - 1;
      o.perm[i] = (int32_t)perm[i];
      rs_row_serial<T>(comp, xs, lo, hi, i, o);
    }
    return;
  }

  // ---- load the extent; pick the pivot = first valid finite value
  constexpr int kPer = RK_CAP / RK_THREADS;  // 6
  T xv[kPer];
  bool ok[kPer], fin[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    int e = r * RK_THREADS + tid;
    bool in = e < len;
    uint64_t cw = in ? comp[e0 + e] : 0ull;
    ok[r] = in && (cw & 1ull);
    xv[r] = ok[r] ? xs[e0 + e] : (T)0;
    fin[r] = ok[r] && Ord<T>::finite(xv[r]);
    s.x[e] = xv[r];
    if (fin[r]) atomicMin(&s.pivot_at, e);
  }
  __syncthreads();
  const int pat = s.pivot_at;
  const double pivot = pat < RK_CAP ? (double)s.x[pat] : 0.0;

  // ---- inclusive prefixes over the extent, 512 elements per round
  double carry_s = 0.0, carry_q = 0.0;
  uint64_t carry_i = 0;
  uint32_t carry_c = 0;
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    int e = r * RK_THREADS + tid;
    double d = fin[r] ? (double)xv[r] - pivot : 0.0;
    double q = d * d;
    uint64_t iv = (kInt && ok[r]) ? (uint64_t)xv[r] : 0ull;
    uint32_t c = (ok[r] ? 1u : 0u) | ((ok[r] && !fin[r]) ? 0x10000u : 0u);
    double is = d, iq = q;
    uint64_t ii = iv;
    uint32_t ic = c;
    for (int off = 1; off < 64; off <<= 1) {
      double ts_ = __shfl_up(is, off, 64), tq = __shfl_up(iq, off, 64);
      uint32_t tc = __shfl_up(ic, off, 64);
      if (lane >= off) { is += ts_; iq += tq; ic += tc; }
      if (kInt) {
        uint64_t ti = __shfl_up(ii, off, 64);
        if (lane >= off) ii += ti;
      }
    }
    if (lane == 63) { s.wave_s[wave] = is; s.wave_q[wave] = iq; s.wave_c[wave] = ic; s.wave_i[wave] = ii; }
    __syncthreads();
    double bs = carry_s, bq = carry_q;
    uint64_t bi = carry_i;
    uint32_t bc = carry_c;
    double tot_s = 0.0, tot_q = 0.0;
    uint64_t tot_i = 0;
    uint32_t tot_c = 0;
    for (int ww = 0; ww < RK_THREADS / 64; ++ww) {
      if (ww < wave) { bs += s.wave_s[ww]; bq += s.wave_q[ww]; bc += s.wave_c[ww]; bi += s.wave_i[ww]; }
      tot_s += s.wave_s[ww]; tot_q += s.wave_q[ww]; tot_c += s.wave_c[ww]; tot_i += s.wave_i[ww];
    }
    s.ps[e] = bs + is;
    s.pq[e] = bq + iq;
    s.pc[e] = bc + ic;
    if (kInt) s.pi[e] = bi + ii;
    carry_s += tot_s; carry_q += tot_q; carry_c += tot_c; carry_i += tot_i;
    __syncthreads();
  }

  // ---- sparse table over 16-row sub-blocks (NULLs hold the identity of min / max)
  for (int b = tid; b < RK_NSB; b += RK_THREADS) {
    T mn = Ord<T>::min_id(), mx = Ord<T>::max_id();
    bool any = false;
    for (int k = 0; k < RK_SB; ++k) {
      int e = b * RK_SB + k;
      uint32_t c = s.pc[e] & 0xffffu, cp = e ? (s.pc[e - 1] & 0xffffu) : 0u;
      if (c != cp) {
        T v = s.x[e];
        if (!any) { mn = v; mx = v; any = true; }
        else {
          if (Ord<T>::lt(v, mn)) mn = v;
          if (Ord<T>::lt(mx, v)) mx = v;
        }
      }
    }
    s.mn[0][b] = mn;
    s.mx[0][b] = mx;
  }
  __syncthreads();
  for (int lv = 1; lv < RK_LEVELS; ++lv) {
    int span = 1 << lv;
    for (int b = tid; b + span <= RK_NSB; b += RK_THREADS) {
      T a0 = s.mn[lv - 1][b], a1 = s.mn[lv - 1][b + span / 2];
      s.mn[lv][b] = Ord<T>::lt(a1, a0) ? a1 : a0;
      T b0 = s.mx[lv - 1][b], b1 = s.mx[lv - 1][b + span / 2];
      s.mx[lv][b] = Ord<T>::lt(b0, b1) ? b1 : b0;
    }
    __syncthreads();
  }

  // ---- per-row frames
  for (int64_t i = t0 + tid; i < t1; i += RK_THREADS) {
    uint64_t cw = comp[i];
    uint64_t k = cw >> low;
    int64_t lo = lower_bound_k(comp, e0, i, frame_start_key(k, pmask, w), low);
    int64_t hi = upper_bound_k(comp, i, e1 + 1, k, low) - 1;
    o.perm[i] = (int32_t)perm[i];
    int a = (int)(lo - e0), b = (int)(hi - e0);
    uint32_t c1 = s.pc[b], c0 = a ? s.pc[a - 1] : 0u;
    uint32_t dc = c1 - c0;
    if (dc >> 16) {  // a non-finite value in the frame: exact sequential evaluation
      rs_row_serial<T>(comp, xs, lo, hi, i, o);
      continue;
    }
    int64_t cnt = dc & 0xffffu;
    double sd = s.ps[b] - (a ? s.ps[a - 1] : 0.0);
    double sq = s.pq[b] - (a ? s.pq[a - 1] : 0.0);
    // min / max over [a, b]: partial sub-blocks serially, whole sub-blocks from the table.
    // NULL slots of s.x are skipped through the valid-count prefix.
    T mn = Ord<T>::min_id(), mx = Ord<T>::max_id();
    bool any = false;
    int sa = a / RK_SB, sb = b / RK_SB;
    int head_end = sa == sb ? b : (sa + 1) * RK_SB - 1;
    for (int e = a; e <= head_end; ++e) {
      uint32_t c = s.pc[e] & 0xffffu, cp = e ? (s.pc[e - 1] & 0xffffu) : 0u;
      if (c != cp) {
        T v = s.x[e];
        if (!any) { mn = v; mx = v; any = true; }
        else {
          if (Ord<T>::lt(v, mn)) mn = v;
          if (Ord<T>::lt(mx, v)) mx = v;
        }
      }
    }
    if (sa != sb) {
      for (int e = sb * RK_SB; e <= b; ++e) {
        uint32_t c = s.pc[e] & 0xffffu, cp = e ? (s.pc[e - 1] & 0xffffu) : 0u;
        if (c != cp) {
          T v = s.x[e];
          if (!any) { mn = v; mx = v; any = true; }
          else {
            if (Ord<T>::lt(v, mn)) mn = v;
            if (Ord<T>::lt(mx, v)) mx = v;
          }
        }
      }
      int m0 = sa + 1, m1 = sb - 1;
      if (m0 <= m1) {
        // whole sub-blocks: valid counts tell whether they hold any value at all
        uint32_t cm = (s.pc[(m1 + 1) * RK_SB - 1] & 0xffffu) - (s.pc[m0 * RK_SB - 1] & 0xffffu);
        if (cm) {
          int lv = 31 - __clz(m1 - m0 + 1);
          T a0 = s.mn[lv][m0], a1 = s.mn[lv][m1 + 1 - (1 << lv)];
          T tmn = Ord<T>::lt(a1, a0) ? a1 : a0;
          T b0 = s.mx[lv][m0], b1 = s.mx[lv][m1 + 1 - (1 << lv)];
          T tmx = Ord<T>::lt(b0, b1) ? b1 : b0;
          if (!any) { mn = tmn; mx = tmx; any = true; }
          else {
            if (Ord<T>::lt(tmn, mn)) mn = tmn;
            if (Ord<T>::lt(mx, tmx)) mx = tmx;
          }
        }
      }
    }
    double nd = (double)cnt;
    double mean = cnt ? pivot + sd / nd : 0.0;
    T sum;
    if constexpr (kInt) sum = (T)(s.pi[b] - (a ? s.pi[a - 1] : 0ull));
    else sum = (T)(nd * pivot + sd);
    double stddev = 0.0;
    if (cnt > 1) {
      bool same = !Ord<T>::lt(mn, mx) && !Ord<T>::lt(mx, mn);  // all values identical: m2 is exactly 0
      double m2 = same ? 0.0 : sq - sd * sd / nd;
      // The prefix difference carries an absolute error of a few ulps of the running total.
      // When the frame's spread is tiny against that total, redo it as a corrected two-pass
      // sum around the frame mean (rare: near-constant frames inside a varying extent).
      if (!same && m2 < 1e-5 * s.pq[b]) {
        double acc = 0.0, lin = 0.0;
        for (int e = a; e <= b; ++e) {
          uint32_t c = s.pc[e] & 0xffffu, cp = e ? (s.pc[e - 1] & 0xffffu) : 0u;
          if (c != cp) {
            double dv = (double)s.x[e] - mean;
            acc += dv * dv;
            lin += dv;
          }
        }
        m2 = acc - lin * lin / nd;
      }
      if (m2 < 0.0) m2 = 0.0;
      stddev = sqrt(m2 / (nd - 1.0));
    }
    rs_store<T>(o, i, cnt, mean, mn, mx, sum, cnt > 1, stddev, (cw & 1ull) != 0, (double)s.x[i - e0]);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_sorted_kernel(const T *x, const uint32_t *perm,
                                                               int64_t n, T *out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = x[perm[i]];
}

struct RsPlan {
  OrderBuffers ob;
  uint64_t *xs;
};

static void rs_plan(const tmp_range_stats_args *a, Workspace &ws, RsPlan *p) {
  order_plan(a->n, ws, &p->ob);
  p->xs = ws.take<uint64_t>((size_t)(a->n > 0 ? a->n : 1));
}

template <typename T>
static int rs_run(const tmp_range_stats_args *a, RsPlan &p, hipStream_t st) {
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts;
  s.prim_div = a->ts_is_seconds ? 1 : 1000000;
  s.n_masks = 1;
  s.masks[0] = a->x_valid;  // NULL pointer = every row valid
  TMP_TRY(order_scan_ranges(s, p.ob, st));
  RangeInfo r;
  TMP_TRY(order_read_ranges(p.ob, &r, st));
  Layout lay;
  TMP_TRY(order_make_layout(s, r, 1, &lay));
  OrderResult res;
  bool sorted = r.unsorted == 0;
  TMP_TRY(order_build(s, lay, sorted, p.ob, &res, st));
  const T *xs = (const T *)a->x;
  if (!sorted) {
    TMP_LAUNCH(gather_sorted_kernel<T>, dim3(grid_for(a->n, kBlock * 4, 16384)), dim3(kBlock), st, (const T *)a->x, res.perm, a->n, (T *)p.xs);
    xs = (const T *)p.xs;
  }
  RsOut<T> o;
  o.perm = a->out_perm; o.mean = a->out_mean; o.count = a->out_count;
  o.mn = (T *)a->out_min; o.mx = (T *)a->out_max; o.sum = (T *)a->out_sum;
  o.stddev = a->out_stddev; o.zscore = a->out_zscore; o.valid = a->out_valid;
  int grid = (int)((a->n + RK_ROWS - 1) / RK_ROWS);
  TMP_LAUNCH(range_stats_kernel<T>, dim3(grid), dim3(RK_THREADS), st, res.comp, res.perm,
                     xs, a->n, lay, (uint64_t)a->range_back_secs, o);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_range_stats_workspace_bytes(const tmp_range_stats_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RsPlan p;
  rs_plan(a, ws, &p);
  return ws.off;
}

extern "C" int tmp_range_stats(const tmp_range_stats_args *a, void *workspace,
                               size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_range_stats: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_range_stats: row count must be in [0, 2^31)");
  TMP_REQUIRE(a->range_back_secs >= 0, TMP_EINVAL, "tmp_range_stats: range_back_secs must be >= 0");
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->x && a->out_perm && a->out_mean && a->out_count && a->out_min && a->out_max &&
                  a->out_sum && a->out_stddev && a->out_zscore && a->out_valid,
              TMP_EINVAL, "tmp_range_stats: null pointer");
  Workspace ws(workspace, workspace_bytes);
  RsPlan p;
  rs_plan(a, ws, &p);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_range_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  return a->x_is_i64 ? rs_run<int64_t>(a, p, st) : rs_run<double>(a, p, st);
}
