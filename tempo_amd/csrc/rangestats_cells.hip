// withRangeStats over (key, second) cells: rolling RANGE-frame statistics of a feed that arrives
// in time order, without ordering the rows (python/tempo/tsdf.py:673-721; frame :563-572, :688:
// rows of the same key whose floor-second lies in [sec - W, sec]).
//
// A frame is a run of W + 1 whole seconds of one key, so it is determined by per-second
// aggregates: count, sum, minimum, maximum and (count, mean, M2) of the second's values -- a table
// of (keys) x (seconds between the first and the last timestamp) cells, small beside the rows.
//   1. two sweeps over the rows as they lie fill the cells (dense_cells.h: counts / sums /
//      extremes, then the squared deviations from each cell's own mean);
//   2. per key the seconds are cut into blocks of W + 1; inside a block the cells are combined
//      from the block's start up to each second ("pre") and from each second to the block's end
//      ("suf").  Counts and sums add, extremes compare, (count, mean, M2) merge by the pairwise
//      update Spark's CentralMomentAgg uses for partial aggregates.  A frame [s - W, s] is either
//      one whole block (pre[s]) or the tail of one block and the head of the next
//      (suf[s - W] + pre[s]): nothing is ever subtracted, so no cancellation, and a combination
//      never spans more than 2 W + 1 seconds;
//   3. a last sweep reads every row's key, second and value, looks its frame up and writes the
//      statistics where the row lies.
// Not taken (the caller then orders the rows): integer metrics, NaN or infinite values (their
// propagation through Spark's recurrences is reproduced by the ordered path only), feeds in no
// time order, too many cells.
#include <string.h>

#include "common.h"
#include "dense_cells.h"
#include "rangestats_common.h"

namespace tmp {

constexpr int64_t kCellMax = 1ll << 24;   // cells the path takes
constexpr int64_t kCellKeys = 4096;
constexpr int64_t kCellMinRows = 1ll << 22;

// Combination of cells: what a run of seconds of one key holds.
// The mean is kept in two parts, mean + lo: `mean` is a value of the first cell that went in (never
// changed by a merge), `lo` the small remainder.  Two runs' means then differ by (difference of two
// nearby doubles, exact) + (difference of two small numbers): the squared difference that enters M2
// is not limited by the rounding of a mean at the level of the values themselves.
struct RunAgg {
  uint32_t n;
  double sum, mean, lo, m2;
  uint64_t mn, mx;  // coded (code_f64); all ones / zero when n == 0
};
__device__ __forceinline__ RunAgg run_empty() {
  RunAgg a;
  a.n = 0; a.sum = 0.0; a.mean = 0.0; a.lo = 0.0; a.m2 = 0.0; a.mn = ~0ull; a.mx = 0ull;
  return a;
}
__device__ __forceinline__ RunAgg run_merge(const RunAgg &a, const RunAgg &b) {
  if (a.n == 0) return b;
  if (b.n == 0) return a;
  RunAgg r;
  r.n = a.n + b.n;
  r.sum = a.sum + b.sum;
  const double na = (double)a.n, nb = (double)b.n, nn = (double)r.n;
  const double delta = (b.mean - a.mean) + (b.lo - a.lo);
  r.mean = a.mean;
  r.lo = a.lo + delta * nb / nn;
  r.m2 = a.m2 + b.m2 + delta * delta * na * nb / nn;
  r.mn = a.mn < b.mn ? a.mn : b.mn;
  r.mx = a.mx > b.mx ? a.mx : b.mx;
  return r;
}

// One combination, one 64-byte line: a row's lookups touch two lines, not a dozen arrays.
struct __attribute__((aligned(64))) RunRec {
  double sum, mean, m2;
  uint64_t mn, mx;
  uint32_t n;
  uint32_t pad_[1];
  double lo;
  uint32_t pad2_[2];
};
static_assert(sizeof(RunRec) == 64, "one combination, one line");
typedef RunRec *RunTab;
__device__ __forceinline__ void run_put(RunTab t, int64_t i, const RunAgg &a) {
  RunRec r;
  r.sum = a.sum; r.mean = a.mean; r.m2 = a.m2; r.mn = a.mn; r.mx = a.mx; r.n = a.n; r.lo = a.lo;
  r.pad_[0] = r.pad2_[0] = r.pad2_[1] = 0;
  t[i] = r;
}
__device__ __forceinline__ RunAgg run_get(const RunRec *t, int64_t i) {
  const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(t + i);
  const ulonglong2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  RunAgg a;
  a.sum = __longlong_as_double((long long)q0.x);
  a.mean = __longlong_as_double((long long)q0.y);
  a.m2 = __longlong_as_double((long long)q1.x);
  a.mn = q1.y;
  a.mx = q2.x;
  a.n = (uint32_t)q2.y;
  a.lo = __longlong_as_double((long long)q3.x);
  return a;
}

// One thread per (key, block of `blk` seconds): pre and suf of the block's cells.
__global__ __launch_bounds__(kBlock) void cell_runs_kernel(StatTab tab, int64_t per_key, int64_t blk,
                                                           int64_t blocks_per_key, int64_t n_keys, RunTab pre,
                                                           RunTab suf) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= blocks_per_key * n_keys) return;
  const int64_t k = j / blocks_per_key, b = j - k * blocks_per_key;
  const int64_t s0 = b * blk, s1 = s0 + blk < per_key ? s0 + blk : per_key;
  const int64_t base = k * per_key;
  auto cell = [&](int64_t s) {
    const int64_t i = base + s;
    RunAgg a = run_empty();
    a.n = tab.cnt[i];
    if (a.n) {
      a.sum = tab.total[i];
      a.mean = tab.sum[i];  // the mean by now (rounded at the level of the values) ...
      a.lo = tab.lin[i] / (double)a.n;  // ... and what the second sweep found it to be off by
      const double m2 = tab.acc[i] - tab.lin[i] * tab.lin[i] / (double)a.n;
      a.m2 = m2 < 0.0 ? 0.0 : m2;
      a.mn = tab.mn[i];
      a.mx = tab.mx[i];
    }
    return a;
  };
  RunAgg run = run_empty();
  for (int64_t s = s0; s < s1; ++s) {
    run = run_merge(run, cell(s));
    run_put(pre, base + s, run);
  }
  run = run_empty();
  for (int64_t s = s1 - 1; s >= s0; --s) {
    run = run_merge(cell(s), run);
    run_put(suf, base + s, run);
  }
}

// Every row's frame and statistics, written where the row lies.  Four rows per thread: their
// inputs, then their records, are requested together before anything is computed from them.
__global__ __launch_bounds__(kBlock) void cell_rows_kernel(const int32_t *key, const int64_t *ts, const double *x,
                                                           const uint8_t *x_valid, int64_t n, int64_t sec_div,
                                                           int64_t smin, int64_t per_key, int64_t blk, int64_t w,
                                                           int64_t n_keys, RunTab pre, RunTab suf, RsOut<double> o) {
  constexpr int U = 4;
  const int64_t step = (int64_t)gridDim.x * blockDim.x * U;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n; i0 += step) {
    int64_t t[U], at[U];
    int32_t k[U];
    double xv[U];
    bool x_ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      at[u] = i < n ? i : n - 1;
      t[u] = ts[at[u]];
      k[u] = key ? key[at[u]] : 0;
      if ((uint32_t)k[u] >= (uint32_t)n_keys) k[u] = (int32_t)n_keys - 1;  // outside the dictionary: read as the last key
      xv[u] = x[at[u]];
      x_ok[u] = !x_valid || x_valid[at[u]];
    }
    int64_t cs[U], cp[U];
    bool two[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = (sec_div == 1 ? t[u] : floordiv_i64(t[u], sec_div)) - smin;
      const int64_t p = s - w > 0 ? s - w : 0;
      const int64_t base = (int64_t)k[u] * per_key;
      cs[u] = base + s;
      cp[u] = base + p;
      // the frame starts in the block before the row's: that block's tail joins in.  (A frame is
      // blk seconds long, so it lies in one block exactly when it starts at a multiple of blk.)
      two[u] = p > 0 && p - floordiv_i64(p, blk) * blk != 0;
    }
    RunAgg a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = run_get(pre, cs[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = run_get(suf, two[u] ? cp[u] : cs[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i >= n) continue;
      const RunAgg f = two[u] ? run_merge(b[u], a[u]) : a[u];
      const int64_t cnt = f.n;
      const double mean = cnt ? f.sum / (double)cnt : 0.0;
      const bool sd_ok = cnt > 1;
      const double sd = sd_ok ? sqrt(f.m2 / (double)(cnt - 1)) : 0.0;
      o.perm[i] = (int32_t)i;
      rs_store<double>(o, i, cnt, mean, __longlong_as_double((long long)decode_f64(f.mn)),
                       __longlong_as_double((long long)decode_f64(f.mx)), f.sum, sd_ok, sd, x_ok[u], xv[u]);
    }
  }
}

void rs_cells_plan(const tmp_range_stats_args *a, Workspace &ws, RsCellsPlan *p) {
  memset(p, 0, sizeof(*p));
  if (!a->any_order || a->x_is_i64 || a->n < (getenv("TEMPO_DENSE_MIN_ROWS") ? dense_min_rows() : kCellMinRows) ||
      plan_keys(a->key, a->n_keys) > kCellKeys || getenv("TEMPO_NO_RANGE_CELLS"))
    return;
  // The tables are sized before the table's time range is known (the caller brings the workspace):
  // for every cell the path may take, about 184 bytes each -- 2.9 GiB at kCellMax.  A call that sets
  // any_order on 2^22 rows or more asks for that much even if it ends on the ordered path.
  const int64_t bound = kCellMax;
  const size_t c = (size_t)bound;
  p->cap = bound;
  p->planned = 1;
  p->sum = ws.take<double>(c); p->total = ws.take<double>(c); p->acc = ws.take<double>(c); p->lin = ws.take<double>(c);
  p->mn = ws.take<uint64_t>(c); p->mx = ws.take<uint64_t>(c);
  for (int d = 0; d < 2; ++d) p->runs[d] = ws.take<RunRec>(c);
  p->rows = ws.take<uint32_t>(c); p->cnt = ws.take<uint32_t>(c);
  p->nonfinite = ws.take<int32_t>(1);
}

int rs_cells_try(const tmp_range_stats_args *a, RsCellsPlan &p, const RangeInfo &r, hipStream_t st, bool *done) {
  *done = false;
  if (!p.planned || sorted_mode(r) == 0) return TMP_OK;
  const int64_t nk = plan_keys(a->key, a->n_keys);
  const uint64_t per_key = (uint64_t)r.prim_max - (uint64_t)r.prim_min + 1;  // seconds first .. last
  if (per_key < 1 || per_key > (uint64_t)p.cap || (uint64_t)nk * per_key > (uint64_t)p.cap) return TMP_OK;
  StatTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.cells = (int64_t)((uint64_t)nk * per_key);
  tab.rows = p.rows; tab.cnt = p.cnt; tab.sum = p.sum; tab.total = p.total; tab.acc = p.acc; tab.lin = p.lin;
  tab.mn = p.mn; tab.mx = p.mx; tab.isum = nullptr; tab.nonfinite = p.nonfinite;
  const size_t c = (size_t)tab.cells;
  TMP_CHECK_HIP(hipMemsetAsync(tab.rows, 0, sizeof(uint32_t) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.cnt, 0, sizeof(uint32_t) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.sum, 0, sizeof(double) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.acc, 0, sizeof(double) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.lin, 0, sizeof(double) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.mx, 0, sizeof(uint64_t) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.mn, 0xff, sizeof(uint64_t) * c, st));
  TMP_CHECK_HIP(hipMemsetAsync(tab.nonfinite, 0, sizeof(int32_t), st));
  ColPtrs cp;
  memset(&cp, 0, sizeof(cp));
  cp.col[0] = a->x;
  cp.valid[0] = a->x_valid;
  const int64_t sec_div = a->ts_is_seconds ? 1 : 1000000;
  // short stretches of rows per block (one step): a second of a feed touches every key once, and a
  // block's table of DS_SLOTS cells should hold its stretch's cells -- also where the table is a
  // few time-ordered runs end to end, each of them that much thinner
  const int grid = grid_for(a->n, DH_THREADS * DH_ROWS, 1 << 20);
  TMP_LAUNCH(dense_stats_kernel<0>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, sec_div, r.prim_min,
             (uint32_t)per_key, 1, cp, tab);
  int32_t nonfinite = 0;
  TMP_CHECK_HIP(hipMemcpyAsync(&nonfinite, tab.nonfinite, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  TMP_CHECK_HIP(hipStreamSynchronize(st));
  if (nonfinite) return TMP_OK;  // NaN / inf: the ordered path reproduces Spark's recurrences
  TMP_LAUNCH(dense_stats_means_kernel, dim3(grid_for(tab.cells, kBlock, 4096)), dim3(kBlock), st, tab, 1);
  TMP_LAUNCH(dense_stats_kernel<1>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, sec_div, r.prim_min,
             (uint32_t)per_key, 1, cp, tab);
  RunTab pre = (RunRec *)p.runs[0], suf = (RunRec *)p.runs[1];
  const int64_t w = a->range_back_secs;
  const int64_t blk = (uint64_t)w + 1 < per_key ? w + 1 : (int64_t)per_key;  // W + 1 seconds, or the whole key
  const int64_t blocks_per_key = ((int64_t)per_key + blk - 1) / blk;
  TMP_LAUNCH(cell_runs_kernel, dim3((unsigned)((blocks_per_key * nk + kBlock - 1) / kBlock)), dim3(kBlock), st, tab,
             (int64_t)per_key, blk, blocks_per_key, nk, pre, suf);
  RsOut<double> o;
  o.perm = a->out_perm; o.mean = a->out_mean; o.count = a->out_count;
  o.mn = (double *)a->out_min; o.mx = (double *)a->out_max; o.sum = (double *)a->out_sum;
  o.stddev = a->out_stddev; o.zscore = a->out_zscore; o.valid = a->out_valid;
  o.vbit = 0;
  TMP_LAUNCH(cell_rows_kernel, dim3(grid_for(a->n, kBlock * 4, 1 << 20)), dim3(kBlock), st, a->key, a->ts,
             (const double *)a->x, a->x_valid, a->n, sec_div, r.prim_min, (int64_t)per_key, blk, w, nk, pre, suf, o);
  TMP_CHECK_HIP(hipGetLastError());
  *done = true;
  return TMP_OK;
}

}  // namespace tmp
