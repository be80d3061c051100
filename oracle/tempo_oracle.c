/*
 * tempo_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the
 * product path (tempo_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * A plain scalar C restatement of the reference's TSDF hot path.  The reference
 * (python/tempo/tsdf.py, python/tempo/resample.py) only builds Spark logical plans; the
 * arithmetic lives in Apache Spark (pyspark==3.2.1, python/requirements.txt:6), which is not
 * vendored.  Each function below therefore restates the reference plan line by line together
 * with the published Spark 3.2.1 semantics of the operators it calls (Window frames,
 * last(ignorenulls), count, CentralMomentAgg's Welford update, window() bucketing ...).
 *
 * Parity status: pinned by the golden vectors of the reference's own tests
 * (python/tests/tsdf_tests.py:164-394 asofJoin, :444-502 withRangeStats to 2 decimals,
 * :580-660 resample; scala EMATests.scala:19-43 for the EMA formula).  withLookbackFeatures
 * and the Python EMA have no reference test: parity unpinned for those two, and for the derived
 * columns of the table writer (tsdf_tests.py:746-787 counts rows only).
 *
 * Conventions: timestamps are int64 microseconds since the epoch (Spark TimestampType) unless
 * ts_is_seconds!=0 (an integral ts column); partition keys are dictionary ids (int32);
 * NULLs are byte masks (1 = valid), a NULL mask pointer means "no NULLs".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
typedef int32_t i32;
typedef uint8_t u8;

/* ------------------------------------------------------------------ helpers */

static i64 floordiv64(i64 a, i64 b) {
  i64 q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

/* Spark Cast double -> int (non-ANSI): NaN -> 0, saturating, truncation toward zero. */
static i32 spark_double_to_int(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (i32)(-2147483647 - 1);
  return (i32)x;
}

typedef int (*cmp_fn)(const void *ctx, i64 a, i64 b);

/* Stable bottom-up merge sort of an index array. */
static int stable_sort_idx(i64 *idx, i64 n, cmp_fn cmp, const void *ctx) {
  if (n < 2) return 0;
  i64 *tmp = (i64 *)malloc((size_t)n * sizeof(i64));
  if (!tmp) return -1;
  i64 *src = idx, *dst = tmp;
  for (i64 w = 1; w < n; w *= 2) {
    for (i64 lo = 0; lo < n; lo += 2 * w) {
      i64 mid = lo + w < n ? lo + w : n;
      i64 hi = lo + 2 * w < n ? lo + 2 * w : n;
      i64 a = lo, b = mid, o = lo;
      while (a < mid && b < hi) {
        if (cmp(ctx, src[b], src[a]) < 0) dst[o++] = src[b++];
        else dst[o++] = src[a++];
      }
      while (a < mid) dst[o++] = src[a++];
      while (b < hi) dst[o++] = src[b++];
    }
    i64 *t = src; src = dst; dst = t;
  }
  if (src != idx) memcpy(idx, src, (size_t)n * sizeof(i64));
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ asofJoin
 * Follows python/tempo/tsdf.py:463-560 (asofJoin), :104-109 (__combineTSDF),
 * :111-162 (__getLastRightRow), :164-190 (__getTimePartitions).
 */
typedef struct {
  i32 key;
  double part;     /* ts_partition (skew mode), else 0 */
  i64 ts;          /* combined_ts, tsdf.py:107 */
  int seq_null;    /* right sequence col is NULL on this row (always for left rows) */
  i64 seq;
  int rec_ind;     /* +1 left, -1 right, tsdf.py:546 */
  i64 src;         /* row number in its own table */
  int is_original; /* tsdf.py:181,187 */
} urow;

typedef struct {
  const urow *rows;
  int has_seq;
  int skew;
} uctx;

static int ucmp(const void *vctx, i64 a, i64 b) {
  const uctx *c = (const uctx *)vctx;
  const urow *x = &c->rows[a], *y = &c->rows[b];
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (c->skew && x->part != y->part) return x->part < y->part ? -1 : 1;
  /* orderBy(combined_ts, [right seq ASC NULLS FIRST], rec_ind)  tsdf.py:117-121 */
  if (x->ts != y->ts) return x->ts < y->ts ? -1 : 1;
  if (c->has_seq) {
    if (x->seq_null != y->seq_null) return x->seq_null ? -1 : 1;
    if (!x->seq_null && x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
  }
  if (x->rec_ind != y->rec_ind) return x->rec_ind < y->rec_ind ? -1 : 1;
  return 0;
}

/* ts_partition and partition_remainder exactly as tsdf.py:178-180 evaluates them in Spark:
 *   ts_col_double = ts.cast("double")            (micros / 1e6)
 *   ts_partition  = lit(v) * (ts_col_double / lit(v)).cast("integer")
 *   remainder     = (ts_col_double - ts_partition) / lit(v)
 * v_is_int selects Spark's int*int (32-bit wrapping) product, else double*int. */
static void time_partition(i64 ts_us, int ts_is_seconds, double v, int v_is_int, double *part,
                           double *rem) {
  double d = ts_is_seconds ? (double)ts_us : (double)ts_us / 1000000.0;
  i32 q = spark_double_to_int(d / v);
  double p;
  if (v_is_int) p = (double)(i32)((uint32_t)(i32)v * (uint32_t)q);
  else p = v * (double)q;
  *part = p;
  *rem = (d - p) / v;
}

static double part_plus_v(double p, double v, int v_is_int) {
  if (v_is_int) return (double)(i32)((uint32_t)(i32)p + (uint32_t)(i32)v);
  return p + v;
}

/*
 * out_idx[c*nL + i] = index of the right row whose column c is joined onto left row i, or -1.
 * any_missing[c]    = 1 if some left row got no value for column c (tsdf.py:142-158).
 * valid_r[c] may be NULL (column has no NULLs).  seq_r may be NULL (no sequence column).
 */
int orc_asof_join(i64 nL, const i32 *key_l, const i64 *ts_l, i64 nR, const i32 *key_r,
                  const i64 *ts_r, const i64 *seq_r, const u8 *seq_r_valid, int n_cols,
                  const u8 *const *valid_r, int skip_nulls, int ts_is_seconds, int skew,
                  double ts_partition_val, int v_is_int, double one_minus_fraction,
                  i64 *out_idx, i32 *any_missing) {
  i64 cap = (nL + nR) * (skew ? 2 : 1);
  urow *rows = (urow *)malloc((size_t)(cap ? cap : 1) * sizeof(urow));
  i64 *idx = (i64 *)malloc((size_t)(cap ? cap : 1) * sizeof(i64));
  i64 *last = (i64 *)malloc((size_t)(n_cols ? n_cols : 1) * sizeof(i64));
  if (!rows || !idx || !last) { free(rows); free(idx); free(last); return -1; }
  i64 n = 0;
  /* union: left rows then right rows (tsdf.py:542-545) */
  for (int side = 0; side < 2; ++side) {
    i64 cnt = side == 0 ? nL : nR;
    for (i64 i = 0; i < cnt; ++i) {
      urow r;
      r.key = side == 0 ? key_l[i] : key_r[i];
      r.ts = side == 0 ? ts_l[i] : ts_r[i];
      r.rec_ind = side == 0 ? 1 : -1;
      r.src = i;
      r.seq_null = 1;
      r.seq = 0;
      if (side == 1 && seq_r && (!seq_r_valid || seq_r_valid[i])) { r.seq_null = 0; r.seq = seq_r[i]; }
      r.part = 0.0;
      r.is_original = 1;
      if (skew) {
        double p, rem;
        time_partition(r.ts, ts_is_seconds, ts_partition_val, v_is_int, &p, &rem);
        r.part = p;
        rows[n++] = r;
        if (rem >= one_minus_fraction) { /* tsdf.py:184-187 */
          urow d = r;
          d.part = part_plus_v(p, ts_partition_val, v_is_int);
          d.is_original = 0;
          rows[n++] = d;
        }
      } else {
        rows[n++] = r;
      }
    }
  }
  for (i64 i = 0; i < n; ++i) idx[i] = i;
  uctx ctx = {rows, seq_r != NULL, skew};
  if (stable_sort_idx(idx, n, ucmp, &ctx)) { free(rows); free(idx); free(last); return -1; }
  for (int c = 0; c < n_cols; ++c) any_missing[c] = 0;
  i64 last_row = -1;
  for (i64 p = 0; p < n; ++p) {
    const urow *r = &rows[idx[p]];
    int new_part = (p == 0);
    if (!new_part) {
      const urow *q = &rows[idx[p - 1]];
      new_part = (q->key != r->key) || (skew && q->part != r->part);
    }
    if (new_part) {
      for (int c = 0; c < n_cols; ++c) last[c] = -1;
      last_row = -1;
    }
    if (r->rec_ind == -1) {
      last_row = r->src;
      for (int c = 0; c < n_cols; ++c)
        if (!valid_r || !valid_r[c] || valid_r[c][r->src]) last[c] = r->src; /* last(col, True) :139 */
    } else if (r->is_original) { /* filter(left_ts isNotNull) :147, is_original==1 :556 */
      for (int c = 0; c < n_cols; ++c) {
        i64 v = skip_nulls ? last[c] : last_row; /* :126-136 for skipNulls=False */
        out_idx[(i64)c * nL + r->src] = v;
        /* non_null_ct == 0  (:142-145, :155) */
        if (last[c] < 0) any_missing[c] = 1;
      }
    }
  }
  free(rows); free(idx); free(last);
  return 0;
}

/* ------------------------------------------------------------------ window ordering
 * __baseWindow (tsdf.py:563-572): partitionBy(partitionCols).orderBy(ts.cast("long")
 * [, seq.cast("long")]).  timestamp.cast("long") = floorDiv(micros, 1e6).
 */
typedef struct {
  const i32 *key;
  const i64 *sec;
  const i64 *seq;
  const u8 *seq_valid;
} wctx;

static int wcmp(const void *vctx, i64 a, i64 b) {
  const wctx *c = (const wctx *)vctx;
  if (c->key && c->key[a] != c->key[b]) return c->key[a] < c->key[b] ? -1 : 1;
  if (c->sec[a] != c->sec[b]) return c->sec[a] < c->sec[b] ? -1 : 1;
  if (c->seq) {
    int na = c->seq_valid && !c->seq_valid[a], nb = c->seq_valid && !c->seq_valid[b];
    if (na != nb) return na ? -1 : 1; /* ASC NULLS FIRST */
    if (!na && c->seq[a] != c->seq[b]) return c->seq[a] < c->seq[b] ? -1 : 1;
  }
  return 0;
}

/* perm[p] = input row at sorted position p; sec_out[p] = its order key.  Ties keep input order. */
int orc_window_order(i64 n, const i32 *key, const i64 *ts, int ts_is_seconds, const i64 *seq,
                     const u8 *seq_valid, i64 *perm, i64 *sec_sorted) {
  i64 *sec = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!sec) return -1;
  for (i64 i = 0; i < n; ++i) {
    sec[i] = ts_is_seconds ? ts[i] : floordiv64(ts[i], 1000000);
    perm[i] = i;
  }
  wctx ctx = {key, sec, seq, seq_valid};
  int rc = stable_sort_idx(perm, n, wcmp, &ctx);
  if (sec_sorted)
    for (i64 p = 0; p < n; ++p) sec_sorted[p] = sec[perm[p]];
  free(sec);
  return rc;
}

/* ------------------------------------------------------------------ withRangeStats
 * tsdf.py:673-721.  RANGE frame [sec - W, sec] (both inclusive, later same-second peers
 * included), aggregates re-evaluated per row in sort order the way Spark's
 * SlidingWindowFunctionFrame does: Sum/Average = sequential fp64 add, count of non-NULL,
 * Min/Max with NaN greatest, stddev_samp = CentralMomentAgg Welford update, NULL for n < 2;
 * zscore = (x - mean) / stddev with NULL on a zero divisor (:716-717).
 * Outputs are in sorted order; perm gives the input row of each output row.
 */
static int spark_lt(double a, double b) { /* a < b with NaN greatest, -0.0 == 0.0 */
  if (a != a) return 0;
  if (b != b) return 1;
  return a < b;
}

int orc_range_stats_f64(i64 n, const i32 *key, const i64 *ts, int ts_is_seconds, const double *x,
                        const u8 *x_valid, i64 range_back_secs, i64 *perm, double *mean,
                        i64 *count, double *mn, double *mx, double *sum, double *stddev,
                        double *zscore, u8 *stat_valid, u8 *stddev_valid, u8 *zscore_valid) {
  i64 *sec = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!sec) return -1;
  if (orc_window_order(n, key, ts, ts_is_seconds, NULL, NULL, perm, sec)) { free(sec); return -1; }
  i64 lo = 0, hi = 0;
  for (i64 p = 0; p < n; ++p) {
    i32 k = key ? key[perm[p]] : 0;
    if (p == 0 || (key && key[perm[p - 1]] != k)) { lo = p; hi = p; }
    while (sec[lo] < sec[p] - range_back_secs) ++lo;
    if (hi < p) hi = p;
    while (hi + 1 < n && (!key || key[perm[hi + 1]] == k) && sec[hi + 1] <= sec[p]) ++hi;
    i64 cnt = 0;
    double s = 0.0, a_min = 0.0, a_max = 0.0, avg = 0.0, m2 = 0.0, nn = 0.0;
    for (i64 j = lo; j <= hi; ++j) {
      i64 r = perm[j];
      if (x_valid && !x_valid[r]) continue;
      double v = x[r];
      if (cnt == 0) { a_min = v; a_max = v; }
      else {
        if (spark_lt(v, a_min)) a_min = v;
        if (spark_lt(a_max, v)) a_max = v;
      }
      ++cnt;
      s += v;
      nn += 1.0;
      double delta = v - avg;
      double delta_n = delta / nn;
      avg += delta_n;
      m2 += delta * (delta - delta_n);
    }
    count[p] = cnt;
    stat_valid[p] = cnt > 0;
    mean[p] = cnt ? s / (double)cnt : 0.0;
    sum[p] = cnt ? s : 0.0;
    mn[p] = cnt ? a_min : 0.0;
    mx[p] = cnt ? a_max : 0.0;
    stddev_valid[p] = cnt > 1;
    stddev[p] = cnt > 1 ? sqrt(m2 / (nn - 1.0)) : 0.0;
    i64 r = perm[p];
    int xv = !x_valid || x_valid[r];
    int zv = xv && cnt > 1 && stddev[p] != 0.0; /* Spark Divide: NULL on zero divisor */
    zscore_valid[p] = (u8)zv;
    zscore[p] = zv ? (x[r] - mean[p]) / stddev[p] : 0.0;
  }
  free(sec);
  return 0;
}

/* Same for int/bigint metric columns: Sum accumulates in int64 (wrapping), Average and
 * stddev cast each value to double, Min/Max stay integral. */
int orc_range_stats_i64(i64 n, const i32 *key, const i64 *ts, int ts_is_seconds, const i64 *x,
                        const u8 *x_valid, i64 range_back_secs, i64 *perm, double *mean,
                        i64 *count, i64 *mn, i64 *mx, i64 *sum, double *stddev, double *zscore,
                        u8 *stat_valid, u8 *stddev_valid, u8 *zscore_valid) {
  i64 *sec = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!sec) return -1;
  if (orc_window_order(n, key, ts, ts_is_seconds, NULL, NULL, perm, sec)) { free(sec); return -1; }
  i64 lo = 0, hi = 0;
  for (i64 p = 0; p < n; ++p) {
    i32 k = key ? key[perm[p]] : 0;
    if (p == 0 || (key && key[perm[p - 1]] != k)) { lo = p; hi = p; }
    while (sec[lo] < sec[p] - range_back_secs) ++lo;
    if (hi < p) hi = p;
    while (hi + 1 < n && (!key || key[perm[hi + 1]] == k) && sec[hi + 1] <= sec[p]) ++hi;
    i64 cnt = 0, a_min = 0, a_max = 0;
    uint64_t s = 0;
    double ds = 0.0, avg = 0.0, m2 = 0.0, nn = 0.0;
    for (i64 j = lo; j <= hi; ++j) {
      i64 r = perm[j];
      if (x_valid && !x_valid[r]) continue;
      i64 v = x[r];
      if (cnt == 0) { a_min = v; a_max = v; }
      else { if (v < a_min) a_min = v; if (v > a_max) a_max = v; }
      ++cnt;
      s += (uint64_t)v;
      ds += (double)v;
      nn += 1.0;
      double delta = (double)v - avg;
      double delta_n = delta / nn;
      avg += delta_n;
      m2 += delta * (delta - delta_n);
    }
    count[p] = cnt;
    stat_valid[p] = cnt > 0;
    mean[p] = cnt ? ds / (double)cnt : 0.0;
    sum[p] = cnt ? (i64)s : 0;
    mn[p] = cnt ? a_min : 0;
    mx[p] = cnt ? a_max : 0;
    stddev_valid[p] = cnt > 1;
    stddev[p] = cnt > 1 ? sqrt(m2 / (nn - 1.0)) : 0.0;
    i64 r = perm[p];
    int xv = !x_valid || x_valid[r];
    int zv = xv && cnt > 1 && stddev[p] != 0.0;
    zscore_valid[p] = (u8)zv;
    zscore[p] = zv ? ((double)x[r] - mean[p]) / stddev[p] : 0.0;
  }
  free(sec);
  return 0;
}

/* ------------------------------------------------------------------ EMA
 * tsdf.py:615-635.  weights[i] = exp_factor * (1 - exp_factor) ** i is evaluated by the caller
 * in Python float arithmetic exactly as tsdf.py:629 does.  The column starts as lit(0) and
 * adds, for i = 0, 1, ..., window-1, weight_i * lag(x, i) when that product is not NULL.
 */
int orc_ema_f64(i64 n, const i32 *key, const i64 *ts, int ts_is_seconds, const i64 *seq,
                const u8 *seq_valid, const double *x, const u8 *x_valid, int window,
                const double *weights, i64 *perm, double *ema) {
  if (orc_window_order(n, key, ts, ts_is_seconds, seq, seq_valid, perm, NULL)) return -1;
  i64 seg = 0;
  for (i64 p = 0; p < n; ++p) {
    if (p == 0 || (key && key[perm[p - 1]] != key[perm[p]])) seg = p;
    volatile double acc = 0.0; /* volatile: keep the left-to-right add order, no contraction */
    for (int i = 0; i < window; ++i) {
      i64 j = p - i;
      if (j < seg) break; /* lag() beyond the partition start is NULL -> contributes 0 */
      i64 r = perm[j];
      if (x_valid && !x_valid[r]) continue;
      volatile double t = weights[i] * x[r];
      acc = acc + t;
    }
    ema[p] = acc;
  }
  return 0;
}

/* ------------------------------------------------------------------ withLookbackFeatures
 * tsdf.py:637-671.  features[p] = rows p-L .. p-1 of the same series (oldest first), each an
 * array of the F feature columns (NULL elements kept).  Dense layout: feat[p][l][f] for
 * l < len[p]; entries l >= len[p] are zero / invalid.
 */
int orc_lookback_f64(i64 n, const i32 *key, const i64 *ts, int ts_is_seconds, const i64 *seq,
                     const u8 *seq_valid, int n_feat, const double *const *x,
                     const u8 *const *x_valid, int lookback, i64 *perm, i32 *len, double *feat,
                     u8 *feat_valid) {
  if (orc_window_order(n, key, ts, ts_is_seconds, seq, seq_valid, perm, NULL)) return -1;
  i64 seg = 0;
  for (i64 p = 0; p < n; ++p) {
    if (p == 0 || (key && key[perm[p - 1]] != key[perm[p]])) seg = p;
    i64 first = p - lookback;
    if (first < seg) first = seg;
    i32 m = (i32)(p - first);
    len[p] = m;
    for (int l = 0; l < lookback; ++l) {
      for (int f = 0; f < n_feat; ++f) {
        i64 o = (p * lookback + l) * n_feat + f;
        if (l < m) {
          i64 r = perm[first + l];
          int v = !x_valid || !x_valid[f] || x_valid[f][r];
          feat_valid[o] = (u8)v;
          feat[o] = v ? x[f][r] : 0.0;
        } else {
          feat_valid[o] = 0;
          feat[o] = 0.0;
        }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ resample
 * python/tempo/resample.py:25-117.  agg_key = window(ts, "<n> <unit>") : tumbling bucket with
 * start = ts - ((ts mod dur) + dur) mod dur  (epoch aligned); groupBy(partitionCols, agg_key).
 * func: 0 floor (min struct(ts, cols...)), 1 min, 2 max, 3 mean, 4 ceil (max struct).
 * Columns are either double (col_is_i64[c]==0) or int64 (==1; also used for order-preserving
 * string codes).  Output buckets are ordered by (key, start).
 *   out_key[b], out_start[b]
 *   func 1/2/3: out_f64[c*cap + b] (doubles; int64 min/max returned in out_i64), out_valid
 *   func 0/4  : out_row[b] = input row chosen by the struct ordering
 * Returns the number of buckets (<= n) or -1.
 */
static int cmp_cell(const void *col, int is_i64, const u8 *valid, i64 a, i64 b) {
  int va = !valid || valid[a], vb = !valid || valid[b];
  if (va != vb) return va ? 1 : -1; /* NULL smallest in struct ordering */
  if (!va) return 0;
  if (is_i64) {
    i64 x = ((const i64 *)col)[a], y = ((const i64 *)col)[b];
    return x < y ? -1 : (x > y ? 1 : 0);
  }
  double x = ((const double *)col)[a], y = ((const double *)col)[b];
  if (x == y) return 0;
  if (x != x) return (y != y) ? 0 : 1; /* NaN greatest */
  if (y != y) return -1;
  return x < y ? -1 : 1;
}

i64 orc_resample(i64 n, const i32 *key, const i64 *ts, i64 dur_us, int func, int n_cols,
                 const void *const *cols, const int *col_is_i64, const u8 *const *col_valid,
                 i32 *out_key, i64 *out_start, double *out_f64, i64 *out_i64, u8 *out_valid,
                 i64 *out_row) {
  i64 *perm = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  i64 *start = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!perm || !start) { free(perm); free(start); return -1; }
  for (i64 i = 0; i < n; ++i) {
    i64 m = ((ts[i] % dur_us) + dur_us) % dur_us;
    start[i] = ts[i] - m;
    perm[i] = i;
  }
  wctx ctx = {key, start, NULL, NULL};
  if (stable_sort_idx(perm, n, wcmp, &ctx)) { free(perm); free(start); return -1; }
  i64 nb = 0;
  i64 cap = n;
  i64 p = 0;
  while (p < n) {
    i64 q = p;
    i64 r0 = perm[p];
    while (q < n && (!key || key[perm[q]] == key[r0]) && start[perm[q]] == start[r0]) ++q;
    out_key[nb] = key ? key[r0] : 0;
    out_start[nb] = start[r0];
    if (func == 0 || func == 4) {
      i64 best = perm[p];
      for (i64 j = p + 1; j < q; ++j) {
        i64 r = perm[j];
        int c = ts[r] < ts[best] ? -1 : (ts[r] > ts[best] ? 1 : 0);
        for (int k = 0; c == 0 && k < n_cols; ++k)
          c = cmp_cell(cols[k], col_is_i64[k], col_valid ? col_valid[k] : NULL, r, best);
        if ((func == 0 && c < 0) || (func == 4 && c > 0)) best = r;
      }
      out_row[nb] = best;
    } else {
      for (int k = 0; k < n_cols; ++k) {
        const u8 *valid = col_valid ? col_valid[k] : NULL;
        i64 cnt = 0, best = -1;
        double s = 0.0;
        for (i64 j = p; j < q; ++j) {
          i64 r = perm[j];
          if (valid && !valid[r]) continue;
          ++cnt;
          s += col_is_i64[k] ? (double)((const i64 *)cols[k])[r] : ((const double *)cols[k])[r];
          if (best < 0) best = r;
          else {
            int c = cmp_cell(cols[k], col_is_i64[k], NULL, r, best);
            if ((func == 1 && c < 0) || (func == 2 && c > 0)) best = r;
          }
        }
        i64 o = (i64)k * cap + nb;
        out_valid[o] = cnt > 0;
        out_f64[o] = 0.0;
        out_i64[o] = 0;
        if (cnt > 0) {
          if (func == 3) out_f64[o] = s / (double)cnt;
          else if (col_is_i64[k]) out_i64[o] = ((const i64 *)cols[k])[best];
          else out_f64[o] = ((const double *)cols[k])[best];
        }
      }
    }
    ++nb;
    p = q;
  }
  free(perm); free(start);
  return nb;
}

/* ------------------------------------------------------------------ withGroupedStats
 * python/tempo/tsdf.py:723-759: groupBy(partitionCols, window(ts, "<n> <unit>")).agg(mean, count,
 * min, max, sum, stddev) per metric.  Same bucket as resample.  Per bucket, values in (key, ts)
 * order (stable): count of non-NULLs; Average = running double sum / count; Min / Max with NaN
 * greatest; Sum in the column's type (long sums wrap); stddev_samp by CentralMomentAgg's
 * update (n, avg, m2), NULL for fewer than two values.  Cells: out[c*cap + b].
 * Returns the number of buckets or -1.
 */
i64 orc_grouped_stats(i64 n, const i32 *key, const i64 *ts, i64 dur_us, int n_cols,
                      const void *const *cols, const int *col_is_i64, const u8 *const *col_valid,
                      i32 *out_key, i64 *out_start, double *out_mean, i64 *out_n, void *out_min,
                      void *out_max, void *out_sum, double *out_stddev) {
  i64 *perm = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  i64 *start = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!perm || !start) { free(perm); free(start); return -1; }
  for (i64 i = 0; i < n; ++i) {
    i64 m = ((ts[i] % dur_us) + dur_us) % dur_us;
    start[i] = ts[i] - m;
    perm[i] = i;
  }
  wctx ctx = {key, ts, NULL, NULL};
  if (stable_sort_idx(perm, n, wcmp, &ctx)) { free(perm); free(start); return -1; }
  i64 nb = 0, cap = n, p = 0;
  while (p < n) {
    i64 q = p, r0 = perm[p];
    while (q < n && (!key || key[perm[q]] == key[r0]) && start[perm[q]] == start[r0]) ++q;
    out_key[nb] = key ? key[r0] : 0;
    out_start[nb] = start[r0];
    for (int k = 0; k < n_cols; ++k) {
      const u8 *valid = col_valid ? col_valid[k] : NULL;
      i64 cnt = 0, best_lo = -1, best_hi = -1;
      double s = 0.0, avg = 0.0, m2 = 0.0, nn = 0.0;
      unsigned long long isum = 0;
      for (i64 j = p; j < q; ++j) {
        i64 r = perm[j];
        if (valid && !valid[r]) continue;
        double v = col_is_i64[k] ? (double)((const i64 *)cols[k])[r] : ((const double *)cols[k])[r];
        ++cnt;
        s += v;
        if (col_is_i64[k]) isum += (unsigned long long)((const i64 *)cols[k])[r];
        nn += 1.0;
        double delta = v - avg, delta_n = delta / nn;
        avg += delta_n;
        m2 += delta * (delta - delta_n);
        if (best_lo < 0) { best_lo = r; best_hi = r; }
        else {
          if (cmp_cell(cols[k], col_is_i64[k], NULL, r, best_lo) < 0) best_lo = r;
          if (cmp_cell(cols[k], col_is_i64[k], NULL, r, best_hi) > 0) best_hi = r;
        }
      }
      i64 o = (i64)k * cap + nb;
      out_n[o] = cnt;
      out_mean[o] = cnt ? s / (double)cnt : 0.0;
      out_stddev[o] = cnt > 1 ? sqrt(m2 / (nn - 1.0)) : 0.0;
      if (col_is_i64[k]) {
        ((i64 *)out_min)[o] = cnt ? ((const i64 *)cols[k])[best_lo] : 0;
        ((i64 *)out_max)[o] = cnt ? ((const i64 *)cols[k])[best_hi] : 0;
        ((i64 *)out_sum)[o] = cnt ? (i64)isum : 0;
      } else {
        ((double *)out_min)[o] = cnt ? ((const double *)cols[k])[best_lo] : 0.0;
        ((double *)out_max)[o] = cnt ? ((const double *)cols[k])[best_hi] : 0.0;
        ((double *)out_sum)[o] = cnt ? s : 0.0;
      }
    }
    ++nb;
    p = q;
  }
  free(perm); free(start);
  return nb;
}

/* ------------------------------------------------------------------ interpolate
 * python/tempo/interpol.py:260-378 on an already sampled table (one row per key and bucket),
 * restated column by column the way the reference builds it:
 *   previous_timestamp = ts, next_timestamp = lead(ts) or ts + freq          (:182-195, :316-321)
 *   per target c: ts_c = value NULL ? NULL : ts; previous_timestamp_c / next_timestamp_c =
 *   last / first non-NULL ts_c over rows <= / >= r; previous_c / next_null_c = last / first
 *   non-NULL value likewise; next_c = lead(value)                            (:197-258, :306-313)
 *   explode(sequence(ts, next_timestamp - freq, freq))                       (:331-336)
 *   is_ts_interpolated, is_interpolated_c and the fill rule per method       (:96-180, :66-94)
 * method: 0 zero, 1 null, 2 bfill, 3 ffill, 4 linear.  Output rows ordered by (key, ts).
 * Returns the number of output rows, -1 on allocation failure, -2 when a gap is shorter than
 * freq (Spark's sequence() would fail), -3 when cap is too small (nothing written).
 */
i64 orc_interpolate(i64 n, const i32 *key, const i64 *ts, i64 freq_us, int method, int n_cols,
                    const double *const *cols, const u8 *const *col_valid, i64 cap, i64 *out_src,
                    i32 *out_key, i64 *out_ts, u8 *out_ts_interp, double *out_val, u8 *out_valid,
                    u8 *out_interp) {
  i64 *perm = (i64 *)malloc((size_t)(n ? n : 1) * sizeof(i64));
  if (!perm) return -1;
  for (i64 i = 0; i < n; ++i) perm[i] = i;
  wctx ctx = {key, ts, NULL, NULL};
  if (stable_sort_idx(perm, n, wcmp, &ctx)) { free(perm); return -1; }
  /* size first */
  i64 total = 0;
  for (i64 p = 0; p < n; ++p) {
    i64 r = perm[p];
    int has_next = p + 1 < n && (!key || key[perm[p + 1]] == key[r]);
    i64 next_ts = has_next ? ts[perm[p + 1]] : ts[r] + freq_us;
    i64 stop = next_ts - freq_us;
    if (stop < ts[r]) { free(perm); return -2; }
    total += (stop - ts[r]) / freq_us + 1;
  }
  if (total > cap) { free(perm); return -3; }
  i64 q = 0;
  for (i64 p = 0; p < n; ++p) {
    i64 r = perm[p];
    int has_next = p + 1 < n && (!key || key[perm[p + 1]] == key[r]);
    i64 next_ts = has_next ? ts[perm[p + 1]] : ts[r] + freq_us;
    for (i64 t = ts[r]; t <= next_ts - freq_us; t += freq_us, ++q) {
      int ts_interp = t != ts[r];
      out_src[q] = r;
      out_key[q] = key ? key[r] : 0;
      out_ts[q] = t;
      out_ts_interp[q] = (u8)ts_interp;
      for (int c = 0; c < n_cols; ++c) {
        const u8 *valid = col_valid ? col_valid[c] : NULL;
        int have = !valid || valid[r];
        int interp = ts_interp || !have;
        /* window columns of row r */
        i64 pj = -1, nj = -1;
        for (i64 j = p; j >= 0 && (!key || key[perm[j]] == key[r]); --j)
          if (!valid || valid[perm[j]]) { pj = perm[j]; break; }
        for (i64 j = p; j < n && (!key || key[perm[j]] == key[r]); ++j)
          if (!valid || valid[perm[j]]) { nj = perm[j]; break; }
        int next_ok = has_next && (!valid || valid[perm[p + 1]]);
        double v = 0.0;
        int ok = 0;
        if (!interp) { v = cols[c][r]; ok = 1; }
        else if (method == 0) { v = 0.0; ok = 1; }
        else if (method == 1) { ok = 0; }
        else if (method == 3) { if (pj >= 0) { v = cols[c][pj]; ok = 1; } }
        else if (method == 2) {
          if (!next_ok && !have) { if (nj >= 0) { v = cols[c][nj]; ok = 1; } }
          else if (next_ok) { v = cols[c][perm[p + 1]]; ok = 1; }
        } else {
          if (!have) {
            if (pj >= 0 && nj >= 0) {
              i64 tp = floordiv64(ts[pj], 1000000), den = floordiv64(ts[nj], 1000000) - tp;
              if (den != 0) {
                v = (cols[c][nj] - cols[c][pj]) / (double)den * (double)(floordiv64(t, 1000000) - tp) + cols[c][pj];
                ok = 1;
              }
            }
          } else if (next_ok) {
            i64 tp = floordiv64(ts[r], 1000000), den = floordiv64(next_ts, 1000000) - tp;
            if (den != 0) {
              v = (cols[c][perm[p + 1]] - cols[c][r]) / (double)den * (double)(floordiv64(t, 1000000) - tp) + cols[c][r];
              ok = 1;
            }
          }
        }
        i64 at = (i64)c * cap + q;
        out_val[at] = ok ? v : 0.0;
        out_valid[at] = (u8)ok;
        out_interp[at] = (u8)interp;
      }
    }
  }
  free(perm);
  return total;
}

/* ------------------------------------------------------------------ table write layout
 * Follows python/tempo/io.py:29-39.  event_dt = to_date(ts) (UTC session).  event_time: the
 * reference casts ts to string -- Spark prints "yyyy-MM-dd HH:mm:ss" plus, when the
 * microseconds are not zero, "." and the six fraction digits with trailing zeros removed --
 * takes the part after the blank, deletes the colons (translate) and casts the rest to double.
 * The text is built and parsed here just like that.  The row order is the linear stand-in for
 * partitionBy("event_dt") + clustering by (partitionCols, event_time): stable sort by
 * (event_dt, key, ts).
 */
#include <stdio.h>

typedef struct {
  const i32 *dt;
  const i32 *key;
  const i64 *ts;
} wl_ctx;

static int wl_cmp(const void *vctx, i64 a, i64 b) {
  const wl_ctx *c = (const wl_ctx *)vctx;
  if (c->dt[a] != c->dt[b]) return c->dt[a] < c->dt[b] ? -1 : 1;
  if (c->key && c->key[a] != c->key[b]) return c->key[a] < c->key[b] ? -1 : 1;
  if (c->ts[a] != c->ts[b]) return c->ts[a] < c->ts[b] ? -1 : 1;
  return 0;
}

int orc_write_layout(i64 n, const i32 *key, const i64 *ts, i32 *out_dt, double *out_time,
                     i64 *out_perm) {
  for (i64 i = 0; i < n; ++i) {
    i64 day = floordiv64(ts[i], 86400000000ll);
    i64 tod = ts[i] - day * 86400000000ll;
    i64 secs = tod / 1000000, micros = tod % 1000000;
    char text[32], stripped[32];
    int len = snprintf(text, sizeof(text), "%02d:%02d:%02d", (int)(secs / 3600),
                       (int)(secs % 3600 / 60), (int)(secs % 60));
    if (micros) {
      len += snprintf(text + len, sizeof(text) - (size_t)len, ".%06d", (int)micros);
      while (text[len - 1] == '0') text[--len] = 0;
    }
    int o = 0;
    for (int k = 0; k < len; ++k)
      if (text[k] != ':') stripped[o++] = text[k];
    stripped[o] = 0;
    out_dt[i] = (i32)day;
    out_time[i] = strtod(stripped, NULL);
    out_perm[i] = i;
  }
  wl_ctx c = {out_dt, key, ts};
  return stable_sort_idx(out_perm, n, wl_cmp, &c);
}
