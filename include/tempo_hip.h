/*
 * tempo_hip.h -- C ABI of libtempo_hip.so, the MI355X (gfx950) engine behind tempo's TSDF
 * hot path.
 *
 * The reference (souvik-databricks/tempo) has no FFI or plugin interface: its TSDF methods
 * build Spark logical plans and Spark executes them.  Each entry point below therefore
 * replaces the Spark physical operators that one TSDF method's plan expands to; the comment
 * on every function cites the reference lines whose plan it stands in for.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer (HBM) unless the comment says "host";
 *  - columns are contiguous 1-D arrays; a partition key is a dictionary id in [0, n_keys);
 *    timestamps are int64 microseconds since the epoch (Spark TimestampType) unless
 *    ts_is_seconds != 0 (an integral ts column);
 *  - SQL NULL is a separate byte mask per column, 1 = valid; a NULL mask pointer = no NULLs;
 *  - the caller owns and allocates inputs, outputs and the workspace
 *    (tmp_*_workspace_bytes); the library allocates nothing and keeps no pointer;
 *  - work is enqueued on `stream` (a hipStream_t); the call returns after enqueueing, except
 *    for one small device->host read of key ranges that the ordering step needs;
 *  - return value: TMP_OK or a negative TMP_E* code, message via tmp_last_error().
 *  - row counts per call are limited to 2^31-1 (row indices are int32);
 *  - ordering works on one 64-bit word per row (key id, time, sequence value, NULL flags).
 *    Sequence values and NULL flags that do not fit next to key and time are handled inside
 *    the library (sorted in passes of their own / carried beside the words); only when key id
 *    and time range alone need more than 64 bits does a call return TMP_EUNSUPPORTED (callers
 *    then split the key dictionary: key order is row order, so the halves concatenate).
 */
#ifndef TEMPO_HIP_H
#define TEMPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMP_ABI_VERSION 2

#define TMP_OK 0
#define TMP_EINVAL (-1)       /* bad argument */
#define TMP_EWORKSPACE (-2)   /* workspace too small */
#define TMP_EHIP (-3)         /* HIP runtime error */
#define TMP_EUNSUPPORTED (-4) /* valid request outside what this build supports */
#define TMP_ECAPACITY (-5)    /* the result has more rows than the caller made room for */
#define TMP_EWIDE (-6)        /* tmp_range_stats: needs the wide workspace (see there) */

int tmp_abi_version(void);
/* Thread-local text of the last error on the calling thread ("" if none). */
const char *tmp_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Ordering: partitionBy(partitionCols).orderBy(...) + SortExec.
 *
 * Replaces the Exchange + Sort that every hot-path plan starts with:
 *   python/tempo/tsdf.py:117-121  (asofJoin: order by combined_ts[, seq], rec_ind)
 *   python/tempo/tsdf.py:563-572  (__baseWindow: order by ts.cast("long")[, seq.cast("long")])
 *   python/tempo/resample.py:63   (groupBy(partitionCols, window(ts)))
 * Output: out_perm[p] = input row at sorted position p (stable: ties keep input order).
 * ts_unit: 0 = order by the exact ts value, 1 = ts is microseconds and the order key is
 * floor(ts / 1e6) (timestamp.cast("long")).  seq may be NULL; seq NULLs sort first.
 */
typedef struct {
  int64_t n;
  const int32_t *key; /* NULL: a single series */
  int32_t n_keys;
  const int64_t *ts;
  int32_t ts_unit;
  const int64_t *seq;
  const uint8_t *seq_valid;
  int32_t *out_perm;
} tmp_sort_args;

size_t tmp_sort_order_workspace_bytes(const tmp_sort_args *a);
int tmp_sort_order(const tmp_sort_args *a, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * AS-OF join match.
 *
 * Replaces the union + Window(last(col, ignorenulls) over rows unboundedPreceding..current)
 * plan of TSDF.asofJoin / __getLastRightRow / __getTimePartitions:
 *   python/tempo/tsdf.py:463-560, :111-162, :164-190.
 * For every left row it returns which right row each joined column comes from, as one record
 * of (1 + n_masks) int32 per left row (row-major, so a row's indices share a cache line):
 *   out_idx[i*(1+n_masks)]     last right row visible to left row i (skipNulls=False value
 *                              rule, and the rule for columns without NULLs), or -1
 *   out_idx[i*(1+n_masks)+1+m] last visible right row whose mask m is valid (skipNulls=True
 *                              rule for a column with NULLs), or -1
 *   out_missing[0]          1 if some left row has no visible right row
 *   out_missing[1+m]        1 if some left row has no visible right row with mask m valid
 *                           (the non_null_ct == 0 warning test, tsdf.py:142-158)
 * "Visible" = same key and ordered before the left row by (ts, seq NULLS FIRST, right before
 * left) -- tsdf.py:117-121; with skew != 0 additionally inside the left row's time bracket or
 * carried over from the previous bracket, evaluated with the same fp64 expressions as
 * tsdf.py:178-185 (ts_partition_val = tsPartitionVal, v_is_int = it was a Python int,
 * one_minus_fraction = 1 - fraction computed by the caller in double).
 * masks: HOST array of n_masks device pointers (<= TMP_MAX_MASKS).
 * Row order of the result: sorted_output == 0 keeps the left table's own row order (record i
 * belongs to left row i; costs one random 4*(1+n_masks)-byte write per row).  sorted_output != 0
 * returns records in (key, ts) order -- the order Spark's window leaves them in, and the order
 * every later window operator wants: record p belongs to left row out_perm[p]; out_key /
 * out_ts (optional) receive that row's key id and timestamp, so only the left payload columns
 * have to be gathered through out_perm.
 */
#define TMP_MAX_MASKS 16
#define TMP_MAX_JOIN_COLS 16

typedef struct {
  int64_t n_left;
  int64_t n_right;
  const int32_t *key_left; /* NULL on both sides: a single series */
  const int64_t *ts_left;
  const int32_t *key_right;
  const int64_t *ts_right;
  const int64_t *seq_right;       /* NULL: right side has no sequence column */
  const uint8_t *seq_right_valid; /* NULL: no NULLs */
  int32_t n_keys;
  int32_t ts_is_seconds;
  int32_t n_masks;
  const uint8_t *const *masks;
  int32_t skew;
  int32_t v_is_int;
  double ts_partition_val;
  double one_minus_fraction;
  int32_t *out_idx;      /* [n_left][1 + n_masks] */
  int32_t sorted_output;
  int32_t *out_perm;     /* [n_left], required when sorted_output != 0 */
  int32_t *out_key;      /* [n_left] or NULL (sorted_output only) */
  int64_t *out_ts;       /* [n_left] or NULL (sorted_output only) */
  /* Fused payload join (sorted_output only, 8-byte element columns, HOST arrays of device
   * pointers, at most TMP_MAX_JOIN_COLS per side).  The merge writes the joined table itself,
   * every store streaming; a side with several payload columns is first packed into row
   * records so that a joined row costs one random read per side:
   *   out_left_cols[c][p]  = left_cols[c][out_perm[p]]
   *   out_right_cols[c][p] = right_cols[c][j], j = the right row chosen by right_col_rule[c]
   *                          (-1: last visible row; m: last visible row with mask m valid)
   *   out_right_valid[c][p] = a row was chosen and (right_col_valid[c] == -1 or mask
   *                          right_col_valid[c] is 1 on that row)
   * With these, out_idx may be NULL. */
  int32_t n_left_cols;
  const void *const *left_cols;
  void *const *out_left_cols;
  int32_t n_right_cols;
  const void *const *right_cols;
  const int32_t *right_col_rule;  /* host */
  const int32_t *right_col_valid; /* host */
  void *const *out_right_cols;
  uint8_t *const *out_right_valid; /* entries may be NULL */
  int32_t *out_missing;  /* [1 + n_masks] */
  /* HOST pointer, optional (sorted_output only).  Non-NULL: the caller can use the left table by
   * reference.  When the call finds the left table already in (key, ts) order -- row p of the
   * result is then left row p -- it stores 1 here and leaves out_perm, out_key, out_ts and
   * out_left_cols unwritten (they would be copies of the caller's own columns); otherwise it
   * stores 0 and writes them as described above. */
  int32_t *left_in_place;
} tmp_asof_args;

/* Streams: every kernel runs on `stream`, with one exception: when both sides of a join have 2^20
 * rows or more (and TEMPO_JOIN_STREAMS=0 is not in the environment), the right side is ordered on
 * a stream of the library's own (one per device, created on the first such call and kept), forked
 * from and joined back into `stream` by events inside the call -- everything the call enqueues is
 * complete once `stream` reaches what is enqueued after it, as for a single-stream call.  Calls
 * that overlap on one device share that stream and take turns on it. */
size_t tmp_asof_join_workspace_bytes(const tmp_asof_args *a);
int tmp_asof_join(const tmp_asof_args *a, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Row gather: with j = idx[i * idx_stride]: out[i] = j >= 0 ? src[j] : 0 ; out_valid[i] =
 * j >= 0 && (src_valid == NULL || src_valid[j]).  elem_bytes in {1, 4, 8}; idx_stride (in
 * int32 elements, >= 1) lets a column of tmp_asof_join's out_idx records be used directly.
 * Replaces the Project that materialises the joined right columns (tsdf.py:139,147) and the
 * row reordering implied by window operators.
 */
int tmp_gather(const void *src, const uint8_t *src_valid, int32_t elem_bytes, const int32_t *idx,
               int64_t idx_stride, int64_t n, void *out, uint8_t *out_valid, void *stream);

/* ---------------------------------------------------------------------------------------
 * withRangeStats: rolling RANGE-frame statistics.
 *
 * Replaces WindowExec over rangeBetween(-rangeBackWindowSecs, 0) with mean/count/min/max/
 * sum/stddev and the derived zscore: python/tempo/tsdf.py:673-721 (frame :704, aggregates
 * :710-717), ordering :563-572.  Frame of a row = rows of the same key whose
 * floor-second lies in [sec - range_back_secs, sec] (later same-second rows included).
 * Outputs are in sorted (key, sec) order; out_perm[p] = input row at output position p.
 * x_is_i64 = 0: x, out_min, out_max, out_sum are double arrays;
 * x_is_i64 = 1: they are int64 arrays (Spark sums int/bigint in int64, min/max stay integral).
 * out_valid[p]: bit0 = mean/min/max/sum not NULL (count > 0), bit1 = stddev not NULL
 * (count > 1), bit2 = zscore not NULL.
 */
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int32_t ts_is_seconds;
  int32_t x_is_i64;
  const void *x;
  const uint8_t *x_valid;
  int64_t range_back_secs;
  int32_t *out_perm;
  double *out_mean;
  int64_t *out_count;
  void *out_min;
  void *out_max;
  void *out_sum;
  double *out_stddev;
  double *out_zscore;
  uint8_t *out_valid;
  /* != 0: the caller takes the result rows in whatever order the call finds cheapest, out_perm
   * saying which input row each one belongs to.  A feed in time order over few (key, second)
   * cells is then not ordered at all (see below) and comes back in its own row order. */
  int32_t any_order;
} tmp_range_stats_args;

/* any_order: a double metric on 2^22 rows or more that arrive in time order (keys interleaved),
 * over at most 2^24 (key, second) cells between the first and the last second and at most 4096
 * keys, is aggregated per cell (two sweeps over the rows as they lie: count, sum, minimum, maximum;
 * then the squared deviations from the cell's mean); the cells of every key are combined forwards
 * and backwards inside blocks of range_back_secs + 1 seconds (counts and sums add, minima and
 * maxima compare, (count, mean, M2) merge as Spark's own CentralMomentAgg merges partial
 * aggregates), so that a row's frame is the union of at most two such runs; a last sweep writes
 * every row's statistics where the row lies (out_perm[i] = i).  Tables with NaN or infinite
 * values, and everything else, take the ordered path.  The sums are atomic adds: last bits of
 * mean / sum / stddev / zscore may differ between two runs.  A call that qualifies by its row and
 * key counts asks for about 2.9 GiB of workspace more (tables for 2^24 cells, sized before the
 * time range is known), whichever path it ends on.
 *
 * Two workspace sizes.  The plain one serves every table whose frames stay inside a block's
 * on-chip region (about 1280 rows back from a row, 256 same-second rows after it).  If some
 * frame is longer, a call with the plain workspace returns TMP_EWIDE (nothing usable has been
 * written); a call with the wide workspace completes those rows through per-chunk prefix
 * arrays in global memory, at a cost that does not grow with the frame length. */
size_t tmp_range_stats_workspace_bytes(const tmp_range_stats_args *a);
size_t tmp_range_stats_wide_workspace_bytes(const tmp_range_stats_args *a);
int tmp_range_stats(const tmp_range_stats_args *a, void *workspace, size_t workspace_bytes,
                    void *stream);

/* withRangeStats over several metric columns of one table (python/tempo/tsdf.py:707-717: the
 * reference appends six aggregates per column of colsToSummarize over ONE window): the rows are
 * ordered once, with every column travelling along, and each column's statistics are then written
 * from that order.  Same results, column by column, as tmp_range_stats on that column alone.
 * x, x_valid, x_is_i64 and the out_* members are HOST arrays of n_cols entries (1 <= n_cols <=
 * TMP_MAX_STATS_COLS); x_valid entries may be NULL; out_min / out_max / out_sum entries point to
 * double or int64 arrays as x_is_i64 says.  out_perm is shared: out_perm[p] = input row at output
 * position p.  Workspace rule as for tmp_range_stats (plain size: TMP_EWIDE when some frame leaves
 * the on-chip region; wide size: every frame length). */
#define TMP_MAX_STATS_COLS 4
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int32_t ts_is_seconds;
  int64_t range_back_secs;
  int32_t n_cols;
  const void *const *x;
  const uint8_t *const *x_valid;
  const int32_t *x_is_i64;
  int32_t *out_perm;
  double *const *out_mean;
  int64_t *const *out_count;
  void *const *out_min;
  void *const *out_max;
  void *const *out_sum;
  double *const *out_stddev;
  double *const *out_zscore;
  uint8_t *const *out_valid;
} tmp_range_stats_cols_args;
size_t tmp_range_stats_cols_workspace_bytes(const tmp_range_stats_cols_args *a);
size_t tmp_range_stats_cols_wide_workspace_bytes(const tmp_range_stats_cols_args *a);
int tmp_range_stats_cols(const tmp_range_stats_cols_args *a, void *workspace, size_t workspace_bytes,
                         void *stream);


/* ---------------------------------------------------------------------------------------
 * EMA: fixed-tap FIR over the previous rows of the same series.
 *
 * Replaces the `window` Lag frames + running sum of python/tempo/tsdf.py:615-635:
 * out[p] = 0 + w[0]*x[p] + w[1]*x[p-1] + ... (left-to-right adds, NULL or missing taps add
 * nothing), rows ordered as tsdf.py:563-572.  weights: HOST array of n_taps doubles computed
 * by the caller as exp_factor * (1 - exp_factor) ** i (tsdf.py:629).  n_taps <= TMP_MAX_TAPS.
 * The array has been copied to the device by the time the call returns (the copy is enqueued
 * ahead of the host read of the key ranges, which waits for the stream): the caller may free
 * or reuse it right away.
 * Output in sorted order with out_perm as above.
 */
#define TMP_MAX_TAPS 1024

typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int32_t ts_is_seconds;
  const int64_t *seq;
  const uint8_t *seq_valid;
  const double *x;
  const uint8_t *x_valid;
  int32_t n_taps;
  const double *weights;
  int32_t *out_perm;
  double *out_ema;
} tmp_ema_args;

size_t tmp_ema_workspace_bytes(const tmp_ema_args *a);
int tmp_ema(const tmp_ema_args *a, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * withLookbackFeatures: trailing-window feature tensor.
 *
 * Replaces collect_list(array(featureCols)) over rowsBetween(-L, -1):
 * python/tempo/tsdf.py:637-671.  For the row at sorted position p with m = min(L, rows before
 * it in its series): out_feat[o][l][f] = x_f of the row at position p - m + l (oldest first).
 * exact_size != 0 keeps only rows with m == L (tsdf.py:668-669): then o counts kept rows,
 * out_rows[o] = input row of output row o, and *out_count (device int64) = number kept.
 * exact_size == 0: o = p for every row, out_len[p] = m, entries l >= m are zero/invalid.
 * x / x_valid: HOST arrays of n_feat device pointers.  out_feat_valid may be NULL when no
 * feature column has NULLs.
 */
#define TMP_MAX_FEATURES 64

typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int32_t ts_is_seconds;
  const int64_t *seq;
  const uint8_t *seq_valid;
  int32_t n_feat;
  const double *const *x;
  const uint8_t *const *x_valid;
  int32_t lookback;
  int32_t exact_size;
  int32_t *out_rows;       /* [n] */
  int32_t *out_len;        /* [n], exact_size == 0 only (may be NULL otherwise) */
  double *out_feat;        /* [n][lookback][n_feat] capacity */
  uint8_t *out_feat_valid; /* same shape, or NULL */
  int64_t *out_count;      /* device scalar */
} tmp_lookback_args;

size_t tmp_lookback_workspace_bytes(const tmp_lookback_args *a);
int tmp_lookback(const tmp_lookback_args *a, void *workspace, size_t workspace_bytes,
                 void *stream);

/* ---------------------------------------------------------------------------------------
 * resample: tumbling-bucket aggregation.
 *
 * Replaces window(ts, "<n> <unit>") + groupBy(partitionCols, agg_key).agg(...):
 * python/tempo/resample.py:25-117 (bucket :33, floor :61-66, mean :67-72, min :73-79,
 * max :80-86, ceil :87-92).  Bucket start = ts - ((ts mod dur_us) + dur_us) mod dur_us.
 * Buckets come out ordered by (key, start); *out_count (device int64) = number of buckets.
 * func: TMP_FUNC_*.  cols: HOST array of n_cols device pointers, col_is_i64[c] selects
 * double or int64 (int64 also carries order-preserving codes of string columns).
 *   mean          -> out_f64[c*n + b]
 *   min / max     -> out_f64 or out_i64 [c*n + b] by column type
 *   floor / ceil  -> out_row[b] = input row with the smallest / largest (ts, cols...) tuple
 *                    under Spark struct ordering (NULL smallest, NaN greatest)
 * out_valid[c*n + b] = result not NULL (mean/min/max).
 * mean of at most 4 columns over at most 2048 keys and 2^20 (key, bucket) cells between the
 * table's first and last bucket, on 2^20 rows or more that arrive in some time order (or fall
 * into at most 512 cells): the rows are not ordered at all; every row adds its values to its
 * cell with double-precision atomic adds, so the last bits of such a mean depend on the order in
 * which the hardware performed the adds and may differ between two runs (the count, the bucket
 * membership and the NULL flags do not).  TEMPO_NO_DENSE_MEAN=1 in the environment keeps every
 * mean on the ordered path.
 * min / max of one or two columns take the same table (integer atomics on order-preserving codes of
 * the values: exact; a NaN extreme comes back as the canonical quiet NaN, and of -0.0 and 0.0 the
 * minimum is -0.0).  TEMPO_NO_DENSE_STATS=1 keeps them on the ordered path.
 * floor / ceil take it too (two sweeps: the cell's earliest / latest timestamp, then the row that
 * carries it); when several rows of a cell tie on that timestamp and there are columns to compare
 * them by, the call is redone the ordered way.  TEMPO_NO_DENSE_PICK=1 keeps the ordered path.
 */
#define TMP_FUNC_FLOOR 0
#define TMP_FUNC_MIN 1
#define TMP_FUNC_MAX 2
#define TMP_FUNC_MEAN 3
#define TMP_FUNC_CEIL 4
#define TMP_MAX_COLS 64

typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int64_t dur_us;
  int32_t func;
  int32_t n_cols;
  const void *const *cols;
  const int32_t *col_is_i64; /* host */
  const uint8_t *const *col_valid;
  int32_t *out_key;   /* [n] capacity */
  int64_t *out_start; /* [n] */
  double *out_f64;    /* [n_cols][n] */
  int64_t *out_i64;   /* [n_cols][n] */
  uint8_t *out_valid; /* [n_cols][n] */
  int32_t *out_row;   /* [n] */
  int64_t *out_count; /* device scalar */
} tmp_resample_args;

size_t tmp_resample_workspace_bytes(const tmp_resample_args *a);
int tmp_resample(const tmp_resample_args *a, void *workspace, size_t workspace_bytes,
                 void *stream);

/* ---------------------------------------------------------------------------------------
 * interpolate: upsample a sampled table to a regular grid and fill the gaps.
 *
 * Replaces the window / explode(sequence()) plan of Interpolation.interpolate,
 * python/tempo/interpol.py:260-378 (helpers :182-258, per-method rules :96-180, the linear
 * expression :66-94).  Input = the sampled table (one row per key and bucket of `freq_us`, what
 * tmp_resample returns).  In (key, ts) order every input row r yields the rows
 *   ts[r], ts[r] + freq, ..., < next_ts[r]      next_ts = ts of the key's next row, or ts[r] + freq
 * (the first one is the row itself, the others are "ts interpolated").  Per target column c a
 * cell is "interpolated" when its row is, or when the input value is NULL; such cells are filled
 * by `method`:
 *   TMP_FILL_ZERO   0            TMP_FILL_NULL   NULL
 *   TMP_FILL_FFILL  the key's last non-NULL value at or before r
 *   TMP_FILL_BFILL  the next row's value (NULL when that is NULL), except that an input row
 *                   whose own and next values are both NULL takes the key's first non-NULL
 *                   value after it (interpol.py:152-170)
 *   TMP_FILL_LINEAR interpol.py:74-87: between the surrounding non-NULL values for a NULL input
 *                   row, between the row and its successor otherwise, on whole seconds
 *                   (unix_timestamp), same fp64 operations in the same order; NULL when an end
 *                   is missing
 * Outputs have *out_count rows (device scalar), ordered by (key, ts).  The call first counts;
 * when the count exceeds `capacity` nothing else is written and TMP_ECAPACITY is returned, so
 * a caller may probe with capacity 0, read *out_count, allocate and call again.
 *   out_src[q]  input row the output row q stems from     out_ts[q]  its timestamp
 *   out_key[q]  its key id                                 out_ts_interp[q]  1 = generated row
 *   out_val[c*capacity + q], out_valid[...] (0 = NULL), out_interp[...] (is_interpolated_<c>)
 * A gap between two rows of a key that is shorter than freq_us (the input was not sampled at
 * this frequency) is rejected with TMP_EINVAL, as Spark's sequence() rejects it.
 */
#define TMP_FILL_ZERO 0
#define TMP_FILL_NULL 1
#define TMP_FILL_BFILL 2
#define TMP_FILL_FFILL 3
#define TMP_FILL_LINEAR 4

typedef struct {
  int64_t n;
  const int32_t *key; /* NULL: a single series */
  int32_t n_keys;
  const int64_t *ts;  /* microseconds */
  int64_t freq_us;
  int32_t method;
  int32_t n_cols;     /* <= TMP_MAX_MASKS */
  const double *const *cols;        /* HOST array of device pointers */
  const uint8_t *const *col_valid;  /* HOST array; entries may be NULL */
  int64_t capacity;
  int32_t *out_src;
  int32_t *out_key;
  int64_t *out_ts;
  uint8_t *out_ts_interp;
  double *out_val;     /* [n_cols][capacity] */
  uint8_t *out_valid;  /* [n_cols][capacity] */
  uint8_t *out_interp; /* [n_cols][capacity] */
  int64_t *out_count;  /* device scalar */
} tmp_interpolate_args;

size_t tmp_interpolate_workspace_bytes(const tmp_interpolate_args *a);
int tmp_interpolate(const tmp_interpolate_args *a, void *workspace, size_t workspace_bytes,
                    void *stream);

/* ---------------------------------------------------------------------------------------
 * withGroupedStats: mean / count / min / max / sum / stddev per (key, tumbling bucket).
 *
 * Replaces groupBy(partitionCols, window(ts_col, "<n> <unit>")).agg(mean, count, min, max,
 * sum, stddev) of python/tempo/tsdf.py:723-759 (the bucket is resample.py's: epoch aligned,
 * start = ts - ((ts mod dur) + dur) mod dur).  Buckets come back ordered by (key, start);
 * *out_count receives their number.  Per column c and bucket b (index c*n + b):
 *   out_n      number of non-NULL values (Spark count)
 *   out_mean   sum as double / out_n                     NULL when out_n == 0
 *   out_min / out_max / out_sum  8-byte cells of the column's own type (int64 sums wrap)
 *                                                        NULL when out_n == 0
 *   out_stddev sample standard deviation                 NULL when out_n < 2
 * NULLs are not flagged separately: out_n says which cells are NULL.
 * One or two columns over few cells on a feed in some time order (the conditions of tmp_resample's
 * mean, with 256 cells where it says 512) are not ordered either: two sweeps over the rows as
 * they lie, atomic adds for the sums (last bits of mean / double sum / stddev may differ between
 * two runs), integer atomics for counts, integer sums, minima and maxima (exact; a NaN extreme
 * comes back as the canonical quiet NaN).  TEMPO_NO_DENSE_STATS=1 keeps the ordered path.
 */
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int64_t dur_us;
  int32_t n_cols;
  const void *const *cols;   /* HOST array of n_cols device pointers (double or int64 columns) */
  const int32_t *col_is_i64; /* host */
  const uint8_t *const *col_valid;
  int32_t *out_key;   /* [n] capacity */
  int64_t *out_start; /* [n] */
  double *out_mean;   /* [n_cols][n] */
  int64_t *out_n;     /* [n_cols][n] */
  void *out_min;      /* [n_cols][n] 8-byte cells */
  void *out_max;
  void *out_sum;
  double *out_stddev; /* [n_cols][n] */
  int64_t *out_count; /* device scalar */
} tmp_grouped_stats_args;

size_t tmp_grouped_stats_workspace_bytes(const tmp_grouped_stats_args *a);
int tmp_grouped_stats(const tmp_grouped_stats_args *a, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Key shuffle, device side: Exchange hashpartitioning(partitionCols).
 *
 * Replaces the shuffle that Window.partitionBy / groupBy imply (tsdf.py:121, :571,
 * resample.py:63) on a multi-GPU node: rows are bucketed by owner = hash(key) % n_parts so
 * that one all-to-all moves every series to the GPU that owns it.
 * out_perm[p] = input row at position p of the bucketed order (stable within a bucket),
 * out_counts[r] (device int64[n_parts]) = rows owned by part r.
 * owner_table (optional, device int32[n_keys]): owner of every key id, for callers that balance
 * the parts by row counts (tmp_key_counts) instead of hashing; entries beyond n_parts - 1 are
 * clamped.  NULL selects the hash.
 *
 * Time-sliced keys (skew AS-OF join only, tsdf.py:164-190): a table entry -(s+1) says that key
 * is too hot for one part and is cut along the time axis.  Its rows go by the index
 * q = (ts.cast("double") / ts_partition_val).cast("integer") of their time bracket -- the same
 * expression the join evaluates -- to slice (q - bracket_min) * split_slices[s] / bracket_count
 * (clamped), owned by slice_owner[split_first[s] + slice].  A left row only ever sees right rows
 * of its own bracket and the carried-over tail of the bracket before it, so slices are
 * independent once that tail has been copied across each slice boundary: halo != 0 routes,
 * instead of the rows, exactly those copies -- rows of sliced keys whose remainder
 * (ts_double - v*q) / v >= one_minus_fraction and whose next bracket belongs to another part --
 * to that part.  In halo mode out_counts[r] counts the copies for part r and only the first
 * sum(out_counts) entries of out_perm are meaningful.  n_parts <= 255 then.
 */
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_parts;
  int32_t *out_perm;
  int64_t *out_counts;
  const int32_t *owner_table;
  int32_t n_keys; /* length of owner_table; ignored without one */
  /* everything below is read only when time-sliced keys are in use (ts != NULL) */
  const int64_t *ts;
  int32_t ts_is_seconds;
  int32_t v_is_int;
  double ts_partition_val;
  double one_minus_fraction;
  int64_t bracket_min;
  int64_t bracket_count;
  int32_t n_split;
  const int32_t *split_slices; /* device int32[n_split] */
  const int32_t *split_first;  /* device int32[n_split] */
  const int32_t *slice_owner;  /* device int32[sum of split_slices] */
  int32_t halo;
} tmp_partition_args;

size_t tmp_hash_partition_workspace_bytes(const tmp_partition_args *a);
int tmp_hash_partition(const tmp_partition_args *a, void *workspace, size_t workspace_bytes,
                       void *stream);

/* tmp_hash_partition for callers that cannot allocate device memory and therefore have no
 * workspace to pass: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  (No counterpart in the reference: Spark allocates its shuffle buffers itself.)
 * Same arguments, outputs and errors as tmp_hash_partition; the scratch (radix ping-pong buffers,
 * digit counters) is a fixed array inside the library's device code, TMP_STATIC_PARTITION_MAX_ROWS
 * * 24 bytes + 4 MiB, held once per GPU the library is used on until the process ends.  Calls
 * with more rows return TMP_ECAPACITY and write nothing.  The call allocates and frees nothing
 * (the first call on a device looks the array's address up, once).  All calls on one GPU share
 * the array: two calls that overlap on the same GPU (different streams or threads) race; calls on
 * one stream, or on different GPUs, do not.
 */
#define TMP_STATIC_PARTITION_MAX_ROWS 1048576
int tmp_hash_partition_static(const tmp_partition_args *a, void *stream);

/* Route and move in one call: what tmp_hash_partition + one tmp_gather per column deliver, without
 * row ids or sort passes, for up to 64 parts.  `route` is read like tmp_hash_partition's
 * argument (out_perm is not used); afterwards out_cols[c][first(r) + j] = cols[c][i_j], i_j the
 * j-th input row routed to part r in input order, first(r) = route.out_counts[0] + ... +
 * route.out_counts[r-1].  Element sizes 1, 4 and 8 bytes; cols / col_bytes / out_cols are HOST
 * arrays of n_cols entries (at most 12).  capacity = rows every output column has room for:
 * route.n always suffices; in halo mode few rows move and a smaller capacity may be offered --
 * TMP_ECAPACITY (with route.out_counts valid) says it was not enough.  Does one small
 * device->host read.
 */
typedef struct {
  tmp_partition_args route;
  int32_t n_cols;
  const void *const *cols;
  const int32_t *col_bytes;
  void *const *out_cols;
  int64_t capacity;
} tmp_partition_move_args;

size_t tmp_partition_move_workspace_bytes(const tmp_partition_move_args *a);
int tmp_partition_move(const tmp_partition_move_args *a, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---------------------------------------------------------------------------------------
 * Narrow wire format for the columns of a key shuffle (what Spark's Exchange serialises for
 * partitionBy(partitionCols), python/tempo/tsdf.py:117-121, :563-572): the key id and up to
 * TMP_WIRE_MAX_MASKS 0/1 flag columns share one word of key_bytes (1, 2 or 4) bytes --
 * bit key_bits + m is flag column m -- and every int64 column k travels as its offset from
 * int_min[k] in int_bytes[k] = 4, 5 or 6 bytes: the low four in wire_lo[k] (uint32), the upper one
 * or two in wire_hi[k] (uint8 / uint16; not used with 4).  tmp_wire_encode reads key / masks / ints
 * and writes wire_key / wire_lo / wire_hi; tmp_wire_decode reads those and writes key / masks /
 * ints: decode(encode(x)) == x bit for bit as long as 0 <= key < 2^key_bits, the flags are 0 / 1
 * and 0 <= ints[k][i] - int_min[k] < 2^(8 * int_bytes[k]).  Elementwise: rows in any order.  All
 * columns are device arrays of n entries; one sweep, 16-byte accesses when every column is
 * 16-byte aligned.
 */
#define TMP_WIRE_MAX_MASKS 8
#define TMP_WIRE_MAX_INTS 8
typedef struct {
  int64_t n;
  int32_t *key;
  int32_t key_bits;
  int32_t key_bytes;
  int32_t n_masks;
  int32_t n_ints;
  uint8_t *masks[TMP_WIRE_MAX_MASKS];
  int64_t *ints[TMP_WIRE_MAX_INTS];
  int64_t int_min[TMP_WIRE_MAX_INTS];
  int32_t int_bytes[TMP_WIRE_MAX_INTS];
  void *wire_key;
  void *wire_lo[TMP_WIRE_MAX_INTS];
  void *wire_hi[TMP_WIRE_MAX_INTS];
} tmp_wire_args;

int tmp_wire_encode(const tmp_wire_args *a, void *stream);
int tmp_wire_decode(const tmp_wire_args *a, void *stream);

/* Rows per key id over every stride-th row (stride 1 = exact counts): what a balanced owner
 * table is computed from -- Spark has no counterpart, its hash partitioning leaves a hot key's
 * task as long as it is (the skew tsdf.py:164-190 works around).  out_counts: device
 * int64[n_keys], overwritten.  Keys outside [0, n_keys) are rejected by the caller's
 * dictionary encoding and are skipped here.  Needs no workspace.
 */
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  int32_t stride;
  int64_t *out_counts;
} tmp_key_counts_args;

size_t tmp_key_counts_workspace_bytes(const tmp_key_counts_args *a);
int tmp_key_counts(const tmp_key_counts_args *a, void *workspace, size_t workspace_bytes,
                   void *stream);
/* The owner function itself (host-callable, same arithmetic as the kernel). */
int32_t tmp_key_owner(int32_t key, int32_t n_parts);

/* ---------------------------------------------------------------------------------------
 * Table write layout: the two derived columns and the row order of a date-partitioned,
 * (key, time)-clustered table.
 *
 * Replaces python/tempo/io.py:29-30 (event_dt = to_date(ts); event_time = the "HH:mm:ss[.f]"
 * part of ts.cast("string") with the colons removed, cast to double), and stands in for
 * partitionBy("event_dt") + "optimize ... zorder by (partitionCols, ..., event_time)" of
 * io.py:35-39 with one linear order: rows sorted by (event_dt, key, ts), ties in input order.
 * Session time zone is UTC.
 *   out_event_dt[i]   days since 1970-01-01 of input row i (floor for times before the epoch)
 *   out_event_time[i] HHMMSS.ffffff of input row i as the double nearest to that decimal
 *   out_perm[p]       input row at position p of the (event_dt, key, ts) order
 *   out_dt_range      device int32[2]: smallest and largest event_dt (untouched when n == 0)
 * key may be NULL (no partition columns).  ts is int64 microseconds.
 */
typedef struct {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int32_t *out_event_dt;
  double *out_event_time;
  int32_t *out_perm;
  int32_t *out_dt_range;
} tmp_write_layout_args;

size_t tmp_write_layout_workspace_bytes(const tmp_write_layout_args *a);
int tmp_write_layout(const tmp_write_layout_args *a, void *workspace, size_t workspace_bytes,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * Arrow C data interface import: one Arrow array (a column chunk) from host memory into a device
 * column, straight out of the Arrow buffers -- no pandas / numpy copy in between.
 *
 * Replaces the DataFrame hand-over of python/tempo/tsdf.py:24-36 (TSDF.__init__ takes a Spark
 * DataFrame; Spark hands partitions over as Arrow batches through mapInArrow / toArrow) for the
 * columns the hot path reads: dictionary-encoded partition keys, timestamp[*] time indices and
 * numeric metrics.  The structs are the ones the Arrow specification fixes (guarded by the
 * specification's own macro, so Arrow's header may be included as well).
 *
 *   array / schema   an exported array and its type.  Supported storage: int8..int64,
 *                    uint8..uint64, float32, float64, timestamp[s|ms|us|ns] ("c C s S i I l L f g
 *                    ts?:..."), and dictionary arrays with integer indices (the indices are
 *                    imported; the dictionary stays with the caller).  array->offset is honoured.
 *   out_kind         element type of out_values:
 *                      TMP_ARROW_I32      int32   (integers, dictionary indices: key ids)
 *                      TMP_ARROW_I64      int64   (integers; timestamps become microseconds,
 *                                                  nanoseconds floored)
 *                      TMP_ARROW_I64_RAW  int64   (timestamps in their own unit)
 *                      TMP_ARROW_F64      double  (floats and integers)
 *   out_values       device array of array->length elements; slots that are NULL hold 0
 *   out_valid        device uint8[length]: 1 = not NULL.  May be NULL when array->null_count == 0.
 *   out_null_count   host int64, may be NULL: the NULLs of this array (counted from the bitmap
 *                    when the producer left null_count at -1)
 * The copies are queued on `stream`; the Arrow buffers must stay alive until the stream has
 * passed them (synchronise before calling the array's release callback).
 */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char *format;
  const char *name;
  const char *metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema **children;
  struct ArrowSchema *dictionary;
  void (*release)(struct ArrowSchema *);
  void *private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void **buffers;
  struct ArrowArray **children;
  struct ArrowArray *dictionary;
  void (*release)(struct ArrowArray *);
  void *private_data;
};
#endif

#define TMP_ARROW_I32 0
#define TMP_ARROW_I64 1
#define TMP_ARROW_F64 2
#define TMP_ARROW_I64_RAW 3

size_t tmp_arrow_import_workspace_bytes(const struct ArrowArray *array, const struct ArrowSchema *schema);
int tmp_arrow_import(const struct ArrowArray *array, const struct ArrowSchema *schema, int32_t out_kind,
                     void *out_values, uint8_t *out_valid, int64_t *out_null_count, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Diagnostics: per-kernel device time (no counterpart in the reference; Spark exposes stage
 * timings through its UI).  While enabled, every kernel launch of the entry points above is
 * bracketed by two HIP events on its launch stream.  tmp_profile_enable(1) creates the event
 * pool the first time (the only call here that creates HIP objects); tmp_profile_get needs the
 * stream to have been synchronised.  Operators may be called from several host threads while
 * it is enabled (launch slots come from an atomic counter); enable / reset / count / get are for
 * one thread while no operator call is in flight.  At most 8192 launches are recorded between
 * resets; tmp_profile_dropped() says how many more were launched and not recorded.
 */
int tmp_profile_enable(int on);
int tmp_profile_reset(void);
int tmp_profile_count(void);
int tmp_profile_dropped(void);
int tmp_profile_get(int i, const char **kernel_name, float *ms);
/* A caller-chosen tag stamped on every launch recorded from now on (e.g. which call it belongs to). */
int tmp_profile_set_tag(int tag);
int tmp_profile_tag(int i);

#ifdef __cplusplus
}
#endif
#endif /* TEMPO_HIP_H */
