// resample: tumbling-bucket aggregation.  Replaces window(ts, "<n> <unit>") +
// groupBy(partitionCols, agg_key).agg(...) of python/tempo/resample.py:25-117.
//
// Rows are ordered by (key, ts); a bucket is a run of rows with equal key and equal
// floor(ts / dur).  Pass 1 flags run heads and counts them per block, pass 2 turns the counts
// into offsets, pass 3 writes each run's first position, and pass 4 gives one wave to each
// bucket: lanes stride over the run, reduce with wave shuffles and lane 0 writes the result.
#include <string.h>

#include "common.h"
#include "dense_cells.h"

namespace tmp {

constexpr int RB_ROWS = 1024;
constexpr int RB_THREADS = 256;
constexpr int BB_TILE = 4096;       // rows per tile of the big-bucket path
constexpr int64_t BB_BIG = 32768;   // buckets longer than this are cut into tiles

// Bucket index of an order key.  dur == 1: the rows were ordered by the bucket index itself
// (operators that do not care about the order inside a bucket), the key is the index.
__device__ __forceinline__ int64_t bucket_index(int64_t prim, int64_t dur) {
  return dur == 1 ? prim : floordiv_i64(prim, dur);
}

__device__ __forceinline__ bool bucket_head(const uint64_t *comp, int64_t i, const Layout &lay,
                                            int64_t dur) {
  if (i == 0) return true;
  uint64_t a = comp[i - 1], b = comp[i];
  if (lay.key_of(a) != lay.key_of(b)) return true;
  return bucket_index(lay.prim_of(a), dur) != bucket_index(lay.prim_of(b), dur);
}

__global__ __launch_bounds__(RB_THREADS) void bucket_count_kernel(const uint64_t *comp, int64_t n,
                                                                  Layout lay, int64_t dur,
                                                                  uint32_t *block_counts) {
  __shared__ uint32_t wsum[RB_THREADS / 64];
  int64_t t0 = (int64_t)blockIdx.x * RB_ROWS;
  uint32_t heads = 0;
  for (int r = 0; r < RB_ROWS / RB_THREADS; ++r) {
    int64_t i = t0 + r * RB_THREADS + threadIdx.x;
    if (i < n && bucket_head(comp, i, lay, dur)) ++heads;
  }
  for (int o = 32; o > 0; o >>= 1) heads += __shfl_down(heads, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = heads;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < RB_THREADS / 64; ++w) t += wsum[w];
    block_counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void bucket_offsets_kernel(uint32_t *block_counts, int nblocks,
                                                              int64_t *out_count, uint32_t *bstart,
                                                              int64_t n) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    int i = base + tid;
    uint32_t v = i < nblocks ? block_counts[i] : 0u;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t wp = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      uint32_t t = wave_tot[w];
      if (w < wave) wp += t;
      tot += t;
    }
    if (i < nblocks) block_counts[i] = carry_s + wp + inc - v;
    __syncthreads();
    if (tid == 0) carry_s += tot;
    __syncthreads();
  }
  if (tid == 0) {
    *out_count = (int64_t)carry_s;
    bstart[carry_s] = (uint32_t)n;  // sentinel: end of the last bucket
  }
}

__global__ __launch_bounds__(RB_THREADS) void bucket_starts_kernel(const uint64_t *comp, int64_t n,
                                                                   Layout lay, int64_t dur, int64_t unit,
                                                                   const uint32_t *block_offsets,
                                                                   uint32_t *bstart, int32_t *out_key,
                                                                   int64_t *out_start) {
  __shared__ uint32_t wsum[RB_THREADS / 64];
  int64_t t0 = (int64_t)blockIdx.x * RB_ROWS;
  uint32_t base = block_offsets[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < RB_ROWS / RB_THREADS; ++r) {
    int64_t i = t0 + r * RB_THREADS + threadIdx.x;
    bool head = i < n && bucket_head(comp, i, lay, dur);
    uint64_t bal = __ballot(head);
    uint32_t before = (uint32_t)__popcll(bal & (lane ? (~0ull >> (64 - lane)) : 0ull));
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wp = 0, tot = 0;
    for (int w = 0; w < RB_THREADS / 64; ++w) {
      if (w < wave) wp += wsum[w];
      tot += wsum[w];
    }
    if (head) {
      uint32_t b = base + wp + before;
      bstart[b] = (uint32_t)i;
      uint64_t c = comp[i];
      out_key[b] = (int32_t)lay.key_of(c);
      // ts - ((ts mod dur) + dur) mod dur, in the timestamp's unit
      out_start[b] = bucket_index(lay.prim_of(c), dur) * dur * unit;
    }
    base += tot;
    __syncthreads();
  }
}


// Spark ordering of one cell against another: NULL smallest, NaN greatest, -0.0 == 0.0.
__device__ __forceinline__ int cmp_cell(const ColPtrs &c, int k, uint32_t a, uint32_t b) {
  bool va = !c.valid[k] || c.valid[k][a], vb = !c.valid[k] || c.valid[k][b];
  if (va != vb) return va ? 1 : -1;
  if (!va) return 0;
  if (c.is_i64[k]) {
    int64_t x = ((const int64_t *)c.col[k])[a], y = ((const int64_t *)c.col[k])[b];
    return x < y ? -1 : (x > y ? 1 : 0);
  }
  double x = ((const double *)c.col[k])[a], y = ((const double *)c.col[k])[b];
  if (x == y) return 0;
  if (x != x) return (y != y) ? 0 : 1;
  if (y != y) return -1;
  return x < y ? -1 : 1;
}

__device__ __forceinline__ bool dlt(double a, double b) {  // a < b, NaN greatest
  if (a != a) return false;
  if (b != b) return true;
  return a < b;
}

// floor / ceil: the row with the smallest / largest (ts, cols...) tuple.  Rows are ordered by
// ts, so only the rows tying with the first / last ts of the bucket compete.
__global__ __launch_bounds__(RB_THREADS) void bucket_pick_kernel(const uint64_t *comp,
                                                                 const uint32_t *perm, Layout lay,
                                                                 const uint32_t *bstart,
                                                                 const int64_t *count, int func,
                                                                 int n_cols, ColPtrs cols,
                                                                 int32_t *out_row) {
  int64_t nb = *count;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int low = lay.low_bits;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
    int64_t lo = bstart[b], hi = bstart[b + 1];  // [lo, hi)
    int64_t best_pos;
    if (func == TMP_FUNC_FLOOR) {
      best_pos = lo;
      uint64_t k0 = comp[lo] >> low;
      for (int64_t j = lo + 1; j < hi && (comp[j] >> low) == k0; ++j) {
        int c = 0;
        for (int k = 0; c == 0 && k < n_cols; ++k) c = cmp_cell(cols, k, perm[j], perm[best_pos]);
        if (c < 0) best_pos = j;
      }
    } else {
      best_pos = hi - 1;
      uint64_t k0 = comp[hi - 1] >> low;
      int64_t first = hi - 1;
      while (first > lo && (comp[first - 1] >> low) == k0) --first;
      best_pos = first;
      for (int64_t j = first + 1; j < hi; ++j) {
        int c = 0;
        for (int k = 0; c == 0 && k < n_cols; ++k) c = cmp_cell(cols, k, perm[j], perm[best_pos]);
        if (c > 0) best_pos = j;
      }
    }
    out_row[b] = (int32_t)perm[best_pos];
  }
}

// Order in which the waves take the buckets.  Buckets are numbered key by key, but the rows of a
// time-ordered feed lie time by time: waves that run side by side should work on the same stretch
// of time for all keys, so that the sectors they gather from are fetched once for all of them.
// Work item j goes to bucket (j mod K) * per + j / K (per = ceil(nb / K), K = number of keys): a
// permutation of [0, K * per) whose items beyond nb are skipped -- exact interleaving when every
// key has the same number of buckets, merely a different order otherwise.
__device__ __forceinline__ int64_t work_bucket(int64_t j, int64_t nb, int64_t n_keys, int64_t per) {
  const int64_t b = (j % n_keys) * per + j / n_keys;
  return b < nb ? b : -1;
}

// mean / min / max: one wave per bucket and column.
__global__ __launch_bounds__(RB_THREADS) void bucket_reduce_kernel(const uint32_t *perm,
                                                                   const uint32_t *bstart,
                                                                   const int64_t *count, int func,
                                                                   int n_cols, ColPtrs cols,
                                                                   int64_t cap, double *out_f64,
                                                                   int64_t *out_i64,
                                                                   uint8_t *out_valid, int64_t n_keys) {
  const int64_t nb = *count;
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t per = (nb + n_keys - 1) / n_keys;
  for (int64_t j = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); j < per * n_keys; j += waves) {
    const int64_t b = work_bucket(j, nb, n_keys, per);
    if (b < 0) continue;
    int64_t lo = bstart[b], hi = bstart[b + 1];
    if (hi - lo > BB_BIG) continue;  // cut into tiles by the big-bucket kernels
    for (int k = 0; k < n_cols; ++k) {
      const uint8_t *valid = cols.valid[k];
      const bool is_i = cols.is_i64[k];
      double sum = 0.0, dbest = 0.0;
      int64_t ibest = 0, cnt = 0;
      bool have = false;
      for (int64_t j = lo + lane; j < hi; j += 64) {
        uint32_t r = perm ? perm[j] : (uint32_t)j;
        if (valid && !valid[r]) continue;
        ++cnt;
        if (is_i) {
          int64_t v = ((const int64_t *)cols.col[k])[r];
          sum += (double)v;
          if (!have || (func == TMP_FUNC_MIN ? v < ibest : v > ibest)) ibest = v;
        } else {
          double v = ((const double *)cols.col[k])[r];
          sum += v;
          if (!have || (func == TMP_FUNC_MIN ? dlt(v, dbest) : dlt(dbest, v))) dbest = v;
        }
        have = true;
      }
      for (int o = 32; o > 0; o >>= 1) {
        double s2 = __shfl_down(sum, o, 64);
        int64_t c2 = __shfl_down(cnt, o, 64);
        double d2 = __shfl_down(dbest, o, 64);
        int64_t i2 = __shfl_down(ibest, o, 64);
        int h2 = __shfl_down((int)have, o, 64);
        sum += s2;
        cnt += c2;
        if (h2) {
          if (is_i) {
            if (!have || (func == TMP_FUNC_MIN ? i2 < ibest : i2 > ibest)) ibest = i2;
          } else {
            if (!have || (func == TMP_FUNC_MIN ? dlt(d2, dbest) : dlt(dbest, d2))) dbest = d2;
          }
          have = true;
        }
      }
      if (lane == 0) {
        int64_t o = (int64_t)k * cap + b;
        out_valid[o] = cnt > 0;
        double f = 0.0;
        int64_t iv = 0;
        if (cnt > 0) {
          if (func == TMP_FUNC_MEAN) f = sum / (double)cnt;
          else if (is_i) iv = ibest;
          else f = dbest;
        }
        out_f64[o] = f;
        out_i64[o] = iv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Big buckets.  A bucket of more than BB_BIG rows is not given to one wave: the ordered rows
// are cut into tiles of BB_TILE, one wave reduces the part of a big bucket inside its tile to
// (count, sum, integer sum, min, max, mean, M2), and one wave per big bucket merges its tiles
// (sums add, min / max compare, (count, mean, M2) by the pairwise variance update).  A tile
// meets at most two big buckets -- the one holding its first row and the one holding its last.
// ------------------------------------------------------------------------------------------
struct BigAgg {
  double sum, mean, m2;  // mean: as an offset from `ref`
  double ref;            // a value of the data the part's sums were taken around
  int64_t cnt;
  uint64_t isum, mn, mx;  // min / max as raw cell bits
  int32_t have;
  int32_t pad;
};

__device__ __forceinline__ void agg_merge(BigAgg &a, const BigAgg &b, bool is_i) {
  if (!b.have) return;
  if (!a.have) { a = b; return; }
  if (is_i) {
    if ((int64_t)b.mn < (int64_t)a.mn) a.mn = b.mn;
    if ((int64_t)b.mx > (int64_t)a.mx) a.mx = b.mx;
  } else {
    if (dlt(__longlong_as_double((long long)b.mn), __longlong_as_double((long long)a.mn))) a.mn = b.mn;
    if (dlt(__longlong_as_double((long long)a.mx), __longlong_as_double((long long)b.mx))) a.mx = b.mx;
  }
  const double na = (double)a.cnt, nb = (double)b.cnt, nn = na + nb;
  // the parts' means differ by little and sit far from 0 when the values do: form the difference
  // from the offsets (the references subtract without rounding when they are close)
  const double delta = (b.ref - a.ref) + (b.mean - a.mean);
  a.mean += delta * (nb / nn);
  a.m2 += b.m2 + delta * delta * (na * nb / nn);
  a.sum += b.sum;
  a.isum += b.isum;
  a.cnt += b.cnt;
}

__device__ __forceinline__ BigAgg agg_shfl_down(const BigAgg &a, int off) {
  BigAgg r;
  r.sum = __shfl_down(a.sum, off, 64);
  r.mean = __shfl_down(a.mean, off, 64);
  r.m2 = __shfl_down(a.m2, off, 64);
  r.ref = __shfl_down(a.ref, off, 64);
  r.cnt = __shfl_down(a.cnt, off, 64);
  r.isum = __shfl_down(a.isum, off, 64);
  r.mn = __shfl_down(a.mn, off, 64);
  r.mx = __shfl_down(a.mx, off, 64);
  r.have = __shfl_down(a.have, off, 64);
  r.pad = 0;
  return r;
}

// Rows [lo, hi) of column k, lanes striding, reduced over the wave; lane 0 holds the result.
__device__ __forceinline__ BigAgg agg_rows(const uint32_t *perm, const ColPtrs &cols, int k, int64_t lo,
                                           int64_t hi, int lane) {
  const uint8_t *valid = cols.valid[k];
  const bool is_i = cols.is_i64[k];
  const int64_t *ci = (const int64_t *)cols.col[k];
  const double *cd = (const double *)cols.col[k];
  // pivot for the squared deviations: the first usable value of the rows (searched 64 rows at a
  // time -- a part that opens with NULLs must not fall back to 0, which would leave the squares of
  // the values themselves to cancel)
  double pivot = 0.0;
  for (int64_t j0 = lo; j0 < hi; j0 += 64) {
    const int64_t j = j0 + lane;
    double v = 0.0;
    bool ok = false;
    if (j < hi) {
      const uint32_t r = perm ? perm[j] : (uint32_t)j;
      if (!valid || valid[r]) {
        v = is_i ? (double)ci[r] : cd[r];
        ok = __builtin_isfinite(v);
      }
    }
    const uint64_t bal = __ballot(ok);
    if (bal) {
      pivot = __shfl(v, (int)__ffsll((long long)bal) - 1, 64);
      break;
    }
  }
  BigAgg a;
  a.ref = pivot;
  a.sum = 0.0; a.mean = 0.0; a.m2 = 0.0; a.cnt = 0; a.isum = 0; a.mn = 0; a.mx = 0; a.have = 0; a.pad = 0;
  double sd = 0.0, sq = 0.0;
  for (int64_t j = lo + lane; j < hi; j += 64) {
    uint32_t r = perm ? perm[j] : (uint32_t)j;
    if (valid && !valid[r]) continue;
    double v;
    uint64_t bits;
    if (is_i) {
      int64_t iv = ci[r];
      v = (double)iv;
      bits = (uint64_t)iv;
      a.isum += bits;
      if (!a.have || iv < (int64_t)a.mn) a.mn = bits;
      if (!a.have || iv > (int64_t)a.mx) a.mx = bits;
    } else {
      v = cd[r];
      bits = (uint64_t)__double_as_longlong(v);
      if (!a.have || dlt(v, __longlong_as_double((long long)a.mn))) a.mn = bits;
      if (!a.have || dlt(__longlong_as_double((long long)a.mx), v)) a.mx = bits;
    }
    a.have = 1;
    ++a.cnt;
    a.sum += v;
    const double d = v - pivot;
    sd += d;
    sq += d * d;
  }
  if (a.cnt > 0) {
    const double n = (double)a.cnt;
    a.mean = sd / n;
    a.m2 = sq - sd * sd / n;
    if (a.m2 < 0.0) a.m2 = 0.0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    BigAgg o = agg_shfl_down(a, off);
    agg_merge(a, o, is_i);
  }
  return a;
}

// part[(tile * 2 + slot) * n_cols + k]; slot 0: the big bucket holding the tile's first row,
// slot 1: the big bucket holding its last row when that is another one.
__global__ __launch_bounds__(RB_THREADS) void bucket_big_tiles_kernel(const uint32_t *perm,
                                                                      const uint32_t *bstart,
                                                                      const int64_t *count, int64_t n,
                                                                      int n_cols, ColPtrs cols,
                                                                      BigAgg *part, int64_t n_keys) {
  const int64_t nb = *count;
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t tiles = (n + BB_TILE - 1) / BB_TILE;
  const int64_t per = (tiles + n_keys - 1) / n_keys;  // tiles are taken key-interleaved, like buckets
  for (int64_t j = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); j < per * n_keys; j += waves) {
    const int64_t t = work_bucket(j, tiles, n_keys, per);
    if (t < 0) continue;
    const int64_t r0 = t * BB_TILE, r1 = r0 + BB_TILE < n ? r0 + BB_TILE : n;
    for (int slot = 0; slot < 2; ++slot) {
      const int64_t row = slot == 0 ? r0 : r1 - 1;
      // the bucket holding `row`: last b with bstart[b] <= row
      int64_t lo = 0, hi = nb;
      while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)bstart[mid] <= row) lo = mid + 1;
        else hi = mid;
      }
      const int64_t b = lo - 1;
      const int64_t blo = bstart[b], bhi = bstart[b + 1];
      const bool big = bhi - blo > BB_BIG;
      const bool dup = slot == 1 && blo <= r0;  // same bucket as slot 0
      const int64_t s0 = blo > r0 ? blo : r0, s1 = bhi < r1 ? bhi : r1;
      for (int k = 0; k < n_cols; ++k) {
        BigAgg a;
        a.ref = 0.0;
        a.sum = 0.0; a.mean = 0.0; a.m2 = 0.0; a.cnt = 0; a.isum = 0; a.mn = 0; a.mx = 0; a.have = 0; a.pad = 0;
        if (big && !dup) a = agg_rows(perm, cols, k, s0, s1, lane);
        if (lane == 0) part[(size_t)(t * 2 + slot) * n_cols + k] = a;
      }
    }
  }
}

// One wave per big bucket: merge its tiles' parts; lane 0 holds the result for column k.
__device__ __forceinline__ BigAgg agg_bucket(const BigAgg *part, int n_cols, int k, bool is_i, int64_t lo,
                                             int64_t hi, int lane) {
  const int64_t t0 = lo / BB_TILE, t1 = (hi - 1) / BB_TILE;
  BigAgg a;
  a.ref = 0.0;
  a.sum = 0.0; a.mean = 0.0; a.m2 = 0.0; a.cnt = 0; a.isum = 0; a.mn = 0; a.mx = 0; a.have = 0; a.pad = 0;
  for (int64_t t = t0 + lane; t <= t1; t += 64) {
    // in its first tile the bucket is slot 0 only if it holds that tile's first row
    const int slot = (t == t0 && lo > t * BB_TILE) ? 1 : 0;
    agg_merge(a, part[(size_t)(t * 2 + slot) * n_cols + k], is_i);
  }
  for (int off = 32; off > 0; off >>= 1) {
    BigAgg o = agg_shfl_down(a, off);
    agg_merge(a, o, is_i);
  }
  return a;
}

__global__ __launch_bounds__(RB_THREADS) void bucket_big_reduce_kernel(const uint32_t *bstart,
                                                                       const int64_t *count, int func,
                                                                       int n_cols, ColPtrs cols,
                                                                       const BigAgg *part, int64_t cap,
                                                                       double *out_f64, int64_t *out_i64,
                                                                       uint8_t *out_valid) {
  const int64_t nb = *count;
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // 64 buckets are tested at a time, one per lane; the big ones among them are then merged by
  // the whole wave, one after the other
  for (int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; base < nb; base += waves * 64) {
    const int64_t mine = base + lane;
    uint64_t todo = __ballot(mine < nb && (int64_t)bstart[mine + 1] - (int64_t)bstart[mine] > BB_BIG);
    for (; todo; todo &= todo - 1) {
    const int64_t b = base + (__ffsll((long long)todo) - 1);
    const int64_t lo = bstart[b], hi = bstart[b + 1];
    for (int k = 0; k < n_cols; ++k) {
      const bool is_i = cols.is_i64[k];
      BigAgg a = agg_bucket(part, n_cols, k, is_i, lo, hi, lane);
      if (lane == 0) {
        int64_t o = (int64_t)k * cap + b;
        out_valid[o] = a.cnt > 0;
        double f = 0.0;
        int64_t iv = 0;
        if (a.cnt > 0) {
          const uint64_t best = func == TMP_FUNC_MIN ? a.mn : a.mx;
          if (func == TMP_FUNC_MEAN) f = a.sum / (double)a.cnt;
          else if (is_i) iv = (int64_t)best;
          else f = __longlong_as_double((long long)best);
        }
        out_f64[o] = f;
        out_i64[o] = iv;
      }
    }
    }
  }
}

// withGroupedStats: one wave per bucket; per column a first sweep for count / sum / min / max
// and a second one, around the mean just found, for the sum of squared deviations.
struct StatsOut {
  double *mean;
  int64_t *n;
  uint64_t *mn;
  uint64_t *mx;
  uint64_t *sum;
  double *stddev;
};

__global__ __launch_bounds__(RB_THREADS) void bucket_stats_kernel(const uint32_t *perm,
                                                                  const uint32_t *bstart,
                                                                  const int64_t *count, int n_cols,
                                                                  ColPtrs cols, int64_t cap,
                                                                  StatsOut o, int64_t n_keys) {
  const int64_t nb = *count;
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t per = (nb + n_keys - 1) / n_keys;
  for (int64_t j = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); j < per * n_keys; j += waves) {
    const int64_t b = work_bucket(j, nb, n_keys, per);
    if (b < 0) continue;
    const int64_t lo = bstart[b], hi = bstart[b + 1];
    if (hi - lo > BB_BIG) continue;  // cut into tiles by the big-bucket kernels
    for (int k = 0; k < n_cols; ++k) {
      const uint8_t *valid = cols.valid[k];
      const bool is_i = cols.is_i64[k];
      const int64_t *ci = (const int64_t *)cols.col[k];
      const double *cd = (const double *)cols.col[k];
      double sum = 0.0, dmn = 0.0, dmx = 0.0;
      uint64_t isum = 0;
      int64_t imn = 0, imx = 0, cnt = 0;
      bool have = false;
      for (int64_t j = lo + lane; j < hi; j += 64) {
        uint32_t r = perm ? perm[j] : (uint32_t)j;
        if (valid && !valid[r]) continue;
        ++cnt;
        if (is_i) {
          int64_t v = ci[r];
          sum += (double)v;
          isum += (uint64_t)v;
          if (!have || v < imn) imn = v;
          if (!have || v > imx) imx = v;
        } else {
          double v = cd[r];
          sum += v;
          if (!have || dlt(v, dmn)) dmn = v;
          if (!have || dlt(dmx, v)) dmx = v;
        }
        have = true;
      }
      for (int off = 32; off > 0; off >>= 1) {
        double s2 = __shfl_down(sum, off, 64);
        uint64_t u2 = __shfl_down(isum, off, 64);
        int64_t c2 = __shfl_down(cnt, off, 64);
        double a2 = __shfl_down(dmn, off, 64), b2 = __shfl_down(dmx, off, 64);
        int64_t p2 = __shfl_down(imn, off, 64), q2 = __shfl_down(imx, off, 64);
        int h2 = __shfl_down((int)have, off, 64);
        sum += s2;
        isum += u2;
        cnt += c2;
        if (h2) {
          if (is_i) {
            if (!have || p2 < imn) imn = p2;
            if (!have || q2 > imx) imx = q2;
          } else {
            if (!have || dlt(a2, dmn)) dmn = a2;
            if (!have || dlt(dmx, b2)) dmx = b2;
          }
          have = true;
        }
      }
      const int64_t n = __shfl(cnt, 0, 64);
      const double total = __shfl(sum, 0, 64);
      const double mean = n > 0 ? total / (double)n : 0.0;
      double acc = 0.0, lin = 0.0;
      if (n > 1) {
        for (int64_t j = lo + lane; j < hi; j += 64) {
          uint32_t r = perm ? perm[j] : (uint32_t)j;
          if (valid && !valid[r]) continue;
          double d = (is_i ? (double)ci[r] : cd[r]) - mean;
          acc += d * d;
          lin += d;
        }
        for (int off = 32; off > 0; off >>= 1) {
          acc += __shfl_down(acc, off, 64);
          lin += __shfl_down(lin, off, 64);
        }
      }
      if (lane == 0) {
        const int64_t at = (int64_t)k * cap + b;
        o.n[at] = n;
        o.mean[at] = mean;
        double sd = 0.0;
        if (n > 1) {
          double m2 = acc - lin * lin / (double)n;
          sd = m2 != m2 ? m2 : sqrt((m2 < 0.0 ? 0.0 : m2) / (double)(n - 1));
        }
        o.stddev[at] = sd;
        if (is_i) {
          o.mn[at] = n > 0 ? (uint64_t)imn : 0ull;
          o.mx[at] = n > 0 ? (uint64_t)imx : 0ull;
          o.sum[at] = n > 0 ? isum : 0ull;
        } else {
          o.mn[at] = n > 0 ? (uint64_t)__double_as_longlong(dmn) : 0ull;
          o.mx[at] = n > 0 ? (uint64_t)__double_as_longlong(dmx) : 0ull;
          o.sum[at] = n > 0 ? (uint64_t)__double_as_longlong(total) : 0ull;
        }
      }
    }
  }
}

__global__ __launch_bounds__(RB_THREADS) void bucket_big_stats_kernel(const uint32_t *bstart,
                                                                      const int64_t *count, int n_cols,
                                                                      ColPtrs cols, const BigAgg *part,
                                                                      int64_t cap, StatsOut o) {
  const int64_t nb = *count;
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; base < nb; base += waves * 64) {
    const int64_t mine = base + lane;
    uint64_t todo = __ballot(mine < nb && (int64_t)bstart[mine + 1] - (int64_t)bstart[mine] > BB_BIG);
    for (; todo; todo &= todo - 1) {
    const int64_t b = base + (__ffsll((long long)todo) - 1);
    const int64_t lo = bstart[b], hi = bstart[b + 1];
    for (int k = 0; k < n_cols; ++k) {
      const bool is_i = cols.is_i64[k];
      BigAgg a = agg_bucket(part, n_cols, k, is_i, lo, hi, lane);
      if (lane == 0) {
        const int64_t at = (int64_t)k * cap + b;
        const int64_t n = a.cnt;
        o.n[at] = n;
        o.mean[at] = n > 0 ? a.sum / (double)n : 0.0;
        double sd = 0.0;
        if (n > 1) sd = a.m2 != a.m2 ? a.m2 : sqrt((a.m2 < 0.0 ? 0.0 : a.m2) / (double)(n - 1));
        o.stddev[at] = sd;
        o.mn[at] = n > 0 ? a.mn : 0ull;
        o.mx[at] = n > 0 ? a.mx : 0ull;
        o.sum[at] = n > 0 ? (is_i ? a.isum : (uint64_t)__double_as_longlong(a.sum)) : 0ull;
      }
    }
    }
  }
}


struct RbPlan {
  OrderBuffers ob;
  uint32_t *block_counts;
  uint32_t *bstart;
  BigAgg *part;  // [tiles][2][n_cols] partial aggregates of big buckets
  int nblocks;
  // up to kMaxPay metric columns are delivered in sorted order by the ordering step (xs), their
  // validity as one byte per row (vs): the bucket kernels then read by position, front to back,
  // instead of fetching every value through a row id
  uint64_t *xs[kMaxPay];
  uint8_t *vs[kMaxPay];
  // mean over few (key, bucket) cells: the cells' row counts, value counts and sums (see dense_mean_kernel)
  uint32_t *d_rows;
  uint32_t *d_cnt;
  double *d_sum;
  StatTab st;  // withGroupedStats over few cells (see dense_stats_kernel); null pointers: not planned
  // floor / ceil over few cells (dense_pick_kernel): rows, coded extreme timestamp, ties, row id, flag
  uint32_t *k_rows, *k_ties, *k_row;
  uint64_t *k_ext;
  int32_t *k_tied;
};

// ------------------------------------------------------------------------------------------
// mean without ordering the rows.  The mean of a bucket does not depend on the order of its rows,
// and a table of (keys) x (buckets between the first and the last timestamp) cells is small when
// the feed covers hours or days in minute buckets: every row adds its values to its cell, and the
// non-empty cells, taken in (key, bucket) order, are the result.  Cells are reached through a
// hash table in LDS, one per block and flushed to the table in global memory when it fills up --
// a block works through a contiguous stretch of rows, which in a time-ordered feed touches a
// handful of cells.  The sums are double-precision atomic adds: their order, and with it the
// last bits of a mean, may differ from run to run.
// ------------------------------------------------------------------------------------------
constexpr int kDenseCols = 4;                  // metric columns per call
constexpr int64_t kDenseGroups = 1 << 20;      // cells of the table
constexpr int64_t kDenseKeys = 2048;           // keys: a stretch of a feed may touch every one of them

struct DenseTab {
  uint32_t *rows;  // [cells]
  uint32_t *cnt;   // [n_cols][cells]; columns with a NULL mask only (the others have `rows` values)
  double *sum;     // [n_cols][cells]
  int64_t cells;
};

__global__ __launch_bounds__(DH_THREADS) void dense_mean_kernel(const int32_t *key, const int64_t *ts, int64_t n,
                                                                int64_t dur, int64_t bmin, uint32_t per_key,
                                                                int n_cols, ColPtrs cols, DenseTab tab) {
  __shared__ uint32_t h_cell[DH_SLOTS];
  __shared__ uint32_t h_rows[DH_SLOTS];
  __shared__ uint32_t h_cnt[kDenseCols][DH_SLOTS];
  __shared__ double h_sum[kDenseCols][DH_SLOTS];
  const int tid = threadIdx.x;
  for (int i = tid; i < DH_SLOTS; i += DH_THREADS) {
    h_cell[i] = DH_EMPTY;
    h_rows[i] = 0;
#pragma unroll
    for (int c = 0; c < kDenseCols; ++c) { h_cnt[c][i] = 0; h_sum[c][i] = 0.0; }
  }
  __syncthreads();
  // the block's stretch of rows, in steps of DH_THREADS * DH_ROWS; the waves run through it on
  // their own (no barrier): a cell that finds no slot within DH_PROBES tries -- the stretch
  // touches more cells than the table holds -- is added to the global table directly
  constexpr int kStep = DH_THREADS * DH_ROWS;
  const int64_t steps = (n + kStep - 1) / kStep;
  const int64_t per_block = (steps + gridDim.x - 1) / gridDim.x;
  const int64_t s0 = (int64_t)blockIdx.x * per_block;
  const int64_t s1 = s0 + per_block < steps ? s0 + per_block : steps;
  for (int64_t st = s0; st < s1; ++st) {
    int64_t row[DH_ROWS];
    int64_t t[DH_ROWS];
    uint32_t k[DH_ROWS];
    double v[DH_ROWS][kDenseCols];
    bool ok[DH_ROWS][kDenseCols];
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      row[r] = st * kStep + r * DH_THREADS + tid;
      const bool live = row[r] < n;
      const int64_t at = live ? row[r] : n - 1;
      t[r] = ts[at];
      k[r] = key ? (uint32_t)key[at] : 0u;
#pragma unroll
      for (int c = 0; c < kDenseCols; ++c) {
        v[r][c] = 0.0;
        ok[r][c] = false;
        if (c >= n_cols) continue;
        const uint8_t *valid = cols.valid[c];
        ok[r][c] = live && (!valid || valid[at]);
        v[r][c] = cols.is_i64[c] ? (double)((const int64_t *)cols.col[c])[at] : ((const double *)cols.col[c])[at];
      }
    }
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      if (row[r] >= n) continue;
      uint32_t g;
      if (!dense_cell_of(k[r], floordiv_i64(t[r], dur) - bmin, per_key, tab.cells, &g)) continue;
      uint32_t slot = (g * 2654435761u) >> 22;  // top 10 bits
      bool found = false;
      for (int probe = 0; probe < DH_PROBES; ++probe) {
        const uint32_t old = atomicCAS(&h_cell[slot], DH_EMPTY, g);
        if (old == DH_EMPTY || old == g) { found = true; break; }
        slot = (slot + 1) & (DH_SLOTS - 1);
      }
      if (found) {
        atomicAdd(&h_rows[slot], 1u);
#pragma unroll
        for (int c = 0; c < kDenseCols; ++c) {
          if (c >= n_cols || !ok[r][c]) continue;
          if (cols.valid[c]) atomicAdd(&h_cnt[c][slot], 1u);  // without NULLs: as many values as rows
          atomicAdd(&h_sum[c][slot], v[r][c]);
        }
      } else {
        atomicAdd(&tab.rows[g], 1u);
#pragma unroll
        for (int c = 0; c < kDenseCols; ++c) {
          if (c >= n_cols || !ok[r][c]) continue;
          if (cols.valid[c]) atomicAdd(&tab.cnt[(size_t)c * tab.cells + g], 1u);
          atomicAdd(&tab.sum[(size_t)c * tab.cells + g], v[r][c]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < DH_SLOTS; i += DH_THREADS) {
    const uint32_t g = h_cell[i];
    if (g == DH_EMPTY) continue;
    atomicAdd(&tab.rows[g], h_rows[i]);
#pragma unroll
    for (int c = 0; c < kDenseCols; ++c) {
      if (c >= n_cols) continue;
      if (cols.valid[c]) {
        if (!h_cnt[c][i]) continue;
        atomicAdd(&tab.cnt[(size_t)c * tab.cells + g], h_cnt[c][i]);
      }
      atomicAdd(&tab.sum[(size_t)c * tab.cells + g], h_sum[c][i]);
    }
  }
}

// The non-empty cells in (key, bucket) order: one block, every thread a contiguous stretch of cells.
__global__ __launch_bounds__(1024) void dense_emit_kernel(DenseTab tab, int n_cols, uint32_t masked, uint32_t per_key, int64_t bmin,
                                                          int64_t dur, int64_t cap, int32_t *out_key,
                                                          int64_t *out_start, double *out_f64, int64_t *out_i64,
                                                          uint8_t *out_valid, int64_t *out_count) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (tab.cells + 1023) / 1024;
  const int64_t c0 = (int64_t)tid * per;
  const int64_t c1 = c0 + per < tab.cells ? c0 + per : tab.cells;
  uint32_t mine = 0;
  for (int64_t g = c0; g < c1; ++g) mine += tab.rows[g] != 0;
  uint32_t inc = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t at = inc - mine;
  uint32_t total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) at += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0) *out_count = (int64_t)total;
  for (int64_t g = c0; g < c1; ++g) {
    if (tab.rows[g] == 0) continue;
    out_key[at] = (int32_t)(g / per_key);
    out_start[at] = (bmin + (int64_t)(g % per_key)) * dur;
    for (int c = 0; c < n_cols; ++c) {
      // values of the cell: counted for a column with NULLs, else the cell's rows
      const uint32_t cnt = ((masked >> c) & 1u) ? tab.cnt[(size_t)c * tab.cells + g] : tab.rows[g];
      const int64_t o = (int64_t)c * cap + at;
      out_valid[o] = cnt > 0;
      out_f64[o] = cnt ? tab.sum[(size_t)c * tab.cells + g] / (double)cnt : 0.0;
      out_i64[o] = 0;
    }
    ++at;
  }
}

// resample(min / max) from the cells: the non-empty ones in (key, bucket) order, one block.
__global__ __launch_bounds__(1024) void dense_extreme_emit_kernel(StatTab tab, int n_cols, ColPtrs cols, int want_max,
                                                                  uint32_t per_key, int64_t bmin, int64_t dur,
                                                                  int64_t cap, int32_t *out_key, int64_t *out_start,
                                                                  double *out_f64, int64_t *out_i64,
                                                                  uint8_t *out_valid, int64_t *out_count) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (tab.cells + 1023) / 1024;
  const int64_t c0 = (int64_t)tid * per;
  const int64_t c1 = c0 + per < tab.cells ? c0 + per : tab.cells;
  uint32_t mine = 0;
  for (int64_t g = c0; g < c1; ++g) mine += tab.rows[g] != 0;
  uint32_t inc = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t at = inc - mine;
  uint32_t total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) at += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0) *out_count = (int64_t)total;
  for (int64_t g = c0; g < c1; ++g) {
    if (tab.rows[g] == 0) continue;
    out_key[at] = (int32_t)(g / per_key);
    out_start[at] = (bmin + (int64_t)(g % per_key)) * dur;
    for (int c = 0; c < n_cols; ++c) {
      const size_t i = (size_t)c * tab.cells + g;
      const bool have = tab.cnt[i] > 0;
      const uint64_t code = want_max ? tab.mx[i] : tab.mn[i];
      const int64_t w = (int64_t)c * cap + at;
      out_valid[w] = have;
      double f = 0.0;
      int64_t iv = 0;
      if (have) {
        if (cols.is_i64[c]) iv = (int64_t)(code ^ (1ull << 63));
        else f = __longlong_as_double((long long)decode_f64(code));
      }
      out_f64[w] = f;
      out_i64[w] = iv;
    }
    ++at;
  }
}

// The non-empty cells in (key, bucket) order with their statistics: one block.
__global__ __launch_bounds__(1024) void dense_stats_emit_kernel(StatTab tab, int n_cols, ColPtrs cols, uint32_t per_key,
                                                                int64_t bmin, int64_t dur, int64_t cap,
                                                                int32_t *out_key, int64_t *out_start, StatsOut o,
                                                                int64_t *out_count) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (tab.cells + 1023) / 1024;
  const int64_t c0 = (int64_t)tid * per;
  const int64_t c1 = c0 + per < tab.cells ? c0 + per : tab.cells;
  uint32_t mine = 0;
  for (int64_t g = c0; g < c1; ++g) mine += tab.rows[g] != 0;
  uint32_t inc = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t at = inc - mine;
  uint32_t total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) at += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0) *out_count = (int64_t)total;
  for (int64_t g = c0; g < c1; ++g) {
    if (tab.rows[g] == 0) continue;
    out_key[at] = (int32_t)(g / per_key);
    out_start[at] = (bmin + (int64_t)(g % per_key)) * dur;
    for (int c = 0; c < n_cols; ++c) {
      const size_t i = (size_t)c * tab.cells + g;
      const int64_t n = tab.cnt[i];
      const int64_t w = (int64_t)c * cap + at;
      o.n[w] = n;
      o.mean[w] = tab.sum[i];
      double sd = 0.0;
      if (n > 1) {
        const double m2 = tab.acc[i] - tab.lin[i] * tab.lin[i] / (double)n;
        sd = m2 != m2 ? m2 : sqrt((m2 < 0.0 ? 0.0 : m2) / (double)(n - 1));
      }
      o.stddev[w] = sd;
      if (cols.is_i64[c]) {
        o.mn[w] = n > 0 ? (tab.mn[i] ^ (1ull << 63)) : 0ull;
        o.mx[w] = n > 0 ? (tab.mx[i] ^ (1ull << 63)) : 0ull;
        o.sum[w] = n > 0 ? tab.isum[i] : 0ull;
      } else {
        o.mn[w] = n > 0 ? decode_f64(tab.mn[i]) : 0ull;
        o.mx[w] = n > 0 ? decode_f64(tab.mx[i]) : 0ull;
        o.sum[w] = n > 0 ? (uint64_t)__double_as_longlong(tab.total[i]) : 0ull;
      }
    }
    ++at;
  }
}

// ------------------------------------------------------------------------------------------
// floor / ceil without ordering the rows: the row with the earliest / latest timestamp of every
// (key, bucket) cell.  Sweep 0 takes the extreme timestamp per cell (integer atomics on its code),
// sweep 1 counts the rows that carry it and keeps the smallest of their row ids.  A cell in which
// several rows tie on that timestamp needs the struct comparison of their columns: the call then
// goes the ordered way (the caller reads the flag the emit kernel raises).
// ------------------------------------------------------------------------------------------
struct PickTab {
  uint32_t *rows;  // [cells]
  uint64_t *ext;   // [cells] coded extreme timestamp (floor: starts at all ones; ceil: at zero)
  uint32_t *ties;  // [cells] rows carrying it
  uint32_t *row;   // [cells] smallest row id among them (starts at all ones)
  int32_t *tied;   // device flag: some cell has more than one such row
  int64_t cells;
};

template <int SWEEP>
__global__ __launch_bounds__(DH_THREADS) void dense_pick_kernel(const int32_t *key, const int64_t *ts, int64_t n,
                                                                int64_t dur, int64_t bmin, uint32_t per_key,
                                                                int want_last, PickTab tab) {
  __shared__ uint32_t h_cell[DH_SLOTS];
  __shared__ uint32_t h_a[DH_SLOTS];   // sweep 0: rows; sweep 1: ties
  __shared__ uint32_t h_row[DH_SLOTS];
  __shared__ uint64_t h_ext[DH_SLOTS];
  const int tid = threadIdx.x;
  for (int i = tid; i < DH_SLOTS; i += DH_THREADS) {
    h_cell[i] = DH_EMPTY; h_a[i] = 0; h_row[i] = ~0u; h_ext[i] = want_last ? 0ull : ~0ull;
  }
  __syncthreads();
  constexpr int kStep = DH_THREADS * DH_ROWS;
  const int64_t steps = (n + kStep - 1) / kStep;
  const int64_t per_block = (steps + gridDim.x - 1) / gridDim.x;
  const int64_t s0 = (int64_t)blockIdx.x * per_block;
  const int64_t s1 = s0 + per_block < steps ? s0 + per_block : steps;
  for (int64_t st = s0; st < s1; ++st) {
    int64_t row[DH_ROWS], t[DH_ROWS];
    uint32_t k[DH_ROWS];
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      row[r] = st * kStep + r * DH_THREADS + tid;
      const int64_t at = row[r] < n ? row[r] : n - 1;
      t[r] = ts[at];
      k[r] = key ? (uint32_t)key[at] : 0u;
    }
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      if (row[r] >= n) continue;
      uint32_t g;
      if (!dense_cell_of(k[r], floordiv_i64(t[r], dur) - bmin, per_key, tab.cells, &g)) continue;
      const uint64_t code = code_i64(t[r]);
      if (SWEEP == 1 && tab.ext[g] != code) continue;
      uint32_t slot = (g * 2654435761u) >> 22;  // top 10 bits
      bool found = false;
      for (int probe = 0; probe < DH_PROBES; ++probe) {
        const uint32_t old = atomicCAS(&h_cell[slot], DH_EMPTY, g);
        if (old == DH_EMPTY || old == g) { found = true; break; }
        slot = (slot + 1) & (DH_SLOTS - 1);
      }
      if (SWEEP == 0) {
        if (found) {
          atomicAdd(&h_a[slot], 1u);
          if (want_last) atomicMax((unsigned long long *)&h_ext[slot], (unsigned long long)code);
          else atomicMin((unsigned long long *)&h_ext[slot], (unsigned long long)code);
        } else {
          atomicAdd(&tab.rows[g], 1u);
          if (want_last) atomicMax((unsigned long long *)&tab.ext[g], (unsigned long long)code);
          else atomicMin((unsigned long long *)&tab.ext[g], (unsigned long long)code);
        }
      } else {
        if (found) {
          atomicAdd(&h_a[slot], 1u);
          atomicMin(&h_row[slot], (uint32_t)row[r]);
        } else {
          atomicAdd(&tab.ties[g], 1u);
          atomicMin(&tab.row[g], (uint32_t)row[r]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < DH_SLOTS; i += DH_THREADS) {
    const uint32_t g = h_cell[i];
    if (g == DH_EMPTY || !h_a[i]) continue;
    if (SWEEP == 0) {
      atomicAdd(&tab.rows[g], h_a[i]);
      if (want_last) atomicMax((unsigned long long *)&tab.ext[g], (unsigned long long)h_ext[i]);
      else atomicMin((unsigned long long *)&tab.ext[g], (unsigned long long)h_ext[i]);
    } else {
      atomicAdd(&tab.ties[g], h_a[i]);
      atomicMin(&tab.row[g], h_row[i]);
    }
  }
}

// The non-empty cells in (key, bucket) order with the row picked for each: one block.
__global__ __launch_bounds__(1024) void dense_pick_emit_kernel(PickTab tab, int ties_matter, uint32_t per_key,
                                                               int64_t bmin, int64_t dur, int32_t *out_key,
                                                               int64_t *out_start, int32_t *out_row,
                                                               int64_t *out_count) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (tab.cells + 1023) / 1024;
  const int64_t c0 = (int64_t)tid * per;
  const int64_t c1 = c0 + per < tab.cells ? c0 + per : tab.cells;
  uint32_t mine = 0;
  for (int64_t g = c0; g < c1; ++g) mine += tab.rows[g] != 0;
  uint32_t inc = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t at = inc - mine;
  uint32_t total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) at += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0) *out_count = (int64_t)total;
  bool tied = false;
  for (int64_t g = c0; g < c1; ++g) {
    if (tab.rows[g] == 0) continue;
    out_key[at] = (int32_t)(g / per_key);
    out_start[at] = (bmin + (int64_t)(g % per_key)) * dur;
    out_row[at] = (int32_t)tab.row[g];
    tied = tied || tab.ties[g] > 1;
    ++at;
  }
  if (ties_matter && tied) *tab.tied = 1;
}

static bool dense_planned(int64_t n, int n_cols, int64_t n_keys, bool mean) {
  return mean && n >= dense_min_rows() && n_cols >= 1 && n_cols <= kDenseCols && n_keys <= kDenseKeys;
}

static void rb_plan(int64_t n, int n_cols, Workspace &ws, RbPlan *p, int64_t n_keys, bool travel = false,
                    bool mean = false, bool stats = false, bool pick = false) {
  const int n_travel = (travel && n_cols >= 1 && n_cols <= kMaxPay) ? n_cols : 0;
  order_plan(n, ws, &p->ob, n_keys, n_travel);
  p->d_rows = p->d_cnt = nullptr;
  p->d_sum = nullptr;
  if (dense_planned(n, n_cols, n_keys, mean)) {
    p->d_sum = ws.take<double>((size_t)kDenseGroups * n_cols);
    p->d_rows = ws.take<uint32_t>((size_t)kDenseGroups);
    p->d_cnt = ws.take<uint32_t>((size_t)kDenseGroups * n_cols);
  }
  p->k_rows = p->k_ties = p->k_row = nullptr;
  p->k_ext = nullptr;
  p->k_tied = nullptr;
  if (pick && n >= dense_min_rows() && n_keys <= kDenseKeys) {
    p->k_ext = ws.take<uint64_t>((size_t)kDenseGroups);
    p->k_rows = ws.take<uint32_t>((size_t)kDenseGroups);
    p->k_ties = ws.take<uint32_t>((size_t)kDenseGroups);
    p->k_row = ws.take<uint32_t>((size_t)kDenseGroups);
    p->k_tied = ws.take<int32_t>(1);
  }
  memset(&p->st, 0, sizeof(p->st));
  if (stats && n >= dense_min_rows() && n_cols >= 1 && n_cols <= kStatCols && n_keys <= kDenseKeys) {
    const size_t cells = (size_t)kDenseGroups, all = cells * (size_t)n_cols;
    p->st.sum = ws.take<double>(all);
    p->st.total = ws.take<double>(all);
    p->st.acc = ws.take<double>(all);
    p->st.lin = ws.take<double>(all);
    p->st.isum = ws.take<uint64_t>(all);
    p->st.mn = ws.take<uint64_t>(all);
    p->st.mx = ws.take<uint64_t>(all);
    p->st.rows = ws.take<uint32_t>(cells);
    p->st.cnt = ws.take<uint32_t>(all);
  }
  for (int c = 0; c < kMaxPay; ++c) {
    p->xs[c] = c < n_travel ? ws.take<uint64_t>((size_t)(n > 0 ? n : 1)) : nullptr;
    p->vs[c] = c < n_travel ? ws.take<uint8_t>((size_t)(n > 0 ? n : 1)) : nullptr;
  }
  size_t cnt = (size_t)(n > 0 ? n : 1);
  p->nblocks = (int)((cnt + RB_ROWS - 1) / RB_ROWS);
  p->block_counts = ws.take<uint32_t>((size_t)p->nblocks);
  p->bstart = ws.take<uint32_t>(cnt + 1);
  size_t tiles = (cnt + BB_TILE - 1) / BB_TILE;
  p->part = ws.take<BigAgg>(tiles * 2 * (size_t)(n_cols > 0 ? n_cols : 1));
}

// Orders the rows for the bucket kernels.  Where the plan has room for them (p.xs) the metric
// columns come out in sorted order too (order_columns): *perm is then null (row j is at position
// j) and cp points at the sorted copies and validity bytes.  Otherwise *perm is the row-id
// permutation and cp the caller's columns.
static int rb_order(const OrderSpec &s0, RbPlan &p, int n_cols, Layout *lay, OrderResult *res, ColPtrs *cp,
                    const uint32_t **perm, hipStream_t st, const RangeInfo *known = nullptr) {
  const uint64_t *src[kMaxPay];
  const uint8_t *val[kMaxPay];
  const bool fits = n_cols >= 1 && n_cols <= kMaxPay;
  for (int k = 0; k < kMaxPay; ++k) {
    src[k] = (fits && k < n_cols) ? (const uint64_t *)cp->col[k] : nullptr;
    val[k] = (fits && k < n_cols) ? cp->valid[k] : nullptr;
  }
  bool traveled = false;
  TMP_TRY(order_columns(s0, p.ob, fits ? n_cols : 0, src, val, p.xs, p.vs, lay, res, &traveled, st, known));
  *perm = res->perm;
  if (traveled) {
    *perm = nullptr;
    for (int k = 0; k < n_cols; ++k) {
      cp->col[k] = p.xs[k];
      if (cp->valid[k]) cp->valid[k] = p.vs[k];
    }
  }
  return TMP_OK;
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_resample_workspace_bytes(const tmp_resample_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RbPlan p;
  const bool reduce = a->func == TMP_FUNC_MIN || a->func == TMP_FUNC_MAX || a->func == TMP_FUNC_MEAN;
  rb_plan(a->n, a->n_cols, ws, &p, plan_keys(a->key, a->n_keys), reduce, a->func == TMP_FUNC_MEAN,
          a->func == TMP_FUNC_MIN || a->func == TMP_FUNC_MAX, !reduce);
  return ws.off;
}

extern "C" int tmp_resample(const tmp_resample_args *a, void *workspace, size_t workspace_bytes,
                            void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_resample: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31) - 1, TMP_EINVAL, "tmp_resample: row count must be in [0, 2^31-1)");
  TMP_REQUIRE(a->dur_us > 0, TMP_EINVAL, "tmp_resample: dur_us must be positive");
  TMP_REQUIRE(a->func >= TMP_FUNC_FLOOR && a->func <= TMP_FUNC_CEIL, TMP_EINVAL, "tmp_resample: bad func");
  TMP_REQUIRE(a->n_cols >= 0 && a->n_cols <= TMP_MAX_COLS, TMP_EINVAL, "tmp_resample: n_cols must be in [0, %d]",
              TMP_MAX_COLS);
  TMP_REQUIRE(a->out_count, TMP_EINVAL, "tmp_resample: out_count is null");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_count, 0, sizeof(int64_t), st));
  if (a->n == 0) return TMP_OK;
  bool reduce = a->func == TMP_FUNC_MIN || a->func == TMP_FUNC_MAX || a->func == TMP_FUNC_MEAN;
  TMP_REQUIRE(a->ts && a->out_key && a->out_start, TMP_EINVAL, "tmp_resample: null pointer");
  TMP_REQUIRE(!reduce || a->n_cols == 0 || (a->out_f64 && a->out_i64 && a->out_valid), TMP_EINVAL,
              "tmp_resample: reduce outputs missing");
  TMP_REQUIRE(reduce || a->out_row, TMP_EINVAL, "tmp_resample: out_row is null");
  TMP_REQUIRE(a->n_cols == 0 || (a->cols && a->col_is_i64 && a->col_valid), TMP_EINVAL,
              "tmp_resample: column arrays missing");
  Workspace ws(workspace, workspace_bytes);
  RbPlan p;
  rb_plan(a->n, a->n_cols, ws, &p, plan_keys(a->key, a->n_keys), reduce, a->func == TMP_FUNC_MEAN,
          a->func == TMP_FUNC_MIN || a->func == TMP_FUNC_MAX, !reduce);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_resample: workspace too small");
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts; s.coarse_ok = 1;
  // min / max / mean do not depend on the order inside a bucket: the rows are ordered by
  // (key, bucket index) only -- far fewer bits to sort, and a feed that is in time order up to
  // jitter inside a bucket needs passes over the key bits alone.  floor / ceil keep the order by ts.
  const int64_t dur = reduce ? 1 : a->dur_us, unit = reduce ? a->dur_us : 1;
  s.prim_div = reduce ? a->dur_us : 1;
  Layout lay;
  OrderResult res;
  ColPtrs cp;
  memset(&cp, 0, sizeof(cp));
  for (int k = 0; k < a->n_cols; ++k) {
    TMP_REQUIRE(a->cols[k] != nullptr, TMP_EINVAL, "tmp_resample: column %d is null", k);
    cp.col[k] = a->cols[k];
    cp.valid[k] = a->col_valid[k];
    cp.is_i64[k] = a->col_is_i64[k] ? 1 : 0;
  }
  const uint32_t *perm = nullptr;
  RangeInfo range;
  const RangeInfo *known = nullptr;
  if (p.k_rows) {
    // floor / ceil over few cells: two sweeps over the rows as they lie, unless rows tie on a cell's
    // extreme timestamp (their columns then decide, which the ordered path works out)
    TMP_TRY(order_scan_ranges(s, p.ob, st));
    TMP_TRY(order_read_ranges(p.ob, &range, st));
    known = &range;
    const int64_t nk = plan_keys(a->key, a->n_keys);
    // floor / ceil order by ts itself: the buckets of the first and the last timestamp
    auto fdiv = [](int64_t x, int64_t d) { int64_t q = x / d; return (x % d != 0 && x < 0) ? q - 1 : q; };  // d > 0
    const int64_t b_lo = fdiv(range.prim_min, a->dur_us), b_hi = fdiv(range.prim_max, a->dur_us);
    const uint64_t per_key = (uint64_t)b_hi - (uint64_t)b_lo + 1;
    const bool local = sorted_mode(range) != 0 || (uint64_t)nk * per_key <= DH_SLOTS / 2;
    if (per_key >= 1 && per_key <= (uint64_t)kDenseGroups && (uint64_t)nk * per_key <= (uint64_t)kDenseGroups && local &&
        !getenv("TEMPO_NO_DENSE_PICK")) {
      PickTab tab;
      tab.cells = (int64_t)((uint64_t)nk * per_key);
      tab.rows = p.k_rows; tab.ext = p.k_ext; tab.ties = p.k_ties; tab.row = p.k_row; tab.tied = p.k_tied;
      const int want_last = a->func == TMP_FUNC_CEIL ? 1 : 0;
      TMP_CHECK_HIP(hipMemsetAsync(tab.rows, 0, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.ties, 0, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.row, 0xff, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.ext, want_last ? 0 : 0xff, sizeof(uint64_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.tied, 0, sizeof(int32_t), st));
      const int grid = grid_for(a->n, DH_THREADS * DH_ROWS * 8, 2048);
      TMP_LAUNCH(dense_pick_kernel<0>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, a->dur_us, b_lo,
                 (uint32_t)per_key, want_last, tab);
      TMP_LAUNCH(dense_pick_kernel<1>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, a->dur_us, b_lo,
                 (uint32_t)per_key, want_last, tab);
      TMP_LAUNCH(dense_pick_emit_kernel, dim3(1), dim3(1024), st, tab, a->n_cols > 0 ? 1 : 0, (uint32_t)per_key, b_lo,
                 a->dur_us, a->out_key, a->out_start, a->out_row, a->out_count);
      int32_t tied = 0;
      TMP_CHECK_HIP(hipMemcpyAsync(&tied, tab.tied, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      TMP_CHECK_HIP(hipStreamSynchronize(st));
      if (!tied) return TMP_OK;
    }
  }
  if (p.st.rows) {
    // min / max over few cells: one sweep over the rows as they lie (see dense_stats_kernel)
    if (!known) {
      TMP_TRY(order_scan_ranges(s, p.ob, st));
      TMP_TRY(order_read_ranges(p.ob, &range, st));
      known = &range;
    }
    const int64_t nk = plan_keys(a->key, a->n_keys);
    const uint64_t per_key = (uint64_t)range.prim_max - (uint64_t)range.prim_min + 1;
    const bool local = sorted_mode(range) != 0 || (uint64_t)nk * per_key <= DS_SLOTS / 2;
    if (per_key >= 1 && per_key <= (uint64_t)kDenseGroups && (uint64_t)nk * per_key <= (uint64_t)kDenseGroups && local &&
        !getenv("TEMPO_NO_DENSE_STATS")) {
      StatTab tab = p.st;
      tab.cells = (int64_t)((uint64_t)nk * per_key);
      const size_t all = (size_t)tab.cells * a->n_cols;
      TMP_CHECK_HIP(hipMemsetAsync(tab.rows, 0, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.cnt, 0, sizeof(uint32_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.sum, 0, sizeof(double) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.isum, 0, sizeof(uint64_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.mx, 0, sizeof(uint64_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.mn, 0xff, sizeof(uint64_t) * all, st));
      TMP_LAUNCH(dense_stats_kernel<0>, dim3(grid_for(a->n, DH_THREADS * DH_ROWS * 8, 2048)), dim3(DH_THREADS), st,
                 a->key, a->ts, a->n, a->dur_us, range.prim_min, (uint32_t)per_key, a->n_cols, cp, tab);
      TMP_LAUNCH(dense_extreme_emit_kernel, dim3(1), dim3(1024), st, tab, a->n_cols, cp,
                 a->func == TMP_FUNC_MAX ? 1 : 0, (uint32_t)per_key, range.prim_min, a->dur_us, a->n, a->out_key,
                 a->out_start, a->out_f64, a->out_i64, a->out_valid, a->out_count);
      TMP_CHECK_HIP(hipGetLastError());
      return TMP_OK;
    }
  }
  if (p.d_rows) {
    // mean over few cells: no ordering at all (the ranges are read either way)
    if (!known) {
      TMP_TRY(order_scan_ranges(s, p.ob, st));
      TMP_TRY(order_read_ranges(p.ob, &range, st));
      known = &range;
    }
    const int64_t nk = plan_keys(a->key, a->n_keys);
    const uint64_t per_key = (uint64_t)range.prim_max - (uint64_t)range.prim_min + 1;  // buckets first .. last
    // rows that lie close together should fall into few cells (what overflows a block's table is
    // added to the global one row by row): a feed in some time order -- or so few cells that a
    // block's table holds them all
    const bool local = sorted_mode(range) != 0 || (uint64_t)nk * per_key <= DH_SLOTS / 2;
    if (per_key >= 1 && per_key <= (uint64_t)kDenseGroups && (uint64_t)nk * per_key <= (uint64_t)kDenseGroups && local &&
        !getenv("TEMPO_NO_DENSE_MEAN")) {
      DenseTab tab;
      tab.cells = (int64_t)((uint64_t)nk * per_key);
      tab.rows = p.d_rows; tab.cnt = p.d_cnt; tab.sum = p.d_sum;
      TMP_CHECK_HIP(hipMemsetAsync(tab.rows, 0, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.cnt, 0, sizeof(uint32_t) * tab.cells * a->n_cols, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.sum, 0, sizeof(double) * tab.cells * a->n_cols, st));
      TMP_LAUNCH(dense_mean_kernel, dim3(grid_for(a->n, DH_THREADS * DH_ROWS * 8, 2048)), dim3(DH_THREADS), st,
                 a->key, a->ts, a->n, a->dur_us, range.prim_min, (uint32_t)per_key, a->n_cols, cp, tab);
      uint32_t masked = 0;
      for (int k = 0; k < a->n_cols; ++k)
        if (cp.valid[k]) masked |= 1u << k;
      TMP_LAUNCH(dense_emit_kernel, dim3(1), dim3(1024), st, tab, a->n_cols, masked, (uint32_t)per_key, range.prim_min,
                 a->dur_us, a->n, a->out_key, a->out_start, a->out_f64, a->out_i64, a->out_valid, a->out_count);
      TMP_CHECK_HIP(hipGetLastError());
      return TMP_OK;
    }
  }
  TMP_TRY(rb_order(s, p, a->n_cols, &lay, &res, &cp, &perm, st, known));
  TMP_LAUNCH(bucket_count_kernel, dim3(p.nblocks), dim3(RB_THREADS), st, res.comp, a->n, lay, dur,
             p.block_counts);
  TMP_LAUNCH(bucket_offsets_kernel, dim3(1), dim3(1024), st, p.block_counts, p.nblocks, a->out_count,
             p.bstart, a->n);
  TMP_LAUNCH(bucket_starts_kernel, dim3(p.nblocks), dim3(RB_THREADS), st, res.comp, a->n, lay, dur, unit,
             p.block_counts, p.bstart, a->out_key, a->out_start);
  if (reduce) {
    if (a->n_cols > 0) {
      TMP_LAUNCH(bucket_reduce_kernel, dim3(grid_for(a->n, RB_THREADS, 8192)), dim3(RB_THREADS), st, perm,
                 p.bstart, a->out_count, a->func, a->n_cols, cp, a->n, a->out_f64, a->out_i64, a->out_valid,
                 (int64_t)(a->key && a->n_keys > 1 ? a->n_keys : 1));
      if (a->n > BB_BIG) {  // only then can a bucket be a big one
        TMP_LAUNCH(bucket_big_tiles_kernel, dim3(grid_for(a->n, BB_TILE * (RB_THREADS / 64), 1 << 16)),
                   dim3(RB_THREADS), st, perm, p.bstart, a->out_count, a->n, a->n_cols, cp, p.part,
                   (int64_t)(a->key && a->n_keys > 1 ? a->n_keys : 1));
        TMP_LAUNCH(bucket_big_reduce_kernel, dim3(grid_for(a->n, BB_BIG * (RB_THREADS / 64), 4096)),
                   dim3(RB_THREADS), st, p.bstart, a->out_count, a->func, a->n_cols, cp, p.part, a->n,
                   a->out_f64, a->out_i64, a->out_valid);
      }
    }
  } else {
    TMP_LAUNCH(bucket_pick_kernel, dim3(grid_for(a->n, RB_THREADS * 4, 8192)), dim3(RB_THREADS), st, res.comp,
               res.perm, lay, p.bstart, a->out_count, a->func, a->n_cols, cp, a->out_row);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

extern "C" size_t tmp_grouped_stats_workspace_bytes(const tmp_grouped_stats_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RbPlan p;
  rb_plan(a->n, a->n_cols, ws, &p, plan_keys(a->key, a->n_keys), true, false, true);
  return ws.off;
}

extern "C" int tmp_grouped_stats(const tmp_grouped_stats_args *a, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_grouped_stats: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31) - 1, TMP_EINVAL,
              "tmp_grouped_stats: row count must be in [0, 2^31-1)");
  TMP_REQUIRE(a->dur_us > 0, TMP_EINVAL, "tmp_grouped_stats: dur_us must be positive");
  TMP_REQUIRE(a->n_cols >= 0 && a->n_cols <= TMP_MAX_COLS, TMP_EINVAL,
              "tmp_grouped_stats: n_cols must be in [0, %d]", TMP_MAX_COLS);
  TMP_REQUIRE(a->out_count, TMP_EINVAL, "tmp_grouped_stats: out_count is null");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_count, 0, sizeof(int64_t), st));
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->out_key && a->out_start, TMP_EINVAL, "tmp_grouped_stats: null pointer");
  TMP_REQUIRE(a->n_cols == 0 || (a->cols && a->col_is_i64 && a->col_valid && a->out_mean && a->out_n &&
                                 a->out_min && a->out_max && a->out_sum && a->out_stddev),
              TMP_EINVAL, "tmp_grouped_stats: column or output arrays missing");
  Workspace ws(workspace, workspace_bytes);
  RbPlan p;
  rb_plan(a->n, a->n_cols, ws, &p, plan_keys(a->key, a->n_keys), true, false, true);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_grouped_stats: workspace too small");
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts; s.coarse_ok = 1;
  s.prim_div = a->dur_us;  // ordered by (key, bucket index): the statistics ignore the order inside a bucket
  Layout lay;
  OrderResult res;
  ColPtrs cp;
  memset(&cp, 0, sizeof(cp));
  for (int k = 0; k < a->n_cols; ++k) {
    TMP_REQUIRE(a->cols[k] != nullptr, TMP_EINVAL, "tmp_grouped_stats: column %d is null", k);
    cp.col[k] = a->cols[k];
    cp.valid[k] = a->col_valid[k];
    cp.is_i64[k] = a->col_is_i64[k] ? 1 : 0;
  }
  const uint32_t *perm = nullptr;
  RangeInfo range;
  const RangeInfo *known = nullptr;
  if (p.st.rows) {
    // few (key, bucket) cells and a feed in some time order: two sweeps over the rows as they lie
    TMP_TRY(order_scan_ranges(s, p.ob, st));
    TMP_TRY(order_read_ranges(p.ob, &range, st));
    known = &range;
    const int64_t nk = plan_keys(a->key, a->n_keys);
    const uint64_t per_key = (uint64_t)range.prim_max - (uint64_t)range.prim_min + 1;
    const bool local = sorted_mode(range) != 0 || (uint64_t)nk * per_key <= DS_SLOTS / 2;
    if (per_key >= 1 && per_key <= (uint64_t)kDenseGroups && (uint64_t)nk * per_key <= (uint64_t)kDenseGroups && local &&
        !getenv("TEMPO_NO_DENSE_STATS")) {
      StatTab tab = p.st;
      tab.cells = (int64_t)((uint64_t)nk * per_key);
      const size_t all = (size_t)tab.cells * a->n_cols;
      TMP_CHECK_HIP(hipMemsetAsync(tab.rows, 0, sizeof(uint32_t) * tab.cells, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.cnt, 0, sizeof(uint32_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.sum, 0, sizeof(double) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.acc, 0, sizeof(double) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.lin, 0, sizeof(double) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.isum, 0, sizeof(uint64_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.mx, 0, sizeof(uint64_t) * all, st));
      TMP_CHECK_HIP(hipMemsetAsync(tab.mn, 0xff, sizeof(uint64_t) * all, st));
      const int grid = grid_for(a->n, DH_THREADS * DH_ROWS * 8, 2048);
      StatsOut o;
      o.mean = a->out_mean; o.n = a->out_n; o.mn = (uint64_t *)a->out_min; o.mx = (uint64_t *)a->out_max;
      o.sum = (uint64_t *)a->out_sum; o.stddev = a->out_stddev;
      TMP_LAUNCH(dense_stats_kernel<0>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, a->dur_us,
                 range.prim_min, (uint32_t)per_key, a->n_cols, cp, tab);
      TMP_LAUNCH(dense_stats_means_kernel, dim3(grid_for((int64_t)all, kBlock, 4096)), dim3(kBlock), st, tab, a->n_cols);
      TMP_LAUNCH(dense_stats_kernel<1>, dim3(grid), dim3(DH_THREADS), st, a->key, a->ts, a->n, a->dur_us,
                 range.prim_min, (uint32_t)per_key, a->n_cols, cp, tab);
      TMP_LAUNCH(dense_stats_emit_kernel, dim3(1), dim3(1024), st, tab, a->n_cols, cp, (uint32_t)per_key,
                 range.prim_min, a->dur_us, a->n, a->out_key, a->out_start, o, a->out_count);
      TMP_CHECK_HIP(hipGetLastError());
      return TMP_OK;
    }
  }
  TMP_TRY(rb_order(s, p, a->n_cols, &lay, &res, &cp, &perm, st, known));
  TMP_LAUNCH(bucket_count_kernel, dim3(p.nblocks), dim3(RB_THREADS), st, res.comp, a->n, lay, (int64_t)1,
             p.block_counts);
  TMP_LAUNCH(bucket_offsets_kernel, dim3(1), dim3(1024), st, p.block_counts, p.nblocks, a->out_count,
             p.bstart, a->n);
  TMP_LAUNCH(bucket_starts_kernel, dim3(p.nblocks), dim3(RB_THREADS), st, res.comp, a->n, lay, (int64_t)1,
             a->dur_us, p.block_counts, p.bstart, a->out_key, a->out_start);
  if (a->n_cols > 0) {
    StatsOut o;
    o.mean = a->out_mean; o.n = a->out_n; o.mn = (uint64_t *)a->out_min; o.mx = (uint64_t *)a->out_max;
    o.sum = (uint64_t *)a->out_sum; o.stddev = a->out_stddev;
    TMP_LAUNCH(bucket_stats_kernel, dim3(grid_for(a->n, RB_THREADS, 8192)), dim3(RB_THREADS), st, perm,
               p.bstart, a->out_count, a->n_cols, cp, a->n, o, (int64_t)(a->key && a->n_keys > 1 ? a->n_keys : 1));
    if (a->n > BB_BIG) {
      TMP_LAUNCH(bucket_big_tiles_kernel, dim3(grid_for(a->n, BB_TILE * (RB_THREADS / 64), 1 << 16)),
                 dim3(RB_THREADS), st, perm, p.bstart, a->out_count, a->n, a->n_cols, cp, p.part,
                   (int64_t)(a->key && a->n_keys > 1 ? a->n_keys : 1));
      TMP_LAUNCH(bucket_big_stats_kernel, dim3(grid_for(a->n, BB_BIG * (RB_THREADS / 64), 4096)),
                 dim3(RB_THREADS), st, p.bstart, a->out_count, a->n_cols, cp, p.part, a->n, o);
    }
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
