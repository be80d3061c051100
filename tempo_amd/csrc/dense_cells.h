// Aggregation of rows into a table of (key, time bucket) cells without ordering them, shared by
// the bucket operators (resample.hip) and the rolling statistics (rangestats_cells.hip).
#pragma once

#include <stdlib.h>

#include "common.h"

namespace tmp {

// Cell of a row: key id x buckets per key + the bucket's offset from the table's first bucket, formed
// in 64 bits.  A key id outside the dictionary is read as the last key -- the rule of every sweep
// over cells, so that all of them put such a row into the same cell; a bucket outside the table's
// range (there is none on well-formed input) has no cell.
__device__ __forceinline__ bool dense_cell_of(uint32_t key, int64_t bucket_off, uint32_t per_key, int64_t cells,
                                              uint32_t *cell) {
  if (bucket_off < 0 || bucket_off >= (int64_t)per_key) return false;
  const uint64_t n_keys = (uint64_t)cells / per_key;
  const uint64_t k = (uint64_t)key < n_keys ? (uint64_t)key : n_keys - 1;
  const uint64_t g = k * per_key + (uint64_t)bucket_off;
  if (g >= (uint64_t)cells) return false;
  *cell = (uint32_t)g;
  return true;
}


struct ColPtrs {
  const void *col[TMP_MAX_COLS];
  const uint8_t *valid[TMP_MAX_COLS];
  uint8_t is_i64[TMP_MAX_COLS];
};

constexpr int DH_SLOTS = 1024;                 // slots of a block's hash table
constexpr int DH_THREADS = 1024;
constexpr int DH_ROWS = 4;                     // rows per thread and step
constexpr int DH_PROBES = 16;                  // slots tried before a row goes to the global table itself
constexpr uint32_t DH_EMPTY = 0xffffffffu;
// Tables below this many rows take the ordered paths (TEMPO_DENSE_MIN_ROWS overrides: tests).
static int64_t dense_min_rows() {
  const char *e = getenv("TEMPO_DENSE_MIN_ROWS");
  return e ? (int64_t)atoll(e) : (int64_t)(1 << 20);
}

// Cells of withGroupedStats without ordering (dense_stats_kernel).
struct StatTab {
  uint32_t *rows;   // [cells]
  uint32_t *cnt;    // [n_cols][cells] values that are not NULL
  double *sum;      // [n_cols][cells] sum of the values as doubles; after the mean pass: the mean
  uint64_t *isum;   // [n_cols][cells] exact sum (integer columns)
  uint64_t *mn;     // [n_cols][cells] coded minimum (starts at all ones)
  uint64_t *mx;     // [n_cols][cells] coded maximum (starts at zero)
  double *total;    // [n_cols][cells] the sum, kept when `sum` becomes the mean
  double *acc;      // [n_cols][cells] sum of squared deviations from the mean
  double *lin;      // [n_cols][cells] sum of deviations
  int64_t cells;
  int32_t *nonfinite;  // optional device flag: sweep 0 raises it on a NaN or infinite double value
};

// ------------------------------------------------------------------------------------------
// withGroupedStats without ordering the rows: the same table of cells, two sweeps over the rows.
// Sweep A adds every row to its cell's count, sum, exact integer sum, minimum and maximum (the
// extremes as order-preserving 64-bit codes, so that they are integer atomics); a pass over the
// cells turns the sums into means; sweep B adds the deviations from the cell's mean and their
// squares (the corrected two-pass form of the ordered path).
// ------------------------------------------------------------------------------------------
constexpr int kStatCols = 2;
constexpr int DS_SLOTS = 512;

// Spark's ordering of doubles as unsigned integers: -inf < ... < -0.0 < 0.0 < ... < inf < NaN.
__device__ __forceinline__ uint64_t code_f64(double v) {
  if (v != v) return ~0ull;
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | (1ull << 63));
}
__device__ __forceinline__ uint64_t decode_f64(uint64_t e) {  // the double's bits
  if (e == ~0ull) return 0x7ff8000000000000ull;
  return (e >> 63) ? (e & ~(1ull << 63)) : ~e;
}
__device__ __forceinline__ uint64_t code_i64(int64_t v) { return (uint64_t)v ^ (1ull << 63); }


// SWEEP 0: counts, sums, extremes.  SWEEP 1: deviations from tab.sum (the means by then).
template <int SWEEP>
__global__ __launch_bounds__(DH_THREADS) void dense_stats_kernel(const int32_t *key, const int64_t *ts, int64_t n,
                                                                 int64_t dur, int64_t bmin, uint32_t per_key,
                                                                 int n_cols, ColPtrs cols, StatTab tab) {
  __shared__ uint32_t h_cell[DS_SLOTS];
  __shared__ uint32_t h_rows[DS_SLOTS];
  __shared__ uint32_t h_cnt[kStatCols][DS_SLOTS];
  __shared__ double h_a[kStatCols][DS_SLOTS];    // sweep 0: sum; sweep 1: squared deviations
  __shared__ double h_b[kStatCols][DS_SLOTS];    // sweep 1: deviations
  __shared__ uint64_t h_i[kStatCols][DS_SLOTS];  // sweep 0: exact integer sum
  __shared__ uint64_t h_mn[kStatCols][DS_SLOTS];
  __shared__ uint64_t h_mx[kStatCols][DS_SLOTS];
  const int tid = threadIdx.x;
  for (int i = tid; i < DS_SLOTS; i += DH_THREADS) {
    h_cell[i] = DH_EMPTY;
    h_rows[i] = 0;
#pragma unroll
    for (int c = 0; c < kStatCols; ++c) {
      h_cnt[c][i] = 0; h_a[c][i] = 0.0; h_b[c][i] = 0.0; h_i[c][i] = 0; h_mn[c][i] = ~0ull; h_mx[c][i] = 0ull;
    }
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
    uint64_t raw[DH_ROWS][kStatCols];
    bool ok[DH_ROWS][kStatCols];
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      row[r] = st * kStep + r * DH_THREADS + tid;
      const bool live = row[r] < n;
      const int64_t at = live ? row[r] : n - 1;
      t[r] = ts[at];
      k[r] = key ? (uint32_t)key[at] : 0u;
#pragma unroll
      for (int c = 0; c < kStatCols; ++c) {
        raw[r][c] = 0;
        ok[r][c] = false;
        if (c >= n_cols) continue;
        const uint8_t *valid = cols.valid[c];
        ok[r][c] = live && (!valid || valid[at]);
        raw[r][c] = ((const uint64_t *)cols.col[c])[at];
      }
    }
#pragma unroll
    for (int r = 0; r < DH_ROWS; ++r) {
      if (row[r] >= n) continue;
      uint32_t g;
      if (!dense_cell_of(k[r], floordiv_i64(t[r], dur) - bmin, per_key, tab.cells, &g)) continue;
      uint32_t slot = (g * 2654435761u) >> 23;  // top 9 bits
      bool found = false;
      for (int probe = 0; probe < DH_PROBES; ++probe) {
        const uint32_t old = atomicCAS(&h_cell[slot], DH_EMPTY, g);
        if (old == DH_EMPTY || old == g) { found = true; break; }
        slot = (slot + 1) & (DS_SLOTS - 1);
      }
      if (SWEEP == 0) {
        if (found) atomicAdd(&h_rows[slot], 1u);
        else atomicAdd(&tab.rows[g], 1u);
      }
#pragma unroll
      for (int c = 0; c < kStatCols; ++c) {
        if (c >= n_cols || !ok[r][c]) continue;
        const bool is_i = cols.is_i64[c] != 0;
        const double v = is_i ? (double)(int64_t)raw[r][c] : __longlong_as_double((long long)raw[r][c]);
        const size_t at = (size_t)c * tab.cells + g;
        if (SWEEP == 0) {
          if (tab.nonfinite && !is_i && !(__builtin_fabs(v) < 1e150) &&  // NaN, infinite, or large enough for squares to overflow
              __atomic_load_n(tab.nonfinite, __ATOMIC_RELAXED) == 0)
            atomicOr(tab.nonfinite, 1);
          const uint64_t code = is_i ? code_i64((int64_t)raw[r][c]) : code_f64(v);
          if (found) {
            atomicAdd(&h_cnt[c][slot], 1u);
            atomicAdd(&h_a[c][slot], v);
            if (is_i) atomicAdd((unsigned long long *)&h_i[c][slot], (unsigned long long)raw[r][c]);
            atomicMin((unsigned long long *)&h_mn[c][slot], (unsigned long long)code);
            atomicMax((unsigned long long *)&h_mx[c][slot], (unsigned long long)code);
          } else {
            atomicAdd(&tab.cnt[at], 1u);
            atomicAdd(&tab.sum[at], v);
            if (is_i) atomicAdd((unsigned long long *)&tab.isum[at], (unsigned long long)raw[r][c]);
            atomicMin((unsigned long long *)&tab.mn[at], (unsigned long long)code);
            atomicMax((unsigned long long *)&tab.mx[at], (unsigned long long)code);
          }
        } else {
          const double d = v - tab.sum[at];  // the cell's mean
          if (found) {
            atomicAdd(&h_a[c][slot], d * d);
            atomicAdd(&h_b[c][slot], d);
          } else {
            atomicAdd(&tab.acc[at], d * d);
            atomicAdd(&tab.lin[at], d);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < DS_SLOTS; i += DH_THREADS) {
    const uint32_t g = h_cell[i];
    if (g == DH_EMPTY) continue;
    if (SWEEP == 0 && h_rows[i]) atomicAdd(&tab.rows[g], h_rows[i]);
#pragma unroll
    for (int c = 0; c < kStatCols; ++c) {
      if (c >= n_cols) continue;
      const size_t at = (size_t)c * tab.cells + g;
      if (SWEEP == 0) {
        if (!h_cnt[c][i]) continue;
        atomicAdd(&tab.cnt[at], h_cnt[c][i]);
        atomicAdd(&tab.sum[at], h_a[c][i]);
        if (cols.is_i64[c]) atomicAdd((unsigned long long *)&tab.isum[at], (unsigned long long)h_i[c][i]);
        atomicMin((unsigned long long *)&tab.mn[at], (unsigned long long)h_mn[c][i]);
        atomicMax((unsigned long long *)&tab.mx[at], (unsigned long long)h_mx[c][i]);
      } else {
        atomicAdd(&tab.acc[at], h_a[c][i]);
        atomicAdd(&tab.lin[at], h_b[c][i]);
      }
    }
  }
}

// Between the sweeps: the cells' sums become their means (the sums are kept beside them).
static __global__ __launch_bounds__(kBlock) void dense_stats_means_kernel(StatTab tab, int n_cols) {
  const int64_t total = tab.cells * n_cols;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const uint32_t cnt = tab.cnt[i];
    tab.total[i] = tab.sum[i];
    tab.sum[i] = cnt ? tab.sum[i] / (double)cnt : 0.0;
  }
}

}  // namespace tmp
