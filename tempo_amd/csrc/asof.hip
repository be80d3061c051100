// AS-OF join match.  Replaces the union + running last(col, ignorenulls) window of
// python/tempo/tsdf.py:463-560 / :111-162 and the bracket rule of :164-190.
//
// Both sides are ordered independently by (key, ts[, seq]) into 64-bit sort words that share
// one bit layout; the per-column validity bits of the right side ride in the low bits of its
// sort words.  Then:
//   1. a wave-ballot prefix scan over the ordered right side gives, for every position and
//      mask, the last position at or before it whose mask bit is set;
//   2. every left row binary-searches its position in the ordered right side (search range
//      narrowed per block), applies the key / bracket visibility tests and scatters the
//      matched right row ids back to the left table's own row order, one record of
//      (1 + masks) ids per left row so that the random write touches one line.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "skew.h"

namespace tmp {

struct AsofPlan {
  OrderBuffers lb, rb;
  int32_t *lastnn;   // [n_masks][n_right]
  int32_t *unit_agg; // [n_masks][kUnits]
  uint64_t *lrec;    // [n_left][stride] packed left payload records (2+ columns)
  uint64_t *rrec;    // [n_right][stride] packed right payload records (2+ columns)
  uint64_t *rsort[kMaxPay];  // right payload columns in sorted order (when the ordering step carries them)
  int64_t *block_bounds;     // [merge blocks + 1]
  uint16_t *cmask;           // [n_right] validity bits in sorted order, when they do not fit the sort words
};

// 8-byte words per packed payload record: a power of two up to 4 (so that a record never
// straddles a 32-byte sector), then multiples of 4.
static inline int rec_stride(int n_cols) {
  if (n_cols <= 2) return n_cols < 1 ? 1 : n_cols;
  return (n_cols + 3) & ~3;
}

constexpr int kUnits = kLastValidUnits;  // waves that each scan one contiguous slice of the right side
constexpr int kUnitBlock = 256;  // 4 waves per block
constexpr int kRowsPerIter = 64 * 4;

struct ScanGeom {
  int units;
  int64_t rows_per_unit;
};

static ScanGeom scan_geom(int64_t n) {
  ScanGeom g;
  int64_t iters = (n + kRowsPerIter - 1) / kRowsPerIter;
  if (iters < 1) iters = 1;
  int64_t per = (iters + kUnits - 1) / kUnits;
  g.rows_per_unit = per * kRowsPerIter;
  g.units = (int)((n + g.rows_per_unit - 1) / g.rows_per_unit);
  if (g.units < 1) g.units = 1;
  return g;
}

// The scans below run front to back; REV mirrors the index space (row i stands for row n-1-i),
// which turns "last set position at or before" into "first set position at or after".
// Phase 1: per unit and mask, the last position whose bit is set (or -1).
template <bool REV>
__global__ __launch_bounds__(kUnitBlock) void lastnn_reduce_kernel(const uint64_t *comp, int64_t n,
                                                                   int n_masks,
                                                                   int64_t rows_per_unit, int units,
                                                                   int32_t *unit_agg, const uint16_t *ext) {
  int unit = blockIdx.x * (kUnitBlock / 64) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (unit >= units) return;
  int64_t begin = (int64_t)unit * rows_per_unit;
  int64_t end = begin + rows_per_unit < n ? begin + rows_per_unit : n;
  int32_t last[TMP_MAX_MASKS];
#pragma unroll
  for (int m = 0; m < TMP_MAX_MASKS; ++m) last[m] = -1;
  for (int64_t base = begin; base < end; base += kRowsPerIter) {
    uint64_t w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t i = base + r * 64 + lane;
      w[r] = i < end ? (ext ? (uint64_t)ext[REV ? n - 1 - i : i] : comp[REV ? n - 1 - i : i]) : 0ull;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int m = 0; m < TMP_MAX_MASKS; ++m) {
        if (m < n_masks) {
          uint64_t bal = __ballot((w[r] >> m) & 1ull);
          if (bal) last[m] = (int32_t)(base + r * 64 + 63 - __clzll((long long)bal));
        }
      }
    }
  }
  if (lane == 0)
    for (int m = 0; m < n_masks; ++m) unit_agg[(size_t)m * kUnits + unit] = last[m];
}

// Phase 2: exclusive prefix-max over the units of one mask (grid.x = n_masks, 1024 threads).
__global__ __launch_bounds__(1024) void lastnn_carry_kernel(int32_t *unit_agg, int units) {
  __shared__ int32_t wave_max_s[16];
  int32_t *agg = unit_agg + (size_t)blockIdx.x * kUnits;
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t v[4], m = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int u = tid * 4 + k;
    v[k] = u < units ? agg[u] : -1;
    m = v[k] > m ? v[k] : m;
  }
  int32_t inc = m;
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc = t > inc ? t : inc;
  }
  if (lane == 63) wave_max_s[wave] = inc;
  __syncthreads();
  int32_t run = -1;
  for (int w = 0; w < wave; ++w) run = wave_max_s[w] > run ? wave_max_s[w] : run;
  int32_t prev = __shfl_up(inc, 1, 64);
  if (lane > 0) run = prev > run ? prev : run;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int u = tid * 4 + k;
    if (u < units) agg[u] = run;
    run = v[k] > run ? v[k] : run;
  }
}

// Phase 3: write lastnn[m][j] = last position <= j whose mask bit m is set, or -1 (REV: the
// first position >= j, or -1).
template <bool REV>
__global__ __launch_bounds__(kUnitBlock) void lastnn_scan_kernel(const uint64_t *comp, int64_t n,
                                                                 int n_masks,
                                                                 int64_t rows_per_unit, int units,
                                                                 const int32_t *unit_agg,
                                                                 int32_t *lastnn, const uint16_t *ext, int sparse) {
  // sparse: entries of rows whose own bit is set (lastnn = the row itself) are left unwritten -- the
  // reader has the row's bits in hand and asks only about rows whose bit is clear
  int unit = blockIdx.x * (kUnitBlock / 64) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (unit >= units) return;
  int64_t begin = (int64_t)unit * rows_per_unit;
  int64_t end = begin + rows_per_unit < n ? begin + rows_per_unit : n;
  int32_t carry[TMP_MAX_MASKS];
#pragma unroll
  for (int m = 0; m < TMP_MAX_MASKS; ++m)
    carry[m] = m < n_masks ? unit_agg[(size_t)m * kUnits + unit] : -1;
  const uint64_t le_mask = ~0ull >> (63 - lane);
  for (int64_t base = begin; base < end; base += kRowsPerIter) {
    uint64_t w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t i = base + r * 64 + lane;
      w[r] = i < end ? (ext ? (uint64_t)ext[REV ? n - 1 - i : i] : comp[REV ? n - 1 - i : i]) : 0ull;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t i = base + r * 64 + lane;
#pragma unroll
      for (int m = 0; m < TMP_MAX_MASKS; ++m) {
        if (m < n_masks) {
          uint64_t bal = __ballot((w[r] >> m) & 1ull);
          uint64_t le = bal & le_mask;
          int32_t j = le ? (int32_t)(base + r * 64 + 63 - __clzll((long long)le)) : carry[m];
          if (i < end && !(sparse && ((w[r] >> m) & 1ull))) {
            if (REV) lastnn[(size_t)m * n + (n - 1 - i)] = j < 0 ? -1 : (int32_t)(n - 1 - j);
            else lastnn[(size_t)m * n + i] = j;
          }
          if (bal) carry[m] = (int32_t)(base + r * 64 + 63 - __clzll((long long)bal));
        }
      }
    }
  }
}

size_t last_valid_scratch_ints() { return (size_t)TMP_MAX_MASKS * kUnits; }

int last_valid_unit_carry(int32_t *unit_agg, int units, int n_masks, hipStream_t st) {
  if (n_masks <= 0 || units <= 0) return TMP_OK;
  TMP_LAUNCH(lastnn_carry_kernel, dim3(n_masks), dim3(1024), st, unit_agg, units);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

int last_valid_scan(const uint64_t *comp, int64_t n, int n_masks, int32_t *unit_agg, int32_t *out,
                    bool reverse, hipStream_t st, const uint16_t *ext, bool sparse) {
  if (n <= 0 || n_masks <= 0) return TMP_OK;
  ScanGeom g = scan_geom(n);
  int blocks = (g.units + (kUnitBlock / 64) - 1) / (kUnitBlock / 64);
  if (reverse) {
    TMP_LAUNCH(lastnn_reduce_kernel<true>, dim3(blocks), dim3(kUnitBlock), st, comp, n, n_masks,
               g.rows_per_unit, g.units, unit_agg, ext);
    TMP_LAUNCH(lastnn_carry_kernel, dim3(n_masks), dim3(1024), st, unit_agg, g.units);
    TMP_LAUNCH(lastnn_scan_kernel<true>, dim3(blocks), dim3(kUnitBlock), st, comp, n, n_masks,
               g.rows_per_unit, g.units, unit_agg, out, ext, sparse ? 1 : 0);
  } else {
    TMP_LAUNCH(lastnn_reduce_kernel<false>, dim3(blocks), dim3(kUnitBlock), st, comp, n, n_masks,
               g.rows_per_unit, g.units, unit_agg, ext);
    TMP_LAUNCH(lastnn_carry_kernel, dim3(n_masks), dim3(1024), st, unit_agg, g.units);
    TMP_LAUNCH(lastnn_scan_kernel<false>, dim3(blocks), dim3(kUnitBlock), st, comp, n, n_masks,
               g.rows_per_unit, g.units, unit_agg, out, ext, sparse ? 1 : 0);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

// Number of right rows ordered at or before the left sort word `key` (low bits dropped).
__device__ __forceinline__ int64_t upper_bound(const uint64_t *comp, int64_t lo, int64_t hi,
                                               uint64_t key, int low_bits) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if ((comp[mid] >> low_bits) <= key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Payload columns of the fused join.  A table with two or more payload columns is first packed
// into row records (pack_rows_kernel) so that the one random access a joined row costs
// brings all its columns; a single column is read where it lies (stride 1).
struct JoinCols {
  int n_left, lstride;
  const uint64_t *lrec;
  uint64_t *lout[TMP_MAX_JOIN_COLS];
  int n, rstride;
  const uint64_t *rrec;
  uint64_t *out[TMP_MAX_JOIN_COLS];
  uint8_t *out_valid[TMP_MAX_JOIN_COLS];
  int8_t rule[TMP_MAX_JOIN_COLS];   // -1: last visible row; m: last visible row with mask m valid
  int8_t vmask[TMP_MAX_JOIN_COLS];  // -1: never NULL once matched; m: NULL where mask m is 0
  // slot[c] >= 0: word of the right payload record that holds column c; -1: the column is the
  // right table's own time column, whose values the ordered sort words already carry
  int8_t slot[TMP_MAX_JOIN_COLS];
  int n_rec;                        // columns held in the records
  // rsorted: the ordering step delivered those columns in the order of the right side's sort
  // words (rs[slot][position]); no records, no row ids
  int rsorted;
  const uint64_t *rs[kMaxPay];
};

struct PackCols {
  int n, stride;
  const uint64_t *src[TMP_MAX_JOIN_COLS];
};

// rec[i][c] = src[c][i]; every column is read once, front to back.  One lane per 16-byte half
// of a record pair of columns, so that a wave's stores cover one contiguous run.
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(PackCols pc, int64_t n, uint64_t *rec) {
  const int pairs = pc.stride >> 1;  // stride is even here (2 or a multiple of 4)
  const int64_t total = n * pairs;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    int64_t i = pairs == 2 ? e >> 1 : e / pairs;
    int c = (int)(e - i * pairs) * 2;
    const uint64_t *pa, *pb;
    if (pairs <= 2) {  // selects instead of a lane-varying index into the argument block
      pa = c == 0 ? pc.src[0] : pc.src[2];
      pb = c == 0 ? pc.src[1] : pc.src[3];
    } else {
      pa = pc.src[c];
      pb = pc.src[c + 1];
    }
    ulonglong2 v;
    v.x = c < pc.n ? pa[i] : 0ull;
    v.y = c + 1 < pc.n ? pb[i] : 0ull;
    *reinterpret_cast<ulonglong2 *>(rec + (size_t)e * 2) = v;
  }
}

// bounds[b] = number of right rows ordered at or before the first left row of merge block b
// (b < n_blocks), bounds[n_blocks] = n_right: one thread per block, so that the 30-step searches
// of all blocks run side by side instead of one per block at the head of its own work.
__global__ __launch_bounds__(kBlock) void merge_bounds_kernel(const uint64_t *comp_l, int64_t n_left,
                                                              const uint64_t *comp_r, int64_t n_right, int low_bits,
                                                              int rows_per_block, int64_t n_blocks, int64_t *bounds) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_blocks) return;
  if (b == n_blocks) { bounds[b] = n_right; return; }
  bounds[b] = upper_bound(comp_r, 0, n_right, comp_l[b * rows_per_block] >> low_bits, low_bits);
}

#ifndef TEMPO_MERGE_ROWS
#define TEMPO_MERGE_ROWS 2
#endif
constexpr int kMergeRows = TEMPO_MERGE_ROWS;     // left rows per thread
constexpr int kMergeRight = 768 * kMergeRows;  // right sort words staged in LDS per block

// Block = 1024 consecutive rows of the ordered left side.  Their positions in the ordered right
// side all fall between the positions of the block's first and last row; that slice of right
// sort words is staged in LDS (when it fits) and every row binary-searches it there.
//
// NC, NM > 0: the launch has at most NC joined columns and NM masks; the loops over them are then
// unrolled, which turns every per-column field of the argument block (rule, slot, pointers) into a
// constant-offset scalar load made once, instead of one indexed load -- and one wait for it -- per
// column, mask and row.  NC = NM = 0: any number, loops as written.
template <int NC, int NM>
__global__ __launch_bounds__(kBlock) void asof_merge_kernel(
    const uint64_t *comp_l, const uint32_t *perm_l, int64_t n_left, const uint64_t *comp_r,
    const uint32_t *perm_r, int64_t n_right, Layout lay, int n_masks, const int32_t *lastnn,
    SkewParams sp, int32_t *out_idx, int32_t *out_missing, int sorted_output, int32_t *out_perm,
    int32_t *out_key, int64_t *out_ts, JoinCols jc, const int64_t *block_bounds, const uint16_t *cmask) {
  __shared__ uint64_t rw[kMergeRight];
  constexpr int kColLoop = NC > 0 ? NC : TMP_MAX_JOIN_COLS, kColUnroll = NC > 0 ? NC : 1;
  constexpr int kMaskLoop = NM > 0 ? NM : TMP_MAX_MASKS, kMaskUnroll = NM > 0 ? NM : 1;
  const int low = lay.low_bits;
  int64_t tile = (int64_t)blockIdx.x * (kBlock * kMergeRows);
  int64_t tile_end = tile + kBlock * kMergeRows < n_left ? tile + kBlock * kMergeRows : n_left;
  // The block's rows fall between the positions of its first row and of the next block's first
  // row in the ordered right side (merge_bounds_kernel); the row just before the first position
  // stays reachable.
  const int64_t b0 = block_bounds[blockIdx.x];
  const int64_t blo = b0 > 0 ? b0 - 1 : 0, bhi = block_bounds[blockIdx.x + 1];
  const bool staged = bhi - blo <= kMergeRight;  // block-uniform
  if (staged) {
    for (int e = threadIdx.x; e < (int)(bhi - blo); e += kBlock) rw[e] = comp_r[blo + e];
    __syncthreads();
  }
  int miss_row = 0;
  uint32_t miss_mask = 0;
  // A row's work is a chain of dependent accesses; what does not depend on the chain is requested
  // ahead of it: the left words and row ids of all of the thread's rows up front, and -- once a
  // row's position in the right side is known -- its last-valid positions of the first four masks
  // together with the payload of the row it matched.
  uint64_t wl_pre[kMergeRows];
  uint32_t src_pre[kMergeRows];
#pragma unroll
  for (int r = 0; r < kMergeRows; ++r) {
    const int64_t i = tile + r * kBlock + threadIdx.x;
    wl_pre[r] = i < tile_end ? comp_l[i] : 0ull;
    src_pre[r] = (i < tile_end && perm_l) ? perm_l[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < kMergeRows; ++r) {
    int64_t i = tile + r * kBlock + threadIdx.x;
    if (i >= tile_end) break;
    uint64_t wl = wl_pre[r];
    uint64_t kl = wl >> low;
    int64_t pos;
    uint64_t wr_row = 0;
    if (staged) {
      int lo = 0, hi = (int)(bhi - blo);
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((rw[mid] >> low) <= kl) lo = mid + 1;
        else hi = mid;
      }
      pos = blo + lo;
      if (lo > 0) wr_row = rw[lo - 1];
    } else {
      pos = upper_bound(comp_r, blo, bhi, kl, low);
      if (pos > blo) wr_row = comp_r[pos - 1];
    }
    int64_t jr = pos - 1;
    uint32_t key_l = lay.key_of(wl);
    bool same = jr >= blo && lay.key_of(wr_row) == key_l;
    double part_l = 0.0;
    if (sp.enabled) part_l = bracket_part_ts(lay.prim_of(wl), sp);
    uint32_t src_row = src_pre[r];
    uint32_t dst = sorted_output ? (uint32_t)i : src_row;
    int32_t jpre[4];
#pragma unroll
    // a matched row whose own bit is set is its own last valid row (the scan does not even write
    // that entry when the bits ride in the words); the array is asked about the others only
    for (int mm = 0; mm < 4; ++mm)
      jpre[mm] = (same && mm < n_masks) ? ((!cmask && ((wr_row >> mm) & 1ull)) ? (int32_t)jr : lastnn[(size_t)mm * n_right + jr])
                                        : -1;
    if (sorted_output && out_perm) {
      if (perm_l) out_perm[i] = (int32_t)src_row;  // null: the ordering step left the row ids in out_perm itself
      if (out_key) out_key[i] = (int32_t)key_l;
      if (out_ts) out_ts[i] = lay.prim_of(wl);
    }
    // jrow: position (ordered right side) of the last visible row, or -1
    const int64_t jrow = (same && (!sp.enabled || bracket_visible(lay.prim_of(wr_row), part_l, sp))) ? jr : -1;
    int32_t row_src = -1;
    int32_t *rec = out_idx ? out_idx + (size_t)dst * (1 + n_masks) : nullptr;
    if (rec) {
      if (jrow >= 0) row_src = (int32_t)perm_r[jrow];
      rec[0] = row_src;
    }
    miss_row |= jrow < 0;
    if (jc.n_left) {  // left payload of this row (sorted_output only)
      const uint64_t *lr = jc.lrec + (size_t)src_row * jc.lstride;
      for (int c = 0; c < jc.n_left; ++c) jc.lout[c][i] = lr[c];
    }
    // the matched row's payload record: up to four columns are held in registers
    const uint64_t *rr = nullptr;
    uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (jc.n) {
      if (jrow >= 0 && jc.n_rec > 0 && jc.rsorted) {
        q0 = jc.rs[0][jrow];
        if (jc.n_rec > 1) q1 = jc.rs[1][jrow];
        if (jc.n_rec > 2) q2 = jc.rs[2][jrow];
        if (jc.n_rec > 3) q3 = jc.rs[3][jrow];
      } else if (jrow >= 0 && jc.n_rec > 0) {
        if (row_src < 0) row_src = (int32_t)perm_r[jrow];
        rr = jc.rrec + (size_t)(uint32_t)row_src * jc.rstride;
        if (jc.rstride == 1) q0 = rr[0];
        else {
          ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(rr);
          q0 = a.x; q1 = a.y;
          if (jc.rstride >= 4) {
            ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(rr + 2);
            q2 = b.x; q3 = b.y;
          }
        }
      }
      // validity bits of the matched row: the low bits of its word, or the separate array
      const uint64_t mrow = cmask ? (jrow >= 0 ? (uint64_t)cmask[jrow] : 0ull) : wr_row;
#pragma unroll kColUnroll
      for (int c = 0; c < kColLoop; ++c) {  // columns that follow the row rule
        if (c >= jc.n) break;
        if (jc.rule[c] >= 0) continue;
        bool okv = jrow >= 0 && (jc.vmask[c] < 0 || ((mrow >> jc.vmask[c]) & 1ull));
        const int sl = jc.slot[c];
        uint64_t v = sl < 0 ? (uint64_t)lay.prim_of(wr_row)
                            : (sl == 0 ? q0 : sl == 1 ? q1 : sl == 2 ? q2 : sl == 3 ? q3 : (rr ? rr[sl] : 0ull));
        jc.out[c][dst] = okv ? v : 0ull;
        if (jc.out_valid[c]) jc.out_valid[c][dst] = okv ? 1 : 0;
      }
    }
#pragma unroll kMaskUnroll
    for (int m = 0; m < kMaskLoop; ++m) {
      // guarded, not left by a break: with the inline asm below the loop only unrolls when its trip
      // count is exact
      if (m >= n_masks) continue;
      int64_t jm = -1;  // position of the last visible row whose mask m is valid, or -1
      uint64_t wm = 0;
      if (same) {
        int32_t j = m == 0 ? jpre[0] : m == 1 ? jpre[1] : m == 2 ? jpre[2] : m == 3 ? jpre[3]
                    : ((!cmask && ((wr_row >> m) & 1ull)) ? (int32_t)jr : lastnn[(size_t)m * n_right + jr]);
        if (j >= 0) {
          // two typed loads, not one load through a pointer chosen at run time (that becomes
          // a flat access, which waits on every outstanding global and LDS operation)
          const bool in_lds = staged && j >= blo;
          uint64_t w_lds = rw[in_lds ? (int)(j - blo) : 0];
          asm volatile("" : "+v"(w_lds));  // keeps the compiler from merging the two loads again
          uint64_t w_glb = 0;
          if (!in_lds) w_glb = comp_r[j];
          wm = in_lds ? w_lds : w_glb;
          if (j == jr) {  // the row rule already judged this row
            if (jrow >= 0) jm = j;
          } else if (lay.key_of(wm) == key_l && (!sp.enabled || bracket_visible(lay.prim_of(wm), part_l, sp))) {
            jm = j;
          }
        }
      }
      if (rec) rec[1 + m] = jm < 0 ? -1 : (jm == jrow && row_src >= 0 ? row_src : (int32_t)perm_r[jm]);
      miss_mask |= (uint32_t)(jm < 0) << m;
      const uint64_t *rm = nullptr;
      const uint64_t mm = cmask ? (jm >= 0 ? (uint64_t)cmask[jm] : 0ull) : wm;
#pragma unroll kColUnroll
      for (int c = 0; c < kColLoop; ++c) {
        if (c >= jc.n) break;
        if (jc.rule[c] != m) continue;
        bool okv = jm >= 0 && (jc.vmask[c] < 0 || ((mm >> jc.vmask[c]) & 1ull));
        uint64_t v = 0;
        if (okv) {
          const int sl = jc.slot[c];
          if (sl < 0) {
            v = (uint64_t)lay.prim_of(wm);
          } else if (jm == jrow) {
            v = sl == 0 ? q0 : sl == 1 ? q1 : sl == 2 ? q2 : sl == 3 ? q3 : rr[sl];
          } else if (jc.rsorted) {
            v = sl == 0 ? jc.rs[0][jm] : sl == 1 ? jc.rs[1][jm] : sl == 2 ? jc.rs[2][jm] : jc.rs[3][jm];
          } else {
            if (!rm) rm = jc.rrec + (size_t)perm_r[jm] * jc.rstride;
            v = rm[sl];
          }
        }
        jc.out[c][dst] = v;
        if (jc.out_valid[c]) jc.out_valid[c][dst] = okv ? 1 : 0;
      }
    }
  }
  // one flag per rule; a wave raises it only while it still reads as clear (the flags are hot:
  // unconditional atomics from every wave serialise on one address)
  const bool lane0 = (threadIdx.x & 63) == 0;
  if (__any(miss_row) && lane0 && __atomic_load_n(&out_missing[0], __ATOMIC_RELAXED) == 0)
    atomicOr(&out_missing[0], 1);
  for (int m = 0; m < n_masks; ++m) {
    int any = __any((miss_mask >> m) & 1u);
    if (any && lane0 && __atomic_load_n(&out_missing[1 + m], __ATOMIC_RELAXED) == 0)
      atomicOr(&out_missing[1 + m], 1);
  }
}

__global__ void fill_i32_kernel(int32_t *p, int64_t n, int32_t v) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// The two sides of a join are ordered independently; the right side's kernels run on a stream of
// the library's own (one per device, created on first use) between two events on the caller's
// stream, so that the bandwidth-bound passes of one side overlap the issue-bound tile kernels of
// the other.  Returns false where the stream or its events cannot be had: the caller's stream
// then carries everything.
struct SideStream {
  hipStream_t st;
  hipEvent_t fork, join;
  bool ok, tried;
  // held by a call from its fork record to its join wait: two host threads joining on one device
  // would otherwise interleave their records and waits on the one event pair
  std::mutex busy;
};
static SideStream *side_stream() {
  static SideStream per_dev[64];
  static std::mutex create;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideStream &x = per_dev[dev];
  std::lock_guard<std::mutex> guard(create);
  if (!x.tried) {
    x.tried = true;
    x.ok = hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
    if (!x.ok) (void)hipGetLastError();
  }
  return x.ok ? &x : nullptr;
}

// Makes the caller's stream wait for what was put on the side stream -- on every way out of the
// join, errors included, so that the workspace is never released under a running kernel.
struct SideJoin {
  SideStream *s;
  hipStream_t st;
  SideJoin(SideStream *side, hipStream_t caller) : s(side), st(caller) {}
  hipError_t join() {
    if (!s) return hipSuccess;
    SideStream *x = s;
    s = nullptr;
    hipError_t e = hipEventRecord(x->join, x->st);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, x->join, 0);
    x->busy.unlock();  // taken where the call forked
    return e;
  }
  ~SideJoin() { (void)join(); }
};

static void asof_plan(const tmp_asof_args *a, Workspace &ws, AsofPlan *p) {
  // one left payload column travels with the rows through the ordering passes
  order_plan(a->n_left, ws, &p->lb, plan_keys(a->key_left, a->n_keys), a->sorted_output && a->n_left_cols == 1);
  int nrec = 0;  // right payload columns other than the right table's own time column
  for (int c = 0; c < (a->sorted_output ? a->n_right_cols : 0); ++c)
    if (a->right_cols && a->right_cols[c] != (const void *)a->ts_right) ++nrec;
  // Carrying the right payload through the right side's passes makes every access of the merge
  // stream, but at today's pass rates it costs more than the merge's one random record read per
  // row saves (5x10^8 + 5x10^8 rows, two 8-byte columns: join 80.1 ms with, 75.2 ms without).
  // TEMPO_RIGHT_CARRY=1 switches it on for measurements.
  const int rcarry = (nrec >= 1 && nrec <= kMaxPay && getenv("TEMPO_RIGHT_CARRY")) ? nrec : 0;
  order_plan(a->n_right, ws, &p->rb, plan_keys(a->key_right, a->n_keys), rcarry);
  for (int c = 0; c < kMaxPay; ++c)
    p->rsort[c] = c < rcarry ? ws.take<uint64_t>((size_t)(a->n_right > 0 ? a->n_right : 1)) : nullptr;
  size_t nm = (size_t)(a->n_masks > 0 ? a->n_masks : 1);
  p->lastnn = ws.take<int32_t>(nm * (size_t)(a->n_right > 0 ? a->n_right : 1));
  p->unit_agg = ws.take<int32_t>(last_valid_scratch_ints());
  p->cmask = ws.take<uint16_t>(a->n_masks > 0 ? (size_t)(a->n_right > 0 ? a->n_right : 1) : 1);
  size_t n_bounds = (size_t)(a->n_left / (kBlock * kMergeRows) + 3);
  if (asof_sorted_bounds_count(a->n_left) > n_bounds) n_bounds = asof_sorted_bounds_count(a->n_left);
  p->block_bounds = ws.take<int64_t>(n_bounds);
  const int nlc = a->sorted_output ? a->n_left_cols : 0;
  p->lrec = ws.take<uint64_t>(nlc >= 2 ? (size_t)rec_stride(nlc) * (size_t)a->n_left : 2);
  p->rrec = ws.take<uint64_t>(nrec >= 2 ? (size_t)rec_stride(nrec) * (size_t)a->n_right : 2);
}

}  // namespace tmp

using namespace tmp;

static int asof_validate(const tmp_asof_args *a) {
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_asof_join: null args");
  TMP_REQUIRE(a->n_left >= 0 && a->n_right >= 0 && a->n_left < (1ll << 31) && a->n_right < (1ll << 31),
              TMP_EINVAL, "tmp_asof_join: row counts must be in [0, 2^31)");
  TMP_REQUIRE(a->n_masks >= 0 && a->n_masks <= TMP_MAX_MASKS, TMP_EINVAL,
              "tmp_asof_join: n_masks must be in [0, %d]", TMP_MAX_MASKS);
  TMP_REQUIRE(a->n_left == 0 || a->n_right == 0 ||
                  (a->key_left == nullptr) == (a->key_right == nullptr), TMP_EINVAL,
              "tmp_asof_join: key columns must be given for both sides or neither");
  TMP_REQUIRE(a->n_left == 0 || a->ts_left, TMP_EINVAL, "tmp_asof_join: ts_left is null");
  TMP_REQUIRE(a->n_right == 0 || a->ts_right, TMP_EINVAL, "tmp_asof_join: ts_right is null");
  TMP_REQUIRE(a->n_masks == 0 || a->masks, TMP_EINVAL, "tmp_asof_join: masks missing");
  const bool fused = a->sorted_output && (a->n_left_cols > 0 || a->n_right_cols > 0);
  TMP_REQUIRE(a->n_left == 0 || a->out_idx || fused, TMP_EINVAL, "tmp_asof_join: out_idx is null");
  TMP_REQUIRE(a->sorted_output || (a->n_left_cols == 0 && a->n_right_cols == 0), TMP_EINVAL,
              "tmp_asof_join: payload columns need sorted_output");
  TMP_REQUIRE(a->n_left_cols >= 0 && a->n_left_cols <= TMP_MAX_JOIN_COLS && a->n_right_cols >= 0 &&
                  a->n_right_cols <= TMP_MAX_JOIN_COLS,
              TMP_EINVAL, "tmp_asof_join: at most %d payload columns per side", TMP_MAX_JOIN_COLS);
  TMP_REQUIRE(a->n_left_cols == 0 || (a->left_cols && a->out_left_cols), TMP_EINVAL,
              "tmp_asof_join: left payload arrays missing");
  TMP_REQUIRE(a->n_right_cols == 0 ||
                  (a->right_cols && a->out_right_cols && a->right_col_rule && a->right_col_valid && a->out_right_valid),
              TMP_EINVAL, "tmp_asof_join: right payload arrays missing");
  for (int c = 0; c < a->n_right_cols; ++c)
    TMP_REQUIRE(a->right_col_rule[c] >= -1 && a->right_col_rule[c] < a->n_masks && a->right_col_valid[c] >= -1 &&
                    a->right_col_valid[c] < a->n_masks,
                TMP_EINVAL, "tmp_asof_join: right column %d refers to a mask that was not given", c);
  TMP_REQUIRE(a->n_left == 0 || !a->sorted_output || a->out_perm, TMP_EINVAL,
              "tmp_asof_join: out_perm is required with sorted_output");
  TMP_REQUIRE(a->out_missing, TMP_EINVAL, "tmp_asof_join: out_missing is null");
  if (a->skew) {
    TMP_REQUIRE(a->ts_partition_val == a->ts_partition_val && a->ts_partition_val != 0.0, TMP_EINVAL,
                "tmp_asof_join: ts_partition_val must be a non-zero number");
  }
  return TMP_OK;
}

extern "C" size_t tmp_asof_join_workspace_bytes(const tmp_asof_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  AsofPlan p;
  asof_plan(a, ws, &p);
  return ws.off;
}

extern "C" int tmp_asof_join(const tmp_asof_args *a, void *workspace, size_t workspace_bytes,
                             void *stream) {
  clear_error();
  TMP_TRY(asof_validate(a));
  hipStream_t st = (hipStream_t)stream;
  if (a->left_in_place) *a->left_in_place = 0;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_missing, 0, sizeof(int32_t) * (1 + a->n_masks), st));
  if (a->n_left == 0) return TMP_OK;
  Workspace ws(workspace, workspace_bytes);
  AsofPlan p;
  asof_plan(a, ws, &p);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE,
              "tmp_asof_join: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
  // the left side's second row-id buffer is the caller's out_perm (when that is aligned like the
  // workspace's own): an ordering that ends there has put the row ids where they belong
  if (a->sorted_output && a->out_perm && ((uintptr_t)a->out_perm & 255) == 0) p.lb.idx[1] = (uint32_t *)a->out_perm;
  if (a->n_right == 0 && !a->sorted_output) {  // nothing to match: every index is -1, everything is missing
    int grid = grid_for(a->n_left, kBlock * 4, 8192);
    TMP_LAUNCH(fill_i32_kernel, dim3(grid), dim3(kBlock), st, a->out_idx, a->n_left * (1 + a->n_masks), -1);
    TMP_LAUNCH(fill_i32_kernel, dim3(1), dim3(64), st, a->out_missing, 1 + a->n_masks, 1);
    TMP_CHECK_HIP(hipGetLastError());
    return TMP_OK;
  }

  OrderSpec ls, rs;
  memset(&ls, 0, sizeof(ls));
  memset(&rs, 0, sizeof(rs));
  ls.n = a->n_left; ls.key = a->key_left; ls.n_keys = a->n_keys; ls.ts = a->ts_left; ls.prim_div = 1;
  rs.n = a->n_right; rs.key = a->key_right; rs.n_keys = a->n_keys; rs.ts = a->ts_right; rs.prim_div = 1;
  rs.seq = a->seq_right; rs.seq_valid = a->seq_right_valid;
  rs.n_masks = a->n_masks;
  ls.coarse_ok = rs.coarse_ok = 1;
  rs.ext_masks_ok = 1;
  for (int m = 0; m < a->n_masks; ++m) rs.masks[m] = a->masks[m];

  // ranges of both sides, one host read each, then one shared bit layout
  TMP_TRY(order_scan_ranges(ls, p.lb, st));
  TMP_TRY(order_scan_ranges(rs, p.rb, st));
  RangeInfo rl, rr;
  TMP_TRY(order_read_ranges(p.lb, &rl, st));
  TMP_TRY(order_read_ranges(p.rb, &rr, st));
  SkewParams sp;
  skew_params_set(&sp, a->skew != 0, a->v_is_int, a->ts_is_seconds, a->ts_partition_val, a->one_minus_fraction);
  // both sides arrive in order: one sweep over the columns, nothing laid out (TEMPO_NO_SORTED_JOIN=1
  // sends such tables down the general path, for comparison)
  if (!rl.unsorted && !rr.unsorted && a->n_right > 0 && !getenv("TEMPO_NO_SORTED_JOIN"))
    return asof_sorted_join(a, sp, p.block_bounds, p.lastnn, p.unit_agg, st);
  RangeInfo both = rr;
  both.prim_min = rl.prim_min < rr.prim_min ? rl.prim_min : rr.prim_min;
  both.prim_max = rl.prim_max > rr.prim_max ? rl.prim_max : rr.prim_max;
  both.not_thousands = rl.not_thousands | rr.not_thousands;
  both.not_millions = rl.not_millions | rr.not_millions;
  Layout lay;
  OrderSpec lay_spec = rs;  // the key field exists if either side carries keys (one may be empty)
  if (!lay_spec.key) lay_spec.key = ls.key;
  TMP_TRY(order_make_layout(lay_spec, both, a->n_masks, &lay));
  // a left table that is in order already can be used by reference (the right one is not, or the
  // one-sweep join above would have taken the call): row p of the result is left row p, and the
  // caller's own key / time / payload columns stand for out_perm, out_key, out_ts, out_left_cols
  const bool left_ref = a->sorted_output && a->left_in_place != nullptr && !rl.unsorted;
  if (left_ref) *a->left_in_place = 1;
  // payload columns of the fused join
  const int nlc = (a->sorted_output && !left_ref) ? a->n_left_cols : 0, nrc = a->sorted_output ? a->n_right_cols : 0;
  JoinCols jc;
  memset(&jc, 0, sizeof(jc));
  jc.n_left = nlc;
  jc.n = a->n_right > 0 ? nrc : 0;
  jc.lstride = rec_stride(nlc);
  PackCols pl, pr;
  memset(&pl, 0, sizeof(pl));
  memset(&pr, 0, sizeof(pr));
  for (int c = 0; c < nlc; ++c) {
    TMP_REQUIRE(a->left_cols[c] && a->out_left_cols[c], TMP_EINVAL, "tmp_asof_join: left payload column %d is null", c);
    pl.src[c] = (const uint64_t *)a->left_cols[c];
    jc.lout[c] = (uint64_t *)a->out_left_cols[c];
  }
  int nrec = 0;
  for (int c = 0; c < nrc; ++c) {
    TMP_REQUIRE(a->n_right == 0 || a->right_cols[c], TMP_EINVAL, "tmp_asof_join: right payload column %d is null", c);
    TMP_REQUIRE(a->out_right_cols[c], TMP_EINVAL, "tmp_asof_join: right output column %d is null", c);
    jc.out[c] = (uint64_t *)a->out_right_cols[c];
    jc.out_valid[c] = a->out_right_valid[c];
    jc.rule[c] = (int8_t)a->right_col_rule[c];
    jc.vmask[c] = (int8_t)a->right_col_valid[c];
    if (a->right_cols[c] == (const void *)a->ts_right) {
      jc.slot[c] = -1;  // the right time column itself: read off the ordered sort words
    } else {
      jc.slot[c] = (int8_t)nrec;
      pr.src[nrec++] = (const uint64_t *)a->right_cols[c];
    }
  }
  jc.n_rec = nrec;
  jc.rstride = rec_stride(nrec);
  if (nrc > 0 && a->n_right == 0) {  // nothing to match: every joined column is NULL
    for (int c = 0; c < nrc; ++c) {
      TMP_CHECK_HIP(hipMemsetAsync(a->out_right_cols[c], 0, sizeof(uint64_t) * a->n_left, st));
      if (a->out_right_valid[c]) TMP_CHECK_HIP(hipMemsetAsync(a->out_right_valid[c], 0, a->n_left, st));
    }
  }
  OrderResult lo, ro;
  // the right side's work up to the merge can go to the side stream (large sides only: below that
  // the two extra event waits cost more than the overlap gives)
  hipStream_t rst = st;
  SideStream *side = nullptr;
  // 5x10^8 + 5x10^8 random-order rows join in 61.8 ms instead of 64.1.  The kernels of the two sides
  // then stretch each other, so per-kernel timings stop adding up: TEMPO_JOIN_STREAMS=0 keeps
  // everything on the caller's stream (profiling runs that want durations which add up).
  const char *two = getenv("TEMPO_JOIN_STREAMS");
  if (!(two && two[0] == '0') && a->n_right >= (1 << 20) && a->n_left >= (1 << 20)) side = side_stream();
  if (side) side->busy.lock();
  SideJoin side_join(side, st);  // from here on every way out joins the side stream and releases it
  if (side) {
    TMP_CHECK_HIP(hipEventRecord(side->fork, st));
    TMP_CHECK_HIP(hipStreamWaitEvent(side->st, side->fork, 0));
    rst = side->st;
  }
  // a single left payload column is handed to the ordering step, which delivers it in (key, ts)
  // order straight into the output column; where it could not (rows already in order: no pass
  // ran) the merge fetches it through the row ids as it does for several columns
  bool left_delivered = false;
  if (nlc == 1) {
    const uint64_t *src1[1] = {pl.src[0]};
    uint64_t *out1[1] = {jc.lout[0]};
    TMP_TRY(order_build(ls, lay, sorted_mode(rl), p.lb, &lo, st, 1, src1, out1, &left_delivered));
  } else {
    TMP_TRY(order_build(ls, lay, sorted_mode(rl), p.lb, &lo, st));
  }
  if (left_delivered) jc.n_left = 0;
  // likewise the right payload columns (up to kMaxPay of them): delivered in the order of the right
  // side's sort words, the merge then reads them by position; otherwise records fetched by row id
  bool right_delivered = false;
  if (nrec >= 1 && nrec <= kMaxPay && a->n_right > 0 && p.rsort[0]) {
    TMP_TRY(order_build(rs, lay, sorted_mode(rr), p.rb, &ro, rst, nrec, pr.src, p.rsort, &right_delivered));
  } else {
    TMP_TRY(order_build(rs, lay, sorted_mode(rr), p.rb, &ro, rst));
  }
  if (right_delivered) {
    jc.rsorted = 1;
    for (int c = 0; c < nrec; ++c) jc.rs[c] = p.rsort[c];
  }
  jc.lrec = nlc == 1 ? pl.src[0] : p.lrec;
  jc.rrec = nrec == 1 ? pr.src[0] : p.rrec;
  if (nlc >= 2) {
    pl.n = nlc; pl.stride = jc.lstride;
    TMP_LAUNCH(pack_rows_kernel, dim3(grid_for(a->n_left * (jc.lstride / 2), kBlock, 8192)), dim3(kBlock), st, pl, a->n_left, p.lrec);
  }
  if (nrec >= 2 && a->n_right > 0 && !right_delivered) {
    pr.n = nrec; pr.stride = jc.rstride;
    TMP_LAUNCH(pack_rows_kernel, dim3(grid_for(a->n_right * (jc.rstride / 2), kBlock, 8192)), dim3(kBlock), rst, pr, a->n_right, p.rrec);
  }

  const uint16_t *cmask = nullptr;
  if (lay.ext_masks && a->n_right > 0) {  // the validity bits did not fit the sort words
    TMP_TRY(order_gather_masks(rs, ro, p.cmask, rst));
    cmask = p.cmask;
  }
  TMP_TRY(last_valid_scan(ro.comp, a->n_right, a->n_masks, p.unit_agg, p.lastnn, false, rst, cmask, cmask == nullptr));
  TMP_CHECK_HIP(side_join.join());
  int grid = (int)((a->n_left + kBlock * kMergeRows - 1) / (kBlock * kMergeRows));
  TMP_LAUNCH(merge_bounds_kernel, dim3((grid + 1 + kBlock - 1) / kBlock), dim3(kBlock), st, lo.comp, a->n_left, ro.comp,
             a->n_right, lay.low_bits, kBlock * kMergeRows, (int64_t)grid, p.block_bounds);
  // the left row ids: not read at all when they already lie in out_perm and no left payload has to be
  // fetched through them (record mode in left row order needs them as destinations)
  const uint32_t *perm_l = lo.perm;
  if (a->sorted_output && !left_ref && (const void *)lo.perm == (const void *)a->out_perm && jc.n_left == 0) perm_l = nullptr;
  {
    ProfScope ps("asof_merge_kernel", st);
    if (jc.n <= 4 && a->n_masks <= 4)
      hipLaunchKernelGGL((asof_merge_kernel<4, 4>), dim3(grid), dim3(kBlock), 0, st, lo.comp, perm_l, a->n_left,
                         ro.comp, ro.perm, a->n_right, lay, a->n_masks, p.lastnn, sp, a->out_idx,
                         a->out_missing, a->sorted_output, left_ref ? nullptr : a->out_perm, left_ref ? nullptr : a->out_key,
                         left_ref ? nullptr : a->out_ts, jc, p.block_bounds, cmask);
    else
      hipLaunchKernelGGL((asof_merge_kernel<0, 0>), dim3(grid), dim3(kBlock), 0, st, lo.comp, perm_l, a->n_left,
                         ro.comp, ro.perm, a->n_right, lay, a->n_masks, p.lastnn, sp, a->out_idx,
                         a->out_missing, a->sorted_output, left_ref ? nullptr : a->out_perm, left_ref ? nullptr : a->out_key,
                         left_ref ? nullptr : a->out_ts, jc, p.block_bounds, cmask);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
