// AS-OF join of two tables that both arrive in (key, ts[, seq NULLS FIRST]) order: one sweep over
// the columns themselves.  Same result as the general path of asof.hip (python/tempo/tsdf.py:463-560,
// :111-162, :164-190) -- which for such input would still lay out sort words and row ids for both
// sides, pack payload records, scan last-valid positions over the whole right side and read all of
// it back in the merge.  Here nothing is built:
//   1. per 512 right rows and NULL mask, the last row whose mask byte is set (reads the mask bytes
//      only), turned into "last valid row before this stretch" by a two-level running maximum;
//   2. per block of 512 left rows, the number of right rows ordered at or before its first row
//      (one binary search over the right key / time columns);
//   3. one kernel: a block stages the slice of right (key, time) its rows can match in LDS, with
//      one bitmap per NULL mask built from wave ballots; every left row searches the slice, applies
//      the key / bracket tests, finds the last valid row per mask in the bitmaps, reads the
//      joined values by position (the positions of neighbouring rows are neighbours) and writes
//      its row of the joined table.  Every column of either table is read once.
// A slice longer than the LDS buffer is worked through in overlapping chunks, so the only limit is
// the row count.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "skew.h"

namespace tmp {

#ifndef TEMPO_SJ_THREADS
#define TEMPO_SJ_THREADS 256
#endif
constexpr int SJ_THREADS = TEMPO_SJ_THREADS;
#ifndef TEMPO_SJ_ROWS
#define TEMPO_SJ_ROWS 2
#endif
constexpr int SJ_ROWS = TEMPO_SJ_ROWS;         // consecutive left rows per thread: 2 or 4
static_assert(SJ_ROWS == 2 || SJ_ROWS == 4, "the wide accesses are written for two or four rows");
constexpr int SJ_TILE = SJ_THREADS * SJ_ROWS;  // left rows per block
#ifndef TEMPO_SJ_CAP
#define TEMPO_SJ_CAP 768
#endif
#ifndef TEMPO_SJ_WAVES
#define TEMPO_SJ_WAVES 8
#endif
constexpr int SJ_CAP = TEMPO_SJ_CAP;           // right rows staged at a time
constexpr int SJ_WORDS = SJ_CAP / 64;
constexpr int SJ_RTILE = 512;                  // right rows per entry of the last-valid table
constexpr int SJ_SEQ = TMP_MAX_MASKS;          // bitmap slot of "the row's sequence value is not NULL"

static_assert(SJ_WORDS <= 63, "one lane per bitmap word plus the closing entry");
static_assert(SJ_CAP % SJ_THREADS == 0, "staging steps cover the buffer exactly");

struct SjSides {
  int64_t n_left, n_right;
  const int32_t *key_l, *key_r;  // both NULL: a single series
  const int64_t *ts_l, *ts_r;
  const uint8_t *seq_valid;      // with has_seq; NULL: every sequence value is present
  int has_seq;
};

struct SjMasks {
  int n;
  const uint8_t *p[TMP_MAX_MASKS];
};

struct SjCols {
  int n_left;
  const uint64_t *lsrc[TMP_MAX_JOIN_COLS];
  uint64_t *lout[TMP_MAX_JOIN_COLS];
  int n;  // joined right columns
  const uint64_t *rsrc[TMP_MAX_JOIN_COLS];
  uint64_t *out[TMP_MAX_JOIN_COLS];
  uint8_t *out_valid[TMP_MAX_JOIN_COLS];
  int8_t rule[TMP_MAX_JOIN_COLS];   // -1: last visible row; m: last visible row with mask m valid
  int8_t vmask[TMP_MAX_JOIN_COLS];  // -1: never NULL once matched; m: NULL where mask m is 0
  // joined column that is the right table's own time column under the row rule, or -1: its values
  // are taken from the staged slice instead of being read a second time
  int ts_col;
};

// Is the right row (kr, tr, sequence present?) ordered at or before the left row (kl, tl)?  Left
// rows carry a NULL sequence value, which sorts first, and of two rows that tie the right one comes
// first: a right row at the left row's own time is visible unless it has a sequence value
// (tsdf.py:117-121).
__device__ __forceinline__ bool sj_before(int32_t kr, int64_t tr, bool seq_set, int32_t kl, int64_t tl) {
  return kr != kl ? kr < kl : (tr != tl ? tr < tl : !seq_set);
}

// bounds[b] = right rows ordered at or before the first left row of block b; bounds[n_blocks] = n_right.
__global__ __launch_bounds__(kBlock) void sj_bounds_kernel(SjSides s, int64_t n_blocks, int64_t *bounds) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_blocks) return;
  if (b == n_blocks) { bounds[b] = s.n_right; return; }
  const int64_t i = b * SJ_TILE;
  const int32_t kl = s.key_l ? s.key_l[i] : 0;
  const int64_t tl = s.ts_l[i];
  int64_t lo = 0, hi = s.n_right;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    const int32_t kr = s.key_r ? s.key_r[mid] : 0;
    const int64_t tr = s.ts_r[mid];
    bool seq_set = false;
    if (s.has_seq && kr == kl && tr == tl) seq_set = !s.seq_valid || s.seq_valid[mid] != 0;
    if (sj_before(kr, tr, seq_set, kl, tl)) lo = mid + 1;
    else hi = mid;
  }
  bounds[b] = lo;
}

struct SjGeom {
  int64_t n_tiles;
  int tiles_per_unit;
  int units;
};
static SjGeom sj_geom(int64_t n_right) {
  SjGeom g;
  g.n_tiles = (n_right + SJ_RTILE - 1) / SJ_RTILE;
  if (g.n_tiles < 1) g.n_tiles = 1;
  g.tiles_per_unit = (int)((g.n_tiles + kLastValidUnits - 1) / kLastValidUnits);
  g.units = (int)((g.n_tiles + g.tiles_per_unit - 1) / g.tiles_per_unit);
  return g;
}

// One wave per unit (a run of tiles): tile_last[m][t] = last row of tile t whose mask m is set, or
// -1; unit_agg[m][unit] = the same over the unit's tiles.
__global__ __launch_bounds__(256) void sj_tile_last_kernel(SjMasks mk, int64_t n, SjGeom g, int32_t *tile_last,
                                                           int32_t *unit_agg) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (unit >= g.units) return;
  const int64_t t0 = (int64_t)unit * g.tiles_per_unit;
  const int64_t t1 = t0 + g.tiles_per_unit < g.n_tiles ? t0 + g.tiles_per_unit : g.n_tiles;
  for (int m = 0; m < mk.n; ++m) {
    const uint8_t *p = mk.p[m];
    int32_t ulast = -1;
    int64_t t = t0;
    if (((uintptr_t)p & 7) == 0) {
      // whole tiles, four at a time: a lane reads eight mask bytes of each, the tile's last set byte
      // is the highest set byte of the highest lane that saw one
      for (; t + 4 <= t1 && (t + 4) * SJ_RTILE <= n; t += 4) {
        uint64_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = reinterpret_cast<const uint64_t *>(p + (t + u) * SJ_RTILE)[lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t bal = __ballot(w[u] != 0);
          int32_t last = -1;
          if (bal) {
            const int top = 63 - __clzll((long long)bal);
            const int byte = (63 - __clzll((long long)(w[u] | 1ull))) >> 3;
            last = (int32_t)((t + u) * SJ_RTILE + top * 8 + __shfl(byte, top, 64));
          }
          if (lane == 0) tile_last[(size_t)m * g.n_tiles + t + u] = last;
          ulast = last >= 0 ? last : ulast;
        }
      }
    }
    for (; t < t1; ++t) {
      const int64_t base = t * SJ_RTILE;
      uint8_t v[SJ_RTILE / 64];
#pragma unroll
      for (int r = 0; r < SJ_RTILE / 64; ++r) {
        const int64_t i = base + r * 64 + lane;
        v[r] = p[i < n ? i : n - 1];
      }
      int32_t last = -1;
#pragma unroll
      for (int r = 0; r < SJ_RTILE / 64; ++r) {
        const uint64_t bal = __ballot(base + r * 64 + lane < n && v[r] != 0);
        if (bal) last = (int32_t)(base + r * 64 + 63 - __clzll((long long)bal));
      }
      if (lane == 0) tile_last[(size_t)m * g.n_tiles + t] = last;
      ulast = last >= 0 ? last : ulast;
    }
    if (lane == 0) unit_agg[(size_t)m * kLastValidUnits + unit] = ulast;
  }
}

// unit_agg now holds, per unit, the last valid row before it: one thread per (unit, mask) walks the
// unit's tiles and replaces each tile's own last row by the last valid row before the tile.
__global__ __launch_bounds__(kBlock) void sj_tile_carry_kernel(int n_masks, SjGeom g, const int32_t *unit_agg,
                                                               int32_t *tile_last) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int unit = (int)(id / n_masks), m = (int)(id % n_masks);
  if (unit >= g.units) return;
  int32_t carry = unit_agg[(size_t)m * kLastValidUnits + unit];
  const int64_t t0 = (int64_t)unit * g.tiles_per_unit;
  const int64_t t1 = t0 + g.tiles_per_unit < g.n_tiles ? t0 + g.tiles_per_unit : g.n_tiles;
  int32_t *row = tile_last + (size_t)m * g.n_tiles;
  for (int64_t t = t0; t < t1; ++t) {
    const int32_t v = row[t];
    row[t] = carry;
    carry = v >= 0 ? v : carry;
  }
}

// SJ_ROWS consecutive elements as the widest accesses that cover them (the caller has checked alignment).
__device__ __forceinline__ void sj_load64(const uint64_t *p, uint64_t v[SJ_ROWS]) {
#pragma unroll
  for (int r = 0; r < SJ_ROWS; r += 2) {
    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p + r);
    v[r] = a.x; v[r + 1] = a.y;
  }
}
__device__ __forceinline__ void sj_store64(uint64_t *p, const uint64_t v[SJ_ROWS]) {
#pragma unroll
  for (int r = 0; r < SJ_ROWS; r += 2) *reinterpret_cast<ulonglong2 *>(p + r) = make_ulonglong2(v[r], v[r + 1]);
}
__device__ __forceinline__ void sj_load32(const int32_t *p, int32_t v[SJ_ROWS]) {
  if constexpr (SJ_ROWS == 4) {
    const int4 a = *reinterpret_cast<const int4 *>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
    const int2 a = *reinterpret_cast<const int2 *>(p);
    v[0] = a.x; v[1] = a.y;
  }
}
__device__ __forceinline__ void sj_store32(int32_t *p, const int32_t v[SJ_ROWS]) {
  if constexpr (SJ_ROWS == 4) *reinterpret_cast<int4 *>(p) = make_int4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<int2 *>(p) = make_int2(v[0], v[1]);
}
__device__ __forceinline__ void sj_store8(uint8_t *p, const uint8_t v[SJ_ROWS]) {
  if constexpr (SJ_ROWS == 4) *reinterpret_cast<uchar4 *>(p) = make_uchar4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<uchar2 *>(p) = make_uchar2(v[0], v[1]);
}

// A thread owns SJ_ROWS consecutive left rows (their columns come and go as 8- or 16-byte accesses when
// `vec` says every pointer involved is aligned for it); consecutive rows have neighbouring places
// in the right table, so only the first of them is found by a full binary search over the staged
// slice, the others by galloping on from their predecessor's place.  What bounds the kernel is the
// number of waves a CU holds (every phase is a round trip to memory or a chain of LDS probes):
// two rows per thread and a 768-row staging buffer keep it at 64 registers and 12 KB of LDS, eight
// waves per SIMD -- 5 x 10^8 + 5 x 10^8 rows: 8.9 ms, against 11.8 ms with four rows and 2048.
//
// NC, NM > 0: at most NC joined columns and NM masks; the loops over them are unrolled, which turns
// the per-column fields of the argument block into one-time scalar loads and keeps the rows' chosen
// positions in registers.  NC = NM = 0: any number.
template <int NC, int NM>
__global__ __launch_bounds__(SJ_THREADS, TEMPO_SJ_WAVES) void sj_merge_kernel(SjSides s, SjMasks mk, SkewParams sp, SjCols jc,
                                                              const int64_t *bounds, const int32_t *tile_carry,
                                                              int64_t n_rtiles, int32_t *out_idx, int32_t *out_missing,
                                                              int sorted_output, int32_t *out_perm, int32_t *out_key,
                                                              int64_t *out_ts, int vec, int by_ref) {
  __shared__ int64_t s_ts[SJ_CAP];
  __shared__ int32_t s_key[SJ_CAP];
  __shared__ uint64_t s_bm[TMP_MAX_MASKS + 1][SJ_WORDS];  // one bit per staged row and mask; slot SJ_SEQ: sequence present
  __shared__ int32_t s_prev[TMP_MAX_MASKS][SJ_WORDS + 1];  // last valid row before word w (the table's rows, or -1)
  __shared__ int32_t s_carry[TMP_MAX_MASKS];               // last valid row before the staged chunk
  constexpr int kColLoop = NC > 0 ? NC : TMP_MAX_JOIN_COLS;
  constexpr int kMaskLoop = NM > 0 ? NM : TMP_MAX_MASKS;
  constexpr int kStage = NM > 0 ? (SJ_CAP / SJ_THREADS < 4 ? SJ_CAP / SJ_THREADS : 4) : 2;  // staging steps whose loads are requested together
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_masks = mk.n;
  const int64_t tile = (int64_t)blockIdx.x * SJ_TILE;
  const int64_t tile_end = tile + SJ_TILE < s.n_left ? tile + SJ_TILE : s.n_left;
  const int64_t i0 = tile + (int64_t)tid * SJ_ROWS;  // this thread's first row
  const int nrow = tile_end - i0 >= SJ_ROWS ? SJ_ROWS : (tile_end > i0 ? (int)(tile_end - i0) : 0);
  const bool wide = vec && nrow == SJ_ROWS;
  // the block's rows fall between the places of its first row and of the next block's first row
  // in the right table; the row just before the first place stays reachable
  const int64_t b0 = bounds[blockIdx.x];
  const int64_t blo = b0 > 0 ? b0 - 1 : 0, bhi = bounds[blockIdx.x + 1];

#ifdef SJ_TIMING
  long long tk[12]; int ntk = 0;
#define SJTICK() do { __syncthreads(); tk[ntk < 11 ? ntk++ : 11] = clock64(); } while (0)
#else
#define SJTICK() do {} while (0)
#endif
  SJTICK();
  int32_t kl[SJ_ROWS];
  int64_t tl[SJ_ROWS];
  if (wide) {
    sj_load64(reinterpret_cast<const uint64_t *>(s.ts_l + i0), reinterpret_cast<uint64_t *>(tl));
#pragma unroll
    for (int r = 0; r < SJ_ROWS; ++r) kl[r] = 0;
    if (s.key_l) sj_load32(s.key_l + i0, kl);
  } else {
#pragma unroll
    for (int r = 0; r < SJ_ROWS; ++r) {
      const int64_t i = i0 + r < tile_end ? i0 + r : tile_end - 1;
      kl[r] = s.key_l ? s.key_l[i] : 0;
      tl[r] = s.ts_l[i];
    }
  }
  // left payload: the table is in order already, its columns pass through (by_ref: the caller uses
  // its own columns, nothing is copied)
  for (int c = 0; c < (by_ref ? 0 : jc.n_left); ++c) {
    if (wide) {
      uint64_t v[SJ_ROWS];
      sj_load64(jc.lsrc[c] + i0, v);
      sj_store64(jc.lout[c] + i0, v);
    } else {
      for (int r = 0; r < nrow; ++r) jc.lout[c][i0 + r] = jc.lsrc[c][i0 + r];
    }
  }
  // last valid row before the slice, per mask: the table's entry for the slice's 512-row stretch,
  // then the bytes between the stretch's start and the slice (all requested at once)
  if (n_masks > 0) {
    const int64_t t0 = blo / SJ_RTILE, base = t0 * SJ_RTILE;
    for (int m = wave; m < n_masks; m += SJ_THREADS / 64) {
      const uint8_t *p = mk.p[m];
      int32_t last = tile_carry[(size_t)m * n_rtiles + t0];
      uint8_t v[SJ_RTILE / 64];
#pragma unroll
      for (int r = 0; r < SJ_RTILE / 64; ++r) {
        const int64_t e = base + r * 64 + lane;
        v[r] = p[e < s.n_right ? e : s.n_right - 1];
      }
#pragma unroll
      for (int r = 0; r < SJ_RTILE / 64; ++r) {
        const uint64_t bal = __ballot(base + r * 64 + lane < blo && v[r] != 0);
        if (bal) last = (int32_t)(base + r * 64 + 63 - __clzll((long long)bal));
      }
      if (lane == 0) s_carry[m] = last;
    }
  }

  SJTICK();  // 1: left rows, pass-through, carry-in
  bool done[SJ_ROWS];
  int32_t jrow[SJ_ROWS];
  int32_t jm[SJ_ROWS][kMaskLoop];
  uint32_t okbits[SJ_ROWS];
  int64_t tjoin[SJ_ROWS];  // time of the row chosen by the row rule
#pragma unroll
  for (int r = 0; r < SJ_ROWS; ++r) {
    done[r] = r >= nrow;
    jrow[r] = -1;
    tjoin[r] = 0;
    okbits[r] = 0;
#pragma unroll
    for (int m = 0; m < kMaskLoop; ++m) jm[r][m] = -1;
  }
  uint32_t miss_row = 0, miss_mask = 0;
  for (int64_t cs = blo;;) {
    const int64_t ce = cs + SJ_CAP < bhi ? cs + SJ_CAP : bhi;
    const int len = (int)(ce - cs);  // 0 only when no right row lies at or before any row of the block
    const bool last_chunk = ce >= bhi;
    const int steps = (len + SJ_THREADS - 1) / SJ_THREADS;
    for (int k0 = 0; k0 < steps; k0 += kStage) {
      int64_t t[kStage];
      int32_t kk[kStage];
      uint8_t mv[kStage][kMaskLoop];
      uint8_t sv[kStage];
#pragma unroll
      for (int u = 0; u < kStage; ++u) {
        const int e = (k0 + u) * SJ_THREADS + tid;
        const int64_t g = e < len ? cs + e : cs;  // a row of the table either way
        t[u] = s.ts_r[g];
        kk[u] = s.key_r ? s.key_r[g] : 0;
#pragma unroll
        for (int m = 0; m < kMaskLoop; ++m) mv[u][m] = m < n_masks ? mk.p[m][g] : 0;
        sv[u] = (s.has_seq && s.seq_valid) ? s.seq_valid[g] : 1;
      }
#pragma unroll
      for (int u = 0; u < kStage; ++u) {
        if (k0 + u >= steps) continue;  // block-uniform
        const int e = (k0 + u) * SJ_THREADS + tid;
        const bool in = e < len;
        if (in) { s_ts[e] = t[u]; s_key[e] = kk[u]; }
#pragma unroll
        for (int m = 0; m < kMaskLoop; ++m) {
          if (m < n_masks) {
            const uint64_t bal = __ballot(in && mv[u][m] != 0);
            if (lane == 0) s_bm[m][(k0 + u) * (SJ_THREADS / 64) + wave] = bal;
          }
        }
        if (s.has_seq) {
          const uint64_t bal = __ballot(in && sv[u] != 0);
          if (lane == 0) s_bm[SJ_SEQ][(k0 + u) * (SJ_THREADS / 64) + wave] = bal;
        }
      }
    }
    __syncthreads();
    SJTICK();  // 2: staged
    const int nwords = (len + 63) >> 6;
    for (int m = wave; m < n_masks; m += SJ_THREADS / 64) {
      int32_t v = -1;
      if (lane < nwords) {
        const uint64_t w = s_bm[m][lane];
        if (w) v = (int32_t)(cs + lane * 64 + 63 - __clzll((long long)w));
      }
      int32_t inc = v;
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc = t > inc ? t : inc;
      }
      const int32_t before = __shfl_up(inc, 1, 64);
      const int32_t c = s_carry[m];
      const int32_t ex = lane == 0 ? c : (before > c ? before : c);
      if (lane <= nwords) s_prev[m][lane] = ex;
    }
    __syncthreads();

    SJTICK();  // 3: prefix
    // ---- places of this thread's rows in the staged chunk (LDS only; a candidate that lies
    // before the chunk is looked up in the table itself)
    int from = -1;  // place of the previous row looked at in this chunk, or -1: none yet
#pragma unroll
    for (int r = 0; r < SJ_ROWS; ++r) {
      if (done[r]) continue;
      auto before_row = [&](int e) {
        const int32_t kr = s_key[e];
        const int64_t tr = s_ts[e];
        bool seq_set = false;
        if (s.has_seq && kr == kl[r] && tr == tl[r]) seq_set = (s_bm[SJ_SEQ][e >> 6] >> (e & 63)) & 1ull;
        return sj_before(kr, tr, seq_set, kl[r], tl[r]);
      };
      int lo = 0, hi = len;
      if (from >= 0) {  // rows are in order: this one's place is at or after its predecessor's
        lo = from;
        int step = 1;
        while (lo + step <= len && before_row(lo + step - 1)) { lo += step; step <<= 1; }
        hi = lo + step - 1 < len ? lo + step - 1 : len;
      }
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (before_row(mid)) lo = mid + 1;
        else hi = mid;
      }
      from = lo;
      if (lo == len && !last_chunk) continue;  // its place lies in a later chunk
      done[r] = true;
      const int q = lo - 1;  // staged index of the last right row at or before this one
      const bool same = q >= 0 && s_key[q] == kl[r];
      double part_l = 0.0;
      if (sp.enabled) part_l = bracket_part_ts(tl[r], sp);
      if (same && (!sp.enabled || bracket_visible(s_ts[q], part_l, sp))) {
        jrow[r] = (int32_t)(cs + q);
        tjoin[r] = s_ts[q];
      }
      miss_row |= jrow[r] < 0;
#pragma unroll
      for (int m = 0; m < kMaskLoop; ++m) {
        if (m >= n_masks) continue;
        int32_t found = -1;
        if (same) {
          const int w = q >> 6;
          const uint64_t bits = s_bm[m][w] & (~0ull >> (63 - (q & 63)));
          const int32_t j = bits ? (int32_t)(cs + w * 64 + 63 - __clzll((long long)bits)) : s_prev[m][w];
          if (j >= 0) {
            if (j == (int32_t)(cs + q)) {  // the row rule already judged this row
              found = jrow[r];
            } else {
              int32_t kj;
              int64_t tj;
              if (j >= cs) {
                kj = s_key[j - cs];
                tj = s_ts[j - cs];
              } else {
                kj = s.key_r ? s.key_r[j] : 0;
                tj = s.ts_r[j];
              }
              if (kj == kl[r] && (!sp.enabled || bracket_visible(tj, part_l, sp))) found = j;
            }
          }
        }
        jm[r][m] = found;
        miss_mask |= (uint32_t)(found < 0) << m;
      }
      // per joined column: was a row chosen, and is the column's own value there not NULL
#pragma unroll
      for (int c = 0; c < kColLoop; ++c) {
        if (c >= jc.n) continue;
        int32_t j = jrow[r];
        const int rl = jc.rule[c];
        if (rl >= 0) {
#pragma unroll
          for (int m = 0; m < kMaskLoop; ++m)
            if (m == rl) j = jm[r][m];
        }
        bool ok = j >= 0;
        const int vm = jc.vmask[c];
        if (ok && vm >= 0) {
          if (j >= cs) ok = (s_bm[vm][(j - cs) >> 6] >> ((j - cs) & 63)) & 1ull;
          else ok = mk.p[vm][j] != 0;
        }
        okbits[r] |= (uint32_t)ok << c;
      }
    }
    SJTICK();  // 4: search + rules
    if (last_chunk) break;
    __syncthreads();  // every row has read the staged chunk
    for (int m = wave; m < n_masks; m += SJ_THREADS / 64)
      if (lane == 0) s_carry[m] = s_prev[m][nwords];
    cs = ce - 1;  // chunks overlap by one row: the row before a place is always staged with it
  }

  // ---- the rows of the joined table: values by position in the right table
  if (sorted_output && nrow > 0 && !by_ref) {
    if (wide) {
      int32_t ids[SJ_ROWS];
#pragma unroll
      for (int r = 0; r < SJ_ROWS; ++r) ids[r] = (int32_t)i0 + r;
      sj_store32(out_perm + i0, ids);
      if (out_key) sj_store32(out_key + i0, kl);
      if (out_ts) sj_store64(reinterpret_cast<uint64_t *>(out_ts + i0), reinterpret_cast<const uint64_t *>(tl));
    } else {
      for (int r = 0; r < nrow; ++r) {
        out_perm[i0 + r] = (int32_t)(i0 + r);
        if (out_key) out_key[i0 + r] = kl[r];
        if (out_ts) out_ts[i0 + r] = tl[r];
      }
    }
  }
  if (out_idx) {
#pragma unroll
    for (int r = 0; r < SJ_ROWS; ++r) {
      if (r >= nrow) continue;
      int32_t *rec = out_idx + (size_t)(i0 + r) * (1 + n_masks);
      rec[0] = jrow[r];
#pragma unroll
      for (int m = 0; m < kMaskLoop; ++m)
        if (m < n_masks) rec[1 + m] = jm[r][m];
    }
  }
  // four columns at a time: their values requested together, then their stores
#pragma unroll
  for (int c0 = 0; c0 < kColLoop; c0 += 4) {
    uint64_t val[4][SJ_ROWS];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + cc;
#pragma unroll
      for (int r = 0; r < SJ_ROWS; ++r) {
        val[cc][r] = 0;
        if (c >= jc.n || !((okbits[r] >> c) & 1u)) continue;
        if (c == jc.ts_col) { val[cc][r] = (uint64_t)tjoin[r]; continue; }
        int32_t j = jrow[r];
        const int rl = jc.rule[c];
        if (rl >= 0) {
#pragma unroll
          for (int m = 0; m < kMaskLoop; ++m)
            if (m == rl) j = jm[r][m];
        }
        val[cc][r] = jc.rsrc[c][j];
      }
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + cc;
      if (c >= jc.n) continue;
      if (wide) {
        sj_store64(jc.out[c] + i0, val[cc]);
        if (jc.out_valid[c]) {
          uint8_t ok[SJ_ROWS];
#pragma unroll
          for (int r = 0; r < SJ_ROWS; ++r) ok[r] = (uint8_t)((okbits[r] >> c) & 1u);
          sj_store8(jc.out_valid[c] + i0, ok);
        }
      } else {
#pragma unroll
        for (int r = 0; r < SJ_ROWS; ++r) {
          if (r >= nrow) continue;
          jc.out[c][i0 + r] = val[cc][r];
          if (jc.out_valid[c]) jc.out_valid[c][i0 + r] = (uint8_t)((okbits[r] >> c) & 1u);
        }
      }
    }
  }
  SJTICK();  // 5: joined values read and written
#ifdef SJ_TIMING
  if (tid == 0 && (blockIdx.x % 49999) == 777)
    printf("SJ block %d len %d: left+carry %lld stage %lld prefix %lld search %lld emit %lld\n", (int)blockIdx.x,
           (int)(bhi - blo), tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
#endif
  // one flag per rule; a wave raises it only while it still reads as clear
  const bool lane0 = lane == 0;
  if (__any(miss_row) && lane0 && __atomic_load_n(&out_missing[0], __ATOMIC_RELAXED) == 0) atomicOr(&out_missing[0], 1);
  for (int m = 0; m < n_masks; ++m) {
    const int any = __any((miss_mask >> m) & 1u);
    if (any && lane0 && __atomic_load_n(&out_missing[1 + m], __ATOMIC_RELAXED) == 0) atomicOr(&out_missing[1 + m], 1);
  }
}

size_t asof_sorted_bounds_count(int64_t n_left) { return (size_t)((n_left + SJ_TILE - 1) / SJ_TILE + 2); }
size_t asof_sorted_table_ints(int64_t n_right, int n_masks) {
  return (size_t)(n_masks > 0 ? n_masks : 1) * (size_t)sj_geom(n_right).n_tiles;
}

int asof_sorted_join(const tmp_asof_args *a, const SkewParams &sp, int64_t *bounds, int32_t *tile_table,
                     int32_t *unit_agg, hipStream_t st) {
  SjSides s;
  memset(&s, 0, sizeof(s));
  s.n_left = a->n_left; s.n_right = a->n_right;
  s.key_l = a->key_left; s.key_r = a->key_right;
  s.ts_l = a->ts_left; s.ts_r = a->ts_right;
  s.has_seq = a->seq_right != nullptr;
  s.seq_valid = a->seq_right_valid;
  TMP_REQUIRE((s.key_l == nullptr) == (s.key_r == nullptr), TMP_EINVAL,
              "tmp_asof_join: key columns must be given for both sides or neither");
  SjMasks mk;
  memset(&mk, 0, sizeof(mk));
  mk.n = a->n_masks;
  for (int m = 0; m < a->n_masks; ++m) {
    TMP_REQUIRE(a->masks[m], TMP_EINVAL, "tmp_asof_join: mask %d is null", m);
    mk.p[m] = a->masks[m];
  }
  SjCols jc;
  memset(&jc, 0, sizeof(jc));
  jc.ts_col = -1;
  if (a->sorted_output) {
    jc.n_left = a->n_left_cols;
    for (int c = 0; c < a->n_left_cols; ++c) {
      TMP_REQUIRE(a->left_cols[c] && a->out_left_cols[c], TMP_EINVAL, "tmp_asof_join: left payload column %d is null", c);
      jc.lsrc[c] = (const uint64_t *)a->left_cols[c];
      jc.lout[c] = (uint64_t *)a->out_left_cols[c];
    }
    jc.n = a->n_right_cols;
    for (int c = 0; c < a->n_right_cols; ++c) {
      TMP_REQUIRE(a->right_cols[c], TMP_EINVAL, "tmp_asof_join: right payload column %d is null", c);
      TMP_REQUIRE(a->out_right_cols[c], TMP_EINVAL, "tmp_asof_join: right output column %d is null", c);
      jc.rsrc[c] = (const uint64_t *)a->right_cols[c];
      jc.out[c] = (uint64_t *)a->out_right_cols[c];
      jc.out_valid[c] = a->out_right_valid[c];
      jc.rule[c] = (int8_t)a->right_col_rule[c];
      jc.vmask[c] = (int8_t)a->right_col_valid[c];
      if (jc.ts_col < 0 && a->right_cols[c] == (const void *)a->ts_right && jc.rule[c] < 0) jc.ts_col = c;
    }
  }
  const SjGeom g = sj_geom(a->n_right);
  if (a->n_masks > 0) {
    TMP_LAUNCH(sj_tile_last_kernel, dim3((g.units + 3) / 4), dim3(256), st, mk, a->n_right, g, tile_table, unit_agg);
    TMP_TRY(last_valid_unit_carry(unit_agg, g.units, a->n_masks, st));
    TMP_LAUNCH(sj_tile_carry_kernel, dim3((unsigned)(((int64_t)g.units * a->n_masks + kBlock - 1) / kBlock)), dim3(kBlock),
               st, a->n_masks, g, unit_agg, tile_table);
  }
  const int64_t n_blocks = (a->n_left + SJ_TILE - 1) / SJ_TILE;
  // 16-byte accesses to the left table's columns and to the joined table need every one of those
  // columns aligned for them (any allocation is; a view that starts inside one may not be)
  uintptr_t a16 = (uintptr_t)s.key_l | (uintptr_t)s.ts_l, a4 = 0;
  const int by_ref = a->sorted_output && a->left_in_place != nullptr;
  if (a->sorted_output && !by_ref) a16 |= (uintptr_t)a->out_perm | (uintptr_t)a->out_key | (uintptr_t)a->out_ts;
  for (int c = 0; c < (by_ref ? 0 : jc.n_left); ++c) a16 |= (uintptr_t)jc.lsrc[c] | (uintptr_t)jc.lout[c];
  for (int c = 0; c < jc.n; ++c) { a16 |= (uintptr_t)jc.out[c]; a4 |= (uintptr_t)jc.out_valid[c]; }
  const int vec = (a16 & 15) == 0 && (a4 & 3) == 0 && !getenv("TEMPO_SORTED_JOIN_SCALAR");
  if (by_ref) *a->left_in_place = 1;
  TMP_LAUNCH(sj_bounds_kernel, dim3((unsigned)((n_blocks + 1 + kBlock - 1) / kBlock)), dim3(kBlock), st, s, n_blocks, bounds);
  {
    ProfScope ps("sj_merge_kernel", st);
    if (jc.n <= 4 && a->n_masks <= 2)
      hipLaunchKernelGGL((sj_merge_kernel<4, 2>), dim3((unsigned)n_blocks), dim3(SJ_THREADS), 0, st, s, mk, sp, jc, bounds,
                         tile_table, g.n_tiles, a->out_idx, a->out_missing, a->sorted_output, a->out_perm, a->out_key,
                         a->out_ts, vec, by_ref);
    else if (jc.n <= 4 && a->n_masks <= 4)
      hipLaunchKernelGGL((sj_merge_kernel<4, 4>), dim3((unsigned)n_blocks), dim3(SJ_THREADS), 0, st, s, mk, sp, jc, bounds,
                         tile_table, g.n_tiles, a->out_idx, a->out_missing, a->sorted_output, a->out_perm, a->out_key,
                         a->out_ts, vec, by_ref);
    else
      hipLaunchKernelGGL((sj_merge_kernel<0, 0>), dim3((unsigned)n_blocks), dim3(SJ_THREADS), 0, st, s, mk, sp, jc, bounds,
                         tile_table, g.n_tiles, a->out_idx, a->out_missing, a->sorted_output, a->out_perm, a->out_key,
                         a->out_ts, vec, by_ref);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

}  // namespace tmp
