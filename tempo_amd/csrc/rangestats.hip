// withRangeStats: rolling RANGE-frame statistics.  Replaces WindowExec over
// rangeBetween(-W, 0) with mean/count/min/max/sum/stddev + zscore
// (python/tempo/tsdf.py:673-721; ordering :563-572).
//
// Rows are ordered by (key, floor-second); the metric's validity bit rides in bit 0 of the
// sort word.  A block owns 1024 consecutive output rows and stages a region of up to 2560 rows
// around them in LDS (32-bit relative sort words + metric, one coalesced read).  Straight from
// the staging registers it reduces every 16-row sub-block (one DPP row) to sum(x - pivot),
// sum((x - pivot)^2), counts, min and max; one wave turns the sums into prefixes over
// sub-blocks, two waves build sparse tables over the minima / maxima.  Every row then finds its
// frame by searching the staged sort words, walks the partial sub-blocks at both ends slot by
// slot and takes the whole sub-blocks in between from the prefixes and the tables.
// Frames that reach outside the region or hold non-finite values take the exact row-sequential
// path (Spark's own Welford recurrence) from global memory.
#include <string.h>

#include "common.h"
#include "rangestats_common.h"

namespace tmp {

constexpr int RK_ROWS = 1024;
// Slots of the staged region: the kernel comes in three sizes, chosen per call from how many rows a
// frame is expected to hold (rs_region_cap) -- everything a block does before its rows' own work
// (staging, sub-block sums, scans) grows with the region, a frame that leaves it sends the column
// down the wide path.
constexpr int RK_CAP_MAX = 2560;
constexpr int RK_THREADS = 1024;
constexpr int RK_FWD = 256;  // rows staged after the tile
#ifndef TEMPO_RK_WPS
#define TEMPO_RK_WPS 8
#endif
constexpr int RK_WAVES_PER_SIMD = TEMPO_RK_WPS;  // 8 = two 1024-thread blocks per CU (at most 64 VGPRs)
#ifndef TEMPO_RK_SB
#define TEMPO_RK_SB 8
#endif
constexpr int RK_SB = TEMPO_RK_SB;      // slots per sub-block (4 or 8)
constexpr int RK_GRP = 16;              // sub-blocks per group (one DPP row of lanes)
constexpr int RK_GLEVELS = 6;           // sparse table over the groups: spans up to 32 of them
static_assert(RK_SB == 4 || RK_SB == 8, "a sub-block is one or two 16-byte LDS reads of sort words");
template <int CAP>
struct RkGeom {
  static constexpr int NSB = CAP / RK_SB;     // sub-blocks of the region
  static constexpr int NGRP = NSB / RK_GRP;   // groups of sub-blocks
  static_assert(CAP <= RK_CAP_MAX && CAP > RK_ROWS + RK_FWD, "region = rows before + the tile + rows after");
  static_assert(NSB % 64 == 0 && NGRP <= 64 && (1 << RK_GLEVELS) > NGRP && NSB <= RK_THREADS,
                "group scans are laid out for whole waves");
};

// ---- DPP moves inside a 16-lane row (used by the min / max scans over groups of sub-blocks)
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int old, int v, bool bound_zero) {
  return bound_zero ? __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true)
                    : __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v, bool bound_zero) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = dpp_i32<CTRL>(lo, lo, bound_zero);
  hi = dpp_i32<CTRL>(hi, hi, bound_zero);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v, bool bound_zero) {
  int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
  lo = dpp_i32<CTRL>(lo, lo, bound_zero);
  hi = dpp_i32<CTRL>(hi, hi, bound_zero);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
// min / max over staged slots.  double: NULL slots hold NaN, and fmin / fmax return the other
// operand for a NaN, so the identity is NaN (frames that hold real NaN / inf values never take
// this path).  int64: identities are the type's extremes.
template <typename T>
struct MinMax;
template <>
struct MinMax<double> {
  static __device__ __forceinline__ double null_slot() { return __builtin_nan(""); }
  static __device__ __forceinline__ double min_identity() { return __builtin_nan(""); }
  static __device__ __forceinline__ double max_identity() { return __builtin_nan(""); }
  // The bare instructions: with a quiet NaN in one operand they return the other one, which is
  // all fmin / fmax are wanted for here; the library calls first pass both operands through a
  // canonicalising v_max_f64 (signalling NaNs, which the staged values never are) -- three
  // instructions where one does.
  static __device__ __forceinline__ double mn(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
  static __device__ __forceinline__ double mx(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
};
template <>
struct MinMax<int64_t> {
  static __device__ __forceinline__ int64_t null_slot() { return 0; }
  static __device__ __forceinline__ int64_t min_identity() { return INT64_MAX; }
  static __device__ __forceinline__ int64_t max_identity() { return INT64_MIN; }
  static __device__ __forceinline__ int64_t mn(int64_t a, int64_t b) { return a < b ? a : b; }
  static __device__ __forceinline__ int64_t mx(int64_t a, int64_t b) { return a > b ? a : b; }
};

// Inclusive scans of min / max along a 16-lane DPP row: forward (row_shr 1, 2, 4, 8) gives every
// lane the extreme of its row up to itself, backward (row_shl) from itself to the row's end.
template <int CTRL>
__device__ __forceinline__ double dpp_any(double v) { return dpp_f64<CTRL>(v, false); }
template <int CTRL>
__device__ __forceinline__ int64_t dpp_any(int64_t v) { return (int64_t)dpp_u64<CTRL>((uint64_t)v, false); }
template <typename T, bool MIN, bool FWD>
__device__ __forceinline__ T group_scan(T v) {
#define G_(c) { T t = dpp_any<c>(v); v = MIN ? MinMax<T>::mn(v, t) : MinMax<T>::mx(v, t); }
  if (FWD) { G_(0x111) G_(0x112) G_(0x114) G_(0x118) }
  else { G_(0x101) G_(0x102) G_(0x104) G_(0x108) }
#undef G_
  return v;
}

constexpr uint32_t RK_SAT = 0x7fffffffu;  // saturated relative sort word

// One staged slot of a double metric into the frame's running sums; NULL slots hold NaN.
__device__ __forceinline__ void slot_add_f64(double v, double pivot, double &sd, double &sq, uint32_t &dc,
                                             double &mn, double &mx) {
  bool okv = v == v;
  double d = okv ? v - pivot : 0.0;
  sd += d;
  sq = __builtin_fma(d, d, sq);
  dc += okv ? 1u : 0u;
  mn = MinMax<double>::mn(mn, v);
  mx = MinMax<double>::mx(mx, v);
}

// (slot numbers are never negative: the unsigned division is one shift)
__device__ __forceinline__ constexpr int xi(int e) { return e + (int)((unsigned)e / (unsigned)RK_SB); }
// The spare element after the first sub-block holds a NULL for good: the place a walk reads for the
// slots beyond its end.
constexpr int RK_NULL_AT = RK_SB;

template <typename T, int RK_CAP>
struct RsShared {
  static constexpr int RK_NSB = RkGeom<RK_CAP>::NSB;
  static constexpr bool kIntT = ((T)0.5 == (T)0);
  uint32_t ck[RK_CAP];       // ((sort word >> low) - base, saturated) << 1 | metric-not-NULL
  // metric; for double a NULL slot holds NaN.  Slot e lives at xi(e): one spare element after every
  // sub-block, so that the threads that each add up one sub-block (lane stride = a sub-block) read
  // from distinct banks -- with the sub-blocks back to back sixteen lanes share a bank
  T x[RK_CAP + RK_NSB + RK_SB / 2];  // (+ the reach of a walk's reads past the last slot)
  double bs[RK_NSB + 1];     // exclusive prefix over sub-blocks of sum(x - pivot)
  double bq[RK_NSB + 1];     // ... of sum((x - pivot)^2)
  uint64_t bi[kIntT ? RK_NSB + 1 : 1];  // ... of the exact integer sum (int64 metrics)
  uint32_t bc[RK_NSB + 1];   // ... of counts: low 16 bits valid, high 16 bits non-finite
  // min / max of whole sub-blocks: per sub-block its own value, the extreme of its group of 16
  // up to it and from it on, and a small sparse table over the group extremes
  T mn0[RK_NSB], mx0[RK_NSB];
  T mn_pre[RK_NSB], mn_suf[RK_NSB], mx_pre[RK_NSB], mx_suf[RK_NSB];
  T gmn[RK_GLEVELS][64], gmx[RK_GLEVELS][64];
  // the two halves of every sub-block on their own (sums of the first half are total - second):
  // a frame's first and last sub-block are then walked slot by slot over at most half their slots
  double hs2[RK_NSB], hq2[RK_NSB];           // second half: sum(x - pivot), sum((x - pivot)^2)
  uint64_t hi2[kIntT ? RK_NSB : 1];          // ... exact integer sum
  uint32_t hc2[RK_NSB];                      // ... counts (as bc)
  T hmn1[RK_NSB], hmx1[RK_NSB], hmn2[RK_NSB], hmx2[RK_NSB];
  int a_lead[RK_ROWS / 16];  // frame start (slot) of every 16th row of the tile
  int pivot_at;
  int nonfinite;  // some staged value is NaN or +-inf
};

// The table's own key / time / NULL-flag columns, for rows that are in (key, time) order already.
struct RsSrc {
  const int32_t *key;    // null: one series
  const int64_t *ts;
  const uint8_t *valid;  // null: no NULLs
};

// GENK: the rows are in order where they lie, and their sort words are not read from `comp` but built
// here from the table's columns (src) -- nothing is laid out beforehand; out_perm is the identity.
// What this form cannot finish (frames that leave the region, frames with NaN / inf) it marks as
// deferred, and the caller runs the ordinary form over laid-out words.
// W32: `comp` holds 32-bit sort words (the ordering step's narrow form); as with GENK, what would need
// the 64-bit words -- the serial evaluation of a frame with NaN / inf -- is left to a second run.
template <typename T, int RK_CAP, bool GENK, bool W32>
__global__ __launch_bounds__(RK_THREADS, RK_WAVES_PER_SIMD) void range_stats_kernel(const uint64_t *comp,
                                                                 const uint32_t *perm,
                                                                 const T *xs, int64_t n, Layout lay,
                                                                 uint64_t w, RsOut<T> o,
                                                                 int32_t *deferred, int n_tiles, RsSrc src) {
  constexpr bool kInt = ((T)0.5 == (T)0);
  constexpr int RK_NSB = RkGeom<RK_CAP>::NSB, RK_NGRP = RkGeom<RK_CAP>::NGRP;
  __shared__ RsShared<T, RK_CAP> s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int low = lay.low_bits;
  const uint64_t pmask = lay.prim_mask();
  // Workgroups go to the 8 XCDs round-robin; neighbouring tiles stage overlapping regions, so the
  // tiles are dealt out such that each XCD (one L2) works through one contiguous eighth of them.
  const int per_xcd = (n_tiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;
  const int64_t t0 = (int64_t)tile * RK_ROWS;
  const int64_t t1 = t0 + RK_ROWS < n ? t0 + RK_ROWS : n;
  // staged region [rs, re): the tile, RK_FWD rows after it (later same-second peers) and as many
  // rows before it as fit
  constexpr int kBack = RK_CAP - RK_ROWS - RK_FWD;
  const int64_t rs = t0 - kBack > 0 ? t0 - kBack : 0;
  const int64_t re = t0 + RK_ROWS + RK_FWD < n ? t0 + RK_ROWS + RK_FWD : n;
  const int len = (int)(re - rs);
  uint64_t base;
  static_assert(!(GENK && W32), "words are either built here or read");
  const uint32_t *comp32 = reinterpret_cast<const uint32_t *>(comp);
  if (GENK) base = plain_word(lay, src.ts[rs], src.key ? (uint32_t)src.key[rs] : 0u, 0u) >> low;
  else if (W32) base = (uint64_t)comp32[rs] >> low;
  else base = comp[rs] >> low;
  if (tid == 0) { s.pivot_at = RK_CAP; s.nonfinite = 0; s.x[RK_NULL_AT] = MinMax<T>::null_slot(); }
  __syncthreads();
#ifdef RK_TIMING
  long long tk[12]; int ntk = 0;
#define RTICK() do { __syncthreads(); tk[ntk++] = clock64(); } while (0)
#else
#define RTICK() do {} while (0)
#endif
  RTICK();

  // ---- stage the region: one coalesced read of sort words and metric values
  constexpr int kPer = (RK_CAP + RK_THREADS - 1) / RK_THREADS;
  T xv[kPer];
  bool ok[kPer], fin[kPer];
  uint64_t wd[kPer];
  // all loads first, none inside a branch (a slot past the region reads the region's last row):
  // loads inside per-slot branches are each waited for before the next is issued
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    const int e = r * RK_THREADS + tid;
    const uint32_t at = (uint32_t)(e < len ? e : len - 1);  // (block-uniform base + 32-bit offset: one address register)
    if (W32) wd[r] = (uint64_t)(comp32 + rs)[at];
    else if (!GENK) wd[r] = (comp + rs)[at];
    xv[r] = (xs + rs)[at];
  }
  if (GENK) {
    int64_t tsv[kPer];
    uint32_t kv[kPer], vv[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int e = r * RK_THREADS + tid;
      const uint32_t at = (uint32_t)(e < len ? e : len - 1);
      tsv[r] = (src.ts + rs)[at];
      kv[r] = src.key ? (uint32_t)(src.key + rs)[at] : 0u;
      vv[r] = src.valid ? (uint32_t)(src.valid + rs)[at] : 1u;
    }
#pragma unroll
    for (int r = 0; r < kPer; ++r) wd[r] = plain_word(lay, tsv[r], kv[r], vv[r] ? 1u : 0u);
  }
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    int e = r * RK_THREADS + tid;
    ok[r] = false;
    fin[r] = false;
    if (e < RK_CAP) {
      uint32_t c = RK_SAT << 1;  // past-the-end slots sort after everything
      if (e < len) {
        const uint64_t word = wd[r];
        uint64_t rel = (word >> low) - base;
        ok[r] = (((uint32_t)word >> o.vbit) & 1u) != 0;  // (the validity bits are the word's lowest)
        c = (uint32_t)(rel < RK_SAT ? rel : RK_SAT) << 1 | (ok[r] ? 1u : 0u);
        fin[r] = ok[r] && Ord<T>::finite(xv[r]);
        if (ok[r] && !fin[r]) s.nonfinite = 1;
      }
      if (!ok[r]) xv[r] = MinMax<T>::null_slot();
      s.ck[e] = c;
      s.x[xi(e)] = xv[r];
      // pivot = first finite value of the region: one LDS atomic per wave, not per slot
      // (and only where it would lower the value: after the first wave's that is almost never)
      uint64_t bal = __ballot(fin[r]);
      if (bal && lane == 0) {
        const int cand = e + (int)__ffsll((long long)bal) - 1;
        if (cand < __hip_atomic_load(&s.pivot_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMin(&s.pivot_at, cand);
      }
    }
  }
  __syncthreads();
  RTICK();
  const int pat = s.pivot_at;
  const double pivot = pat < RK_CAP ? (double)s.x[xi(pat)] : 0.0;

  // first slot at or after the frame start of the row whose relative sort word is krel, searched
  // in slots [lo, hi]; the row's full word is base + krel (rows with a saturated krel are deferred)
  // Rows of the key the region starts in (all rows of most tiles) have their time field at
  // base's + krel, so the frame starts w before the row or at the region's first slot: 32-bit
  // arithmetic.  same_key = the largest krel of that key; rows of later keys take the general form.
  const uint64_t room = pmask - (base & pmask);
  const uint32_t same_key = (uint32_t)(room < RK_SAT ? room : RK_SAT);
  const uint32_t w32 = (uint32_t)(w < RK_SAT ? w : RK_SAT);
  auto frame_first = [&](uint32_t krel, int lo, int hi) {
    uint32_t srel;
    if (krel <= same_key) {
      srel = krel > w32 ? krel - w32 : 0u;
    } else {
      const uint64_t kstart = frame_start_key(base + krel, pmask, w);
      const uint64_t srel64 = kstart > base ? kstart - base : 0;
      srel = (uint32_t)(srel64 < RK_SAT ? srel64 : RK_SAT);
    }
    const uint32_t bound = srel << 1;  // staged words carry the validity bit below the relative word
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s.ck[mid] < bound) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  // ---- per 8-slot sub-block aggregates: one thread per sub-block adds its eight staged slots up
  //      in registers (a few waves' worth of plain adds instead of cross-lane scans in every wave);
  //      the next wave meanwhile searches the frame start of every 16th row over the whole
  //      region -- frame starts are monotone, so the rows in between search only between two of these
  if (wave == (RK_NSB + 63) / 64) {
    const int64_t il = t0 + lane * 16;
    if (il < t1) {
      const int lil = (int)(il - rs);
      s.a_lead[lane] = frame_first(s.ck[lil] >> 1, 0, lil);
    }
  }
  if (tid < RK_NSB) {
    const int b = tid;
    double sd = 0.0, sq = 0.0;
    uint32_t sc = 0;
    uint64_t si = 0;
    T vmn = MinMax<T>::min_identity(), vmx = MinMax<T>::max_identity();
    double sd1 = 0.0, sq1 = 0.0;
    uint32_t sc1 = 0;
    uint64_t si1 = 0;
    T vmn1 = vmn, vmx1 = vmx;
    bool plain = false;  // a double metric without NaN / inf in the region: a slot is NULL exactly when it holds NaN
    if constexpr (!kInt) plain = s.nonfinite == 0;
    if (plain) {
      if constexpr (!kInt) {
#pragma unroll
        for (int j = 0; j < RK_SB; ++j) {
          slot_add_f64(s.x[b * (RK_SB + 1) + j], pivot, sd, sq, sc, vmn, vmx);
          if (j == RK_SB / 2 - 1) {  // the first half is complete: keep it and start the second
            sd1 = sd; sq1 = sq; sc1 = sc; vmn1 = vmn; vmx1 = vmx;
            sd = 0.0; sq = 0.0; sc = 0;
            vmn = MinMax<T>::min_identity(); vmx = MinMax<T>::max_identity();
          }
        }
      }
    } else {
      uint32_t cs[RK_SB];
#pragma unroll
      for (int q4 = 0; q4 < RK_SB / 4; ++q4) {
        const uint4 c4 = *reinterpret_cast<const uint4 *>(&s.ck[b * RK_SB + 4 * q4]);
        cs[4 * q4] = c4.x; cs[4 * q4 + 1] = c4.y; cs[4 * q4 + 2] = c4.z; cs[4 * q4 + 3] = c4.w;
      }
#pragma unroll
      for (int j = 0; j < RK_SB; ++j) {
        const T v = s.x[b * (RK_SB + 1) + j];
        const bool okj = (cs[j] & 1u) != 0;
        const bool finj = okj && Ord<T>::finite(v);
        const double d = finj ? (double)v - pivot : 0.0;
        sd += d;
        sq = __builtin_fma(d, d, sq);
        sc += (okj ? 1u : 0u) | ((okj && !finj) ? 0x10000u : 0u);
        vmn = MinMax<T>::mn(vmn, okj ? v : MinMax<T>::min_identity());
        vmx = MinMax<T>::mx(vmx, okj ? v : MinMax<T>::max_identity());
        if (kInt) si += okj ? (uint64_t)v : 0ull;
        if (j == RK_SB / 2 - 1) {  // the first half is complete: keep it and start the second
          sd1 = sd; sq1 = sq; sc1 = sc; si1 = si; vmn1 = vmn; vmx1 = vmx;
          sd = 0.0; sq = 0.0; sc = 0; si = 0;
          vmn = MinMax<T>::min_identity(); vmx = MinMax<T>::max_identity();
        }
      }
    }
    s.hs2[b] = sd; s.hq2[b] = sq; s.hc2[b] = sc;
    if (kInt) s.hi2[b] = si;
    s.hmn1[b] = vmn1; s.hmx1[b] = vmx1; s.hmn2[b] = vmn; s.hmx2[b] = vmx;
    sd += sd1; sq += sq1; sc += sc1; si += si1;
    vmn = MinMax<T>::mn(vmn, vmn1); vmx = MinMax<T>::mx(vmx, vmx1);
    s.bs[b + 1] = sd;  // shifted by one: turned into exclusive prefixes below
    s.bq[b + 1] = sq;
    s.bc[b + 1] = sc;
    if (kInt) s.bi[b + 1] = si;
    s.mn0[b] = vmn;
    s.mx0[b] = vmx;
  }
  __syncthreads();
  RTICK();

  // ---- wave 0: prefixes over the sub-blocks; waves 1 / 2: the sparse tables (wave-local work,
  //      LDS keeps one wave's accesses in order)
  if (wave == 0) {
    constexpr int kEach = (RK_NSB + 63) / 64;
    double vs[kEach], vq[kEach];
    uint32_t vc[kEach];
    uint64_t vi[kEach];
    double ts_ = 0.0, tq = 0.0;
    uint32_t tc = 0;
    uint64_t ti = 0;
#pragma unroll
    for (int j = 0; j < kEach; ++j) {
      int b = lane * kEach + j;
      bool in = b < RK_NSB;
      vs[j] = in ? s.bs[b + 1] : 0.0;
      vq[j] = in ? s.bq[b + 1] : 0.0;
      vc[j] = in ? s.bc[b + 1] : 0u;
      vi[j] = (kInt && in) ? s.bi[b + 1] : 0ull;
      ts_ += vs[j]; tq += vq[j]; tc += vc[j]; ti += vi[j];
    }
    double is = ts_, iq = tq;
    uint32_t ic = tc;
    uint64_t ii = ti;
    for (int off = 1; off < 64; off <<= 1) {
      double a = __shfl_up(is, off, 64), b = __shfl_up(iq, off, 64);
      uint32_t c = __shfl_up(ic, off, 64);
      uint64_t d = kInt ? __shfl_up(ii, off, 64) : 0ull;
      if (lane >= off) { is += a; iq += b; ic += c; ii += d; }
    }
    // exclusive over lanes: the lane below's inclusive value, not "inclusive - own" -- a lane whose own
    // sub-blocks hold far larger values than everything before them would round the small prefix away
    double rs_ = __shfl_up(is, 1, 64), rq = __shfl_up(iq, 1, 64);
    if (lane == 0) { rs_ = 0.0; rq = 0.0; }
    uint32_t rc = ic - tc;
    uint64_t ri = ii - ti;
#pragma unroll
    for (int j = 0; j < kEach; ++j) {
      int b = lane * kEach + j;
      if (b < RK_NSB) {
        rs_ += vs[j]; rq += vq[j]; rc += vc[j]; ri += vi[j];
        s.bs[b + 1] = rs_;
        s.bq[b + 1] = rq;
        s.bc[b + 1] = rc;
        if (kInt) s.bi[b + 1] = ri;
      }
    }
    if (lane == 0) {
      s.bs[0] = 0.0; s.bq[0] = 0.0; s.bc[0] = 0u;
      if (kInt) s.bi[0] = 0ull;
    }
  } else {
    // waves 1..15 share the group scans: chunk c < RK_NSB / 64 scans 64 minima, the others 64
    // maxima; lane = sub-block
    constexpr int kChunks = RK_NSB / 64;
    for (int c = wave - 1; c < 2 * kChunks; c += RK_THREADS / 64 - 1) {
      const bool is_min = c < kChunks;
      const int b = (c % kChunks) * 64 + lane;
      if (is_min) {
        const T v = s.mn0[b];
        const T pre = group_scan<T, true, true>(v);
        s.mn_pre[b] = pre;
        s.mn_suf[b] = group_scan<T, true, false>(v);
        if ((lane & 15) == 15) s.gmn[0][b / RK_GRP] = pre;
      } else {
        const T v = s.mx0[b];
        const T pre = group_scan<T, false, true>(v);
        s.mx_pre[b] = pre;
        s.mx_suf[b] = group_scan<T, false, false>(v);
        if ((lane & 15) == 15) s.gmx[0][b / RK_GRP] = pre;
      }
    }
  }
  __syncthreads();
  // sparse table over the group extremes, in registers: wave 0 the minima, wave 1 the maxima;
  if (wave < 2) {
    T v = lane < RK_NGRP ? (wave == 0 ? s.gmn[0][lane] : s.gmx[0][lane])
                         : (wave == 0 ? MinMax<T>::min_identity() : MinMax<T>::max_identity());
    for (int lv = 1; lv < RK_GLEVELS; ++lv) {
      T t = __shfl_down(v, 1 << (lv - 1), 64);
      v = wave == 0 ? MinMax<T>::mn(v, t) : MinMax<T>::mx(v, t);
      if (wave == 0) s.gmn[lv][lane] = v;
      else s.gmx[lv][lane] = v;
    }
  }
  __syncthreads();
  RTICK();

  // ---- one output row per thread
  const int64_t i = t0 + tid;
#ifndef RK_TIMING
  if (i >= t1) return;
#endif
  const int li = (int)(i - rs);
  const uint32_t cki = s.ck[li];
  const uint32_t krel = cki >> 1;
  int a, b;
  {
    // first slot at or after the frame start: between the starts found for the 16th rows around
    const int g = tid >> 4;
    int hi = li;
    if (g + 1 < RK_ROWS / 16 && t0 + (g + 1) * 16 < t1) hi = s.a_lead[g + 1] < li ? s.a_lead[g + 1] : li;
    a = frame_first(krel, s.a_lead[g], hi);
    // last slot of the row's own second: gallop forward (usually the row itself or close)
    int step = 1;
    int lo = li + 1;
    hi = li + 1;
    const uint32_t peer_max = cki | 1u;  // the largest staged word of the row's own second
    while (hi < len && s.ck[hi] <= peer_max) { lo = hi + 1; hi += step; step <<= 1; }
    if (hi > len) hi = len;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (s.ck[mid] <= peer_max) lo = mid + 1;
      else hi = mid;
    }
    b = lo - 1;
  }
  RTICK();
  if (GENK) o.perm[i] = (int32_t)i;
  else if (perm) o.perm[i] = (int32_t)perm[i];  // null: the ordering step left the row ids in o.perm itself
  const bool open_left = a == 0 && rs > 0;          // the frame may start before the region
  const bool open_right = b == len - 1 && re < n;   // peers may continue after the region
  bool slow = open_left || open_right || krel >= RK_SAT;

  // head / tail sub-blocks slot by slot, whole sub-blocks from the prefixes and tables
  double sd = 0.0, sq = 0.0;
  uint32_t dc = 0;
  uint64_t isum = 0;
  T mn = MinMax<T>::min_identity(), mx = MinMax<T>::max_identity();
  const int sa = (int)((unsigned)a / (unsigned)RK_SB), sb = (int)((unsigned)b / (unsigned)RK_SB);  // 0 <= a <= b
  if (!slow) {
    // first and last sub-block: slot by slot up to the middle or the end of the sub-block,
    // whichever comes first; the half beyond the middle comes from the half aggregates
    constexpr int kHalf = RK_SB / 2;
    const bool head_half = sa != sb && (a & (RK_SB - 1)) < kHalf;   // head stops at the middle, + second half
    const bool tail_half = sa != sb && (b & (RK_SB - 1)) >= kHalf;  // first half + tail from the middle on
    // both walks are at most half a sub-block long: a frame inside one sub-block is cut after its
    // first four slots
    const int head_end = sa == sb ? (b < a + kHalf - 1 ? b : a + kHalf - 1)
                                  : (head_half ? sa * RK_SB + kHalf - 1 : (sa + 1) * RK_SB - 1);
    const int tail_begin = sa == sb ? head_end + 1 : (tail_half ? sb * RK_SB + kHalf : sb * RK_SB);
    if constexpr (!kInt) {
      if (!s.nonfinite) {
        // every staged value is finite and NULL slots hold NaN: no per-slot tests needed.  All
        // eight slots are read at once (one beyond a walk reads as NULL and adds nothing) instead
        // of one LDS round trip per step of two loops whose lengths differ from lane to lane.
        // A walk stays inside one sub-block, so its slots lie side by side in the padded array; a
        // slot beyond the walk's end is read from the element that always holds a NULL.
        // A walk stays inside one sub-block, so its slots lie side by side in the padded array: four
        // reads off one address each.  What lies beyond the walk's end is turned into a NULL by
        // replacing its upper half with a NaN's (one select per slot).
        const double *ph = &s.x[xi(a)], *pt = &s.x[xi(tail_begin < RK_CAP ? tail_begin : RK_CAP - 1)];
        const int nh = head_end - a + 1, nt = b - tail_begin + 1;  // slots in the walks (0 .. kHalf)
        double hv[kHalf], tv[kHalf];
#pragma unroll
        for (int u = 0; u < kHalf; ++u) { hv[u] = ph[u]; tv[u] = pt[u]; }
        constexpr int kNanHi = 0x7ff80000;
#pragma unroll
        for (int u = 0; u < kHalf; ++u) {
          hv[u] = __hiloint2double(u < nh ? __double2hiint(hv[u]) : kNanHi, __double2loint(hv[u]));
          tv[u] = __hiloint2double(u < nt ? __double2hiint(tv[u]) : kNanHi, __double2loint(tv[u]));
        }
#pragma unroll
        for (int u = 0; u < kHalf; ++u) slot_add_f64(hv[u], pivot, sd, sq, dc, mn, mx);
#pragma unroll
        for (int u = 0; u < kHalf; ++u) slot_add_f64(tv[u], pivot, sd, sq, dc, mn, mx);
      }
    }
    if (kInt || s.nonfinite) {
      for (int e = a; e <= b; e = (e == head_end ? tail_begin : e + 1)) {
        if (s.ck[e] & 1u) {
          T v = s.x[xi(e)];
          if (Ord<T>::finite(v)) {
            double d = (double)v - pivot;
            sd += d;
            sq += d * d;
            ++dc;
            if (kInt) isum += (uint64_t)v;
            mn = MinMax<T>::mn(mn, v);
            mx = MinMax<T>::mx(mx, v);
          } else {
            dc += 0x10000u;
          }
        }
      }
    }
    if (head_half) {
      sd += s.hs2[sa];
      sq += s.hq2[sa];
      dc += s.hc2[sa];
      if (kInt) isum += s.hi2[sa];
      mn = MinMax<T>::mn(mn, s.hmn2[sa]);
      mx = MinMax<T>::mx(mx, s.hmx2[sa]);
    }
    if (tail_half) {  // first half = whole sub-block - second half
      sd += (s.bs[sb + 1] - s.bs[sb]) - s.hs2[sb];
      sq += (s.bq[sb + 1] - s.bq[sb]) - s.hq2[sb];
      dc += (s.bc[sb + 1] - s.bc[sb]) - s.hc2[sb];
      if (kInt) isum += (s.bi[sb + 1] - s.bi[sb]) - s.hi2[sb];
      mn = MinMax<T>::mn(mn, s.hmn1[sb]);
      mx = MinMax<T>::mx(mx, s.hmx1[sb]);
    }
    RTICK();
    const int m0 = sa + 1, m1 = sb - 1;
    if (m0 <= m1) {
      sd += s.bs[m1 + 1] - s.bs[m0];
      sq += s.bq[m1 + 1] - s.bq[m0];
      dc += s.bc[m1 + 1] - s.bc[m0];
      if (kInt) isum += s.bi[m1 + 1] - s.bi[m0];
      const int g0 = m0 / RK_GRP, g1 = m1 / RK_GRP;
      if (g0 == g1) {  // inside one group: its sub-blocks one by one (at most 16)
        for (int sbk = m0; sbk <= m1; ++sbk) {
          mn = MinMax<T>::mn(mn, s.mn0[sbk]);
          mx = MinMax<T>::mx(mx, s.mx0[sbk]);
        }
      } else {
        mn = MinMax<T>::mn(mn, MinMax<T>::mn(s.mn_suf[m0], s.mn_pre[m1]));
        mx = MinMax<T>::mx(mx, MinMax<T>::mx(s.mx_suf[m0], s.mx_pre[m1]));
        const int inner = g1 - g0 - 1;  // whole groups in between
        if (inner > 0) {
          const int lv = 31 - __clz(inner);
          mn = MinMax<T>::mn(mn, MinMax<T>::mn(s.gmn[lv][g0 + 1], s.gmn[lv][g1 - (1 << lv)]));
          mx = MinMax<T>::mx(mx, MinMax<T>::mx(s.gmx[lv][g0 + 1], s.gmx[lv][g1 - (1 << lv)]));
        }
      }
    }
    slow = (dc >> 16) != 0;  // a non-finite value inside the frame
  }
  RTICK();
  // A frame that leaves the staged region (or a region too wide for 32-bit relative words) is
  // not evaluated here: the row is marked and the wide path completes it.
  const bool defer = open_left || open_right || krel >= RK_SAT || ((GENK || W32) && slow);  // (no 64-bit words laid out for the serial path)
  if (__any(defer) && lane == 0 && __atomic_load_n(deferred, __ATOMIC_RELAXED) == 0) atomicOr(deferred, 1);
  if (defer) {
    o.valid[i] = kDeferred;
    return;
  }
  if (slow) {
    // a non-finite value inside the frame: exact sequential evaluation (Spark's own recurrence)
    if constexpr (!GENK && !W32) rs_row_serial<T>(comp, xs, rs + a, rs + b, i, o);
    return;
  }
  const int64_t cnt = dc & 0xffffu;
  const double nd = (double)cnt;
  // 1 / n and 1 / (n - 1) off one reciprocal (the instruction runs at a quarter of the rate of a
  // multiplication): r = 1 / (n (n - 1)), the product being exact
  const double nm1 = nd - 1.0;
  const double r_both = cnt > 1 ? fast_rcp(nd * nm1) : 1.0;
  const double inv_n = cnt > 1 ? r_both * nm1 : (cnt ? 1.0 : 0.0);
  const double mean = pivot + sd * inv_n;  // (an empty frame stores 0 below)
  T sum;
  if constexpr (kInt) sum = (T)isum;
  else sum = (T)(nd * pivot + sd);
  double stddev = 0.0, inv_sd = 0.0;
  if (cnt > 1) {
    const bool same = mn == mx;  // all values identical: Spark's m2 is exactly 0
    double m2 = same ? 0.0 : sq - sd * sd * inv_n;
    // The prefix difference carries an absolute error of a few ulps of the running total.
    // When the frame's spread is tiny against that total, redo it as a corrected two-pass
    // sum around the frame mean (rare: near-constant frames inside a varying region).
    // (the running total up to and including the frame's last sub-block: where the frame ends in the
    // first half of that sub-block its part was taken as whole - second half, at the size of the whole)
    if (!same && m2 < 1e-5 * (s.bq[sb + 1] + sq)) {
      double acc = 0.0, lin = 0.0;
      for (int e = a; e <= b; ++e) {
        if (s.ck[e] & 1u) {
          double dv = (double)s.x[xi(e)] - mean;
          acc += dv * dv;
          lin += dv;
        }
      }
      m2 = acc - lin * lin / nd;
    }
    if (m2 < 0.0) m2 = 0.0;
    stddev = sqrt_and_rcp(m2 * (r_both * nd), &inv_sd);
  }
  // rs_store's arithmetic, with the reciprocal form of the zscore division
  {
    const bool okk = cnt > 0, sd_ok = cnt > 1;
    const uint32_t ti = (uint32_t)tid;  // the tile's first row is block-uniform: one offset register serves all columns
    (o.count + t0)[ti] = cnt;
    (o.mean + t0)[ti] = okk ? mean : 0.0;
    (o.mn + t0)[ti] = okk ? mn : (T)0;
    (o.mx + t0)[ti] = okk ? mx : (T)0;
    (o.sum + t0)[ti] = sum;        // an empty frame: 0 * pivot + 0
    (o.stddev + t0)[ti] = stddev;  // 0 unless the frame holds two values
    const bool z_ok = (cki & 1u) && sd_ok && stddev != 0.0;  // Spark Divide: NULL on a zero divisor
    (o.zscore + t0)[ti] = z_ok ? ((double)s.x[xi(li)] - mean) * inv_sd : 0.0;
    (o.valid + t0)[ti] = (uint8_t)((okk ? 1 : 0) | (sd_ok ? 2 : 0) | (z_ok ? 4 : 0));
  }
#ifdef RK_TIMING
  RTICK();
  if (tid == 0 && (blockIdx.x % 99991) == 777)
    printf("R tile %d: stage %lld aggr %lld scans %lld search %lld headtail %lld middle %lld epilogue %lld\n", tile,
           tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6]);
#endif
}

struct RsPlan {
  OrderBuffers ob;
  uint64_t *xs;
  int32_t *deferred;
  RsWidePlan wide;
  RsCellsPlan cells;
};

static void rs_plan(const tmp_range_stats_args *a, Workspace &ws, RsPlan *p, bool with_wide) {
  order_plan(a->n, ws, &p->ob, plan_keys(a->key, a->n_keys), 1);  // the metric travels with the rows
  p->xs = ws.take<uint64_t>((size_t)(a->n > 0 ? a->n : 1));
  p->deferred = ws.take<int32_t>(1);
  if (with_wide) rs_wide_plan(a->n, a->x_is_i64 != 0, ws, &p->wide);
  rs_cells_plan(a, ws, &p->cells);
}

static void rs_spec(const tmp_range_stats_args *a, OrderSpec *s) {
  memset(s, 0, sizeof(*s));
  s->n = a->n; s->key = a->key; s->n_keys = a->n_keys; s->ts = a->ts;
  s->prim_div = a->ts_is_seconds ? 1 : 1000000;
  s->n_masks = 1;
  s->masks[0] = a->x_valid;  // NULL pointer = every row valid
}

// One metric column in sorted order -> its seven output columns: the region kernel, then (when some
// frame left a block's region, which costs one small blocking read to learn) the wide path.
// Slots of the region the kernel stages for these rows: the smallest of the three sizes whose reach
// back (the region less the tile and the rows after it) is at least 1.25 x the rows a frame holds if
// the rows are spread evenly over the keys and the time span.  Rows bunched more than that leave the
// region and take the wide path -- as they do, later, with the largest region.  TEMPO_RK_CAP pins a size.
static int rs_region_cap(int64_t n, int64_t n_keys, const Layout &lay, uint64_t w) {
  if (const char *e = getenv("TEMPO_RK_CAP")) {
    const int c = atoi(e);
    if (c == 1536 || c == 2048 || c == 2560) return c;
  }
  const double span = (double)((uint64_t)lay.prim_max - (uint64_t)lay.prim_min) + 1.0;
  const double est = ((double)w + 1.0) * (double)n / (span * (double)(n_keys > 0 ? n_keys : 1));
  if (est * 1.25 <= 1536 - RK_ROWS - RK_FWD) return 1536;
  if (est * 1.25 <= 2048 - RK_ROWS - RK_FWD) return 2048;
  return 2560;
}

template <typename T, bool GENK, bool W32 = false>
static void rs_launch(int cap, int grid, hipStream_t st, const OrderResult &res, const uint32_t *perm, const T *xs, int64_t n,
                      const Layout &lay, uint64_t w, const RsOut<T> &o, int32_t *deferred_flag, int n_tiles, const RsSrc &src) {
  ProfScope ps(GENK ? "range_stats_kernel<T>+words" : (W32 ? "range_stats_kernel<T>/words32" : "range_stats_kernel<T>"), st);
  switch (cap) {
    case 1536:
      hipLaunchKernelGGL((range_stats_kernel<T, 1536, GENK, W32>), dim3(grid), dim3(RK_THREADS), 0, st, res.comp, perm, xs, n, lay,
                         w, o, deferred_flag, n_tiles, src);
      break;
    case 2048:
      hipLaunchKernelGGL((range_stats_kernel<T, 2048, GENK, W32>), dim3(grid), dim3(RK_THREADS), 0, st, res.comp, perm, xs, n, lay,
                         w, o, deferred_flag, n_tiles, src);
      break;
    default:
      hipLaunchKernelGGL((range_stats_kernel<T, 2560, GENK, W32>), dim3(grid), dim3(RK_THREADS), 0, st, res.comp, perm, xs, n, lay,
                         w, o, deferred_flag, n_tiles, src);
  }
}

constexpr int kRsNeedWords = 1000;  // rs_column with `src`: some rows need the ordinary form over laid-out words

// src != nullptr: the rows are in order where they lie; the kernel builds the sort words itself (res is
// not used).  Returns kRsNeedWords if some row could not be finished that way.
template <typename T>
static int rs_column(const OrderResult &res, const T *xs, int64_t n, int64_t n_keys, const Layout &lay, uint64_t w,
                     const RsOut<T> &o, int32_t *deferred_flag, bool have_wide, RsWidePlan &wide, hipStream_t st,
                     const RsSrc *src = nullptr) {
  const int n_tiles = (int)((n + RK_ROWS - 1) / RK_ROWS);
  int grid = (n_tiles + 7) / 8 * 8;
  const uint32_t *perm = (src || (const void *)res.perm == (const void *)o.perm) ? nullptr : res.perm;
  RsSrc none;
  memset(&none, 0, sizeof(none));
  // the region picked from the expected rows per frame; if some frame left it after all (rows bunched
  // where the estimate assumed an even spread), once more with the largest region before anything else
  int cap = rs_region_cap(n, n_keys, lay, w);
  for (;;) {
    TMP_CHECK_HIP(hipMemsetAsync(deferred_flag, 0, sizeof(int32_t), st));
    if (src) rs_launch<T, true>(cap, grid, st, res, perm, xs, n, lay, w, o, deferred_flag, n_tiles, *src);
    else if (res.narrow) rs_launch<T, false, true>(cap, grid, st, res, perm, xs, n, lay, w, o, deferred_flag, n_tiles, none);
    else rs_launch<T, false>(cap, grid, st, res, perm, xs, n, lay, w, o, deferred_flag, n_tiles, none);
    TMP_CHECK_HIP(hipGetLastError());
    int32_t deferred = 0;
    TMP_CHECK_HIP(hipMemcpyAsync(&deferred, deferred_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TMP_CHECK_HIP(hipStreamSynchronize(st));
    if (!deferred) return TMP_OK;
    if (cap == RK_CAP_MAX || getenv("TEMPO_RK_CAP")) break;
    cap = RK_CAP_MAX;
  }
  if (src || res.narrow) return kRsNeedWords;
  if (!have_wide) {
    set_error("tmp_range_stats: some frames are longer than the on-chip region; call again with a workspace of "
              "tmp_range_stats_wide_workspace_bytes() bytes");
    return TMP_EWIDE;
  }
  return rs_wide_run<T>(res.comp, xs, n, lay, w, o, wide, st);
}

template <typename T>
static int rs_run(const tmp_range_stats_args *a, RsPlan &p, bool have_wide, hipStream_t st,
                  const RangeInfo *known = nullptr) {
  OrderSpec s;
  rs_spec(a, &s);
  RangeInfo r;
  if (known) {
    r = *known;
  } else {
    TMP_TRY(order_scan_ranges(s, p.ob, st));
    TMP_TRY(order_read_ranges(p.ob, &r, st));
  }
  Layout lay;
  TMP_TRY(order_make_layout(s, r, 1, &lay));
  OrderResult res;
  RsOut<T> o;
  o.perm = a->out_perm; o.mean = a->out_mean; o.count = a->out_count;
  o.mn = (T *)a->out_min; o.mx = (T *)a->out_max; o.sum = (T *)a->out_sum;
  o.stddev = a->out_stddev; o.zscore = a->out_zscore; o.valid = a->out_valid;
  o.vbit = 0;
  // rows that arrive in (key, second) order: nothing is laid out, the statistics kernel reads the
  // table's own columns (TEMPO_NO_STATS_IN_PLACE=1: the words are laid out first, as for any table)
  if (sorted_mode(r) == 1 && a->n > 0 && !getenv("TEMPO_NO_STATS_IN_PLACE")) {
    RsSrc src;
    src.key = a->key; src.ts = a->ts; src.valid = a->x_valid;
    res.comp = nullptr; res.perm = nullptr; res.narrow = false;
    const int rc = rs_column<T>(res, (const T *)a->x, a->n, a->n_keys, lay, (uint64_t)a->range_back_secs, o, p.deferred,
                                have_wide, p.wide, st, &src);
    if (rc != kRsNeedWords) return rc;
  }
  // the metric column rides along with the last radix pass; if no pass runs the rows are
  // already in order and the column is used where it is
  bool carried = false;
  const uint64_t *pay_src[1] = {(const uint64_t *)a->x};
  uint64_t *pay_out[1] = {p.xs};
  // sort words that fit 32 bits may come as such (half the bytes of that stream, written and read
  // once each); if some row then needs the 64-bit words after all, the rows are laid out once more
  p.ob.narrow_words = true;
  TMP_TRY(order_build(s, lay, sorted_mode(r), p.ob, &res, st, 1, pay_src, pay_out, &carried));
  const T *xs = carried ? (const T *)p.xs : (const T *)a->x;
  int rc = rs_column<T>(res, xs, a->n, a->n_keys, lay, (uint64_t)a->range_back_secs, o, p.deferred, have_wide, p.wide, st);
  if (rc != kRsNeedWords) return rc;
  p.ob.narrow_words = false;
  TMP_TRY(order_build(s, lay, sorted_mode(r), p.ob, &res, st, 1, pay_src, pay_out, &carried));
  xs = carried ? (const T *)p.xs : (const T *)a->x;
  return rs_column<T>(res, xs, a->n, a->n_keys, lay, (uint64_t)a->range_back_secs, o, p.deferred, have_wide, p.wide, st);
}

// Several metric columns over one ordering (tmp_range_stats_cols).
struct RsColsPlan {
  OrderBuffers ob;
  uint64_t *xs[TMP_MAX_STATS_COLS];
  int32_t *deferred;
  RsWidePlan wide;
};

static void rs_cols_plan(const tmp_range_stats_cols_args *a, Workspace &ws, RsColsPlan *p, bool with_wide) {
  const int nc = a->n_cols < 1 ? 1 : (a->n_cols > TMP_MAX_STATS_COLS ? TMP_MAX_STATS_COLS : a->n_cols);
  order_plan(a->n, ws, &p->ob, plan_keys(a->key, a->n_keys), nc);
  for (int c = 0; c < TMP_MAX_STATS_COLS; ++c)
    p->xs[c] = c < nc ? ws.take<uint64_t>((size_t)(a->n > 0 ? a->n : 1)) : nullptr;
  p->deferred = ws.take<int32_t>(1);
  bool any_i64 = false;
  for (int c = 0; c < nc; ++c) any_i64 = any_i64 || (a->x_is_i64 && a->x_is_i64[c]);
  if (with_wide) rs_wide_plan(a->n, any_i64, ws, &p->wide);  // one set of buffers, used column after column
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_range_stats_workspace_bytes(const tmp_range_stats_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RsPlan p;
  rs_plan(a, ws, &p, false);
  return ws.off;
}

extern "C" size_t tmp_range_stats_wide_workspace_bytes(const tmp_range_stats_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RsPlan p;
  rs_plan(a, ws, &p, true);
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
  // a workspace large enough for the wide path enables it; the plain size serves every table
  // whose frames stay inside the on-chip region
  const bool have_wide = workspace_bytes >= tmp_range_stats_wide_workspace_bytes(a);
  Workspace ws(workspace, workspace_bytes);
  RsPlan p;
  rs_plan(a, ws, &p, have_wide);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_range_stats: workspace too small");
  // the second row-id buffer of the ordering step is the caller's out_perm: when the order ends up there
  // (a single pass does), the row ids are where they belong and the kernel neither reads nor copies them
  if (((uintptr_t)a->out_perm & 255) == 0) p.ob.idx[1] = (uint32_t *)a->out_perm;  // aligned like the workspace's own
  hipStream_t st = (hipStream_t)stream;
  RangeInfo range;
  const RangeInfo *known = nullptr;
  if (p.cells.planned) {  // any_order: a time-ordered feed over few (key, second) cells is not ordered at all
    OrderSpec s;
    rs_spec(a, &s);
    TMP_TRY(order_scan_ranges(s, p.ob, st));
    TMP_TRY(order_read_ranges(p.ob, &range, st));
    known = &range;
    bool done = false;
    TMP_TRY(rs_cells_try(a, p.cells, range, st, &done));
    if (done) return TMP_OK;
  }
  return a->x_is_i64 ? rs_run<int64_t>(a, p, have_wide, st, known) : rs_run<double>(a, p, have_wide, st, known);
}

extern "C" size_t tmp_range_stats_cols_workspace_bytes(const tmp_range_stats_cols_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RsColsPlan p;
  rs_cols_plan(a, ws, &p, false);
  return ws.off;
}

extern "C" size_t tmp_range_stats_cols_wide_workspace_bytes(const tmp_range_stats_cols_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  RsColsPlan p;
  rs_cols_plan(a, ws, &p, true);
  return ws.off;
}

extern "C" int tmp_range_stats_cols(const tmp_range_stats_cols_args *a, void *workspace, size_t workspace_bytes,
                                    void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_range_stats_cols: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_range_stats_cols: row count must be in [0, 2^31)");
  TMP_REQUIRE(a->range_back_secs >= 0, TMP_EINVAL, "tmp_range_stats_cols: range_back_secs must be >= 0");
  TMP_REQUIRE(a->n_cols >= 1 && a->n_cols <= TMP_MAX_STATS_COLS, TMP_EINVAL,
              "tmp_range_stats_cols: n_cols must be in [1, %d]", TMP_MAX_STATS_COLS);
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->x && a->x_valid && a->x_is_i64 && a->out_perm && a->out_mean && a->out_count && a->out_min &&
                  a->out_max && a->out_sum && a->out_stddev && a->out_zscore && a->out_valid,
              TMP_EINVAL, "tmp_range_stats_cols: null pointer");
  for (int c = 0; c < a->n_cols; ++c)
    TMP_REQUIRE(a->x[c] && a->out_mean[c] && a->out_count[c] && a->out_min[c] && a->out_max[c] && a->out_sum[c] &&
                    a->out_stddev[c] && a->out_zscore[c] && a->out_valid[c],
                TMP_EINVAL, "tmp_range_stats_cols: column %d has a null pointer", c);
  const bool have_wide = workspace_bytes >= tmp_range_stats_cols_wide_workspace_bytes(a);
  Workspace ws(workspace, workspace_bytes);
  RsColsPlan p;
  rs_cols_plan(a, ws, &p, have_wide);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_range_stats_cols: workspace too small");
  if (((uintptr_t)a->out_perm & 255) == 0) p.ob.idx[1] = (uint32_t *)a->out_perm;  // see tmp_range_stats
  hipStream_t st = (hipStream_t)stream;
  // one ordering for all columns: column c's "not NULL" flag rides in bit c of the sort words and
  // its values travel with the rows
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts;
  s.prim_div = a->ts_is_seconds ? 1 : 1000000;
  s.n_masks = a->n_cols;
  for (int c = 0; c < a->n_cols; ++c) s.masks[c] = a->x_valid[c];
  RangeInfo r;
  TMP_TRY(order_scan_ranges(s, p.ob, st));
  TMP_TRY(order_read_ranges(p.ob, &r, st));
  Layout lay;
  TMP_TRY(order_make_layout(s, r, a->n_cols, &lay));
  OrderResult res;
  bool carried = false;
  const uint64_t *pay_src[TMP_MAX_STATS_COLS];
  for (int c = 0; c < a->n_cols; ++c) pay_src[c] = (const uint64_t *)a->x[c];
  TMP_TRY(order_build(s, lay, sorted_mode(r), p.ob, &res, st, a->n_cols, pay_src, p.xs, &carried));
  for (int c = 0; c < a->n_cols; ++c) {
    const void *xs = carried ? (const void *)p.xs[c] : a->x[c];
    int rc;
    if (a->x_is_i64[c]) {
      RsOut<int64_t> o;
      o.perm = a->out_perm; o.mean = a->out_mean[c]; o.count = a->out_count[c];
      o.mn = (int64_t *)a->out_min[c]; o.mx = (int64_t *)a->out_max[c]; o.sum = (int64_t *)a->out_sum[c];
      o.stddev = a->out_stddev[c]; o.zscore = a->out_zscore[c]; o.valid = a->out_valid[c];
      o.vbit = c;
      rc = rs_column<int64_t>(res, (const int64_t *)xs, a->n, a->n_keys, lay, (uint64_t)a->range_back_secs, o, p.deferred,
                              have_wide, p.wide, st);
    } else {
      RsOut<double> o;
      o.perm = a->out_perm; o.mean = a->out_mean[c]; o.count = a->out_count[c];
      o.mn = (double *)a->out_min[c]; o.mx = (double *)a->out_max[c]; o.sum = (double *)a->out_sum[c];
      o.stddev = a->out_stddev[c]; o.zscore = a->out_zscore[c]; o.valid = a->out_valid[c];
      o.vbit = c;
      rc = rs_column<double>(res, (const double *)xs, a->n, a->n_keys, lay, (uint64_t)a->range_back_secs, o, p.deferred,
                             have_wide, p.wide, st);
    }
    TMP_TRY(rc);
  }
  return TMP_OK;
}
