// Ordering step shared by every op: (key, primary[, seq]) -> 64-bit sort word -> stable LSD
// radix sort of (sort word, row id) pairs.  Replaces Spark's Exchange + SortExec in front of
// WindowExec / aggregates (python/tempo/tsdf.py:117-121, :563-572; resample.py:63).
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace tmp {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void clear_error() { g_err[0] = 0; }

// ---- per-kernel event timing (diagnostic; one device at a time).  Launch slots are handed out
//      by an atomic counter, so operators called from several host threads each keep their own
//      slots; launches beyond the pool are counted, not recorded.  Enable / reset / read-out are
//      for one thread while no operator call is in flight.
constexpr int kProfSlots = 8192;
static std::atomic<bool> g_prof_on{false};
static std::atomic<int> g_prof_n{0};        // slots handed out (may run past kProfSlots)
static hipEvent_t *g_prof_ev = nullptr;  // 2 * kProfSlots events, created on first enable
static const char *g_prof_name[kProfSlots];
static int g_prof_tag[kProfSlots];
static int g_prof_cur_tag = 0;

ProfScope::ProfScope(const char *name, hipStream_t s) : slot(-1), st(s) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  const int got = g_prof_n.fetch_add(1, std::memory_order_relaxed);
  if (got >= kProfSlots) return;  // pool exhausted: tmp_profile_dropped() reports how many
  slot = got;
  g_prof_name[slot] = name;
  g_prof_tag[slot] = g_prof_cur_tag;
  (void)hipEventRecord(g_prof_ev[2 * slot], st);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof_ev[2 * slot + 1], st);
}

// ------------------------------------------------------------------------------------------
// Pass 0: value ranges + "already in order" test, one streaming read of key/ts/seq.
// ------------------------------------------------------------------------------------------
__global__ void range_init_kernel(RangeInfo *r) {
  r->prim_min = INT64_MAX;
  r->prim_max = INT64_MIN;
  r->seq_min = INT64_MAX;
  r->seq_max = INT64_MIN;
  r->unsorted = 0;
  r->time_unsorted = 0;
  r->not_thousands = 0;
  r->not_millions = 0;
  r->n_desc = 0;
  r->key_pairs = 0;
  r->key_viol = 0;
  r->key_hist = 0;
  for (int i = 0; i < kMaxDesc; ++i) r->desc_pos[i] = 0;
}

// Rows of this wave's current step at which time steps back (`down`): counted, and the first few
// noted, while the count is still small enough to matter (a table in random order stops touching
// the counter after the first waves).
__device__ __forceinline__ void note_descents(bool down, int64_t i, RangeInfo *out) {
  const uint64_t bal = __ballot(down);
  if (!bal) return;
  const int lane = threadIdx.x & 63;
  const int first = __ffsll((long long)bal) - 1;
  int base = kMaxDesc + 1;
  if (lane == first && __atomic_load_n(&out->n_desc, __ATOMIC_RELAXED) <= kMaxDesc)
    base = atomicAdd(&out->n_desc, (int)__popcll(bal));
  base = __shfl(base, first, 64);
  if (down) {
    const int slot = base + (int)__popcll(bal & ((1ull << lane) - 1));
    if (slot < kMaxDesc) out->desc_pos[slot] = i;
  }
}

__device__ __forceinline__ int64_t wave_min(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) {
    int64_t t = __shfl_down(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int64_t wave_max(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) {
    int64_t t = __shfl_down(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// x is a multiple of 1000 (= 8 * 125) / of 1000000 (= 64 * 15625): low bits clear and the odd
// factor divides |x|, tested by multiplying with its inverse modulo 2^64 -- no division.
__device__ __forceinline__ bool multiple_of_1000(int64_t x) {
  uint64_t u = x < 0 ? 0ull - (uint64_t)x : (uint64_t)x;
  return (u & 7ull) == 0 && u * 0x1cac083126e978d5ull <= 0x20c49ba5e353f7cull;
}
__device__ __forceinline__ bool multiple_of_1000000(int64_t x) {
  uint64_t u = x < 0 ? 0ull - (uint64_t)x : (uint64_t)x;
  return (u & 63ull) == 0 && u * 0x790fb65668c26139ull <= 0x431bde82d7b63ull;
}

__global__ __launch_bounds__(kBlock) void range_scan_kernel(int64_t n, const int32_t *key,
                                                            const int64_t *ts, int64_t div,
                                                            const int64_t *seq,
                                                            const uint8_t *seq_valid,
                                                            RangeInfo *out, int skip_if_key_hist) {
  if (skip_if_key_hist && out->key_hist) return;  // the form that also counts the keys is doing the job
  int64_t pmin = INT64_MAX, pmax = INT64_MIN, smin = INT64_MAX, smax = INT64_MIN;
  int bad = 0, tbad = 0, nk = 0, nm = 0;
  // granularity is looked for in raw timestamp columns only, and a wave stops looking once one
  // of its lanes has met a value that is not a whole millisecond (wave-uniform branch)
  bool gran = div == 1;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t first = 0;  // rows below `first` are covered by the four-rows-per-thread loop
  if (!seq && (((uintptr_t)ts | (uintptr_t)key) & 15) == 0) {
    const int64_t n4 = n >> 2;
    first = n4 << 2;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += stride) {
      longlong2 a = reinterpret_cast<const longlong2 *>(ts)[2 * c];
      longlong2 b = reinterpret_cast<const longlong2 *>(ts)[2 * c + 1];
      int4 k = key ? reinterpret_cast<const int4 *>(key)[c] : make_int4(0, 0, 0, 0);
      int64_t t[4] = {a.x, a.y, b.x, b.y};
      int32_t kk[4] = {k.x, k.y, k.z, k.w};
      int64_t tp = c > 0 ? ts[4 * c - 1] : t[0];
      int32_t kp = (c > 0 && key) ? key[4 * c - 1] : kk[0];
      if (div != 1) {
        tp = floordiv_i64(tp, div);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = floordiv_i64(t[j], div);
      }
      if (c == 0) tp = t[0];
      bool down[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (gran) {
          nk |= !multiple_of_1000(t[j]);
          nm |= !multiple_of_1000000(t[j]);
        }
        pmin = t[j] < pmin ? t[j] : pmin;
        pmax = t[j] > pmax ? t[j] : pmax;
        bool tdown = tp > t[j];
        tbad |= tdown;
        bad |= kp != kk[j] ? kp > kk[j] : tdown;
        tp = t[j];
        kp = kk[j];
        down[j] = tdown;
      }
      if (gran && __any(nk)) gran = false;
      if (__any(down[0] | down[1] | down[2] | down[3])) {
#pragma unroll
        for (int j = 0; j < 4; ++j) note_descents(down[j], 4 * c + j, out);
      }
    }
  }
  for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int64_t p = div == 1 ? ts[i] : floordiv_i64(ts[i], div);
    if (gran) {
      nk |= !multiple_of_1000(p);
      nm |= !multiple_of_1000000(p);
      if (__any(nk)) gran = false;
    }
    pmin = p < pmin ? p : pmin;
    pmax = p > pmax ? p : pmax;
    bool sv = false;
    int64_t s = 0;
    if (seq) {
      sv = !seq_valid || seq_valid[i];
      if (sv) {
        s = seq[i];
        smin = s < smin ? s : smin;
        smax = s > smax ? s : smax;
      }
    }
    bool down = false;
    if (i > 0) {  // lexicographic (key, primary, seq NULLS FIRST) against the previous row
      int32_t k0 = key ? key[i - 1] : 0, k1 = key ? key[i] : 0;
      int64_t p0 = div == 1 ? ts[i - 1] : floordiv_i64(ts[i - 1], div);
      int tcmp = p0 > p ? 1 : (p0 < p ? -1 : 0);  // (primary, seq) of the previous row vs this one
      if (tcmp == 0 && seq) {
        bool sv0 = !seq_valid || seq_valid[i - 1];
        if (sv0 != sv) tcmp = sv0 ? 1 : -1;  // NULL sorts first
        else if (sv) {
          int64_t s0 = seq[i - 1];
          tcmp = s0 > s ? 1 : (s0 < s ? -1 : 0);
        }
      }
      tbad |= tcmp > 0;
      down = tcmp > 0;
      if (k0 != k1) bad |= k0 > k1;
      else bad |= tcmp > 0;
    }
    if (__any(down)) note_descents(down, i, out);
  }
  pmin = wave_min(pmin);
  pmax = wave_max(pmax);
  smin = wave_min(smin);
  smax = wave_max(smax);
  bad = __any(bad);
  tbad = __any(tbad);
  if (div != 1) nk = 1;
  nk = __any(nk);
  nm = __any(nm) | nk;
  // Every wave reports into one small record: each field is read first and only touched by an
  // atomic when this wave would change it, which after the first few waves is almost never.
  if ((threadIdx.x & 63) == 0) {
    if (pmin < __atomic_load_n(&out->prim_min, __ATOMIC_RELAXED)) atomicMin((long long *)&out->prim_min, (long long)pmin);
    if (pmax > __atomic_load_n(&out->prim_max, __ATOMIC_RELAXED)) atomicMax((long long *)&out->prim_max, (long long)pmax);
    if (smin < __atomic_load_n(&out->seq_min, __ATOMIC_RELAXED)) atomicMin((long long *)&out->seq_min, (long long)smin);
    if (smax > __atomic_load_n(&out->seq_max, __ATOMIC_RELAXED)) atomicMax((long long *)&out->seq_max, (long long)smax);
    if (bad && !__atomic_load_n(&out->unsorted, __ATOMIC_RELAXED)) atomicOr(&out->unsorted, 1);
    if (tbad && !__atomic_load_n(&out->time_unsorted, __ATOMIC_RELAXED)) atomicOr(&out->time_unsorted, 1);
    if (nk && !__atomic_load_n(&out->not_thousands, __ATOMIC_RELAXED)) atomicOr(&out->not_thousands, 1);
    if (nm && !__atomic_load_n(&out->not_millions, __ATOMIC_RELAXED)) atomicOr(&out->not_millions, 1);
  }
}

// range_scan_kernel for a table with a key column and no sequence column, in the block geometry of
// the first radix pass (block b owns tiles_per_block tiles of RH_TILE consecutive rows): besides the
// ranges it leaves counts[digit][block] of the key column's low bits (key & mask) -- the histogram a
// pass over the key field needs -- so that a time-ordered feed is not read a second time for it.
// The key and time columns are 16-byte aligned (the caller has checked).
constexpr int RH_THREADS = 1024;

// 1024 pairs of adjacent rows spread over the table: key_hist = 1 if time steps back in none of them.
// Steers which of the two scans runs; what it guesses wrong costs time, never correctness.
__global__ __launch_bounds__(256) void time_order_sample_kernel(int64_t n, const int64_t *ts, RangeInfo *out) {
  __shared__ int bad_s;
  if (threadIdx.x == 0) bad_s = 0;
  __syncthreads();
  bool bad = false;
  const int64_t step = (n - 1) / 1024;
  for (int j = 0; j < 4; ++j) {
    const int64_t p = (int64_t)(threadIdx.x * 4 + j) * step;
    if (p + 1 < n) bad |= ts[p] > ts[p + 1];
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) bad_s = 1;
  __syncthreads();
  if (threadIdx.x == 0) out->key_hist = bad_s ? 0 : 1;
}

template <int RH_TILE>
__global__ __launch_bounds__(RH_THREADS) void range_scan_hist_kernel(int64_t n, const int32_t *key, const int64_t *ts,
                                                                     int64_t div, RangeInfo *out, uint32_t *counts,
                                                                     int tiles_per_block, uint32_t mask) {
  constexpr int RH_WAVES = RH_THREADS / 64;
  if (!out->key_hist) return;  // the sample says the rows are not time-ordered: the plain scan runs instead
  __shared__ uint32_t hist[RH_WAVES][256];
  for (int i = threadIdx.x; i < RH_WAVES * 256; i += RH_THREADS) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  int64_t pmin = INT64_MAX, pmax = INT64_MIN;
  int bad = 0, tbad = 0, nk = 0, nm = 0;
  bool gran = div == 1;
  const int64_t begin = (int64_t)blockIdx.x * tiles_per_block * RH_TILE;
  int64_t end = begin + (int64_t)tiles_per_block * RH_TILE;
  if (end > n) end = n;
  const int64_t c_end = end >> 2;  // whole groups of four rows (begin is a multiple of four)
  for (int64_t c = (begin >> 2) + threadIdx.x; c < c_end; c += RH_THREADS) {
    longlong2 a = reinterpret_cast<const longlong2 *>(ts)[2 * c];
    longlong2 b = reinterpret_cast<const longlong2 *>(ts)[2 * c + 1];
    int4 k = reinterpret_cast<const int4 *>(key)[c];
    int64_t t[4] = {a.x, a.y, b.x, b.y};
    int32_t kk[4] = {k.x, k.y, k.z, k.w};
    int64_t tp = c > 0 ? ts[4 * c - 1] : t[0];
    int32_t kp = c > 0 ? key[4 * c - 1] : kk[0];
    if (div != 1) {
      tp = floordiv_i64(tp, div);
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = floordiv_i64(t[j], div);
    }
    if (c == 0) tp = t[0];
    bool down[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (gran) {
        nk |= !multiple_of_1000(t[j]);
        nm |= !multiple_of_1000000(t[j]);
      }
      pmin = t[j] < pmin ? t[j] : pmin;
      pmax = t[j] > pmax ? t[j] : pmax;
      bool tdown = tp > t[j];
      tbad |= tdown;
      bad |= kp != kk[j] ? kp > kk[j] : tdown;
      tp = t[j];
      kp = kk[j];
      down[j] = tdown;
    }
    // keys that come in long stretches (a table ordered by key) would send all 64 lanes to one
    // counter: a wave whose lanes all hold one key adds its rows up in a single update
    const uint32_t k_first = (uint32_t)__builtin_amdgcn_readfirstlane(kk[0]) & mask;
    bool one_key = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) one_key &= ((uint32_t)kk[j] & mask) == k_first;
    if (__all(one_key)) {
      const uint64_t act = __ballot(true);
      if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&hist[wave][k_first], 4u * (uint32_t)__popcll(act));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&hist[wave][(uint32_t)kk[j] & mask], 1u);
    }
    if (gran && __any(nk)) gran = false;
    if (__any(down[0] | down[1] | down[2] | down[3])) {
#pragma unroll
      for (int j = 0; j < 4; ++j) note_descents(down[j], 4 * c + j, out);
    }
  }
  // the last rows of a table whose length is not a multiple of four belong to the last block
  if (end == n) {
    for (int64_t i = (c_end << 2) + threadIdx.x; i < n; i += RH_THREADS) {
      int64_t p = div == 1 ? ts[i] : floordiv_i64(ts[i], div);
      if (gran) {
        nk |= !multiple_of_1000(p);
        nm |= !multiple_of_1000000(p);
      }
      pmin = p < pmin ? p : pmin;
      pmax = p > pmax ? p : pmax;
      bool down = false;
      if (i > 0) {
        const int32_t k0 = key[i - 1], k1 = key[i];
        const int64_t p0 = div == 1 ? ts[i - 1] : floordiv_i64(ts[i - 1], div);
        down = p0 > p;
        tbad |= down;
        bad |= k0 != k1 ? k0 > k1 : down;
      }
      atomicAdd(&hist[wave][(uint32_t)key[i] & mask], 1u);
      if (__any(down)) note_descents(down, i, out);
    }
  }
  pmin = wave_min(pmin);
  pmax = wave_max(pmax);
  bad = __any(bad);
  tbad = __any(tbad);
  if (div != 1) nk = 1;
  nk = __any(nk);
  nm = __any(nm) | nk;
  if ((threadIdx.x & 63) == 0) {
    if (pmin < __atomic_load_n(&out->prim_min, __ATOMIC_RELAXED)) atomicMin((long long *)&out->prim_min, (long long)pmin);
    if (pmax > __atomic_load_n(&out->prim_max, __ATOMIC_RELAXED)) atomicMax((long long *)&out->prim_max, (long long)pmax);
    if (bad && !__atomic_load_n(&out->unsorted, __ATOMIC_RELAXED)) atomicOr(&out->unsorted, 1);
    if (tbad && !__atomic_load_n(&out->time_unsorted, __ATOMIC_RELAXED)) atomicOr(&out->time_unsorted, 1);
    if (nk && !__atomic_load_n(&out->not_thousands, __ATOMIC_RELAXED)) atomicOr(&out->not_thousands, 1);
    if (nm && !__atomic_load_n(&out->not_millions, __ATOMIC_RELAXED)) atomicOr(&out->not_millions, 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < 256; d += RH_THREADS) {
    uint32_t c = 0;
    for (int w = 0; w < RH_WAVES; ++w) c += hist[w][d];
    counts[(size_t)d * gridDim.x + blockIdx.x] = c;
  }
}

// ------------------------------------------------------------------------------------------
// Pass 1: build the sort word and the identity row ids.
// ------------------------------------------------------------------------------------------
struct MaskPtrs {
  const uint8_t *p[TMP_MAX_MASKS];
};

// Everything needed to build the sort word of input row i.
struct GenArgs {
  const int32_t *key;
  const int64_t *ts;
  const int64_t *seq;
  const uint8_t *seq_valid;
  Layout lay;
  int n_masks;
  MaskPtrs masks;
  int id_bits;  // > 0: the radix passes carry the row id in that many bits below the word
  // bucketed ordering: the top field holds base(key) + floor(nb(key) * rel / (rel_max + 1)) -- the
  // row's bucket -- instead of the key id (rel = the row's time field, inv = ceil(2^64 / (rel_max + 1)));
  // key_pack[key] = {nb, base}, one 8-byte gather per row
  const uint2 *key_pack;
  uint64_t inv;
  uint32_t last_key;  // n_keys - 1: key ids beyond the dictionary are read as the last key
  int vec4;           // the columns are aligned for make_words4 (set where the words are laid out)
};

__device__ __forceinline__ uint64_t make_word(const GenArgs &g, int64_t i) {
  const Layout &lay = g.lay;
  int64_t p = lay.prim_div == 1 ? g.ts[i] : floordiv_i64(g.ts[i], lay.prim_div);
  uint64_t rel = (uint64_t)(p - lay.prim_min);
  if (lay.prim_gran != 1) rel = (uint64_t)floordiv_i64((int64_t)rel, lay.prim_gran);
  uint64_t w = rel << lay.prim_shift();
  if (g.key_pack) {
    uint32_t k = g.key ? (uint32_t)g.key[i] : 0u;
    k = k < g.last_key ? k : g.last_key;
    const uint2 kp = g.key_pack[k];
    const uint32_t nb = kp.x;
    uint64_t q = __umul64hi(rel * nb, g.inv);
    if (q >= nb) q = nb - 1;
    w |= (uint64_t)(kp.y + (uint32_t)q) << lay.key_shift();
  } else if (lay.key_bits) {
    w |= (uint64_t)(uint32_t)g.key[i] << lay.key_shift();
  }
  if (lay.seq_bits && g.seq) {
    bool sv = !g.seq_valid || g.seq_valid[i];
    if (sv) w |= (lay.seq_wide ? 1ull : (uint64_t)(g.seq[i] - lay.seq_min + 1)) << lay.seq_shift();
  }
  for (int m = 0; m < g.n_masks; ++m)
    if (!g.masks.p[m] || g.masks.p[m][i]) w |= 1ull << m;
  return w;
}

// The (time, sequence) part of row i's sort word, as a number: ordering by it is ordering by
// (primary, seq NULLS FIRST).  Not for layouts with seq_wide.
__device__ __forceinline__ uint64_t time_of(const Layout &lay, int64_t ts, bool seq_ok, int64_t seq) {
  int64_t p = lay.prim_div == 1 ? ts : floordiv_i64(ts, lay.prim_div);
  uint64_t rel = (uint64_t)(p - lay.prim_min);
  if (lay.prim_gran != 1) rel = (uint64_t)floordiv_i64((int64_t)rel, lay.prim_gran);
  uint64_t t = rel << lay.seq_bits;
  if (seq_ok) t |= (uint64_t)(seq - lay.seq_min + 1);
  return t;
}
__device__ __forceinline__ uint64_t make_time(const GenArgs &g, int64_t i) {
  const bool has_seq = g.lay.seq_bits && g.seq;
  const bool sv = has_seq && (!g.seq_valid || g.seq_valid[i]);
  return time_of(g.lay, g.ts[i], sv, sv ? g.seq[i] : 0);
}

// Sort words of the four rows i .. i + 3 (i a multiple of 4, all four inside the table), with
// 16-byte loads of the key and time columns and 4-byte loads of the NULL masks; the caller has
// checked that those columns are aligned for it (gen_vector_ok).
__device__ __forceinline__ void make_words4(const GenArgs &g, int64_t i, uint64_t w[4]) {
  const Layout &lay = g.lay;
  const longlong2 ta = *reinterpret_cast<const longlong2 *>(g.ts + i);
  const longlong2 tb = *reinterpret_cast<const longlong2 *>(g.ts + i + 2);
  const int64_t t[4] = {ta.x, ta.y, tb.x, tb.y};
  int4 kv = make_int4(0, 0, 0, 0);
  if (g.key) kv = *reinterpret_cast<const int4 *>(g.key + i);
  const uint32_t kk[4] = {(uint32_t)kv.x, (uint32_t)kv.y, (uint32_t)kv.z, (uint32_t)kv.w};
  uint64_t rel[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t p = lay.prim_div == 1 ? t[u] : floordiv_i64(t[u], lay.prim_div);
    rel[u] = (uint64_t)(p - lay.prim_min);
    if (lay.prim_gran != 1) rel[u] = (uint64_t)floordiv_i64((int64_t)rel[u], lay.prim_gran);
    w[u] = rel[u] << lay.prim_shift();
  }
  if (g.key_pack) {
    uint32_t nb[4], base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t k = kk[u] < g.last_key ? kk[u] : g.last_key;
      const uint2 kp = g.key_pack[k];
      nb[u] = kp.x;
      base[u] = kp.y;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint64_t q = __umul64hi(rel[u] * nb[u], g.inv);
      if (q >= nb[u]) q = nb[u] - 1;
      w[u] |= (uint64_t)(base[u] + (uint32_t)q) << lay.key_shift();
    }
  } else if (lay.key_bits) {
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] |= (uint64_t)kk[u] << lay.key_shift();
  }
  if (lay.seq_bits && g.seq) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool sv = !g.seq_valid || g.seq_valid[i + u];
      if (sv) w[u] |= (lay.seq_wide ? 1ull : (uint64_t)(g.seq[i + u] - lay.seq_min + 1)) << lay.seq_shift();
    }
  }
  for (int m = 0; m < g.n_masks; ++m) {
    uint32_t v = 0x01010101u;
    if (g.masks.p[m]) v = *reinterpret_cast<const uint32_t *>(g.masks.p[m] + i);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((v >> (8 * u)) & 0xffu) w[u] |= 1ull << m;
  }
}

// Whether make_words4 may be used on these columns (host side).
static bool gen_vector_ok(const GenArgs &g) {
  uintptr_t a16 = (uintptr_t)g.ts | (uintptr_t)g.key;
  uintptr_t a4 = 0;
  for (int m = 0; m < g.n_masks; ++m) a4 |= (uintptr_t)g.masks.p[m];
  return (a16 & 15) == 0 && (a4 & 3) == 0;
}

__global__ __launch_bounds__(kBlock) void make_comp_kernel(int64_t n, GenArgs g, uint64_t *comp,
                                                           uint32_t *idx) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    comp[i] = make_word(g, i);
    if (idx) idx[i] = (uint32_t)i;
  }
}

// ------------------------------------------------------------------------------------------
// LSD radix sort of (sort word, row id) pairs, 8-bit digits, three kernels per digit:
//   histogram per block -> exclusive scan of the [digit][block] table -> stable scatter.
// A block owns a contiguous run of tiles; inside a tile the rank of a row among the rows with
// the same digit comes from wave-wide ballots (64 lanes) plus per-wave LDS counters, and the
// tile is first reordered in LDS so that global stores go out in runs of equal digit.
// ------------------------------------------------------------------------------------------
constexpr int RS_MAX_GRID = 1024;
constexpr int RS_MAX_BITS = 9;  // widest digit; passes are 8 bits wide unless 9 saves a pass

struct SortGeom {
  int grid;
  int tiles_per_block;
};
#ifndef TEMPO_FUSED_ITEMS
#define TEMPO_FUSED_ITEMS 8
#endif
constexpr int kFusedItems = TEMPO_FUSED_ITEMS;  // rows per thread of a first pass that also builds the words (their inputs need registers)

static SortGeom sort_geom(int64_t n, int tile) {
  int64_t tiles = (n + tile - 1) / tile;
  if (tiles < 1) tiles = 1;
  SortGeom g;
  g.tiles_per_block = (int)((tiles + RS_MAX_GRID - 1) / RS_MAX_GRID);
  g.grid = (int)((tiles + g.tiles_per_block - 1) / g.tiles_per_block);
  return g;
}

// GEN: the launch is the first pass's histogram and builds the sort words on the way (it writes
// them to `words`; the row ids of a first pass are the positions and are never stored).
// GEN == 2: the words also take the row id into their low gen.id_bits bits.
template <int RS_THREADS, int RS_ITEMS, int RS_BITS, int GEN>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                                int shift, uint32_t mask,
                                                                int tiles_per_block,
                                                                uint32_t *__restrict__ counts, GenArgs gen,
                                                                uint64_t *__restrict__ words,
                                                                const int32_t *run_if) {
  constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
  constexpr int RS_WAVES = RS_THREADS / 64;
  constexpr int RS_RADIX = 1 << RS_BITS;
  if (run_if && !*run_if) return;  // a pass of the stand-by sort: runs only if the bucketed one gave up
  __shared__ uint32_t hist[RS_WAVES][RS_RADIX];
  for (int i = threadIdx.x; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&hist[0][0])[i] = 0;
  __syncthreads();
  int64_t begin = (int64_t)blockIdx.x * tiles_per_block * RS_TILE;
  int64_t end = begin + (int64_t)tiles_per_block * RS_TILE;
  if (end > n) end = n;
  int wave = threadIdx.x >> 6;
  if (GEN == 3) {
    // the digit comes from the key column alone (a pass over key bits of rows whose words the
    // scatter builds itself): 4 bytes per row are read and nothing is written
    const int ks = gen.lay.key_shift();
    for (int64_t i = begin + (int64_t)threadIdx.x * 4; i < end; i += (int64_t)RS_THREADS * 4) {
      if (gen.vec4 && i + 4 <= end) {
        const int4 kv = *reinterpret_cast<const int4 *>(gen.key + i);
        const uint32_t kk[4] = {(uint32_t)kv.x, (uint32_t)kv.y, (uint32_t)kv.z, (uint32_t)kv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) atomicAdd(&hist[wave][(uint32_t)(((uint64_t)kk[u] << ks) >> shift) & mask], 1u);
      } else {
        for (int64_t r = i; r < end && r < i + 4; ++r)
          atomicAdd(&hist[wave][(uint32_t)(((uint64_t)(uint32_t)gen.key[r] << ks) >> shift) & mask], 1u);
      }
    }
  } else if (GEN && gen.vec4) {
    // four consecutive rows per thread, wide loads and stores (begin is a multiple of the tile)
    for (int64_t i = begin + (int64_t)threadIdx.x * 4; i < end; i += (int64_t)RS_THREADS * 4) {
      uint64_t w[4];
      if (i + 4 <= end) {
        make_words4(gen, i, w);
        if (GEN == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = (w[u] << gen.id_bits) | (uint64_t)(i + u);
        }
        *reinterpret_cast<ulonglong2 *>(words + i) = make_ulonglong2(w[0], w[1]);
        *reinterpret_cast<ulonglong2 *>(words + i + 2) = make_ulonglong2(w[2], w[3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) atomicAdd(&hist[wave][(uint32_t)(w[u] >> shift) & mask], 1u);
      } else {
        for (int64_t r = i; r < end; ++r) {
          uint64_t x = make_word(gen, r);
          if (GEN == 2) x = (x << gen.id_bits) | (uint64_t)r;
          words[r] = x;
          atomicAdd(&hist[wave][(uint32_t)(x >> shift) & mask], 1u);
        }
      }
    }
  } else {
  // four rows per thread and round: all their loads are issued before the first store or counter
  // update, so a wave keeps four requests per column in flight instead of one
  constexpr int UN = 4;
  for (int64_t i0 = begin + threadIdx.x; i0 < end; i0 += (int64_t)UN * RS_THREADS) {
    uint64_t w[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + (int64_t)u * RS_THREADS;
      w[u] = 0;
      if (i < end) {
        if (GEN) {
          w[u] = make_word(gen, i);
          if (GEN == 2) w[u] = (w[u] << gen.id_bits) | (uint64_t)i;
        } else {
          w[u] = keys[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + (int64_t)u * RS_THREADS;
      if (i < end) {
        if (GEN) words[i] = w[u];
        atomicAdd(&hist[wave][(uint32_t)(w[u] >> shift) & mask], 1u);
      }
    }
  }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RS_RADIX; d += RS_THREADS) {
    uint32_t c = 0;
    for (int w = 0; w < RS_WAVES; ++w) c += hist[w][d];
    counts[(size_t)d * gridDim.x + blockIdx.x] = c;
  }
}

// Exclusive scan of `total` uint32 counters in place, one block, 16 counters per thread and
// round (four 16-byte loads; `counts` comes from the 256-byte aligned workspace).
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *counts, int total,
                                                          const int32_t *run_if) {
  constexpr int kPer = 16;
  if (run_if && !*run_if) return;
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < total; base += 1024 * kPer) {
    int i = base + tid * kPer;
    uint32_t v[kPer];
    if (i + kPer <= total) {
#pragma unroll
      for (int k = 0; k < kPer; k += 4) {
        uint4 q = *reinterpret_cast<const uint4 *>(counts + i + k);
        v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) v[k] = (i + k < total) ? counts[i + k] : 0u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) s += v[k];
    uint32_t inc = s;
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
    uint32_t run = carry_s + wp + (inc - s);
    if (i + kPer <= total) {
#pragma unroll
      for (int k = 0; k < kPer; k += 4) {
        uint4 q;
        q.x = run; run += v[k];
        q.y = run; run += v[k + 1];
        q.z = run; run += v[k + 2];
        q.w = run; run += v[k + 3];
        *reinterpret_cast<uint4 *>(counts + i + k) = q;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if (i + k < total) counts[i + k] = run;
        run += v[k];
      }
    }
    __syncthreads();
    if (tid == 0) carry_s += tot;
    __syncthreads();
  }
}

int scan_u32_inplace(uint32_t *counts, int total, hipStream_t st) {
  TMP_LAUNCH(radix_scan_kernel, dim3(1), dim3(1024), st, counts, total, (const int32_t *)nullptr);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

// NPAY > 0: the pass also moves up to NPAY 8-byte payload columns, read through the row id
// (src[row id]) and written next to the pair -- used on the last pass so that ordered copies
// of payload columns cost no separate random gather (the scatter writes them in digit runs).
// FIRST: the pass is the first one, its row ids are the positions themselves (not read).
// PACK: the words carry the row id in their low id_bits bits (narrow sort words leave room for
// it), so a pass moves one 8-byte stream instead of a word and an id.  1 = packed in and out
// (every pass but the last), 2 = packed in, and out come the plain word and the row id, both
// written from the one staged copy (the last pass).  0 = separate row ids.
// GENW: the sort words are not read but built here from the table's columns (first pass of rows
// whose digit the histogram took from the key column): no pass writes or re-reads plain words.
// NARROW (with GENW): the words fit 32 bits and go out as uint32, one per row, into kout's memory.
template <int RS_THREADS, int RS_ITEMS, int NPAY, int RS_BITS, bool FIRST, int PACK, bool GENW = false, bool NARROW = false>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, int64_t n,
    int shift, uint32_t mask, int nbits, int tiles_per_block, const uint32_t *offsets,
    PayPtrs pp, int id_bits, const int32_t *run_if, GenArgs gen) {
  static_assert(!GENW || (FIRST && PACK == 0 && NPAY <= 0), "words are built in the plain first pass only");
  static_assert(!NARROW || GENW, "32-bit words leave the pass that builds them, and no other pass reads them");
  // NPAY < 0: -NPAY columns that are already in the order of the rows being moved (read by
  // position, no dependent load: the larger plain tile serves, nothing is requested ahead)
  constexpr bool PAY = NPAY != 0;
  constexpr bool POS = NPAY < 0;
  constexpr int NP = NPAY < 0 ? -NPAY : NPAY;
  if (run_if && !*run_if) return;
  const uint64_t id_mask = PACK ? ((1ull << id_bits) - 1) : 0ull;
  constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
  constexpr int RS_WAVES = RS_THREADS / 64;
  constexpr int RS_RADIX = 1 << RS_BITS;
  static_assert(RS_RADIX <= RS_THREADS, "one thread per digit");
  // One staging buffer, used in turn for the sort words, the row ids and each payload column:
  // a tile is as large as the words alone allow, which makes the runs of equal digit -- the
  // unit the scattered stores go out in -- that much longer.
  __shared__ uint64_t sbuf[RS_TILE];
  uint32_t *sbuf32 = reinterpret_cast<uint32_t *>(sbuf);
  // A wave's lanes hand counts to each other through whist from one item to the next: those
  // accesses are relaxed wavefront-scope atomics (plain ds_read / ds_write, kept in program
  // order).  Not volatile: volatile LDS accesses are emitted as flat operations, which also
  // wait on the global loads in flight.
  __shared__ uint32_t whist[RS_WAVES][RS_RADIX];
  __shared__ uint32_t dstart[RS_RADIX];
  __shared__ uint32_t goff[RS_RADIX];
  __shared__ uint32_t wtot[RS_RADIX / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  if (tid < RS_RADIX) goff[tid] = offsets[(size_t)tid * gridDim.x + blockIdx.x];
  // Payload passes read their payload through the row ids, a dependent chain of loads: there
  // the pairs of the next tile are requested one tile ahead.  Plain passes use the registers
  // for a larger tile instead.
  constexpr bool PREFETCH = PAY && !POS;
  uint64_t k[RS_ITEMS], kn[PREFETCH ? RS_ITEMS : 1];
  uint32_t v[RS_ITEMS], vn[PREFETCH ? RS_ITEMS : 1];
  if (PREFETCH) {
    int64_t base0 = (int64_t)blockIdx.x * tiles_per_block * RS_TILE;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      int64_t g = base0 + wave * (64 * RS_ITEMS) + it * 64 + lane;
      bool valid = g < n;
      kn[PREFETCH ? it : 0] = valid ? kin[g] : 0ull;
      vn[PREFETCH ? it : 0] = (valid && !PACK) ? (FIRST ? (uint32_t)g : vin[g]) : 0u;
    }
  }
  for (int t = 0; t < tiles_per_block; ++t) {
    int64_t tile_base = ((int64_t)blockIdx.x * tiles_per_block + t) * RS_TILE;
    if (tile_base >= n) break;
    int tile_n = (int)((n - tile_base) < RS_TILE ? (n - tile_base) : RS_TILE);
    for (int i = tid; i < RS_WAVES * RS_RADIX; i += RS_THREADS) whist[i / RS_RADIX][i % RS_RADIX] = 0;
    __syncthreads();
    uint64_t pay[(PAY && !POS) ? NP : 1][(PAY && !POS) ? RS_ITEMS : 1];
    uint32_t rank[RS_ITEMS];
    uint32_t cnt_inc = 0;
#ifdef RS_TIMING
    long long tk[12]; int ntk = 0;
#define STICK() do { __syncthreads(); tk[ntk++] = clock64(); } while (0)
#else
#define STICK() do {} while (0)
#endif
    STICK();
    if (PREFETCH) {
#pragma unroll
      for (int it = 0; it < RS_ITEMS; ++it) {
        k[it] = kn[PREFETCH ? it : 0];
        v[it] = PACK ? (uint32_t)(k[it] & id_mask) : vn[PREFETCH ? it : 0];
      }
      if (t + 1 < tiles_per_block) {
        int64_t next_base = tile_base + RS_TILE;
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
          int64_t g = next_base + wave * (64 * RS_ITEMS) + it * 64 + lane;
          bool valid = g < n;
          kn[PREFETCH ? it : 0] = valid ? kin[g] : 0ull;
          vn[PREFETCH ? it : 0] = (valid && !PACK) ? (FIRST ? (uint32_t)g : vin[g]) : 0u;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < RS_ITEMS; ++it) {
        int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
        if (!GENW) k[it] = pos < tile_n ? kin[tile_base + pos] : 0ull;
      }
      if (GENW) {
        // every load first and none inside a per-row branch (a slot past the tile reads the tile's
        // last row): a wave then has all its requests in flight at once
        // (uniform base pointers and 32-bit offsets: one address register per load)
        const int64_t *ts_t = gen.ts + tile_base;
        const int32_t *key_t = gen.key + tile_base;
        int64_t tsv[RS_ITEMS];
        uint32_t kv[RS_ITEMS], mb[RS_ITEMS];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
          const uint32_t pos = (uint32_t)(wave * (64 * RS_ITEMS) + it * 64 + lane);
          const uint32_t at = pos < (uint32_t)tile_n ? pos : (uint32_t)tile_n - 1u;
          tsv[it] = ts_t[at];
          kv[it] = (uint32_t)key_t[at];
          mb[it] = 0;
        }
        for (int m = 0; m < gen.n_masks; ++m) {
          const uint8_t *mp = gen.masks.p[m];
          if (mp) {
            mp += tile_base;
#pragma unroll
            for (int it = 0; it < RS_ITEMS; ++it) {
              const uint32_t pos = (uint32_t)(wave * (64 * RS_ITEMS) + it * 64 + lane);
              const uint32_t at = pos < (uint32_t)tile_n ? pos : (uint32_t)tile_n - 1u;
              mb[it] |= (mp[at] ? 1u : 0u) << m;
            }
          } else {
#pragma unroll
            for (int it = 0; it < RS_ITEMS; ++it) mb[it] |= 1u << m;
          }
        }
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
          const int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
          k[it] = pos < tile_n ? plain_word(gen.lay, tsv[it], kv[it], mb[it]) : 0ull;
        }
      }
    }
    STICK();
    if (PAY && !POS) {
#pragma unroll
      for (int it = 0; it < RS_ITEMS; ++it) {
        int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
#pragma unroll
        for (int c = 0; c < NP; ++c)
          pay[c][(PAY && !POS) ? it : 0] = (pos < tile_n && c < pp.n) ? pp.src[c][v[it]] : 0ull;
      }
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
      bool valid = pos < tile_n;
      uint32_t d = (uint32_t)(k[it] >> shift) & mask;
      // peers = lanes holding the same digit: per bit one ballot, then "keep the lanes that
      // agree with me" as xnor with the bit spread over a word (bit set: keep the ballot,
      // clear: keep its complement).  All RS_BITS bits, unrolled: bits above the pass's width
      // are zero in every lane and keep everything.
      const uint64_t vb = __ballot(valid);
      uint32_t plo = (uint32_t)vb, phi = (uint32_t)(vb >> 32);
#pragma unroll
      for (int b = 0; b < RS_BITS; ++b) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // 0 or ~0
        const uint64_t bal = __ballot(m != 0u);
        plo &= ~((uint32_t)bal ^ m);
        phi &= ~((uint32_t)(bal >> 32) ^ m);
      }
      const uint64_t peers = ((uint64_t)phi << 32) | plo;
      uint32_t prior = __hip_atomic_load(&whist[wave][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      uint32_t cnt = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
      bool leader = valid && (lane == (__ffsll((long long)peers) - 1));
      if (leader)
        __hip_atomic_store(&whist[wave][d], prior + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      rank[it] = prior + (uint32_t)__popcll(peers & lt_mask);
    }
    __syncthreads();
    STICK();
    // digit `tid` (first 256 threads): exclusive prefix of its count over the waves, then over
    // the digits
    uint32_t cnt_d = 0;
    if (tid < RS_RADIX) {
      for (int w = 0; w < RS_WAVES; ++w) {
        uint32_t c = whist[w][tid];
        whist[w][tid] = cnt_d;
        cnt_d += c;
      }
      uint32_t inc = cnt_d;
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t tt = __shfl_up(inc, o, 64);
        if (lane >= o) inc += tt;
      }
      if (lane == 63) wtot[wave] = inc;
      cnt_inc = inc;
    }
    __syncthreads();
    if (tid < RS_RADIX) {
      uint32_t wp = 0;
      for (int w = 0; w < wave; ++w) wp += wtot[w];
      dstart[tid] = wp + cnt_inc - cnt_d;
    }
    __syncthreads();
    STICK();
    // words: to their place in the tile's digit order, then out in runs
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
      if (pos < tile_n) {
        uint32_t d = (uint32_t)(k[it] >> shift) & mask;
        rank[it] += dstart[d] + whist[wave][d];  // from here on: the item's slot in the tile
        sbuf[rank[it]] = k[it];
      }
    }
    __syncthreads();
    STICK();
    uint32_t gpos[RS_ITEMS];  // global position of the tile slot tid + j * RS_THREADS
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
      int p = tid + j * RS_THREADS;
      if (p < tile_n) {
        uint64_t kk = sbuf[p];
        uint32_t d = (uint32_t)(kk >> shift) & mask;
        gpos[j] = goff[d] + ((uint32_t)p - dstart[d]);
        if (NARROW) reinterpret_cast<uint32_t *>(kout)[gpos[j]] = (uint32_t)kk;
        else kout[gpos[j]] = PACK == 2 ? kk >> id_bits : kk;
      }
    }
    if (PACK == 2) {  // the row ids, from the same staged copy, as a stream of their own
#pragma unroll
      for (int j = 0; j < RS_ITEMS; ++j) {
        int p = tid + j * RS_THREADS;
        if (p < tile_n) vout[gpos[j]] = (uint32_t)(sbuf[p] & id_mask);
      }
    }
    STICK();
    if (PACK == 0) {
      __syncthreads();
      // row ids through the same buffer (read only now unless the payload needed them: fewer
      // registers live while the tile is ranked)
#pragma unroll
      for (int it = 0; it < RS_ITEMS; ++it) {
        int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
        if (!PREFETCH) v[it] = pos < tile_n ? (FIRST ? (uint32_t)(tile_base + pos) : vin[tile_base + pos]) : 0u;
      }
#pragma unroll
      for (int it = 0; it < RS_ITEMS; ++it) {
        int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
        if (pos < tile_n) sbuf32[rank[it]] = v[it];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RS_ITEMS; ++j) {
        int p = tid + j * RS_THREADS;
        if (p < tile_n) vout[gpos[j]] = sbuf32[p];
      }
    }
    STICK();
    if (PAY && !POS) {
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        if (c < pp.n) {  // uniform
          __syncthreads();
#pragma unroll
          for (int it = 0; it < RS_ITEMS; ++it) {
            int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
            if (pos < tile_n) sbuf[rank[it]] = pay[c][(PAY && !POS) ? it : 0];
          }
          __syncthreads();
#pragma unroll
          for (int j = 0; j < RS_ITEMS; ++j) {
            int p = tid + j * RS_THREADS;
            if (p < tile_n) pp.out[c][gpos[j]] = sbuf[p];
          }
        }
      }
    }
    if (POS) {
      // columns that travel with the rows: each is read by position only now (nothing of it was
      // live while the tile was ranked) and goes through the staging buffer like the words did
#pragma unroll 1
      for (int c = 0; c < pp.n; ++c) {
        const uint64_t *src = pp.src[c];
        uint64_t *out = pp.out[c];
        uint64_t pv[RS_ITEMS];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
          int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
          pv[it] = pos < tile_n ? src[tile_base + pos] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
          int pos = wave * (64 * RS_ITEMS) + it * 64 + lane;
          if (pos < tile_n) sbuf[rank[it]] = pv[it];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
          int p = tid + j * RS_THREADS;
          if (p < tile_n) out[gpos[j]] = sbuf[p];
        }
      }
    }
    __syncthreads();
    if (tid < RS_RADIX) goff[tid] += cnt_d;
    __syncthreads();
#ifdef RS_TIMING
    STICK();
    if (tid == 0 && t == 1 && (blockIdx.x % 300) == 7)
      printf("X npay %d bits %d first %d pack %d: load %lld rank %lld scan %lld lds %lld outw %lld ids %lld pay %lld\n", NPAY, RS_BITS,
             (int)FIRST, PACK, tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6]);
#endif
  }
}


// ------------------------------------------------------------------------------------------
// Bucketed ordering of large tables in random order: two or three partition passes instead of
// one pass per 8 bits of the sort word.
//
// Every row gets a bucket number that is monotone in (key, time): a key owns a run of
// nb(key) ~ rows(key) / A buckets and a row falls into the one its time selects
// (key_base + floor(nb * (t - t_min) / (t_max - t_min + 1))).  Where the rows of a key are spread
// evenly over time a bucket holds about A rows.  The sort word carries the bucket number in its
// top field in place of the key id; stable LSD passes over that field alone (bucket_bits, 17-18
// bits for 5x10^8 rows) group the rows by bucket, and a tile kernel then finishes the order
// inside LDS: a block takes the buckets that start inside its 8192-row stretch (at most 16384
// rows), restores the key id in the top field (bucket_key), sorts by the bits in which the
// tile's words actually differ and writes the words and row ids out in one contiguous run.
//
// A bucket that turns out longer than a tile can absorb (rows bunched in time, massive
// duplicates) raises a device flag; the plain LSD passes are always enqueued behind the tile
// kernel, every one of their kernels returning at once unless that flag is set, so the result
// is the same either way and the host never waits.
// ------------------------------------------------------------------------------------------
constexpr int TS_THREADS = 1024;
constexpr int TS_ITEMS = 16;
constexpr int TS_CAP = TS_THREADS * TS_ITEMS;  // most rows one tile can hold
constexpr int TS_TILE = TS_CAP / 2;            // stretch of rows whose bucket starts a block owns
constexpr int TS_IDX_BITS = 14;                // bits_for(TS_CAP - 1)
// Tables below this many rows take the plain passes (TEMPO_BUCKET_MIN_ROWS overrides: tests).
static int64_t bucket_min_rows() {
  const char *e = getenv("TEMPO_BUCKET_MIN_ROWS");
  return e ? (int64_t)atoll(e) : (int64_t)(1 << 22);
}
constexpr int kCountStride = 16;               // every 16th row is counted per key
constexpr int kCountLdsKeys = 12288;

struct BucketGeom {
  bool ok;
  int bits;         // width of the bucket field
  int64_t avg;      // rows per bucket the map aims at
  int64_t max_buckets;
};

static BucketGeom bucket_geometry(int64_t n, int64_t n_keys) {
  BucketGeom g;
  g.ok = false; g.bits = 0; g.avg = 0; g.max_buckets = 0;
  if (n < bucket_min_rows() || n_keys < 1) return g;
  for (int b = 10; b <= 26; ++b) {
    // every key owns at least one bucket; the counts are estimates from every 16th row
    const int64_t room = (1ll << b) - n_keys - 2;
    if (room <= 0) continue;
    const int64_t a = (n + kCountStride + room - 1) / room;
    if (a <= TS_TILE * 55 / 100) {
      g.ok = a >= 64;  // more keys than buckets worth having: the plain passes do as well
      g.bits = b;
      g.avg = a;
      g.max_buckets = 1ll << b;
      return g;
    }
  }
  return g;
}

__global__ __launch_bounds__(kBlock) void key_sample_kernel(const int32_t *__restrict__ key, int64_t n_samples,
                                                              uint32_t n_keys, unsigned long long *rows) {
  __shared__ unsigned int hist[kCountLdsKeys];
  const bool local = n_keys <= (uint32_t)kCountLdsKeys;
  if (local) {
    for (uint32_t i = threadIdx.x; i < n_keys; i += kBlock) hist[i] = 0;
    __syncthreads();
  }
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_samples; s += step) {
    const uint32_t k = key ? (uint32_t)key[s * kCountStride] : 0u;
    if (k >= n_keys) continue;
    if (local) atomicAdd(&hist[k], 1u);
    else atomicAdd(&rows[k], 1ull);
  }
  if (local) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_keys; i += kBlock)
      if (hist[i]) atomicAdd(&rows[i], (unsigned long long)hist[i]);
  }
}

// nb(key) = max(1, ceil(estimated rows / avg)), into both arrays (key_base is scanned afterwards)
__global__ void bucket_nb_kernel(const unsigned long long *rows, uint32_t n_keys, uint64_t avg, uint32_t *key_nb,
                                 uint32_t *key_base) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_keys) return;
  uint64_t est = rows[k] * (uint64_t)kCountStride;
  uint64_t nb = (est + avg - 1) / avg;
  if (nb < 1) nb = 1;
  key_nb[k] = (uint32_t)nb;
  key_base[k] = (uint32_t)nb;
}

// bucket_key[b] = the key whose run of buckets holds b (largest k with key_base[k] <= b); and
// key_pack[k] = {key_nb[k], key_base[k]}, the form the sort words are built from
__global__ void bucket_key_kernel(const uint32_t *key_base, const uint32_t *key_nb, uint32_t n_keys,
                                  uint32_t max_buckets, uint32_t *bucket_key, uint2 *key_pack) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t k = b; k < n_keys; k += gridDim.x * blockDim.x) key_pack[k] = make_uint2(key_nb[k], key_base[k]);
  if (b >= max_buckets) return;
  const uint32_t total = key_base[n_keys - 1] + key_nb[n_keys - 1];
  if (b >= total) return;
  uint32_t lo = 0, hi = n_keys;  // first k with key_base[k] > b
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (key_base[mid] <= b) lo = mid + 1;
    else hi = mid;
  }
  bucket_key[b] = lo - 1;
}

// bounds[g] = first position p >= g * tile at which a bucket starts (p == 0, or the top field
// differs from the row before), looked for up to g * tile + cap; n when the table ends first,
// INT64_MAX when no bucket starts within reach.  One wave per entry, g in [0, n_tiles].
__global__ __launch_bounds__(kBlock) void tile_bounds_kernel(const uint64_t *w, int64_t n, int top_shift, int tile,
                                                             int cap, int64_t n_tiles, int64_t *bounds) {
  const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (g > n_tiles) return;
  const int64_t from = g * tile;
  int64_t found = INT64_MAX;
  if (from >= n) {
    found = n;
  } else {
    const int64_t lim = from + cap + 1 < n ? from + cap + 1 : n;
#pragma unroll 1
    for (int64_t base = from; base < lim; base += 256) {
      const int64_t p0 = base + lane * 4;
      int hit = 4;  // first of this lane's four positions that starts a bucket
      if (p0 < lim) {
        uint64_t prev = p0 > 0 ? w[p0 - 1] >> top_shift : ~0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (p0 + r < lim) {
            const uint64_t cur = w[p0 + r] >> top_shift;
            if (hit == 4 && (p0 + r == 0 || cur != prev)) hit = r;
            prev = cur;
          }
        }
      }
      const uint64_t bal = __ballot(hit < 4);
      if (bal) {
        const int first = __ffsll((long long)bal) - 1;
        found = base + first * 4 + __shfl(hit, first, 64);
        break;
      }
    }
    if (found == INT64_MAX && lim == n) found = n;
  }
  if (lane == 0) bounds[g] = found;
}

// ---- a few time-ordered runs laid end to end (sorted mode 3) ---------------------------------
// The runs are merged by (time, sequence), ties in input order, TS_CAP output rows per tile: for
// every tile boundary d = t * TS_CAP the cut (c_0 .. c_R-1), sum = d, says how many rows of every
// run come before it.  It is found by bisecting on the time value v of the d-th merged row --
// "how many rows are <= v" is one binary search per run, run r in lane r of the wave that owns the
// boundary -- and handing the rows equal to v out in run order.  Every 64th boundary is searched
// over whole runs; the ones in between only between their two neighbours' cuts.
struct RunArgs {
  int n_runs;
  int64_t start[kMaxRuns + 1];
};
constexpr int kCutStride = 64;

// rows of [base + lo, base + hi) -- time-ordered -- whose time is < v (le false) or <= v (le true), plus lo
__device__ __forceinline__ int64_t run_rank(const GenArgs &g, int64_t base, int64_t lo, int64_t hi, uint64_t v, bool le) {
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    const uint64_t t = make_time(g, base + mid);
    if (le ? t <= v : t < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(kBlock) void run_cuts_kernel(GenArgs g, RunArgs ra, int64_t n, int tile, int64_t n_tiles,
                                                          int fine, int64_t *cuts) {
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  // coarse launch: boundaries 0, 64, 128, ... and the last one; fine launch: all the others
  int64_t t;
  if (!fine) {
    const int64_t n_coarse = (n_tiles + kCutStride - 1) / kCutStride + 1;
    if (wv >= n_coarse) return;
    t = wv * kCutStride < n_tiles ? wv * kCutStride : n_tiles;
  } else {
    if (wv > n_tiles) return;
    t = wv;
    if (t % kCutStride == 0 || t == n_tiles) return;
  }
  const bool act = lane < ra.n_runs;
  const int64_t base = act ? ra.start[lane] : 0;
  const int64_t len = act ? ra.start[lane + 1] - base : 0;
  const int64_t d = t * tile < n ? t * tile : n;
  int64_t lo = 0, hi = len;
  if (fine) {
    const int64_t t0 = t / kCutStride * kCutStride;
    const int64_t t1 = t0 + kCutStride < n_tiles ? t0 + kCutStride : n_tiles;
    if (act) { lo = cuts[t0 * kMaxRuns + lane]; hi = cuts[t1 * kMaxRuns + lane]; }
  }
  int64_t cut = lo;
  if (d >= n) {
    cut = len;
  } else if (d > 0) {
    // the d-th merged row's time lies between the smallest time at the lower cuts and the largest
    // before the upper cuts
    uint64_t vlo = ~0ull, vhi = 0ull;
    if (act && lo < len) vlo = make_time(g, base + lo);
    if (act && hi > 0) vhi = make_time(g, base + hi - 1);
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t a = __shfl_xor(vlo, o, 64), b = __shfl_xor(vhi, o, 64);
      vlo = a < vlo ? a : vlo;
      vhi = b > vhi ? b : vhi;
    }
    while (vlo < vhi) {  // smallest v with (rows <= v) > d
      const uint64_t mid = vlo + ((vhi - vlo) >> 1);
      const int64_t mine = act ? run_rank(g, base, lo, hi, mid, true) : 0;
      // the counts are monotone in v: each answer also narrows where the next search has to look
      if (wave_sum_i64(mine) > d) { vhi = mid; hi = mine; }
      else { vlo = mid + 1; lo = mine; }
    }
    const int64_t lt = act ? run_rank(g, base, lo, hi, vlo, false) : 0;
    const int64_t le = act ? run_rank(g, base, lo, hi, vlo, true) : 0;
    const int64_t rem = d - wave_sum_i64(lt);  // rows equal to v that still come before the boundary
    int64_t cap = le - lt, inc = cap;
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    const int64_t before = inc - cap;  // equal rows of earlier runs
    int64_t take = rem - before;
    take = take < 0 ? 0 : (take > cap ? cap : take);
    cut = lt + take;
  }
  if (lane < kMaxRuns) cuts[t * kMaxRuns + lane] = act ? cut : 0;
}

// TS_BINS bins for the interpolation placement of a tile, two 16-bit counters per LDS word.
constexpr int TS_BINS = TS_CAP / 2;
constexpr int TS_DENSE = 64;  // a bin with more slots than this sends its tile to the radix passes

// MERGE: the tile's rows are not one stretch of bucketed pairs but the slices of a few time-ordered
// runs between two cuts (run_cuts_kernel); they are read from the table's own columns, sorted by
// (time, sequence) -- ties in input order, as the slots are filled run by run -- and written out as
// sort words (key included) and row ids.
template <int NT, int NI, bool MERGE>
__global__ __launch_bounds__(NT) void tile_sort_kernel(const uint64_t *win, const uint32_t *idin, uint64_t *wout,
                                                       uint32_t *idout, int64_t n, int top_shift, int low_bits,
                                                       const uint32_t *bucket_key, const int64_t *bounds,
                                                       int32_t *fallback, PayPtrs pp, int force_radix, GenArgs gen,
                                                       RunArgs ra) {
  constexpr int CAP = NT * NI, TILE = CAP / 2, NW = NT / 64;
  static_assert(NW * 256 * 4 == TS_BINS * 2 && CAP == TS_CAP, "aux serves both the radix counters and the bins");
  __shared__ uint64_t sbuf[CAP];
  __shared__ uint32_t aux[NW * 256];  // radix passes: [wave][digit] counters; placement: packed bins
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t wtot[NW];
  __shared__ uint64_t red_min[1], red_max[1];
  __shared__ int dense_s;
  uint32_t(*whist)[256] = reinterpret_cast<uint32_t(*)[256]>(aux);
  uint32_t *sbuf32 = reinterpret_cast<uint32_t *>(sbuf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  __shared__ int64_t run_src[kMaxRuns];  // MERGE: first row of every run's slice ...
  __shared__ int run_off[kMaxRuns + 1];  // ... and the slot its slice starts at
  int64_t s, e;
#ifdef TS_TIMING
  long long tk[16]; int ntk = 0;
#define TICK() do { __syncthreads(); tk[ntk++] = clock64(); } while (0)
#else
#define TICK() do {} while (0)
#endif
  TICK();
  if (tid == 0) { red_min[0] = ~0ull; red_max[0] = 0ull; }
  if (MERGE) {
    // bounds = the cuts: tile t holds, of run r, the rows cuts[t][r] .. cuts[t + 1][r]
    s = (int64_t)blockIdx.x * CAP;
    e = s + CAP < n ? s + CAP : n;
    if (tid < 64) {
      const bool act = tid < ra.n_runs;
      const int64_t c0 = act ? bounds[(int64_t)blockIdx.x * kMaxRuns + tid] : 0;
      const int64_t c1 = act ? bounds[((int64_t)blockIdx.x + 1) * kMaxRuns + tid] : 0;
      int len = (int)(c1 - c0), inc = len;
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (tid >= o) inc += up;
      }
      if (tid < kMaxRuns) {
        run_src[tid] = act ? ra.start[tid] + c0 : 0;
        run_off[tid] = inc - len;
      }
      if (tid == kMaxRuns - 1) run_off[kMaxRuns] = inc;
    }
    __syncthreads();
  } else {
    const int64_t t0 = (int64_t)blockIdx.x * TILE;
    const int64_t t1 = t0 + TILE < n ? t0 + TILE : n;
    // the block owns the buckets that start in [t0, t1): rows [s, e)
    s = bounds[blockIdx.x];
    if (s >= t1) return;  // a bucket that began earlier covers the whole stretch
    e = bounds[blockIdx.x + 1];
    if (e == INT64_MAX || e - s > CAP) {
      if (tid == 0) atomicOr(fallback, 1);
      return;
    }
    __syncthreads();
  }
  const int m = (int)(e - s);
  // wave w works on slots [w * chunk, (w + 1) * chunk), nit items per lane
  const int nit = (m + NT - 1) / NT;
  const int chunk = nit * 64;
  const uint64_t field_mask = (top_shift >= 64) ? ~0ull : ((1ull << top_shift) - 1);
  uint64_t wv[NI];
  uint32_t idv[NI];
  uint32_t kv[MERGE ? NI : 1];  // MERGE: key id and NULL flags of the row, as they sit below the time
  const int kv_bits = MERGE ? gen.lay.key_bits + gen.n_masks : 0;
  uint64_t mn = ~0ull, mx = 0ull;
  if (tid == 0) dense_s = force_radix;
  // the loads of all the thread's slots are issued before anything is computed from them
  if (MERGE) {
    const bool has_seq = gen.lay.seq_bits && gen.seq;
    const bool has_kv = kv_bits <= 31;  // bit 31 of kv: the row's sequence value is not NULL
    const bool has_key = gen.lay.key_bits && gen.key && has_kv;
    TICK();
    // slot -> row: a lane's slots are 64 apart and mostly inside one run's slice, where the rows
    // are 64 apart too; the run is looked up again only when some lane of the wave leaves its slice
    {
      int next_off = 0;       // slot at which the lane's current slice ends
      int64_t row_minus = 0;  // row of slot j = row_minus + j inside the current slice
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = wave * chunk + it * 64 + lane;
        const bool live = it < nit && j < m;
        kv[it] = 0;
        if (__any(live && j >= next_off)) {
          // the last run whose slice starts at or before slot j (slices of unused runs start at the end)
          static_assert(kMaxRuns == 16, "four halving steps");
          const int jj = live ? j : 0;
          int r = jj >= run_off[8] ? 8 : 0;
          r += jj >= run_off[r + 4] ? 4 : 0;
          r += jj >= run_off[r + 2] ? 2 : 0;
          r += jj >= run_off[r + 1] ? 1 : 0;
          row_minus = run_src[r] - run_off[r];
          next_off = run_off[r + 1];
        }
        idv[it] = live ? (uint32_t)(row_minus + j) : 0u;
      }
    }
    TICK();
    // no branch around a load (an unused slot reads row 0): loads inside per-slot branches would
    // each be waited for before the next is issued
#pragma unroll
    for (int it = 0; it < NI; ++it) wv[it] = (uint64_t)gen.ts[idv[it]];
    if (has_key) {
#pragma unroll
      for (int it = 0; it < NI; ++it) kv[it] = (uint32_t)gen.key[idv[it]] << gen.n_masks;
    }
    if (has_kv) {
      for (int mk = 0; mk < gen.n_masks; ++mk) {
        const uint8_t *mp = gen.masks.p[mk];
        if (mp) {
#pragma unroll
          for (int h = 0; h < NI; h += NI / 2) {  // two batches, to bound the registers held at once
            uint8_t f[NI / 2];
#pragma unroll
            for (int it = 0; it < NI / 2; ++it) f[it] = mp[idv[h + it]];
#pragma unroll
            for (int it = 0; it < NI / 2; ++it) kv[h + it] |= (f[it] ? 1u : 0u) << mk;
          }
        } else {
#pragma unroll
          for (int it = 0; it < NI; ++it) kv[it] |= 1u << mk;
        }
      }
    }
    if (has_seq) {
      if (gen.seq_valid) {
        uint8_t f[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) f[it] = gen.seq_valid[idv[it]];
#pragma unroll
        for (int it = 0; it < NI; ++it) kv[it] |= (f[it] ? 1u : 0u) << 31;
      } else {
#pragma unroll
        for (int it = 0; it < NI; ++it) kv[it] |= 1u << 31;
      }
    }
    TICK();
    {
      // floor(ts / prim_div) - prim_min in double arithmetic where that is exact (times below 2^52,
      // a result below 2^31): product, truncation, and one fused remainder that says whether the
      // truncated quotient is one off -- a fraction of the instructions of the integer route
      const Layout &lay = gen.lay;
      const bool quick = lay.prim_div >= 2 && lay.prim_div <= (1ll << 40) && lay.prim_gran == 1 && lay.prim_bits <= 31;
      const double bd = (double)lay.prim_div, inv = 1.0 / bd, pmin = (double)lay.prim_min;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int64_t ts = (int64_t)wv[it];
        if (quick && (uint64_t)ts < (1ull << 52)) {
          const double a = (double)ts;
          double q = trunc(a * inv);
          const double r = __builtin_fma(-q, bd, a);
          q += r < 0.0 ? -1.0 : (r >= bd ? 1.0 : 0.0);
          wv[it] = (uint64_t)(uint32_t)(q - pmin) << lay.seq_bits;
        } else {
          wv[it] = time_of(lay, ts, false, 0);
        }
      }
    }
    TICK();
    if (has_seq) {
      // the sequence values join in two batches, to bound the registers held at once
#pragma unroll
      for (int h = 0; h < NI; h += NI / 2) {
        int64_t sq[NI / 2];
#pragma unroll
        for (int it = 0; it < NI / 2; ++it) sq[it] = gen.seq[idv[h + it]];
#pragma unroll
        for (int it = 0; it < NI / 2; ++it)
          if (kv[h + it] >> 31) wv[h + it] |= (uint64_t)(sq[it] - gen.lay.seq_min + 1);
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      kv[it] &= 0x7fffffffu;
      if (it < nit && j < m) {
        mn = wv[it] < mn ? wv[it] : mn;
        mx = wv[it] > mx ? wv[it] : mx;
      }
    }
  } else {
    uint32_t top[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      const bool live = it < nit && j < m;
      wv[it] = live ? win[s + j] : 0ull;
      idv[it] = live ? idin[s + j] : 0u;
    }
    if (bucket_key && top_shift < 64) {
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = wave * chunk + it * 64 + lane;
        top[it] = (it < nit && j < m) ? bucket_key[wv[it] >> top_shift] : 0u;
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      if (it < nit && j < m) {
        uint64_t v = wv[it];
        if (bucket_key && top_shift < 64) v = ((uint64_t)top[it] << top_shift) | (v & field_mask);
        wv[it] = v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
      }
    }
  }
  TICK();
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t a = __shfl_down(mn, o, 64), b = __shfl_down(mx, o, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (lane == 0) {
    atomicMin((unsigned long long *)&red_min[0], (unsigned long long)mn);
    atomicMax((unsigned long long *)&red_max[0], (unsigned long long)mx);
  }
  for (int i = tid; i < NW * 256; i += NT) aux[i] = 0;
  __syncthreads();
  mn = red_min[0]; mx = red_max[0];
  // MERGE: key and flags ride below the time value where the tile's time span leaves room for them
  // (they are then read with the time, in row order); otherwise they are fetched by row id at the end
  bool ride = false;
  const uint64_t tmin = MERGE ? mn : 0ull;  // MERGE: times count from the tile's first
  if (MERGE) {
    const int tspan = mx == mn ? 0 : 64 - __clzll((long long)(mx - mn));
    ride = kv_bits <= 31 && tspan + kv_bits + TS_IDX_BITS <= 64;
    if (ride) low_bits = kv_bits;
    mx = (mx - mn) << low_bits;
    mn = 0;
  }
  const uint64_t low_mask = (1ull << low_bits) - 1;
  const uint64_t wmin = mn & ~low_mask;
  // bits in which the tile's words differ, above the low bits that are carried but not compared
  const uint64_t range = (mx - wmin) >> low_bits;
  const int span = range ? 64 - __clzll((long long)range) : 0;
  if (span + low_bits + TS_IDX_BITS > 64) {  // no room for the slot number next to the word
    if (tid == 0) atomicOr(fallback, 1);
    return;
  }
  // from here on a slot is (word - wmin) << TS_IDX_BITS | slot number: all distinct, and their
  // order is the stable order of the words
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int j = wave * chunk + it * 64 + lane;
    if (MERGE) wv[it] = ((wv[it] - tmin) << low_bits) | (ride ? (uint64_t)kv[it] : 0ull);
    wv[it] = ((wv[it] - wmin) << TS_IDX_BITS) | (uint64_t)j;
  }
  TICK();
  // fewer distinct values than a bin may hold per slot: some bin is crowded for certain
  bool dense = force_radix != 0 || (range + 1) * (uint64_t)TS_DENSE < (uint64_t)m;
  if (!dense && span > 0) {
    // ---- interpolation placement: bin = floor(TS_BINS * value / (range + 1)) is monotone in the
    // value; where the values are spread evenly a bin holds a slot or two.  Count per bin, scan,
    // drop every slot into its bin (any order), then rank it among the slots of its bin.
    const double scale = (double)TS_BINS / ((double)range + 1.0);
    const int val_shift = TS_IDX_BITS + low_bits;
    const uint64_t cmp_mask = ~(low_mask << TS_IDX_BITS);
    uint32_t q[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      q[it] = 0;
      if (it < nit && j < m) {
        uint32_t b = (uint32_t)((double)(wv[it] >> val_shift) * scale);
        b = b < (uint32_t)(TS_BINS - 1) ? b : (uint32_t)(TS_BINS - 1);
        q[it] = b;
        atomicAdd(&aux[b >> 1], 1u << ((b & 1u) * 16));
      }
    }
    __syncthreads();
    {
      // exclusive scan over the bins: thread t owns the TS_BINS / NT bins of aux[t * W .. t * W + W)
      constexpr int W = TS_BINS / 2 / NT;  // words per thread
      uint32_t c[2 * W];
      uint32_t tot = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const uint32_t pkd = aux[tid * W + k];
        c[2 * k] = pkd & 0xffffu;
        c[2 * k + 1] = pkd >> 16;
        tot += c[2 * k] + c[2 * k + 1];
      }
      uint32_t inc = tot;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t tt = __shfl_up(inc, o, 64);
        if (lane >= o) inc += tt;
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      uint32_t run = inc - tot;
      for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const uint32_t a0 = run;
        run += c[2 * k];
        const uint32_t a1 = run;
        run += c[2 * k + 1];
        aux[tid * W + k] = a0 | (a1 << 16);
      }
    }
    __syncthreads();
    // drop: the bin's counter hands out its slots and ends up at the bin's end (= the next bin's start)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      if (it < nit && j < m) {
        const uint32_t sh = (q[it] & 1u) * 16;
        const uint32_t old = atomicAdd(&aux[q[it] >> 1], 1u << sh);
        sbuf[(old >> sh) & 0xffffu] = wv[it];
      }
    }
    __syncthreads();
    bool crowded = false;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      if (it < nit && j < m) {
        const uint32_t b = q[it];
        const uint32_t hi = (aux[b >> 1] >> ((b & 1u) * 16)) & 0xffffu;
        const uint32_t lo = b ? (aux[(b - 1) >> 1] >> (((b - 1) & 1u) * 16)) & 0xffffu : 0u;
        uint32_t pos = lo;
        if (hi - lo > (uint32_t)TS_DENSE) {
          crowded = true;
        } else {
          // (value, slot number) decides; the carried low bits in between must not
          const uint64_t mine = wv[it] & cmp_mask;
          for (uint32_t t = lo; t < hi; ++t) pos += (sbuf[t] & cmp_mask) < mine ? 1u : 0u;
        }
        q[it] = pos;
      }
    }
    if (__any(crowded) && lane == 0) dense_s = 1;
    __syncthreads();
    dense = dense_s != 0;
    if (!dense) {
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = wave * chunk + it * 64 + lane;
        if (it < nit && j < m) sbuf[q[it]] = wv[it];
      }
    }
  }
  TICK();
  if (dense || span == 0) {
    // ---- radix passes (values bunched inside the tile, or nothing to sort: one plain copy)
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int j = wave * chunk + it * 64 + lane;
      if (it < nit && j < m) sbuf[j] = wv[it];
    }
    const int passes = (span + 7) / 8;
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
      const int shift = TS_IDX_BITS + low_bits + 8 * p;
      __syncthreads();
      for (int i = tid; i < NW * 256; i += NT) aux[i] = 0;
      __syncthreads();  // also orders the previous pass's (or the initial) writes of sbuf before the reads
      // rank of every slot among the slots of its wave with the same digit; only the digit is read
      // here, the words stay in LDS until the scatter, so few registers are live while ranking
      uint32_t rk[NI];  // rank | digit << 16, then the slot's destination
#pragma unroll
      for (int it = 0; it < NI; ++it) rk[it] = (uint32_t)(sbuf[wave * chunk + it * 64 + lane] >> shift) & 255u;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        if (it < nit) {  // wave-uniform
          const int j = wave * chunk + it * 64 + lane;
          const bool valid = j < m;
          const uint32_t d = valid ? rk[it] : 0u;
          const uint64_t vb = __ballot(valid);  // the valid lanes are the first ones
          const uint32_t dprev = (uint32_t)__shfl_up((int)d, 1, 64);
          uint32_t cnt, before;  // lanes of the wave with this digit, and how many of them come first
          bool leader;
          if (__all(!valid || lane == 0 || d >= dprev)) {
            // digits already in order across the wave (slices of sorted runs): equal digits sit
            // next to each other, so a lane's peers are the stretch between two changes of digit
            const uint64_t starts = __ballot(valid && (lane == 0 || d != dprev));
            const uint64_t upto = lt_mask | (1ull << lane);
            const int first = 63 - __clzll((long long)(starts & upto));
            const uint64_t above = (starts | ~vb) & ~upto;
            const int next = above ? __ffsll((long long)above) - 1 : 64;
            cnt = (uint32_t)(next - first);
            before = (uint32_t)(lane - first);
            leader = valid && lane == first;
          } else {
            uint32_t plo = (uint32_t)vb, phi = (uint32_t)(vb >> 32);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // 0 or ~0
              const uint64_t bal = __ballot(mb != 0u);
              plo &= ~((uint32_t)bal ^ mb);
              phi &= ~((uint32_t)(bal >> 32) ^ mb);
            }
            const uint64_t peers = ((uint64_t)phi << 32) | plo;
            cnt = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
            before = (uint32_t)__popcll(peers & lt_mask);
            leader = valid && (lane == (__ffsll((long long)peers) - 1));
          }
          uint32_t prior = __hip_atomic_load(&whist[wave][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (leader)
            __hip_atomic_store(&whist[wave][d], prior + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          rk[it] = (prior + before) | (d << 16);
        }
      }
      __syncthreads();
      uint32_t cnt_d = 0, cnt_inc = 0;
      if (tid < 256) {
        for (int w = 0; w < NW; ++w) {
          const uint32_t c = whist[w][tid];
          whist[w][tid] = cnt_d;
          cnt_d += c;
        }
        uint32_t inc = cnt_d;
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t tt = __shfl_up(inc, o, 64);
          if (lane >= o) inc += tt;
        }
        if (lane == 63) wtot[wave] = inc;
        cnt_inc = inc;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t wp = 0;
        for (int w = 0; w < wave; ++w) wp += wtot[w];
        dstart[tid] = wp + cnt_inc - cnt_d;
      }
      __syncthreads();
      // in place: every slot is taken into registers before any slot is rewritten
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = wave * chunk + it * 64 + lane;
        const uint32_t d = (rk[it] >> 16) & 255u;
        wv[it] = sbuf[j];
        rk[it] = dstart[d] + whist[wave][d] + (rk[it] & 0xffffu);
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = wave * chunk + it * 64 + lane;
        if (it < nit && j < m) sbuf[rk[it]] = wv[it];
      }
    }
  }
  __syncthreads();
  TICK();
  // ---- out: words in order; the row ids change places through the same buffer
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int j = tid + it * NT;
    wv[it] = j < m ? sbuf[j] : 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int j = wave * chunk + it * 64 + lane;
    if (it < nit && j < m) sbuf32[j] = idv[it];
  }
  __syncthreads();
  const uint64_t idx_mask = (1ull << TS_IDX_BITS) - 1;
#pragma unroll
  for (int it = 0; it < NI; ++it) idv[it] = sbuf32[(uint32_t)(wv[it] & idx_mask)];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int j = tid + it * NT;
    if (j < m) {
      const uint64_t pk = wv[it];
      const uint32_t id = idv[it];
      uint64_t w = (pk >> TS_IDX_BITS) + wmin;
      if (MERGE) {
        if (ride) {
          const uint64_t x = w & low_mask;
          w = (((w >> low_bits) + tmin) << gen.lay.seq_shift()) | (x & ((1ull << gen.n_masks) - 1));
          if (gen.lay.key_bits) w |= (x >> gen.n_masks) << gen.lay.key_shift();
        } else {
          w = make_word(gen, (int64_t)id);
        }
      }
      wout[s + j] = w;
      idout[s + j] = id;
      if (!pp.positional) {
#pragma unroll
        for (int c = 0; c < kMaxPay; ++c)
          if (c < pp.n) pp.out[c][s + j] = pp.src[c][id];
      }
    }
  }
  if (pp.positional) {
    // payload that travelled with the pairs: each column is read as it lies (coalesced) into the
    // tile buffer and leaves in sorted order from there, instead of one 8-byte fetch per row
    // from wherever its slot was
#pragma unroll 1
    for (int c = 0; c < pp.n; ++c) {
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = tid + it * NT;
        if (j < m) sbuf[j] = pp.src[c][s + j];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int j = tid + it * NT;
        if (j < m) pp.out[c][s + j] = sbuf[(uint32_t)(wv[it] & idx_mask)];
      }
    }
  }
  TICK();
#ifdef TS_TIMING
  if (tid == 0 && (blockIdx.x % 9973) == 500)
    if (ntk >= 10) {
      printf("M tile %d m %d dense %d: cuts %lld lookup %lld loads %lld time_of %lld rest %lld minmax %lld interp %lld radix %lld out %lld\n", (int)blockIdx.x, m, (int)dense,
           tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6], tk[8] - tk[7], tk[9] - tk[8]);
    } else {
      printf("S tile %d m %d dense %d: load %lld minmax %lld interp %lld radix %lld out %lld\n", (int)blockIdx.x, m, (int)dense,
           tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
    }
#endif
}

// Copies the stand-by sort's result to where the tile kernel would have left it (only when the
// two end in different halves of the ping-pong buffers, and only if the stand-by ran).
__global__ void copy_pairs_if_kernel(const uint64_t *win, const uint32_t *idin, uint64_t *wout, uint32_t *idout,
                                     int64_t n, const int32_t *run_if) {
  if (!*run_if) return;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    wout[i] = win[i];
    idout[i] = idin[i];
  }
}

// Wide sequence columns: words that hold just the sequence field (0 = NULL, else value - min + 1)
// with identity row ids, and -- after the passes over those -- the real sort words taken in the
// order the row ids then stand in.
__global__ __launch_bounds__(kBlock) void seq_words_kernel(int64_t n, const int64_t *seq, const uint8_t *seq_valid,
                                                           int64_t seq_min, uint64_t *words, uint32_t *idx) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool sv = !seq_valid || seq_valid[i];
    words[i] = sv ? (uint64_t)(seq[i] - seq_min) + 1ull : 0ull;
    idx[i] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(kBlock) void make_comp_by_id_kernel(int64_t n, GenArgs g, const uint32_t *idx,
                                                                 uint64_t *comp) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
    comp[p] = make_word(g, (int64_t)idx[p]);
}

__global__ __launch_bounds__(kBlock) void gather_masks_kernel(int64_t n, const uint32_t *perm, int n_masks,
                                                              MaskPtrs masks, uint16_t *out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t i = perm[p];
    uint32_t bits = 0;
    for (int m = 0; m < n_masks; ++m)
      if (!masks.p[m] || masks.p[m][i]) bits |= 1u << m;
    out[p] = (uint16_t)bits;
  }
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
void order_plan(int64_t n, Workspace &ws, OrderBuffers *b, int64_t n_keys, int carry_cols) {
  size_t cnt = (size_t)(n > 0 ? n : 1);
  b->comp[0] = ws.take<uint64_t>(cnt);
  b->comp[1] = ws.take<uint64_t>(cnt);
  b->idx[0] = ws.take<uint32_t>(cnt);
  b->idx[1] = ws.take<uint32_t>(cnt);
  b->counts = ws.take<uint32_t>(((size_t)1 << RS_MAX_BITS) * RS_MAX_GRID);
  b->range = ws.take<RangeInfo>(1);
  b->key_rows = nullptr;
  b->key_nb = b->key_base = b->bucket_key = nullptr;
  b->fallback = nullptr;
  b->tile_bounds = nullptr;
  b->max_buckets = 0;
  b->run_if = nullptr;
  b->carry_cols = 0;
  b->key_hist_n = 0;
  b->narrow_words = false;
  b->key_hist_mask = 0;
  b->key_hist_of = nullptr;
  for (int c = 0; c < 2 * kMaxPay; ++c) b->carry[c] = nullptr;
  b->n_runs = 0;
  b->run_cuts = ws.take<int64_t>((size_t)(n / TS_CAP + 3) * kMaxRuns);
  b->fallback = ws.take<int32_t>(1);
  const BucketGeom g = bucket_geometry(n, n_keys);
  if (g.ok) {
    b->key_rows = ws.take<unsigned long long>((size_t)n_keys);
    b->key_nb = ws.take<uint32_t>((size_t)n_keys);
    b->key_base = ws.take<uint32_t>((size_t)n_keys);
    b->bucket_key = ws.take<uint32_t>((size_t)g.max_buckets);
    b->tile_bounds = ws.take<int64_t>((size_t)(n / TS_TILE + 3));
    b->max_buckets = g.max_buckets;
    b->carry_cols = carry_cols < kMaxPay ? carry_cols : kMaxPay;
    for (int c = 0; c < 2 * b->carry_cols; ++c) b->carry[c] = ws.take<uint64_t>(cnt);
  }
}

// ------------------------------------------------------------------------------------------
// Is every key's own row sequence in order?  A sample: kKoWindows windows of kKoRows consecutive
// rows each, one wave per window.  The wave walks its window 64 rows at a time and keeps, per key
// seen so far, the (primary, seq) of that key's latest row in a small LDS hash table; every row
// that has a predecessor of its key inside the window is one pair, a pair that steps back one
// violation.  The result only steers which ordering is tried first -- it proves nothing.
// ------------------------------------------------------------------------------------------
constexpr int kKoWindows = 512;
constexpr int kKoRows = 1024;
constexpr int kKoSlots = 2048;
constexpr int64_t kKoMinRows = 1 << 20;  // smaller tables: not worth a look

__global__ __launch_bounds__(64) void key_order_sample_kernel(int64_t n, const int32_t *key, const int64_t *ts,
                                                              int64_t div, const int64_t *seq, const uint8_t *seq_valid,
                                                              RangeInfo *out) {
  __shared__ uint32_t tkey[kKoSlots];
  __shared__ int64_t tprim[kKoSlots], tseq[kKoSlots];
  const int lane = threadIdx.x;
  for (int i = lane; i < kKoSlots; i += 64) tkey[i] = 0xffffffffu;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int64_t start = (int64_t)blockIdx.x * (n / kKoWindows);
  if (start + kKoRows > n) start = n - kKoRows;
  int pairs = 0, viol = 0;
  for (int bt = 0; bt < kKoRows / 64; ++bt) {
    const int64_t i = start + bt * 64 + lane;
    const uint32_t k = (uint32_t)key[i];
    const int64_t p = div == 1 ? ts[i] : floordiv_i64(ts[i], div);
    // NULL sequence values sort first: below every real one
    const bool sv = seq && (!seq_valid || seq_valid[i]);
    const int64_t q = seq ? (sv ? seq[i] : INT64_MIN) : 0;
    // lanes of this step with my key
    uint64_t peers = ~0ull;
    for (int b = 0; b < 32; ++b) {
      const bool bit = (k >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint64_t before = peers & lt_mask;
    bool have = false;
    int64_t pp = 0, pq = 0;
    const int prev = before ? 63 - __clzll((long long)before) : lane;
    const int64_t sp = __shfl(p, prev, 64), sq = __shfl(q, prev, 64);
    uint32_t h = (k * 2654435761u) >> 21;  // 11 bits
    if (before) {
      have = true; pp = sp; pq = sq;
    } else {
      for (int probe = 0; probe < kKoSlots; ++probe) {
        const uint32_t kk = tkey[h];
        if (kk == k) { have = true; pp = tprim[h]; pq = tseq[h]; break; }
        if (kk == 0xffffffffu) break;
        h = (h + 1) & (kKoSlots - 1);
      }
    }
    if (have) {
      ++pairs;
      if (p < pp || (p == pp && q < pq)) ++viol;
    }
    // the key's last lane of this step leaves its values for the steps to come (keys of one step
    // are distinct among these lanes; the window holds fewer rows than the table has slots)
    if (!(peers & ~lt_mask & ~(1ull << lane))) {
      h = (k * 2654435761u) >> 21;
      for (int probe = 0; probe < kKoSlots; ++probe) {
        const uint32_t old = atomicCAS(&tkey[h], 0xffffffffu, k);
        if (old == 0xffffffffu || old == k) { tprim[h] = p; tseq[h] = q; break; }
        h = (h + 1) & (kKoSlots - 1);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    pairs += __shfl_xor(pairs, o, 64);
    viol += __shfl_xor(viol, o, 64);
  }
  if (lane == 0) {
    if (pairs) atomicAdd(&out->key_pairs, pairs);
    if (viol) atomicAdd(&out->key_viol, viol);
  }
}

// Raises *flag when the words are not in order (compared above their low bits).
__global__ __launch_bounds__(kBlock) void ordered_check_kernel(const uint64_t *w, int64_t n, int low_bits, int32_t *flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride)
    bad |= (w[i] >> low_bits) < (w[i - 1] >> low_bits);
  if (__any(bad) && (threadIdx.x & 63) == 0 && __atomic_load_n(flag, __ATOMIC_RELAXED) == 0) atomicOr(flag, 1);
}

int order_scan_ranges(const OrderSpec &s, OrderBuffers &b, hipStream_t st) {
  TMP_LAUNCH(range_init_kernel, dim3(1), dim3(1), st, b.range);
  b.key_hist_n = 0;
  if (s.n > 0) {
    // A large table with a short key dictionary and no sequence column may turn out to be a
    // time-ordered feed, whose one pass over the key field wants the histogram of the key column: the
    // scan reads that column anyway and leaves the histogram behind (sort_geom of the pass that uses it).
    const int key_bits = s.key ? bits_for((uint64_t)(s.n_keys > 1 ? s.n_keys - 1 : 0)) : 0;
    const bool with_hist = s.key && !s.seq && key_bits >= 1 && key_bits <= 8 && s.n >= (1 << 20) &&
                           (((uintptr_t)s.ts | (uintptr_t)s.key) & 15) == 0 && !getenv("TEMPO_NO_FUSED_WORDS") &&
                           !getenv("TEMPO_NO_SCAN_HIST");
    if (with_hist) {
      const SortGeom g = sort_geom(s.n, 1024 * kFusedItems);
      const uint32_t mask = (1u << key_bits) - 1;
      // which of the two scans does the work is settled on the device, by a sample of adjacent rows
      TMP_LAUNCH(time_order_sample_kernel, dim3(1), dim3(256), st, s.n, s.ts, b.range);
      {
        ProfScope ps("range_scan_kernel+keys", st);
        hipLaunchKernelGGL((range_scan_hist_kernel<1024 * kFusedItems>), dim3(g.grid), dim3(RH_THREADS), 0, st, s.n, s.key,
                           s.ts, s.prim_div, b.range, b.counts, g.tiles_per_block, mask);
      }
      b.key_hist_n = s.n;  // (dropped again when the record read back says the plain scan ran)
      b.key_hist_mask = mask;
      b.key_hist_of = s.key;
    }
    {
      int grid = grid_for(s.n, kBlock * 16, 4096);
      TMP_LAUNCH(range_scan_kernel, dim3(grid), dim3(kBlock), st, s.n, s.key, s.ts,
                 s.prim_div, s.seq, s.seq_valid, b.range, with_hist ? 1 : 0);
    }
    if (s.key && s.n >= kKoMinRows && !getenv("TEMPO_NO_KEY_ORDER"))
      TMP_LAUNCH(key_order_sample_kernel, dim3(kKoWindows), dim3(64), st, s.n, s.key, s.ts, s.prim_div, s.seq,
                 s.seq_valid, b.range);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

int order_read_ranges(OrderBuffers &b, RangeInfo *host, hipStream_t st) {
  TMP_CHECK_HIP(hipMemcpyAsync(host, b.range, sizeof(RangeInfo), hipMemcpyDeviceToHost, st));
  TMP_CHECK_HIP(hipStreamSynchronize(st));
  if (!host->key_hist) b.key_hist_n = 0;
  b.n_runs = 0;
  if (sorted_mode(*host) == 3) {  // a few time-ordered runs end to end: where each starts
    int64_t pos[kMaxDesc];
    const int d = host->n_desc;
    for (int i = 0; i < d; ++i) pos[i] = host->desc_pos[i];
    for (int i = 1; i < d; ++i)  // insertion sort of at most 15 positions
      for (int j = i; j > 0 && pos[j - 1] > pos[j]; --j) { int64_t t = pos[j]; pos[j] = pos[j - 1]; pos[j - 1] = t; }
    b.n_runs = d + 1;
    b.run_start[0] = 0;
    for (int i = 0; i < d; ++i) b.run_start[i + 1] = pos[i];
  }
  return TMP_OK;
}

int order_make_layout(const OrderSpec &s, const RangeInfo &r, int low_bits, Layout *out) {
  Layout l;
  l.low_bits = low_bits;
  l.prim_div = s.prim_div;
  l.prim_min = r.prim_min <= r.prim_max ? r.prim_min : 0;
  l.prim_max = r.prim_min <= r.prim_max ? r.prim_max : 0;
  uint64_t prange = r.prim_min <= r.prim_max ? (uint64_t)r.prim_max - (uint64_t)r.prim_min : 0;
  l.prim_gran = 1;
  if (s.coarse_ok) l.prim_gran = !r.not_millions ? 1000000 : (!r.not_thousands ? 1000 : 1);
  prange /= (uint64_t)l.prim_gran;
  l.prim_bits = bits_for(prange);
  l.seq_bits = 0;
  l.seq_min = 0;
  if (s.seq) {
    if (r.seq_min <= r.seq_max) {
      l.seq_min = r.seq_min;
      uint64_t srange = (uint64_t)r.seq_max - (uint64_t)r.seq_min;
      TMP_REQUIRE(srange < (1ull << 62), TMP_EUNSUPPORTED, "sequence column range too wide");
      l.seq_bits = bits_for(srange + 1);
    } else {
      l.seq_bits = 1;  // every sequence value is NULL
    }
  }
  l.key_bits = s.key ? bits_for((uint64_t)(s.n_keys > 1 ? s.n_keys - 1 : 0)) : 0;
  l.seq_wide = 0;
  l.seq_full_bits = l.seq_bits;
  l.ext_masks = 0;
  if (l.total_bits() > 64 && l.seq_bits > 1) {  // sequence values sorted in passes of their own
    l.seq_wide = 1;
    l.seq_bits = 1;
  }
  if (l.total_bits() > 64 && l.low_bits > 0 && s.ext_masks_ok) {  // validity bits beside the words
    l.ext_masks = l.low_bits;
    l.low_bits = 0;
  }
  TMP_REQUIRE(l.total_bits() <= 64, TMP_EUNSUPPORTED,
              "sort word needs %d bits (key %d + time %d + sequence %d + masks %d) > 64",
              l.total_bits(), l.key_bits, l.prim_bits, l.seq_bits, l.low_bits);
  *out = l;
  return TMP_OK;
}

// One stable LSD pass (histogram, scan, scatter) over `nbits` bits at `lo`.  gen != nullptr: the
// pass is the first one over freshly laid out rows -- its histogram kernel builds the sort words
// (into b.comp[cur]) and its row ids are the positions.
template <int THREADS, int ITEMS, int NPAY, int BITS, int PACK>
static void radix_scatter_launch(OrderBuffers &b, int64_t n, int cur, int shift, uint32_t mask, int nbits,
                                 hipStream_t st, const PayPtrs &pp, bool first, const SortGeom &g,
                                 int id_bits) {
  GenArgs none;
  memset(&none, 0, sizeof(none));
  if (first)
    hipLaunchKernelGGL((radix_scatter_kernel<THREADS, ITEMS, NPAY, BITS, true, PACK>), dim3(g.grid), dim3(THREADS),
                       0, st, b.comp[cur], b.idx[cur], b.comp[cur ^ 1], b.idx[cur ^ 1], n, shift, mask, nbits,
                       g.tiles_per_block, b.counts, pp, id_bits, b.run_if, none);
  else
    hipLaunchKernelGGL((radix_scatter_kernel<THREADS, ITEMS, NPAY, BITS, false, PACK>), dim3(g.grid), dim3(THREADS),
                       0, st, b.comp[cur], b.idx[cur], b.comp[cur ^ 1], b.idx[cur ^ 1], n, shift, mask, nbits,
                       g.tiles_per_block, b.counts, pp, id_bits, b.run_if, none);
}

// Whether a first pass may take its digits from the key column and build the words in the
// scatter: the pass lies inside the key field of plain (not bucketed, not id-carrying) words.
static bool genw_ok(const GenArgs *gen, int lo, int pack) {
  return gen && gen->key && !gen->key_pack && gen->id_bits == 0 && pack == 0 && lo >= gen->lay.key_shift() &&
         !(gen->lay.seq_bits && gen->seq) &&
         !getenv("TEMPO_NO_FUSED_WORDS");
}

// The first pass of rows that are time-ordered already, over (part of) the key field: histogram of
// the key column, then one sweep that builds the words, moves them, the row ids (= positions) and
// the columns that travel with the rows (NPAY = 0 or -1).
template <int THREADS, int ITEMS, int NPAY, int BITS>
static void radix_first_pass_fused(OrderBuffers &b, int64_t n, int cur, int lo, int nbits, hipStream_t st,
                                   const PayPtrs &pp, const GenArgs &gen, bool narrow = false) {
  SortGeom g = sort_geom(n, THREADS * ITEMS);
  const uint32_t mask = (1u << nbits) - 1;
  // the range scan may have left this very histogram in b.counts (same rows, same tiles, the digit
  // being the key's low bits: the pass starts at the key field's first bit)
  const bool have_hist = b.key_hist_n == n && b.key_hist_mask == mask && b.key_hist_of == (const void *)gen.key && BITS == 8 && THREADS * ITEMS == 1024 * kFusedItems &&
                         lo == gen.lay.key_shift() && !b.run_if;
  b.key_hist_n = 0;
  {
    ProfScope ps("radix_hist_kernel+keys", st);
    if (!have_hist)
      hipLaunchKernelGGL((radix_hist_kernel<THREADS, ITEMS, BITS, 3>), dim3(g.grid), dim3(THREADS), 0, st,
                         (const uint64_t *)nullptr, n, lo, mask, g.tiles_per_block, b.counts, gen, (uint64_t *)nullptr, b.run_if);
    TMP_LAUNCH(radix_scan_kernel, dim3(1), dim3(1024), st, b.counts, (1 << BITS) * g.grid, b.run_if);
  }
  {
    ProfScope ps(NPAY == 0 ? "radix_scatter_kernel+words" : (narrow ? "radix_scatter_kernel+words32+carried" : "radix_scatter_kernel+words+carried"), st);
    if (narrow)
      hipLaunchKernelGGL((radix_scatter_kernel<THREADS, ITEMS, NPAY, BITS, true, 0, true, true>), dim3(g.grid), dim3(THREADS),
                         0, st, b.comp[cur], b.idx[cur], b.comp[cur ^ 1], b.idx[cur ^ 1], n, lo, mask, nbits,
                         g.tiles_per_block, b.counts, pp, 0, b.run_if, gen);
    else
      hipLaunchKernelGGL((radix_scatter_kernel<THREADS, ITEMS, NPAY, BITS, true, 0, true>), dim3(g.grid), dim3(THREADS),
                         0, st, b.comp[cur], b.idx[cur], b.comp[cur ^ 1], b.idx[cur ^ 1], n, lo, mask, nbits,
                         g.tiles_per_block, b.counts, pp, 0, b.run_if, gen);
  }
}

// One stable LSD pass (histogram, scan, scatter) over `nbits` bits at `lo`.  gen != nullptr: the
// pass is the first one over freshly laid out rows -- its histogram kernel builds the sort words
// (into b.comp[cur]) and its row ids are the positions.  pack: 0 = words and row ids travel
// separately, 1 / 2 = the words carry the row id in their low id_bits bits (see the kernel).
template <int THREADS, int ITEMS, int NPAY, int BITS>
static void radix_one_pass_t(OrderBuffers &b, int64_t n, int cur, int lo, int nbits, hipStream_t st,
                             const PayPtrs &pp, const GenArgs *gen, bool hist_done, int pack, int id_bits) {
  SortGeom g = sort_geom(n, THREADS * ITEMS);
  uint32_t mask = (1u << nbits) - 1;
  const int shift = lo + (pack ? id_bits : 0);
  GenArgs none;
  memset(&none, 0, sizeof(none));
  b.key_hist_n = 0;  // whatever the range scan left in b.counts is about to be overwritten
  if (!hist_done) {
    ProfScope ps(gen ? "radix_hist_kernel+words" : "radix_hist_kernel", st);
    if (gen && gen->id_bits)
      hipLaunchKernelGGL((radix_hist_kernel<THREADS, ITEMS, BITS, 2>), dim3(g.grid), dim3(THREADS), 0, st,
                         (const uint64_t *)nullptr, n, shift, mask, g.tiles_per_block, b.counts, *gen, b.comp[cur], b.run_if);
    else if (gen)
      hipLaunchKernelGGL((radix_hist_kernel<THREADS, ITEMS, BITS, 1>), dim3(g.grid), dim3(THREADS), 0, st,
                         (const uint64_t *)nullptr, n, shift, mask, g.tiles_per_block, b.counts, *gen, b.comp[cur], b.run_if);
    else
      hipLaunchKernelGGL((radix_hist_kernel<THREADS, ITEMS, BITS, 0>), dim3(g.grid), dim3(THREADS), 0, st,
                         b.comp[cur], n, shift, mask, g.tiles_per_block, b.counts, none, (uint64_t *)nullptr, b.run_if);
    TMP_LAUNCH(radix_scan_kernel, dim3(1), dim3(1024), st, b.counts, (1 << BITS) * g.grid, b.run_if);
  }
  {
    ProfScope ps(NPAY == 0 ? (pack ? "radix_scatter_kernel+ids_in_words" : "radix_scatter_kernel")
                           : (NPAY < 0 ? "radix_scatter_kernel+carried" : "radix_scatter_kernel+payload"), st);
    const bool first = gen != nullptr;
    if (pack == 2)
      radix_scatter_launch<THREADS, ITEMS, NPAY, BITS, 2>(b, n, cur, shift, mask, nbits, st, pp, first, g, id_bits);
    else if (pack == 1 && NPAY == 0)
      radix_scatter_launch<THREADS, ITEMS, 0, BITS, 1>(b, n, cur, shift, mask, nbits, st, pp, first, g, id_bits);
    else
      radix_scatter_launch<THREADS, ITEMS, NPAY, BITS, 0>(b, n, cur, shift, mask, nbits, st, pp, first, g, 0);
  }
}

#ifndef TEMPO_RS_ITEMS
#define TEMPO_RS_ITEMS 12
#endif
constexpr int kPlainItems = TEMPO_RS_ITEMS;  // rows per thread of the passes that move pairs (and travelling columns) only

template <int THREADS, int ITEMS, int NPAY>
static void radix_one_pass(OrderBuffers &b, int64_t n, int cur, int lo, int nbits, hipStream_t st,
                           const PayPtrs &pp, const GenArgs *gen, bool hist_done, int pack, int id_bits) {
  if (nbits <= 8)
    radix_one_pass_t<THREADS, ITEMS, NPAY, 8>(b, n, cur, lo, nbits, st, pp, gen, hist_done, pack, id_bits);
  else
    radix_one_pass_t<THREADS, ITEMS, NPAY, RS_MAX_BITS>(b, n, cur, lo, nbits, st, pp, gen, hist_done, pack, id_bits);
}

// Stable LSD passes over bits [lo_bit, hi_bit) of the pairs in b.comp[0] / b.idx[0] -- or, with
// `gen`, of the rows themselves: the first pass then lays out the sort words on the way and
// takes the positions as row ids (lo_bit < hi_bit is the caller's business then).  Digits are
// 8 bits wide, or a mix of 8 and 9 where that takes one pass fewer.  With payload columns the
// last pass also delivers out[c][p] = src[c][perm[p]] (one column per launch; more columns
// replay the last pass); *carried tells whether a pass ran at all (no pass: the order is the
// identity and the outputs are untouched).
static int radix_passes(OrderBuffers &b, int64_t n, int lo_bit, int hi_bit, OrderResult *res,
                        hipStream_t st, int n_pay, const uint64_t *const *pay_src,
                        uint64_t *const *pay_out, bool *carried, const GenArgs *gen,
                        uint64_t *const *carry_buf = nullptr, int *carry_at = nullptr, int start_cur = 0) {
  int cur = start_cur;
  bool narrow = false;
  if (carried) *carried = false;
  PayPtrs none;
  memset(&none, 0, sizeof(none));
  const int total = hi_bit > lo_bit ? hi_bit - lo_bit : 0;
  int passes = (total + 7) / 8;
  if ((total + RS_MAX_BITS - 1) / RS_MAX_BITS < passes) passes = (total + RS_MAX_BITS - 1) / RS_MAX_BITS;
  int lo = lo_bit;
  // Narrow sort words leave room for the row id below them: the passes then move one 8-byte
  // stream, and the last pass hands out plain words and row ids again.  Only when the words are
  // laid out here (gen), so that nobody else ever sees the packed form.
  const int id_bits = bits_for((uint64_t)(n > 1 ? n - 1 : 1));
  const bool packed = gen != nullptr && passes > 1 && hi_bit + id_bits <= 64 && !carry_buf;
  GenArgs gen_packed;
  if (packed) {
    gen_packed = *gen;
    gen_packed.id_bits = id_bits;
  }
  for (int p = 0; p < passes; ++p) {
    // spread the bits evenly: the first (total % passes) digits are one bit wider
    int nbits = total / passes + (p < total % passes ? 1 : 0);
    bool last = p + 1 == passes;
    const GenArgs *g = p == 0 ? (packed ? &gen_packed : gen) : nullptr;
    const int pack = packed ? (last ? 2 : 1) : 0;
    if (carry_buf) {
      // the payload column moves with the rows in every pass: read where the rows are read,
      // written where they are written
      PayPtrs pp;
      memset(&pp, 0, sizeof(pp));
      pp.n = n_pay;
      pp.positional = 1;
      for (int c = 0; c < n_pay; ++c) {
        pp.src[c] = p == 0 ? pay_src[c] : carry_buf[2 * c + ((p - 1) & 1)];
        // the last pass delivers the column where the caller wants it, if the caller says where
        pp.out[c] = (last && pay_out) ? pay_out[c] : carry_buf[2 * c + (p & 1)];
      }
      if (p == 0 && genw_ok(g, lo, pack)) {
        if (nbits <= 8) radix_first_pass_fused<1024, kFusedItems, -1, 8>(b, n, cur, lo, nbits, st, pp, *g);
        else radix_first_pass_fused<1024, kFusedItems, -1, RS_MAX_BITS>(b, n, cur, lo, nbits, st, pp, *g);
      } else {
        radix_one_pass<1024, kPlainItems, -1>(b, n, cur, lo, nbits, st, pp, g, false, pack, id_bits);
      }
      if (carry_at) *carry_at = p & 1;
    } else if (last && n_pay > 0) {
      // the last pass also fetches the payload through the row ids: one column with six rows per
      // thread, two to kMaxPay columns with four (the registers that hold the fetched values bound it)
      PayPtrs pp;
      memset(&pp, 0, sizeof(pp));
      pp.n = n_pay;
      for (int c = 0; c < n_pay; ++c) { pp.src[c] = pay_src[c]; pp.out[c] = pay_out[c]; }
      if (p == 0 && genw_ok(g, lo, pack)) {
        // the only pass: the row ids are the positions, so the payload is read where the rows are
        // read (no dependent load), and the words are built on the way -- as 32-bit words where the
        // caller takes them and they fit (nothing else will read them: no stand-by passes are enqueued)
        pp.positional = 1;
        narrow = b.narrow_words && nbits <= 8 && g->lay.total_bits() <= 32 && !b.run_if && !getenv("TEMPO_NO_NARROW_WORDS");
        if (nbits <= 8) radix_first_pass_fused<1024, kFusedItems, -1, 8>(b, n, cur, lo, nbits, st, pp, *g, narrow);
        else radix_first_pass_fused<1024, kFusedItems, -1, RS_MAX_BITS>(b, n, cur, lo, nbits, st, pp, *g);
      } else if (n_pay == 1)
        radix_one_pass<1024, 6, 1>(b, n, cur, lo, nbits, st, pp, g, false, pack, id_bits);
      else
        radix_one_pass<1024, 4, kMaxPay>(b, n, cur, lo, nbits, st, pp, g, false, pack, id_bits);
      if (carried) *carried = true;
    } else if (p == 0 && genw_ok(g, lo, pack)) {
      if (nbits <= 8) radix_first_pass_fused<1024, kFusedItems, 0, 8>(b, n, cur, lo, nbits, st, none, *g);
      else radix_first_pass_fused<1024, kFusedItems, 0, RS_MAX_BITS>(b, n, cur, lo, nbits, st, none, *g);
    } else {
      radix_one_pass<1024, kPlainItems, 0>(b, n, cur, lo, nbits, st, none, g, false, pack, id_bits);
    }
    lo += nbits;
    cur ^= 1;
  }
  TMP_CHECK_HIP(hipGetLastError());
  res->comp = b.comp[cur];
  res->perm = b.idx[cur];
  res->narrow = narrow;
  return TMP_OK;
}

int order_radix_passes(OrderBuffers &b, int64_t n, int lo_bit, int hi_bit, OrderResult *res,
                       hipStream_t st, int n_pay, const uint64_t *const *pay_src,
                       uint64_t *const *pay_out, bool *carried) {
  return radix_passes(b, n, lo_bit, hi_bit, res, st, n_pay, pay_src, pay_out, carried, nullptr);
}

static int passes_for(int bits) {
  int passes = (bits + 7) / 8;
  if ((bits + RS_MAX_BITS - 1) / RS_MAX_BITS < passes) passes = (bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  return passes;
}

// The bucketed ordering serves a full sort of a large table whose buffers were planned for it,
// when the bucket field fits the word and the passes saved outweigh the tile kernel.
static bool bucketed_wanted(const OrderSpec &s, const Layout &lay, int sorted, const OrderBuffers &b, BucketGeom *g) {
  if (sorted != 0 || !b.key_nb || !b.fallback || lay.seq_wide) return false;
  if (getenv("TEMPO_PLAIN_SORT")) return false;  // diagnostic switch: always the plain LSD passes
  const int64_t nk = plan_keys(s.key, s.n_keys);
  *g = bucket_geometry(s.n, nk);
  if (!g->ok || g->max_buckets > b.max_buckets) return false;
  if (lay.low_bits + lay.seq_bits + lay.prim_bits + g->bits > 64) return false;
  if (lay.prim_bits + g->bits > 63) return false;  // (time field) x (buckets of a key) is formed in 64 bits
  return passes_for(lay.total_bits() - lay.low_bits) >= passes_for(g->bits) + 2;
}

// out[c][j] = src[c][perm[j]]: payload columns for an order that was not produced by a radix pass
// (a pass would have delivered them).
__global__ __launch_bounds__(kBlock) void pay_gather_kernel(int64_t n, const uint32_t *perm, PayPtrs pp) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const uint32_t r = perm[j];
    for (int c = 0; c < pp.n; ++c) pp.out[c][j] = pp.src[c][r];
  }
}

int order_build(const OrderSpec &s, const Layout &lay, int sorted, OrderBuffers &b,
                OrderResult *res, hipStream_t st, int n_pay, const uint64_t *const *pay_src,
                uint64_t *const *pay_out, bool *carried) {
  if (carried) *carried = false;
  res->comp = b.comp[0];
  res->perm = b.idx[0];
  res->narrow = false;
  // 32-bit words are on offer to one route only (a time-ordered feed whose key field takes one pass):
  // every other route re-reads its words, checks them or has stand-by passes that write 64-bit ones
  const bool narrow_wanted = b.narrow_words;
  b.narrow_words = false;
  if (s.n <= 0) return TMP_OK;
  GenArgs gen;
  memset(&gen, 0, sizeof(gen));
  gen.key = s.key; gen.ts = s.ts; gen.seq = s.seq; gen.seq_valid = s.seq_valid;
  gen.lay = lay;
  gen.n_masks = lay.ext_masks ? 0 : s.n_masks;  // validity bits beside the words: not in them
  for (int m = 0; m < TMP_MAX_MASKS; ++m) gen.masks.p[m] = m < s.n_masks ? s.masks[m] : nullptr;
  gen.vec4 = gen_vector_ok(gen) ? 1 : 0;
  const int lo_bit = sorted == 2 ? lay.key_shift() : lay.low_bits, hi_bit = lay.total_bits();
  if (sorted == 1 || lo_bit >= hi_bit) {  // no pass to run: the words in input order are the result
    // (the row ids go to the second id buffer: where a caller has put its own output array in that
    // buffer's place -- as for an ordering that ends there -- they are where they belong at once)
    int grid = grid_for(s.n, kBlock * 4, 8192);
    TMP_LAUNCH(make_comp_kernel, dim3(grid), dim3(kBlock), st, s.n, gen, b.comp[0], b.idx[1]);
    TMP_CHECK_HIP(hipGetLastError());
    res->perm = b.idx[1];
    return TMP_OK;
  }
  if (sorted == 3 && (lay.seq_wide || !b.run_cuts || b.n_runs < 2 || b.n_runs > kMaxRuns ||
                      lay.prim_bits + lay.seq_bits + TS_IDX_BITS > 64))
    sorted = 0;
  if (sorted == 3) {
    // a few time-ordered runs end to end: merged by (time, sequence) tile by tile straight from the
    // table's columns, then -- like any time-ordered feed -- stable passes over the key bits alone
    RunArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n_runs = b.n_runs;
    for (int r = 0; r < b.n_runs; ++r) ra.start[r] = b.run_start[r];
    ra.start[b.n_runs] = s.n;
    const int64_t n_tiles = (s.n + TS_CAP - 1) / TS_CAP;
    const int64_t n_coarse = (n_tiles + kCutStride - 1) / kCutStride + 1;
    TMP_LAUNCH(run_cuts_kernel, dim3((unsigned)((n_coarse + 3) / 4)), dim3(kBlock), st, gen, ra, s.n, TS_CAP, n_tiles, 0,
               b.run_cuts);
    TMP_LAUNCH(run_cuts_kernel, dim3((unsigned)((n_tiles + 1 + 3) / 4)), dim3(kBlock), st, gen, ra, s.n, TS_CAP, n_tiles, 1,
               b.run_cuts);
    PayPtrs none;
    memset(&none, 0, sizeof(none));
    {
      ProfScope ps("tile_merge_kernel", st);
      hipLaunchKernelGGL((tile_sort_kernel<TS_THREADS, TS_ITEMS, true>), dim3((unsigned)n_tiles), dim3(TS_THREADS), 0, st,
                         (const uint64_t *)nullptr, (const uint32_t *)nullptr, b.comp[1], b.idx[1], s.n, 64, 0,
                         (const uint32_t *)nullptr, b.run_cuts, b.fallback, none, 0, gen, ra);
    }
    TMP_CHECK_HIP(hipGetLastError());
    if (lay.key_shift() >= hi_bit) {  // no key: the merged order is the result
      res->comp = b.comp[1];
      res->perm = b.idx[1];
      if (n_pay > 0) {  // no pass follows that would bring the payload columns along: fetch them
        PayPtrs pp;
        memset(&pp, 0, sizeof(pp));
        pp.n = n_pay;
        for (int c = 0; c < n_pay; ++c) { pp.src[c] = pay_src[c]; pp.out[c] = pay_out[c]; }
        TMP_LAUNCH(pay_gather_kernel, dim3(grid_for(s.n, kBlock * 4, 8192)), dim3(kBlock), st, s.n, res->perm, pp);
        TMP_CHECK_HIP(hipGetLastError());
        if (carried) *carried = true;
      }
      return TMP_OK;
    }
    return radix_passes(b, s.n, lay.key_shift(), hi_bit, res, st, n_pay, pay_src, pay_out, carried, nullptr, nullptr,
                        nullptr, 1);
  }
  if (sorted == 4) {
    // every key's rows look ordered: stable passes over the key bits alone, as for a time-ordered
    // feed, then a check of the result; the full sort stands by behind the check's flag
    const int key_passes = passes_for(hi_bit - lay.key_shift());
    if (lay.key_bits > 0 && !lay.seq_wide && b.fallback && !b.run_if &&
        passes_for(hi_bit - lay.low_bits) >= key_passes + 1) {
      TMP_CHECK_HIP(hipMemsetAsync(b.fallback, 0, sizeof(int32_t), st));
      OrderResult quick;
      TMP_TRY(radix_passes(b, s.n, lay.key_shift(), hi_bit, &quick, st, n_pay, pay_src, pay_out, carried, &gen));
      TMP_LAUNCH(ordered_check_kernel, dim3(grid_for(s.n, kBlock * 8, 8192)), dim3(kBlock), st, quick.comp, s.n,
                 lay.low_bits, b.fallback);
      b.run_if = b.fallback;
      OrderResult plain;
      int rc = radix_passes(b, s.n, lay.low_bits, hi_bit, &plain, st, n_pay, pay_src, pay_out, nullptr, &gen);
      if (rc == TMP_OK && plain.comp != quick.comp) {
        const int at = quick.comp == b.comp[0] ? 0 : 1;
        TMP_LAUNCH(copy_pairs_if_kernel, dim3(grid_for(s.n, kBlock * 4, 8192)), dim3(kBlock), st, plain.comp, plain.perm,
                   b.comp[at], b.idx[at], s.n, b.run_if);
      }
      b.run_if = nullptr;
      TMP_TRY(rc);
      TMP_CHECK_HIP(hipGetLastError());
      *res = quick;
      return TMP_OK;
    }
    sorted = 0;
  }
  if (lay.seq_wide && s.seq && sorted == 0) {
    // the sequence values first (stable passes over words that hold nothing else), then the sort
    // words -- laid out in the order those passes left, through the row ids -- over their own bits
    const int grid = grid_for(s.n, kBlock * 4, 8192);
    TMP_LAUNCH(seq_words_kernel, dim3(grid), dim3(kBlock), st, s.n, s.seq, s.seq_valid, lay.seq_min, b.comp[0], b.idx[0]);
    OrderResult by_seq;
    TMP_TRY(radix_passes(b, s.n, 0, lay.seq_full_bits, &by_seq, st, 0, nullptr, nullptr, nullptr, nullptr));
    const int at = by_seq.comp == b.comp[0] ? 0 : 1;
    TMP_LAUNCH(make_comp_by_id_kernel, dim3(grid), dim3(kBlock), st, s.n, gen, b.idx[at], b.comp[at]);
    return radix_passes(b, s.n, lo_bit, hi_bit, res, st, n_pay, pay_src, pay_out, carried, nullptr, nullptr, nullptr, at);
  }
  BucketGeom bg;
  if (!bucketed_wanted(s, lay, sorted, b, &bg)) {
    // a time-ordered feed whose key field takes more than one pass: the travelling columns move with
    // the rows through every pass (buffers planned for it) instead of being fetched through the row
    // ids at the end -- streams in place of a gather over the whole column
    if (sorted == 2 && n_pay >= 1 && n_pay <= b.carry_cols && pay_out && passes_for(hi_bit - lo_bit) > 1 &&
        !getenv("TEMPO_NO_CARRY_TIME")) {
      int at = 0;
      TMP_TRY(radix_passes(b, s.n, lo_bit, hi_bit, res, st, n_pay, pay_src, pay_out, nullptr, &gen, b.carry, &at));
      if (carried) *carried = true;
      return TMP_OK;
    }
    b.narrow_words = narrow_wanted && sorted == 2;
    const int rc = radix_passes(b, s.n, lo_bit, hi_bit, res, st, n_pay, pay_src, pay_out, carried, &gen);
    b.narrow_words = false;
    return rc;
  }

  // ---- bucket map: sampled rows per key -> buckets per key -> first bucket per key -> key per bucket
  const uint32_t nk = (uint32_t)plan_keys(s.key, s.n_keys);
  TMP_CHECK_HIP(hipMemsetAsync(b.key_rows, 0, sizeof(unsigned long long) * nk, st));
  TMP_CHECK_HIP(hipMemsetAsync(b.fallback, 0, sizeof(int32_t), st));
  const int64_t n_samples = (s.n + kCountStride - 1) / kCountStride;
  TMP_LAUNCH(key_sample_kernel, dim3(grid_for(n_samples, kBlock * 16, 1024)), dim3(kBlock), st, s.key, n_samples,
             nk, b.key_rows);
  TMP_LAUNCH(bucket_nb_kernel, dim3((nk + kBlock - 1) / kBlock), dim3(kBlock), st, b.key_rows, nk, (uint64_t)bg.avg,
             b.key_nb, b.key_base);
  TMP_TRY(scan_u32_inplace(b.key_base, (int)nk, st));
  TMP_LAUNCH(bucket_key_kernel, dim3((unsigned)((bg.max_buckets + kBlock - 1) / kBlock)), dim3(kBlock), st, b.key_base,
             b.key_nb, nk, (uint32_t)bg.max_buckets, b.bucket_key, reinterpret_cast<uint2 *>(b.key_rows));
  GenArgs bgen = gen;
  bgen.key_pack = reinterpret_cast<const uint2 *>(b.key_rows);  // the sampled counts are spent by now
  bgen.last_key = nk - 1;
  const uint64_t rel_count = ((uint64_t)lay.prim_max - (uint64_t)lay.prim_min) / (uint64_t)lay.prim_gran + 1;  // R
  bgen.inv = rel_count <= 1 ? 0ull : (~0ull / rel_count) + 1;                                                // ceil(2^64 / R)
  // ---- partition passes over the bucket field, then the tiles
  OrderResult part;
  TMP_REQUIRE(n_pay <= kMaxPay, TMP_EUNSUPPORTED, "ordering: at most %d payload column(s) per call", kMaxPay);
  // a payload column either travels with the rows through every partition pass (buffers planned
  // for it: all its reads and writes stream) or is fetched through the row ids at the very end
  bool carry = n_pay >= 1 && n_pay <= b.carry_cols;
  int carry_at = 0;
  TMP_TRY(radix_passes(b, s.n, lay.key_shift(), lay.key_shift() + bg.bits, &part, st, carry ? n_pay : 0,
                       carry ? pay_src : nullptr, nullptr, nullptr, &bgen, carry ? b.carry : nullptr, &carry_at));
  const int at = part.comp == b.comp[0] ? 0 : 1;
  PayPtrs pp;
  memset(&pp, 0, sizeof(pp));
  pp.n = n_pay;
  for (int c = 0; c < n_pay; ++c) { pp.src[c] = pay_src[c]; pp.out[c] = pay_out[c]; }
  if (carry) {
    pp.positional = 1;
    for (int c = 0; c < n_pay; ++c) pp.src[c] = b.carry[2 * c + carry_at];
  }
  const int64_t n_tiles = (s.n + TS_TILE - 1) / TS_TILE;
  TMP_LAUNCH(tile_bounds_kernel, dim3((unsigned)((n_tiles + 1 + 3) / 4)), dim3(kBlock), st, b.comp[at], s.n,
             lay.key_shift(), TS_TILE, TS_CAP, n_tiles, b.tile_bounds);
  {
    ProfScope ps("tile_sort_kernel", st);
    RunArgs no_runs;
    memset(&no_runs, 0, sizeof(no_runs));
    hipLaunchKernelGGL((tile_sort_kernel<TS_THREADS, TS_ITEMS, false>), dim3((unsigned)n_tiles), dim3(TS_THREADS), 0, st,
                       b.comp[at], b.idx[at], b.comp[at ^ 1], b.idx[at ^ 1], s.n, lay.key_shift(), lay.low_bits,
                       b.bucket_key, b.tile_bounds, b.fallback, pp, getenv("TEMPO_TILE_RADIX") ? 1 : 0, gen, no_runs);
  }
  TMP_CHECK_HIP(hipGetLastError());
  // ---- stand-by: the plain passes, every kernel of which returns at once unless a tile gave up
  b.run_if = b.fallback;
  OrderResult plain;
  int rc = radix_passes(b, s.n, lo_bit, hi_bit, &plain, st, n_pay, pay_src, pay_out, nullptr, &gen);
  if (rc == TMP_OK && plain.comp != b.comp[at ^ 1]) {
    TMP_LAUNCH(copy_pairs_if_kernel, dim3(grid_for(s.n, kBlock * 4, 8192)), dim3(kBlock), st, plain.comp, plain.perm,
               b.comp[at ^ 1], b.idx[at ^ 1], s.n, b.run_if);
  }
  b.run_if = nullptr;
  TMP_TRY(rc);
  TMP_CHECK_HIP(hipGetLastError());
  res->comp = b.comp[at ^ 1];
  res->perm = b.idx[at ^ 1];
  if (carried) *carried = n_pay > 0;
  return TMP_OK;
}

int order_gather_masks(const OrderSpec &s, const OrderResult &res, uint16_t *out, hipStream_t st) {
  if (s.n <= 0) return TMP_OK;
  MaskPtrs mp;
  for (int m = 0; m < TMP_MAX_MASKS; ++m) mp.p[m] = m < s.n_masks ? s.masks[m] : nullptr;
  TMP_LAUNCH(gather_masks_kernel, dim3(grid_for(s.n, kBlock * 4, 8192)), dim3(kBlock), st, s.n, res.perm, s.n_masks, mp,
             out);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

// out[j] = bit `slot` of the sort word at sorted position j.
__global__ __launch_bounds__(kBlock) void mask_bytes_kernel(const uint64_t *comp, int64_t n, int slot, uint8_t *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
    out[j] = (uint8_t)((comp[j] >> slot) & 1ull);
}

int order_columns(const OrderSpec &s0, OrderBuffers &b, int n_cols, const uint64_t *const *cols,
                  const uint8_t *const *valid, uint64_t *const *xs, uint8_t *const *vs, Layout *lay,
                  OrderResult *res, bool *traveled, hipStream_t st, const RangeInfo *known) {
  *traveled = false;
  if (n_cols >= 1 && n_cols <= kMaxPay && xs && xs[0] && s0.n_masks + n_cols <= TMP_MAX_MASKS) {
    OrderSpec s = s0;
    int slot_of[kMaxPay];
    for (int k = 0; k < n_cols; ++k) {
      slot_of[k] = -1;
      if (valid[k]) { slot_of[k] = s.n_masks; s.masks[s.n_masks++] = valid[k]; }
    }
    RangeInfo r;
    if (known) {
      r = *known;
    } else {
      TMP_TRY(order_scan_ranges(s, b, st));
      TMP_TRY(order_read_ranges(b, &r, st));
    }
    if (order_make_layout(s, r, s.n_masks, lay) == TMP_OK) {
      bool carried = false;
      TMP_TRY(order_build(s, *lay, sorted_mode(r), b, res, st, n_cols, cols, xs, &carried));
      if (carried) {
        for (int k = 0; k < n_cols; ++k)
          if (slot_of[k] >= 0)
            TMP_LAUNCH(mask_bytes_kernel, dim3(grid_for(s.n, kBlock * 8, 8192)), dim3(kBlock), st, res->comp, s.n,
                       slot_of[k], vs[k]);
        TMP_CHECK_HIP(hipGetLastError());
        *traveled = true;
      }
      return TMP_OK;
    }
    clear_error();  // the masks do not fit next to key and time: order without them
  }
  return order_single(s0, b, lay, res, st, known);
}

int order_single(const OrderSpec &s, OrderBuffers &b, Layout *lay, OrderResult *res,
                 hipStream_t st, const RangeInfo *known) {
  RangeInfo r;
  if (known) {
    r = *known;
  } else {
    TMP_TRY(order_scan_ranges(s, b, st));
    TMP_TRY(order_read_ranges(b, &r, st));
  }
  TMP_TRY(order_make_layout(s, r, s.n_masks, lay));
  return order_build(s, *lay, sorted_mode(r), b, res, st);
}

}  // namespace tmp

using namespace tmp;

extern "C" int tmp_profile_enable(int on) {
  clear_error();
  if (on && !g_prof_ev) {
    g_prof_ev = new hipEvent_t[2 * kProfSlots];
    for (int i = 0; i < 2 * kProfSlots; ++i) TMP_CHECK_HIP(hipEventCreate(&g_prof_ev[i]));
  }
  g_prof_on = on != 0;
  return TMP_OK;
}
static int prof_recorded() {
  const int n = g_prof_n.load(std::memory_order_relaxed);
  return n < kProfSlots ? n : kProfSlots;
}
extern "C" int tmp_profile_reset(void) { g_prof_n = 0; return TMP_OK; }
extern "C" int tmp_profile_count(void) { return prof_recorded(); }
extern "C" int tmp_profile_dropped(void) { return g_prof_n.load(std::memory_order_relaxed) - prof_recorded(); }
extern "C" int tmp_profile_set_tag(int tag) { g_prof_cur_tag = tag; return TMP_OK; }
extern "C" int tmp_profile_tag(int i) { return (i >= 0 && i < prof_recorded()) ? g_prof_tag[i] : -1; }
extern "C" int tmp_profile_get(int i, const char **name, float *ms) {
  clear_error();
  TMP_REQUIRE(i >= 0 && i < prof_recorded() && name && ms, TMP_EINVAL, "tmp_profile_get: bad index");
  *name = g_prof_name[i];
  TMP_CHECK_HIP(hipEventElapsedTime(ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
  return TMP_OK;
}

extern "C" int tmp_abi_version(void) { return TMP_ABI_VERSION; }
extern "C" const char *tmp_last_error(void) { return g_err; }

__global__ void copy_perm_kernel(const uint32_t *in, int32_t *out, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (int32_t)in[i];
}

static int sort_spec(const tmp_sort_args *a, OrderSpec *s) {
  TMP_REQUIRE(a && a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_sort_order: bad row count");
  TMP_REQUIRE(a->n == 0 || (a->ts && a->out_perm), TMP_EINVAL, "tmp_sort_order: null pointer");
  TMP_REQUIRE(a->ts_unit == 0 || a->ts_unit == 1, TMP_EINVAL, "tmp_sort_order: bad ts_unit");
  memset(s, 0, sizeof(*s));
  s->n = a->n;
  s->key = a->key;
  s->n_keys = a->n_keys;
  s->ts = a->ts;
  s->prim_div = a->ts_unit == 1 ? 1000000 : 1;
  s->seq = a->seq;
  s->seq_valid = a->seq_valid;
  return TMP_OK;
}

extern "C" size_t tmp_sort_order_workspace_bytes(const tmp_sort_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  OrderBuffers b;
  order_plan(a->n, ws, &b, plan_keys(a->key, a->n_keys));
  return ws.off;
}

extern "C" int tmp_sort_order(const tmp_sort_args *a, void *workspace, size_t workspace_bytes,
                              void *stream) {
  clear_error();
  OrderSpec s;
  TMP_TRY(sort_spec(a, &s));
  if (a->n == 0) return TMP_OK;
  Workspace ws(workspace, workspace_bytes);
  OrderBuffers b;
  order_plan(a->n, ws, &b, plan_keys(a->key, a->n_keys));
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_sort_order: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Layout lay;
  OrderResult res;
  TMP_TRY(order_single(s, b, &lay, &res, st));
  TMP_LAUNCH(copy_perm_kernel, dim3(grid_for(a->n, kBlock * 4, 8192)), dim3(kBlock), st,
                     res.perm, a->out_perm, a->n);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
