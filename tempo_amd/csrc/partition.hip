// Device side of the key shuffle: bucket rows by owner = mix(key) % n_parts so that a single
// all-to-all sends every series to the GPU that owns it.  Replaces Spark's
// Exchange hashpartitioning(partitionCols) (python/tempo/tsdf.py:121, :571; resample.py:63).
// The bucketing is one stable radix pass of the ordering step with the owner as the digit.
#include <string.h>

#include <mutex>

#include "common.h"
#include "skew.h"

namespace tmp {

__host__ __device__ __forceinline__ uint32_t mix_key(uint32_t k) {
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  k *= 0xc2b2ae35u;
  k ^= k >> 16;
  return k;
}

// Time-sliced ownership of hot keys (see tmp_partition_args in tempo_hip.h).
struct SliceArgs {
  const int64_t *ts;
  SkewParams sp;
  int64_t bracket_min;
  int64_t bracket_count;
  uint32_t n_split;
  const int32_t *split_slices;
  const int32_t *split_first;
  const int32_t *slice_owner;
  int halo;
};

__device__ __forceinline__ uint32_t clamp_owner(int32_t t, uint32_t n_parts) {
  return t < 0 ? 0u : ((uint32_t)t < n_parts ? (uint32_t)t : n_parts - 1);
}

__device__ __forceinline__ uint32_t slice_owner_of(const SliceArgs &sl, uint32_t slot, int64_t q,
                                                    uint32_t n_parts) {
  const int64_t S = sl.split_slices[slot];
  int64_t slice = (q - sl.bracket_min) * S / sl.bracket_count;
  slice = slice < 0 ? 0 : (slice >= S ? S - 1 : slice);
  return clamp_owner(sl.slice_owner[sl.split_first[slot] + slice], n_parts);
}

// Part of one row; `counted` = the row goes anywhere at all.  In halo mode the result is the part
// that needs a COPY of the row, or n_parts ("none", not counted).
// Balanced ownership: a small table, read by every row (stays in cache).  The read is
// unconditional (keys outside the table read entry 0 and get 0), so the lookups of several rows can
// be in flight together.
__device__ __forceinline__ int32_t route_entry(uint32_t k, const int32_t *__restrict__ table, uint32_t n_table) {
  const int32_t t = table[k < n_table ? k : 0u];
  return k < n_table ? t : 0;
}

// The part a table entry stands for (see route_row).
__device__ __forceinline__ uint32_t route_from_entry(int32_t t, int64_t i, uint32_t n_parts, const SliceArgs &sl,
                                                     bool &counted) {
  counted = true;
  const bool sliced = t < 0 && sl.ts && (uint32_t)(-(t + 1)) < sl.n_split;
  if (!sliced) {
    if (sl.halo) { counted = false; return n_parts; }
    return clamp_owner(t, n_parts);
  }
  const uint32_t slot = (uint32_t)(-(t + 1));
  const int64_t ts = sl.ts[i];
  const int32_t q = bracket_index(ts, sl.sp);
  const uint32_t o = slice_owner_of(sl, slot, q, n_parts);
  if (!sl.halo) return o;
  // the carried-over tail of bracket q is read by bracket q + 1
  const double d = bracket_seconds(ts, sl.sp);
  const double part = bracket_part(d, sl.sp);
  const bool tail = (d - part) / sl.sp.v >= sl.sp.one_minus_fraction;
  const uint32_t o2 = slice_owner_of(sl, slot, (int64_t)q + 1, n_parts);
  counted = tail && o2 != o;
  return counted ? o2 : n_parts;
}

__device__ __forceinline__ uint32_t route_row(uint32_t k, int64_t i, uint32_t n_parts,
                                              const int32_t *__restrict__ table, uint32_t n_table,
                                              const SliceArgs &sl, bool &counted) {
  if (!table) { counted = true; return mix_key(k) % n_parts; }
  return route_from_entry(route_entry(k, table, n_table), i, n_parts, sl, counted);
}

// Owner of every row (the digit of the bucketing pass) and the rows per owner.
__global__ __launch_bounds__(kBlock) void owner_kernel(const int32_t *key, int64_t n, uint32_t n_parts,
                                                       const int32_t *__restrict__ table, uint32_t n_table,
                                                       SliceArgs sl, uint64_t *comp, uint32_t *idx,
                                                       unsigned long long *counts) {
  __shared__ unsigned int hist[256];
  for (int i = threadIdx.x; i < 256; i += kBlock) hist[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool few = n_parts <= 64;  // lane p of every wave counts the rows of part p in a register
  unsigned int mine = 0;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t start = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // whole waves iterate together (the ballots below need every lane present)
  for (int64_t base = start - lane; base < n; base += stride) {
    const int64_t i = base + lane;
    const bool in = i < n;
    uint32_t o = 0;
    bool counted = false;
    if (in) {
      o = route_row((uint32_t)key[i], i, n_parts, table, n_table, sl, counted);
      comp[i] = o;
      idx[i] = (uint32_t)i;
    }
    if (few) {
      for (uint32_t p = 0; p < n_parts; ++p) {
        uint64_t b = __ballot(counted && o == p);
        if ((uint32_t)lane == p) mine += (unsigned int)__popcll(b);
      }
    } else if (counted) {
      atomicAdd(&hist[o], 1u);
    }
  }
  if (few && mine) atomicAdd(&hist[lane], mine);
  __syncthreads();
  for (int i = threadIdx.x; i < (int)n_parts; i += kBlock)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

// ---- route and move in one go (at most 64 parts) ------------------------------------------
// Every wave owns a contiguous range of rows.  Pass 1 notes each row's part in a byte and counts
// the wave's rows per part; an exclusive scan over [part][wave] turns the counts into where each
// wave's rows of each part start in the bucketed order; pass 2 walks the same ranges and copies
// every column's value to its place (grouped by part in a strip of LDS per wave first) -- no row
// ids, no sort passes, no barriers.
// Rows per lane and step, most waves, waves per SIMD.  Measured on 5x10^8 rows of (key, ts, fp64)
// with 10^4 random keys hashed to 8 parts, count + move (tools/bench_route.py): 8.4 ms with 8192
// waves, keys and table entries waited for row group by row group and every column stored
// straight from the lanes that read it; 7.8 ms as it stands (16384 waves, a step's keys requested
// a step ahead, rows grouped by part in LDS before they are stored).  One part: 5.4 -> 4.8 ms.
#ifndef TEMPO_RM_U
#define TEMPO_RM_U 4
#endif
#ifndef TEMPO_RM_WAVES
#define TEMPO_RM_WAVES 16384
#endif
#ifndef TEMPO_RM_OCC
#define TEMPO_RM_OCC 8
#endif
constexpr int kMoveWavesPerBlock = kBlock / 64;
constexpr uint8_t kNoPart = 255;

struct MoveGeom {
  int grid;
  int waves;
  int64_t rows_per_wave;  // multiple of 64
};

static MoveGeom move_geom(int64_t n) {
  MoveGeom g;
  int64_t want = (n + 64 * 64 - 1) / (64 * 64);  // at least 64 steps per wave when there is work
  g.waves = (int)(want < 1 ? 1 : (want > TEMPO_RM_WAVES ? TEMPO_RM_WAVES : want));
  g.waves = (g.waves + kMoveWavesPerBlock - 1) / kMoveWavesPerBlock * kMoveWavesPerBlock;
  g.grid = g.waves / kMoveWavesPerBlock;
  int64_t per = (n + g.waves - 1) / g.waves;
  g.rows_per_wave = (per + 63) / 64 * 64;
  if (g.rows_per_wave < 64) g.rows_per_wave = 64;
  return g;
}

__global__ __launch_bounds__(kBlock, 8) void route_count_kernel(const int32_t *key, int64_t n, uint32_t n_parts,
                                                             const int32_t *__restrict__ table,
                                                             uint32_t n_table, SliceArgs sl,
                                                             int64_t rows_per_wave, int n_waves,
                                                             uint8_t *__restrict__ part_of,
                                                             uint32_t *__restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kMoveWavesPerBlock + (threadIdx.x >> 6);
  const int64_t lo = (int64_t)wave * rows_per_wave;
  const int64_t hi = lo + rows_per_wave < n ? lo + rows_per_wave : n;
  uint32_t mine = 0;  // lane p: rows of part p in this wave's range
  // Eight rows per lane and step.  A step's keys are requested while the step before is being
  // worked on and its table lookups are issued together: a wave otherwise waits out one memory
  // round trip per row group for the key and another for the table entry.
  constexpr int U = 8;
  uint32_t k[U];
#pragma unroll
  for (int u = 0; u < U; ++u) k[u] = 0;
  if (lo < hi) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = lo + u * 64 + lane;
      k[u] = (uint32_t)key[i < hi ? i : hi - 1];
    }
  }
  for (int64_t base = lo; base < hi; base += 64 * U) {
    uint32_t kn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kn[u] = 0;
    const int64_t nb = base + 64 * U;
    if (nb < hi) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = nb + u * 64 + lane;
        kn[u] = (uint32_t)key[i < hi ? i : hi - 1];
      }
    }
    int32_t t[U];
    if (table) {
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = route_entry(k[u], table, n_table);
    }
    uint32_t o[U];
    bool counted[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 64 + lane;
      const bool in = i < hi;
      bool c = true;
      o[u] = table ? route_from_entry(t[u], in ? i : hi - 1, n_parts, sl, c) : mix_key(k[u]) % n_parts;
      counted[u] = c && in;
      if (!counted[u]) o[u] = n_parts;
      if (in) part_of[i] = counted[u] ? (uint8_t)o[u] : kNoPart;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      for (uint32_t p = 0; p < n_parts; ++p) {
        const uint64_t b = __ballot(o[u] == p);
        if ((uint32_t)lane == p) mine += (uint32_t)__popcll(b);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = kn[u];
  }
  if ((uint32_t)lane < n_parts) counts[(size_t)lane * n_waves + wave] = mine;
}

// After the scan over n_parts * n_waves + 1 entries (the last one starts at zero, so it ends up
// holding the grand total): rows per part.
__global__ void part_totals_kernel(const uint32_t *offsets, int n_waves, uint32_t n_parts,
                                   unsigned long long *out) {
  const uint32_t p = threadIdx.x;
  if (p < n_parts) out[p] = offsets[(size_t)(p + 1) * n_waves] - offsets[(size_t)p * n_waves];
}

constexpr int kMoveCols = 12;
struct MoveCols {
  int n;
  const void *src[kMoveCols];
  void *dst[kMoveCols];
  int bytes[kMoveCols];
};

__global__ __launch_bounds__(kBlock, TEMPO_RM_OCC) void route_move_kernel(int64_t n, uint32_t n_parts, int64_t rows_per_wave,
                                                            int n_waves, const uint8_t *__restrict__ part_of,
                                                            const uint32_t *__restrict__ offsets,
                                                            MoveCols mc) {
  // Four rows per lane and step.  A step's rows leave grouped by part: every wave has a strip of
  // LDS of its own in which the step's values are put in (part, row) order, so that neighbouring
  // lanes store to neighbouring places -- runs of rows per part instead of a few rows per part
  // and store instruction.  LDS keeps one wave's accesses in order, so no barrier is involved.
  constexpr int U = TEMPO_RM_U;
  constexpr int kStep = 64 * U;
  __shared__ uint64_t val_s[kMoveWavesPerBlock][kStep];
  __shared__ uint32_t to_s[kMoveWavesPerBlock][kStep];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int wave = blockIdx.x * kMoveWavesPerBlock + wib;
  uint64_t *val = val_s[wib];
  uint32_t *to = to_s[wib];
  const int64_t lo = (int64_t)wave * rows_per_wave;
  const int64_t hi = lo + rows_per_wave < n ? lo + rows_per_wave : n;
  // lane p: where this wave's next row of part p goes
  uint32_t run = (uint32_t)lane < n_parts ? offsets[(size_t)lane * n_waves + wave] : 0u;
  const uint64_t below = (1ull << lane) - 1;
  uint32_t o[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = lo + u * 64 + lane;
    o[u] = i < hi ? part_of[i] : kNoPart;
  }
  // one column's values of this step's rows (rows that go nowhere are not read)
  auto fetch = [&](int c, int64_t base, uint64_t(&v)[U]) {
    const void *src = mc.src[c];
    const int bytes = mc.bytes[c];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = 0;
      if (o[u] == kNoPart) continue;
      const int64_t i = base + u * 64 + lane;
      if (bytes == 8) v[u] = ((const uint64_t *)src)[i];
      else if (bytes == 4) v[u] = ((const uint32_t *)src)[i];
      else v[u] = ((const uint8_t *)src)[i];
    }
  };
  for (int64_t base = lo; base < hi; base += kStep) {
    uint32_t on[U], pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // the next step's parts, requested now
      const int64_t i = base + kStep + u * 64 + lane;
      on[u] = i < hi ? part_of[i] : kNoPart;
    }
    uint64_t v[U];
    if (mc.n > 0) fetch(0, base, v);
    // place of every row: in the bucketed order (through the strip) and inside the strip
    uint32_t moved = 0;  // rows of this step placed so far (the same in every lane)
#pragma unroll
    for (int u = 0; u < U; ++u) pos[u] = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
      uint32_t start = __shfl(run, (int)p);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t b = __ballot(o[u] == p);
        const uint32_t before = (uint32_t)__popcll(b & below);
        if (o[u] == p) {
          pos[u] = moved + before;
          to[moved + before] = start + before;
        }
        const uint32_t c = (uint32_t)__popcll(b);
        start += c;
        moved += c;
      }
      if ((uint32_t)lane == p) run = start;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t dest[U];
#pragma unroll
    for (int u = 0; u < U; ++u) dest[u] = (uint32_t)(u * 64 + lane) < moved ? to[u * 64 + lane] : 0u;
    for (int c = 0; c < mc.n; ++c) {
      uint64_t vn[U];
      if (c + 1 < mc.n) fetch(c + 1, base, vn);  // the next column is on its way while this one leaves
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (o[u] != kNoPart) val[pos[u]] = v[u];
      __builtin_amdgcn_wave_barrier();
      void *dst = mc.dst[c];
      const int bytes = mc.bytes[c];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((uint32_t)(u * 64 + lane) >= moved) continue;
        const uint64_t w = val[u * 64 + lane];
        if (bytes == 8) ((uint64_t *)dst)[dest[u]] = w;
        else if (bytes == 4) ((uint32_t *)dst)[dest[u]] = (uint32_t)w;
        else ((uint8_t *)dst)[dest[u]] = (uint8_t)w;
      }
      if (c + 1 < mc.n) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = vn[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) o[u] = on[u];
  }
}

// Rows per key over every stride-th row.  Small dictionaries are counted in LDS per block and
// flushed once; a counter is touched through an atomic only by the lanes that hold the key, and
// the flush skips empty counters.
constexpr int kCountLds = 12288;

__global__ __launch_bounds__(kBlock) void key_counts_kernel(const int32_t *__restrict__ key, int64_t n_samples,
                                                            int64_t stride, uint32_t n_keys,
                                                            unsigned long long *counts) {
  __shared__ unsigned int hist[kCountLds];
  const bool local = n_keys <= (uint32_t)kCountLds;
  if (local) {
    for (uint32_t i = threadIdx.x; i < n_keys; i += kBlock) hist[i] = 0;
    __syncthreads();
  }
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_samples; s += step) {
    const uint32_t k = (uint32_t)key[s * stride];
    if (k >= n_keys) continue;
    if (local) atomicAdd(&hist[k], 1u);
    else atomicAdd(&counts[k], 1ull);
  }
  if (local) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_keys; i += kBlock)
      if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
  }
}

__global__ void copy_u32_i32_kernel(const uint32_t *in, int32_t *out, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (int32_t)in[i];
}

}  // namespace tmp

using namespace tmp;

extern "C" int32_t tmp_key_owner(int32_t key, int32_t n_parts) {
  return n_parts > 0 ? (int32_t)(mix_key((uint32_t)key) % (uint32_t)n_parts) : 0;
}

static int slice_args(const tmp_partition_args *a, SliceArgs *out) {
  SliceArgs sl;
  memset(&sl, 0, sizeof(sl));
  if (a->owner_table && a->ts) {
    TMP_REQUIRE(a->n_split >= 0 && (a->n_split == 0 || (a->split_slices && a->split_first && a->slice_owner)),
                TMP_EINVAL, "partition: sliced keys need split_slices, split_first and slice_owner");
    TMP_REQUIRE(a->ts_partition_val > 0 && a->bracket_count >= 1, TMP_EINVAL,
                "partition: sliced keys need ts_partition_val > 0 and bracket_count >= 1");
    TMP_REQUIRE(!a->halo || a->n_parts <= 255, TMP_EINVAL, "partition: halo mode takes at most 255 parts");
    sl.ts = a->ts;
    skew_params_set(&sl.sp, 1, a->v_is_int, a->ts_is_seconds, a->ts_partition_val, a->one_minus_fraction);
    sl.bracket_min = a->bracket_min;
    sl.bracket_count = a->bracket_count;
    sl.n_split = (uint32_t)a->n_split;
    sl.split_slices = a->split_slices;
    sl.split_first = a->split_first;
    sl.slice_owner = a->slice_owner;
    sl.halo = a->halo != 0;
  } else {
    TMP_REQUIRE(!a->halo, TMP_EINVAL, "partition: halo mode needs an owner table and ts");
  }
  *out = sl;
  return TMP_OK;
}

extern "C" size_t tmp_hash_partition_workspace_bytes(const tmp_partition_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  OrderBuffers b;
  order_plan(a->n, ws, &b);
  return ws.off;
}

static int hash_partition_run(const tmp_partition_args *a, void *workspace, size_t workspace_bytes,
                              void *stream, const char *who) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "%s: null args", who);
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "%s: bad row count", who);
  TMP_REQUIRE(a->n_parts >= 1 && a->n_parts <= 256, TMP_EINVAL, "%s: n_parts must be in [1, 256]", who);
  TMP_REQUIRE(a->out_counts, TMP_EINVAL, "%s: out_counts is null", who);
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_counts, 0, sizeof(int64_t) * a->n_parts, st));
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->key && a->out_perm, TMP_EINVAL, "%s: null pointer", who);
  Workspace ws(workspace, workspace_bytes);
  OrderBuffers b;
  order_plan(a->n, ws, &b);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "%s: workspace too small", who);
  TMP_REQUIRE(!a->owner_table || a->n_keys >= 1, TMP_EINVAL, "%s: owner_table needs n_keys >= 1", who);
  SliceArgs sl;
  TMP_TRY(slice_args(a, &sl));
  TMP_LAUNCH(owner_kernel, dim3(grid_for(a->n, kBlock * 8, 2048)), dim3(kBlock), st, a->key, a->n,
             (uint32_t)a->n_parts, a->owner_table, (uint32_t)(a->owner_table ? a->n_keys : 0), sl,
             b.comp[0], b.idx[0], (unsigned long long *)a->out_counts);
  OrderResult res;
  // halo mode buckets into n_parts + 1 digits, the last one being "no copy"
  TMP_TRY(order_radix_passes(b, a->n, 0, bits_for((uint64_t)a->n_parts - (sl.halo ? 0 : 1)), &res, st));
  TMP_LAUNCH(copy_u32_i32_kernel, dim3(grid_for(a->n, kBlock * 4, 8192)), dim3(kBlock), st, res.perm,
             a->out_perm, a->n);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

extern "C" int tmp_hash_partition(const tmp_partition_args *a, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  return hash_partition_run(a, workspace, workspace_bytes, stream, "tmp_hash_partition");
}

// ---- tmp_hash_partition without a workspace argument: the radix ping-pong buffers and digit
// counters live in a device array of this code object.  Every GPU the library runs on holds its
// own copy of the array at its own address (until the process ends); host code only ever uses
// the per-device address hipGetSymbolAddress returns, never the array's name.
constexpr int64_t kStaticMaxRows = TMP_STATIC_PARTITION_MAX_ROWS;
// what the ordering step takes for kStaticMaxRows rows (two 8-byte and two 4-byte arrays per
// row) plus 4 MiB for its digit counters and small records; the entry point checks the exact
// need of every call against this
constexpr size_t kStaticScratchBytes = (size_t)kStaticMaxRows * 24 + ((size_t)4 << 20);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[kMaxDevices];  // [device ordinal], filled on the first call per device

extern "C" int tmp_hash_partition_static(const tmp_partition_args *a, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_hash_partition_static: null args");
  // the argument errors tmp_hash_partition reports come first, under the same code
  TMP_REQUIRE(a->n >= 0, TMP_EINVAL, "tmp_hash_partition_static: row count must be in [0, 2^31)");
  TMP_REQUIRE(a->n == 0 || (a->key && a->out_perm), TMP_EINVAL, "tmp_hash_partition_static: null pointer");
  TMP_REQUIRE(a->n <= kStaticMaxRows, TMP_ECAPACITY,
              "tmp_hash_partition_static: %lld rows exceed the built-in scratch (at most %lld rows); "
              "use tmp_hash_partition with a workspace", (long long)a->n, (long long)kStaticMaxRows);
  const size_t need = tmp_hash_partition_workspace_bytes(a);
  TMP_REQUIRE(need <= kStaticScratchBytes, TMP_ECAPACITY,
              "tmp_hash_partition_static: needs %zu bytes of scratch, %zu are built in", need, kStaticScratchBytes);
  int dev = -1;
  TMP_CHECK_HIP(hipGetDevice(&dev));
  TMP_REQUIRE(dev >= 0 && dev < kMaxDevices, TMP_EUNSUPPORTED, "tmp_hash_partition_static: device ordinal %d", dev);
  void *ptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {  // once per device, with that device current
      void *p = nullptr;
      TMP_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      TMP_REQUIRE(p != nullptr, TMP_EHIP, "tmp_hash_partition_static: no device address for the scratch array");
      g_scratch_ptr[dev] = p;
    }
    ptr = g_scratch_ptr[dev];
  }
  // Nothing in the scratch has to start at zero: the routing kernel writes every word / row id,
  // each pass's histogram kernel writes every counter before the scan reads it.
  return hash_partition_run(a, ptr, kStaticScratchBytes, stream, "tmp_hash_partition_static");
}

extern "C" size_t tmp_key_counts_workspace_bytes(const tmp_key_counts_args *) { return 0; }

extern "C" int tmp_key_counts(const tmp_key_counts_args *a, void *, size_t, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_key_counts: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_key_counts: bad row count");
  TMP_REQUIRE(a->n_keys >= 1 && a->out_counts, TMP_EINVAL, "tmp_key_counts: n_keys >= 1 and out_counts needed");
  TMP_REQUIRE(a->stride >= 1, TMP_EINVAL, "tmp_key_counts: stride must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_counts, 0, sizeof(int64_t) * (size_t)a->n_keys, st));
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->key, TMP_EINVAL, "tmp_key_counts: null key column");
  const int64_t n_samples = (a->n + a->stride - 1) / a->stride;
  // few blocks when the counters live in LDS: each one flushes up to n_keys counters
  TMP_LAUNCH(key_counts_kernel, dim3(grid_for(n_samples, kBlock * 64, 1024)), dim3(kBlock), st, a->key,
             n_samples, (int64_t)a->stride, (uint32_t)a->n_keys, (unsigned long long *)a->out_counts);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

static void move_plan(int64_t n, int n_parts, Workspace &ws, MoveGeom *g, uint8_t **part_of,
                      uint32_t **offsets) {
  *g = move_geom(n);
  *part_of = ws.take<uint8_t>((size_t)(n > 0 ? n : 1));
  *offsets = ws.take<uint32_t>((size_t)n_parts * g->waves + 1);
}

extern "C" size_t tmp_partition_move_workspace_bytes(const tmp_partition_move_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  MoveGeom g;
  uint8_t *part_of;
  uint32_t *offsets;
  move_plan(a->route.n, a->route.n_parts > 0 ? a->route.n_parts : 1, ws, &g, &part_of, &offsets);
  return ws.off;
}

extern "C" int tmp_partition_move(const tmp_partition_move_args *a, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_partition_move: null args");
  const tmp_partition_args *r = &a->route;
  TMP_REQUIRE(r->n >= 0 && r->n < (1ll << 31), TMP_EINVAL, "tmp_partition_move: bad row count");
  TMP_REQUIRE(r->n_parts >= 1 && r->n_parts <= 64, TMP_EUNSUPPORTED,
              "tmp_partition_move: 1 to 64 parts (use tmp_hash_partition + tmp_gather beyond)");
  TMP_REQUIRE(r->out_counts, TMP_EINVAL, "tmp_partition_move: out_counts is null");
  TMP_REQUIRE(a->n_cols >= 0 && a->n_cols <= kMoveCols, TMP_EUNSUPPORTED,
              "tmp_partition_move: at most %d columns per call", kMoveCols);
  TMP_REQUIRE(a->capacity >= 0, TMP_EINVAL, "tmp_partition_move: negative capacity");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(r->out_counts, 0, sizeof(int64_t) * r->n_parts, st));
  if (r->n == 0) return TMP_OK;
  TMP_REQUIRE(r->key, TMP_EINVAL, "tmp_partition_move: null key column");
  TMP_REQUIRE(!r->owner_table || r->n_keys >= 1, TMP_EINVAL, "tmp_partition_move: owner_table needs n_keys >= 1");
  MoveCols mc;
  memset(&mc, 0, sizeof(mc));
  mc.n = a->n_cols;
  for (int c = 0; c < a->n_cols; ++c) {
    TMP_REQUIRE(a->cols && a->out_cols && a->col_bytes && a->cols[c] && a->out_cols[c], TMP_EINVAL,
                "tmp_partition_move: null column %d", c);
    TMP_REQUIRE(a->col_bytes[c] == 1 || a->col_bytes[c] == 4 || a->col_bytes[c] == 8, TMP_EUNSUPPORTED,
                "tmp_partition_move: column %d has %d-byte elements (1, 4 or 8 supported)", c, a->col_bytes[c]);
    mc.src[c] = a->cols[c];
    mc.dst[c] = a->out_cols[c];
    mc.bytes[c] = a->col_bytes[c];
  }
  SliceArgs sl;
  TMP_TRY(slice_args(r, &sl));
  Workspace ws(workspace, workspace_bytes);
  MoveGeom g;
  uint8_t *part_of;
  uint32_t *offsets;
  move_plan(r->n, r->n_parts, ws, &g, &part_of, &offsets);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_partition_move: workspace too small");
  const int n_entries = r->n_parts * g.waves + 1;
  TMP_CHECK_HIP(hipMemsetAsync(offsets + n_entries - 1, 0, sizeof(uint32_t), st));
  TMP_LAUNCH(route_count_kernel, dim3(g.grid), dim3(kBlock), st, r->key, r->n, (uint32_t)r->n_parts,
             r->owner_table, (uint32_t)(r->owner_table ? r->n_keys : 0), sl, g.rows_per_wave, g.waves,
             part_of, offsets);
  TMP_TRY(scan_u32_inplace(offsets, n_entries, st));
  TMP_LAUNCH(part_totals_kernel, dim3(1), dim3(64), st, offsets, g.waves, (uint32_t)r->n_parts,
             (unsigned long long *)r->out_counts);
  // the copies may only run when everything fits: in halo mode the number of rows is not known
  // beforehand, so the total is read back first (one small device->host read)
  uint32_t total = 0;
  TMP_CHECK_HIP(hipMemcpyAsync(&total, offsets + n_entries - 1, sizeof(total), hipMemcpyDeviceToHost, st));
  TMP_CHECK_HIP(hipStreamSynchronize(st));
  TMP_REQUIRE((int64_t)total <= a->capacity, TMP_ECAPACITY,
              "tmp_partition_move: %u rows to move, room for %lld", total, (long long)a->capacity);
  if (total > 0 && a->n_cols > 0)
    TMP_LAUNCH(route_move_kernel, dim3(g.grid), dim3(kBlock), st, r->n, (uint32_t)r->n_parts,
               g.rows_per_wave, g.waves, part_of, offsets, mc);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
