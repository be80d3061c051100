// withLookbackFeatures: trailing-window feature tensor.  Replaces
// collect_list(array(featureCols)) over rowsBetween(-L, -1) and the size filter of
// python/tempo/tsdf.py:637-671.
//
// Rows are ordered as __baseWindow does (tsdf.py:563-572).  Pass 1 finds, per row, how many
// rows of its own series precede it (capped at L) and counts the rows that are kept; pass 2
// turns the per-block counts into output offsets; pass 3 writes the dense [rows][L][F] tensor
// with consecutive lanes on consecutive output elements (the operator is write-bound).
#include <string.h>

#include "common.h"

namespace tmp {

constexpr int LB_ROWS = 1024;
constexpr int LB_THREADS = 256;

// rows of the same series before position p, capped at `lookback`
__device__ __forceinline__ int lookback_len(const uint64_t *comp, int64_t p, int lookback, int ks,
                                            bool has_key) {
  int64_t lo = p - lookback < 0 ? 0 : p - lookback;
  if (!has_key) return (int)(p - lo);
  uint64_t key = comp[p] >> ks;
  int64_t hi = p;  // first position in [lo, p] whose key equals key
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if ((comp[mid] >> ks) < key) lo = mid + 1;
    else hi = mid;
  }
  return (int)(p - lo);
}

__global__ __launch_bounds__(LB_THREADS) void lookback_count_kernel(const uint64_t *comp, int64_t n,
                                                                    Layout lay, int lookback,
                                                                    int exact, uint8_t *len8,
                                                                    int32_t *len32,
                                                                    uint32_t *block_counts) {
  __shared__ uint32_t wsum[LB_THREADS / 64];
  const int ks = lay.key_shift();
  const bool has_key = lay.key_bits > 0;
  int64_t t0 = (int64_t)blockIdx.x * LB_ROWS;
  uint32_t kept = 0;
  for (int r = 0; r < LB_ROWS / LB_THREADS; ++r) {
    int64_t p = t0 + r * LB_THREADS + threadIdx.x;
    if (p < n) {
      int m = lookback_len(comp, p, lookback, ks, has_key);
      if (len32) len32[p] = m;
      if (len8) len8[p] = (uint8_t)(m == lookback);
      kept += (!exact || m == lookback) ? 1u : 0u;
    }
  }
  for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < LB_THREADS / 64; ++w) t += wsum[w];
    block_counts[blockIdx.x] = t;
  }
}

// exclusive scan of block_counts in place (one block), total -> *out_count
__global__ __launch_bounds__(1024) void lookback_offsets_kernel(uint32_t *block_counts, int nblocks,
                                                                int64_t *out_count) {
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
  if (tid == 0) *out_count = (int64_t)carry_s;
}

// Output slot of every kept row: row_pos[o] = sorted position, out_rows[o] = input row.
__global__ __launch_bounds__(LB_THREADS) void lookback_slots_kernel(const uint64_t *comp,
                                                                    const uint32_t *perm, int64_t n,
                                                                    int exact, const uint8_t *full,
                                                                    const uint32_t *block_offsets,
                                                                    uint32_t *row_pos,
                                                                    int32_t *out_rows) {
  __shared__ uint32_t wsum[LB_THREADS / 64];
  int64_t t0 = (int64_t)blockIdx.x * LB_ROWS;
  uint32_t base = block_offsets[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < LB_ROWS / LB_THREADS; ++r) {
    int64_t p = t0 + r * LB_THREADS + threadIdx.x;
    bool keep = p < n && (!exact || full[p]);
    uint64_t bal = __ballot(keep);
    uint32_t before = (uint32_t)__popcll(bal & (lane ? (~0ull >> (64 - lane)) : 0ull));
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wp = 0, tot = 0;
    for (int w = 0; w < LB_THREADS / 64; ++w) {
      if (w < wave) wp += wsum[w];
      tot += wsum[w];
    }
    if (keep) {
      uint32_t o = base + wp + before;
      row_pos[o] = (uint32_t)p;
      out_rows[o] = (int32_t)perm[p];
    }
    base += tot;
    __syncthreads();
  }
}

struct FeatPtrs {
  const double *x[TMP_MAX_FEATURES];
  const uint8_t *v[TMP_MAX_FEATURES];
};

// One thread per output element; element e = (o * L + l) * F + f.
__global__ __launch_bounds__(LB_THREADS) void lookback_write_kernel(
    const uint32_t *row_pos, const uint32_t *perm, const int32_t *len32, const int64_t *count,
    int lookback, int n_feat, int exact, FeatPtrs fp, double *out, uint8_t *out_valid) {
  const int64_t per_row = (int64_t)lookback * n_feat;
  const int64_t total = *count * per_row;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int64_t o = e / per_row;
    int rem = (int)(e - o * per_row);
    int l = rem / n_feat, f = rem - l * n_feat;
    int64_t p = row_pos[o];
    int m = exact ? lookback : len32[p];
    double val = 0.0;
    uint8_t ok = 0;
    if (l < m) {
      uint32_t src = perm[p - m + l];
      ok = fp.v[f] ? fp.v[f][src] : 1;
      val = ok ? fp.x[f][src] : 0.0;
    }
    out[e] = val;
    if (out_valid) out_valid[e] = ok;
  }
}

struct LbPlan {
  OrderBuffers ob;
  uint8_t *full;
  uint32_t *block_counts;
  uint32_t *row_pos;
  int nblocks;
};

static void lb_plan(const tmp_lookback_args *a, Workspace &ws, LbPlan *p) {
  order_plan(a->n, ws, &p->ob);
  size_t cnt = (size_t)(a->n > 0 ? a->n : 1);
  p->nblocks = (int)((cnt + LB_ROWS - 1) / LB_ROWS);
  p->full = ws.take<uint8_t>(cnt);
  p->block_counts = ws.take<uint32_t>((size_t)p->nblocks);
  p->row_pos = ws.take<uint32_t>(cnt);
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_lookback_workspace_bytes(const tmp_lookback_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  LbPlan p;
  lb_plan(a, ws, &p);
  return ws.off;
}

extern "C" int tmp_lookback(const tmp_lookback_args *a, void *workspace, size_t workspace_bytes,
                            void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_lookback: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_lookback: row count must be in [0, 2^31)");
  TMP_REQUIRE(a->n_feat >= 1 && a->n_feat <= TMP_MAX_FEATURES, TMP_EINVAL,
              "tmp_lookback: n_feat must be in [1, %d]", TMP_MAX_FEATURES);
  TMP_REQUIRE(a->lookback >= 1, TMP_EINVAL, "tmp_lookback: lookback must be >= 1");
  TMP_REQUIRE(a->out_count, TMP_EINVAL, "tmp_lookback: out_count is null");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_count, 0, sizeof(int64_t), st));
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->x && a->x_valid && a->out_rows && a->out_feat, TMP_EINVAL,
              "tmp_lookback: null pointer");
  TMP_REQUIRE(a->exact_size || a->out_len, TMP_EINVAL, "tmp_lookback: out_len is required when exact_size == 0");
  Workspace ws(workspace, workspace_bytes);
  LbPlan p;
  lb_plan(a, ws, &p);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_lookback: workspace too small");
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts;
  s.prim_div = a->ts_is_seconds ? 1 : 1000000;
  s.seq = a->seq; s.seq_valid = a->seq_valid;
  Layout lay;
  OrderResult res;
  TMP_TRY(order_single(s, p.ob, &lay, &res, st));
  FeatPtrs fp;
  for (int f = 0; f < TMP_MAX_FEATURES; ++f) {
    fp.x[f] = f < a->n_feat ? a->x[f] : nullptr;
    fp.v[f] = f < a->n_feat ? a->x_valid[f] : nullptr;
    TMP_REQUIRE(f >= a->n_feat || fp.x[f], TMP_EINVAL, "tmp_lookback: feature column %d is null", f);
  }
  int exact = a->exact_size != 0;
  TMP_LAUNCH(lookback_count_kernel, dim3(p.nblocks), dim3(LB_THREADS), st, res.comp, a->n, lay, a->lookback,
             exact, p.full, exact ? nullptr : a->out_len, p.block_counts);
  TMP_LAUNCH(lookback_offsets_kernel, dim3(1), dim3(1024), st, p.block_counts, p.nblocks, a->out_count);
  TMP_LAUNCH(lookback_slots_kernel, dim3(p.nblocks), dim3(LB_THREADS), st, res.comp, res.perm, a->n, exact,
             p.full, p.block_counts, p.row_pos, a->out_rows);
  int64_t max_elems = a->n * (int64_t)a->lookback * a->n_feat;
  TMP_LAUNCH(lookback_write_kernel, dim3(grid_for(max_elems, LB_THREADS * 8, 1 << 16)), dim3(LB_THREADS), st,
             p.row_pos, res.perm, a->out_len, a->out_count, a->lookback, a->n_feat, exact, fp, a->out_feat,
             a->out_feat_valid);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
