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

// One thread per output element; element e = (o * L + l) * F + f.  General form (64-bit index
// arithmetic per element): used when a row's L * F elements are too many for the blocked form.
__global__ __launch_bounds__(LB_THREADS) void lookback_write_wide_kernel(
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
      uint32_t src = perm ? perm[p - m + l] : (uint32_t)(p - m + l);
      ok = fp.v[f] ? fp.v[f][src] : 1;
      val = ok ? fp.x[f][src] : 0.0;
    }
    out[e] = val;
    if (out_valid) out_valid[e] = ok;
  }
}

// Blocked form: a block owns 256 output rows, keeps their positions and history lengths in LDS
// and walks their L * F elements in output order (stores coalesce); the element -> (row, l, f)
// split is 32-bit arithmetic with float reciprocals, exact after one correction step because
// every index stays below 2^22.
constexpr int LBW_ROWS = 256;
constexpr uint32_t LBW_MAX_PER_ROW = 16384;

__device__ __forceinline__ uint32_t div_small(uint32_t q, uint32_t d, float inv, uint32_t *rem) {
  uint32_t o = (uint32_t)((float)q * inv);
  int32_t r = (int32_t)(q - o * d);
  if (r < 0) { --o; r += (int32_t)d; }
  else if (r >= (int32_t)d) { ++o; r -= (int32_t)d; }
  *rem = (uint32_t)r;
  return o;
}

__global__ __launch_bounds__(LB_THREADS) void lookback_write_kernel(
    const uint32_t *row_pos, const uint32_t *perm, const int32_t *len32, const int64_t *count,
    int lookback, int n_feat, int exact, FeatPtrs fp, double *out, uint8_t *out_valid,
    float inv_per_row, float inv_feat, int cap) {
  extern __shared__ __align__(16) unsigned char lb_smem[];
  __shared__ uint32_t pos_s[LBW_ROWS];
  __shared__ int m_s[LBW_ROWS];
  const int64_t n_out = *count;
  const int64_t o0 = (int64_t)blockIdx.x * LBW_ROWS;
  if (o0 >= n_out) return;
  const int rows = (int)(n_out - o0 < LBW_ROWS ? n_out - o0 : LBW_ROWS);
  for (int t = threadIdx.x; t < rows; t += LB_THREADS) {
    const uint32_t p = row_pos[o0 + t];
    pos_s[t] = p;
    m_s[t] = exact ? lookback : len32[p];
  }
  __syncthreads();
  // The histories of the block's rows are one stretch of the ordered table (a history starts no
  // earlier than the one before it): when it fits, its feature values are brought into LDS once,
  // each through one read of the row id, instead of once per output element.
  const uint32_t first = pos_s[0] - (uint32_t)m_s[0];
  const uint32_t span = pos_s[rows - 1] - first;  // slots first .. pos - 1 of the last row
  const bool staged = cap > 0 && span <= (uint32_t)cap;
  double *xv = (double *)lb_smem;                      // [n_feat][cap]
  uint8_t *okv = lb_smem + (size_t)cap * n_feat * 8;   // [n_feat][cap]
  if (staged) {
    for (uint32_t sl = threadIdx.x; sl < span; sl += LB_THREADS) {
      const uint32_t src = perm ? perm[first + sl] : first + sl;
      for (int f = 0; f < n_feat; ++f) {
        const uint8_t ok = fp.v[f] ? fp.v[f][src] : 1;
        okv[(size_t)f * cap + sl] = ok;
        xv[(size_t)f * cap + sl] = ok ? fp.x[f][src] : 0.0;
      }
    }
    __syncthreads();
  }
  const uint32_t per_row = (uint32_t)lookback * (uint32_t)n_feat;
  const uint32_t total = (uint32_t)rows * per_row;
  double *ob = out + o0 * per_row;
  uint8_t *vb = out_valid ? out_valid + o0 * per_row : nullptr;
  // element q of the block = (row o, lag l, feature f); the split is divided out once per thread
  // and then advanced: a step moves q by a fixed amount, i.e. (o, rem) by its quotient and
  // remainder, and q + 1 is the next feature (or lag, or row)
  auto fetch = [&](uint32_t o, uint32_t l, uint32_t f, double &val, uint8_t &ok) {
    const int m = m_s[o];
    val = 0.0;
    ok = 0;
    if ((int)l < m) {
      const uint32_t at = pos_s[o] - (uint32_t)m + l;
      if (staged) {
        ok = okv[(size_t)f * cap + (at - first)];
        val = xv[(size_t)f * cap + (at - first)];
      } else {
        const uint32_t src = perm ? perm[at] : at;
        ok = fp.v[f] ? fp.v[f][src] : 1;
        val = ok ? fp.x[f][src] : 0.0;
      }
    }
  };
  // two elements per thread and step, stored as one 16-byte word where the output allows it
  const bool wide = (((uintptr_t)ob) & 15) == 0;
  constexpr uint32_t kStride = LB_THREADS * 2;
  uint32_t dummy;
  const uint32_t step_o = div_small(kStride, per_row, inv_per_row, &dummy), step_rem = dummy;
  uint32_t rem;
  uint32_t o = div_small(threadIdx.x * 2, per_row, inv_per_row, &rem);
  for (uint32_t q = threadIdx.x * 2; q < total; q += kStride) {
    uint32_t f;
    const uint32_t l = div_small(rem, (uint32_t)n_feat, inv_feat, &f);
    double v0, v1 = 0.0;
    uint8_t k0, k1 = 0;
    fetch(o, l, f, v0, k0);
    const bool two = q + 1 < total;
    if (two) {
      uint32_t o1 = o, l1 = l, f1 = f + 1;
      if (f1 == (uint32_t)n_feat) { f1 = 0; ++l1; }
      if (l1 == (uint32_t)lookback) { l1 = 0; ++o1; }
      fetch(o1, l1, f1, v1, k1);
    }
    o += step_o;
    rem += step_rem;
    if (rem >= per_row) { rem -= per_row; ++o; }
    if (two && wide) {
      *reinterpret_cast<double2 *>(ob + q) = make_double2(v0, v1);
    } else {
      ob[q] = v0;
      if (two) ob[q + 1] = v1;
    }
    if (vb) {
      vb[q] = k0;
      if (two) vb[q + 1] = k1;
    }
  }
}

struct LbPlan {
  OrderBuffers ob;
  uint8_t *full;
  uint32_t *block_counts;
  uint32_t *row_pos;
  int nblocks;
  // up to kMaxPay feature columns come out of the ordering step in sorted order (values and
  // validity bytes): the write kernels then read a history as one contiguous stretch
  uint64_t *xs[kMaxPay];
  uint8_t *vs[kMaxPay];
};

static void lb_plan(const tmp_lookback_args *a, Workspace &ws, LbPlan *p) {
  const int n_travel = (a->n_feat >= 1 && a->n_feat <= kMaxPay) ? a->n_feat : 0;
  order_plan(a->n, ws, &p->ob, plan_keys(a->key, a->n_keys), n_travel);
  size_t cnt = (size_t)(a->n > 0 ? a->n : 1);
  for (int c = 0; c < kMaxPay; ++c) {
    p->xs[c] = c < n_travel ? ws.take<uint64_t>(cnt) : nullptr;
    p->vs[c] = c < n_travel ? ws.take<uint8_t>(cnt) : nullptr;
  }
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
  FeatPtrs fp;
  for (int f = 0; f < TMP_MAX_FEATURES; ++f) {
    fp.x[f] = f < a->n_feat ? a->x[f] : nullptr;
    fp.v[f] = f < a->n_feat ? a->x_valid[f] : nullptr;
    TMP_REQUIRE(f >= a->n_feat || fp.x[f], TMP_EINVAL, "tmp_lookback: feature column %d is null", f);
  }
  const bool fits = a->n_feat <= kMaxPay;
  const uint64_t *src[kMaxPay];
  const uint8_t *val[kMaxPay];
  for (int f = 0; f < kMaxPay; ++f) {
    src[f] = (fits && f < a->n_feat) ? (const uint64_t *)fp.x[f] : nullptr;
    val[f] = (fits && f < a->n_feat) ? fp.v[f] : nullptr;
  }
  bool traveled = false;
  TMP_TRY(order_columns(s, p.ob, fits ? a->n_feat : 0, src, val, p.xs, p.vs, &lay, &res, &traveled, st));
  const uint32_t *feat_perm = res.perm;  // how the write kernels find a history's values
  if (traveled) {
    feat_perm = nullptr;
    for (int f = 0; f < a->n_feat; ++f) {
      fp.x[f] = (const double *)p.xs[f];
      if (fp.v[f]) fp.v[f] = p.vs[f];
    }
  }
  int exact = a->exact_size != 0;
  TMP_LAUNCH(lookback_count_kernel, dim3(p.nblocks), dim3(LB_THREADS), st, res.comp, a->n, lay, a->lookback,
             exact, p.full, exact ? nullptr : a->out_len, p.block_counts);
  TMP_LAUNCH(lookback_offsets_kernel, dim3(1), dim3(1024), st, p.block_counts, p.nblocks, a->out_count);
  TMP_LAUNCH(lookback_slots_kernel, dim3(p.nblocks), dim3(LB_THREADS), st, res.comp, res.perm, a->n, exact,
             p.full, p.block_counts, p.row_pos, a->out_rows);
  int64_t max_elems = a->n * (int64_t)a->lookback * a->n_feat;
  const int64_t per_row = (int64_t)a->lookback * a->n_feat;
  if (per_row <= (int64_t)LBW_MAX_PER_ROW) {
    // LDS room for the staged histories: 9 bytes per slot and feature, at most ~60 KB; pointless
    // when it cannot even hold one row's history plus a few rows
    int cap = 61440 / (9 * a->n_feat);
    cap = cap > 4096 ? 4096 : cap / 16 * 16;
    if (cap < a->lookback + 64) cap = 0;
    // no more than a block of dense rows needs (256 rows + one history + slack): the smaller the
    // staging area, the more blocks a CU holds
    const int want = (a->lookback + LBW_ROWS + 128 + 15) / 16 * 16;
    if (cap > want) cap = want;
    ProfScope ps("lookback_write_kernel", st);
    hipLaunchKernelGGL(lookback_write_kernel, dim3((unsigned)((a->n + LBW_ROWS - 1) / LBW_ROWS)), dim3(LB_THREADS),
                       (size_t)cap * a->n_feat * 9, st, p.row_pos, feat_perm, a->out_len, a->out_count, a->lookback,
                       a->n_feat, exact, fp, a->out_feat, a->out_feat_valid, 1.0f / (float)per_row,
                       1.0f / (float)a->n_feat, cap);
  }
  else
    TMP_LAUNCH(lookback_write_wide_kernel, dim3(grid_for(max_elems, LB_THREADS * 8, 1 << 16)), dim3(LB_THREADS), st,
               p.row_pos, feat_perm, a->out_len, a->out_count, a->lookback, a->n_feat, exact, fp, a->out_feat,
               a->out_feat_valid);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
