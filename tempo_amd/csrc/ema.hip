// EMA: fixed-tap FIR over the previous rows of the same series.  Replaces the `window` Lag
// frames and the running when/otherwise sum of python/tempo/tsdf.py:615-635.
//
// Rows are ordered as __baseWindow does (tsdf.py:563-572).  A block stages the metric values
// (NULL -> 0) and key ids of its 1024 rows plus a (taps - 1)-row halo in LDS and works out where
// the series of every staged row starts.  A thread then owns four consecutive rows: one value
// read from LDS serves a tap of each of them, and every row adds its w[k] * x terms in the
// reference's order k = 0, 1, 2, ... (no fused multiply-add: the library is built with
// -ffp-contract=off), stopping where its series starts -- lag() beyond that is NULL.
#include <string.h>

#include "common.h"

namespace tmp {

constexpr int EM_ROWS = 1024;
constexpr int EM_THREADS = 256;
constexpr int EM_PER = EM_ROWS / EM_THREADS;  // rows per thread

__global__ __launch_bounds__(EM_THREADS) void ema_kernel(const uint64_t *comp, const uint32_t *perm,
                                                         const double *xs, const double *x_orig,
                                                         int64_t n, Layout lay, int taps,
                                                         const double *weights, int32_t *out_perm,
                                                         double *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int wave_last[EM_THREADS / 64];
  const int halo = taps > 0 ? taps - 1 : 0;
  const int cap = halo + EM_ROWS;
  double *xv = (double *)smem;            // [cap] metric, 0 for NULL
  double *w = xv + cap;                   // [taps]
  int *st = (int *)(w + (taps > 0 ? taps : 1));  // [cap] key id, then: first slot of the row's series
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * EM_ROWS;
  const int64_t t1 = t0 + EM_ROWS < n ? t0 + EM_ROWS : n;
  const int64_t h0 = t0 - halo;  // may be negative
  const int total = (int)(t1 - h0);
  const int ks = lay.key_shift();
  const bool has_key = lay.key_bits > 0;
  for (int e = tid; e < total; e += EM_THREADS) {
    int64_t j = h0 + e;
    double v = 0.0;
    int kid = -1;  // rows before the table: a series of their own
    if (j >= 0) {
      uint64_t c = comp[j];
      kid = has_key ? (int)(uint32_t)(c >> ks) : 0;
      if (c & 1ull) v = xs ? xs[j] : x_orig[perm[j]];
    }
    xv[e] = v;
    st[e] = kid;
  }
  for (int k = tid; k < taps; k += EM_THREADS) w[k] = weights[k];
  __syncthreads();

  // ---- first slot of every slot's series: a running maximum over "a series starts here" marks.
  //      Thread t owns the slots [t * per, (t + 1) * per); slot 0 counts as a start (a series that
  //      began before the halo is at least `taps` rows long for every row of the tile).
  const int per = (total + EM_THREADS - 1) / EM_THREADS;
  const int lo = tid * per, hi = lo + per < total ? lo + per : total;
  int prev = lo > 0 && lo <= total ? st[lo - 1] : -2;
  int last = -1;  // last start at or before the slot, inside this thread's range
  {
    int p = prev;
    for (int e = lo; e < hi; ++e) {
      const int kid = st[e];
      if (e == 0 || kid != p) last = e;
      p = kid;
    }
  }
  int inc = last;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o && t > inc) inc = t;
  }
  if (lane == 63) wave_last[wave] = inc;
  int carry = __shfl_up(inc, 1, 64);
  if (lane == 0) carry = -1;
  __syncthreads();  // every thread has read the key ids it needs; st is rewritten below
  for (int wv = 0; wv < wave; ++wv) carry = carry > wave_last[wv] ? carry : wave_last[wv];
  {
    int p = prev, run = carry;
    // second sweep over the same slots: key ids are still in place for this thread's own range
    for (int e = lo; e < hi; ++e) {
      const int kid = st[e];
      if (e == 0 || kid != p) run = e;
      p = kid;
      st[e] = run;
    }
  }
  __syncthreads();

  // ---- four consecutive rows per thread
  const int e0 = halo + tid * EM_PER;
  const int64_t i0 = t0 + (int64_t)tid * EM_PER;
  if (i0 >= t1) return;
  int m[EM_PER];
  double acc[EM_PER];
#pragma unroll
  for (int r = 0; r < EM_PER; ++r) {
    const int e = e0 + r;
    int len = e < total ? e - st[e] + 1 : 0;  // rows of the series up to and including this one
    m[r] = len < taps ? len : taps;
    acc[r] = 0.0;
  }
  bool deep = true;  // every one of the thread's rows has `taps` rows of its series behind it
#pragma unroll
  for (int r = 0; r < EM_PER; ++r) deep = deep && m[r] == taps;
  if (__all(deep) && taps >= EM_PER) {
    // the common case, without the per-term test: row r takes w[d + r] * x in step d, the weights
    // slide through registers (one LDS read of a weight per step instead of one per row)
    static_assert(EM_PER == 4, "four weights in flight");
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = w[0];  // w[d], w[d + 1], w[d + 2], w[d + 3]; 0 outside [0, taps)
    // steps d = -3 .. -1: only the rows whose tap index d + r is not negative take part
    {
      const double xa = xv[e0 + 3], xb = xv[e0 + 2], xc = xv[e0 + 1];
      acc[3] = acc[3] + w[0] * xa;                               // d = -3: k = 0 for r = 3
      acc[2] = acc[2] + w[0] * xb; acc[3] = acc[3] + w[1] * xb;  // d = -2
      acc[1] = acc[1] + w[0] * xc; acc[2] = acc[2] + w[1] * xc; acc[3] = acc[3] + w[2] * xc;  // d = -1
      w0 = w[0]; w1 = w[1]; w2 = w[2]; w3 = w[3];
    }
    int d = 0;
    for (; d + EM_PER <= taps; ++d) {  // all four tap indices inside [0, taps)
      const double x = xv[e0 - d];
      acc[0] = acc[0] + w0 * x;
      acc[1] = acc[1] + w1 * x;
      acc[2] = acc[2] + w2 * x;
      acc[3] = acc[3] + w3 * x;
      w0 = w1; w1 = w2; w2 = w3;
      w3 = d + EM_PER < taps ? w[d + EM_PER] : 0.0;
    }
    for (; d < taps; ++d) {  // the last three steps: tap indices run off the end for the later rows
      const double x = xv[e0 - d];
#pragma unroll
      for (int r = 0; r < EM_PER; ++r) {
        const int k = d + r;  // uniform over the block
        if (k < taps) acc[r] = acc[r] + w[k] * x;
      }
    }
  } else {
    for (int d = -(EM_PER - 1); d < taps; ++d) {
      const int at = e0 - d;
      const double x = at < total ? xv[at] : 0.0;
#pragma unroll
      for (int r = 0; r < EM_PER; ++r) {
        const int k = d + r;  // uniform over the block
        if (k >= 0 && k < taps) {
          const double t = w[k] * x;
          if (k < m[r]) acc[r] = acc[r] + t;
        }
      }
    }
  }
  // (the ordering step may have left the row ids in out_perm itself: nothing to copy then)
  const bool ids_in_place = (const void *)perm == (const void *)out_perm;
  const bool vec = i0 + EM_PER <= t1 && (((uintptr_t)out | (uintptr_t)out_perm | (uintptr_t)perm) & 15) == 0;
  if (vec) {
    *reinterpret_cast<double2 *>(out + i0) = make_double2(acc[0], acc[1]);
    *reinterpret_cast<double2 *>(out + i0 + 2) = make_double2(acc[2], acc[3]);
    if (!ids_in_place) {
      const uint4 pv = *reinterpret_cast<const uint4 *>(perm + i0);
      *reinterpret_cast<uint4 *>(out_perm + i0) = pv;
    }
  } else {
#pragma unroll
    for (int r = 0; r < EM_PER; ++r)
      if (i0 + r < t1) {
        out[i0 + r] = acc[r];
        if (!ids_in_place) out_perm[i0 + r] = (int32_t)perm[i0 + r];
      }
  }
}

struct EmaPlan {
  OrderBuffers ob;
  double *xs;
  double *w;
};

static void ema_plan(const tmp_ema_args *a, Workspace &ws, EmaPlan *p) {
  order_plan(a->n, ws, &p->ob, plan_keys(a->key, a->n_keys), 1);  // the metric travels with the rows
  p->xs = ws.take<double>((size_t)(a->n > 0 ? a->n : 1));
  p->w = ws.take<double>(TMP_MAX_TAPS);
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_ema_workspace_bytes(const tmp_ema_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  EmaPlan p;
  ema_plan(a, ws, &p);
  return ws.off;
}

extern "C" int tmp_ema(const tmp_ema_args *a, void *workspace, size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_ema: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_ema: row count must be in [0, 2^31)");
  TMP_REQUIRE(a->n_taps >= 0 && a->n_taps <= TMP_MAX_TAPS, TMP_EINVAL, "tmp_ema: n_taps must be in [0, %d]",
              TMP_MAX_TAPS);
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->x && a->out_perm && a->out_ema && (a->n_taps == 0 || a->weights), TMP_EINVAL,
              "tmp_ema: null pointer");
  Workspace ws(workspace, workspace_bytes);
  EmaPlan p;
  ema_plan(a, ws, &p);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_ema: workspace too small");
  // the second row-id buffer of the ordering step is the caller's out_perm: when the order ends up there
  // (a single pass does), the row ids are where they belong and the kernel neither reads nor copies them
  if (((uintptr_t)a->out_perm & 255) == 0) p.ob.idx[1] = (uint32_t *)a->out_perm;  // aligned like the workspace's own
  hipStream_t st = (hipStream_t)stream;
  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n; s.key = a->key; s.n_keys = a->n_keys; s.ts = a->ts;
  s.prim_div = a->ts_is_seconds ? 1 : 1000000;
  s.seq = a->seq; s.seq_valid = a->seq_valid;
  s.n_masks = 1;
  s.masks[0] = a->x_valid;
  Layout lay;
  OrderResult res;
  int taps = a->n_taps;  // zero taps: the column stays lit(0)
  // The tap weights go to the device ahead of the range scan: the host read of the ranges below
  // waits for the stream, so the caller's (pageable) array has been consumed before this call
  // returns and may be released right after it.
  if (taps > 0)
    TMP_CHECK_HIP(hipMemcpyAsync(p.w, a->weights, sizeof(double) * taps, hipMemcpyHostToDevice, st));
  TMP_TRY(order_scan_ranges(s, p.ob, st));
  RangeInfo r;
  TMP_TRY(order_read_ranges(p.ob, &r, st));
  TMP_TRY(order_make_layout(s, r, 1, &lay));
  bool carried = false;
  const uint64_t *pay_src[1] = {(const uint64_t *)a->x};
  uint64_t *pay_out[1] = {(uint64_t *)p.xs};
  TMP_TRY(order_build(s, lay, sorted_mode(r), p.ob, &res, st, 1, pay_src, pay_out, &carried));
  const double *xs = carried ? p.xs : a->x;
  size_t shm = (size_t)((taps > 0 ? taps - 1 : 0) + EM_ROWS) * 12 + (size_t)(taps > 0 ? taps : 1) * 8;
  int grid = (int)((a->n + EM_ROWS - 1) / EM_ROWS);
  {
    ProfScope ps("ema_kernel", st);
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(EM_THREADS), shm, st, res.comp, res.perm, xs, a->x, a->n,
                       lay, taps, p.w, a->out_perm, a->out_ema);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
