// EMA: fixed-tap FIR over the previous rows of the same series.  Replaces the `window` Lag
// frames and the running when/otherwise sum of python/tempo/tsdf.py:615-635.
//
// Rows are ordered as __baseWindow does (tsdf.py:563-572).  A block stages the sort words and
// metric values of its 1024 rows plus a (taps - 1)-row halo in LDS; each row then walks back
// through its taps while the key field still matches, adding w[k] * x in the reference's
// left-to-right order (no fused multiply-add: the library is built with -ffp-contract=off).
#include <string.h>

#include "common.h"

namespace tmp {

constexpr int EM_ROWS = 1024;
constexpr int EM_THREADS = 256;

__global__ __launch_bounds__(EM_THREADS) void ema_kernel(const uint64_t *comp, const uint32_t *perm,
                                                         const double *xs, const double *x_orig,
                                                         int64_t n, Layout lay, int taps,
                                                         const double *weights, int32_t *out_perm,
                                                         double *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int halo = taps > 0 ? taps - 1 : 0;
  uint64_t *cw = (uint64_t *)smem;                 // [halo + EM_ROWS]
  double *xv = (double *)(cw + halo + EM_ROWS);    // [halo + EM_ROWS]
  double *w = xv + halo + EM_ROWS;                 // [taps]
  const int64_t t0 = (int64_t)blockIdx.x * EM_ROWS;
  const int64_t t1 = t0 + EM_ROWS < n ? t0 + EM_ROWS : n;
  const int64_t h0 = t0 - halo;  // may be negative
  const int total = (int)(t1 - h0);
  for (int e = threadIdx.x; e < total; e += EM_THREADS) {
    int64_t j = h0 + e;
    uint64_t c = 0;
    double v = 0.0;
    if (j >= 0) {
      c = comp[j];
      if (c & 1ull) v = xs ? xs[j] : x_orig[perm[j]];
    }
    cw[e] = c;
    xv[e] = v;
  }
  for (int k = threadIdx.x; k < taps; k += EM_THREADS) w[k] = weights[k];
  __syncthreads();
  const int ks = lay.key_shift();
  const bool has_key = lay.key_bits > 0;
  for (int64_t i = t0 + threadIdx.x; i < t1; i += EM_THREADS) {
    int e = (int)(i - h0);
    uint64_t ci = cw[e];
    uint64_t key_i = has_key ? (ci >> ks) : 0;
    double acc = 0.0;
    for (int k = 0; k < taps; ++k) {
      int64_t j = i - k;
      if (j < 0) break;
      uint64_t cj = cw[e - k];
      if (has_key && (cj >> ks) != key_i) break;  // lag() beyond the series start is NULL
      if (cj & 1ull) {
        double t = w[k] * xv[e - k];
        acc = acc + t;
      }
    }
    out[i] = acc;
    out_perm[i] = (int32_t)perm[i];
  }
}

struct EmaPlan {
  OrderBuffers ob;
  double *xs;
  double *w;
};

static void ema_plan(const tmp_ema_args *a, Workspace &ws, EmaPlan *p) {
  order_plan(a->n, ws, &p->ob);
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
  TMP_TRY(order_scan_ranges(s, p.ob, st));
  RangeInfo r;
  TMP_TRY(order_read_ranges(p.ob, &r, st));
  TMP_TRY(order_make_layout(s, r, 1, &lay));
  bool carried = false;
  const uint64_t *pay_src[1] = {(const uint64_t *)a->x};
  uint64_t *pay_out[1] = {(uint64_t *)p.xs};
  TMP_TRY(order_build(s, lay, sorted_mode(r), p.ob, &res, st, 1, pay_src, pay_out, &carried));
  const double *xs = carried ? p.xs : a->x;
  int taps = a->n_taps;  // zero taps: the column stays lit(0)
  if (taps > 0)
    TMP_CHECK_HIP(hipMemcpyAsync(p.w, a->weights, sizeof(double) * taps, hipMemcpyHostToDevice, st));
  size_t shm = (size_t)((taps > 0 ? taps - 1 : 0) + EM_ROWS) * 16 + (size_t)(taps > 0 ? taps : 1) * 8;
  int grid = (int)((a->n + EM_ROWS - 1) / EM_ROWS);
  {
    ProfScope ps("ema_kernel", st);
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(EM_THREADS), shm, st, res.comp, res.perm, xs, a->x, a->n,
                       lay, taps, p.w, a->out_perm, a->out_ema);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
