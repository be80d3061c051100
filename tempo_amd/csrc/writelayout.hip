// Layout of a date-partitioned, (key, time)-clustered table: the derived columns event_dt and
// event_time of python/tempo/io.py:29-30 and the row order that partitionBy("event_dt")
// (io.py:35) plus the clustering by (partitionCols, event_time) (io.py:39) ask for, as one
// linear order (event_dt, key, ts).
//
// The order comes from the ordering step of every other operator: rows are sorted by (key, ts),
// then one or two more stable radix passes over the day number move whole days together without
// disturbing the (key, ts) order inside a day.
#include <string.h>

#include "common.h"

namespace tmp {

constexpr int64_t kDayUs = 86400000000ll;

// event_dt: to_date(ts) as days since the epoch.  event_time: Spark prints the timestamp as
// "yyyy-MM-dd HH:mm:ss[.ffffff]" (trailing zeros of the fraction dropped), the reference keeps
// the part after the blank, removes the colons and casts the rest to double.  That decimal is
// (HHMMSS * 10^6 + microseconds) / 10^6; numerator and denominator are exact doubles, so the
// correctly rounded quotient is the double a decimal parser returns for the same text.
__global__ __launch_bounds__(kBlock) void event_cols_kernel(int64_t n, const int64_t *__restrict__ ts,
                                                            int32_t *__restrict__ out_dt,
                                                            double *__restrict__ out_time) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t t = ts[i];
    const int64_t day = floordiv_i64(t, kDayUs);
    const int64_t tod = t - day * kDayUs;  // [0, 86.4e9)
    const int64_t secs = floordiv_i64(tod, 1000000);
    const int64_t micros = tod - secs * 1000000;
    const int32_t s32 = (int32_t)secs;
    const int32_t h = s32 / 3600, m = (s32 % 3600) / 60, s = s32 % 60;
    const int64_t hhmmss = (int64_t)h * 10000 + m * 100 + s;
    if (out_dt) out_dt[i] = (int32_t)day;
    if (out_time) out_time[i] = (double)(hhmmss * 1000000 + micros) / 1000000.0;
  }
}

// Sorted sort word -> day number relative to the first day (the digit of the extra passes).
__global__ __launch_bounds__(kBlock) void day_digit_kernel(int64_t n, uint64_t *comp, Layout lay,
                                                           int64_t day_min) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    comp[i] = (uint64_t)(floordiv_i64(lay.prim_of(comp[i]), kDayUs) - day_min);
}

__global__ void perm_out_kernel(const uint32_t *in, int32_t *out, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (int32_t)in[i];
}

__global__ void day_range_kernel(int32_t *out, int32_t lo, int32_t hi) {
  out[0] = lo;
  out[1] = hi;
}

static int64_t host_floordiv(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_write_layout_workspace_bytes(const tmp_write_layout_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  OrderBuffers b;
  order_plan(a->n, ws, &b, plan_keys(a->key, a->n_keys));
  return ws.off;
}

extern "C" int tmp_write_layout(const tmp_write_layout_args *a, void *workspace,
                                size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_write_layout: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_write_layout: bad row count");
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->ts && a->out_perm, TMP_EINVAL, "tmp_write_layout: null pointer");
  TMP_REQUIRE(!a->key || a->n_keys >= 1, TMP_EINVAL, "tmp_write_layout: n_keys must be >= 1");
  Workspace ws(workspace, workspace_bytes);
  OrderBuffers b;
  order_plan(a->n, ws, &b, plan_keys(a->key, a->n_keys));
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_write_layout: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(a->n, kBlock * 4, 8192);
  if (a->out_event_dt || a->out_event_time)
    TMP_LAUNCH(event_cols_kernel, dim3(grid), dim3(kBlock), st, a->n, a->ts, a->out_event_dt,
               a->out_event_time);

  OrderSpec s;
  memset(&s, 0, sizeof(s));
  s.n = a->n;
  s.key = a->key;
  s.n_keys = a->n_keys;
  s.ts = a->ts;
  s.prim_div = 1;
  s.coarse_ok = 1;
  TMP_TRY(order_scan_ranges(s, b, st));
  RangeInfo r;
  TMP_TRY(order_read_ranges(b, &r, st));
  Layout lay;
  TMP_TRY(order_make_layout(s, r, 0, &lay));
  OrderResult res;
  TMP_TRY(order_build(s, lay, sorted_mode(r), b, &res, st));

  const int64_t day_min = host_floordiv(r.prim_min, kDayUs), day_max = host_floordiv(r.prim_max, kDayUs);
  if (a->out_dt_range)
    TMP_LAUNCH(day_range_kernel, dim3(1), dim3(1), st, a->out_dt_range, (int32_t)day_min,
               (int32_t)day_max);
  if (day_max > day_min) {
    // the ordered pair of buffers becomes the input side of the extra passes
    // (words and row ids may have ended in different halves of their ping-pong pairs)
    const int at = res.comp == b.comp[0] ? 0 : 1, at_ids = res.perm == b.idx[0] ? 0 : 1;
    OrderBuffers b2 = b;
    b2.comp[0] = b.comp[at]; b2.idx[0] = b.idx[at_ids];
    b2.comp[1] = b.comp[at ^ 1]; b2.idx[1] = b.idx[at_ids ^ 1];
    TMP_LAUNCH(day_digit_kernel, dim3(grid), dim3(kBlock), st, a->n, b2.comp[0], lay, day_min);
    TMP_TRY(order_radix_passes(b2, a->n, 0, bits_for((uint64_t)(day_max - day_min)), &res, st));
  }
  TMP_LAUNCH(perm_out_kernel, dim3(grid), dim3(kBlock), st, res.perm, a->out_perm, a->n);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
