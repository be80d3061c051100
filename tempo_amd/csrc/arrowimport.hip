// Arrow C data interface import: a column chunk goes from the Arrow buffers in host memory to a
// device column (python/tempo/tsdf.py:24-36 takes a Spark DataFrame; its partitions reach a
// Python worker as Arrow batches).  The value buffer is copied once, host to device; where the
// stored type is not the column's type a kernel converts it on the device, and the validity
// bitmap is expanded to one byte per row there as well.
#include <cstring>

#include "common.h"

namespace tmp {

enum ArrowStore { AS_I8, AS_U8, AS_I16, AS_U16, AS_I32, AS_U32, AS_I64, AS_U64, AS_F32, AS_F64, AS_BAD };

struct ArrowType {
  ArrowStore store;
  int width;
  int64_t ts_mul;  // timestamps: microseconds = value * ts_mul / ts_div (floor); 0 = not a timestamp
  int64_t ts_div;
};

static ArrowType parse_format(const char *f) {
  ArrowType t{AS_BAD, 0, 0, 1};
  if (!f) return t;
  if (f[0] == 't' && f[1] == 's' && f[2] && f[3] == ':') {
    t.store = AS_I64; t.width = 8; t.ts_mul = 1;
    switch (f[2]) {
      case 's': t.ts_mul = 1000000; break;
      case 'm': t.ts_mul = 1000; break;
      case 'u': break;
      case 'n': t.ts_div = 1000; break;
      default: t.store = AS_BAD;
    }
    return t;
  }
  if (f[1] != 0) return t;
  switch (f[0]) {
    case 'c': t.store = AS_I8; t.width = 1; break;
    case 'C': t.store = AS_U8; t.width = 1; break;
    case 's': t.store = AS_I16; t.width = 2; break;
    case 'S': t.store = AS_U16; t.width = 2; break;
    case 'i': t.store = AS_I32; t.width = 4; break;
    case 'I': t.store = AS_U32; t.width = 4; break;
    case 'l': t.store = AS_I64; t.width = 8; break;
    case 'L': t.store = AS_U64; t.width = 8; break;
    case 'f': t.store = AS_F32; t.width = 4; break;
    case 'g': t.store = AS_F64; t.width = 8; break;
    default: break;
  }
  return t;
}

// out[i] = convert(in[i]); timestamps scaled to microseconds (floor); NULL slots become 0.
// bits: the validity bitmap from the byte that holds the first row on, bit_off = that row's bit.
template <typename S, typename D>
__global__ __launch_bounds__(kBlock) void arrow_convert_kernel(const S *in, D *out, int64_t n, int64_t mul, int64_t div,
                                                               const uint8_t *bits, int bit_off, uint8_t *valid) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    bool ok = true;
    if (bits) {
      const int64_t b = i + bit_off;
      ok = (bits[b >> 3] >> (b & 7)) & 1;
    }
    if (valid) valid[i] = ok ? 1 : 0;
    if (in) {
      D v = (D)in[i];
      if (mul != 1) v = (D)((int64_t)v * mul);
      if (div != 1) v = (D)floordiv_i64((int64_t)v, div);
      out[i] = ok ? v : (D)0;
    } else if (!ok) {
      out[i] = (D)0;  // the values were copied in place: only the NULL slots are touched
    }
  }
}

template <typename D>
static int launch_convert(ArrowStore store, const void *in, D *out, int64_t n, int64_t mul, int64_t div,
                          const uint8_t *bits, int bit_off, uint8_t *valid, hipStream_t st) {
  const int grid = grid_for(n, kBlock * 4, 16384);
#define TMP_CONV(S)                                                                                        \
  TMP_LAUNCH((arrow_convert_kernel<S, D>), dim3(grid), dim3(kBlock), st, (const S *)in, out, n, mul, div, \
             bits, bit_off, valid)
  switch (store) {
    case AS_I8: TMP_CONV(int8_t); break;
    case AS_U8: TMP_CONV(uint8_t); break;
    case AS_I16: TMP_CONV(int16_t); break;
    case AS_U16: TMP_CONV(uint16_t); break;
    case AS_I32: TMP_CONV(int32_t); break;
    case AS_U32: TMP_CONV(uint32_t); break;
    case AS_I64: TMP_CONV(int64_t); break;
    case AS_U64: TMP_CONV(uint64_t); break;
    case AS_F32: TMP_CONV(float); break;
    case AS_F64: TMP_CONV(double); break;
    default: return TMP_EUNSUPPORTED;
  }
#undef TMP_CONV
  return TMP_OK;
}

struct ArrowPlan {
  uint8_t *stage;  // the value buffer's slice, when it has to be converted
  uint8_t *bits;   // the bitmap's slice
  size_t bitmap_bytes;
};

static ArrowPlan arrow_plan(const ArrowArray *a, const ArrowType &t, Workspace &ws) {
  ArrowPlan p{};
  p.stage = ws.take<uint8_t>((size_t)a->length * (size_t)t.width);
  p.bitmap_bytes = (size_t)(((a->offset & 7) + a->length + 7) >> 3);
  p.bits = ws.take<uint8_t>(p.bitmap_bytes);
  return p;
}

}  // namespace tmp

using namespace tmp;

extern "C" size_t tmp_arrow_import_workspace_bytes(const struct ArrowArray *array, const struct ArrowSchema *schema) {
  if (!array || !schema || array->length <= 0) return 256;
  ArrowType t = parse_format(schema->format);
  if (t.store == AS_BAD) return 256;
  Workspace ws(nullptr, 0);
  arrow_plan(array, t, ws);
  return ws.off + 256;
}

extern "C" int tmp_arrow_import(const struct ArrowArray *array, const struct ArrowSchema *schema, int32_t out_kind,
                                void *out_values, uint8_t *out_valid, int64_t *out_null_count, void *workspace,
                                size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(array && schema, TMP_EINVAL, "tmp_arrow_import: null array or schema");
  TMP_REQUIRE(array->length >= 0 && array->offset >= 0, TMP_EINVAL, "tmp_arrow_import: negative length or offset");
  TMP_REQUIRE(out_kind >= TMP_ARROW_I32 && out_kind <= TMP_ARROW_I64_RAW, TMP_EINVAL, "tmp_arrow_import: unknown out_kind");
  const ArrowType t = parse_format(schema->format);
  TMP_REQUIRE(t.store != AS_BAD, TMP_EUNSUPPORTED,
              "tmp_arrow_import: only integer, float32/64 and timestamp storage is imported");
  const bool is_float = t.store == AS_F32 || t.store == AS_F64;
  TMP_REQUIRE(!(is_float && out_kind != TMP_ARROW_F64), TMP_EUNSUPPORTED,
              "tmp_arrow_import: a floating-point array goes to a double column only");
  TMP_REQUIRE(!(schema->dictionary && is_float), TMP_EINVAL, "tmp_arrow_import: dictionary indices must be integers");
  TMP_REQUIRE(!(t.ts_mul && out_kind == TMP_ARROW_I32), TMP_EUNSUPPORTED,
              "tmp_arrow_import: timestamps need a 64-bit column");
  const int64_t n = array->length;
  const uint8_t *bitmap = (array->n_buffers >= 1 && array->buffers) ? (const uint8_t *)array->buffers[0] : nullptr;
  const bool may_have_nulls = bitmap && array->null_count != 0;
  // the NULL count, from the bitmap when the producer did not say
  int64_t nulls = array->null_count;
  if (nulls < 0) {
    nulls = 0;
    if (bitmap)
      for (int64_t i = 0; i < n; ++i) {
        const int64_t b = array->offset + i;
        nulls += !((bitmap[b >> 3] >> (b & 7)) & 1);
      }
  }
  TMP_REQUIRE(nulls == 0 || bitmap, TMP_EINVAL, "tmp_arrow_import: the array counts NULLs but has no validity bitmap");
  TMP_REQUIRE(nulls <= n, TMP_EINVAL, "tmp_arrow_import: null_count exceeds the length");
  if (out_null_count) *out_null_count = nulls;
  if (n == 0) return TMP_OK;
  TMP_REQUIRE(array->n_buffers >= 2 && array->buffers && array->buffers[1], TMP_EINVAL,
              "tmp_arrow_import: the array has no value buffer");
  TMP_REQUIRE(out_values, TMP_EINVAL, "tmp_arrow_import: null output");
  TMP_REQUIRE(out_valid || nulls == 0, TMP_EINVAL, "tmp_arrow_import: the array has NULLs and out_valid is null");
  hipStream_t st = (hipStream_t)stream;
  Workspace ws(workspace, workspace_bytes);
  ArrowPlan p = arrow_plan(array, t, ws);
  TMP_REQUIRE(!ws.overflow() && p.stage && p.bits, TMP_EWORKSPACE, "tmp_arrow_import: workspace too small");
  const uint8_t *values = (const uint8_t *)array->buffers[1] + (size_t)array->offset * (size_t)t.width;
  const bool scale = t.ts_mul != 0 && out_kind != TMP_ARROW_I64_RAW && (t.ts_mul != 1 || t.ts_div != 1);
  const int out_width = out_kind == TMP_ARROW_I32 ? 4 : 8;
  // stored type == column type: the value buffer is copied to its place and nothing converts it
  const bool in_place = !scale && t.width == out_width &&
                        ((out_kind == TMP_ARROW_F64) == is_float) && t.store != AS_F32;
  const uint8_t *dev_bits = nullptr;
  int bit_off = 0;
  if (nulls != 0 && may_have_nulls) {
    TMP_CHECK_HIP(hipMemcpyAsync(p.bits, bitmap + (array->offset >> 3), p.bitmap_bytes, hipMemcpyHostToDevice, st));
    dev_bits = p.bits;
    bit_off = (int)(array->offset & 7);
  }
  const void *dev_in = nullptr;
  if (in_place) {
    TMP_CHECK_HIP(hipMemcpyAsync(out_values, values, (size_t)n * (size_t)t.width, hipMemcpyHostToDevice, st));
  } else {
    TMP_CHECK_HIP(hipMemcpyAsync(p.stage, values, (size_t)n * (size_t)t.width, hipMemcpyHostToDevice, st));
    dev_in = p.stage;
  }
  if (!dev_in && !dev_bits) {  // nothing to convert, no NULLs
    if (out_valid) TMP_CHECK_HIP(hipMemsetAsync(out_valid, 1, (size_t)n, st));
    return TMP_OK;
  }
  const int64_t mul = scale ? t.ts_mul : 1, div = scale ? t.ts_div : 1;
  int rc;
  {
    ProfScope ps("arrow_convert_kernel", st);
    if (out_kind == TMP_ARROW_I32)
      rc = launch_convert<int32_t>(t.store, dev_in, (int32_t *)out_values, n, mul, div, dev_bits, bit_off, out_valid, st);
    else if (out_kind == TMP_ARROW_F64)
      rc = launch_convert<double>(t.store, dev_in, (double *)out_values, n, mul, div, dev_bits, bit_off, out_valid, st);
    else
      rc = launch_convert<int64_t>(t.store, dev_in, (int64_t *)out_values, n, mul, div, dev_bits, bit_off, out_valid, st);
  }
  TMP_REQUIRE(rc == TMP_OK, rc, "tmp_arrow_import: unsupported storage type");
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
