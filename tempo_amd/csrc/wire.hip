// Narrow and widen the columns of a key shuffle (the Exchange behind partitionBy(partitionCols),
// python/tempo/tsdf.py:117-121 and :563-572): on the way out the key id and the 0/1 NULL-flag columns
// share one word of 1, 2 or 4 bytes, every int64 column travels as its offset from a minimum in 4, 5
// or 6 bytes; on arrival the columns are restored bit for bit.  One sweep each way, four rows per
// thread where the columns are aligned for it.
#include <string.h>

#include "common.h"

namespace tmp {

struct WireCols {
  int64_t n;
  int32_t *key;            // plain side
  int key_bits, key_bytes, n_masks, n_ints;
  uint8_t *mask[TMP_WIRE_MAX_MASKS];
  int64_t *ints[TMP_WIRE_MAX_INTS];
  int64_t int_min[TMP_WIRE_MAX_INTS];
  int int_bytes[TMP_WIRE_MAX_INTS];
  void *w_key;             // wire side
  uint32_t *w_lo[TMP_WIRE_MAX_INTS];
  void *w_hi[TMP_WIRE_MAX_INTS];
};

__device__ __forceinline__ void store_key_word(void *p, int bytes, int64_t i, uint32_t v) {
  if (bytes == 1) ((uint8_t *)p)[i] = (uint8_t)v;
  else if (bytes == 2) ((uint16_t *)p)[i] = (uint16_t)v;
  else ((uint32_t *)p)[i] = v;
}
__device__ __forceinline__ uint32_t load_key_word(const void *p, int bytes, int64_t i) {
  if (bytes == 1) return ((const uint8_t *)p)[i];
  if (bytes == 2) return ((const uint16_t *)p)[i];
  return ((const uint32_t *)p)[i];
}

template <bool ENCODE>
__device__ __forceinline__ void wire_row(const WireCols &c, int64_t i) {
  if (ENCODE) {
    uint32_t word = (uint32_t)c.key[i];
    for (int m = 0; m < c.n_masks; ++m) word |= (c.mask[m][i] ? 1u : 0u) << (c.key_bits + m);
    store_key_word(c.w_key, c.key_bytes, i, word);
    for (int k = 0; k < c.n_ints; ++k) {
      const uint64_t rel = (uint64_t)(c.ints[k][i] - c.int_min[k]);
      c.w_lo[k][i] = (uint32_t)rel;
      if (c.int_bytes[k] == 5) ((uint8_t *)c.w_hi[k])[i] = (uint8_t)(rel >> 32);
      else if (c.int_bytes[k] == 6) ((uint16_t *)c.w_hi[k])[i] = (uint16_t)(rel >> 32);
    }
  } else {
    const uint32_t word = load_key_word(c.w_key, c.key_bytes, i);
    c.key[i] = (int32_t)(c.key_bits < 32 ? word & (uint32_t)((1ull << c.key_bits) - 1) : word);
    for (int m = 0; m < c.n_masks; ++m) c.mask[m][i] = (uint8_t)((word >> (c.key_bits + m)) & 1u);
    for (int k = 0; k < c.n_ints; ++k) {
      uint64_t rel = c.w_lo[k][i];
      if (c.int_bytes[k] == 5) rel |= (uint64_t)((const uint8_t *)c.w_hi[k])[i] << 32;
      else if (c.int_bytes[k] == 6) rel |= (uint64_t)((const uint16_t *)c.w_hi[k])[i] << 32;
      c.ints[k][i] = (int64_t)rel + c.int_min[k];
    }
  }
}

// Rows one by one: columns at any alignment, and the last n % 4 rows of the vector form.
template <bool ENCODE>
__global__ __launch_bounds__(kBlock) void wire_rows_kernel(WireCols c, int64_t from) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += stride) wire_row<ENCODE>(c, i);
}

// Four consecutive rows per thread: 16-byte accesses of the 4- and 8-byte columns, 4-byte ones of the
// flag bytes and of one-byte key words (every column 16-byte aligned; rows [0, n - n % 4)).
template <bool ENCODE>
__global__ __launch_bounds__(kBlock) void wire_rows4_kernel(WireCols c) {
  const int64_t n4 = c.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t key_mask = c.key_bits < 32 ? (uint32_t)((1ull << c.key_bits) - 1) : ~0u;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    uint32_t word[4];
    if (ENCODE) {
      const int4 kv = reinterpret_cast<const int4 *>(c.key)[q];
      word[0] = (uint32_t)kv.x; word[1] = (uint32_t)kv.y; word[2] = (uint32_t)kv.z; word[3] = (uint32_t)kv.w;
      for (int m = 0; m < c.n_masks; ++m) {
        const uint32_t f = reinterpret_cast<const uint32_t *>(c.mask[m])[q];
#pragma unroll
        for (int u = 0; u < 4; ++u) word[u] |= (((f >> (8 * u)) & 0xffu) ? 1u : 0u) << (c.key_bits + m);
      }
      if (c.key_bytes == 1)
        reinterpret_cast<uint32_t *>(c.w_key)[q] = (word[0] & 0xffu) | ((word[1] & 0xffu) << 8) | ((word[2] & 0xffu) << 16) | (word[3] << 24);
      else if (c.key_bytes == 2)
        reinterpret_cast<uint2 *>(c.w_key)[q] = make_uint2((word[0] & 0xffffu) | (word[1] << 16), (word[2] & 0xffffu) | (word[3] << 16));
      else
        reinterpret_cast<uint4 *>(c.w_key)[q] = make_uint4(word[0], word[1], word[2], word[3]);
      for (int k = 0; k < c.n_ints; ++k) {
        const longlong2 a = reinterpret_cast<const longlong2 *>(c.ints[k])[2 * q];
        const longlong2 b = reinterpret_cast<const longlong2 *>(c.ints[k])[2 * q + 1];
        const uint64_t r[4] = {(uint64_t)(a.x - c.int_min[k]), (uint64_t)(a.y - c.int_min[k]), (uint64_t)(b.x - c.int_min[k]),
                               (uint64_t)(b.y - c.int_min[k])};
        reinterpret_cast<uint4 *>(c.w_lo[k])[q] = make_uint4((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        if (c.int_bytes[k] == 5)
          reinterpret_cast<uint32_t *>(c.w_hi[k])[q] = (uint32_t)((r[0] >> 32) & 0xff) | (uint32_t)(((r[1] >> 32) & 0xff) << 8) |
                                                       (uint32_t)(((r[2] >> 32) & 0xff) << 16) | (uint32_t)(((r[3] >> 32) & 0xff) << 24);
        else if (c.int_bytes[k] == 6)
          reinterpret_cast<uint2 *>(c.w_hi[k])[q] = make_uint2((uint32_t)((r[0] >> 32) & 0xffff) | (uint32_t)(((r[1] >> 32) & 0xffff) << 16),
                                                               (uint32_t)((r[2] >> 32) & 0xffff) | (uint32_t)(((r[3] >> 32) & 0xffff) << 16));
      }
    } else {
      if (c.key_bytes == 1) {
        const uint32_t f = reinterpret_cast<const uint32_t *>(c.w_key)[q];
#pragma unroll
        for (int u = 0; u < 4; ++u) word[u] = (f >> (8 * u)) & 0xffu;
      } else if (c.key_bytes == 2) {
        const uint2 f = reinterpret_cast<const uint2 *>(c.w_key)[q];
        word[0] = f.x & 0xffffu; word[1] = f.x >> 16; word[2] = f.y & 0xffffu; word[3] = f.y >> 16;
      } else {
        const uint4 f = reinterpret_cast<const uint4 *>(c.w_key)[q];
        word[0] = f.x; word[1] = f.y; word[2] = f.z; word[3] = f.w;
      }
      reinterpret_cast<int4 *>(c.key)[q] = make_int4((int)(word[0] & key_mask), (int)(word[1] & key_mask), (int)(word[2] & key_mask),
                                                     (int)(word[3] & key_mask));
      for (int m = 0; m < c.n_masks; ++m) {
        uint32_t f = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) f |= ((word[u] >> (c.key_bits + m)) & 1u) << (8 * u);
        reinterpret_cast<uint32_t *>(c.mask[m])[q] = f;
      }
      for (int k = 0; k < c.n_ints; ++k) {
        const uint4 lo = reinterpret_cast<const uint4 *>(c.w_lo[k])[q];
        uint64_t r[4] = {lo.x, lo.y, lo.z, lo.w};
        if (c.int_bytes[k] == 5) {
          const uint32_t h = reinterpret_cast<const uint32_t *>(c.w_hi[k])[q];
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] |= (uint64_t)((h >> (8 * u)) & 0xffu) << 32;
        } else if (c.int_bytes[k] == 6) {
          const uint2 h = reinterpret_cast<const uint2 *>(c.w_hi[k])[q];
          r[0] |= (uint64_t)(h.x & 0xffffu) << 32; r[1] |= (uint64_t)(h.x >> 16) << 32;
          r[2] |= (uint64_t)(h.y & 0xffffu) << 32; r[3] |= (uint64_t)(h.y >> 16) << 32;
        }
        reinterpret_cast<longlong2 *>(c.ints[k])[2 * q] = make_longlong2((int64_t)r[0] + c.int_min[k], (int64_t)r[1] + c.int_min[k]);
        reinterpret_cast<longlong2 *>(c.ints[k])[2 * q + 1] = make_longlong2((int64_t)r[2] + c.int_min[k], (int64_t)r[3] + c.int_min[k]);
      }
    }
  }
}

static int wire_check(const tmp_wire_args *a, WireCols *c) {
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_wire: null args");
  TMP_REQUIRE(a->n >= 0, TMP_EINVAL, "tmp_wire: n must not be negative");
  TMP_REQUIRE(a->n_masks >= 0 && a->n_masks <= TMP_WIRE_MAX_MASKS && a->n_ints >= 0 && a->n_ints <= TMP_WIRE_MAX_INTS, TMP_EINVAL,
              "tmp_wire: at most %d flag columns and %d integer columns", TMP_WIRE_MAX_MASKS, TMP_WIRE_MAX_INTS);
  TMP_REQUIRE(a->key_bits >= 1 && a->key_bits <= 32 && a->key_bits + a->n_masks <= 32, TMP_EINVAL,
              "tmp_wire: key_bits + n_masks must fit 32 bits");
  TMP_REQUIRE(a->key_bytes == 1 || a->key_bytes == 2 || a->key_bytes == 4, TMP_EINVAL, "tmp_wire: key_bytes must be 1, 2 or 4");
  TMP_REQUIRE(a->key_bits + a->n_masks <= 8 * a->key_bytes, TMP_EINVAL, "tmp_wire: key word too narrow for key_bits + n_masks");
  memset(c, 0, sizeof(*c));
  c->n = a->n; c->key = a->key; c->key_bits = a->key_bits; c->key_bytes = a->key_bytes;
  c->n_masks = a->n_masks; c->n_ints = a->n_ints; c->w_key = a->wire_key;
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->key && a->wire_key, TMP_EINVAL, "tmp_wire: key column missing");
  for (int m = 0; m < a->n_masks; ++m) {
    TMP_REQUIRE(a->masks[m], TMP_EINVAL, "tmp_wire: flag column %d is null", m);
    c->mask[m] = a->masks[m];
  }
  for (int k = 0; k < a->n_ints; ++k) {
    TMP_REQUIRE(a->ints[k] && a->wire_lo[k], TMP_EINVAL, "tmp_wire: integer column %d is null", k);
    TMP_REQUIRE(a->int_bytes[k] == 4 || a->int_bytes[k] == 5 || a->int_bytes[k] == 6, TMP_EINVAL,
                "tmp_wire: integer column %d: 4, 5 or 6 bytes", k);
    TMP_REQUIRE(a->int_bytes[k] == 4 || a->wire_hi[k], TMP_EINVAL, "tmp_wire: integer column %d: upper part missing", k);
    c->ints[k] = a->ints[k]; c->int_min[k] = a->int_min[k]; c->int_bytes[k] = a->int_bytes[k];
    c->w_lo[k] = (uint32_t *)a->wire_lo[k]; c->w_hi[k] = a->int_bytes[k] == 4 ? nullptr : a->wire_hi[k];
  }
  return TMP_OK;
}

static bool wire_aligned(const WireCols &c) {
  uintptr_t all = (uintptr_t)c.key | (uintptr_t)c.w_key;
  for (int m = 0; m < c.n_masks; ++m) all |= (uintptr_t)c.mask[m];
  for (int k = 0; k < c.n_ints; ++k) all |= (uintptr_t)c.ints[k] | (uintptr_t)c.w_lo[k] | (uintptr_t)c.w_hi[k];
  return (all & 15) == 0;
}

template <bool ENCODE>
static int wire_run(const tmp_wire_args *a, hipStream_t st) {
  WireCols c;
  TMP_TRY(wire_check(a, &c));
  if (c.n == 0) return TMP_OK;
  int64_t from = 0;
  if (wire_aligned(c) && c.n >= 4) {
    ProfScope ps(ENCODE ? "wire_encode_kernel" : "wire_decode_kernel", st);
    hipLaunchKernelGGL((wire_rows4_kernel<ENCODE>), dim3(grid_for(c.n >> 2, kBlock, 1 << 16)), dim3(kBlock), 0, st, c);
    from = (c.n >> 2) << 2;
  }
  if (from < c.n) {
    ProfScope ps(ENCODE ? "wire_encode_kernel" : "wire_decode_kernel", st);
    hipLaunchKernelGGL((wire_rows_kernel<ENCODE>), dim3(grid_for(c.n - from, kBlock, 1 << 16)), dim3(kBlock), 0, st, c, from);
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

}  // namespace tmp

using namespace tmp;

extern "C" int tmp_wire_encode(const tmp_wire_args *a, void *stream) {
  clear_error();
  return wire_run<true>(a, (hipStream_t)stream);
}

extern "C" int tmp_wire_decode(const tmp_wire_args *a, void *stream) {
  clear_error();
  return wire_run<false>(a, (hipStream_t)stream);
}
