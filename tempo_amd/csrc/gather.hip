// Row gather by int32 row ids: the Project that materialises joined / reordered columns
// (python/tempo/tsdf.py:139,147) once the match or sort permutation is known.
#include "common.h"

namespace tmp {

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_kernel(const T *src, const uint8_t *src_valid,
                                                        const int32_t *idx, int64_t idx_stride,
                                                        int64_t n, T *out, uint8_t *out_valid) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t j = idx[i * idx_stride];
    bool ok = j >= 0;
    T v = ok ? src[j] : (T)0;
    if (ok && src_valid) ok = src_valid[j] != 0;
    out[i] = ok ? v : (T)0;
    if (out_valid) out_valid[i] = ok ? 1 : 0;
  }
}

}  // namespace tmp

using namespace tmp;

extern "C" int tmp_gather(const void *src, const uint8_t *src_valid, int32_t elem_bytes,
                          const int32_t *idx, int64_t idx_stride, int64_t n, void *out,
                          uint8_t *out_valid, void *stream) {
  clear_error();
  TMP_REQUIRE(n >= 0 && idx_stride >= 1, TMP_EINVAL, "tmp_gather: bad row count or index stride");
  if (n == 0) return TMP_OK;
  TMP_REQUIRE(src && idx && out, TMP_EINVAL, "tmp_gather: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int grid = grid_for(n, kBlock * 4, 16384);
  switch (elem_bytes) {
    case 1:
      TMP_LAUNCH(gather_kernel<uint8_t>, dim3(grid), dim3(kBlock), st, (const uint8_t *)src,
                         src_valid, idx, idx_stride, n, (uint8_t *)out, out_valid);
      break;
    case 4:
      TMP_LAUNCH(gather_kernel<uint32_t>, dim3(grid), dim3(kBlock), st, (const uint32_t *)src,
                         src_valid, idx, idx_stride, n, (uint32_t *)out, out_valid);
      break;
    case 8:
      TMP_LAUNCH(gather_kernel<uint64_t>, dim3(grid), dim3(kBlock), st, (const uint64_t *)src,
                         src_valid, idx, idx_stride, n, (uint64_t *)out, out_valid);
      break;
    default:
      TMP_REQUIRE(false, TMP_EINVAL, "tmp_gather: elem_bytes must be 1, 4 or 8");
  }
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
