// Device side of the key shuffle: bucket rows by owner = mix(key) % n_parts so that a single
// all-to-all sends every series to the GPU that owns it.  Replaces Spark's
// Exchange hashpartitioning(partitionCols) (python/tempo/tsdf.py:121, :571; resample.py:63).
// The bucketing is one stable radix pass of the ordering step with the owner as the digit.
#include <string.h>

#include "common.h"

namespace tmp {

__host__ __device__ __forceinline__ uint32_t mix_key(uint32_t k) {
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  k *= 0xc2b2ae35u;
  k ^= k >> 16;
  return k;
}

__global__ __launch_bounds__(kBlock) void owner_kernel(const int32_t *key, int64_t n, uint32_t n_parts,
                                                       uint64_t *comp, uint32_t *idx,
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
    if (in) {
      o = mix_key((uint32_t)key[i]) % n_parts;
      comp[i] = o;
      idx[i] = (uint32_t)i;
    }
    if (few) {
      for (uint32_t p = 0; p < n_parts; ++p) {
        uint64_t b = __ballot(in && o == p);
        if ((uint32_t)lane == p) mine += (unsigned int)__popcll(b);
      }
    } else if (in) {
      atomicAdd(&hist[o], 1u);
    }
  }
  if (few && mine) atomicAdd(&hist[lane], mine);
  __syncthreads();
  for (int i = threadIdx.x; i < (int)n_parts; i += kBlock)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
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

extern "C" size_t tmp_hash_partition_workspace_bytes(const tmp_partition_args *a) {
  if (!a) return 0;
  Workspace ws(nullptr, 0);
  OrderBuffers b;
  order_plan(a->n, ws, &b);
  return ws.off;
}

extern "C" int tmp_hash_partition(const tmp_partition_args *a, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  clear_error();
  TMP_REQUIRE(a != nullptr, TMP_EINVAL, "tmp_hash_partition: null args");
  TMP_REQUIRE(a->n >= 0 && a->n < (1ll << 31), TMP_EINVAL, "tmp_hash_partition: bad row count");
  TMP_REQUIRE(a->n_parts >= 1 && a->n_parts <= 256, TMP_EINVAL,
              "tmp_hash_partition: n_parts must be in [1, 256]");
  TMP_REQUIRE(a->out_counts, TMP_EINVAL, "tmp_hash_partition: out_counts is null");
  hipStream_t st = (hipStream_t)stream;
  TMP_CHECK_HIP(hipMemsetAsync(a->out_counts, 0, sizeof(int64_t) * a->n_parts, st));
  if (a->n == 0) return TMP_OK;
  TMP_REQUIRE(a->key && a->out_perm, TMP_EINVAL, "tmp_hash_partition: null pointer");
  Workspace ws(workspace, workspace_bytes);
  OrderBuffers b;
  order_plan(a->n, ws, &b);
  TMP_REQUIRE(workspace && !ws.overflow(), TMP_EWORKSPACE, "tmp_hash_partition: workspace too small");
  TMP_LAUNCH(owner_kernel, dim3(grid_for(a->n, kBlock * 8, 2048)), dim3(kBlock), st, a->key, a->n,
             (uint32_t)a->n_parts, b.comp[0], b.idx[0], (unsigned long long *)a->out_counts);
  OrderResult res;
  TMP_TRY(order_radix_passes(b, a->n, 0, bits_for((uint64_t)a->n_parts - 1), &res, st));
  TMP_LAUNCH(copy_u32_i32_kernel, dim3(grid_for(a->n, kBlock * 4, 8192)), dim3(kBlock), st, res.perm,
             a->out_perm, a->n);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}
