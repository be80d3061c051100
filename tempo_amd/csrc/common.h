// Shared host/device helpers of libtempo_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "tempo_hip.h"

namespace tmp {

void set_error(const char *fmt, ...);
void clear_error();

#define TMP_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::tmp::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return TMP_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

#define TMP_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::tmp::set_error(__VA_ARGS__);  \
      return (code);                  \
    }                                 \
  } while (0)

#define TMP_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != TMP_OK) return rc_; \
  } while (0)

// Bump allocator over the caller's workspace.  With base == nullptr it only measures, so the
// same planning code serves tmp_*_workspace_bytes and the real call.
struct Workspace {
  char *base;
  size_t size;
  size_t off;
  Workspace(void *b, size_t s) : base((char *)b), size(s), off(0) {}
  template <typename T>
  T *take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    size_t at = off;
    off += bytes;
    if (!base) return nullptr;
    if (off > size) return nullptr;
    return (T *)(base + at);
  }
  bool measuring() const { return base == nullptr; }
  bool overflow() const { return base != nullptr && off > size; }
};

inline int bits_for(uint64_t max_value) {  // bits needed to hold values in [0, max_value]
  int b = 0;
  while (max_value) { ++b; max_value >>= 1; }
  return b;
}

// Bit layout of the 64-bit sort word ("composite"), most significant field first:
//   [ key id | primary order key - prim_min | sequence field | low bits ]
// primary = ts, or floor(ts / prim_div); sequence field = 0 for NULL, seq - seq_min + 1
// otherwise (NULLS FIRST); the low bits are carried through the sort but never compared.
// prim_gran > 1: every primary value is a multiple of it (microsecond columns that hold whole
// milliseconds or seconds) and the field stores (primary - prim_min) / prim_gran -- 10 or 20
// bits less to sort.
struct Layout {
  int low_bits;
  int seq_bits;
  int prim_bits;
  int key_bits;
  int64_t prim_min;
  int64_t prim_max;
  int64_t seq_min;
  int64_t prim_div;
  int64_t prim_gran;
  // Wide tables.  seq_wide: the sequence values do not fit next to key and time; the word then
  // keeps one bit for them (0 = NULL, which sorts first, 1 = a value) and the ordering step sorts
  // by the values themselves in passes of their own (seq_full_bits wide) before the word's.
  // ext_masks > 0: the validity bits of that many payload columns do not fit either; low_bits is
  // 0 and they are handed to the consumers as a separate array in sorted order.
  int seq_wide;
  int seq_full_bits;
  int ext_masks;
  __host__ __device__ int seq_shift() const { return low_bits; }
  __host__ __device__ int prim_shift() const { return low_bits + seq_bits; }
  __host__ __device__ int key_shift() const { return low_bits + seq_bits + prim_bits; }
  __host__ __device__ int total_bits() const { return low_bits + seq_bits + prim_bits + key_bits; }
  __host__ __device__ uint64_t prim_mask() const {
    return prim_bits >= 64 ? ~0ull : ((1ull << prim_bits) - 1);
  }
  __device__ int64_t prim_of(uint64_t comp) const {
    return prim_min + (int64_t)((comp >> prim_shift()) & prim_mask()) * prim_gran;
  }
  __device__ uint32_t key_of(uint64_t comp) const {
    return key_bits == 0 ? 0u : (uint32_t)(comp >> key_shift());
  }
};

// Values the ordering step reads back from the device before it can lay out the sort word.
struct RangeInfo {
  int64_t prim_min;
  int64_t prim_max;
  int64_t seq_min;
  int64_t seq_max;
  int32_t unsorted;       // rows are not in (key, primary, seq) order
  int32_t time_unsorted;  // rows are not even in (primary, seq) order, keys ignored
  int32_t not_thousands;  // some primary value is not a multiple of 1000
  int32_t not_millions;   // ... of 1000000
  // rows at which (primary, seq) steps back: a table that is a handful of time-ordered runs laid
  // end to end (what a rank holds after the key shuffle) has few of them.  n_desc counts them while
  // it is <= kMaxDesc; desc_pos holds the first kMaxDesc positions found, in no particular order.
  int32_t n_desc;
  // windows of consecutive rows, looked at key by key: pairs of successive rows of one key seen,
  // and how many of them step back in (primary, seq).  Many pairs and no step back: every key's
  // rows are probably in order already (a feed of interleaved devices whose clocks are not aligned).
  int32_t key_pairs;
  int32_t key_viol;
  // set by a sample of adjacent rows taken before the scan: the rows look time-ordered, and the scan
  // took the form that also leaves the key column's histogram for the first radix pass
  int32_t key_hist;
  int64_t desc_pos[15];
};
constexpr int kMaxDesc = 15;   // at most kMaxDesc + 1 = 16 runs are merged; more: a full sort
constexpr int kMaxRuns = kMaxDesc + 1;

struct OrderSpec {
  int64_t n;
  const int32_t *key;
  int32_t n_keys;
  const int64_t *ts;
  int64_t prim_div;  // 1, or 1000000 for timestamp.cast("long")
  const int64_t *seq;
  const uint8_t *seq_valid;
  int n_masks;
  const uint8_t *masks[TMP_MAX_MASKS];
  int coarse_ok;  // the caller reads times only through Layout::prim_of: prim_gran may be used
  int ext_masks_ok;  // the caller can take the validity bits as a separate array (order_gather_masks)
};

struct OrderBuffers {
  uint64_t *comp[2];
  uint32_t *idx[2];
  uint32_t *counts;
  RangeInfo *range;  // device
  // bucketed ordering (see order.hip); all null when the table is too small for it
  unsigned long long *key_rows;  // [n_keys] sampled rows per key
  uint32_t *key_nb;              // [n_keys] buckets of the key
  uint32_t *key_base;            // [n_keys] first bucket of the key
  uint32_t *bucket_key;          // [max_buckets] key id of every bucket
  int32_t *fallback;             // device flag: some tile did not fit, the plain passes redo the job
  int64_t *tile_bounds;          // [tiles + 1] first bucket start at or after every tile's first row
  int64_t max_buckets;
  const int32_t *run_if;         // host-side switch: non-null while the stand-by passes are enqueued
  int n_runs;                    // sorted mode 3: the runs found by the range scan ...
  int64_t run_start[17];         // ... run r = rows [run_start[r], run_start[r + 1])
  int64_t *run_cuts;             // [tiles + 1][kMaxRuns] rows of every run before each output tile
  uint64_t *carry[2 * 4];        // [column][2] ping-pong copies of payload columns that travel with the rows
  int carry_cols;
  // host side: `counts` holds the histogram of (key & key_hist_mask) over key_hist_n rows in the block
  // geometry of the first pass that builds its own words (left there by the range scan); 0 = it does not
  int64_t key_hist_n;
  uint32_t key_hist_mask;
  const void *key_hist_of;  // the key column it was taken over
  // host side, set by the caller: sort words that fit 32 bits may be delivered as 32-bit words where
  // the ordering step has a form for it (OrderResult::narrow says whether they were)
  bool narrow_words;
};

struct OrderResult {
  const uint64_t *comp;  // sorted sort words
  const uint32_t *perm;  // perm[p] = input row at sorted position p
  bool narrow;           // comp holds the words' low 32 bits, one uint32 per row (their upper halves are zero)
};

// n_keys > 0 also reserves what the bucketed ordering of a large table needs.
// carry_cols: also two row-sized buffers per payload column that moves with the rows in every pass.
void order_plan(int64_t n, Workspace &ws, OrderBuffers *b, int64_t n_keys = 0, int carry_cols = 0);
// Size of the key dictionary as the ordering step sees it (a table without keys is one series).
inline int64_t plan_keys(const void *key, int64_t n_keys) { return key ? (n_keys > 0 ? n_keys : 1) : 1; }
int order_scan_ranges(const OrderSpec &s, OrderBuffers &b, hipStream_t st);
int order_read_ranges(OrderBuffers &b, RangeInfo *host, hipStream_t st);
int order_make_layout(const OrderSpec &s, const RangeInfo &r, int low_bits, Layout *out);
// sorted: 0 = full sort, 1 = rows already in (key, primary, seq) order (no passes), 2 = rows in
// (primary, seq) order with interleaved keys (a time-ordered feed): the stable passes over the
// key bits alone finish the job.
// Payload columns that ride along with the last radix pass (8-byte elements).
constexpr int kMaxPay = 4;
struct PayPtrs {
  int n;
  int positional;  // src is already in the order of the rows being moved (src[position], not src[row id])
  const uint64_t *src[kMaxPay];
  uint64_t *out[kMaxPay];
};
// pay_src / pay_out (optional, n_pay columns of 8-byte elements): delivered in sorted order
// along with the last radix pass, pay_out[c][p] = pay_src[c][perm[p]]; *carried reports
// whether that happened (it does not when the rows need no pass).
int order_build(const OrderSpec &s, const Layout &lay, int sorted, OrderBuffers &b,
                OrderResult *res, hipStream_t st, int n_pay = 0,
                const uint64_t *const *pay_src = nullptr, uint64_t *const *pay_out = nullptr,
                bool *carried = nullptr);
// 3 = a few time-ordered runs end to end: merged by time first, then as 2.
// 4 = every key's rows look ordered (sampled): the passes over the key bits are tried on their own
//     and the result is checked; the full sort stands by.
constexpr int kKeyPairsEvidence = 4096;
inline int sorted_mode(const RangeInfo &r) {
  if (!r.unsorted) return 1;
  if (!r.time_unsorted) return 2;
  if (r.n_desc >= 1 && r.n_desc <= kMaxDesc) return 3;
  return (r.key_pairs >= kKeyPairsEvidence && r.key_viol == 0) ? 4 : 0;
}
int order_radix_passes(OrderBuffers &b, int64_t n, int lo_bit, int hi_bit, OrderResult *res,
                       hipStream_t st, int n_pay = 0, const uint64_t *const *pay_src = nullptr,
                       uint64_t *const *pay_out = nullptr, bool *carried = nullptr);
// One-call form for ops that order a single table.
// known: the table's ranges as order_scan_ranges / order_read_ranges gave them for this very
// OrderSpec (NULL masks aside); the scan is then not repeated.
int order_single(const OrderSpec &s, OrderBuffers &b, Layout *lay, OrderResult *res,
                 hipStream_t st, const RangeInfo *known = nullptr);
// Orders one table and, where it can, delivers n_cols (1..kMaxPay) 8-byte columns in sorted order
// as well: xs[c][p] = cols[c][perm[p]], and for the columns that have a NULL mask vs[c][p] = its
// byte at that row (the masks ride in the low bits of the sort words and are spread out to bytes
// afterwards).  *traveled says whether that happened -- not when the rows were in order already
// (perm is then the identity) or the masks do not fit the word; the order itself is always built.
// `b` must have been planned with carry_cols >= n_cols for the columns to travel with a bucketed sort.
int order_columns(const OrderSpec &s0, OrderBuffers &b, int n_cols, const uint64_t *const *cols,
                  const uint8_t *const *valid, uint64_t *const *xs, uint8_t *const *vs, Layout *lay,
                  OrderResult *res, bool *traveled, hipStream_t st, const RangeInfo *known = nullptr);

// Layouts with ext_masks: out[p] = validity bits (bit m = mask m) of the row at sorted position p.
int order_gather_masks(const OrderSpec &s, const OrderResult &res, uint16_t *out, hipStream_t st);

// Per mask bit m < n_masks of the sort words: out[m*n + j] = last position <= j whose bit is set
// (reverse: first position >= j), or -1.  unit_agg: last_valid_scratch_ints() int32 of scratch.
size_t last_valid_scratch_ints();
// ext: the mask bits as a separate array (Layout::ext_masks) instead of the words' low bits.
int last_valid_scan(const uint64_t *comp, int64_t n, int n_masks, int32_t *unit_agg, int32_t *out,
                    bool reverse, hipStream_t st, const uint16_t *ext = nullptr, bool sparse = false);
// The scan works in kLastValidUnits slices ("units"); unit_agg is laid out [mask][kLastValidUnits].
constexpr int kLastValidUnits = 4096;
// In place, per mask m < n_masks: unit_agg[m][u] (last set position inside unit u, or -1) becomes the
// last set position before unit u, or -1 (u < units <= kLastValidUnits).
int last_valid_unit_carry(int32_t *unit_agg, int units, int n_masks, hipStream_t st);
// Exclusive scan of `total` uint32 counters in place.
int scan_u32_inplace(uint32_t *counts, int total, hipStream_t st);

// AS-OF join of two tables that are both in (key, ts[, seq]) order already (asof_sorted.hip): no
// sort words, no row ids; reads the argument block's columns directly.  bounds:
// asof_sorted_bounds_count(n_left) int64; tile_table: asof_sorted_table_ints(n_right, n_masks) int32;
// unit_agg: last_valid_scratch_ints() int32.
struct SkewParams;
size_t asof_sorted_bounds_count(int64_t n_left);
size_t asof_sorted_table_ints(int64_t n_right, int n_masks);
int asof_sorted_join(const tmp_asof_args *a, const SkewParams &sp, int64_t *bounds, int32_t *tile_table,
                     int32_t *unit_agg, hipStream_t st);

// floor(a / b), b > 0 expected (negative b falls through to the exact path).  For divisors up
// to 2^40 the quotient comes from a double-precision estimate that exact integer remainder steps
// correct -- a handful of instructions instead of a 64-bit integer division: one step where
// 0 <= a < 2^52 (the product is then within one of the quotient), two over the rest of int64.
__device__ __forceinline__ int64_t floordiv_i64(int64_t a, int64_t b) {
  if (b >= 2 && b <= (1ll << 40)) {
    const double inv = 1.0 / (double)b;
    if ((uint64_t)a < (1ull << 52)) {
      int64_t q = (int64_t)((double)a * inv);
      const int64_t r = a - q * b;
      if (r < 0) --q;
      else if (r >= b) ++q;
      return q;
    }
    // estimate: off by at most ~2^11 / b + 1 over the whole int64 range
    int64_t q = (int64_t)floor((double)a * inv);
    int64_t r = (int64_t)((uint64_t)a - (uint64_t)q * (uint64_t)b);  // exact: |r| < 2^52
    int64_t q2 = (int64_t)floor((double)r * inv);                     // off by at most 1
    q += q2;
    r -= q2 * b;
    if (r < 0) { --q; r += b; }
    if (r >= b) { ++q; }
    return q;
  }
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

// The sort word of a row without a sequence column, from its loaded values (key id, time, NULL
// mask bits): what make_word gives for plain (not bucketed) words.
__device__ __forceinline__ uint64_t plain_word(const Layout &lay, int64_t ts, uint32_t key, uint32_t mask_bits) {
  const int64_t p = lay.prim_div == 1 ? ts : floordiv_i64(ts, lay.prim_div);
  uint64_t rel = (uint64_t)(p - lay.prim_min);
  if (lay.prim_gran != 1) rel = (uint64_t)floordiv_i64((int64_t)rel, lay.prim_gran);
  uint64_t w = (rel << lay.prim_shift()) | mask_bits;
  if (lay.key_bits) w |= (uint64_t)key << lay.key_shift();
  return w;
}

// Optional per-kernel timing (tmp_profile_* in tempo_hip.h): a pair of HIP events around every
// launch, recorded on the launch stream.  Off by default; costs nothing when off.
struct ProfScope {
  int slot;
  hipStream_t st;
  ProfScope(const char *name, hipStream_t s);
  ~ProfScope();
};

#define TMP_LAUNCH(kernel, grid, block, st, ...)                       \
  do {                                                                 \
    ::tmp::ProfScope prof_scope_(#kernel, st);                         \
    hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);       \
  } while (0)

constexpr int kBlock = 256;
inline int grid_for(int64_t n, int per_block, int max_grid = 1 << 20) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > max_grid) g = max_grid;
  return (int)g;
}

}  // namespace tmp
