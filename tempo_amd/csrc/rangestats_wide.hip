// withRangeStats for frames of any length (python/tempo/tsdf.py:673-721).  The region kernel of
// rangestats.hip serves frames that fit a block's LDS region; rows whose frame reaches further
// are marked there and completed here in O(1) random reads per row, however long the frame:
//
//   * the ordered rows are cut into chunks of 1024; one pass writes, per row, the inclusive
//     within-chunk prefixes of (x - pivot), (x - pivot)^2, the counts, the wrapping integer
//     sum, and the prefix / suffix minima and maxima of its chunk; per chunk, its totals;
//   * chunk totals get exclusive prefixes inside superblocks of 64 chunks (pivot = a value of
//     the superblock, so prefix differences stay well conditioned), chunk minima / maxima a
//     sparse table over all chunks;
//   * a marked row finds its frame [A, B] by binary search in the ordered sort words and
//     assembles it from the tail of A's chunk, the head of B's chunk and the whole chunks in
//     between, one (count, mean, M2) piece per superblock touched, merged with the pairwise
//     update of the variance (no subtraction of large sums across pivots).
// Frames that hold NaN / +-inf values still take the exact row-sequential evaluation.
#include <string.h>

#include "rangestats_common.h"

namespace tmp {

constexpr int WK_CHUNK = 1024;  // rows per chunk = threads per block
constexpr int WK_SUPER = 64;    // chunks per superblock

template <typename T>
struct WideOps;
template <>
struct WideOps<double> {
  static constexpr bool kInt = false;
  // NULL slots and identities are NaN: fmin / fmax return the other operand
  static __device__ __forceinline__ double ident_min() { return __builtin_nan(""); }
  static __device__ __forceinline__ double ident_max() { return __builtin_nan(""); }
  static __device__ __forceinline__ double mn(double a, double b) { return fmin(a, b); }
  static __device__ __forceinline__ double mx(double a, double b) { return fmax(a, b); }
  static __device__ __forceinline__ uint64_t bits(double v) { return (uint64_t)__double_as_longlong(v); }
  static __device__ __forceinline__ double val(uint64_t b) { return __longlong_as_double((long long)b); }
};
template <>
struct WideOps<int64_t> {
  static constexpr bool kInt = true;
  static __device__ __forceinline__ int64_t ident_min() { return INT64_MAX; }
  static __device__ __forceinline__ int64_t ident_max() { return INT64_MIN; }
  static __device__ __forceinline__ int64_t mn(int64_t a, int64_t b) { return a < b ? a : b; }
  static __device__ __forceinline__ int64_t mx(int64_t a, int64_t b) { return a > b ? a : b; }
  static __device__ __forceinline__ uint64_t bits(int64_t v) { return (uint64_t)v; }
  static __device__ __forceinline__ int64_t val(uint64_t b) { return (int64_t)b; }
};

void rs_wide_plan(int64_t n, bool is_i64, Workspace &ws, RsWidePlan *p) {
  size_t rows = (size_t)(n > 0 ? n : 1);
  int64_t chunks = (int64_t)((rows + WK_CHUNK - 1) / WK_CHUNK);
  int levels = 1;
  while ((1ll << levels) <= chunks) ++levels;  // table levels 0 .. levels-1
  p->chunks = chunks;
  p->levels = levels;
  p->ps = ws.take<double>(rows);
  p->pq = ws.take<double>(rows);
  p->pc = ws.take<uint32_t>(rows);
  p->pi = ws.take<uint64_t>(is_i64 ? rows : 1);
  p->pmn = ws.take<uint64_t>(rows);
  p->smn = ws.take<uint64_t>(rows);
  p->pmx = ws.take<uint64_t>(rows);
  p->smx = ws.take<uint64_t>(rows);
  size_t nc = (size_t)chunks;
  p->tot_s = ws.take<double>(nc);
  p->tot_q = ws.take<double>(nc);
  p->tot_c = ws.take<uint64_t>(nc);
  p->tot_i = ws.take<uint64_t>(nc);
  p->exc_s = ws.take<double>(nc);
  p->exc_q = ws.take<double>(nc);
  p->exc_c = ws.take<uint64_t>(nc);
  p->exc_i = ws.take<uint64_t>(nc);
  p->pivot = ws.take<double>((nc + WK_SUPER - 1) / WK_SUPER);
  p->cmean = ws.take<double>(nc);
  p->cm2 = ws.take<double>(nc);
  p->tab_mn = ws.take<uint64_t>(nc * (size_t)levels);
  p->tab_mx = ws.take<uint64_t>(nc * (size_t)levels);
}

// pivot[s] = the first finite non-NULL value of superblock s (0 if it holds none, and then no
// value of it is ever added up): one wave per superblock.  The search goes on past the first
// chunk: after a long run of NULLs a pivot of 0 would leave sums of squares of the values
// themselves, and with them the variance of values far from 0, to cancellation.
template <typename T>
__global__ __launch_bounds__(256) void wide_pivot_kernel(const uint64_t *comp, const T *xs, int64_t n,
                                                         int64_t supers, double *pivot, int vbit) {
  int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (s >= supers) return;
  int64_t base = s * (int64_t)WK_SUPER * WK_CHUNK;
  double pv = 0.0;
  bool found = false;
  for (int r = 0; r < WK_SUPER * WK_CHUNK / 64 && !found; ++r) {
    int64_t i = base + (int64_t)r * 64 + lane;
    if (base + (int64_t)r * 64 >= n) break;
    bool ok = false;
    T v = (T)0;
    if (i < n && ((comp[i] >> vbit) & 1ull)) {
      v = xs[i];
      ok = Ord<T>::finite(v);
    }
    uint64_t bal = __ballot(ok);
    if (bal) {
      int src = (int)__ffsll((long long)bal) - 1;
      pv = __shfl((double)v, src, 64);
      found = true;
    }
  }
  if (lane == 0) pivot[s] = pv;
}

// Inclusive scan over the 1024 threads of a block (forward, or backward with REV), operation
// given by `op`; `lds` holds 16 entries.  Every thread calls it.
template <bool REV, typename V, typename Op>
__device__ __forceinline__ V block_scan(V v, Op op, V *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    V t = REV ? __shfl_down(v, o, 64) : __shfl_up(v, o, 64);
    bool take = REV ? lane + o < 64 : lane >= o;
    if (take) v = op(t, v);
  }
  __syncthreads();  // lds may still be read by the previous scan
  if (lane == (REV ? 0 : 63)) lds[wave] = v;
  __syncthreads();
  if (REV) {
    for (int wv = 15; wv > wave; --wv) v = op(lds[wv], v);
  } else {
    for (int wv = 0; wv < wave; ++wv) v = op(lds[wv], v);
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(WK_CHUNK) void wide_chunk_kernel(const uint64_t *comp, const T *xs, int64_t n,
                                                              RsWidePlan p, int vbit) {
  using W = WideOps<T>;
  __shared__ double lds_d[16];
  __shared__ uint64_t lds_u[16];
  __shared__ uint32_t lds_c[16];
  __shared__ T lds_t[16];
  __shared__ double chunk_mean_s;
  __shared__ double dev_sq_s[WK_CHUNK / 64];
  const int64_t c = blockIdx.x;
  const int64_t i = c * WK_CHUNK + threadIdx.x;
  const bool in = i < n;
  bool ok = false, fin = false;
  T v = (T)0;
  if (in && ((comp[i] >> vbit) & 1ull)) {
    ok = true;
    v = xs[i];
    fin = Ord<T>::finite(v);
  }
  const double pv = p.pivot[c / WK_SUPER];
  const double d = fin ? (double)v - pv : 0.0;
  auto addd = [](double a, double b) { return a + b; };
  auto addu = [](uint64_t a, uint64_t b) { return a + b; };
  auto addc = [](uint32_t a, uint32_t b) { return a + b; };
  auto mnop = [](T a, T b) { return W::mn(a, b); };
  auto mxop = [](T a, T b) { return W::mx(a, b); };
  const double ps = block_scan<false>(d, addd, lds_d);
  const double pq = block_scan<false>(d * d, addd, lds_d);
  const uint32_t pc = block_scan<false>((ok ? 1u : 0u) | ((ok && !fin) ? 0x10000u : 0u), addc, lds_c);
  uint64_t pi = 0;
  if (W::kInt) pi = block_scan<false>(ok ? (uint64_t)v : (uint64_t)0, addu, lds_u);
  // only finite values compete for min / max (frames with others are evaluated row by row)
  const T vmn = fin ? v : W::ident_min(), vmx = fin ? v : W::ident_max();
  const T pmn = block_scan<false>(vmn, mnop, lds_t);
  const T smn = block_scan<true>(vmn, mnop, lds_t);
  const T pmx = block_scan<false>(vmx, mxop, lds_t);
  const T smx = block_scan<true>(vmx, mxop, lds_t);
  if (in) {
    p.ps[i] = ps;
    p.pq[i] = pq;
    p.pc[i] = pc;
    if (W::kInt) p.pi[i] = pi;
    p.pmn[i] = W::bits(pmn);
    p.smn[i] = W::bits(smn);
    p.pmx[i] = W::bits(pmx);
    p.smx[i] = W::bits(smx);
  }
  // the chunk's last row holds its totals
  const int64_t last = (c + 1) * WK_CHUNK < n ? (c + 1) * WK_CHUNK - 1 : n - 1;
  // the chunk's own mean, and the squared deviations from it added up around it (no large sums are
  // subtracted): what a frame whose spread is tiny against the prefixes is put together from
  const uint32_t n_fin = (pc & 0xffffu) - (pc >> 16);
  if (i == last) chunk_mean_s = n_fin ? ps / (double)n_fin : 0.0;
  __syncthreads();
  const double dev = fin ? d - chunk_mean_s : 0.0;
  // (totals only: one reduction per wave and sixteen partial sums, not two more scans of the block)
  double dev1 = dev, dev2 = dev * dev;
  for (int o = 32; o > 0; o >>= 1) {
    dev1 += __shfl_xor(dev1, o, 64);
    dev2 += __shfl_xor(dev2, o, 64);
  }
  __syncthreads();  // lds_d may still be read by the last scan
  if ((threadIdx.x & 63) == 0) { lds_d[threadIdx.x >> 6] = dev1; dev_sq_s[threadIdx.x >> 6] = dev2; }
  __syncthreads();
  if (i == last) {
    dev1 = 0.0; dev2 = 0.0;
    for (int wv = 0; wv < WK_CHUNK / 64; ++wv) { dev1 += lds_d[wv]; dev2 += dev_sq_s[wv]; }
    const double nf = (double)n_fin;
    p.cmean[c] = n_fin ? chunk_mean_s + dev1 / nf : 0.0;  // as an offset from the superblock's pivot: small, hence finely resolved
    p.cm2[c] = n_fin ? dev2 - dev1 * dev1 / nf : 0.0;
    p.tot_s[c] = ps;
    p.tot_q[c] = pq;
    p.tot_c[c] = (uint64_t)(pc & 0xffffu) | ((uint64_t)(pc >> 16) << 32);
    p.tot_i[c] = pi;
    p.tab_mn[c] = W::bits(pmn);
    p.tab_mx[c] = W::bits(pmx);
  }
}

// Exclusive prefix of the chunk totals inside each superblock: one wave per superblock.
__global__ __launch_bounds__(256) void wide_super_kernel(RsWidePlan p, int64_t supers) {
  int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (s >= supers) return;
  int64_t c = s * WK_SUPER + lane;
  bool in = c < p.chunks;
  double vs = in ? p.tot_s[c] : 0.0, vq = in ? p.tot_q[c] : 0.0;
  uint64_t vc = in ? p.tot_c[c] : 0ull, vi = in ? p.tot_i[c] : 0ull;
  double is = vs, iq = vq;
  uint64_t ic = vc, ii = vi;
  for (int o = 1; o < 64; o <<= 1) {
    double a = __shfl_up(is, o, 64), b = __shfl_up(iq, o, 64);
    uint64_t cc = __shfl_up(ic, o, 64), dd = __shfl_up(ii, o, 64);
    if (lane >= o) { is += a; iq += b; ic += cc; ii += dd; }
  }
  // (exclusive = the lane below's inclusive value; "inclusive - own" would round a small prefix away
  // where the chunk's own total is far larger)
  double es = __shfl_up(is, 1, 64), eq = __shfl_up(iq, 1, 64);
  if (lane == 0) { es = 0.0; eq = 0.0; }
  if (in) {
    p.exc_s[c] = es;
    p.exc_q[c] = eq;
    p.exc_c[c] = ic - vc;
    p.exc_i[c] = ii - vi;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void wide_table_kernel(uint64_t *tab_mn, uint64_t *tab_mx, int64_t chunks,
                                                         int lv) {
  using W = WideOps<T>;
  const int64_t span = 1ll << lv;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c + span <= chunks; c += stride) {
    const uint64_t *pm = tab_mn + (size_t)(lv - 1) * chunks, *px = tab_mx + (size_t)(lv - 1) * chunks;
    tab_mn[(size_t)lv * chunks + c] = W::bits(W::mn(W::val(pm[c]), W::val(pm[c + span / 2])));
    tab_mx[(size_t)lv * chunks + c] = W::bits(W::mx(W::val(px[c]), W::val(px[c + span / 2])));
  }
}

// One piece of a frame into the running (count, mean, M2): the pairwise variance update.  The
// running mean is kept as an offset from the first piece's pivot (`ref`): the difference of two
// pieces' means is then formed from small numbers (pivots of neighbouring superblocks subtract
// without rounding), not from two means that each carry the rounding of the values' own level.
__device__ __forceinline__ void piece_merge(double n_p, double s_p, double q_p, double pivot_p, double &N,
                                            double &ref, double &rmean, double &M2, double &sum) {
  if (n_p <= 0.0) return;
  double m2_p = q_p - s_p * s_p / n_p;
  if (m2_p < 0.0) m2_p = 0.0;
  sum += n_p * pivot_p + s_p;
  if (N == 0.0) {
    N = n_p; ref = pivot_p; rmean = s_p / n_p; M2 = m2_p;
    return;
  }
  const double mean_p = (pivot_p - ref) + s_p / n_p;
  const double delta = mean_p - rmean, nn = N + n_p;
  rmean += delta * (n_p / nn);
  M2 += m2_p + delta * delta * (N * n_p / nn);
  N = nn;
}

template <typename T>
__global__ __launch_bounds__(256) void wide_rows_kernel(const uint64_t *comp, const T *xs, int64_t n,
                                                        Layout lay, uint64_t w, RsOut<T> o, RsWidePlan p) {
  using W = WideOps<T>;
  const int low = lay.low_bits;
  const uint64_t pmask = lay.prim_mask();
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (o.valid[i] != kDeferred) continue;
    const uint64_t wi = comp[i];
    const uint64_t k = wi >> low;
    const uint64_t kstart = frame_start_key(k, pmask, w);
    const int64_t A = lower_bound_k(comp, 0, i, kstart, low);
    const int64_t B = upper_bound_k(comp, i, n, k, low) - 1;
    const int64_t cA = A / WK_CHUNK, cB = B / WK_CHUNK;
    const bool a_first = (A % WK_CHUNK) == 0;
    double N = 0.0, ref = 0.0, rmean = 0.0, M2 = 0.0, dsum = 0.0;
    double qmag = 0.0;  // size of the prefix values of sum((x - pivot)^2) that the frame's M2 was taken from
    uint64_t cnt = 0, nonfin = 0, isum = 0;
    T mn = W::ident_min(), mx = W::ident_max();
    // everything of chunk c up to and including row r, minus everything before row `from`
    if (cA == cB) {
      const uint32_t c1 = p.pc[B], c0 = a_first ? 0u : p.pc[A - 1];
      cnt = (c1 & 0xffffu) - (c0 & 0xffffu);
      nonfin = (c1 >> 16) - (c0 >> 16);
      const double s = p.ps[B] - (a_first ? 0.0 : p.ps[A - 1]);
      const double q = p.pq[B] - (a_first ? 0.0 : p.pq[A - 1]);
      qmag += p.pq[B];
      if (W::kInt) isum = p.pi[B] - (a_first ? 0ull : p.pi[A - 1]);
      piece_merge((double)(cnt - nonfin), s, q, p.pivot[cA / WK_SUPER], N, ref, rmean, M2, dsum);
      if (!nonfin)
        for (int64_t j = A; j <= B; ++j)
          if ((comp[j] >> o.vbit) & 1ull) { T v = xs[j]; mn = W::mn(mn, v); mx = W::mx(mx, v); }
    } else {
      {  // head: rows A .. end of chunk cA
        const uint64_t tc = p.tot_c[cA];
        const uint32_t c0 = a_first ? 0u : p.pc[A - 1];
        const uint64_t hv = (tc & 0xffffffffull) - (c0 & 0xffffu), hn = (tc >> 32) - (c0 >> 16);
        cnt += hv; nonfin += hn;
        if (W::kInt) isum += p.tot_i[cA] - (a_first ? 0ull : p.pi[A - 1]);
        qmag += p.tot_q[cA];
        piece_merge((double)(hv - hn), p.tot_s[cA] - (a_first ? 0.0 : p.ps[A - 1]),
                    p.tot_q[cA] - (a_first ? 0.0 : p.pq[A - 1]), p.pivot[cA / WK_SUPER], N, ref, rmean, M2, dsum);
        mn = W::mn(mn, W::val(p.smn[A]));
        mx = W::mx(mx, W::val(p.smx[A]));
      }
      const int64_t m0 = cA + 1, m1 = cB - 1;
      if (m0 <= m1) {  // whole chunks, one piece per superblock touched
        for (int64_t s = m0 / WK_SUPER; s <= m1 / WK_SUPER; ++s) {
          const int64_t lo = m0 > s * WK_SUPER ? m0 : s * WK_SUPER;
          const int64_t hi = m1 < s * WK_SUPER + WK_SUPER - 1 ? m1 : s * WK_SUPER + WK_SUPER - 1;
          const uint64_t pc = p.exc_c[hi] + p.tot_c[hi] - p.exc_c[lo];
          const uint64_t pv = pc & 0xffffffffull, pn = pc >> 32;
          cnt += pv; nonfin += pn;
          if (W::kInt) isum += p.exc_i[hi] + p.tot_i[hi] - p.exc_i[lo];
          qmag += p.exc_q[hi] + p.tot_q[hi];
          piece_merge((double)(pv - pn), p.exc_s[hi] + p.tot_s[hi] - p.exc_s[lo],
                      p.exc_q[hi] + p.tot_q[hi] - p.exc_q[lo], p.pivot[s], N, ref, rmean, M2, dsum);
        }
        const int lv = 63 - __clzll((long long)(m1 - m0 + 1));
        const uint64_t *tm = p.tab_mn + (size_t)lv * p.chunks, *tx = p.tab_mx + (size_t)lv * p.chunks;
        mn = W::mn(mn, W::mn(W::val(tm[m0]), W::val(tm[m1 + 1 - (1ll << lv)])));
        mx = W::mx(mx, W::mx(W::val(tx[m0]), W::val(tx[m1 + 1 - (1ll << lv)])));
      }
      {  // tail: start of chunk cB .. B
        const uint32_t c1 = p.pc[B];
        const uint64_t tv = c1 & 0xffffu, tn = c1 >> 16;
        cnt += tv; nonfin += tn;
        if (W::kInt) isum += p.pi[B];
        qmag += p.pq[B];
        piece_merge((double)(tv - tn), p.ps[B], p.pq[B], p.pivot[cB / WK_SUPER], N, ref, rmean, M2, dsum);
        mn = W::mn(mn, W::val(p.pmn[B]));
        mx = W::mx(mx, W::val(p.pmx[B]));
      }
    }
    if (nonfin) {  // NaN / inf in the frame: Spark's own recurrence, value by value
      rs_row_serial<T>(comp, xs, A, B, i, o);
      continue;
    }
    T sum;
    if constexpr (W::kInt) sum = (T)isum;
    else sum = (T)dsum;
    const bool x_ok = ((wi >> o.vbit) & 1ull) != 0;
    const bool same = cnt > 1 && mn == mx;  // all values identical: Spark's m2 is exactly 0
    // The prefix differences carry an absolute error of a few ulps of the prefixes themselves.  A frame
    // whose spread is tiny against them (two nearly equal values of a sparse series; a metric that sits
    // on a level far from the superblock's pivot) is added up again around its own mean: the rows of
    // the chunks it covers in part one by one, the chunks it covers whole from their own deviations.
    if (cnt > 1 && !same && M2 < 1e-4 * qmag) {
      // everything as an offset from `ref` (the first piece's pivot): the values less ref and the
      // pivots less ref are differences of neighbours, formed without rounding; the centre is ref + rmean
      double acc = 0.0, lin = 0.0;
      for (int64_t c = cA; c <= cB; ++c) {
        const int64_t c0 = c * WK_CHUNK, c1 = c0 + WK_CHUNK - 1 < n - 1 ? c0 + WK_CHUNK - 1 : n - 1;
        const int64_t lo = A > c0 ? A : c0, hi = B < c1 ? B : c1;
        if (lo == c0 && hi == c1) {  // a whole chunk: its own mean and deviations, moved to the frame's mean
          const uint64_t tc = p.tot_c[c];
          const double nc = (double)((tc & 0xffffffffull) - (tc >> 32));
          const double dm = ((p.pivot[c / WK_SUPER] - ref) + p.cmean[c]) - rmean;
          acc += p.cm2[c] + nc * dm * dm;
          lin += nc * dm;
        } else {  // part of a chunk (the frame's two ends at most): row by row
          for (int64_t j = lo; j <= hi; ++j)
            if ((comp[j] >> o.vbit) & 1ull) {
              const double dv = ((double)xs[j] - ref) - rmean;
              acc += dv * dv;
              lin += dv;
            }
        }
      }
      M2 = acc - lin * lin / N;
      if (M2 < 0.0) M2 = 0.0;
    }
    const double sd = cnt > 1 ? (same ? 0.0 : sqrt(M2 / (N - 1.0))) : 0.0;
    rs_store<T>(o, i, (int64_t)cnt, cnt ? ref + rmean : 0.0, mn, mx, sum, cnt > 1, sd, x_ok, x_ok ? (double)xs[i] : 0.0);
  }
}

template <typename T>
int rs_wide_run(const uint64_t *comp, const T *xs, int64_t n, const Layout &lay, uint64_t w,
                const RsOut<T> &o, RsWidePlan &p, hipStream_t st) {
  const int64_t supers = (p.chunks + WK_SUPER - 1) / WK_SUPER;
  TMP_LAUNCH(wide_pivot_kernel<T>, dim3((unsigned)((supers + 3) / 4)), dim3(256), st, comp, xs, n, supers, p.pivot, o.vbit);
  TMP_LAUNCH(wide_chunk_kernel<T>, dim3((unsigned)p.chunks), dim3(WK_CHUNK), st, comp, xs, n, p, o.vbit);
  TMP_LAUNCH(wide_super_kernel, dim3((unsigned)((supers + 3) / 4)), dim3(256), st, p, supers);
  for (int lv = 1; lv < p.levels; ++lv)
    TMP_LAUNCH(wide_table_kernel<T>, dim3(grid_for(p.chunks, 256, 4096)), dim3(256), st, p.tab_mn, p.tab_mx,
               p.chunks, lv);
  TMP_LAUNCH(wide_rows_kernel<T>, dim3(grid_for(n, 256, 1 << 16)), dim3(256), st, comp, xs, n, lay, w, o, p);
  TMP_CHECK_HIP(hipGetLastError());
  return TMP_OK;
}

template int rs_wide_run<double>(const uint64_t *, const double *, int64_t, const Layout &, uint64_t,
                                 const RsOut<double> &, RsWidePlan &, hipStream_t);
template int rs_wide_run<int64_t>(const uint64_t *, const int64_t *, int64_t, const Layout &, uint64_t,
                                  const RsOut<int64_t> &, RsWidePlan &, hipStream_t);

}  // namespace tmp
