// Latent statistics on device: the step right after the hot path in analysis.py.
//   veex_column_quantiles     np.quantile(z_mean, q, axis=0)                  reference analysis.py:243-244,298
//   veex_band_outside_counts  the TP / FP counting loops of histogram_25_75 and auc_roc
//                                                                             reference analysis.py:255-264,323-333
// z is the [N][L] fp32 matrix the encoder wrote (row-major, one column per latent unit).  HBM-bound integer /
// compare work: N*L*4 bytes are read once per radix pass (4 passes) resp. once per counting launch.
//
// Quantiles: exact order statistics by a most-significant-digit radix SELECT on the order-preserving integer
// image of the floats -- no sort, nothing allocated: one block per column keeps, for every requested rank, the
// key prefix found so far and a 256-bin LDS histogram of the next digit among the elements that share it.
// Ranks with the same prefix share one histogram.  The interpolation then follows NumPy 1.24 (the reference's
// pin, requirements.txt) operation for operation: fp32 difference of the two neighbours, float64 weight.
#include "veex_internal.h"

#pragma clang fp contract(off)   // NumPy rounds the product and the sum separately: no FMA in the interpolation

#define LS_MAX_RANKS 48
#define LS_THREADS 1024

struct LSRanks {
  int nr;                       // distinct ranks
  int rank[LS_MAX_RANKS];       // ascending
};
struct LSLerp {                 // per quantile: the two neighbours (indices into the rank list) and the weight
  int nq;
  int lo[LS_MAX_RANKS / 2], hi[LS_MAX_RANKS / 2];
  double gamma[LS_MAX_RANKS / 2];
};

__device__ __forceinline__ unsigned ls_key(float f) {   // monotone float -> unsigned
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ls_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// order[r*L + col] = value of rank r (0-based, ascending) in column col
__global__ __launch_bounds__(LS_THREADS) void column_select_kernel(const float* __restrict__ z, int N, int L,
                                                                  const LSRanks rk, float* __restrict__ order) {
  __shared__ unsigned hist[LS_MAX_RANKS][256];
  __shared__ unsigned prefix[LS_MAX_RANKS];
  __shared__ int remain[LS_MAX_RANKS];     // rank inside the group of elements sharing the prefix
  __shared__ int group[LS_MAX_RANKS];      // first rank index with the same prefix (its histogram is the shared one)
  __shared__ int ngroups_s;
  __shared__ int glist[LS_MAX_RANKS];      // distinct group leaders
  const int col = blockIdx.x, tid = threadIdx.x;
  const int nr = rk.nr;
  if (tid < nr) { prefix[tid] = 0u; remain[tid] = rk.rank[tid]; }
  __syncthreads();

  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid == 0) {   // group ranks by prefix (ranks are ascending, so equal prefixes are adjacent)
      int ng = 0;
      for (int r = 0; r < nr; ++r) {
        if (r > 0 && prefix[r] == prefix[r - 1]) group[r] = group[r - 1];
        else { group[r] = r; glist[ng++] = r; }
      }
      ngroups_s = ng;
    }
    for (int i = tid; i < nr * 256; i += LS_THREADS) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const int ng = ngroups_s;
    const unsigned himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < N; i += LS_THREADS) {
      const unsigned k = ls_key(z[(long long)i * L + col]);
      const unsigned hi = k & himask;
      for (int g = 0; g < ng; ++g) {
        const int r = glist[g];
        if (hi == prefix[r]) { atomicAdd(&hist[r][(k >> shift) & 255u], 1u); break; }   // prefixes are distinct
      }
    }
    __syncthreads();
    if (tid < nr) {
      const unsigned* h = hist[group[tid]];
      int rem = remain[tid], b = 0;
      for (; b < 255; ++b) {
        const int c = (int)h[b];
        if (rem < c) break;
        rem -= c;
      }
      prefix[tid] |= (unsigned)b << shift;
      remain[tid] = rem;
    }
    __syncthreads();
  }
  if (tid < nr) order[(long long)tid * L + col] = ls_unkey(prefix[tid]);
}

// NumPy 1.24 _lerp on fp32 neighbours with a float64 weight: a + (b - a)*t, and b - (b - a)*(1 - t) where t >= 0.5
__global__ void quantile_lerp_kernel(const float* __restrict__ order, const LSLerp lp, int L, double* __restrict__ out) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lp.nq * L) return;
  const int q = i / L, col = i - q * L;
  const float a = order[(long long)lp.lo[q] * L + col], b = order[(long long)lp.hi[q] * L + col];
  // (plain operators under `fp contract(off)`: the __dmul_rn / __dadd_rn wrappers are ordinary * and + to the
  //  compiler and were fused into v_fma_f64)
  const float d = b - a;
  const double t = lp.gamma[q];
  const double dd = (double)d;
  double v;
  if (t >= 0.5) { const double u = 1.0 - t; const double pr = dd * u; v = (double)b - pr; }
  else          { const double pr = dd * t; v = (double)a + pr; }
  out[i] = v;
}

// np.median of a float32 column (analysis.py:576): the mean of the two middle order statistics, formed like
// np.mean of a 2-element float32 array -- fp32 sum, then the division by 2; the middle element itself when N is odd
__global__ void median_pair_kernel(const float* __restrict__ order, int two, int L, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= L) return;
  const float a = order[col];
  if (!two) { out[col] = a; return; }
  const float b = order[L + col];
  const float sum = a + b;
  out[col] = sum / 2.0f;
}

// counts[col*nb + k] = #{ i : z[i][col] <= lo[col*nb + k]  or  z[i][col] >= hi[col*nb + k] }   (float64 compares)
__global__ __launch_bounds__(256) void band_outside_kernel(const float* __restrict__ z, int N, int L, int nb,
                                                           const double* __restrict__ lo, const double* __restrict__ hi,
                                                           int* __restrict__ counts) {
  const int col = blockIdx.x, k = blockIdx.y;
  const double l = lo[col * nb + k], h = hi[col * nb + k];
  int c = 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const double v = (double)z[(long long)i * L + col];
    c += (v <= l || v >= h) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[col * nb + k] = s[0] + s[1] + s[2] + s[3];
}

extern "C" {

long long veex_column_quantiles_scratch_bytes(int nq, int L) {
  if (nq < 1 || L < 1) return -1;
  return (long long)2 * nq * L * 4;   // the order statistics [<= 2*nq][L] floats
}

// q: HOST array of nq quantile orders in [0, 1]; out: DEVICE [nq][L] float64; scratch: DEVICE bytes as above.
int veex_column_quantiles(const float* z, int N, int L, const double* q, int nq, double* out, void* scratch,
                          hipStream_t st) {
  if (z == nullptr || q == nullptr || out == nullptr || scratch == nullptr || N < 1 || L < 1 || nq < 1) {
    veex_set_error("column_quantiles: bad arguments"); return -1;
  }
  if (2 * nq > LS_MAX_RANKS) { veex_set_error("column_quantiles: at most %d quantiles per call", LS_MAX_RANKS / 2); return -1; }
  // NumPy 'linear': virtual index q*(n-1), neighbours floor and floor+1 (clipped), weight = fractional part
  LSLerp lp; lp.nq = nq;
  LSRanks rk; rk.nr = 0;
  int all[LS_MAX_RANKS];
  for (int i = 0; i < nq; ++i) {
    if (!(q[i] >= 0.0 && q[i] <= 1.0)) { veex_set_error("column_quantiles: q[%d] outside [0, 1]", i); return -1; }
    const double v = q[i] * (double)(N - 1);
    double f = (double)(long long)v;   // floor (v >= 0)
    long long p = (long long)f, nx = p + 1;
    if (p > N - 1) p = N - 1;
    if (nx > N - 1) nx = N - 1;
    lp.gamma[i] = v - f;
    all[2 * i] = (int)p; all[2 * i + 1] = (int)nx;
  }
  // distinct ascending ranks
  int sorted[LS_MAX_RANKS];
  for (int i = 0; i < 2 * nq; ++i) sorted[i] = all[i];
  for (int i = 1; i < 2 * nq; ++i) { const int v = sorted[i]; int j = i - 1; while (j >= 0 && sorted[j] > v) { sorted[j + 1] = sorted[j]; --j; } sorted[j + 1] = v; }
  for (int i = 0; i < 2 * nq; ++i) if (rk.nr == 0 || sorted[i] != rk.rank[rk.nr - 1]) rk.rank[rk.nr++] = sorted[i];
  for (int i = 0; i < nq; ++i) {
    for (int r = 0; r < rk.nr; ++r) { if (rk.rank[r] == all[2 * i]) lp.lo[i] = r; if (rk.rank[r] == all[2 * i + 1]) lp.hi[i] = r; }
  }
  float* order = static_cast<float*>(scratch);
  hipLaunchKernelGGL(column_select_kernel, dim3(L), dim3(LS_THREADS), 0, st, z, N, L, rk, order);
  VEEX_CHECK_LAUNCH();
  hipLaunchKernelGGL(quantile_lerp_kernel, dim3((nq * L + 255) / 256), dim3(256), 0, st, order, lp, L, out);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// np.median(z[:, j]) for every column j (NaN-free input): out DEVICE [L] fp32; scratch as for one quantile.
int veex_column_medians(const float* z, int N, int L, float* out, void* scratch, hipStream_t st) {
  if (z == nullptr || out == nullptr || scratch == nullptr || N < 1 || L < 1) { veex_set_error("column_medians: bad arguments"); return -1; }
  LSRanks rk;
  rk.rank[0] = (N - 1) / 2; rk.rank[1] = N / 2;
  rk.nr = (rk.rank[0] == rk.rank[1]) ? 1 : 2;
  float* order = static_cast<float*>(scratch);
  hipLaunchKernelGGL(column_select_kernel, dim3(L), dim3(LS_THREADS), 0, st, z, N, L, rk, order);
  VEEX_CHECK_LAUNCH();
  hipLaunchKernelGGL(median_pair_kernel, dim3((L + 255) / 256), dim3(256), 0, st, order, rk.nr == 2, L, out);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_band_outside_counts(const float* z, int N, int L, int nb, const double* lo, const double* hi, int* counts,
                             hipStream_t st) {
  if (z == nullptr || lo == nullptr || hi == nullptr || counts == nullptr || N < 0 || L < 1 || nb < 1) {
    veex_set_error("band_outside_counts: bad arguments"); return -1;
  }
  hipLaunchKernelGGL(band_outside_kernel, dim3(L, nb), dim3(256), 0, st, z, N, L, nb, lo, hi, counts);
  VEEX_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
