// BatchNormalization on BATCH statistics -- the alternate reading of models.py:152-153,222,265 behind
// veex_net_set_semantics(VEEX_BN_BATCH) (the reference never forwards `training`, so the default is the
// moving statistics; nothing in the reference tree can pin which of the two Keras 2.13 really does).  A correctness
// hedge, not a tuned path: plain HBM-bound passes, fp64 accumulation, deterministic (no atomics).
//   forward : per-channel mean / biased variance of the post-ReLU activation -> scale = gamma * rstd,
//             shift = beta - mean * scale for the consumers' affine prologue; moving statistics updated with
//             momentum 0.99 (unbiased variance for the 4-D layers, biased for dec.dense's 2-D layer);
//   backward: through the statistics, da = gamma * rstd * (v - mean(v) - xhat * mean(v * xhat)), then the ReLU mask;
//             dgamma = sum v * xhat, dbeta = sum v, dbias = sum of the masked da.
#include "veex_internal.h"

#define BNB_TX 64   // columns per block
#define BNB_TY 4    // row lanes per block

// partial[chunk][0][c] = sum_r v[r][c], partial[chunk][1][c] = sum_r v[r][c] * a[r][c] over the chunk's rows (fp64)
__global__ __launch_bounds__(BNB_TX * BNB_TY) void bnb_colstats_kernel(const float* __restrict__ v, const float* __restrict__ a,
                                                                     long long rows, int C, long long rows_per_chunk,
                                                                     double* __restrict__ partial) {
  __shared__ double red[2][BNB_TY][BNB_TX];
  const int tx = threadIdx.x & (BNB_TX - 1), ty = threadIdx.x / BNB_TX;
  const int c = blockIdx.y * BNB_TX + tx;
  const long long r0 = (long long)blockIdx.x * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (long long r = r0 + ty; r < r1; r += BNB_TY) {
      const float x = v[r * C + c], y = a[r * C + c];
      s1 += (double)x;
      s2 += (double)x * (double)y;
    }
  red[0][ty][tx] = s1; red[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < BNB_TY; ++k) { t1 += red[0][k][tx]; t2 += red[1][k][tx]; }
    partial[((long long)blockIdx.x * 2 + 0) * C + c] = t1;
    partial[((long long)blockIdx.x * 2 + 1) * C + c] = t2;
  }
}

__global__ __launch_bounds__(256) void bnb_fwd_finalize_kernel(const double* __restrict__ partial, int chunks, int C, double N,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, int unbiased, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ rstd,
                                                               float* __restrict__ mu, float* __restrict__ moving_mean,
                                                               float* __restrict__ moving_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < chunks; ++k) { s1 += partial[((long long)k * 2) * C + c]; s2 += partial[((long long)k * 2 + 1) * C + c]; }
  const double mean = s1 / N;
  double var = s2 / N - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * rs;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - mean * sc);
  rstd[c] = (float)rs;
  mu[c] = (float)mean;
  const double vu = (unbiased && N > 1.0) ? var * (N / (N - 1.0)) : var;
  moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
  moving_var[c] = (float)((double)moving_var[c] * momentum + vu * (1.0 - (double)momentum));
}

// coef[0][c] = gamma * rstd, coef[1][c] = mean(v), coef[2][c] = rstd * mean(v * xhat); dgamma = sum v * xhat, dbeta = sum v
__global__ __launch_bounds__(256) void bnb_bwd_finalize_kernel(const double* __restrict__ partial, int chunks, int C, double N,
                                                               const float* __restrict__ gamma, const float* __restrict__ mu,
                                                               const float* __restrict__ rstd, float* __restrict__ coef,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < chunks; ++k) { s1 += partial[((long long)k * 2) * C + c]; s2 += partial[((long long)k * 2 + 1) * C + c]; }
  const double rs = (double)rstd[c];
  const double sxh = (s2 - (double)mu[c] * s1) * rs;     // sum v * xhat
  coef[c] = (float)((double)gamma[c] * rs);
  coef[C + c] = (float)(s1 / N);
  coef[2 * C + c] = (float)(rs * sxh / N);
  if (dgamma != nullptr) dgamma[c] = (float)sxh;
  if (dbeta != nullptr) dbeta[c] = (float)s1;
}

// out = (a > 0) ? A * (v - c1 - (a - mu) * K) : 0 ;  partial[chunk][c] = column sums of out (the bias gradient)
__global__ __launch_bounds__(BNB_TX * BNB_TY) void bnb_bwd_apply_kernel(const float* __restrict__ v, const float* __restrict__ a,
                                                                      const float* __restrict__ mu, const float* __restrict__ coef,
                                                                      float* __restrict__ out, long long rows, int C,
                                                                      long long rows_per_chunk, double* __restrict__ partial) {
  __shared__ double red[BNB_TY][BNB_TX];
  const int tx = threadIdx.x & (BNB_TX - 1), ty = threadIdx.x / BNB_TX;
  const int c = blockIdx.y * BNB_TX + tx;
  const long long r0 = (long long)blockIdx.x * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  double s = 0.0;
  if (c < C) {
    const float A = coef[c], c1 = coef[C + c], K = coef[2 * C + c], m = mu[c];
    for (long long r = r0 + ty; r < r1; r += BNB_TY) {
      const float x = a[r * C + c];
      const float d = A * (v[r * C + c] - c1 - (x - m) * K);
      const float o = (x > 0.f) ? d : 0.f;
      out[r * C + c] = o;
      s += (double)o;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < BNB_TY; ++k) t += red[k][tx];
    partial[(long long)blockIdx.x * C + c] = t;
  }
}

__global__ __launch_bounds__(256) void bnb_sum_rows_kernel(const double* __restrict__ partial, int chunks, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += partial[(long long)k * C + c];
  out[c] = (float)s;
}

static void bnb_split(long long rows, int C, int& chunks, long long& per) {
  const int col_blocks = (C + BNB_TX - 1) / BNB_TX;
  long long want = 2048 / col_blocks;
  if (want < 1) want = 1;
  per = (rows + want - 1) / want;
  if (per < 16) per = 16;
  chunks = (int)((rows + per - 1) / per);
}

long long veex_bn_batch_scratch_doubles(long long rows, int C) {
  int chunks; long long per;
  bnb_split(rows, C, chunks, per);
  return (long long)chunks * 2 * C;
}

// act: [rows][C] post-ReLU activation.  scratch: veex_bn_batch_scratch_doubles(rows, C) doubles.
int veex_bn_batch_forward(const float* act, long long rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                          int unbiased, float* scale, float* shift, float* rstd, float* mu, float* moving_mean, float* moving_var,
                          double* scratch, hipStream_t st) {
  if (act == nullptr || rows < 1 || C < 1 || scratch == nullptr) { veex_set_error("bn_batch_forward: bad arguments"); return -1; }
  int chunks; long long per;
  bnb_split(rows, C, chunks, per);
  hipLaunchKernelGGL(bnb_colstats_kernel, dim3(chunks, (C + BNB_TX - 1) / BNB_TX), dim3(BNB_TX * BNB_TY), 0, st, act, act, rows, C, per, scratch);
  hipLaunchKernelGGL(bnb_fwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, chunks, C, (double)rows, gamma, beta, eps,
                     momentum, unbiased, scale, shift, rstd, mu, moving_mean, moving_var);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// v: [rows][C] gradient wrt the BatchNorm output; a: the activation; out (may alias v): d(pre-activation).
// coef: 3 * C floats of scratch.
int veex_bn_batch_backward(const float* v, const float* a, long long rows, int C, const float* gamma, const float* mu,
                           const float* rstd, float* out, float* dgamma, float* dbeta, float* dbias, float* coef, double* scratch,
                           hipStream_t st) {
  if (v == nullptr || a == nullptr || out == nullptr || rows < 1 || C < 1 || scratch == nullptr || coef == nullptr) {
    veex_set_error("bn_batch_backward: bad arguments"); return -1;
  }
  int chunks; long long per;
  bnb_split(rows, C, chunks, per);
  const dim3 grid(chunks, (C + BNB_TX - 1) / BNB_TX);
  hipLaunchKernelGGL(bnb_colstats_kernel, grid, dim3(BNB_TX * BNB_TY), 0, st, v, a, rows, C, per, scratch);
  hipLaunchKernelGGL(bnb_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, chunks, C, (double)rows, gamma, mu, rstd,
                     coef, dgamma, dbeta);
  hipLaunchKernelGGL(bnb_bwd_apply_kernel, grid, dim3(BNB_TX * BNB_TY), 0, st, v, a, mu, coef, out, rows, C, per, scratch);
  if (dbias != nullptr) hipLaunchKernelGGL(bnb_sum_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, chunks, C, dbias);
  VEEX_CHECK_LAUNCH();
  return 0;
}
