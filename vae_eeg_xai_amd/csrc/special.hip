// HBM-bound kernels of the VAE hot path (everything that is not MFMA work):
//   conv1 (Cin=1, K=9) forward / weight-grad / input-grad        models.py:219-223
//   deconv6 (Cout=1) fused with BCE-from-logits + per-sample reduction, and its backward
//                                                               models.py:300-301,156-160
//   reparameterisation + analytic KL (Philox or injected epsilon) models.py:10-22,161-162
//   BatchNorm(inference)/ReLU backward over [rows][cols], column-sum finalisation
//   split-K / weight-grad slab reduction, TF-form Adam, loss finalisation.
// All of them stream each operand once with 16-byte accesses and reduce with wavefront
// shuffles + one LDS pass; cross-block sums go through partial slabs (no float atomics, so
// results are bitwise reproducible run to run).
#include "veex_internal.h"

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

#define IMG 40
#define C1H 20          // conv1 output 20x20
#define D5H 39          // deconv5 output 39x39 (deconv6 input)
#define NCH 64          // channels of conv1 output / deconv6 input

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------
// conv1: x [B][40][40] -> a1 [B][20][20][64] = relu(conv3x3 s2 SAME(pad 0/1) + bias)
// One work unit = ten output rows of one sample (200 pixels): its 21 input rows are staged in LDS with coalesced
// 16-byte loads, then thread = (pixel lane, 4 channels) walks 13 pixels, 9x4 weights in registers, taps read from LDS,
// one float4 store per pixel.  HBM-bound on the 52 MB it writes; the first version read its nine taps from global
// memory per pixel and was bound by that latency (20.6 us = 2.7 TB/s at batch 512, profiles/r01_seq_f32.txt).
// ---------------------------------------------------------------------------------------
#define C1_UROWS 10                       // output rows per unit
#define C1_UPIX (C1_UROWS * C1H)          // 200 pixels per unit
#define C1_IROWS (2 * C1_UROWS + 1)       // 21 input rows (the last one is the zero row below the image for the lower half)
#define C1_LDW 44                         // padded LDS row: column 40 is the zero column right of the image

// stage input rows [2*oy0, 2*oy0 + 21) of one 40x40 sample into xs[21][44]; rows / columns past the image are zeros
__device__ __forceinline__ void conv1_stage(const float* __restrict__ xsrc, int oy0, float* __restrict__ xs) {
  for (int i = threadIdx.x; i < C1_IROWS * (C1_LDW / 4); i += 256) {
    const int r = i / (C1_LDW / 4), c4 = i - r * (C1_LDW / 4);
    const int iy = 2 * oy0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy < IMG && c4 < IMG / 4) v = *reinterpret_cast<const float4*>(xsrc + iy * IMG + 4 * c4);
    *reinterpret_cast<float4*>(xs + r * C1_LDW + 4 * c4) = v;
  }
}

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int nunits, const float* __restrict__ bscale,
                                                        const float* __restrict__ bshift, __bf16* __restrict__ out_bf,
                                                        float* __restrict__ out_bn) {
  __shared__ __attribute__((aligned(16))) float xs[C1_IROWS * C1_LDW];
  const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(W + t * NCH + 4 * cq);
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * cq);
  float4 osc = make_float4(1.f, 1.f, 1.f, 1.f), osh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (out_bf != nullptr || out_bn != nullptr) {   // consumer copies: this layer's BatchNorm applied (what conv2 reads)
    osc = *reinterpret_cast<const float4*>(bscale + 4 * cq);
    osh = *reinterpret_cast<const float4*>(bshift + 4 * cq);
  }
  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const int n = unit >> 1, oy0 = (unit & 1) * C1_UROWS;
    __syncthreads();                       // the previous unit's taps have been read
    conv1_stage(x + (long long)n * (IMG * IMG), oy0, xs);
    __syncthreads();
    const long long pix0 = (long long)n * (C1H * C1H) + oy0 * C1H;
#pragma unroll 2
    for (int lp = pg; lp < C1_UPIX; lp += 16) {
      const int ry = lp / C1H, ox = lp - ry * C1H;
      const float* t0 = xs + (2 * ry) * C1_LDW + 2 * ox;
      float4 acc = b4;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = t0[ky * C1_LDW + kx];
          const float4 ww = w[ky * 3 + kx];
          acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y);
          acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
        }
      }
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
      *reinterpret_cast<float4*>(out + (pix0 + lp) * NCH + 4 * cq) = acc;
      if (out_bf != nullptr) {
        bf16x4_t h;
        h[0] = (__bf16)fmaf(acc.x, osc.x, osh.x); h[1] = (__bf16)fmaf(acc.y, osc.y, osh.y);
        h[2] = (__bf16)fmaf(acc.z, osc.z, osh.z); h[3] = (__bf16)fmaf(acc.w, osc.w, osh.w);
        *reinterpret_cast<bf16x4_t*>(out_bf + (pix0 + lp) * NCH + 4 * cq) = h;
      }
      if (out_bn != nullptr)
        *reinterpret_cast<float4*>(out_bn + (pix0 + lp) * NCH + 4 * cq) =
            make_float4(fmaf(acc.x, osc.x, osh.x), fmaf(acc.y, osc.y, osh.y), fmaf(acc.z, osc.z, osh.z), fmaf(acc.w, osc.w, osh.w));
    }
  }
}

// dW1[tap][co] partials: slab[block][9][64].  Same units and LDS staging as conv1_fwd; the only global loads of
// the pixel loop are the float4 gradients, four of them in flight per thread.
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                          float* __restrict__ slab, int nunits) {
  __shared__ __attribute__((aligned(16))) float red[16 * 9 * NCH];     // also the staged input rows (21 x 44 floats)
  float* xs = red;
  const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int TRIPS = (C1_UPIX + 15) / 16;     // 13 pixels per thread and unit, the last trip half empty
  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const int n = unit >> 1, oy0 = (unit & 1) * C1_UROWS;
    const float* gsrc = dpre + ((long long)n * (C1H * C1H) + oy0 * C1H) * NCH + 4 * cq;
    float4 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const float4*>(gsrc + (long long)(pg + 16 * u) * NCH);
    __syncthreads();
    conv1_stage(x + (long long)n * (IMG * IMG), oy0, xs);
    __syncthreads();
#pragma unroll
    for (int tr = 0; tr < TRIPS; ++tr) {
      const int lp = pg + 16 * tr;
      const float4 gc = g[tr & 3];
      if (tr + 4 < TRIPS) {
        const int lq = lp + 64;
        g[tr & 3] = (lq < C1_UPIX) ? *reinterpret_cast<const float4*>(gsrc + (long long)lq * NCH) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (lp < C1_UPIX) {
        const int ry = lp / C1H, ox = lp - ry * C1H;
        const float* t0 = xs + (2 * ry) * C1_LDW + 2 * ox;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = t0[ky * C1_LDW + kx];
            float4& a = acc[ky * 3 + kx];
            a.x = fmaf(v, gc.x, a.x); a.y = fmaf(v, gc.y, a.y); a.z = fmaf(v, gc.z, a.z); a.w = fmaf(v, gc.w, a.w);
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(red + (pg * 9 + t) * NCH + 4 * cq) = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * NCH; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) s += red[g2 * 9 * NCH + i];
    slab[(long long)blockIdx.x * (9 * NCH) + i] = s;
  }
}

// sum over the 16 lanes of a DPP row with four v_add_f32 ... dpp (quad xor 1, quad xor 2, half-row mirror, row
// mirror) instead of four ds_bpermute round trips through the LDS pipe; every lane of the row gets the total
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return v;
}

// dx[n][iy][ix] = sum_{ky,kx,co} dpre1[n][(iy-ky)/2][(ix-kx)/2][co] * W1[ky][kx][co]  (attribution only)
// One block per sample, dpre1 streamed exactly once:
//   phase 1: per conv1 output pixel the 9 tap dot products over 64 channels -> T[400][9] in LDS.  Tiles of 64
//            pixels are staged through LDS with coalesced 16-byte loads (the next tile's loads are in flight while
//            this one is multiplied); lane = pixel, wave = channel quarter, so the 9x16 weights of a wave are
//            wave-uniform (scalar registers) and a pixel costs 144 v_fmac per lane with no cross-lane reduction
//            (the 16-lane DPP version spent as much on the row sums as on the products: VALU-bound at 2x the
//            memory time); the four quarter sums meet in LDS.
//   phase 2: per input pixel gather its <= 4 contributions from T, times mul (gradient*input), store.
#define C1D_TP 64            // pixels per tile
#define C1D_LD 68            // padded tile row (floats): 17 sixteen-byte slots, conflict-free ds_read_b128 per lane
__global__ __launch_bounds__(256) void conv1_dgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ W,
                                                          const float* __restrict__ mul, float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float sX[C1D_TP * C1D_LD];
  __shared__ float sP[3][C1D_TP * 9];
  __shared__ float sT[C1H * C1H * 9];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int qtr = __builtin_amdgcn_readfirstlane(tid >> 6);   // channel quarter = wave
  const int pl = tid & 63;
  const float* wq = W + 16 * qtr;                              // W[t*64 + c]
  const float4* src4 = reinterpret_cast<const float4*>(dpre + (long long)n * (C1H * C1H * NCH));
  constexpr int NPIX = C1H * C1H, NT = (NPIX + C1D_TP - 1) / C1D_TP;
  // staging: tile pixel i>>4, float4 column i&15 for i = tid + 256*k; two tiles of loads in flight per block
  struct Stage { float4 v[4]; };
  auto load_tile = [&](Stage& sg, int t) {
    const int t0 = t * C1D_TP;
    const int np = min(C1D_TP, NPIX - t0);   // <= 0 past the end: nothing is loaded
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k;
      sg.v[k] = ((i >> 4) < np) ? src4[t0 * 16 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  Stage s0, s1;
  load_tile(s0, 0);
  load_tile(s1, 1);
  // the multipliers of phase 2 (gradient * input) are fetched now: a load per output inside that loop costs its
  // full latency seven times per block
  float mreg[(IMG * IMG + 255) / 256];
#pragma unroll
  for (int k = 0; k < (IMG * IMG + 255) / 256; ++k) {
    const int o = tid + 256 * k;
    mreg[k] = (mul != nullptr && o < IMG * IMG) ? mul[(long long)n * (IMG * IMG) + o] : 1.f;
  }
  auto step = [&](Stage& sg, int t) {
    const int t0 = t * C1D_TP;
    const int np = min(C1D_TP, NPIX - t0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k;
      *reinterpret_cast<float4*>(sX + (i >> 4) * C1D_LD + 4 * (i & 15)) = sg.v[k];
    }
    __syncthreads();
    load_tile(sg, t + 2);
    float x[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(sX + pl * C1D_LD + 16 * qtr + 4 * c);
      x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    float a[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) sacc = fmaf(x[c], wq[tp * NCH + c], sacc);
      a[tp] = sacc;
    }
    if (qtr > 0) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) sP[qtr - 1][pl * 9 + tp] = a[tp];
    }
    __syncthreads();
    if (qtr == 0 && pl < np) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
        sT[(t0 + pl) * 9 + tp] = ((a[tp] + sP[0][pl * 9 + tp]) + sP[1][pl * 9 + tp]) + sP[2][pl * 9 + tp];
    }
    // (the next tile's sX / sP writes come after this tile's reads: sX is read before the barrier above and
    //  rewritten after it; sP is read here and rewritten only after the next tile's first barrier)
  };
  for (int t = 0; t < NT; t += 2) {
    step(s0, t);
    if (t + 1 < NT) step(s1, t + 1);
  }
  __syncthreads();
  constexpr int NO = (IMG * IMG + 255) / 256;
#pragma unroll
  for (int k = 0; k < NO; ++k) {
    const int o = tid + 256 * k;
    if (o >= IMG * IMG) break;
    const int iy = o / IMG, ix = o - iy * IMG;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= C1H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= C1H) continue;
        s += sT[((ty >> 1) * C1H + (tx >> 1)) * 9 + ky * 3 + kx];
      }
    }
    dx[(long long)n * (IMG * IMG) + o] = s * mreg[k];
  }
}

// ---------------------------------------------------------------------------------------
// deconv6 + sigmoid + BCE-from-logits, one block per sample.
//   logits[oy][ox] = b + sum_{ky,kx in {0,1}} sum_ci y5[oy-ky][ox-kx][ci] * W6[ky][kx][ci],  y5 = BN5(a5)
// phase 1: per input pixel the 4 tap dot products (16 lanes x 4 channels, shuffle reduce) -> LDS
// phase 2: per output pixel gather <=4 of them, BCE, dlogit = (sigmoid - t)/B, block reduction.
// ---------------------------------------------------------------------------------------
#define D6_THREADS 512
// PROBS: the clipped-probability form of keras.backend.binary_crossentropy (SURVEY A.4's fallback, behind
// veex_net_set_semantics) instead of the from-logits form the reference is read to take
template <bool PROBS>
__global__ __launch_bounds__(D6_THREADS) void deconv6_bce_fwd_kernel(
    const float* __restrict__ a5, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ W6, const float* __restrict__ b6, const float* __restrict__ target,
    float* __restrict__ logits, float* __restrict__ recon, float* __restrict__ dlogit,
    float* __restrict__ rec_b, float* __restrict__ dlsum_b, float inv_batch) {
  __shared__ __attribute__((aligned(16))) float d[D5H * D5H * 4];
  __shared__ float redbuf[2 * (D6_THREADS / 64)];
  const int n = blockIdx.x;
  const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
  constexpr int NPG = D6_THREADS / 16;
  float4 w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const float4*>(W6 + t * NCH + 4 * cq);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale != nullptr) {
    sc = *reinterpret_cast<const float4*>(scale + 4 * cq);
    sh = *reinterpret_cast<const float4*>(shift + 4 * cq);
  }
  const float* src = a5 + (long long)n * (D5H * D5H * NCH);
  constexpr int NIT = (D5H * D5H + NPG - 1) / NPG;
  // the targets of phase 2 are fetched now (a load per output inside that loop pays its full latency each time)
  constexpr int NO = (IMG * IMG + D6_THREADS - 1) / D6_THREADS;
  float treg[NO];
#pragma unroll
  for (int k = 0; k < NO; ++k) {
    const int o = threadIdx.x + D6_THREADS * k;
    treg[k] = (target != nullptr && o < IMG * IMG) ? target[(long long)n * (IMG * IMG) + o] : 0.f;
  }
  constexpr int PF = 4;   // pixels in flight per lane
  for (int it0 = 0; it0 < NIT; it0 += PF) {
    float4 av[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int pix = (it0 + u) * NPG + pg;
      av[u] = (pix < D5H * D5H) ? *reinterpret_cast<const float4*>(src + pix * NCH + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int pix = (it0 + u) * NPG + pg;
      float4 a = av[u];
      a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
      float s[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = a.x * w[t].x;
        v = fmaf(a.y, w[t].y, v); v = fmaf(a.z, w[t].z, v); v = fmaf(a.w, w[t].w, v);
        s[t] = row16_sum(v);
      }
      if (cq == 0 && pix < D5H * D5H) *reinterpret_cast<float4*>(d + pix * 4) = make_float4(s[0], s[1], s[2], s[3]);
    }
  }
  __syncthreads();
  const float bias = b6[0];
  float bce_sum = 0.f, dl_sum = 0.f;
#pragma unroll
  for (int k = 0; k < NO; ++k) {
    const int o = threadIdx.x + D6_THREADS * k;
    if (o >= IMG * IMG) break;
    const int oy = o / IMG, ox = o - oy * IMG;
    float a = bias;
#pragma unroll
    for (int ky = 0; ky < 2; ++ky) {
      const int iy = oy - ky;
      if (iy < 0 || iy >= D5H) continue;
#pragma unroll
      for (int kx = 0; kx < 2; ++kx) {
        const int ix = ox - kx;
        if (ix < 0 || ix >= D5H) continue;
        a += d[(iy * D5H + ix) * 4 + ky * 2 + kx];
      }
    }
    const long long oi = (long long)n * (IMG * IMG) + o;
    const float e = expf(-fabsf(a));
    const float sg = (a >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
    if (logits != nullptr) logits[oi] = a;
    if (recon != nullptr) recon[oi] = sg;
    if (target != nullptr) {
      const float t = treg[k];
      float dl;
      if (PROBS) {   // -(t log(p + 1e-7) + (1 - t) log(1 - p + 1e-7)), p = clip(sigmoid(a), 1e-7, 1 - 1e-7); no gradient through an active clip
        const float ce = 1e-7f;
        const float pr = fminf(fmaxf(sg, ce), 1.f - ce);
        bce_sum -= t * logf(pr + ce) + (1.f - t) * logf(1.f - pr + ce);
        const bool inside = sg >= ce && sg <= 1.f - ce;
        dl = inside ? ((1.f - t) / (1.f - pr + ce) - t / (pr + ce)) * sg * (1.f - sg) * inv_batch : 0.f;
      } else {
        bce_sum += fmaxf(a, 0.f) - a * t + log1pf(e);
        dl = (sg - t) * inv_batch;
      }
      dl_sum += dl;
      if (dlogit != nullptr) dlogit[oi] = dl;
    }
  }
  bce_sum = wave_sum(bce_sum);
  dl_sum = wave_sum(dl_sum);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { redbuf[wv] = bce_sum; redbuf[D6_THREADS / 64 + wv] = dl_sum; }
  __syncthreads();
  if (threadIdx.x == 0 && rec_b != nullptr) {
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < D6_THREADS / 64; ++i) { s1 += redbuf[i]; s2 += redbuf[D6_THREADS / 64 + i]; }
    rec_b[n] = s1;
    if (dlsum_b != nullptr) dlsum_b[n] = s2;
  }
}

// deconv6 backward fused with BN5/ReLU backward, one block per sample.
//   g[iy][ix][ci]   = sum_{ky,kx} dlogit[iy+ky][ix+kx] * W6[ky][kx][ci]      (= dL/d y5)
//   dpre5           = (a5 > 0) ? g * scale5 : 0
//   slab[n][0..3]   = sum_pix y5 * dlogit[tap]   (dW6 partial),  slab[n][4..6] = S1,S2,S3 of BN5 backward
__global__ __launch_bounds__(D6_THREADS) void deconv6_bwd_kernel(
    const float* __restrict__ dlogit, const float* __restrict__ a5, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ W6, float* __restrict__ dpre5,
    float* __restrict__ slab, __bf16* __restrict__ dpre5_bf, int raw) {
  // raw != 0 (batch-statistics BatchNorm, bn_batch.hip): dpre5 receives g = dL/d y5 itself -- no ReLU mask, no scale
  __shared__ __attribute__((aligned(16))) float dl[IMG * IMG];
  __shared__ __attribute__((aligned(16))) float red[(D6_THREADS / 16) * 7 * NCH];
  const int n = blockIdx.x;
  const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
  constexpr int NPG = D6_THREADS / 16;
  for (int i = threadIdx.x; i < IMG * IMG; i += D6_THREADS) dl[i] = dlogit[(long long)n * (IMG * IMG) + i];
  float4 w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const float4*>(W6 + t * NCH + 4 * cq);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale != nullptr) {
    sc = *reinterpret_cast<const float4*>(scale + 4 * cq);
    sh = *reinterpret_cast<const float4*>(shift + 4 * cq);
  }
  float4 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float* src = a5 + (long long)n * (D5H * D5H * NCH);
  float* dst = dpre5 + (long long)n * (D5H * D5H * NCH);
  constexpr int PF = 4;   // pixels in flight per lane (48 trips of one dependent 16-byte load each otherwise)
  for (int pix0 = pg; pix0 < D5H * D5H; pix0 += PF * NPG) {
    float4 av[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int pix = pix0 + u * NPG;
      av[u] = (pix < D5H * D5H) ? *reinterpret_cast<const float4*>(src + pix * NCH + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
    const int pix = pix0 + u * NPG;
    if (pix >= D5H * D5H) break;
    const int iy = pix / D5H, ix = pix - iy * D5H;
    const float4 a = av[u];
    float4 y;
    y.x = fmaf(a.x, sc.x, sh.x); y.y = fmaf(a.y, sc.y, sh.y); y.z = fmaf(a.z, sc.z, sh.z); y.w = fmaf(a.w, sc.w, sh.w);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float dv = dl[(iy + (t >> 1)) * IMG + ix + (t & 1)];
      g.x = fmaf(dv, w[t].x, g.x); g.y = fmaf(dv, w[t].y, g.y); g.z = fmaf(dv, w[t].z, g.z); g.w = fmaf(dv, w[t].w, g.w);
      acc[t].x = fmaf(dv, y.x, acc[t].x); acc[t].y = fmaf(dv, y.y, acc[t].y);
      acc[t].z = fmaf(dv, y.z, acc[t].z); acc[t].w = fmaf(dv, y.w, acc[t].w);
    }
    acc[4].x += g.x; acc[4].y += g.y; acc[4].z += g.z; acc[4].w += g.w;
    acc[5].x = fmaf(g.x, a.x, acc[5].x); acc[5].y = fmaf(g.y, a.y, acc[5].y);
    acc[5].z = fmaf(g.z, a.z, acc[5].z); acc[5].w = fmaf(g.w, a.w, acc[5].w);
    float4 o;
    o.x = (a.x > 0.f) ? g.x : 0.f; o.y = (a.y > 0.f) ? g.y : 0.f;
    o.z = (a.z > 0.f) ? g.z : 0.f; o.w = (a.w > 0.f) ? g.w : 0.f;
    acc[6].x += o.x; acc[6].y += o.y; acc[6].z += o.z; acc[6].w += o.w;
    o.x *= sc.x; o.y *= sc.y; o.z *= sc.z; o.w *= sc.w;
    if (raw) o = g;
    if (dpre5 != nullptr) *reinterpret_cast<float4*>(dst + pix * NCH + 4 * cq) = o;
    if (dpre5_bf != nullptr) {
      bf16x4_t h;
      h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
      *reinterpret_cast<bf16x4_t*>(dpre5_bf + (long long)n * (D5H * D5H * NCH) + pix * NCH + 4 * cq) = h;
    }
    }
  }
#pragma unroll
  for (int t = 0; t < 7; ++t) *reinterpret_cast<float4*>(red + (pg * 7 + t) * NCH + 4 * cq) = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 7 * NCH; i += D6_THREADS) {
    float s = 0.f;
    for (int g2 = 0; g2 < NPG; ++g2) s += red[g2 * 7 * NCH + i];
    slab[(long long)n * (7 * NCH) + i] = s;
  }
}

// ---------------------------------------------------------------------------------------
// Philox-4x32-10 (Salmon et al. 2011) + Box-Muller; element e of the (sample, latent) grid
// uses counter e>>2 and output lane e&3, so epsilon is independent of batch split / world size.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float philox_normal(unsigned long long e, unsigned k0, unsigned k1) {
  unsigned c0 = (unsigned)((e >> 2) & 0xFFFFFFFFull), c1 = (unsigned)(e >> 34), c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const int sel = (int)(e & 3ull);
  const unsigned ua = (sel < 2) ? c0 : c2, ub = (sel < 2) ? c1 : c3;
  const float u1 = ((float)ua + 0.5f) * (1.0f / 4294967296.0f);
  const float u2 = ((float)ub + 0.5f) * (1.0f / 4294967296.0f);
  // (float)ua rounds to 2^32 for the top 128 values -> u1 could be 1.0: log(1)=0, still finite
  const float rad = sqrtf(-2.0f * logf(fminf(u1, 0.99999994f)));
  const float th = 6.28318530717958647692f * u2;
  return (sel & 1) ? rad * sinf(th) : rad * cosf(th);
}

// z = mu + exp(0.5*lv)*eps ; kl_b = -0.5 * sum_l (1 + lv - mu^2 - exp(lv));  one wave per sample.
// mulv: [B][2L] (z_mean | z_log_var).  eps: read when use_philox==0, written otherwise.
__global__ __launch_bounds__(64) void reparam_kl_fwd_kernel(const float* __restrict__ mulv, float* __restrict__ eps,
                                                            float* __restrict__ z, float* __restrict__ kl_b, int L,
                                                            int use_philox, unsigned long long seed,
                                                            unsigned long long first_sample,
                                                            const unsigned long long* __restrict__ counter) {
  // counter (nullable): a device word added to first_sample -- a captured graph replays with fresh epsilon because
  // the Philox counter is READ at run time instead of being frozen into the launch (veex_net_set_sample_counter)
  if (counter != nullptr) first_sample += counter[0];
  const int b = blockIdx.x;
  float kl = 0.f;
  for (int l = threadIdx.x; l < L; l += 64) {
    const float mu = mulv[(long long)b * 2 * L + l];
    const float lv = mulv[(long long)b * 2 * L + L + l];
    float e;
    if (use_philox) {
      e = philox_normal((first_sample + (unsigned long long)b) * (unsigned long long)L + l,
                        (unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32));
      eps[(long long)b * L + l] = e;
    } else {
      e = eps[(long long)b * L + l];
    }
    z[(long long)b * L + l] = fmaf(expf(0.5f * lv), e, mu);
    kl += -0.5f * (1.f + lv - mu * mu - expf(lv));
  }
  kl = wave_sum(kl);
  if (threadIdx.x == 0 && kl_b != nullptr) kl_b[b] = kl;
}

// d mu = dz + mu/B ;  d lv = dz*0.5*exp(0.5 lv)*eps + 0.5*(exp(lv)-1)/B     
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ mulv,
                                                             const float* __restrict__ eps, float* __restrict__ dmulv,
                                                             int B, int L, float inv_batch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i - b * L;
  const float mu = mulv[(long long)b * 2 * L + l];
  const float lv = mulv[(long long)b * 2 * L + L + l];
  const float g = dz[i], e = eps[i];
  dmulv[(long long)b * 2 * L + l] = fmaf(mu, inv_batch, g);
  dmulv[(long long)b * 2 * L + L + l] = g * 0.5f * expf(0.5f * lv) * e + 0.5f * (expf(lv) - 1.f) * inv_batch;
}

// ---------------------------------------------------------------------------------------
// BatchNorm (inference mode) helpers
// ---------------------------------------------------------------------------------------
__global__ void bn_prepare_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ mean, const float* __restrict__ var,
                                  float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ rstd,
                                  int n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r = 1.0f / sqrtf(var[i] + eps);
  const float s = gamma[i] * r;
  scale[i] = s;
  shift[i] = fmaf(-mean[i], s, beta[i]);
  rstd[i] = r;
}

// Elementwise BN/ReLU backward over g[R][C]: out = (aux>0) ? g*scale[c] : 0, with per-row-chunk
// column sums S1 = sum g, S2 = sum g*aux, S3 = sum masked g  -> partials[chunk][3][C].
// aux == nullptr: no mask, S2 = 0 (plain column sum, e.g. bias gradient of a linear layer).
#define BRB_ROWS 8
__global__ __launch_bounds__(256) void bnrelu_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ aux,
                                                              const float* __restrict__ e_scale, float* __restrict__ out,
                                                              float* __restrict__ partials, int R, int C, int cmask) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * BRB_ROWS, r1 = min(R, r0 + BRB_ROWS);
  const float sc = (e_scale != nullptr) ? e_scale[c & cmask] : 1.f;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  float gv[BRB_ROWS], av[BRB_ROWS];   // all loads of the chunk first: the rows are independent
#pragma unroll
  for (int k = 0; k < BRB_ROWS; ++k) {
    const long long i = (long long)(r0 + k) * C + c;
    const bool live = r0 + k < r1;
    gv[k] = live ? g[i] : 0.f;
    av[k] = (live && aux != nullptr) ? aux[i] : 1.f;
  }
#pragma unroll
  for (int k = 0; k < BRB_ROWS; ++k) {
    if (r0 + k >= r1) break;
    const long long i = (long long)(r0 + k) * C + c;
    const float v = gv[k], a = av[k];
    if (aux != nullptr) s2 = fmaf(v, a, s2);
    const bool on = a > 0.f;
    s1 += v;
    s3 += on ? v : 0.f;
    if (out != nullptr) out[i] = on ? v * sc : 0.f;
  }
  if (partials != nullptr) {
    float* pp = partials + (long long)blockIdx.y * 3 * C + c;
    pp[0] = s1; pp[C] = s2; pp[2 * C] = s3;
  }
}

// Column-sum finalisation: channel ch collects every column j with (j % nch) == ch over all
// T partial rows.  dbeta = S1, dgamma = rstd*(S2 - mean*S1), dbias = escale*S3.
// partial element (t,k,j) lives at partials[t*row_stride + k*ncols + j].  One block per channel.
__device__ __forceinline__ void colsum_finalize_body(const float* __restrict__ partials, int T, long long row_stride,
                                                     int ncols, int nch, const float* __restrict__ e_scale,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ dbias, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int ch, float* red) {
  const int nrep = ncols / nch;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int total = T * nrep;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int t = i / nrep, rep = i - t * nrep;
    const float* pp = partials + (long long)t * row_stride + rep * nch + ch;
    s1 += pp[0]; s2 += pp[ncols]; s3 += pp[2 * ncols];
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = s1; red[4 + wv] = s2; red[8 + wv] = s3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0] + red[1] + red[2] + red[3];
    s2 = red[4] + red[5] + red[6] + red[7];
    s3 = red[8] + red[9] + red[10] + red[11];
    if (dbias != nullptr) dbias[ch] = (e_scale != nullptr) ? e_scale[ch] * s3 : s3;
    if (dbeta != nullptr) dbeta[ch] = s1;
    if (dgamma != nullptr) dgamma[ch] = rstd[ch] * (s2 - mean[ch] * s1);
  }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partials, int T,
                                                              long long row_stride, int ncols, int nch,
                                                              const float* __restrict__ e_scale,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float* __restrict__ dbias, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ float red[3 * 4];
  colsum_finalize_body(partials, T, row_stride, ncols, nch, e_scale, mean, rstd, dbias, dgamma, dbeta, blockIdx.x, red);
}

// First stage for tall partial matrices (an EPI_BWD launch leaves one row per 64-row tile: 11552 rows of 3*Cout at
// batch 512, which one block per channel reads 4 strided bytes at a time): chunk c of `rpc` rows -> one row,
// out[c][W], read coalesced (W < 256: 256/W rows in flight, folded through LDS).  The finalisation then runs
// on the <= 256 compacted rows.  Sums in a fixed order: deterministic.
__device__ __forceinline__ void colsum_compact_body(const float* __restrict__ partials, int T, long long row_stride,
                                                    int W, int rpc, float* __restrict__ out, int chunk, int cgroup,
                                                    float* red) {
  const int tid = threadIdx.x;
  const int R = (W >= 256) ? 1 : 256 / W;
  const int r = (W >= 256) ? 0 : tid / W;
  const int col = (W >= 256) ? cgroup * 256 + tid : tid - r * W;
  const bool live = (r < R) && (col < W);
  const int t0 = chunk * rpc, t1 = min(T, t0 + rpc);
  float s = 0.f;
  if (live) {
    const float* pp = partials + col;
    int t = t0 + r;
    for (; t + 3 * R < t1; t += 4 * R) {
      const float v0 = pp[(long long)t * row_stride], v1 = pp[(long long)(t + R) * row_stride];
      const float v2 = pp[(long long)(t + 2 * R) * row_stride], v3 = pp[(long long)(t + 3 * R) * row_stride];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; t < t1; t += R) s += pp[(long long)t * row_stride];
  }
  if (R > 1) {
    red[tid] = live ? s : 0.f;
    __syncthreads();
    if (tid < W) {
      float a = red[tid];
      for (int k = 1; k < R; ++k) a += red[k * W + tid];
      out[(long long)chunk * W + tid] = a;
    }
  } else if (live) {
    out[(long long)chunk * W + col] = s;
  }
}
// both stages for a whole table of partial matrices (one backward pass): block -> entry by its first block
__global__ __launch_bounds__(256) void colsum_compact_batch_kernel(const ColsumBatch b, float* __restrict__ scratch) {
  __shared__ float red[256];
  int k = 0;
  for (int i = 1; i < b.n; ++i) if ((int)blockIdx.x >= b.e[i].blk_c) k = i;   // blk_c ascending; empty entries repeat it
  while (k > 0 && b.e[k].nchunks == 0) --k;
  const ColsumEntry& e = b.e[k];
  const int local = (int)blockIdx.x - e.blk_c;
  colsum_compact_body(e.partials, e.T, e.row_stride, 3 * e.ncols, e.rpc, scratch + e.scr_off, local / e.cgroups,
                      local % e.cgroups, red);
}
__global__ __launch_bounds__(256) void colsum_finalize_batch_kernel(const ColsumBatch b, const float* __restrict__ scratch) {
  __shared__ float red[3 * 4];
  int k = 0;
  for (int i = 1; i < b.n; ++i) if ((int)blockIdx.x >= b.e[i].blk_f) k = i;
  const ColsumEntry& e = b.e[k];
  const bool cmp = e.nchunks > 0;
  colsum_finalize_body(cmp ? scratch + e.scr_off : e.partials, cmp ? e.nchunks : e.T,
                       cmp ? 3LL * e.ncols : e.row_stride, e.ncols, e.nch, e.e_scale, e.mean, e.rstd, e.dbias, e.dgamma,
                       e.dbeta, (int)blockIdx.x - e.blk_f, red);
}

// out[i] = epi( sum_s slab[s*stride + i] ),  epi = (+bias[i % ncols]) (relu)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, int nsplit, long long stride,
                                                       long long n, float* __restrict__ out,
                                                       const float* __restrict__ bias, int ncols, int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {   // four independent loads in flight, added in slab order
      const float v0 = slab[(long long)k * stride + i], v1 = slab[(long long)(k + 1) * stride + i];
      const float v2 = slab[(long long)(k + 2) * stride + i], v3 = slab[(long long)(k + 3) * stride + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nsplit; ++k) s += slab[(long long)k * stride + i];
    if (bias != nullptr) s += bias[(int)(i % ncols)];
    if (relu) s = fmaxf(s, 0.f);
    out[i] = s;
  }
}

// Same sums in the same order (slab 0, 1, 2, ... per element), four elements per thread and eight slabs of
// 16-byte loads in flight: the plain kernel keeps 4 bytes per thread in flight and reaches ~3 TB/s on the
// 50 MB of a weight-gradient launch.  Needs n % 4 == 0, stride % 4 == 0, no epilogue.
__global__ __launch_bounds__(256) void slab_sum_vec_kernel(const float4* __restrict__ slab, int nsplit, long long stride4,
                                                           long long n4, float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(long long)(k + u) * stride4 + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; k < nsplit; ++k) {
    const float4 v = slab[(long long)k * stride4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[i] = s;
}

// Same result for "many slabs, few elements": 64 columns x LY slab lanes per block; every lane sums
// its slabs in increasing order, then the LY lane sums are added in lane order (fixed order -> reproducible).
// LY = 16 for the few-column case; LY = 4 (256-thread blocks, which find a free slot beside a GEMM of the
// other stream far sooner than a 1024-thread block) otherwise.
template <int LY>
__global__ __launch_bounds__(64 * LY) void slab_sum_tall_kernel(const float* __restrict__ slab, int nsplit,
                                                             long long stride, long long n, float* __restrict__ out,
                                                             const float* __restrict__ bias, int ncols, int relu) {
  __shared__ float red[LY][64];
  const int cx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + cx;
  float s = 0.f;
  if (i < n) {   // four independent loads in flight, added in slab order
    int k = ly;
    for (; k + 3 * LY < nsplit; k += 4 * LY) {
      const float v0 = slab[(long long)k * stride + i], v1 = slab[(long long)(k + LY) * stride + i];
      const float v2 = slab[(long long)(k + 2 * LY) * stride + i], v3 = slab[(long long)(k + 3 * LY) * stride + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nsplit; k += LY) s += slab[(long long)k * stride + i];
  }
  red[ly][cx] = s;
  __syncthreads();
  if (ly == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LY; ++k) t += red[k][cx];
    if (bias != nullptr) t += bias[(int)(i % ncols)];
    if (relu) t = fmaxf(t, 0.f);
    out[i] = t;
  }
}

// ---------------------------------------------------------------------------------------
// keras.optimizers.legacy.Adam: epsilon outside the bias correction.
// state[0] = step count t (as float bits of an int), alpha written to state_alpha.
// ---------------------------------------------------------------------------------------
__global__ void adam_prepare_kernel(int* __restrict__ step, float* __restrict__ alpha, double lr, double b1, double b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int t = step[0] + 1;
    step[0] = t;
    const double a = lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t));
    alpha[0] = (float)a;
  }
}

__global__ __launch_bounds__(256) void adam_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long long n4,
                                                         const float* __restrict__ alpha_p, float gscale, float b1,
                                                         float omb1, float b2, float omb2, float eps) {
  const float alpha = alpha_p[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    gg.x *= gscale; gg.y *= gscale; gg.z *= gscale; gg.w *= gscale;
    mm.x = b1 * mm.x + omb1 * gg.x; mm.y = b1 * mm.y + omb1 * gg.y;
    mm.z = b1 * mm.z + omb1 * gg.z; mm.w = b1 * mm.w + omb1 * gg.w;
    vv.x = b2 * vv.x + omb2 * gg.x * gg.x; vv.y = b2 * vv.y + omb2 * gg.y * gg.y;
    vv.z = b2 * vv.z + omb2 * gg.z * gg.z; vv.w = b2 * vv.w + omb2 * gg.w * gg.w;
    pp.x -= alpha * mm.x / (sqrtf(vv.x) + eps); pp.y -= alpha * mm.y / (sqrtf(vv.y) + eps);
    pp.z -= alpha * mm.z / (sqrtf(vv.z) + eps); pp.w -= alpha * mm.w / (sqrtf(vv.w) + eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

// losses[0..2] = total, rec, kl (sums over the local batch / div; div = the local batch unless the caller set a divisor); metrics[0..2] += those, metrics[3] += 1;
// db6 = sum_b dlsum_b (bias gradient of deconv6).  Single block.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ rec_b, const float* __restrict__ kl_b,
                                                            const float* __restrict__ dlsum_b, int B, double div,
                                                            float* __restrict__ losses, float* __restrict__ metrics,
                                                            float* __restrict__ db6) {
  __shared__ double red[3 * 4];
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) {
    s1 += (double)rec_b[i];
    if (kl_b != nullptr) s2 += (double)kl_b[i];
    if (dlsum_b != nullptr) s3 += (double)dlsum_b[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = s1; red[4 + wv] = s2; red[8 + wv] = s3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0] + red[1] + red[2] + red[3];
    s2 = red[4] + red[5] + red[6] + red[7];
    s3 = red[8] + red[9] + red[10] + red[11];
    const float rec = (float)(s1 / div), kl = (float)(s2 / div);
    if (losses != nullptr) { losses[0] = rec + kl; losses[1] = rec; losses[2] = kl; }
    if (metrics != nullptr) { metrics[0] += rec + kl; metrics[1] += rec; metrics[2] += kl; metrics[3] += 1.f; }
    if (db6 != nullptr) db6[0] = (float)s3;
  }
}

// Attribution seed: d(pre-activation of the encoder dense) for target latent unit j.
//  VAE: dpre[b][i] = (h[b][i] > 0) ? Whead[i][j] : 0      (z_mean column j of the [100][2L] head matrix)
//  AE : dpre[b][i] = (i == j && h[b][i] > 0) ? 1 : 0
__global__ __launch_bounds__(256) void attr_seed_kernel(const float* __restrict__ h, const float* __restrict__ whead,
                                                        float* __restrict__ dpre, int B, int H, int ldw, int unit,
                                                        int is_ae) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * H) return;
  const int col = i % H;
  const bool on = h[i] > 0.f;
  float v;
  if (is_ae) v = (col == unit && on) ? 1.f : 0.f;
  else v = on ? whead[col * ldw + unit] : 0.f;
  dpre[i] = v;
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
static inline int grid_for(long long work, int per_block, int cap) {
  long long g = (work + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int veex_conv1_fwd(const float* x, const float* W, const float* bias, float* out, int B, hipStream_t st) {
  return veex_conv1_fwd_copies(x, W, bias, out, nullptr, nullptr, nullptr, nullptr, B, st);
}

int veex_conv1_fwd_bf16(const float* x, const float* W, const float* bias, float* out, const float* bn_scale,
                        const float* bn_shift, void* out_bf16, int B, hipStream_t st) {
  return veex_conv1_fwd_copies(x, W, bias, out, bn_scale, bn_shift, nullptr, out_bf16, B, st);
}

int veex_conv1_fwd_copies(const float* x, const float* W, const float* bias, float* out, const float* bn_scale,
                          const float* bn_shift, float* out_bn, void* out_bf16, int B, hipStream_t st) {
  if ((out_bf16 != nullptr || out_bn != nullptr) && (bn_scale == nullptr || bn_shift == nullptr)) {
    veex_set_error("conv1_fwd: a consumer copy needs the BatchNorm affine"); return -1;
  }
  const int nunits = 2 * B;    // ten output rows of one sample each
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3(grid_for(nunits, 1, 2048)), dim3(256), 0, st, x, W, bias, out, nunits, bn_scale,
                     bn_shift, reinterpret_cast<__bf16*>(out_bf16), out_bn);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_conv1_wgrad_blocks(int B) { return grid_for(2LL * B, 1, 1024); }   // units of ten output rows; at most 1024 slab rows

int veex_conv1_wgrad(const float* x, const float* dpre, float* slab, int B, hipStream_t st) {
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(veex_conv1_wgrad_blocks(B)), dim3(256), 0, st, x, dpre, slab, 2 * B);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_conv1_dgrad(const float* dpre, const float* W, const float* mul, float* dx, int B, hipStream_t st) {
  if (dpre == nullptr || W == nullptr || dx == nullptr || B < 1) { veex_set_error("conv1_dgrad: bad arguments"); return -1; }
  hipLaunchKernelGGL(conv1_dgrad_kernel, dim3(B), dim3(256), 0, st, dpre, W, mul, dx);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_deconv6_bce_fwd_form(const float* a5, const float* scale, const float* shift, const float* W6, const float* b6,
                               const float* target, float* logits, float* recon, float* dlogit, float* rec_b,
                               float* dlsum_b, int B, float inv_batch, int bce_form, hipStream_t st) {
  if (bce_form != 0 && bce_form != 1) { veex_set_error("deconv6_bce_fwd: bce_form must be 0 (logits) or 1 (probabilities)"); return -1; }
  if (bce_form == 1)
    hipLaunchKernelGGL(deconv6_bce_fwd_kernel<true>, dim3(B), dim3(D6_THREADS), 0, st, a5, scale, shift, W6, b6, target, logits,
                       recon, dlogit, rec_b, dlsum_b, inv_batch);
  else
    hipLaunchKernelGGL(deconv6_bce_fwd_kernel<false>, dim3(B), dim3(D6_THREADS), 0, st, a5, scale, shift, W6, b6, target, logits,
                       recon, dlogit, rec_b, dlsum_b, inv_batch);
  VEEX_CHECK_LAUNCH();
  return 0;
}
int veex_deconv6_bce_fwd(const float* a5, const float* scale, const float* shift, const float* W6, const float* b6,
                          const float* target, float* logits, float* recon, float* dlogit, float* rec_b,
                          float* dlsum_b, int B, float inv_batch, hipStream_t st) {
  return veex_deconv6_bce_fwd_form(a5, scale, shift, W6, b6, target, logits, recon, dlogit, rec_b, dlsum_b, B, inv_batch, 0, st);
}

int veex_deconv6_bwd(const float* dlogit, const float* a5, const float* scale, const float* shift, const float* W6,
                      float* dpre5, float* slab, int B, hipStream_t st) {
  return veex_deconv6_bwd_bf16(dlogit, a5, scale, shift, W6, dpre5, nullptr, slab, B, st);
}

int veex_deconv6_bwd_bf16(const float* dlogit, const float* a5, const float* scale, const float* shift, const float* W6,
                          float* dpre5, void* dpre5_bf16, float* slab, int B, hipStream_t st) {
  if (dpre5 == nullptr && dpre5_bf16 == nullptr) { veex_set_error("deconv6_bwd: no output buffer"); return -1; }
  hipLaunchKernelGGL(deconv6_bwd_kernel, dim3(B), dim3(D6_THREADS), 0, st, dlogit, a5, scale, shift, W6, dpre5, slab,
                     reinterpret_cast<__bf16*>(dpre5_bf16), 0);
  VEEX_CHECK_LAUNCH();
  return 0;
}
// internal (batch-statistics BatchNorm): dL/d y5 itself into g5, dW6 partials into slab rows 0-3 as usual
int veexi_deconv6_bwd_raw(const float* dlogit, const float* a5, const float* scale, const float* shift, const float* W6,
                          float* g5, float* slab, int B, hipStream_t st) {
  hipLaunchKernelGGL(deconv6_bwd_kernel, dim3(B), dim3(D6_THREADS), 0, st, dlogit, a5, scale, shift, W6, g5, slab,
                     static_cast<__bf16*>(nullptr), 1);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_reparam_kl_fwd_ctr(const float* mulv, float* eps, float* z, float* kl_b, int B, int L, int use_philox,
                             unsigned long long seed, unsigned long long first_sample, const unsigned long long* counter,
                             hipStream_t st) {
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(B), dim3(64), 0, st, mulv, eps, z, kl_b, L, use_philox, seed,
                     first_sample, counter);
  VEEX_CHECK_LAUNCH();
  return 0;
}
int veex_reparam_kl_fwd(const float* mulv, float* eps, float* z, float* kl_b, int B, int L, int use_philox,
                         unsigned long long seed, unsigned long long first_sample, hipStream_t st) {
  return veex_reparam_kl_fwd_ctr(mulv, eps, z, kl_b, B, L, use_philox, seed, first_sample, nullptr, st);
}
__global__ void counter_add_kernel(unsigned long long* ctr, long long inc) { ctr[0] += (unsigned long long)inc; }
int veex_counter_add(unsigned long long* counter, long long inc, hipStream_t st) {
  if (counter == nullptr) { veex_set_error("counter_add: null counter"); return -1; }
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, st, counter, inc);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_reparam_kl_bwd(const float* dz, const float* mulv, const float* eps, float* dmulv, int B, int L,
                         float inv_batch, hipStream_t st) {
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, dz, mulv, eps, dmulv, B, L,
                     inv_batch);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_bn_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float* scale,
                     float* shift, float* rstd, int n, float eps, hipStream_t st) {
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var, scale, shift,
                     rstd, n, eps);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_bnrelu_bwd_rows_chunks(int R) { return (R + BRB_ROWS - 1) / BRB_ROWS; }

int veex_bnrelu_bwd_rows(const float* g, const float* aux, const float* e_scale, float* out, float* partials, int R,
                          int C, int e_channels, hipStream_t st) {
  if (e_scale != nullptr && (e_channels <= 0 || (e_channels & (e_channels - 1)) != 0)) {
    veex_set_error("bnrelu_bwd_rows: e_channels must be a power of two");
    return -1;
  }
  if (R < 1 || C < 1 || veex_bnrelu_bwd_rows_chunks(R) > 65535) {
    veex_set_error("bnrelu_bwd_rows: %d rows x %d columns (at most %d rows per call: split larger inputs)", R, C, 65535 * BRB_ROWS);
    return -1;
  }
  const int cmask = (e_scale != nullptr) ? e_channels - 1 : 0;
  hipLaunchKernelGGL(bnrelu_bwd_rows_kernel, dim3((C + 255) / 256, veex_bnrelu_bwd_rows_chunks(R)), dim3(256), 0, st, g, aux,
                     e_scale, out, partials, R, C, cmask);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_colsum_finalize(const float* partials, int T, long long row_stride, int ncols, int nch, const float* e_scale,
                          const float* mean, const float* rstd, float* dbias, float* dgamma, float* dbeta,
                          hipStream_t st) {
  if (nch <= 0 || ncols % nch != 0) {
    veex_set_error("colsum_finalize: channel count %d must divide ncols %d", nch, ncols);
    return -1;
  }
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(nch), dim3(256), 0, st, partials, T, row_stride, ncols, nch, e_scale,
                     mean, rstd, dbias, dgamma, dbeta);
  VEEX_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

// All finalisations of one pass: tall matrices (T >= 512) are compacted first (one launch for all of them), then one
// launch finalises every channel of every entry.  Fills the grid bookkeeping of `b`.
int veex_colsum_batch(ColsumBatch& b, float* scratch, long long scratch_n, hipStream_t st) {
  if (b.n < 1) return 0;
  int nc = 0, nf = 0;
  long long off = 0;
  for (int i = 0; i < b.n; ++i) {
    ColsumEntry& e = b.e[i];
    if (e.nch <= 0 || e.ncols % e.nch != 0) { veex_set_error("colsum_batch: channel count %d must divide ncols %d", e.nch, e.ncols); return -1; }
    const int W = 3 * e.ncols;
    e.rpc = 1; e.nchunks = 0; e.cgroups = 1; e.blk_c = nc; e.blk_f = nf; e.scr_off = off;
    if (e.T >= 512 && scratch != nullptr) {
      const int rpc = (e.T + 255) / 256, nchunks = (e.T + rpc - 1) / rpc;
      if (off + (long long)nchunks * W <= scratch_n) {
        e.rpc = rpc; e.nchunks = nchunks; e.cgroups = (W >= 256) ? (W + 255) / 256 : 1;
        nc += nchunks * e.cgroups; off += (long long)nchunks * W;
      }
    }
    nf += e.nch;
  }
  if (nc > 0) {
    hipLaunchKernelGGL(colsum_compact_batch_kernel, dim3(nc), dim3(256), 0, st, b, scratch);
    VEEX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(colsum_finalize_batch_kernel, dim3(nf), dim3(256), 0, st, b, (const float*)scratch);
  VEEX_CHECK_LAUNCH();
  return 0;
}

extern "C" {

int veex_slab_sum(const float* slab, int nsplit, long long stride, long long n, float* out, const float* bias,
                   int ncols, int relu, hipStream_t st) {
  if (nsplit > 32 && n <= 4096)
    hipLaunchKernelGGL(slab_sum_tall_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, slab, nsplit, stride, n,
                       out, bias, ncols, relu);
  else if (nsplit > 32 && n <= 65536)
    hipLaunchKernelGGL(slab_sum_tall_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slab, nsplit, stride, n,
                       out, bias, ncols, relu);
  else if (bias == nullptr && !relu && nsplit >= 4 && (n & 3) == 0 && (stride & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(slab_sum_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(slab), nsplit, stride / 4, n / 4, reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, slab, nsplit, stride, n, out,
                       bias, ncols, relu);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_adam(float* p, const float* g, float* m, float* v, long long n, int* step, float* alpha, double lr,
              double gscale, double b1, double b2, double eps, hipStream_t st) {
  if (n % 4 != 0) { veex_set_error("adam: element count must be a multiple of 4"); return -1; }
  if (p == nullptr || g == nullptr || m == nullptr || v == nullptr || step == nullptr || alpha == nullptr) {
    veex_set_error("adam: null argument"); return -1;
  }
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, step, alpha, lr, b1, b2);
  // (1-beta) is formed in double and rounded once (1.f - 0.999f is off by 1.3e-5 relative)
  hipLaunchKernelGGL(adam_apply_kernel, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, st, p, g, m, v, n / 4, alpha,
                     (float)gscale, (float)b1, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veexi_loss_finalize(const float* rec_b, const float* kl_b, const float* dlsum_b, int B, double div, float* losses,
                        float* metrics, float* db6, hipStream_t st) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, rec_b, kl_b, dlsum_b, B, div, losses, metrics, db6);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veexi_attr_seed(const float* h, const float* whead, float* dpre, int B, int H, int ldw, int unit, int is_ae,
                    hipStream_t st) {
  hipLaunchKernelGGL(attr_seed_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, h, whead, dpre, B, H, ldw, unit,
                     is_ae);
  VEEX_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
