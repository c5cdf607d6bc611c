// Weight-gradient GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   dW[tap][cb][cs] = sum_{n,py,px} Big(n, s*py+ky-pb, s*px+kx-pb, cb) * Small(n,py,px,cs)
//
// Conv2D (kernel (kh,kw,Cin,Cout)):            Big = layer input,  Small = d(pre-activation)
// Conv2DTranspose (kernel (kh,kw,Cout,Cin)):   Big = d(pre-act),   Small = layer input
// Dense (kernel (in,out)):                     1 tap, Big = input, Small = d(pre-act)
// (the reference obtains these from tape.gradient, models.py:166).
//
// The reduction runs over pixels (B*Hs*Ws, up to ~10^6), the output is small, so the grid is
// split along the reduction; every split writes its own [M][Cs] slab with plain stores and a
// second kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics).
// Either operand may carry the producer's BatchNorm affine (applied while staging, in-bounds
// pixels only).
#include "veex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 wld4(const float* ptr, bool ok, int nvalid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    if (vec && nvalid >= 4) {
      v = *reinterpret_cast<const float4*>(ptr);
    } else {
      if (nvalid > 0) v.x = ptr[0];
      if (nvalid > 1) v.y = ptr[1];
      if (nvalid > 2) v.z = ptr[2];
      if (nvalid > 3) v.w = ptr[3];
    }
  }
  return v;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(const WGParams p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A_ELEMS = VEEX_BK * LDA, B_ELEMS = VEEX_BK * LDB;
  constexpr int QM = BM / 4, QN = BN / 4;
  constexpr int RPA = 256 / QM, RPB = 256 / QN;       // reduction rows staged per pass
  constexpr int APASS = VEEX_BK / RPA, BPASS = VEEX_BK / RPB;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(APASS >= 1 && BPASS >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][16][BM+4]  (reduction row major, m contiguous)
  float* Bs = smem + 2 * A_ELEMS;    // [2][16][BN+4]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bmt, bnt, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bmt, bnt, kz);
  const int m0 = bmt * BM, n0 = bnt * BN;
  const int r_begin = kz * p.rows_per_split;
  const int r_end = min(p.R, r_begin + p.rows_per_split);
  const int nchunks = (r_end > r_begin) ? (r_end - r_begin + VEEX_BK - 1) / VEEX_BK : 0;

  // A loader: m-quad fixed per thread
  const int amq = tid % QM, arow = tid / QM;
  const int am = m0 + 4 * amq;
  const bool vecA = ((p.Cb & 3) == 0);
  const bool vecB = ((p.Cs & 3) == 0);
  int a_ky[4], a_kx[4], a_cb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = am + e;
    const int tap = m / p.Cb;
    a_cb[e] = m - tap * p.Cb;
    a_ky[e] = tap / p.kw;
    a_kx[e] = tap - a_ky[e] * p.kw;
  }
  const int bnq = tid % QN, brow = tid / QN;
  const int bn = n0 + 4 * bnq;
  const int HWs = p.Hs * p.Ws;

  float4 ra[APASS], rb[BPASS];

  auto load_chunk = [&](int c) {
    const int rbase = r_begin + c * VEEX_BK;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int r = rbase + arow + RPA * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < r_end && am < p.M) {
        const int n = r / HWs, rem = r - n * HWs;
        const int py = rem / p.Ws, px = rem - py * p.Ws;
        if (vecA) {
          const int by = p.stride * py + a_ky[0] - p.pb, bx = p.stride * px + a_kx[0] - p.pb;
          if (by >= 0 && by < p.Hb && bx >= 0 && bx < p.Wb) {
            const int pix = by * p.Wb + bx;
            v = *reinterpret_cast<const float4*>(p.Big + ((long long)(n * p.Hb * p.Wb + pix)) * p.Cb + a_cb[0]);
            if (p.b_scale != nullptr) {
              const int aidx = (a_cb[0] + (p.b_aff_spatial ? pix * p.Cb : 0)) & p.b_aff_mask;
              const float4 sc = *reinterpret_cast<const float4*>(p.b_scale + aidx);
              const float4 sh = *reinterpret_cast<const float4*>(p.b_shift + aidx);
              v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
              v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            }
          }
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (am + e >= p.M) continue;
            const int by = p.stride * py + a_ky[e] - p.pb, bx = p.stride * px + a_kx[e] - p.pb;
            if (by < 0 || by >= p.Hb || bx < 0 || bx >= p.Wb) continue;
            const int pix = by * p.Wb + bx;
            float x = p.Big[((long long)(n * p.Hb * p.Wb + pix)) * p.Cb + a_cb[e]];
            if (p.b_scale != nullptr) {
              const int aidx = (a_cb[e] + (p.b_aff_spatial ? pix * p.Cb : 0)) & p.b_aff_mask;
              x = fmaf(x, p.b_scale[aidx], p.b_shift[aidx]);
            }
            t[e] = x;
          }
          v = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int r = rbase + brow + RPB * i;
      const bool ok = r < r_end && bn < p.Cs;
      float4 v = wld4(p.Small + (long long)r * p.Cs + bn, ok, p.Cs - bn, vecB);
      if (p.s_scale != nullptr && ok) {
        const int pix = p.s_aff_spatial ? (r % HWs) : 0;
        const int aidx = (bn + pix * p.Cs) & p.s_aff_mask;
        const float4 sc = wld4(p.s_scale + aidx, true, p.Cs - bn, vecB);
        const float4 sh = wld4(p.s_shift + aidx, true, p.Cs - bn, vecB);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
      rb[i] = v;
    }
  };

  auto store_chunk = [&](int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<float4*>(as + (arow + RPA * i) * LDA + 4 * amq) = ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      *reinterpret_cast<float4*>(bs + (brow + RPB * i) * LDB + 4 * bnq) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    const bool more = (c + 1 < nchunks);
    if (more) load_chunk(c + 1);
    const float* as = As + cur * A_ELEMS + (8 * lh) * LDA + wm * WTM + l31;
    const float* bs = Bs + cur * B_ELEMS + (8 * lh) * LDB + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = as[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = bs[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(cur ^ 1);
    __syncthreads();
  }

  float* outp = p.Out + (long long)kz * p.M * p.Cs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= p.Cs) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) outp[(long long)m * p.Cs + col] = acc[i][j][r];
      }
    }
  }
}


// ---------------------------------------------------------------------------
// Fast path: channel counts % 4 == 0, operands < 2 GiB, R < 2^24.  Every global access is a
// 16-byte raw buffer load with the offset forced out of range for padding pixels and rows past
// the split, so the reduction loop is one basic block.  Row index -> (n, py, px) uses an exact
// float-reciprocal division (r < 2^24).
// AFFB: Big carries a per-channel BatchNorm affine; AFFS: 0 none, 1 per-channel, 2 flattened.
// ---------------------------------------------------------------------------
typedef unsigned int v4u __attribute__((__vector_size__(16)));
typedef float v4f __attribute__((__vector_size__(16)));
#define VEEX_OOB 0xFFFFFF00u

__device__ __forceinline__ float4 wbload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}

// q = r / d, rem = r % d for 0 <= r < 2^24, 1 <= d < 2^24
__device__ __forceinline__ void fdivmod(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

template <int BM, int BN, int WM, int WN, bool AFFB, int AFFS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WGParams p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A_ELEMS = VEEX_BK * LDA, B_ELEMS = VEEX_BK * LDB;
  constexpr int QM = BM / 4, QN = BN / 4;
  constexpr int RPA = 256 / QM, RPB = 256 / QN;
  constexpr int APASS = VEEX_BK / RPA, BPASS = VEEX_BK / RPB;
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bmt, bnt, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bmt, bnt, kz);
  const int m0 = bmt * BM, n0 = bnt * BN;
  const int r_begin = kz * p.rows_per_split;
  const int r_end = min(p.R, r_begin + p.rows_per_split);
  const int nchunks = (r_end > r_begin) ? (r_end - r_begin + VEEX_BK - 1) / VEEX_BK : 0;

  const __amdgpu_buffer_rsrc_t rBig = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Big), 0, (int)((unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rSmall = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Small), 0, (int)((unsigned)p.R * p.Cs * 4u), 0x00020000);

  const int amq = tid % QM, arow = tid / QM;
  const int am = m0 + 4 * amq;
  const bool am_ok = am < p.M;
  const int atap = am / p.Cb;
  const int acb = am - atap * p.Cb;
  const int aky = atap / p.kw, akx = atap - aky * p.kw;
  const int bnq = tid % QN, brow = tid / QN;
  const int bn = n0 + 4 * bnq;
  const bool bn_ok = bn < p.Cs;
  const int HWs = p.Hs * p.Ws;
  const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;

  float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (AFFB && am_ok) {
    bsc = *reinterpret_cast<const float4*>(p.b_scale + (acb & p.b_aff_mask));
    bsh = *reinterpret_cast<const float4*>(p.b_shift + (acb & p.b_aff_mask));
  }
  float4 ssc = make_float4(1.f, 1.f, 1.f, 1.f), ssh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (AFFS == 1 && bn_ok) {
    ssc = *reinterpret_cast<const float4*>(p.s_scale + (bn & p.s_aff_mask));
    ssh = *reinterpret_cast<const float4*>(p.s_shift + (bn & p.s_aff_mask));
  }
  const unsigned saff_bytes = (AFFS == 2) ? (unsigned)(p.s_aff_mask + 1) * 4u : 0u;
  const __amdgpu_buffer_rsrc_t rSS = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFFS == 2 ? p.s_scale : p.Small), 0, (int)saff_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rSH = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFFS == 2 ? p.s_shift : p.Small), 0, (int)saff_bytes, 0x00020000);

  float4 ra[APASS], rb[BPASS], rssc[AFFS == 2 ? BPASS : 1], rssh[AFFS == 2 ? BPASS : 1];
  bool aok[APASS], bok[BPASS];

  auto load_chunk = [&](int c) {
    const int rbase = r_begin + c * VEEX_BK;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int r = rbase + arow + RPA * i;
      int n, rem, py, px;
      fdivmod(r, HWs, invHWs, n, rem);
      fdivmod(rem, p.Ws, invWs, py, px);
      const int by = p.stride * py + aky - p.pb, bx = p.stride * px + akx - p.pb;
      const bool ok = am_ok && r < r_end && by >= 0 && by < p.Hb && bx >= 0 && bx < p.Wb;
      const unsigned off = ok ? (unsigned)((((n * p.Hb + by) * p.Wb + bx) * p.Cb + acb) * 4) : VEEX_OOB;
      ra[i] = wbload4(rBig, off);
      aok[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int r = rbase + brow + RPB * i;
      const bool ok = bn_ok && r < r_end;
      rb[i] = wbload4(rSmall, ok ? (unsigned)((r * p.Cs + bn) * 4) : VEEX_OOB);
      bok[i] = ok;
      if (AFFS == 2) {
        int n, rem;
        fdivmod(r, HWs, invHWs, n, rem);
        const unsigned ao = ok ? (unsigned)((rem * p.Cs + bn) * 4) : VEEX_OOB;
        rssc[i] = wbload4(rSS, ao);
        rssh[i] = wbload4(rSH, ao);
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float4 v = ra[i];
      if (AFFB) {
        v.x = aok[i] ? fmaf(v.x, bsc.x, bsh.x) : 0.f; v.y = aok[i] ? fmaf(v.y, bsc.y, bsh.y) : 0.f;
        v.z = aok[i] ? fmaf(v.z, bsc.z, bsh.z) : 0.f; v.w = aok[i] ? fmaf(v.w, bsc.w, bsh.w) : 0.f;
      }
      *reinterpret_cast<float4*>(as + (arow + RPA * i) * LDA + 4 * amq) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      float4 v = rb[i];
      if (AFFS != 0) {
        const float4 sc = (AFFS == 2) ? rssc[AFFS == 2 ? i : 0] : ssc;
        const float4 sh = (AFFS == 2) ? rssh[AFFS == 2 ? i : 0] : ssh;
        v.x = bok[i] ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = bok[i] ? fmaf(v.y, sc.y, sh.y) : 0.f;
        v.z = bok[i] ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = bok[i] ? fmaf(v.w, sc.w, sh.w) : 0.f;
      }
      *reinterpret_cast<float4*>(bs + (brow + RPB * i) * LDB + 4 * bnq) = v;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(0);        // nchunks == 0: every row >= r_end -> zeros
  store_chunk(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    load_chunk(c + 1);  // past the end: rows >= r_end -> out-of-range offsets -> zeros, never read
    const float* as = As + cur * A_ELEMS + (8 * lh) * LDA + wm * WTM + l31;
    const float* bs = Bs + cur * B_ELEMS + (8 * lh) * LDB + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = as[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = bs[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    store_chunk(cur ^ 1);
    __syncthreads();
  }

  float* outp = p.Out + (long long)kz * p.M * p.Cs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= p.Cs) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) outp[(long long)m * p.Cs + col] = acc[i][j][r];
      }
    }
  }
}

int veex_wg_tiles(const WGParams& p, int* bm, int* bn) {
  *bn = (p.Cs > 64) ? 128 : 64;
  *bm = (p.M > 64) ? 128 : 64;
  const int t = veex_tune_int("VEEX_WG_TILE", 0);   // tuning: 1 = 64x64, 2 = 128x64, 3 = 64x128
  if (t == 1) { *bm = 64; *bn = 64; }
  else if (t == 2) { *bn = 64; }
  else if (t == 3) { *bm = 64; }
  const int mt = (p.M + *bm - 1) / *bm, nt = (p.Cs + *bn - 1) / *bn;
  return mt * nt;
}

template <int BM, int BN>
static int wg_launch_cfg(const WGParams& pin, bool fast, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  p.remap = veex_tune_int("VEEX_XCD_WG", 1);   // see wgrad_gemm_v3.hip
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  constexpr size_t lds = 2 * VEEX_BK * ((BM + 4) + (BN + 4)) * sizeof(float);
  if (!fast) {
    hipLaunchKernelGGL((wgrad_generic_kernel<BM, BN, 2, 2>), grid, dim3(256), lds, st, p);
  } else {
    const int affs = (p.s_scale == nullptr) ? 0 : (p.s_aff_spatial ? 2 : 1);
    const bool affb = p.b_scale != nullptr;
    if (!affb && affs == 0) hipLaunchKernelGGL((wgrad_kernel<BM, BN, 2, 2, false, 0>), grid, dim3(256), lds, st, p);
    else if (!affb && affs == 1) hipLaunchKernelGGL((wgrad_kernel<BM, BN, 2, 2, false, 1>), grid, dim3(256), lds, st, p);
    else if (!affb && affs == 2) hipLaunchKernelGGL((wgrad_kernel<BM, BN, 2, 2, false, 2>), grid, dim3(256), lds, st, p);
    else if (affb && affs == 0) hipLaunchKernelGGL((wgrad_kernel<BM, BN, 2, 2, true, 0>), grid, dim3(256), lds, st, p);
    else { veex_set_error("wgrad: affine on both operands is not supported"); return -1; }
  }
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_launch_wgrad(const WGParams& p, hipStream_t st) {
  if (p.stride != 1 && p.stride != 2) { veex_set_error("wgrad: stride %d unsupported", p.stride); return -1; }
  if (p.M != p.kh * p.kw * p.Cb || p.R != p.Nb * p.Hs * p.Ws || p.nsplit < 1 ||
      (p.rows_per_split % VEEX_BK) != 0 || (long long)p.rows_per_split * p.nsplit < p.R) {
    veex_set_error("wgrad: inconsistent descriptor (M=%d R=%d nsplit=%d rows_per_split=%d)", p.M, p.R, p.nsplit, p.rows_per_split);
    return -1;
  }
  if (p.kh * p.kw > 1 && (p.Cb & 3) != 0) { veex_set_error("wgrad: multi-tap layers need Cb %% 4 == 0"); return -1; }
  if (veex_wgrad_bf16_ok(p)) return veex_launch_wgrad_bf16(p, st);
  int bm, bn;
  veex_wg_tiles(p, &bm, &bn);
  const long long big_bytes = (long long)p.Nb * p.Hb * p.Wb * p.Cb * 4, small_bytes = (long long)p.R * p.Cs * 4;
  const bool fast = (p.Cb % 4 == 0) && (p.Cs % 4 == 0) && big_bytes < 0x7FFF0000LL && small_bytes < 0x7FFF0000LL &&
                    p.R < (1 << 24) && !(p.b_scale != nullptr && p.b_aff_spatial) &&
                    !(p.b_scale != nullptr && p.s_scale != nullptr) &&
                    (p.s_scale == nullptr || !p.s_aff_spatial || (p.s_aff_mask + 1) == p.Hs * p.Ws * p.Cs);
  if (fast && veex_wgrad_x3_ok(p, bm, bn)) return veex_launch_wgrad_x3(p, bm, bn, st);
  if (fast && veex_tune_int("VEEX_WGV3", 1) && veex_wgrad_v3_ok(p)) return veex_launch_wgrad_v3(p, bm, bn, st);
  if (bm == 128 && bn == 128) return wg_launch_cfg<128, 128>(p, fast, st);
  if (bm == 128 && bn == 64) return wg_launch_cfg<128, 64>(p, fast, st);
  if (bm == 64 && bn == 128) return wg_launch_cfg<64, 128>(p, fast, st);
  return wg_launch_cfg<64, 64>(p, fast, st);
}
