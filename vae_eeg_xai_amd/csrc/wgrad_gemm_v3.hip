// Weight-gradient GEMM on the exact-f32 matrix cores, lean reduction loop (wgrad_gemm.hip has the general
// kernel, the operation and the data layout).
//
// The f32 MFMA does not co-issue with other vector instructions on gfx950 (gather_gemm_v3.hip,
// profiles/r01_micro_mfma_issue.txt), and the general kernel spends ~135 VALU per 32 MFMAs turning a
// reduction row into (sample, y, x), bounds and an address for every load.  Here a block first writes a
// table of its reduction rows into LDS -- per row the byte offset of the pixel under tap (0,0) and an
// inverted tap-validity mask -- and the loop reads it back: one ds_read_b64 + 3 VALU per A load (offset +
// the thread's constant tap/channel displacement, sign-extended mask bit ORed in: out of range -> zeros),
// nothing per B load (constant per-thread offset + a wave-uniform SGPR offset; gfx950 range-checks
// voffset + soffset, tools/micro/soffset_check.hip).  Rows past the split get an all-invalid mask, so
// the A side is zero there and whatever the B side reads is multiplied away.
// Constraints (else the general kernel): Cb % 4 == 0, Cs % 4 == 0, operands < 2 GiB, <= 31 taps,
// rows_per_split <= VEEX_WG3_MAX_ROWS.
#include "veex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f_ __attribute__((__vector_size__(16)));
#define WG3_OOB 0xFFFFFF00u

__device__ __forceinline__ float4 w3load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const v4f_ f = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}

// AFFS: 0 none, 1 per-channel BatchNorm on `small`, 2 BatchNorm over the flattened (h,w,c) vector of `small`
template <int BM, int BN, bool AFFB, int AFFS>
__global__ __launch_bounds__(256) void wgrad_v3_kernel(const WGParams p) {
  constexpr int BK = VEEX_BK;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
  constexpr int QM = BM / 4, QN = BN / 4;
  constexpr int RPA = 256 / QM, RPB = 256 / QN;
  constexpr int APASS = BK / RPA, BPASS = BK / RPB;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  int2* tab = reinterpret_cast<int2*>(Bs + 2 * B_ELEMS);   // [nchunks * BK] (row byte offset, inverted tap mask)
  int* tabS = reinterpret_cast<int*>(tab + ((p.rows_per_split + BK - 1) / BK) * BK);   // AFFS == 2: byte offset of the row's pixel in the affine

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lh = lane >> 5, l31 = lane & 31;

  int bmt, bnt, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bmt, bnt, kz);
  const int m0 = bmt * BM, n0 = bnt * BN;
  const int r_begin = kz * p.rows_per_split;
  const int r_end = min(p.R, r_begin + p.rows_per_split);
  const int nchunks = (r_end > r_begin) ? (r_end - r_begin + BK - 1) / BK : 0;

  // ---- row table ----
  {
    // (float reciprocals + one correction step instead of two 32-bit integer divisions, and the tap mask composed
    //  from a column mask and the valid kernel rows: ~50 instead of ~200 instructions per row -- with up to 12
    //  rows per thread the table cost 8 % of a block's MFMA time)
    const int HWs = p.Hs * p.Ws;
    const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;   // rows < 2^24 (veex_wgrad_v3_ok)
    for (int t = tid; t < nchunks * BK; t += 256) {
      const int r = r_begin + t;
      int off = 0;
      unsigned inv = 0xFFFFFFFFu;
      if (r < r_end) {
        int n = (int)((float)r * invHWs), rem = r - n * HWs;
        if (rem < 0) { n -= 1; rem += HWs; }
        if (rem >= HWs) { n += 1; rem -= HWs; }
        int py = (int)((float)rem * invWs), px = rem - py * p.Ws;
        if (px < 0) { py -= 1; px += p.Ws; }
        if (px >= p.Ws) { py += 1; px -= p.Ws; }
        const int y0 = p.stride * py - p.pb, x0 = p.stride * px - p.pb;
        off = (((n * p.Hb + y0) * p.Wb + x0) * p.Cb) * 4;   // may be "virtual" (negative) at the border
        unsigned xb = 0u, mk = 0u;
        for (int kx = 0; kx < p.kw; ++kx) if (x0 + kx >= 0 && x0 + kx < p.Wb) xb |= 1u << kx;
        for (int ky = 0; ky < p.kh; ++ky) if (y0 + ky >= 0 && y0 + ky < p.Hb) mk |= xb << (ky * p.kw);
        inv = ~mk;
      }
      tab[t] = make_int2(off, (int)inv);
      if (AFFS == 2) tabS[t] = (r % HWs) * p.Cs * 4;
    }
  }

  const __amdgpu_buffer_rsrc_t rBig = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Big), 0, (int)((unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rSmall = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Small), 0, (int)((unsigned)p.R * p.Cs * 4u), 0x00020000);

  // ---- A loader: m-quad (tap, 4 channels) fixed per thread, reduction rows arow + RPA*i of a chunk ----
  const int amq = tid % QM, arow = tid / QM;
  const int am = m0 + 4 * amq;
  const bool am_ok = am < p.M;
  const int atap = am_ok ? am / p.Cb : 31;          // bit 31 of every mask is set: zeros
  const int acb = am - (am / p.Cb) * p.Cb;
  const int aky = atap / p.kw, akx = atap - aky * p.kw;
  const int a_const = am_ok ? ((aky * p.Wb + akx) * p.Cb + acb) * 4 : 0;
  // ---- B loader ----
  const int bnq = tid % QN, brow = tid / QN;
  const int bn = n0 + 4 * bnq;
  const bool bn_ok = bn < p.Cs;
  unsigned b_voff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) b_voff[i] = bn_ok ? (unsigned)(((brow + RPB * i) * p.Cs + bn) * 4) : WG3_OOB;

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
  const unsigned bn4 = bn_ok ? (unsigned)(bn * 4) : WG3_OOB;

  float4 ra[APASS], rb[BPASS], rssc[AFFS == 2 ? BPASS : 1], rssh[AFFS == 2 ? BPASS : 1];
  unsigned akeep[APASS];   // AFFB: all ones where the pixel is real (the BatchNorm shift applies), else 0

  auto load_chunk = [&](int c) {
    const int2* tp = tab + c * BK + arow;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int2 e = tp[RPA * i];
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(e.y, atap, 1);   // 0 or 0xFFFFFFFF
      ra[i] = w3load(rBig, (unsigned)(e.x + a_const) | m, 0);
      if (AFFB) akeep[i] = ~m;
    }
    const int soff = (r_begin + c * BK) * p.Cs * 4;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      rb[i] = w3load(rSmall, b_voff[i], soff);
      if (AFFS == 2) {
        const unsigned ao = (unsigned)tabS[c * BK + brow + RPB * i] + bn4;
        rssc[i] = w3load(rSS, ao, 0);
        rssh[i] = w3load(rSH, ao, 0);
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float4 v = ra[i];
      if (AFFB) {   // padding pixels / rows past the split stay exactly 0
        v.x = __uint_as_float(__float_as_uint(fmaf(v.x, bsc.x, bsh.x)) & akeep[i]);
        v.y = __uint_as_float(__float_as_uint(fmaf(v.y, bsc.y, bsh.y)) & akeep[i]);
        v.z = __uint_as_float(__float_as_uint(fmaf(v.z, bsc.z, bsh.z)) & akeep[i]);
        v.w = __uint_as_float(__float_as_uint(fmaf(v.w, bsc.w, bsh.w)) & akeep[i]);
      }
      *reinterpret_cast<float4*>(as + (arow + RPA * i) * LDA + 4 * amq) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      float4 v = rb[i];
      if (AFFS != 0) {   // rows past the split / the tensor: the A side is 0 there, any finite value will do
        const float4 sc = (AFFS == 2) ? rssc[AFFS == 2 ? i : 0] : ssc, sh = (AFFS == 2) ? rssh[AFFS == 2 ? i : 0] : ssh;
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
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

  __syncthreads();   // row table complete
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    // past the end: the table has no such rows -> clamp to the last chunk (loaded, never stored to a live buffer)
    load_chunk(min(c + 1, nchunks - 1));
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

template <int BM, int BN>
static int wg3_launch(const WGParams& pin, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  // XCD-grouping the tap tiles of one row split (they read the same rows of `big`, shifted) cuts the HBM
  // fetches ~5x; -1 % once no CU holds a block more than the others (net.hip wgrad_split)
  p.remap = veex_tune_int("VEEX_XCD_WG", 1);
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  const int tab_rows = (p.rows_per_split + VEEX_BK - 1) / VEEX_BK * VEEX_BK;
  const bool affb = p.b_scale != nullptr;
  const int affs = (p.s_scale == nullptr) ? 0 : (p.s_aff_spatial ? 2 : 1);
  const size_t lds = 2 * VEEX_BK * ((BM + 4) + (BN + 4)) * sizeof(float) + (size_t)tab_rows * (sizeof(int2) + (affs == 2 ? sizeof(int) : 0));
  if (!affb && affs == 0) hipLaunchKernelGGL((wgrad_v3_kernel<BM, BN, false, 0>), grid, dim3(256), lds, st, p);
  else if (!affb && affs == 1) hipLaunchKernelGGL((wgrad_v3_kernel<BM, BN, false, 1>), grid, dim3(256), lds, st, p);
  else if (!affb && affs == 2) hipLaunchKernelGGL((wgrad_v3_kernel<BM, BN, false, 2>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((wgrad_v3_kernel<BM, BN, true, 0>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_wgrad_v3_ok(const WGParams& p) {
  const long long big_bytes = (long long)p.Nb * p.Hb * p.Wb * p.Cb * 4, small_bytes = (long long)(p.R + VEEX_BK) * p.Cs * 4;
  if (p.Cb % 4 != 0 || p.Cs % 4 != 0 || p.kh * p.kw > 31) return 0;
  if (big_bytes >= 0x7FFF0000LL || small_bytes >= 0x7FFF0000LL) return 0;
  if (p.b_scale != nullptr && (p.s_scale != nullptr || p.b_aff_spatial)) return 0;
  if (p.s_scale != nullptr && p.s_aff_spatial && (p.s_aff_mask + 1) != p.Hs * p.Ws * p.Cs) return 0;
  if (p.rows_per_split > VEEX_WG3_MAX_ROWS) return 0;
  if (p.R >= (1 << 24)) return 0;   // the row table divides with float reciprocals
  return 1;
}

// p: validated by veex_launch_wgrad
int veex_launch_wgrad_v3(const WGParams& p, int bm, int bn, hipStream_t st) {
  if (bm == 128 && bn == 128) return wg3_launch<128, 128>(p, st);
  if (bm == 128 && bn == 64) return wg3_launch<128, 64>(p, st);
  if (bm == 64 && bn == 128) return wg3_launch<64, 128>(p, st);
  return wg3_launch<64, 64>(p, st);
}
