// Weight-gradient GEMM with fp32 results on the bf16 matrix cores: the split-fp32 arithmetic of gather_gemm_x3.hip
// (a = a1 + a2 + a3 in bf16 terms, six partial products per product, fp32 accumulate) on the operation, row table and
// loaders of wgrad_gemm_v3.hip:
//   dW[tap][cb][cs] = sum_r Big(n, s*py+ky-pb, s*px+kx-pb, cb) * Small(r, cs),   r = (n,py,px)
// (the kernel gradients tape.gradient returns for Conv2D / Conv2DTranspose, reference models.py:166).
//
// The reduction index r is the MFMA K dimension while memory is channel-contiguous.  Nothing is transposed while a
// chunk is staged: a thread loads 4 channels of one reduction row (16 bytes), applies the producer's BatchNorm, splits
// the four values and stores 8 bytes into each of three ROW-MAJOR bf16 planes [r][channel]; the MFMA operands (8
// consecutive r of one channel per lane) come out of those planes through ds_read_b64_tr_b16, gfx950's transposing LDS
// read (a 16-lane group reads a 4-row x 16-column block column-major).  Rows are unpadded; the 16-byte pieces of a row
// are XOR-swizzled by the row so that both the 8-byte stores and the transposed reads are free of bank conflicts:
//   128-channel tiles (256-byte rows): piece ^= (row & 3) << 2;   64-channel tiles (128-byte rows): piece ^= (row >> 1 & 1) << 2.
// K chunk = 16 rows = one MFMA k step; 2 x 3 planes x 16 rows x (BM + BN) x 2 B = 48 KB at 128 x 128: three workgroups
// per CU, whose steps interleave (a workgroup's own waves meet at a barrier every chunk).
// Constraints: those of wgrad_gemm_v3.hip, and tiles of 128 x 128 or 128 x 64.
#include "veex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f_ __attribute__((__vector_size__(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define WGX_OOB 0xFFFFFF00u
#define WGX_LDS __attribute__((address_space(3)))

__device__ __forceinline__ float4 wxload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const v4f_ f = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}
// v = h1 + h2 + h3 (+ <= 2^-24 |v|): four values -> three packed 8-byte pieces
__device__ __forceinline__ void wx_split4(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
  const float vv[4] = {v.x, v.y, v.z, v.w};
  unsigned short h1[4], h2[4], h3[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 t1 = (__bf16)vv[e];
    const float r1 = vv[e] - (float)t1;     // exact
    const __bf16 t2 = (__bf16)r1;
    const float r2 = r1 - (float)t2;        // exact
    const __bf16 t3 = (__bf16)r2;
    h1[e] = __builtin_bit_cast(unsigned short, t1);
    h2[e] = __builtin_bit_cast(unsigned short, t2);
    h3[e] = __builtin_bit_cast(unsigned short, t3);
  }
  p1 = make_uint2((unsigned)h1[0] | ((unsigned)h1[1] << 16), (unsigned)h1[2] | ((unsigned)h1[3] << 16));
  p2 = make_uint2((unsigned)h2[0] | ((unsigned)h2[1] << 16), (unsigned)h2[2] | ((unsigned)h2[3] << 16));
  p3 = make_uint2((unsigned)h3[0] | ((unsigned)h3[1] << 16), (unsigned)h3[2] | ((unsigned)h3[3] << 16));
}

// AFFS: 0 none, 1 per-channel BatchNorm on `small`, 2 BatchNorm over the flattened (h,w,c) vector of `small`
template <int BM, int BN, bool AFFB, int AFFS>
__global__ __launch_bounds__(256) void wgrad_x3_kernel(const WGParams p) {
  constexpr int BK = VEEX_BK;
  static_assert(BK == 16, "one v_mfma_f32_32x32x16_bf16 k step per chunk");
  static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "row swizzles exist for 64- and 128-channel tiles");
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int AROWB = BM * 2, BROWB = BN * 2;                  // bytes per LDS row
  constexpr int A_PL = BK * AROWB, B_PL = BK * BROWB;            // bytes per plane
  constexpr int QM = BM / 4, QN = BN / 4;
  constexpr int RPA = 256 / QM, RPB = 256 / QN;
  constexpr int APASS = BK / RPA, BPASS = BK / RPB;
  static_assert(APASS >= 1 && BPASS >= 1, "tile too narrow for 256 threads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                                     // [buf][plane][BK][BM] bf16
  unsigned char* Bs = smem + 2 * 3 * A_PL;                      // [buf][plane][BK][BN] bf16
  int* tab = reinterpret_cast<int*>(Bs + 2 * 3 * B_PL);         // [nchunks * BK] packed rows (below)
  int* tabS = tab + ((p.rows_per_split + BK - 1) / BK) * BK;   // AFFS == 2: byte offset of the row's pixel in the affine

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

  // ---- row table (the idea of wgrad_gemm_v3.hip, 4 bytes per row so that 1 200 rows still leave room for three workgroups
  // per CU): (pixel under tap (0,0) minus the pixel of the split's first row) << 9 | inverted tap-validity mask ----
  const int HWs = p.Hs * p.Ws;
  const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;   // rows < 2^24 (veex_wgrad_x3_ok)
  auto row_pixel = [&](int r, int& y0, int& x0) {
    int n = (int)((float)r * invHWs), rem = r - n * HWs;
    if (rem < 0) { n -= 1; rem += HWs; }
    if (rem >= HWs) { n += 1; rem -= HWs; }
    int py = (int)((float)rem * invWs), px = rem - py * p.Ws;
    if (px < 0) { py -= 1; px += p.Ws; }
    if (px >= p.Ws) { py += 1; px -= p.Ws; }
    y0 = p.stride * py - p.pb; x0 = p.stride * px - p.pb;
    return (n * p.Hb + y0) * p.Wb + x0;        // may be "virtual" (negative / outside its image) at the border
  };
  int y0b, x0b;
  const int pix_base = row_pixel(min(r_begin, p.R - 1), y0b, x0b);     // (block-uniform)
  for (int t = tid; t < nchunks * BK; t += 256) {
    const int r = r_begin + t;
    int e = 0x1FF;
    if (r < r_end) {
      int y0, x0;
      const int pix = row_pixel(r, y0, x0);
      unsigned xb = 0u, mk = 0u;
      for (int kx = 0; kx < p.kw; ++kx) if (x0 + kx >= 0 && x0 + kx < p.Wb) xb |= 1u << kx;
      for (int ky = 0; ky < p.kh; ++ky) if (y0 + ky >= 0 && y0 + ky < p.Hb) mk |= xb << (ky * p.kw);
      e = ((pix - pix_base) << 9) | (int)(~mk & 0x1FFu);
    }
    tab[t] = e;
    if (AFFS == 2) tabS[t] = (r % HWs) * p.Cs * 4;
  }

  const __amdgpu_buffer_rsrc_t rBig = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Big), 0, (int)((unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rSmall = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Small), 0, (int)((unsigned)p.R * p.Cs * 4u), 0x00020000);

  // ---- A loader: m-quad (tap, 4 channels) fixed per thread, reduction rows arow + RPA*i of a chunk ----
  const int amq = tid % QM, arow = tid / QM;
  const int am = m0 + 4 * amq;
  const bool am_ok = am < p.M;
  const int atap = am_ok ? am / p.Cb : 0;
  const unsigned a_dead = am_ok ? 0u : 0xFFFFFFFFu;  // a thread past M loads zeros
  const int acb = am - (am / p.Cb) * p.Cb;
  const int aky = atap / p.kw, akx = atap - aky * p.kw;
  const int cb4 = p.Cb * 4;
  const int a_const = am_ok ? (pix_base * p.Cb + (aky * p.Wb + akx) * p.Cb + acb) * 4 : 0;
  // ---- B loader ----
  const int bnq = tid % QN, brow = tid / QN;
  const int bn = n0 + 4 * bnq;
  const bool bn_ok = bn < p.Cs;
  unsigned b_voff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) b_voff[i] = bn_ok ? (unsigned)(((brow + RPB * i) * p.Cs + bn) * 4) : WGX_OOB;

  // ---- LDS store offsets: row-major planes, 16-byte piece (2 quads) swizzled by the row ----
  auto swz = [](int row, int width) { return width == 128 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; };
  int a_st[APASS], b_st[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int row = arow + RPA * i;
    a_st[i] = row * AROWB + 16 * ((amq >> 1) ^ swz(row, BM)) + 8 * (amq & 1);
  }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int row = brow + RPB * i;
    b_st[i] = row * BROWB + 16 * ((bnq >> 1) ^ swz(row, BN)) + 8 * (bnq & 1);
  }
  // ---- transposed operand reads: 16-lane group g reads the block of rows 8*(g>>1) + 4*t .. + 3 (t = 0, 1: two reads per
  // operand), columns tile base + 16*(g&1) .. + 15; lane 4q + pq of the group supplies the address of row q, columns
  // 4*pq .. 4*pq + 3 and receives column (lane & 15), rows 0..3: the 8 k of lane half lh, channel l31 of the MFMA operand
  const int g = lane >> 4, tq = (lane & 15) >> 2, tpq = lane & 3;
  int a_rd[MT], b_rd[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = 8 * (g >> 1) + tq;                                   // (+ 4 for the second read: same swizzle)
    const int piece = (wm * WTM + 32 * i) / 8 + 2 * (g & 1) + (tpq >> 1);
    a_rd[i] = row * AROWB + 16 * (piece ^ swz(row, BM)) + 8 * (tpq & 1);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = 8 * (g >> 1) + tq;
    const int piece = (wn * WTN + 32 * j) / 8 + 2 * (g & 1) + (tpq >> 1);
    b_rd[j] = row * BROWB + 16 * (piece ^ swz(row, BN)) + 8 * (tpq & 1);
  }

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
  const unsigned bn4 = bn_ok ? (unsigned)(bn * 4) : WGX_OOB;

  float4 ra[APASS], rb[BPASS], rssc[AFFS == 2 ? BPASS : 1], rssh[AFFS == 2 ? BPASS : 1];
  unsigned akeep[APASS];   // AFFB: all ones where the pixel is real (the BatchNorm shift applies), else 0

  auto load_chunk = [&](int c) {
    const int* tp = tab + c * BK + arow;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int e = tp[RPA * i];
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(e, atap, 1) | a_dead;   // 0 or 0xFFFFFFFF
      ra[i] = wxload(rBig, (unsigned)(__mul24(e >> 9, cb4) + a_const) | m, 0);
      if (AFFB) akeep[i] = ~m;
    }
    const int soff = (r_begin + c * BK) * p.Cs * 4;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      rb[i] = wxload(rSmall, b_voff[i], soff);
      if (AFFS == 2) {
        const unsigned ao = (unsigned)tabS[c * BK + brow + RPB * i] + bn4;
        rssc[i] = wxload(rSS, ao, 0);
        rssh[i] = wxload(rSH, ao, 0);
      }
    }
  };

  auto store_chunk = [&](int buf) {
    unsigned char* as = As + buf * 3 * A_PL;
    unsigned char* bs = Bs + buf * 3 * B_PL;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float4 v = ra[i];
      if (AFFB) {   // padding pixels / rows past the split stay exactly 0
        v.x = __uint_as_float(__float_as_uint(fmaf(v.x, bsc.x, bsh.x)) & akeep[i]);
        v.y = __uint_as_float(__float_as_uint(fmaf(v.y, bsc.y, bsh.y)) & akeep[i]);
        v.z = __uint_as_float(__float_as_uint(fmaf(v.z, bsc.z, bsh.z)) & akeep[i]);
        v.w = __uint_as_float(__float_as_uint(fmaf(v.w, bsc.w, bsh.w)) & akeep[i]);
      }
      uint2 p1, p2, p3;
      wx_split4(v, p1, p2, p3);
      *reinterpret_cast<uint2*>(as + a_st[i]) = p1;
      *reinterpret_cast<uint2*>(as + A_PL + a_st[i]) = p2;
      *reinterpret_cast<uint2*>(as + 2 * A_PL + a_st[i]) = p3;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      float4 v = rb[i];
      if (AFFS != 0) {   // rows past the split / the tensor: the A side is 0 there, any finite value will do
        const float4 sc = (AFFS == 2) ? rssc[AFFS == 2 ? i : 0] : ssc, sh = (AFFS == 2) ? rssh[AFFS == 2 ? i : 0] : ssh;
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
      uint2 p1, p2, p3;
      wx_split4(v, p1, p2, p3);
      *reinterpret_cast<uint2*>(bs + b_st[i]) = p1;
      *reinterpret_cast<uint2*>(bs + B_PL + b_st[i]) = p2;
      *reinterpret_cast<uint2*>(bs + 2 * B_PL + b_st[i]) = p3;
    }
  };

  auto tr_read = [](const unsigned char* base) {      // 8 consecutive k (two 4-row blocks) of this lane's channel
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WGX_LDS v4s*)(base));
    return lo;
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
    const unsigned char* as = As + cur * 3 * A_PL;
    const unsigned char* bs = Bs + cur * 3 * B_PL;
    bf16x8 av[3][MT], bv[3][NT];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const v4s lo = tr_read(as + pl * A_PL + a_rd[i]), hi = tr_read(as + pl * A_PL + a_rd[i] + 4 * AROWB);
        av[pl][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const v4s lo = tr_read(bs + pl * B_PL + b_rd[j]), hi = tr_read(bs + pl * B_PL + b_rd[j] + 4 * BROWB);
        bv[pl][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
    // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t]][i], bv[PB[t]][j], acc[i][j], 0, 0, 0);
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
static int wgx_launch(const WGParams& pin, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  p.remap = 1;     // tap tiles of one row split on one XCD (wgrad_gemm_v3.hip)
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  const int tab_rows = (p.rows_per_split + VEEX_BK - 1) / VEEX_BK * VEEX_BK;
  const bool affb = p.b_scale != nullptr;
  const int affs = (p.s_scale == nullptr) ? 0 : (p.s_aff_spatial ? 2 : 1);
  const size_t lds = (size_t)2 * 3 * VEEX_BK * (BM + BN) * 2 + (size_t)tab_rows * (sizeof(int) + (affs == 2 ? sizeof(int) : 0));
  static bool attr_done = false;   // function attributes are per process; set them once, outside any capture
  if (!attr_done) {
    const int cap = 2 * 3 * VEEX_BK * (BM + BN) * 2 + VEEX_WG3_MAX_ROWS * 8;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<BM, BN, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<BM, BN, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<BM, BN, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<BM, BN, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
      veex_set_error("wgrad(x3): cannot raise the dynamic LDS limit"); return -3;
    }
    attr_done = true;
  }
  if (!affb && affs == 0) hipLaunchKernelGGL((wgrad_x3_kernel<BM, BN, false, 0>), grid, dim3(256), lds, st, p);
  else if (!affb && affs == 1) hipLaunchKernelGGL((wgrad_x3_kernel<BM, BN, false, 1>), grid, dim3(256), lds, st, p);
  else if (!affb && affs == 2) hipLaunchKernelGGL((wgrad_x3_kernel<BM, BN, false, 2>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((wgrad_x3_kernel<BM, BN, true, 0>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// the split kernel takes the launch (p.x3 set by the caller; p validated by veex_launch_wgrad; bm x bn from veex_wg_tiles)
int veex_wgrad_x3_ok(const WGParams& p, int bm, int bn) {
  if (!p.x3 || p.bf16) return 0;
  if (bm != 128 || (bn != 128 && bn != 64)) return 0;
  if (p.kh * p.kw > 9 || p.Cb > 4096) return 0;          // 9 mask bits per table row; the row's pixel delta times Cb * 4 in 24-bit operands
  if ((long long)p.rows_per_split * p.stride * p.stride + 4LL * p.Wb * (p.kh + p.stride) >= (1LL << 21)) return 0;   // pixel delta within a split
  return veex_wgrad_v3_ok(p);
}

int veex_launch_wgrad_x3(const WGParams& p, int bm, int bn, hipStream_t st) {
  if (bn == 128) return wgx_launch<128, 128>(p, st);
  return wgx_launch<128, 64>(p, st);
}
