// Weight-gradient GEMM of the bf16 engine on bf16 consumer copies (WGParams::BigBf / SmallBf, BatchNorm already applied),
// staged WITHOUT a transpose: the layout idea of wgrad_gemm_x3.hip with one plane.
//   dW[tap][cb][cs] = sum_r Big(n, s*py+ky-pb, s*px+kx-pb, cb) * Small(r, cs),   r = (n,py,px)
// (the kernel gradients tape.gradient returns for Conv2D / Conv2DTranspose, reference models.py:166).
//
// wgrad_bf16c_kernel (wgrad_gemm_bf16.hip) turns every 8-row x 8-channel piece in registers before it writes [channel][k]
// LDS tiles, and that staging VALU work bounds it.  Here a thread copies 8 channels of one reduction row (16 bytes)
// straight into a ROW-MAJOR tile [r][channel]; the MFMA operands (8 consecutive r of one channel per lane) come out of it
// through ds_read_b64_tr_b16, gfx950's transposing LDS read.  Rows are unpadded, their 16-byte pieces XOR-swizzled by
// the row (256-byte rows: piece ^= (row & 3) << 2; 128-byte rows: piece ^= (row >> 1 & 1) << 2): the 16-byte stores and
// the transposed reads are free of bank conflicts.  K chunk = BK rows (BK / 16 MFMA k steps); 2 x BK x (BM + BN) x 2 B of
// LDS = 64 KB at 128 x 128 x 64 (the shipped depth) plus the 4-byte-per-row table of wgrad_gemm_x3.hip: two workgroups per CU
// (32 rows deep and three per CU measures the same).
#include "veex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int v4u_ __attribute__((__vector_size__(16)));
#define WGT_OOB 0xFFFFFF00u
#define WGT_LDS __attribute__((address_space(3)))
#define WGT_MAX_ROWS VEEX_WGT_MAX_ROWS

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void wgrad_bf16t_kernel(const WGParams p) {
  static_assert(BK % 16 == 0 && BM == 128 && (BN == 128 || BN == 64), "row swizzles exist for 64- and 128-channel tiles");
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int AROWB = BM * 2, BROWB = BN * 2;                  // bytes per LDS row
  constexpr int A_BYTES = BK * AROWB, B_BYTES = BK * BROWB;      // bytes per buffer
  constexpr int OM = BM / 8, ON = BN / 8;                        // 16-byte pieces (8 channels) per row
  constexpr int RPA = 256 / OM, RPB = 256 / ON;
  constexpr int APASS = BK / RPA, BPASS = (BK + RPB - 1) / RPB;
  static_assert(APASS >= 1 && BK % RPA == 0, "chunk too short for 256 threads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                                     // [buf][BK][BM] bf16
  unsigned char* Bs = smem + 2 * A_BYTES;                       // [buf][BK][BN] bf16
  int* tab = reinterpret_cast<int*>(Bs + 2 * B_BYTES);          // [nchunks * BK] packed rows (below)

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

  // ---- row table: (pixel under tap (0,0) minus the pixel of the split's first row) << 9 | inverted tap-validity mask ----
  const int HWs = p.Hs * p.Ws;
  const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;   // rows < 2^24 (veex_wgrad_bf16t_ok)
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
  }

  const __amdgpu_buffer_rsrc_t rBig = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.BigBf), 0, (int)((unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rSmall = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.SmallBf), 0, (int)((unsigned)p.R * p.Cs * 2u), 0x00020000);

  // ---- A loader: channel octet (tap, 8 channels) fixed per thread, reduction rows arow + RPA*i of a chunk ----
  const int aoc = tid % OM, arow = tid / OM;
  const int am = m0 + 8 * aoc;
  const bool am_ok = am < p.M;
  const int atap = am_ok ? am / p.Cb : 0;
  const unsigned a_dead = am_ok ? 0u : 0xFFFFFFFFu;  // a thread past M loads zeros
  const int acb = am - (am / p.Cb) * p.Cb;
  const int aky = atap / p.kw, akx = atap - aky * p.kw;
  const int cb2 = p.Cb * 2;
  const int a_const = am_ok ? (pix_base * p.Cb + (aky * p.Wb + akx) * p.Cb + acb) * 2 : 0;
  // ---- B loader ----
  const int boc = tid % ON, brow = tid / ON;
  const int bn = n0 + 8 * boc;
  const bool bn_ok = bn < p.Cs;
  unsigned b_voff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) b_voff[i] = (bn_ok && brow + RPB * i < BK) ? (unsigned)(((brow + RPB * i) * p.Cs + bn) * 2) : WGT_OOB;

  // ---- LDS store offsets: row-major tile, 16-byte piece swizzled by the row ----
  auto swz = [](int row, int width) { return width == 128 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; };
  int a_st[APASS], b_st[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int row = arow + RPA * i;
    a_st[i] = row * AROWB + 16 * (aoc ^ swz(row, BM));
  }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int row = brow + RPB * i;
    b_st[i] = row * BROWB + 16 * (boc ^ swz(row, BN));
  }
  // ---- transposed operand reads (wgrad_gemm_x3.hip): 16-lane group g reads rows 8*(g>>1) + 4*t .. + 3 (t = 0, 1), columns
  // tile base + 16*(g&1) .. + 15; lane 4q + pq of the group supplies the address of row q, columns 4*pq .. 4*pq + 3
  const int g = lane >> 4, tq = (lane & 15) >> 2, tpq = lane & 3;
  int a_rd[MT], b_rd[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = 8 * (g >> 1) + tq;                                   // (+ 4, + 16 ks: same swizzle)
    const int piece = (wm * WTM + 32 * i) / 8 + 2 * (g & 1) + (tpq >> 1);
    a_rd[i] = row * AROWB + 16 * (piece ^ swz(row, BM)) + 8 * (tpq & 1);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = 8 * (g >> 1) + tq;
    const int piece = (wn * WTN + 32 * j) / 8 + 2 * (g & 1) + (tpq >> 1);
    b_rd[j] = row * BROWB + 16 * (piece ^ swz(row, BN)) + 8 * (tpq & 1);
  }

  v4u_ ra[APASS], rb[BPASS];

  auto load_chunk = [&](int c) {
    const int* tp = tab + c * BK + arow;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int e = tp[RPA * i];
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(e, atap, 1) | a_dead;   // 0 or 0xFFFFFFFF
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rBig, (unsigned)(__mul24(e >> 9, cb2) + a_const) | m, 0, 0);
    }
    const int soff = (r_begin + c * BK) * p.Cs * 2;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rSmall, b_voff[i], soff, 0);
  };

  auto store_chunk = [&](int buf) {
    unsigned char* as = As + buf * A_BYTES;
    unsigned char* bs = Bs + buf * B_BYTES;
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<v4u_*>(as + a_st[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      if (BK % RPB == 0 || brow + RPB * i < BK) *reinterpret_cast<v4u_*>(bs + b_st[i]) = rb[i];
  };

  auto tr8 = [](const unsigned char* base, int rowb) {      // 8 consecutive k (two 4-row blocks) of this lane's channel
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WGT_LDS v4s*)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WGT_LDS v4s*)(base + 4 * rowb));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
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
    const unsigned char* as = As + cur * A_BYTES;
    const unsigned char* bs = Bs + cur * B_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = tr8(as + a_rd[i] + 16 * ks * AROWB, AROWB);
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = tr8(bs + b_rd[j] + 16 * ks * BROWB, BROWB);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
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

template <int BM, int BN, int BK>
static int wgt_launch(const WGParams& pin, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  p.remap = 1;     // tap tiles of one row split on one XCD (wgrad_gemm_v3.hip)
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  const int tab_rows = (p.rows_per_split + BK - 1) / BK * BK;
  const size_t lds = (size_t)2 * BK * (BM + BN) * 2 + (size_t)tab_rows * sizeof(int);
  static bool attr_done = false;   // function attributes are per process; set them once, outside any capture
  if (!attr_done) {
    const int cap = 2 * BK * (BM + BN) * 2 + (WGT_MAX_ROWS + BK) * 4;
    if (cap > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16t_kernel<BM, BN, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
      veex_set_error("wgrad(bf16t): cannot raise the dynamic LDS limit"); return -3;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_bf16t_kernel<BM, BN, BK>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// the transposing-read kernel takes a launch the copy kernel would take (veex_wgrad_bf16c_ok) when its row table fits
int veex_wgrad_bf16t_ok(const WGParams& p) {
  if (!veex_tune_int("VEEX_WGBT", 1)) return 0;
  if (p.kh * p.kw > 9 || p.Cb > 4096 || p.rows_per_split > WGT_MAX_ROWS) return 0;
  if ((long long)p.rows_per_split * p.stride * p.stride + 4LL * p.Wb * (p.kh + p.stride) >= (1LL << 21)) return 0;   // pixel delta within a split
  return 1;
}

int veex_launch_wgrad_bf16t(const WGParams& p, hipStream_t st) {
  const int bk = veex_tune_int("VEEX_WGBT_BK", 64);
  if (p.Cs > 64) return bk == 64 ? wgt_launch<128, 128, 64>(p, st) : wgt_launch<128, 128, 32>(p, st);
  return bk == 64 ? wgt_launch<128, 64, 64>(p, st) : wgt_launch<128, 64, 32>(p, st);
}
