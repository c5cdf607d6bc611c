// bf16 variant of the weight-gradient GEMM (wgrad_gemm.hip): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
//   dW[tap][cb][cs] = sum_r Big(n, s*py+ky-pb, s*px+kx-pb, cb) * Small(r, cs),   r = (n,py,px)
// (the kernel gradients tape.gradient returns for Conv2D / Conv2DTranspose, reference models.py:166).
//
// The reduction index r is the MFMA K dimension, but memory is channel-contiguous, so both operands
// are transposed while they are staged: a thread loads a 4-channel x 8-row piece with eight 16-byte
// buffer loads (offsets forced out of range for padding pixels / rows past the split), applies the
// producer's BatchNorm affine, rounds to bf16 and writes four 16-byte k-runs into [channel][k] LDS
// tiles; threads 0..BM-1 stage Big, threads BM..BM+BN-1 stage Small.  LDS row rho = e*(T/4) + quad
// holds channel 4*quad + e of a T-channel tile: the 8 lanes of a ds_write_b128 group then hit 8
// distinct 16-byte bank groups (rows 80 B apart), and the MFMA reads rows rho = base + lane
// (conflict-free ds_read_b128); the permutation is undone when the accumulators are stored.
#include "gg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, bool AFFB, bool AFFS>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WGParams p) {
  constexpr int BK = 32, LDKH = BK + 8;
  constexpr int WM = 2, WN = 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int QM = BM / 4, QN = BN / 4;
  constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;
  static_assert(BM + BN <= 256 && BM % 64 == 0 && BN % 64 == 0, "one 4-channel x 8-row piece per thread");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);      // [2][BM][40]
  __bf16* Bs = As + 2 * A_ELEMS;                         // [2][BN][40]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the role below is per wave
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bmt, bnt, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bmt, bnt, kz);
  const int m0 = bmt * BM, n0 = bnt * BN;
  const int r_begin = kz * p.rows_per_split;
  const int r_end = min(p.R, r_begin + p.rows_per_split);
  const int nchunks = (r_end > r_begin) ? (r_end - r_begin + BK - 1) / BK : 0;

  // ---- role of this thread (uniform per wave: BM, BN are multiples of 64) ----
  const bool isA = wave * 64 < BM;
  const bool active = wave * 64 < BM + BN;
  const int u = isA ? tid : tid - BM;
  const int QT = isA ? QM : QN;
  const int quad = u % QT, rg = u / QT;          // 4-channel group, 8-row group (0..3)
  const int ch0 = (isA ? m0 : n0) + 4 * quad;    // first of the 4 channels (A: index into M = tap*Cb + cb)
  int C, H, W, str, pb, ky = 0, kx = 0, cch;
  bool ch_ok;
  if (isA) {
    C = p.Cb; H = p.Hb; W = p.Wb; str = p.stride; pb = p.pb;
    ch_ok = active && ch0 < p.M;
    const int tap = ch0 / p.Cb;
    cch = ch0 - tap * p.Cb;
    ky = tap / p.kw; kx = tap - ky * p.kw;
  } else {
    C = p.Cs; H = p.Hs; W = p.Ws; str = 1; pb = 0;
    ch_ok = active && ch0 < p.Cs;
    cch = ch0;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(isA ? p.Big : p.Small), 0,
      (int)(isA ? (unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 4u : (unsigned)p.R * p.Cs * 4u), 0x00020000);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (AFFB && isA && ch_ok) {
    sc = *reinterpret_cast<const float4*>(p.b_scale + (cch & p.b_aff_mask));
    sh = *reinterpret_cast<const float4*>(p.b_shift + (cch & p.b_aff_mask));
  }
  if (AFFS && !isA && ch_ok) {
    sc = *reinterpret_cast<const float4*>(p.s_scale + (cch & p.s_aff_mask));
    sh = *reinterpret_cast<const float4*>(p.s_shift + (cch & p.s_aff_mask));
  }
  const bool has_aff = isA ? AFFB : AFFS;
  const int HWs = p.Hs * p.Ws;
  const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;
  __bf16* const lds_dst = (isA ? As : Bs) + quad * LDKH + 8 * rg;     // + buf*ELEMS + e*QT*LDKH
  const int lds_estride = QT * LDKH;
  const int lds_elems = isA ? A_ELEMS : B_ELEMS;

  float4 ra[8];
  unsigned okmask = 0u;

  auto load_chunk = [&](int c) {
    const int r0 = r_begin + c * BK + 8 * rg;
    int n, rem, py, px;
    {
      int q1 = (int)((float)r0 * invHWs); int m1 = r0 - q1 * HWs;
      if (m1 < 0) { q1 -= 1; m1 += HWs; }
      if (m1 >= HWs) { q1 += 1; m1 -= HWs; }
      n = q1; rem = m1;
      int q2 = (int)((float)rem * invWs); int m2 = rem - q2 * p.Ws;
      if (m2 < 0) { q2 -= 1; m2 += p.Ws; }
      if (m2 >= p.Ws) { q2 += 1; m2 -= p.Ws; }
      py = q2; px = m2;
    }
    okmask = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int by = str * py + ky - pb, bx = str * px + kx - pb;
      const bool ok = ch_ok && (r0 + k) < r_end && by >= 0 && by < H && bx >= 0 && bx < W;
      ra[k] = bload4(rsrc, ok ? (unsigned)((((n * H + by) * W + bx) * C + cch) * 4) : VEEX_OOB);
      okmask |= ok ? (1u << k) : 0u;
      // next row of the (n, py, px) walk
      px += 1;
      if (px == p.Ws) { px = 0; py += 1; if (py == p.Hs) { py = 0; n += 1; } }
    }
  };

  auto store_chunk = [&](int buf) {
    if (!active) return;
    float vx[8], vy[8], vz[8], vw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float4 v = ra[k];
      if (has_aff) {   // padding pixels / rows past the split stay 0 (the shift applies to real pixels only)
        const bool ok = (okmask >> k) & 1u;
        v.x = ok ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = ok ? fmaf(v.y, sc.y, sh.y) : 0.f;
        v.z = ok ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = ok ? fmaf(v.w, sc.w, sh.w) : 0.f;
      }
      vx[k] = v.x; vy[k] = v.y; vz[k] = v.z; vw[k] = v.w;
    }
    bf16x8 h0, h1, h2, h3;
#pragma unroll
    for (int k = 0; k < 8; ++k) { h0[k] = (__bf16)vx[k]; h1[k] = (__bf16)vy[k]; h2[k] = (__bf16)vz[k]; h3[k] = (__bf16)vw[k]; }
    __bf16* d = lds_dst + buf * lds_elems;
    *reinterpret_cast<bf16x8*>(d) = h0;
    *reinterpret_cast<bf16x8*>(d + lds_estride) = h1;
    *reinterpret_cast<bf16x8*>(d + 2 * lds_estride) = h2;
    *reinterpret_cast<bf16x8*>(d + 3 * lds_estride) = h3;
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
    const __bf16* as = As + cur * A_ELEMS + (wm * WTM + l31) * LDKH + 8 * lh;
    const __bf16* bs = Bs + cur * B_ELEMS + (wn * WTN + l31) * LDKH + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDKH + 16 * ks);
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDKH + 16 * ks);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    store_chunk(cur ^ 1);
    __syncthreads();
  }

  // tile row rho -> channel 4*(rho % (T/4)) + rho / (T/4)
  float* outp = p.Out + (long long)kz * p.M * p.Cs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int rn = wn * WTN + j * 32 + l31;
    const int col = n0 + 4 * (rn % QN) + rn / QN;
    if (col >= p.Cs) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rm = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + 4 * (rm % QM) + rm / QM;
        if (m < p.M) outp[(long long)m * p.Cs + col] = acc[i][j][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Variant for bf16 consumer copies (WGParams::BigBf / SmallBf, written by the producers' epilogues with
// the BatchNorm already applied): no affine, no conversion.  A thread stages an 8-row x 8-channel
// piece: eight 16-byte loads, an in-register 8x8 transpose of 16-bit elements, eight 16-byte k-runs
// into the [channel][k] LDS tile.  K chunk = 64 rows: waves 0-1 stage Big (16 octets x 8 row groups),
// waves 2-3 stage Small, every wave multiplies.  LDS rows are 64 bf16 + 8 pad = 144 B = 9 sixteen-byte
// slots (odd): both the permuted-row writes (rho = e*(T/8) + octet) and the operand reads are
// conflict-free.  Row -> pixel address and tap validity come from a windowed LDS row table (512 rows, rebuilt
// every 8 chunks; the wgrad_gemm_v3.hip idea): 3 VALU per 16-byte load instead of ~20 (the staging VALU work,
// not the MFMAs, bounded this kernel).
// ---------------------------------------------------------------------------
#define WGBC_TABW 512   // rows per window of the row table (8 chunks)
template <int BM, int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_bf16c_kernel(const WGParams p) {
  constexpr int BK = 64, LDKH = BK + 8;
  constexpr int WM = 2, WN = 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int OM = BM / 8, ON = BN / 8;            // channel octets per tile
  constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;
  static_assert(BM == 128 && (BN == 128 || BN == 64), "waves 0-1 stage A, waves 2-3 stage B");
  static_assert(WGBC_TABW % BK == 0, "a chunk never straddles two table windows");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);      // [2][BM][72]
  __bf16* Bs = As + 2 * A_ELEMS;                         // [2][BN][72]
  int2* tab = reinterpret_cast<int2*>(Bs + 2 * B_ELEMS); // [WGBC_TABW] (byte offset of the row's pixel in Big, ~valid taps)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bmt, bnt, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bmt, bnt, kz);
  const int m0 = bmt * BM, n0 = bnt * BN;
  const int r_begin = kz * p.rows_per_split;
  const int r_end = min(p.R, r_begin + p.rows_per_split);
  const int nchunks = (r_end > r_begin) ? (r_end - r_begin + BK - 1) / BK : 0;

  const bool isA = wave < 2;
  const int u = isA ? tid : tid - 128;
  const int OT = isA ? OM : ON;
  const int oct = u % OT, rg = u / OT;                   // channel octet, 8-row group
  const bool active = rg < 8;                            // BN = 64: wave 3 has nothing to stage
  const int ch0 = (isA ? m0 : n0) + 8 * oct;
  // A (Big) threads: the tap and channel of the octet are fixed, the pixel of a row comes from the row table:
  //   address = tab[row].x + a_const, forced out of range (-> zeros) by the row's inverted tap mask
  // B (Small) threads: rows are linear in memory: a constant per-row offset + the chunk's base in the scalar offset
  int atap = 31, a_const = 0;      // bit 31 of every mask is set: zeros
  unsigned b_voff[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) b_voff[k] = VEEX_OOB;
  if (isA) {
    if (active && ch0 < p.M) {
      atap = ch0 / p.Cb;
      const int cch = ch0 - atap * p.Cb;
      const int ky = atap / p.kw, kx = atap - ky * p.kw;
      a_const = ((ky * p.Wb + kx) * p.Cb + cch) * 2;
    }
  } else if (active && ch0 < p.Cs) {
#pragma unroll
    for (int k = 0; k < 8; ++k) b_voff[k] = (unsigned)(((8 * rg + k) * p.Cs + ch0) * 2);
  }
  // one resource per role, each built from kernel arguments only: a per-thread select of the two made the
  // descriptor a VGPR value and every load a readfirstlane "waterfall" loop (~12 extra instructions per load)
  const __amdgpu_buffer_rsrc_t rBig = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.BigBf), 0, (int)((unsigned)p.Nb * p.Hb * p.Wb * p.Cb * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rSmall = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.SmallBf), 0, (int)((unsigned)p.R * p.Cs * 2u), 0x00020000);
  const int HWs = p.Hs * p.Ws;
  const float invHWs = 1.0f / (float)HWs, invWs = 1.0f / (float)p.Ws;
  __bf16* const lds_dst = (isA ? As : Bs) + oct * LDKH + 8 * rg;     // + buf*ELEMS + e*OT*LDKH
  const int lds_estride = OT * LDKH;
  const int lds_elems = isA ? A_ELEMS : B_ELEMS;

  // window w of the row table: rows r_begin + w*TABW ... ; rows past the split read zeros for every tap
  auto build_window = [&](int w) {
    for (int t = tid; t < WGBC_TABW; t += 256) {
      const int r = r_begin + w * WGBC_TABW + t;
      int off = 0;
      unsigned inv = 0xFFFFFFFFu;
      if (r < r_end) {
        int q1 = (int)((float)r * invHWs); int m1 = r - q1 * HWs;
        if (m1 < 0) { q1 -= 1; m1 += HWs; }
        if (m1 >= HWs) { q1 += 1; m1 -= HWs; }
        int q2 = (int)((float)m1 * invWs); int m2 = m1 - q2 * p.Ws;
        if (m2 < 0) { q2 -= 1; m2 += p.Ws; }
        if (m2 >= p.Ws) { q2 += 1; m2 -= p.Ws; }
        const int y0 = p.stride * q2 - p.pb, x0 = p.stride * m2 - p.pb;
        off = (((q1 * p.Hb + y0) * p.Wb + x0) * p.Cb) * 2;   // may be "virtual" (negative) at the border
        unsigned xb = 0u, mk = 0u;
        for (int kx = 0; kx < p.kw; ++kx) if (x0 + kx >= 0 && x0 + kx < p.Wb) xb |= 1u << kx;
        for (int ky = 0; ky < p.kh; ++ky) if (y0 + ky >= 0 && y0 + ky < p.Hb) mk |= xb << (ky * p.kw);
        inv = ~mk;
      }
      tab[t] = make_int2(off, (int)inv);
    }
  };

  struct Stage { v4u ra[8]; };

  auto load_chunk = [&](Stage& sg, int c) {
    v4u (&ra)[8] = sg.ra;
    if ((c & (WGBC_TABW / BK - 1)) == 0 && c > 0) {   // every earlier chunk has read its table entries
      __syncthreads();
      build_window(c / (WGBC_TABW / BK));
      __syncthreads();
    }
    if (isA) {
      const int2* tp = tab + ((c * BK) & (WGBC_TABW - 1)) + 8 * rg;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int2 e = tp[k];
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe(e.y, atap, 1);   // 0 or 0xFFFFFFFF
        ra[k] = __builtin_amdgcn_raw_buffer_load_b128(rBig, (unsigned)(e.x + a_const) | m, 0, 0);
      }
    } else {
      const int soff = (r_begin + c * BK) * p.Cs * 2;
#pragma unroll
      for (int k = 0; k < 8; ++k) ra[k] = __builtin_amdgcn_raw_buffer_load_b128(rSmall, b_voff[k], soff, 0);
    }
  };

  // ra[k][d] = channels (2d, 2d+1) of row k  ->  per channel the 8 rows as four dwords (k0k1, k2k3, k4k5, k6k7)
  auto store_chunk = [&](const Stage& sg, int buf) {
    if (!active) return;
    const v4u (&ra)[8] = sg.ra;
    __bf16* d = lds_dst + buf * lds_elems;
#pragma unroll
    for (int dp = 0; dp < 4; ++dp) {
      v4u lo, hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned a = ra[2 * j][dp], b = ra[2 * j + 1][dp];
        lo[j] = (a & 0xFFFFu) | (b << 16);
        hi[j] = (a >> 16) | (b & 0xFFFF0000u);
      }
      *reinterpret_cast<v4u*>(d + (2 * dp) * lds_estride) = lo;
      *reinterpret_cast<v4u*>(d + (2 * dp + 1) * lds_estride) = hi;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int cur) {
    const __bf16* as = As + cur * A_ELEMS + (wm * WTM + l31) * LDKH + 8 * lh;
    const __bf16* bs = Bs + cur * B_ELEMS + (wn * WTN + l31) * LDKH + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDKH + 16 * ks);
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDKH + 16 * ks);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  // two chunks of global loads in flight (a chunk's MFMAs take ~0.2 us, a load far longer), one barrier per chunk;
  // chunks past the end read zeros (every table row >= r_end is masked) and are never multiplied
  build_window(0);
  __syncthreads();
  Stage s0, s1;
  load_chunk(s0, 0);
  store_chunk(s0, 0);
  load_chunk(s0, 1);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    load_chunk(s1, c + 2);
    compute(0);
    store_chunk(s0, 1);
    __syncthreads();
    if (c + 1 >= nchunks) break;   // uniform
    load_chunk(s0, c + 3);
    compute(1);
    store_chunk(s1, 0);
    __syncthreads();
  }

  // tile row rho -> channel 8*(rho % (T/8)) + rho / (T/8)
  float* outp = p.Out + (long long)kz * p.M * p.Cs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int rn = wn * WTN + j * 32 + l31;
    const int col = n0 + 8 * (rn % ON) + rn / ON;
    if (col >= p.Cs) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rm = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + 8 * (rm % OM) + rm / OM;
        if (m < p.M) outp[(long long)m * p.Cs + col] = acc[i][j][r];
      }
    }
  }
}

template <int BM, int BN>
static int wgbc_launch(const WGParams& pin, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  p.remap = veex_tune_int("VEEX_XCD_WG", 1);   // -7 %: the tap tiles of one row split share their rows of `big` in one L2
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  constexpr size_t lds = 2 * (BM + BN) * 72 * 2 + WGBC_TABW * sizeof(int2);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16c_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      veex_set_error("wgrad(bf16): cannot raise the dynamic LDS limit to %zu bytes", lds); return -3;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_bf16c_kernel<BM, BN>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN>
static int wgb_launch(const WGParams& pin, hipStream_t st) {
  WGParams p = pin;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.Cs + BN - 1) / BN; p.gz = p.nsplit;
  p.remap = veex_tune_int("VEEX_XCD_WG", 0);
  dim3 grid((unsigned)p.gx * p.gy * p.gz, 1, 1);
  constexpr size_t lds = 2 * (BM + BN) * 40 * 2;
  const bool affb = p.b_scale != nullptr, affs = p.s_scale != nullptr;
  if (!affb && !affs) hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, false, false>), grid, dim3(256), lds, st, p);
  else if (affb && !affs) hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, true, false>), grid, dim3(256), lds, st, p);
  else if (!affb && affs) hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, false, true>), grid, dim3(256), lds, st, p);
  else { veex_set_error("wgrad: affine on both operands is not supported"); return -1; }
  VEEX_CHECK_LAUNCH();
  return 0;
}

// 1 if the bf16 kernel can run this (validated) descriptor, else the caller uses the fp32 kernel
int veex_wgrad_bf16c_ok(const WGParams& p);
int veex_wgrad_bf16_ok(const WGParams& p) {
  if (veex_wgrad_bf16c_ok(p)) return 1;
  const long long big_bytes = (long long)p.Nb * p.Hb * p.Wb * p.Cb * 4, small_bytes = (long long)p.R * p.Cs * 4;
  return p.bf16 && p.M >= 128 && p.Cs >= 64 && (p.Cb % 4 == 0) && (p.Cs % 4 == 0) && (p.rows_per_split % 32 == 0) &&
         big_bytes < 0x7FFF0000LL && small_bytes < 0x7FFF0000LL && p.R < (1 << 24) &&
         !(p.b_scale != nullptr && p.b_aff_spatial) && !(p.s_scale != nullptr && p.s_aff_spatial);
}

// 1 if both operands come as bf16 consumer copies the copy kernel can take
int veex_wgrad_bf16c_ok(const WGParams& p) {
  const long long big_bytes = (long long)p.Nb * p.Hb * p.Wb * p.Cb * 2, small_bytes = (long long)p.R * p.Cs * 2;
  return p.bf16 && p.BigBf != nullptr && p.SmallBf != nullptr && p.M >= 128 && p.Cs >= 64 && (p.Cb % 8 == 0) &&
         (p.Cs % 8 == 0) && (p.rows_per_split % 64 == 0) && big_bytes < 0x7FFF0000LL && small_bytes < 0x7FFF0000LL &&
         p.R < (1 << 24);
}

int veex_launch_wgrad_bf16(const WGParams& p, hipStream_t st) {
  if (veex_wgrad_bf16c_ok(p)) {
    if (veex_wgrad_bf16t_ok(p)) return veex_launch_wgrad_bf16t(p, st);     // no transpose while staging (wgrad_gemm_bf16t.hip)
    return (p.Cs > 64) ? wgbc_launch<128, 128>(p, st) : wgbc_launch<128, 64>(p, st);
  }
  if (p.Cs > 64) return wgb_launch<128, 128>(p, st);
  return wgb_launch<128, 64>(p, st);
}
