// Implicit-GEMM engine on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: bit-exact k-ordered fmaf chain, 64 FLOP/clk/SIMD).
//
// One kernel template serves every dense contraction of the VAE hot path that is not a
// weight gradient:
//   mode 0 "conv-gather":   Conv2D forward (reference models.py:219-235), Dense forward
//                           (models.py:238-241,264), Conv2DTranspose input-gradient
//   mode 1 "deconv-gather": Conv2DTranspose forward (models.py:270-298), Conv2D / Dense
//                           input-gradient.  Stride 2 is decomposed into 4 sub-pixel phases so
//                           that no MFMA is spent on the zeros of a zero-inserted input.
//
// Data layout: activations NHWC in HBM (channel-contiguous = GEMM-K contiguous), weights in
// the Keras shapes the reference uses ((kh,kw,Cin,Cout) / (kh,kw,Cout,Cin) / (in,out)).
// A tile [BM][16] and B tile are register-staged into double-buffered LDS (one barrier per
// K-chunk); the producer layer's BatchNorm (inference mode) is applied to A while
// staging, so every activation is stored once (post-ReLU) and read once per tap.
// Epilogues: FWD = +bias, ReLU;  BWD = BatchNorm/ReLU backward of the layer whose output
// gradient this GEMM produces (writes d(pre-activation) and per-tile column sums for dgamma,
// dbeta, dbias);  RAW = split-K partial slab.
#include "veex_internal.h"
#include <stdlib.h>
#include <algorithm>

#include "gg_common.h"

template <int BM, int BN, int WM, int WN, bool MODE1>
__global__ __launch_bounds__(256) void gather_gemm_generic_kernel(const GGParams p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int APASS = BM / 64;
  constexpr int A_ELEMS = BM * VEEX_LDK;
  constexpr int LDN = BN + 4;
  constexpr int B_ELEMS = MODE1 ? BN * VEEX_LDK : VEEX_BK * LDN;
  constexpr int BPASS = MODE1 ? BN / 64 : (VEEX_BK * BN / 4) / 256;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BPASS >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  int* s_off = reinterpret_cast<int*>(Bs + 2 * B_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  // ---- block -> (phase, m tile, n tile, k split) ----
  int bnt, btile, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
  int tile = btile, ph = 0;
  if (p.nphase == 4) {
    // phases are concatenated, 4-tap phase first: the heavy tiles start first and the 1-tap tiles
    // fill the tail (interleaving the phases measured 5-18 % slower)
    while (ph < 3 && tile >= p.tile_prefix[ph + 1]) ++ph;
  }
  const int py = ph >> 1, px = ph & 1;
  const int Hp = p.Hp[py], Wp = p.Wp[px];
  const int ostep = (p.nphase == 4) ? 2 : 1;
  const int oy0 = p.o0[py], ox0 = p.o0[px];
  const int Mph = p.Nb * Hp * Wp;
  const int m0 = (tile - p.tile_prefix[ph]) * BM;
  const int n0 = bnt * BN;

  int kys = 0, kxs = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (p.nphase == 4) {
    kys = py; kxs = px; kstep = 2;
    nky = (p.kh - py + 1) >> 1;
    nkx = (p.kw - px + 1) >> 1;
  }
  const int cpt = (p.Cin + VEEX_BK - 1) / VEEX_BK;
  const int ntaps = nky * nkx;
  const int nchunks = ntaps * cpt;
  const int c_begin = kz * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);

  const int HWp = Hp * Wp;
  // output offsets (pixel index) of the tile rows, for the epilogue
  if (tid < BM) {
    int m = m0 + tid, off = -1;
    if (m < Mph) {
      int n = m / HWp, rem = m - n * HWp;
      int j = rem / Wp, i = rem - j * Wp;
      off = (n * p.Hout + ostep * j + oy0) * p.Wout + ostep * i + ox0;
    }
    s_off[tid] = off;
  }

  // ---- loader state: this thread stages rows (tid>>2)+64*i, k-quad (tid&3) of the A tile ----
  const int q = tid & 3, lr = tid >> 2;
  int a_base[APASS], a_y[APASS], a_x[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int m = m0 + lr + 64 * i;
    a_base[i] = -1; a_y[i] = 0; a_x[i] = 0;
    if (m < Mph) {
      int n = m / HWp, rem = m - n * HWp;
      int j = rem / Wp, ii = rem - j * Wp;
      int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      a_base[i] = n * p.Hin * p.Win;
      if (MODE1) { a_y[i] = oy + p.pb; a_x[i] = ox + p.pb; }
      else       { a_y[i] = p.stride * oy - p.pb; a_x[i] = p.stride * ox - p.pb; }
    }
  }
  const bool vecA = ((p.Cin & 3) == 0);
  const bool vecW = MODE1 ? vecA : ((p.Cout & 3) == 0);
  const int sshift = p.stride - 1;   // stride in {1,2}

  float4 ra[APASS], rb[BPASS];

  auto load_chunk = [&](int c) {
    // tap_inner: channel chunk outer, tap inner -- the shifted re-reads of one channel slice follow
    // each other closely (L2 hits instead of Infinity-Cache / HBM re-fetches)
    const int cc = p.tap_inner ? c / ntaps : c % cpt;
    const int tapi = p.tap_inner ? c - cc * ntaps : c / cpt;
    const int ia = tapi / nkx, ib = tapi - ia * nkx;
    const int ky = kys + kstep * ia, kx = kxs + kstep * ib;
    const int tap = ky * p.kw + kx;
    const int ci0 = cc * VEEX_BK;
    const int ci = ci0 + 4 * q;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      bool ok = a_base[i] >= 0;
      int iy, ix;
      if (MODE1) {
        int ty = a_y[i] - ky, tx = a_x[i] - kx;
        ok = ok && ty >= 0 && tx >= 0;
        iy = ty >> sshift; ix = tx >> sshift;
      } else {
        iy = a_y[i] + ky; ix = a_x[i] + kx;
      }
      ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && ci < p.Cin;
      const int pix = iy * p.Win + ix;
      const float* ptr = p.A + ((long long)(a_base[i] + pix)) * p.Cin + ci;
      float4 v = ld4(ptr, ok, p.Cin - ci, vecA);
      if (p.a_scale != nullptr && ok) {
        const int aidx = (ci + (p.aff_spatial ? pix * p.Cin : 0)) & p.aff_mask;
        const float4 sc = ld4(p.a_scale + aidx, true, p.Cin - ci, vecA);
        const float4 sh = ld4(p.a_shift + aidx, true, p.Cin - ci, vecA);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        // lanes beyond Cin keep 0: scale/shift loads there returned 0 too
      }
      ra[i] = v;
    }
    if (MODE1) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + lr + 64 * i;
        const bool ok = n < p.Cout && ci < p.Cin;
        const float* ptr = p.W + ((long long)tap * p.Cout + n) * p.Cin + ci;
        rb[i] = ld4(ptr, ok, p.Cin - ci, vecW);
      }
    } else {
      constexpr int QN = BN / 4;
      constexpr int RPP = 256 / QN;
      const int nq = tid % QN, kr = tid / QN;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int k = kr + RPP * i;
        const int cik = ci0 + k;
        const int n = n0 + 4 * nq;
        const bool ok = cik < p.Cin && n < p.Cout;
        const float* ptr = p.W + ((long long)tap * p.Cin + cik) * p.Cout + n;
        rb[i] = ld4(ptr, ok, p.Cout - n, vecW);
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<float4*>(as + (lr + 64 * i) * VEEX_LDK + 4 * q) = ra[i];
    if (MODE1) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i)
        *reinterpret_cast<float4*>(bs + (lr + 64 * i) * VEEX_LDK + 4 * q) = rb[i];
    } else {
      constexpr int QN = BN / 4;
      constexpr int RPP = 256 / QN;
      const int nq = tid % QN, kr = tid / QN;
#pragma unroll
      for (int i = 0; i < BPASS; ++i)
        *reinterpret_cast<float4*>(bs + (kr + RPP * i) * LDN + 4 * nq) = rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk(0);
  }
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) & 1;
    const bool more = (c + 1 < c_end);
    if (more) load_chunk(c + 1);
    const float* as = As + cur * A_ELEMS + (wm * WTM + l31) * VEEX_LDK + 8 * lh;
    const float* bs = MODE1 ? (Bs + cur * B_ELEMS + (wn * WTN + l31) * VEEX_LDK + 8 * lh)
                            : (Bs + cur * B_ELEMS + (8 * lh) * LDN + wn * WTN + l31);
    // K order inside a chunk is permuted: lanes 0-31 feed k = 0..7, lanes 32-63 k = 8..15
    // (A and B use the same permutation, so the sum is unchanged).
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float av[MT][4], bv[NT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(as + i * 32 * VEEX_LDK + 4 * half);
        av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (MODE1) {
          const float4 t = *reinterpret_cast<const float4*>(bs + j * 32 * VEEX_LDK + 4 * half);
          bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) bv[j][kk] = bs[(4 * half + kk) * LDN + j * 32];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][kk], bv[j][kk], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(cur ^ 1);
    __syncthreads();
  }

  gg_epilogue<BM, BN, WM, WN>(p, acc, s_off, As, btile, kz, n0);
}


// ---------------------------------------------------------------------------
// Fast path (all vector dims % 4 == 0, every operand < 4 GiB).  Differences to the generic kernel:
//  * every global access is a 16-byte raw buffer load whose offset is forced out of range for
//    padding pixels / K tails / rows past M, so the hardware returns zeros: no branch around any
//    load, the K loop is ONE basic block and hipcc interleaves loads, LDS traffic and the 64-cycle
//    f32 MFMAs freely;
//  * per-row addressing is (row base) + (wave-uniform tap delta) + (chunk offset); validity of a
//    (row, tap) pair is one bit of a per-row mask computed once before the loop.
// AFF: 0 none, 1 per-channel producer BatchNorm, 2 BatchNorm over the flattened (h,w,c) vector.
// ---------------------------------------------------------------------------
// V2: the chunk -> (tap, channel slice) decomposition is carried incrementally in scalar registers (no
// integer division in the K loop: -2 % over the GEMMs of a step).  Tried on top of it and dropped (batch 512,
// tools/ab_env.py): the per-channel BatchNorm staged in LDS (+1 %), s_setprio around the MFMAs (+2 %), channel-
// slice pairs innermost for L1 line re-use (-0.5 %).
template <int BM, int BN, int WM, int WN, bool MODE1, int AFF, int BK, int DBG = 0, int V2 = 0>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GGParams p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int LDK = BK + 4;          // padded k-stride of the [row][k] LDS tiles
  constexpr int QK = BK / 4;           // 16-byte quads per tile row
  constexpr int RP = 256 / QK;         // tile rows staged per pass
  constexpr int APASS = BM / RP;
  constexpr int A_ELEMS = BM * LDK;
  constexpr int LDN = BN + 4;
  constexpr int B_ELEMS = MODE1 ? BN * LDK : BK * LDN;
  constexpr int BPASS = MODE1 ? BN / RP : (BK * BN / 4) / 256;
  constexpr int QN = BN / 4;
  constexpr int RPP = 256 / QN;
  static_assert(APASS >= 1 && BPASS >= 1, "tile too small for this BK");
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  int* s_off = reinterpret_cast<int*>(Bs + 2 * B_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bnt, btile, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
  int tile = btile, ph = 0;
  if (p.nphase == 4) {
    // phases are concatenated, 4-tap phase first: the heavy tiles start first and the 1-tap tiles
    // fill the tail (interleaving the phases measured 5-18 % slower)
    while (ph < 3 && tile >= p.tile_prefix[ph + 1]) ++ph;
  }
  const int py = ph >> 1, px = ph & 1;
  const int Hp = p.Hp[py], Wp = p.Wp[px];
  const int ostep = (p.nphase == 4) ? 2 : 1;
  const int oy0 = p.o0[py], ox0 = p.o0[px];
  const int Mph = p.Nb * Hp * Wp;
  const int m0 = (tile - p.tile_prefix[ph]) * BM;
  const int n0 = bnt * BN;

  int kys = 0, kxs = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (p.nphase == 4) {
    kys = py; kxs = px; kstep = 2;
    nky = (p.kh - py + 1) >> 1;
    nkx = (p.kw - px + 1) >> 1;
  }
  const int cpt = (p.Cin + BK - 1) / BK;
  const int ntaps = nky * nkx;
  const int nchunks = ntaps * cpt;
  const int c_begin = kz * p.chunks_per_split;
  const int c_end = (DBG & 32) ? c_begin : min(nchunks, c_begin + p.chunks_per_split);
  const int HWp = Hp * Wp;
  if (tid < BM) {
    int m = m0 + tid, off = -1;
    if (m < Mph) {
      int n = m / HWp, rem = m - n * HWp;
      int j = rem / Wp, i = rem - j * Wp;
      off = (n * p.Hout + ostep * j + oy0) * p.Wout + ostep * i + ox0;
    }
    s_off[tid] = (off < 0) ? (int)0x80000000 : off * p.Cout * 4;   // byte offset of the fp32 output row (gg_epilogue_b)
  }

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, (int)((unsigned)p.Nb * p.Hin * p.Win * p.Cin * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.W), 0, (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 4u), 0x00020000);
  const unsigned aff_bytes = (AFF != 0) ? (unsigned)(p.aff_mask + 1) * 4u : 0u;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_scale : p.A), 0, (int)aff_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_shift : p.A), 0, (int)aff_bytes, 0x00020000);

  // ---- per-row state of the A loader: rows (tid/QK)+RP*i, k-quad (tid%QK) ----
  const int q = tid % QK, lr = tid / QK;
  int a_rowbase[APASS];        // byte offset of (row pixel at phase tap (0,0), channel 4q); may be "virtual"
  int a_nbase[APASS];          // byte offset of the sample (for AFF==2)
  unsigned a_mask[APASS];      // bit t set <=> tap index t of this phase reads an in-bounds pixel
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + lr + RP * i;
    a_rowbase[i] = 0; a_mask[i] = 0u; a_nbase[i] = 0;
    if (m < Mph) {
      const int n = m / HWp, rem = m - n * HWp;
      const int j = rem / Wp, ii = rem - j * Wp;
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      int by, bx;   // input pixel read by phase tap (0,0)
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      a_nbase[i] = n * p.Hin * p.Win * p.Cin * 4;
      a_rowbase[i] = a_nbase[i] + ((by * p.Win + bx) * p.Cin + 4 * q) * 4;
      unsigned mk = 0u;
      for (int ia = 0; ia < nky; ++ia)
        for (int ib = 0; ib < nkx; ++ib) {
          const int iy = MODE1 ? by - ia : by + ia, ix = MODE1 ? bx - ib : bx + ib;
          bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
          if (MODE1) ok = ok && (oy + p.pb - (kys + kstep * ia)) >= 0 && (ox + p.pb - (kxs + kstep * ib)) >= 0;
          if (ok) mk |= 1u << (ia * nkx + ib);
        }
      a_mask[i] = mk;
    }
  }
  // B loader state
  const int nq = tid % QN, kr = tid / QN;
  int b_rowbase[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    if (MODE1) {   // [tap][Cout][Cin]: row n, quad q along Cin
      const int n = n0 + lr + RP * i;
      b_ok[i] = n < p.Cout;
      b_rowbase[i] = (n * p.Cin + 4 * q) * 4;
    } else {       // [tap][Cin][Cout]: row k, quad nq along Cout
      const int n = n0 + 4 * nq;
      b_ok[i] = n < p.Cout;
      b_rowbase[i] = ((kr + RPP * i) * p.Cout + n) * 4;
    }
  }
  const int tap_bytes = p.Cin * p.Cout * 4;

  struct Stage {   // one K-chunk in flight from global memory (registers)
    float4 ra[APASS], rb[BPASS], rsc[AFF == 2 ? APASS : 1], rsh[AFF == 2 ? APASS : 1];
    bool rok[APASS];
  };

  // V2: running (tap, channel slice) of the next chunk to load; chunks are loaded strictly in order
  int it_c = c_begin, it_cc = 0, it_tapi = 0, it_ia = 0, it_ib = 0;
  if (V2) {
    it_cc = c_begin / ntaps; it_tapi = c_begin - it_cc * ntaps;
    it_ia = it_tapi / nkx; it_ib = it_tapi - it_ia * nkx;
  }

  auto load_chunk = [&](Stage& st, int c, bool cvalid) {
    int cc, tapi, ia, ib;
    if (V2) {
      cc = it_cc; tapi = it_tapi; ia = it_ia; ib = it_ib; cvalid = it_c < c_end;
      ++it_c; ++it_tapi; ++it_ib;
      if (it_ib == nkx) { it_ib = 0; ++it_ia; }
      if (it_tapi == ntaps) { it_tapi = 0; it_ia = 0; it_ib = 0; ++it_cc; }
    } else {
      cc = p.tap_inner ? c / ntaps : c % cpt;   // see the generic kernel
      tapi = p.tap_inner ? c - cc * ntaps : c / cpt;
      ia = tapi / nkx; ib = tapi - ia * nkx;
    }
    const int tap = (kys + kstep * ia) * p.kw + (kxs + kstep * ib);
    const int ci0 = cc * BK;
    const int dpix = ia * p.Win + ib;
    const int delta = (MODE1 ? -dpix : dpix) * p.Cin * 4 + ci0 * 4;
    const bool kq = cvalid && (ci0 + 4 * q < p.Cin);
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = kq && ((a_mask[i] >> tapi) & 1u);
      unsigned off = ok ? (unsigned)(a_rowbase[i] + delta) : VEEX_OOB;
      if (DBG & 64) off = ok ? (off & 0xFF0u) : VEEX_OOB;   // timing-only: every A load hits the same 4 KiB
      st.ra[i] = bload4(rA, off);
      st.rok[i] = ok;
      if (AFF == 2) {
        const unsigned ao = ok ? (unsigned)(a_rowbase[i] + delta - a_nbase[i]) : VEEX_OOB;
        st.rsc[i] = bload4(rS, ao);
        st.rsh[i] = bload4(rH, ao);
      }
    }
    if (AFF == 1) {
      const unsigned ao = kq ? (unsigned)(((ci0 + 4 * q) & p.aff_mask) * 4) : VEEX_OOB;
      st.rsc[0] = bload4(rS, ao);
      st.rsh[0] = bload4(rH, ao);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      bool ok;
      unsigned off;
      if (MODE1) {
        ok = kq && b_ok[i];
        off = (unsigned)(tap * tap_bytes + b_rowbase[i] + ci0 * 4);
      } else {
        ok = cvalid && b_ok[i] && (ci0 + kr + RPP * i < p.Cin);
        off = (unsigned)(tap * tap_bytes + ci0 * p.Cout * 4 + b_rowbase[i]);
      }
      st.rb[i] = bload4(rW, ok ? off : VEEX_OOB);
    }
  };

  auto store_chunk = [&](const Stage& st, int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float4 v = st.ra[i];
      if (AFF != 0) {
        const float4 sc = st.rsc[AFF == 2 ? i : 0], sh = st.rsh[AFF == 2 ? i : 0];
        // padding pixels must stay 0 (the shift only applies to real pixels)
        v.x = st.rok[i] ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = st.rok[i] ? fmaf(v.y, sc.y, sh.y) : 0.f;
        v.z = st.rok[i] ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = st.rok[i] ? fmaf(v.w, sc.w, sh.w) : 0.f;
      }
      *reinterpret_cast<float4*>(as + (lr + RP * i) * LDK + 4 * q) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      if (MODE1) *reinterpret_cast<float4*>(bs + (lr + RP * i) * LDK + 4 * q) = st.rb[i];
      else       *reinterpret_cast<float4*>(bs + (kr + RPP * i) * LDN + 4 * nq) = st.rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K order inside a chunk is permuted: lanes 0-31 feed k = 0..BK/2-1, lanes 32-63 the upper half
  // (A and B use the same permutation, so the sum is unchanged).
  auto compute = [&](int cur) {
    const float* as = As + cur * A_ELEMS + (wm * WTM + l31) * LDK + (BK / 2) * lh;
    const float* bs = MODE1 ? (Bs + cur * B_ELEMS + (wn * WTN + l31) * LDK + (BK / 2) * lh)
                            : (Bs + cur * B_ELEMS + ((BK / 2) * lh) * LDN + wn * WTN + l31);
#pragma unroll
    for (int half = 0; half < BK / 8; ++half) {
      float av[MT][4], bv[NT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(as + i * 32 * LDK + 4 * half);
        av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (MODE1) {
          const float4 t = *reinterpret_cast<const float4*>(bs + j * 32 * LDK + 4 * half);
          bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) bv[j][kk] = bs[(4 * half + kk) * LDN + j * 32];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (DBG & 2) acc[i][j][kk] += av[i][kk] * bv[j][kk];   // timing-only build: no MFMA
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][kk], bv[j][kk], acc[i][j], 0, 0, 0);
          }
    }
  };

  // Software pipeline, two chunks of global loads in flight: while chunk c is multiplied out of LDS,
  // chunk c+1 sits in registers (issued one iteration earlier) and chunk c+2 is being issued.  Loads
  // past the end have every offset out of range (zeros, never read).  One barrier per chunk.
  Stage s0, s1;
  load_chunk(s0, c_begin, c_begin < c_end);
  store_chunk(s0, 0);
  load_chunk(s0, c_begin + 1, c_begin + 1 < c_end);
  __syncthreads();
  for (int c = c_begin; c < c_end; c += 2) {
    if (!(DBG & 1)) load_chunk(s1, c + 2, c + 2 < c_end);
    compute(0);
    if (!(DBG & 8)) store_chunk(s0, 1);
    if (!(DBG & 4)) __syncthreads();
    // (an odd chunk count multiplies one extra all-zero chunk: cheaper than a second loop exit,
    //  which makes hipcc keep two copies of the accumulators)
    if (!(DBG & 1)) load_chunk(s0, c + 3, c + 3 < c_end);
    compute(1);
    if (!(DBG & 8)) store_chunk(s1, 0);
    if (!(DBG & 4)) __syncthreads();
  }
  if (DBG & 16) {   // timing-only: one store per thread keeps the accumulators alive
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][jj][r];
    p.Out[(long long)btile * 256 + tid] = t;
    return;
  }
  gg_epilogue_b<BM, BN, WM, WN>(p, acc, reinterpret_cast<const unsigned*>(s_off), As, btile, kz, n0);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Rows per tile.  Measured at batch 512 (tools/ab_env.py VEEX_BM=64,128): the fp32 engine is faster with
// 64x64 tiles on every layer (-3..-45 %: 8 resident blocks per CU overlap each other's prologues,
// barriers and epilogues; the f32 MFMA is slow enough that operand re-use is not the limit); the bf16
// engine keeps 128-row tiles for forward launches (operand re-use matters at 16x the MFMA rate) and
// takes 64x64 for the heavier BatchNorm/ReLU-backward epilogue and for launches of < 256 blocks.
static bool gg_fast_ok_fwd(const GGParams& p);
static bool gg_x3_dispatch(const GGParams& p, bool fast);
int veex_gg_bm(const GGParams& p) {
  const long long M = (long long)p.Nb * p.Hout * p.Wout;
  if (gg_x3_dispatch(p, gg_fast_ok_fwd(p))) return veex_gg_x3_bm(p);
  if (veex_gg_v4_ok(p)) return 32;   // small launches: 32x32 tiles, K split over the waves of a block (gather_gemm_v4.hip)
  if (M < 4096) return 64;
  const int force = veex_tune_int("VEEX_BM", 0);
  if (force == 64 || force == 128) return force;
  const bool nobf = p.Wbf == nullptr || p.wplanes == 3;   // (a split-fp32 launch the split kernel does not take runs on the fp32 kernels)
  if (nobf && veex_env_int("VEEX_GGV3", 1) && gg_fast_ok_fwd(p) && veex_gg_v3_ok(p)) return veex_gg_v3_bm(p);   // lean fp32 kernel
  if (nobf || p.Cin % 32 != 0) return 64;
  const long long blocks = ((M + 127) / 128) * ((p.Cout > 64) ? (p.Cout + 127) / 128 : 1);
  // BatchNorm/ReLU-backward epilogue: 64-row tiles, except the stride-1 multi-tap input gradients (deconv2, deconv4:
  // long K loops), where 128-row tiles measure -10..-27 % since the epilogue went branch-free
  const bool bwd64 = p.epi == VEEX_EPI_BWD && !(p.stride == 1 && p.kh * p.kw > 1);
  return (bwd64 || blocks < 256) ? 64 : 128;
}

static int gg_bn(int Cout, int bm) {
  if (bm == 64) return 64;
  return (Cout > 64) ? 128 : 64;
}

static bool gg_fast_ok(const GGParams& p);
static bool gg_bf16_dispatch(const GGParams& p, bool fast);
// the launch goes to the halo-tile bf16 kernel (halo_gemm_bf16.hip)
static bool gg_halo_dispatch(const GGParams& p) { return gg_bf16_dispatch(p, gg_fast_ok(p)) && veex_hg_ok(p); }
int veex_gg_m_tiles(GGParams& p) {
  const int t = veex_gg_m_tiles_bm(p, veex_gg_bm(p));
  return gg_halo_dispatch(p) ? veex_hg_m_tiles(p) : t;   // (the phase tables are filled either way)
}

int veex_gg_m_tiles_bm(GGParams& p, int bm) {
  const long long Mtot = (long long)p.Nb * p.Hout * p.Wout;
  if (p.mode == 1 && p.stride == 2) {
    p.nphase = 4;
    for (int par = 0; par < 2; ++par) {
      const int o0 = (par + p.pb) & 1;   // smallest o >= 0 with (o + pb) % 2 == par
      p.o0[par] = o0;
      p.Hp[par] = (p.Hout > o0) ? (p.Hout - o0 + 1) / 2 : 0;
      p.Wp[par] = (p.Wout > o0) ? (p.Wout - o0 + 1) / 2 : 0;
    }
    p.tile_prefix[0] = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const long long mph = (long long)p.Nb * p.Hp[ph >> 1] * p.Wp[ph & 1];
      p.tile_prefix[ph + 1] = p.tile_prefix[ph] + (int)((mph + bm - 1) / bm);
    }
    return p.tile_prefix[4];
  }
  p.nphase = 1;
  p.o0[0] = p.o0[1] = 0;
  p.Hp[0] = p.Hp[1] = p.Hout;
  p.Wp[0] = p.Wp[1] = p.Wout;
  p.tile_prefix[0] = 0;
  const int t = (int)((Mtot + bm - 1) / bm);
  for (int ph = 1; ph <= 4; ++ph) p.tile_prefix[ph] = t;
  return t;
}

template <int BM, int BN, bool MODE1, int BK>
static constexpr size_t gg_lds_bytes() {
  return (2 * BM * (BK + 4) + 2 * (MODE1 ? BN * (BK + 4) : BK * (BN + 4))) * sizeof(float) + BM * sizeof(int);
}

template <typename K>
static int gg_set_lds(K kernel, size_t lds) {
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      veex_set_error("gather_gemm: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return -3;
    }
  }
  return 0;
}

template <int BM, int BN, bool MODE1, int BK>
static int gg_launch_fast(const GGParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = gg_lds_bytes<BM, BN, MODE1, BK>();
  const int aff = (p.a_scale == nullptr) ? 0 : (p.aff_spatial ? 2 : 1);
  static bool attr_done = false;   // function attributes are per process; set them once, outside any capture
  if (!attr_done) {
    if (gg_set_lds(gather_gemm_kernel<BM, BN, 2, 2, MODE1, 0, BK>, lds) || gg_set_lds(gather_gemm_kernel<BM, BN, 2, 2, MODE1, 1, BK>, lds) ||
        gg_set_lds(gather_gemm_kernel<BM, BN, 2, 2, MODE1, 2, BK>, lds)) return -3;
    attr_done = true;
  }
  if constexpr (BM == 64 && BN == 64 && BK == 16) {   // V2 of the K loop (see the kernel)
    if (aff == 0) hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 0, BK, 0, 1>), grid, dim3(256), lds, st, p);
    else if (aff == 1) hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 1, BK, 0, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 2, BK, 0, 1>), grid, dim3(256), lds, st, p);
    VEEX_CHECK_LAUNCH();
    return 0;
  }
  if (aff == 0) hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 0, BK>), grid, dim3(256), lds, st, p);
  else if (aff == 1) hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 1, BK>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, 2, 2, MODE1, 2, BK>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN>
static int gg_launch_cfg(GGParams& p, int mtiles, bool fast, hipStream_t st) {
  const int ntiles = (p.Cout + BN - 1) / BN;
  // XCD-contiguous block order: neutral on single-phase layers, but on the 4-phase (stride-2 deconv)
  // launches whole XCDs would get only 4-tap or only 1-tap tiles (+30..50 % time): round-robin there
  const int xr = veex_tune_int("VEEX_XCD", -1);
  p.remap = (xr < 0) ? (p.nphase == 1) : xr;
  p.tap_inner = 1;   // taps innermost: measured -1..-4 % on every layer
  p.gx = ntiles; p.gy = mtiles; p.gz = p.nsplit;
  dim3 grid((unsigned)ntiles * mtiles * p.nsplit, 1, 1);
  if (fast) {
    const int want32 = veex_tune_int("VEEX_BK", 16) == 32;
    const bool bk32 = want32 && ((p.kh * p.kw == 1) || (p.Cin % 32 == 0));
    const int bk = bk32 ? 32 : 16;
    if (p.nsplit > 1) {
      const int nchunks = p.kh * p.kw * ((p.Cin + bk - 1) / bk);
      p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
    }
    if (bk32) return (p.mode == 1) ? gg_launch_fast<BM, BN, true, 32>(p, grid, st) : gg_launch_fast<BM, BN, false, 32>(p, grid, st);
    return (p.mode == 1) ? gg_launch_fast<BM, BN, true, 16>(p, grid, st) : gg_launch_fast<BM, BN, false, 16>(p, grid, st);
  }
  if (p.nsplit > 1) {
    const int nchunks = p.kh * p.kw * ((p.Cin + VEEX_BK - 1) / VEEX_BK);
    p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
  }
  constexpr size_t lds1 = gg_lds_bytes<BM, BN, true, VEEX_BK>();
  constexpr size_t lds0 = gg_lds_bytes<BM, BN, false, VEEX_BK>();
  if (p.mode == 1)
    hipLaunchKernelGGL((gather_gemm_generic_kernel<BM, BN, 2, 2, true>), grid, dim3(256), lds1, st, p);
  else
    hipLaunchKernelGGL((gather_gemm_generic_kernel<BM, BN, 2, 2, false>), grid, dim3(256), lds0, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

static bool gg_fast_ok(const GGParams& p);
static bool gg_fast_ok_fwd(const GGParams& p) { return gg_fast_ok(p); }
// fast path: 16-byte buffer loads everywhere, 32-bit byte offsets, tap-validity bit mask
static bool gg_fast_ok(const GGParams& p) {
  const long long Mtot = (long long)p.Nb * p.Hout * p.Wout;
  const long long a_bytes = (long long)p.Nb * p.Hin * p.Win * p.Cin * 4;
  const long long w_bytes = (long long)p.kh * p.kw * p.Cin * p.Cout * 4;
  const bool vec_ok = (p.Cin % 4 == 0) && (p.mode == 1 || p.Cout % 4 == 0);
  return vec_ok && a_bytes < 0x7FFF0000LL && w_bytes < 0x7FFF0000LL && Mtot * p.Cout * 4 < 0x7FFF0000LL && p.kh * p.kw <= 32 &&
         (p.a_scale == nullptr || !p.aff_spatial || (p.aff_mask + 1) == p.Hin * p.Win * p.Cin);
}
// the launch goes to the bf16 engine (gather_gemm_bf16.hip)
static bool gg_x3_dispatch(const GGParams& p, bool fast) {
  const long long Mtot = (long long)p.Nb * p.Hout * p.Wout;
  return p.Wbf != nullptr && p.wplanes == 3 && fast && Mtot >= 4096 && Mtot < (1LL << 24) && Mtot * p.Cout * 4 < 0x7FFF0000LL && p.Cin % 32 == 0 &&
         p.Abf == nullptr && p.OutBf == nullptr && veex_gg_x3_cfg(p) >= 0;
}
static bool gg_bf16_dispatch(const GGParams& p, bool fast) {
  const long long Mtot = (long long)p.Nb * p.Hout * p.Wout;
  return p.Wbf != nullptr && p.wplanes != 3 && fast && Mtot >= 4096 && Mtot < (1LL << 24) && Mtot * p.Cout * 4 < 0x7FFF0000LL && p.Cin % 32 == 0;
}
// 1 if this launch would run on the bf16 engine reading only the bf16 copy of its A operand (then nobody reads the
// fp32 A), and 2 if in addition it can write its bf16 output copy alone (Out may be null)
int veex_gg_bf16_only(const GGParams& p) {
  if (!gg_bf16_dispatch(p, gg_fast_ok(p)) || p.Abf == nullptr) return 0;
  return (p.OutBf != nullptr && p.epi != VEEX_EPI_RAW) ? 2 : 1;
}

int veex_launch_gather_gemm(const GGParams& pin, hipStream_t st) {
  GGParams p = pin;
#ifdef VEEX_DEBUG_TIMING   // timing experiments only (make TIMING=1; tools/ab_env.py): results are wrong by construction,
  if (const int tm = veex_env_int("VEEX_TIMING", 0)) {   // so the shipped library does not contain the switch at all
    if (tm & 1) p.a_scale = p.a_shift = nullptr;          //   1: no BatchNorm prologue
    if (tm & 2) { p.epi = VEEX_EPI_RAW; p.partials = nullptr; p.aux = nullptr; }   // 2: plain store epilogue
  }
#endif
  if (p.stride != 1 && p.stride != 2) { veex_set_error("gather_gemm: stride %d unsupported", p.stride); return -1; }
  if (p.Nb <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.Hout <= 0 || p.Wout <= 0 || p.Hin <= 0 || p.Win <= 0) {
    veex_set_error("gather_gemm: bad geometry"); return -1;
  }
  if (p.kh * p.kw > 1 && (p.Cin % VEEX_BK) != 0) {
    veex_set_error("gather_gemm: multi-tap layers need Cin %% 16 == 0 (got %d)", p.Cin); return -1;
  }  // (the K chunk is 32 when Cin %% 32 == 0 or the layer has a single tap, else 16)
  if (p.a_scale && ((p.aff_mask + 1) & p.aff_mask)) { veex_set_error("gather_gemm: affine size must be a power of two"); return -1; }
  const long long Mtot = (long long)p.Nb * p.Hout * p.Wout;
  if (Mtot * p.Cout >= (1LL << 40) || Mtot >= (1LL << 31)) { veex_set_error("gather_gemm: problem too large"); return -1; }
  const int mtiles = veex_gg_m_tiles(p);
  if (p.nsplit < 1) p.nsplit = 1;
  if (p.nsplit > 1 && (p.nphase != 1 || p.epi != VEEX_EPI_RAW)) {
    veex_set_error("gather_gemm: split-K needs RAW epilogue and a single phase"); return -1;
  }
  if (p.nsplit == 1) p.chunks_per_split = 1 << 30;
  const int bm = veex_gg_bm(p);
  const int bn = gg_bn(p.Cout, bm);
  const bool fast = gg_fast_ok(p);
  if (gg_x3_dispatch(p, fast)) return veex_launch_gather_gemm_x3(p, mtiles, st);
  if (gg_halo_dispatch(p)) return veex_launch_halo_gemm_bf16(p, st);
  if (gg_bf16_dispatch(p, fast)) return veex_launch_gather_gemm_bf16(p, mtiles, st);
  if (p.Out == nullptr) { veex_set_error("gather_gemm: null output (only the bf16 engine can write the bf16 copy alone)"); return -1; }
  if (bm == 32) return veex_launch_gather_gemm_v4(p, mtiles, st);
  if (fast && veex_env_int("VEEX_GGV3", 1) && veex_gg_v3_ok(p) && bm == veex_gg_v3_bm(p)) return veex_launch_gather_gemm_v3(p, mtiles, st);
  if (bm == 128 && bn == 128 && fast) return gg_launch_cfg<128, 128>(p, mtiles, true, st);
  if (bm == 128 && bn == 64 && fast) return gg_launch_cfg<128, 64>(p, mtiles, true, st);
  if (bm == 64 && fast) return gg_launch_cfg<64, 64>(p, mtiles, true, st);
  // generic (scalar-capable) kernel is instantiated for the 64x64 tile only
  if (bm != 64) {
    // re-plan the phase tables for 64-row tiles is not possible here (callers sized partials by
    // veex_gg_m_tiles), so large-M problems must satisfy the fast-path constraints
    veex_set_error("gather_gemm: M >= 4096 requires channel counts %% 4 == 0 and operands < 2 GiB");
    return -1;
  }
  return gg_launch_cfg<64, 64>(p, mtiles, false, st);
}

// Timing-only variants for profiling (outputs are wrong by construction): dbg 1 = no global loads in the
// K loop, dbg 2 = no MFMA.  Only the 128x64 mode-1 per-channel-affine and 128x128 mode-0 no-affine shapes.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// returns nothing useful in out; flops = grid*4 waves*iters*32 MFMAs*4096
extern "C" int veexi_debug_mfma_peak(float* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
  VEEX_CHECK_LAUNCH();
  return 0;
}

#define DBG_CASE(BM_, BN_, M1_, AFF_, D_) \
  case D_: hipLaunchKernelGGL((gather_gemm_kernel<BM_, BN_, 2, 2, M1_, AFF_, 16, D_, 1>), grid, dim3(256), lds, st, p); break;

// Timing-only variants for profiling (outputs are wrong by construction). dbg bits: 1 = no global loads in
// the K loop, 2 = no MFMA, 4 = no barrier, 8 = no LDS store, 16 = no epilogue.  64x64 tiles: mode 1 with a
// per-channel affine (a Conv2DTranspose forward) or mode 0 without (its input gradient).
extern "C" int veexi_debug_gather_gemm(const GGParams* pin, int dbg, hipStream_t st) {
  GGParams p = *pin;
  const int mtiles = veex_gg_m_tiles_bm(p, 64);
  p.nsplit = 1; p.chunks_per_split = 1 << 30;
  p.gx = (p.Cout + 63) / 64; p.gy = mtiles; p.gz = 1;
  p.remap = (p.nphase == 1); p.tap_inner = 1;
  dim3 grid(mtiles * p.gx, 1, 1);
  constexpr size_t lds = gg_lds_bytes<64, 64, true, 16>();
  if (p.mode == 1) {
    switch (dbg) {
      DBG_CASE(64, 64, true, 1, 0) DBG_CASE(64, 64, true, 1, 1) DBG_CASE(64, 64, true, 1, 16) DBG_CASE(64, 64, true, 1, 64)
      DBG_CASE(64, 64, true, 1, 13) DBG_CASE(64, 64, true, 1, 29) DBG_CASE(64, 64, true, 1, 17)
      default: return -1;
    }
  } else {
    switch (dbg) {
      DBG_CASE(64, 64, false, 0, 0) DBG_CASE(64, 64, false, 0, 1) DBG_CASE(64, 64, false, 0, 16) DBG_CASE(64, 64, false, 0, 64)
      DBG_CASE(64, 64, false, 0, 13) DBG_CASE(64, 64, false, 0, 29) DBG_CASE(64, 64, false, 0, 17)
      default: return -1;
    }
  }
  VEEX_CHECK_LAUNCH();
  return 0;
}
