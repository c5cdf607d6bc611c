// Lean bf16 instance of the implicit-GEMM engine (gather_gemm_bf16.hip has the general one).
//
// Why: rocprofv3 --pmc over the 21 bf16 launches of a train step (profiles/r02_pmc_gemms_bf16.txt) showed 5-40 VALU and
// 3-39 SALU instructions per v_mfma_f32_32x32x16_bf16 in the general kernel -- per-row address arithmetic, per-chunk
// tap bookkeeping with scalar divisions, per-element epilogue stores -- against a budget of ~5 vector instructions an
// MFMA of 32 cycles can hide (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): MFMA utilisation 0.08-0.19.
// This kernel is the fp32 lean kernel's address scheme (gather_gemm_v3.hip) on bf16 consumer copies:
//   * A (bf16, the producer's BatchNorm already applied) is staged with plain 16-byte copies whose address is a
//     per-row constant VGPR | validity mask + a wave-uniform SGPR displacement per chunk (tap and channel slice);
//     padding pixels, rows past M and chunks past K fall out of the buffer's range and arrive as zeros;
//   * chunk -> (tap, channel slice) is carried incrementally in SGPRs over the tile's set of valid taps (taps that are
//     padding for every row of a tile are skipped: row_order 1, as in the fp32 kernel);
//   * B = packed bf16 weights [tap][Cout][Cin], constant VGPR offset + SGPR offset;
//   * NST chunks of global loads are in flight per wave (register stages), not two: with the address work gone the
//     launches turned out to be bound by load latency -- a 128x128 tile of deconv2 forward took 1 200 cycles per
//     chunk of 256 MFMA cycles with one block per CU -- so the depth of the prefetch is what buys time;
//   * the epilogues are the shared gg_epilogue_b (channels across lanes: a store instruction writes whole 128-byte
//     lines).  Tried and dropped: the MFMA with swapped operands, so that a lane holds four consecutive channels of
//     one pixel and stores 16 bytes at once -- 4x fewer store instructions, but 64 lanes x 16 bytes in 64 different
//     lines per instruction: forward launches took 1.3-3x longer (deconv5 forward 70 -> 208 us).
// Per K chunk of 32 and wave (64x64 register tile): 8 MFMAs beside 8 ds_read_b128, 4 ds_write_b128, 4 buffer loads
// and ~6 VALU.  Same tiles, phase decomposition, results and constraints as the general kernel with ABF (A from a bf16
// copy, no affine), nsplit == 1.
// Reference ops: Conv2D / Conv2DTranspose forward (models.py:219-298) and their input gradients (models.py:166).
#include "gg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4u bload4us(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

__device__ __forceinline__ void b3_divmod(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

// NST: register stages = chunks of global loads in flight
template <int BM, int BN, bool MODE1, int BK, int NST>
__global__ __launch_bounds__(256) void gather_gemm_bf16_v3_kernel(const GGParams p) {
  constexpr int LDKH = BK + 8;                 // bf16 per LDS row: 80 B (BK 32) / 144 B (BK 64): conflict-free b128 reads
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int PQ = BK / 8;                   // 16-byte pieces per row of a chunk
  constexpr int RPP = 256 / PQ;                // rows staged per pass
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;
  static_assert(APASS >= 1 && BPASS >= 1 && MT >= 1 && NT >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Bs = As + 2 * A_ELEMS;
  unsigned* s_offb = reinterpret_cast<unsigned*>(Bs + 2 * B_ELEMS);   // [BM] byte offsets of the fp32 output rows
  unsigned* s_tmk = s_offb + BM;                                      // valid-tap set of the tile

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lh = lane >> 5, l31 = lane & 31;

  int bnt, btile, kz;
  if (p.remap == 2) {
    // one image row per group of tiles (row_order 1): tile t of EVERY image row goes to XCD t % 8 (workgroups are dealt
    // round-robin over the XCDs), image rows in order: the input rows a tile's taps share with the rows above and
    // below meet in ONE L2.  The grid is padded to a multiple of 8 tiles per image row; surplus blocks leave at once.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int rest = local / p.gx, gq = (p.row_tiles + 7) >> 3;
    bnt = local - rest * p.gx;
    const int y = rest / gq, g8 = rest - y * gq;
    const int t = g8 * 8 + xcd;
    if (t >= p.row_tiles) return;          // uniform for the block, before any barrier
    btile = y * p.row_tiles + t;
    kz = 0;
  } else {
    veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
  }
  bnt = __builtin_amdgcn_readfirstlane(bnt);       // wave-uniform: what derives from them stays in SGPRs
  btile = __builtin_amdgcn_readfirstlane(btile);
  int tile = btile, ph = 0;
  if (p.nphase == 4) {
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
  const int cpt = p.Cin / BK;
  const int ntaps = nky * nkx;
  const int D1 = p.row_order ? p.Nb * Wp : Hp * Wp;
  const float invD1 = 1.0f / (float)D1, invWp = 1.0f / (float)Wp;   // rows < 2^24 (checked by the launcher)

  if (tid == 0) *s_tmk = 0u;

  // the per-row offset is taken at the tap with the lowest address; at the top-left border that pixel lies before the
  // tensor, so every per-row offset is shifted up by `bias` bytes and the resource base down by as much (never
  // dereferenced there: such a (row, tap) pair is masked)
  const int bias_pix = MODE1 ? ((nky - 1) * p.Win + (nkx - 1)) : p.pb * (p.Win + 1);
  const int bias_bytes = bias_pix * p.Cin * 2;
  const unsigned a_total = (unsigned)p.Nb * p.Hin * p.Win * p.Cin * 2u;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.Abf)) - bias_bytes, 0, (int)(a_total + (unsigned)bias_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.Wbf), 0, (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 2u), 0x00020000);

  // ---- A loader: rows (tid / PQ) + RPP * i, 16-byte piece tid % PQ ----
  const int q = tid % PQ, lr = tid / PQ;
  unsigned a_voff[APASS], a_inv[APASS];     // a_inv bit t set <=> tap t must read zeros
  unsigned my_taps = 0u;
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + lr + RPP * i;
    unsigned out_off = 0x80000000u;         // + any column offset stays out of range: the store is dropped
    a_voff[i] = 0u; a_inv[i] = 0xFFFFFFFFu;
    if (m < Mph) {
      int a, rem, b, ii;
      b3_divmod(m, D1, invD1, a, rem);
      b3_divmod(rem, Wp, invWp, b, ii);
      const int n = p.row_order ? b : a, j = p.row_order ? a : b;
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      out_off = (unsigned)(((n * p.Hout + oy) * p.Wout + ox) * p.Cout) * 4u;
      int by, bx;   // input pixel read by phase tap (0,0)
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      const int nbase = n * p.Hin * p.Win * p.Cin * 2;
      const int inner = ((MODE1 ? (by * p.Win + bx) : ((by + p.pb) * p.Win + bx + p.pb)) * p.Cin + 8 * q) * 2;
      a_voff[i] = (unsigned)(nbase + inner);
      unsigned xb = 0u, mk = 0u;
      for (int ib = 0; ib < nkx; ++ib) {
        const int ix = MODE1 ? bx - ib : bx + ib;
        bool ok = ix >= 0 && ix < p.Win;
        if (MODE1) ok = ok && (ox + p.pb - (kxs + kstep * ib)) >= 0;
        if (ok) xb |= 1u << ib;
      }
      for (int ia = 0; ia < nky; ++ia) {
        const int iy = MODE1 ? by - ia : by + ia;
        bool ok = iy >= 0 && iy < p.Hin;
        if (MODE1) ok = ok && (oy + p.pb - (kys + kstep * ia)) >= 0;
        if (ok) mk |= xb << (ia * nkx);
      }
      a_inv[i] = ~mk;   // bits >= ntaps (bit 31 included: "chunk past the end") stay set
      my_taps |= mk;
    }
    if (q == 0) s_offb[lr + RPP * i] = out_off;
  }
  __syncthreads();
  if (q == 0 && my_taps != 0u) atomicOr(s_tmk, my_taps);
  __syncthreads();
  const unsigned all_taps = (ntaps >= 32) ? 0xFFFFFFFFu : ((1u << ntaps) - 1u);
  const unsigned tile_mk = (unsigned)__builtin_amdgcn_readfirstlane((int)(*s_tmk & all_taps));
  const int nvalid = __builtin_popcount(tile_mk);

  // ---- B loader: rows (tid / PQ) + RPP * i of the packed weights [tap][Cout][Cin], piece tid % PQ ----
  unsigned b_voff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int n = n0 + lr + RPP * i;
    b_voff[i] = (n < p.Cout) ? (unsigned)((n * p.Cin + 8 * q) * 2) : VEEX_OOB;
  }

  struct Stage { v4u ra[APASS], rb[BPASS]; };

#ifdef VEEX_DEBUG_TIMING   // timing experiments (results wrong): 1 = no K loop, 2 = no epilogue, 4 = no MFMA, 8 = no LDS
  const int c_end = (p.tap_inner & 1) ? 0 : nvalid * cpt;
#else
  const int c_end = nvalid * cpt;
#endif
  // running (tap, channel slice) of the next chunk to load: per channel slice the set bits of tile_mk, ascending
  int it_c = 0, it_cc = 0;
  unsigned it_bits = tile_mk;
  const int inv_nkx = __builtin_amdgcn_readfirstlane(65536 / nkx + 1);   // tapi / nkx for tapi < 32
  // per-tap displacements, tap t in lane t of two VGPRs (v_readlane with the chunk's tap gives them back as scalars:
  // the scalar multiplies of a per-chunk decode were ~60 SALU instructions per chunk and wave)
  int vtab_d, vtab_w;
  {
    const int ia = (l31 * inv_nkx) >> 16, ib = l31 - ia * nkx;
    const int dpix = MODE1 ? ((nky - 1 - ia) * p.Win + (nkx - 1 - ib)) : (ia * p.Win + ib);
    vtab_d = dpix * p.Cin * 2;
    vtab_w = ((kys + kstep * ia) * p.kw + (kxs + kstep * ib)) * p.Cin * p.Cout * 2;
  }

  // Branch-free on purpose: with `cvalid ? x : 0` on the scalar values the compiler built basic blocks around the
  // multiplies, and at their joins it waited for ALL outstanding loads (s_waitcnt vmcnt(0) in the K loop: the
  // prefetch depth was worth nothing).  A chunk past the end reads zeros: every offset is forced out of range.
  auto load_chunk = [&](Stage& st) {
    const unsigned inval = (it_c < c_end) ? 0u : 0xFFFFFFFFu;
    const int tapi = __builtin_ctz((it_bits & ~inval) | 0x80000000u);        // 31 past the end: bit 31 of a_inv is always set
    const int ci0b = it_cc * (BK * 2);
    const int delta = (__builtin_amdgcn_readlane(vtab_d, tapi) + ci0b) & (int)~inval;
    const int wsoff = (__builtin_amdgcn_readlane(vtab_w, tapi) + ci0b) & (int)~inval;
    ++it_c;
    it_bits &= it_bits - 1u;
    const unsigned wrap = (it_bits == 0u) ? 0xFFFFFFFFu : 0u;
    it_bits |= tile_mk & wrap;
    it_cc -= (int)wrap;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(a_inv[i], tapi, 1);   // 0 or 0xFFFFFFFF
      st.ra[i] = bload4us(rA, a_voff[i] | m, delta);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) st.rb[i] = bload4us(rW, b_voff[i] | inval, wsoff);
  };

  auto store_chunk = [&](const Stage& st, int buf) {
    __bf16* as = As + buf * A_ELEMS + lr * LDKH + 8 * q;
    __bf16* bs = Bs + buf * B_ELEMS + lr * LDKH + 8 * q;
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<v4u*>(as + RPP * i * LDKH) = st.ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<v4u*>(bs + RPP * i * LDKH) = st.rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // lane l: A[row l&31][k = 8*(l>>5) + e], B[col l&31][k = 8*(l>>5) + e], e = 0..7 (one ds_read_b128 each)
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
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // NST chunks of global loads in flight, LDS double-buffered, one barrier per chunk.  At the step of chunk cc (in LDS
  // buffer cc & 1) the registers hold chunks cc+1 .. cc+NST-1; the stage chunk cc came from is free and takes chunk
  // cc+NST.  The loop is unrolled so that stage and buffer indices are compile-time constants.
  constexpr int UNR = (NST % 2 == 0) ? NST : 2 * NST;
  Stage sg[NST];
#pragma unroll
  for (int u = 0; u < NST; ++u) load_chunk(sg[u]);
  store_chunk(sg[0], 0);
  __syncthreads();
  for (int c = 0; c < c_end; c += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (c + u < c_end) {                       // uniform
        load_chunk(sg[u % NST]);                 // chunk c + u + NST
#ifdef VEEX_DEBUG_TIMING
        if (!(p.tap_inner & 4)) compute(u & 1);
        if (!(p.tap_inner & 8)) store_chunk(sg[(u + 1) % NST], (u + 1) & 1);
        else if (sg[(u + 1) % NST].ra[0][0] == 0x12345u) acc[0][0][0] += 1.f;
#else
        compute(u & 1);
        store_chunk(sg[(u + 1) % NST], (u + 1) & 1);
#endif
        __syncthreads();
      }
    }
  }

#ifdef VEEX_DEBUG_TIMING
  if (p.tap_inner & 2) { if (acc[0][0][0] == 12345.f) p.Out[0] = 1.f; return; }
#endif
  gg_epilogue_b<BM, BN, 2, 2>(p, acc, s_offb, reinterpret_cast<float*>(As), btile, 0, n0);
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int BM, int BN, int BK>
static constexpr size_t gb3_lds_bytes() { return (size_t)(2 * (BM + BN) * (BK + 8)) * 2 + (BM + 4) * sizeof(unsigned); }

template <int BM, int BN, bool MODE1, int BK, int NST>
static int gb3_launch3(const GGParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = gb3_lds_bytes<BM, BN, BK>();
  static bool attr_done = false;   // function attributes are per process; set once, outside any capture (warm-up call)
  if (!attr_done && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_bf16_v3_kernel<BM, BN, MODE1, BK, NST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      veex_set_error("gather_gemm(bf16 lean): cannot raise the dynamic LDS limit to %zu bytes", lds); return -3;
    }
  }
  attr_done = true;
  hipLaunchKernelGGL((gather_gemm_bf16_v3_kernel<BM, BN, MODE1, BK, NST>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, bool MODE1>
static int gb3_launch(const GGParams& p, dim3 grid, int bk, int nst, hipStream_t st) {
  if (bk == 64) return (nst >= 3) ? gb3_launch3<BM, BN, MODE1, 64, 3>(p, grid, st) : gb3_launch3<BM, BN, MODE1, 64, 2>(p, grid, st);
  if (nst >= 4) return gb3_launch3<BM, BN, MODE1, 32, 4>(p, grid, st);
  if (nst == 3) return gb3_launch3<BM, BN, MODE1, 32, 3>(p, grid, st);
  return gb3_launch3<BM, BN, MODE1, 32, 2>(p, grid, st);
}

// 1: the lean kernel can run this launch
int veex_gg_bf16_v3_ok(const GGParams& p) {
  const long long a_bytes = (long long)p.Nb * p.Hin * p.Win * p.Cin * 2;
  const long long bias = ((long long)(p.kh - 1) * p.Win + p.kw + (long long)p.pb * (p.Win + 1)) * p.Cin * 2;
  const long long o_bytes = (long long)p.Nb * p.Hout * p.Wout * p.Cout * 4;
  if (p.Abf == nullptr || p.Wbf == nullptr || p.nsplit > 1) return 0;
  if (p.Cin % 32 != 0 || p.Cout % 4 != 0 || p.kh * p.kw > 31) return 0;
  if ((long long)p.Nb * p.Hout * p.Wout >= (1LL << 24)) return 0;
  if (a_bytes + bias >= 0x7FFF0000LL || o_bytes >= 0x7FFF0000LL) return 0;
  if (p.epi == VEEX_EPI_BWD && p.e_scale != nullptr && p.epi_cmask + 1 < 1) return 0;
  return 1;
}

// p: validated by veex_launch_gather_gemm, phase tables filled for `bm`-row tiles (veex_gg_m_tiles)
int veex_launch_gather_gemm_bf16_v3(GGParams& p, int mtiles, int bm, hipStream_t st) {
  const int bn = (bm == 128 && p.Cout > 64) ? 128 : 64;
  const int ntiles = (p.Cout + bn - 1) / bn;
  const int xr = veex_env_int("VEEX_XCD", -1);
  p.remap = (xr < 0) ? (p.nphase == 1) : xr;
  p.gx = ntiles; p.gy = mtiles; p.gz = 1;
  unsigned blocks = (unsigned)ntiles * mtiles;
  // one image row per tile group wherever a launch has vertically padded taps to skip
  const bool vpad = (p.mode == 1) ? true : (p.pb > 0 || (p.Hout - 1) * p.stride + p.kh - 1 - p.pb >= p.Hin);
  p.row_order = (p.kh > 1 && vpad) ? (veex_env_int("VEEX_ROWORD", 1) != 0 ? 1 : 0) : 0;
  if (p.row_order == 1 && xr < 0) {
    const long long rowlen = (long long)p.Nb * p.Wout;
    p.remap = 0;
    if (p.nphase == 1 && rowlen % bm == 0 && (long long)mtiles == p.Hout * (rowlen / bm)) {
      p.remap = 2; p.row_tiles = (int)(rowlen / bm);
      blocks = (unsigned)ntiles * p.Hout * 8u * (unsigned)((p.row_tiles + 7) / 8);
    }
  }
  dim3 grid(blocks, 1, 1);
  // K chunks of 64 channels where the K loop per tap is long and the grid small (as the general kernel)
  const int bfbk = veex_env_int("VEEX_BFBK", 0);
  const bool auto64 = p.Cin >= 128 && (long long)blocks <= 1024;
  const int bk = (p.Cin % 64 == 0 && (bfbk == 64 || (bfbk == 0 && auto64))) ? 64 : 32;
#ifdef VEEX_DEBUG_TIMING
  p.tap_inner = veex_env_int("VEEX_BFDBG", 0);
#endif
  const int nst = veex_env_int("VEEX_BFNST", 4);   // register stages (chunks of loads in flight): 2, 3, 4 (BK 64: 2, 3)
  const bool m1 = p.mode == 1;
  if (bm == 64) return m1 ? gb3_launch<64, 64, true>(p, grid, bk, nst, st) : gb3_launch<64, 64, false>(p, grid, bk, nst, st);
  if (bn == 128) return m1 ? gb3_launch<128, 128, true>(p, grid, bk, nst, st) : gb3_launch<128, 128, false>(p, grid, bk, nst, st);
  return m1 ? gb3_launch<128, 64, true>(p, grid, bk, nst, st) : gb3_launch<128, 64, false>(p, grid, bk, nst, st);
}
