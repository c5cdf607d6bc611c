// Lean 64x64 instance of the fp32 implicit-GEMM engine (gather_gemm.hip has the general one).
//
// Why a second kernel: on gfx950 the exact-f32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles) does NOT overlap
// with other vector instructions of the same SIMD -- measured with tools/micro/mfma_issue.hip: every
// VALU / LDS / VMEM instruction issued beside the MFMAs adds ~4 cycles to the MFMA time, with 1 or 7
// waves per SIMD alike (profiles/r01_micro_mfma_issue.txt).  The K loop of the general kernel carries
// ~30 such instructions per 8 MFMAs (512 cycles); this one carries 15 (forward, BatchNorm prologue) /
// 10 (input gradient):
//   * the gather address of a (row, tap) pair is  voffset | validity  +  an SGPR offset:  the per-row
//     byte offset never changes, the tap / channel-slice displacement is wave-uniform and rides in the
//     buffer instruction's scalar offset (gfx950 range-checks voffset + soffset), and an invalid pair (padding
//     pixel, row past M, chunk past K) ORs the sign-extended bit of an inverted tap mask into the
//     offset, which sends it out of range: the load returns zeros.  2 VALU per A load, 0 per B load;
//   * the producer BatchNorm is 4 v_fma: its shift is fetched through the same masked offset, so a
//     padding pixel gets shift 0 and stays exactly 0 without a select;
//   * chunk -> (tap, channel slice) is carried incrementally in SGPRs (scalar instructions are free beside MFMAs);
//   * taps that are padding for EVERY row of a tile are not executed at all: the rows of a tile run over
//     (image row, sample, column) (GGParams::row_order = 1), so a tile shares its image row and with it the
//     vertical validity of the taps; the valid-tap set of the tile (OR over its rows, one LDS word) is walked with
//     s_ff1.  Skipped products are exact zeros, so the result is bit-identical.
// Constraints (else the general kernel runs): Cin % 16 == 0, operands < 2 GiB, <= 31 taps.
// Same data layout, epilogues and results as gather_gemm_kernel (k order inside a chunk included).
#include "gg_common.h"

__device__ __forceinline__ float4 bload4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}

// q = r / d, rem = r % d for 0 <= r < 2^24, 1 <= d < 2^24: float reciprocal + one correction step instead of the
// ~40-instruction 32-bit integer division (four of them per thread in the prologue of every tile)
__device__ __forceinline__ void g3_divmod(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

// BPK (mode 0 only): the weights come as the packed copy GGParams::Wt = [tap][Cout][Cin] (K contiguous, the layout mode 1
// has natively), so the B tile is staged and read exactly as in mode 1; same values in the same k order: same bits.
// NWM x NWN waves of one 32x32 accumulator each: tile 32*NWM rows x 32*NWN columns, 64*NWM*NWN threads.  2 x 2 is the
// 64x64 tile; 4 x 2 / 2 x 4 / 4 x 4 move 25 / 25 / 50 % fewer operand bytes from L2 per MFMA at the same registers per
// wave (ablation, profiles/r02_f32_lean_ablation_*.txt: with the MFMAs removed a 64x64 launch still takes half its
// time, almost all of it in the operand loads -- 1.7 GB from L2 for deconv4 forward).
template <bool MODE1, int AFF, bool BPK = false, int NWM = 2, int NWN = 2>
__global__ __launch_bounds__(64 * NWM * NWN) __attribute__((amdgpu_waves_per_eu(7, 8))) void gather_gemm_v3_kernel(const GGParams p) {
  constexpr int BM = 32 * NWM, BN = 32 * NWN, BK = 16;
  static_assert((NWM == 2 && NWN == 2) || MODE1 || BPK, "wide tiles stage B as [n][k] rows");
  constexpr int LDK = BK + 4;
  constexpr int A_ELEMS = BM * LDK;
  constexpr int LDN = BN + 4;
  constexpr bool BROW = MODE1 || BPK;      // B tile as [n][k] rows
  constexpr int B_ELEMS = BROW ? BN * LDK : BK * LDN;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  unsigned* s_offb = reinterpret_cast<unsigned*>(Bs + 2 * B_ELEMS);   // [BM] output row byte offsets
  unsigned* s_tmk = s_offb + BM;                                      // valid-tap set of the tile

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int lh = lane >> 5, l31 = lane & 31;
  // staging roles, uniform per wave: thread t stages 16-byte piece t % 4 of A row t / 4 (t < 4*BM) and of B row t / 4 (t < 4*BN)
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  constexpr int NWAVES = NWM * NWN;
  const bool hasA = (BM / 16 >= NWAVES) || swave < BM / 16, hasB = (BN / 16 >= NWAVES) || swave < BN / 16;

  int bnt, btile, kz;
  if (p.remap == 2) {
    // row_order 1, one phase: the tiles (image row y, sample group g) of ALL y go to XCD g % 8 (workgroups are dealt
    // round-robin over the XCDs), in y-major order: a sample group's input rows are shared by three consecutive
    // output rows, and meet in ONE L2 (deconv4 forward: 291 -> 94 MB fetched, -11 % time vs the chunked order).
    // row_order 2 (any number of phases): the same with "output pixel" in place of "image row": tile = pixel * G + g
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int rest = local / p.gx, gq = p.row_tiles >> 3;
    bnt = local - rest * p.gx;
    const int y = rest / gq, g8 = rest - y * gq;
    btile = y * p.row_tiles + g8 * 8 + xcd;
    kz = 0;
  } else {
    veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
  }
  // (say so: once the compiler lost track of their uniformity, every scalar offset derived from them moved into
  //  VGPRs and each buffer load of the K loop became a readfirstlane "waterfall" loop)
  bnt = __builtin_amdgcn_readfirstlane(bnt);
  btile = __builtin_amdgcn_readfirstlane(btile);
  kz = __builtin_amdgcn_readfirstlane(kz);
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
  // row_order >= 2: the tile is ONE output pixel (uy, ux in phase coordinates) of sample group ug -- wave-uniform, so
  // everything that depends on the pixel alone (tap validity, input / output pixel offsets) is scalar arithmetic
  int uy = 0, ux = 0, ug = 0;
  if (p.row_order >= 2) {
    const int t = tile - p.tile_prefix[ph], G = p.row_tiles;
    int pix, g, yi, xi;
    g3_divmod(t, G, 1.0f / (float)G, pix, g);
    g3_divmod(pix, Wp, 1.0f / (float)Wp, yi, xi);
    if (p.row_order == 3) {
      // interior pixels (every tap valid) first, in raster order, then the border rows / columns from the inside out:
      // the longest tiles are dispatched first and the short border tiles fill the tail of the launch (tile lengths
      // differ 9:1 here), while the raster order of the interior keeps the L2 reuse of a plain sweep
      const int bl = p.perm_lo, bh = p.perm_hi;
      auto perm = [bl, bh](int k, int n) {
        const int ni = n - bl - bh;
        if (k < ni) return k + bl;
        int r = k - ni;
        const int pairs = bl < bh ? bl : bh;
        if (r < 2 * pairs) return (r & 1) ? n - bh + (r >> 1) : bl - 1 - (r >> 1);
        r -= 2 * pairs;
        return (bl > bh) ? bl - 1 - pairs - r : n - bh + pairs + r;
      };
      yi = perm(yi, Hp); xi = perm(xi, Wp);
    }
    uy = __builtin_amdgcn_readfirstlane(yi);
    ux = __builtin_amdgcn_readfirstlane(xi);
    ug = __builtin_amdgcn_readfirstlane(g);
  }
  const int n0 = bnt * BN;

  int kys = 0, kxs = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (p.nphase == 4) {
    kys = py; kxs = px; kstep = 2;
    nky = (p.kh - py + 1) >> 1;
    nkx = (p.kw - px + 1) >> 1;
  }
  const int cpt = p.Cin / BK;
  const int ntaps = nky * nkx;
  // row m of the phase -> (sample n, phase row j, phase column i): (n, j, i) order, (j, n, i) with row_order 1,
  // (j, i, n) with row_order 2 / 3
  const int D1 = p.row_order ? p.Nb * Wp : Hp * Wp;
  const int D2 = (p.row_order >= 2) ? p.Nb : Wp;
  const float invD1 = 1.0f / (float)D1, invD2 = 1.0f / (float)D2;   // rows < 2^24 (veex_gg_v3_ok)
  auto row_nji = [&](int m, int& n, int& j, int& i) {
    int a, rem, b, c;
    g3_divmod(m, D1, invD1, a, rem);
    g3_divmod(rem, D2, invD2, b, c);
    if (p.row_order >= 2)      { j = a; i = b; n = c; }
    else if (p.row_order == 1) { j = a; n = b; i = c; }
    else                       { n = a; j = b; i = c; }
  };

  if (tid == 0) *s_tmk = 0u;

  // Bias: the scalar offset of a tap must be >= 0, so the per-row offset is taken at the tap with the lowest
  // address; for rows at the top-left border that pixel lies before the tensor, hence every per-row offset
  // is shifted up by `bias` bytes and the resource base down by the same amount (never dereferenced there:
  // such a (row, tap) pair is masked).
  const int bias_pix = MODE1 ? ((nky - 1) * p.Win + (nkx - 1)) : p.pb * (p.Win + 1);
  const int bias_bytes = bias_pix * p.Cin * 4;
  const unsigned a_total = (unsigned)p.Nb * p.Hin * p.Win * p.Cin * 4u;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)) - bias_bytes, 0, (int)(a_total + (unsigned)bias_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(BPK ? p.Wt : p.W), 0, (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 4u), 0x00020000);
  const unsigned aff_bytes = (AFF != 0) ? (unsigned)(p.aff_mask + 1) * 4u : 0u;
  const int aff_bias = (AFF == 2) ? bias_bytes : 0;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(AFF != 0 ? p.a_scale : p.A)) - aff_bias, 0, (int)(aff_bytes + (unsigned)aff_bias), 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(AFF != 0 ? p.a_shift : p.A)) - aff_bias, 0, (int)(aff_bytes + (unsigned)aff_bias), 0x00020000);

  // ---- A loader: row tid/4, k-quad tid%4 ----
  const int q = tid & 3, lr = tid >> 2;
  unsigned a_voff = 0u, a_voff2 = 0u, a_inv = 0xFFFFFFFFu;   // a_inv bit t set <=> tap t must read zeros
  unsigned umk = 0u;                                          // row_order >= 2: the tile's valid taps (scalar)
  if (p.row_order >= 2) {
    // one pixel per tile (Nb % BM == 0: every row is a sample): pixel geometry and tap validity in scalar registers,
    // per lane only the sample's base offset -- no divisions, no per-lane mask loops, no LDS atomics
    const int oy = ostep * uy + oy0, ox = ostep * ux + ox0;
    int by, bx;
    if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
    else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
    unsigned xb = 0u;
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
      if (ok) umk |= xb << (ia * nkx);
    }
    const int pix_in = (MODE1 ? (by * p.Win + bx) : ((by + p.pb) * p.Win + bx + p.pb)) * p.Cin * 4;   // scalar
    const int pix_out = (oy * p.Wout + ox) * p.Cout * 4;
    const int n = ug * BM + lr;
    unsigned out_off = 0x80000000u;
    if (hasA) {
      out_off = (unsigned)(n * (p.Hout * p.Wout * p.Cout * 4) + pix_out);
      a_voff2 = (unsigned)(pix_in + 16 * q);
      a_voff = (unsigned)(n * (p.Hin * p.Win * p.Cin * 4)) + a_voff2;
      a_inv = ~umk;
    }
    if (q == 0 && hasA) s_offb[lr] = out_off;
  } else {
    const int m = m0 + lr;
    unsigned out_off = 0x80000000u;   // + any column offset stays out of range: the store is dropped
    if (m < Mph && hasA) {
      int n, j, ii;
      row_nji(m, n, j, ii);
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      out_off = (unsigned)(((n * p.Hout + oy) * p.Wout + ox) * p.Cout) * 4u;
      int by, bx;   // input pixel read by phase tap (0,0)
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      // lowest-address tap: MODE1 (by-(nky-1), bx-(nkx-1)), MODE0 (by, bx); plus the bias
      const int nbase = n * p.Hin * p.Win * p.Cin * 4;
      const int inner = ((MODE1 ? (by * p.Win + bx) : ((by + p.pb) * p.Win + bx + p.pb)) * p.Cin + 4 * q) * 4;
      a_voff = (unsigned)(nbase + inner);
      a_voff2 = (unsigned)inner;
      // tap (ia, ib) is valid iff its row is and its column is: a column mask, shifted in for every valid row
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
      a_inv = ~mk;   // bits >= ntaps (bit 31 included: "chunk past the end") stay set
    }
    if (q == 0 && hasA) s_offb[lr] = out_off;   // the epilogue's output row byte offsets (read after the K loop's barriers)
  }
  // valid-tap set of the tile: split-K launches keep every tap (their chunk ranges are planned on the host)
  const unsigned all_taps = (ntaps >= 32) ? 0xFFFFFFFFu : ((1u << ntaps) - 1u);
  unsigned tile_mk;
  if (p.row_order >= 2) {
    tile_mk = (unsigned)__builtin_amdgcn_readfirstlane((int)(umk & all_taps));
  } else {
    __syncthreads();
    if (q == 0 && ~a_inv != 0u) atomicOr(s_tmk, ~a_inv);
    __syncthreads();
    tile_mk = (p.nsplit > 1) ? all_taps : (unsigned)__builtin_amdgcn_readfirstlane((int)(*s_tmk & all_taps));
  }
  const int nvalid = __builtin_popcount(tile_mk);
  // ---- B loader ----
  unsigned b_voff;
  if (BROW) {    // [tap][Cout][Cin]: row n0 + tid/4, quad q along Cin
    const int n = n0 + lr;
    b_voff = (n < p.Cout && hasB) ? (unsigned)((n * p.Cin + 4 * q) * 4) : VEEX_OOB;
  } else {       // [tap][Cin][Cout]: row tid/16, quad tid%16 along Cout
    const int n = n0 + 4 * (tid & 15);
    b_voff = (n < p.Cout) ? (unsigned)(((tid >> 4) * p.Cout + n) * 4) : VEEX_OOB;
  }
  const int tap_bytes = p.Cin * p.Cout * 4;
  const unsigned s_voff = (unsigned)(16 * q);   // AFF == 1: this thread's 4 channels of the slice

  struct Stage { float4 ra, rb, rsc, rsh; };

  const int nchunks = nvalid * cpt;
  const int c_begin = kz * p.chunks_per_split;
#ifdef VEEX_DEBUG_TIMING   // timing experiments (results wrong): 1 = no K loop, 2 = no epilogue, 4 = no MFMA, 8 = no LDS stores, 16 = no loads
  const int c_end = (p.tap_inner & 1) ? c_begin : min(nchunks, c_begin + p.chunks_per_split);
#else
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);
#endif
  // running (tap, channel slice) of the next chunk to load; chunks are loaded strictly in order: per channel slice
  // the set bits of tile_mk in ascending order
  int it_c = c_begin;
  int it_cc = (nvalid > 0) ? c_begin / nvalid : 0;
  unsigned it_bits = tile_mk;
  for (int k = c_begin - it_cc * nvalid; k > 0; --k) it_bits &= it_bits - 1u;   // (split-K: tile_mk is the full set)
  const int inv_nkx = __builtin_amdgcn_readfirstlane(65536 / nkx + 1);   // tapi / nkx for tapi < 32

  auto load_chunk = [&](Stage& st) {
    const bool cvalid = it_c < c_end;
    const int tapi = cvalid ? __builtin_ctz(it_bits | 0x80000000u) : 31;     // bit 31 of a_inv is always set: zeros
    const int ia = (tapi * inv_nkx) >> 16, ib = tapi - ia * nkx;
    const int ci0b = cvalid ? it_cc * (BK * 4) : 0;
    const int tap = cvalid ? (kys + kstep * ia) * p.kw + (kxs + kstep * ib) : 0;
    const int dpix = MODE1 ? ((nky - 1 - ia) * p.Win + (nkx - 1 - ib)) : (ia * p.Win + ib);
    const int delta = cvalid ? dpix * p.Cin * 4 + ci0b : 0;
    ++it_c;
    it_bits &= it_bits - 1u;
    if (it_bits == 0u) { it_bits = tile_mk; ++it_cc; }

    const unsigned m = (unsigned)__builtin_amdgcn_sbfe(a_inv, tapi, 1);   // 0 or 0xFFFFFFFF
    // every thread issues every load: a thread without the role has its offset out of range (zeros, no memory
    // access).  Branching around the loads made the compiler wait for ALL outstanding loads at the joins.
    st.ra = bload4s(rA, a_voff | m, delta);
    if (AFF == 1) {
      st.rsc = bload4s(rS, s_voff | m, ci0b);
      st.rsh = bload4s(rH, s_voff | m, ci0b);
    } else if (AFF == 2) {
      st.rsc = bload4s(rS, a_voff2 | m, delta);
      st.rsh = bload4s(rH, a_voff2 | m, delta);
    }
    st.rb = bload4s(rW, b_voff, tap * tap_bytes + (BROW ? ci0b : ci0b * p.Cout));
  };

  auto store_chunk = [&](const Stage& st, int buf) {
    float* as = As + buf * A_ELEMS;
    float* bs = Bs + buf * B_ELEMS;
    if (hasA) {
      float4 v = st.ra;
      if (AFF != 0) {   // padding pixels: value 0, shift 0 -> stay exactly 0
        v.x = fmaf(v.x, st.rsc.x, st.rsh.x); v.y = fmaf(v.y, st.rsc.y, st.rsh.y);
        v.z = fmaf(v.z, st.rsc.z, st.rsh.z); v.w = fmaf(v.w, st.rsc.w, st.rsh.w);
      }
      *reinterpret_cast<float4*>(as + lr * LDK + 4 * q) = v;
    }
    if (hasB) {
      if (BROW) *reinterpret_cast<float4*>(bs + lr * LDK + 4 * q) = st.rb;
      else      *reinterpret_cast<float4*>(bs + (tid >> 4) * LDN + 4 * (tid & 15)) = st.rb;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // K order inside a chunk is permuted: lanes 0-31 feed k = 0..7, lanes 32-63 k = 8..15
  auto compute = [&](int cur) {
    const float* as = As + cur * A_ELEMS + (wm * 32 + l31) * LDK + 8 * lh;
    const float* bs = BROW ? (Bs + cur * B_ELEMS + (wn * 32 + l31) * LDK + 8 * lh)
                           : (Bs + cur * B_ELEMS + (8 * lh) * LDN + wn * 32 + l31);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float av[4], bv[4];
      const float4 t = *reinterpret_cast<const float4*>(as + 4 * half);
      av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
      if (BROW) {
        const float4 u = *reinterpret_cast<const float4*>(bs + 4 * half);
        bv[0] = u.x; bv[1] = u.y; bv[2] = u.z; bv[3] = u.w;
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bv[kk] = bs[(4 * half + kk) * LDN];
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
    }
  };

  // two chunks of global loads in flight, one barrier per chunk (see gather_gemm_kernel)
  Stage s0, s1;
  load_chunk(s0);
  store_chunk(s0, 0);
  load_chunk(s0);
  __syncthreads();
#ifdef VEEX_DEBUG_TIMING
  for (int c = c_begin; c < c_end; c += 2) {
    if (!(p.tap_inner & 16)) load_chunk(s1);
    if (!(p.tap_inner & 4)) compute(0);
    if (!(p.tap_inner & 8)) store_chunk(s0, 1); else if (s0.ra.x == 12345.f) acc[0] += 1.f;
    __syncthreads();
    if (!(p.tap_inner & 16)) load_chunk(s0);
    if (!(p.tap_inner & 4)) compute(1);
    if (!(p.tap_inner & 8)) store_chunk(s1, 0); else if (s1.ra.x == 12345.f) acc[0] += 1.f;
    __syncthreads();
  }
  if (p.tap_inner & 2) { if (acc[0] == 12345.f) p.Out[0] = 1.f; return; }
#else
  for (int c = c_begin; c < c_end; c += 2) {
    load_chunk(s1);
    compute(0);
    store_chunk(s0, 1);
    __syncthreads();
    load_chunk(s0);
    compute(1);
    store_chunk(s1, 0);
    __syncthreads();
  }
#endif

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  const int col = n0 + wn * 32 + l31;
  const bool colok = col < p.Cout;
  float* outp = p.Out;
  if (p.epi == VEEX_EPI_RAW) outp += (long long)kz * p.slab_stride;
  const unsigned out_bytes = (unsigned)p.Nb * p.Hout * p.Wout * p.Cout * 4u;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);
  const unsigned col4 = (unsigned)col * 4u;
  float cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
  if (colok) {
    if (p.epi == VEEX_EPI_FWD) {
      const float bia = (p.bias != nullptr) ? p.bias[col] : 0.f;
      // the BatchNorm-applied copy for the fp32 consumers of this activation (zero records without one: stores dropped)
      const bool has_bn = p.OutBn != nullptr;
      const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(has_bn ? p.OutBn : outp, 0, has_bn ? (int)out_bytes : 0, 0x00020000);
      const float osc = has_bn ? p.o_scale[col & p.o_cmask] : 1.f, osh = has_bn ? p.o_shift[col & p.o_cmask] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 ro = *reinterpret_cast<const uint4*>(s_offb + wm * 32 + 8 * g + 4 * lh);
        const unsigned rof[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[4 * g + e] + bia;
          if (p.relu) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[e] + col4, 0, 0);
          if (has_bn) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(v, osc, osh)), rN, rof[e] + col4, 0, 0);
        }
      }
    } else if (p.epi == VEEX_EPI_BWD) {
      const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.aux), 0, (int)out_bytes, 0x00020000);
      const float esc = (p.e_scale != nullptr) ? p.e_scale[col & p.epi_cmask] : 1.f;
      // all 16 activations first: loads and stores go through different descriptors the compiler cannot tell apart,
      // so four load-use-store groups would pay four memory latencies per tile
      unsigned rof[16];
      float a[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 ro = *reinterpret_cast<const uint4*>(s_offb + wm * 32 + 8 * g + 4 * lh);
        rof[4 * g] = ro.x; rof[4 * g + 1] = ro.y; rof[4 * g + 2] = ro.z; rof[4 * g + 3] = ro.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)   // rows past M: out of range -> 0 (and their accumulators are 0)
        a[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, rof[r] + col4, 0, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        const float vm = (!p.relu || a[r] > 0.f) ? v : 0.f;
        cs1 += v;
        cs2 = fmaf(v, a[r], cs2);
        cs3 += vm;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vm * esc), rO, rof[r] + col4, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 ro = *reinterpret_cast<const uint4*>(s_offb + wm * 32 + 8 * g + 4 * lh);
        const unsigned rof[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[4 * g + e];   // (bit-casting the vector element in place stored acc[0] sixteen times)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[e] + col4, 0, 0);
        }
      }
    }
  }

  if (p.epi == VEEX_EPI_BWD && p.partials != nullptr) {
    // column sums: lane halves -> the two waves sharing the columns -> one row of the partial slab
    float* s_red = As;   // [NWM][BN][3]; safe: every wave passed the loop's last barrier
    cs1 += __shfl_xor(cs1, 32);
    cs2 += __shfl_xor(cs2, 32);
    cs3 += __shfl_xor(cs3, 32);
    if (lh == 0) {
      const int cl = wn * 32 + l31;
      s_red[(wm * BN + cl) * 3 + 0] = cs1;
      s_red[(wm * BN + cl) * 3 + 1] = cs2;
      s_red[(wm * BN + cl) * 3 + 2] = cs3;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int w = 0; w < NWM; ++w) {     // in wave order: 2 x 2 sums exactly as before
        t1 += s_red[(w * BN + tid) * 3 + 0];
        t2 += s_red[(w * BN + tid) * 3 + 1];
        t3 += s_red[(w * BN + tid) * 3 + 2];
      }
      float* pp = p.partials + (long long)btile * 3 * p.Cout + n0 + tid;
      pp[0] = t1;
      pp[p.Cout] = t2;
      pp[2 * p.Cout] = t3;
    }
  }
}

template <bool MODE1, bool BPK, int NWM, int NWN>
static void gg3_launch(const GGParams& p, int aff, dim3 grid, hipStream_t st) {
  constexpr int BM = 32 * NWM, BN = 32 * NWN;
  constexpr size_t lds = (2 * BM * 20 + 2 * ((MODE1 || BPK) ? BN * 20 : 16 * 68)) * sizeof(float) + (BM + 4) * sizeof(unsigned);
  const dim3 blk(64 * NWM * NWN);
  if (aff == 0) hipLaunchKernelGGL((gather_gemm_v3_kernel<MODE1, 0, BPK, NWM, NWN>), grid, blk, lds, st, p);
  else if (aff == 1) hipLaunchKernelGGL((gather_gemm_v3_kernel<MODE1, 1, BPK, NWM, NWN>), grid, blk, lds, st, p);
  else hipLaunchKernelGGL((gather_gemm_v3_kernel<MODE1, 2, BPK, NWM, NWN>), grid, blk, lds, st, p);
}

// 1: the lean kernel can run this launch (64-row tiles already chosen by the caller)
int veex_gg_v3_ok(const GGParams& p) {
  const long long a_bytes = (long long)p.Nb * p.Hin * p.Win * p.Cin * 4;
  const long long bias = ((long long)(p.kh - 1) * p.Win + p.kw + (long long)p.pb * (p.Win + 1)) * p.Cin * 4;
  const long long o_bytes = (long long)p.Nb * p.Hout * p.Wout * p.Cout * 4;
  const long long w_bytes = (long long)p.kh * p.kw * p.Cin * p.Cout * 4;
  if (p.Cin % 16 != 0 || p.kh * p.kw > 31) return 0;
  if ((long long)p.Nb * p.Hout * p.Wout >= (1LL << 24)) return 0;
  if (p.mode == 0 && p.Cout % 4 != 0) return 0;
  if (a_bytes + bias >= 0x7FFF0000LL || o_bytes >= 0x7FFF0000LL || w_bytes >= 0x7FFF0000LL) return 0;
  if (p.OutBf != nullptr || p.Abf != nullptr) return 0;
  if (p.a_scale != nullptr && p.aff_spatial && (p.aff_mask + 1) != p.Hin * p.Win * p.Cin) return 0;
  if (p.a_scale != nullptr && !p.aff_spatial && (p.aff_mask + 1) < p.Cin) return 0;
  if (p.epi == VEEX_EPI_BWD && p.e_scale != nullptr && p.epi_cmask + 1 < 1) return 0;
  return 1;
}

// Tile of the lean kernel: 64 x 64 (veex_gg_bm is the one source of the row count: callers size the column partials
// by it).  128-row / 128-column tiles were measured slower on every layer (profiles/r02_f32_tiles.txt) and are gone.
int veex_gg_v3_bm(const GGParams&) { return 64; }

// p: validated by veex_launch_gather_gemm, phase tables filled for veex_gg_bm(p)-row tiles
int veex_launch_gather_gemm_v3(GGParams& p, int mtiles, hipStream_t st) {
  constexpr int bm = 64, bn = 64;
  const int ntiles = (p.Cout + bn - 1) / bn;
  if (p.nsplit > 1) {
    const int nchunks = p.kh * p.kw * (p.Cin / 16);
    p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
  }
  const int xr = veex_tune_int("VEEX_XCD", -1);
  p.remap = (xr < 0) ? (p.nphase == 1) : xr;
  p.gx = ntiles; p.gy = mtiles; p.gz = p.nsplit;
  dim3 grid((unsigned)ntiles * mtiles * p.nsplit, 1, 1);
  // one image row per tile wherever a launch has vertically padded taps to skip (A/B at batch 512: -2...-8 % on
  // those launches, +3 % on a launch without padding, whose tiles then only lose locality)
  const bool vpad = (p.mode == 1) ? true : (p.pb > 0 || (p.Hout - 1) * p.stride + p.kh - 1 - p.pb >= p.Hin);
  const int ro = veex_tune_int("VEEX_ROWORD", 2);
  p.row_order = (p.kh > 1 && p.nsplit == 1 && vpad) ? (ro > 1 ? 1 : ro) : 0;
  // one output PIXEL per tile (its rows = bm consecutive samples) when the batch is a multiple of the tile height:
  // every row of the tile then has the same valid taps, vertically AND horizontally, and the tile executes exactly
  // the in-bounds taps (deconv4 forward: 2 601 tap-pixels instead of the 2 907 that vertical skipping leaves)
  if (p.row_order == 1 && ro == 2 && p.Nb % bm == 0 && mtiles % (p.Nb / bm) == 0) {
    // interior-first pixel order for one-phase launches (the 4-phase launches already run their longest phase first).
    // Border widths = output rows / columns with fewer than kh valid taps, low and high side (square maps).
    p.row_order = 2;
    if (p.nphase == 1 && veex_tune_int("VEEX_TILEPERM", 1)) {
      int lo, hi;
      if (p.mode == 1) { lo = p.kh - 1 - p.pb; hi = p.Hout - p.Hin + p.pb; }                       // stride 1 here
      else { lo = (p.pb + p.stride - 1) / p.stride; hi = p.Hout - 1 - (p.Hin - p.kh + p.pb) / p.stride; }
      if (lo >= 0 && hi >= 0 && lo + hi < p.Hout && lo + hi < p.Wout && p.Hout == p.Wout && p.Hin == p.Win && lo + hi > 0) {
        p.row_order = 3; p.perm_lo = lo; p.perm_hi = hi;
      }
    }
    p.row_tiles = p.Nb / bm;
    if (xr < 0) {
      p.remap = 0;
      if ((p.Nb / bm) % 8 == 0) { p.remap = 2; p.row_tiles = p.Nb / bm; }
    }
  }
  if (p.row_order == 1 && xr < 0) {
    // the chunked XCD order would split a sample group's image rows over the XCDs: group-per-XCD order where the
    // tiles line up with the image rows, plain round-robin otherwise
    const long long rowlen = (long long)p.Nb * p.Wout;
    p.remap = 0;
    if (p.nphase == 1 && rowlen % bm == 0 && (rowlen / bm) % 8 == 0 && (long long)mtiles == p.Hout * (rowlen / bm)) {
      p.remap = 2; p.row_tiles = (int)(rowlen / bm);
    }
  }
  const int aff = (p.a_scale == nullptr) ? 0 : (p.aff_spatial ? 2 : 1);
#ifdef VEEX_DEBUG_TIMING
  p.tap_inner = veex_tune_int("VEEX_F32DBG", 0);
#endif
  const bool bpk = p.mode == 0 && p.Wt != nullptr && veex_tune_int("VEEX_WPACKF", 1);
  if (p.mode == 1) gg3_launch<true, false, 2, 2>(p, aff, grid, st);
  else if (bpk) gg3_launch<false, true, 2, 2>(p, aff, grid, st);
  else gg3_launch<false, false, 2, 2>(p, aff, grid, st);
  VEEX_CHECK_LAUNCH();
  return 0;
}
