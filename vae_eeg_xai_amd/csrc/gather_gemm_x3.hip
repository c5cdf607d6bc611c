// fp32 implicit GEMM on the bf16 matrix cores: every fp32 operand is split into three bf16 terms
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)          (3 x 8 = 24 significand bits)
// and the product a*b is accumulated in fp32 from the six partial products of order <= 2
//     a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1).
// Each partial product of two bf16 values is exact in fp32; the three dropped ones are <= 2^-24 of |a*b| each, the size
// of the one rounding a fused fp32 multiply-add makes anyway.  The result is an fp32 dot product -- not a bf16 one --
// at 6 MFMAs of v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 of v_mfma_f32_32x32x2_f32 at twice the latency: the
// matrix pipe has 16/6 = 2.7x the fp32 rate on it (417 TFLOP/s against 157).
//
// Same gather addressing, phases, producer-BatchNorm prologue and epilogues as gather_gemm_bf16.hip (reference ops:
// Conv2D / Conv2DTranspose forward, models.py:219-298, and their input gradients).  Activations are fp32 in HBM and are
// split while a tile is staged into LDS (after the BN affine); the weights come pre-split as three packed bf16 planes
// [tap][Cin/16][plane][Cout][16] (veex_pack_weights_bf16x3, once per step from the fp32 master weights: a K chunk's
// weight tile is one contiguous run per plane).
// Tiles from 64 x 64 x 32 (4 waves) to 256 x 128 x 16 (8 waves, 74 KB of LDS: 3 planes x 2 buffers), chosen per layer;
// LDS rows are unpadded with their 16-byte pieces XOR-swizzled by the row (no bank conflict on either side).
#include "gg_common.h"
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 x3_bload4u(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4u f = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void x3_fdivmod24(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}
// v = h1 + h2 + h3 (+ <= 2^-24 |v|)
__device__ __forceinline__ void x3_split(float v, __bf16& h1, __bf16& h2, __bf16& h3) {
  h1 = (__bf16)v;
  const float r1 = v - (float)h1;     // exact
  h2 = (__bf16)r1;
  const float r2 = r1 - (float)h2;    // exact
  h3 = (__bf16)r2;
}

// DBG (a `make TIMING=1` build only; results are WRONG by construction): 1 no global loads, 2 no split arithmetic (one
// conversion per element), 4 no LDS stores, 8 no MFMAs (the operand reads stay), 16 no operand reads (the MFMAs stay).
// OPT bit 0: unpadded LDS rows with the 16-byte pieces of a row XOR-swizzled by the row (conflict-free ds_read_b128 AND
// ds_write_b128; the padded rows' stores are 2-way); bit 1: a step converts and stores the next chunk BEFORE its MFMAs in
// program order (its loads were issued a whole step earlier), so the scheduler may spread the stores under the MFMAs.
template <int BM, int BN, int BK, int WM, int WN, bool MODE1, int AFF, int NST, int OPT = 0, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN) void gather_gemm_x3_kernel(const GGParams p) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr bool SWZ = (OPT & 1) != 0, STORE_FIRST = (OPT & 2) != 0;
  constexpr int LDKH = SWZ ? BK : BK + 8;   // bf16 elements per LDS row (padded: 80 / 48 B, conflict-free b128 reads)
  // swizzle of a row's 16-byte pieces: 64-byte rows (BK 32) by (row >> 2) & 3, 32-byte rows (BK 16) by (row >> 3) & 1
  auto swz = [](int row) { return SWZ ? (BK == 32 ? (row >> 2) & 3 : (row >> 3) & 1) : 0; };
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int AQ = BK / 8;           // A: 32-byte pieces (8 fp32 channels = two 16-byte loads) per row
  constexpr int ARP = NTHR / AQ, APASS = BM / ARP;
  constexpr int BPIECES = (BK / 16) * 3 * BN * 2;           // B: 16-byte pieces (8 bf16) of the chunk's three planes
  constexpr int BPASS = (BPIECES + NTHR - 1) / NTHR;
  constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;   // per plane
  static_assert(APASS >= 1 && BM % ARP == 0 && MT >= 1 && NT >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);          // [buf][plane][BM][LDKH]
  __bf16* Bs = As + 2 * 3 * A_ELEMS;                         // [buf][plane][BN][LDKH]
  int* s_off = reinterpret_cast<int*>(Bs + 2 * 3 * B_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bnt, btile, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
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
  const int n0 = bnt * BN;

  int kys = 0, kxs = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (p.nphase == 4) {
    kys = py; kxs = px; kstep = 2;
    nky = (p.kh - py + 1) >> 1;
    nkx = (p.kw - px + 1) >> 1;
  }
  const int cpt = p.Cin / BK;
  const int ntaps = nky * nkx;
  const int nchunks = ntaps * cpt;
  const int c_begin = kz * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);
  const int HWp = Hp * Wp;
  const float invHWp = 1.0f / (float)HWp, invWp = 1.0f / (float)Wp;

  if (tid < BM) {
    int m = m0 + tid, off = -1;
    if (m < Mph) {
      int n, rem, j, i;
      x3_fdivmod24(m, HWp, invHWp, n, rem);
      x3_fdivmod24(rem, Wp, invWp, j, i);
      off = (n * p.Hout + ostep * j + oy0) * p.Wout + ostep * i + ox0;
    }
    s_off[tid] = (off < 0) ? (int)0x80000000 : off * p.Cout * 4;
  }

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, (int)((unsigned)p.Nb * p.Hin * p.Win * p.Cin * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.Wbf), 0, (int)(3u * (unsigned)p.kh * p.kw * p.Cin * p.Cout * 2u), 0x00020000);
  const unsigned aff_bytes = (AFF != 0) ? (unsigned)(p.aff_mask + 1) * 4u : 0u;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_scale : p.A), 0, (int)aff_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_shift : p.A), 0, (int)aff_bytes, 0x00020000);

  // ---- A loader: rows (tid/AQ)+ARP*i, 32-byte piece (tid%AQ) = 8 channels of the chunk ----
  const int q = tid % AQ, lr = tid / AQ;
  int a_rowbase[APASS], a_nbase[APASS];
  unsigned a_mask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + lr + ARP * i;
    a_rowbase[i] = 0; a_mask[i] = 0u; a_nbase[i] = 0;
    if (m < Mph) {
      int n, rem, j, ii;
      x3_fdivmod24(m, HWp, invHWp, n, rem);
      x3_fdivmod24(rem, Wp, invWp, j, ii);
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      int by, bx;
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      a_nbase[i] = n * p.Hin * p.Win * p.Cin * 4;
      a_rowbase[i] = a_nbase[i] + ((by * p.Win + bx) * p.Cin) * 4 + 32 * q;
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
  // ---- B loader: the chunk's weights are ONE contiguous run per (16-channel half, plane) in the packed layout
  // [tap][Cin/16][plane][Cout][16]: piece pi = tid + NTHR*i -> (half h, plane, row, 16-byte piece bq) ----
  int b_goff[BPASS], b_lds[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int pi = tid + NTHR * i, bq = pi & 1, vr = pi >> 1;
    const int t = vr / BN, row = vr - t * BN, h = t / 3, pl = t - 3 * h;
    b_ok[i] = pi < BPIECES && n0 + row < p.Cout;
    b_goff[i] = ((h * 3 + pl) * p.Cout + n0 + row) * 32 + bq * 16;
    b_lds[i] = pl * B_ELEMS + row * LDKH + ((h * 2 + bq) ^ swz(row)) * 8;
  }
  const int c16_bytes = 3 * p.Cout * 32;          // one 16-channel slice of one tap, three planes
  const int tap_c16 = p.Cin / 16;

  struct Stage {
    float4 ra[APASS][2], rsc[AFF == 2 ? APASS : 1][2], rsh[AFF == 2 ? APASS : 1][2];
    uint4 rb[BPASS];
    bool rok[APASS];
  };

  int it_cc = c_begin / ntaps;
  int it_tapi = c_begin - it_cc * ntaps;
  int it_ia = it_tapi / nkx, it_ib = it_tapi - it_ia * nkx;
  it_cc = __builtin_amdgcn_readfirstlane(it_cc); it_tapi = __builtin_amdgcn_readfirstlane(it_tapi);
  it_ia = __builtin_amdgcn_readfirstlane(it_ia); it_ib = __builtin_amdgcn_readfirstlane(it_ib);

  auto load_chunk = [&](Stage& st, bool cvalid) {
    const int cc = it_cc, tapi = it_tapi, ia = it_ia, ib = it_ib;
    ++it_tapi; ++it_ib;
    if (it_ib == nkx) { it_ib = 0; ++it_ia; }
    if (it_tapi == ntaps) { it_tapi = 0; it_ia = 0; it_ib = 0; ++it_cc; }
    const int tap = (kys + kstep * ia) * p.kw + (kxs + kstep * ib);
    const int ci0 = cc * BK;
    const int dpix = ia * p.Win + ib;
    const int delta = ((MODE1 ? -dpix : dpix) * p.Cin + ci0) * 4;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = cvalid && ((a_mask[i] >> tapi) & 1u);
      const unsigned aoff = ok ? (unsigned)(a_rowbase[i] + delta) : VEEX_OOB;
      if (DBG & 1) {
        asm volatile("" : "=v"(st.ra[i][0].x), "=v"(st.ra[i][0].y), "=v"(st.ra[i][0].z), "=v"(st.ra[i][0].w) : "v"(aoff));
        asm volatile("" : "=v"(st.ra[i][1].x), "=v"(st.ra[i][1].y), "=v"(st.ra[i][1].z), "=v"(st.ra[i][1].w) : "v"(aoff));
      } else {
        st.ra[i][0] = bload4(rA, aoff);
        st.ra[i][1] = bload4(rA, aoff + 16u);
      }
      st.rok[i] = ok;
      if (AFF == 2) {
        const unsigned ao = ok ? (unsigned)(a_rowbase[i] + delta - a_nbase[i]) : VEEX_OOB;
        st.rsc[i][0] = bload4(rS, ao); st.rsc[i][1] = bload4(rS, ao + 16u);
        st.rsh[i][0] = bload4(rH, ao); st.rsh[i][1] = bload4(rH, ao + 16u);
      }
    }
    if (AFF == 1) {
      const unsigned ao = cvalid ? (unsigned)(((ci0 + 8 * q) & p.aff_mask) * 4) : VEEX_OOB;
      st.rsc[0][0] = bload4(rS, ao); st.rsc[0][1] = bload4(rS, ao + 16u);
      st.rsh[0][0] = bload4(rH, ao); st.rsh[0][1] = bload4(rH, ao + 16u);
    }
    const int wbase = (tap * tap_c16 + (ci0 >> 4)) * c16_bytes;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const bool ok = cvalid && b_ok[i];
      const unsigned boff = ok ? (unsigned)(wbase + b_goff[i]) : VEEX_OOB;
      if (DBG & 1) asm volatile("" : "=v"(st.rb[i].x), "=v"(st.rb[i].y), "=v"(st.rb[i].z), "=v"(st.rb[i].w) : "v"(boff));
      else st.rb[i] = x3_bload4u(rW, boff);
    }
  };

  auto store_chunk = [&](const Stage& st, int buf) {
    __bf16* as = As + buf * 3 * A_ELEMS;
    __bf16* bs = Bs + buf * 3 * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      bf16x8 h1, h2, h3;
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        float4 v = st.ra[i][hq];
        if (AFF != 0) {
          const float4 sc = st.rsc[AFF == 2 ? i : 0][hq], sh = st.rsh[AFF == 2 ? i : 0][hq];
          v.x = st.rok[i] ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = st.rok[i] ? fmaf(v.y, sc.y, sh.y) : 0.f;
          v.z = st.rok[i] ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = st.rok[i] ? fmaf(v.w, sc.w, sh.w) : 0.f;
        }
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 t1, t2, t3;
          if (DBG & 2) { t1 = (__bf16)vv[e]; t2 = t1; t3 = t1; }
          else x3_split(vv[e], t1, t2, t3);
          h1[4 * hq + e] = t1; h2[4 * hq + e] = t2; h3[4 * hq + e] = t3;
        }
      }
      __bf16* d = as + (lr + ARP * i) * LDKH + 8 * (q ^ swz(lr + ARP * i));
      if (DBG & 4) {
        asm volatile("" : : "v"(h1), "v"(h2), "v"(h3), "v"(d));
      } else {
        *reinterpret_cast<bf16x8*>(d) = h1;
        *reinterpret_cast<bf16x8*>(d + A_ELEMS) = h2;
        *reinterpret_cast<bf16x8*>(d + 2 * A_ELEMS) = h3;
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      if (BPIECES % NTHR == 0 || tid + NTHR * i < BPIECES) {
        if (DBG & 4) asm volatile("" : : "v"(st.rb[i].x), "v"(st.rb[i].y), "v"(st.rb[i].z), "v"(st.rb[i].w), "v"(bs + b_lds[i]));
        else *reinterpret_cast<uint4*>(bs + b_lds[i]) = st.rb[i];
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
    // (tile origins are multiples of 32 rows: the swizzle of a lane's row depends on l31 alone)
    const __bf16* as = As + cur * 3 * A_ELEMS + (wm * WTM + l31) * LDKH;
    const __bf16* bs = Bs + cur * 3 * B_ELEMS + (wn * WTN + l31) * LDKH;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int ko = 8 * ((2 * ks + lh) ^ swz(l31));
      bf16x8 av[3][MT], bv[3][NT];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (DBG & 16) asm volatile("" : "=v"(av[pl][i]) : "v"(as + pl * A_ELEMS + i * 32 * LDKH + ko));
          else av[pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * A_ELEMS + i * 32 * LDKH + ko);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (DBG & 16) asm volatile("" : "=v"(bv[pl][j]) : "v"(bs + pl * B_ELEMS + j * 32 * LDKH + ko));
          else bv[pl][j] = *reinterpret_cast<const bf16x8*>(bs + pl * B_ELEMS + j * 32 * LDKH + ko);
        }
      }
      if (DBG & 8) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int i = 0; i < MT; ++i) asm volatile("" : : "v"(av[pl][i]));
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" : : "v"(bv[pl][j]));
        }
        continue;
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
    }
  };

  // NST - 1 .. NST chunks of global loads in flight (register stages; the gather latency under load is several chunk
  // times); one barrier per chunk.  Step j computes chunk j out of LDS buffer j & 1, stores chunk j + 1 (stage
  // (j + 1) % NST) into the other buffer and issues the loads of chunk j + NST into the stage chunk j came from.
  Stage st[NST];
  load_chunk(st[0], c_begin < c_end);
  store_chunk(st[0], 0);
#pragma unroll
  for (int k = 1; k < NST; ++k) load_chunk(st[k], c_begin + k < c_end);
  if (STORE_FIRST) load_chunk(st[0], c_begin + NST < c_end);   // (stage j % NST holds chunk j; a step refills the stage it stored)
  __syncthreads();
  constexpr int U = (NST % 2 == 0) ? NST : 2 * NST;
  for (int c = c_begin; c < c_end; c += U) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (c + k < c_end) {            // (wave-uniform)
        if (STORE_FIRST) {
          store_chunk(st[(k + 1) % NST], (k + 1) & 1);
          load_chunk(st[(k + 1) % NST], c + k + 1 + NST < c_end);
          compute(k & 1);
        } else {
          load_chunk(st[k % NST], c + k + NST < c_end);
          compute(k & 1);
          store_chunk(st[(k + 1) % NST], (k + 1) & 1);
        }
        __syncthreads();
      }
    }
  }
  gg_epilogue_b<BM, BN, WM, WN>(p, acc, reinterpret_cast<const unsigned*>(s_off), reinterpret_cast<float*>(As), btile, kz, n0);
}

// ---------------------------------------------------------------------------
// packed weights, three planes: dst[plane][tap][c][r] (transpose) or dst[plane][tap][r][c]
// ---------------------------------------------------------------------------
// dst[tap][K/16][plane][N][16]: logical weight W_g[tap][n][k] (n = GEMM output channel, k = GEMM reduction channel) at
// (((tap * K/16 + k/16) * 3 + plane) * N + n) * 16 + k % 16 -- a K chunk's tile of the three planes is contiguous.
// src fp32 [tap][R][C]: transpose != 0 -> n = column, k = row (N = C, K = R); else n = row, k = column.
struct Pack3Job { const float* src; __bf16* dst; int taps, R, C, transpose, block0, blocks; };
struct Pack3Jobs { int n; Pack3Job j[16]; };
// One thread per (tap, 16-channel slice, n): 16 source values (a 64-byte run, or a stride-C column whose neighbours the
// neighbouring threads read) -> one 32-byte row of each plane; consecutive threads write consecutive rows.
__global__ __launch_bounds__(256) void pack_weights_bf16x3_multi_kernel(const Pack3Jobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const Pack3Job J = jobs.j[k];
  const int N = J.transpose ? J.C : J.R, K = J.transpose ? J.R : J.C, K16 = K / 16;
  const long long per = (long long)J.R * J.C, items = (long long)J.taps * K16 * N;
  for (long long i = (long long)(blockIdx.x - J.block0) * 256 + threadIdx.x; i < items; i += (long long)J.blocks * 256) {
    const int nn = (int)(i % N);
    const long long r = i / N;
    const int k16 = (int)(r % K16), t = (int)(r / K16);
    const float* src = J.src + t * per;
    float v[16];
    if (J.transpose) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = src[(long long)(k16 * 16 + j) * J.C + nn];
    } else {
      const float4* s4 = reinterpret_cast<const float4*>(src + (long long)nn * J.C + k16 * 16);   // C % 16 == 0
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 f = s4[j]; v[4 * j] = f.x; v[4 * j + 1] = f.y; v[4 * j + 2] = f.z; v[4 * j + 3] = f.w; }
    }
    bf16x8 h[3][2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      __bf16 t1, t2, t3;
      x3_split(v[j], t1, t2, t3);
      h[0][j >> 3][j & 7] = t1; h[1][j >> 3][j & 7] = t2; h[2][j >> 3][j & 7] = t3;
    }
    __bf16* d = J.dst + (((long long)t * K16 + k16) * 3 * N + nn) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<bf16x8*>(d + (long long)pl * N * 16) = h[pl][0];
      *reinterpret_cast<bf16x8*>(d + (long long)pl * N * 16 + 8) = h[pl][1];
    }
  }
}
int veex_pack_weights_bf16x3_multi(int n, const float* const* src, void* const* dst, const int* taps, const int* rows,
                                   const int* cols, const int* transpose, hipStream_t st) {
  if (n < 1 || n > 16) { veex_set_error("pack_weights_bf16x3_multi: 1..16 jobs"); return -1; }
  for (int i = 0; i < n; ++i)
    if ((transpose[i] ? rows[i] : cols[i]) % 16 != 0 || cols[i] % 4 != 0) { veex_set_error("pack_weights_bf16x3: the reduction dimension must be a multiple of 16 (and the columns of 4)"); return -1; }
  Pack3Jobs jobs; jobs.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const long long items = (long long)taps[i] * rows[i] * cols[i] / 16;   // one thread each
    int b = (int)((items + 255) / 256);
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    jobs.j[i].src = src[i]; jobs.j[i].dst = reinterpret_cast<__bf16*>(dst[i]);
    jobs.j[i].taps = taps[i]; jobs.j[i].R = rows[i]; jobs.j[i].C = cols[i]; jobs.j[i].transpose = transpose[i];
    jobs.j[i].block0 = total; jobs.j[i].blocks = b;
    total += b;
  }
  hipLaunchKernelGGL(pack_weights_bf16x3_multi_kernel, dim3(total), dim3(256), 0, st, jobs);
  VEEX_CHECK_LAUNCH();
  return 0;
}
extern "C" int veex_pack_weights_bf16x3(const float* src, void* dst, int taps, int rows, int cols, int transpose, hipStream_t st) {
  if (src == nullptr || dst == nullptr || taps < 1 || rows < 1 || cols < 1) { veex_set_error("pack_weights_bf16x3: bad arguments"); return -1; }
  return veex_pack_weights_bf16x3_multi(1, &src, &dst, &taps, &rows, &cols, &transpose, st);
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int BM, int BN, int BK, int OPT>
static constexpr size_t x3_lds_bytes() { return (size_t)(2 * 3 * (BM + BN) * (BK + ((OPT & 1) ? 0 : 8))) * 2 + BM * sizeof(int); }

template <int BM, int BN, int BK, int WM, int WN, bool MODE1, int NST, int OPT>
static int x3_launch_n(const GGParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = x3_lds_bytes<BM, BN, BK, OPT>();
  static_assert(x3_lds_bytes<BM, BN, BK, OPT>() <= 160 * 1024, "LDS of one CU");
  const int aff = (p.a_scale == nullptr) ? 0 : (p.aff_spatial ? 2 : 1);
  static bool attr_done = false;   // function attributes are per process; set them once, outside any capture
  if (!attr_done && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 0, NST, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 1, NST, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 2, NST, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      veex_set_error("gather_gemm(x3): cannot raise the dynamic LDS limit"); return -3;
    }
  }
  attr_done = true;
  const dim3 blk(64 * WM * WN);
#ifdef VEEX_DEBUG_TIMING
  if constexpr (OPT == 1) if (const int dbg = veex_tune_int("VEEX_X3_DBG", 0)) {      // ablations (AFF-less instantiation whatever the launch asks)
    static bool dbg_attr = false;
#define X3_DBG_CASE(D) case D: { auto kfn = gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 0, NST, OPT, D>; \
      if (!dbg_attr && lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL(kfn, grid, blk, lds, st, p); break; }
    switch (dbg) { X3_DBG_CASE(1) X3_DBG_CASE(2) X3_DBG_CASE(4) X3_DBG_CASE(8) X3_DBG_CASE(16) X3_DBG_CASE(3) X3_DBG_CASE(7) X3_DBG_CASE(24)
      default: veex_set_error("VEEX_X3_DBG: 1, 2, 3, 4, 7, 8, 16 or 24"); return -1; }
#undef X3_DBG_CASE
    VEEX_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (aff == 0) hipLaunchKernelGGL((gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 0, NST, OPT>), grid, blk, lds, st, p);
  else if (aff == 1) hipLaunchKernelGGL((gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 1, NST, OPT>), grid, blk, lds, st, p);
  else hipLaunchKernelGGL((gather_gemm_x3_kernel<BM, BN, BK, WM, WN, MODE1, 2, NST, OPT>), grid, blk, lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, int BK, int WM, int WN, bool MODE1>
static int x3_launch(const GGParams& p, dim3 grid, hipStream_t st) {
  // register stages: 2, 3 and 4 chunks of loads in flight measure the same on every layer (profiles/r03_x3_sweep.txt)
#ifdef VEEX_DEBUG_TIMING
  switch (veex_tune_int("VEEX_X3_OPT", 1)) {      // (A/B of the kernel's OPT bits)
    case 0: return x3_launch_n<BM, BN, BK, WM, WN, MODE1, 2, 0>(p, grid, st);
    case 2: return x3_launch_n<BM, BN, BK, WM, WN, MODE1, 2, 2>(p, grid, st);
    case 3: return x3_launch_n<BM, BN, BK, WM, WN, MODE1, 2, 3>(p, grid, st);
    default: break;
  }
#endif
  // swizzled rows: 1-3 % on most layers and an extra workgroup per CU for the 16-deep tiles; converting and storing the
  // next chunk before the step's MFMAs (bit 1) measured 2-8 % slower (profiles/r03_x3_sweep_swizzled.txt)
  return x3_launch_n<BM, BN, BK, WM, WN, MODE1, 2, 1>(p, grid, st);
}

// Tile configuration: 0 = 64 x 64 x 32 (4 waves), 2 = 128 x 128 x 32 (8 waves, one workgroup per CU), 3 = 256 x 128 x 16
// (8 waves), 4 = 256 x 64 x 32 (8 waves), 5 = 128 x 128 x 16 (4 waves, three workgroups per CU), 6 = 128 x 64 x 16 (4 waves,
// four per CU), 7 = 128 x 64 x 32 (4 waves, two per CU), 8 = 256 x 64 x 16 (8 waves, two per CU); -1 = the launch stays on
// the fp32-MFMA kernel.  Measured per layer at batch 512 with the swizzled LDS rows (tools/x3_try.py,
// profiles/r03_x3_sweep_swizzled.txt; the padded-row sweep is profiles/r03_x3_sweep.txt); geometries outside the table
// take the size rule below.
struct X3Known { int mode, stride, Hin, Cin, Cout, cfg; };
static const X3Known X3_KNOWN[] = {
    {0, 2, 20, 64, 128, 5},    // enc.conv2 forward            53 us (fp32 MFMA kernel: 80)
    {1, 2, 10, 128, 64, 6},    // enc.conv2 input gradient     60 (73)
    {0, 2, 10, 128, 256, 2},   // enc.conv3 forward            48 (70)
    {1, 2, 5, 256, 128, 6},    // enc.conv3 input gradient     55 (64)
    {1, 2, 4, 256, 256, 2},    // dec.deconv1 forward          59 (74)
    {0, 2, 8, 256, 256, 7},    // dec.deconv1 input gradient   62 (88)
    {1, 1, 8, 256, 128, 2},    // dec.deconv2 forward         100 (137)
    {0, 1, 8, 128, 256, 5},    // dec.deconv2 input gradient  102 (136); tile 3 is 99 on its own, 5 is 0.8 % better on the step
                               //   (49 KB of LDS beside the weight gradients of the side stream; tools/x3_step_tiles.sh)
    {1, 2, 8, 128, 128, 5},    // dec.deconv3 forward          71 (91)
    {0, 2, 17, 128, 128, 2},   // dec.deconv3 input gradient   57 (81)
    {1, 1, 17, 128, 64, 8},    // dec.deconv4 forward         146 (180)
    {0, 1, 19, 64, 128, 5},    // dec.deconv4 input gradient  125 (183)
    {1, 2, 19, 64, 64, 6},     // dec.deconv5 forward         105-112 (138)
    {0, 2, 39, 64, 64, 7},     // dec.deconv5 input gradient  104 (127)
};
static int x3_bm_of(int cfg) { return cfg == 0 ? 64 : (cfg == 2 || cfg == 5 || cfg == 6 || cfg == 7) ? 128 : 256; }
static int x3_bn_of(int cfg) { return (cfg == 0 || cfg == 4 || cfg == 6 || cfg == 7 || cfg == 8) ? 64 : 128; }
int veex_gg_x3_cfg(const GGParams& p) {
  const long long M = (long long)p.Nb * p.Hout * p.Wout;
  int cfg = -2;
  const int force = veex_tune_int("VEEX_X3_CFG", -2);
  if (force >= -1 && !(p.Cout <= 64 && force >= 0 && x3_bn_of(force) == 128)) cfg = force;
  if (cfg == -2 && p.kh == 3 && p.kw == 3)
    for (const X3Known& e : X3_KNOWN)
      if (e.mode == p.mode && e.stride == p.stride && e.Hin == p.Hin && e.Cin == p.Cin && e.Cout == p.Cout) {
        cfg = e.cfg;
#ifdef VEEX_DEBUG_TIMING
        char name[24];      // VEEX_X3_T<row of the table>: one layer's tile inside the whole step
        snprintf(name, sizeof(name), "VEEX_X3_T%d", (int)(&e - X3_KNOWN));
        cfg = veex_tune_int(name, cfg);
#endif
      }
  if (cfg == -1) return -1;
  // the table's tile (or the largest) only while it still gives every CU a workgroup; else smaller tiles
  auto blocks = [&](int c) { return ((M + x3_bm_of(c) - 1) / x3_bm_of(c)) * ((p.Cout + x3_bn_of(c) - 1) / x3_bn_of(c)); };
  // (a table entry was measured with the grid it has at batch 512 -- enc.conv3 forward runs 200 workgroups of 128 x 128
  //  faster than 800 of 64 x 64 -- and is kept down to half of that; the size rule wants every CU to have a workgroup)
  const long long floor_blocks = (cfg == -2) ? 256 : 96;
  if (cfg == -2) cfg = (p.Cout > 64) ? 2 : 4;
  if (force < 0 && blocks(cfg) < floor_blocks) cfg = (p.Cout > 64 && blocks(5) >= 256) ? 5 : 0;
  return cfg;
}

// tile rows of the split kernel for this launch (veex_gg_bm is the one source: callers size the column partials by it)
int veex_gg_x3_bm(const GGParams& p) { return x3_bm_of(veex_gg_x3_cfg(p)); }

// p: validated by veex_launch_gather_gemm, phase tables filled for veex_gg_x3_bm(p)-row tiles, p.Wbf = three planes
int veex_launch_gather_gemm_x3(GGParams& p, int mtiles, hipStream_t st) {
  if (p.Cin % 32 != 0) { veex_set_error("gather_gemm(x3): Cin %% 32 == 0 required (got %d)", p.Cin); return -1; }
  if ((long long)p.Nb * p.Hout * p.Wout >= (1LL << 24)) { veex_set_error("gather_gemm(x3): more than 2^24 output pixels"); return -1; }
  if ((long long)p.Nb * p.Hout * p.Wout * p.Cout * 4 >= 0x7FFF0000LL) { veex_set_error("gather_gemm(x3): output of 2 GiB or more"); return -1; }
  if (3LL * p.kh * p.kw * p.Cin * p.Cout * 2 >= 0x7FFF0000LL) { veex_set_error("gather_gemm(x3): weights of 2 GiB or more"); return -1; }
  const int cfg = veex_gg_x3_cfg(p);
  if (cfg < 0) { veex_set_error("gather_gemm(x3): launch not taken"); return -1; }
  const int bn = x3_bn_of(cfg), bk = (cfg == 3 || cfg == 5 || cfg == 6 || cfg == 8) ? 16 : 32;
  const int ntiles = (p.Cout + bn - 1) / bn;
  if (p.nsplit > 1) {
    const int nchunks = p.kh * p.kw * (p.Cin / bk);
    p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
  }
  p.remap = (p.nphase == 1);
  p.gx = ntiles; p.gy = mtiles; p.gz = p.nsplit;
  dim3 grid((unsigned)ntiles * mtiles * p.nsplit, 1, 1);
  const bool m1 = p.mode == 1;
  switch (cfg) {
    case 2: return m1 ? x3_launch<128, 128, 32, 2, 4, true>(p, grid, st) : x3_launch<128, 128, 32, 2, 4, false>(p, grid, st);
    case 3: return m1 ? x3_launch<256, 128, 16, 4, 2, true>(p, grid, st) : x3_launch<256, 128, 16, 4, 2, false>(p, grid, st);
    case 4: return m1 ? x3_launch<256, 64, 32, 4, 2, true>(p, grid, st) : x3_launch<256, 64, 32, 4, 2, false>(p, grid, st);
    case 5: return m1 ? x3_launch<128, 128, 16, 2, 2, true>(p, grid, st) : x3_launch<128, 128, 16, 2, 2, false>(p, grid, st);
    case 6: return m1 ? x3_launch<128, 64, 16, 2, 2, true>(p, grid, st) : x3_launch<128, 64, 16, 2, 2, false>(p, grid, st);
    case 7: return m1 ? x3_launch<128, 64, 32, 2, 2, true>(p, grid, st) : x3_launch<128, 64, 32, 2, 2, false>(p, grid, st);
    case 8: return m1 ? x3_launch<256, 64, 16, 4, 2, true>(p, grid, st) : x3_launch<256, 64, 16, 4, 2, false>(p, grid, st);
    default: return m1 ? x3_launch<64, 64, 32, 2, 2, true>(p, grid, st) : x3_launch<64, 64, 32, 2, 2, false>(p, grid, st);
  }
}
