// bf16 variant of the implicit-GEMM engine (gather_gemm.hip): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Same gather addressing, phase decomposition, fused producer-BatchNorm prologue and epilogues as the
// fp32 engine (reference ops: Conv2D / Conv2DTranspose forward, models.py:219-298, and their input
// gradients).  Differences:
//   * activations stay fp32 in HBM (written once by the fp32 epilogues); they are rounded to bf16
//     (RNE, v_cvt_pk_bf16_f32) while a tile is staged into LDS, after the producer's BN affine;
//   * weights come from a packed bf16 copy [tap][Cout][Cin] (K contiguous) that
//     veex_pack_weights_bf16 makes from the fp32 master weights once per step, so the B tile is a
//     plain 16-byte copy and both operands are read from LDS with one ds_read_b128 per 8 k values;
//   * K chunk = 32 channels of one tap (Cin % 32 == 0 is required).
// LDS rows are 32 bf16 + 8 pad = 80 B: the b128 operand reads of a 16-lane group fall on 16 distinct
// 4-bank groups (20r mod 64), i.e. conflict-free.
#include "gg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 bload4u(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4u f = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(f[0], f[1], f[2], f[3]);
}

// q = r / d, rem = r % d for 0 <= r < 2^24, 1 <= d < 2^24 (float reciprocal + one correction step: a
// fraction of the ~45 instructions of a 32-bit integer division, which matter here because a bf16 tile's
// whole K loop is only a few thousand cycles)
__device__ __forceinline__ void fdivmod24(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

// ABF: A comes from a bf16 consumer copy (p.Abf, BatchNorm already applied by its producer's epilogue):
// the A tile is staged with plain 16-byte copies -- no affine, no conversion (AFF must be 0).
template <int BM, int BN, int WM, int WN, bool MODE1, int AFF, bool ABF, int BK = 32>
__global__ __launch_bounds__(256) void gather_gemm_bf16_kernel(const GGParams p) {
  constexpr int LDKH = BK + 8;         // bf16 elements per LDS row
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int AQ = ABF ? BK / 8 : BK / 4;   // 16-byte pieces per A row: 8 bf16 or 4 fp32 each
  constexpr int ARP = 256 / AQ;        // A rows staged per pass
  constexpr int APASS = BM / ARP;
  constexpr int AEB = ABF ? 2 : 4;     // bytes per A element in HBM
  static_assert(!ABF || AFF == 0, "a bf16 consumer copy already carries the BatchNorm");
  constexpr int BQ = BK / 8;           // B: 16-byte pieces per row
  constexpr int BRP = 256 / BQ;        // B rows staged per pass
  constexpr int BPASS = BN / BRP;
  constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(APASS >= 1 && BPASS >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Bs = As + 2 * A_ELEMS;
  int* s_off = reinterpret_cast<int*>(Bs + 2 * B_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bnt, btile, kz;
  veex_block_coords(p.remap, p.gx, p.gy, bnt, btile, kz);
  bnt = __builtin_amdgcn_readfirstlane(bnt);     // wave-uniform: everything derived from them stays in SGPRs
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
  const float invHWp = 1.0f / (float)HWp, invWp = 1.0f / (float)Wp;   // rows < 2^24 (checked by the launcher)

  if (tid < BM) {
    int m = m0 + tid, off = -1;
    if (m < Mph) {
      int n, rem, j, i;
      fdivmod24(m, HWp, invHWp, n, rem);
      fdivmod24(rem, Wp, invWp, j, i);
      off = (n * p.Hout + ostep * j + oy0) * p.Wout + ostep * i + ox0;
    }
    s_off[tid] = (off < 0) ? (int)0x80000000 : off * p.Cout * 4;   // byte offset of the fp32 output row (gg_epilogue_b)
  }

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(ABF ? p.Abf : static_cast<const void*>(p.A)), 0,
      (int)((unsigned)p.Nb * p.Hin * p.Win * p.Cin * (unsigned)AEB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.Wbf), 0, (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 2u), 0x00020000);
  const unsigned aff_bytes = (AFF != 0) ? (unsigned)(p.aff_mask + 1) * 4u : 0u;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_scale : p.A), 0, (int)aff_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AFF != 0 ? p.a_shift : p.A), 0, (int)aff_bytes, 0x00020000);

  // ---- A loader: rows (tid/AQ)+ARP*i, 16-byte piece (tid%AQ) of the 32-channel chunk ----
  const int q = tid % AQ, lr = tid / AQ;
  int a_rowbase[APASS], a_nbase[APASS];
  unsigned a_mask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + lr + ARP * i;
    a_rowbase[i] = 0; a_mask[i] = 0u; a_nbase[i] = 0;
    if (m < Mph) {
      int n, rem, j, ii;
      fdivmod24(m, HWp, invHWp, n, rem);
      fdivmod24(rem, Wp, invWp, j, ii);
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      int by, bx;   // input pixel read by phase tap (0,0)
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      a_nbase[i] = n * p.Hin * p.Win * p.Cin * AEB;
      a_rowbase[i] = a_nbase[i] + ((by * p.Win + bx) * p.Cin) * AEB + 16 * q;
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
  // ---- B loader: rows (tid>>2)+64*i of the packed weights, 16-byte piece (tid&3) = 8 k values ----
  const int bq = tid % BQ, br = tid / BQ;
  int b_rowbase[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int n = n0 + br + BRP * i;
    b_ok[i] = n < p.Cout;
    b_rowbase[i] = (n * p.Cin + 8 * bq) * 2;
  }
  const int tap_bytes = p.Cin * p.Cout * 2;

  struct Stage {
    float4 ra[APASS], rsc[AFF == 2 ? APASS : 1], rsh[AFF == 2 ? APASS : 1];
    uint4 rb[BPASS];
    bool rok[APASS];
  };

  // running (channel slice, tap) of the next chunk to load: chunks are loaded strictly in order, channel slice outer /
  // tap inner.  (Recomputing them per chunk cost three scalar integer divisions = ~55 SALU instructions and, with the
  // timing switches that used to sit in this loop, ~10 branches per 4-8 MFMAs: the loop was bound by instruction
  // issue, not by the MFMAs or the loads.)
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
    const int delta = ((MODE1 ? -dpix : dpix) * p.Cin + ci0) * AEB;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = cvalid && ((a_mask[i] >> tapi) & 1u);
      const unsigned aoff = ok ? (unsigned)(a_rowbase[i] + delta) : VEEX_OOB;
      st.ra[i] = bload4(rA, aoff);
      st.rok[i] = ok;
      if (AFF == 2) {
        const unsigned ao = ok ? (unsigned)(a_rowbase[i] + delta - a_nbase[i]) : VEEX_OOB;
        st.rsc[i] = bload4(rS, ao);
        st.rsh[i] = bload4(rH, ao);
      }
    }
    if (AFF == 1) {
      const unsigned ao = cvalid ? (unsigned)(((ci0 + 4 * q) & p.aff_mask) * 4) : VEEX_OOB;
      st.rsc[0] = bload4(rS, ao);
      st.rsh[0] = bload4(rH, ao);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const bool ok = cvalid && b_ok[i];
      st.rb[i] = bload4u(rW, ok ? (unsigned)(tap * tap_bytes + b_rowbase[i] + ci0 * 2) : VEEX_OOB);
    }
  };

  auto store_chunk = [&](const Stage& st, int buf) {
    __bf16* as = As + buf * A_ELEMS;
    __bf16* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      if constexpr (ABF) {   // 8 bf16 as loaded (out-of-range pieces arrive as zeros)
        *reinterpret_cast<float4*>(as + (lr + ARP * i) * LDKH + 8 * q) = st.ra[i];
        continue;
      }
      float4 v = st.ra[i];
      if (AFF != 0) {
        const float4 sc = st.rsc[AFF == 2 ? i : 0], sh = st.rsh[AFF == 2 ? i : 0];
        v.x = st.rok[i] ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = st.rok[i] ? fmaf(v.y, sc.y, sh.y) : 0.f;
        v.z = st.rok[i] ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = st.rok[i] ? fmaf(v.w, sc.w, sh.w) : 0.f;
      }
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(as + (lr + ARP * i) * LDKH + 4 * q) = h;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      *reinterpret_cast<uint4*>(bs + (br + BRP * i) * LDKH + 8 * bq) = st.rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // lane l: A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7 (one ds_read_b128 each)
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

  // two chunks of global loads in flight (see gather_gemm_kernel); one barrier per chunk
  Stage s0, s1;
  load_chunk(s0, c_begin < c_end);
  store_chunk(s0, 0);
  load_chunk(s0, c_begin + 1 < c_end);
  __syncthreads();
  for (int c = c_begin; c < c_end; c += 2) {
    load_chunk(s1, c + 2 < c_end);
    compute(0);
    store_chunk(s0, 1);
    __syncthreads();
    load_chunk(s0, c + 3 < c_end);
    compute(1);
    store_chunk(s1, 0);
    __syncthreads();
  }
  gg_epilogue_b<BM, BN, WM, WN>(p, acc, reinterpret_cast<const unsigned*>(s_off), reinterpret_cast<float*>(As), btile, kz, n0);
}

// ---------------------------------------------------------------------------
// packed bf16 weights
// ---------------------------------------------------------------------------
// dst[tap][c][r] = bf16(src[tap][r][c]) (transpose != 0) or dst[tap][r][c] = bf16(src[tap][r][c])
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                int taps, int R, int C, int transpose) {
  const long long n = (long long)taps * R * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (!transpose) { dst[i] = (__bf16)src[i]; continue; }
    const long long per = (long long)R * C;
    const long long t = i / per, rem = i - t * per;
    const int c = (int)(rem / R), r = (int)(rem - (long long)c * R);     // dst index = (t, c, r)
    dst[i] = (__bf16)src[t * per + (long long)r * C + c];
  }
}

// all packed copies of one step in ONE launch (14 tiny launches cost ~4.4 us each on the critical path)
struct PackJob { const float* src; __bf16* dst; int taps, R, C, transpose, block0, blocks; };
struct PackJobs { int n; PackJob j[16]; };
__global__ __launch_bounds__(256) void pack_weights_bf16_multi_kernel(const PackJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const PackJob J = jobs.j[k];
  const long long n = (long long)J.taps * J.R * J.C, per = (long long)J.R * J.C;
  for (long long i = (long long)(blockIdx.x - J.block0) * 256 + threadIdx.x; i < n; i += (long long)J.blocks * 256) {
    if (!J.transpose) { J.dst[i] = (__bf16)J.src[i]; continue; }
    const long long t = i / per, rem = i - t * per;
    const int c = (int)(rem / J.R), r = (int)(rem - (long long)c * J.R);
    J.dst[i] = (__bf16)J.src[t * per + (long long)r * J.C + c];
  }
}
int veex_pack_weights_bf16_multi(int n, const float* const* src, void* const* dst, const int* taps, const int* rows,
                                 const int* cols, const int* transpose, hipStream_t st) {
  if (n < 1 || n > 16) { veex_set_error("pack_weights_bf16_multi: 1..16 jobs"); return -1; }
  PackJobs jobs; jobs.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const long long e = (long long)taps[i] * rows[i] * cols[i];
    int b = (int)((e + 256 * 8 - 1) / (256 * 8));
    if (b < 1) b = 1;
    if (b > 256) b = 256;
    jobs.j[i].src = src[i]; jobs.j[i].dst = reinterpret_cast<__bf16*>(dst[i]);
    jobs.j[i].taps = taps[i]; jobs.j[i].R = rows[i]; jobs.j[i].C = cols[i]; jobs.j[i].transpose = transpose[i];
    jobs.j[i].block0 = total; jobs.j[i].blocks = b;
    total += b;
  }
  hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3(total), dim3(256), 0, st, jobs);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// fp32 transposed copies for the lean fp32 kernel's mode-0 launches (GGParams::Wt): dst[t][c][r] = src[t][r][c]
struct TrJob { const float* src; float* dst; int taps, R, C, block0, blocks; };
struct TrJobs { int n; TrJob j[16]; };
__global__ __launch_bounds__(256) void transpose_weights_f32_multi_kernel(const TrJobs jobs) {
  __shared__ float tile[32][33];
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const TrJob J = jobs.j[k];
  // one 32x32 tile of one tap per block step: coalesced 128-byte reads along C and writes along R
  const int tr = (J.R + 31) / 32, tc = (J.C + 31) / 32, per_tap = tr * tc, total = J.taps * per_tap;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
  for (int t = blockIdx.x - J.block0; t < total; t += J.blocks) {
    const int tap = t / per_tap, rem = t - tap * per_tap, br = rem / tc, bc = rem - br * tc;
    const float* s = J.src + (long long)tap * J.R * J.C;
    float* d = J.dst + (long long)tap * J.R * J.C;
#pragma unroll
    for (int y = ly; y < 32; y += 8) {
      const int r = br * 32 + y, c = bc * 32 + lx;
      tile[y][lx] = (r < J.R && c < J.C) ? s[(long long)r * J.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int y = ly; y < 32; y += 8) {
      const int c = bc * 32 + y, r = br * 32 + lx;
      if (r < J.R && c < J.C) d[(long long)c * J.R + r] = tile[lx][y];
    }
    __syncthreads();
  }
}
int veex_transpose_weights_f32_multi(int n, const float* const* src, float* const* dst, const int* taps, const int* rows,
                                     const int* cols, hipStream_t st) {
  if (n < 1 || n > 16) { veex_set_error("transpose_weights_f32_multi: 1..16 jobs"); return -1; }
  TrJobs jobs; jobs.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const int tiles = taps[i] * ((rows[i] + 31) / 32) * ((cols[i] + 31) / 32);
    int b = (tiles + 1) / 2;
    if (b < 1) b = 1;
    if (b > 144) b = 144;
    jobs.j[i].src = src[i]; jobs.j[i].dst = dst[i]; jobs.j[i].taps = taps[i]; jobs.j[i].R = rows[i]; jobs.j[i].C = cols[i];
    jobs.j[i].block0 = total; jobs.j[i].blocks = b;
    total += b;
  }
  hipLaunchKernelGGL(transpose_weights_f32_multi_kernel, dim3(total), dim3(256), 0, st, jobs);
  VEEX_CHECK_LAUNCH();
  return 0;
}

extern "C" int veex_transpose_weights_f32(const float* src, float* dst, int taps, int rows, int cols, hipStream_t st) {
  if (src == nullptr || dst == nullptr || taps < 1 || rows < 1 || cols < 1) { veex_set_error("transpose_weights_f32: bad arguments"); return -1; }
  return veex_transpose_weights_f32_multi(1, &src, &dst, &taps, &rows, &cols, st);
}

extern "C" int veex_pack_weights_bf16(const float* src, void* dst, int taps, int rows, int cols, int transpose,
                                      hipStream_t st) {
  if (src == nullptr || dst == nullptr || taps < 1 || rows < 1 || cols < 1) { veex_set_error("pack_weights_bf16: bad arguments"); return -1; }
  const long long n = (long long)taps * rows * cols;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, st, src, reinterpret_cast<__bf16*>(dst), taps, rows, cols, transpose);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int BM, int BN, int BK = 32>
static constexpr size_t ggb_lds_bytes() { return (2 * BM * (BK + 8) + 2 * BN * (BK + 8)) * 2 + BM * sizeof(int); }

template <int BM, int BN, bool MODE1>
static int ggb_launch(const GGParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = ggb_lds_bytes<BM, BN>();
  static_assert(lds <= 64 * 1024, "fits the default dynamic LDS limit");
  const int aff = (p.a_scale == nullptr) ? 0 : (p.aff_spatial ? 2 : 1);
  if (p.Abf != nullptr) {
    // K chunks of 64 channels (half the barriers and loop overhead, twice the LDS, so fewer resident blocks):
    // measured per layer at B=512 it wins where the K loop per tap is long (Cin >= 128) and the launch is at most
    // ~4 blocks per CU, and loses on the large grids.  VEEX_BFBK=32/64 forces one of them.
    const int bfbk = veex_tune_int("VEEX_BFBK", 0);
    const bool auto64 = p.Cin >= 128 && (long long)grid.x <= 1024;
    if (p.Cin % 64 == 0 && (bfbk == 64 || (bfbk == 0 && auto64))) {
      constexpr size_t lds64 = ggb_lds_bytes<BM, BN, 64>();
      static bool attr_done = false;
      if (!attr_done && lds64 > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 0, true, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64) != hipSuccess) {
          veex_set_error("gather_gemm(bf16): cannot raise the dynamic LDS limit"); return -3;
        }
      }
      attr_done = true;
      hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 0, true, 64>), grid, dim3(256), lds64, st, p);
    } else {
      hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 0, true>), grid, dim3(256), lds, st, p);
    }
  }
  else if (aff == 0) hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 0, false>), grid, dim3(256), lds, st, p);
  else if (aff == 1) hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 1, false>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM, BN, 2, 2, MODE1, 2, false>), grid, dim3(256), lds, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// p: validated by veex_launch_gather_gemm, phase tables filled (veex_gg_m_tiles), p.Wbf != nullptr
int veex_launch_gather_gemm_bf16(GGParams& p, int mtiles, hipStream_t st) {
  if (p.Cin % 32 != 0) { veex_set_error("gather_gemm(bf16): Cin %% 32 == 0 required (got %d)", p.Cin); return -1; }
  if ((long long)p.Nb * p.Hout * p.Wout >= (1LL << 24)) { veex_set_error("gather_gemm(bf16): more than 2^24 output pixels"); return -1; }
  if ((long long)p.Nb * p.Hout * p.Wout * p.Cout * 4 >= 0x7FFF0000LL) { veex_set_error("gather_gemm(bf16): output of 2 GiB or more"); return -1; }
  const int bm = veex_gg_bm(p);
  const int bn = (bm == 128 && p.Cout > 64) ? 128 : 64;
  const int ntiles = (p.Cout + bn - 1) / bn;
  if (p.nsplit > 1) {
    const int nchunks = p.kh * p.kw * (p.Cin / 32);
    p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
  }
  const int xr = veex_tune_int("VEEX_XCD", -1);
  p.remap = (xr < 0) ? (p.nphase == 1) : xr;
  p.gx = ntiles; p.gy = mtiles; p.gz = p.nsplit;
  dim3 grid((unsigned)ntiles * mtiles * p.nsplit, 1, 1);
  if (bm == 64) return (p.mode == 1) ? ggb_launch<64, 64, true>(p, grid, st) : ggb_launch<64, 64, false>(p, grid, st);
  if (bn == 128) return (p.mode == 1) ? ggb_launch<128, 128, true>(p, grid, st) : ggb_launch<128, 128, false>(p, grid, st);
  return (p.mode == 1) ? ggb_launch<128, 64, true>(p, grid, st) : ggb_launch<128, 64, false>(p, grid, st);
}
