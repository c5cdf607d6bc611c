// Small-launch instance of the fp32 implicit-GEMM engine: 32x32 tiles, the K loop of a tile split over the four waves
// of its block.
//
// Why: with few rows (small batches -- the reference trains at 32 .. 128 -- and serving: predict / encode / decode /
// attribute on a handful of maps) a launch of the 64x64 kernel has a few hundred tiles for 2 048 block slots, and
// each tile walks its whole K loop (72 .. 144 chunks of 8 dependent MFMAs, one barrier each) alone: the launch lasts
// as long as one tile's chain, 30 .. 40 us, with the chip almost idle.  Here a tile is 32 rows x 32
// columns, and the four waves of the block each take every fourth K chunk of it:
//   * 4x the tiles and 4x shorter dependent chains per launch;
//   * operands go from L2 straight into the MFMA fragment layout (lane = row / column, 8 consecutive k per lane half:
//     two 16-byte loads per operand and chunk): no LDS staging and NO barrier in the K loop, two chunks in flight;
//   * the four partial accumulators are summed through LDS in a fixed order (deterministic), and wave 0 runs the
//     same fused epilogues as the 64x64 kernel (bias / ReLU / BatchNorm-applied copy; BatchNorm / ReLU backward with
//     per-tile column sums).
// Same address scheme as gather_gemm_v3.hip: constant per-row offset, tap / channel-slice displacement in the scalar
// offset, validity ORed in (out of range -> zeros), taps that are padding for the whole tile skipped.
// Needs the weights as [tap][Cout][Cin] rows (mode 1 natively, mode 0 through GGParams::Wt), no affine prologue
// (producers write BatchNorm-applied copies), a fused (non-split) epilogue.  The K order differs from the 64x64 kernel
// (tap-major, four interleaved partial sums), so results agree with it to fp32 rounding, not bit for bit.
#include "gg_common.h"

__device__ __forceinline__ float4 v4_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void v4_divmod(int r, int d, float inv, int& qo, int& rem) {   // 0 <= r < 2^24, 1 <= d < 2^24
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

template <bool MODE1>
__global__ __launch_bounds__(256) void gather_gemm_v4_kernel(const GGParams p) {
  constexpr int BM = 32, BN = 32, BK = 16;
  __shared__ float s_acc[3 * 16 * 64];   // accumulators of waves 1..3
  __shared__ unsigned s_offb[BM];        // output row byte offsets for the epilogue

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, l31 = lane & 31;
  const int btile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)p.gx));
  const int bnt = __builtin_amdgcn_readfirstlane((int)(blockIdx.x - (unsigned)btile * (unsigned)p.gx));
  int ph = 0;
  if (p.nphase == 4) {
    while (ph < 3 && btile >= p.tile_prefix[ph + 1]) ++ph;
  }
  const int py = ph >> 1, px = ph & 1;
  const int Hp = p.Hp[py], Wp = p.Wp[px];
  const int ostep = (p.nphase == 4) ? 2 : 1;
  const int oy0 = p.o0[py], ox0 = p.o0[px];
  const int Mph = p.Nb * Hp * Wp;
  const int m0 = (btile - p.tile_prefix[ph]) * BM;
  const int n0 = bnt * BN;
  int kys = 0, kxs = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (p.nphase == 4) {
    kys = py; kxs = px; kstep = 2;
    nky = (p.kh - py + 1) >> 1;
    nkx = (p.kw - px + 1) >> 1;
  }
  const int cpt = p.Cin / BK;
  const int ntaps = nky * nkx;

  // as in gather_gemm_v3.hip: per-row offsets are taken at the tap with the lowest address, shifted up by `bias` bytes
  const int bias_pix = MODE1 ? ((nky - 1) * p.Win + (nkx - 1)) : p.pb * (p.Win + 1);
  const int bias_bytes = bias_pix * p.Cin * 4;
  const unsigned a_total = (unsigned)p.Nb * p.Hin * p.Win * p.Cin * 4u;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)) - bias_bytes, 0, (int)(a_total + (unsigned)bias_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(MODE1 ? p.W : p.Wt), 0, (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 4u), 0x00020000);

  // ---- A fragment: lane = row m0 + l31, k = 8 * lh .. 8 * lh + 7 of the chunk ----
  unsigned a_voff = 0u, a_inv = 0xFFFFFFFFu;     // a_inv bit t set <=> tap t must read zeros
  {
    const int m = m0 + l31;
    unsigned out_off = 0x80000000u;              // + any column offset stays out of range: the store is dropped
    if (m < Mph) {
      int n, rem, j, ii;
      v4_divmod(m, Hp * Wp, 1.0f / (float)(Hp * Wp), n, rem);
      v4_divmod(rem, Wp, 1.0f / (float)Wp, j, ii);
      const int oy = ostep * j + oy0, ox = ostep * ii + ox0;
      out_off = (unsigned)(((n * p.Hout + oy) * p.Wout + ox) * p.Cout) * 4u;
      int by, bx;
      if (MODE1) { by = (oy + p.pb - kys) >> (p.stride - 1); bx = (ox + p.pb - kxs) >> (p.stride - 1); }
      else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
      const int nbase = n * p.Hin * p.Win * p.Cin * 4;
      const int inner = ((MODE1 ? (by * p.Win + bx) : ((by + p.pb) * p.Win + bx + p.pb)) * p.Cin + 8 * lh) * 4;
      a_voff = (unsigned)(nbase + inner);
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
      a_inv = ~mk;
    }
    if (wave == 0 && lh == 0) s_offb[l31] = out_off;
  }
  const unsigned a_voff1 = a_voff + 16u;          // second 16-byte piece (kept apart: the mask is ORed in afterwards)
  // valid-tap set of the tile: OR over its rows (both lane halves hold the same rows)
  unsigned any = ~a_inv;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) any |= (unsigned)__shfl_xor((int)any, o);
  const unsigned all_taps = (ntaps >= 32) ? 0xFFFFFFFFu : ((1u << ntaps) - 1u);
  const unsigned tile_mk = (unsigned)__builtin_amdgcn_readfirstlane((int)(any & all_taps));
  const int nchunks = __builtin_popcount(tile_mk) * cpt;

  // ---- B fragment: lane = column n0 + l31 of the [tap][Cout][Cin] weights, the same k as A ----
  const int col = n0 + l31;
  const bool colok = col < p.Cout;
  const unsigned b_voff = colok ? (unsigned)((col * p.Cin + 8 * lh) * 4) : VEEX_OOB;
  const unsigned b_voff1 = colok ? b_voff + 16u : VEEX_OOB;
  const int tap_bytes = p.Cin * p.Cout * 4;

  // this wave's chunks: wave, wave + 4, ... in tap-major order (chunk c = valid tap c / cpt, channel slice c % cpt)
  // (advanced with scalar selects, not branches: a branch between loads makes the compiler wait for all of them)
  int it_cc = wave;
  unsigned it_bits = tile_mk;
  auto normalise = [&]() {           // it_cc < cpt + 4 on entry: at most four taps further (cpt >= 1)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool adv = it_cc >= cpt;
      it_cc = adv ? it_cc - cpt : it_cc;
      it_bits = adv ? (it_bits & (it_bits - 1u)) : it_bits;
    }
  };
  normalise();
  const int my_chunks = (nchunks > wave) ? (nchunks - wave + 3) >> 2 : 0;
  const int inv_nkx = __builtin_amdgcn_readfirstlane(65536 / nkx + 1);   // tapi / nkx for tapi < 32

  struct Stage { float4 a0, a1, b0, b1; };
  auto load_chunk = [&](Stage& st) {
    const bool valid = it_bits != 0u;
    const int tapi = valid ? __builtin_ctz(it_bits) : 31;                 // bit 31 of a_inv is always set: zeros
    const int ia = (tapi * inv_nkx) >> 16, ib = tapi - ia * nkx;
    const int tap = (kys + kstep * ia) * p.kw + (kxs + kstep * ib);
    const int dpix = MODE1 ? ((nky - 1 - ia) * p.Win + (nkx - 1 - ib)) : (ia * p.Win + ib);
    const int delta = valid ? dpix * p.Cin * 4 + it_cc * (BK * 4) : 0;
    const int wsoff = valid ? tap * tap_bytes + it_cc * (BK * 4) : 0x7FFFFFFF;   // past the end: zeros on both sides
    it_cc += 4;
    normalise();
    const unsigned m = (unsigned)__builtin_amdgcn_sbfe(a_inv, tapi, 1);   // 0 or 0xFFFFFFFF
    st.a0 = v4_bload(rA, a_voff | m, delta);
    st.a1 = v4_bload(rA, a_voff1 | m, delta);
    st.b0 = v4_bload(rW, b_voff, wsoff);
    st.b1 = v4_bload(rW, b_voff1, wsoff);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto compute = [&](const Stage& st) {
    const float av[8] = {st.a0.x, st.a0.y, st.a0.z, st.a0.w, st.a1.x, st.a1.y, st.a1.z, st.a1.w};
    const float bv[8] = {st.b0.x, st.b0.y, st.b0.z, st.b0.w, st.b1.x, st.b1.y, st.b1.z, st.b1.w};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
  };

  // no LDS and no barrier inside: each wave streams its own fragments, two chunks in flight (an odd count multiplies
  // one all-zero chunk)
  Stage s0, s1;
  load_chunk(s0);
  for (int c = 0; c < my_chunks; c += 2) {
    load_chunk(s1);
    __builtin_amdgcn_sched_barrier(0);   // keep the next chunk's loads ABOVE this chunk's MFMAs (the scheduler sank them)
    compute(s0);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(s0);
    __builtin_amdgcn_sched_barrier(0);
    compute(s1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- the four partial sums, in wave order ----
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += s_acc[(w * 16 + r) * 64 + lane];
  }

  // ---- epilogue (wave 0): C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  if (!colok) return;
  const unsigned out_bytes = (unsigned)p.Nb * p.Hout * p.Wout * p.Cout * 4u;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.Out, 0, (int)out_bytes, 0x00020000);
  const unsigned col4 = (unsigned)col * 4u;
  unsigned rof[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint4 ro = *reinterpret_cast<const uint4*>(s_offb + 8 * g + 4 * lh);
    rof[4 * g] = ro.x; rof[4 * g + 1] = ro.y; rof[4 * g + 2] = ro.z; rof[4 * g + 3] = ro.w;
  }
  if (p.epi == VEEX_EPI_FWD) {
    const float bia = (p.bias != nullptr) ? p.bias[col] : 0.f;
    const bool has_bn = p.OutBn != nullptr;
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(has_bn ? p.OutBn : p.Out, 0, has_bn ? (int)out_bytes : 0, 0x00020000);
    const float osc = has_bn ? p.o_scale[col & p.o_cmask] : 1.f, osh = has_bn ? p.o_shift[col & p.o_cmask] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bia;
      if (p.relu) v = fmaxf(v, 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[r] + col4, 0, 0);
      if (has_bn) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(v, osc, osh)), rN, rof[r] + col4, 0, 0);
    }
  } else {   // VEEX_EPI_BWD: BatchNorm / ReLU backward of the producing layer + column sums of this tile
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.aux), 0, (int)out_bytes, 0x00020000);
    const float esc = (p.e_scale != nullptr) ? p.e_scale[col & p.epi_cmask] : 1.f;
    float a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)   // rows past M: out of range -> 0 (and their accumulators are 0)
      a[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, rof[r] + col4, 0, 0));
    float cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r];
      const float vm = (!p.relu || a[r] > 0.f) ? v : 0.f;
      cs1 += v;
      cs2 = fmaf(v, a[r], cs2);
      cs3 += vm;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vm * esc), rO, rof[r] + col4, 0, 0);
    }
    if (p.partials != nullptr) {
      cs1 += __shfl_xor(cs1, 32);
      cs2 += __shfl_xor(cs2, 32);
      cs3 += __shfl_xor(cs3, 32);
      if (lh == 0) {
        float* pp = p.partials + (long long)btile * 3 * p.Cout + col;
        pp[0] = cs1;
        pp[p.Cout] = cs2;
        pp[2 * p.Cout] = cs3;
      }
    }
  }
}

// 1: this launch is small enough for the 32x32 split-K kernel to win and fits its constraints (veex_gg_bm then is 32:
// callers size the column partials by it)
int veex_gg_v4_ok(const GGParams& p) {
  if (!veex_env_int("VEEX_GGV4", 1)) return 0;
  // (wplanes == 3: the split-fp32 engine's weight planes ride along on every descriptor; a launch its kernel does not
  //  take is an fp32 launch like any other)
  if ((p.Wbf != nullptr && p.wplanes != 3) || p.Abf != nullptr || p.OutBf != nullptr || p.a_scale != nullptr) return 0;
  if (p.epi == VEEX_EPI_RAW || p.nsplit > 1 || p.Out == nullptr) return 0;
  if (p.mode == 0 && p.Wt == nullptr) return 0;
  if (p.mode == 1 && p.stride != 1 && p.stride != 2) return 0;
  if (p.epi == VEEX_EPI_BWD && p.aux == nullptr) return 0;
  if (!veex_gg_v3_ok(p)) return 0;                       // Cin % 16, <= 31 taps, operands < 2 GiB, rows < 2^24
  // Threshold = blocks the 64x64 kernel would launch.  Whole train step, A/B on one box (profiles/r02_small_launch_ab.txt):
  //   batch 32: off 0.706 ms, < 64: 0.641, < 128: 0.559, < 256: 0.526, < 512: 0.530
  //   batch 64: off 0.770 ms,          < 128: 0.717, < 256: 0.700, < 512: 0.733
  //   batch 128: off 0.988 ms,         < 128: 0.992, < 256: 1.003, < 512: 1.025;   batch >= 256: no launch below 256
  const long long M = (long long)p.Nb * p.Hout * p.Wout;
  const long long blocks64 = ((M + 63) / 64) * ((p.Cout + 63) / 64);
  return blocks64 < veex_tune_int("VEEX_V4_BLOCKS", 256);
}

// p: validated by veex_launch_gather_gemm, phase tables filled for 32-row tiles
int veex_launch_gather_gemm_v4(GGParams& p, int mtiles, hipStream_t st) {
  const int ntiles = (p.Cout + 31) / 32;
  p.gx = ntiles; p.gy = mtiles; p.gz = 1;
  p.row_order = 0; p.remap = 0;
  const dim3 grid((unsigned)ntiles * (unsigned)mtiles, 1, 1);
  if (p.mode == 1) hipLaunchKernelGGL((gather_gemm_v4_kernel<true>), grid, dim3(256), 0, st, p);
  else             hipLaunchKernelGGL((gather_gemm_v4_kernel<false>), grid, dim3(256), 0, st, p);
  VEEX_CHECK_LAUNCH();
  return 0;
}
