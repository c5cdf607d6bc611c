// Halo-tile implicit GEMM on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the Conv2D / Conv2DTranspose forwards of
// models.py:225-235,270-298 and their input gradients (tape.gradient, models.py:166) in the bf16 engine.
//
// The general bf16 kernel (gather_gemm_bf16.hip) fetches the A operand of every (tap, channel chunk) from L2 again:
// a 3x3 layer reads each input element nine times, and its K loop runs at the rate a CU can gather 16-byte pieces
// from L2 (~29 B/clk), not at the MFMA rate.  Here a workgroup owns G whole samples (or one BAND of the
// output rows of one sample) and one tile of output channels:
//   * the bf16 input pixels it needs ([pixel][channel] rows padded by 16 B so that the b128 operand reads of 16
//     consecutive pixels fall on 16 different 4-bank groups) are staged in LDS ONCE -- all channels, or one slice of
//     CH channels at a time (slice outer, taps inner; single-phase layers only);
//   * every tap -- and every one of the four sub-pixel phases of a stride-2 transposed convolution -- reads its A
//     fragments from that image: lane = output row, its LDS address = (address of the input pixel of tap (0,0)) +
//     (tap displacement), or the address of a zero row when the tap falls into the padding (one v_cndmask per
//     fragment and tap, no pad ring in LDS);
//   * only the weights stream: [BN][64-channel] chunks of the packed [tap][Cout][Cin] copy, register-staged into a
//     double-buffered LDS tile, one barrier per chunk.
// Rows of a workgroup: (sample, y, x) of one phase, up to 32*MT*WM of them, dealt to the WM wave rows fragment by
// fragment (round-robin, so that a phase that fills only part of the slots still spreads over all SIMDs); the
// accumulators are flushed through the fused epilogue after each phase (bias + ReLU + bf16 BatchNorm copy, or
// BatchNorm/ReLU backward + column partial sums), so a stride-2 layer needs no more registers than a stride-1 one.
// Prologue (staging) and epilogue of a workgroup are not overlapped with its own K loop: the planner sizes the
// workgroups so that two or three of them share a CU (LDS <= 80 / 53 KB, <= 128 VGPRs) wherever the layer allows.
#include "gg_common.h"
#include <stdio.h>
#include <string.h>
#include <array>
#include <map>
#include <mutex>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HG_MAX_BANDS 8
struct HGPlan {
  int G;          // samples per workgroup (1 when nbands > 1)
  int groups;     // ceil(Nb / G)
  int nbands;     // bands of output rows per sample
  int ntn;        // tiles of BN output channels
  int CH;         // channels of the halo resident at a time (a divisor of Cin, multiple of 64)
  int rowbytes;   // LDS bytes per halo pixel: CH * 2 + 16
  int zoff;       // LDS byte offset of the zero row
  int lds_b;      // LDS byte offset of the two weight buffers
  int lds_tab;    // LDS byte offset of the row tables (abase, amask, offb: BM words each) and the column-sum scratch
  int cshift;     // log2(CH / 8): 16-byte pieces per halo pixel
  int mt, nt;     // template instance
  int res;        // workgroups the planner expects to share a CU
  int stagger;    // start delay of the 2nd (3rd) resident workgroup of a CU, in s_sleep(127) units of ~3.4 us / 2.4 GHz
  int iy0[HG_MAX_BANDS], iy1[HG_MAX_BANDS];     // input rows [iy0, iy1) a band needs
  int j0[4][HG_MAX_BANDS + 1];                  // phase ph, band b: phase rows [j0[ph][b], j0[ph][b+1])
  size_t lds_bytes;
  double cost;    // planner's estimate (cycles per CU)
  int dbg;        // make TIMING=1 builds only (VEEX_HG_DBG): 1 no MFMA / operand reads, 2 no epilogue, 4 no halo staging, 8 no weight chunks, 16 no barrier per chunk, 32 no row tables, 64 return at once -- results wrong
};

__device__ __forceinline__ void hg_fdivmod(int r, int d, float inv, int& qo, int& rem) {
  int q = (int)((float)r * inv);
  int m = r - q * d;
  if (m < 0) { q -= 1; m += d; }
  if (m >= d) { q += 1; m -= d; }
  qo = q; rem = m;
}

__device__ __forceinline__ uint4 hg_bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4u f = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(f[0], f[1], f[2], f[3]);
}

struct HGPhase {   // geometry of one output-parity phase (the whole layer when nphase == 1); all wave-uniform
  int py, px, Hp, Wp, oy0, ox0, ostep, kys, kxs, kstep, nky, nkx, ntaps;
};
__host__ __device__ __forceinline__ HGPhase hg_phase(const GGParams& p, int ph) {
  HGPhase g;
  g.py = ph >> 1; g.px = ph & 1;
  g.Hp = p.Hp[g.py]; g.Wp = p.Wp[g.px];
  g.oy0 = p.o0[g.py]; g.ox0 = p.o0[g.px];
  if (p.nphase == 4) {
    g.ostep = 2; g.kys = g.py; g.kxs = g.px; g.kstep = 2;
    g.nky = (p.kh - g.py + 1) >> 1; g.nkx = (p.kw - g.px + 1) >> 1;
  } else {
    g.ostep = 1; g.kys = 0; g.kxs = 0; g.kstep = 1; g.nky = p.kh; g.nkx = p.kw;
  }
  g.ntaps = g.nky * g.nkx;
  return g;
}

template <bool MODE1, int MT, int NT, int WM, int WN, int OCC>
__global__ __launch_bounds__(64 * WM * WN, OCC) void halo_gemm_bf16_kernel(const GGParams p, const HGPlan h) {
  constexpr int NTH = 64 * WM * WN, BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr int KC = 64, BROW = KC * 2 + 16;          // weight chunk: 64 channels per row, rows padded to 144 B
  constexpr int BPT = (BN * 8 + NTH - 1) / NTH;       // 16-byte weight pieces per thread and chunk
  constexpr int SU = 4;                               // halo pieces in flight per thread while staging
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Bs = smem + h.lds_b;
  int* const s_abase = reinterpret_cast<int*>(smem + h.lds_tab);
  unsigned* const s_amask = reinterpret_cast<unsigned*>(s_abase + BM);
  unsigned* const s_offb = s_amask + BM;
  float* const s_red = reinterpret_cast<float*>(s_offb + BM);   // [WM][BN][3]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;

  int bnt, unit, kz;
  veex_block_coords(1, h.ntn, h.groups * h.nbands, bnt, unit, kz);
  bnt = __builtin_amdgcn_readfirstlane(bnt);
  unit = __builtin_amdgcn_readfirstlane(unit);
  const int grp = unit / h.nbands, band = unit - grp * h.nbands;
  const int n0c = bnt * BN;                 // first output channel of the tile
  const int s0 = grp * h.G;                 // first sample
  const int Gv = min(h.G, p.Nb - s0);       // valid samples of this group
  const int iy0 = h.iy0[band], HB = h.iy1[band] - iy0;   // staged input rows of each sample
  const int HBW = HB * p.Win;
  const unsigned pix_bytes = (unsigned)p.Cin * 2u;
  const int nsl = p.Cin / h.CH;             // halo slices (1 unless the layer is single-phase)
  const int ncs = h.CH / KC;                // weight chunks per tap and slice


  // ---- weight chunk producer: strictly in the order the consumer walks: (slice | phase), tap, channel chunk ----
  int pr_ph = 0, pr_sl = 0, pr_cs = 0, pr_ia = 0, pr_ib = 0;
  HGPhase pg = hg_phase(p, 0);
  unsigned b_goff[BPT], b_loff[BPT];
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int piece = tid + u * NTH;
    const int n = piece >> 3, part = piece & 7;
    const bool ok = (piece < BN * 8) && (n0c + n < p.Cout);
    b_goff[u] = ok ? (unsigned)(((n0c + n) * p.Cin) * 2 + part * 16) : VEEX_OOB;
    b_loff[u] = (unsigned)(n * BROW + part * 16);
  }
  const unsigned tap_bytes = (unsigned)p.Cin * p.Cout * 2u;
  // The chunk loads are inline asm: issued by the compiler's own buffer-load builtin, the load that is in flight across
  // the loop's back edge made its wait-count pass put `s_waitcnt vmcnt(0)` in front of the operand reads at the TOP of
  // every step (it cannot tell which registers the pending load targets once paths merge) -- the prefetch then hid
  // nothing.  The asm load is invisible to that pass; chunk_wait() below is its only wait.  (The compiler's own
  // vmcnt(N) waits stay correct beside it: an extra load in the queue can only make them wait longer.)
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i rWs, rAs;     // buffer descriptors of the packed weights and of the bf16 input copy (range-checked, raw)
  {
    const unsigned long long aa = reinterpret_cast<unsigned long long>(p.Abf);
    rAs[0] = (int)(unsigned)aa; rAs[1] = (int)(unsigned)((aa >> 32) & 0xFFFFu);
    rAs[2] = (int)((unsigned)p.Nb * p.Hin * p.Win * pix_bytes); rAs[3] = 0x00020000;
    const unsigned long long wa = reinterpret_cast<unsigned long long>(p.Wbf);
    rWs[0] = (int)(unsigned)wa; rWs[1] = (int)(unsigned)((wa >> 32) & 0xFFFFu);
    rWs[2] = (int)((unsigned)p.kh * p.kw * p.Cin * p.Cout * 2u); rWs[3] = 0x00020000;
  }
  v4u breg[BPT];
  auto chunk_wait = [&]() {
#pragma unroll
    for (int u = 0; u < BPT; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(breg[u]) : : "memory");
  };
  auto load_chunk = [&]() {
    const bool valid = pr_ph < p.nphase;
    const int tap = (pg.kys + pg.kstep * pr_ia) * p.kw + (pg.kxs + pg.kstep * pr_ib);
    const unsigned soff = valid ? (unsigned)tap * tap_bytes + (unsigned)(pr_sl * h.CH + pr_cs * KC) * 2u : 0u;
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      const unsigned off = valid ? b_goff[u] + soff : VEEX_OOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(breg[u]) : "v"(off), "s"(rWs) : "memory");
    }
    if (++pr_cs == ncs) {
      pr_cs = 0;
      if (++pr_ib == pg.nkx) { pr_ib = 0; ++pr_ia; }
      if (pr_ia == pg.nky) {
        pr_ia = 0;
        if (++pr_sl == nsl) {          // (nsl > 1 only with one phase)
          pr_sl = 0; ++pr_ph;
          if (pr_ph < p.nphase) pg = hg_phase(p, pr_ph);
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
    unsigned char* bs = Bs + buf * (BN * BROW);
    chunk_wait();
#pragma unroll
    for (int u = 0; u < BPT; ++u)
      if (BN * 8 % NTH == 0 || tid + u * NTH < BN * 8) *reinterpret_cast<v4u*>(bs + b_loff[u]) = breg[u];
  };

  // ---- halo staging: pixels (sample, iy0 .. iy1) are one contiguous run in HBM; slice sl takes CH channels of each ----
  auto stage_halo = [&](int sl, auto&& while_in_flight) {
    const int npix = Gv * HBW;                           // (samples past Nb are simply not staged: their rows are never read)
    const int npieces = npix << h.cshift;
    const unsigned gbase = ((unsigned)s0 * p.Hin * p.Win + (unsigned)iy0 * p.Win) * pix_bytes + (unsigned)(sl * h.CH) * 2u;
    const int cmask = (1 << h.cshift) - 1;
    bool first = true;
    for (int q0 = tid; q0 < npieces; q0 += NTH * SU) {
      v4u v[SU];     // (asm loads, like the weight chunks: compiler-visible loads inside the loop nest made its
                     //  wait-count pass guard the operand registers of every K step with vmcnt waits)
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int q = q0 + u * NTH;
        const unsigned off = q < npieces ? gbase + (unsigned)(q >> h.cshift) * pix_bytes + (unsigned)(q & cmask) * 16u : VEEX_OOB;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v[u]) : "v"(off), "s"(rAs) : "memory");
      }
      if (first) { while_in_flight(); first = false; }
#pragma unroll
      for (int u = 0; u < SU; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[u]) : : "memory");
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int q = q0 + u * NTH;
        if (q < npieces) *reinterpret_cast<v4u*>(smem + (q >> h.cshift) * h.rowbytes + (q & cmask) * 16) = v[u];
      }
    }
    if (first) while_in_flight();     // (a thread without a piece of the halo)
  };

#ifdef VEEX_DEBUG_TIMING
#define HG_DBG(bit) (h.dbg & (bit))
#else
#define HG_DBG(bit) 0
#endif
  if (HG_DBG(64)) return;
  // Workgroups that start together on one CU would stage, multiply and store in lockstep -- memory idle during the K
  // loops, matrix cores idle during staging and epilogues, on the whole chip at once.  The workgroups dispatched into
  // a CU's 2nd / 3rd slot (blocks go round-robin over the 8 XCDs and, in order, over an XCD's 32 CUs: a guess about
  // placement, for speed only) start later by about one staging time, so that their staging runs beside the first
  // one's K loop and the first one's epilogue beside their K loop.
  if (h.stagger > 0 && h.res > 1) {
    const int slot = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) >> 5) % h.res);
    for (int i = 0; i < slot * h.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  auto row_tables = [&](int ph) {
    const HGPhase g = hg_phase(p, ph);
    const int j0 = h.j0[ph][band], nj = h.j0[ph][band + 1] - j0;
    const int RS = nj * g.Wp;
    const int rows = Gv * RS;
    {
      const float invRS = 1.0f / (float)max(RS, 1), invWp = 1.0f / (float)max(g.Wp, 1);
      for (int r = tid; r < BM; r += NTH) {
        int abase = 0; unsigned mask = 0u, offb = 0x80000000u;
        if (r < rows && !HG_DBG(32)) {
          int s, rem, j, i;
          hg_fdivmod(r, RS, invRS, s, rem);
          hg_fdivmod(rem, g.Wp, invWp, j, i);
          const int oy = g.ostep * (j0 + j) + g.oy0, ox = g.ostep * i + g.ox0;
          offb = (unsigned)((((s0 + s) * p.Hout + oy) * p.Wout + ox) * p.Cout) * 4u;
          int by, bx;
          if (MODE1) { by = (oy + p.pb - g.kys) >> (p.stride - 1); bx = (ox + p.pb - g.kxs) >> (p.stride - 1); }
          else       { by = p.stride * oy - p.pb; bx = p.stride * ox - p.pb; }
          abase = ((s * HB + by - iy0) * p.Win + bx) * h.rowbytes;
          for (int ia = 0; ia < g.nky; ++ia)
            for (int ib = 0; ib < g.nkx; ++ib) {
              const int iy = MODE1 ? by - ia : by + ia, ix = MODE1 ? bx - ib : bx + ib;
              bool ok = iy >= iy0 && iy < iy0 + HB && ix >= 0 && ix < p.Win;   // (the band's window covers every in-image tap)
              if (MODE1) ok = ok && (oy + p.pb - (g.kys + g.kstep * ia)) >= 0 && (ox + p.pb - (g.kxs + g.kstep * ib)) >= 0;
              if (ok) mask |= 1u << (ia * g.nkx + ib);
            }
        }
        s_abase[r] = abase; s_amask[r] = mask; s_offb[r] = offb;
      }
    }
  };
  load_chunk();   // chunk 0 in flight while the halo is staged
  if (!HG_DBG(4)) stage_halo(0, [&]() { row_tables(0); }); else row_tables(0);
  for (int q = tid; q * 16 < h.rowbytes; q += NTH) *reinterpret_cast<uint4*>(smem + h.zoff + q * 16) = make_uint4(0u, 0u, 0u, 0u);
  store_chunk(0);
  load_chunk();   // chunk 1
  int cur = 0;

  const unsigned out_bytes = (unsigned)p.Nb * p.Hout * p.Wout * p.Cout * 4u;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.Out != nullptr ? p.Out : const_cast<float*>(p.A), 0, p.Out != nullptr ? (int)out_bytes : 0, 0x00020000);
  const bool has_bf = p.OutBf != nullptr;
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(has_bf ? p.OutBf : const_cast<float*>(p.A), 0, has_bf ? (int)(out_bytes >> 1) : 0, 0x00020000);
  const bool bwd = p.epi == VEEX_EPI_BWD;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bwd ? p.aux : p.A), 0, bwd ? (int)out_bytes : 0, 0x00020000);

  for (int ph = 0; ph < p.nphase; ++ph) {
    const HGPhase g = hg_phase(p, ph);
    const int j0 = h.j0[ph][band], nj = h.j0[ph][band + 1] - j0;
    const int RS = nj * g.Wp;                 // rows per sample in this phase and band
    const int rows = Gv * RS;
    // ---- row tables of this phase (phase 0: computed in the prologue, while the halo loads were in flight) ----
    if (ph > 0) {
      __syncthreads();      // the previous phase's epilogue has read its tables
      row_tables(ph);
    }
    __syncthreads();                  // tables (and, in phase 0, the halo and weight chunk 0) are visible
    int abase[MT]; unsigned amask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int r = (i * WM + wm) * 32 + l31;
      abase[i] = s_abase[r] + lh * 16; amask[i] = s_amask[r];
    }
    // fragments of this wave that hold rows of this phase (wave-uniform): fragment f = i * WM + wm covers rows 32f..
    const int nfrag = (rows + 31) >> 5;
    const int mt_act = __builtin_amdgcn_readfirstlane(nfrag > wm ? min(MT, (nfrag - wm + WM - 1) / WM) : 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned brow0 = (unsigned)((wn * NT * 32 + l31) * BROW + lh * 16);
    for (int sl = 0; sl < nsl; ++sl) {
      if (sl > 0) {                   // next channel slice of the halo (every wave has passed the last chunk's barrier)
        stage_halo(sl, []() {});
        __syncthreads();
      }
      int ti = 0;
      for (int ia = 0; ia < g.nky; ++ia)
        for (int ib = 0; ib < g.nkx; ++ib, ++ti) {
          const int dpix = ia * p.Win + ib;
          const int delta = (MODE1 ? -dpix : dpix) * h.rowbytes;
          int addr[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) addr[i] = ((amask[i] >> ti) & 1u) ? abase[i] + delta : h.zoff + lh * 16;
          for (int cs = 0; cs < ncs; ++cs) {
            const unsigned char* bs = Bs + cur * (BN * BROW) + brow0;
            // (fragments past the phase's rows read the zero row: a per-fragment skip put a branch and a wait in front
            //  of every MFMA; a wave without any row of the phase skips the chunk as a whole)
            // The operand reads of k-step ks + 1 are issued before the MFMAs of k-step ks (two named register sets, the
            // order pinned): left to itself the compiler, asked for <= 128 registers, read one fragment, waited, and
            // multiplied -- a workgroup alone on its CU then ran at half the MFMA rate.
            if (!HG_DBG(1) && mt_act > 0) {
              const unsigned char* as = smem + cs * (KC * 2);
              bf16x8 a0[MT], b0[NT], a1[MT], b1[NT];
#define HG_LOADK(AV, BV, KS)                                                                                        \
  {                                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) BV[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * BROW + (KS) * 32); \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AV[i] = *reinterpret_cast<const bf16x8*>(as + addr[i] + (KS) * 32);      \
  }
#define HG_MFMAK(AV, BV)                                                                                            \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                   \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV[i], BV[j], acc[i][j], 0, 0, 0);                       \
  }
              HG_LOADK(a0, b0, 0)
              __builtin_amdgcn_sched_barrier(0);
              HG_LOADK(a1, b1, 1)
              __builtin_amdgcn_sched_barrier(0);
              HG_MFMAK(a0, b0)
              __builtin_amdgcn_sched_barrier(0);
              HG_LOADK(a0, b0, 2)
              __builtin_amdgcn_sched_barrier(0);
              HG_MFMAK(a1, b1)
              __builtin_amdgcn_sched_barrier(0);
              HG_LOADK(a1, b1, 3)
              __builtin_amdgcn_sched_barrier(0);
              HG_MFMAK(a0, b0)
              __builtin_amdgcn_sched_barrier(0);
              HG_MFMAK(a1, b1)
#undef HG_LOADK
#undef HG_MFMAK
            }
            // next chunk (loaded one step ago) into the other buffer -- last read in the previous step, before its barrier
            if (!HG_DBG(8)) { store_chunk(cur ^ 1); load_chunk(); }
            if (!HG_DBG(16)) __syncthreads();
            cur ^= 1;
          }
        }
    }

    // ---- epilogue of the phase ----
    float cs1[NT], cs2[NT], cs3[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; cs3[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0c + (wn * NT + j) * 32 + l31;
      if (col >= p.Cout) continue;
      if (HG_DBG(2)) { if (acc[0][j][0] == 1.2345e-30f) s_red[0] = 1.f; continue; }
      const unsigned col4 = (unsigned)col * 4u, col2 = (unsigned)col * 2u;
      float bia = 0.f, esc = 1.f, osc = 1.f, osh = 0.f;
      if (!bwd && p.bias != nullptr) bia = p.bias[col];
      if (bwd && p.e_scale != nullptr) esc = p.e_scale[col & p.epi_cmask];
      if (has_bf && !bwd && p.o_scale != nullptr) { osc = p.o_scale[col & p.o_cmask]; osh = p.o_shift[col & p.o_cmask]; }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i >= mt_act) continue;
        const unsigned* ofs = s_offb + (i * WM + wm) * 32 + 4 * lh;     // rows 8q + 4 lh + (0..3) of the fragment
        if (!bwd) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 ro = *reinterpret_cast<const uint4*>(ofs + 8 * q);
            const unsigned rof[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[i][j][4 * q + e] + bia;
              if (p.relu) v = fmaxf(v, 0.f);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[e] + col4, 0, 0);
              if (has_bf) {
                const __bf16 hv = (__bf16)fmaf(v, osc, osh);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rB, (rof[e] >> 1) + col2, 0, 0);
              }
            }
          }
        } else {
          float a[16];     // the fragment's 16 activations in flight at once: one memory round trip per fragment
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 ro = *reinterpret_cast<const uint4*>(ofs + 8 * q);
            a[4 * q + 0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, ro.x + col4, 0, 0));
            a[4 * q + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, ro.y + col4, 0, 0));
            a[4 * q + 2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, ro.z + col4, 0, 0));
            a[4 * q + 3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, ro.w + col4, 0, 0));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 ro = *reinterpret_cast<const uint4*>(ofs + 8 * q);
            const unsigned rof[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[i][j][4 * q + e];
              const float vm = (!p.relu || a[4 * q + e] > 0.f) ? v : 0.f;
              cs1[j] += v;
              cs2[j] = fmaf(v, a[4 * q + e], cs2[j]);
              cs3[j] += vm;
              const float dv = vm * esc;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rO, rof[e] + col4, 0, 0);
              if (has_bf) {
                const __bf16 hv = (__bf16)dv;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rB, (rof[e] >> 1) + col2, 0, 0);
              }
            }
          }
        }
      }
    }
    if (bwd && p.partials != nullptr) {   // column sums: lane halves -> the WM wave rows -> one row of the partial slab
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        cs1[j] += __shfl_xor(cs1[j], 32);
        cs2[j] += __shfl_xor(cs2[j], 32);
        cs3[j] += __shfl_xor(cs3[j], 32);
        if (lh == 0) {
          const int cl = (wn * NT + j) * 32 + l31;
          s_red[(wm * BN + cl) * 3 + 0] = cs1[j];
          s_red[(wm * BN + cl) * 3 + 1] = cs2[j];
          s_red[(wm * BN + cl) * 3 + 2] = cs3[j];
        }
      }
      __syncthreads();
      if (tid < BN && n0c + tid < p.Cout) {
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t1 += s_red[(w * BN + tid) * 3 + 0];
          t2 += s_red[(w * BN + tid) * 3 + 1];
          t3 += s_red[(w * BN + tid) * 3 + 2];
        }
        float* pp = p.partials + (long long)(unit * p.nphase + ph) * 3 * p.Cout + n0c + tid;
        pp[0] = t1;
        pp[p.Cout] = t2;
        pp[2 * p.Cout] = t3;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// planning and launch (host)
// ---------------------------------------------------------------------------
static constexpr int HG_WM = 4, HG_WN = 2;
static constexpr size_t HG_LDS_MAX = 160 * 1024;

// Fills the band tables of plan h for (G, nbands) and returns the largest row count of a (phase, band), or -1 when a
// band's input window cannot be built.  p: phase tables filled.
static int hg_bands(const GGParams& p, HGPlan& h) {
  int rmax = 0;
  for (int b = 0; b < h.nbands; ++b) { h.iy0[b] = p.Hin; h.iy1[b] = 0; }
  for (int ph = 0; ph < p.nphase; ++ph) {
    const HGPhase g = hg_phase(p, ph);
    for (int b = 0; b <= h.nbands; ++b) h.j0[ph][b] = (int)(((long long)g.Hp * b) / h.nbands);
    for (int b = 0; b < h.nbands; ++b) {
      const int ja = h.j0[ph][b], jb = h.j0[ph][b + 1];
      const int r = (jb - ja) * g.Wp * h.G;
      if (r > rmax) rmax = r;
      if (jb <= ja) continue;
      const int oya = g.ostep * ja + g.oy0, oyb = g.ostep * (jb - 1) + g.oy0;
      int lo, hi;   // input rows touched by in-image taps
      if (p.mode == 1) {
        lo = ((oya + p.pb - g.kys) >> (p.stride - 1)) - (g.nky - 1);
        hi = (oyb + p.pb - g.kys) >> (p.stride - 1);
      } else {
        lo = p.stride * oya - p.pb;
        hi = p.stride * oyb - p.pb + g.nky - 1;
      }
      if (lo < 0) lo = 0;
      if (hi > p.Hin - 1) hi = p.Hin - 1;
      if (lo < h.iy0[b]) h.iy0[b] = lo;
      if (hi + 1 > h.iy1[b]) h.iy1[b] = hi + 1;
    }
  }
  for (int b = 0; b < h.nbands; ++b) if (h.iy1[b] <= h.iy0[b]) { h.iy0[b] = 0; h.iy1[b] = 1; }
  return rmax;
}

// Picks samples per workgroup / bands per sample, the resident channel slice and the channel tile by a small cost
// model: MFMA cycles of the fullest SIMD against the weight bytes the workgroup pulls through L2 (~24 B/clk per CU in
// practice), plus staging and epilogue, which only overlap with another workgroup's K loop when two or more fit on a
// CU.  VEEX_HG_PLAN forces a choice (tests, A/B sweeps).  Returns false when the launch does not fit.
static bool hg_plan_search(const GGParams& pin, HGPlan& best);
// The search walks a few hundred candidate plans: far too slow for the launch path (a whole-step call asks for the
// plan of each launch three times), so plans are cached per geometry; a re-read of the switches drops the cache.
static bool hg_plan(const GGParams& p, HGPlan& best) {
  struct Entry { bool ok; HGPlan h; };
  static std::mutex mu;
  static std::map<std::array<int, 11>, Entry> cache;
  static int gen = -1;
  const std::array<int, 11> key = {p.mode, p.stride, p.pb, p.kh, p.Nb, p.Hin, p.Win, p.Cin, p.Hout, p.Cout, p.epi == VEEX_EPI_BWD ? 1 : 0};
  std::lock_guard<std::mutex> lk(mu);
  const int g = veex_env_generation();
  if (g != gen) { cache.clear(); gen = g; }
  auto it = cache.find(key);
  if (it == cache.end()) {
    Entry e; memset(&e.h, 0, sizeof(e.h));
    e.ok = (p.kh == p.kw && p.Hout == p.Wout && p.Hin == p.Win) ? hg_plan_search(p, e.h) : false;
    it = cache.emplace(key, e).first;
  }
  best = it->second.h;
  return it->second.ok;
}
// Measured choices for the launches of the reference network (tools/hg_sweep.py at batch 512, MI355X,
// profiles/r03_halo_sweep.txt): {mode, stride, Hin, Cin, Cout} -> {G, bands, CH, NT}; G = 0: the general kernel is
// faster for this launch (stride-2 layers on tiny images re-use a staged pixel at most 9/4 times).  The cost model
// below ranks plans of ONE layer usefully but does not predict where the general kernel wins, so a geometry that is
// not in this table goes to the general kernel unless VEEX_HALO=2 asks for the model's plan.
struct HGKnown { int mode, stride, Hin, Cin, Cout, G, bands, CH, NT; };
static const HGKnown HG_KNOWN[] = {
    {0, 2, 20, 64, 128, 2, 1, 64, 2},    // enc.conv2 forward
    {1, 2, 10, 128, 64, 1, 1, 128, 1},   // enc.conv2 input gradient
    {0, 2, 10, 128, 256, 4, 1, 128, 2},  // enc.conv3 forward
    {1, 2, 5, 256, 128, 0, 0, 0, 0},     // enc.conv3 input gradient
    {1, 2, 4, 256, 256, 0, 0, 0, 0},     // dec.deconv1 forward
    {0, 2, 8, 256, 256, 4, 1, 128, 2},   // dec.deconv1 input gradient
    {1, 1, 8, 256, 128, 4, 1, 256, 1},   // dec.deconv2 forward
    {0, 1, 8, 128, 256, 4, 1, 64, 1},    // dec.deconv2 input gradient
    {1, 2, 8, 128, 128, 3, 1, 128, 1},   // dec.deconv3 forward
    {0, 2, 17, 128, 128, 0, 0, 0, 0},    // dec.deconv3 input gradient
    {1, 1, 17, 128, 64, 1, 1, 64, 1},    // dec.deconv4 forward
    {0, 1, 19, 64, 128, 1, 1, 64, 1},    // dec.deconv4 input gradient
    {1, 2, 19, 64, 64, 1, 2, 64, 1},     // dec.deconv5 forward
    {0, 2, 39, 64, 64, 0, 0, 0, 0},      // dec.deconv5 input gradient
};
// VEEX_HG_PLAN forces a plan instead of the table's (tests, tools/hg_sweep.py): decimal digits G*100000 + bands*10000 +
// (CH/32)*100 + NT*10 + OCC, each field 0 = leave it to the search (vae_eeg_xai_amd/_lib.py hg_plan_code)
struct HGForce { int G, bands, CH, NT, occ; };
static HGForce hg_forced() {
  const int v = veex_env_int("VEEX_HG_PLAN", 0);
  return HGForce{v / 100000, (v / 10000) % 10, ((v / 100) % 100) * 32, (v / 10) % 10, v % 10};
}
static bool hg_plan_search(const GGParams& pin, HGPlan& best) {
  GGParams p = pin;
  int kG = 0, kB = 0, kC = 0, kN = 0;
  const HGForce f = hg_forced();
  {
    const bool forced = f.G || f.bands || f.CH || f.NT;
    const HGKnown* k = nullptr;
    for (const HGKnown& e : HG_KNOWN)
      if (e.mode == p.mode && e.stride == p.stride && e.Hin == p.Hin && e.Cin == p.Cin && e.Cout == p.Cout && p.kh == 3) k = &e;
    if (!forced && veex_env_int("VEEX_HALO", 1) != 2) {
      if (k == nullptr || k->G == 0) return false;
      kG = k->G; kB = k->bands; kC = k->CH; kN = k->NT;
      if (kG > p.Nb) kG = p.Nb;
    }
  }
  if (p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
  if (p.Cout % 64 != 0 || p.kh * p.kw > 9 || p.kh != p.kw) return false;
  veex_gg_m_tiles_bm(p, 64);   // fills nphase, Hp, Wp, o0
  const int fG = kG ? kG : f.G, fB = kB ? kB : f.bands;
  const int fC = kC ? kC : f.CH, fN = kN ? kN : f.NT;
  bool found = false;
  for (int nt = 1; nt <= 2; ++nt) {
    const int bn = 32 * nt * HG_WN;
    if (bn > p.Cout || (fN && fN != nt)) continue;
    const int ntn = (p.Cout + bn - 1) / bn;
    for (int CH = 64; CH <= p.Cin; CH *= 2) {
      if ((p.nphase != 1 && CH != p.Cin) || (fC && fC != CH)) continue;
      const int rowbytes = CH * 2 + 16;
      int cshift = 0; while ((8 << cshift) < CH) ++cshift;
      for (int cfg = 0; cfg < 16 + HG_MAX_BANDS - 1; ++cfg) {
        HGPlan h; memset(&h, 0, sizeof(h));
        h.G = cfg < 16 ? cfg + 1 : 1;
        h.nbands = cfg < 16 ? 1 : cfg - 14;          // 2 .. HG_MAX_BANDS
        if ((fG && fG != h.G) || (fB && fB != h.nbands)) continue;
        if (h.G > p.Nb) continue;
        const int rmax = hg_bands(p, h);
        const int mt = (rmax + 32 * HG_WM - 1) / (32 * HG_WM);
        if (mt < 1 || mt > 4) continue;
        const int bm = 32 * mt * HG_WM;
        int hbmax = 0;
        for (int b = 0; b < h.nbands; ++b) hbmax = (h.iy1[b] - h.iy0[b]) > hbmax ? (h.iy1[b] - h.iy0[b]) : hbmax;
        const size_t halo = (size_t)h.G * hbmax * p.Win * rowbytes;
        const size_t lds_b = (halo + rowbytes + 15) & ~(size_t)15;
        const size_t lds_tab = lds_b + 2 * (size_t)bn * 144;
        const size_t lds = lds_tab + 3 * (size_t)bm * 4 + (size_t)HG_WM * bn * 3 * 4;
        if (lds > HG_LDS_MAX) continue;
        const int groups = (p.Nb + h.G - 1) / h.G;
        // cycles of one workgroup
        double mfma = 0.0, o_elems = 0.0;
        for (int ph = 0; ph < p.nphase; ++ph) {
          const HGPhase g = hg_phase(p, ph);
          double fr = 0.0;   // average over bands of the fullest SIMD's fragments (wave rows wm and wm + 2 share a SIMD pair)
          for (int b = 0; b < h.nbands; ++b) {
            const int rows = (h.j0[ph][b + 1] - h.j0[ph][b]) * g.Wp * h.G;
            const int nfrag = (rows + 31) / 32;
            const int a = (nfrag + HG_WM - 1) / HG_WM, c = (nfrag + HG_WM - 3) / HG_WM;
            fr += a + (c > 0 ? c : 0);
            o_elems += (double)rows * bn;
          }
          mfma += fr / h.nbands * nt * g.ntaps * (p.Cin / 16) * 32.0;
        }
        o_elems /= h.nbands;
        const double w_bytes = (double)p.kh * p.kw * p.Cin * bn * 2.0;
        const double a_bytes = (double)halo * (p.Cin / CH);
        const double loop = (mfma > w_bytes / 24.0 ? mfma : w_bytes / 24.0) + 400.0 * p.kh * p.kw * (p.Cin / 64);   // + a barrier per chunk
        const double edge = a_bytes / 24.0 + o_elems * (p.epi == VEEX_EPI_BWD ? 10.0 : 6.0) / 12.0 + 4000.0;
        const int regs_ok2 = (mt * nt <= 3);
        int res = (int)(HG_LDS_MAX / lds);
        if (res > (regs_ok2 ? 2 : 1)) res = regs_ok2 ? 2 : 1;
        const long long wgs = (long long)groups * h.nbands * ntn;
        const double per_cu = (double)((wgs + 255) / 256);
        const double cost = per_cu * (res >= 2 ? (loop > edge ? loop : edge) * 1.1 : loop + edge) + (res >= 2 ? edge : 0.0);
        if (!found || cost < best.cost) {
          found = true;
          best = h;
          best.res = res;
          best.stagger = veex_tune_int("VEEX_HG_STAGGER", 0);   // measured: every unit of delay is paid in full (profiles/r03_halo_stagger.txt); off
          best.groups = groups; best.ntn = ntn; best.CH = CH; best.rowbytes = rowbytes; best.zoff = (int)halo;
          best.lds_b = (int)lds_b; best.lds_tab = (int)lds_tab; best.cshift = cshift; best.mt = mt; best.nt = nt;
          best.lds_bytes = lds; best.cost = cost;
        }
      }
    }
  }
  return found;
}

int veex_hg_ok(const GGParams& p) {
  if (!veex_env_int("VEEX_HALO", 1)) return 0;
  if (p.Wbf == nullptr || p.Abf == nullptr || p.OutBn != nullptr) return 0;   // (a_scale is moot: the bf16 copy carries the BatchNorm)
  if (p.epi != VEEX_EPI_FWD && p.epi != VEEX_EPI_BWD) return 0;
  if (p.nsplit > 1 || p.kh * p.kw < 4) return 0;
  if ((long long)p.Nb * p.Hin * p.Win * p.Cin * 2 >= 0x7FFF0000LL) return 0;
  if ((long long)p.Nb * p.Hout * p.Wout * p.Cout * 4 >= 0x7FFF0000LL) return 0;
  if ((long long)p.Nb * p.Hout * p.Wout >= (1LL << 24)) return 0;
  HGPlan h;
  if (!hg_plan(p, h)) return 0;
  // callers size the partials slab for 32-row tiles of the general kernels: never write more rows than that
  GGParams q = p;
  const int cap = veex_gg_m_tiles_bm(q, 32);
  return (h.groups * h.nbands * q.nphase <= cap) ? 1 : 0;
}

// rows of the partials slab an EPI_BWD launch writes: one per (sample group, band, phase)
int veex_hg_m_tiles(const GGParams& p) {
  HGPlan h;
  if (!hg_plan(p, h)) return -1;
  GGParams q = p;
  veex_gg_m_tiles_bm(q, 64);
  return h.groups * h.nbands * q.nphase;
}

// OCC = waves per SIMD the register allocation must allow: 4 = two 512-thread workgroups per CU (<= 128 VGPRs)
template <bool MODE1, int MT, int NT, int OCC>
static int hg_launch(const GGParams& p, const HGPlan& h, hipStream_t st) {
  auto kern = halo_gemm_bf16_kernel<MODE1, MT, NT, HG_WM, HG_WN, OCC>;
  static bool attr_done = false;   // per instantiation, once per process, outside any capture (the first call of a shape)
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HG_LDS_MAX) != hipSuccess) {
      veex_set_error("halo_gemm(bf16): cannot raise the dynamic LDS limit"); return -3;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(h.groups * h.nbands * h.ntn)), dim3(64 * HG_WM * HG_WN), h.lds_bytes, st, p, h);
  VEEX_CHECK_LAUNCH();
  return 0;
}

// p: validated by veex_launch_gather_gemm, veex_hg_ok(p) holds
int veex_launch_halo_gemm_bf16(GGParams& p, hipStream_t st) {
  HGPlan h;
  if (!hg_plan(p, h)) { veex_set_error("halo_gemm(bf16): launch does not fit"); return -1; }
  veex_gg_m_tiles_bm(p, 64);   // phase tables (nphase, Hp, Wp, o0)
#ifdef VEEX_DEBUG_TIMING
  h.dbg = veex_tune_int("VEEX_HG_DBG", 0);
#endif
  if (veex_tune_int("VEEX_HG_VERBOSE", 0))
    fprintf(stderr, "[halo] mode %d s%d %dx%dx%d -> %dx%dx%d epi %d: G %d bands %d CH %d mt %d nt %d lds %zu wgs %d cost %.0f\n", p.mode, p.stride,
            p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.Cout, p.epi, h.G, h.nbands, h.CH, h.mt, h.nt, h.lds_bytes, h.groups * h.nbands * h.ntn, h.cost);
  const bool m1 = p.mode == 1;
  const int occ = hg_forced().occ ? hg_forced().occ : 4;
#define HG_CASE(MT_, NT_) if (h.mt == MT_ && h.nt == NT_) return m1 ? hg_launch<true, MT_, NT_, 2>(p, h, st) : hg_launch<false, MT_, NT_, 2>(p, h, st);
#define HG_CASE2(MT_, NT_) if (h.mt == MT_ && h.nt == NT_ && occ == 4) return m1 ? hg_launch<true, MT_, NT_, 4>(p, h, st) : hg_launch<false, MT_, NT_, 4>(p, h, st);
  HG_CASE2(1, 1) HG_CASE2(2, 1) HG_CASE2(3, 1) HG_CASE2(1, 2)
  HG_CASE(1, 1) HG_CASE(2, 1) HG_CASE(3, 1) HG_CASE(4, 1)
  HG_CASE(1, 2) HG_CASE(2, 2) HG_CASE(3, 2) HG_CASE(4, 2)
#undef HG_CASE
#undef HG_CASE2
  veex_set_error("halo_gemm(bf16): no instance for mt %d nt %d", h.mt, h.nt);
  return -1;
}

// A/B sweeps (tools/hg_sweep.py): the plan this launch would get -> out[0..7] = G, bands, CH, mt, nt, LDS bytes,
// workgroups, modelled cycles; returns 0 when the halo kernel does not take the launch
int veex_hg_describe(const GGParams& p, long long* out) {
  if (!veex_hg_ok(p)) return 0;
  HGPlan h;
  hg_plan(p, h);
  out[0] = h.G; out[1] = h.nbands; out[2] = h.CH; out[3] = h.mt; out[4] = h.nt; out[5] = (long long)h.lds_bytes;
  out[6] = (long long)h.groups * h.nbands * h.ntn; out[7] = (long long)h.cost;
  return 1;
}
