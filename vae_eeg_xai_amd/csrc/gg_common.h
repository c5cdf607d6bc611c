// Shared device helpers of the implicit-GEMM kernels (gather_gemm.hip, gather_gemm_bf16.hip).
#ifndef VEEX_GG_COMMON_H
#define VEEX_GG_COMMON_H
#include "veex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* ptr, bool ok, int nvalid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    if (vec && nvalid >= 4) {
      v = *reinterpret_cast<const float4*>(ptr);
    } else {
      if (nvalid > 0) v.x = ptr[0];
      if (nvalid > 1) v.y = ptr[1];
      if (nvalid > 2) v.z = ptr[2];
      if (nvalid > 3) v.w = ptr[3];
    }
  }
  return v;
}

// Shared epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gg_epilogue(const GGParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                            const int* s_off, float* As, int tile, int kz, int n0) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;
  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float cs1[NT], cs2[NT], cs3[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; cs3[j] = 0.f; }

  float* outp = p.Out;
  if (p.epi == VEEX_EPI_RAW) outp += (long long)kz * p.slab_stride;
  __bf16* const obf = (p.epi == VEEX_EPI_RAW) ? nullptr : reinterpret_cast<__bf16*>(p.OutBf);

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    const bool colok = col < p.Cout;
    float bia = 0.f, esc = 1.f, osc = 1.f, osh = 0.f;
    if (colok) {
      if (p.epi == VEEX_EPI_FWD && p.bias != nullptr) bia = p.bias[col];
      if (p.epi == VEEX_EPI_BWD && p.e_scale != nullptr) esc = p.e_scale[col & p.epi_cmask];
      if ((obf != nullptr || p.OutBn != nullptr) && p.epi == VEEX_EPI_FWD && p.o_scale != nullptr) { osc = p.o_scale[col & p.o_cmask]; osh = p.o_shift[col & p.o_cmask]; }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int off = s_off[rl];
        if (off < 0 || !colok) continue;
        const long long addr = (long long)off * p.Cout + col;
        float v = acc[i][j][r];
        if (p.epi == VEEX_EPI_FWD) {
          v += bia;
          if (p.relu) v = fmaxf(v, 0.f);
          outp[addr] = v;
          if (obf != nullptr) obf[addr] = (__bf16)fmaf(v, osc, osh);
          if (p.OutBn != nullptr) p.OutBn[addr] = fmaf(v, osc, osh);
        } else if (p.epi == VEEX_EPI_BWD) {
          const float a = p.aux[addr];
          const bool on = (!p.relu) || (a > 0.f);
          cs1[j] += v;
          cs2[j] = fmaf(v, a, cs2[j]);
          cs3[j] += on ? v : 0.f;
          const float dv = on ? v * esc : 0.f;
          outp[addr] = dv;
          if (obf != nullptr) obf[addr] = (__bf16)dv;
        } else {
          outp[addr] = v;
        }
      }
    }
  }

  if (p.epi == VEEX_EPI_BWD && p.partials != nullptr) {
    // column sums: lane halves -> waves sharing the columns (wm) -> one row of the partial slab
    float* s_red = As;   // [WM][BN][3]; safe: every wave passed the loop's last barrier
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      cs1[j] += __shfl_xor(cs1[j], 32);
      cs2[j] += __shfl_xor(cs2[j], 32);
      cs3[j] += __shfl_xor(cs3[j], 32);
      if (lh == 0) {
        const int cl = wn * WTN + j * 32 + l31;
        s_red[(wm * BN + cl) * 3 + 0] = cs1[j];
        s_red[(wm * BN + cl) * 3 + 1] = cs2[j];
        s_red[(wm * BN + cl) * 3 + 2] = cs3[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += s_red[(w * BN + tid) * 3 + 0];
        t2 += s_red[(w * BN + tid) * 3 + 1];
        t3 += s_red[(w * BN + tid) * 3 + 2];
      }
      float* pp = p.partials + (long long)tile * 3 * p.Cout + n0 + tid;
      pp[0] = t1;
      pp[p.Cout] = t2;
      pp[2 * p.Cout] = t3;
    }
  }
}

// Branch-free epilogue on BYTE row offsets (the fast kernels; outputs < 2 GiB).  s_offb[row] = byte offset of
// the fp32 output row (pixel * Cout * 4), or 0x80000000 for a row past M: every access is a raw buffer access at
// row offset + column offset, so rows past M fall out of range (loads give 0, stores are dropped) and there is no
// per-element branch, LDS read or 64-bit address arithmetic.  The bf16 copy lives at half the fp32 byte offset.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gg_epilogue_b(const GGParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                              const unsigned* s_offb, float* As, int tile, int kz, int n0) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5, l31 = lane & 31;
  float cs1[NT], cs2[NT], cs3[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; cs3[j] = 0.f; }

  float* outp = p.Out;
  if (p.epi == VEEX_EPI_RAW) outp += (long long)kz * p.slab_stride;
  const unsigned out_bytes = (unsigned)p.Nb * p.Hout * p.Wout * p.Cout * 4u;
  // Out == nullptr (bf16 engine, only the bf16 copy is consumed): zero records, every fp32 store is dropped
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(outp, 0, p.Out != nullptr ? (int)out_bytes : 0, 0x00020000);
  const bool has_bf = (p.epi != VEEX_EPI_RAW) && (p.OutBf != nullptr);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(has_bf ? p.OutBf : static_cast<void*>(outp), 0,
                                                                      has_bf ? (int)(out_bytes >> 1) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.epi == VEEX_EPI_BWD ? p.aux : outp), 0, p.epi == VEEX_EPI_BWD ? (int)out_bytes : 0, 0x00020000);
  const bool has_bn = (p.epi == VEEX_EPI_FWD) && (p.OutBn != nullptr);
  const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(has_bn ? p.OutBn : outp, 0, has_bn ? (int)out_bytes : 0, 0x00020000);

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= p.Cout) continue;     // (only the last column tile of a Cout that is not a multiple of 32)
    const unsigned col4 = (unsigned)col * 4u, col2 = (unsigned)col * 2u;
    float bia = 0.f, esc = 1.f, osc = 1.f, osh = 0.f;
    if (p.epi == VEEX_EPI_FWD && p.bias != nullptr) bia = p.bias[col];
    if (p.epi == VEEX_EPI_BWD && p.e_scale != nullptr) esc = p.e_scale[col & p.epi_cmask];
    if ((has_bf || has_bn) && p.epi == VEEX_EPI_FWD && p.o_scale != nullptr) { osc = p.o_scale[col & p.o_cmask]; osh = p.o_shift[col & p.o_cmask]; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 ro = *reinterpret_cast<const uint4*>(s_offb + wm * WTM + i * 32 + 8 * g + 4 * lh);
        const unsigned rof[4] = {ro.x, ro.y, ro.z, ro.w};
        if (p.epi == VEEX_EPI_FWD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * g + e] + bia;
            if (p.relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[e] + col4, 0, 0);
            if (has_bf) {
              const __bf16 h = (__bf16)fmaf(v, osc, osh);
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rB, (rof[e] >> 1) + col2, 0, 0);
            }
            if (has_bn) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(v, osc, osh)), rN, rof[e] + col4, 0, 0);
          }
        } else if (p.epi == VEEX_EPI_BWD) {
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, rof[e] + col4, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][4 * g + e];
            const float vm = (!p.relu || a[e] > 0.f) ? v : 0.f;
            cs1[j] += v;
            cs2[j] = fmaf(v, a[e], cs2[j]);
            cs3[j] += vm;
            const float dv = vm * esc;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rO, rof[e] + col4, 0, 0);
            if (has_bf) {
              const __bf16 h = (__bf16)dv;
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rB, (rof[e] >> 1) + col2, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][4 * g + e];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rO, rof[e] + col4, 0, 0);
          }
        }
      }
    }
  }

  if (p.epi == VEEX_EPI_BWD && p.partials != nullptr) {
    float* s_red = As;   // [WM][BN][3]; safe: every wave passed the loop's last barrier
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      cs1[j] += __shfl_xor(cs1[j], 32);
      cs2[j] += __shfl_xor(cs2[j], 32);
      cs3[j] += __shfl_xor(cs3[j], 32);
      if (lh == 0) {
        const int cl = wn * WTN + j * 32 + l31;
        s_red[(wm * BN + cl) * 3 + 0] = cs1[j];
        s_red[(wm * BN + cl) * 3 + 1] = cs2[j];
        s_red[(wm * BN + cl) * 3 + 2] = cs3[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += s_red[(w * BN + tid) * 3 + 0];
        t2 += s_red[(w * BN + tid) * 3 + 1];
        t3 += s_red[(w * BN + tid) * 3 + 2];
      }
      float* pp = p.partials + (long long)tile * 3 * p.Cout + n0 + tid;
      pp[0] = t1;
      pp[p.Cout] = t2;
      pp[2 * p.Cout] = t3;
    }
  }
}

typedef unsigned int v4u __attribute__((__vector_size__(16)));
typedef float v4f __attribute__((__vector_size__(16)));
#define VEEX_OOB 0xFFFFFF00u

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}


// launch-side helpers shared by the fp32 and bf16 engines
int veex_launch_gather_gemm_bf16(GGParams& p, int mtiles, hipStream_t st);   // gather_gemm_bf16.hip
int veex_gg_x3_cfg(const GGParams& p);                                         // -1: the split kernel does not take the launch
int veex_gg_x3_bm(const GGParams& p);                                          // gather_gemm_x3.hip
int veex_launch_gather_gemm_x3(GGParams& p, int mtiles, hipStream_t st);
int veex_hg_ok(const GGParams& p);                                             // halo_gemm_bf16.hip: the halo-tile kernel takes it
int veex_hg_m_tiles(const GGParams& p);                                        // rows of its partials slab (sample groups x phases)
int veex_launch_halo_gemm_bf16(GGParams& p, hipStream_t st);
int veex_hg_describe(const GGParams& p, long long* out8);
int veex_gg_v3_ok(const GGParams& p);                                          // gather_gemm_v3.hip
int veex_launch_gather_gemm_v3(GGParams& p, int mtiles, hipStream_t st);
int veex_gg_v3_bm(const GGParams& p);                                          // tile rows the lean fp32 kernel uses
int veex_gg_v4_ok(const GGParams& p);                                          // gather_gemm_v4.hip (small launches)
int veex_launch_gather_gemm_v4(GGParams& p, int mtiles, hipStream_t st);

#endif
