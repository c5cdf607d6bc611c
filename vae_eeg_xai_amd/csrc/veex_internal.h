// Internal (non-ABI) declarations shared by the veex HIP translation units.
// gfx950 / CDNA4 only.  The public C-ABI is include/veex.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define VEEX_BK 16          // K-chunk of every MFMA GEMM tile (floats)
#define VEEX_LDK 20         // padded k-stride of [row][k] LDS tiles (16 + one b128)

enum { VEEX_EPI_FWD = 0, VEEX_EPI_BWD = 1, VEEX_EPI_RAW = 2 };

// ---------------------------------------------------------------------------
// Gather-GEMM (implicit GEMM for Conv2D / Conv2DTranspose / Dense, forward and dgrad)
//   Out[m][n] = sum_{tap, ci} Aval(m, tap, ci) * Wval(tap, ci, n)
// mode 0 (conv-gather):   in = stride*out + k - pb     (Conv2D fwd, Conv2DTranspose dgrad, Dense)
// mode 1 (deconv-gather): in = (out + pb - k)/stride   (Conv2DTranspose fwd, Conv2D dgrad, Dense dgrad)
// Weight memory: mode 0 -> [tap][Cin][Cout]  (n contiguous);  mode 1 -> [tap][Cout][Cin] (k contiguous)
// ---------------------------------------------------------------------------
struct GGParams {
  const float* A;        // gathered operand, NHWC [Nb][Hin][Win][Cin]
  const float* W;
  const void* Wbf;        // optional packed bf16 weights [tap][Cout][Cin]: run the bf16-MFMA kernel
  int wplanes;            // 3: Wbf holds the three-term split [plane][tap][Cout][Cin] of the fp32 weights: run the
                          // split-fp32 kernel (gather_gemm_x3.hip: fp32 result on the bf16 matrix cores); else 0 / 1
  const float* Wt;        // optional fp32 copy of W as [tap][Cout][Cin] for a MODE 0 launch of the lean fp32 kernel (whose
                          // B fragment then is one 16-byte LDS read per four MFMAs, like mode 1, instead of four 4-byte ones)
  float* Out;            // NHWC [Nb][Hout][Wout][Cout]; RAW: slab base [nsplit][same]
  const float* a_scale;  // optional per-channel affine applied to in-bounds A loads (producer's BN)
  const float* a_shift;
  const float* bias;     // EPI_FWD (nullable)
  const float* aux;      // EPI_BWD: post-ReLU activation of the layer being differentiated
  const float* e_scale;  // EPI_BWD: BN scale of that layer (nullable -> 1)
  float* partials;       // EPI_BWD: [m_tiles][3][Cout] column partial sums (nullable)
  int Nb, Hin, Win, Cin, Hout, Wout, Cout;
  int kh, kw, stride, pb;
  int mode;
  int aff_mask, aff_spatial;
  int epi, relu, epi_cmask;
  int nsplit, chunks_per_split;
  int nphase;            // 1, or 4 for stride-2 deconv-gather (sub-pixel phases)
  int tile_prefix[5];    // cumulative m-tiles per phase
  int Hp[2], Wp[2];      // rows / cols of each parity class (nphase==1: Hp[0]=Hout, Wp[0]=Wout)
  int o0[2];             // first output coordinate of parity class p
  long long slab_stride; // elements per split slab (RAW)
  int gx, gy, gz;        // logical grid (n tiles, m tiles, k splits), filled by the launcher
  int remap;             // 1: XCD-aware block order (veex_block_coords)
  int tap_inner;         // general fp32 kernel: K-chunk order, 1 = channel chunk outer / tap inner, 0 = tap outer (the lean and bf16 kernels are always tap-inner)
  // bf16 "consumer copies" (bf16 compute only): the epilogue can also write what the NEXT bf16 GEMM
  // reads -- bf16(o_scale[c]*out + o_shift[c]) for a forward launch (this layer's BatchNorm applied),
  // bf16(out) for a gradient -- and the loader can read such a copy (no affine, no conversion).
  const void* Abf;       // optional bf16 copy of A (BatchNorm already applied): replaces A / a_scale / a_shift
  void* OutBf;           // optional bf16 copy of Out
  const float* o_scale;  // EPI_FWD: BatchNorm of THIS layer for OutBf (nullable: plain rounding)
  const float* o_shift;
  int o_cmask;           // o_scale index = col & o_cmask
  float* OutBn;          // optional (EPI_FWD): fp32 copy o_scale[c]*out + o_shift[c] -- THIS layer's BatchNorm applied once by
                         // the producer, so that the fp32 GEMMs that consume the activation (next forward, weight gradient)
                         // stage it without an affine prologue (and zero padding is zero again without a masked shift)
  int row_order;         // lean fp32 kernel: 0 = tile rows run over (n, y, x), 1 = over (y, n, x) -- one image row per tile,
                         // 2 = over (y, x, n) with Nb % tile rows == 0 -- one output pixel per tile, 3 = the same with the
                         // interior pixels first in raster order, then the border rows / columns (longest tiles first)
  int perm_lo, perm_hi;  // row_order 3: output rows / columns at the low / high border that have padded taps
  int row_tiles;         // m tiles per image row (row_order 1, remap 2) / per pixel (row_order 2, 3); with remap == 2 a multiple
                         // of 8: XCD x owns the tiles g % 8 == x of every row / pixel
};

// ---------------------------------------------------------------------------
// Weight-gradient GEMM:  dW[tap][cb][cs] = sum_{n,py,px} Big(n, s*py+ky-pb, s*px+kx-pb, cb) * Small(n,py,px,cs)
// Conv2D: Big = layer input, Small = d(pre-activation);  Conv2DTranspose: Big = d(pre-act), Small = layer input.
// ---------------------------------------------------------------------------
struct WGParams {
  const float* Big;
  const float* Small;
  float* Out;            // [nsplit][M][Cs] partial slabs (nsplit==1: the gradient itself)
  const float* b_scale; const float* b_shift;
  const float* s_scale; const float* s_shift;
  int b_aff_mask, b_aff_spatial, s_aff_mask, s_aff_spatial;
  int Nb, Hb, Wb, Cb, Hs, Ws, Cs;
  int kh, kw, stride, pb;
  int R;                 // Nb*Hs*Ws reduction rows
  int rows_per_split;    // multiple of VEEX_BK
  int M;                 // kh*kw*Cb
  int nsplit;
  int bf16;              // 1: bf16 MFMA (operands rounded while staging); 2: + both operands as bf16 copies
  const void* BigBf;     // optional bf16 copies [rows][C] of Big / Small (BatchNorm already applied to the
  const void* SmallBf;   // activation side): staged with 16-byte copies + an in-register 8x8 transpose
  int gx, gy, gz;        // logical grid (m tiles, n tiles, row splits), filled by the launcher
  int remap;
  int x3;                // 1 (bf16 == 0): fp32 results from three-term bf16 splits on the bf16 MFMA (wgrad_gemm_x3.hip)
};

// Logical block coordinates of a 1-D launch, XCD-aware: workgroups are placed round-robin over the
// 8 XCDs (linear id % 8), each with its own 4 MiB L2, so CONSECUTIVE logical blocks -- which share
// operand tiles (the same rows for every n tile / tap tile, halo rows between neighbouring m tiles)
// -- are mapped onto ONE XCD and dispatched close in time.  Speed only: any placement is correct.
#ifdef __HIPCC__
__device__ __forceinline__ void veex_block_coords(int remap, int gx, int gy, int& bx, int& by, int& bz) {
  const unsigned total = gridDim.x, lin = blockIdx.x;
  const unsigned xcd = lin & 7u, local = lin >> 3;
  const unsigned q = total >> 3, r = total & 7u;
  const unsigned logical = remap ? xcd * q + (xcd < r ? xcd : r) + local : lin;
  const unsigned t = logical / (unsigned)gx;
  bx = (int)(logical - t * (unsigned)gx);
  bz = (int)(t / (unsigned)gy);
  by = (int)(t - (unsigned)bz * (unsigned)gy);
}
#endif

int veex_env_generation(void);                  // changes whenever veexi_debug_reload_env() made the switches stale
int veex_env_int(const char* name, int dflt);   // read once per process and frozen; veexi_debug_reload_env() re-reads
// Tuning knobs of the A/B sweeps behind DESIGN 5.1's table: they exist only in the tuning build (make TIMING=1).  In
// the shipped library every one of them is its default, a constant the compiler folds (its other branch is not built).
#ifdef VEEX_DEBUG_TIMING
inline int veex_tune_int(const char* name, int dflt) { return veex_env_int(name, dflt); }
#else
inline int veex_tune_int(const char*, int dflt) { return dflt; }
#endif

// launchers (host side; enqueue only, no sync, no allocation)
int veex_launch_gather_gemm(const GGParams& p, hipStream_t st);
int veex_launch_wgrad(const WGParams& p, hipStream_t st);
int veex_gg_m_tiles(GGParams& p);   // fills phase tables; returns total m tiles for the chosen config
int veex_gg_m_tiles_bm(GGParams& p, int bm);   // same for a given tile height (64: upper bound for sizing)
int veex_gg_bm(const GGParams& p);
int veex_gg_bf16_only(const GGParams& p);   // gather_gemm.hip: 0 no, 1 reads only the bf16 A copy, 2 and may write the bf16 copy alone
int veex_wgrad_bf16c_ok(const WGParams& p);  // wgrad_gemm_bf16.hip: the copy kernel (reads only the two bf16 copies) takes it
int veex_wg_tiles(const WGParams& p, int* bm, int* bn);
int veex_wgrad_bf16_ok(const WGParams& p);                       // wgrad_gemm_bf16.hip
#define VEEX_WGT_MAX_ROWS 4096                                    // rows_per_split the LDS row table of wgrad_gemm_bf16t.hip holds (16 KB)
#define VEEX_WG3_MAX_ROWS 3072                                    // rows_per_split the LDS row table of wgrad_gemm_v3.hip holds
#define VEEX_PARTIALS2_N (256LL * 768)   // floats: 256 compacted rows of 3 x 256 channels (one entry of veex_colsum_batch)
// Every column-sum finalisation of one backward pass in two launches (instead of up to 17 of ~5 us each):
// the entries are collected while the pass is enqueued and run once at its end (only Adam reads the results).
#define VEEX_COLSUM_MAX 12
struct ColsumEntry {
  const float* partials; int T; long long row_stride; int ncols, nch;     // as veex_colsum_finalize
  const float *e_scale, *mean, *rstd; float *dbias, *dgamma, *dbeta;
  int rpc, nchunks, cgroups;      // compaction: rows per chunk, chunks (0: not compacted), 256-column groups
  int blk_c, blk_f;               // first block of this entry in the compaction / finalisation grid
  long long scr_off;              // its compacted rows in the scratch
};
struct ColsumBatch { int n; ColsumEntry e[VEEX_COLSUM_MAX]; };
int veex_colsum_batch(ColsumBatch& b, float* scratch, long long scratch_n, hipStream_t st);   // special.hip
int veex_wgrad_v3_ok(const WGParams& p);                         // wgrad_gemm_v3.hip
int veex_launch_wgrad_v3(const WGParams& p, int bm, int bn, hipStream_t st);
int veex_wgrad_bf16t_ok(const WGParams& p);                      // wgrad_gemm_bf16t.hip (p passes veex_wgrad_bf16c_ok)
int veex_launch_wgrad_bf16t(const WGParams& p, hipStream_t st);
int veex_wgrad_x3_ok(const WGParams& p, int bm, int bn);         // wgrad_gemm_x3.hip
int veex_launch_wgrad_x3(const WGParams& p, int bm, int bn, hipStream_t st);
int veex_launch_wgrad_bf16(const WGParams& p, hipStream_t st);
extern "C" int veex_conv1_fwd_bf16(const float* x, const float* W, const float* bias, float* out, const float* bn_scale,
                                   const float* bn_shift, void* out_bf16, int B, hipStream_t st);
extern "C" int veex_conv1_fwd_copies(const float* x, const float* W, const float* bias, float* out, const float* bn_scale,
                                     const float* bn_shift, float* out_bn, void* out_bf16, int B, hipStream_t st);
extern "C" int veex_deconv6_bwd_bf16(const float* dlogit, const float* a5, const float* scale, const float* shift,
                                     const float* W6, float* dpre5, void* dpre5_bf16, float* slab, int B, hipStream_t st);
int veex_pack_weights_bf16_multi(int n, const float* const* src, void* const* dst, const int* taps, const int* rows,
                                 const int* cols, const int* transpose, hipStream_t st);
extern "C" int veex_pack_weights_bf16(const float* src, void* dst, int taps, int rows, int cols, int transpose, hipStream_t st);
// fp32 transposed copies dst[tap][c][r] = src[tap][r][c] of up to 16 kernels in one launch (GGParams::Wt)
int veex_transpose_weights_f32_multi(int n, const float* const* src, float* const* dst, const int* taps, const int* rows,
                                     const int* cols, hipStream_t st);

// bn_batch.hip: BatchNormalization on batch statistics (veex_net_set_semantics, VEEX_BN_BATCH)
long long veex_bn_batch_scratch_doubles(long long rows, int C);
int veex_bn_batch_forward(const float* act, long long rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                          int unbiased, float* scale, float* shift, float* rstd, float* mu, float* moving_mean, float* moving_var,
                          double* scratch, hipStream_t st);
int veex_bn_batch_backward(const float* v, const float* a, long long rows, int C, const float* gamma, const float* mu,
                           const float* rstd, float* out, float* dgamma, float* dbeta, float* dbias, float* coef, double* scratch,
                           hipStream_t st);

void veex_set_error(const char* fmt, ...);

#define VEEX_CHECK_LAUNCH()                                        \
  do {                                                             \
    hipError_t e__ = hipGetLastError();                            \
    if (e__ != hipSuccess) {                                       \
      veex_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return -3;                                                   \
    }                                                              \
  } while (0)
