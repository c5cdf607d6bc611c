/* veex.h -- C-ABI of libveex.so: the MI355X (gfx950) implementation of the VAE / autoencoder
 * hot path of alessandroquirile/VAE-EEG-XAI.
 *
 * The reference has no FFI: its boundary is the Python class API of models.py /
 * models_standard.py, underneath which TensorFlow/Keras kernels run.  Each entry point below
 * therefore cites the reference lines whose Keras call it replaces.  The Python side
 * (vae_eeg_xai_amd/) binds these symbols with ctypes and keeps the reference's class API.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless named host_*; the caller owns all
 *    memory; nothing here allocates, frees or synchronises; every kernel is enqueued on the
 *    hipStream_t passed in (so calls can be captured into a HIP graph);
 *  - return 0 on success, negative VEEX_E* otherwise; veex_last_error() returns a
 *    thread-local message;
 *  - activations are NHWC, weights have the Keras shapes of the reference:
 *    Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin), Dense (in,out).
 */
#ifndef VEEX_H
#define VEEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* veexStream_t; /* == hipStream_t */

#define VEEX_OK 0
#define VEEX_EINVAL (-1)
#define VEEX_ENOMEM (-2)   /* caller-provided workspace too small */
#define VEEX_ELAUNCH (-3)

#define VEEX_VARIANT_VAE 0 /* models.py: Encoder / Decoder / VAE */
#define VEEX_VARIANT_AE 1  /* models_standard.py: EncoderStandard / DecoderStandard / Autoencoder */

/* Version of the descriptor structs and entry points declared here; veex_abi_version() returns the version the
 * LIBRARY was built with.  A caller compiled against an older header must refuse to run when the two differ:
 * VeexGemmDesc grew (weight_t, out_bn) in version 2 and the library reads those fields.  From version 3 on the structs
 * carry struct_size, so growth at the end (weight_planes, round 3) does not change the version: fields at or beyond the
 * caller's struct_size are not read. */
#define VEEX_ABI_VERSION 3
const char* veex_last_error(void);
int veex_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Operator level
 * ---------------------------------------------------------------------------------------- */

/* Implicit-GEMM convolution family on the fp32 matrix cores.
 * mode 0 "conv-gather"  : in = stride*out + k - pad_begin.  Conv2D forward (models.py:219-235,
 *                         models_standard.py:89-105), Dense forward (models.py:238-241,264),
 *                         Conv2DTranspose input-gradient (tape.gradient, models.py:166).
 *                         Weight memory [tap][Cin][Cout].
 * mode 1 "deconv-gather": in = (out + pad_begin - k)/stride.  Conv2DTranspose forward
 *                         (models.py:270-298), Conv2D / Dense input-gradient.
 *                         Weight memory [tap][Cout][Cin].
 * pad_begin is TF's SAME pad-before of the (corresponding forward) convolution, i.e. the
 * crop-at-start of a Conv2DTranspose.
 * epilogue 0: out = relu?(acc + bias)            (the Keras layer's bias + activation)
 * epilogue 1: BatchNorm(inference)/ReLU backward of the producing layer:
 *             out = (aux > 0) ? acc * e_scale[c] : 0, column sums of acc, acc*aux, masked acc
 *             per m-tile into partials[m_tiles][3][Cout] (see veex_colsum_finalize)
 * epilogue 2: raw split-K slabs out[nsplit][...] (sum with veex_slab_sum) */
typedef struct VeexGemmDesc {
  const float* in;       /* [N][Hin][Win][Cin] */
  const float* weight;
  float* out;            /* [N][Hout][Wout][Cout] */
  const float* in_scale; /* optional BatchNorm affine of the producer, applied to in-bounds loads */
  const float* in_shift;
  const float* bias;     /* epilogue 0, optional */
  const float* aux;      /* epilogue 1 */
  const float* e_scale;  /* epilogue 1, optional */
  float* partials;       /* epilogue 1, optional */
  int32_t N, Hin, Win, Cin, Hout, Wout, Cout;
  int32_t kh, kw, stride, pad_begin;
  int32_t mode;
  int32_t affine_channels; /* power of two; index = (ci [+ pixel*Cin]) & (affine_channels-1) */
  int32_t affine_spatial;  /* 1: BatchNorm ran over the flattened (h,w,c) vector (models.py:263-268) */
  int32_t epilogue, relu;
  int32_t e_channels;      /* epilogue 1: e_scale index = col & (e_channels-1) */
  int32_t nsplit;          /* epilogue 2 */
  const void* in_bf16;     /* optional bf16 consumer copy of `in` with its BatchNorm already applied (an out_bf16 of
                              the producer): read instead of in / in_scale / in_shift by the bf16 kernel */
  void* out_bf16;          /* optional: also write bf16(out_scale[c]*out + out_shift[c]) (epilogue 0; this layer's
                              BatchNorm, nullable = plain rounding) or bf16(out) (epilogue 1) */
  const float* out_scale; const float* out_shift;
  int32_t out_channels;    /* power of two; out_scale index = col & (out_channels-1) */
  int32_t struct_size;     /* sizeof(VeexGemmDesc) of the caller's header, or 0 = this header's.  Fields that start at
                              or beyond struct_size are taken as NULL instead of being read (a caller built against
                              the version-1 header, which ended after weight_bf16, passes a shorter struct).
                              Zero-initialise the whole descriptor before filling it. */
  const void* weight_bf16; /* optional packed bf16 weights [tap][Cout][Cin] (veex_pack_weights_bf16): selects the
                              bf16-MFMA kernel (fp32 accumulate) when Cin % 32 == 0 and N*Hout*Wout >= 4096 */
  const float* weight_t;   /* optional, mode 0, fp32 engine: the same weights as [tap][Cout][Cin] (veex_transpose_weights_f32);
                              same result bit for bit, the kernel's weight fragment then is one 16-byte LDS read */
  float* out_bn;           /* optional, epilogue 0: also write the fp32 copy out_scale[c]*out + out_shift[c] (this layer's
                              BatchNorm applied once by the producer): the fp32 GEMMs that consume the activation then
                              take it as `in` / `small` / `big` WITHOUT in_scale / in_shift -- same values bit for bit,
                              8 of 15 vector instructions less per K chunk of the consumer */
  int32_t weight_planes;   /* 3: weight_bf16 holds the three-term bf16 split of the fp32 weights, [plane][tap][Cout][Cin]
                              (veex_pack_weights_bf16x3): the launch computes an fp32 result -- operands split into three
                              bf16 terms, six partial products per multiply, fp32 accumulate -- on the bf16 matrix cores
                              (DESIGN.md 3.5).  0 / 1: weight_bf16 is the single rounded copy (bf16 engine). */
} VeexGemmDesc;
int veex_gather_gemm(const VeexGemmDesc* d, veexStream_t stream);
/* bf16 copy of one layer's kernel for veex_gather_gemm: src fp32 [taps][rows][cols] ->
 * dst bf16 [taps][rows][cols] (transpose=0) or [taps][cols][rows] (transpose=1).  Keras Conv2D kernels
 * (kh,kw,Cin,Cout) need transpose=1 for the forward (mode 0) and 0 for the input gradient (mode 1);
 * Conv2DTranspose kernels (kh,kw,Cout,Cin) the other way round. */
int veex_pack_weights_bf16(const float* src, void* dst, int taps, int rows, int cols, int transpose, veexStream_t s);
/* the three-term split of the same copy: dst bf16 [3][taps][..][..], src = plane 0 + plane 1 + plane 2 to 2^-24 relative */
int veex_pack_weights_bf16x3(const float* src, void* dst, int taps, int rows, int cols, int transpose, veexStream_t s);
/* fp32 transposed copy of one layer's kernel: dst[tap][c][r] = src[tap][r][c] (VeexGemmDesc::weight_t) */
int veex_transpose_weights_f32(const float* src, float* dst, int taps, int rows, int cols, veexStream_t s);
/* rows of the partials slab an epilogue-1 launch of this descriptor writes */
int veex_gather_gemm_m_tiles(const VeexGemmDesc* d);

/* Weight gradient (tape.gradient wrt kernel, models.py:166):
 *   dW[tap][cb][cs] = sum_{n,py,px} big(n, stride*py+ky-pad_begin, stride*px+kx-pad_begin, cb) * small(n,py,px,cs)
 * Conv2D: big = layer input, small = d(pre-activation); Conv2DTranspose: big = d(pre-act), small = input.
 * out: [nsplit][kh*kw*Cb][Cs] slabs (nsplit==1: the gradient). rows_per_split multiple of 16. */
typedef struct VeexWgradDesc {
  const float* big;
  const float* small;
  float* out;
  const float* big_scale; const float* big_shift;     /* optional producer BatchNorm affine */
  const float* small_scale; const float* small_shift;
  int32_t big_affine_channels, big_affine_spatial, small_affine_channels, small_affine_spatial;
  int32_t N, Hb, Wb, Cb, Hs, Ws, Cs;
  int32_t kh, kw, stride, pad_begin;
  int32_t nsplit, rows_per_split;
  int32_t bf16;            /* 1: round both operands to bf16 while staging, bf16 MFMA, fp32 accumulate
                              (rows_per_split multiple of 32, Cb % 4 == 0, Cs % 4 == 0);
                              2: fp32 result on the bf16 MFMA -- both operands split into three bf16 terms, six partial
                              products per multiply, fp32 accumulate (kh*kw*Cb >= 65, else the fp32 MFMA) */
  const void* big_bf16;    /* optional bf16 consumer copies of big / small (BatchNorm already applied to the */
  const void* small_bf16;  /* activation side): both given + bf16=1 -> copy kernel (rows_per_split % 64 == 0) */
} VeexWgradDesc;
int veex_wgrad_gemm(const VeexWgradDesc* d, veexStream_t stream);
/* The row split veex_net_train_fwd_bwd would use for this launch (fills nsplit / rows_per_split from the
 * geometry and d->bf16; the caller sizes `out` as nsplit * kh*kw*Cb * Cs floats).  Host only. */
int veex_wgrad_plan(const VeexWgradDesc* d, int32_t* nsplit, int32_t* rows_per_split);

/* conv1 of the encoder, Conv2D(64,3,strides=2,'same',relu) on the 40x40x1 map (models.py:219-221). */
int veex_conv1_fwd(const float* x, const float* w, const float* bias, float* out, int batch, veexStream_t s);
/* same, also writing the bf16 consumer copy bf16(bn_scale*out + bn_shift) that a bf16 conv2 reads (nullable) */
int veex_conv1_fwd_bf16(const float* x, const float* w, const float* bias, float* out, const float* bn_scale,
                        const float* bn_shift, void* out_bf16, int batch, veexStream_t s);
/* same with both optional consumer copies: out_bn = fp32 bn_scale*out + bn_shift (what an fp32 conv2 and its weight
 * gradient read without an affine prologue), out_bf16 as above */
int veex_conv1_fwd_copies(const float* x, const float* w, const float* bias, float* out, const float* bn_scale,
                          const float* bn_shift, float* out_bn, void* out_bf16, int batch, veexStream_t s);
int veex_conv1_wgrad_blocks(int batch);
int veex_conv1_wgrad(const float* x, const float* dpre, float* slab /*[blocks][576]*/, int batch, veexStream_t s);
/* d/dx through conv1 (attribution; no reference call site); mul != NULL gives gradient*input */
int veex_conv1_dgrad(const float* dpre, const float* w, const float* mul, float* dx, int batch, veexStream_t s);

/* deconv6 = Conv2DTranspose(1,2,'valid',sigmoid) (models.py:300-301) fused with
 * keras.losses.binary_crossentropy + reduce_sum(axis=(1,2)) (models.py:156-160).
 * Any of logits/recon/dlogit/rec_b/dlsum_b/target may be NULL. dlogit = (sigmoid-target)*inv_batch. */
int veex_deconv6_bce_fwd(const float* a5, const float* scale, const float* shift, const float* w6, const float* b6,
                         const float* target, float* logits, float* recon, float* dlogit, float* rec_b,
                         float* dlsum_b, int batch, float inv_batch, veexStream_t s);
/* same with the form of the cross-entropy selectable: VEEX_BCE_LOGITS (what veex_deconv6_bce_fwd computes) or
 * VEEX_BCE_PROBS, keras.backend.binary_crossentropy on the clipped probabilities -- see veex_net_set_semantics */
int veex_deconv6_bce_fwd_form(const float* a5, const float* scale, const float* shift, const float* w6, const float* b6,
                              const float* target, float* logits, float* recon, float* dlogit, float* rec_b,
                              float* dlsum_b, int batch, float inv_batch, int bce_form, veexStream_t s);
/* backward of deconv6 fused with BatchNorm/ReLU backward of deconv5; slab [batch][7][64]:
 * rows 0-3 dW6 partials, rows 4-6 the three column sums of veex_colsum_finalize */
int veex_deconv6_bwd(const float* dlogit, const float* a5, const float* scale, const float* shift, const float* w6,
                     float* dpre5, float* slab, int batch, veexStream_t s);

/* same, also writing bf16(dpre5) for the bf16 input- and weight-gradient GEMMs of deconv5 (nullable) */
int veex_deconv6_bwd_bf16(const float* dlogit, const float* a5, const float* scale, const float* shift, const float* w6,
                          float* dpre5, void* dpre5_bf16, float* slab, int batch, veexStream_t s);

/* sample() (models.py:10-22) fused with the analytic KL term (models.py:161-162).
 * mulv = [B][2L] (z_mean | z_log_var). use_philox=0: eps is read (injected epsilon);
 * use_philox=1: eps is generated (Philox-4x32-10 + Box-Muller, counter = global element index)
 * and written back for the backward pass. */
int veex_reparam_kl_fwd(const float* mulv, float* eps, float* z, float* kl_b, int batch, int latent, int use_philox,
                        unsigned long long seed, unsigned long long first_sample, veexStream_t s);
/* same; `counter` (nullable) is a DEVICE word that the kernel adds to first_sample when it runs: a launch captured
 * into a HIP graph then draws fresh epsilon on every replay (veex_net_set_sample_counter).  veex_counter_add:
 * counter[0] += inc on the stream (one thread; the step's "samples seen" bookkeeping inside a captured graph). */
int veex_reparam_kl_fwd_ctr(const float* mulv, float* eps, float* z, float* kl_b, int batch, int latent, int use_philox,
                            unsigned long long seed, unsigned long long first_sample,
                            const unsigned long long* counter, veexStream_t s);
int veex_counter_add(unsigned long long* counter, long long inc, veexStream_t s);
int veex_reparam_kl_bwd(const float* dz, const float* mulv, const float* eps, float* dmulv, int batch, int latent,
                        float inv_batch, veexStream_t s);

/* BatchNormalization in inference mode (models.py:222 etc.): scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale, rstd = 1/sqrt(var+eps) */
int veex_bn_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float* scale,
                    float* shift, float* rstd, int n, float eps, veexStream_t s);
int veex_bnrelu_bwd_rows_chunks(int rows);
int veex_bnrelu_bwd_rows(const float* g, const float* aux, const float* e_scale, float* out, float* partials,
                         int rows, int cols, int e_channels, veexStream_t s);
int veex_colsum_finalize(const float* partials, int T, long long row_stride, int ncols, int nch, const float* e_scale,
                         const float* mean, const float* rstd, float* dbias, float* dgamma, float* dbeta,
                         veexStream_t s);
int veex_slab_sum(const float* slab, int nsplit, long long stride, long long n, float* out, const float* bias,
                  int ncols, int relu, veexStream_t s);

/* keras.optimizers.legacy.Adam (train.py:9,137,193): p -= alpha_t*m/(sqrt(v)+eps), alpha_t =
 * lr*sqrt(1-b2^t)/(1-b1^t); step[0] (device int) is incremented; gscale multiplies g (1/world). */
int veex_adam(float* p, const float* g, float* m, float* v, long long n, int* step, float* alpha, double lr,
              double gscale, double b1, double b2, double eps, veexStream_t s);

/* Either side of the path.  Bit-identical to the reference's NumPy float32 expressions.
 * veex_minmax: global min / max of x[n] -> minmax[2] (partials: veex_minmax_partial_blocks(n)*2 floats);
 * veex_normalize: (x - min) / (max - min), train.py:99-100;  veex_denormalize: x*(max-min)+min,
 * signal_reconstruction.py:92-95;  veex_electrode_readout: out[b][e] = maps[b][pix[e]] (optionally
 * denormalized, minmax nullable), the pixel-per-electrode gather of indian_functions.py:146-160
 * (pix[e] = row*width + col; the host validates 0 <= pix[e] < map_elems). */
int veex_minmax_partial_blocks(long long n);
int veex_minmax(const float* x, long long n, float* partials, float* minmax, veexStream_t s);
/* veex_minmax for callers that cannot allocate -- code under HIP-graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  It has no counterpart in the reference (whose NumPy min/max allocate freely).  Same
 * kernels and bit-identical result as veex_minmax; the per-block partials live in a 16 KB file-scope device array of
 * the library (one copy per GPU, held until the process ends) instead of a caller buffer.  Limits: n <= 2^32
 * floats per call (larger n is rejected with VEEX_EINVAL); calls that overlap on ONE GPU (different streams) share
 * the array and race -- serialise them; the first call on a device resolves the array's address and must not be
 * made under stream capture.  The CURRENT device of the calling thread must be the GPU that owns x and the stream
 * (each GPU has its own copy of the array at its own address; the copy is looked up by the current device, and a
 * device pointer x that belongs to another GPU is refused with VEEX_EINVAL).  Calls no hipMalloc / hipFree. */
int veex_minmax_static(const float* x, long long n, float* minmax, veexStream_t s);
int veex_normalize(const float* x, const float* minmax, float* out, long long n, veexStream_t s);
int veex_denormalize(const float* x, const float* minmax, float* out, long long n, veexStream_t s);
int veex_electrode_readout(const float* maps, const int* pix, const float* minmax, float* out, int batch,
                           int map_elems, int n_electrodes, veexStream_t s);

/* Latent statistics, the step right after the path in analysis.py.  z: DEVICE [N][L] fp32
 * (z_mean as the encoder wrote it), one column per latent unit.
 * veex_column_quantiles: np.quantile(z, q, axis=0) as NumPy 1.24 (the reference's pin) computes it for a
 *   float32 array and Python-float orders -- 'linear' method, fp32 neighbour difference, float64 weight and
 *   result (analysis.py:243-244,298).  q: HOST array of nq <= 24 orders in [0,1]; out: DEVICE [nq][L] float64;
 *   scratch: DEVICE, veex_column_quantiles_scratch_bytes(nq, L) bytes.  Exact order statistics by radix select.
 * veex_band_outside_counts: counts[j*nb + k] = #{i : z[i][j] <= lo[j*nb+k] or z[i][j] >= hi[j*nb+k]} with
 *   float64 compares -- the TP (blink rows) / FP (no-blink rows) counters of histogram_25_75 and auc_roc
 *   (analysis.py:255-264,323-333); FN / TN are N minus these.  lo, hi: DEVICE [L][nb] float64.
 * veex_column_medians: np.median(z[:, j]) of every column as NumPy computes it for a float32 column -- the fp32
 *   mean of the two middle order statistics (the middle one for odd N) -- the value latent masking writes into
 *   a unit (analysis.py:574-577).  out: DEVICE [L] fp32; scratch: veex_column_quantiles_scratch_bytes(1, L). */
long long veex_column_quantiles_scratch_bytes(int nq, int L);
int veex_column_medians(const float* z, int N, int L, float* out, void* scratch, veexStream_t s);
int veex_column_quantiles(const float* z, int N, int L, const double* q, int nq, double* out, void* scratch,
                          veexStream_t s);
int veex_band_outside_counts(const float* z, int N, int L, int nb, const double* lo, const double* hi, int* counts,
                             veexStream_t s);

/* ------------------------------------------------------------------------------------------
 * Network level: the whole Encoder / Decoder / VAE.train_step graph as one enqueue.
 * ---------------------------------------------------------------------------------------- */
typedef struct VeexNet VeexNet; /* host-side plan only; owns no device memory */

VeexNet* veex_net_create(int latent_dim, int variant);
void veex_net_destroy(VeexNet* net);
long long veex_net_param_floats(const VeexNet* net);  /* trainable flat buffer (multiple of 4) */
long long veex_net_state_floats(const VeexNet* net);  /* BatchNorm moving mean | var */
int veex_net_num_tensors(const VeexNet* net);
/* tensor i in Keras Model.get_weights() order (train.py:287, analysis.py:127): encoder trainable, encoder
 * BatchNorm moving statistics, decoder trainable, decoder moving statistics ([KERAS-SEM], parity unpinned; the
 * metric trackers' scalars Keras appends are not tensors of the net). offset is into the trainable buffer
 * (trainable=1) or the state buffer (trainable=0). For the VAE the two latent heads are stored
 * as one [100][2L] matrix: part = 0 whole tensor, 1 = columns [0,L), 2 = columns [L,2L). */
int veex_net_tensor_info(const VeexNet* net, int i, char* name, int name_cap, long long* offset, int* ndim,
                         long long* shape4, int* trainable, int* part);
long long veex_net_workspace_floats(const VeexNet* net, int batch);
/* named workspace regions (activations "act.<layer>", gradients "dpre.<layer>", "z", "mulv", "eps", ...) for tests */
long long veex_net_ws_offset(const VeexNet* net, int batch, const char* name, long long* nfloats);

/* arithmetic of the conv / deconv contractions of a network: exact fp32 MFMA (default) or bf16 MFMA with
 * fp32 accumulation, fp32 storage and fp32 master weights (north_star "fp32 reference plus bf16") */
#define VEEX_COMPUTE_F32 0
#define VEEX_COMPUTE_BF16 1
/* fp32 results on the bf16 matrix cores: every operand of the large conv / deconv GEMMs is split into three bf16 terms
 * and each product accumulated in fp32 from six partial products.  Same tolerance class as
 * VEEX_COMPUTE_F32 (the parity tests run both at 1e-5); layers the split kernel does not win on stay on the fp32 MFMA. */
#define VEEX_COMPUTE_F32X3 2
int veex_net_set_compute(VeexNet* net, int compute);

/* The two Keras 2.13 behaviours of models.py:152-160 that nothing in the reference tree can pin (parity unpinned;
 * SURVEY Appendix A.3 / A.4), each with its alternative one switch away.  Defaults = the survey's reading.
 *  bn_mode  VEEX_BN_INFERENCE: BatchNormalization normalises with its moving statistics inside train_step too (the
 *           reference never forwards `training`);  VEEX_BN_BATCH: batch mean / biased variance, gradients through the
 *           statistics, moving statistics updated with momentum 0.99 by veex_net_train_fwd_bwd (which then WRITES
 *           the state buffer); encode / decode / forward / attribute keep using the moving statistics.  fp32 engine.
 *  bce_form VEEX_BCE_LOGITS: sigmoid_cross_entropy_with_logits (Keras finds the sigmoid's cached logits);
 *           VEEX_BCE_PROBS: -(t log(p + 1e-7) + (1 - t) log(1 - p + 1e-7)), p = clip(sigmoid(a), 1e-7, 1 - 1e-7). */
#define VEEX_BN_INFERENCE 0
#define VEEX_BN_BATCH 1
#define VEEX_BCE_LOGITS 0
#define VEEX_BCE_PROBS 1
int veex_net_set_semantics(VeexNet* net, int bn_mode, int bce_form);

/* Optional second stream (caller-owned, same device): veex_net_train_fwd_bwd then runs every weight-gradient GEMM
 * and its slab reduction there, concurrently with the input-gradient chain on the main stream, and makes the main
 * stream wait for it before returning (fork/join with eight events created by THIS call, so the step itself
 * creates nothing and can run under stream capture).
 * NULL (default) = everything on the one stream passed to the call. */
int veex_net_set_side_stream(VeexNet* net, veexStream_t side);

/* Data parallelism in two buckets (no reference call site: the reference is single-process).  The gradients of the
 * decoder's kernels and biases -- one contiguous range of the flat buffer, veex_net_grad_bucket -- are final long
 * before veex_net_train_fwd_bwd's last kernel: with a caller-owned event set here, the call records it (on the side
 * stream when there is one, else on the call's stream) at the point where that range is complete, so that the
 * caller can start its all-reduce on a communication stream (hipStreamWaitEvent) while the encoder's backward pass
 * still runs, and reduce the rest after the call.  The gradients are the same bits either way.  NULL = off. */
int veex_net_set_bucket_event(VeexNet* net, void* hip_event);
int veex_net_grad_bucket(const VeexNet* net, long long* first_float, long long* n_floats);

/* Optional device-side Philox sample counter (no reference call site: tf.random.normal is unseeded): when set, the
 * whole-step entry points draw epsilon for sample index *device_counter + first_sample + b instead of
 * first_sample + b, reading the word at RUN time -- so a train step captured into a HIP graph (a fixed small batch,
 * where enqueueing ~60 launches costs the host more than the GPU needs to run them) replays with fresh epsilon.
 * NULL (default) = first_sample by value only. */
int veex_net_set_sample_counter(VeexNet* net, const unsigned long long* device_counter);

/* Divisor of the batch-mean loss and of its gradients (models.py:156-163 reduce_mean over the batch).  0 (default):
 * the batch of the call.  Data parallelism with unequal shards sets global_batch / world on every rank, so that the
 * SUM all-reduce of the flat gradient buffer followed by veex_adam's gscale = 1/world is the mean over the GLOBAL
 * batch (no reference call site: the reference is single-process). */
int veex_net_set_batch_divisor(VeexNet* net, double divisor);

#define VEEX_F_PHILOX 1      /* generate epsilon on device (else read it from the eps argument) */
#define VEEX_F_ZMEAN_DECODE 2 /* decode z_mean instead of the sampled z (not the reference's VAE.call) */

/* Encoder.call (models.py:244-253) / EncoderStandard.call (models_standard.py:110-116).
 * outputs (nullable): z_mean [B][L], z_log_var [B][L], z [B][L] (AE: z = bottleneck, others unused) */
int veex_net_encode(const VeexNet* net, const float* params, const float* state, const float* x, float* eps_io,
                    int batch, int flags, unsigned long long seed, unsigned long long first_sample, float* ws,
                    long long ws_floats, float* z_mean, float* z_log_var, float* z, veexStream_t s);
/* Decoder.call (models.py:303-312): z [B][L] -> recon [B][40][40][1] (and/or logits) */
int veex_net_decode(const VeexNet* net, const float* params, const float* state, const float* z, int batch, float* ws,
                    long long ws_floats, float* recon, float* logits, veexStream_t s);
/* VAE.call / test_step (models.py:135-138,182-207): forward + losses. losses[3] = total, rec, kl;
 * metrics[4] (nullable) += losses, count */
int veex_net_forward(const VeexNet* net, const float* params, const float* state, const float* x, float* eps_io,
                     int batch, int flags, unsigned long long seed, unsigned long long first_sample, float* ws,
                     long long ws_floats, float* recon, float* losses, float* metrics, veexStream_t s);
/* VAE.train_step up to tape.gradient (models.py:148-166): forward, losses, all gradients into
 * grads (same layout as params). The optimizer step is veex_adam on the flat buffers, after the
 * caller's (optional) all-reduce of grads. */
int veex_net_train_fwd_bwd(const VeexNet* net, const float* params, const float* state, const float* x, float* eps_io,
                           int batch, int flags, unsigned long long seed, unsigned long long first_sample, float* ws,
                           long long ws_floats, float* grads, float* losses, float* metrics, veexStream_t s);
/* Gradient attribution (north_star item 3; no reference call site): out[u][b][40][40] =
 * d z_mean[b][units[u]] / d x[b]  (AE: the bottleneck unit), times x when grad_x_input. */
int veex_net_attribute(const VeexNet* net, const float* params, const float* state, const float* x, int batch,
                       const int* host_units, int n_units, int grad_x_input, float* ws, long long ws_floats,
                       float* out, veexStream_t s);

#ifdef __cplusplus
}
#endif
#endif /* VEEX_H */
