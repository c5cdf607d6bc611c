// Host-side plan of the reference network (models.py:211-312, models_standard.py:81-175) and the
// whole-step entry points of the C-ABI: one call enqueues every kernel of Encoder.call,
// Decoder.call, VAE.train_step (forward + loss + tape.gradient) or the attribution backward on
// the caller's stream.  No allocation, no synchronisation: all device memory is the caller's.
#include "veex_internal.h"
#include "../../include/veex.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void veex_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* veex_last_error(void) { return g_err; }

// Tuning switches from the environment.  Each variable is read ONCE per process, at its first use, and
// frozen: a change of the environment between veex_net_workspace_floats and the run cannot make the plan and the
// launch disagree.  The A/B tools flip a switch inside one process through veexi_debug_reload_env().
struct EnvVal { bool set; int v; };
static std::mutex g_env_mu;
static std::map<std::string, EnvVal> g_env;
int veex_env_int(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  auto it = g_env.find(name);
  if (it == g_env.end()) {
    const char* e = getenv(name);
    EnvVal v; v.set = (e != nullptr && *e); v.v = v.set ? atoi(e) : 0;
    it = g_env.emplace(name, v).first;
  }
  return it->second.set ? it->second.v : dflt;
}
static int g_env_gen = 0;
int veex_env_generation(void) { std::lock_guard<std::mutex> lk(g_env_mu); return g_env_gen; }   // bumps when the switches are re-read: cached plans are stale
extern "C" void veexi_debug_reload_env(void) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  g_env.clear();
  ++g_env_gen;
}
extern "C" int veex_abi_version(void) { return VEEX_ABI_VERSION; }

extern "C" int veexi_deconv6_bwd_raw(const float*, const float*, const float*, const float*, const float*, float*, float*, int, hipStream_t);
extern "C" int veexi_loss_finalize(const float*, const float*, const float*, int, double, float*, float*, float*, hipStream_t);
extern "C" int veexi_attr_seed(const float*, const float*, float*, int, int, int, int, int, hipStream_t);

#define TRY(x)            \
  do {                    \
    int rc__ = (x);       \
    if (rc__ != 0) return rc__; \
  } while (0)

static inline int pow2_or_zero(int c) { return (c > 0 && (c & (c - 1)) == 0); }

// ------------------------------------------------------------------------------------------
// operator-level wrappers (descriptor -> internal params)
// ------------------------------------------------------------------------------------------
static int fill_gg(const VeexGemmDesc* d, GGParams& p) {
  if (d == nullptr) { veex_set_error("null descriptor"); return VEEX_EINVAL; }
  memset(&p, 0, sizeof(p));
  // fields the caller's header does not have (struct_size, include/veex.h) are not read
  const size_t have = (d->struct_size > 0) ? (size_t)d->struct_size : sizeof(VeexGemmDesc);
  if (have < offsetof(VeexGemmDesc, weight_bf16) || have > sizeof(VeexGemmDesc)) { veex_set_error("gather_gemm: struct_size %d is not a VeexGemmDesc of any version", d->struct_size); return VEEX_EINVAL; }
  const bool has_wbf = have >= offsetof(VeexGemmDesc, weight_bf16) + sizeof(void*);
  const bool has_wt = have >= offsetof(VeexGemmDesc, weight_t) + sizeof(void*);
  const bool has_obn = have >= offsetof(VeexGemmDesc, out_bn) + sizeof(void*);
  const bool has_wpl = have >= offsetof(VeexGemmDesc, weight_planes) + sizeof(int32_t);
  p.A = d->in; p.W = d->weight; p.Wbf = has_wbf ? d->weight_bf16 : nullptr; p.Wt = (has_wt && d->mode == 0) ? d->weight_t : nullptr; p.Out = d->out;
  p.wplanes = has_wpl ? d->weight_planes : 0;
  p.Abf = d->in_bf16; p.OutBf = d->out_bf16; p.o_scale = d->out_scale; p.o_shift = d->out_shift;
  p.OutBn = (has_obn && d->epilogue == VEEX_EPI_FWD) ? d->out_bn : nullptr;
  if (p.OutBn != nullptr && p.o_scale == nullptr) { veex_set_error("gather_gemm: out_bn needs out_scale / out_shift"); return VEEX_EINVAL; }
  if (p.o_scale != nullptr) {
    if (!pow2_or_zero(d->out_channels) || p.o_shift == nullptr) { veex_set_error("gather_gemm: out_channels must be a power of two and out_shift non-null"); return VEEX_EINVAL; }
    p.o_cmask = d->out_channels - 1;
  }
  p.a_scale = d->in_scale; p.a_shift = d->in_shift;
  p.bias = d->bias; p.aux = d->aux; p.e_scale = d->e_scale; p.partials = d->partials;
  p.Nb = d->N; p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin;
  p.Hout = d->Hout; p.Wout = d->Wout; p.Cout = d->Cout;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pb = d->pad_begin;
  p.mode = d->mode;
  if (p.a_scale != nullptr) {
    if (!pow2_or_zero(d->affine_channels) || p.a_shift == nullptr) {
      veex_set_error("gather_gemm: affine_channels must be a power of two and in_shift non-null"); return VEEX_EINVAL;
    }
    p.aff_mask = d->affine_channels - 1;
  }
  p.aff_spatial = d->affine_spatial;
  p.epi = d->epilogue; p.relu = d->relu;
  if (p.epi == VEEX_EPI_BWD) {
    if (p.aux == nullptr) { veex_set_error("gather_gemm: epilogue 1 needs aux"); return VEEX_EINVAL; }
    if (p.e_scale != nullptr && !pow2_or_zero(d->e_channels)) {
      veex_set_error("gather_gemm: e_channels must be a power of two"); return VEEX_EINVAL;
    }
    p.epi_cmask = (p.e_scale != nullptr) ? d->e_channels - 1 : 0;
  }
  if (p.mode != 0 && p.mode != 1) { veex_set_error("gather_gemm: mode must be 0 or 1"); return VEEX_EINVAL; }
  if (p.A == nullptr || p.W == nullptr || p.Out == nullptr) { veex_set_error("gather_gemm: null operand"); return VEEX_EINVAL; }
  p.nsplit = d->nsplit > 0 ? d->nsplit : 1;
  if (p.epi != VEEX_EPI_RAW && p.nsplit != 1) { veex_set_error("gather_gemm: nsplit > 1 needs epilogue 2"); return VEEX_EINVAL; }
  const int cpt = (p.Cin + VEEX_BK - 1) / VEEX_BK;
  const int nchunks = p.kh * p.kw * cpt;
  p.chunks_per_split = (nchunks + p.nsplit - 1) / p.nsplit;
  p.slab_stride = (long long)p.Nb * p.Hout * p.Wout * p.Cout;
  return 0;
}

extern "C" int veex_gather_gemm(const VeexGemmDesc* d, veexStream_t stream) {
  GGParams p;
  TRY(fill_gg(d, p));
  return veex_launch_gather_gemm(p, stream);
}

int veex_hg_describe(const GGParams& p, long long* out8);   // halo_gemm_bf16.hip
// measurement aid: the halo-tile plan of this launch (0 = the halo kernel does not take it)
extern "C" int veexi_debug_hg_plan(const VeexGemmDesc* d, long long* out8) {
  GGParams p;
  if (fill_gg(d, p) != 0 || out8 == nullptr) return -1;
  return veex_hg_describe(p, out8);
}

extern "C" int veex_gather_gemm_m_tiles(const VeexGemmDesc* d) {
  GGParams p;
  if (fill_gg(d, p) != 0) return -1;
  return veex_gg_m_tiles(p);
}

static int fill_wg(const VeexWgradDesc* d, WGParams& p) {
  if (d == nullptr) { veex_set_error("null descriptor"); return VEEX_EINVAL; }
  memset(&p, 0, sizeof(p));
  p.Big = d->big; p.Small = d->small; p.Out = d->out;
  p.b_scale = d->big_scale; p.b_shift = d->big_shift; p.s_scale = d->small_scale; p.s_shift = d->small_shift;
  if (p.b_scale != nullptr) {
    if (!pow2_or_zero(d->big_affine_channels)) { veex_set_error("wgrad: big_affine_channels must be a power of two"); return VEEX_EINVAL; }
    p.b_aff_mask = d->big_affine_channels - 1;
  }
  if (p.s_scale != nullptr) {
    if (!pow2_or_zero(d->small_affine_channels)) { veex_set_error("wgrad: small_affine_channels must be a power of two"); return VEEX_EINVAL; }
    p.s_aff_mask = d->small_affine_channels - 1;
  }
  p.b_aff_spatial = d->big_affine_spatial; p.s_aff_spatial = d->small_affine_spatial;
  p.Nb = d->N; p.Hb = d->Hb; p.Wb = d->Wb; p.Cb = d->Cb; p.Hs = d->Hs; p.Ws = d->Ws; p.Cs = d->Cs;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pb = d->pad_begin;
  p.R = p.Nb * p.Hs * p.Ws;
  p.M = p.kh * p.kw * p.Cb;
  p.nsplit = d->nsplit; p.rows_per_split = d->rows_per_split;
  p.x3 = (d->bf16 == 2) ? 1 : 0;                 // 2: split-fp32 (fp32 results on the bf16 MFMA)
  p.bf16 = p.x3 ? 0 : d->bf16;
  p.BigBf = d->big_bf16; p.SmallBf = d->small_bf16;
  if (p.Big == nullptr || p.Small == nullptr || p.Out == nullptr) { veex_set_error("wgrad: null operand"); return VEEX_EINVAL; }
  return 0;
}

extern "C" int veex_wgrad_gemm(const VeexWgradDesc* d, veexStream_t stream) {
  WGParams p;
  TRY(fill_wg(d, p));
  return veex_launch_wgrad(p, stream);
}

// ------------------------------------------------------------------------------------------
// network plan
// ------------------------------------------------------------------------------------------
static const float BN_EPS = 1e-3f;  // keras BatchNormalization default

struct TensorInfo {
  std::string name;
  long long offset;
  int ndim;
  long long shape[4];
  int trainable;
  int part;  // 0 whole, 1/2 = first/second half of the columns of the [100][2L] head matrix
};

struct ConvLayer {  // one Conv2D / Conv2DTranspose / Dense with optional BatchNorm after its ReLU
  const char* name;
  int Cin, Cout, k, stride, pb, Hin, Hout;
  long long w, b;   // offsets into the trainable buffer
  int bn;           // channel offset into the BN arrays, -1 = none
  int bnC;
};

struct VeexNet {
  int L, variant, H1;
  int compute;                  // VEEX_COMPUTE_F32 / VEEX_COMPUTE_BF16
  double batch_div = 0.0;       // divisor of the batch-mean loss; 0 = the batch of the call (veex_net_set_batch_divisor)
  hipEvent_t bucket_ev = nullptr;   // optional caller-owned event: "the decoder's gradients are final" (veex_net_set_bucket_event)
  const unsigned long long* sample_counter = nullptr;   // optional device word added to first_sample (veex_net_set_sample_counter)
  int bn_mode = VEEX_BN_INFERENCE, bce_form = VEEX_BCE_LOGITS;   // the two unpinned Keras readings (veex_net_set_semantics)
  hipStream_t side = nullptr;   // optional caller-owned second stream for the weight-gradient GEMMs
  mutable hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  mutable int ev_next = 0;      // fork / join events (host objects, created by veex_net_set_side_stream)
  // timing probe (bench.py's roofline object): two timing events recorded around ONE forward launch of the next
  // whole-step call, on the launch stream (veexi_debug_probe_arm / veexi_debug_probe_ms)
  mutable hipEvent_t probe_ev[2] = {nullptr, nullptr};
  mutable std::string probe_layer;
  mutable int probe_state = 0;  // 0 off, 1 armed, 2 recorded
  long long n_params, n_state;
  int n_bn;                     // total BatchNorm channels
  long long gamma_off, beta_off;  // trainable buffer
  std::vector<TensorInfo> tensors;
  ConvLayer conv[3];    // encoder convs
  ConvLayer dense;      // encoder dense (as 5x5 VALID conv over (5,5,256))
  long long wh, bh;     // VAE heads [100][2L], [2L]
  ConvLayer ddense;     // decoder dense L -> 4096
  ConvLayer dc[6];      // decoder deconvs
};

static long long align4(long long v) { return (v + 3) & ~3LL; }
static long long align64(long long v) { return (v + 63) & ~63LL; }

extern "C" VeexNet* veex_net_create(int L, int variant) {
  if (L < 1 || L > 4096 || (variant != VEEX_VARIANT_VAE && variant != VEEX_VARIANT_AE)) {
    veex_set_error("net_create: bad latent_dim %d / variant %d", L, variant);
    return nullptr;
  }
  VeexNet* n = new VeexNet();
  n->L = L; n->variant = variant; n->compute = VEEX_COMPUTE_F32;
  n->H1 = (variant == VEEX_VARIANT_VAE) ? 100 : L;
  long long off = 0;
  int bnoff = 0;
  auto take = [&](long long cnt) { long long o = off; off = align4(off + cnt); return o; };
  auto takebn = [&](int c) { int o = bnoff; bnoff += c; return o; };

  // encoder convs (models.py:219-235): k3 s2 SAME -> pad_before 0 on even inputs
  const int ecin[3] = {1, 64, 128}, ecout[3] = {64, 128, 256}, ehin[3] = {40, 20, 10};
  const char* ename[3] = {"enc.conv1", "enc.conv2", "enc.conv3"};
  for (int i = 0; i < 3; ++i) {
    ConvLayer& c = n->conv[i];
    c.name = ename[i]; c.Cin = ecin[i]; c.Cout = ecout[i]; c.k = 3; c.stride = 2; c.pb = 0;
    c.Hin = ehin[i]; c.Hout = ehin[i] / 2;
    c.w = take(9LL * c.Cin * c.Cout); c.b = take(c.Cout);
    c.bn = takebn(c.Cout); c.bnC = c.Cout;
  }
  {
    ConvLayer& c = n->dense;  // Dense(6400 -> H1, relu) (models.py:238 / models_standard.py:108)
    c.name = "enc.dense"; c.Cin = 256; c.Cout = n->H1; c.k = 5; c.stride = 1; c.pb = 0; c.Hin = 5; c.Hout = 1;
    c.w = take(6400LL * n->H1); c.b = take(n->H1); c.bn = -1; c.bnC = 0;
  }
  if (variant == VEEX_VARIANT_VAE) { n->wh = take(100LL * 2 * L); n->bh = take(2LL * L); }
  else { n->wh = n->bh = -1; }
  {
    ConvLayer& c = n->ddense;  // Dense(L -> 4096, relu) + BN(4096) (models.py:263-266)
    c.name = "dec.dense"; c.Cin = L; c.Cout = 4096; c.k = 1; c.stride = 1; c.pb = 0; c.Hin = 1; c.Hout = 1;
    c.w = take((long long)L * 4096); c.b = take(4096); c.bn = takebn(4096); c.bnC = 4096;
  }
  // decoder deconvs (models.py:270-301); pb = crop-at-start
  const int dcin[6] = {256, 256, 128, 128, 64, 64}, dcout[6] = {256, 128, 128, 64, 64, 1};
  const int dk[6] = {3, 3, 3, 3, 3, 2}, ds[6] = {2, 1, 2, 1, 2, 1}, dpb[6] = {0, 1, 0, 0, 0, 0};
  const int dhin[6] = {4, 8, 8, 17, 19, 39}, dhout[6] = {8, 8, 17, 19, 39, 40};
  const char* dname[6] = {"dec.deconv1", "dec.deconv2", "dec.deconv3", "dec.deconv4", "dec.deconv5", "dec.deconv6"};
  for (int i = 0; i < 6; ++i) {
    ConvLayer& c = n->dc[i];
    c.name = dname[i]; c.Cin = dcin[i]; c.Cout = dcout[i]; c.k = dk[i]; c.stride = ds[i]; c.pb = dpb[i];
    c.Hin = dhin[i]; c.Hout = dhout[i];
    c.w = take((long long)dk[i] * dk[i] * dcin[i] * dcout[i]); c.b = take(dcout[i]);
    if (i < 5) { c.bn = takebn(c.Cout); c.bnC = c.Cout; } else { c.bn = -1; c.bnC = 0; }
  }
  n->n_bn = bnoff;
  n->gamma_off = take(bnoff);
  n->beta_off = take(bnoff);
  n->n_params = off;
  n->n_state = 2LL * bnoff;  // mean | var

  // Tensor table in Keras Model.get_weights() order ([KERAS-SEM], parity unpinned): Model.weights
  // concatenates layer.variables of the tracked sub-layers, and a Layer's variables are its trainable weights (layer
  // creation order) followed by its non-trainable ones -- so: encoder trainable (18 / AE 14), encoder moving
  // statistics (6), decoder trainable (26), decoder moving statistics (12).  The flat trainable / state buffers
  // keep their own layout; only this table's order is the interchange order.
  auto add = [&](const std::string& name, long long o, std::initializer_list<long long> shp, int tr, int part = 0) {
    TensorInfo t; t.name = name; t.offset = o; t.ndim = (int)shp.size(); t.trainable = tr; t.part = part;
    int i = 0; for (long long s : shp) t.shape[i++] = s; for (; i < 4; ++i) t.shape[i] = 1;
    n->tensors.push_back(t);
  };
  auto add_bn_tr = [&](const ConvLayer& c) {
    add(std::string(c.name) + ".gamma", n->gamma_off + c.bn, {c.bnC}, 1);
    add(std::string(c.name) + ".beta", n->beta_off + c.bn, {c.bnC}, 1);
  };
  auto add_bn_st = [&](const ConvLayer& c) {
    add(std::string(c.name) + ".mean", c.bn, {c.bnC}, 0);
    add(std::string(c.name) + ".var", (long long)n->n_bn + c.bn, {c.bnC}, 0);
  };
  for (int i = 0; i < 3; ++i) {
    const ConvLayer& c = n->conv[i];
    add(std::string(c.name) + ".kernel", c.w, {3, 3, c.Cin, c.Cout}, 1);
    add(std::string(c.name) + ".bias", c.b, {c.Cout}, 1);
    add_bn_tr(c);
  }
  add("enc.dense.kernel", n->dense.w, {6400, n->H1}, 1);
  add("enc.dense.bias", n->dense.b, {n->H1}, 1);
  if (variant == VEEX_VARIANT_VAE) {
    add("enc.z_mean.kernel", n->wh, {100, L}, 1, 1);
    add("enc.z_mean.bias", n->bh, {L}, 1, 1);
    add("enc.z_log_var.kernel", n->wh, {100, L}, 1, 2);
    add("enc.z_log_var.bias", n->bh, {L}, 1, 2);
  }
  for (int i = 0; i < 3; ++i) add_bn_st(n->conv[i]);
  add("dec.dense.kernel", n->ddense.w, {L, 4096}, 1);
  add("dec.dense.bias", n->ddense.b, {4096}, 1);
  add_bn_tr(n->ddense);
  for (int i = 0; i < 6; ++i) {
    const ConvLayer& c = n->dc[i];
    add(std::string(c.name) + ".kernel", c.w, {c.k, c.k, c.Cout, c.Cin}, 1);
    add(std::string(c.name) + ".bias", c.b, {c.Cout}, 1);
    if (c.bn >= 0) add_bn_tr(c);
  }
  add_bn_st(n->ddense);
  for (int i = 0; i < 5; ++i) add_bn_st(n->dc[i]);
  return n;
}

extern "C" void veex_net_destroy(VeexNet* n) {
  if (n == nullptr) return;
  for (int i = 0; i < 8; ++i) if (n->ev[i] != nullptr) (void)hipEventDestroy(n->ev[i]);
  for (int i = 0; i < 2; ++i) if (n->probe_ev[i] != nullptr) (void)hipEventDestroy(n->probe_ev[i]);
  delete n;
}
// Measurement aid, not part of the product API: time ONE forward launch (layer = "enc.conv2" ... "dec.deconv5") inside
// the next veex_net_train_fwd_bwd / veex_net_forward call with HIP events on the stream the kernel is launched on.
extern "C" int veexi_debug_probe_arm(VeexNet* n, const char* layer) {
  if (n == nullptr || layer == nullptr) { veex_set_error("probe_arm: null argument"); return VEEX_EINVAL; }
  for (int i = 0; i < 2; ++i)
    if (n->probe_ev[i] == nullptr && hipEventCreate(&n->probe_ev[i]) != hipSuccess) {
      n->probe_ev[i] = nullptr; veex_set_error("probe_arm: cannot create a timing event"); return VEEX_ELAUNCH;
    }
  n->probe_layer = layer;
  n->probe_state = 1;
  return 0;
}
extern "C" int veexi_debug_probe_ms(VeexNet* n, float* ms) {
  if (n == nullptr || ms == nullptr) { veex_set_error("probe_ms: null argument"); return VEEX_EINVAL; }
  if (n->probe_state != 2) { veex_set_error("probe_ms: the armed layer was not launched"); n->probe_state = 0; return VEEX_EINVAL; }
  n->probe_state = 0;
  if (hipEventSynchronize(n->probe_ev[1]) != hipSuccess ||
      hipEventElapsedTime(ms, n->probe_ev[0], n->probe_ev[1]) != hipSuccess) { veex_set_error("probe_ms: event query failed"); return VEEX_ELAUNCH; }
  return 0;
}
extern "C" int veex_net_set_side_stream(VeexNet* n, veexStream_t side) {
  if (n == nullptr) { veex_set_error("set_side_stream: null net"); return VEEX_EINVAL; }
  // the fork / join events are created HERE, not on first use inside veex_net_train_fwd_bwd: that call may run under
  // stream capture, and it creates nothing (host objects included)
  if (side != nullptr) {
    for (int i = 0; i < 8; ++i) {
      if (n->ev[i] == nullptr && hipEventCreateWithFlags(&n->ev[i], hipEventDisableTiming) != hipSuccess) {
        n->ev[i] = nullptr;
        veex_set_error("set_side_stream: cannot create the fork / join events");
        return VEEX_ELAUNCH;
      }
    }
  }
  n->side = side;
  return 0;
}
extern "C" int veex_net_set_compute(VeexNet* n, int compute) {
  if (n == nullptr || (compute != VEEX_COMPUTE_F32 && compute != VEEX_COMPUTE_BF16 && compute != VEEX_COMPUTE_F32X3)) { veex_set_error("set_compute: bad arguments"); return VEEX_EINVAL; }
  if (compute == VEEX_COMPUTE_BF16 && n->bn_mode == VEEX_BN_BATCH) { veex_set_error("set_compute: batch-statistics BatchNorm runs on the fp32 engine only"); return VEEX_EINVAL; }
  n->compute = compute;
  return 0;
}
extern "C" int veex_net_set_semantics(VeexNet* n, int bn_mode, int bce_form) {
  if (n == nullptr || (bn_mode != VEEX_BN_INFERENCE && bn_mode != VEEX_BN_BATCH) || (bce_form != VEEX_BCE_LOGITS && bce_form != VEEX_BCE_PROBS)) {
    veex_set_error("set_semantics: bad arguments"); return VEEX_EINVAL;
  }
  if (bn_mode == VEEX_BN_BATCH && n->compute == VEEX_COMPUTE_BF16) { veex_set_error("set_semantics: batch-statistics BatchNorm runs on the fp32 engine only"); return VEEX_EINVAL; }
  n->bn_mode = bn_mode; n->bce_form = bce_form;
  return 0;
}
extern "C" int veex_net_set_bucket_event(VeexNet* n, void* event) {
  if (n == nullptr) { veex_set_error("set_bucket_event: null net"); return VEEX_EINVAL; }
  n->bucket_ev = static_cast<hipEvent_t>(event);
  return 0;
}
extern "C" int veex_net_grad_bucket(const VeexNet* n, long long* first, long long* count) {
  if (n == nullptr || first == nullptr || count == nullptr) { veex_set_error("grad_bucket: null argument"); return VEEX_EINVAL; }
  *first = n->ddense.w;                      // dec.dense kernel ... dec.deconv6 bias: contiguous in the flat buffer
  *count = n->gamma_off - n->ddense.w;
  return 0;
}
extern "C" int veex_net_set_sample_counter(VeexNet* n, const unsigned long long* device_counter) {
  if (n == nullptr) { veex_set_error("set_sample_counter: null net"); return VEEX_EINVAL; }
  n->sample_counter = device_counter;
  return 0;
}
extern "C" int veex_net_set_batch_divisor(VeexNet* n, double divisor) {
  if (n == nullptr || !(divisor >= 0.0) || divisor > 1e12) { veex_set_error("set_batch_divisor: bad arguments"); return VEEX_EINVAL; }
  n->batch_div = divisor;
  return 0;
}
extern "C" long long veex_net_param_floats(const VeexNet* n) { return n ? n->n_params : -1; }
extern "C" long long veex_net_state_floats(const VeexNet* n) { return n ? n->n_state : -1; }
extern "C" int veex_net_num_tensors(const VeexNet* n) { return n ? (int)n->tensors.size() : -1; }
extern "C" int veex_net_tensor_info(const VeexNet* n, int i, char* name, int cap, long long* offset, int* ndim,
                                    long long* shape4, int* trainable, int* part) {
  if (n == nullptr || i < 0 || i >= (int)n->tensors.size()) { veex_set_error("tensor_info: bad index"); return VEEX_EINVAL; }
  const TensorInfo& t = n->tensors[i];
  if (name != nullptr && cap > 0) { strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (offset) *offset = t.offset;
  if (ndim) *ndim = t.ndim;
  if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
  if (trainable) *trainable = t.trainable;
  if (part) *part = t.part;
  return 0;
}

// ------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------
struct WsEntry { std::string name; long long off, n; };

struct Ws {
  std::vector<WsEntry> e;
  long long total = 0;
  long long add(const std::string& name, long long n) {
    WsEntry x; x.name = name; x.off = total; x.n = n;
    e.push_back(x);
    total = align64(total + n);
    return x.off;
  }
  long long get(const char* name, long long* n = nullptr) const {
    for (const auto& x : e) if (x.name == name) { if (n) *n = x.n; return x.off; }
    return -1;
  }
};

struct SplitPlan { int nsplit; int per; };  // per = rows_per_split (wgrad) or unused

static SplitPlan wgrad_split(const WGParams& pin) {
  int bm, bn;
  const int tiles = veex_wg_tiles(pin, &bm, &bn);
  const int bk = (pin.bf16 == 2) ? 64 : pin.bf16 ? 32 : VEEX_BK;   // reduction rows per K chunk of the kernel that will run
  const int chunks = (pin.R + bk - 1) / bk;
  // Blocks per launch the row split aims at: what fits on the chip at once (3 per CU), equally long, so the launch
  // lasts as long as the fullest CU -- never exceed the target (774 blocks = a 4th block on six CUs = +33 %).
  // (the bf16 copy kernel holds 72 KB of LDS: 2 blocks per CU)
  // (bf16 copy kernel, 72 KB of LDS: ONE block per CU measured best inside the step -- 1.209 -> 1.176 ms at batch 512
  //  against two per CU, -0.8 % at batch 2 048: half the slab bytes, and the input-gradient launches keep the other slot)
  // (transposing-read bf16 kernel, wgrad_gemm_bf16t.hip, 64 KB + row table: two per CU -- 512 and 768 measure the same
  //  inside the step, 256 is 6 % slower on it; profiles/r03_bf16_wgrad_tr.txt)
  const bool bt = pin.bf16 == 2 && veex_tune_int("VEEX_WGBT", 1) != 0;
  // (split-fp32 kernel, wgrad_gemm_x3.hip: 768 = three per CU is its best stand-alone, 512 is 1.1 % better on the step --
  //  the input-gradient launches of the main stream keep a slot; 384 / 640 / 1 024 are worse; profiles/r03_x3_wgrad.txt)
  const int target = veex_tune_int("VEEX_WG_BLOCKS", (pin.bf16 == 2) ? (bt ? 512 : 256) : (pin.x3 ? 512 : 768));
  int ns = veex_tune_int("VEEX_WG_FLOOR", 1) ? target / tiles : (target + tiles - 1) / tiles;
  const int maxs = (chunks / 4 > 0) ? chunks / 4 : 1;
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  const int cps = (chunks + ns - 1) / ns;
  SplitPlan s; s.per = cps * bk;
  if (!pin.bf16 && s.per > VEEX_WG3_MAX_ROWS) s.per = VEEX_WG3_MAX_ROWS;   // keep the lean kernel's row table in LDS
  if (bt && s.per > VEEX_WGT_MAX_ROWS) s.per = VEEX_WGT_MAX_ROWS;          // and the transposing-read bf16 kernel's
  s.nsplit = (pin.R + s.per - 1) / s.per;
  return s;
}

extern "C" int veex_wgrad_plan(const VeexWgradDesc* d, int32_t* nsplit, int32_t* rows_per_split) {
  if (d == nullptr || nsplit == nullptr || rows_per_split == nullptr) { veex_set_error("wgrad_plan: null argument"); return VEEX_EINVAL; }
  WGParams p; memset(&p, 0, sizeof(p));
  p.Nb = d->N; p.Hb = d->Hb; p.Wb = d->Wb; p.Cb = d->Cb; p.Hs = d->Hs; p.Ws = d->Ws; p.Cs = d->Cs;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pb = d->pad_begin;
  p.R = p.Nb * p.Hs * p.Ws; p.M = p.kh * p.kw * p.Cb;
  p.x3 = (d->bf16 == 2) ? 1 : 0;
  p.bf16 = (d->bf16 && !p.x3) ? ((d->big_bf16 != nullptr && d->small_bf16 != nullptr) ? 2 : 1) : 0;
  if (p.R < 1 || p.M < 1 || p.Cs < 1) { veex_set_error("wgrad_plan: bad geometry"); return VEEX_EINVAL; }
  const SplitPlan s = wgrad_split(p);
  *nsplit = s.nsplit; *rows_per_split = s.per;
  return 0;
}

static int gg_split(int mtiles, int Cout, int chunks) {
  const int ntiles = (Cout + 63) / 64;
  const int tiles = mtiles * ntiles;
  int ns = (512 + tiles - 1) / tiles;
  const int maxs = (chunks / 4 > 0) ? chunks / 4 : 1;
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  const int cps = (chunks + ns - 1) / ns;
  return (chunks + cps - 1) / cps;
}

struct Plan {
  Ws ws;
  long long bn_scale, bn_shift, bn_rstd, bn_mu, bn_coef, bn_scratch;
  long long act_conv[3], act_h, mulv, eps, z;
  long long act_dd, act_dc[5], dlogit, rec_b, kl_b, dlsum_b;
  long long d_conv[3], d_h, dmulv, dz, d_dd, g_dd, d_dc[5];
  long long slab, slab_side, partials, partials2, losses_tmp;
  long long slab_n, partials_n, partials2_n;
  long long wpack[7][2];
  long long wpackf[7];   // fp32 engine: transposed fp32 copy [tap][Cout_gemm][Cin_gemm] of the big kernels for their mode-0 launches
  // bf16 consumer copies (bf16 compute only; -1 otherwise): BatchNorm-applied activations and gradients
  // fp32 engine: BatchNorm-applied fp32 copies of the activations a GEMM consumes (GGParams::OutBn); -1 in bf16 mode
  long long bn_conv[3], bn_dd, bn_dc[4];
  long long bf_act_conv[2], bf_act_dd, bf_act_dc[4], bf_d_dc[5], bf_d_conv[3];   // bf16 weight copies (floats offset): layers conv2, conv3, deconv1..5; [0] as stored, [1] transposed
};

static long long act_floats(const ConvLayer& c, int B) { return (long long)B * c.Hout * c.Hout * c.Cout; }

// descriptors shared by planning (sizes) and execution
static void wg_desc_deconv(const VeexNet* n, int i, int B, WGParams& p) {
  const ConvLayer& c = n->dc[i];
  memset(&p, 0, sizeof(p));
  // 2: both operands come as bf16 consumer copies.  The split-fp32 engine runs wgrad_gemm_x3.hip (row-major planes read
  // through the transposing LDS read; two earlier split kernels that transposed while staging were 20-75 % slower than
  // wgrad_gemm_v3.hip, then on a par with it: profiles/r03_x3_wgrad.txt)
  p.bf16 = (n->compute == VEEX_COMPUTE_BF16) ? 2 : 0;
  p.x3 = (n->compute == VEEX_COMPUTE_F32X3) ? 1 : 0;
  p.Nb = B; p.Hb = c.Hout; p.Wb = c.Hout; p.Cb = c.Cout; p.Hs = c.Hin; p.Ws = c.Hin; p.Cs = c.Cin;
  p.kh = c.k; p.kw = c.k; p.stride = c.stride; p.pb = c.pb;
  p.R = B * c.Hin * c.Hin; p.M = c.k * c.k * c.Cout;
}
static void wg_desc_conv(const ConvLayer& c, int B, WGParams& p, int bf16 = 0, int x3 = 0) {
  memset(&p, 0, sizeof(p));
  p.bf16 = bf16; p.x3 = x3;
  p.Nb = B; p.Hb = c.Hin; p.Wb = c.Hin; p.Cb = c.Cin; p.Hs = c.Hout; p.Ws = c.Hout; p.Cs = c.Cout;
  p.kh = c.k; p.kw = c.k; p.stride = c.stride; p.pb = c.pb;
  p.R = B * c.Hout * c.Hout; p.M = c.k * c.k * c.Cin;
}

static long long gg_tiles_for(int B, int Hout, int Cout, int mode, int stride, int pb, int bm = 64) {
  GGParams p; memset(&p, 0, sizeof(p));
  p.Nb = B; p.Hout = Hout; p.Wout = Hout; p.Cout = Cout; p.mode = mode; p.stride = stride; p.pb = pb;
  return veex_gg_m_tiles_bm(p, bm);   // smallest tile the launcher may pick: an upper bound on the tile count
}

// the seven MFMA-bound layers that have a bf16 path: 0,1 = enc.conv2/3, 2..6 = dec.deconv1..5
static const ConvLayer& big_layer(const VeexNet* n, int i) { return (i < 2) ? n->conv[i + 1] : n->dc[i - 2]; }

static void make_plan(const VeexNet* n, int B, Plan& P) {
  Ws& w = P.ws;
  const int L = n->L, H1 = n->H1;
  P.bn_scale = w.add("bn.scale", n->n_bn);
  P.bn_shift = w.add("bn.shift", n->n_bn);
  P.bn_rstd = w.add("bn.rstd", n->n_bn);
  P.bn_mu = w.add("bn.mu", n->n_bn);                     // batch means (VEEX_BN_BATCH)
  P.bn_coef = w.add("bn.coef", 3LL * 4096);              // backward coefficients of one layer (largest: dec.dense)
  {
    long long d = 0;
    auto upd2 = [&](long long rows, int C) { const long long v = veex_bn_batch_scratch_doubles(rows, C); if (v > d) d = v; };
    for (int i = 0; i < 3; ++i) upd2((long long)B * n->conv[i].Hout * n->conv[i].Hout, n->conv[i].Cout);
    upd2(B, 4096);
    for (int i = 0; i < 5; ++i) upd2((long long)B * n->dc[i].Hout * n->dc[i].Hout, n->dc[i].Cout);
    P.bn_scratch = w.add("bn.scratch", 2 * d);           // doubles, as floats
  }
  for (int i = 0; i < 3; ++i) P.act_conv[i] = w.add(std::string("act.") + n->conv[i].name, act_floats(n->conv[i], B));
  P.act_h = w.add("act.enc.dense", (long long)B * H1);
  if (n->variant == VEEX_VARIANT_VAE) {
    P.mulv = w.add("mulv", (long long)B * 2 * L);
    P.eps = w.add("eps", (long long)B * L);
    P.z = w.add("z", (long long)B * L);
  } else {
    P.mulv = P.eps = -1;
    P.z = P.act_h;
  }
  P.act_dd = w.add("act.dec.dense", (long long)B * 4096);
  for (int i = 0; i < 5; ++i) P.act_dc[i] = w.add(std::string("act.") + n->dc[i].name, act_floats(n->dc[i], B));
  P.dlogit = w.add("dlogit", (long long)B * 1600);
  P.rec_b = w.add("rec_b", B);
  P.kl_b = w.add("kl_b", B);
  P.dlsum_b = w.add("dlsum_b", B);
  P.losses_tmp = w.add("losses_tmp", 4);
  for (int i = 0; i < 3; ++i) P.d_conv[i] = w.add(std::string("dpre.") + n->conv[i].name, act_floats(n->conv[i], B));
  P.d_h = w.add("dpre.enc.dense", (long long)B * H1);
  if (n->variant == VEEX_VARIANT_VAE) P.dmulv = w.add("dmulv", (long long)B * 2 * L); else P.dmulv = -1;
  P.dz = w.add("dz", (long long)B * L);
  P.g_dd = w.add("g.dec.dense", (long long)B * 4096);
  P.d_dd = w.add("dpre.dec.dense", (long long)B * 4096);
  for (int i = 0; i < 5; ++i) P.d_dc[i] = w.add(std::string("dpre.") + n->dc[i].name, act_floats(n->dc[i], B));

  // scratch sizes: slabs (split-K / wgrad / conv1 / deconv6) and column partials
  long long slab = 0, part = 0;
  auto upd = [](long long& m, long long v) { if (v > m) m = v; };
  WGParams wp;
  for (int i = 0; i < 5; ++i) { wg_desc_deconv(n, i, B, wp); upd(slab, (long long)wgrad_split(wp).nsplit * wp.M * wp.Cs); }
  for (int i = 1; i < 3; ++i) { wg_desc_conv(n->conv[i], B, wp, n->compute == VEEX_COMPUTE_BF16 ? 2 : 0, n->compute == VEEX_COMPUTE_F32X3 ? 1 : 0); upd(slab, (long long)wgrad_split(wp).nsplit * wp.M * wp.Cs); }
  wg_desc_conv(n->dense, B, wp); upd(slab, (long long)wgrad_split(wp).nsplit * wp.M * wp.Cs);
  wg_desc_conv(n->ddense, B, wp); upd(slab, (long long)wgrad_split(wp).nsplit * wp.M * wp.Cs);
  if (n->variant == VEEX_VARIANT_VAE) {
    memset(&wp, 0, sizeof(wp));
    wp.Nb = B; wp.Hb = wp.Wb = wp.Hs = wp.Ws = 1; wp.Cb = 100; wp.Cs = 2 * L; wp.kh = wp.kw = 1; wp.stride = 1;
    wp.R = B; wp.M = 100;
    upd(slab, (long long)wgrad_split(wp).nsplit * wp.M * wp.Cs);
  }
  upd(slab, (long long)veex_conv1_wgrad_blocks(B) * 576);
  upd(slab, (long long)B * 448);
  {  // enc dense forward split-K and dz split-K
    const int mt = (int)gg_tiles_for(B, 1, H1, 0, 1, 0);
    upd(slab, (long long)gg_split(mt, H1, 25 * 16) * B * H1);
    const int mt2 = (int)gg_tiles_for(B, 1, L, 1, 1, 0);
    upd(slab, (long long)gg_split(mt2, L, 4096 / VEEX_BK) * B * L);
  }
  // partials: every producer of one backward pass keeps its own rows until the batched finalisation at the end of
  // the pass (take_partials): EPI_BWD launches (m_tiles * 3 * Cout), row-chunk launches, the deconv6 backward rows
  auto acc = [](long long& m, long long v) { m += (v + 63) & ~63LL; };
  for (int i = 1; i < 5; ++i)  // dgrad of deconv i+1 produces gradient of deconv i's output
    acc(part, gg_tiles_for(B, n->dc[i - 1].Hout, n->dc[i - 1].Cout, 0, 1, 0, 32) * 3 * n->dc[i - 1].Cout);
  acc(part, gg_tiles_for(B, 1, 6400, 1, 1, 0, 32) * 3 * 6400);
  acc(part, gg_tiles_for(B, 10, 128, 1, 2, 0, 32) * 3 * 128);
  acc(part, gg_tiles_for(B, 20, 64, 1, 2, 0, 32) * 3 * 64);
  acc(part, gg_tiles_for(B, 1, H1, 1, 1, 0, 32) * 3 * H1);
  acc(part, (long long)veex_bnrelu_bwd_rows_chunks(B) * 3 * 4096);
  acc(part, (long long)veex_bnrelu_bwd_rows_chunks(B) * 3 * (2 * L > H1 ? 2 * L : H1));
  acc(part, (long long)B * 448);
  P.slab_n = slab; P.partials_n = part;
  P.slab = w.add("scratch.slab", slab);
  P.slab_side = w.add("scratch.slab_side", slab);   // weight-gradient slabs when they run on the side stream
  P.partials = w.add("scratch.partials", part);
  P.partials2_n = (long long)VEEX_COLSUM_MAX * VEEX_PARTIALS2_N;
  P.partials2 = w.add("scratch.partials2", P.partials2_n);   // compacted partial rows (veex_colsum_batch)
  P.bf_act_conv[0] = P.bf_act_conv[1] = P.bf_act_dd = -1;
  for (int i = 0; i < 4; ++i) P.bf_act_dc[i] = -1;
  for (int i = 0; i < 5; ++i) P.bf_d_dc[i] = -1;
  for (int i = 0; i < 3; ++i) P.bf_d_conv[i] = -1;
  P.bn_conv[0] = P.bn_conv[1] = P.bn_conv[2] = P.bn_dd = -1;
  for (int i = 0; i < 4; ++i) P.bn_dc[i] = -1;
  if (n->compute != VEEX_COMPUTE_BF16) {
    for (int i = 0; i < 3; ++i) P.bn_conv[i] = w.add(std::string("bn.") + n->conv[i].name, act_floats(n->conv[i], B));
    P.bn_dd = w.add("bn.dec.dense", (long long)B * 4096);
    for (int i = 0; i < 4; ++i) P.bn_dc[i] = w.add(std::string("bn.") + n->dc[i].name, act_floats(n->dc[i], B));
  }
  if (n->compute == VEEX_COMPUTE_BF16) {
    auto half = [](long long elems) { return (elems + 1) / 2; };
    for (int i = 0; i < 2; ++i) P.bf_act_conv[i] = w.add(std::string("bf.act.") + n->conv[i].name, half(act_floats(n->conv[i], B)));
    P.bf_act_dd = w.add("bf.act.dec.dense", half((long long)B * 4096));
    for (int i = 0; i < 4; ++i) P.bf_act_dc[i] = w.add(std::string("bf.act.") + n->dc[i].name, half(act_floats(n->dc[i], B)));
    for (int i = 0; i < 5; ++i) P.bf_d_dc[i] = w.add(std::string("bf.dpre.") + n->dc[i].name, half(act_floats(n->dc[i], B)));
    for (int i = 1; i < 3; ++i) P.bf_d_conv[i] = w.add(std::string("bf.dpre.") + n->conv[i].name, half(act_floats(n->conv[i], B)));
  }
  for (int i = 0; i < 7; ++i) {
    const ConvLayer& l = big_layer(n, i);
    const long long halfs = (long long)l.k * l.k * l.Cin * l.Cout;
    const long long planes = (n->compute == VEEX_COMPUTE_F32X3) ? 3 : 1;   // split-fp32: three bf16 planes per packed copy
    P.wpack[i][0] = w.add(std::string("wpack.") + l.name + ".a", (planes * halfs + 1) / 2);
    P.wpack[i][1] = w.add(std::string("wpack.") + l.name + ".t", (planes * halfs + 1) / 2);
    P.wpackf[i] = w.add(std::string("wpackf.") + l.name, halfs);
  }
}

extern "C" long long veex_net_workspace_floats(const VeexNet* n, int B) {
  if (n == nullptr || B < 1) { veex_set_error("workspace_floats: bad arguments"); return -1; }
  Plan P; make_plan(n, B, P);
  return P.ws.total;
}

extern "C" long long veex_net_ws_offset(const VeexNet* n, int B, const char* name, long long* nfloats) {
  if (n == nullptr || B < 1 || name == nullptr) return -1;
  Plan P; make_plan(n, B, P);
  return P.ws.get(name, nfloats);
}

// ------------------------------------------------------------------------------------------
// execution helpers
// ------------------------------------------------------------------------------------------
struct Ctx {
  Ctx() { fin.n = 0; }
  hipStream_t side = nullptr;   // non-null: weight-gradient GEMMs are forked onto this stream
  bool forked = false;
  const VeexNet* n;
  const float* P;     // params
  const float* S;     // state (mean | var)
  float* ws;
  Plan plan;
  int B;
  hipStream_t st;
  long long part_cur = 0;       // bump cursor into scratch.partials: every producer of one pass keeps its own rows
  ColsumBatch fin;              // column-sum finalisations collected during the backward pass (flush_colsum)
  bool batch_bn = false;        // this call normalises with batch statistics (train_fwd_bwd under VEEX_BN_BATCH)
  bool have_wt_enc = false, have_wt_bwd = false;   // packed fp32 weight copies made by this call (run_transpose_weights)
  // divisor of the batch mean (models.py:156-163): the local batch, or the caller's (a rank's share of a global batch)
  double div() const { return (n->batch_div > 0.0) ? n->batch_div : (double)B; }
  float inv_div() const { return (n->batch_div > 0.0) ? (float)(1.0 / n->batch_div) : 1.0f / B; }
  const float* scale(int bn) const { return ws + plan.bn_scale + bn; }
  const float* shift(int bn) const { return ws + plan.bn_shift + bn; }
  const float* rstd(int bn) const { return ws + plan.bn_rstd + bn; }
  const float* mean(int bn) const { return batch_bn ? ws + plan.bn_mu + bn : S + bn; }
};

static int check_common(const VeexNet* n, const void* params, const void* state, int B, const float* ws,
                        long long ws_floats) {
  if (n == nullptr || params == nullptr || state == nullptr || ws == nullptr) { veex_set_error("null argument"); return VEEX_EINVAL; }
  if (B < 1 || B > (1 << 20)) { veex_set_error("bad batch %d", B); return VEEX_EINVAL; }
  const long long need = veex_net_workspace_floats(n, B);
  if (ws_floats < need) { veex_set_error("workspace too small: %lld floats given, %lld needed", ws_floats, need); return VEEX_ENOMEM; }
  return 0;
}

// bf16 copies of the seven big kernels from the fp32 master weights (once per call: the weights change every step)
// fp32 engine: transposed copies for the mode-0 launches of the lean kernel (the Conv2D forwards; with `bwd` also the
// Conv2DTranspose input gradients), one launch per call -- the weights change every step
static int run_transpose_weights(Ctx& c, bool enc, bool bwd) {
  // split-fp32 engine from batch 256 up: every mode-0 launch of the big layers runs on the split kernel (N*Hout*Wout >= 4096
  // for all of them), which reads its own packed planes -- nobody would read the transposed fp32 copies
  if (c.n->compute == VEEX_COMPUTE_F32X3 && c.B >= 256) { c.have_wt_enc = c.have_wt_bwd = false; return 0; }
  const float* src[7]; float* dst[7]; int taps[7], rows[7], cols[7];
  int k = 0;
  for (int i = 0; i < 7; ++i) {
    const bool conv = i < 2;
    if (conv ? !enc : !bwd) continue;
    const ConvLayer& l = big_layer(c.n, i);
    // Conv2D kernel (kh,kw,Cin,Cout): rows Cin; Conv2DTranspose kernel (kh,kw,Cout,Cin): rows Cout
    src[k] = c.P + l.w; dst[k] = c.ws + c.plan.wpackf[i]; taps[k] = l.k * l.k;
    rows[k] = conv ? l.Cin : l.Cout; cols[k] = conv ? l.Cout : l.Cin;
    ++k;
  }
  c.have_wt_enc = enc; c.have_wt_bwd = bwd;
  if (k == 0) return 0;
  return veex_transpose_weights_f32_multi(k, src, dst, taps, rows, cols, c.st);
}
// the packed fp32 copy of big layer i, if this call made it
static const float* packed_wt(const Ctx& c, int i) {
  if (c.n->compute == VEEX_COMPUTE_BF16) return nullptr;
  return ((i < 2) ? c.have_wt_enc : c.have_wt_bwd) ? c.ws + c.plan.wpackf[i] : nullptr;
}

int veex_pack_weights_bf16x3_multi(int n, const float* const* src, void* const* dst, const int* taps, const int* rows,
                                   const int* cols, const int* transpose, hipStream_t st);   // gather_gemm_x3.hip
static int run_pack_weights(Ctx& c) {
  if (c.n->compute != VEEX_COMPUTE_BF16 && c.n->compute != VEEX_COMPUTE_F32X3) return 0;
  const float* src[14]; void* dst[14]; int taps[14], rows[14], cols[14], tr[14];
  for (int i = 0; i < 7; ++i) {
    const ConvLayer& l = big_layer(c.n, i);
    // Conv2D kernels are (kh,kw,Cin,Cout), Conv2DTranspose kernels (kh,kw,Cout,Cin)
    for (int t = 0; t < 2; ++t) {
      const int k = 2 * i + t;
      src[k] = c.P + l.w; dst[k] = c.ws + c.plan.wpack[i][t]; taps[k] = l.k * l.k;
      rows[k] = (i < 2) ? l.Cin : l.Cout; cols[k] = (i < 2) ? l.Cout : l.Cin; tr[k] = t;
    }
  }
  if (c.n->compute == VEEX_COMPUTE_F32X3) return veex_pack_weights_bf16x3_multi(14, src, dst, taps, rows, cols, tr, c.st);
  return veex_pack_weights_bf16_multi(14, src, dst, taps, rows, cols, tr, c.st);
}
// packed weights [tap][Cout_gemm][Cin_gemm] for the forward (fwd=1) or input-gradient GEMM of big layer i
static const void* packed_w(const Ctx& c, int i, int fwd) {
  if (c.n->compute != VEEX_COMPUTE_BF16 && c.n->compute != VEEX_COMPUTE_F32X3) return nullptr;
  const int conv = (i < 2);
  const int which = (conv == (fwd != 0)) ? 1 : 0;   // conv fwd / deconv dgrad: transposed copy
  return c.ws + c.plan.wpack[i][which];
}

// pointer to a bf16 consumer copy region (nullptr in fp32 mode)
static void* bfp(const Ctx& c, long long off) { return (off < 0) ? nullptr : static_cast<void*>(c.ws + off); }
// pointer to a BatchNorm-applied fp32 copy (nullptr in bf16 mode, or with VEEX_BNCOPY=0: consumers then apply the affine)
// (nor with batch statistics: a producer cannot apply a normalisation whose statistics are its own output's)
static float* bnp(const Ctx& c, long long off) { return (off < 0 || c.batch_bn || !veex_tune_int("VEEX_BNCOPY", 1)) ? nullptr : c.ws + off; }

// VEEX_BN_BATCH: statistics of layer l's activation -> its scale / shift / rstd / mean slots, moving statistics updated
static int bn_batch_forward(Ctx& c, const ConvLayer& l, const float* act) {
  const VeexNet* n = c.n;
  const long long rows = (long long)c.B * l.Hout * l.Hout * l.Cout / l.bnC;   // dec.dense: Hout = 1, bnC = 4096 -> B rows
  float* S = const_cast<float*>(c.S);   // documented: veex_net_train_fwd_bwd writes the state buffer in this mode
  return veex_bn_batch_forward(act, rows, l.bnC, c.P + n->gamma_off + l.bn, c.P + n->beta_off + l.bn, BN_EPS, 0.99f,
                               /*unbiased=*/l.Hout > 1 || l.k > 1, c.ws + c.plan.bn_scale + l.bn, c.ws + c.plan.bn_shift + l.bn,
                               c.ws + c.plan.bn_rstd + l.bn, c.ws + c.plan.bn_mu + l.bn, S + l.bn, S + n->n_bn + l.bn,
                               reinterpret_cast<double*>(c.ws + c.plan.bn_scratch), c.st);
}
// VEEX_BN_BATCH: v = dL/d(BatchNorm output of layer l) -> dpre (may alias v) + dgamma / dbeta / dbias of layer l
static int bn_batch_backward(Ctx& c, const ConvLayer& l, const float* v, const float* act, float* dpre, float* G) {
  const VeexNet* n = c.n;
  const long long rows = (long long)c.B * l.Hout * l.Hout * l.Cout / l.bnC;
  return veex_bn_batch_backward(v, act, rows, l.bnC, c.P + n->gamma_off + l.bn, c.ws + c.plan.bn_mu + l.bn,
                                c.ws + c.plan.bn_rstd + l.bn, dpre, G + n->gamma_off + l.bn, G + n->beta_off + l.bn, G + l.b,
                                c.ws + c.plan.bn_coef, reinterpret_cast<double*>(c.ws + c.plan.bn_scratch), c.st);
}

static int run_bn_prepare(Ctx& c) {
  TRY(run_pack_weights(c));
  const VeexNet* n = c.n;
  return veex_bn_prepare(c.P + n->gamma_off, c.P + n->beta_off, c.S, c.S + n->n_bn, c.ws + c.plan.bn_scale,
                         c.ws + c.plan.bn_shift, c.ws + c.plan.bn_rstd, n->n_bn, BN_EPS, c.st);
}

// forward GEMM layer: in (with producer BN) -> out = relu?(conv + bias)
static int fwd_gemm(Ctx& c, const ConvLayer& l, int mode, const float* in, int in_bn, int in_bnC, int aff_spatial,
                    float* out, int relu, const void* wp = nullptr, const void* abf = nullptr, void* outbf = nullptr,
                    const float* wt = nullptr, const float* in_bncopy = nullptr, float* out_bncopy = nullptr) {
  GGParams p; memset(&p, 0, sizeof(p));
  p.A = in; p.W = c.P + l.w; p.Wbf = wp; p.Wt = wt; p.Out = out; p.bias = c.P + l.b;
  p.wplanes = (c.n->compute == VEEX_COMPUTE_F32X3) ? 3 : 0;
  p.Abf = abf; p.OutBf = outbf;
  if (l.bn >= 0) p.OutBn = out_bncopy;
  if ((outbf != nullptr || p.OutBn != nullptr) && l.bn >= 0) { p.o_scale = c.scale(l.bn); p.o_shift = c.shift(l.bn); p.o_cmask = l.bnC - 1; }
  if (in_bncopy != nullptr) { p.A = in_bncopy; in_bn = -1; }   // the producer already applied its BatchNorm
  if (in_bn >= 0) { p.a_scale = c.scale(in_bn); p.a_shift = c.shift(in_bn); p.aff_mask = in_bnC - 1; p.aff_spatial = aff_spatial; }
  p.Nb = c.B; p.Hin = l.Hin; p.Win = l.Hin; p.Cin = l.Cin; p.Hout = l.Hout; p.Wout = l.Hout; p.Cout = l.Cout;
  p.kh = l.k; p.kw = l.k; p.stride = l.stride; p.pb = l.pb; p.mode = mode;
  p.epi = VEEX_EPI_FWD; p.relu = relu; p.nsplit = 1;
  const bool probe = c.n->probe_state == 1 && c.n->probe_layer == l.name;
  if (probe && hipEventRecord(c.n->probe_ev[0], c.st) != hipSuccess) { veex_set_error("probe: event record failed"); return VEEX_ELAUNCH; }
  const int rc = veex_launch_gather_gemm(p, c.st);
  if (probe) {
    if (hipEventRecord(c.n->probe_ev[1], c.st) != hipSuccess) { veex_set_error("probe: event record failed"); return VEEX_ELAUNCH; }
    c.n->probe_state = 2;
  }
  if (rc == 0 && c.batch_bn && l.bn >= 0) return bn_batch_forward(c, l, out);
  return rc;
}

static int encoder_forward(Ctx& c, const float* x) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  float* ws = c.ws;
  TRY(veex_conv1_fwd_copies(x, c.P + n->conv[0].w, c.P + n->conv[0].b, ws + P.act_conv[0], c.scale(n->conv[0].bn),
                            c.shift(n->conv[0].bn), bnp(c, P.bn_conv[0]), bfp(c, P.bf_act_conv[0]), c.B, c.st));
  if (c.batch_bn) TRY(bn_batch_forward(c, n->conv[0], ws + P.act_conv[0]));
  TRY(fwd_gemm(c, n->conv[1], 0, ws + P.act_conv[0], n->conv[0].bn, 64, 0, ws + P.act_conv[1], 1, packed_w(c, 0, 1),
               bfp(c, P.bf_act_conv[0]), bfp(c, P.bf_act_conv[1]), packed_wt(c, 0), bnp(c, P.bn_conv[0]), bnp(c, P.bn_conv[1])));
  TRY(fwd_gemm(c, n->conv[2], 0, ws + P.act_conv[1], n->conv[1].bn, 128, 0, ws + P.act_conv[2], 1, packed_w(c, 1, 1),
               bfp(c, P.bf_act_conv[1]), nullptr, packed_wt(c, 1), bnp(c, P.bn_conv[1]), bnp(c, P.bn_conv[2])));
  {  // Dense(6400 -> H1, relu) as a 5x5 VALID conv, split-K over taps/channels
    const ConvLayer& l = n->dense;
    GGParams p; memset(&p, 0, sizeof(p));
    p.A = ws + P.act_conv[2]; p.W = c.P + l.w; p.Out = ws + P.slab;
    p.a_scale = c.scale(n->conv[2].bn); p.a_shift = c.shift(n->conv[2].bn); p.aff_mask = 255;
    if (const float* y = bnp(c, P.bn_conv[2])) { p.A = y; p.a_scale = p.a_shift = nullptr; p.aff_mask = 0; }
    p.Nb = c.B; p.Hin = 5; p.Win = 5; p.Cin = 256; p.Hout = 1; p.Wout = 1; p.Cout = l.Cout;
    p.kh = 5; p.kw = 5; p.stride = 1; p.pb = 0; p.mode = 0; p.epi = VEEX_EPI_RAW;
    const int chunks = 25 * 16;
    GGParams q = p;
    const int mt = veex_gg_m_tiles(q);
    p.nsplit = gg_split(mt, l.Cout, chunks);
    p.chunks_per_split = (chunks + p.nsplit - 1) / p.nsplit;
    p.slab_stride = (long long)c.B * l.Cout;
    if ((long long)p.nsplit * p.slab_stride > P.slab_n) { veex_set_error("internal: slab scratch too small (enc.dense)"); return VEEX_ENOMEM; }
    TRY(veex_launch_gather_gemm(p, c.st));
    TRY(veex_slab_sum(ws + P.slab, p.nsplit, p.slab_stride, p.slab_stride, ws + P.act_h, c.P + l.b, l.Cout, 1, c.st));
  }
  if (n->variant == VEEX_VARIANT_VAE) {  // both heads as one [100][2L] GEMM (models.py:240-241)
    GGParams p; memset(&p, 0, sizeof(p));
    p.A = ws + P.act_h; p.W = c.P + n->wh; p.Out = ws + P.mulv; p.bias = c.P + n->bh;
    p.Nb = c.B; p.Hin = p.Win = p.Hout = p.Wout = 1; p.Cin = 100; p.Cout = 2 * n->L;
    p.kh = p.kw = 1; p.stride = 1; p.mode = 0; p.epi = VEEX_EPI_FWD; p.relu = 0; p.nsplit = 1;
    TRY(veex_launch_gather_gemm(p, c.st));
  }
  return 0;
}

// decoder up to deconv5's activation
static int decoder_trunk(Ctx& c, const float* z) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  float* ws = c.ws;
  TRY(fwd_gemm(c, n->ddense, 0, z, -1, 0, 0, ws + P.act_dd, 1, nullptr, nullptr, bfp(c, P.bf_act_dd), nullptr, nullptr,
               bnp(c, P.bn_dd)));
  const float* in = ws + P.act_dd;
  const float* in_y = bnp(c, P.bn_dd);
  const void* in_bf = bfp(c, P.bf_act_dd);
  int in_bn = n->ddense.bn, in_bnC = 4096, spatial = 1;
  for (int i = 0; i < 5; ++i) {
    void* out_bf = (i < 4) ? bfp(c, P.bf_act_dc[i]) : nullptr;   // deconv5's consumer (deconv6) reads fp32
    float* out_y = (i < 4) ? bnp(c, P.bn_dc[i]) : nullptr;       // (and applies BatchNorm 5 itself)
    TRY(fwd_gemm(c, n->dc[i], 1, in, in_bn, in_bnC, spatial, ws + P.act_dc[i], 1, packed_w(c, i + 2, 1), in_bf, out_bf,
                 nullptr, in_y, out_y));
    in = ws + P.act_dc[i]; in_y = out_y; in_bf = out_bf; in_bn = n->dc[i].bn; in_bnC = n->dc[i].Cout; spatial = 0;
  }
  return 0;
}

static int deconv6_forward(Ctx& c, const float* target, float* logits, float* recon, float* dlogit, float* rec_b,
                           float* dlsum_b) {
  const VeexNet* n = c.n;
  return veex_deconv6_bce_fwd_form(c.ws + c.plan.act_dc[4], c.scale(n->dc[4].bn), c.shift(n->dc[4].bn), c.P + n->dc[5].w,
                                   c.P + n->dc[5].b, target, logits, recon, dlogit, rec_b, dlsum_b, c.B, c.inv_div(),
                                   n->bce_form, c.st);
}

static hipEvent_t next_event(const VeexNet* n) {
  const int i = n->ev_next;
  n->ev_next = (i + 1) & 7;
  return n->ev[i];   // created by veex_net_set_side_stream
}
// side stream waits for everything enqueued on the main stream so far
static int fork_side(Ctx& c) {
  hipEvent_t e = next_event(c.n);
  if (e == nullptr || hipEventRecord(e, c.st) != hipSuccess || hipStreamWaitEvent(c.side, e, 0) != hipSuccess) {
    veex_set_error("side stream fork failed"); return VEEX_ELAUNCH;
  }
  c.forked = true;
  return 0;
}
// main stream waits for the side stream (before the caller reads the gradients)
static int join_side(Ctx& c) {
  if (c.side == nullptr || !c.forked) return 0;
  hipEvent_t e = next_event(c.n);
  if (e == nullptr || hipEventRecord(e, c.side) != hipSuccess || hipStreamWaitEvent(c.st, e, 0) != hipSuccess) {
    veex_set_error("side stream join failed"); return VEEX_ELAUNCH;
  }
  c.forked = false;
  return 0;
}

// Weight gradient of one layer.  With a side stream the GEMM and its slab reduction run there,
// concurrently with the input-gradient chain on the main stream (they only share read-only inputs;
// measured -12..-14 % per dgrad+wgrad pair): fork after the producer of its operands, join at the end.
// layer: name for the timing probe ("<layer>.wgrad" arms the GEMM launch, on the stream it runs on)
static int run_wgrad(Ctx& c, WGParams& p, float* dW, const char* layer = nullptr) {
  const SplitPlan s = wgrad_split(p);
  p.nsplit = s.nsplit; p.rows_per_split = s.per;
  const long long n = (long long)p.M * p.Cs;
  hipStream_t st = c.st;
  float* slab = c.ws + c.plan.slab;
  if (c.side != nullptr) { TRY(fork_side(c)); st = c.side; slab = c.ws + c.plan.slab_side; }
  if (s.nsplit > 1 && (long long)s.nsplit * n > c.plan.slab_n) { veex_set_error("internal: slab scratch too small"); return VEEX_ENOMEM; }
  p.Out = (s.nsplit == 1) ? dW : slab;
  const bool probe = layer != nullptr && c.n->probe_state == 1 && c.n->probe_layer == std::string(layer) + ".wgrad";
  if (probe && hipEventRecord(c.n->probe_ev[0], st) != hipSuccess) { veex_set_error("probe: event record failed"); return VEEX_ELAUNCH; }
  TRY(veex_launch_wgrad(p, st));
  if (probe) {
    if (hipEventRecord(c.n->probe_ev[1], st) != hipSuccess) { veex_set_error("probe: event record failed"); return VEEX_ELAUNCH; }
    c.n->probe_state = 2;
  }
  if (s.nsplit == 1) return 0;
  return veex_slab_sum(slab, s.nsplit, n, n, dW, nullptr, 1, 0, st);
}

// The two GEMMs that consume d(pre-activation) of decoder layer i: its weight gradient and its input gradient
static void deconv_wgrad_desc(Ctx& c, int i, WGParams& w) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  const ConvLayer& prod = (i == 0) ? n->ddense : n->dc[i - 1];
  wg_desc_deconv(n, i, c.B, w);   // big = d(pre-act) of this layer, small = its input (with the producer's BN)
  w.Big = c.ws + P.d_dc[i]; w.Small = (i == 0) ? c.ws + P.act_dd : c.ws + P.act_dc[i - 1];
  w.BigBf = bfp(c, P.bf_d_dc[i]); w.SmallBf = (i == 0) ? bfp(c, P.bf_act_dd) : bfp(c, P.bf_act_dc[i - 1]);
  w.s_scale = c.scale(prod.bn); w.s_shift = c.shift(prod.bn); w.s_aff_mask = prod.bnC - 1; w.s_aff_spatial = (i == 0);
  if (const float* y = bnp(c, (i == 0) ? P.bn_dd : P.bn_dc[i - 1])) {   // the producer's BatchNorm-applied copy: no affine
    w.Small = y; w.s_scale = w.s_shift = nullptr; w.s_aff_mask = 0; w.s_aff_spatial = 0;
  }
}
static void deconv_dgrad_desc(Ctx& c, int i, GGParams& p) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  const ConvLayer& l = n->dc[i];
  memset(&p, 0, sizeof(p));
  p.A = c.ws + P.d_dc[i]; p.W = c.P + l.w; p.Wbf = packed_w(c, i + 2, 0); p.Wt = packed_wt(c, i + 2);
  p.wplanes = (c.n->compute == VEEX_COMPUTE_F32X3) ? 3 : 0;
  p.Abf = bfp(c, P.bf_d_dc[i]); p.OutBf = (i > 0) ? bfp(c, P.bf_d_dc[i - 1]) : nullptr;
  p.Nb = c.B; p.Hin = p.Win = l.Hout; p.Cin = l.Cout; p.Hout = p.Wout = l.Hin; p.Cout = l.Cin;
  p.kh = p.kw = l.k; p.stride = l.stride; p.pb = l.pb; p.mode = 0;
}
// bf16 engine: true when both of them read only the bf16 copy, so that the producer need not write the fp32 one
// (372 MB less to write per step at batch 512)
static bool dpre_dc_f32_unused(Ctx& c, int i) {
  if (c.n->compute != VEEX_COMPUTE_BF16 || !veex_tune_int("VEEX_BF_SKIPF32", 1)) return false;
  WGParams w; deconv_wgrad_desc(c, i, w);
  const SplitPlan s = wgrad_split(w);
  w.nsplit = s.nsplit; w.rows_per_split = s.per;
  if (!veex_wgrad_bf16c_ok(w)) return false;
  GGParams p; deconv_dgrad_desc(c, i, p);
  p.epi = (i > 0) ? VEEX_EPI_BWD : VEEX_EPI_FWD;
  return veex_gg_bf16_only(p) >= 1;
}

static float* take_partials(Ctx& c, long long nfloats) {
  if (c.part_cur + nfloats > c.plan.partials_n) { veex_set_error("internal: partials scratch too small"); return nullptr; }
  float* p = c.ws + c.plan.partials + c.part_cur;
  c.part_cur += (nfloats + 63) & ~63LL;
  return p;
}
static int defer_colsum(Ctx& c, const float* partials, int T, long long row_stride, int ncols, int nch, const float* e_scale,
                        const float* mean, const float* rstd, float* dbias, float* dgamma, float* dbeta) {
  if (c.fin.n >= VEEX_COLSUM_MAX) { veex_set_error("internal: column-sum table full"); return VEEX_EINVAL; }
  ColsumEntry& e = c.fin.e[c.fin.n++];
  memset(&e, 0, sizeof(e));
  e.partials = partials; e.T = T; e.row_stride = row_stride; e.ncols = ncols; e.nch = nch;
  e.e_scale = e_scale; e.mean = mean; e.rstd = rstd; e.dbias = dbias; e.dgamma = dgamma; e.dbeta = dbeta;
  if (!veex_tune_int("VEEX_COLSUM_DEFER", 1)) {   // A/B switch: finalise right away (two launches per entry)
    const int rc = veex_colsum_batch(c.fin, c.ws + c.plan.partials2, c.plan.partials2_n, c.st);
    c.fin.n = 0;
    return rc;
  }
  return 0;
}
static int flush_colsum(Ctx& c) {
  const int rc = veex_colsum_batch(c.fin, c.ws + c.plan.partials2, c.plan.partials2_n, c.st);
  c.fin.n = 0;
  return rc;
}

// dgrad GEMM with BN/ReLU-backward epilogue + finalisation of the producing layer's bias/gamma/beta grads
static int dgrad_bwd(Ctx& c, GGParams& p, const ConvLayer& prod, const float* aux, float* dpre, float* G, int want_param_grads,
                     bool f32_unused = false) {
  const VeexNet* n = c.n;
  if (c.batch_bn && prod.bn >= 0 && want_param_grads) {
    // batch statistics: the gradient wrt the BatchNorm output first (plain epilogue), then the backward through the
    // statistics, which needs the complete column sums before it can form a single element
    p.Out = dpre; p.epi = VEEX_EPI_FWD; p.relu = 0; p.bias = nullptr; p.nsplit = 1;
    TRY(veex_launch_gather_gemm(p, c.st));
    return bn_batch_backward(c, prod, dpre, aux, dpre, G);
  }
  p.Out = dpre; p.aux = aux; p.epi = VEEX_EPI_BWD; p.relu = 1; p.nsplit = 1;
  if (prod.bn >= 0) { p.e_scale = c.scale(prod.bn); p.epi_cmask = prod.bnC - 1; }
  if (f32_unused && veex_gg_bf16_only(p) == 2) p.Out = nullptr;   // only the bf16 copy of this gradient is read
  GGParams q = p;
  const int mt = veex_gg_m_tiles(q);
  p.partials = nullptr;
  if (want_param_grads) {
    p.partials = take_partials(c, (long long)mt * 3 * p.Cout);
    if (p.partials == nullptr) return VEEX_ENOMEM;
  }
  TRY(veex_launch_gather_gemm(p, c.st));
  if (!want_param_grads) return 0;
  const int nch = (prod.bn >= 0) ? prod.bnC : p.Cout;
  return defer_colsum(c, p.partials, mt, 3LL * p.Cout, p.Cout, nch,
                      prod.bn >= 0 ? c.scale(prod.bn) : nullptr, prod.bn >= 0 ? c.mean(prod.bn) : nullptr,
                      prod.bn >= 0 ? c.rstd(prod.bn) : nullptr, G + prod.b,
                      prod.bn >= 0 ? G + n->gamma_off + prod.bn : nullptr,
                      prod.bn >= 0 ? G + n->beta_off + prod.bn : nullptr);
}

// encoder backward from d(pre-activation of enc.dense) in ws+d_h down to d(pre-act conv1) in ws+d_conv[0]
static int encoder_backward(Ctx& c, const float* x, float* G, int want_param_grads) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  float* ws = c.ws;
  const int H1 = n->H1;
  if (want_param_grads) {  // dW of enc.dense: big = BN3(a3) (5,5,256), small = d_h
    WGParams w; wg_desc_conv(n->dense, c.B, w);
    w.Big = ws + P.act_conv[2]; w.Small = ws + P.d_h;
    w.b_scale = c.scale(n->conv[2].bn); w.b_shift = c.shift(n->conv[2].bn); w.b_aff_mask = 255;
    if (const float* y = bnp(c, P.bn_conv[2])) { w.Big = y; w.b_scale = w.b_shift = nullptr; w.b_aff_mask = 0; }
    TRY(run_wgrad(c, w, G + n->dense.w));
  }
  {  // g3 = d_h . W^T as a plain GEMM over the flattened (h,w,c) vector; epilogue = BN3/ReLU backward
    GGParams p; memset(&p, 0, sizeof(p));
    p.A = ws + P.d_h; p.W = c.P + n->dense.w;
    p.Nb = c.B; p.Hin = p.Win = p.Hout = p.Wout = 1; p.Cin = H1; p.Cout = 6400;
    p.kh = p.kw = 1; p.stride = 1; p.mode = 1;
    p.OutBf = bfp(c, P.bf_d_conv[2]);
    TRY(dgrad_bwd(c, p, n->conv[2], ws + P.act_conv[2], ws + P.d_conv[2], G, want_param_grads));
  }
  for (int i = 2; i >= 1; --i) {
    const ConvLayer& l = n->conv[i];
    if (want_param_grads) {
      WGParams w; wg_desc_conv(l, c.B, w, n->compute == VEEX_COMPUTE_BF16 ? 2 : 0, n->compute == VEEX_COMPUTE_F32X3 ? 1 : 0);
      w.Big = ws + P.act_conv[i - 1]; w.Small = ws + P.d_conv[i];
      w.BigBf = bfp(c, P.bf_act_conv[i - 1]); w.SmallBf = bfp(c, P.bf_d_conv[i]);
      w.b_scale = c.scale(n->conv[i - 1].bn); w.b_shift = c.shift(n->conv[i - 1].bn); w.b_aff_mask = n->conv[i - 1].Cout - 1;
      if (const float* y = bnp(c, P.bn_conv[i - 1])) { w.Big = y; w.b_scale = w.b_shift = nullptr; w.b_aff_mask = 0; }
      TRY(run_wgrad(c, w, G + l.w, l.name));
    }
    GGParams p; memset(&p, 0, sizeof(p));
    p.A = ws + P.d_conv[i]; p.W = c.P + l.w; p.Wbf = packed_w(c, i - 1, 0);
    p.wplanes = (c.n->compute == VEEX_COMPUTE_F32X3) ? 3 : 0;
    p.Abf = bfp(c, P.bf_d_conv[i]); p.OutBf = (i == 2) ? bfp(c, P.bf_d_conv[1]) : nullptr;
    p.Nb = c.B; p.Hin = p.Win = l.Hout; p.Cin = l.Cout; p.Hout = p.Wout = l.Hin; p.Cout = l.Cin;
    p.kh = p.kw = 3; p.stride = 2; p.pb = l.pb; p.mode = 1;
    TRY(dgrad_bwd(c, p, n->conv[i - 1], ws + P.act_conv[i - 1], ws + P.d_conv[i - 1], G, want_param_grads));
  }
  if (want_param_grads) {
    const int nb = veex_conv1_wgrad_blocks(c.B);
    TRY(veex_conv1_wgrad(x, ws + P.d_conv[0], ws + P.slab, c.B, c.st));
    TRY(veex_slab_sum(ws + P.slab, nb, 576, 576, G + n->conv[0].w, nullptr, 1, 0, c.st));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// public network entry points
// ------------------------------------------------------------------------------------------
static int encode_impl(Ctx& c, const float* x, float* eps_io, int flags, unsigned long long seed,
                       unsigned long long first_sample, int want_kl) {
  const VeexNet* n = c.n;
  const Plan& P = c.plan;
  TRY(encoder_forward(c, x));
  if (n->variant == VEEX_VARIANT_VAE) {
    const int philox = (flags & VEEX_F_PHILOX) ? 1 : 0;
    float* eps = c.ws + P.eps;
    if (!philox) {
      if (eps_io == nullptr) { veex_set_error("epsilon buffer required unless VEEX_F_PHILOX is set"); return VEEX_EINVAL; }
      if (hipMemcpyAsync(eps, eps_io, sizeof(float) * (size_t)c.B * n->L, hipMemcpyDeviceToDevice, c.st) != hipSuccess) {
        veex_set_error("epsilon copy failed"); return VEEX_ELAUNCH;
      }
    }
    TRY(veex_reparam_kl_fwd_ctr(c.ws + P.mulv, eps, c.ws + P.z, want_kl ? c.ws + P.kl_b : nullptr, c.B, n->L, philox, seed,
                                first_sample, philox ? n->sample_counter : nullptr, c.st));
    if (philox && eps_io != nullptr) {
      if (hipMemcpyAsync(eps_io, eps, sizeof(float) * (size_t)c.B * n->L, hipMemcpyDeviceToDevice, c.st) != hipSuccess) {
        veex_set_error("epsilon copy failed"); return VEEX_ELAUNCH;
      }
    }
  }
  return 0;
}

extern "C" int veex_net_encode(const VeexNet* n, const float* params, const float* state, const float* x, float* eps_io,
                               int B, int flags, unsigned long long seed, unsigned long long first_sample, float* ws,
                               long long ws_floats, float* z_mean, float* z_log_var, float* z, veexStream_t st) {
  TRY(check_common(n, params, state, B, ws, ws_floats));
  if (x == nullptr) { veex_set_error("null input"); return VEEX_EINVAL; }
  Ctx c; c.n = n; c.P = params; c.S = state; c.ws = ws; c.B = B; c.st = st;
  make_plan(n, B, c.plan);
  TRY(run_bn_prepare(c));
  if (n->compute != VEEX_COMPUTE_BF16) TRY(run_transpose_weights(c, true, false));
  TRY(encode_impl(c, x, eps_io, flags, seed, first_sample, 0));
  const int L = n->L;
  if (n->variant == VEEX_VARIANT_VAE) {
    // split [B][2L] into the two heads with strided 2-D copies
    if (z_mean != nullptr &&
        hipMemcpy2DAsync(z_mean, sizeof(float) * L, ws + c.plan.mulv, sizeof(float) * 2 * L, sizeof(float) * L, B,
                         hipMemcpyDeviceToDevice, st) != hipSuccess) { veex_set_error("copy failed"); return VEEX_ELAUNCH; }
    if (z_log_var != nullptr &&
        hipMemcpy2DAsync(z_log_var, sizeof(float) * L, ws + c.plan.mulv + L, sizeof(float) * 2 * L, sizeof(float) * L, B,
                         hipMemcpyDeviceToDevice, st) != hipSuccess) { veex_set_error("copy failed"); return VEEX_ELAUNCH; }
  }
  if (z != nullptr &&
      hipMemcpyAsync(z, ws + c.plan.z, sizeof(float) * (size_t)B * L, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    veex_set_error("copy failed"); return VEEX_ELAUNCH;
  }
  return 0;
}

extern "C" int veex_net_decode(const VeexNet* n, const float* params, const float* state, const float* z, int B,
                               float* ws, long long ws_floats, float* recon, float* logits, veexStream_t st) {
  TRY(check_common(n, params, state, B, ws, ws_floats));
  if (z == nullptr) { veex_set_error("null input"); return VEEX_EINVAL; }
  Ctx c; c.n = n; c.P = params; c.S = state; c.ws = ws; c.B = B; c.st = st;
  make_plan(n, B, c.plan);
  TRY(run_bn_prepare(c));
  TRY(decoder_trunk(c, z));
  return deconv6_forward(c, nullptr, logits, recon, nullptr, nullptr, nullptr);
}

extern "C" int veex_net_forward(const VeexNet* n, const float* params, const float* state, const float* x, float* eps_io,
                                int B, int flags, unsigned long long seed, unsigned long long first_sample, float* ws,
                                long long ws_floats, float* recon, float* losses, float* metrics, veexStream_t st) {
  TRY(check_common(n, params, state, B, ws, ws_floats));
  if (x == nullptr) { veex_set_error("null input"); return VEEX_EINVAL; }
  Ctx c; c.n = n; c.P = params; c.S = state; c.ws = ws; c.B = B; c.st = st;
  make_plan(n, B, c.plan);
  const Plan& P = c.plan;
  TRY(run_bn_prepare(c));
  if (n->compute != VEEX_COMPUTE_BF16) TRY(run_transpose_weights(c, true, false));
  TRY(encode_impl(c, x, eps_io, flags, seed, first_sample, 1));
  const float* zdec = ws + P.z;
  if ((flags & VEEX_F_ZMEAN_DECODE) && n->variant == VEEX_VARIANT_VAE) {
    if (hipMemcpy2DAsync(ws + P.dz, sizeof(float) * n->L, ws + P.mulv, sizeof(float) * 2 * n->L, sizeof(float) * n->L, B,
                         hipMemcpyDeviceToDevice, st) != hipSuccess) { veex_set_error("copy failed"); return VEEX_ELAUNCH; }
    zdec = ws + P.dz;
  }
  TRY(decoder_trunk(c, zdec));
  TRY(deconv6_forward(c, x, nullptr, recon, nullptr, ws + P.rec_b, nullptr));
  return veexi_loss_finalize(ws + P.rec_b, n->variant == VEEX_VARIANT_VAE ? ws + P.kl_b : nullptr, nullptr, B, c.div(),
                             losses, metrics, nullptr, st);
}

extern "C" int veex_net_train_fwd_bwd(const VeexNet* n, const float* params, const float* state, const float* x,
                                      float* eps_io, int B, int flags, unsigned long long seed,
                                      unsigned long long first_sample, float* ws, long long ws_floats, float* G,
                                      float* losses, float* metrics, veexStream_t st) {
  TRY(check_common(n, params, state, B, ws, ws_floats));
  if (x == nullptr || G == nullptr) { veex_set_error("null input"); return VEEX_EINVAL; }
  Ctx c; c.n = n; c.P = params; c.S = state; c.ws = ws; c.B = B; c.st = st;
  if (n->side != nullptr && n->side != st) c.side = n->side;
  make_plan(n, B, c.plan);
  const Plan& P = c.plan;
  const int L = n->L, H1 = n->H1;
  const bool vae = (n->variant == VEEX_VARIANT_VAE);
  c.batch_bn = (n->bn_mode == VEEX_BN_BATCH);     // (encode / decode / forward / attribute keep the moving statistics)
  if (c.batch_bn && B < 2) { veex_set_error("train_fwd_bwd: batch-statistics BatchNorm needs a batch of at least 2"); return VEEX_EINVAL; }

  // ---- forward (models.py:152-163) ----
  TRY(run_bn_prepare(c));
  if (n->compute != VEEX_COMPUTE_BF16) TRY(run_transpose_weights(c, true, true));
  TRY(encode_impl(c, x, eps_io, flags, seed, first_sample, 1));
  TRY(decoder_trunk(c, ws + P.z));
  TRY(deconv6_forward(c, x, nullptr, nullptr, ws + P.dlogit, ws + P.rec_b, ws + P.dlsum_b));
  TRY(veexi_loss_finalize(ws + P.rec_b, vae ? ws + P.kl_b : nullptr, ws + P.dlsum_b, B, c.div(), losses, metrics,
                          G + n->dc[5].b, st));

  // ---- backward (tape.gradient, models.py:166) ----
  {  // deconv6 + BN5/ReLU backward
    const ConvLayer& l5 = n->dc[4];
    float* d6 = take_partials(c, (long long)B * 448);   // rows 0-3 dW6 partials, rows 4-6 the three column sums
    if (d6 == nullptr) return VEEX_ENOMEM;
    float* d5 = (bfp(c, P.bf_d_dc[4]) != nullptr && dpre_dc_f32_unused(c, 4)) ? nullptr : ws + P.d_dc[4];
    if (c.batch_bn) {
      TRY(veexi_deconv6_bwd_raw(ws + P.dlogit, ws + P.act_dc[4], c.scale(l5.bn), c.shift(l5.bn), c.P + n->dc[5].w,
                                ws + P.d_dc[4], d6, B, st));
      TRY(veex_slab_sum(d6, B, 448, 256, G + n->dc[5].w, nullptr, 1, 0, st));
      TRY(bn_batch_backward(c, l5, ws + P.d_dc[4], ws + P.act_dc[4], ws + P.d_dc[4], G));
    } else {
    TRY(veex_deconv6_bwd_bf16(ws + P.dlogit, ws + P.act_dc[4], c.scale(l5.bn), c.shift(l5.bn), c.P + n->dc[5].w,
                              d5, bfp(c, P.bf_d_dc[4]), d6, B, st));
    TRY(veex_slab_sum(d6, B, 448, 256, G + n->dc[5].w, nullptr, 1, 0, st));
    TRY(defer_colsum(c, d6 + 256, B, 448, 64, 64, c.scale(l5.bn), c.mean(l5.bn), c.rstd(l5.bn), G + l5.b,
                     G + n->gamma_off + l5.bn, G + n->beta_off + l5.bn));
    }
  }
  for (int i = 4; i >= 0; --i) {
    const ConvLayer& l = n->dc[i];
    const ConvLayer& prod = (i == 0) ? n->ddense : n->dc[i - 1];
    {
      WGParams w; deconv_wgrad_desc(c, i, w);
      TRY(run_wgrad(c, w, G + l.w, l.name));
    }
    GGParams p; deconv_dgrad_desc(c, i, p);
    if (i > 0) {
      TRY(dgrad_bwd(c, p, prod, ws + P.act_dc[i - 1], ws + P.d_dc[i - 1], G, 1, dpre_dc_f32_unused(c, i - 1)));
    } else {
      // BatchNorm of dec.dense runs over the flattened 4096 features: plain dgrad, then row-wise BN/ReLU backward
      p.Out = ws + P.g_dd; p.epi = VEEX_EPI_FWD; p.relu = 0; p.bias = nullptr; p.nsplit = 1;
      TRY(veex_launch_gather_gemm(p, st));
      if (c.batch_bn) { TRY(bn_batch_backward(c, prod, ws + P.g_dd, ws + P.act_dd, ws + P.d_dd, G)); continue; }
      const int chunks = veex_bnrelu_bwd_rows_chunks(B);
      float* part = take_partials(c, (long long)chunks * 3 * 4096);
      if (part == nullptr) return VEEX_ENOMEM;
      TRY(veex_bnrelu_bwd_rows(ws + P.g_dd, ws + P.act_dd, c.scale(prod.bn), ws + P.d_dd, part, B, 4096, 4096, st));
      TRY(defer_colsum(c, part, chunks, 3LL * 4096, 4096, 4096, c.scale(prod.bn), c.mean(prod.bn),
                       c.rstd(prod.bn), G + prod.b, G + n->gamma_off + prod.bn, G + n->beta_off + prod.bn));
    }
  }
  {  // dec.dense: dW = z^T . dpre ; dz = dpre . W^T (split-K)
    WGParams w; wg_desc_conv(n->ddense, B, w);
    w.Big = ws + P.z; w.Small = ws + P.d_dd;
    TRY(run_wgrad(c, w, G + n->ddense.w));
    GGParams p; memset(&p, 0, sizeof(p));
    p.A = ws + P.d_dd; p.W = c.P + n->ddense.w; p.Out = ws + P.slab;
    p.Nb = B; p.Hin = p.Win = p.Hout = p.Wout = 1; p.Cin = 4096; p.Cout = L;
    p.kh = p.kw = 1; p.stride = 1; p.mode = 1; p.epi = VEEX_EPI_RAW;
    GGParams q = p;
    const int mt = veex_gg_m_tiles(q);
    const int chunks = 4096 / VEEX_BK;
    p.nsplit = gg_split(mt, L, chunks);
    p.chunks_per_split = (chunks + p.nsplit - 1) / p.nsplit;
    p.slab_stride = (long long)B * L;
    if ((long long)p.nsplit * p.slab_stride > P.slab_n) { veex_set_error("internal: slab scratch too small (dz)"); return VEEX_ENOMEM; }
    TRY(veex_launch_gather_gemm(p, st));
    TRY(veex_slab_sum(ws + P.slab, p.nsplit, p.slab_stride, p.slab_stride, ws + P.dz, nullptr, 1, 0, st));
  }
  if (n->bucket_ev != nullptr) {
    // Data parallelism, early bucket: every kernel / bias gradient of the decoder (veex_net_grad_bucket) is final once
    // the column sums collected so far are finalised and the weight-gradient GEMMs enqueued so far on the side stream
    // have run.  The event is recorded on the side stream behind a fork from here, so the main stream does not wait.
    TRY(flush_colsum(c));
    hipStream_t es = st;
    if (c.side != nullptr) { TRY(fork_side(c)); es = c.side; }
    if (hipEventRecord(n->bucket_ev, es) != hipSuccess) { veex_set_error("bucket event record failed"); return VEEX_ELAUNCH; }
  }
  if (vae) {
    TRY(veex_reparam_kl_bwd(ws + P.dz, ws + P.mulv, ws + P.eps, ws + P.dmulv, B, L, c.inv_div(), st));
    {  // head biases: column sums of dmulv
      const int chunks = veex_bnrelu_bwd_rows_chunks(B);
      float* part = take_partials(c, (long long)chunks * 3 * 2 * L);
      if (part == nullptr) return VEEX_ENOMEM;
      TRY(veex_bnrelu_bwd_rows(ws + P.dmulv, nullptr, nullptr, nullptr, part, B, 2 * L, 0, st));
      TRY(defer_colsum(c, part, chunks, 3LL * 2 * L, 2 * L, 2 * L, nullptr, nullptr, nullptr, G + n->bh, nullptr, nullptr));
    }
    {  // dWh = h^T . dmulv
      WGParams w; memset(&w, 0, sizeof(w));
      w.Big = ws + P.act_h; w.Small = ws + P.dmulv;
      w.Nb = B; w.Hb = w.Wb = w.Hs = w.Ws = 1; w.Cb = 100; w.Cs = 2 * L; w.kh = w.kw = 1; w.stride = 1;
      w.R = B; w.M = 100;
      TRY(run_wgrad(c, w, G + n->wh));
    }
    {  // d h = dmulv . Wh^T, ReLU backward of enc.dense, bias gradient of enc.dense
      GGParams p; memset(&p, 0, sizeof(p));
      p.A = ws + P.dmulv; p.W = c.P + n->wh;
      p.Nb = B; p.Hin = p.Win = p.Hout = p.Wout = 1; p.Cin = 2 * L; p.Cout = 100;
      p.kh = p.kw = 1; p.stride = 1; p.mode = 1;
      TRY(dgrad_bwd(c, p, n->dense, ws + P.act_h, ws + P.d_h, G, 1));
    }
  } else {  // AE: z is the ReLU bottleneck itself (models_standard.py:108)
    const int chunks = veex_bnrelu_bwd_rows_chunks(B);
    float* part = take_partials(c, (long long)chunks * 3 * H1);
    if (part == nullptr) return VEEX_ENOMEM;
    TRY(veex_bnrelu_bwd_rows(ws + P.dz, ws + P.act_h, nullptr, ws + P.d_h, part, B, H1, 0, st));
    TRY(defer_colsum(c, part, chunks, 3LL * H1, H1, H1, nullptr, nullptr, nullptr, G + n->dense.b, nullptr, nullptr));
  }
  TRY(encoder_backward(c, x, G, 1));
  TRY(flush_colsum(c));
  return join_side(c);
}

extern "C" int veex_net_attribute(const VeexNet* n, const float* params, const float* state, const float* x, int B,
                                  const int* units, int n_units, int grad_x_input, float* ws, long long ws_floats,
                                  float* out, veexStream_t st) {
  TRY(check_common(n, params, state, B, ws, ws_floats));
  if (x == nullptr || out == nullptr || units == nullptr || n_units < 1) { veex_set_error("attribute: bad arguments"); return VEEX_EINVAL; }
  for (int u = 0; u < n_units; ++u)
    if (units[u] < 0 || units[u] >= n->L) { veex_set_error("attribute: unit %d out of range", units[u]); return VEEX_EINVAL; }
  Ctx c; c.n = n; c.P = params; c.S = state; c.ws = ws; c.B = B; c.st = st;
  make_plan(n, B, c.plan);
  const Plan& P = c.plan;
  TRY(run_bn_prepare(c));
  if (n->compute != VEEX_COMPUTE_BF16) TRY(run_transpose_weights(c, true, false));
  TRY(encoder_forward(c, x));
  const bool vae = (n->variant == VEEX_VARIANT_VAE);
  for (int u = 0; u < n_units; ++u) {
    TRY(veexi_attr_seed(ws + P.act_h, vae ? params + n->wh : nullptr, ws + P.d_h, B, n->H1, 2 * n->L, units[u],
                        vae ? 0 : 1, st));
    TRY(encoder_backward(c, x, nullptr, 0));
    TRY(veex_conv1_dgrad(ws + P.d_conv[0], params + n->conv[0].w, grad_x_input ? x : nullptr,
                         out + (long long)u * B * 1600, B, st));
  }
  return 0;
}
