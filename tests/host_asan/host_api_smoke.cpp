// Host-side AddressSanitizer smoke of libveex's HOST-ONLY entry points ("host ASan build of the shim").
// Built by `make -C vae_eeg_xai_amd/csrc asan` against an ASan-instrumented build of the library (host code only:
// -Xarch_host -fsanitize=address; device code is not instrumented) and run on the CPU by tests/test_host_asan.py.
// Touches no GPU: plan / table / geometry / argument-checking paths, i.e. everything the library does on the host
// before it enqueues (string tables, the workspace planner, the row-split planner, the error buffer, the env cache).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/veex.h"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int one_net(int latent, int variant, int compute) {
  VeexNet* n = veex_net_create(latent, variant);
  REQUIRE(n != nullptr);
  REQUIRE(veex_net_set_compute(n, compute) == 0);
  const long long np = veex_net_param_floats(n), ns = veex_net_state_floats(n);
  REQUIRE(np > 0 && np % 4 == 0 && ns > 0);
  const int nt = veex_net_num_tensors(n);
  REQUIRE(nt > 0);
  long long covered_tr = 0, covered_st = 0;
  for (int i = 0; i < nt; ++i) {
    char name[8];                       // deliberately short: the library must truncate, not overrun
    char full[96];
    long long off = -1, shape[4] = {0, 0, 0, 0};
    int ndim = 0, tr = -1, part = -1;
    REQUIRE(veex_net_tensor_info(n, i, name, (int)sizeof(name), &off, &ndim, shape, &tr, &part) == 0);
    REQUIRE(std::strlen(name) < sizeof(name));
    REQUIRE(veex_net_tensor_info(n, i, full, (int)sizeof(full), &off, &ndim, shape, &tr, &part) == 0);
    REQUIRE(ndim >= 1 && ndim <= 4 && off >= 0 && (tr == 0 || tr == 1));
    long long cnt = 1;
    for (int d = 0; d < ndim; ++d) cnt *= shape[d];
    REQUIRE(off + (part == 0 ? cnt : 0) <= (tr ? np : ns));
    if (part == 0 || part == 1) (tr ? covered_tr : covered_st) += (part == 0 ? cnt : 2 * cnt);
  }
  REQUIRE(covered_tr <= np && covered_tr > np - 8 && covered_st == ns);
  REQUIRE(veex_net_tensor_info(n, nt, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr) != 0);   // out of range
  for (int B : {1, 3, 40, 64, 512, 1024, 4096}) {
    const long long w = veex_net_workspace_floats(n, B);
    REQUIRE(w > 0);
    long long nf = 0;
    const long long o = veex_net_ws_offset(n, B, "act.dec.deconv5", &nf);
    REQUIRE(o >= 0 && nf == (long long)B * 39 * 39 * 64 && o + nf <= w);
    REQUIRE(veex_net_ws_offset(n, B, "no.such.region", &nf) < 0);
  }
  REQUIRE(veex_net_workspace_floats(n, 0) < 0);
  REQUIRE(veex_net_set_batch_divisor(n, 128.0) == 0 && veex_net_set_batch_divisor(n, -1.0) != 0);
  // argument checking happens before anything is enqueued: null buffers are refused with a message
  REQUIRE(veex_net_train_fwd_bwd(n, nullptr, nullptr, nullptr, nullptr, 8, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr) != 0);
  REQUIRE(std::strlen(veex_last_error()) > 0);
  veex_net_destroy(n);
  return 0;
}

int main() {
  for (int variant : {VEEX_VARIANT_VAE, VEEX_VARIANT_AE})
    for (int latent : {2, 28, 128})
      for (int compute : {VEEX_COMPUTE_F32, VEEX_COMPUTE_BF16})
        if (one_net(latent, variant, compute) != 0) return 1;
  REQUIRE(veex_net_create(0, VEEX_VARIANT_VAE) == nullptr);
  REQUIRE(veex_net_create(28, 7) == nullptr);
  // the row-split planner and the tile counter (host arithmetic only)
  VeexWgradDesc w; std::memset(&w, 0, sizeof(w));
  w.N = 512; w.Hb = w.Wb = 19; w.Cb = 64; w.Hs = w.Ws = 17; w.Cs = 128; w.kh = w.kw = 3; w.stride = 1;
  int32_t ns = 0, per = 0;
  REQUIRE(veex_wgrad_plan(&w, &ns, &per) == 0 && ns >= 1 && per % 16 == 0 && (long long)ns * per >= 512LL * 17 * 17);
  REQUIRE(veex_wgrad_plan(nullptr, &ns, &per) != 0);
  VeexGemmDesc g; std::memset(&g, 0, sizeof(g));
  float dummy = 0.f;
  g.in = g.weight = &dummy; g.out = &dummy;      // never dereferenced: the call only counts tiles
  g.N = 512; g.Hin = g.Win = 8; g.Cin = 128; g.Hout = g.Wout = 17; g.Cout = 128; g.kh = g.kw = 3; g.stride = 2; g.mode = 1;
  g.nsplit = 1;
  REQUIRE(veex_gather_gemm_m_tiles(&g) > 0);
  g.mode = 5;
  REQUIRE(veex_gather_gemm_m_tiles(&g) < 0);
  std::puts("host ASan smoke: ok");
  return 0;
}
