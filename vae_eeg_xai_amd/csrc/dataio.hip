// The steps either side of the hot path, HBM-bound, fp32:
//   normalize(x)   = (x - min(x)) / (max(x) - min(x)) over the WHOLE array      reference train.py:99-100
//   denormalize    = x_norm * (max - min) + min                                  reference signal_reconstruction.py:92-95
//   electrode read-out: one pixel per electrode gathered from every 40x40 map    reference indian_functions.py:146-160
// The arithmetic is the reference's fp32 expression op for op (subtract, IEEE divide; multiply,
// add -- no FMA contraction), so results are bit-identical to NumPy float32.
#include "veex_internal.h"
#include <float.h>
#include <mutex>

#pragma clang fp contract(off)

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// NumPy's min / max PROPAGATE NaN (np.min of an array that holds a NaN is NaN; train.py:99-100 then normalises the
// whole set to NaN), fminf / fmaxf DROP it: every element is also tested for v != v, and a block / the final fold that
// saw one writes NaN for both extrema.
__device__ __forceinline__ bool block_any(bool flag, int* s_flag) {   // s_flag: one LDS word, zeroed by thread 0 here
  if (threadIdx.x == 0) *s_flag = 0;
  __syncthreads();
  if (__any(flag) && (threadIdx.x & 63) == 0) atomicOr(s_flag, 1);
  __syncthreads();
  return *s_flag != 0;
}

// pass 1: per-block (min, max) partials [blocks][2]; min/max are exact whatever the order
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  float lo = FLT_MAX, hi = -FLT_MAX;
  bool bad = false;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x4[i];
    lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
    bad |= (v != v);
  }
  lo = wave_min(lo); hi = wave_max(hi);
  __shared__ float s[8];
  __shared__ int s_bad;
  const bool any_bad = block_any(bad, &s_bad);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[w] = lo; s[4 + w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float qnan = __uint_as_float(0x7FC00000u);
    part[2 * blockIdx.x] = any_bad ? qnan : fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
    part[2 * blockIdx.x + 1] = any_bad ? qnan : fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
  }
}

// pass 2: one block folds the partials into minmax[2]
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int blocks, float* __restrict__ minmax) {
  float lo = FLT_MAX, hi = -FLT_MAX;
  bool bad = false;
  for (int i = threadIdx.x; i < blocks; i += 256) {
    const float a = part[2 * i], b = part[2 * i + 1];
    lo = fminf(lo, a); hi = fmaxf(hi, b);
    bad |= (a != a) | (b != b);
  }
  lo = wave_min(lo); hi = wave_max(hi);
  __shared__ float s[8];
  __shared__ int s_bad;
  const bool any_bad = block_any(bad, &s_bad);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[w] = lo; s[4 + w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float qnan = __uint_as_float(0x7FC00000u);
    minmax[0] = any_bad ? qnan : fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
    minmax[1] = any_bad ? qnan : fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
  }
}

// out = (x - lo) / (hi - lo)   (train.py:100; a constant array gives 0/0 = NaN like NumPy)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, const float* __restrict__ minmax,
                                                        float* __restrict__ out, long long n) {
  const float lo = minmax[0], range = minmax[1] - minmax[0];
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = x4[i];
    v.x = (v.x - lo) / range; v.y = (v.y - lo) / range; v.z = (v.z - lo) / range; v.w = (v.w - lo) / range;
    o4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    out[i] = (x[i] - lo) / range;
  }
}

// out[b][e] = maps[b][pix[e]] * (hi - lo) + lo   (denormalize fused into the electrode gather;
// minmax == nullptr: plain gather).  One wave per map: 64 lanes cover up to 64 electrodes per pass.
__global__ __launch_bounds__(256) void electrode_readout_kernel(const float* __restrict__ maps, const int* __restrict__ pix,
                                                                const float* __restrict__ minmax, float* __restrict__ out,
                                                                int batch, int map_elems, int n_el) {
  const int lane = threadIdx.x & 63;
  float lo = 0.f, range = 1.f;
  const bool den = minmax != nullptr;
  if (den) { lo = minmax[0]; range = minmax[1] - minmax[0]; }
  for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < batch; b += gridDim.x * 4) {
    const float* m = maps + (long long)b * map_elems;
    for (int e = lane; e < n_el; e += 64) {
      float v = m[pix[e]];
      if (den) { v = v * range; v = v + lo; }
      out[(long long)b * n_el + e] = v;
    }
  }
}

// out = x * (hi - lo) + lo  (signal_reconstruction.py:94), whole maps
__global__ __launch_bounds__(256) void denormalize_kernel(const float* __restrict__ x, const float* __restrict__ minmax,
                                                          float* __restrict__ out, long long n) {
  const float lo = minmax[0], range = minmax[1] - minmax[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = x[i] * range;
    out[i] = v + lo;
  }
}

static inline int blocks_for(long long work, long long per_block, int cap) {
  long long b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

extern "C" {

int veex_minmax_partial_blocks(long long n) { return blocks_for(n, 256 * 16, 2048); }

int veex_minmax(const float* x, long long n, float* partials, float* minmax, hipStream_t st) {
  if (x == nullptr || partials == nullptr || minmax == nullptr || n < 1) { veex_set_error("minmax: bad arguments"); return -1; }
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) { veex_set_error("minmax: x must be 16-byte aligned"); return -1; }
  const int blocks = veex_minmax_partial_blocks(n);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, partials);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, st, partials, blocks, minmax);
  VEEX_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

// ---- veex_minmax for callers that cannot allocate (HIP-graph capture, tight serving loops): the per-block partials
// live in a file-scope device array instead of a caller-provided buffer.  16 KB per GPU, held until the process
// ends.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.
// Nothing has to start at zero: pass 1 overwrites partial rows [0, blocks), pass 2 reads exactly those.
#define VEEX_MINMAX_STATIC_BLOCKS 2048                 /* = the cap of veex_minmax_partial_blocks */
#define VEEX_MINMAX_STATIC_MAX_ELEMS (1LL << 32)       /* stated limit: 2^32 floats (16 GiB) per call */
#define VEEX_MAX_DEVICES 64
static __device__ __align__(256) float scratch[2 * VEEX_MINMAX_STATIC_BLOCKS];
static std::mutex g_scratch_mu;
static float* g_scratch_ptr[VEEX_MAX_DEVICES];         // per device: the array's address ON THAT GPU (not the host-side name)

// The array of the GPU that owns `x`: the kernels run on the stream's device and must be handed THAT device's copy
// of the array.  The caller's current device must be that GPU too (hipGetSymbolAddress and the launch both act on the
// current device); a mismatch is refused instead of launching with another GPU's address.
static int scratch_of_device_owning(const float* x, float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VEEX_MAX_DEVICES) { veex_set_error("minmax_static: no current device"); return -3; }
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, x) == hipSuccess && attr.type == hipMemoryTypeDevice && attr.device != dev) {
    veex_set_error("minmax_static: x lives on GPU %d but the current device is %d (set the device that owns x and the stream)", attr.device, dev);
    return -1;
  }
  (void)hipGetLastError();   // an unregistered pointer leaves a sticky error behind; the launch check below must not see it
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {   // first call on this device (outside stream capture): resolve the symbol once
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) { veex_set_error("minmax_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e)); return -3; }
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

extern "C" {

int veex_minmax_static(const float* x, long long n, float* minmax, hipStream_t st) {
  if (x == nullptr || minmax == nullptr || n < 1) { veex_set_error("minmax_static: bad arguments"); return -1; }
  if (n > VEEX_MINMAX_STATIC_MAX_ELEMS) { veex_set_error("minmax_static: %lld elements exceed the limit of %lld per call", n, (long long)VEEX_MINMAX_STATIC_MAX_ELEMS); return -1; }
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) { veex_set_error("minmax_static: x must be 16-byte aligned"); return -1; }
  const int blocks = veex_minmax_partial_blocks(n);
  if (blocks > VEEX_MINMAX_STATIC_BLOCKS) { veex_set_error("minmax_static: internal: %d blocks exceed the scratch array", blocks); return -1; }
  float* part = nullptr;
  const int rc = scratch_of_device_owning(x, &part);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, st, part, blocks, minmax);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_normalize(const float* x, const float* minmax, float* out, long long n, hipStream_t st) {
  if (x == nullptr || out == nullptr || minmax == nullptr || n < 1) { veex_set_error("normalize: bad arguments"); return -1; }
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) { veex_set_error("normalize: buffers must be 16-byte aligned"); return -1; }
  hipLaunchKernelGGL(normalize_kernel, dim3(blocks_for(n, 256 * 16, 4096)), dim3(256), 0, st, x, minmax, out, n);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_denormalize(const float* x, const float* minmax, float* out, long long n, hipStream_t st) {
  if (x == nullptr || out == nullptr || minmax == nullptr || n < 1) { veex_set_error("denormalize: bad arguments"); return -1; }
  hipLaunchKernelGGL(denormalize_kernel, dim3(blocks_for(n, 256 * 8, 4096)), dim3(256), 0, st, x, minmax, out, n);
  VEEX_CHECK_LAUNCH();
  return 0;
}

int veex_electrode_readout(const float* maps, const int* pix, const float* minmax, float* out, int batch, int map_elems,
                           int n_electrodes, hipStream_t st) {
  if (maps == nullptr || pix == nullptr || out == nullptr || batch < 1 || map_elems < 1 || n_electrodes < 1) {
    veex_set_error("electrode_readout: bad arguments"); return -1;
  }
  hipLaunchKernelGGL(electrode_readout_kernel, dim3(blocks_for(batch, 4, 4096)), dim3(256), 0, st, maps, pix, minmax, out,
                     batch, map_elems, n_electrodes);
  VEEX_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
