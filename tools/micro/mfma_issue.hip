// Micro-benchmark: how much MFMA throughput do co-issued VALU / LDS / SALU instructions cost on gfx950?
// Each wave runs ITER x { NM dependent v_mfma_f32_32x32x2_f32 on ACC accumulators, NV independent v_fma_f32,
// NL ds_read_b128, NS s_add }.  Build: hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, int ACC, int NS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float4 lds[1024];
  f32x16 acc[ACC];
  for (int a = 0; a < ACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float4 l = make_float4(0, 0, 0, 0);
  int s = 0;
  lds[threadIdx.x] = make_float4(x, x, x, x);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m % ACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m % ACC], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j % 8]) : "v"(y));
#pragma unroll
      for (int j = 0; j < NS / 8; ++j) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      float4 t;
      asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(threadIdx.x * 16 + j * 4096)));
      l.x += t.x;
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float t = l.x + s;
  for (int a = 0; a < ACC; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
  for (int j = 0; j < 8; ++j) t += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NV, int NL, int ACC, int NS>
void run(const char* name, float* out, int wps) {
  const int iters = 2000;
  const int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD when all resident
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NL, ACC, NS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_cycles = (double)wps * iters * 8 * 64;   // per SIMD
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-34s waves/SIMD=%d  %.3f ms  MFMA util %.3f  (extra cycles per 8-MFMA group per wave: %.1f)\n", name, wps, ms,
         mfma_cycles / cycles, (cycles - mfma_cycles) / ((double)wps * iters));
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int wps : {1, 2, 4, 7}) {
    run<0, 0, 1, 0>("8 MFMA (1 acc)", out, wps);
    run<0, 0, 2, 0>("8 MFMA (2 acc)", out, wps);
    run<16, 0, 1, 0>("8 MFMA + 16 VALU", out, wps);
    run<32, 0, 1, 0>("8 MFMA + 32 VALU", out, wps);
    run<64, 0, 1, 0>("8 MFMA + 64 VALU", out, wps);
    run<0, 4, 1, 0>("8 MFMA + 4 ds_read_b128", out, wps);
    run<32, 4, 1, 0>("8 MFMA + 32 VALU + 4 ds_read_b128", out, wps);
    run<0, 0, 1, 32>("8 MFMA + 32 SALU", out, wps);
    run<0, 0, 1, 64>("8 MFMA + 64 SALU", out, wps);
    run<0, 0, 2, 64>("8 MFMA (2 acc) + 64 SALU", out, wps);
  }
  return 0;
}
