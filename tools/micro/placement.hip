// Where does the dispatcher put the workgroups of a 512-thread, 66 KB-LDS launch?  Prints, per physical CU, the
// block ids it ran (speed heuristics only -- HIP promises nothing about placement).
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/placement.hip -o tools/micro/placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void census(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    smem[0] = 1;
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);        // stay resident so that later blocks must share / queue
}
// the same with 128 VGPRs allocated (v127 touched): do two such 512-thread workgroups really share a CU?
__global__ __launch_bounds__(512) void census128(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = 0; smem[0] = 1; }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512, lds = argc > 2 ? atoi(argv[2]) : 67872;
  unsigned* d; hipMalloc(&d, 1024 * 8 + blocks * 8);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(census, dim3(blocks), dim3(512), lds, 0, d, 20);
  std::vector<unsigned> h(2 * blocks);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    cu[key].push_back(b);
  }
  printf("%d blocks, %d bytes LDS: %zu distinct CUs\n", blocks, lds, cu.size());
  hipFuncSetAttribute((const void*)census128, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which)
    for (int nb : {256, 512, 768, 1024}) {
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (which) hipLaunchKernelGGL(census128, dim3(nb), dim3(512), lds, 0, d, 20);
        else hipLaunchKernelGGL(census, dim3(nb), dim3(512), lds, 0, d, 20);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("  %s, %4d blocks of 20 x s_sleep(127): %.1f us\n", which ? "128 VGPRs" : "few VGPRs", nb, ms * 1e3);
    }
  int n = 0;
  for (auto& kv : cu) {
    if (n++ % 16 == 0) { printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  }
  return 0;
}
