// Is the SGPR offset of a raw buffer load part of the range check on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* buf, float* out, int nrec_bytes, int soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, nrec_bytes, 0x00020000);
  unsigned v0 = __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, soff, 0);
  out[threadIdx.x] = __uint_as_float(v0);
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  float *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 64 * 4); hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  float r[64];
  // 128 bytes of records (32 floats); voffset = lane*4 (< 128 for lanes < 32); soffset = 64 bytes
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 128, 64);
  hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  printf("num_records=128B soffset=64B: lane 8 -> %.0f (124 = in range), lane 20 (voff 80 + 64 = 144 >= 128) -> %.0f (0 = soffset checked, 136 = not checked), lane 40 (voff 160) -> %.0f\n", r[8], r[20], r[40]);
  return 0;
}
