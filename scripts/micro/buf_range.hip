#include <hip/hip_runtime.h>
#include <stdio.h>
// does the raw-buffer range check include the SGPR offset?  num_records = 64 bytes; stores at voffset 0 with soffset 128
__global__ void k(float* y, int so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, 64, 0x00020000);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u, r, 0, so, 0);          // voffset 0, soffset 128 -> byte 128 (out of the 64)
    __builtin_amdgcn_raw_buffer_store_b32(0x40000000u, r, 128 + 4, 0, 0);     // voffset 132 -> out of range for sure
    __builtin_amdgcn_raw_buffer_store_b32(0x40400000u, r, 8, 0, 0);           // in range
    unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, 0, so + 16, 0);      // load through soffset 144
    y[60] = __uint_as_float(v);
  }
}
int main() {
  float* y; hipMalloc(&y, 4096); 
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = -7.f;
  hipMemcpy(y, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, y, 128);
  hipMemcpy(h, y, 4096, hipMemcpyDeviceToHost);
  printf("y[32] (soffset store at byte 128) = %g   y[33] (voffset 132) = %g   y[2] = %g   load via soffset 144 -> %g (memory holds %g)\n", h[32], h[33], h[2], h[60], h[36]);
  return 0;
}
