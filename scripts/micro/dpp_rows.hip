// which lane does a 16-lane-row DPP shift read?  (hipcc --offload-arch=gfx950 dpp_rows.hip -o dpp_rows && ./dpp_rows)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int v = threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, false);        // row_shl:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x121, 0xf, 0xf, false);  // row_ror:1
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, true);   // row_shl:1 bound_ctrl
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"row_shl:1", "row_shr:1", "row_ror:1", "row_shl:1 bc"};
  for (int a = 0; a < 4; ++a) { printf("%-14s", nm[a]); for (int i = 0; i < 20; ++i) printf(" %d", h[a * 64 + i]); printf("\n"); }
  return 0;
}
