#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
  uint32_t v = threadIdx.x * 10u + 7u;
  uint32_t n = (uint32_t)__builtin_amdgcn_update_dpp(0xdead, (int)v, 0x130, 0xf, 0xf, false);
  uint32_t d = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true) - v;
  out[threadIdx.x] = n; out[64 + threadIdx.x] = d;
}
int main() {
  uint32_t* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); uint32_t h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%u:%u/%d ", i, h[i], (int)h[64+i]); printf("\n"); return 0;
}
