#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float buf[4][2][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float __attribute__((address_space(1)))* g = (const float __attribute__((address_space(1)))*)(src + wave * 1024 + lane * 4);
  for (int s = 0; s < 2; s++)
    __builtin_amdgcn_global_load_lds(g + s * 256, (float __attribute__((address_space(3)))*)&buf[wave][s][0], 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // unaligned 5-float read from LDS
  const float* p = &buf[wave][1][0] + ((lane * 3 + shift) & 127);
  typedef f4 f4u __attribute__((aligned(4)));
  f4 v = *reinterpret_cast<const f4u*>(p);
  float w = p[4];
  out[threadIdx.x] = v.x + v.y * 2 + v.z * 3 + v.w * 4 + w * 5;
}
int main() {
  float *d, *o; hipMalloc(&d, 4096 * 4); hipMalloc(&o, 256 * 4);
  float h[4096]; for (int i = 0; i < 4096; i++) h[i] = (float)i; hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 256>>>(d, o, 1); float r[256]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; t++) { int wave = t >> 6, lane = t & 63; int b = wave * 1024 + 256 + ((lane * 3 + 1) & 127);
    float e = b + (b + 1) * 2.f + (b + 2) * 3.f + (b + 3) * 4.f + (b + 4) * 5.f; if (r[t] != e) { if (bad < 5) printf("t %d got %f exp %f\n", t, r[t], e); bad++; } }
  printf("bad %d\n", bad); return 0;
}
