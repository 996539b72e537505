// Micro-benchmark (measurement tool, not product code): HBM read bandwidth on gfx950 for
//  (a) an ideal contiguous grid-stride float4 read, and
//  (b) the mix kernel's access pattern: workgroup (block b, group g) reads, for 64 tracks, one 2 KiB row
//      per channel (4 waves x 1 KiB) from 64 x 2 separate arrays, K blocks contiguous per array.
// build: hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw ; run: ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_contig(const f4* __restrict__ p, size_t n4, float* out) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f4 v = p[i];
    acc += v;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int U>
__global__ __launch_bounds__(256) void k_rows(const float* const* __restrict__ chans, int tracks_per_group, int frames_per_block,
                                              float* out) {
  const int b = blockIdx.x, g = blockIdx.y;
  const int c = threadIdx.x >> 7, j0 = (threadIdx.x & 127) * 4;
  f4 acc = {0, 0, 0, 0};
  for (int t0 = 0; t0 < tracks_per_group; t0 += U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float* base = chans[((size_t)(g * tracks_per_group + t0 + u)) * 2 + c];
      v[u] = *reinterpret_cast<const f4*>(base + (size_t)b * frames_per_block + j0);
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// the window (linear resample) load shape: lane L reads 16 B at float offset floor(L*4*0.91875)+off and 4 B behind it
template <int U, bool W4>
__global__ __launch_bounds__(256) void k_window(const float* const* __restrict__ chans, int tracks_per_group, int frames_per_block,
                                                int off, float* out) {
  const int b = blockIdx.x, g = blockIdx.y;
  const int c = threadIdx.x >> 7, j0 = (threadIdx.x & 127) * 4;
  const int src0 = (int)((double)((size_t)b * frames_per_block + j0) * 0.91875) + off;
  f4 acc = {0, 0, 0, 0};
  for (int t0 = 0; t0 < tracks_per_group; t0 += U) {
    f4 v[U]; float w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float* base = chans[((size_t)(g * tracks_per_group + t0 + u)) * 2 + c];
      v[u] = *reinterpret_cast<const f4u*>(base + src0);
      w[u] = W4 ? base[src0 + 4] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) { acc += v[u]; acc.x += w[u]; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// the same samples through LDS: every wave fetches its contiguous, 16-B aligned source span (1 KiB) with one
// direct-to-LDS load per track (no VGPRs held while in flight, D tracks deep), then each lane picks its 5 taps
// one __shared__ array per ring slot (distinct objects): the compiler's waitcnt pass can then tell which
// direct-to-LDS load a later ds_read depends on, instead of waiting for vmcnt(0) before every LDS read
template <int S>
__device__ __forceinline__ float* ring_slot(int wave) {
  __shared__ __attribute__((aligned(16))) float buf[4][256];
  return buf[wave];
}
template <int D>
__global__ __launch_bounds__(256) void k_lds(const float* const* __restrict__ chans, int tracks_per_group, int frames_per_block,
                                             int off, float* out) {
  const int b = blockIdx.x, g = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = threadIdx.x >> 7, j0 = (threadIdx.x & 127) * 4;
  const int jw = (wave & 1) * 256;
  const int src0 = (int)((double)((size_t)b * frames_per_block + j0) * 0.91875) + off;
  const int base = ((int)((double)((size_t)b * frames_per_block + jw) * 0.91875) + off) & ~3;
  const int o = src0 - base;
  f4 acc = {0, 0, 0, 0};
  auto issue = [&](int t, float* slot) {
    const float* p = chans[((size_t)(g * tracks_per_group + t)) * 2 + c] + base + lane * 4;
    __builtin_amdgcn_global_load_lds((const float __attribute__((address_space(1)))*)p,
                                     (float __attribute__((address_space(3)))*)slot, 16, 0, 0);
  };
  auto use = [&](const float* slot) {
    const float* q = slot + o;
    acc.x += q[0]; acc.y += q[1]; acc.z += q[2]; acc.w += q[3]; acc.x += q[4];
  };
  static_assert(D == 4 || D == 8, "slots are spelled out");
  float* s0 = ring_slot<0>(wave); float* s1 = ring_slot<1>(wave); float* s2 = ring_slot<2>(wave); float* s3 = ring_slot<3>(wave);
  float* s4 = ring_slot<4>(wave); float* s5 = ring_slot<5>(wave); float* s6 = ring_slot<6>(wave); float* s7 = ring_slot<7>(wave);
  // prologue: D-1 loads in flight
  issue(0, s0); issue(1, s1); issue(2, s2);
  if (D == 8) { issue(3, s3); issue(4, s4); issue(5, s5); issue(6, s6); }
  const int T = tracks_per_group;   // multiple of D
  for (int t0 = 0; t0 < T; t0 += D) {
    const bool more = t0 + D < T;
    if (D == 4) {
      issue(t0 + 3, s3); use(s0);
      if (more) issue(t0 + 4, s0); use(s1);
      if (more) issue(t0 + 5, s1); use(s2);
      if (more) issue(t0 + 6, s2); use(s3);
    } else {
      issue(t0 + 7, s7); use(s0);
      if (more) issue(t0 + 8, s0); use(s1);
      if (more) issue(t0 + 9, s1); use(s2);
      if (more) issue(t0 + 10, s2); use(s3);
      if (more) issue(t0 + 11, s3); use(s4);
      if (more) issue(t0 + 12, s4); use(s5);
      if (more) issue(t0 + 13, s5); use(s6);
      if (more) issue(t0 + 14, s6); use(s7);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int WARM = argc > 1 ? atoi(argv[1]) : 3;   // launches before the timed ones (60+: the steady clocks)

  const int N = 4096, K = 256, F = 512, G = 64;
  const size_t frames = (size_t)K * F;
  std::vector<float*> h(N * 2);
  for (auto& p : h) { hipMalloc(&p, frames * 4 + 256); hipMemset(p, 0, frames * 4); }
  float** d; hipMalloc(&d, h.size() * sizeof(float*)); hipMemcpy(d, h.data(), h.size() * sizeof(float*), hipMemcpyHostToDevice);
  float* big; const size_t bytes = (size_t)N * 2 * frames * 4; hipMalloc(&big, bytes); hipMemset(big, 0, bytes);
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < WARM; i++) launch();
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
  };
  timeit("contiguous grid-stride x2048", [&] { hipLaunchKernelGGL(k_contig, dim3(2048), dim3(256), 0, 0, (const f4*)big, bytes / 16, out); });
  timeit("contiguous grid-stride x8192", [&] { hipLaunchKernelGGL(k_contig, dim3(8192), dim3(256), 0, 0, (const f4*)big, bytes / 16, out); });
  timeit("rows U=2 (b fastest)", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, out); });
  timeit("rows U=4 (b fastest)", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, out); });
  timeit("rows U=8 (b fastest)", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, out); });
  timeit("rows U=16 (b fastest)", [&] { hipLaunchKernelGGL(k_rows<16>, dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, out); });
  const double wb = bytes * 0.91875;
  auto timeit2 = [&](const char* name, auto launch) {
    for (int i = 0; i < WARM; i++) launch();
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %7.1f GB/s (unique bytes)\n", name, ms / R, wb / (ms / R * 1e-3) / 1e9);
  };
  timeit2("window U=2 no w4", [&] { hipLaunchKernelGGL((k_window<2, false>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  timeit2("window U=2 + w4", [&] { hipLaunchKernelGGL((k_window<2, true>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  timeit2("window U=4 + w4", [&] { hipLaunchKernelGGL((k_window<4, true>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  timeit2("window U=8 + w4", [&] { hipLaunchKernelGGL((k_window<8, true>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  timeit2("lds ring D=4", [&] { hipLaunchKernelGGL((k_lds<4>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  timeit2("lds ring D=8", [&] { hipLaunchKernelGGL((k_lds<8>), dim3(K, N / G), dim3(256), 0, 0, (const float* const*)d, G, F, 1, out); });
  return 0;
}
