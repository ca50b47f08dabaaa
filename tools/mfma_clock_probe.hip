// Probe: raw v_mfma_f32_32x32x2_f32 rate and the in-kernel shader clock, at several grid sizes.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock_probe mfma_clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int nacc>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* stamps, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 8
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    if constexpr (nacc > 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    if constexpr (nacc > 2) { a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int nacc>
__global__ __launch_bounds__(256) void probe16(float* out, unsigned long long* stamps, int iters) {
  f32x4v a[8];
  for (int k = 0; k < 8; ++k) a[k] = f32x4v{0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < nacc; ++k) a[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[k], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int k = 0; k < nacc; ++k) s += a[k][0] + a[k][1] + a[k][2] + a[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
typedef double f64x4v __attribute__((ext_vector_type(4)));
template <int nacc>
__global__ __launch_bounds__(256) void probe64(float* out, unsigned long long* stamps, int iters) {
  f64x4v a[8];
  for (int k = 0; k < 8; ++k) a[k] = f64x4v{0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3 + 1.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < nacc; ++k) a[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[k], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int k = 0; k < nacc; ++k) s += a[k][0] + a[k][1] + a[k][2] + a[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&st, 4096 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nacc : {1, 2, 4}) for (int blocks : {256, 512, 1024}) for (int iters : {8192}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); if (nacc == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, st, iters); else if (nacc == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, st, iters); else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, st, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
      double clk = (double)h[0] / (double)h[1] * 100.0;   // MHz
      double mf = (double)blocks * 4 * iters * nacc, flop = mf * 4096.0;
      if (rep == 2) printf("nacc=%d blocks=%5d iters=%6d  %9.1f us  %7.1f TF/s  cyc/mfma/wave=%6.1f  in-kernel clock %6.0f MHz\n", nacc, blocks, iters, ms * 1e3, flop / (ms * 1e-3) / 1e12,
             (double)h[0] / ((double)iters * nacc), clk);
    }
  }
  for (int nacc : {1, 2, 4, 8}) for (int blocks : {256, 1024}) {
    int iters = 8192;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (nacc == 1) hipLaunchKernelGGL(probe16<1>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      else if (nacc == 2) hipLaunchKernelGGL(probe16<2>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      else if (nacc == 4) hipLaunchKernelGGL(probe16<4>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      else hipLaunchKernelGGL(probe16<8>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
      double mf = (double)blocks * 4 * iters * nacc, flop = mf * 2048.0;
      if (rep == 2) printf("16x16x4 nacc=%d blocks=%5d  %9.1f us  %7.1f TF/s  cyc/mfma/wave=%6.1f  clock %6.0f MHz\n", nacc, blocks, ms * 1e3, flop / (ms * 1e-3) / 1e12,
             (double)h[0] / ((double)iters * nacc), (double)h[0] / (double)h[1] * 100.0);
    }
  }
  for (int nacc : {1, 4, 8}) for (int blocks : {256, 1024}) {
    int iters = 8192;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (nacc == 1) hipLaunchKernelGGL(probe64<1>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      else if (nacc == 4) hipLaunchKernelGGL(probe64<4>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      else hipLaunchKernelGGL(probe64<8>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
      double mf = (double)blocks * 4 * iters * nacc, flop = mf * 2048.0;
      if (rep == 2) printf("f64 16x16x4 nacc=%d blocks=%5d  %9.1f us  %7.1f TF/s  cyc/mfma/wave=%6.1f  clock %6.0f MHz\n", nacc, blocks, ms * 1e3, flop / (ms * 1e-3) / 1e12,
             (double)h[0] / ((double)iters * nacc), (double)h[0] / (double)h[1] * 100.0);
    }
  }
  return 0;
}
