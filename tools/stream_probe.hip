// Probe for the planned sample-owned low-resolution kernel: every workgroup streams the SAME weight arena
// (as all samples need all weights) from L2 / Infinity Cache while feeding v_mfma_f32_16x16x4_f32 at the planned intensity:
// per wave and chunk 4 x global_load_dwordx4 (4 KB) -> 16 MFMAs.  Reports the time to walk `mb` MB per workgroup for several grids.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4v gf32x4v;
constexpr int RING = 4;
template <int MFMAS>
__global__ __launch_bounds__(512, 1) void stream(const float* __restrict__ w, long long chunks, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  const float a = threadIdx.x * 1e-3f;
  // wave w takes chunks w, w + 8, ...; a chunk = 4 rows of 1 KB (64 lanes x 16 B)
  f32x4v ring[RING][4];
  long long c = wave;
  auto issue = [&](f32x4v* r, long long ci) {
    const float* p = w + ci * 1024 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *(const gf32x4v*)(unsigned long long)(p + j * 256);
  };
#pragma unroll
  for (int r = 0; r < RING; ++r) if (c + 8 * r < chunks) issue(ring[r], c + 8 * r);
  for (; c < chunks; c += 8 * RING) {
#pragma unroll
    for (int r = 0; r < RING; ++r) {
      if (c + 8 * r < chunks) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < MFMAS / 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[r][j][i & 3], acc[i], 0, 0, 0);
        if (c + 8 * (r + RING) < chunks) issue(ring[r], c + 8 * (r + RING));
      }
    }
  }
  float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 9.5;
  const long long chunks = (long long)(mb * 1e6 / 4096);
  float* w; float* out;
  hipMalloc(&w, chunks * 4096 + 4096); hipMemset(w, 0, chunks * 4096 + 4096); hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {64, 128, 256}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(stream<16>, dim3(grid), dim3(512), 0, 0, w, chunks, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("16 MFMA/chunk grid %3d: %.1f us for %.1f MB per workgroup = %.1f GB/s per workgroup, %.2f TB/s aggregate, %.1f TFLOP/s\n", grid, ms * 1e3, mb,
                           mb * 1e6 / (ms * 1e-3) / 1e9, grid * mb * 1e6 / (ms * 1e-3) / 1e12, grid * (double)chunks * 16 * 2048 / (ms * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(stream<4>, dim3(grid), dim3(512), 0, 0, w, chunks, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf(" 4 MFMA/chunk grid %3d: %.1f us = %.1f GB/s per workgroup, %.2f TB/s aggregate\n", grid, ms * 1e3, mb * 1e6 / (ms * 1e-3) / 1e9, grid * mb * 1e6 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
