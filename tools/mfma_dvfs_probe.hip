// Probe: which clock does the chip HOLD under a sustained fp32-MFMA load, and does it depend on the MFMA shape?
// (MI355X_MICROARCH.md "DVFS give-back": MFMA-dense loops on random data run well under 2.4 GHz; items 6 and 7.)
// A wave owns a 64x64 output tile and walks K in steps of 16: 32 x v_mfma_f32_32x32x2_f32 or 64 x v_mfma_f32_16x16x4_f32 per step
// (2048 MFMA cycles either way), operands either fixed in registers or re-read from LDS with 8 ds_read_b128 per step (the
// traffic of seg_gemm's 128x128 tile).  Two workgroups of four waves per CU, >= 2 s of back-to-back launches before the
// measured ones, data random or all zero.  In-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_dvfs_probe mfma_dvfs_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* out, unsigned long long* stamps, int iters) {
  __shared__ f32x4 sm[2 * 4 * 256 + 64];                    // [operand][read][thread]
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * 4 * 256; i += 256) sm[i] = reinterpret_cast<const f32x4*>(src)[(blockIdx.x & 7) * 2048 + i];
  __syncthreads();
  f32x4 fa[4], fb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { fa[r] = sm[r * 256 + tid]; fb[r] = sm[(4 + r) * 256 + tid]; }
  f32x16 acc32[4]; f32x4 acc16[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (LDS) {
      const int rot = (it & 15) * 4;                        // a moving address, so the reads cannot be hoisted
#pragma unroll
      for (int r = 0; r < 4; ++r) { fa[r] = sm[r * 256 + ((tid + rot) & 255)]; fb[r] = sm[(4 + r) * 256 + ((tid + rot) & 255)]; }
    }
    if constexpr (SHAPE == 32) {
      // 2 x 2 tiles of 32x32, K = 16 as 8 MFMAs of k = 2 each: fa[2a + h][e] is the operand of row block a, k step 4h + e
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc32[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * a + h][e], fb[2 * b + h][e], acc32[a * 2 + b], 0, 0, 0);
    } else {
      // 4 x 4 tiles of 16x16, K = 16 as 4 MFMAs of k = 4 each
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc16[a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][e], fb[b][e], acc16[a * 4 + b], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 32) { for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r]; }
  else { for (int i = 0; i < 16; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3]; }
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, bool LDS>
void run(const char* name, const float* src, float* out, unsigned long long* st, const char* data) {
  const int blocks = 512, iters = 4096;                     // 2048 cycles x 4096 = 8.4 M cycles per wave and launch, two waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto w0 = std::chrono::steady_clock::now();
  int warm = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 2.5) {
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, src, out, st, iters);
    hipDeviceSynchronize(); warm += 8;
  }
  hipEventRecord(e0);
  const int reps = 16;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, src, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> clk(blocks), cyc(blocks);
  for (int b = 0; b < blocks; ++b) { clk[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0; cyc[b] = (double)h[2 * b] / iters; }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double flop = (double)blocks * 4 * iters * 64.0 * 64.0 * 16.0 * 2.0;
  printf("%-9s %-7s %-6s  %8.2f ms/launch  %6.1f TF/s  in-kernel clock %5.0f MHz (median; min %5.0f max %5.0f)  cycles per K step and wave %6.0f (MFMA floor with two waves per SIMD: 4096)  [%d warm-up launches]\n",
         name, LDS ? "lds" : "regs", data, ms, flop / (ms * 1e-3) / 1e12, clk[blocks / 2], clk[0], clk[blocks - 1], cyc[blocks / 2], warm);
  fflush(stdout);
}

int main() {
  float *src, *out; unsigned long long* st;
  const size_t n = 8 * 2048 * 4;
  hipMalloc(&src, n * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&st, 512 * 16);
  std::vector<float> h(n);
  for (int pass = 0; pass < 2; ++pass) {
    srand(1234);
    for (size_t i = 0; i < n; ++i) h[i] = pass == 0 ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    const char* data = pass == 0 ? "random" : "zeros";
    run<32, false>("32x32x2", src, out, st, data);
    run<16, false>("16x16x4", src, out, st, data);
    run<32, true>("32x32x2", src, out, st, data);
    run<16, true>("16x16x4", src, out, st, data);
  }
  return 0;
}
