// stand-alone timing probe for the 8x8 linear-attention core: phases timed with s_memtime in block 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct AttnArgs { const float* qkv; float* out; int B, n, ldq, ldo, heads; float scale; };
__global__ __launch_bounds__(256) void k(const AttnArgs a, unsigned long long* st) {
  __shared__ __attribute__((aligned(16))) float sq[64 * 32], sk[64 * 32], sv[64 * 32];
  __shared__ float pctx[4][32][33];
  __shared__ float qs[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  const float* base = a.qkv + (long long)b * a.ldq + h * 32;
  const long long pstride = (long long)a.B * a.ldq;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int idx = tid; idx < a.n * 8; idx += 256) {
    const int i = idx >> 3, c4 = (idx & 7) * 4;
    const float* r = base + i * pstride + c4;
    reinterpret_cast<float4*>(sq)[idx] = *reinterpret_cast<const float4*>(r);
    reinterpret_cast<float4*>(sk)[idx] = *reinterpret_cast<const float4*>(r + hd);
    reinterpret_cast<float4*>(sv)[idx] = *reinterpret_cast<const float4*>(r + 2 * hd);
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float kmax = -INFINITY;
  for (int i = half; i < a.n; i += 2) kmax = fmaxf(kmax, sk[i * 32 + d]);
  kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
  float ksum = 0.f;
  for (int i = half; i < a.n; i += 2) ksum += expf(sk[i * 32 + d] - kmax);
  ksum += __shfl_xor(ksum, 32);
  const float kinv = 1.0f / ksum;
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  const int per = (a.n + 3) / 4, i_lo = wave * per, i_hi = min(a.n, i_lo + per);
  float c[16];
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  for (int i = i_lo; i < i_hi; ++i) {
    const float kv = expf(sk[i * 32 + d] - kmax) * kinv;
    const float* vrow = sv + i * 32 + half * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = fmaf(kv, vrow[e], c[e]);
  }
  for (int e = 0; e < 16; ++e) pctx[wave][d][half * 16 + e] = c[e];
  __syncthreads();
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  float col[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) col[dd] = ((pctx[0][dd][d] + pctx[1][dd][d]) + pctx[2][dd][d]) + pctx[3][dd][d];
  for (int i0 = i_lo; i0 < i_hi; i0 += 2) {
    const int i = i0 + half;
    const bool ok = i < i_hi;
    const float qv = ok ? sq[i * 32 + d] : 0.f;
    float m = qv;
    for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e = expf(qv - m);
    float s = e;
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    qs[wave][half][d] = e / s * a.scale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float o = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) o = fmaf(col[dd], qs[wave][half][dd], o);
    if (ok) a.out[((long long)i * a.B + b) * a.ldo + h * 32 + d] = o;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 512) { st[0] = t1 - t0; st[1] = t2 - t1; st[2] = t3 - t2; st[3] = t4 - t3; }
}
int main() {
  const int B = 256, n = 64, heads = 4, ldq = 384, ldo = 128;
  float *qkv, *out; unsigned long long* st;
  CK(hipMalloc(&qkv, (size_t)n * B * ldq * 4)); CK(hipMalloc(&out, (size_t)n * B * ldo * 4)); CK(hipMalloc(&st, 64));
  std::vector<float> h((size_t)n * B * ldq); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  AttnArgs a{qkv, out, B, n, ldq, ldo, heads, 0.1767767f};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(B * heads), dim3(256), 0, 0, a, st);
  hipEventRecord(e0); for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(B * heads), dim3(256), 0, 0, a, st); hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long s4[4]; CK(hipMemcpy(s4, st, 32, hipMemcpyDeviceToHost));
  printf("%.2f us per launch; block 512 cycles: stage %llu, k-stats %llu, ctx %llu, out %llu\n", ms * 1000 / 20, s4[0], s4[1], s4[2], s4[3]);
  return 0;
}
