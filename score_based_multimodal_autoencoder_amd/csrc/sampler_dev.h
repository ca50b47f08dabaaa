// sampler_dev.h -- device helpers of the sampling loop's update arithmetic, shared by sampler.hip (the update kernels) and pixrun.hip (the EM
// predictor applied by the kernel that ends the score net, round 4).  Every function that does floating-point arithmetic switches FMA
// contraction off in its own body: the update agrees with PyTorch's unfused fp32 ops to the last bit wherever libm is not involved, and it is
// the same bits whichever kernel applies it.
#pragma once
#include <hip/hip_runtime.h>

#include "device.h"

namespace sbm {

// ---- Philox4x32-10 ------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four N(0,1) for elements [4*e4, 4*e4+4) of global sample `gs` at (step, draw)
__device__ __forceinline__ float4 normal4(unsigned long long seed, unsigned long long gs, unsigned e4, unsigned step, unsigned draw) {
#pragma clang fp contract(off)
  unsigned r[4];
  philox4x32_10(e4, (unsigned)gs, (unsigned)(gs >> 32) ^ (draw << 24), step, (unsigned)seed, (unsigned)(seed >> 32), r);
  const float u0 = ((r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (r[1] >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (r[3] >> 8) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float sa, ca, sb, cb;
  sincospif(2.0f * u1, &sa, &ca); sincospif(2.0f * u3, &sb, &cb);
  return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

__device__ __forceinline__ float4 load_noise4(const StepArgs& a, int b, int e, unsigned step) {
  if (a.noise) return *reinterpret_cast<const float4*>(a.noise + (long long)step * a.noise_step_stride + a.noise_draw_off + (long long)b * a.CHW + e);
  return normal4(a.seed, a.sample_offset + (unsigned long long)b, (unsigned)(e >> 2), step, (unsigned)a.draw);
}

// ---- SDE coefficients (per sample) ---------------------------------------------------------------
struct Coef { float dcoef, g; };
__device__ __forceinline__ Coef sde_coef(const SdeParams& s, float t) {
#pragma clang fp contract(off)
  Coef c;
  if (s.kind == 2) {                        // VESDE.sde sde_helper2.py:445-450
    const float sigma = s.b0 * powf(s.ve_ratio, t);
    c.dcoef = 0.f; c.g = sigma * s.ve_gconst;
    return c;
  }
  const float beta_t = s.b0 + t * s.b1_minus_b0;       // VPSDE.sde sde_helper2.py:352-356
  c.dcoef = -0.5f * beta_t;
  if (s.kind == 0) c.g = sqrtf(beta_t);
  else {                                               // subVPSDE.sde sde_helper2.py:402-407
    const float e = expf((-2.0f * s.b0) * t - s.b1_minus_b0 * (t * t));
    c.g = sqrtf(beta_t * (1.0f - e));
  }
  return c;
}

// the new state in the score net's input layout: element e = (c, y, x .. x + 3) of sample b (W % 4 == 0, no spatial padding)
__device__ __forceinline__ void store_pbc(const StepArgs& a, int b, int e, const float4 o) {
  const int HW = a.pbc_H * a.pbc_W, c = e / HW, r = e - c * HW, y = r / a.pbc_W, x = r - y * a.pbc_W;
  if (a.pbc_rowpack) {                     // one "pixel" per image row, channel x * Cv + c (plan.h CvtOp::rowpack)
    float* d = a.pbc_out + ((long long)y * a.B + b) * a.pbc_ld + x * a.pbc_Cv + c;
    d[0] = o.x; d[a.pbc_Cv] = o.y; d[2 * a.pbc_Cv] = o.z; d[3 * a.pbc_Cv] = o.w;
  } else {
    float* d = a.pbc_out + ((long long)(y * a.pbc_W + x) * a.B + b) * a.pbc_ld + c;
    const long long ps = (long long)a.B * a.pbc_ld;
    d[0] = o.x; d[ps] = o.y; d[2 * ps] = o.z; d[3 * ps] = o.w;
  }
}
// Every workgroup of the launch has read the step counter when it arrives here; the last one to arrive moves it on and re-arms the ticket.
__device__ __forceinline__ void advance_by_last_workgroup(const StepArgs& a, int stepi) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(a.adv_ticket, 1u);
    if (t == gridDim.x - 1) { *a.adv_ticket = 0u; *a.adv_counter = stepi + 1; }
  }
}

// em_predictor + RSDE.sde on four elements (reference sde_helper2.py:45-52, 277-317): the new mean and the new state
__device__ __forceinline__ void em_update4(const StepArgs& a, const Coef c, const float4 x, const float4 s, const float4 z, float4* mean, float4* out) {
#pragma clang fp contract(off)
  const float g2 = c.g * c.g;
  const float pf = a.sde.probability_flow ? 0.5f : 1.0f;
  const float gz = a.sde.probability_flow ? 0.0f : c.g * a.sde.sqrt_mdt;
  float4 m, o;
#define SBM_EM(C) { const float drift = (c.dcoef * x.C) - (g2 * s.C) * pf; m.C = x.C + drift * a.sde.dt; o.C = m.C + gz * z.C; }
  SBM_EM(x) SBM_EM(y) SBM_EM(z) SBM_EM(w)
#undef SBM_EM
  *mean = m; *out = o;
}

}  // namespace sbm
