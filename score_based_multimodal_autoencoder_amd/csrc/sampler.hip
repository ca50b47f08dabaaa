// sampler.hip -- elementwise kernels of the reverse-SDE sampling loop (HBM-bound, 16-byte coalesced).
//
//   em_step        em_predictor + RSDE.sde   (reference sde_helper2.py:45-52, 277-317, 352-356, 402-407, 445-450)
//   langevin_step  one corrector iteration   (reference sde_helper2.py:96-101)
//   philox normal  counter-based N(0,1) keyed by (seed, GLOBAL sample index, step, draw): the stream a
//                  sample sees does not depend on how the batch is sharded over GPUs.
// The arithmetic follows the reference's operator order with FMA contraction disabled, so that the
// update agrees with PyTorch's unfused fp32 ops to the last bit wherever libm is not involved.
#include <hip/hip_runtime.h>

#include "device.h"
#include "sampler_dev.h"

namespace sbm {

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void em_step_kernel(const StepArgs a) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per thread
  const int q4 = a.CHW >> 2;
  const int stepi = a.step_counter ? *a.step_counter : a.step_base;
  if (idx >= (long long)a.B * q4) { if (a.adv_counter) advance_by_last_workgroup(a, stepi); return; }
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  const float t = a.t ? a.t[b] : a.timesteps[stepi];
  const Coef c = sde_coef(a.sde, t);
  const float4 x = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  const float4 s = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!a.sde.probability_flow) z = load_noise4(a, b, e, (unsigned)stepi);
  float4 m, o;
  em_update4(a, c, x, s, z, &m, &o);
  *reinterpret_cast<float4*>(a.mean_out + (long long)b * a.CHW + e) = m;
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
  if (a.pbc_out) store_pbc(a, b, e, o);
  if (a.adv_counter) advance_by_last_workgroup(a, stepi);
}

// per-sample L2 norms of the score and of the corrector's noise: one workgroup per sample
__global__ __launch_bounds__(256) void norms_kernel(const StepArgs a) {
  __shared__ float rs[256], rn[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int stepi = a.step_counter ? *a.step_counter : a.step_base;
  float s = 0.f, n = 0.f;
  for (int e = tid * 4; e < a.CHW; e += 1024) {
    const float4 g = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
    const float4 z = load_noise4(a, b, e, (unsigned)stepi);
    s += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    n += z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
  }
  rs[tid] = s; rn[tid] = n;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if (tid < o) { rs[tid] += rs[tid + o]; rn[tid] += rn[tid + o]; } __syncthreads(); }
  if (tid == 0) { a.norms[2 * b] = sqrtf(rs[0]); a.norms[2 * b + 1] = sqrtf(rn[0]); }
}

__global__ __launch_bounds__(256) void langevin_update_kernel(const StepArgs a) {
  __shared__ float rs[256], rn[256];
  const int tid = threadIdx.x;
  // batch-mean norms (sde_helper2.py:97-98): every workgroup reduces the B per-sample norms itself
  float gnorm, nnorm;
  if (a.norm_sums) {       // exact sharded mode: the batch means are over the GLOBAL batch (sums all-reduced by the host side)
    gnorm = (float)(a.norm_sums[0] / (double)a.B_global); nnorm = (float)(a.norm_sums[1] / (double)a.B_global);
  } else {
    float s = 0.f, n = 0.f;
    for (int i = tid; i < a.B; i += 256) { s += a.norms[2 * i]; n += a.norms[2 * i + 1]; }
    rs[tid] = s; rn[tid] = n;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) { rs[tid] += rs[tid + o]; rn[tid] += rn[tid + o]; } __syncthreads(); }
    gnorm = rs[0] / (float)a.B; nnorm = rn[0] / (float)a.B;
  }
  const long long idx = (long long)blockIdx.x * blockDim.x + tid;
  const int q4 = a.CHW >> 2;
  const int stepi = a.step_counter ? *a.step_counter : a.step_base;
  if (idx >= (long long)a.B * q4) { if (a.adv_counter) advance_by_last_workgroup(a, stepi); return; }
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  const float t = a.t ? a.t[b] : a.timesteps[stepi];
  float alpha = 1.0f;
  if (a.sde.kind != 2) {                                            // sde_helper2.py:56-60
    int ai = (int)((t * (float)(a.sde.N - 1)) / 1.0f);
    ai = ai < 0 ? 0 : (ai > a.sde.N - 1 ? a.sde.N - 1 : ai);
    alpha = a.alphas[ai];
  }
  const float r = (a.sde.snr * nnorm) / gnorm;
  const float step = ((r * r) * 2.0f) * alpha;                      // sde_helper2.py:99
  const float sq = sqrtf(step * 2.0f);
  const float4 x = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  const float4 g = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
  const float4 z = load_noise4(a, b, e, (unsigned)stepi);
  float4 m, o;
  m.x = x.x + step * g.x; m.y = x.y + step * g.y; m.z = x.z + step * g.z; m.w = x.w + step * g.w;
  o.x = m.x + sq * z.x; o.y = m.y + sq * z.y; o.z = m.z + sq * z.z; o.w = m.w + sq * z.w;
  *reinterpret_cast<float4*>(a.mean_out + (long long)b * a.CHW + e) = m;
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
  if (a.pbc_out) store_pbc(a, b, e, o);
  if (a.adv_counter) advance_by_last_workgroup(a, stepi);
}

// Fixed-step Langevin update of the alternative samplers: x_out = x + mask_c * (a * score + b * noise) per modality
// channel c (annealed Langevin fid_upd10.py:93-97, score/sigma first, a = alpha, b = c*sqrt(2 alpha); MLP-score Langevin
// fid_upd10.py:331, a = lr1, b = lr2).  Observed modalities (mask bit clear) pass through unchanged.
__global__ __launch_bounds__(256) void axpy_noise_kernel(const StepArgs a, float ca, float cb, float sdiv, unsigned mod_mask, int per_mod) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q4 = a.CHW >> 2;
  if (idx >= (long long)a.B * q4) return;
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  const float4 x = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  float4 o = x;
  if ((mod_mask >> (e / per_mod)) & 1u) {
    float4 s = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
    s.x = s.x / sdiv; s.y = s.y / sdiv; s.z = s.z / sdiv; s.w = s.w / sdiv;             // score / sigma_i (fid_upd10.py:93); exact for sdiv = 1
    const float4 z = load_noise4(a, b, e, (unsigned)a.step_base);
    o.x = (x.x + ca * s.x) + cb * z.x; o.y = (x.y + ca * s.y) + cb * z.y; o.z = (x.z + ca * s.z) + cb * z.z; o.w = (x.w + ca * s.w) + cb * z.w;
  }
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
}
// The same update inside a device-resident loop (sbm_langevin_run): the iteration index is the device step counter, and the iteration's
// coefficients come from a table row [a_0 .. a_{n_mod-1} | b_0 .. b_{n_mod-1} | score_div] -- one captured launch serves every iteration.
__global__ __launch_bounds__(256) void axpy_noise_tab_kernel(const StepArgs a, const float* __restrict__ coef, unsigned mod_mask, int n_mod, int per_mod) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q4 = a.CHW >> 2;
  if (idx >= (long long)a.B * q4) return;
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  const int step = *a.step_counter, mod = e / per_mod;
  const float* row = coef + (long long)step * (2 * n_mod + 1);
  const float4 x = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  float4 o = x;
  if ((mod_mask >> mod) & 1u) {
    const float ca = row[mod], cb = row[n_mod + mod], sdiv = row[2 * n_mod];
    float4 s = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
    s.x = s.x / sdiv; s.y = s.y / sdiv; s.z = s.z / sdiv; s.w = s.w / sdiv;
    const float4 z = load_noise4(a, b, e, (unsigned)step);
    o.x = (x.x + ca * s.x) + cb * z.x; o.y = (x.y + ca * s.y) + cb * z.y; o.z = (x.z + ca * s.z) + cb * z.z; o.w = (x.w + ca * s.w) + cb * z.w;
  }
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
}
void launch_axpy_noise_tab(const StepArgs& a, const float* coef, unsigned mod_mask, int n_mod, int per_mod, hipStream_t st) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(axpy_noise_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, coef, mod_mask, n_mod, per_mod);
}
void launch_axpy_noise(const StepArgs& a, float ca, float cb, float sdiv, unsigned mod_mask, int per_mod, hipStream_t st) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(axpy_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, ca, cb, sdiv, mod_mask, per_mod);
}

// One ancestral step of the DiffuseVAE refiner's DDPM loop (eval_lat_celeba_hq_all.py:107-118):
//   mean = sqrt_recip_alpha_t * (x - beta_t * eps / sqrt(1 - alphabar_t));  x_out = mean + sigma_t * z   (sigma_t = 0 at t = 0: no draw)
__global__ __launch_bounds__(256) void ddpm_step_kernel(const StepArgs a, float r, float beta, float s1m, float sigma) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q4 = a.CHW >> 2;
  if (idx >= (long long)a.B * q4) return;
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  const float4 x = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  const float4 s = *reinterpret_cast<const float4*>(a.score + (long long)b * a.CHW + e);
  float4 o;
  o.x = r * (x.x - (beta * s.x) / s1m); o.y = r * (x.y - (beta * s.y) / s1m); o.z = r * (x.z - (beta * s.z) / s1m); o.w = r * (x.w - (beta * s.w) / s1m);
  if (sigma != 0.f) {
    const float4 z = load_noise4(a, b, e, (unsigned)a.step_base);
    o.x += sigma * z.x; o.y += sigma * z.y; o.z += sigma * z.z; o.w += sigma * z.w;
  }
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
}
void launch_ddpm_step(const StepArgs& a, float r, float beta, float s1m, float sigma, hipStream_t st) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, r, beta, s1m, sigma);
}

// Re-noising of the observed modalities in the conditional loop (train_poly_unet_cont.py:356-364,
// train_lat_celebhq_unet_cont2.py:293-301): x_out[:, c] = mean_coef(t) * z_obs[:, c] + std(t) * z_obs[:, c] for observed c
// (the reference multiplies std by z itself, not by fresh noise -- reproduced knowingly), x[:, c] otherwise.
__global__ __launch_bounds__(256) void renoise_kernel(const StepArgs a, const float* z_obs, unsigned obs_mask, int per_mod, int plain) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q4 = a.CHW >> 2;
  if (idx >= (long long)a.B * q4) return;
  const int b = (int)(idx / q4), e = (int)(idx - (long long)b * q4) * 4;
  float4 o = *reinterpret_cast<const float4*>(a.x + (long long)b * a.CHW + e);
  if ((obs_mask >> (e / per_mod)) & 1u) {
    float mc = 1.f, sd = 0.f;
    const float t = plain ? 0.f : (a.t ? a.t[b] : a.timesteps[a.step_counter ? *a.step_counter : a.step_base]);
    if (plain) { /* noise_obs off, and the final write-back: the clean observed latent (train_lat_celebhq_unet_cont2.py:300-301) */ }
    else if (a.sde.kind == 2) sd = a.sde.b0 * powf(a.sde.ve_ratio, t);
    else {
      const float lmc = (-0.25f * (t * t)) * a.sde.b1_minus_b0 - (0.5f * t) * a.sde.b0;     // sde_helper2.py:359
      mc = expf(lmc);
      const float e2 = expf(2.0f * lmc);
      sd = a.sde.kind == 0 ? sqrtf(1.0f - e2) : 1.0f - e2;
    }
    const float4 z = *reinterpret_cast<const float4*>(z_obs + (long long)b * a.CHW + e);
    if (plain) o = z;
    else { o.x = mc * z.x + sd * z.x; o.y = mc * z.y + sd * z.y; o.z = mc * z.z + sd * z.z; o.w = mc * z.w + sd * z.w; }
  }
  *reinterpret_cast<float4*>(a.x_out + (long long)b * a.CHW + e) = o;
}
void launch_renoise(const StepArgs& a, const float* z_obs, unsigned obs_mask, int per_mod, hipStream_t st, int plain) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(renoise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, z_obs, obs_mask, per_mod, plain);
}

__global__ void fill_time_kernel(float* t_out, const float* timesteps, const int* step_counter, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) t_out[i] = timesteps[*step_counter];
}
__global__ void advance_kernel(int* step_counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *step_counter += 1; }

__global__ void fill_normal_kernel(float* out, long long q4, int B, unsigned long long seed, unsigned long long sample_offset, unsigned step, unsigned draw, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * q4) return;
  const int b = (int)(idx / q4); const long long e4 = idx - (long long)b * q4;
  float4 z = normal4(seed, sample_offset + (unsigned long long)b, (unsigned)e4, step, draw);
  z.x *= scale; z.y *= scale; z.z *= scale; z.w *= scale;
  *reinterpret_cast<float4*>(out + ((long long)b * q4 + e4) * 4) = z;
}

void launch_em_step(const StepArgs& a, hipStream_t st) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(em_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
void launch_langevin_step(const StepArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(norms_kernel, dim3(a.B), dim3(256), 0, st, a);
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
// Exact sharded corrector (sde_helper2.py:97-99 takes the MEAN over the whole batch): a shard's part of the two sums in fp64.
// A double holds the sum of a few thousand floats of similar magnitude exactly, so the total does not depend on how the batch
// is split nor on the order of the additions: sharded and unsharded runs agree bit for bit.
__global__ __launch_bounds__(256) void norm_sums_kernel(const float* norms, int B, double* out) {
  __shared__ double rs[256], rn[256];
  const int tid = threadIdx.x;
  double s = 0.0, n = 0.0;
  for (int i = tid; i < B; i += 256) { s += (double)norms[2 * i]; n += (double)norms[2 * i + 1]; }
  rs[tid] = s; rn[tid] = n;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if (tid < o) { rs[tid] += rs[tid + o]; rn[tid] += rn[tid + o]; } __syncthreads(); }
  if (tid == 0) { out[0] = rs[0]; out[1] = rn[0]; }
}
void launch_langevin_norms(const StepArgs& a, double* sums_out, hipStream_t st) {
  hipLaunchKernelGGL(norms_kernel, dim3(a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(norm_sums_kernel, dim3(1), dim3(256), 0, st, a.norms, a.B, sums_out);
}
void launch_langevin_update(const StepArgs& a, hipStream_t st) {
  const long long n = (long long)a.B * (a.CHW >> 2);
  hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
void launch_fill_time(float* t_out, const float* timesteps, const int* step_counter, int B, hipStream_t st) {
  hipLaunchKernelGGL(fill_time_kernel, dim3((B + 255) / 256), dim3(256), 0, st, t_out, timesteps, step_counter, B);
}
__global__ void set_counter_kernel(int* step_counter, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *step_counter = v; }
void launch_set_counter(int* step_counter, int value, hipStream_t st) { hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(64), 0, st, step_counter, value); }
void launch_advance(int* step_counter, hipStream_t st) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, step_counter); }
void launch_fill_normal(float* out, long long n_per_sample, int B, unsigned long long seed, unsigned long long sample_offset, unsigned step, unsigned draw, float scale, hipStream_t st) {
  const long long q4 = n_per_sample >> 2, n = (long long)B * q4;
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, q4, B, seed, sample_offset, step, draw, scale);
}

}  // namespace sbm
