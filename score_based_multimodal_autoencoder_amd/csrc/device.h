// device.h -- internal interface between the launcher (kernels.hip), the sampler kernels
// (sampler.hip) and the C ABI (api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.h"

namespace sbm {

// Enqueue every op of `plan` on `st`.  Returns nullptr or a static error string.  No allocation, no
// synchronisation: safe to capture into a hipGraph.
const char* run_plan(const Plan& plan, const float* params_dev, float* ws_dev, const float* ext_in, const float* t_dev,
                     float* ext_out, hipStream_t st);

const char* launch_op(const Plan& plan, const Op& op, const float* params_dev, float* ws_dev, const float* ext_in, float* ext_out,
                      hipStream_t st);

extern unsigned long long* g_gemm_dbg;

// SDE description shared by the step kernels (mirrors oracle/sde.py SDECfg)
struct SdeParams {
  int kind;          // 0 VP, 1 subVP, 2 VE
  float b0, b1;      // beta_min/max or sigma_min/max, rounded to fp32
  float b1_minus_b0; // fl32(b1 - b0) computed in double first (PyTorch wrapped-number promotion)
  float ve_ratio;    // fl32(sigma_max / sigma_min)
  float ve_gconst;   // fl32(sqrt(2 (ln sigma_max - ln sigma_min)))
  float dt;          // fl32(-1/N)
  float sqrt_mdt;    // fl32(sqrt(1/N))
  float snr;         // target_snr
  int N;
  int probability_flow;
};

struct StepArgs {
  const float* x;          // (B, CHW) state
  const float* score;      // (B, CHW)
  const float* noise;      // injected noise base, or null -> in-kernel Philox.  The draw used is
                           // noise + step * noise_step_stride + noise_draw_off (floats), shape (B, CHW)
  long long noise_step_stride, noise_draw_off;
  const float* t;          // (B,) per-sample time, or null -> timesteps[*step_counter]
  const float* timesteps;  // (N,) device
  const int* step_counter; // device scalar (graph-replayable step index)
  const float* alphas;     // (N,) device: 1 - linspace(b0/N, b1/N, N)  (corrector)
  float* x_out; float* mean_out;
  float* norms;            // (B, 2) scratch: per-sample ||score||, ||noise||
  unsigned long long seed; unsigned long long sample_offset;   // Philox key / global sample index of row 0
  int step_base;           // step index used when step_counter is null
  int draw;                // which draw of the step this is (0 = predictor, 1.. = corrector iterations)
  int B, CHW;
  SdeParams sde;
};

void launch_em_step(const StepArgs& a, hipStream_t st);
void launch_langevin_step(const StepArgs& a, hipStream_t st);   // norms + update (two kernels)
void launch_fill_time(float* t_out, const float* timesteps, const int* step_counter, int B, hipStream_t st);
void launch_advance(int* step_counter, hipStream_t st);
void launch_fill_normal(float* out, long long n_per_sample, int B, unsigned long long seed, unsigned long long sample_offset,
                        unsigned step, unsigned draw, float scale, hipStream_t st);
void launch_set_counter(int* step_counter, int value, hipStream_t st);

}  // namespace sbm
