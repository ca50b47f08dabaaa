// device.h -- internal interface between the launcher (kernels.hip), the sampler kernels
// (sampler.hip) and the C ABI (api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "plan.h"

namespace sbm {

// erf for the exact GELU (nn.GELU(): unet_model.py:98,108,225).  The ocml erff is ~70 instructions with a divergent range split, and
// the GELU epilogue of a 64x64 tile was 13 k cycles of a 52 k-cycle workgroup; this one is branch-free: both ranges are evaluated
// (|x| < 0.921875: x + x P5(x^2); else 1 - exp(-t Q6(t)), t = min(|x|, 4)) and selected.  Coefficients fitted here (Chebyshev nodes)
// and checked against scipy in float32 arithmetic: |erf error| <= 1.2e-7 over the reals, |GELU error| <= 4.5e-7 at |x| ~ 5 (one ulp of
// the result).  -DSBM_EXACT_ERF restores erff.
__device__ __forceinline__ float sbm_erf(float x) {
#ifdef SBM_EXACT_ERF
  return erff(x);
#else
  const float t = fabsf(x), s = x * x;
  float p = -0.0006008737254887819f;
  p = fmaf(p, s, 0.004997011739760637f); p = fmaf(p, s, -0.026769496500492096f); p = fmaf(p, s, 0.11281904578208923f);
  p = fmaf(p, s, -0.3761250376701355f); p = fmaf(p, s, 0.12837915122509003f);
  const float small = fmaf(p, x, x);
  const float tc = fminf(t, 4.0f);
  float q = 1.6357038475689478e-05f;
  q = fmaf(q, tc, -0.0003699551452882588f); q = fmaf(q, tc, 0.0038104963023215532f); q = fmaf(q, tc, -0.02405124343931675f);
  q = fmaf(q, tc, 0.1064765453338623f); q = fmaf(q, tc, 0.6350749731063843f); q = fmaf(q, tc, 1.1286484003067017f);
  const float e = __builtin_amdgcn_exp2f(-(q * tc) * 1.4426950408889634f);
  const float large = copysignf(1.0f - e, x);
  return t < 0.921875f ? small : large;
#endif
}

// The same function on a PAIR of values with packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per issue slot,
// lane for lane the operations of sbm_erf, hence bit-identical results): the GELU epilogues are VALU-issue-bound (a wave64 instruction
// takes four cycles and two waves share a SIMD), so halving the polynomial's instruction count is worth more than its latency.
typedef float sbm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sbm_f32x2 sbm_fma2(sbm_f32x2 a, sbm_f32x2 b, float c) { return __builtin_elementwise_fma(a, b, (sbm_f32x2){c, c}); }
__device__ __forceinline__ sbm_f32x2 sbm_erf2(sbm_f32x2 x) {
#ifdef SBM_EXACT_ERF
  return (sbm_f32x2){erff(x.x), erff(x.y)};
#else
  const sbm_f32x2 t = __builtin_elementwise_abs(x), s = x * x;
  sbm_f32x2 p = {-0.0006008737254887819f, -0.0006008737254887819f};
  p = sbm_fma2(p, s, 0.004997011739760637f); p = sbm_fma2(p, s, -0.026769496500492096f); p = sbm_fma2(p, s, 0.11281904578208923f);
  p = sbm_fma2(p, s, -0.3761250376701355f); p = sbm_fma2(p, s, 0.12837915122509003f);
  const sbm_f32x2 small = __builtin_elementwise_fma(p, x, x);
  const sbm_f32x2 tc = __builtin_elementwise_min(t, (sbm_f32x2){4.0f, 4.0f});
  sbm_f32x2 q = {1.6357038475689478e-05f, 1.6357038475689478e-05f};
  q = sbm_fma2(q, tc, -0.0003699551452882588f); q = sbm_fma2(q, tc, 0.0038104963023215532f); q = sbm_fma2(q, tc, -0.02405124343931675f);
  q = sbm_fma2(q, tc, 0.1064765453338623f); q = sbm_fma2(q, tc, 0.6350749731063843f); q = sbm_fma2(q, tc, 1.1286484003067017f);
  const sbm_f32x2 a = -(q * tc) * 1.4426950408889634f;
  const float e0 = __builtin_amdgcn_exp2f(a.x), e1 = __builtin_amdgcn_exp2f(a.y);
  sbm_f32x2 r;
  r.x = t.x < 0.921875f ? small.x : copysignf(1.0f - e0, x.x);
  r.y = t.y < 0.921875f ? small.y : copysignf(1.0f - e1, x.y);
  return r;
#endif
}
// GELU(erf) of a pair: 0.5 v (1 + erf(v / sqrt 2)), the operation order of the scalar form
__device__ __forceinline__ sbm_f32x2 sbm_gelu2(sbm_f32x2 v) { return 0.5f * v * (1.0f + sbm_erf2(v * 0.70710678118654752440f)); }

// seg_gemm kernel arguments: a small by-value part used inside the K loop, and the rest in device memory.
struct GemmCold {
  const float* bias;          // [Npad] (+ p * bias_pix_stride) or null
  const float* R;             // residual view origin or null
  const float2* gn_part;      // GroupNorm partial sums of source 1 [gn_nparts][B]
  const float* gn_s; const float* gn_b1;   // folded-GroupNorm tables [P_out][Npad] (plan.h)
  float2* stat_out;           // partial (sum, sumsq) of the stored rows [P * ntiles][B] or null
  long long bias_pix_stride;
  int gn_nparts; float gn_inv_count;
  int ldr, act, ldo, nstore, out_nchw, N, crop_y, crop_x, Hn, Wn, Wout;
};
struct GemmHot {
  const float* A; const float* A2;      // source 1 / 2 view origins
  const float* W; const float* W2;      // [tap][Npad][Kp] / [Npad][Kp2]
  const int4* desc; const int4* hdr;    // host-built chunk tables (plan.h ChunkDesc / PixHdr)
  const GemmCold* cold;
  float* out;                           // PBC view origin, or the external NCHW output
  int B, lda, lda2, Kp, Kp2, Npad, mtiles, ntiles;
  int split; float* part; unsigned* cnt;   // split-K (GemmOp::split): partial tiles and arrival counters in the workspace
  unsigned long long* dbg;              // diagnostic cycle stamps (null in production)
};

// Sampler control block (device memory, one per score net): the step counter every per-step kernel reads, and the time-path
// table -- in the sampling loop all samples share the step's time, so the time MLP and the per-block condition GEMM are
// tabulated once for all N steps and the depthwise kernels index the table by the step counter.
struct StepCtl { int step; int table_mode; int tab_ld; int pad; const float* table; };
struct DwArgs {
  const float* in; float* out; const float* w; const float* bias; const float* cond; float2* stat;
  const StepCtl* ctl;        // null for models without a sampler
  int B, H, W, C, ldi, ldo, cond_ld, cond_col, lanes, spb, slabs;   // cond_col: column of this block inside a table row; lanes = threads per sample, spb = samples per block
};
// the row of time conditions sample b reads: its own row of the per-forward buffer, or the step's row of the table
__device__ __forceinline__ const float* dw_cond_row(const DwArgs& d, int b) {
  if (d.ctl && d.ctl->table_mode) return d.ctl->table + (long long)d.ctl->step * d.ctl->tab_ld + d.cond_col;
  return d.cond + (long long)b * d.cond_ld;
}
struct AttnArgs { const float* qkv; float* out; int B, n, ldq, ldo, heads; float scale; };
struct GnResArgs { const float* u; const float* x; float* out; const float2* part; const float* gamma; const float* beta; int B, P, C, ldu, ldx, ldo, nparts; float inv_count; };
DwArgs make_dw_args(const Plan& plan, const DwOp& o, const float* prm, float* ws, const StepCtl* ctl);
AttnArgs make_attn_args(const AttnOp& o, float* ws);
GnResArgs make_gnres_args(const GnResOp& o, const float* prm, float* ws);

// ---- XCD-resident chain kernel (chain.hip): a run of low-resolution ops executed by ONE launch ----
struct ChainGemm {
  GemmHot g; GemmCold cold;
  unsigned tapmask;          // weight taps source 1 actually reads (zero-padding taps never are): prefetch hint
  int tap_floats;            // floats per tap block = Npad * Kp
  int w2_floats;             // floats of the source-2 weight block (0 if none)
  int wpu;                   // 1: short K (<= 8 chunks per unit): one unit per WAVE, no K split (chain.hip chain_gemm_wpu)
  int rounds;                // unit schedule for a 32-workgroup cluster: sched[r * 32 + slot] = unit or -1, r < rounds
  const int* sched;          // (longest-processing-time assignment by chunk count; null -> plain striding)
};
struct ChainOp {
  int kind; int P; int pad0, pad1;
  union { ChainGemm gm; DwArgs d; AttnArgs a; GnResArgs r; };
};
constexpr int CHAIN_MAX_OPS = 96;   // ops per launch (their descriptors are staged in LDS)
struct ChainSync {          // all zero between launches
  unsigned count[8];        // workgroups registered per XCD
  unsigned arrived;         // total registered
  unsigned done;            // workgroups that finished (the last one resets the registration state)
  unsigned abort_flag;      // set when a barrier timed out (results are then invalid); sticky until the host resets the block
  unsigned pad[5];
  unsigned bar[16 * 16];    // one barrier counter per cluster = (XCD, group), 64 bytes apart
  unsigned left[16 * 16];   // workgroups of the cluster that have finished (the last one resets bar)
};
struct ChainCtl {           // launch-invariant controls, passed by value
  unsigned* abort_host;     // host-mapped word the kernel also raises on a time-out (the API polls it without synchronising)
  unsigned spin_limit;      // barrier polls before giving up (0 = default, ~0.1 s)
  unsigned debug_skip;      // test knob: one workgroup never arrives at its barriers, so that they time out
};

// ---- row kernel (rowrun.hip): the one-position level, four samples per workgroup, activations in LDS ----
constexpr int ROW_R = 4;            // samples per workgroup (= the rows of v_mfma_f32_4x4x1_16b_f32)
constexpr int ROW_STAGE = 100;      // pseudo-op: copy a workspace tensor (and its GroupNorm partial sums) into LDS
struct RowEnt { int w_off; int a_lds; int nch; int src2; };   // weight float offset, LDS offset of the source pixel slab, K chunks of 8, source 2?
struct RowOpD {
  int kind;                          // OP_GEMM, OP_DWCONV7, OP_GN_RES, ROW_STAGE
  int P_out, Npad, nstore, act, ldo, nblk;
  int gn_stat, stat_out;             // LDS statistics slots: consumed (GroupNorm of source 1 / of u), produced; -1 none
  int in_lds, out_lds, out_pix, res_lds, res_pix;
  int ent_first[4], ent_cnt[4], res_p2[4];
  float gn_inv;
  const float* Wt; const float* Wt2; const RowEnt* ent;
  const float* bias; const float* gn_s; const float* gn_b1;
  float* out_g; float2* stat_g;
  DwArgs dw; GnResArgs gr;
  const float* st_src; const float2* st_part; int st_ld, st_C, st_P, st_lds, st_pix, st_stat, st_nparts;
  int pf_next; int pf_lines[7]; const float* pf_ptr[7];   // L2 prefetch (rowrun.hip, prefetch role): this op's weight spans and epilogue vectors (128-byte lines)
};
struct DevPlan;
const char* launch_rowrun(const DevPlan& dp, int first, int count, hipStream_t st);
void rowrun_release(const Plan* plan);
const char* rowrun_prepare(const DevPlan& dp);
extern unsigned long long* g_row_dbg;
int rowrun_program_ops(const DevPlan& dp, int first, int count, int* kinds, int cap);   // diagnostics: kind * 1000 + passes per program op

struct PixState;   // pixrun.hip: fragment-order weights and the programs of a model's low-resolution runs

// A plan resolved against its device arenas.
struct DevPlan {
  const Plan* plan = nullptr;
  const float* prm = nullptr;   // device params arena
  float* ws = nullptr;          // device workspace arena
  std::vector<GemmHot> hot;     // per op (valid for OP_GEMM)
  const float* aux = nullptr;   // per-call auxiliary input (UNetModel z), copied into ws[plan.aux_off] by run_plan
  const ChainOp* chain_ops = nullptr;   // device array, one entry per plan op (filled for chain ops)
  ChainSync* chain_sync = nullptr;
  int chain_grid = 256, chain_block = 512;
  ChainCtl chain_ctl{};
  const StepCtl* ctl = nullptr;         // sampler control block (score nets)
  bool use_pix = false;                 // the chain-level ops run as sample-owned workgroups (pixrun.hip) instead of the chain kernel
  PixState* pix = nullptr;              // ... and their state (owned by the model handle; released by pixrun_release)
};
// pixrun.hip: the 4x4 / 2x2 / 1x1 levels, one sample per workgroup, activations in LDS
struct StepArgs;
const char* launch_pixrun(const DevPlan& dp, int first, int count, hipStream_t st, float* ext_out = nullptr, const StepArgs* em = nullptr);   // ext_out: the plan's external NCHW output, for a run that ends in it; em: the sampler's EM update, applied by the run that ends the plan
bool pixrun_can_fuse_em(const DevPlan& dp);
const char* pixrun_prepare(DevPlan& dp, const float* host_params, double* frag_mb);   // host_params: the plan's packed weights (host copy)
void pixrun_release(DevPlan& dp);
extern unsigned long long* g_pix_dbg;
int pixrun_program_ops(const DevPlan& dp, int first, int count, int* kinds, int cap);   // diagnostics: kind * 1000 + taps per program op
const char* launch_chain(const DevPlan& dp, int first, int count, hipStream_t st, float* ext_out = nullptr, const StepArgs* em = nullptr);
const char* chain_launch_geometry(int cus, int* grid, int* block);   // verifies that the whole grid can be resident
int chain_variant_groups();
bool qkv_linattn_fusable(const Plan& plan, int i);            // ops i (to_qkv GEMM) and i + 1 (LinearAttention core)
const char* launch_qkv_linattn(const DevPlan& dp, int i, hipStream_t st);
bool attn_block_fusable(const Plan& plan, int i);             // ops i .. i + 3: the whole block as one attnblock_kernel launch
const char* launch_attn_block(const DevPlan& dp, int i, hipStream_t st);
void pack_attn_block_fragments(Plan* plan, int i);            // host: fragment-order copies of to_qkv's and to_out.0's weights (GemmOp::wt_off of ops i, i + 2)
const char* make_gemm_args(const Plan& plan, const GemmOp& o, const float* prm, float* ws, const GemmCold* cold_dev, GemmHot* hot, GemmCold* cold);

// Enqueue every op of the plan on `st`.  Returns nullptr or a static error string.  No allocation, no
// synchronisation: safe to capture into a hipGraph.
const char* run_plan(const DevPlan& dp, const float* ext_in, const float* t_dev, float* ext_out, hipStream_t st, int time_ops = 0, const StepArgs* em = nullptr);   // em: see launch_pixrun   // time_ops: 0 all ops, 1 skip the time path, 2 only the time path; + 4: skip the input conversion op
const char* launch_op(const DevPlan& dp, int op_index, const float* ext_in, float* ext_out, hipStream_t st);

extern unsigned long long* g_gemm_dbg;
extern unsigned long long* g_chain_dbg;

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
  const double* norm_sums; // exact sharded corrector: (2,) sums of the per-sample norms over the GLOBAL batch (null: reduce the local batch)
  long long B_global;      // ... and the global batch size the means divide by
  unsigned long long seed; unsigned long long sample_offset;   // Philox key / global sample index of row 0
  int step_base;           // step index used when step_counter is null
  int draw;                // which draw of the step this is (0 = predictor, 1.. = corrector iterations)
  int B, CHW;
  SdeParams sde;
  // Sampler loop only.  pbc_out: the new state is ALSO stored in the score net's converted input layout (the plan's NCHW -> PBC op, row-packed
  // or plain), so the next step's forward skips its conversion launch.  adv_counter: the LAST workgroup of this launch to finish advances
  // the step counter (every workgroup has read it by then), which replaces the one-thread advance launch behind the step.
  float* pbc_out; int pbc_ld, pbc_Cv, pbc_rowpack, pbc_H, pbc_W;
  int* adv_counter; unsigned* adv_ticket;
};

// classifier / energy guidance MLP (guidance.hip): forward + input-VJP of cl_out.mean() + in-place score update
struct ClgArgs {
  const float* x; int C, per_mod, ch1, ch2;      // joint state (B, C, per_mod): the pair is x[:, ch1], x[:, ch2] ...
  const float* x_pair;                           // ... or an explicit (B, 2 * size_z) matrix (forward(x, t) of the module)
  const float* t; int id1, id2;
  float* score; float cl_s; unsigned upd_mask;   // score (B, C, per_mod) updated in place: bit 0 -> ch1, bit 1 -> ch2
  float* grad_out;                               // (B, 2 * size_z) or null
  float* out;                                    // (B, n_class) or null
  const float* Wt_T; const float* bt; const float* emb; const float* W1; const float* W1_T; const float* b1;
  const float* W2; const float* W2_T; const float* b2; const float* W3; const float* b3; const float* freq;
  int B, size_z, n_class, D0, n_emb;
};
const char* launch_clg(const ClgArgs& a, hipStream_t st);
void clg_prepare();
std::vector<TensorSpec> clg_tensor_specs(int variant, int n_mod, int size_z, int n_class);
bool clg_pack(int variant, int n_mod, int size_z, int n_class, const float* const* t, std::vector<float>* out, long long* off, std::string* err);

void launch_em_step(const StepArgs& a, hipStream_t st);
void launch_ddpm_step(const StepArgs& a, float sqrt_recip_alpha, float beta, float sqrt_one_minus_abar, float sigma, hipStream_t st);
void launch_langevin_step(const StepArgs& a, hipStream_t st);   // norms + update (two kernels)
void launch_langevin_norms(const StepArgs& a, double* sums_out, hipStream_t st);   // norms + their fp64 sums (2,)
void launch_langevin_update(const StepArgs& a, hipStream_t st);  // update alone (a.norm_sums set, or a.norms filled)
void launch_fill_time(float* t_out, const float* timesteps, const int* step_counter, int B, hipStream_t st);
void launch_advance(int* step_counter, hipStream_t st);
void launch_fill_normal(float* out, long long n_per_sample, int B, unsigned long long seed, unsigned long long sample_offset,
                        unsigned step, unsigned draw, float scale, hipStream_t st);
void arm_profile_events(hipEvent_t start, hipEvent_t stop);
// Raise a kernel's dynamic-LDS limit once per (function, device): the attribute is per device, so a process-wide flag would leave a second
// device at the 64 KB default.  Cheap when already done (one device query + a set lookup under a mutex).
void ensure_dynamic_lds(const void* fn, int bytes);
void prepare_layer_kernels();          // kernels.hip: dynamic-LDS attributes of the per-layer kernels (called by sbm_model_create)
extern thread_local hipEvent_t g_prof_start, g_prof_stop;   // armed by arm_profile_events (kernels.hip)
void launch_set_counter(int* step_counter, int value, hipStream_t st);
void launch_axpy_noise(const StepArgs& a, float ca, float cb, float sdiv, unsigned mod_mask, int per_mod, hipStream_t st);
void launch_axpy_noise_tab(const StepArgs& a, const float* coef, unsigned mod_mask, int n_mod, int per_mod, hipStream_t st);
void launch_renoise(const StepArgs& a, const float* z_obs, unsigned obs_mask, int per_mod, hipStream_t st, int plain = 0);   // plain: copy the clean observed latents

}  // namespace sbm
