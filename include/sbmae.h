/* sbmae.h -- C ABI of libsbmae_hip.so: the MI355X-native latent score-model sampling path.
 *
 * The reference (DanielMitiku/score_based_multimodal_autoencoder) is pure Python/PyTorch and has no
 * FFI layer; its boundary for this path is the module-level Python call surface.  Every entry point
 * below names the reference call it stands in for (file:line relative to the reference tree).  The
 * Python host side (the modules of score_based_multimodal_autoencoder_amd) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only (no torch types).  All device buffers are fp32,
 * contiguous, 16-byte aligned and owned by the caller.  `stream` is a hipStream_t passed as void*
 * (NULL = the legacy default stream).  All work is enqueued asynchronously on `stream`; nothing
 * synchronises except create/destroy, sbm_model_profile, and sbm_sampler_run / sbm_cond_sampler_run, which run on a PRIVATE stream
 * of the model handle (ordered after the caller's stream at entry and before it at exit by events) and synchronise that stream
 * when they replay a captured hipGraph (use_graph, n_run > 1: the graph is released before returning) or when elapsed_ms is
 * requested.  A chain-kernel barrier that timed out in an earlier call (not every workgroup resident) is reported by the next
 * sbm_model_forward / sbm_sampler_run / sbm_model_check on the same handle (and by sbm_model_destroy), through a host-mapped flag:
 * until then the calls in between return 0 with invalid results -- call sbm_model_check(m, 1, stream) before trusting an output
 * that no later call on the handle will vouch for.
 * Every function returns 0 on success and a non-zero code on failure; sbm_last_error() then
 * returns a message (thread-local).  No C++ exception crosses this boundary.  A model handle is not
 * thread-safe; use one handle per device and per batch size.
 */
#ifndef SBMAE_H
#define SBMAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbm_model sbm_model;

enum sbm_kind {
  SBM_UNET = 0,    /* unet_model.Unet (ConvNeXt score net)                 unet_model.py:189-323 */
  SBM_RESDEC = 1,  /* h_vae_model_copy.ResVAE/ResAE/ResVAEN/ResAEN.decoder  h_vae_model_copy.py:123-126,166-169,489-492,576-579 */
  SBM_ATTRDEC = 2, /* h_vae_model.CelebAAttrNewBN[AE].decoder               h_vae_model.py:768-769,888-890 */
  SBM_UNET_OA = 3, /* unet_openai.UNetModel (conditional CelebA score net)   unet_openai.py:361-575 */
  SBM_MLP = 4,     /* lat_sm2_model.LSMPoly64_sm (MLP score, DSM Langevin)   lat_sm2_model.py:348-371; cfg [n_mod, size_z].
                    * RESTRICTION: n_mod * size_z must be a multiple of 32 (the GEMM's K granule; sbm_model_create fails otherwise).  The
                    * reference accepts any width; its callers use 64-wide latents (n_mod * 64). */
  SBM_RESENC = 5   /* h_vae_model_copy.ResVAE/ResAE/ResVAEN/ResAEN.encoder: image (B,img_ch,s,s) -> (B, 2*size_z) = [mu | logvar]
                      h_vae_model_copy.py:41-71,379-409; cfg as SBM_RESDEC */,
  SBM_ATTRENC = 6  /* h_vae_model.CelebAAttrNewBN[AE].encoder: attributes (B, att_size) -> (B, 2*size_z) = [mu | logvar] (AE: (B, size_z))
                      h_vae_model.py:718-736,756-759, 845-863,884-887; cfg [size_z, att_size, ae] */,
  SBM_CLG = 7      /* lat_sm2_model.ClwithTime2 / ClwithTime3: the time-conditioned pairwise classifier / energy MLP used for guidance
                      (train_lat_celebhq_unet_cont2.py:487, train_poly_unet_cont.py:587; the reference imports these names but its
                      lat_sm2_model.py no longer defines them -- the architecture is this package's); cfg [variant 2|3, n_mod = 2, size_z, n_class] */
};

/* Integer configuration vectors (`cfg`):
 *   SBM_UNET    [dim, channels, init_dim|0, out_dim|0, with_time_emb, convnext_mult, H, W, n_mults, mults...]
 *               = the constructor arguments of unet_model.Unet (unet_model.py:190-201) + input size
 *   SBM_RESDEC  [variant(0 ResVAE/ResAE, 1 ResVAEN/ResAEN), size_in, size_z, img_ch, n_enc, n_dec,
 *                enc_channel_list (4 ints each)..., dec_channel_list (4 ints each)...]
 *               = the constructor arguments of ResVAE (h_vae_model_copy.py:92)
 *   SBM_ATTRDEC [size_z, att_size]   (h_vae_model.py:713)
 *   SBM_UNET_OA [in_channels, model_channels, out_channels, num_res_blocks, num_heads, z_dim, use_z, H, W,
 *                n_mult, channel_mult..., n_attn, attention_resolutions...]  (unet_openai.py:383-401);
 *               sbm_model_create's `in_pitch` selects the variant: 1 = forward(x, t, z=z), 0 = forward(x, t)
 */

const char* sbm_last_error(void);
int sbm_abi_version(void);

/* state_dict entries a model consumes, in the order `host_tensors` must be passed to
 * sbm_model_create.  Names are the reference's state_dict keys (torch `load_state_dict` contract,
 * train_poly_unet_cont.py:563,579).  sbm_spec returns the element count (or -1) and writes the
 * NUL-terminated key into `name`. */
int sbm_spec_count(int kind, const int* cfg);
long sbm_spec(int kind, const int* cfg, int index, char* name, int name_capacity);

/* Build a model for a fixed batch size: packs the weights (host fp32 pointers, reference layouts)
 * into the device arena and plans the fused kernel sequence.  Replaces module construction +
 * load_state_dict + .to(device) (train_poly_unet_cont.py:548-580).
 * `in_pitch`: decoders only -- row pitch in floats of the latent matrix handed to forward (the
 * reference decodes the strided slice z_out[:, ind].view(B, size_z), train_poly_unet_cont.py:303). */
int sbm_model_create(int kind, const int* cfg, int batch, int in_pitch, const float* const* host_tensors,
                     int n_tensors, sbm_model** out);
/* Releases the handle.  Waits for the whole device first (hipDeviceSynchronize): forwards and step kernels are enqueued
 * asynchronously on the caller's streams, and a handle dropped right behind such a call must not free the arenas under them.
 * Returns non-zero (message in sbm_last_error) when a HIP call of the tear-down failed, or when a chain-kernel barrier time-out
 * was still unreported on this handle: then every result produced on it since its last successful check is invalid.
 * The handle is gone in every case.  Replaces `del model` / interpreter exit of the reference. */
int sbm_model_destroy(sbm_model* m);

/* Chain-kernel health of a handle (models whose low-resolution levels run in the XCD-resident chain kernel; always 0 for the
 * others): non-zero when one of its barriers has timed out since the last successful check -- ALL results produced on the handle
 * since then are invalid (once a launch has aborted, every later launch returns at once until the state is reset, which this call
 * does).  `sync` != 0 first waits for `stream` and for the model's private stream, so the answer covers all work enqueued so far;
 * with sync == 0 the call only reads a host-mapped word.  sbm_model_forward / sbm_sampler_run make the same (unsynchronised) check
 * at entry, sbm_sampler_run also at exit whenever it synchronised anyway (use_graph or elapsed_ms). */
int sbm_model_check(sbm_model* m, int sync, void* stream);

/* info[0..9] = n_ops, workspace bytes, parameter bytes, dense FLOP per call, executed FLOP per call,
 *              n seg_gemm launches, batch, dense FLOP per call computed by seg_gemm launches,
 *              1 if the low-resolution levels run in the sample-owned pixel kernel (pixrun.hip), MB of its fragment-order weights */
int sbm_model_info(const sbm_model* m, double* info);
/* Why the low-resolution levels of this Unet do not run in the sample-owned pixel kernel (NUL-terminated; empty when they do): the
 * choice between that kernel and the legacy per-layer / chain kernels is made once in sbm_model_create and reported here. */
int sbm_model_note(const sbm_model* m, char* buf, int capacity);

/* One forward pass.
 *   SBM_UNET:    in = x (B,C,H,W), t = time (B,), out = score (B,C,H,W)        == Unet.forward(x, time), unet_model.py:275
 *   SBM_RESDEC:  in = z rows (B x size_z, pitch in_pitch), t = NULL, out (B,img_ch,S,S)  == .decoder(z)
 *   SBM_ATTRDEC: in = z rows, out (B, att_size)                                 == .decoder(z)          */
int sbm_model_forward(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, void* stream);

/* UNetModel.forward(x, timesteps, z=z_cond) (unet_openai.py:538): aux_dev = z (B, z_dim) for a model built with in_pitch=1. */
int sbm_model_forward_aux(sbm_model* m, const float* in_dev, const float* t_dev, const float* aux_dev, float* out_dev, void* stream);

/* Conditioning vector used by subsequent sbm_sampler_run calls on a with_z model: score_fn(x, t) == UNetModel(x, t, z=z_cond)
 * (the semantics train_lat_celebhq_unet_cont2_cond.py:225-226 intends; as shipped those calls raise TypeError -- SURVEY D8). */
int sbm_model_set_aux(sbm_model* m, const float* aux_dev);

/* Per-kernel timing of one forward with HIP events around every launch (eager, `reps` repetitions
 * after one warm-up).  ms_by_kind[k] / launches_by_kind[k] for k = plan op kind (0 = seg_gemm,
 * 1 = dwconv7, 2 = linattn, 3 = attn, 4 = gn_res, 5 = sinus, 6 = nchw_to_pbc, 7 = bilinear); sums per forward. */
int sbm_model_profile(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps,
                      double* ms_by_kind, int* launches_by_kind, int n_kinds);
/* As sbm_model_profile; sampler_mix != 0 times the launch mix of one step of sbm_sampler_run (the time path -- sinusoid, time MLP,
 * per-block condition GEMM -- is tabulated once per run there and not launched per step). */
int sbm_model_profile_mix(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps, int sampler_mix,
                          double* ms_by_kind, int* launches_by_kind, int n_kinds);
/* Dense-convention GEMM FLOP per forward executed by seg_gemm launches (out6[0]) and by chain_kernel launches (out6[1]);
 * out6[2], out6[3]: the parts of those that belong to the time path (tabulated once per sampler run, not launched per step);
 * out6[4]: by the row kernel (the one-position level); out6[5]: inside linattn_qkv_kernel (to_qkv). */
int sbm_model_gemm_flops(const sbm_model* m, double* out6);

/* SDE description == the reference's SDE objects (sde_helper2.py:329-473). kind: 0 VPSDE, 1 subVPSDE, 2 VESDE;
 * (b0,b1) = (beta_min,beta_max) or (sigma_min,sigma_max). */
typedef struct sbm_sde_cfg {
  int kind;
  double b0, b1;
  int N;
  double target_snr;
  int probability_flow;
} sbm_sde_cfg;

/* One Euler-Maruyama predictor update given the score  == em_predictor minus its score_fn call
 * (sde_helper2.py:45-52; PolyMNIST copy train_poly_unet_cont.py:96-102).
 * noise == NULL: draw N(0,1) in-kernel from Philox keyed by (seed, sample_offset + row, step, draw); `draw` numbers the draws of one
 * step in execution order (sbm_sampler_run: corrector iterations first for order 0, the predictor first for order 1). */
int sbm_em_step(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t,
                float* x_out, float* mean_out, int B, int CHW, uint64_t seed, uint64_t sample_offset, int step,
                int draw, void* stream);

/* One Langevin corrector iteration given the score  == the loop body of corrector (sde_helper2.py:96-101).
 * alphas: device (N,) = 1 - linspace(b0/N, b1/N, N) (ignored for VE); norms_scratch: device 2*B floats. */
int sbm_langevin_step(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t,
                      const float* alphas, float* x_out, float* mean_out, float* norms_scratch, int B, int CHW,
                      uint64_t seed, uint64_t sample_offset, int step, int draw, void* stream);

/* The same iteration in two halves, for a batch sharded over ranks (SURVEY 8(e) "exact mode"): the step size uses the MEAN of the
 * per-sample norms over the whole batch (sde_helper2.py:97-99), so between the halves the caller all-reduces two numbers.
 *   sbm_langevin_norms:  sums_out[0] = sum_b ||score_b||, sums_out[1] = sum_b ||noise_b|| over THIS shard's B rows, in fp64 (exact for
 *                        batches of this size, hence independent of the split); norms_scratch: device 2*B floats
 *   sbm_langevin_update: the update with norm_sums = those sums added over all shards and B_global = the global batch size. */
int sbm_langevin_norms(const float* score, const float* noise, float* norms_scratch, double* sums_out, int B, int CHW, uint64_t seed,
                       uint64_t sample_offset, int step, int draw, void* stream);
int sbm_langevin_update(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t, const float* alphas,
                        float* x_out, float* mean_out, const double* norm_sums, long B_global, int B, int CHW, uint64_t seed,
                        uint64_t sample_offset, int step, int draw, void* stream);

/* Classifier / energy guidance: for the pair of modalities (ch1, ch2) of the joint state x (B, C, per_mod),
 *   grad = d mean(cl_g(cat[x[:, ch1], x[:, ch2]].view(B, -1), t [, id1, id2])) / d (the pair)      (B, 2, per_mod)
 *   score[:, ch1] -= cl_s * grad[:, 0]  (upd_mask bit 0),   score[:, ch2] -= cl_s * grad[:, 1]  (bit 1)      in place
 * == the `if cl_g is not None and given` branches of RSDE.sde (sde_helper2.py:283-312) and corrector (:65-94); the PolyMNIST variant
 * (train_poly_unet_cont.py:73-89, 129-146) updates only the predicted modality (upd_mask = 2).  `mean` is over B * n_class outputs,
 * so the gradient carries the reference's 1 / (B * n_class).  grad_out (B, 2 * per_mod) optional; score_inout may be NULL then.
 * sbm_clg_forward: cl_g(x_pair, t [, id1, id2]) -> out (B, n_class). */
int sbm_clg_forward(sbm_model* m, const float* x_pair, const float* t, int id1, int id2, float* out, int B, void* stream);
int sbm_clg_guide(sbm_model* m, const float* x, const float* t, int ch1, int ch2, int id1, int id2, float cl_s, unsigned upd_mask,
                  float* score_inout, float* grad_out, int B, int C, int per_mod, void* stream);

/* Fixed-step Langevin update of the alternative samplers, on the joint state (B, n_mod, per_mod):
 *   x_out[:, m] = x[:, m] + a * (score[:, m] / score_div) + b * noise[:, m]  for modalities m with bit m of mod_mask set, x[:, m] otherwise.
 *   annealed Langevin (fid_upd10.gen_ae :97, eval_lat_celeba_hq_all.evaluate :258-275): a = alpha, score_div = sigma_i, b = c*sqrt(2*alpha)
 *   MLP-score Langevin (fid_upd10.calc_poly_uncond :331): a = lr1, score_div = 1, b = lr2.   noise == NULL: in-kernel Philox. */
int sbm_langevin_fixed_step(const float* x, const float* score, const float* noise, float* x_out, float a, float b, float score_div,
                            unsigned mod_mask, int B, int n_mod, int per_mod, uint64_t seed, uint64_t sample_offset, int step, int draw, void* stream);

/* Device-resident fixed-step Langevin loops as ONE call: the annealed chain over a time-conditioned score net (SBM_UNET / SBM_UNET_OA;
 * fid_upd10.gen_ae :84-99, eval_lat_celeba_hq_all.evaluate :258-275) and the MLP-score chain (SBM_MLP; fid_upd10.calc_poly_uncond
 * :311-339).  For k in [0, n_iter):
 *   score = net(z [, times[k]]);   z[:, m] += a[k][m] * (score[:, m] / score_div[k]) + b[k][m] * noise_k[:, m]   for bit m of mod_mask set.
 * times (n_iter,; NULL for a net without time input) and coef (n_iter, 2*n_mod + 1) = rows [a_0 .. a_{n_mod-1} | b_0 .. b_{n_mod-1} |
 * score_div] are DEVICE arrays; the iteration index lives in device memory, a net's time path is tabulated once for all iterations, and
 * one captured hipGraph {score net, update, advance} is replayed n_iter times (use_graph != 0).  tail: NULL, or a HOST row of the same
 * layout applied once more with the LAST iteration's score (calc_poly_uncond's closing step, :336-339; Philox draw 1 of that iteration).
 * noise (n_iter [+ 1 with tail], B, n_mod*per_mod) or NULL: in-kernel Philox keyed (seed, sample_offset + row, step = k, draw = 0) --
 * the stream sbm_langevin_fixed_step(step = k) uses, so the loop reproduces the per-iteration calls bit for bit.  z_inout
 * (B, n_mod, per_mod) is updated in place.  elapsed_ms (optional): loop time by HIP events. */
int sbm_langevin_run(sbm_model* m, const float* times, const float* coef, unsigned mod_mask, int n_mod, int n_iter, const float* noise,
                     float* z_inout, uint64_t seed, uint64_t sample_offset, int use_graph, const float* tail, float* elapsed_ms, void* stream);

/* Conditional loop: re-noise the OBSERVED modalities at time t, x_out[:, m] = mean(t)*z_obs[:, m] + std(t)*z_obs[:, m]
 * for bit m of obs_mask set (the reference multiplies std by z itself, train_poly_unet_cont.py:363-364), x[:, m] otherwise;
 * mean/std = SDE.marginal_prob (sde_helper2.py:358-362, 409-413, 452-455). */
int sbm_renoise_observed(const sbm_sde_cfg* sde, const float* x, const float* z_obs, const float* t, float* x_out, unsigned obs_mask,
                         int B, int n_mod, int per_mod, void* stream);

/* One ancestral step of the DiffuseVAE refiner's DDPM loop (eval_lat_celeba_hq_all.py:107-118), on (B, chw) rows:
 *   x_out = sqrt_recip_alpha_t * (x - beta_t * eps / sqrt(1 - alphabar_t)) + sigma_t * z,  sigma_t = sqrt(posterior_variance_t), 0 at t = 0.
 * eps = sm_model(cat[x, vae_out], t).  noise == NULL: z from the in-kernel Philox stream keyed by (seed, global sample, step). */
int sbm_ddpm_step(const float* x, const float* eps, const float* noise, float* x_out, float sqrt_recip_alpha, float beta, float sqrt_one_minus_abar,
                  float sigma, int B, long chw, uint64_t seed, uint64_t sample_offset, int step, void* stream);

/* out[b, :] = scale * N(0,1) from the same Philox stream (prior draw: SDE.prior_sampling, sde_helper2.py:364,415,457) */
int sbm_fill_normal(float* out, long n_per_sample, int B, uint64_t seed, uint64_t sample_offset, int step, int draw,
                    float scale, void* stream);

typedef struct sbm_sampler_cfg {
  int pc;             /* run the Langevin corrector                         (uncond_sampler `pc`) */
  int n_steps;        /* corrector iterations per step                     (`n_steps`) */
  int order;          /* 0: corrector -> predictor (sde_helper2.py:121-126); 1: predictor -> corrector (train_poly_unet_cont.py:180-186) */
  int use_graph;      /* capture one step into a hipGraph and replay it */
  uint64_t seed;      /* Philox key when noise == NULL */
  uint64_t sample_offset; /* global index of row 0 (batch sharding) */
} sbm_sampler_cfg;

/* The whole sampling loop for steps [first_step, first_step + n_run): per step the score net forward
 * and the predictor / corrector updates  == uncond_sampler's for-loop (sde_helper2.py:120-126).
 *   timesteps: device (N,) = linspace(T, eps, N);  alphas: device (N,) or NULL when pc == 0
 *   noise: NULL, or device [(N * draws_per_step)][B][CHW] injected draws in execution order
 *          (draws_per_step = 1 + pc*n_steps)
 *   x_inout: state, updated in place;  x_mean_out: the last executed sub-step's mean (what the
 *   reference returns, sde_helper2.py:128)
 *   elapsed_ms: NULL, or receives the HIP-event time of the loop on the stream it ran on (synchronises). */
int sbm_sampler_run(sbm_model* score, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const float* timesteps,
                    const float* alphas, const float* noise, float* x_inout, float* x_mean_out, int first_step,
                    int n_run, float* elapsed_ms, void* stream);

/* The conditional loop of the evaluation callers, fused: per step re-noise (noise_obs) or restore the observed
 * modalities' latents, then predictor [+ corrector], all captured as ONE step like sbm_sampler_run; at the end the observed channels
 * of x_mean_out are the clean z_obs.  == the loops of train_lat_celebhq_unet_cont2.calc_perf (:286-316) and
 * train_poly_unet_cont.evaluate (:356-383), which run predictor -> corrector (cfg->order = 1).
 *   z_obs: device (B, n_mod, per_mod) -- only the channels in obs_mask are read;  x_inout: the joint state, observed channels ignored. */
typedef struct sbm_cond_cfg {
  unsigned obs_mask;    /* bit m set: modality m is observed (given) */
  int noise_obs;        /* 1: observed latents enter each step as mean(t) z + std(t) z (the reference multiplies std by z itself) */
  int n_mod;            /* modalities = channels of the joint state */
} sbm_cond_cfg;
int sbm_cond_sampler_run(sbm_model* score, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const sbm_cond_cfg* cond, const float* z_obs,
                         const float* timesteps, const float* alphas, const float* noise, float* x_inout, float* x_mean_out,
                         int first_step, int n_run, float* elapsed_ms, void* stream);

/* The same loop with classifier / energy guidance INSIDE the captured step: after every score evaluation -- the predictor's
 * and each corrector iteration's -- every job subtracts cl_s * d cl_g(cat[x[:, ch1], x[:, ch2]], t).mean() / dx from the score of its two
 * modalities (upd_mask bit 0 / 1: update ch1 / ch2), at the x and t of that evaluation: RSDE.sde sde_helper2.py:283-312 and corrector
 * :65-94, pairs fixed for the run (the CelebA scripts' '01' / '02' / '12' form; the PolyMNIST variant draws a random pair per call on the
 * host and stays a host-driven loop).  jobs[j].clg: an SBM_CLG handle (ClwithTime2 / ClwithTime3; id1 / id2 are used by variant 3). */
typedef struct sbm_guide_job { sbm_model* clg; int ch1, ch2, id1, id2; unsigned upd_mask; } sbm_guide_job;
int sbm_guided_cond_sampler_run(sbm_model* score, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const sbm_cond_cfg* cond, const float* z_obs,
                                const sbm_guide_job* jobs, int n_jobs, float cl_s, const float* timesteps, const float* alphas, const float* noise,
                                float* x_inout, float* x_mean_out, int first_step, int n_run, float* elapsed_ms, void* stream);

/* Debug entry points used only by tests/tools: run the first n_ops ops of the plan (synchronises), read back
 * workspace floats, name an op ("<kind>:<reference layer>").  Not part of the drop-in surface. */
int sbm_debug_forward_partial(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int n_ops, void* stream);
int sbm_debug_forward_range(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int first, int last, void* stream);
long sbm_debug_workspace_floats(const sbm_model* m);
int sbm_debug_read_workspace(sbm_model* m, long off, long n, float* host);
int sbm_debug_op_name(const sbm_model* m, int i, char* name, int cap);
int sbm_debug_chain_sync(sbm_model* m, unsigned* out, int n_words);
int sbm_debug_chain_ctl(sbm_model* m, unsigned spin_limit, int skip_arrive);   /* barrier poll limit (0 = default); 1 = force a barrier time-out */
int sbm_debug_chain_stamps(sbm_model* m, int first, int count, unsigned long long* out_host);
int sbm_debug_row_stamps(sbm_model* m, int first, int count, unsigned long long* out_host, int* kinds, int cap, int* n_out);
/* the same for a run of pixel-kernel ops (csrc/pixrun.hip: the 4x4 / 2x2 levels as sample-owned workgroups); kinds[i] = op kind * 1000 + taps */
int sbm_debug_pix_stamps(sbm_model* m, int first, int count, unsigned long long* out_host, int* kinds, int cap, int* n_out);
int sbm_debug_gemm_stamps(sbm_model* m, const float* in_dev, float* out_dev, int i, unsigned long long* stamps_host);
int sbm_debug_profile_ops(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps,
                          double* ms_per_op, double* flops_per_op, int n_ops);

/* ---- evaluation statistics right behind the path: pytorch_fid/fid_score.py on device features, fp64 ----
 * The Inception network is not part of this library (its weights are not available offline): the entry points start from the
 * (n, dims) feature rows the reference collects in `pred_arr` (fid_score.py:127-153, 163-184). */

/* sum[d] += sum_i feats[i][d], gram[d][e] += sum_i feats[i][d] * feats[i][e], both fp64 on the device, feats fp32 (row pitch ld).
 * One call per batch of features; a rank accumulates its shard, the two buffers are then summed over ranks (one all-reduce of
 * dims + dims^2 doubles) -- what replaces growing `pred_arr` and np.mean / np.cov over it (fid_score.py:248-250, 273-279). */
int sbm_fid_accumulate(const float* feats_dev, long n, int dims, long ld, double* sum_dev, double* gram_dev, void* stream);

/* mu = sum / n, sigma = (gram - n mu mu^T) / (n - 1): np.mean(act, axis=0), np.cov(act, rowvar=False) (fid_score.py:249-250) */
int sbm_fid_finalize_stats(const double* sum_dev, const double* gram_dev, long n, int dims, double* mu_dev, double* sigma_dev, void* stream);

/* calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps) (fid_score.py:187-240) on device fp64 operands.
 * Tr sqrtm(sigma1 sigma2) is taken as Tr sqrtm(S sigma2 S), S = sqrtm(sigma1), both roots by coupled Newton-Schulz iterations of
 * fp64 MFMA GEMMs stopped when the trace is stationary (at most max_iters each, 0 = default 60); if a root is not finite the
 * iteration is repeated with eps added to both diagonals, as the reference does (:224-229).
 * work_dev: sbm_fid_work_doubles(dims) doubles of device scratch.  out_host[5] (HOST memory) = {d^2, Tr sqrtm, iterations of the
 * first root, of the second root, 1 if eps was used}.  Synchronises the stream. */
long sbm_fid_work_doubles(int dims);
int sbm_fid_frechet(const double* mu1_dev, const double* sigma1_dev, const double* mu2_dev, const double* sigma2_dev, int dims, double eps,
                    int max_iters, double* work_dev, double* out_host, void* stream);

/* test hook: C[M][N] = alpha * At^T B + beta * C + diag * I with At [K][M], B [K][N] row-major fp64 (the GEMM the entries above use);
 * K < 0: the first operand is A [M][-K] row-major instead (C = alpha * A B + ...) */
int sbm_debug_dgemm_tn(const double* At_dev, const double* B_dev, double* C_dev, int M, int N, int K, double alpha, double beta, double diag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SBMAE_H */
