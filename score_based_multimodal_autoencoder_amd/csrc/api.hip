// api.hip -- the C ABI of include/sbmae.h: model handles (plan + device arenas), forward, per-kernel
// profiling, SDE step entry points and the hipGraph-replayed sampling loop.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/sbmae.h"
#include "device.h"
#include "plan.h"

using namespace sbm;

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
static int fail_hip(const char* what, hipError_t e) { g_err = std::string(what) + ": " + hipGetErrorString(e); return 2; }
namespace sbm {
int api_fail(const std::string& m) { return fail(m); }                       // for the other translation units of the library (fid.hip)
int api_fail_hip(const char* what, hipError_t e) { return fail_hip(what, e); }
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail_hip(#x, e_); } while (0)

struct sbm_model {
  int kind = 0;
  Plan plan;
  float* d_params = nullptr;
  float* d_ws = nullptr;
  GemmCold* d_cold = nullptr;    // one per op (only the GEMM entries are meaningful)
  ChainOp* d_chain_ops = nullptr; // one per op (filled for ops marked op.chain)
  ChainSync* d_chain_sync = nullptr;
  int* d_chain_sched = nullptr;   // unit schedules of the chain GEMMs
  unsigned* h_abort = nullptr;    // host-mapped word the chain kernel raises when a barrier times out (polled without synchronising)
  const float* aux_persistent = nullptr;   // z_cond used by sbm_sampler_run (sbm_model_set_aux)
  DevPlan dp;
  double pix_frag_mb = 0.0;       // fragment-order weight copies of the pixel kernel (MB), 0 when it is not in use
  std::string pix_reason;         // why a Unet does not use the pixel kernel (empty when it does)
  // sampler scratch (score nets only)
  float* d_score = nullptr;
  float* d_norms = nullptr;
  int* d_counter = nullptr;       // StepCtl block (step counter first)
  float* d_time_table = nullptr;  // [table_rows][cond_ld] time-path outputs for every step of the current sampler run
  size_t time_table_floats = 0;
  long long clg_off[13] = {0};    // SBM_CLG: offsets of the packed guidance-MLP tensors inside d_params (guidance.hip clg_pack)
  int clg_cfg[4] = {0, 0, 0, 0};  // variant, n_mod, size_z, n_class
  hipStream_t own = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
};

static UnetCfg unet_cfg_from_ints(const int* c) {
  UnetCfg u; u.dim = c[0]; u.channels = c[1]; u.init_dim = c[2]; u.out_dim = c[3]; u.with_time_emb = c[4]; u.convnext_mult = c[5];
  u.H = c[6]; u.W = c[7]; u.n_mults = c[8] < 8 ? c[8] : 8;
  for (int i = 0; i < u.n_mults; ++i) u.dim_mults[i] = c[9 + i];
  return u;
}
static ResDecCfg resdec_cfg_from_ints(const int* c) {
  ResDecCfg r; r.variant = c[0]; r.size_in = c[1]; r.size_z = c[2]; r.img_ch = c[3]; r.n_enc = c[4] < 8 ? c[4] : 8; r.n_dec = c[5] < 8 ? c[5] : 8;
  const int* e = c + 6;
  for (int i = 0; i < r.n_enc; ++i) for (int j = 0; j < 4; ++j) r.enc[i][j] = *e++;
  for (int i = 0; i < r.n_dec; ++i) for (int j = 0; j < 4; ++j) r.dec[i][j] = *e++;
  return r;
}
static UnetOACfg unet_oa_cfg_from_ints(const int* c) {
  UnetOACfg u; u.in_channels = c[0]; u.model_channels = c[1]; u.out_channels = c[2]; u.num_res_blocks = c[3]; u.num_heads = c[4];
  u.z_dim = c[5]; u.use_z = c[6]; u.H = c[7]; u.W = c[8]; u.n_mult = c[9] < 8 ? c[9] : 8;
  for (int i = 0; i < u.n_mult; ++i) u.channel_mult[i] = c[10 + i];
  u.n_attn = c[10 + u.n_mult] < 8 ? c[10 + u.n_mult] : 8;
  for (int i = 0; i < u.n_attn; ++i) u.attn_res[i] = c[11 + u.n_mult + i];
  return u;
}
static std::vector<TensorSpec> specs_of(int kind, const int* cfg) {
  if (kind == SBM_UNET_OA) return unet_oa_tensor_specs(unet_oa_cfg_from_ints(cfg));
  if (kind == SBM_MLP) { MlpCfg c; c.n_mod = cfg[0]; c.size_z = cfg[1]; return mlp_tensor_specs(c); }
  if (kind == SBM_RESENC) return resenc_tensor_specs(resdec_cfg_from_ints(cfg));
  if (kind == SBM_ATTRENC) { AttrEncCfg c; c.size_z = cfg[0]; c.att_size = cfg[1]; c.ae = cfg[2]; return attrenc_tensor_specs(c); }
  if (kind == SBM_UNET) return unet_tensor_specs(unet_cfg_from_ints(cfg));
  if (kind == SBM_CLG) return clg_tensor_specs(cfg[0], cfg[1], cfg[2], cfg[3]);
  if (kind == SBM_RESDEC) return resdec_tensor_specs(resdec_cfg_from_ints(cfg));
  if (kind == SBM_ATTRDEC) { AttrDecCfg a; a.size_z = cfg[0]; a.att_size = cfg[1]; return attrdec_tensor_specs(a); }
  return {};
}

static SdeParams make_sde(const sbm_sde_cfg& c) {
  SdeParams s{};
  s.kind = c.kind; s.b0 = (float)c.b0; s.b1 = (float)c.b1; s.b1_minus_b0 = (float)(c.b1 - c.b0);
  s.ve_ratio = (float)(c.b1 / c.b0);
  s.ve_gconst = c.kind == 2 ? (float)std::sqrt(2.0 * (std::log(c.b1) - std::log(c.b0))) : 0.f;
  s.dt = (float)(-1.0 / c.N); s.sqrt_mdt = (float)std::sqrt(1.0 / c.N);
  s.snr = (float)c.target_snr; s.N = c.N; s.probability_flow = c.probability_flow;
  return s;
}

// A chain-kernel barrier that timed out in an EARLIER call (not every workgroup resident -- e.g. the GPU shared with another
// process) raised the host-mapped flag.  Every entry point that launches the kernel looks at it first, without synchronising:
// the synchronisation state is reset (in stream order) and the call fails, naming the earlier results as invalid.
static int chain_prior_abort(sbm_model* m, hipStream_t st, const char* who) {
  if (!m->h_abort || !__atomic_load_n(m->h_abort, __ATOMIC_RELAXED)) return 0;
  (void)hipMemsetAsync(m->d_chain_sync, 0, sizeof(ChainSync), st);
  __atomic_store_n(m->h_abort, 0u, __ATOMIC_RELAXED);
  return fail(std::string(who) + ": a chain-kernel barrier timed out in an earlier call on this model (not every workgroup was resident?); "
              "all results produced on the handle since the last successful check are invalid.  The synchronisation state has been reset; "
              "repeat the calls.");
}

extern "C" {

const char* sbm_last_error(void) { return g_err.c_str(); }
int sbm_abi_version(void) { return 3; }

int sbm_spec_count(int kind, const int* cfg) { return (int)specs_of(kind, cfg).size(); }

long sbm_spec(int kind, const int* cfg, int index, char* name, int cap) {
  std::vector<TensorSpec> s = specs_of(kind, cfg);
  if (index < 0 || index >= (int)s.size() || cap <= 0) return -1;
  std::strncpy(name, s[index].name.c_str(), cap - 1); name[cap - 1] = 0;
  return (long)s[index].numel();
}

// Release a handle.  Work of this model may still be in flight: sbm_model_forward and the single-kernel entry points enqueue on the
// CALLER's stream and return at once, so a handle dropped right behind such a call (a temporary module going out of scope in
// Python) is destroyed under running kernels that read the weights / workspace arenas freed here.  Destroy therefore waits for the
// whole device first.  (Round 2 recorded what happens without that wait -- gpurun_out/t7_r2.log, DESIGN.md section 1 "Streams".)
// Every HIP status is kept: the first failures are joined into sbm_last_error and the call returns non-zero; a chain-kernel barrier
// time-out that no later call on the handle has reported yet is reported here.
int sbm_model_destroy(sbm_model* m) {
  if (!m) return 0;
  std::string errs;
  auto note = [&](const char* what, hipError_t e) {
    if (e == hipSuccess) return;
    if (!errs.empty()) errs += "; ";
    errs += what; errs += ": "; errs += hipGetErrorString(e);
  };
  note("hipDeviceSynchronize", hipDeviceSynchronize());
  const bool aborted = m->h_abort && __atomic_load_n(m->h_abort, __ATOMIC_RELAXED) != 0u;
  rowrun_release(&m->plan);
  pixrun_release(m->dp);
  if (m->own) note("hipStreamSynchronize(model stream)", hipStreamSynchronize(m->own));
  if (m->d_params) note("hipFree(params)", hipFree(m->d_params));
  if (m->d_ws) note("hipFree(workspace)", hipFree(m->d_ws));
  if (m->d_cold) note("hipFree(cold args)", hipFree(m->d_cold));
  if (m->d_chain_ops) note("hipFree(chain ops)", hipFree(m->d_chain_ops));
  if (m->d_chain_sync) note("hipFree(chain sync)", hipFree(m->d_chain_sync));
  if (m->d_chain_sched) note("hipFree(chain schedule)", hipFree(m->d_chain_sched));
  if (m->h_abort) note("hipHostFree(abort word)", hipHostFree(m->h_abort));
  if (m->d_score) note("hipFree(score)", hipFree(m->d_score));
  if (m->d_norms) note("hipFree(norms)", hipFree(m->d_norms));
  if (m->d_counter) note("hipFree(step control)", hipFree(m->d_counter));
  if (m->d_time_table) note("hipFree(time table)", hipFree(m->d_time_table));
  if (m->ev_in) note("hipEventDestroy", hipEventDestroy(m->ev_in));
  if (m->ev_out) note("hipEventDestroy", hipEventDestroy(m->ev_out));
  if (m->ev_t0) note("hipEventDestroy", hipEventDestroy(m->ev_t0));
  if (m->ev_t1) note("hipEventDestroy", hipEventDestroy(m->ev_t1));
  if (m->own) note("hipStreamDestroy", hipStreamDestroy(m->own));
  delete m;
  if (aborted) {
    std::string msg = "sbm_model_destroy: a chain-kernel barrier timed out on this model and no call reported it: every result produced on "
                      "the handle since its last successful sbm_model_check / forward / sampler call is invalid";
    if (!errs.empty()) msg += " (also: " + errs + ")";
    return fail(msg);
  }
  if (!errs.empty()) return fail("sbm_model_destroy: " + errs);
  return 0;
}

// Has a chain-kernel barrier timed out since the last check?  Reads the host-mapped word; `sync` != 0 waits for the model's private
// stream and for `stream` first, so that the answer covers everything enqueued so far.  On a time-out the synchronisation state is
// reset and the call fails: ALL results produced on this handle since the last successful check are invalid (an aborted launch
// makes every later launch return at once until the state is reset).
int sbm_model_check(sbm_model* m, int sync, void* stream) {
  if (!m) return fail("sbm_model_check: bad arguments");
  if (sync) {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (m->own) HIPCHK(hipStreamSynchronize(m->own));
  }
  if (!m->h_abort || !__atomic_load_n(m->h_abort, __ATOMIC_RELAXED)) return 0;
  (void)hipMemsetAsync(m->d_chain_sync, 0, sizeof(ChainSync), (hipStream_t)stream);
  __atomic_store_n(m->h_abort, 0u, __ATOMIC_RELAXED);
  return fail("sbm_model_check: a chain-kernel barrier timed out on this model (not every workgroup was resident?): all results produced on "
              "the handle since the last successful check are invalid.  The synchronisation state has been reset; repeat the calls.");
}

static int model_create_impl(int kind, const int* cfg, int batch, int in_pitch, const float* const* host_tensors, int n_tensors, sbm_model** out, bool pix_mode, std::string* pix_reason);

// The low-resolution levels of a Unet (4x4 and smaller) run as sample-owned workgroups (pixrun.hip) whenever that kernel can take them
// (op shapes, LDS, weight volume: decided HERE, once, never at launch time); otherwise -- or with SBM_PIX=0 -- the model is planned
// again the legacy way (XCD-resident chain kernel for large batches + row kernel, per-layer launches for small ones).  The reason
// is kept on the handle (sbm_model_info) and printed with SBM_PIX_VERBOSE.
int sbm_model_create(int kind, const int* cfg, int batch, int in_pitch, const float* const* host_tensors, int n_tensors, sbm_model** out) {
  const char* pe = std::getenv("SBM_PIX");
  const bool try_pix = kind == SBM_UNET && !(pe && std::atoi(pe) == 0);
  std::string reason = try_pix ? "" : (kind == SBM_UNET ? "disabled by SBM_PIX=0" : "");
  if (try_pix) {
    const int rc = model_create_impl(kind, cfg, batch, in_pitch, host_tensors, n_tensors, out, true, &reason);
    if (rc == 0 || reason.empty()) return rc;          // built, or failed for a reason that has nothing to do with the pixel kernel
    if (std::getenv("SBM_PIX_VERBOSE")) fprintf(stderr, "sbmae: low-resolution levels stay with the legacy kernels: %s\n", reason.c_str());
  }
  const int rc = model_create_impl(kind, cfg, batch, in_pitch, host_tensors, n_tensors, out, false, nullptr);
  if (rc == 0 && *out) (*out)->pix_reason = reason;
  return rc;
}

static int model_create_impl(int kind, const int* cfg, int batch, int in_pitch, const float* const* host_tensors, int n_tensors, sbm_model** out, bool pix_mode, std::string* pix_reason) {
  if (!cfg || !host_tensors || !out || batch <= 0) return fail("sbm_model_create: bad arguments");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("sbm_model_create: no HIP device (this library has no CPU path)");
  if ((int)specs_of(kind, cfg).size() != n_tensors) return fail("sbm_model_create: tensor count does not match sbm_spec_count");
  sbm_model* m = new (std::nothrow) sbm_model();
  if (!m) return fail("out of host memory");
  m->kind = kind;
  prepare_layer_kernels();
  if (kind == SBM_CLG) {                    // guidance MLP: no plan, just the packed matrices
    std::vector<float> packed; std::string err;
    if (!clg_pack(cfg[0], cfg[1], cfg[2], cfg[3], host_tensors, &packed, m->clg_off, &err)) { delete m; return fail("sbm_model_create: " + err); }
    for (int i = 0; i < 4; ++i) m->clg_cfg[i] = cfg[i];
    hipError_t ce;
    if ((ce = hipMalloc(&m->d_params, packed.size() * sizeof(float))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(params)", ce); }
    if ((ce = hipMemcpy(m->d_params, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemcpy(params)", ce); }
    m->plan.B = batch;
    *out = m;
    return 0;
  }
  bool ok = false;
  plan_set_pix_mode(pix_mode ? 1 : 0);
  if (kind == SBM_UNET) ok = build_unet_plan(unet_cfg_from_ints(cfg), batch, host_tensors, &m->plan);
  else if (kind == SBM_RESDEC) ok = build_resdec_plan(resdec_cfg_from_ints(cfg), batch, in_pitch, host_tensors, &m->plan);
  else if (kind == SBM_ATTRDEC) { AttrDecCfg a; a.size_z = cfg[0]; a.att_size = cfg[1]; ok = build_attrdec_plan(a, batch, in_pitch, host_tensors, &m->plan); }
  else if (kind == SBM_UNET_OA) ok = build_unet_oa_plan(unet_oa_cfg_from_ints(cfg), batch, in_pitch, host_tensors, &m->plan);
  else if (kind == SBM_MLP) { MlpCfg c; c.n_mod = cfg[0]; c.size_z = cfg[1]; ok = build_mlp_plan(c, batch, host_tensors, &m->plan); }
  else if (kind == SBM_RESENC) ok = build_resenc_plan(resdec_cfg_from_ints(cfg), batch, host_tensors, &m->plan);
  else if (kind == SBM_ATTRENC) { AttrEncCfg c; c.size_z = cfg[0]; c.att_size = cfg[1]; c.ae = cfg[2]; ok = build_attrenc_plan(c, batch, host_tensors, &m->plan); }
  else m->plan.error = "unknown model kind";
  plan_set_pix_mode(0);
  if (!ok) { std::string e = m->plan.error; delete m; return fail("sbm_model_create: " + e); }
  for (int i = 0; i + 1 < (int)m->plan.ops.size(); ++i) if (qkv_linattn_fusable(m->plan, i)) m->plan.ops[i].fuse_next = 1;
  for (int i = 0; i + 3 < (int)m->plan.ops.size(); ++i)
    if (attn_block_fusable(m->plan, i)) {
      m->plan.ops[i].fuse_next = 2; m->plan.ops[i + 2].fuse_next = m->plan.ops[i + 3].fuse_next = -1;
      pack_attn_block_fragments(&m->plan, i);          // appends the block's weights in the kernel's per-lane order to the parameter arena (before its upload)
    }
  const size_t pbytes = m->plan.params.size() * sizeof(float), wbytes = (size_t)m->plan.ws_floats * sizeof(float);
  hipError_t e;
  if ((e = hipMalloc(&m->d_params, pbytes ? pbytes : 16)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(params)", e); }
  if ((e = hipMalloc(&m->d_ws, wbytes ? wbytes : 16)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(workspace)", e); }
  if ((e = hipMemcpy(m->d_params, m->plan.params.data(), pbytes, hipMemcpyHostToDevice)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemcpy(params)", e); }
  if ((e = hipMemset(m->d_ws, 0, wbytes)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemset(workspace)", e); }
  std::vector<float> host_params;
  host_params.swap(m->plan.params);                   // the plan no longer needs its host copy; the pixel kernel's packer (below) still does
  if (kind == SBM_UNET || kind == SBM_UNET_OA || kind == SBM_MLP) {
    if ((e = hipMalloc(&m->d_counter, 64)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(counter)", e); }
    (void)hipMemset(m->d_counter, 0, 64);
    m->dp.ctl = reinterpret_cast<const StepCtl*>(m->d_counter);
  }
  {
    // resolve every GEMM op into kernel arguments once; the cold halves live in device memory
    const size_t nops = m->plan.ops.size();
    std::vector<GemmCold> cold(nops);
    if ((e = hipMalloc(&m->d_cold, sizeof(GemmCold) * (nops ? nops : 1))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(cold args)", e); }
    m->dp.plan = &m->plan; m->dp.prm = m->d_params; m->dp.ws = m->d_ws; m->dp.hot.resize(nops);
    for (size_t i = 0; i < nops; ++i) {
      if (m->plan.ops[i].kind != OP_GEMM) { std::memset(&cold[i], 0, sizeof(GemmCold)); continue; }
      const char* ge = make_gemm_args(m->plan, m->plan.ops[i].g, m->d_params, m->d_ws, m->d_cold + i, &m->dp.hot[i], &cold[i]);
      if (ge) { std::string msg = std::string(ge) + " (" + m->plan.ops[i].name + ")"; sbm_model_destroy(m); return fail("sbm_model_create: " + msg); }
    }
    if ((e = hipMemcpy(m->d_cold, cold.data(), sizeof(GemmCold) * nops, hipMemcpyHostToDevice)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemcpy(cold args)", e); }
    // programs of the row kernel (rowrun.hip).  A level too wide for the LDS (activations of four samples) stays in the chain kernel:
    // same descriptors, same numerics class, decided here once -- never at launch time.
    if (pix_mode) {
      // sample-owned kernel: fragment-order copies of the low-resolution weights + the programs of its runs.  A "pixel kernel: ..." failure
      // sends the caller back to the legacy plan.
      if (const char* pe = pixrun_prepare(m->dp, host_params.data(), &m->pix_frag_mb)) { if (pix_reason) *pix_reason = pe; std::string msg = pe; sbm_model_destroy(m); return fail("sbm_model_create: " + msg); }
      m->dp.use_pix = true;
    }
    std::vector<float>().swap(host_params);
    if (const char* re = pix_mode ? nullptr : rowrun_prepare(m->dp)) {
      if (std::string(re).find("do not fit") == std::string::npos) { std::string msg = re; sbm_model_destroy(m); return fail("sbm_model_create: " + msg); }
      rowrun_release(&m->plan);
      for (Op& op : m->plan.ops) if (op.row) { op.row = 0; op.chain = 1; }
    }
    // op table of the XCD-resident chain kernel (chain.hip)
    bool any_chain = false;
    std::vector<ChainOp> cops(nops);
    std::vector<int> sched_all; std::vector<size_t> sched_at(nops, (size_t)-1);
    for (size_t i = 0; i < nops; ++i) {
      const Op& op = m->plan.ops[i];
      std::memset(&cops[i], 0, sizeof(ChainOp));
      if (!op.chain || pix_mode) continue;              // pixel mode: the marked ops never run in the chain kernel
      any_chain = true;
      cops[i].kind = op.kind;
      if (op.kind == OP_GEMM) {
        const GemmOp& go = op.g;
        cops[i].gm.g = m->dp.hot[i]; cops[i].gm.cold = cold[i]; cops[i].P = go.Hout * go.Wout;
        unsigned mask = 0;
        if (go.KH * go.KW <= 32 && go.w_pix_stride == 0)
          for (int pp = 0; pp < cops[i].P; ++pp)
            for (int ky = 0; ky < go.KH; ++ky) for (int kx = 0; kx < go.KW; ++kx)
              if (go.a_bcast || tap_source_pixel(pp / go.Wout, pp % go.Wout, ky, kx, go.stride, go.pad, go.transposed, go.up, go.a.H, go.a.W) >= 0) mask |= 1u << (ky * go.KW + kx);
        cops[i].gm.tapmask = mask; cops[i].gm.tap_floats = go.Npad * go.Kp; cops[i].gm.w2_floats = go.has2 ? go.Npad * go.Kp2 : 0;
        // unit schedule: longest-processing-time over 32 workgroups, cost = K chunks of the unit's pixel
        const int P = cops[i].P, nblk = ((go.out_nchw ? go.N : go.out.C) + 31) / 32, units = P * nblk;
        std::vector<int> cost(P, 0);
        for (int pp = 0; pp < P; ++pp) {
          int taps = 0;
          for (int ky = 0; ky < go.KH; ++ky) for (int kx = 0; kx < go.KW; ++kx)
            if (go.a_bcast || tap_source_pixel(pp / go.Wout, pp % go.Wout, ky, kx, go.stride, go.pad, go.transposed, go.up, go.a.H, go.a.W) >= 0) ++taps;
          cost[pp] = taps * (go.Kp / 32) + (go.has2 ? go.Kp2 / 32 : 0);
        }
        { int mx = 0; for (int pp = 0; pp < P; ++pp) mx = std::max(mx, cost[pp]); cops[i].gm.wpu = (mx <= 8 && units >= 128 && !std::getenv("SBM_NO_WPU")) ? 1 : 0; }   // pays only with >= 4 units per workgroup
        std::vector<int> order(units);
        for (int u = 0; u < units; ++u) order[u] = u;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a / nblk] > cost[b / nblk]; });
        std::vector<std::vector<int>> bins(32); std::vector<long> load(32, 0);
        for (int u : order) { int best = 0; for (int k = 1; k < 32; ++k) if (load[k] < load[best]) best = k; bins[best].push_back(u); load[best] += cost[u / nblk]; }
        size_t rounds = 0; for (auto& b : bins) rounds = std::max(rounds, b.size());
        sched_at[i] = sched_all.size(); cops[i].gm.rounds = (int)rounds;
        for (size_t r = 0; r < rounds; ++r) for (int k = 0; k < 32; ++k) sched_all.push_back(r < bins[k].size() ? bins[k][r] : -1);
      }
      else if (op.kind == OP_DWCONV7) cops[i].d = make_dw_args(m->plan, op.dw, m->d_params, m->d_ws, m->dp.ctl);
      else if (op.kind == OP_LINATTN || op.kind == OP_ATTN) cops[i].a = make_attn_args(op.at, m->d_ws);
      else if (op.kind == OP_GN_RES) cops[i].r = make_gnres_args(op.gr, m->d_params, m->d_ws);
      else { sbm_model_destroy(m); return fail("sbm_model_create: op kind cannot run in the chain kernel (" + op.name + ")"); }
    }
    if (any_chain) {
      if ((e = hipMalloc(&m->d_chain_ops, sizeof(ChainOp) * nops)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(chain ops)", e); }
      if ((e = hipMalloc(&m->d_chain_sync, sizeof(ChainSync))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(chain sync)", e); }
      if ((e = hipMemset(m->d_chain_sync, 0, sizeof(ChainSync))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemset(chain sync)", e); }
      if ((e = hipMalloc(&m->d_chain_sched, sizeof(int) * (sched_all.size() + 1))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(chain schedule)", e); }
      if ((e = hipMemcpy(m->d_chain_sched, sched_all.data(), sizeof(int) * sched_all.size(), hipMemcpyHostToDevice)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemcpy(chain schedule)", e); }
      for (size_t i = 0; i < nops; ++i) if (sched_at[i] != (size_t)-1) cops[i].gm.sched = m->d_chain_sched + sched_at[i];
      if ((e = hipMemcpy(m->d_chain_ops, cops.data(), sizeof(ChainOp) * nops, hipMemcpyHostToDevice)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMemcpy(chain ops)", e); }
      int dev = 0, cus = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      m->dp.chain_ops = m->d_chain_ops; m->dp.chain_sync = m->d_chain_sync;
      // The kernel's barriers need the whole grid resident at once: the grid is (workgroups per CU the occupancy calculator
      // grants this kernel) x CUs, verified here -- a device that cannot hold it gets an error, never a different numeric path.
      if (cus <= 0) { sbm_model_destroy(m); return fail("sbm_model_create: cannot read the CU count"); }
      if (const char* ge = chain_launch_geometry(cus, &m->dp.chain_grid, &m->dp.chain_block)) { std::string msg = ge; sbm_model_destroy(m); return fail("sbm_model_create: " + msg); }
      if ((e = hipHostMalloc(reinterpret_cast<void**>(&m->h_abort), 64, hipHostMallocMapped)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipHostMalloc(abort flag)", e); }
      *m->h_abort = 0;
      unsigned* dev_view = nullptr;
      if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_view), m->h_abort, 0)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipHostGetDevicePointer", e); }
      m->dp.chain_ctl.abort_host = dev_view;
    }
  }
  if (kind == SBM_UNET || kind == SBM_UNET_OA || kind == SBM_MLP) {
    const size_t sbytes = (size_t)batch * m->plan.out_C * m->plan.out_H * m->plan.out_W * sizeof(float);
    if ((e = hipMalloc(&m->d_score, sbytes)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(score)", e); }
    if ((e = hipMalloc(&m->d_norms, (size_t)batch * 2 * sizeof(float))) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipMalloc(norms)", e); }
  }
  if ((e = hipStreamCreateWithFlags(&m->own, hipStreamNonBlocking)) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipStreamCreate", e); }
  (void)hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming); (void)hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming);
  (void)hipEventCreate(&m->ev_t0); (void)hipEventCreate(&m->ev_t1);
  if ((e = hipDeviceSynchronize()) != hipSuccess) { sbm_model_destroy(m); return fail_hip("hipDeviceSynchronize", e); }
  *out = m;
  return 0;
}

int sbm_model_info(const sbm_model* m, double* info) {
  if (!m || !info) return fail("sbm_model_info: bad arguments");
  int ng = 0; for (auto& op : m->plan.ops) ng += op.kind == OP_GEMM;
  info[0] = (double)m->plan.ops.size(); info[1] = (double)m->plan.ws_floats * 4; info[2] = 0;
  info[3] = m->plan.flops_dense; info[4] = m->plan.flops_exec; info[5] = ng; info[6] = m->plan.B; info[7] = m->plan.flops_dense_gemm;
  info[8] = m->dp.use_pix ? 1.0 : 0.0; info[9] = m->pix_frag_mb;
  return 0;
}

// Why a Unet's low-resolution levels do NOT run in the sample-owned pixel kernel (empty string: they do, or the model is not a Unet).
// The choice is made once, in sbm_model_create, and kept here: never silent, never at launch time.
int sbm_model_note(const sbm_model* m, char* buf, int cap) {
  if (!m || !buf || cap <= 0) return fail("sbm_model_note: bad arguments");
  std::strncpy(buf, m->pix_reason.c_str(), cap - 1); buf[cap - 1] = 0;
  return 0;
}

int sbm_model_set_aux(sbm_model* m, const float* aux_dev) {
  if (!m) return fail("sbm_model_set_aux: bad arguments");
  if (m->plan.aux_off < 0 && aux_dev) return fail("sbm_model_set_aux: this model variant takes no auxiliary input");
  m->aux_persistent = aux_dev;
  return 0;
}

int sbm_model_forward_aux(sbm_model* m, const float* in_dev, const float* t_dev, const float* aux_dev, float* out_dev, void* stream) {
  if (!m || !in_dev || !out_dev) return fail("sbm_model_forward_aux: bad arguments");
  if ((m->plan.aux_off >= 0) != (aux_dev != nullptr)) return fail("sbm_model_forward_aux: this model variant was built with_z / without z; pass z accordingly");
  m->dp.aux = aux_dev;
  const int rc = sbm_model_forward(m, in_dev, t_dev, out_dev, stream);
  m->dp.aux = nullptr;
  return rc;
}

int sbm_model_forward(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, void* stream) {
  if (!m || !in_dev || !out_dev) return fail("sbm_model_forward: bad arguments");
  if (m->plan.needs_time && !t_dev) return fail("sbm_model_forward: this score net needs a time vector");
  if (int rc = chain_prior_abort(m, (hipStream_t)stream, "sbm_model_forward")) return rc;
  const char* e = run_plan(m->dp, in_dev, t_dev, out_dev, (hipStream_t)stream);
  return e ? fail(std::string("sbm_model_forward: ") + e) : 0;
}

// ---- debug entry points (tests/tools only): run the first n_ops ops, read back the workspace ----
int sbm_debug_forward_partial(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int n_ops, void* stream) {
  if (!m || !in_dev || !out_dev) return fail("sbm_debug_forward_partial: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (m->plan.time_off >= 0 && t_dev) HIPCHK(hipMemcpyAsync(m->d_ws + m->plan.time_off, t_dev, sizeof(float) * m->plan.B, hipMemcpyDeviceToDevice, st));
  for (int i = 0; i < n_ops && i < (int)m->plan.ops.size(); ++i) {
    const char* e = launch_op(m->dp, (int)i, in_dev, out_dev, st);
    if (e) return fail(e);
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
int sbm_debug_forward_range(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int first, int last, void* stream) {
  if (!m || !in_dev || !out_dev) return fail("sbm_debug_forward_range: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (first == 0 && m->plan.time_off >= 0 && t_dev) HIPCHK(hipMemcpyAsync(m->d_ws + m->plan.time_off, t_dev, sizeof(float) * m->plan.B, hipMemcpyDeviceToDevice, st));
  const int nops = (int)m->plan.ops.size();
  for (int i = first; i < last && i < nops;) {
    int n = 1;
    const char* e;
    if (m->plan.ops[i].chain) {                      // like run_plan: consecutive chain ops of the range share one launch
      while (i + n < last && i + n < nops && m->plan.ops[i + n].chain) ++n;
      e = launch_chain(m->dp, i, n, st, out_dev);
    } else if (m->plan.ops[i].row) {
      while (i + n < last && i + n < nops && m->plan.ops[i + n].row) ++n;
      e = launch_rowrun(m->dp, i, n, st);
    } else {
      e = launch_op(m->dp, i, in_dev, out_dev, st);
    }
    if (e) return fail(e);
    i += n;
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
// run op `i` alone with the diagnostic stamp buffer armed; stamps[0..7] = cycles {prologue, loop, epilogue, wait, issue,
// compute} of the middle block's wave 0, chunk count, grid size
int sbm_debug_gemm_stamps(sbm_model* m, const float* in_dev, float* out_dev, int i, unsigned long long* stamps_host) {
  if (!m || i < 0 || i >= (int)m->plan.ops.size() || m->plan.ops[i].kind != OP_GEMM) return fail("sbm_debug_gemm_stamps: not a GEMM op");
  unsigned long long* d = nullptr;
  const char* tl = getenv("SBM_GEMM_TIMELINE");          // diagnostics: file that receives every workgroup's (start, end, where) triple
  const size_t bytes = 128 + (tl ? 6 * 8 * 65536 : 0);
  HIPCHK(hipMalloc(&d, bytes)); HIPCHK(hipMemset(d, 0, bytes));
  if (tl) { const unsigned long long one = 1; HIPCHK(hipMemcpy(d + 15, &one, 8, hipMemcpyHostToDevice)); }
  g_gemm_dbg = d;
  const char* e = launch_op(m->dp, i, in_dev, out_dev, m->own);
  g_gemm_dbg = nullptr;
  if (e) return fail(e);
  HIPCHK(hipStreamSynchronize(m->own));
  HIPCHK(hipMemcpy(stamps_host, d, 128, hipMemcpyDeviceToHost));
  if (tl) {
    std::vector<unsigned long long> h(16 + 6 * 65536);
    HIPCHK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(tl, "a")) {
      const int grid = (int)h[7] < 65536 ? (int)h[7] : 65536;
      fprintf(f, "op %d grid %d\n", i, grid);
      for (int b = 0; b < grid; ++b) fprintf(f, "%d %llu %llu %llx %llu %llu %llu %llu\n", b, h[16 + 6 * b], h[17 + 6 * b], h[18 + 6 * b], h[19 + 6 * b], h[20 + 6 * b], h[21 + 6 * b] >> 32, h[21 + 6 * b] & 0xffffffffull);
      fclose(f);
    }
  }
  (void)hipFree(d);
  return 0;
}
long sbm_debug_workspace_floats(const sbm_model* m) { return m ? (long)m->plan.ws_floats : -1; }
int sbm_debug_read_workspace(sbm_model* m, long off, long n, float* host) {
  if (!m || !host || off < 0 || off + n > (long)m->plan.ws_floats) return fail("sbm_debug_read_workspace: bad range");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, m->d_ws + off, sizeof(float) * n, hipMemcpyDeviceToHost));
  return 0;
}
int sbm_debug_op_name(const sbm_model* m, int i, char* name, int cap) {
  if (!m || i < 0 || i >= (int)m->plan.ops.size() || cap <= 0) return fail("sbm_debug_op_name: bad index");
  std::string s = std::to_string(m->plan.ops[i].kind) + ":" + m->plan.ops[i].name;
  std::strncpy(name, s.c_str(), cap - 1); name[cap - 1] = 0;
  return 0;
}

// test knobs of the chain kernel: barrier poll limit (0 = default) and "one workgroup never arrives" (forces the time-out path)
int sbm_debug_chain_ctl(sbm_model* m, unsigned spin_limit, int skip_arrive) {
  if (!m) return fail("sbm_debug_chain_ctl: bad arguments");
  m->dp.chain_ctl.spin_limit = spin_limit; m->dp.chain_ctl.debug_skip = skip_arrive ? 1u : 0u;
  return 0;
}

// chain-kernel synchronisation state (count[8], arrived, done, abort_flag, pad[5], bar[128], left[128]); all zero between launches
int sbm_debug_chain_sync(sbm_model* m, unsigned* out, int n_words) {
  if (!m || !out || n_words <= 0) return fail("sbm_debug_chain_sync: bad arguments");
  std::memset(out, 0, sizeof(unsigned) * n_words);
  if (!m->d_chain_sync) return 0;
  HIPCHK(hipDeviceSynchronize());
  const size_t n = sizeof(ChainSync) < sizeof(unsigned) * n_words ? sizeof(ChainSync) : sizeof(unsigned) * n_words;
  HIPCHK(hipMemcpy(out, m->d_chain_sync, n, hipMemcpyDeviceToHost));
  return 0;
}

// run chain ops [first, first+count) in one launch with time stamps armed: out[0] = start, out[1+2k] = workgroup 0 of the first
// cluster finished op k, out[2+2k] = its cluster passed the barrier behind op k (s_memtime ticks, ~2.4 GHz)
int sbm_debug_chain_stamps(sbm_model* m, int first, int count, unsigned long long* out_host) {
  if (!m || !out_host || count <= 0 || count > 256) return fail("sbm_debug_chain_stamps: bad arguments");
  unsigned long long* d = nullptr;
  if (count > CHAIN_MAX_OPS) return fail("sbm_debug_chain_stamps: too many ops for one launch");
  HIPCHK(hipMalloc(&d, sizeof(unsigned long long) * 1024));
  HIPCHK(hipMemset(d, 0, sizeof(unsigned long long) * 1024));
  g_chain_dbg = d;
  const char* e = launch_chain(m->dp, first, count, m->own, m->d_score);
  g_chain_dbg = nullptr;
  if (e) { (void)hipFree(d); return fail(e); }
  HIPCHK(hipStreamSynchronize(m->own));
  HIPCHK(hipMemcpy(out_host, d, sizeof(unsigned long long) * (1 + 10 * count), hipMemcpyDeviceToHost));   // + 8 trace slots per op
  (void)hipFree(d);
  return 0;
}

// run row-kernel ops [first, first + count) in one launch with time stamps armed: out[i] = program op i starts (workgroup 0), out[n] = end;
// kinds[i] = op kind * 1000 + GEMM passes.  Returns the number of program ops (staging pseudo-ops included) through *n_out.
int sbm_debug_row_stamps(sbm_model* m, int first, int count, unsigned long long* out_host, int* kinds, int cap, int* n_out) {
  if (!m || !out_host || !kinds || !n_out || cap <= 0) return fail("sbm_debug_row_stamps: bad arguments");
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(unsigned long long) * 256));
  HIPCHK(hipMemset(d, 0, sizeof(unsigned long long) * 256));
  for (int rep = 0; rep < 3; ++rep) {
    g_row_dbg = d;
    const char* e = launch_rowrun(m->dp, first, count, m->own);
    g_row_dbg = nullptr;
    if (e) { (void)hipFree(d); return fail(e); }
    HIPCHK(hipStreamSynchronize(m->own));
  }
  const int n = rowrun_program_ops(m->dp, first, count, kinds, cap);
  if (n < 0 || n + 1 > 256) { (void)hipFree(d); return fail("sbm_debug_row_stamps: no such program"); }
  HIPCHK(hipMemcpy(out_host, d, sizeof(unsigned long long) * (n + 1 < cap ? n + 1 : cap), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  *n_out = n;
  return 0;
}

// the same for the pixel kernel (pixrun.hip): stamps of workgroup 0 at the start of every program op of the run [first, first + count)
int sbm_debug_pix_stamps(sbm_model* m, int first, int count, unsigned long long* out_host, int* kinds, int cap, int* n_out) {
  if (!m || !out_host || !kinds || !n_out || cap <= 0) return fail("sbm_debug_pix_stamps: bad arguments");
  if (!m->dp.use_pix) return fail("sbm_debug_pix_stamps: the pixel kernel is not in use for this model");
  for (int i = first; i < first + count; ++i) if (i < 0 || i >= (int)m->plan.ops.size() || !m->plan.ops[i].chain || m->plan.ops[i].time_path) return fail("sbm_debug_pix_stamps: not a run of pixel-kernel ops");
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(unsigned long long) * 512));
  HIPCHK(hipMemset(d, 0, sizeof(unsigned long long) * 512));
  for (int rep = 0; rep < 3; ++rep) {
    g_pix_dbg = d;
    const char* e = launch_pixrun(m->dp, first, count, m->own, m->d_score);
    g_pix_dbg = nullptr;
    if (e) { (void)hipFree(d); return fail(e); }
    HIPCHK(hipStreamSynchronize(m->own));
  }
  const int n = pixrun_program_ops(m->dp, first, count, kinds, cap);
  if (n < 0 || n + 1 > 128) { (void)hipFree(d); return fail("sbm_debug_pix_stamps: no such program (or more than 127 program ops)"); }
  HIPCHK(hipMemcpy(out_host, d, sizeof(unsigned long long) * (cap < 512 ? cap : 512), hipMemcpyDeviceToHost));      // op stamps [0, n]; per GEMM pass (in program order) four phase stamps from 128
  (void)hipFree(d);
  *n_out = n;
  return 0;
}

// per-op variant of sbm_model_profile: ms_per_op[i], flops_per_op[i], grid_per_op[i] for i < n_ops
int sbm_debug_profile_ops(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps, double* ms_per_op, double* flops_per_op, int n_ops) {
  if (!m || !in_dev || !out_dev || reps <= 0 || !ms_per_op) return fail("sbm_debug_profile_ops: bad arguments");
  hipStream_t st = m->own;
  const char* e = run_plan(m->dp, in_dev, t_dev, out_dev, st);
  if (e) return fail(e);
  HIPCHK(hipStreamSynchronize(st));
  const size_t nops = m->plan.ops.size();
  std::vector<hipEvent_t> ev(2 * nops);
  for (auto& x : ev) HIPCHK(hipEventCreate(&x));
  for (int i = 0; i < n_ops; ++i) { ms_per_op[i] = 0; if (flops_per_op) flops_per_op[i] = i < (int)nops ? m->plan.ops[i].flops : 0; }
  for (int r = 0; r < reps; ++r) {
    for (size_t i = 0; i < nops; ++i) {
      arm_profile_events(ev[2 * i], ev[2 * i + 1]);
      e = launch_op(m->dp, (int)i, in_dev, out_dev, st);
      arm_profile_events(nullptr, nullptr);
      if (e) return fail(e);
    }
    HIPCHK(hipStreamSynchronize(st));
    for (size_t i = 0; i < nops && (int)i < n_ops; ++i) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); ms_per_op[i] += ms / reps; }
  }
  for (auto& x : ev) (void)hipEventDestroy(x);
  return 0;
}

// Launches are issued exactly as run_plan issues them (a run of chain ops = ONE chain_kernel launch) and each carries its
// own event pair; bucket k < 11 = op kind k launched as its own kernel, bucket 11 = chain_kernel, 13 = rowrun_kernel, 14 = pixrun_kernel.
int sbm_model_profile(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps, double* ms_by_kind, int* launches_by_kind, int n_kinds) {
  return sbm_model_profile_mix(m, in_dev, t_dev, out_dev, reps, 0, ms_by_kind, launches_by_kind, n_kinds);
}

// sampler_mix != 0: the launch mix of one step of sbm_sampler_run, where the time path comes from the per-run table and is not launched
int sbm_model_profile_mix(sbm_model* m, const float* in_dev, const float* t_dev, float* out_dev, int reps, int sampler_mix, double* ms_by_kind, int* launches_by_kind, int n_kinds) {
  if (!m || !in_dev || !out_dev || reps <= 0 || !ms_by_kind || !launches_by_kind) return fail("sbm_model_profile: bad arguments");
  const bool skip_time = sampler_mix && m->plan.cond_off >= 0 && !std::getenv("SBM_NO_TIME_TABLE");
  for (int k = 0; k < n_kinds; ++k) { ms_by_kind[k] = 0; launches_by_kind[k] = 0; }
  hipStream_t st = m->own;
  const char* e = run_plan(m->dp, in_dev, t_dev, out_dev, st);       // warm-up
  if (e) return fail(e);
  HIPCHK(hipStreamSynchronize(st));
  const int nops = (int)m->plan.ops.size();
  std::vector<hipEvent_t> ev(2 * (size_t)nops);
  for (auto& x : ev) HIPCHK(hipEventCreate(&x));
  std::vector<int> bucket;
  for (int r = 0; r < reps; ++r) {
    if (m->plan.time_off >= 0 && t_dev) HIPCHK(hipMemcpyAsync(m->d_ws + m->plan.time_off, t_dev, sizeof(float) * m->plan.B, hipMemcpyDeviceToDevice, st));
    bucket.clear();
    for (int i = 0; i < nops;) {
      int n = 1;
      if (skip_time && m->plan.ops[i].time_path) { ++i; continue; }
      const size_t l = bucket.size();
      arm_profile_events(ev[2 * l], ev[2 * l + 1]);            // events ride on the kernel's own dispatch: begin -> end of the kernel
      if (m->plan.ops[i].fuse_next == 2) {
        e = launch_attn_block(m->dp, i, st);
        bucket.push_back(OP_LINATTN);
        n = 4;
      } else if (m->plan.ops[i].fuse_next == 1) {
        e = launch_qkv_linattn(m->dp, i, st);
        bucket.push_back(OP_LINATTN);
        n = 2;
      } else if (m->plan.ops[i].chain) {
        while (i + n < nops && m->plan.ops[i + n].chain && !(skip_time && m->plan.ops[i + n].time_path)) ++n;
        if (n > CHAIN_MAX_OPS) n = CHAIN_MAX_OPS;              // one launch per event pair
        e = launch_chain(m->dp, i, n, st, out_dev);
        bucket.push_back(m->dp.use_pix ? 14 : 11);
      } else if (m->plan.ops[i].row) {
        while (i + n < nops && m->plan.ops[i + n].row) ++n;
        e = launch_rowrun(m->dp, i, n, st);
        bucket.push_back(13);
      } else {
        e = launch_op(m->dp, i, in_dev, out_dev, st);
        bucket.push_back(m->plan.ops[i].kind);
      }
      arm_profile_events(nullptr, nullptr);
      if (e) return fail(e);
      i += n;
    }
    HIPCHK(hipStreamSynchronize(st));
    for (size_t l = 0; l < bucket.size(); ++l) {
      float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev[2 * l], ev[2 * l + 1]));
      const int k = bucket[l];
      if (k < n_kinds) { ms_by_kind[k] += ms / reps; if (r == 0) launches_by_kind[k] += 1; }
    }
  }
  for (auto& x : ev) (void)hipEventDestroy(x);
  return 0;
}

// Dense-convention GEMM FLOP (2*MAC, zero-padding taps counted) per forward, split by the kernel that executes them:
// out[0] = seg_gemm launches, out[1] = chain_kernel launches; out[2], out[3] = the parts of out[0], out[1] that belong to the
// time path (not launched per step inside sbm_sampler_run).
int sbm_model_gemm_flops(const sbm_model* m, double* out6) {
  if (!m || !out6) return fail("sbm_model_gemm_flops: bad arguments");
  double chain = 0.0, chain_t = 0.0, seg_t = m->plan.flops_dense_cond, other = 0.0, row = 0.0;
  for (const Op& op : m->plan.ops) {
    if (op.kind != OP_GEMM) continue;
    if (op.chain) { chain += op.flops_dense; if (op.time_path) chain_t += op.flops_dense; }
    else if (op.row) row += op.flops_dense;
    else if (op.time_path) seg_t += op.flops_dense;
    else if (op.fuse_next) other += op.flops_dense;           // executed inside linattn_qkv_kernel
  }
  out6[0] = m->plan.flops_dense_gemm - chain - other - row; out6[1] = chain; out6[2] = seg_t; out6[3] = chain_t; out6[4] = row; out6[5] = other;
  return 0;
}

static StepArgs step_args(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t, float* x_out, float* mean_out,
                          int B, int CHW, uint64_t seed, uint64_t sample_offset, int step, int draw) {
  StepArgs a{};
  a.x = x; a.score = score; a.noise = noise; a.t = t; a.x_out = x_out; a.mean_out = mean_out; a.B = B; a.CHW = CHW;
  a.seed = seed; a.sample_offset = sample_offset; a.step_base = step; a.draw = draw; a.sde = make_sde(*sde);
  return a;
}

int sbm_em_step(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t, float* x_out, float* mean_out,
                int B, int CHW, uint64_t seed, uint64_t sample_offset, int step, int draw, void* stream) {
  if (!sde || !x || !score || !t || !x_out || !mean_out || B <= 0 || CHW <= 0) return fail("sbm_em_step: bad arguments");
  if (CHW % 4) return fail("sbm_em_step: C*H*W must be a multiple of 4");
  StepArgs a = step_args(sde, x, score, noise, t, x_out, mean_out, B, CHW, seed, sample_offset, step, draw);
  launch_em_step(a, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_langevin_step(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t, const float* alphas,
                      float* x_out, float* mean_out, float* norms_scratch, int B, int CHW, uint64_t seed, uint64_t sample_offset, int step, int draw, void* stream) {
  if (!sde || !x || !score || !t || !x_out || !mean_out || !norms_scratch || B <= 0 || CHW <= 0) return fail("sbm_langevin_step: bad arguments");
  if (sde->kind != 2 && !alphas) return fail("sbm_langevin_step: VP/subVP need the alphas table");
  if (CHW % 4) return fail("sbm_langevin_step: C*H*W must be a multiple of 4");
  StepArgs a = step_args(sde, x, score, noise, t, x_out, mean_out, B, CHW, seed, sample_offset, step, draw);
  a.alphas = alphas; a.norms = norms_scratch;
  launch_langevin_step(a, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- classifier / energy guidance ----
// the guidance network's weights and shape into a launch argument block (pointers stay valid while the handle lives)
static int clg_bind(sbm_model* m, ClgArgs* pa, const char* who) {
  ClgArgs& a = *pa;
  if (!m || m->kind != SBM_CLG) return fail(std::string(who) + ": not a guidance model handle");
  const float* p = m->d_params; const long long* o = m->clg_off;
  a.Wt_T = p + o[0]; a.bt = p + o[1]; a.emb = o[2] >= 0 ? p + o[2] : nullptr; a.W1 = p + o[3]; a.W1_T = p + o[4]; a.b1 = p + o[5];
  a.W2 = p + o[6]; a.W2_T = p + o[7]; a.b2 = p + o[8]; a.W3 = p + o[9]; a.b3 = p + o[10]; a.freq = p + o[11];
  a.D0 = (int)o[12]; a.size_z = m->clg_cfg[2]; a.n_class = m->clg_cfg[3]; a.n_emb = m->clg_cfg[0] == 3 ? 1 : 0;
  if (a.n_emb && (a.id1 < 0 || a.id1 >= 10 || a.id2 < 0 || a.id2 >= 10)) return fail(std::string(who) + ": modality ids must be in [0, 10)");
  return 0;
}

static int clg_run(sbm_model* m, ClgArgs a, void* stream, const char* who) {
  if (!a.t || a.B <= 0) return fail(std::string(who) + ": bad arguments");
  if (int rc = clg_bind(m, &a, who)) return rc;
  const char* e = launch_clg(a, (hipStream_t)stream);
  if (e) return fail(std::string(who) + ": " + e);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_clg_forward(sbm_model* m, const float* x_pair, const float* t, int id1, int id2, float* out, int B, void* stream) {
  if (!x_pair || !out) return fail("sbm_clg_forward: bad arguments");
  ClgArgs a{}; a.x_pair = x_pair; a.t = t; a.id1 = id1; a.id2 = id2; a.out = out; a.B = B;
  return clg_run(m, a, stream, "sbm_clg_forward");
}

int sbm_clg_guide(sbm_model* m, const float* x, const float* t, int ch1, int ch2, int id1, int id2, float cl_s, unsigned upd_mask,
                  float* score_inout, float* grad_out, int B, int C, int per_mod, void* stream) {
  if (!m || !x || (!score_inout && !grad_out)) return fail("sbm_clg_guide: bad arguments");
  if (m->kind == SBM_CLG && per_mod != m->clg_cfg[2]) return fail("sbm_clg_guide: a modality's latent size must equal the MLP's size_z");
  if (ch1 < 0 || ch1 >= C || ch2 < 0 || ch2 >= C) return fail("sbm_clg_guide: modality channel out of range");
  ClgArgs a{}; a.x = x; a.C = C; a.per_mod = per_mod; a.ch1 = ch1; a.ch2 = ch2; a.t = t; a.id1 = id1; a.id2 = id2;
  a.score = score_inout; a.cl_s = cl_s; a.upd_mask = upd_mask; a.grad_out = grad_out; a.B = B;
  return clg_run(m, a, stream, "sbm_clg_guide");
}

int sbm_langevin_norms(const float* score, const float* noise, float* norms_scratch, double* sums_out, int B, int CHW, uint64_t seed,
                       uint64_t sample_offset, int step, int draw, void* stream) {
  if (!score || !norms_scratch || !sums_out || B <= 0 || CHW <= 0 || CHW % 4) return fail("sbm_langevin_norms: bad arguments (C*H*W must be a positive multiple of 4)");
  StepArgs a{};
  a.score = score; a.noise = noise; a.norms = norms_scratch; a.B = B; a.CHW = CHW; a.seed = seed; a.sample_offset = sample_offset; a.step_base = step; a.draw = draw;
  launch_langevin_norms(a, sums_out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_langevin_update(const sbm_sde_cfg* sde, const float* x, const float* score, const float* noise, const float* t, const float* alphas,
                        float* x_out, float* mean_out, const double* norm_sums, long B_global, int B, int CHW, uint64_t seed, uint64_t sample_offset,
                        int step, int draw, void* stream) {
  if (!sde || !x || !score || !t || !x_out || !mean_out || !norm_sums || B <= 0 || B_global < B || CHW <= 0 || CHW % 4) return fail("sbm_langevin_update: bad arguments");
  if (sde->kind != 2 && !alphas) return fail("sbm_langevin_update: VP/subVP need the alphas table");
  StepArgs a = step_args(sde, x, score, noise, t, x_out, mean_out, B, CHW, seed, sample_offset, step, draw);
  a.alphas = alphas; a.norm_sums = norm_sums; a.B_global = B_global;
  launch_langevin_update(a, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_langevin_fixed_step(const float* x, const float* score, const float* noise, float* x_out, float a, float b, float score_div, unsigned mod_mask, int B,
                            int n_mod, int per_mod, uint64_t seed, uint64_t sample_offset, int step, int draw, void* stream) {
  if (!x || !score || !x_out || B <= 0 || n_mod <= 0 || n_mod > 32 || per_mod <= 0 || per_mod % 4 || !(score_div != 0.f)) return fail("sbm_langevin_fixed_step: bad arguments");
  StepArgs s{};
  s.x = x; s.score = score; s.noise = noise; s.x_out = x_out; s.B = B; s.CHW = n_mod * per_mod; s.seed = seed; s.sample_offset = sample_offset;
  s.step_base = step; s.draw = draw;
  launch_axpy_noise(s, a, b, score_div, mod_mask, per_mod, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_ddpm_step(const float* x, const float* eps, const float* noise, float* x_out, float sqrt_recip_alpha, float beta, float sqrt_one_minus_abar,
                  float sigma, int B, long chw, uint64_t seed, uint64_t sample_offset, int step, void* stream) {
  if (!x || !eps || !x_out || B <= 0 || chw <= 0 || chw % 4 || chw > 0x7fffffffL || !(sqrt_one_minus_abar > 0.f)) return fail("sbm_ddpm_step: bad arguments (chw must be a positive multiple of 4)");
  StepArgs s{};
  s.x = x; s.score = eps; s.noise = noise; s.x_out = x_out; s.B = B; s.CHW = (int)chw; s.seed = seed; s.sample_offset = sample_offset; s.step_base = step; s.draw = 0;
  launch_ddpm_step(s, sqrt_recip_alpha, beta, sqrt_one_minus_abar, sigma, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_renoise_observed(const sbm_sde_cfg* sde, const float* x, const float* z_obs, const float* t, float* x_out, unsigned obs_mask, int B, int n_mod,
                         int per_mod, void* stream) {
  if (!sde || !x || !z_obs || !t || !x_out || B <= 0 || n_mod <= 0 || n_mod > 32 || per_mod <= 0 || per_mod % 4) return fail("sbm_renoise_observed: bad arguments");
  StepArgs s{};
  s.x = x; s.t = t; s.x_out = x_out; s.B = B; s.CHW = n_mod * per_mod; s.sde = make_sde(*sde);
  launch_renoise(s, z_obs, obs_mask, per_mod, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_fill_normal(float* out, long n_per_sample, int B, uint64_t seed, uint64_t sample_offset, int step, int draw, float scale, void* stream) {
  if (!out || n_per_sample <= 0 || n_per_sample % 4 || B <= 0) return fail("sbm_fill_normal: bad arguments (n_per_sample must be a positive multiple of 4)");
  launch_fill_normal(out, n_per_sample, B, seed, sample_offset, (unsigned)step, (unsigned)draw, scale, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// The plan's input conversion op when the sampler's update kernels can write its output themselves (sampler.hip store_pbc): an unconditional
// loop over a state without spatial padding and with rows of a multiple of four elements; -1 otherwise (the forward then converts as usual).
static int sampler_cvt_op(const sbm_model* m, bool conditional) {
  if (conditional || std::getenv("SBM_NO_FUSED_CVT")) return -1;
  const Plan& P = m->plan;
  for (int i = 0; i < (int)P.ops.size(); ++i) {
    if (P.ops[i].kind != OP_NCHW_TO_PBC) continue;
    const CvtOp& cv = P.ops[i].cv;
    const int Ho = cv.out.H, Wo = cv.rowpack ? cv.Wo : cv.out.W;
    if (cv.pad_y || cv.pad_x || Ho != cv.Hn || Wo != cv.Wn || (cv.Wn & 3) || cv.C != P.out_C || cv.Hn != P.out_H || cv.Wn != P.out_W) return -1;
    return i;
  }
  return -1;
}

// one full sampling step enqueued on `st` (graph-capturable: no allocation, no sync, step index read on device)
static const char* enqueue_step(sbm_model* m, const SdeParams& sp, const sbm_sampler_cfg& c, const float* timesteps, const float* alphas, const float* noise,
                                float* x, float* x_mean, hipStream_t st, bool time_table, const sbm_cond_cfg* cond = nullptr, const float* z_obs = nullptr,
                                const std::vector<ClgArgs>* guide = nullptr) {
  const Plan& P = m->plan;
  const int B = P.B, CHW = P.out_C * P.out_H * P.out_W;
  const int dps = 1 + (c.pc ? c.n_steps : 0);
  float* tbuf = m->d_ws + P.time_off;
  StepArgs a{};
  a.x = x; a.score = m->d_score; a.noise = noise; a.t = nullptr; a.timesteps = timesteps; a.step_counter = m->d_counter; a.alphas = alphas;
  a.x_out = x; a.mean_out = x_mean; a.norms = m->d_norms; a.seed = c.seed; a.sample_offset = c.sample_offset; a.B = B; a.CHW = CHW; a.sde = sp;
  a.noise_step_stride = (long long)dps * B * CHW;
  const bool guided = guide && !guide->empty();
  // The update kernels keep the score net's converted input current (sampler.hip store_pbc), so the forward skips its conversion launch:
  // unconditional loops only (the conditional ones rewrite x in front of the forward).  sampler_run_impl converts the initial state once.
  const int cvt = sampler_cvt_op(m, cond != nullptr);
  if (cvt >= 0) {
    const CvtOp& cv = P.ops[cvt].cv;
    a.pbc_out = m->d_ws + cv.out.off; a.pbc_ld = cv.out.ld; a.pbc_rowpack = cv.rowpack; a.pbc_Cv = cv.rowpack ? cv.Cpad : cv.out.C; a.pbc_H = cv.Hn; a.pbc_W = cv.Wn;
  }
  const int skip = cvt >= 0 ? 4 : 0;
  // The EM predictor rides on the kernel that ends the score net when that is a pixel-kernel run with the score in LDS (pixrun.hip kernel tail):
  // no guidance (it rewrites the score in between), 8x8 states.  SBM_NO_FUSED_EM=1: the update kernel of its own.
  const bool fuse_em = !guided && !std::getenv("SBM_NO_FUSED_EM") && P.out_H * P.out_W == 64 && pixrun_can_fuse_em(m->dp);
  auto score_eval = [&](const StepArgs* em = nullptr) -> const char* {
    const char* e;
    if (time_table) {
      if (guided) launch_fill_time(tbuf, timesteps, m->d_counter, B, st);   // the guidance networks take the step's time per sample (the score net reads the table)
      e = run_plan(m->dp, x, tbuf, m->d_score, st, 1 | skip, em);           // the time path comes from the table
    } else {
      launch_fill_time(tbuf, timesteps, m->d_counter, B, st);
      e = run_plan(m->dp, x, tbuf, m->d_score, st, skip, em);
    }
    if (e) return e;
    if (guided)                    // classifier / energy guidance on the score just evaluated, at the same x and t (sde_helper2.py:283-312, 65-94)
      for (ClgArgs a : *guide) { a.x = x; a.t = tbuf; a.score = m->d_score; if ((e = launch_clg(a, st))) return e; }
    return nullptr;
  };
  // the step's LAST update kernel also advances the step counter (its last workgroup, sampler.hip advance_by_last_workgroup)
  unsigned* ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(m->d_counter) + 48);
  auto corrector = [&](int first_draw, bool last) -> const char* {
    for (int i = 0; i < c.n_steps; ++i) {
      const char* e = score_eval(); if (e) return e;
      a.draw = first_draw + i; a.noise_draw_off = (long long)(first_draw + i) * B * CHW;
      const bool adv = last && i == c.n_steps - 1;
      a.adv_counter = adv ? m->d_counter : nullptr; a.adv_ticket = adv ? ticket : nullptr;
      launch_langevin_step(a, st);
    }
    return nullptr;
  };
  auto predictor = [&](int draw, bool last) -> const char* {
    a.draw = draw; a.noise_draw_off = (long long)draw * B * CHW;
    a.adv_counter = last ? m->d_counter : nullptr; a.adv_ticket = last ? ticket : nullptr;
    if (fuse_em) return score_eval(&a);
    const char* e = score_eval(); if (e) return e;
    launch_em_step(a, st);
    return nullptr;
  };
  const char* e = nullptr;
  if (cond) {
    // conditional loop (train_lat_celebhq_unet_cont2.py:292-303, train_poly_unet_cont.py:356-368): the observed modalities enter every
    // step as their latents re-noised to level t (noise_obs) or clean; what the previous step wrote there is discarded
    StepArgs r{};
    r.x = x; r.x_out = x; r.t = nullptr; r.timesteps = timesteps; r.step_counter = m->d_counter; r.B = B; r.CHW = CHW; r.sde = sp;
    launch_renoise(r, z_obs, cond->obs_mask, CHW / cond->n_mod, st, cond->noise_obs ? 0 : 1);
  }
  if (c.order == 0) {                      // sde_helper2.uncond_sampler: corrector, then predictor
    if (c.pc && (e = corrector(0, false))) return e;
    if ((e = predictor(c.pc ? c.n_steps : 0, true))) return e;
  } else {                                 // train_poly_unet_cont.uncond_sampler: predictor, then corrector
    if ((e = predictor(0, !c.pc))) return e;
    if (c.pc && (e = corrector(1, true))) return e;
  }
  return nullptr;
}

struct TableModeGuard {        // whatever happens in a loop, later forwards read their own time inputs again
  sbm_model* m; hipStream_t st; bool armed = false;
  ~TableModeGuard() { if (armed) { (void)hipMemsetAsync(reinterpret_cast<char*>(m->d_counter) + 4, 0, 4, st); (void)hipStreamSynchronize(st); } }
};
struct GraphGuard {            // every exit path (HIPCHK returns included) releases the captured step; the stream is drained first
  hipStream_t st; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  ~GraphGuard() { if (exec || graph) (void)hipStreamSynchronize(st); if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
};

// Every sample of a sampling loop shares the iteration's time input: tabulate the time path (sinusoid, time MLP, per-block condition
// GEMM) for all N iterations once, B at a time through the very same ops (hence bit-identical to the per-forward path); the depthwise
// kernels then index the table by the device step counter and the per-iteration launches of the time path disappear.  Leaves the
// model in table mode (the caller's guard switches it back).
static int build_time_table(sbm_model* m, const float* times_dev, int N, const float* x_any, hipStream_t st, const char* who) {
  const int B = m->plan.B, ld = m->plan.cond_ld;
  const size_t need = (size_t)N * ld;
  if (need > m->time_table_floats) {
    if (m->d_time_table) (void)hipFree(m->d_time_table);
    m->d_time_table = nullptr; m->time_table_floats = 0;
    HIPCHK(hipMalloc(&m->d_time_table, need * sizeof(float)));
    m->time_table_floats = need;
  }
  for (int k0 = 0; k0 < N; k0 += B) {
    const int n = N - k0 < B ? N - k0 : B;
    HIPCHK(hipMemcpyAsync(m->d_ws + m->plan.time_off, times_dev + k0, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
    if (const char* e = run_plan(m->dp, x_any, m->d_ws + m->plan.time_off, m->d_score, st, 2)) return fail(std::string(who) + ": " + e);
    HIPCHK(hipMemcpyAsync(m->d_time_table + (size_t)k0 * ld, m->d_ws + m->plan.cond_off, sizeof(float) * (size_t)n * ld, hipMemcpyDeviceToDevice, st));
  }
  StepCtl h{}; h.step = 0; h.table_mode = 1; h.tab_ld = ld; h.pad = 0; h.table = m->d_time_table;
  HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(m->d_counter) + 4, reinterpret_cast<const char*>(&h) + 4, sizeof(StepCtl) - 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));                            // `h` is a stack object
  return 0;
}

static int sampler_run_impl(sbm_model* m, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const sbm_cond_cfg* cond, const float* z_obs,
                            const float* timesteps, const float* alphas, const float* noise, float* x_inout, float* x_mean_out, int first_step,
                            int n_run, float* elapsed_ms, void* stream, const std::vector<ClgArgs>* guide = nullptr) {
  if (!m || (m->kind != SBM_UNET && m->kind != SBM_UNET_OA) || !sde || !cfg || !timesteps || !x_inout || !x_mean_out) return fail("sbm_sampler_run: bad arguments");
  if (cond) {
    if (!z_obs || cond->n_mod <= 0 || cond->n_mod > 32 || cond->n_mod != m->plan.out_C) return fail("sbm_cond_sampler_run: n_mod must equal the score net's channel count; z_obs must be given");
    if (((m->plan.out_H * m->plan.out_W)) % 4) return fail("sbm_cond_sampler_run: a modality's latent size must be a multiple of 4");
  }
  if (m->plan.aux_off >= 0 && !m->aux_persistent) return fail("sbm_sampler_run: the score net was built with z conditioning; call sbm_model_set_aux first");
  m->dp.aux = m->aux_persistent;
  if (!m->plan.needs_time) return fail("sbm_sampler_run: the score net has no time embedding");
  if (first_step < 0 || n_run < 0 || first_step + n_run > sde->N) return fail("sbm_sampler_run: step range outside [0, N)");
  if (cfg->pc && cfg->n_steps <= 0) return fail("sbm_sampler_run: n_steps must be positive when pc is set");
  if (cfg->pc && sde->kind != 2 && !alphas) return fail("sbm_sampler_run: VP/subVP corrector needs the alphas table");
  const int CHW = m->plan.out_C * m->plan.out_H * m->plan.out_W;
  if (CHW % 4) return fail("sbm_sampler_run: C*H*W must be a multiple of 4");
  if (m->plan.in_C != m->plan.out_C) return fail("sbm_sampler_run: score net must map the state onto itself");
  const SdeParams sp = make_sde(*sde);
  hipStream_t caller = (hipStream_t)stream, st = m->own;
  if (int rc = chain_prior_abort(m, st, "sbm_sampler_run")) return rc;
  HIPCHK(hipEventRecord(m->ev_in, caller));
  HIPCHK(hipStreamWaitEvent(st, m->ev_in, 0));
  launch_set_counter(m->d_counter, first_step, st);
  TableModeGuard guard{m, st};
  GraphGuard gg{st};
  hipGraph_t& graph = gg.graph; hipGraphExec_t& exec = gg.exec;
  const char* e = nullptr;
  // Every sample of the loop shares the step's time: tabulate the time path (sinusoid, time MLP, per-block condition GEMM)
  // for all N steps once, B steps per pass through the very same ops; the depthwise kernels then index the table by the
  // step counter and the per-step launches of the time path disappear.
  const bool time_table = m->plan.cond_off >= 0 && m->plan.time_off >= 0 && !std::getenv("SBM_NO_TIME_TABLE");
  if (time_table) {
    if (int rc = build_time_table(m, timesteps, sde->N, x_inout, st, "sbm_sampler_run")) return rc;
    guard.armed = true;
  }
  { const int cvt = sampler_cvt_op(m, cond != nullptr);      // the loop's forwards skip the conversion: the initial state is converted here, once
    if (cvt >= 0 && (e = launch_op(m->dp, cvt, x_inout, nullptr, st))) return fail(std::string("sbm_sampler_run: ") + e); }
  if (cfg->use_graph && n_run > 1) {
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    e = enqueue_step(m, sp, *cfg, timesteps, alphas, noise, x_inout, x_mean_out, st, time_table, cond, z_obs, guide);
    hipError_t ce = hipStreamEndCapture(st, &graph);
    if (e) return fail(std::string("sbm_sampler_run: ") + e);
    if (ce != hipSuccess) return fail_hip("hipStreamEndCapture", ce);
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  }
  if (elapsed_ms) HIPCHK(hipEventRecord(m->ev_t0, st));
  for (int i = 0; i < n_run; ++i) {
    if (exec) { hipError_t le = hipGraphLaunch(exec, st); if (le != hipSuccess) return fail_hip("hipGraphLaunch", le); }
    else if ((e = enqueue_step(m, sp, *cfg, timesteps, alphas, noise, x_inout, x_mean_out, st, time_table, cond, z_obs, guide))) return fail(std::string("sbm_sampler_run: ") + e);
  }
  if (cond && n_run > 0) {                 // the observed modalities leave as their clean latents (:314-316 keep z[mod] for mod in given)
    StepArgs r{};
    r.x = x_mean_out; r.x_out = x_mean_out; r.B = m->plan.B; r.CHW = CHW; r.sde = sp;
    launch_renoise(r, z_obs, cond->obs_mask, CHW / cond->n_mod, st, 1);
  }
  if (elapsed_ms) HIPCHK(hipEventRecord(m->ev_t1, st));
  if (time_table) { HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(m->d_counter) + 4, 0, 4, st)); guard.armed = false; }      // back to per-forward time inputs
  HIPCHK(hipEventRecord(m->ev_out, st));
  HIPCHK(hipStreamWaitEvent(caller, m->ev_out, 0));
  if (elapsed_ms || exec) {
    hipError_t se = hipStreamSynchronize(st);
    if (se != hipSuccess) return fail_hip("hipStreamSynchronize", se);
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, m->ev_t0, m->ev_t1));
    if (m->h_abort && __atomic_load_n(m->h_abort, __ATOMIC_RELAXED)) {    // the stream is idle here: the flag is final for this run
      (void)hipMemset(m->d_chain_sync, 0, sizeof(ChainSync));
      __atomic_store_n(m->h_abort, 0u, __ATOMIC_RELAXED);
      return fail("sbm_sampler_run: a chain-kernel barrier timed out (not every workgroup was resident?); results are invalid");
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int sbm_sampler_run(sbm_model* m, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const float* timesteps, const float* alphas, const float* noise,
                    float* x_inout, float* x_mean_out, int first_step, int n_run, float* elapsed_ms, void* stream) {
  return sampler_run_impl(m, sde, cfg, nullptr, nullptr, timesteps, alphas, noise, x_inout, x_mean_out, first_step, n_run, elapsed_ms, stream);
}

int sbm_cond_sampler_run(sbm_model* m, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const sbm_cond_cfg* cond, const float* z_obs,
                         const float* timesteps, const float* alphas, const float* noise, float* x_inout, float* x_mean_out, int first_step,
                         int n_run, float* elapsed_ms, void* stream) {
  if (!cond) return fail("sbm_cond_sampler_run: bad arguments");
  return sampler_run_impl(m, sde, cfg, cond, z_obs, timesteps, alphas, noise, x_inout, x_mean_out, first_step, n_run, elapsed_ms, stream);
}

// The conditional loop with classifier / energy guidance inside the captured step: after EVERY score evaluation (predictor and each
// corrector iteration) the listed guidance networks subtract cl_s * d cl_g(cat[x_m1, x_m2], t).mean() / dx from the score slices of
// their pair, at the x and t of that evaluation -- sde_helper2.py:283-312 (RSDE.sde) and :65-94 (corrector).
int sbm_guided_cond_sampler_run(sbm_model* m, const sbm_sde_cfg* sde, const sbm_sampler_cfg* cfg, const sbm_cond_cfg* cond, const float* z_obs,
                                const sbm_guide_job* jobs, int n_jobs, float cl_s, const float* timesteps, const float* alphas, const float* noise,
                                float* x_inout, float* x_mean_out, int first_step, int n_run, float* elapsed_ms, void* stream) {
  if (!m || !cond || !jobs || n_jobs <= 0 || n_jobs > 64) return fail("sbm_guided_cond_sampler_run: bad arguments");
  const int C = m->plan.out_C, per_mod = m->plan.out_H * m->plan.out_W;
  std::vector<ClgArgs> guide((size_t)n_jobs);
  for (int j = 0; j < n_jobs; ++j) {
    const sbm_guide_job& jb = jobs[j];
    ClgArgs& a = guide[(size_t)j];
    a = ClgArgs{};
    if (!jb.clg || jb.clg->kind != SBM_CLG) return fail("sbm_guided_cond_sampler_run: job without a guidance model handle");
    if (per_mod != jb.clg->clg_cfg[2]) return fail("sbm_guided_cond_sampler_run: a modality's latent size must equal the guidance MLP's size_z");
    if (jb.ch1 < 0 || jb.ch1 >= C || jb.ch2 < 0 || jb.ch2 >= C) return fail("sbm_guided_cond_sampler_run: modality channel out of range");
    a.C = C; a.per_mod = per_mod; a.ch1 = jb.ch1; a.ch2 = jb.ch2; a.id1 = jb.id1; a.id2 = jb.id2; a.cl_s = cl_s; a.upd_mask = jb.upd_mask; a.B = m->plan.B;
    if (int rc = clg_bind(jb.clg, &a, "sbm_guided_cond_sampler_run")) return rc;
  }
  clg_prepare();
  return sampler_run_impl(m, sde, cfg, cond, z_obs, timesteps, alphas, noise, x_inout, x_mean_out, first_step, n_run, elapsed_ms, stream, &guide);
}

// Device-resident fixed-step Langevin loops: annealed Langevin over a time-conditioned score net (fid_upd10.gen_ae :84-99,
// eval_lat_celeba_hq_all.evaluate :258-275) and the MLP-score chain (fid_upd10.calc_poly_uncond :311-339).  n_iter iterations of
//   score = net(z [, times[k]]);  z[:, m] += a[k][m] * score[:, m] / score_div[k] + b[k][m] * N   (m in mod_mask)
// with the iteration index in device memory, the time path (if the net has one) tabulated once for all iterations, and ONE captured
// graph {score net, update, advance} replayed n_iter times.  coef rows: [a_0..a_{n_mod-1} | b_0..b_{n_mod-1} | score_div].
// tail (host, optional): one more update with the LAST score and its own row -- calc_poly_uncond's closing step (:336-339).
int sbm_langevin_run(sbm_model* m, const float* times, const float* coef, unsigned mod_mask, int n_mod, int n_iter, const float* noise, float* z_inout,
                     uint64_t seed, uint64_t sample_offset, int use_graph, const float* tail, float* elapsed_ms, void* stream) {
  if (!m || (m->kind != SBM_UNET && m->kind != SBM_UNET_OA && m->kind != SBM_MLP) || !coef || !z_inout || n_iter < 0 || n_mod <= 0 || n_mod > 32) return fail("sbm_langevin_run: bad arguments");
  const int B = m->plan.B, CHW = m->plan.out_C * m->plan.out_H * m->plan.out_W;
  if (m->plan.in_C != m->plan.out_C || m->plan.in_H != m->plan.out_H || m->plan.in_W != m->plan.out_W) return fail("sbm_langevin_run: score net must map the state onto itself");
  if (CHW % n_mod || (CHW / n_mod) % 4) return fail("sbm_langevin_run: the state must split into n_mod modalities of a multiple of 4 floats");
  const bool timed = m->plan.needs_time != 0;
  if (timed && (!times || m->plan.cond_off < 0 || m->plan.time_off < 0)) return fail("sbm_langevin_run: a time-conditioned score net needs the times array and a tabulated time path");
  if (m->plan.aux_off >= 0 && !m->aux_persistent) return fail("sbm_langevin_run: the score net was built with z conditioning; call sbm_model_set_aux first");
  if (tail && n_iter == 0) return fail("sbm_langevin_run: a closing step needs at least one iteration (it reuses the last score)");
  m->dp.aux = m->aux_persistent;
  hipStream_t caller = (hipStream_t)stream, st = m->own;
  if (int rc = chain_prior_abort(m, st, "sbm_langevin_run")) return rc;
  HIPCHK(hipEventRecord(m->ev_in, caller));
  HIPCHK(hipStreamWaitEvent(st, m->ev_in, 0));
  launch_set_counter(m->d_counter, 0, st);
  TableModeGuard guard{m, st};
  GraphGuard gg{st};
  if (timed && n_iter > 0) {
    if (int rc = build_time_table(m, times, n_iter, z_inout, st, "sbm_langevin_run")) return rc;
    guard.armed = true;
  }
  StepArgs a{};
  a.x = z_inout; a.x_out = z_inout; a.score = m->d_score; a.noise = noise; a.noise_step_stride = (long long)B * CHW; a.step_counter = m->d_counter;
  a.B = B; a.CHW = CHW; a.seed = seed; a.sample_offset = sample_offset; a.draw = 0;
  auto enqueue = [&]() -> const char* {
    if (const char* e = timed ? run_plan(m->dp, z_inout, m->d_ws + m->plan.time_off, m->d_score, st, 1) : run_plan(m->dp, z_inout, nullptr, m->d_score, st, 0)) return e;
    launch_axpy_noise_tab(a, coef, mod_mask, n_mod, CHW / n_mod, st);              // element-wise: in place is safe
    launch_advance(m->d_counter, st);
    return nullptr;
  };
  if (use_graph && n_iter > 1) {
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const char* e = enqueue();
    hipError_t ce = hipStreamEndCapture(st, &gg.graph);
    if (e) return fail(std::string("sbm_langevin_run: ") + e);
    if (ce != hipSuccess) return fail_hip("hipStreamEndCapture", ce);
    HIPCHK(hipGraphInstantiate(&gg.exec, gg.graph, nullptr, nullptr, 0));
  }
  if (elapsed_ms) HIPCHK(hipEventRecord(m->ev_t0, st));
  for (int i = 0; i < n_iter; ++i) {
    if (gg.exec) { hipError_t le = hipGraphLaunch(gg.exec, st); if (le != hipSuccess) return fail_hip("hipGraphLaunch", le); }
    else if (const char* e = enqueue()) return fail(std::string("sbm_langevin_run: ") + e);
  }
  if (tail) {                              // the closing update: the last iteration's score once more, Philox draw 1 of that iteration
    StepArgs t = a;
    t.step_counter = nullptr; t.step_base = n_iter - 1; t.draw = 1;
    if (noise) { t.noise = noise + (long long)n_iter * B * CHW; t.noise_step_stride = 0; }
    for (int md = 0; md < n_mod; ++md)
      if ((mod_mask >> md) & 1u) launch_axpy_noise(t, tail[md], tail[n_mod + md], tail[2 * n_mod], 1u << md, CHW / n_mod, st);
  }
  if (elapsed_ms) HIPCHK(hipEventRecord(m->ev_t1, st));
  if (guard.armed) { HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(m->d_counter) + 4, 0, 4, st)); guard.armed = false; }
  HIPCHK(hipEventRecord(m->ev_out, st));
  HIPCHK(hipStreamWaitEvent(caller, m->ev_out, 0));
  if (elapsed_ms || gg.exec) {
    hipError_t se = hipStreamSynchronize(st);
    if (se != hipSuccess) return fail_hip("hipStreamSynchronize", se);
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, m->ev_t0, m->ev_t1));
    if (m->h_abort && __atomic_load_n(m->h_abort, __ATOMIC_RELAXED)) {
      (void)hipMemset(m->d_chain_sync, 0, sizeof(ChainSync));
      __atomic_store_n(m->h_abort, 0u, __ATOMIC_RELAXED);
      return fail("sbm_langevin_run: a chain-kernel barrier timed out (not every workgroup was resident?); results are invalid");
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
