// guidance.hip -- classifier / energy guidance of the conditional sampler:
//
//     new_x   = cat[x[:, m1], x[:, m2]]                              (B, 2, size_z)
//     cl_out  = cl_g(new_x.view(B, -1), t [, id1, id2])              (B, n_class)
//     grad    = d cl_out.mean() / d new_x                            (torch.autograd.grad in the reference)
//     score[:, m1] -= cl_s * grad[:, 0];  score[:, m2] -= cl_s * grad[:, 1]
//                                             reference sde_helper2.py:283-312 (RSDE.sde), :65-94 (corrector);
//                                             PolyMNIST random-pair variant train_poly_unet_cont.py:73-89, 129-146
//
// cl_g is a time-conditioned MLP over two latent slices (lat_sm2_model.ClwithTime2 / ClwithTime3 of this package: the reference
// imports those names but no longer defines them -- SURVEY D6):
//     temb = Softplus(Linear(64 -> 128)(sinusoid_64(t)));   e1, e2 = mod_emb[id1], mod_emb[id2]   (ClwithTime3 only)
//     h    = Softplus(Linear(D0 -> 512)([x | temb | e1 | e2]));  h = Softplus(Linear(512 -> 256)(h));  out = Linear(256 -> n_class)(h)
//
// The forward AND the input-VJP run in ONE kernel: a workgroup carries TS samples through the three layers with the
// activations in LDS, then walks back (g = W^T g * sigmoid(pre)), and subtracts cl_s * grad from the score in place.  The
// matrices are stored both ways ([in][out] for the forward sweep, [out][in] = the state_dict layout for the backward sweep) so
// that consecutive threads always read consecutive floats.  HBM/L2-bound by the ~2 MB of weights each workgroup streams.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "device.h"

namespace sbm {

namespace {

constexpr int CLG_TS = 8;        // samples per workgroup
constexpr int CLG_NT = 1024;     // threads: every mat-vec phase is cut into (output neuron) x (K slice) tasks, so that a thread's serial,
                                 // load-latency-bound K loop is 64 ... 128 long instead of 256 ... 512 (r03: 130 -> see DESIGN 6c)
constexpr int CLG_TE = 64, CLG_TH = 128, CLG_EMB = 32, CLG_H1 = 512, CLG_H2 = 256;
constexpr int CLG_PART = CLG_NT * CLG_TS;      // partial sums [K slice][output][sample]

__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }            // nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float dsoftplus_f(float v) { if (v > 20.f) return 1.f; const float z = expf(v); return z / (z + 1.0f); }   // its backward

// One mat-vec phase for the workgroup's TS samples: out[n][s] = sum_k w(k, n) * in[k][s], n < outs, k < K.  Thread (n, slice) walks its
// K slice (consecutive threads read consecutive weights), the slices meet in `part` and are added in slice order by thread n, which
// hands the sums to `fin(n, acc)`.  Two barriers per phase.
template <class W, class F>
__device__ __forceinline__ void clg_phase(int outs, int K, const float* in, float* part, W w, F fin) {
  const int tid = threadIdx.x;
  int ksn = 1;
  while (ksn * 2 * outs <= CLG_NT && ksn * 2 <= K) ksn *= 2;
  for (int n0 = 0; n0 < outs; n0 += CLG_NT) {                 // (outs <= CLG_NT for every layer of this network: one round)
    const int cnt = outs - n0 < CLG_NT ? outs - n0 : CLG_NT;
    const int n = tid % cnt, ks = tid / cnt;
    if (ks < ksn) {
      float acc[CLG_TS];
#pragma unroll
      for (int s = 0; s < CLG_TS; ++s) acc[s] = 0.f;
      const int k0 = (int)((long long)K * ks / ksn), k1 = (int)((long long)K * (ks + 1) / ksn);
#pragma unroll 4
      for (int k = k0; k < k1; ++k) {
        const float wv = w(k, n0 + n);
        const float4 v0 = *reinterpret_cast<const float4*>(in + k * CLG_TS), v1 = *reinterpret_cast<const float4*>(in + k * CLG_TS + 4);
        acc[0] = fmaf(wv, v0.x, acc[0]); acc[1] = fmaf(wv, v0.y, acc[1]); acc[2] = fmaf(wv, v0.z, acc[2]); acc[3] = fmaf(wv, v0.w, acc[3]);
        acc[4] = fmaf(wv, v1.x, acc[4]); acc[5] = fmaf(wv, v1.y, acc[5]); acc[6] = fmaf(wv, v1.z, acc[6]); acc[7] = fmaf(wv, v1.w, acc[7]);
      }
      float4* pp = reinterpret_cast<float4*>(part + ((long long)ks * cnt + n) * CLG_TS);
      pp[0] = make_float4(acc[0], acc[1], acc[2], acc[3]); pp[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    if (tid < cnt) {
      float acc[CLG_TS];
#pragma unroll
      for (int s = 0; s < CLG_TS; ++s) acc[s] = 0.f;
      for (int q = 0; q < ksn; ++q) {
        const float4* pp = reinterpret_cast<const float4*>(part + ((long long)q * cnt + tid) * CLG_TS);
        const float4 v0 = pp[0], v1 = pp[1];
        acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w; acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
      }
      fin(n0 + tid, acc);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(CLG_NT) void clg_kernel(const ClgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // LDS: a0 [D0][TS] | a1 [H1][TS] | s1 [H1][TS] | a2 [H2][TS] | s2 [H2][TS] | te [TE][TS] | part [NT][TS]
  float* a0 = sm;
  float* a1 = a0 + a.D0 * CLG_TS;
  float* s1 = a1 + CLG_H1 * CLG_TS;
  float* a2 = s1 + CLG_H1 * CLG_TS;
  float* s2 = a2 + CLG_H2 * CLG_TS;
  float* te = s2 + CLG_H2 * CLG_TS;
  float* part = te + CLG_TE * CLG_TS;
  const int tid = threadIdx.x, b0 = blockIdx.x * CLG_TS;
  const int nx = 2 * a.size_z;
  // ---- inputs: the two latent slices, the sinusoid of t, the modality embeddings
  for (int idx = tid; idx < nx * CLG_TS; idx += CLG_NT) {
    const int j = idx / CLG_TS, s = idx - j * CLG_TS, b = min(b0 + s, a.B - 1);
    const int ch = j < a.size_z ? a.ch1 : a.ch2, e = j < a.size_z ? j : j - a.size_z;
    a0[j * CLG_TS + s] = a.x_pair ? a.x_pair[(long long)b * nx + j] : a.x[((long long)b * a.C + ch) * a.per_mod + e];
  }
  for (int idx = tid; idx < CLG_TE * CLG_TS; idx += CLG_NT) {
    const int j = idx / CLG_TS, s = idx - j * CLG_TS, b = min(b0 + s, a.B - 1), half = CLG_TE / 2;
    const float arg = a.t[b] * a.freq[j < half ? j : j - half];
    te[j * CLG_TS + s] = j < half ? sinf(arg) : cosf(arg);                      // [sin | cos], as unet_model's embedding
  }
  if (a.n_emb) {
    for (int idx = tid; idx < 2 * CLG_EMB * CLG_TS; idx += CLG_NT) {
      const int j = idx / CLG_TS, s = idx - j * CLG_TS;
      const int id = j < CLG_EMB ? a.id1 : a.id2;
      a0[(nx + CLG_TH + j) * CLG_TS + s] = a.emb[id * CLG_EMB + (j < CLG_EMB ? j : j - CLG_EMB)];
    }
  }
  __syncthreads();
  // ---- time MLP: temb = softplus(Wt te + bt) -> a0[nx .. nx + TH)
  clg_phase(CLG_TH, CLG_TE, te, part, [&](int k, int n) { return a.Wt_T[k * CLG_TH + n]; },
            [&](int n, const float* acc) {
#pragma unroll
              for (int s = 0; s < CLG_TS; ++s) a0[(nx + n) * CLG_TS + s] = softplus_f(acc[s] + a.bt[n]);
            });
  // ---- layer 1: D0 -> 512
  clg_phase(CLG_H1, a.D0, a0, part, [&](int k, int n) { return a.W1_T[(long long)k * CLG_H1 + n]; },
            [&](int n, const float* acc) {
              const float bb = a.b1[n];
#pragma unroll
              for (int s = 0; s < CLG_TS; ++s) { a1[n * CLG_TS + s] = softplus_f(acc[s] + bb); s1[n * CLG_TS + s] = dsoftplus_f(acc[s] + bb); }
            });
  // ---- layer 2: 512 -> 256
  clg_phase(CLG_H2, CLG_H1, a1, part, [&](int k, int n) { return a.W2_T[k * CLG_H2 + n]; },
            [&](int n, const float* acc) {
              const float bb = a.b2[n];
#pragma unroll
              for (int s = 0; s < CLG_TS; ++s) { a2[n * CLG_TS + s] = softplus_f(acc[s] + bb); s2[n * CLG_TS + s] = dsoftplus_f(acc[s] + bb); }
            });
  // ---- layer 3: 256 -> n_class (forward output, when asked for)
  if (a.out) {
    for (int idx = tid; idx < a.n_class * CLG_TS; idx += CLG_NT) {
      const int c = idx / CLG_TS, s = idx - c * CLG_TS;
      float acc = a.b3[c];
      for (int k = 0; k < CLG_H2; ++k) acc = fmaf(a.W3[c * CLG_H2 + k], a2[k * CLG_TS + s], acc);
      if (b0 + s < a.B) a.out[(long long)(b0 + s) * a.n_class + c] = acc;
    }
  }
  if (!a.score && !a.grad_out) return;
  __syncthreads();                                               // the forward output has read a2; it becomes g2 below
  // ---- backward of cl_out.mean(): d out[b][c] = 1 / (B * n_class) for every (b, c)
  //      g2 = (W3^T 1) * c0 * sigmoid(pre2)            -> overwrites a2
  const float c0 = 1.0f / ((float)a.B * (float)a.n_class);
  for (int n = tid; n < CLG_H2; n += CLG_NT) {
    float w = 0.f;
    for (int c = 0; c < a.n_class; ++c) w += a.W3[c * CLG_H2 + n];
    w *= c0;
#pragma unroll
    for (int s = 0; s < CLG_TS; ++s) a2[n * CLG_TS + s] = w * s2[n * CLG_TS + s];
  }
  __syncthreads();
  //      g1 = (W2^T g2) * sigmoid(pre1)                -> overwrites a1
  clg_phase(CLG_H1, CLG_H2, a2, part, [&](int n, int k) { return a.W2[n * CLG_H1 + k]; },
            [&](int k, const float* acc) {
#pragma unroll
              for (int s = 0; s < CLG_TS; ++s) a1[k * CLG_TS + s] = acc[s] * s1[k * CLG_TS + s];
            });
  //      gx = W1[:, :2 size_z]^T g1 ; score[:, m] -= cl_s * gx
  clg_phase(nx, CLG_H1, a1, part, [&](int k, int j) { return a.W1[(long long)k * a.D0 + j]; },
            [&](int j, const float* acc) {
              const int half = j < a.size_z ? 0 : 1, e = j - half * a.size_z, ch = half ? a.ch2 : a.ch1;
#pragma unroll
              for (int s = 0; s < CLG_TS; ++s) {
                const int b = b0 + s;
                if (b >= a.B) continue;
                if (a.grad_out) a.grad_out[(long long)b * nx + j] = acc[s];
                if (a.score && ((a.upd_mask >> half) & 1)) {
                  float* p = a.score + ((long long)b * a.C + ch) * a.per_mod + e;
                  *p = *p - a.cl_s * acc[s];
                }
              }
            });
}

}  // namespace

size_t clg_lds_bytes(int D0) { return sizeof(float) * ((size_t)CLG_TS * (size_t)(D0 + 2 * CLG_H1 + 2 * CLG_H2 + CLG_TE) + CLG_PART); }

// once per device, outside any stream capture (the fused guided loop calls it before it captures its step)
void clg_prepare() { ensure_dynamic_lds(reinterpret_cast<const void*>(clg_kernel), 160 * 1024); }

const char* launch_clg(const ClgArgs& a, hipStream_t st) {
  const size_t lds = clg_lds_bytes(a.D0);
  if (lds > 160 * 1024) return "guidance: the input width does not fit the LDS tile";
  ensure_dynamic_lds(reinterpret_cast<const void*>(clg_kernel), 160 * 1024);
  hipLaunchKernelGGL(clg_kernel, dim3((a.B + CLG_TS - 1) / CLG_TS), dim3(CLG_NT), lds, st, a);
  return nullptr;
}

// state_dict entries of ClwithTime2 (variant 2) / ClwithTime3 (variant 3), in the order sbm_model_create expects
std::vector<TensorSpec> clg_tensor_specs(int variant, int n_mod, int size_z, int n_class) {
  const int D0 = n_mod * size_z + CLG_TH + (variant == 3 ? 2 * CLG_EMB : 0);
  std::vector<TensorSpec> v;
  v.push_back({"time_mlp.0.weight", {CLG_TH, CLG_TE}}); v.push_back({"time_mlp.0.bias", {CLG_TH}});
  if (variant == 3) v.push_back({"mod_emb.weight", {10, CLG_EMB}});
  v.push_back({"layers.0.weight", {CLG_H1, D0}}); v.push_back({"layers.0.bias", {CLG_H1}});
  v.push_back({"layers.2.weight", {CLG_H2, CLG_H1}}); v.push_back({"layers.2.bias", {CLG_H2}});
  v.push_back({"layers.4.weight", {n_class, CLG_H2}}); v.push_back({"layers.4.bias", {n_class}});
  return v;
}

// Pack into one float array; offsets (floats) returned through `off`: Wt_T, bt, emb, W1, W1_T, b1, W2, W2_T, b2, W3, b3, freq
bool clg_pack(int variant, int n_mod, int size_z, int n_class, const float* const* t, std::vector<float>* out, long long* off, std::string* err) {
  if (n_mod != 2) { *err = "guidance MLP: n_mod must be 2 (a pair of latent slices)"; return false; }
  if (variant != 2 && variant != 3) { *err = "guidance MLP: variant must be 2 (ClwithTime2) or 3 (ClwithTime3)"; return false; }
  if (size_z <= 0 || size_z % 4 || n_class <= 0 || n_class > 64) { *err = "guidance MLP: bad size_z / n_class"; return false; }
  const int D0 = n_mod * size_z + CLG_TH + (variant == 3 ? 2 * CLG_EMB : 0);
  int k = 0;
  const float* Wt = t[k++]; const float* bt = t[k++];
  const float* emb = variant == 3 ? t[k++] : nullptr;
  const float* W1 = t[k++]; const float* b1 = t[k++]; const float* W2 = t[k++]; const float* b2 = t[k++]; const float* W3 = t[k++]; const float* b3 = t[k++];
  auto push = [&](const float* p, size_t n) { long long o = (long long)out->size(); out->insert(out->end(), p, p + n); while (out->size() % 4) out->push_back(0.f); return o; };
  auto push_T = [&](const float* p, int rows, int cols) {          // [rows][cols] -> [cols][rows]
    std::vector<float> tr((size_t)rows * cols);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) tr[(size_t)c * rows + r] = p[(size_t)r * cols + c];
    return push(tr.data(), tr.size());
  };
  off[0] = push_T(Wt, CLG_TH, CLG_TE); off[1] = push(bt, CLG_TH);
  off[2] = emb ? push(emb, 10 * CLG_EMB) : -1;
  off[3] = push(W1, (size_t)CLG_H1 * D0); off[4] = push_T(W1, CLG_H1, D0); off[5] = push(b1, CLG_H1);
  off[6] = push(W2, (size_t)CLG_H2 * CLG_H1); off[7] = push_T(W2, CLG_H2, CLG_H1); off[8] = push(b2, CLG_H2);
  off[9] = push(W3, (size_t)n_class * CLG_H2); off[10] = push(b3, n_class);
  std::vector<float> fr(CLG_TE / 2);                             // unet_model.py:42-44 frequencies for dim = 64
  const float kk = (float)(-(std::log(10000.0) / (CLG_TE / 2 - 1)));
  for (int j = 0; j < CLG_TE / 2; ++j) { const float arg = (float)j * kk; fr[j] = (float)std::exp((double)arg); }
  off[11] = push(fr.data(), fr.size());
  off[12] = D0;
  return true;
}

}  // namespace sbm
