// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the plan ops of plan.h, and their launcher.
//
// Dominant kernel: seg_gemm<BM,BN> -- an fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM convolution
// over the PBC layout: one workgroup computes out[p][m0:m0+BM][n0:n0+BN] by walking the VALID taps
// of output pixel p (zero-padding taps are skipped) and, per tap, the input channels in chunks of 32.
// A (activations) and B (weights) chunks are row-major with K contiguous, staged global -> VGPR ->
// LDS with a one-chunk register prefetch, GroupNorm(1,C) applied on the fly to A while it passes
// through registers, bias / residual / activation / GroupNorm partial statistics fused in the epilogue.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "plan.h"
#include "device.h"

namespace sbm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DEVINL __device__ __forceinline__

DEVINL float act_apply(int act, float v) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_LRELU02: return v >= 0.f ? v : 0.2f * v;
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------------
// seg_gemm
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A;      // source 1 base (already offset to the view origin)
  const float* W;      // [tap][Npad][Ca]
  const float* A2;     // source 2 base or null
  const float* W2;     // [Npad][C2]
  const float* bias;   // [Npad] or null
  const float* R;      // residual base or null
  float* out;          // PBC view origin, or NCHW base
  const float2* gn_part; const float* gn_gamma; const float* gn_beta;
  float2* stat_out;
  long long w_pix_stride, bias_pix_stride;
  int B, Hout, Wout;
  int lda, Ca, Ha, Wa, up, KH, KW, stride, pad, transposed, a_bcast;
  int gn_nparts; float gn_inv_count;
  int lda2, C2, W2w, up2;
  int ldr, Wr;
  int act, ldo, nstore, Npad, out_nchw, N, crop_y, crop_x, Hn, Wn;
  int mtiles, ntiles;
};

constexpr int BK = 32;        // K chunk
constexpr int LDS_LD = 36;    // padded LDS row stride (floats): conflict-free ds_read_b128 / ds_write_b128

template <int BM, int BN>
__global__ __launch_bounds__(256) void seg_gemm(const GemmArgs g) {
  constexpr int WM = BM / 64, WN = BN / 64;      // 32x32 MFMA tiles per wave in M and N (waves are 2 x 2)
  constexpr int RA = BM / 32, RB = BN / 32;      // float4 staging loads per thread for A and B
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
  __shared__ int s_tap[64], s_src[64];
  __shared__ int s_nvalid;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  int bid = blockIdx.x;
  const int nt = bid % g.ntiles; bid /= g.ntiles;
  const int mt = bid % g.mtiles; const int p = bid / g.mtiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int oy = p / g.Wout, ox = p - oy * g.Wout;

  // ---- valid taps of this output pixel (wave 0 compacts them with a ballot) ----
  if (wave == 0) {
    const int ntaps = g.KH * g.KW;
    int q = -1;
    if (lane < ntaps) {
      const int ky = lane / g.KW, kx = lane - ky * g.KW;
      q = g.a_bcast ? 0 : tap_source_pixel(oy, ox, ky, kx, g.stride, g.pad, g.transposed, g.up, g.Ha, g.Wa);
    }
    const unsigned long long mask = __ballot(q >= 0);
    if (q >= 0) {
      const int pos = __popcll(mask & ((1ull << lane) - 1ull));
      s_tap[pos] = lane; s_src[pos] = q;
    }
    if (lane == 0) s_nvalid = __popcll(mask);
  }

  // ---- staging role of this thread: rows sr + 32*j, k offset skq ----
  const int sr = tid >> 3, skq = (tid & 7) * 4;

  // ---- GroupNorm(1,C) statistics of my A rows from the producer's partial sums ----
  float gmean[RA], grstd[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) { gmean[j] = 0.f; grstd[j] = 1.f; }
  if (g.gn_part) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int b = m0 + sr + 32 * j;
      double s = 0.0, q = 0.0;
      if (b < g.B)
        for (int i = (tid & 7); i < g.gn_nparts; i += 8) { const float2 v = g.gn_part[(long long)i * g.B + b]; s += v.x; q += v.y; }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      const double mean = s * (double)g.gn_inv_count;
      double var = q * (double)g.gn_inv_count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      gmean[j] = (float)mean; grstd[j] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  __syncthreads();

  const int nvalid = s_nvalid;
  const int cpt1 = (g.Ca + BK - 1) / BK;
  const int n1 = nvalid * cpt1;
  const int total = n1 + (g.A2 ? (g.C2 + BK - 1) / BK : 0);
  const int p2 = (oy / g.up2) * g.W2w + (ox / g.up2);
  const float* Wp = g.W + (long long)p * g.w_pix_stride;

  float4 ra[RA], rb[RB];
  auto load_chunk = [&](int i) {
    const float* abase; const float* wbase; int lda, ldw, klen; bool gn;
    if (i < n1) {
      const int vi = i / cpt1, kc = (i - vi * cpt1) * BK;
      abase = g.A + (long long)s_src[vi] * g.B * g.lda + kc;
      wbase = Wp + (long long)s_tap[vi] * g.Npad * g.Ca + kc;
      lda = g.lda; ldw = g.Ca; klen = g.Ca - kc; gn = g.gn_part != nullptr;
      if (gn && skq < klen) {
        const float4 ga = *reinterpret_cast<const float4*>(g.gn_gamma + kc + skq);
        const float4 be = *reinterpret_cast<const float4*>(g.gn_beta + kc + skq);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          const int b = m0 + sr + 32 * j;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (b < g.B) {
            v = *reinterpret_cast<const float4*>(abase + (long long)b * lda + skq);
            v.x = ((v.x - gmean[j]) * grstd[j]) * ga.x + be.x;
            v.y = ((v.y - gmean[j]) * grstd[j]) * ga.y + be.y;
            v.z = ((v.z - gmean[j]) * grstd[j]) * ga.z + be.z;
            v.w = ((v.w - gmean[j]) * grstd[j]) * ga.w + be.w;
          }
          ra[j] = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          const int b = m0 + sr + 32 * j;
          ra[j] = (b < g.B && skq < klen) ? *reinterpret_cast<const float4*>(abase + (long long)b * lda + skq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
      const int kc = (i - n1) * BK;
      abase = g.A2 + (long long)p2 * g.B * g.lda2 + kc;
      wbase = g.W2 + kc;
      lda = g.lda2; ldw = g.C2; klen = g.C2 - kc;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int b = m0 + sr + 32 * j;
        ra[j] = (b < g.B && skq < klen) ? *reinterpret_cast<const float4*>(abase + (long long)b * lda + skq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int n = n0 + sr + 32 * j;
      rb[j] = (n < g.Npad && skq < klen) ? *reinterpret_cast<const float4*>(wbase + (long long)n * ldw + skq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf) {
    float* la = lds + buf * (BM + BN) * LDS_LD;
    float* lb = la + BM * LDS_LD;
#pragma unroll
    for (int j = 0; j < RA; ++j) *reinterpret_cast<float4*>(la + (sr + 32 * j) * LDS_LD + skq) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j) *reinterpret_cast<float4*>(lb + (sr + 32 * j) * LDS_LD + skq) = rb[j];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  if (total > 0) { load_chunk(0); store_chunk(0); }
  __syncthreads();
  for (int i = 0; i < total; ++i) {
    const int buf = i & 1;
    if (i + 1 < total) load_chunk(i + 1);
    const float* la = lds + buf * (BM + BN) * LDS_LD + (wave_m * 32 * WM + fi) * LDS_LD + fh * 4;
    const float* lb = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wave_n * 32 * WN + fi) * LDS_LD + fh * 4;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 fa[WM], fb[WN];
#pragma unroll
      for (int a = 0; a < WM; ++a) fa[a] = *reinterpret_cast<const float4*>(la + a * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int b = 0; b < WN; ++b) fb[b] = *reinterpret_cast<const float4*>(lb + b * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
    if (i + 1 < total) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: out = act(acc + bias + r); optional per-row partial (sum, sumsq) ----
  const float* biasp = g.bias ? g.bias + (long long)p * g.bias_pix_stride : nullptr;
  float* srow = lds;                       // [2 (wave_n)][BM][2] floats, reused after the main loop
  const bool want_stats = g.stat_out != nullptr;
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    float rs[16], rq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rs[r] = 0.f; rq[r] = 0.f; }
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int n = n0 + wave_n * 32 * WN + b * 32 + fi;
      const bool ncol = n < g.nstore;
      const float bv = (biasp && n < g.Npad) ? biasp[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wave_m * 32 * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int bb = m0 + ml;
        float v = acc[a][b][r] + bv;
        if (ncol && bb < g.B) {
          if (g.R) v += g.R[((long long)p2 * g.B + bb) * g.ldr + n];
          v = act_apply(g.act, v);
          if (g.out_nchw) {
            const int y = oy - g.crop_y, x = ox - g.crop_x;
            if (y >= 0 && y < g.Hn && x >= 0 && x < g.Wn) g.out[(((long long)bb * g.N + n) * g.Hn + y) * g.Wn + x] = v;
          } else {
            g.out[((long long)p * g.B + bb) * g.ldo + n] = v;
          }
          rs[r] += v; rq[r] += v * v;
        }
      }
    }
    if (want_stats) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = rs[r], q = rq[r];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if (fi == 0) {
          const int ml = wave_m * 32 * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          srow[(wave_n * BM + ml) * 2 + 0] = s; srow[(wave_n * BM + ml) * 2 + 1] = q;
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BM && m0 + tid < g.B) {
      const float s = srow[tid * 2] + srow[(BM + tid) * 2], q = srow[tid * 2 + 1] + srow[(BM + tid) * 2 + 1];
      g.stat_out[((long long)p * g.ntiles + nt) * g.B + m0 + tid] = make_float2(s, q);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + time condition, GroupNorm partials per (pixel, sample)
// ------------------------------------------------------------------------------------------------
struct DwArgs {
  const float* in; float* out; const float* w; const float* bias; const float* cond; float2* stat;
  int B, H, W, C, ldi, ldo, cond_ld, lanes, spb;   // lanes = threads per sample, spb = samples per block
};

__global__ __launch_bounds__(256) void dwconv7_kernel(const DwArgs d) {
  __shared__ float2 red[256];
  const int tid = threadIdx.x;
  const int p = blockIdx.x, oy = p / d.W, ox = p - oy * d.W;
  const int sl = tid / d.lanes, cl = tid - sl * d.lanes;
  const int b = blockIdx.y * d.spb + sl;
  const bool active = sl < d.spb && b < d.B;
  const int ky0 = max(0, 3 - oy), ky1 = min(7, d.H + 3 - oy), kx0 = max(0, 3 - ox), kx1 = min(7, d.W + 3 - ox);
  float s = 0.f, q = 0.f;
  if (active) {
    for (int c4 = cl; c4 * 4 < d.C; c4 += d.lanes) {
      const int c = c4 * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ky = ky0; ky < ky1; ++ky)
        for (int kx = kx0; kx < kx1; ++kx) {
          const int qp = (oy - 3 + ky) * d.W + (ox - 3 + kx);
          const float4 x = *reinterpret_cast<const float4*>(d.in + ((long long)qp * d.B + b) * d.ldi + c);
          const float4 w = *reinterpret_cast<const float4*>(d.w + (ky * 7 + kx) * d.C + c);
          acc.x = fmaf(x.x, w.x, acc.x); acc.y = fmaf(x.y, w.y, acc.y); acc.z = fmaf(x.z, w.z, acc.z); acc.w = fmaf(x.w, w.w, acc.w);
        }
      const float4 bi = *reinterpret_cast<const float4*>(d.bias + c);
      acc.x += bi.x; acc.y += bi.y; acc.z += bi.z; acc.w += bi.w;
      if (d.cond) {
        const float4 cv = *reinterpret_cast<const float4*>(d.cond + (long long)b * d.cond_ld + c);
        acc.x += cv.x; acc.y += cv.y; acc.z += cv.z; acc.w += cv.w;
      }
      *reinterpret_cast<float4*>(d.out + ((long long)p * d.B + b) * d.ldo + c) = acc;
      s += acc.x + acc.y + acc.z + acc.w;
      q += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    }
  }
  if (d.stat) {
    red[tid] = make_float2(s, q);
    __syncthreads();
    if (active && cl == 0) {
      float ss = 0.f, qq = 0.f;
      for (int i = 0; i < d.lanes; ++i) { ss += red[tid + i].x; qq += red[tid + i].y; }
      d.stat[(long long)p * d.B + b] = make_float2(ss, qq);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LinearAttention core (unet_model.py:169-175): one wave per (sample, head); lane = head dim d,
// the two 32-lane halves take alternate pixels.
// ------------------------------------------------------------------------------------------------
struct AttnArgs { const float* qkv; float* out; int B, n, ldq, ldo, heads; float scale; };

__global__ __launch_bounds__(64) void linattn_kernel(const AttnArgs a) {
  __shared__ float ctx[32][33];
  __shared__ float qs[2][32];
  const int lane = threadIdx.x, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  const float* base = a.qkv + (long long)b * a.ldq + h * 32;
  const long long pstride = (long long)a.B * a.ldq;
  // pass 1: k softmax over n -> max and sum per d
  float kmax = -INFINITY;
  for (int i = half; i < a.n; i += 2) kmax = fmaxf(kmax, base[i * pstride + hd + d]);
  kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
  float ksum = 0.f;
  for (int i = half; i < a.n; i += 2) ksum += expf(base[i * pstride + hd + d] - kmax);
  ksum += __shfl_xor(ksum, 32);
  const float kinv = 1.0f / ksum;
  // pass 2: ctx[d][e] = sum_n softmax(k)[d][n] * v[e][n]; this lane accumulates e in [16*half, 16*half+16)... over ALL n
  float c[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  for (int i = 0; i < a.n; ++i) {
    const float kv = expf(base[i * pstride + hd + d] - kmax) * kinv;
    const float* vrow = base + i * pstride + 2 * hd + half * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = fmaf(kv, vrow[e], c[e]);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) ctx[d][half * 16 + e] = c[e];
  __syncthreads();
  // lane now owns output channel e = d: column e of ctx
  float col[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) col[dd] = ctx[dd][d];
  // pass 3: out[e][n] = sum_d ctx[d][e] * softmax_d(q)[d][n] * scale
  for (int i0 = 0; i0 < a.n; i0 += 2) {
    const int i = i0 + half;
    const bool ok = i < a.n;
    const float qv = ok ? base[i * pstride + d] : 0.f;
    float m = qv;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e = expf(qv - m);
    float s = e;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    qs[half][d] = e / s * a.scale;
    __syncthreads();
    float o = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) o = fmaf(col[dd], qs[half][dd], o);
    if (ok) a.out[((long long)i * a.B + b) * a.ldo + h * 32 + d] = o;
    __syncthreads();
  }
}

// softmax attention core (unet_model.py:141-148), n <= 64: one wave per (sample, head); lane = query i
__global__ __launch_bounds__(64) void attn_kernel(const AttnArgs a) {
  __shared__ float sk[32][65], sv[32][65];
  const int lane = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  const float* base = a.qkv + (long long)b * a.ldq + h * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int idx = lane; idx < 32 * a.n; idx += 64) {
    const int j = idx >> 5, dd = idx & 31;
    sk[dd][j] = base[j * pstride + hd + dd]; sv[dd][j] = base[j * pstride + 2 * hd + dd];
  }
  __syncthreads();
  if (lane < a.n) {
    float q[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) q[dd] = base[lane * pstride + dd] * a.scale;
    float m = -INFINITY;
    for (int j = 0; j < a.n; ++j) { float t = 0.f; for (int dd = 0; dd < 32; ++dd) t = fmaf(q[dd], sk[dd][j], t); m = fmaxf(m, t); }
    float s = 0.f, o[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) o[dd] = 0.f;
    for (int j = 0; j < a.n; ++j) {
      float t = 0.f; for (int dd = 0; dd < 32; ++dd) t = fmaf(q[dd], sk[dd][j], t);
      const float e = expf(t - m); s += e;
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) o[dd] = fmaf(e, sv[dd][j], o[dd]);
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) a.out[((long long)lane * a.B + b) * a.ldo + h * 32 + dd] = o[dd] * inv;
  }
}

// ------------------------------------------------------------------------------------------------
// y = GroupNorm(1,C)(u) * gamma + beta + x   -- one workgroup per sample
// ------------------------------------------------------------------------------------------------
struct GnResArgs { const float* u; const float* x; float* out; const float2* part; const float* gamma; const float* beta; int B, P, C, ldu, ldx, ldo, nparts; float inv_count; };

__global__ __launch_bounds__(256) void gn_res_kernel(const GnResArgs g) {
  __shared__ double rs[256], rq[256];
  __shared__ float s_mean, s_rstd;
  const int tid = threadIdx.x, b = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int i = tid; i < g.nparts; i += 256) { const float2 v = g.part[(long long)i * g.B + b]; s += v.x; q += v.y; }
  rs[tid] = s; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; } __syncthreads(); }
  if (tid == 0) {
    const double mean = rs[0] * (double)g.inv_count; double var = rq[0] * (double)g.inv_count - mean * mean; var = var < 0.0 ? 0.0 : var;
    s_mean = (float)mean; s_rstd = (float)(1.0 / sqrt(var + 1e-5));
  }
  __syncthreads();
  const float mean = s_mean, rstd = s_rstd;
  const int c4n = g.C / 4;
  for (int idx = tid; idx < g.P * c4n; idx += 256) {
    const int p = idx / c4n, c = (idx - p * c4n) * 4;
    const long long row = (long long)p * g.B + b;
    const float4 u = *reinterpret_cast<const float4*>(g.u + row * g.ldu + c);
    const float4 x = *reinterpret_cast<const float4*>(g.x + row * g.ldx + c);
    const float4 ga = *reinterpret_cast<const float4*>(g.gamma + c);
    const float4 be = *reinterpret_cast<const float4*>(g.beta + c);
    float4 y;
    y.x = ((u.x - mean) * rstd) * ga.x + be.x + x.x; y.y = ((u.y - mean) * rstd) * ga.y + be.y + x.y;
    y.z = ((u.z - mean) * rstd) * ga.z + be.z + x.z; y.w = ((u.w - mean) * rstd) * ga.w + be.w + x.w;
    *reinterpret_cast<float4*>(g.out + row * g.ldo + c) = y;
  }
}

// ------------------------------------------------------------------------------------------------
// small boundary kernels
// ------------------------------------------------------------------------------------------------
__global__ void sinus_kernel(const float* t, const float* freq, float* out, int B, int dim, int ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * ld) return;
  const int b = idx / ld, j = idx - b * ld, half = dim / 2;
  float v = 0.f;
  if (j < half) v = sinf(t[b] * freq[j]);
  else if (j < 2 * half) v = cosf(t[b] * freq[j - half]);
  out[idx] = v;
}

__global__ void nchw_to_pbc_kernel(const float* in, float* out, int B, int C, int Hn, int Wn, int pad_y, int pad_x, int Ho, int Wo, int ld, int Cv) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ho * Wo * B * Cv;
  if (idx >= total) return;
  const int c = idx % Cv; long long r = idx / Cv; const int b = r % B; const int p = r / B;
  const int y = p / Wo - pad_y, x = p % Wo - pad_x;
  float v = 0.f;
  if (c < C && y >= 0 && y < Hn && x >= 0 && x < Wn) v = in[(((long long)b * C + c) * Hn + y) * Wn + x];
  out[((long long)p * B + b) * ld + c] = v;
}

struct BilArgs { const float* in; float* out; int B, Hi, Wi, Ho, Wo, C, ldi, ldo, scale; };
__global__ void bilinear_kernel(const BilArgs o) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = o.C / 4;
  const long long total = (long long)o.Ho * o.Wo * o.B * c4n;
  if (idx >= total) return;
  const int c = (idx % c4n) * 4; long long r = idx / c4n; const int b = r % o.B; const int p = r / o.B;
  const int oy = p / o.Wo, ox = p - oy * o.Wo;
  const float sy = fmaxf((oy + 0.5f) / o.scale - 0.5f, 0.f), sx = fmaxf((ox + 0.5f) / o.scale - 0.5f, 0.f);
  const int y0 = min((int)sy, o.Hi - 1), x0 = min((int)sx, o.Wi - 1), y1 = min(y0 + 1, o.Hi - 1), x1 = min(x0 + 1, o.Wi - 1);
  const float ly = sy - y0, lx = sx - x0;
  auto at = [&](int y, int x) { return *reinterpret_cast<const float4*>(o.in + ((long long)(y * o.Wi + x) * o.B + b) * o.ldi + c); };
  const float4 a00 = at(y0, x0), a01 = at(y0, x1), a10 = at(y1, x0), a11 = at(y1, x1);
  auto mix = [&](float p00, float p01, float p10, float p11) { const float top = p00 * (1.f - lx) + p01 * lx, bot = p10 * (1.f - lx) + p11 * lx; return top * (1.f - ly) + bot * ly; };
  float4 v; v.x = mix(a00.x, a01.x, a10.x, a11.x); v.y = mix(a00.y, a01.y, a10.y, a11.y); v.z = mix(a00.z, a01.z, a10.z, a11.z); v.w = mix(a00.w, a01.w, a10.w, a11.w);
  *reinterpret_cast<float4*>(o.out + ((long long)p * o.B + b) * o.ldo + c) = v;
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
static const char* launch_gemm(const GemmOp& o, const float* prm, float* ws, const float* ext_in, float* ext_out, hipStream_t st) {
  GemmArgs g{};
  g.A = (o.a_ext ? ext_in : ws) + o.a.off;
  g.W = prm + o.w_off;
  g.A2 = o.has2 ? ws + o.a2.off : nullptr; g.W2 = o.has2 ? prm + o.w2_off : nullptr;
  g.bias = o.bias_off >= 0 ? prm + o.bias_off : nullptr;
  g.R = o.has_res ? ws + o.r.off : nullptr;
  g.out = o.out_nchw ? ext_out : ws + o.out.off;
  g.gn_part = o.gn ? reinterpret_cast<const float2*>(ws + o.gn_part_off) : nullptr;
  g.gn_gamma = o.gn ? prm + o.gn_gamma_off : nullptr; g.gn_beta = o.gn ? prm + o.gn_beta_off : nullptr;
  g.stat_out = o.stat_off >= 0 ? reinterpret_cast<float2*>(ws + o.stat_off) : nullptr;
  g.w_pix_stride = o.w_pix_stride; g.bias_pix_stride = o.bias_pix_stride;
  g.B = o.B; g.Hout = o.Hout; g.Wout = o.Wout;
  g.lda = o.a.ld; g.Ca = o.a.C; g.Ha = o.a.H; g.Wa = o.a.W; g.up = o.up; g.KH = o.KH; g.KW = o.KW; g.stride = o.stride; g.pad = o.pad;
  g.transposed = o.transposed; g.a_bcast = o.a_bcast;
  g.gn_nparts = o.gn_nparts; g.gn_inv_count = o.gn_inv_count;
  g.lda2 = o.a2.ld; g.C2 = o.a2.C; g.up2 = o.up2; g.W2w = o.has2 ? o.a2.W : (o.has_res ? o.r.W : 1);
  g.ldr = o.r.ld; g.Wr = o.r.W;
  g.act = o.act; g.ldo = o.out.ld; g.nstore = o.out_nchw ? o.N : o.out.C; g.Npad = o.Npad; g.out_nchw = o.out_nchw; g.N = o.N;
  g.crop_y = o.crop_y; g.crop_x = o.crop_x; g.Hn = o.Hn; g.Wn = o.Wn;
  g.mtiles = ceil_div(o.B, o.BM); g.ntiles = ceil_div(g.nstore, o.BN);
  if (o.KH * o.KW > 64) return "seg_gemm: more than 64 taps";
  if ((o.a.C & 3) || (o.a.ld & 3) || (o.has2 && ((o.a2.C & 3) || (o.a2.ld & 3)))) return "seg_gemm: channel counts / pitches must be multiples of 4";
  const long long blocks = (long long)o.Hout * o.Wout * g.mtiles * g.ntiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return "seg_gemm: bad grid";
  dim3 grid((unsigned)blocks), block(256);
  if (o.BM == 64 && o.BN == 64) hipLaunchKernelGGL((seg_gemm<64, 64>), grid, block, 0, st, g);
  else if (o.BM == 128 && o.BN == 64) hipLaunchKernelGGL((seg_gemm<128, 64>), grid, block, 0, st, g);
  else if (o.BM == 128 && o.BN == 128) hipLaunchKernelGGL((seg_gemm<128, 128>), grid, block, 0, st, g);
  else return "seg_gemm: unsupported tile";
  return nullptr;
}

const char* launch_op(const Plan& plan, const Op& op, const float* prm, float* ws, const float* ext_in, float* ext_out, hipStream_t st) {
  switch (op.kind) {
    case OP_GEMM: return launch_gemm(op.g, prm, ws, ext_in, ext_out, st);
    case OP_DWCONV7: {
      const DwOp& o = op.dw; DwArgs d{};
      d.in = ws + o.in.off; d.out = ws + o.out.off; d.w = prm + o.w_off; d.bias = prm + o.bias_off;
      d.cond = o.has_cond ? ws + o.cond_off : nullptr; d.stat = o.stat_off >= 0 ? reinterpret_cast<float2*>(ws + o.stat_off) : nullptr;
      d.B = o.B; d.H = o.in.H; d.W = o.in.W; d.C = o.in.C; d.ldi = o.in.ld; d.ldo = o.out.ld; d.cond_ld = o.cond_ld;
      d.lanes = o.in.C / 4 < 256 ? o.in.C / 4 : 256; d.spb = 256 / d.lanes;
      dim3 grid(o.in.H * o.in.W, ceil_div(o.B, d.spb));
      hipLaunchKernelGGL(dwconv7_kernel, grid, dim3(256), 0, st, d);
    } break;
    case OP_LINATTN: case OP_ATTN: {
      const AttnOp& o = op.at; AttnArgs a{};
      a.qkv = ws + o.qkv.off; a.out = ws + o.out.off; a.B = o.B; a.n = o.qkv.H * o.qkv.W; a.ldq = o.qkv.ld; a.ldo = o.out.ld; a.heads = o.heads; a.scale = o.scale;
      if (o.dh != 32) return "attention: dim_head must be 32";
      if (op.kind == OP_ATTN && a.n > 64) return "softmax attention: more than 64 positions";
      if (op.kind == OP_LINATTN) hipLaunchKernelGGL(linattn_kernel, dim3(o.B * o.heads), dim3(64), 0, st, a);
      else hipLaunchKernelGGL(attn_kernel, dim3(o.B * o.heads), dim3(64), 0, st, a);
    } break;
    case OP_GN_RES: {
      const GnResOp& o = op.gr; GnResArgs g{};
      g.u = ws + o.u.off; g.x = ws + o.x.off; g.out = ws + o.out.off; g.part = reinterpret_cast<const float2*>(ws + o.part_off);
      g.gamma = prm + o.gamma_off; g.beta = prm + o.beta_off; g.B = o.B; g.P = o.u.H * o.u.W; g.C = o.u.C;
      g.ldu = o.u.ld; g.ldx = o.x.ld; g.ldo = o.out.ld; g.nparts = o.nparts; g.inv_count = o.inv_count;
      hipLaunchKernelGGL(gn_res_kernel, dim3(o.B), dim3(256), 0, st, g);
    } break;
    case OP_SINUS: {
      const SinusOp& o = op.si; const int n = o.B * o.ld;
      hipLaunchKernelGGL(sinus_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, ws + plan.time_off, prm + o.freq_off, ws + o.out_off, o.B, o.dim, o.ld);
    } break;
    case OP_NCHW_TO_PBC: {
      const CvtOp& o = op.cv; const long long n = (long long)o.out.H * o.out.W * o.B * o.out.C;
      hipLaunchKernelGGL(nchw_to_pbc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ext_in, ws + o.out.off, o.B, o.C, o.Hn, o.Wn, o.pad_y, o.pad_x, o.out.H, o.out.W, o.out.ld, o.out.C);
    } break;
    case OP_BILINEAR: {
      const BilinearOp& o = op.bl; BilArgs a{};
      a.in = ws + o.in.off; a.out = ws + o.out.off; a.B = o.B; a.Hi = o.in.H; a.Wi = o.in.W; a.Ho = o.out.H; a.Wo = o.out.W; a.C = o.in.C; a.ldi = o.in.ld; a.ldo = o.out.ld; a.scale = o.scale;
      const long long n = (long long)a.Ho * a.Wo * a.B * (a.C / 4);
      hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    } break;
    default: return "unknown op";
  }
  return nullptr;
}

const char* run_plan(const Plan& plan, const float* prm, float* ws, const float* ext_in, const float* t_dev, float* ext_out, hipStream_t st) {
  if (plan.time_off >= 0 && t_dev && t_dev != ws + plan.time_off)
    if (hipMemcpyAsync(ws + plan.time_off, t_dev, sizeof(float) * plan.B, hipMemcpyDeviceToDevice, st) != hipSuccess) return "time copy failed";
  for (const Op& op : plan.ops) {
    const char* e = launch_op(plan, op, prm, ws, ext_in, ext_out, st);
    if (e) return e;
  }
  if (hipGetLastError() != hipSuccess) return "kernel launch failed";
  return nullptr;
}

}  // namespace sbm
