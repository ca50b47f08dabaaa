// tests/interp.cpp -- TEST-ONLY scalar host interpreter of a Plan.
//
// Built with g++ into libsbmae_planref.so (never linked into the product library).  It executes the
// op descriptors of plan.h with plain loops on host memory, in the same PBC layout and with the same
// partial-statistics plumbing as the HIP kernels, so that the plan builder / weight packer can be
// checked against the oracle on a machine without a GPU, and so that each HIP kernel can be tested
// against the descriptor semantics op by op.  It is not a fallback: nothing in the product path
// references it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.h"

namespace sbm {

static inline float act_apply(int act, float v) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + (float)std::erf((double)v * 0.70710678118654752440));
    case ACT_LRELU02: return v >= 0.f ? v : 0.2f * v;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_SIGMOID: return (float)(1.0 / (1.0 + std::exp(-(double)v)));
    case ACT_SILU: return (float)((double)v / (1.0 + std::exp(-(double)v)));
    case ACT_SOFTPLUS: return v > 20.f ? v : (float)std::log1p(std::exp((double)v));
    case ACT_LRELU01: return v >= 0.f ? v : 0.1f * v;
    default: return v;
  }
}

static void gn_mean_rstd(const float* ws, int64_t part_off, int nparts, int B, int b, float inv_count, float* mean, float* rstd) {
  double s = 0, q = 0;
  for (int i = 0; i < nparts; ++i) {
    s += ws[part_off + ((int64_t)i * B + b) * 2 + 0];
    q += ws[part_off + ((int64_t)i * B + b) * 2 + 1];
  }
  double m = s * inv_count;
  double var = q * inv_count - m * m;
  if (var < 0) var = 0;
  *mean = (float)m;
  *rstd = (float)(1.0 / std::sqrt(var + 1e-5));
}

static void run_gemm(const GemmOp& g, const float* prm, float* ws, const float* ext_in, float* ext_out) {
  // Walks the SAME host-built chunk tables the HIP kernel reads (hdr/desc), so the tables are tested on the CPU.
  const int B = g.B, P = g.Hout * g.Wout;
  const int nstore = g.out_nchw ? g.N : g.out.C;
  const int ntiles = ceil_div(nstore, g.BN);
  const float* A1 = (g.a_ext ? ext_in : ws) + g.a.off;
  const float* A2 = g.has2 ? ws + g.a2.off : nullptr;
  const int* hdr = reinterpret_cast<const int*>(prm + g.hdr_off);
  const int* desc = reinterpret_cast<const int*>(prm + g.desc_off);
  std::vector<float> mean(B, 0.f), rstd(B, 1.f);
  if (g.gn) for (int b = 0; b < B; ++b) gn_mean_rstd(ws, g.gn_part_off, g.gn_nparts, B, b, g.gn_inv_count, &mean[b], &rstd[b]);
  std::vector<float> acc(nstore), acc2(nstore);
  for (int p = 0; p < P; ++p) {
    const int oy = p / g.Wout, ox = p % g.Wout;
    const int first = hdr[4 * p], total = hdr[4 * p + 1], p2 = hdr[4 * p + 3];
    const float* bias = g.bias_off >= 0 ? prm + g.bias_off + (int64_t)p * g.bias_pix_stride : nullptr;
    for (int b = 0; b < B; ++b) {
      std::fill(acc.begin(), acc.end(), 0.f); std::fill(acc2.begin(), acc2.end(), 0.f);
      for (int c = 0; c < total; ++c) {
        const int* d = desc + 4 * (first + c);
        const bool src2 = d[3] & 1;
        const float* arow = (src2 ? A2 : A1) + d[0] + (int64_t)b * (src2 ? g.a2.ld : g.a.ld);
        const float* wbase = prm + (src2 ? g.w2_off : g.w_off) + d[1];
        const int ldw = src2 ? g.Kp2 : g.Kp;
        // the kernel multiplies the whole 32-wide chunk; beyond the true channel count the weights are zero
        const int kc = src2 ? d[1] : (d[1] % g.Kp);
        const int klen = std::min(32, (src2 ? g.a2.C : g.a.C) - kc);
        std::vector<float>& dst = src2 ? acc2 : acc;
        for (int n = 0; n < nstore; ++n) {
          const float* w = wbase + (int64_t)n * ldw;
          float s = dst[n];
          for (int k = 0; k < klen; ++k) s = std::fmaf(arow[k], w[k], s);
          dst[n] = s;
        }
      }
      // folded GroupNorm (plan.h): source 1 is scaled by rstd and corrected with the host tables S / B1
      for (int n = 0; n < nstore; ++n) {
        float v = acc[n];
        if (g.gn) v = rstd[b] * v + (prm[g.gn_b1_off + (int64_t)p * g.Npad + n] - (mean[b] * rstd[b]) * prm[g.gn_s_off + (int64_t)p * g.Npad + n]);
        acc[n] = v + acc2[n];
      }
      std::vector<double> tsum(ntiles, 0.0), tsq(ntiles, 0.0);
      for (int n = 0; n < nstore; ++n) {
        float v = acc[n];
        if (bias) v += bias[n];
        if (g.has_res) v += ws[g.r.off + ((int64_t)p2 * B + b) * g.r.ld + n];
        v = act_apply(g.act, v);
        if (g.out_nchw) {
          const int y = oy - g.crop_y, x = ox - g.crop_x;
          if (y >= 0 && y < g.Hn && x >= 0 && x < g.Wn) ext_out[(((int64_t)b * g.N + n) * g.Hn + y) * g.Wn + x] = v;
        } else {
          ws[g.out.off + ((int64_t)p * B + b) * g.out.ld + n] = v;
        }
        tsum[n / g.BN] += v; tsq[n / g.BN] += (double)v * v;
      }
      if (g.stat_off >= 0)
        for (int t = 0; t < ntiles; ++t) {
          ws[g.stat_off + (((int64_t)p * ntiles + t) * B + b) * 2 + 0] = (float)tsum[t];
          ws[g.stat_off + (((int64_t)p * ntiles + t) * B + b) * 2 + 1] = (float)tsq[t];
        }
    }
  }
}

static void run_dw(const DwOp& d, const float* prm, float* ws) {
  const int B = d.B, H = d.in.H, W = d.in.W, C = d.in.C;
  const int slabs = d.stat_slabs > 1 ? d.stat_slabs : 1;          // > 1: one partial per (32-channel slab, sample) over all pixels
  std::vector<double> ss((size_t)slabs * B, 0.0), qs((size_t)slabs * B, 0.0);
  for (int p = 0; p < H * W; ++p) {
    const int oy = p / W, ox = p % W;
    for (int b = 0; b < B; ++b) {
      double s = 0, q = 0;
      float* o = ws + d.out.off + ((int64_t)p * B + b) * d.out.ld;
      for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int ky = 0; ky < 7; ++ky) for (int kx = 0; kx < 7; ++kx) {
          const int iy = oy - 3 + ky, ix = ox - 3 + kx;
          if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
          acc = std::fmaf(ws[d.in.off + ((int64_t)(iy * W + ix) * B + b) * d.in.ld + c], prm[d.w_off + (int64_t)(ky * 7 + kx) * C + c], acc);
        }
        acc += prm[d.bias_off + c];
        if (d.has_cond) acc += ws[d.cond_off + (int64_t)b * d.cond_ld + c];
        o[c] = acc; s += acc; q += (double)acc * acc;
        if (slabs > 1) { ss[(size_t)(c / 32) * B + b] += acc; qs[(size_t)(c / 32) * B + b] += (double)acc * acc; }
      }
      if (d.stat_off >= 0 && slabs == 1) { ws[d.stat_off + ((int64_t)p * B + b) * 2] = (float)s; ws[d.stat_off + ((int64_t)p * B + b) * 2 + 1] = (float)q; }
    }
  }
  if (d.stat_off >= 0 && slabs > 1)
    for (int i = 0; i < slabs * B; ++i) { ws[d.stat_off + (int64_t)i * 2] = (float)ss[i]; ws[d.stat_off + (int64_t)i * 2 + 1] = (float)qs[i]; }
}

static void run_attn(const AttnOp& a, bool linear, float* ws) {
  const int B = a.B, n = a.qkv.H * a.qkv.W, hd = a.heads * a.dh, dh = a.dh;
  std::vector<double> q((size_t)dh * n), k((size_t)dh * n), v((size_t)dh * n);
  for (int b = 0; b < B; ++b) for (int h = 0; h < a.heads; ++h) {
    for (int d = 0; d < dh; ++d) for (int i = 0; i < n; ++i) {
      const float* row = ws + a.qkv.off + ((int64_t)i * B + b) * a.qkv.ld;
      q[(size_t)d * n + i] = row[h * dh + d]; k[(size_t)d * n + i] = row[hd + h * dh + d]; v[(size_t)d * n + i] = row[2 * hd + h * dh + d];
    }
    if (linear) {
      // q softmax over d, k softmax over n, q *= scale; ctx[d][e] = sum_n k[d][n] v[e][n]; out[e][n] = sum_d ctx[d][e] q[d][n]
      for (int i = 0; i < n; ++i) {
        double m = -1e300, s = 0; for (int d = 0; d < dh; ++d) m = std::max(m, q[(size_t)d * n + i]);
        for (int d = 0; d < dh; ++d) { q[(size_t)d * n + i] = std::exp(q[(size_t)d * n + i] - m); s += q[(size_t)d * n + i]; }
        for (int d = 0; d < dh; ++d) q[(size_t)d * n + i] = q[(size_t)d * n + i] / s * a.scale;
      }
      for (int d = 0; d < dh; ++d) {
        double m = -1e300, s = 0; for (int i = 0; i < n; ++i) m = std::max(m, k[(size_t)d * n + i]);
        for (int i = 0; i < n; ++i) { k[(size_t)d * n + i] = std::exp(k[(size_t)d * n + i] - m); s += k[(size_t)d * n + i]; }
        for (int i = 0; i < n; ++i) k[(size_t)d * n + i] /= s;
      }
      std::vector<double> ctx((size_t)dh * dh, 0.0);
      for (int d = 0; d < dh; ++d) for (int e = 0; e < dh; ++e) { double s = 0; for (int i = 0; i < n; ++i) s += k[(size_t)d * n + i] * v[(size_t)e * n + i]; ctx[(size_t)d * dh + e] = s; }
      for (int e = 0; e < dh; ++e) for (int i = 0; i < n; ++i) {
        double s = 0; for (int d = 0; d < dh; ++d) s += ctx[(size_t)d * dh + e] * q[(size_t)d * n + i];
        ws[a.out.off + ((int64_t)i * B + b) * a.out.ld + h * dh + e] = (float)s;
      }
    } else {
      // sim[i][j] = sum_d scale*q[d][i] k[d][j]; softmax_j; out[i][d] = sum_j attn[i][j] v[d][j]
      std::vector<double> sim(n);
      for (int i = 0; i < n; ++i) {
        double m = -1e300, s = 0;
        for (int j = 0; j < n; ++j) { double t = 0; for (int d = 0; d < dh; ++d) t += q[(size_t)d * n + i] * a.scale * k[(size_t)d * n + j]; sim[j] = t; m = std::max(m, t); }
        for (int j = 0; j < n; ++j) { sim[j] = std::exp(sim[j] - m); s += sim[j]; }
        for (int d = 0; d < dh; ++d) {
          double t = 0; for (int j = 0; j < n; ++j) t += sim[j] / s * v[(size_t)d * n + j];
          ws[a.out.off + ((int64_t)i * B + b) * a.out.ld + h * dh + d] = (float)t;
        }
      }
    }
  }
}

static void run_gn_res(const GnResOp& g, const float* prm, float* ws, float* ext_out) {
  const int B = g.B, P = g.u.H * g.u.W, C = g.u.C;
  for (int b = 0; b < B; ++b) {
    float mean, rstd; gn_mean_rstd(ws, g.part_off, g.nparts, B, b, g.inv_count, &mean, &rstd);
    for (int p = 0; p < P; ++p) for (int c = 0; c < C; ++c) {
      float u = ws[g.u.off + ((int64_t)p * B + b) * g.u.ld + c];
      float y = ((u - mean) * rstd) * prm[g.gamma_off + c] + prm[g.beta_off + c];
      (g.out_ext ? ext_out : ws + g.out.off)[((int64_t)p * B + b) * g.out.ld + c] = y + ws[g.x.off + ((int64_t)p * B + b) * g.x.ld + c];
    }
  }
}

static void run_bilinear(const BilinearOp& o, float* ws) {
  const int B = o.B, Hi = o.in.H, Wi = o.in.W, Ho = o.out.H, Wo = o.out.W, C = o.in.C, s = o.scale;
  for (int oy = 0; oy < Ho; ++oy) for (int ox = 0; ox < Wo; ++ox) {
    float sy = std::max((oy + 0.5f) / s - 0.5f, 0.f), sx = std::max((ox + 0.5f) / s - 0.5f, 0.f);
    int y0 = std::min((int)sy, Hi - 1), x0 = std::min((int)sx, Wi - 1);
    int y1 = std::min(y0 + 1, Hi - 1), x1 = std::min(x0 + 1, Wi - 1);
    float ly = sy - y0, lx = sx - x0;
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) {
      auto at = [&](int y, int x) { return ws[o.in.off + ((int64_t)(y * Wi + x) * B + b) * o.in.ld + c]; };
      float top = at(y0, x0) * (1.f - lx) + at(y0, x1) * lx, bot = at(y1, x0) * (1.f - lx) + at(y1, x1) * lx;
      ws[o.out.off + ((int64_t)(oy * Wo + ox) * B + b) * o.out.ld + c] = top * (1.f - ly) + bot * ly;
    }
  }
}

void interpret(const Plan& plan, std::vector<float>& ws, const float* in, const float* t, float* out, int n_ops = 1 << 30, const float* aux = nullptr) {
  if (plan.aux_off >= 0 && aux) std::memcpy(&ws[plan.aux_off], aux, sizeof(float) * plan.B * plan.aux_C);
  const float* prm = plan.params.data();
  if (plan.time_off >= 0 && t) std::memcpy(&ws[plan.time_off], t, sizeof(float) * plan.B);
  int done = 0;
  for (const Op& op : plan.ops) {
    if (done++ >= n_ops) break;
    switch (op.kind) {
      case OP_GEMM: run_gemm(op.g, prm, ws.data(), in, out); break;
      case OP_DWCONV7: run_dw(op.dw, prm, ws.data()); break;
      case OP_LINATTN: run_attn(op.at, true, ws.data()); break;
      case OP_ATTN: run_attn(op.at, false, ws.data()); break;
      case OP_GN_RES: run_gn_res(op.gr, prm, ws.data(), out); break;
      case OP_SINUS: {
        const SinusOp& s = op.si; const int half = s.dim / 2;
        for (int b = 0; b < s.B; ++b) for (int j = 0; j < half; ++j) {
          float arg = ws[plan.time_off + b] * prm[s.freq_off + j];
          ws[s.out_off + (int64_t)b * s.ld + (s.cos_first ? half : 0) + j] = (float)std::sin((double)arg);
          ws[s.out_off + (int64_t)b * s.ld + (s.cos_first ? 0 : half) + j] = (float)std::cos((double)arg);
        }
        for (int b = 0; b < s.B; ++b) for (int j = 2 * half; j < s.ld; ++j) ws[s.out_off + (int64_t)b * s.ld + j] = 0.f;
      } break;
      case OP_NCHW_TO_PBC: {
        const CvtOp& c = op.cv; const View& o = c.out;
        if (c.rowpack) {
          for (int y0 = 0; y0 < o.H; ++y0) for (int b = 0; b < c.B; ++b) for (int k = 0; k < o.ld; ++k) {
            const int xs = k / c.Cpad, ch = k % c.Cpad, y = y0 - c.pad_y, x = xs - c.pad_x;
            float v = 0.f;
            if (xs < c.Wo && ch < c.C && y >= 0 && y < c.Hn && x >= 0 && x < c.Wn) v = in[(((int64_t)b * c.C + ch) * c.Hn + y) * c.Wn + x];
            ws[o.off + ((int64_t)y0 * c.B + b) * o.ld + k] = v;
          }
          break;
        }
        for (int p = 0; p < o.H * o.W; ++p) for (int b = 0; b < c.B; ++b) for (int ch = 0; ch < o.C; ++ch) {
          int y = p / o.W - c.pad_y, x = p % o.W - c.pad_x;
          float v = 0.f;
          if (ch < c.C && y >= 0 && y < c.Hn && x >= 0 && x < c.Wn) v = in[(((int64_t)b * c.C + ch) * c.Hn + y) * c.Wn + x];
          ws[o.off + ((int64_t)p * c.B + b) * o.ld + ch] = v;
        }
      } break;
      case OP_IM2COL: {                                  // pixel mode: init_conv's input matrix from the row-packed image (plan.h Im2colOp)
        const Im2colOp& o = op.ic; const int taps = o.KH * o.KW;
        for (int p = 0; p < o.H * o.W; ++p) for (int b = 0; b < o.B; ++b) for (int k = 0; k < o.out.ld; ++k) {
          const int c = k / taps, r = k % taps, iy = p / o.W + r / o.KW - o.pad, ix = p % o.W + r % o.KW - o.pad;
          float v = 0.f;
          if (c < o.C && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W) v = ws[o.in.off + ((int64_t)iy * o.B + b) * o.in.ld + ix * o.Cpad + c];
          ws[o.out.off + ((int64_t)p * o.B + b) * o.out.ld + k] = v;
        }
      } break;
      case OP_BILINEAR: run_bilinear(op.bl, ws.data()); break;
      case OP_AVGPOOL: {
        const BilinearOp& o = op.bl; const int k = o.scale, Ho = o.out.H, Wo = o.out.W, Wi = o.in.W;
        for (int oy = 0; oy < Ho; ++oy) for (int ox = 0; ox < Wo; ++ox) for (int b = 0; b < o.B; ++b) for (int c = 0; c < o.in.C; ++c) {
          float s = 0.f;
          for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) s += ws[o.in.off + ((int64_t)((oy * k + i) * Wi + ox * k + j) * o.B + b) * o.in.ld + c];
          ws[o.out.off + ((int64_t)(oy * Wo + ox) * o.B + b) * o.out.ld + c] = s / (float)(k * k);
        }
      } break;
      case OP_GN32: {
        const Gn32Op& o = op.gn; const int P = o.in.H * o.in.W, C = o.in.C, cg = C / o.groups;
        for (int b = 0; b < o.B; ++b) for (int g = 0; g < o.groups; ++g) {
          double s = 0, q = 0;
          for (int p = 0; p < P; ++p) for (int c = g * cg; c < (g + 1) * cg; ++c) { const double v = ws[o.in.off + ((int64_t)p * o.B + b) * o.in.ld + c]; s += v; q += v * v; }
          const double mean = s / (P * cg); double var = q / (P * cg) - mean * mean; if (var < 0) var = 0;
          const double rstd = 1.0 / std::sqrt(var + 1e-5);
          for (int p = 0; p < P; ++p) for (int c = g * cg; c < (g + 1) * cg; ++c) {
            float y = (float)(((double)ws[o.in.off + ((int64_t)p * o.B + b) * o.in.ld + c] - mean) * rstd) * prm[o.gamma_off + c] + prm[o.beta_off + c];
            if (o.silu) y = act_apply(ACT_SILU, y);
            ws[o.out.off + ((int64_t)p * o.B + b) * o.out.ld + c] = y;
          }
        }
      } break;
      case OP_ATTN_QKV: {
        const AttnQkvOp& o = op.aq; const int T = o.qkv.H * o.qkv.W, ch = o.chh;
        std::vector<double> w((size_t)T * T);
        for (int b = 0; b < o.B; ++b) for (int h = 0; h < o.heads; ++h) {
          auto Q = [&](int c, int t) { return (double)ws[o.qkv.off + ((int64_t)t * o.B + b) * o.qkv.ld + h * 3 * ch + c]; };
          auto K = [&](int c, int t) { return (double)ws[o.qkv.off + ((int64_t)t * o.B + b) * o.qkv.ld + h * 3 * ch + ch + c]; };
          auto V = [&](int c, int t) { return (double)ws[o.qkv.off + ((int64_t)t * o.B + b) * o.qkv.ld + h * 3 * ch + 2 * ch + c]; };
          for (int t = 0; t < T; ++t) {
            double m = -1e300;
            for (int u = 0; u < T; ++u) { double a = 0; for (int c = 0; c < ch; ++c) a += Q(c, t) * o.scale * K(c, u) * o.scale; w[(size_t)t * T + u] = a; m = std::max(m, a); }
            double z = 0; for (int u = 0; u < T; ++u) { w[(size_t)t * T + u] = std::exp(w[(size_t)t * T + u] - m); z += w[(size_t)t * T + u]; }
            for (int u = 0; u < T; ++u) w[(size_t)t * T + u] /= z;
          }
          for (int t = 0; t < T; ++t) for (int c = 0; c < ch; ++c) {
            double a = 0; for (int u = 0; u < T; ++u) a += w[(size_t)t * T + u] * V(c, u);
            ws[o.out.off + ((int64_t)t * o.B + b) * o.out.ld + h * ch + c] = (float)a;
          }
        }
      } break;
      default: break;
    }
  }
}

}  // namespace sbm

// ------------------------------------------------------------------------------------------------
// C test API (ctypes)
// ------------------------------------------------------------------------------------------------
using namespace sbm;

struct RefHandle { Plan plan; std::vector<float> ws; };
static thread_local std::string g_err;

static UnetCfg unet_cfg_from_ints(const int* c) {
  // [dim, channels, init_dim, out_dim, with_time_emb, convnext_mult, H, W, n_mults, mults...]
  UnetCfg u; u.dim = c[0]; u.channels = c[1]; u.init_dim = c[2]; u.out_dim = c[3]; u.with_time_emb = c[4]; u.convnext_mult = c[5];
  u.H = c[6]; u.W = c[7]; u.n_mults = c[8];
  for (int i = 0; i < u.n_mults && i < 8; ++i) u.dim_mults[i] = c[9 + i];
  return u;
}
static UnetOACfg unet_oa_cfg_from_ints(const int* c) {
  // [in_ch, model_ch, out_ch, num_res_blocks, num_heads, z_dim, use_z, H, W, n_mult, mults..., n_attn, attn_res...]
  UnetOACfg u; u.in_channels = c[0]; u.model_channels = c[1]; u.out_channels = c[2]; u.num_res_blocks = c[3]; u.num_heads = c[4];
  u.z_dim = c[5]; u.use_z = c[6]; u.H = c[7]; u.W = c[8]; u.n_mult = c[9] < 8 ? c[9] : 8;
  for (int i = 0; i < u.n_mult; ++i) u.channel_mult[i] = c[10 + i];
  u.n_attn = c[10 + u.n_mult] < 8 ? c[10 + u.n_mult] : 8;
  for (int i = 0; i < u.n_attn; ++i) u.attn_res[i] = c[11 + u.n_mult + i];
  return u;
}
static ResDecCfg resdec_cfg_from_ints(const int* c) {
  // [variant, size_in, size_z, img_ch, n_enc, n_dec, enc(4*n_enc)..., dec(4*n_dec)...]
  ResDecCfg r; r.variant = c[0]; r.size_in = c[1]; r.size_z = c[2]; r.img_ch = c[3]; r.n_enc = c[4]; r.n_dec = c[5];
  const int* e = c + 6;
  for (int i = 0; i < r.n_enc; ++i) for (int j = 0; j < 4; ++j) r.enc[i][j] = *e++;
  for (int i = 0; i < r.n_dec; ++i) for (int j = 0; j < 4; ++j) r.dec[i][j] = *e++;
  return r;
}

extern "C" {

const char* sbmref_error() { return g_err.c_str(); }

void* sbmref_build(int kind, const int* cfg, int B, int z_ld, const float* const* tensors) {
  RefHandle* h = new RefHandle();
  bool ok = false;
  // SBMREF_PIX=1: the plan as the product builds it for the sample-owned pixel kernel (marking, statistics layouts, pinned buffers), so
  // that the op-by-op comparison runs against the same descriptors
  if (kind == 0) { plan_set_pix_mode(std::getenv("SBMREF_PIX") ? 1 : 0); ok = build_unet_plan(unet_cfg_from_ints(cfg), B, tensors, &h->plan); plan_set_pix_mode(0); }
  else if (kind == 1) ok = build_resdec_plan(resdec_cfg_from_ints(cfg), B, z_ld, tensors, &h->plan);
  else if (kind == 2) { AttrDecCfg a; a.size_z = cfg[0]; a.att_size = cfg[1]; ok = build_attrdec_plan(a, B, z_ld, tensors, &h->plan); }
  else if (kind == 3) ok = build_unet_oa_plan(unet_oa_cfg_from_ints(cfg), B, z_ld, tensors, &h->plan);
  else if (kind == 4) { MlpCfg c; c.n_mod = cfg[0]; c.size_z = cfg[1]; ok = build_mlp_plan(c, B, tensors, &h->plan); }
  else if (kind == 5) ok = build_resenc_plan(resdec_cfg_from_ints(cfg), B, tensors, &h->plan);
  else if (kind == 6) { AttrEncCfg c; c.size_z = cfg[0]; c.att_size = cfg[1]; c.ae = cfg[2]; ok = build_attrenc_plan(c, B, tensors, &h->plan); }
  if (!ok) { g_err = h->plan.error.empty() ? "unknown plan kind" : h->plan.error; delete h; return nullptr; }
  h->ws.assign((size_t)h->plan.ws_floats, std::nanf(""));      // poison: every read must be preceded by a write
  return h;
}

int sbmref_num_specs(int kind, const int* cfg) {
  if (kind == 0) return (int)unet_tensor_specs(unet_cfg_from_ints(cfg)).size();
  if (kind == 1) return (int)resdec_tensor_specs(resdec_cfg_from_ints(cfg)).size();
  if (kind == 3) return (int)unet_oa_tensor_specs(unet_oa_cfg_from_ints(cfg)).size();
  if (kind == 4) { MlpCfg c; c.n_mod = cfg[0]; c.size_z = cfg[1]; return (int)mlp_tensor_specs(c).size(); }
  if (kind == 5) return (int)resenc_tensor_specs(resdec_cfg_from_ints(cfg)).size();
  if (kind == 6) { AttrEncCfg c; c.size_z = cfg[0]; c.att_size = cfg[1]; c.ae = cfg[2]; return (int)attrenc_tensor_specs(c).size(); }
  AttrDecCfg a; a.size_z = cfg[0]; a.att_size = cfg[1];
  return (int)attrdec_tensor_specs(a).size();
}

long sbmref_spec(int kind, const int* cfg, int i, char* name, int cap) {
  std::vector<TensorSpec> s;
  if (kind == 0) s = unet_tensor_specs(unet_cfg_from_ints(cfg));
  else if (kind == 1) s = resdec_tensor_specs(resdec_cfg_from_ints(cfg));
  else if (kind == 3) s = unet_oa_tensor_specs(unet_oa_cfg_from_ints(cfg));
  else if (kind == 4) { MlpCfg c; c.n_mod = cfg[0]; c.size_z = cfg[1]; s = mlp_tensor_specs(c); }
  else if (kind == 5) s = resenc_tensor_specs(resdec_cfg_from_ints(cfg));
  else if (kind == 6) { AttrEncCfg c; c.size_z = cfg[0]; c.att_size = cfg[1]; c.ae = cfg[2]; s = attrenc_tensor_specs(c); }
  else { AttrDecCfg a; a.size_z = cfg[0]; a.att_size = cfg[1]; s = attrdec_tensor_specs(a); }
  if (i < 0 || i >= (int)s.size()) return -1;
  std::strncpy(name, s[i].name.c_str(), cap - 1); name[cap - 1] = 0;
  return (long)s[i].numel();
}

int sbmref_run(void* hv, const float* in, const float* t, float* out) {
  RefHandle* h = (RefHandle*)hv;
  interpret(h->plan, h->ws, in, t, out);
  return 0;
}

// info: [n_ops, ws_floats, params_floats, flops_dense, flops_exec, n_gemm]
void sbmref_info(void* hv, double* o) {
  RefHandle* h = (RefHandle*)hv;
  int ng = 0; for (auto& op : h->plan.ops) ng += op.kind == OP_GEMM;
  o[0] = (double)h->plan.ops.size(); o[1] = (double)h->plan.ws_floats; o[2] = (double)h->plan.params.size();
  o[3] = h->plan.flops_dense; o[4] = h->plan.flops_exec; o[5] = ng;
}

int sbmref_run_aux(void* hv, const float* in, const float* t, const float* aux, float* out) {
  RefHandle* h = (RefHandle*)hv;
  interpret(h->plan, h->ws, in, t, out, 1 << 30, aux);
  return 0;
}
int sbmref_run_partial(void* hv, const float* in, const float* t, float* out, int n_ops) {
  RefHandle* h = (RefHandle*)hv;
  interpret(h->plan, h->ws, in, t, out, n_ops);
  return 0;
}
int sbmref_run_range(void* hv, const float* in, const float* t, float* out, int first, int last) {
  RefHandle* h = (RefHandle*)hv;
  Plan sub = h->plan;                      // copy of descriptors only (params vector copied too; test-only)
  if (first > 0) sub.time_off = h->plan.time_off;
  std::vector<Op> ops(h->plan.ops.begin() + first, h->plan.ops.begin() + std::min<size_t>(last, h->plan.ops.size()));
  sub.ops.swap(ops);
  interpret(sub, h->ws, in, first == 0 ? t : nullptr, out);
  return 0;
}
int sbmref_op_name(void* hv, int i, char* name, int cap) {
  RefHandle* h = (RefHandle*)hv;
  if (i < 0 || i >= (int)h->plan.ops.size()) return 1;
  std::string s = std::to_string(h->plan.ops[i].kind) + ":" + h->plan.ops[i].name;
  std::strncpy(name, s.c_str(), cap - 1); name[cap - 1] = 0;
  return 0;
}
const float* sbmref_workspace(void* hv) { return ((RefHandle*)hv)->ws.data(); }

void sbmref_free(void* hv) { delete (RefHandle*)hv; }

}  // extern "C"
