// pixrun.hip -- the 4x4 and 2x2 levels of the score net as launches of SAMPLE-OWNED workgroups.
//
// The XCD-resident chain kernel (chain.hip) carries a 32-sample tile through the low-resolution ops with a cluster barrier after every
// op: 57 barriers and as many first-touch weight latencies per step, 0.57 ms for 21 GFLOP.  Nothing in the network mixes samples, so
// this kernel gives a workgroup ONE sample and lets it run through the whole op list alone: no barrier ever leaves the workgroup, the
// sample's activations live in LDS from the first op to the last, and the level's weights are streamed from L2 -- all workgroups of an
// XCD walk the same bytes in the same order, so the L2 serves 31 of 32 readers (tools/stream_probe.hip: 256 workgroups x 9.5 MB in
// 178 us while issuing the MFMAs below = 13.7 TB/s out of the L2s, 109 TFLOP/s).
//
//   rows      : the PIXELS of the sample are the M rows of v_mfma_f32_16x16x4_f32 (16 pixels at 4x4; at 2x2 rows 4..15 are idle)
//   activations: LDS slab T[k >> 4][row][16], the 16 channels of a chunk stored at position (k & 3) * 4 + ((k >> 2) & 3): lane
//               (row = l & 15, g = l >> 4) reads ONE float4 = A[row][16 c + 4 j + g], j = 0..3, the operands of four MFMAs
//   taps      : a convolution tap is a row gather: lane's source row = src[tap][row] (or the zero row) -- padding, stride, transposed
//               convolution and nearest upsampling are all just row tables built on the host
//   weights   : transposed copies [tap][k][n] (n contiguous): lane (q = l & 15, g) loads W[16 c + 4 j + g][n0 + 4 q .. + 3], j = 0..3;
//               accumulator i of four holds column n0 + 4 q + i (rows 4 g + v in VGPR v)
//   waves     : a pass covers 1, 2 or 4 blocks of 64 columns; the waves of a block split K and meet in LDS
//   epilogue  : 512 threads = 16 rows x 32 column pairs per block; folded GroupNorm, bias, residual, activation, LDS + global store
//
// Every output is also stored to the workspace in the plan's [P][B][ld] layout: skip connections, the ops behind the run and the
// op-by-op debug tools find it where the other paths would have put it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "device.h"

namespace sbm {

namespace {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4p gf32x4p;

constexpr int PNT = 512, PNW = 8;
constexpr int PIX_STAGE = 101;
constexpr int P_RING = 4;                 // weight chunks (16 k x 64 columns = 4 dwordx4 per lane) in flight per wave
constexpr int P_ZERO_F = 64, P_RED_F = PNW * 16 * 64, P_STAT_SLOTS = 64, P_SCR_F = 4352, P_MAXTAP = 20, P_TAP_I = 12;

struct PixTap { int w_off, a_lds, a_cs, nch, src2, pad0, pad1, pad2; int src[4]; };   // src: 16 signed bytes, source row of output row r (-1: none)
static_assert(sizeof(PixTap) == P_TAP_I * 4, "PixTap is staged to LDS as 12 ints");

struct PixOpD {
  int kind;                              // OP_GEMM, OP_DWCONV7, OP_GN_RES, OP_LINATTN, PIX_STAGE
  int P_out, Npad, nstore, act, ldo, nblk;
  int gn_stat, stat_out;                 // LDS statistics slots consumed / produced, -1 none
  int in_lds, in_cs, out_lds, out_cs, res_lds, res_cs;
  int res_row[4];                        // 16 signed bytes: residual row of output row r
  int tap_first, tap_cnt;
  const float* pf_ptr; int pf_lines;     // L2 warm-up: the first weight lines of the NEXT GEMM op of the program (touched under this op's epilogue)
  int k_ntap1, k_nch1, k_nch2;           // source-1 taps and their common chunk count; chunks of the source-2 tap (0: none)
  float gn_inv;
  const float* Wt; const float* Wt2; const PixTap* taps;
  const float* bias; const float* gn_s; const float* gn_b1;
  float* out_g; float2* stat_g;
  DwArgs dw; GnResArgs gr; AttnArgs at;
  int x_lds, x_cs;                       // GN_RES: the residual input x
  const float* st_src; const float2* st_part; int st_ld, st_C, st_P, st_lds, st_cs, st_stat, st_nparts;
};

__device__ __forceinline__ float pix_act(int act, float v) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + sbm_erf(v * 0.70710678118654752440f));
    case ACT_LRELU02: return v >= 0.f ? v : 0.2f * v;
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ACT_SILU: return v / (1.0f + expf(-v));
    case ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
    case ACT_LRELU01: return v >= 0.f ? v : 0.1f * v;
    default: return v;
  }
}
// float offset of channel k of row r inside a slab with chunk stride cs
__device__ __forceinline__ int pix_at(int k, int r, int cs) { return (k >> 4) * cs + r * 16 + (k & 3) * 4 + ((k >> 2) & 3); }
__device__ __forceinline__ int sbyte(const int* w, int r) { return (int)(signed char)((w[r >> 2] >> (8 * (r & 3))) & 0xff); }

}  // namespace

__global__ __launch_bounds__(PNT, 1) void pixrun_kernel(const PixOpD* __restrict__ ops, int count, int B, int act_floats, const PixTap* __restrict__ taps, int ntaps, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* zero = lds + act_floats;                        // P_ZERO_F zeros: the source of rows a tap does not reach
  float* red = zero + P_ZERO_F;                          // [8 waves][16 rows][64 columns]
  float* stat = red + P_RED_F;                           // [slots][2]
  float* scr = stat + P_STAT_SLOTS * 2;                  // scratch (LinearAttention contexts, reductions)
  int* tap_all = reinterpret_cast<int*>(scr + P_SCR_F);  // the tap tables of the whole program, staged once
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int q = lane & 15, g = lane >> 4;
  unsigned sink = 0;
  for (int i = tid; i < act_floats + P_ZERO_F + P_RED_F; i += PNT) lds[i] = 0.f;
  for (int i = tid; i < ntaps * P_TAP_I; i += PNT) tap_all[i] = reinterpret_cast<const int*>(taps)[i];
  __syncthreads();
  for (int k = 0; k < count; ++k) {
    const PixOpD& op = ops[k];
    const int kind = op.kind;
    if (dbg && b == 0 && tid == 0) dbg[k] = __builtin_amdgcn_s_memtime();
    if (kind == PIX_STAGE) {
      const int C4 = op.st_C >> 2;
      for (int idx = tid; idx < op.st_P * C4; idx += PNT) {
        const int c4 = idx % C4, p = idx / C4;
        const float4 v = *reinterpret_cast<const float4*>(op.st_src + ((long long)p * B + b) * op.st_ld + 4 * c4);
        float* d = lds + op.st_lds + ((4 * c4) >> 4) * op.st_cs + p * 16 + (c4 & 3);     // k = 4 c4 + e: position e * 4 + (c4 & 3)
        d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
      }
      if (op.st_stat >= 0 && tid == 0) {
        double s = 0.0, qq = 0.0;
        for (int i = 0; i < op.st_nparts; ++i) { const float2 v = op.st_part[(long long)i * B + b]; s += (double)v.x; qq += (double)v.y; }
        stat[op.st_stat * 2] = (float)s; stat[op.st_stat * 2 + 1] = (float)qq;
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_DWCONV7) {
      // depth-wise 7x7 over an image of at most 4x4: the window covers the whole image, so an output walks the source pixels in raster
      // order (= tap order).  Thread = (channel, every tpc-th pixel).
      const DwArgs& d = op.dw;
      const int C = d.C, P = d.H * d.W, tpc = PNT / C;
      const int c = tid % C, pg = tid / C;
      const float bi = d.bias[c];
      const float cv = d.cond ? dw_cond_row(d, b)[c] : 0.f;
      float s = 0.f, qq = 0.f;
      if (pg < tpc) {
        for (int p = pg; p < P; p += tpc) {
          const int py = p / d.W, px = p - py * d.W;
          float acc = 0.f;
          for (int q0 = 0; q0 < P; q0 += 4) {             // P is 4 or 16 (or 1): four independent weight loads in flight
            float xv[4], wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int qp = min(q0 + j, P - 1), qy = qp / d.W, qx = qp - qy * d.W;
              wv[j] = d.w[((qy - py + 3) * 7 + (qx - px + 3)) * C + c];
              xv[j] = lds[op.in_lds + pix_at(c, qp, op.in_cs)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (q0 + j < P) acc = fmaf(xv[j], wv[j], acc);
          }
          acc += bi;
          if (d.cond) acc += cv;
          lds[op.out_lds + pix_at(c, p, op.out_cs)] = acc;
          if (op.out_g) d.out[((long long)p * B + b) * d.ldo + c] = acc;
          s += acc; qq += acc * acc;
        }
      }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) { s += __shfl_xor(s, m); qq += __shfl_xor(qq, m); }
      if (lane == 0) { scr[wave] = s; scr[PNW + wave] = qq; }
      __syncthreads();
      if (tid == 0) {
        float ss = 0.f, q2 = 0.f;
        for (int w = 0; w < PNW; ++w) { ss += scr[w]; q2 += scr[PNW + w]; }
        if (op.stat_out >= 0) { stat[op.stat_out * 2] = ss; stat[op.stat_out * 2 + 1] = q2; }
        if (d.stat && op.stat_g) {                       // the plan's layout is one partial per pixel: the whole sum goes to pixel 0
          d.stat[b] = make_float2(ss, q2);
          for (int p = 1; p < P; ++p) d.stat[(long long)p * B + b] = make_float2(0.f, 0.f);
        }
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_GN_RES) {
      const GnResArgs& gr = op.gr;
      const float su = stat[op.gn_stat * 2], sq = stat[op.gn_stat * 2 + 1];
      const double mean_d = (double)su * (double)gr.inv_count;
      double var = (double)sq * (double)gr.inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
      const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + 1e-5));
      for (int idx = tid; idx < gr.P * gr.C; idx += PNT) {
        const int c = idx % gr.C, p = idx / gr.C;
        const float u = lds[op.in_lds + pix_at(c, p, op.in_cs)], x = lds[op.x_lds + pix_at(c, p, op.x_cs)];
        const float y = ((u - mean) * rstd) * gr.gamma[c] + gr.beta[c] + x;
        lds[op.out_lds + pix_at(c, p, op.out_cs)] = y;
        if (op.out_g) gr.out[((long long)p * B + b) * gr.ldo + c] = y;
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_LINATTN) {
      // LinearAttention core (unet_model.py:169-175) of one sample: n <= 16 positions, heads x 32 channels; q | k | v in one slab
      const AttnArgs& a = op.at;
      const int n = a.n, hd = a.heads * 32, ics = op.in_cs;
      if (tid < hd) {                                    // k: softmax over the positions, per channel
        float m = -INFINITY;
        for (int p = 0; p < n; ++p) m = fmaxf(m, lds[op.in_lds + pix_at(hd + tid, p, ics)]);
        float z = 0.f;
        for (int p = 0; p < n; ++p) { const float e = expf(lds[op.in_lds + pix_at(hd + tid, p, ics)] - m); lds[op.in_lds + pix_at(hd + tid, p, ics)] = e; z += e; }
        const float inv = 1.0f / z;
        for (int p = 0; p < n; ++p) lds[op.in_lds + pix_at(hd + tid, p, ics)] *= inv;
      } else if (tid >= 256 && tid < 256 + n * a.heads) {   // q: softmax over the head's channels, per position; scaled
        const int t = tid - 256, p = t / a.heads, h = t - p * a.heads;
        float m = -INFINITY;
        for (int dd = 0; dd < 32; ++dd) m = fmaxf(m, lds[op.in_lds + pix_at(h * 32 + dd, p, ics)]);
        float z = 0.f;
        for (int dd = 0; dd < 32; ++dd) { const float e = expf(lds[op.in_lds + pix_at(h * 32 + dd, p, ics)] - m); lds[op.in_lds + pix_at(h * 32 + dd, p, ics)] = e; z += e; }
        const float inv = a.scale / z;
        for (int dd = 0; dd < 32; ++dd) lds[op.in_lds + pix_at(h * 32 + dd, p, ics)] *= inv;
      }
      __syncthreads();
      // context[h][d][e] = sum_p k[p][h, d] v[p][h, e]: thread = (h, d, 8 consecutive e)
      for (int t = tid; t < a.heads * 32 * 4; t += PNT) {
        const int h = t >> 7, dd = (t >> 2) & 31, e0 = (t & 3) * 8;
        float cx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cx[e] = 0.f;
        for (int p = 0; p < n; ++p) {
          const float kv = lds[op.in_lds + pix_at(hd + h * 32 + dd, p, ics)];
#pragma unroll
          for (int e = 0; e < 8; ++e) cx[e] = fmaf(kv, lds[op.in_lds + pix_at(2 * hd + h * 32 + e0 + e, p, ics)], cx[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) scr[(h * 32 + dd) * 33 + e0 + e] = cx[e];
      }
      __syncthreads();
      // out[p][h, e] = sum_d q[p][h, d] context[h][d][e]: thread = (p, channel)
      for (int t = tid; t < n * hd; t += PNT) {
        const int ch = t % hd, p = t / hd, h = ch >> 5, e = ch & 31;
        float o = 0.f;
        for (int dd = 0; dd < 32; ++dd) o = fmaf(lds[op.in_lds + pix_at(h * 32 + dd, p, ics)], scr[(h * 32 + dd) * 33 + e], o);
        lds[op.out_lds + pix_at(ch, p, op.out_cs)] = o;
        if (op.out_g) a.out[((long long)p * B + b) * a.ldo + ch] = o;
      }
      __syncthreads();
      continue;
    }
    // ---------------------------------------------------------------- GEMM
#define PSTAMP(ph) do { if (dbg && b == 0 && tid == 0 && k < 30) dbg[64 + k * 6 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
    PSTAMP(0);
    const int* tapl = tap_all + op.tap_first * P_TAP_I;
    const bool gn = op.gn_stat >= 0;
    float mean = 0.f, rstd = 1.f, inv_r = 1.f;
    if (gn) {
      const float su = stat[op.gn_stat * 2], sq = stat[op.gn_stat * 2 + 1];
      const double mean_d = (double)su * (double)op.gn_inv;
      double var = (double)sq * (double)op.gn_inv - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
      mean = (float)mean_d; rstd = rsqrtf((float)var + 1e-5f); inv_r = sqrtf((float)var + 1e-5f);
    }
    const int nb = (op.nstore + 63) >> 6, ntap = op.tap_cnt, npad = op.Npad;
    // At most four output rows (the 2x2 level): v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 outer products per issue, rows = the four
    // pixels, one k per lane group g -- at a quarter of the 16x16x4 issue time per weight byte: these layers become L2-streaming bound.
    // The four lane groups hold partial sums over disjoint k and meet in the LDS reduction with the waves' K slices.
    const bool small = __builtin_amdgcn_readfirstlane(op.P_out) <= 4;
    const int erow = tid >> 5, ec2 = (tid & 31) * 2;     // epilogue: this thread's row and column pair inside a block
    const int rrow = op.res_lds >= 0 ? sbyte(op.res_row, erow) : 0;
    float sacc = 0.f, qacc = 0.f;
    for (int cb0 = 0; cb0 < nb;) {
      const int rem = nb - cb0, par = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
      const int ksh = par == 4 ? 1 : (par == 2 ? 2 : 3);   // waves per block = K slices = 8 / par, as a shift (no integer divisions in the set-up)
      const int ksplit = 1 << ksh, wblk = wave >> ksh, wk = wave & (ksplit - 1);
      const int col = min((cb0 + wblk) * 64 + 4 * q, npad - 4);
      // epilogue operands of this thread's outputs, asked for before the K loop
      float2 e_b[4], e_s[4], e_g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e_b[j] = e_s[j] = e_g[j] = make_float2(0.f, 0.f);
        const int n = (cb0 + j) * 64 + ec2;
        if (j < par && n < op.nstore && erow < op.P_out) {
          if (op.bias) e_b[j] = *reinterpret_cast<const float2*>(op.bias + n);
          if (gn) { e_s[j] = *reinterpret_cast<const float2*>(op.gn_s + (long long)erow * npad + n); e_g[j] = *reinterpret_cast<const float2*>(op.gn_b1 + (long long)erow * npad + n); }
        }
      }
      f32x4p acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4p{0.f, 0.f, 0.f, 0.f};
      // chunks of this wave in this pass (all source-1 taps have the same depth: no walk over the tap table here)
      const int nk = op.k_ntap1 * (op.k_nch1 > wk ? (op.k_nch1 - wk + ksplit - 1) >> ksh : 0) + (op.k_nch2 > wk ? (op.k_nch2 - wk + ksplit - 1) >> ksh : 0);
      // Two cursors walk this wave's chunk list (tap by tap, chunks wk, wk + ksplit, ... of each): L issues the weight loads P_RING chunks
      // ahead of C, which consumes them.  Per chunk a cursor costs one pointer increment; the tap's fields are read from LDS once per tap.
      struct CurL { const float* w; int left, t; };
      struct CurC { int a, stepa, left, t, src2; };
      const long long wstep = (long long)ksplit * 16 * npad;
      auto next_l = [&](CurL& u) {
        for (;;) {
          ++u.t;
          if (u.t >= ntap) { u.left = 1 << 30; u.w = op.Wt; return; }     // cannot happen (nk counts exactly the chunks below): never walk off the table
          const int* tp = tapl + u.t * P_TAP_I;
          const int nch = __builtin_amdgcn_readfirstlane(tp[3]);
          if (nch <= wk) continue;                         // this wave has no chunk of that tap
          u.left = (nch - wk + ksplit - 1) >> ksh;
          u.w = (__builtin_amdgcn_readfirstlane(tp[4]) ? op.Wt2 : op.Wt) + __builtin_amdgcn_readfirstlane(tp[0]) + (long long)(16 * wk + g) * npad + col;
          return;
        }
      };
      auto next_c = [&](CurC& u) {
        for (;;) {
          ++u.t;
          if (u.t >= ntap) { u.left = 1 << 30; u.a = act_floats; u.stepa = 0; u.src2 = 0; return; }
          const int* tp = tapl + u.t * P_TAP_I;
          const int nch = __builtin_amdgcn_readfirstlane(tp[3]);
          if (nch <= wk) continue;
          u.left = (nch - wk + ksplit - 1) >> ksh;
          u.src2 = __builtin_amdgcn_readfirstlane(tp[4]);
          const int srow = sbyte(tp + 8, small ? (lane & 3) : q);
          const int acs = __builtin_amdgcn_readfirstlane(tp[2]);
          if (srow >= 0) { u.a = __builtin_amdgcn_readfirstlane(tp[1]) + srow * 16 + g * 4 + wk * acs; u.stepa = ksplit * acs; }
          else { u.a = act_floats + g * 4; u.stepa = 0; }
          return;
        }
      };
      CurL L{nullptr, 0, -1};
      CurC C{0, 0, 0, -1, 0};
      f32x4p ring[P_RING][4];
      auto issue = [&](f32x4p* r) {
        if (L.left == 0) next_l(L);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = *(const gf32x4p*)(unsigned long long)(L.w + (long long)(4 * j) * npad);
        L.w += wstep; --L.left;
      };
      auto consume = [&](const f32x4p* r) {
        if (C.left == 0) next_c(C);
        float4 a = *reinterpret_cast<const float4*>(lds + C.a);
        C.a += C.stepa; --C.left;
        if (gn && C.src2) { a.x *= inv_r; a.y *= inv_r; a.z *= inv_r; a.w *= inv_r; }
        const float av[4] = {a.x, a.y, a.z, a.w};
        if (small) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], r[j][i], acc[i], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], r[j][i], acc[i], 0, 0, 0);
        }
      };
#pragma unroll
      for (int r = 0; r < P_RING; ++r) if (r < nk) issue(ring[r]);
      PSTAMP(1);
      for (int i = 0; i < nk; i += P_RING) {
#pragma unroll
        for (int r = 0; r < P_RING; ++r) {
          if (i + r < nk) {
            consume(ring[r]);
            if (i + r + P_RING < nk) issue(ring[r]);
          }
        }
      }
      PSTAMP(2);
      // L2 warm-up for the next GEMM: its ring fill would otherwise be a chip-wide first touch on this op's critical path; the touches
      // fly under the reduction and the epilogue below (results only feed `sink`)
      float pf0 = 0.f, pf1 = 0.f;
      if (cb0 + par >= nb && op.pf_ptr) {
        if (tid < op.pf_lines) pf0 = op.pf_ptr[(long long)tid * 32];
        if (tid + PNT < op.pf_lines) pf1 = op.pf_ptr[(long long)(tid + PNT) * 32];
      }
      // partials: red[wave][row 4 g + v][column 4 q + i]
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<float4*>(red + (wave * 16 + 4 * g + v) * 64 + 4 * q) = make_float4(acc[0][v], acc[1][v], acc[2][v], acc[3][v]);
      __syncthreads();
      PSTAMP(3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < par) {
        const int n = (cb0 + j) * 64 + ec2;
        float t0 = 0.f, t1 = 0.f;
        if (small) {                                     // rows 4 gg + erow of a wave's slab are the four k groups' partials of row erow
          if (erow < 4)
            for (int w = 0; w < ksplit; ++w)
#pragma unroll
              for (int gg = 0; gg < 4; ++gg) { const float2 v = *reinterpret_cast<const float2*>(red + ((j * ksplit + w) * 16 + 4 * gg + erow) * 64 + ec2); t0 += v.x; t1 += v.y; }
        } else {
          for (int w = 0; w < ksplit; ++w) { const float2 v = *reinterpret_cast<const float2*>(red + ((j * ksplit + w) * 16 + erow) * 64 + ec2); t0 += v.x; t1 += v.y; }
        }
        const bool ok = n < op.nstore && erow < op.P_out;
        float o0 = 0.f, o1 = 0.f;
        if (ok) {
          if (gn) { t0 = t0 * rstd + (e_g[j].x - (mean * rstd) * e_s[j].x); t1 = t1 * rstd + (e_g[j].y - (mean * rstd) * e_s[j].y); }
          t0 += e_b[j].x; t1 += e_b[j].y;
          if (op.res_lds >= 0) {
            t0 += lds[op.res_lds + pix_at(n, rrow, op.res_cs)]; t1 += lds[op.res_lds + pix_at(n + 1, rrow, op.res_cs)];
          }
          o0 = pix_act(op.act, t0); o1 = pix_act(op.act, t1);
          if (op.out_lds >= 0) { lds[op.out_lds + pix_at(n, erow, op.out_cs)] = o0; lds[op.out_lds + pix_at(n + 1, erow, op.out_cs)] = o1; }
          if (op.out_g) *reinterpret_cast<float2*>(op.out_g + ((long long)erow * B + b) * op.ldo + n) = make_float2(o0, o1);
          sacc += o0 + o1; qacc += o0 * o0 + o1 * o1;
        }
        if (op.stat_g) {                                  // the plan's partial layout: one (sum, sumsq) per (pixel, 32-column block)
          float s32 = o0 + o1, q32 = o0 * o0 + o1 * o1;
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) { s32 += __shfl_xor(s32, m); q32 += __shfl_xor(q32, m); }
          const int nbk = n >> 5;
          if ((tid & 15) == 0 && nbk < op.nblk && erow < op.P_out) op.stat_g[((long long)erow * op.nblk + nbk) * B + b] = make_float2(s32, q32);
        }
        }
      }
      __syncthreads();                                    // red (and the taps, after the last pass) are reused
      PSTAMP(4);
      sink ^= __float_as_uint(pf0) ^ __float_as_uint(pf1);
      cb0 += par;
    }
    if (op.stat_out >= 0) {
      float ss = sacc, q2 = qacc;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) { ss += __shfl_xor(ss, m); q2 += __shfl_xor(q2, m); }
      if (lane == 0) { scr[wave] = ss; scr[PNW + wave] = q2; }
      __syncthreads();
      if (tid == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int w = 0; w < PNW; ++w) { a0 += scr[w]; a1 += scr[PNW + w]; }
        stat[op.stat_out * 2] = a0; stat[op.stat_out * 2 + 1] = a1;
      }
      __syncthreads();
    }
  }
  if (dbg && b == 0 && tid == 0) dbg[count] = __builtin_amdgcn_s_memtime();
  if (sink == 0x9e3779b9u && dbg) dbg[255] = 1;          // consumes the warm-up touches
}

// ------------------------------------------------------------------------------------------------
// host: build the program of one run of chain-level ops, launch it
// ------------------------------------------------------------------------------------------------
namespace {

struct PSlot { long long base; int ld; int P; int rows; int lds; int t0, t1; };

int rows_of(int P) { return P <= 16 ? 16 : (P + 15) / 16 * 16; }
void set_sbyte(int* w, int r, int v) { w[r >> 2] = (w[r >> 2] & ~(0xff << (8 * (r & 3)))) | ((v & 0xff) << (8 * (r & 3))); }

struct PBuilder {
  const Plan& plan; const DevPlan& dp; int first, count;
  std::vector<PSlot> slots; int act_floats = 0;
  std::vector<PixOpD> ops; std::vector<PixTap> taps;
  std::map<long long, int> stat_of;
  std::map<int, int> produced;
  const std::vector<PSlot>* preset = nullptr;
  std::vector<int> touched;
  const char* err = nullptr;

  // LDS float offset of (row 0, channel 0 of the view) and the slab's chunk stride
  int slab(const View& v, int* cs, int* slot_index) {
    const int P = v.H * v.W;
    for (size_t i = 0; i < slots.size(); ++i) {
      PSlot& s = slots[i];
      if (s.ld == v.ld && s.P == P && v.off >= s.base && v.off < s.base + s.ld) {
        *cs = s.rows * 16; if (slot_index) *slot_index = (int)i;
        touched.push_back((int)i);
        if ((v.off - s.base) & 15) err = "pixel kernel: a view does not start on a 16-channel boundary";
        return s.lds + (int)((v.off - s.base) >> 4) * s.rows * 16;
      }
    }
    long long base = v.off;
    for (int i = 0; i < (int)plan.ops.size(); ++i) {
      const Op& o = plan.ops[i];
      auto look = [&](const View& w) { if (w.ld == v.ld && w.H * w.W == P && w.off < base && base - w.off < v.ld) base = w.off; };
      if (o.kind == OP_GEMM) { look(o.g.a); look(o.g.out); if (o.g.has2) look(o.g.a2); if (o.g.has_res) look(o.g.r); }
      else if (o.kind == OP_DWCONV7) { look(o.dw.in); look(o.dw.out); }
      else if (o.kind == OP_GN_RES) { look(o.gr.u); look(o.gr.x); look(o.gr.out); }
      else if (o.kind == OP_LINATTN) { look(o.at.qkv); look(o.at.out); }
    }
    PSlot s{base, v.ld, P, rows_of(P), act_floats, (int)ops.size(), (int)ops.size()};
    if (v.ld & 15) err = "pixel kernel: a tensor's channel pitch is not a multiple of 16";
    if (preset) {
      for (const PSlot& ps : *preset) if (ps.base == base && ps.ld == v.ld && ps.P == P) s.lds = ps.lds;
    } else {
      act_floats += s.rows * s.ld;
    }
    slots.push_back(s);
    touched.push_back((int)slots.size() - 1);
    *cs = s.rows * 16; if (slot_index) *slot_index = (int)slots.size() - 1;
    if ((v.off - s.base) & 15) err = "pixel kernel: a view does not start on a 16-channel boundary";
    return s.lds + (int)((v.off - s.base) >> 4) * s.rows * 16;
  }
  // An op that writes only PART of a buffer (one half of a skip concatenation) while the rest was produced outside the run: the whole
  // buffer is staged first, so that later readers of the full width find the other half.
  void mark_produced(const View& v) {
    int cs, si; slab(v, &cs, &si);
    if (!produced.count(si) && (v.off != slots[si].base || v.C < slots[si].ld)) {
      PixOpD st{}; st.kind = PIX_STAGE;
      const PSlot& s = slots[si];
      st.st_src = dp.ws + s.base; st.st_ld = s.ld; st.st_C = s.ld; st.st_P = s.P; st.st_lds = s.lds; st.st_cs = s.rows * 16; st.st_stat = -1;
      ops.push_back(st);
    }
    produced[si] = 1;
  }
  int stat_slot(long long off) {
    auto it = stat_of.find(off);
    if (it != stat_of.end()) return it->second;
    const int s = (int)stat_of.size();
    stat_of[off] = s;
    return s;
  }
  int need(const View& v, int* cs, long long stat_off = -1, int nparts = 0) {
    int si; const int off = slab(v, cs, &si);
    const bool have_stat = stat_off < 0 || stat_of.count(stat_off);
    if (produced.count(si) && have_stat) return off;
    PixOpD st{}; st.kind = PIX_STAGE;
    const PSlot& s = slots[si];
    st.st_src = dp.ws + s.base; st.st_ld = s.ld; st.st_C = s.ld; st.st_P = s.P; st.st_lds = s.lds; st.st_cs = s.rows * 16; st.st_stat = -1;
    if (produced.count(si)) st.st_P = 0;
    if (stat_off >= 0 && !stat_of.count(stat_off)) {
      st.st_stat = stat_slot(stat_off); st.st_part = reinterpret_cast<const float2*>(dp.ws + stat_off); st.st_nparts = nparts;
    }
    ops.push_back(st);
    produced[si] = 1;
    return off;
  }
};

struct PProgram { PixOpD* d_ops = nullptr; PixTap* d_taps = nullptr; int n_ops = 0; int act_floats = 0; int ntaps = 0; };
std::map<std::pair<const void*, long long>, PProgram> g_pprograms;

size_t pix_lds_bytes(int act_floats, int ntaps) { return sizeof(float) * ((size_t)act_floats + P_ZERO_F + P_RED_F + P_STAT_SLOTS * 2 + P_SCR_F) + sizeof(int) * (size_t)(ntaps + 1) * P_TAP_I; }

const char* pixrun_build(const DevPlan& dp, int first, int count) {
  const Plan& plan = *dp.plan;
  const auto key = std::make_pair((const void*)&plan, (long long)first * 4096 + count);
  if (g_pprograms.count(key)) return nullptr;
  auto fill = [&](PBuilder& b) -> const char* {
    for (int i = first; i < first + count; ++i) {
      const Op& op = plan.ops[i];
      b.touched.clear();
      PixOpD d{};
      d.kind = op.kind; d.gn_stat = -1; d.stat_out = -1; d.out_lds = -1; d.res_lds = -1; d.in_lds = -1;
      if (op.kind == OP_DWCONV7) {
        const DwOp& o = op.dw;
        if (o.in.H > 4 || o.in.W > 4 || (PNT % o.in.C) || o.in.C > PNT) return "pixel kernel: depth-wise op outside the supported shapes";
        d.in_lds = b.need(o.in, &d.in_cs);
        d.out_lds = b.slab(o.out, &d.out_cs, nullptr); b.mark_produced(o.out);
        d.dw = make_dw_args(plan, o, dp.prm, dp.ws, dp.ctl);
        if (o.stat_off >= 0) d.stat_out = b.stat_slot(o.stat_off);
      } else if (op.kind == OP_GN_RES) {
        const GnResOp& o = op.gr;
        if (o.u.H * o.u.W > 16 || o.out_ext) return "pixel kernel: unsupported GroupNorm+residual op";
        d.in_lds = b.need(o.u, &d.in_cs, o.part_off, o.nparts);
        d.x_lds = b.need(o.x, &d.x_cs);
        d.gn_stat = b.stat_slot(o.part_off);
        d.out_lds = b.slab(o.out, &d.out_cs, nullptr); b.mark_produced(o.out);
        d.gr = make_gnres_args(o, dp.prm, dp.ws);
      } else if (op.kind == OP_LINATTN) {
        const AttnOp& o = op.at;
        if (o.qkv.H * o.qkv.W > 16 || o.dh != 32 || o.heads * 32 > 256 || o.heads * 32 * 33 > P_SCR_F) return "pixel kernel: unsupported attention op";
        d.in_lds = b.need(o.qkv, &d.in_cs);
        d.out_lds = b.slab(o.out, &d.out_cs, nullptr); b.mark_produced(o.out);
        d.at = make_attn_args(o, dp.ws);
      } else if (op.kind == OP_GEMM) {
        const GemmOp& g = op.g;
        const int P_out = g.Hout * g.Wout, P_in = g.a.H * g.a.W;
        if (P_out > 16 || P_in > 64 || g.a_ext || g.out_nchw || g.w_pix_stride || g.bias_pix_stride || g.a_bcast || g.a_rows || g.wt_off < 0) return "pixel kernel: unsupported GEMM op";
        if ((g.Kp & 15) || (g.has2 && (g.Kp2 & 15)) || (g.Npad & 3) || (g.out.C & 1)) return "pixel kernel: K must be a multiple of 16";
        int acs; const int a_lds = b.need(g.a, &acs, g.gn ? g.gn_part_off : -1, g.gn_nparts);
        int a2cs = 0, a2_lds = -1;
        if (g.has2) a2_lds = b.need(g.a2, &a2cs);
        if (g.has_res) d.res_lds = b.need(g.r, &d.res_cs);
        d.P_out = P_out; d.Npad = g.Npad; d.nstore = g.out.C; d.act = g.act; d.ldo = g.out.ld;
        d.Wt = dp.prm + g.wt_off; d.Wt2 = g.has2 ? dp.prm + g.wt2_off : nullptr;
        d.bias = g.bias_off >= 0 ? dp.prm + g.bias_off : nullptr;
        if (g.gn) { d.gn_stat = b.stat_slot(g.gn_part_off); d.gn_inv = g.gn_inv_count; d.gn_s = dp.prm + g.gn_s_off; d.gn_b1 = dp.prm + g.gn_b1_off; }
        d.out_g = dp.ws + g.out.off;
        d.out_lds = b.slab(g.out, &d.out_cs, nullptr); b.mark_produced(g.out);
        if (g.stat_off >= 0) { d.stat_out = b.stat_slot(g.stat_off); d.stat_g = reinterpret_cast<float2*>(dp.ws + g.stat_off); d.nblk = (g.out.C + 31) / 32; }
        d.tap_first = (int)b.taps.size();
        const int w2 = g.has2 ? g.a2.W : (g.has_res ? g.r.W : 1);
        for (int ky = 0; ky < g.KH; ++ky) for (int kx = 0; kx < g.KW; ++kx) {
          PixTap t{}; t.w_off = (int)((long long)(ky * g.KW + kx) * g.Kp * g.Npad); t.a_lds = a_lds; t.a_cs = acs; t.nch = g.Kp / 16; t.src2 = 0;
          bool any = false;
          for (int r = 0; r < 16; ++r) {
            int qp = -1;
            if (r < P_out) qp = tap_source_pixel(r / g.Wout, r % g.Wout, ky, kx, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W);
            set_sbyte(t.src, r, qp);
            any |= qp >= 0;
          }
          if (any) b.taps.push_back(t);
        }
        for (int r = 0; r < 16; ++r) set_sbyte(d.res_row, r, r < P_out ? ((r / g.Wout) / g.up2) * w2 + ((r % g.Wout) / g.up2) : 0);
        if (g.has2) {
          PixTap t{}; t.w_off = 0; t.a_lds = a2_lds; t.a_cs = a2cs; t.nch = g.Kp2 / 16; t.src2 = 1;
          for (int r = 0; r < 16; ++r) set_sbyte(t.src, r, r < P_out ? ((r / g.Wout) / g.up2) * w2 + ((r % g.Wout) / g.up2) : -1);
          b.taps.push_back(t);
        }
        d.tap_cnt = (int)b.taps.size() - d.tap_first;
        d.k_ntap1 = d.tap_cnt - (g.has2 ? 1 : 0); d.k_nch1 = g.Kp / 16; d.k_nch2 = g.has2 ? g.Kp2 / 16 : 0;
        if (d.tap_cnt > P_MAXTAP) return "pixel kernel: too many taps";
      } else {
        return "pixel kernel: op kind not supported";
      }
      // Results leave the launch (workspace store) only when something OUTSIDE the run reads them: a later op of the plan, or nothing
      // in this run at all (a debug sub-range).  Statistics likewise.
      {
        auto reads = [&](const Op& o, const View& v) {
          // same buffer (same pitch and pixel count, origins within one row of each other -- the test PBuilder::slab uses): conservative
          auto hit = [&](const View& w) { return w.ld == v.ld && w.H * w.W == v.H * v.W && w.off - v.off < v.ld && v.off - w.off < v.ld; };
          if (o.kind == OP_GEMM) return hit(o.g.a) || (o.g.has2 && hit(o.g.a2)) || (o.g.has_res && hit(o.g.r));
          if (o.kind == OP_DWCONV7) return hit(o.dw.in);
          if (o.kind == OP_GN_RES) return hit(o.gr.u) || hit(o.gr.x);
          if (o.kind == OP_LINATTN || o.kind == OP_ATTN) return hit(o.at.qkv);
          return true;                                   // any other kind: assume it reads everything
        };
        const View* outv = op.kind == OP_GEMM ? &op.g.out : (op.kind == OP_DWCONV7 ? &op.dw.out : (op.kind == OP_GN_RES ? &op.gr.out : &op.at.out));
        bool outside = i == (int)plan.ops.size() - 1;
        for (int j = 0; j < (int)plan.ops.size() && !outside; ++j) if ((j < first || j >= first + count) && j > i && reads(plan.ops[j], *outv)) outside = true;
        long long soff = op.kind == OP_GEMM ? op.g.stat_off : (op.kind == OP_DWCONV7 ? op.dw.stat_off : -1);
        bool stat_outside = false;
        for (int j = 0; j < (int)plan.ops.size() && soff >= 0; ++j)
          if ((j < first || j >= first + count) && ((plan.ops[j].kind == OP_GEMM && plan.ops[j].g.gn && plan.ops[j].g.gn_part_off == soff) || (plan.ops[j].kind == OP_GN_RES && plan.ops[j].gr.part_off == soff))) stat_outside = true;
        if (count == 1) { outside = true; stat_outside = true; }
        static float dummy_out;                            // non-null marker for the element-wise ops (they store through their own argument blocks)
        if (op.kind == OP_GEMM) { if (!outside) d.out_g = nullptr; }
        else d.out_g = outside ? &dummy_out : nullptr;
        if (!stat_outside) d.stat_g = nullptr; else if (op.kind == OP_DWCONV7) d.stat_g = reinterpret_cast<float2*>(&dummy_out);
      }
      if (b.err) return b.err;
      for (int si : b.touched) b.slots[si].t1 = std::max(b.slots[si].t1, (int)b.ops.size());
      b.ops.push_back(d);
    }
    return nullptr;
  };
  PBuilder b0{plan, dp, first, count};
  if (const char* e = fill(b0)) return e;
  std::vector<PSlot> packed = b0.slots;
  {
    std::vector<size_t> order(packed.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return packed[x].t0 < packed[y].t0; });
    std::vector<size_t> placed;
    int top = 0;
    for (size_t oi : order) {
      PSlot& sl = packed[oi];
      const int size = sl.rows * sl.ld + sl.rows * 16;                 // + one chunk of slack: K is walked to the weights' padded depth
      int off = 0;
      bool moved = true;
      while (moved) {
        moved = false;
        for (size_t pj : placed) {
          const PSlot& o = packed[pj];
          const int osize = o.rows * o.ld + o.rows * 16;
          const bool live = !(o.t1 < sl.t0 || sl.t1 < o.t0);
          if (live && off < o.lds + osize && o.lds < off + size) { off = o.lds + osize; moved = true; }
        }
      }
      sl.lds = off;
      placed.push_back(oi);
      top = std::max(top, off + size);
    }
    b0.act_floats = top;
  }
  PBuilder b{plan, dp, first, count};
  b.preset = &packed; b.act_floats = b0.act_floats;
  if (const char* e = fill(b)) return e;
  for (size_t i = 0; i < b.ops.size(); ++i) {           // warm-up links: a GEMM op names the first tap's weights of the next GEMM op
    b.ops[i].pf_ptr = nullptr; b.ops[i].pf_lines = 0;
    if (b.ops[i].kind != OP_GEMM) continue;
    for (size_t j = i + 1; j < b.ops.size(); ++j) {
      if (b.ops[j].kind != OP_GEMM) continue;
      const PixTap& t0 = b.taps[b.ops[j].tap_first];
      const long long lines = (long long)t0.nch * 16 * b.ops[j].Npad / 32;
      b.ops[i].pf_ptr = (t0.src2 ? b.ops[j].Wt2 : b.ops[j].Wt) + t0.w_off; b.ops[i].pf_lines = (int)std::min<long long>(lines, 2 * PNT);
      break;
    }
  }
  if (b.stat_of.size() > (size_t)P_STAT_SLOTS) return "pixel kernel: too many statistics slots";
  if (pix_lds_bytes(b.act_floats, (int)b.taps.size()) > 160 * 1024) return "pixel kernel: the activations do not fit the LDS";
  PProgram p;
  p.n_ops = (int)b.ops.size(); p.act_floats = b.act_floats; p.ntaps = (int)b.taps.size();
  if (hipMalloc(&p.d_taps, sizeof(PixTap) * (b.taps.size() + 1)) != hipSuccess) return "pixel kernel: hipMalloc failed";
  if (hipMemcpy(p.d_taps, b.taps.data(), sizeof(PixTap) * b.taps.size(), hipMemcpyHostToDevice) != hipSuccess) return "pixel kernel: hipMemcpy failed";
  for (PixOpD& d : b.ops) if (d.kind == OP_GEMM) d.taps = p.d_taps;
  if (hipMalloc(&p.d_ops, sizeof(PixOpD) * b.ops.size()) != hipSuccess) return "pixel kernel: hipMalloc failed";
  if (hipMemcpy(p.d_ops, b.ops.data(), sizeof(PixOpD) * b.ops.size(), hipMemcpyHostToDevice) != hipSuccess) return "pixel kernel: hipMemcpy failed";
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pixrun_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  g_pprograms.emplace(key, p);
  return nullptr;
}

}  // namespace

unsigned long long* g_pix_dbg = nullptr;

const char* launch_pixrun(const DevPlan& dp, int first, int count, hipStream_t st) {
  const Plan& plan = *dp.plan;
  const auto key = std::make_pair((const void*)&plan, (long long)first * 4096 + count);
  auto it = g_pprograms.find(key);
  if (it == g_pprograms.end()) {
    if (const char* e = pixrun_build(dp, first, count)) return e;
    it = g_pprograms.find(key);
  }
  const PProgram& p = it->second;
  const dim3 grid(plan.B), block(PNT);
  const size_t ldsb = pix_lds_bytes(p.act_floats, p.ntaps);
  if (g_prof_start) hipExtLaunchKernelGGL(pixrun_kernel, grid, block, ldsb, st, g_prof_start, g_prof_stop, 0, p.d_ops, p.n_ops, plan.B, p.act_floats, p.d_taps, p.ntaps, g_pix_dbg);
  else hipLaunchKernelGGL(pixrun_kernel, grid, block, ldsb, st, p.d_ops, p.n_ops, plan.B, p.act_floats, p.d_taps, p.ntaps, g_pix_dbg);
  return nullptr;
}

int pixrun_program_ops(const DevPlan& dp, int first, int count, int* kinds, int cap) {
  auto it = g_pprograms.find(std::make_pair((const void*)dp.plan, (long long)first * 4096 + count));
  if (it == g_pprograms.end()) return -1;
  std::vector<PixOpD> h(it->second.n_ops);
  if (hipMemcpy(h.data(), it->second.d_ops, sizeof(PixOpD) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int i = 0; i < (int)h.size() && i < cap; ++i) kinds[i] = h[i].kind * 1000 + (h[i].kind == OP_GEMM ? h[i].tap_cnt : 0);
  return (int)h.size();
}

// Build the programs of every maximal run of chain ops (model creation).  Any failure leaves those ops to the chain kernel.
const char* pixrun_prepare(const DevPlan& dp) {
  const Plan& plan = *dp.plan;
  for (int i = 0; i < (int)plan.ops.size();) {
    if (!plan.ops[i].chain || plan.ops[i].time_path) { ++i; continue; }
    int n = 1;
    while (i + n < (int)plan.ops.size() && plan.ops[i + n].chain && !plan.ops[i + n].time_path) ++n;
    if (const char* e = pixrun_build(dp, i, n)) return e;
    i += n;
  }
  return nullptr;
}

void pixrun_release(const Plan* plan) {
  for (auto it = g_pprograms.begin(); it != g_pprograms.end();) {
    if (it->first.first == (const void*)plan) { (void)hipFree(it->second.d_ops); (void)hipFree(it->second.d_taps); it = g_pprograms.erase(it); }
    else ++it;
  }
}

}  // namespace sbm
