// rowrun.hip -- the one-position (1x1) level of the score net as ONE launch of independent workgroups.
//
// Inside the XCD-resident chain kernel (chain.hip) the 25 ops of the 1x1 level -- the bottleneck of the UNet: ConvNeXt blocks
// whose 3x3 convolutions see a single pixel, the folded attention, the down/up-sampling convolutions around them -- cost ~6 us each:
// a cluster barrier plus a chain of dependent L2 round trips for a GEMM of 2-4 MFLOP per sample tile.  The arithmetic is nothing
// (40 MFLOP per 32 samples); the 3.2 MB of weights are.  So this kernel turns the level around: a workgroup owns FOUR samples and
// carries them through every op of the level by itself -- activations live in LDS from the first op to the last, no barrier ever
// leaves the workgroup -- streaming the weights once per workgroup (64 workgroups at B = 256: 8 per XCD, which share the lines in
// that XCD's L2).  The matrix instruction is v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per issue = 4 samples x
// 64 output columns x one k, at the same 64 FLOP/clk/SIMD as the 32x32 tile but with M = 4 instead of 32 rows.
//
//   weights   : transposed copies [tap][k][n] (n contiguous), so a wave reads 512 contiguous bytes per (k, 128-column block):
//               lane l (q = l & 31, phase = l >> 5) loads W[k0 + 4 phase + j][n0 + 4 q .. + 3], j < 4, as four dwordx4
//   activations: LDS image T[k >> 2][sample][k & 3]: lane l reads its phase's four k of sample l & 3 with one ds_read_b128
//   mapping   : D[vgpr v][lane l] = sample v x column n0 + 4 q + i for the i-th of four accumulators; the two 32-lane phases hold
//               disjoint k and meet in the LDS reduction together with the 8 waves' K slices
//   epilogue  : 512 threads = 4 samples x 128 columns; folded GroupNorm, bias, residual, activation, LDS + global store, statistics
//
// Every output is ALSO stored to the workspace in the plan's [P][B][ld] layout (fire and forget): consumers outside the level and
// the op-by-op debug tools find it where the other paths would have put it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "device.h"

namespace sbm {

namespace {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4r gf32x4r;
typedef int i32x4r __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4r const_i32x4r;

constexpr int RNT = 512, RNW = 8;
constexpr int RED_F = RNW * 2 * ROW_R * 128;          // floats: per wave and phase a [sample][128 columns] partial
constexpr int STAT_SLOTS = 64;
constexpr int RING = 4;                                // weight chunks (8 k x 128 columns = 4 dwordx4 per lane) in flight per wave

__device__ __forceinline__ float row_act(int act, float v) {
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

// LDS float offset of element (channel k, sample s) of a tensor slab
__device__ __forceinline__ int tix(int k, int s) { return (k >> 2) * (4 * ROW_R) + s * 4 + (k & 3); }

struct Chunk { f32x4r b[4]; };
// wave-uniform 16-byte table entry through the scalar cache
__device__ __forceinline__ RowEnt ent_load(const RowEnt* p) {
  const i32x4r v = *(const_i32x4r*)(unsigned long long)p;
  return RowEnt{v.x, v.y, v.z, v.w};
}
}  // namespace

__global__ __launch_bounds__(RNT, 2) void rowrun_kernel(const RowOpD* __restrict__ ops, int count, int B, int act_floats, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + act_floats;                        // [RNW][2][ROW_R][128]
  float* stat = red + RED_F;                            // [STAT_SLOTS][ROW_R][2]  (sum, sumsq)
  float* scr = stat + STAT_SLOTS * ROW_R * 2;           // 2 * RNT floats of scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_compute = (B + ROW_R - 1) / ROW_R;
  if ((int)blockIdx.x >= n_compute) {
    // PREFETCH role.  Every parameter of the level is touched once per step and always from beyond L2 (the step streams ~45 MB
    // through a 4 MB L2), and a compute workgroup meets them as a chain of dependent first touches: ~1.5 us each, three per op.
    // The workgroups behind the compute ones (dealt round-robin over the XCDs like them) run through the ops' weight spans in
    // program order and touch one dword per 128-byte line, 8 loads in flight per thread, so that the lines are in THEIR XCD's L2
    // (the level's 3.2 MB fit) when the compute workgroups of that XCD ask.  Pure hint: results never depend on it.
    const int pw = (int)blockIdx.x - n_compute, npw = (int)gridDim.x - n_compute;
    const int xslot = pw >> 3, per_xcd = (npw + 7) >> 3;             // workgroups pw, pw + 8, ... share an XCD: split the lines among them
    unsigned sink = 0;
    for (int k = 0; k < count; ++k) {
      const RowOpD& op = ops[k];
      if (op.kind == OP_DWCONV7 || op.kind == OP_GN_RES) {       // the small vectors of the element-wise ops (one line per 32 channels)
        const int C = op.kind == OP_DWCONV7 ? op.dw.C : op.gr.C;
        const float* v0 = op.kind == OP_DWCONV7 ? op.dw.w + 24 * C : op.gr.gamma;
        const float* v1 = op.kind == OP_DWCONV7 ? op.dw.bias : op.gr.beta;
        const float* v2 = (op.kind == OP_DWCONV7 && op.dw.ctl && op.dw.ctl->table_mode) ? dw_cond_row(op.dw, 0) : nullptr;
        const int L = xslot * RNT + tid;
        if (xslot == 0 && L < (C + 31) / 32) { sink ^= __float_as_uint(v0[L * 32]) ^ __float_as_uint(v1[L * 32]); if (v2) sink ^= __float_as_uint(v2[L * 32]); }
        continue;
      }
      if (op.kind != OP_GEMM) continue;
      for (int sI = 0; sI < 7; ++sI) {
        const float* base = op.pf_ptr[sI];
        const int lines = op.pf_lines[sI];
        for (int L0 = xslot * RNT + tid; L0 < lines; L0 += per_xcd * RNT * 8) {
          unsigned v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int L = L0 + u * per_xcd * RNT; v[u] = L < lines ? __float_as_uint(base[(long long)L * 32]) : 0u; }
#pragma unroll
          for (int u = 0; u < 8; ++u) sink ^= v[u];
        }
      }
    }
    if (sink == 0x9e3779b9u && dbg) dbg[255] = 1;                    // consumes the touched words
    return;
  }
  const int b0 = blockIdx.x * ROW_R;
  const int ph = lane >> 5, q = lane & 31, ls = lane & 3;
  // K is walked in chunks of 8 up to the weights' padded depth (zero rows): the activation image must hold finite values there
  for (int i = tid; i < act_floats + RED_F; i += RNT) lds[i] = 0.f;
  __syncthreads();
  for (int k = 0; k < count; ++k) {
    const RowOpD& op = ops[k];
    const int kind = op.kind;
    if (dbg && blockIdx.x == 0 && tid == 0) dbg[k] = __builtin_amdgcn_s_memtime();          // diagnostics: op k starts (workgroup 0)
    if (kind == ROW_STAGE) {
      // global [P][B][ld] tensor (C channels of the view) -> LDS slot, for inputs produced outside this launch
      const int C4 = op.st_C >> 2;
      for (int idx = tid; idx < op.st_P * C4 * ROW_R; idx += RNT) {
        const int s = idx % ROW_R, r = idx / ROW_R, c4 = r % C4, p = r / C4;
        const int b = min(b0 + s, B - 1);
        const float4 v = *reinterpret_cast<const float4*>(op.st_src + ((long long)p * B + b) * op.st_ld + 4 * c4);
        *reinterpret_cast<float4*>(lds + op.st_lds + p * op.st_pix + c4 * (4 * ROW_R) + s * 4) = v;
      }
      if (op.st_stat >= 0 && tid < ROW_R) {             // and its GroupNorm partials (producer outside: sum them)
        const int b = min(b0 + tid, B - 1);
        double s = 0.0, qq = 0.0;
        for (int i = 0; i < op.st_nparts; ++i) { const float2 v = op.st_part[(long long)i * B + b]; s += (double)v.x; qq += (double)v.y; }
        stat[(op.st_stat * ROW_R + tid) * 2] = (float)s; stat[(op.st_stat * ROW_R + tid) * 2 + 1] = (float)qq;
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_DWCONV7) {
      // one position: the 7x7 depth-wise window sees the centre tap only -> h = w[3][3] * x + bias + time condition
      const DwArgs& d = op.dw;
      const int C4 = d.C >> 2;
      float s = 0.f, qq = 0.f;
      for (int idx = tid; idx < C4 * ROW_R; idx += RNT) {
        const int sm = idx % ROW_R, c4 = idx / ROW_R, b = min(b0 + sm, B - 1);
        const float4 x = *reinterpret_cast<const float4*>(lds + op.in_lds + c4 * (4 * ROW_R) + sm * 4);
        const float4 w = *reinterpret_cast<const float4*>(d.w + 24 * d.C + 4 * c4);
        const float4 bi = *reinterpret_cast<const float4*>(d.bias + 4 * c4);
        float4 h = make_float4(fmaf(x.x, w.x, 0.f) + bi.x, fmaf(x.y, w.y, 0.f) + bi.y, fmaf(x.z, w.z, 0.f) + bi.z, fmaf(x.w, w.w, 0.f) + bi.w);
        if (d.cond) { const float4 cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + 4 * c4); h.x += cv.x; h.y += cv.y; h.z += cv.z; h.w += cv.w; }
        *reinterpret_cast<float4*>(lds + op.out_lds + c4 * (4 * ROW_R) + sm * 4) = h;
        if (b0 + sm < B) *reinterpret_cast<float4*>(d.out + (long long)(b0 + sm) * d.ldo + 4 * c4) = h;
        s += (h.x + h.y) + (h.z + h.w); qq += (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
      }
      // statistics per sample: thread t holds sample t % ROW_R (RNT is a multiple of ROW_R): shuffle tree over the lanes of equal
      // sample, then the eight waves through LDS in fixed order
#pragma unroll
      for (int m = ROW_R; m < 64; m <<= 1) { s += __shfl_xor(s, m); qq += __shfl_xor(qq, m); }
      if (lane < ROW_R) { scr[wave * ROW_R + lane] = s; scr[RNT + wave * ROW_R + lane] = qq; }
      __syncthreads();
      if (tid < ROW_R) {
        float ss = 0.f, q2 = 0.f;
        for (int w = 0; w < RNW; ++w) { ss += scr[w * ROW_R + tid]; q2 += scr[RNT + w * ROW_R + tid]; }
        if (op.stat_out >= 0) { stat[(op.stat_out * ROW_R + tid) * 2] = ss; stat[(op.stat_out * ROW_R + tid) * 2 + 1] = q2; }
        if (d.stat && b0 + tid < B) d.stat[b0 + tid] = make_float2(ss, q2);
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_GN_RES) {
      // y = GroupNorm(1,C)(u) * gamma + beta + x   (LinearAttention tail + Residual) on one position
      const GnResArgs& g = op.gr;
      const int C4 = g.C >> 2;
      for (int idx = tid; idx < C4 * ROW_R; idx += RNT) {
        const int sm = idx % ROW_R, c4 = idx / ROW_R;
        const float su = stat[(op.gn_stat * ROW_R + sm) * 2], sq = stat[(op.gn_stat * ROW_R + sm) * 2 + 1];
        const double mean_d = (double)su * (double)g.inv_count;
        double var = (double)sq * (double)g.inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
        const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float4 u = *reinterpret_cast<const float4*>(lds + op.in_lds + c4 * (4 * ROW_R) + sm * 4);
        const float4 x = *reinterpret_cast<const float4*>(lds + op.res_lds + c4 * (4 * ROW_R) + sm * 4);
        const float4 ga = *reinterpret_cast<const float4*>(g.gamma + 4 * c4), be = *reinterpret_cast<const float4*>(g.beta + 4 * c4);
        float4 y;
        y.x = ((u.x - mean) * rstd) * ga.x + be.x + x.x; y.y = ((u.y - mean) * rstd) * ga.y + be.y + x.y;
        y.z = ((u.z - mean) * rstd) * ga.z + be.z + x.z; y.w = ((u.w - mean) * rstd) * ga.w + be.w + x.w;
        *reinterpret_cast<float4*>(lds + op.out_lds + c4 * (4 * ROW_R) + sm * 4) = y;
        if (b0 + sm < B) *reinterpret_cast<float4*>(g.out + (long long)(b0 + sm) * g.ldo + 4 * c4) = y;
      }
      __syncthreads();
      continue;
    }
    // ---------------------------------------------------------------- GEMM
    const bool gn = op.gn_stat >= 0;
    float inv_r = 1.f;                                  // 1 / rstd of this lane's sample: source-2 rows are pre-divided (see chain.hip)
    if (gn) {
      const float su = stat[(op.gn_stat * ROW_R + ls) * 2], sq = stat[(op.gn_stat * ROW_R + ls) * 2 + 1];
      const double mean_d = (double)su * (double)op.gn_inv;
      double var = (double)sq * (double)op.gn_inv - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
      inv_r = sqrtf((float)var + 1e-5f);
    }
    const int ncb = (op.nstore + 127) >> 7;
    const float* Wt = op.Wt; const float* Wt2 = op.Wt2; const int npad = op.Npad;
    // A pass covers one or two 128-column blocks: with two, waves 0-3 take the first and waves 4-7 the second, each splitting K
    // four ways (N = 256: one K loop, one pair of barriers per layer instead of two).
    for (int po = 0; po < op.P_out; ++po) {
      const RowEnt* ent = op.ent + op.ent_first[po];
      const int ne = op.ent_cnt[po];
      float sacc = 0.f, qacc = 0.f;                     // statistics of this thread's outputs over the column blocks
      for (int cb0 = 0; cb0 < ncb; cb0 += 2) {
        const int par = ncb - cb0 >= 2 ? 2 : 1;         // column blocks in this pass
        const int ksplit = RNW / par;                   // waves per block = K slices
        const int wblk = par == 2 ? (wave >> 2) : 0, wk = par == 2 ? (wave & 3) : wave;
        const int cb = cb0 + wblk;
        const int col = min(cb * 128 + 4 * q, op.Npad - 4);
        // epilogue operands of this thread's outputs (sample ev, column ec of block cb0 + j): asked for before the K loop
        const int ev = tid >> 7, ec = tid & 127;
        float e_bias[2] = {0.f, 0.f}, e_gs[2] = {0.f, 0.f}, e_gb[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = (cb0 + j) * 128 + ec;
          if (j < par && n < op.nstore) {
            if (op.bias) e_bias[j] = op.bias[n];
            if (gn) { e_gs[j] = op.gn_s[(long long)po * op.Npad + n]; e_gb[j] = op.gn_b1[(long long)po * op.Npad + n]; }
          }
        }
        f32x4r acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4r{0.f, 0.f, 0.f, 0.f};
        // this wave's chunk list: entries in order, chunks wk, wk + ksplit, ... of each (chunks do not carry across entries);
        // RING of them in flight.  The loads are ordinary (compiler-tracked) loads: an asm-issued ring would put in-flight
        // destinations through the phi copies of this loop's control flow, i.e. copy registers the data has not reached yet.
        Chunk ring[RING];
        struct Cur { int e, c; RowEnt en; };
        auto cur_init = [&](Cur& u) { u.e = 0; u.c = wk; if (ne > 0) u.en = ent_load(ent); };
        auto cur_fix = [&](Cur& u) {                   // move to the next entry that has a chunk for this wave
          while (u.e < ne && u.c >= u.en.nch) { ++u.e; u.c = wk; if (u.e < ne) u.en = ent_load(ent + u.e); }
          return u.e < ne;
        };
        int nk = 0;                                     // chunks of this wave in this pass
        for (int e = 0; e < ne; ++e) { const int nch = ent_load(ent + e).nch; nk += nch > wk ? (nch - wk + ksplit - 1) / ksplit : 0; }
        Cur L, C;
        cur_init(L); cur_init(C);
        auto issue = [&](Chunk& ck) {
          cur_fix(L);
          const float* w = (L.en.src2 ? Wt2 : Wt) + L.en.w_off + (long long)(8 * L.c + 4 * ph) * npad + col;
#pragma unroll
          for (int j = 0; j < 4; ++j) ck.b[j] = *(const gf32x4r*)(unsigned long long)(w + (long long)j * npad);     // global_load (vmcnt only), not flat
          L.c += ksplit;
        };
        auto consume = [&](const Chunk& ck) {
          cur_fix(C);
          float4 a = *reinterpret_cast<const float4*>(lds + C.en.a_lds + (2 * C.c + ph) * (4 * ROW_R) + ls * 4);
          if (gn && C.en.src2) { a.x *= inv_r; a.y *= inv_r; a.z *= inv_r; a.w *= inv_r; }
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], ck.b[j][i], acc[i], 0, 0, 0);
          }
          C.c += ksplit;
        };
#pragma unroll
        for (int r = 0; r < RING; ++r) if (r < nk) issue(ring[r]);
        for (int i = 0; i < nk; i += RING) {
#pragma unroll
          for (int r = 0; r < RING; ++r) {
            if (i + r < nk) {
              consume(ring[r]);
              if (i + r + RING < nk) issue(ring[r]);
            }
          }
        }
        // partials -> LDS: red[wave][phase][sample v][column 4 q + i]   (waves 4-7 hold the second block when par == 2)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          *reinterpret_cast<float4*>(red + ((wave * 2 + ph) * ROW_R + v) * 128 + 4 * q) = make_float4(acc[0][v], acc[1][v], acc[2][v], acc[3][v]);
        __syncthreads();
        // reduce + epilogue: thread = (sample ev, column ec) of each block of the pass
        float mean = 0.f, rstd = 1.f;
        if (gn) {
          const float su = stat[(op.gn_stat * ROW_R + ev) * 2], sq = stat[(op.gn_stat * ROW_R + ev) * 2 + 1];
          const double mean_d = (double)su * (double)op.gn_inv;
          double var = (double)sq * (double)op.gn_inv - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
          mean = (float)mean_d; rstd = rsqrtf((float)var + 1e-5f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j >= par) break;
          const int n = (cb0 + j) * 128 + ec;
          const int w0 = par == 2 ? j * 8 : 0, nw = par == 2 ? 8 : 16;          // (wave, phase) partial slabs of this block
          float t = 0.f;
          for (int w = 0; w < nw; ++w) t += red[((w0 + w) * ROW_R + ev) * 128 + ec];
          const bool ok = n < op.nstore;
          float o = 0.f;
          if (ok) {
            if (gn) t = t * rstd + (e_gb[j] - (mean * rstd) * e_gs[j]);
            t += e_bias[j];
            if (op.res_lds >= 0) t += lds[op.res_lds + op.res_p2[po] * op.res_pix + tix(n, ev)];
            o = row_act(op.act, t);
            if (op.out_lds >= 0) lds[op.out_lds + po * op.out_pix + tix(n, ev)] = o;
            if (b0 + ev < B) op.out_g[((long long)po * B + b0 + ev) * op.ldo + n] = o;
            sacc += o; qacc += o * o;
          }
          if (op.stat_g) {                              // the plan's partial layout: one (sum, sumsq) per 32-column block
            float s32 = ok ? o : 0.f, q32 = ok ? o * o : 0.f;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) { s32 += __shfl_xor(s32, m); q32 += __shfl_xor(q32, m); }
            const int nb = n >> 5;
            if ((ec & 31) == 0 && nb < op.nblk && b0 + ev < B) op.stat_g[((long long)po * op.nblk + nb) * B + b0 + ev] = make_float2(s32, q32);
          }
        }
        __syncthreads();                                // red is reused by the next pass
      }
      if (op.stat_out >= 0) {
        // per-sample statistics of the whole output tensor (all pixels): wave reduction (fixed shuffle tree), then the two waves of a
        // sample through LDS, in fixed order
        float ss = sacc, q2 = qacc;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { ss += __shfl_xor(ss, m); q2 += __shfl_xor(q2, m); }
        if (lane == 0) { scr[wave] = ss; scr[RNW + wave] = q2; }
        __syncthreads();
        if (tid < ROW_R) {
          const float a0 = scr[2 * tid] + scr[2 * tid + 1], a1 = scr[RNW + 2 * tid] + scr[RNW + 2 * tid + 1];
          float* st = stat + (op.stat_out * ROW_R + tid) * 2;
          if (po == 0) { st[0] = a0; st[1] = a1; } else { st[0] += a0; st[1] += a1; }
        }
        __syncthreads();
      }
    }
  }
  if (dbg && blockIdx.x == 0 && tid == 0) dbg[count] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------------------------------------
// host: build the program of one run of row ops, launch it
// ------------------------------------------------------------------------------------------------
namespace {

struct Slot { long long base; int ld; int P; int lds; int t0, t1; };   // workspace buffer <-> LDS slab; [t0, t1] = program ops that touch it

struct Builder {
  const Plan& plan; const DevPlan& dp; int first, count;
  std::vector<Slot> slots; int act_floats = 0;
  std::vector<RowOpD> ops; std::vector<RowEnt> ents;
  std::map<long long, int> stat_of;                                  // workspace stat offset -> LDS stat slot
  std::map<long long, int> produced;                                 // slot index -> produced inside the run
  std::string err;
  const std::vector<Slot>* preset = nullptr;                         // second pass: slabs packed by live interval (see rowrun_build)
  std::vector<int> touched;                                          // slabs the op under construction refers to

  // the LDS slab of a view (allocated on first sight); returns LDS float offset of (pixel 0, channel 0 of the view)
  int slab(const View& v, int* pix_stride, int* slot_index) {
    const int P = v.H * v.W;
    for (size_t i = 0; i < slots.size(); ++i) {
      Slot& s = slots[i];
      if (s.ld == v.ld && s.P == P && v.off >= s.base && v.off < s.base + s.ld) {
        *pix_stride = s.ld * ROW_R; if (slot_index) *slot_index = (int)i;
        touched.push_back((int)i);
        return s.lds + (int)(v.off - s.base) * ROW_R;
      }
    }
    // a view into the middle of a buffer seen first (second half of a concat): the buffer starts at a multiple of its row
    long long base = v.off;
    // concat halves: off = buffer + pad8(co); recover the buffer start from the other views of the plan with this ld
    for (int i = 0; i < (int)plan.ops.size(); ++i) {
      const Op& o = plan.ops[i];
      auto look = [&](const View& w) { if (w.ld == v.ld && w.H * w.W == P && w.off < base && base - w.off < v.ld) base = w.off; };
      if (o.kind == OP_GEMM) { look(o.g.a); look(o.g.out); if (o.g.has2) look(o.g.a2); if (o.g.has_res) look(o.g.r); }
      else if (o.kind == OP_DWCONV7) { look(o.dw.in); look(o.dw.out); }
      else if (o.kind == OP_GN_RES) { look(o.gr.u); look(o.gr.x); look(o.gr.out); }
    }
    Slot s{base, v.ld, P, act_floats, (int)ops.size(), (int)ops.size()};
    if (preset) {
      for (const Slot& ps : *preset) if (ps.base == base && ps.ld == v.ld && ps.P == P) s.lds = ps.lds;
    } else {
      act_floats += P * v.ld * ROW_R;
    }
    slots.push_back(s);
    touched.push_back((int)slots.size() - 1);
    *pix_stride = s.ld * ROW_R; if (slot_index) *slot_index = (int)slots.size() - 1;
    return s.lds + (int)(v.off - s.base) * ROW_R;
  }
  bool is_produced(const View& v) {
    int ps, si; slab(v, &ps, &si);
    return produced.count(si) != 0;
  }
  void mark_produced(const View& v) { int ps, si; slab(v, &ps, &si); produced[si] = 1; }
  int stat_slot(long long off) {
    auto it = stat_of.find(off);
    if (it != stat_of.end()) return it->second;
    const int s = (int)stat_of.size();
    stat_of[off] = s;
    return s;
  }
  // make sure the tensor behind `v` is in LDS: produced earlier in this run, or staged from the workspace now
  int need(const View& v, int* pix, long long stat_off = -1, int nparts = 0) {
    int si; const int off = slab(v, pix, &si);
    const bool have_stat = stat_off < 0 || stat_of.count(stat_off);
    if (produced.count(si) && have_stat) return off;
    RowOpD st{}; st.kind = ROW_STAGE;
    const Slot& s = slots[si];
    st.st_src = dp.ws + s.base; st.st_ld = s.ld; st.st_C = s.ld; st.st_P = s.P; st.st_lds = s.lds; st.st_pix = s.ld * ROW_R; st.st_stat = -1;
    if (produced.count(si)) { st.st_P = 0; }                        // only the statistics are missing
    if (stat_off >= 0 && !stat_of.count(stat_off)) {
      st.st_stat = stat_slot(stat_off); st.st_part = reinterpret_cast<const float2*>(dp.ws + stat_off); st.st_nparts = nparts;
    }
    ops.push_back(st);
    produced[si] = 1;
    return off;
  }
};

struct Program { RowOpD* d_ops = nullptr; RowEnt* d_ents = nullptr; int n_ops = 0; int act_floats = 0; };
std::map<std::pair<const void*, long long>, Program> g_programs;   // (plan, first * 4096 + count) -> uploaded program
std::mutex g_programs_mu;      // the map is process-wide (handles of different threads share it): every access holds this

}  // namespace

size_t rowrun_lds_bytes(int act_floats) { return sizeof(float) * ((size_t)act_floats + RED_F + STAT_SLOTS * ROW_R * 2 + 2 * RNT); }

// Build the program of ops [first, first + count) of the plan, all marked op.row (allocates and uploads: not capturable).
static const char* rowrun_build(const DevPlan& dp, int first, int count) {
  const Plan& plan = *dp.plan;
  const auto key = std::make_pair((const void*)&plan, (long long)first * 4096 + count);
  std::lock_guard<std::mutex> lock(g_programs_mu);
  if (g_programs.count(key)) return nullptr;
  {
    auto fill = [&](Builder& b) -> const char* {
    for (int i = first; i < first + count; ++i) {
      const Op& op = plan.ops[i];
      if (!op.row) return "row kernel: op is not marked for it";
      b.touched.clear();
      RowOpD d{};
      d.kind = op.kind; d.gn_stat = -1; d.stat_out = -1; d.out_lds = -1; d.res_lds = -1; d.in_lds = -1;
      int pix;
      if (op.kind == OP_DWCONV7) {
        const DwOp& o = op.dw;
        if (o.in.H * o.in.W != 1) return "row kernel: depth-wise op over more than one position";
        d.in_lds = b.need(o.in, &pix);
        d.out_lds = b.slab(o.out, &pix, nullptr); b.mark_produced(o.out);
        d.dw = make_dw_args(plan, o, dp.prm, dp.ws, dp.ctl);
        if (o.stat_off >= 0) d.stat_out = b.stat_slot(o.stat_off);
      } else if (op.kind == OP_GN_RES) {
        const GnResOp& o = op.gr;
        if (o.u.H * o.u.W != 1 || o.out_ext) return "row kernel: unsupported GroupNorm+residual op";
        d.in_lds = b.need(o.u, &pix, o.part_off, o.nparts);
        d.res_lds = b.need(o.x, &pix);
        d.gn_stat = b.stat_slot(o.part_off);
        d.out_lds = b.slab(o.out, &pix, nullptr); b.mark_produced(o.out);
        d.gr = make_gnres_args(o, dp.prm, dp.ws);
      } else if (op.kind == OP_GEMM) {
        const GemmOp& g = op.g;
        const int P_out = g.Hout * g.Wout, P_in = g.a.H * g.a.W;
        if (P_out > 4 || P_in > 4 || g.a_ext || g.out_nchw || g.w_pix_stride || g.bias_pix_stride || g.a_bcast || g.a_rows || g.wt_off < 0) return "row kernel: unsupported GEMM op";
        if ((g.Kp & 7) || (g.has2 && (g.Kp2 & 7)) || (g.Npad & 3)) return "row kernel: K must be a multiple of 8";
        int apix; const int a_lds = b.need(g.a, &apix, g.gn ? g.gn_part_off : -1, g.gn_nparts);
        int a2pix = 0, a2_lds = -1;
        if (g.has2) a2_lds = b.need(g.a2, &a2pix);
        int rpix = 0;
        if (g.has_res) { d.res_lds = b.need(g.r, &rpix); d.res_pix = rpix; }
        d.P_out = P_out; d.Npad = g.Npad; d.nstore = g.out.C; d.act = g.act; d.ldo = g.out.ld;
        d.Wt = dp.prm + g.wt_off; d.Wt2 = g.has2 ? dp.prm + g.wt2_off : nullptr;
        d.bias = g.bias_off >= 0 ? dp.prm + g.bias_off : nullptr;
        if (g.gn) { d.gn_stat = b.stat_slot(g.gn_part_off); d.gn_inv = g.gn_inv_count; d.gn_s = dp.prm + g.gn_s_off; d.gn_b1 = dp.prm + g.gn_b1_off; }
        d.out_g = dp.ws + g.out.off;
        int opix; d.out_lds = b.slab(g.out, &opix, nullptr); d.out_pix = opix; b.mark_produced(g.out);
        if (g.stat_off >= 0) { d.stat_out = b.stat_slot(g.stat_off); d.stat_g = reinterpret_cast<float2*>(dp.ws + g.stat_off); d.nblk = (g.out.C + 31) / 32; }
        for (int po = 0; po < P_out; ++po) {
          const int oy = po / g.Wout, ox = po % g.Wout;
          d.ent_first[po] = (int)b.ents.size();
          for (int ky = 0; ky < g.KH; ++ky) for (int kx = 0; kx < g.KW; ++kx) {
            const int qp = tap_source_pixel(oy, ox, ky, kx, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W);
            if (qp < 0) continue;
            b.ents.push_back(RowEnt{(int)((long long)(ky * g.KW + kx) * g.Kp * g.Npad), a_lds + qp * apix, g.Kp / 8, 0});
          }
          const int w2 = g.has2 ? g.a2.W : (g.has_res ? g.r.W : 1);
          const int p2 = (oy / g.up2) * w2 + (ox / g.up2);
          d.res_p2[po] = p2;
          if (g.has2) b.ents.push_back(RowEnt{0, a2_lds + p2 * a2pix, g.Kp2 / 8, 1});
          d.ent_cnt[po] = (int)b.ents.size() - d.ent_first[po];
        }
      } else {
        return "row kernel: op kind not supported";
      }
      for (int si : b.touched) b.slots[si].t1 = std::max(b.slots[si].t1, (int)b.ops.size());   // live until this op (staging ops sit before it)
      b.ops.push_back(d);
    }
    return nullptr;
    };
    // pass 1: every tensor of the run gets a slab of its own, which records the program ops that touch it; then the slabs are packed
    // first-fit by live interval (two tensors whose intervals do not overlap may share LDS: ops run one after the other, a workgroup
    // barrier between them), and pass 2 rebuilds the descriptors on the packed offsets.
    Builder b0{plan, dp, first, count};
    if (const char* e = fill(b0)) return e;
    std::vector<Slot> packed = b0.slots;
    {
      std::vector<size_t> order(packed.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return packed[x].t0 < packed[y].t0; });
      std::vector<size_t> placed;
      int top = 0;
      for (size_t oi : order) {
        Slot& sl = packed[oi];
        const int size = sl.P * sl.ld * ROW_R + 32 * ROW_R;            // + slack: a slab is read up to the padded K behind its last channel
        int off = 0;
        bool moved = true;
        while (moved) {
          moved = false;
          for (size_t pj : placed) {
            const Slot& o = packed[pj];
            const int osize = o.P * o.ld * ROW_R + 32 * ROW_R;
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
    Builder b{plan, dp, first, count};
    b.preset = &packed; b.act_floats = b0.act_floats;
    if (const char* e = fill(b)) return e;
    // prefetch links: every op names the next GEMM op behind it; a GEMM op lists the weight spans it will stream
    for (size_t i = 0; i < b.ops.size(); ++i) {
      RowOpD& d = b.ops[i];
      d.pf_next = -1;
      for (int sI = 0; sI < 7; ++sI) { d.pf_ptr[sI] = nullptr; d.pf_lines[sI] = 0; }
      if (d.kind != OP_GEMM) continue;
      std::vector<std::pair<const float*, int>> spans;
      for (int po = 0; po < d.P_out; ++po)
        for (int e = d.ent_first[po]; e < d.ent_first[po] + d.ent_cnt[po]; ++e) {
          const RowEnt& en = b.ents[e];
          const float* base = (en.src2 ? d.Wt2 : d.Wt) + en.w_off;
          bool seen = false;
          for (auto& sp : spans) seen |= sp.first == base;
          if (!seen && spans.size() < 5) spans.push_back({base, en.nch * 8 * d.Npad / 32});
        }
      for (size_t sI = 0; sI < spans.size(); ++sI) { d.pf_ptr[sI] = spans[sI].first; d.pf_lines[sI] = spans[sI].second; }
      // the epilogue's vectors: bias and the folded-GroupNorm tables [P_out][Npad], one after the other in the arena as emitted
      if (d.gn_s) { d.pf_ptr[5] = d.gn_s; d.pf_lines[5] = (d.P_out * d.Npad + 31) / 32; d.pf_ptr[6] = d.gn_b1; d.pf_lines[6] = (d.P_out * d.Npad + 31) / 32; }
      else if (d.bias) { d.pf_ptr[5] = d.bias; d.pf_lines[5] = (d.Npad + 31) / 32; }
    }
    for (size_t i = 0; i < b.ops.size(); ++i)
      for (size_t j = i + 1; j < b.ops.size(); ++j) if (b.ops[j].kind == OP_GEMM) { b.ops[i].pf_next = (int)j; break; }
    if (b.stat_of.size() > (size_t)STAT_SLOTS) return "row kernel: too many statistics slots";
    if (rowrun_lds_bytes(b.act_floats) > 160 * 1024) return "row kernel: the level's activations do not fit the LDS";
    Program p;
    p.n_ops = (int)b.ops.size(); p.act_floats = b.act_floats;
    if (hipMalloc(&p.d_ents, sizeof(RowEnt) * (b.ents.size() + 1)) != hipSuccess) return "row kernel: hipMalloc failed";
    if (hipMemcpy(p.d_ents, b.ents.data(), sizeof(RowEnt) * b.ents.size(), hipMemcpyHostToDevice) != hipSuccess) return "row kernel: hipMemcpy failed";
    for (RowOpD& d : b.ops) if (d.kind == OP_GEMM) d.ent = p.d_ents;
    if (hipMalloc(&p.d_ops, sizeof(RowOpD) * b.ops.size()) != hipSuccess) return "row kernel: hipMalloc failed";
    if (hipMemcpy(p.d_ops, b.ops.data(), sizeof(RowOpD) * b.ops.size(), hipMemcpyHostToDevice) != hipSuccess) return "row kernel: hipMemcpy failed";
    ensure_dynamic_lds(reinterpret_cast<const void*>(rowrun_kernel), 160 * 1024);
    g_programs.emplace(key, p);
  }
  return nullptr;
}

// Launch ops [first, first + count) (one rowrun_kernel).  The programs of the plan's own runs exist since model creation; a debug
// sub-range is built on first use (synchronous: never under a stream capture).
const char* launch_rowrun(const DevPlan& dp, int first, int count, hipStream_t st) {
  const Plan& plan = *dp.plan;
  const auto key = std::make_pair((const void*)&plan, (long long)first * 4096 + count);
  Program p;
  for (int attempt = 0;; ++attempt) {
    {
      std::lock_guard<std::mutex> lock(g_programs_mu);
      auto it = g_programs.find(key);
      if (it != g_programs.end()) { p = it->second; break; }
    }
    if (attempt) return "row kernel: program missing after build";
    if (const char* e = rowrun_build(dp, first, count)) return e;
  }
  const int n_compute = (plan.B + ROW_R - 1) / ROW_R;
  const int n_prefetch = n_compute >= 8 ? 16 : 0;                    // two per XCD behind the compute workgroups (pure hint)
  const dim3 grid(n_compute + n_prefetch), block(RNT);
  const size_t ldsb = rowrun_lds_bytes(p.act_floats);
  if (g_prof_start) hipExtLaunchKernelGGL(rowrun_kernel, grid, block, ldsb, st, g_prof_start, g_prof_stop, 0, p.d_ops, p.n_ops, plan.B, p.act_floats, g_row_dbg);
  else hipLaunchKernelGGL(rowrun_kernel, grid, block, ldsb, st, p.d_ops, p.n_ops, plan.B, p.act_floats, g_row_dbg);
  return nullptr;
}

unsigned long long* g_row_dbg = nullptr;     // diagnostics only (sbm_debug_row_stamps)
int rowrun_program_ops(const DevPlan& dp, int first, int count, int* kinds, int cap) {
  std::lock_guard<std::mutex> lock(g_programs_mu);
  auto it = g_programs.find(std::make_pair((const void*)dp.plan, (long long)first * 4096 + count));
  if (it == g_programs.end()) return -1;
  std::vector<RowOpD> h(it->second.n_ops);
  if (hipMemcpy(h.data(), it->second.d_ops, sizeof(RowOpD) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int i = 0; i < (int)h.size() && i < cap; ++i) kinds[i] = h[i].kind * 1000 + (h[i].kind == OP_GEMM ? h[i].P_out * ((h[i].nstore + 127) / 128) : 0);
  return (int)h.size();
}

// Build the programs of every maximal run of row ops (model creation: never inside a stream capture).
const char* rowrun_prepare(const DevPlan& dp) {
  const Plan& plan = *dp.plan;
  for (int i = 0; i < (int)plan.ops.size();) {
    if (!plan.ops[i].row) { ++i; continue; }
    int n = 1;
    while (i + n < (int)plan.ops.size() && plan.ops[i + n].row) ++n;
    if (const char* e = rowrun_build(dp, i, n)) return e;
    i += n;
  }
  return nullptr;
}

// release the programs of a plan (model destruction)
void rowrun_release(const Plan* plan) {
  std::lock_guard<std::mutex> lock(g_programs_mu);
  for (auto it = g_programs.begin(); it != g_programs.end();) {
    if (it->first.first == (const void*)plan) { (void)hipFree(it->second.d_ops); (void)hipFree(it->second.d_ents); it = g_programs.erase(it); }
    else ++it;
  }
}

}  // namespace sbm
