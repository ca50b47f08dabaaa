// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the plan ops of plan.h, and their launcher.
//
// Dominant kernel: seg_gemm<BM,BN> -- an fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM convolution
// over the PBC layout: one workgroup computes out[p][m0:m0+BM][n0:n0+BN] by walking the VALID taps
// of output pixel p (zero-padding taps are skipped) and, per tap, the input channels in chunks of 32.
// A (activations) and B (weights) chunks are row-major with K contiguous, staged global -> VGPR ->
// LDS with a one-chunk register prefetch, GroupNorm(1,C) applied on the fly to A while it passes
// through registers, bias / residual / activation / GroupNorm partial statistics fused in the epilogue.
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include <utility>

#include <cstdint>
#include <cstdio>

#include "plan.h"
#include "device.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <cstring>
#include <vector>

// Profiling hook: when an event pair is armed (sbm_model_profile), the next launch carries it on its own dispatch packet, so
// hipEventElapsedTime(start, stop) is the kernel's begin->end time (what rocprofv3 --kernel-trace reports), free of the
// dispatch latency that a hipEventRecord pair around the launch would include.
namespace sbm {
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
void ensure_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, dev})) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.insert({fn, dev});
}

void arm_profile_events(hipEvent_t start, hipEvent_t stop) { g_prof_start = start; g_prof_stop = stop; }
}
#define SBM_KLAUNCH(kern, grid, block, shmem, st, ...)                                                                      \
  do {                                                                                                                     \
    if (sbm::g_prof_start) hipExtLaunchKernelGGL(kern, grid, block, shmem, st, sbm::g_prof_start, sbm::g_prof_stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                                                    \
  } while (0)

namespace sbm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DEVINL __device__ __forceinline__

DEVINL float act_apply(int act, float v) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + sbm_erf(v * 0.70710678118654752440f));
    case ACT_LRELU02: return v >= 0.f ? v : 0.2f * v;
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ACT_SILU: return v / (1.0f + expf(-v));
    case ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));      // nn.Softplus(beta=1, threshold=20)
    case ACT_LRELU01: return v >= 0.f ? v : 0.1f * v;
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------------
// seg_gemm  (LDS-DMA ring, host-built chunk tables)
// ------------------------------------------------------------------------------------------------
constexpr int BK = 32;        // K chunk = one 128-byte LDS row per operand row

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4 const_i32x4;   // constant address space: uniform loads become s_load

// Wave-uniform 16-byte table read through the scalar cache (SMEM counts on lgkmcnt, never on vmcnt, so it cannot
// make hipcc drain the LDS-DMA ring the way an ordinary global_load would).
DEVINL int4 scalar_load16(const int4* p) {
  const i32x4 v = *(const_i32x4*)(unsigned long long)p;
  return make_int4(v.x, v.y, v.z, v.w);
}

// Fragment reads are inline asm: hipcc must not see an LDS read while LDS-DMA is in flight, or it drains the
// whole ring (s_waitcnt vmcnt(0)) in front of it.  `lds_issue` starts N ds_read_b128, `lds_wait` is the
// matching s_waitcnt lgkmcnt(0) that names every destination "+v" (cdna_hip_programming.md 5.7 item 1, form (ii)),
// so that the MFMAs consuming them cannot be scheduled above the wait and the next group's reads can be
// issued before the current group's MFMAs.  Outputs are early-clobber: a destination must not share a register
// with the address of a later read of the same statement (the data may land before that read issues).
template <int N> DEVINL void lds_issue(f32x4* o, const unsigned* a);
template <> DEVINL void lds_issue<4>(f32x4* o, const unsigned* a) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
}
template <> DEVINL void lds_issue<6>(f32x4* o, const unsigned* a) {
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]) : "memory");
}
template <> DEVINL void lds_issue<8>(f32x4* o, const unsigned* a) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
               "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}
template <> DEVINL void lds_issue<10>(f32x4* o, const unsigned* a) { lds_issue<4>(o, a); lds_issue<6>(o + 4, a + 4); }
template <> DEVINL void lds_issue<12>(f32x4* o, const unsigned* a) { lds_issue<6>(o, a); lds_issue<6>(o + 6, a + 6); }

template <int N> DEVINL void lds_wait(f32x4* o);
template <> DEVINL void lds_wait<4>(f32x4* o) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])::"memory"); }
template <> DEVINL void lds_wait<6>(f32x4* o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5])::"memory");
}
template <> DEVINL void lds_wait<8>(f32x4* o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])::"memory");
}
template <> DEVINL void lds_wait<10>(f32x4* o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(o[8]), "+v"(o[9])::"memory");
}
template <> DEVINL void lds_wait<12>(f32x4* o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(o[8]), "+v"(o[9]),
               "+v"(o[10]), "+v"(o[11])::"memory");
}
template <int N> DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One workgroup = 4 waves (2 x 2) computing out[p][m0:m0+BM][n0:n0+BN].
//
// K is walked in 32-wide chunks listed by the host (GemmHot::desc): valid taps of source 1, then source 2; zero
// padding taps never appear.  LDS ring of NST stages; a stage holds one chunk: BM A rows + BN B rows of 128 bytes,
// filled by LDS-DMA (global_load_lds_dwordx4: each wave-instruction writes 1 KiB = 8 rows).  Rows are unpadded --
// the DMA write is lane-linear -- so the 16-byte slots of a row are XOR-swizzled by (row >> 1) & 7 on the SOURCE
// address and on the fragment read, which makes ds_read_b128 conflict-free.  GroupNorm(1,C) on source 1 is folded
// (plan.h): gamma lives in the weights, the accumulator rows are scaled by rstd after the last source-1 chunk and
// the epilogue adds B1[p][n] - mean*rstd*S[p][n]; the K loop itself is a pure GEMM.
// Every wave issues the same number IA of DMA instructions per chunk, so `s_waitcnt vmcnt((NST-2)*IA)` retires
// exactly chunk i while NST-2 younger chunks stay in flight across the (raw) barrier.
// Rows beyond the batch / beyond Npad are CLAMPED, K is padded to 32 with zero weights: the loop has no predicate.
template <int BM, int BN, int NST, bool GN>
__global__ __launch_bounds__(256) void seg_gemm(const GemmHot g) {
  constexpr int WM = BM / 64, WN = BN / 64;          // 32x32 MFMA tiles per wave in M and N (waves are 2 x 2)
  constexpr int ROWS = BM + BN;
  constexpr int NI = ROWS / 32;                      // DMA wave-instructions per wave per chunk (8 rows each)
  constexpr int RA = BM / 32;                        // the first RA of them carry A rows
  constexpr int STAGE_F = ROWS * 32;
  constexpr int IA = NI;
  constexpr int PER = WM + WN;                       // ds_read_b128 per k-group of 8
  constexpr int NRD = 2 * PER;                       // ... per asm statement (two k-groups)
  constexpr int EP_LD = 32 * WN + 4;                 // epilogue transpose tile row stride (floats)
  constexpr int GN_SCRATCH = GN ? 256 * 4 + BM * 2 : 0;      // fp64 partial sums [256/BM][BM][2] + (mean, rstd)[BM]
  constexpr int RING_F = NST * STAGE_F + GN_SCRATCH;
  constexpr int LDS_F = RING_F > 4 * 32 * EP_LD + 4 * BM ? RING_F : 4 * 32 * EP_LD + 4 * BM;
  __shared__ __attribute__((aligned(1024))) float lds[LDS_F];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;
  unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0, st_wait = 0, st_comp = 0, st_a = 0, st_b = 0;
  unsigned long long st_r0 = 0;
  if (g.dbg) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  int bid = blockIdx.x;
  int ks = 0;
  if (g.split > 1) { ks = bid % g.split; bid /= g.split; }     // the splits of a tile are neighbours in launch order
  const int tile_id = bid;
  const int nt = bid % g.ntiles; bid /= g.ntiles;
  const int mt = bid % g.mtiles; const int p = bid / g.mtiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int fi = lane & 31, fh = lane >> 5;

  const int4 hd = scalar_load16(g.hdr + p);                        // {first descriptor, chunk count, source-1 chunks, p2}
  const int4* __restrict__ dsc = g.desc + hd.x;
  const int total = hd.y;
  int c0 = 0, c1 = total;                             // this workgroup's chunk range (split-K: an even share of the pixel's list)
  if (g.split > 1) { c0 = total * ks / g.split; c1 = total * (ks + 1) / g.split; }

  // ---- per-lane DMA source row offsets (loop invariant).  Lane l of instruction q = wave + 4j writes LDS row
  //      r = 8q + (l >> 3), physical slot l & 7, and fetches logical slot ls = (l & 7) ^ ((r >> 1) & 7), which
  //      does not depend on j. ----
  const int rbase = 8 * wave + (lane >> 3);
  const int ls4 = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 4;
  int offs1[NI], offs2[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = j < RA ? min(m0 + rbase + 32 * j, g.B - 1) : min(n0 + rbase + 32 * (j - RA), g.Npad - 1);
    offs1[j] = row * (j < RA ? g.lda : g.Kp) + ls4;
    offs2[j] = row * (j < RA ? g.lda2 : g.Kp2) + ls4;
  }

  // The DMA of one chunk is issued in IA pieces so that the pieces can be spread between the MFMA groups of the
  // chunk being computed.  Piece 0 fetches the chunk's descriptor (scalar load) and resolves the bases.
  int is_c = c0, is_stage = 0;
  int4 is_next = scalar_load16(dsc + c0);                 // descriptor of the next chunk to issue, loaded one chunk ahead
  const float* is_a = g.A; const float* is_w = g.W;
  bool is_src2 = false;
  float* is_sbase = lds;
  auto issue_piece = [&](int j) {
    if (j == 0) {
      const int4 d = is_next;
      ++is_c;
      is_next = scalar_load16(dsc + (is_c < total ? is_c : 0));
      is_src2 = (d.w & 1) != 0;
      is_a = (is_src2 ? g.A2 : g.A) + d.x;
      is_w = (is_src2 ? g.W2 : g.W) + d.y;
      is_sbase = lds + is_stage * STAGE_F + wave * 256;
      is_stage = is_stage + 1 == NST ? 0 : is_stage + 1;
    }
    const int o = is_src2 ? offs2[j] : offs1[j];
    __builtin_amdgcn_global_load_lds((glb_void*)((j < RA ? is_a : is_w) + o), (lds_void*)(is_sbase + j * 1024), 16, 0, 0);
  };

  // ---- GroupNorm(1,C) statistics of the block's BM rows from the producer's partial sums: all 256 threads read
  //      partials (256/BM threads per row).  The first 16 loads per thread are issued BEFORE the ring prefill so that
  //      their latency hides behind the DMA issue; the sums are reduced through a scratch area behind the ring. ----
  if (g.dbg) st_t1 = __builtin_amdgcn_s_memtime();
  const GemmCold* __restrict__ cold = g.cold;
  constexpr int PL = 256 / BM;
  const int gr_r = tid % BM, gr_pl = tid / BM;
  float2 gv[16];
  int gn_nparts = 0;
  const float2* gp = nullptr;
  if constexpr (GN) {
    gn_nparts = cold->gn_nparts;
    gp = cold->gn_part + min(m0 + gr_r, g.B - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u)          // clamped index, masked at summation: no select may sit between the loads
      gv[u] = gp[(long long)min(gr_pl + u * PL, gn_nparts - 1) * g.B];
  }

  // ---- prefill the ring ----
  int issued = c0;
  for (; issued < c1 && issued < c0 + NST - 1; ++issued) {
#pragma unroll
    for (int j = 0; j < IA; ++j) issue_piece(j);
  }
  unsigned long long st_p1 = 0, st_p2 = 0, st_p3 = 0, st_p4 = 0;
  if (g.dbg) st_p1 = __builtin_amdgcn_s_memtime();

  float rrow[GN ? WM : 1][16];
  if constexpr (GN) {
    const int r = gr_r, pl = gr_pl;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) if (pl + u * PL < gn_nparts) { s += (double)gv[u].x; q += (double)gv[u].y; }
    for (int i = pl + 16 * PL; i < gn_nparts; i += 16 * PL) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = gp[(long long)min(i + u * PL, gn_nparts - 1) * g.B];
#pragma unroll
      for (int u = 0; u < 16; ++u) if (i + u * PL < gn_nparts) { s += (double)v[u].x; q += (double)v[u].y; }
    }
    if (g.dbg) st_p2 = __builtin_amdgcn_s_memtime();
    double* red = reinterpret_cast<double*>(lds + NST * STAGE_F);            // [PL][BM][2]
    red[(pl * BM + r) * 2] = s; red[(pl * BM + r) * 2 + 1] = q;
    __syncthreads();
    float2* mr = reinterpret_cast<float2*>(lds + NST * STAGE_F + PL * BM * 4);   // [BM] (mean, rstd)
    if (tid < BM) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int k = 0; k < PL; ++k) { ss += red[(k * BM + tid) * 2]; qq += red[(k * BM + tid) * 2 + 1]; }
      const double inv = (double)cold->gn_inv_count;
      const double mean = ss * inv;
      double var = qq * inv - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      mr[tid] = make_float2((float)mean, rsqrtf((float)var + 1e-5f));
    }
    __syncthreads();
    // rstd of the 16 accumulator rows this lane holds per 32x32 tile (row = (r&3) + 8(r>>2) + 4 fh)
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) rrow[a][r] = mr[wave_m * 32 * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh].y;
  }

  if (g.dbg) st_p3 = __builtin_amdgcn_s_memtime();
  // ---- consumer side: per-lane LDS byte addresses of the fragments (loop invariant within a stage) ----
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_void*)lds;
  unsigned fa_off[WM][4], fb_off[WN][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
    for (int a = 0; a < WM; ++a) { const int r = wave_m * 32 * WM + a * 32 + fi; fa_off[a][kk] = r * 128 + (((2 * kk + fh) ^ ((r >> 1) & 7)) * 16); }
#pragma unroll
    for (int b = 0; b < WN; ++b) { const int r = BM + wave_n * 32 * WN + b * 32 + fi; fb_off[b][kk] = r * 128 + (((2 * kk + fh) ^ ((r >> 1) & 7)) * 16); }
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto addrs = [&](unsigned sb, int kp, unsigned* ad) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int kk = kp * 2 + h2;
#pragma unroll
      for (int a = 0; a < WM; ++a) ad[h2 * PER + a] = sb + fa_off[a][kk];
#pragma unroll
      for (int b = 0; b < WN; ++b) ad[h2 * PER + WM + b] = sb + fb_off[b][kk];
    }
  };
  // consume one k-pair (two groups of 8 k); behind each group's MFMAs issue DMA pieces gq, gq + 4, ... of the chunk
  // NST - 1 ahead (gq = 2 kp + h2)
  auto consume = [&](const f32x4* rd, int kp, bool do_issue) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x4 fa[WM], fb[WN];
#pragma unroll
      for (int a = 0; a < WM; ++a) fa[a] = rd[h2 * PER + a];
#pragma unroll
      for (int b = 0; b < WN; ++b) fb[b] = rd[h2 * PER + WM + b];
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
      if (do_issue) {
#pragma unroll
        for (int j = kp * 2 + h2; j < IA; j += 4) issue_piece(j);
      }
    }
  };

  if (g.dbg) st_p4 = __builtin_amdgcn_s_memtime();
  int cstage = 0;
  // The K loop proper.  It runs twice: over the source-1 chunks, then (after the folded-GroupNorm row scaling, which
  // must not sit inside the loop or hipcc keeps the accumulators in VGPRs and copies them to AGPRs every trip)
  // over the source-2 chunks.
  auto run_chunks = [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      if (g.dbg) st_a = __builtin_amdgcn_s_memtime();
      if (issued - (i + 1) >= NST - 2) wait_vmcnt<(NST - 2) * IA>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();            // chunk i landed for every wave; everyone is done with chunk i-1
      if (g.dbg) { st_b = __builtin_amdgcn_s_memtime(); st_wait += st_b - st_a; }
      const bool do_issue = issued < c1;
      if (do_issue) ++issued;
      const unsigned sb = lds_base + (unsigned)(cstage * STAGE_F * 4);
      cstage = cstage + 1 == NST ? 0 : cstage + 1;
      f32x4 r0[NRD], r1[NRD]; unsigned ad[NRD];
      addrs(sb, 0, ad); lds_issue<NRD>(r0, ad);
      addrs(sb, 1, ad);
      lds_wait<NRD>(r0);
      lds_issue<NRD>(r1, ad);                   // the second k-pair's reads fly under the first pair's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      consume(r0, 0, do_issue);
      __builtin_amdgcn_sched_barrier(0);
      lds_wait<NRD>(r1);
      __builtin_amdgcn_sched_barrier(0);
      consume(r1, 1, do_issue);
      if (g.dbg) { st_a = __builtin_amdgcn_s_memtime(); st_comp += st_a - st_b; }
    }
  };
  if constexpr (GN) {
    run_chunks(c0, min(c1, hd.z));
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] *= rrow[a][r];
    run_chunks(max(c0, hd.z), c1);
  } else {
    run_chunks(c0, c1);
  }
  if (g.dbg) st_t2 = __builtin_amdgcn_s_memtime();
  __syncthreads();      // all DMA retired (last wait was vmcnt(0)); LDS is reused by the epilogue

  // ---- split-K: park the partial tile in the workspace; the LAST workgroup of the tile to arrive sums all partials in split
  //      order (so the result does not depend on arrival order) and goes on to the epilogue.  The splits of a tile sit on
  //      different XCDs, whose L2s are not coherent with each other: partials are written and read with DEVICE-scope accesses
  //      (sc1: write-through / read-through at the L2, like the arrival counter) -- device-scope FENCES would write back and
  //      invalidate the whole L2 under the other workgroups' operand streams (measured: +80 us per launch). ----
  if (g.split > 1) {
    constexpr int TILE_F = WM * WN * 16 * 256;
    float* pt = g.part + (long long)tile_id * g.split * TILE_F;
    float* mine = pt + (long long)ks * TILE_F;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __hip_atomic_store(mine + ((a * WN + b) * 16 + r) * 256 + tid, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_vmcnt<0>();                                   // the write-throughs are acknowledged
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(g.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)(g.split - 1);
      if (last) __hip_atomic_store(g.cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // clean for the next launch
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int s2 = 0; s2 < g.split; ++s2) {
      const float* src = pt + (long long)s2 * TILE_F;
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[a][b][r] += __hip_atomic_load(src + ((a * WN + b) * 16 + r) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- epilogue: out = act(acc + bias + r) and per-row partial (sum, sumsq).  Each wave transposes its 32-row
  //      tile through LDS; then 8*WN consecutive lanes own one row (each lane 4 consecutive columns), so every global
  //      access is a full 128*WN-byte row segment.  A lane's column is fixed across the passes: bias-like vectors
  //      (folded-GroupNorm S / B1) are loaded once, and all residual loads are issued before the first store. ----
  const GemmCold c = *cold;
  const int p2 = hd.w;
  const int oy = p / c.Wout, ox = p - oy * c.Wout;
  const float* biasp = c.bias ? c.bias + (long long)p * c.bias_pix_stride : nullptr;
  float* tile = lds + wave * 32 * EP_LD;
  float* srow = lds + 4 * 32 * EP_LD;        // [2 (wave_n)][BM][2]
  const bool want_stats = c.stat_out != nullptr;
  constexpr int LPR = 8 * WN, RPI = 64 / LPR, NPASS = 32 / RPI;    // lanes per row, rows per pass, passes
  const int ecol = (lane % LPR) * 4, erow0 = lane / LPR;
  const int n = n0 + wave_n * 32 * WN + ecol;                       // this lane's first global column
  const bool ncol = n < c.nstore;
  float4 gs = make_float4(0.f, 0.f, 0.f, 0.f), gb = gs;
  if constexpr (GN) {
    if (ncol && !c.out_nchw) {
      gs = *reinterpret_cast<const float4*>(c.gn_s + (long long)p * g.Npad + n);
      gb = *reinterpret_cast<const float4*>(c.gn_b1 + (long long)p * g.Npad + n);
    }
  }
  const float2* mrp = reinterpret_cast<const float2*>(lds + NST * STAGE_F + (GN ? 256 * 4 : 0));
#pragma unroll
  for (int a = 0; a < WM; ++a) {
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int nb = n0 + wave_n * 32 * WN + b * 32 + fi;
      const float bv = (biasp && nb < g.Npad) ? biasp[nb] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_LD + b * 32 + fi] = acc[a][b][r] + bv;
    }
    // wave-local: LDS operations of one wave complete in order, no barrier needed
    const int mbase = wave_m * 32 * WM + a * 32;
    if (!c.out_nchw) {
      float4 rv[NPASS];
      if (c.R) {
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
          const int bb = m0 + mbase + t * RPI + erow0;
          rv[t] = (ncol && bb < g.B) ? *reinterpret_cast<const float4*>(c.R + ((long long)p2 * g.B + bb) * c.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int t = 0; t < NPASS; ++t) {
        const int row = t * RPI + erow0, bb = m0 + mbase + row;
        float4 v = *reinterpret_cast<const float4*>(tile + row * EP_LD + ecol);
        float s = 0.f, q = 0.f;
        if (ncol && bb < g.B) {
          if constexpr (GN) {
            const float2 mrv = mrp[mbase + row]; const float mrs = mrv.x * mrv.y;
            v.x += gb.x - mrs * gs.x; v.y += gb.y - mrs * gs.y; v.z += gb.z - mrs * gs.z; v.w += gb.w - mrs * gs.w;
          }
          if (c.R) { v.x += rv[t].x; v.y += rv[t].y; v.z += rv[t].z; v.w += rv[t].w; }
          if (c.act == ACT_GELU) {                     // packed pair form: half the VALU issue slots, bit-identical (device.h: sbm_gelu2)
            const sbm_f32x2 g0 = sbm_gelu2((sbm_f32x2){v.x, v.y}), g1 = sbm_gelu2((sbm_f32x2){v.z, v.w});
            v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
          } else if (c.act != ACT_NONE) {
            v.x = act_apply(c.act, v.x); v.y = act_apply(c.act, v.y); v.z = act_apply(c.act, v.z); v.w = act_apply(c.act, v.w);
          }
          *reinterpret_cast<float4*>(g.out + ((long long)p * g.B + bb) * c.ldo + n) = v;
          s = (v.x + v.y) + (v.z + v.w); q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if (want_stats) {
#pragma unroll
          for (int o = 1; o < LPR; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
          if ((lane % LPR) == 0) { srow[(wave_n * BM + mbase + row) * 2] = s; srow[(wave_n * BM + mbase + row) * 2 + 1] = q; }
        }
      }
    } else {
      // external NCHW output (final layers; N is tiny): per-element scatter
      const int y = oy - c.crop_y, x = ox - c.crop_x;
#pragma unroll
      for (int t = 0; t < NPASS; ++t) {
        const int row = t * RPI + erow0, bb = m0 + mbase + row;
        const float4 v = *reinterpret_cast<const float4*>(tile + row * EP_LD + ecol);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float mrs = 0.f;
        if constexpr (GN) { const float2 mrv = mrp[mbase + row]; mrs = mrv.x * mrv.y; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (bb < g.B && n + e < c.nstore && y >= 0 && y < c.Hn && x >= 0 && x < c.Wn) {
            float u = vv[e];
            if constexpr (GN) u += c.gn_b1[(long long)p * g.Npad + n + e] - mrs * c.gn_s[(long long)p * g.Npad + n + e];
            if (c.R) u += c.R[((long long)p2 * g.B + bb) * c.ldr + n + e];
            g.out[(((long long)bb * c.N + n + e) * c.Hn + y) * c.Wn + x] = act_apply(c.act, u);
          }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BM && m0 + tid < g.B) {
      const float s = srow[tid * 2] + srow[(BM + tid) * 2], q = srow[tid * 2 + 1] + srow[(BM + tid) * 2 + 1];
      c.stat_out[((long long)p * g.ntiles + nt) * g.B + m0 + tid] = make_float2(s, q);
    }
  }
  if (g.dbg && tid == 0 && g.dbg[15]) {        // timeline mode (SBM_GEMM_TIMELINE): every workgroup's span on the 100 MHz clock and where it ran
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* e = g.dbg + 16 + 6 * blockIdx.x;
    e[0] = st_r0; e[1] = __builtin_amdgcn_s_memrealtime(); e[2] = ((unsigned long long)xcc << 32) | hw; e[3] = (st_wait & 0xffffffffull) | ((__builtin_amdgcn_s_memtime() - st_t0) << 32); e[4] = st_comp; e[5] = ((unsigned long long)(c1 - c0) << 32) | (unsigned)(st_t2 - st_t1 > 0xffffffffull ? 0xffffffffull : st_t2 - st_t1);
  }
  if (g.dbg && tid == 0 && (int)blockIdx.x == (int)(gridDim.x / 2)) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    g.dbg[0] = st_t1 - st_t0; g.dbg[1] = st_t2 - st_t1; g.dbg[2] = t3 - st_t2; g.dbg[3] = st_wait; g.dbg[4] = __builtin_amdgcn_s_memrealtime() - st_r0; g.dbg[5] = st_comp;      // [4]: the same span as [0]+[1]+[2] on the 100 MHz clock
    g.dbg[8] = st_p1 - st_t1; g.dbg[9] = st_p2 - st_p1; g.dbg[10] = st_p3 - st_p2; g.dbg[11] = st_p4 - st_p3;
    g.dbg[6] = (unsigned long long)total; g.dbg[7] = gridDim.x;
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + time condition, GroupNorm partials per (pixel, sample)
// ------------------------------------------------------------------------------------------------

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
      for (int ky = ky0; ky < ky1; ++ky) {
        // the 7 taps of a row as one batch of independent loads (out-of-image taps: clamped address, zero weight)
        float4 x[7], w[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const bool in = kx >= kx0 && kx < kx1;
          const int qp = (oy - 3 + ky) * d.W + min(max(ox - 3 + kx, 0), d.W - 1);
          x[kx] = *reinterpret_cast<const float4*>(d.in + ((long long)qp * d.B + b) * d.ldi + c);
          w[kx] = in ? *reinterpret_cast<const float4*>(d.w + (ky * 7 + kx) * d.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
          if (kx >= kx0 && kx < kx1) { acc.x = fmaf(x[kx].x, w[kx].x, acc.x); acc.y = fmaf(x[kx].y, w[kx].y, acc.y); acc.z = fmaf(x[kx].z, w[kx].z, acc.z); acc.w = fmaf(x[kx].w, w[kx].w, acc.w); }
      }
      const float4 bi = *reinterpret_cast<const float4*>(d.bias + c);
      acc.x += bi.x; acc.y += bi.y; acc.z += bi.z; acc.w += bi.w;
      if (d.cond) {
        const float4 cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + c);
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

// W == 8 variant: a thread owns (sample, channel quad) and FOUR horizontally adjacent outputs (half an output row), so that the
// 7 input pixels of a window row are loaded once for the four windows that share them (12 loads per output instead of ~30).
// Taps are accumulated in the same (ky, kx) order as dwconv7_kernel: the results are bit-identical.
template <int HALF>
__device__ __forceinline__ void dw8_half_row(const DwArgs& d, int oy, int b, int c, float4* acc) {
  const int ky0 = max(0, 3 - oy), ky1 = min(7, d.H + 3 - oy);
  constexpr int XB = HALF ? 1 : 0;                               // first input column of the 7 this half row can touch
  for (int ky = ky0; ky < ky1; ++ky) {
    const int iy = oy - 3 + ky;
    float4 x[7], w[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      x[j] = *reinterpret_cast<const float4*>(d.in + ((long long)(iy * 8 + XB + j) * d.B + b) * d.ldi + c);
      w[j] = *reinterpret_cast<const float4*>(d.w + (ky * 7 + j) * d.C + c);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        constexpr int dummy = 0; (void)dummy;
        const int j = o + kx - (HALF ? 0 : 3);                    // input column (relative to XB) under tap kx of output 4*HALF + o
        if (j >= 0 && j < 7) {
          acc[o].x = fmaf(x[j].x, w[kx].x, acc[o].x); acc[o].y = fmaf(x[j].y, w[kx].y, acc[o].y);
          acc[o].z = fmaf(x[j].z, w[kx].z, acc[o].z); acc[o].w = fmaf(x[j].w, w[kx].w, acc[o].w);
        }
      }
  }
}

__global__ __launch_bounds__(256) void dwconv7_w8_kernel(const DwArgs d) {
  __shared__ float2 red[4][256];
  const int tid = threadIdx.x;
  const int oy = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int sl = tid / d.lanes, cl = tid - sl * d.lanes;         // lanes == C/4 here: one channel quad per thread
  const int b = blockIdx.y * d.spb + sl;
  const bool active = sl < d.spb && b < d.B;
  const int c = cl * 4;
  float4 acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    if (half) dw8_half_row<1>(d, oy, b, c, acc); else dw8_half_row<0>(d, oy, b, c, acc);
    const float4 bi = *reinterpret_cast<const float4*>(d.bias + c);
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.cond) cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + c);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      acc[o].x += bi.x; acc[o].y += bi.y; acc[o].z += bi.z; acc[o].w += bi.w;
      if (d.cond) { acc[o].x += cv.x; acc[o].y += cv.y; acc[o].z += cv.z; acc[o].w += cv.w; }
      const int p = oy * 8 + half * 4 + o;
      *reinterpret_cast<float4*>(d.out + ((long long)p * d.B + b) * d.ldo + c) = acc[o];
    }
  }
  if (d.stat) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      red[o][tid] = active ? make_float2(acc[o].x + acc[o].y + acc[o].z + acc[o].w, acc[o].x * acc[o].x + acc[o].y * acc[o].y + acc[o].z * acc[o].z + acc[o].w * acc[o].w)
                           : make_float2(0.f, 0.f);
    __syncthreads();
    if (active && cl < 4) {                                      // four threads of the sample, one output pixel each
      float ss = 0.f, qq = 0.f;
      for (int i = 0; i < d.lanes; ++i) { ss += red[cl][tid - cl + i].x; qq += red[cl][tid - cl + i].y; }
      d.stat[(long long)(oy * 8 + half * 4 + cl) * d.B + b] = make_float2(ss, qq);
    }
  }
}

// 8 x 8 variant with the sample's whole image in LDS: one workgroup per sample stages the 64 pixels x C channels (coalesced 16-byte
// loads, one L2 round trip) and the 49 x C weights, then a thread owns (channel quad, output row) and walks the window rows: per
// row 8 pixel + 7 weight LDS reads feed 8 outputs x 7 taps.  ~15 LDS reads per 56 tap-FMAs instead of ~14 global loads per 28.
// Taps accumulate in (ky, kx) order and the statistics in channel order, as in dwconv7_kernel.
__global__ __launch_bounds__(256) void dwconv7_t8_kernel(const DwArgs d) {
  extern __shared__ __attribute__((aligned(16))) float4 dw_sm[];
  const int C4 = d.C >> 2, b = blockIdx.x, tid = threadIdx.x;
  float4* tile = dw_sm;                 // [64][C4]
  float4* wt = dw_sm + 64 * C4;         // [49][C4]
  for (int idx = tid; idx < 64 * C4; idx += 256) {
    const int p = idx / C4, c4 = idx - p * C4;
    tile[idx] = *reinterpret_cast<const float4*>(d.in + ((long long)p * d.B + b) * d.ldi + 4 * c4);
  }
  for (int idx = tid; idx < 49 * C4; idx += 256) wt[idx] = *reinterpret_cast<const float4*>(d.w + 4 * idx);
  __syncthreads();
  const int c4 = tid % C4, oy = tid / C4;
  const bool active = tid < 8 * C4;
  float4 acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    const int ky0 = max(0, 3 - oy), ky1 = min(7, 8 + 3 - oy);
    for (int ky = ky0; ky < ky1; ++ky) {
      const int iy = oy - 3 + ky;
      float4 x[8], w[7];
#pragma unroll
      for (int ix = 0; ix < 8; ++ix) x[ix] = tile[(iy * 8 + ix) * C4 + c4];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) w[kx] = wt[(ky * 7 + kx) * C4 + c4];
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int ix = o - 3 + kx;
          if (ix >= 0 && ix < 8) {
            acc[o].x = fmaf(x[ix].x, w[kx].x, acc[o].x); acc[o].y = fmaf(x[ix].y, w[kx].y, acc[o].y);
            acc[o].z = fmaf(x[ix].z, w[kx].z, acc[o].z); acc[o].w = fmaf(x[ix].w, w[kx].w, acc[o].w);
          }
        }
    }
    const float4 bi = *reinterpret_cast<const float4*>(d.bias + 4 * c4);
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.cond) cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + 4 * c4);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      acc[o].x += bi.x; acc[o].y += bi.y; acc[o].z += bi.z; acc[o].w += bi.w;
      if (d.cond) { acc[o].x += cv.x; acc[o].y += cv.y; acc[o].z += cv.z; acc[o].w += cv.w; }
      *reinterpret_cast<float4*>(d.out + ((long long)(oy * 8 + o) * d.B + b) * d.ldo + 4 * c4) = acc[o];
    }
  }
  if (d.stat) {
    __syncthreads();                                              // the image is dead: its LDS holds the per-thread partials [64][C4]
    float2* red = reinterpret_cast<float2*>(dw_sm);
    if (active) {
#pragma unroll
      for (int o = 0; o < 8; ++o)
        red[(oy * 8 + o) * C4 + c4] = make_float2(acc[o].x + acc[o].y + acc[o].z + acc[o].w, acc[o].x * acc[o].x + acc[o].y * acc[o].y + acc[o].z * acc[o].z + acc[o].w * acc[o].w);
    }
    __syncthreads();
    if (tid < 64) {
      float ss = 0.f, qq = 0.f;
      for (int i = 0; i < C4; ++i) { ss += red[tid * C4 + i].x; qq += red[tid * C4 + i].y; }
      d.stat[(long long)tid * d.B + b] = make_float2(ss, qq);
    }
  }
}

// 16 x 16 variant: one workgroup per (sample, 32-channel slab) stages the slab of the sample's image (256 pixels x 8 channel quads,
// 32 KB) and the 49 x 8 weight quads in LDS; a thread owns (channel quad, output row, half row = 8 outputs) and walks the window rows:
// 11 pixel + 7 weight LDS reads feed 8 outputs x 7 taps.  GroupNorm partial sums are per (pixel, slab) -- DwOp::stat_slabs, the
// consumer just sees more partials -- reduced over the slab's 8 quads by wave shuffles.  Taps accumulate in (ky, kx) order as in
// dwconv7_kernel.
__global__ __launch_bounds__(256) void dwconv7_t16_kernel(const DwArgs d) {
  constexpr int RP = 16 * 8 + 1;                    // float4 pitch of an image row in LDS (+1: rows of one wave spread over the banks)
  __shared__ __attribute__((aligned(16))) float4 tile[16 * RP];
  __shared__ __attribute__((aligned(16))) float4 wt[49 * 8];
  const int b = blockIdx.x, slab = blockIdx.y, tid = threadIdx.x;
  const int cbase = slab * 32;
  const int nq = min(8, (d.C - cbase) >> 2);        // channel quads of this slab (the last slab may be narrower)
  for (int idx = tid; idx < 256 * 8; idx += 256) {
    const int p = idx >> 3, q = idx & 7;
    if (q < nq) tile[(p >> 4) * RP + (p & 15) * 8 + q] = *reinterpret_cast<const float4*>(d.in + ((long long)p * d.B + b) * d.ldi + cbase + 4 * q);
  }
  for (int idx = tid; idx < 49 * 8; idx += 256) if ((idx & 7) < nq) wt[idx] = *reinterpret_cast<const float4*>(d.w + (long long)(idx >> 3) * d.C + cbase + 4 * (idx & 7));
  __syncthreads();
  const int q = tid & 7, oy = (tid >> 3) & 15, half = tid >> 7;
  const bool live = q < nq;
  const int ox0 = half * 8;
  float4 acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ky0 = max(0, 3 - oy), ky1 = live ? min(7, 16 + 3 - oy) : 0;
  for (int ky = ky0; ky < ky1; ++ky) {
    const int iy = oy - 3 + ky;
    float4 x[14], w[7];
#pragma unroll
    for (int j = 0; j < 14; ++j) {                  // window columns ox0 - 3 + j; columns outside the image are never used below
      const int ix = min(max(ox0 - 3 + j, 0), 15);
      x[j] = tile[iy * RP + ix * 8 + q];
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) w[kx] = wt[(ky * 7 + kx) * 8 + q];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int ix = ox0 + o - 3 + kx;
        if (ix >= 0 && ix < 16) {
          const float4 xv = x[o + kx];
          acc[o].x = fmaf(xv.x, w[kx].x, acc[o].x); acc[o].y = fmaf(xv.y, w[kx].y, acc[o].y);
          acc[o].z = fmaf(xv.z, w[kx].z, acc[o].z); acc[o].w = fmaf(xv.w, w[kx].w, acc[o].w);
        }
      }
  }
  const int c = cbase + 4 * (live ? q : 0);
  const float4 bi = live ? *reinterpret_cast<const float4*>(d.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.cond && live) cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + c);
  float s = 0.f, qq = 0.f;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    acc[o].x += bi.x; acc[o].y += bi.y; acc[o].z += bi.z; acc[o].w += bi.w;
    if (d.cond) { acc[o].x += cv.x; acc[o].y += cv.y; acc[o].z += cv.z; acc[o].w += cv.w; }
    const int p = oy * 16 + ox0 + o;
    if (live) *reinterpret_cast<float4*>(d.out + ((long long)p * d.B + b) * d.ldo + c) = acc[o];
    s += (acc[o].x + acc[o].y) + (acc[o].z + acc[o].w);
    qq += (acc[o].x * acc[o].x + acc[o].y * acc[o].y) + (acc[o].z * acc[o].z + acc[o].w * acc[o].w);
  }
  if (d.stat) {                                     // ONE partial per (slab, sample): the workgroup holds all 256 pixels of its 32 channels
    __shared__ float2 red[4];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { s += __shfl_xor(s, m); qq += __shfl_xor(qq, m); }
    if ((tid & 63) == 0) red[tid >> 6] = make_float2(s, qq);
    __syncthreads();
    if (tid == 0) d.stat[(long long)slab * d.B + b] = make_float2((red[0].x + red[1].x) + (red[2].x + red[3].x), (red[0].y + red[1].y) + (red[2].y + red[3].y));
  }
}

// ------------------------------------------------------------------------------------------------
// LinearAttention core (unet_model.py:169-175): one wave per (sample, head); lane = head dim d,
// the two 32-lane halves take alternate pixels.
// ------------------------------------------------------------------------------------------------

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

// Same arithmetic, in the same order, with q/k/v of the (sample, head) staged through LDS by coalesced float4 loads (one
// latency instead of a dependent load per pixel and pass).  n <= 64.
__global__ __launch_bounds__(64) void linattn_lds_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sq[64 * 32], sk[64 * 32], sv[64 * 32];      // [pixel][d]
  __shared__ float ctx[32][33];
  __shared__ float qs[2][32];
  const int lane = threadIdx.x, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  const float* base = a.qkv + (long long)b * a.ldq + h * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int idx = lane; idx < a.n * 8; idx += 64) {
    const int i = idx >> 3, c4 = (idx & 7) * 4;
    const float* r = base + i * pstride + c4;
    reinterpret_cast<float4*>(sq)[idx] = *reinterpret_cast<const float4*>(r);
    reinterpret_cast<float4*>(sk)[idx] = *reinterpret_cast<const float4*>(r + hd);
    reinterpret_cast<float4*>(sv)[idx] = *reinterpret_cast<const float4*>(r + 2 * hd);
  }
  __syncthreads();
  float kmax = -INFINITY;
  for (int i = half; i < a.n; i += 2) kmax = fmaxf(kmax, sk[i * 32 + d]);
  kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
  float ksum = 0.f;
  for (int i = half; i < a.n; i += 2) ksum += expf(sk[i * 32 + d] - kmax);
  ksum += __shfl_xor(ksum, 32);
  const float kinv = 1.0f / ksum;
  float c[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  for (int i = 0; i < a.n; ++i) {
    const float kv = expf(sk[i * 32 + d] - kmax) * kinv;
    const float* vrow = sv + i * 32 + half * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = fmaf(kv, vrow[e], c[e]);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) ctx[d][half * 16 + e] = c[e];
  __syncthreads();
  float col[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) col[dd] = ctx[dd][d];
  for (int i0 = 0; i0 < a.n; i0 += 2) {
    const int i = i0 + half;
    const bool ok = i < a.n;
    const float qv = ok ? sq[i * 32 + d] : 0.f;
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

// 16 < n <= 64: four waves per (sample, head).  q/k/v staged once in LDS; both softmaxes are then applied IN PLACE with every
// exponential evaluated exactly once (256 threads share them: expf is the expensive instruction sequence here): k over the
// pixels per channel (8 thread groups per channel, combined through LDS), q over the channels per pixel (4 threads per pixel).
// Each wave accumulates the context over its quarter of the pixels, the four partial contexts are added in wave order, and
// each wave produces its quarter of the output pixels.
__global__ __launch_bounds__(256) void linattn_lds4_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float la_sm[];      // q, k, v of the (sample, head): 3 x [n pixels][32] (launch: 3 * n * 128 bytes)
  float* sq = la_sm; float* sk = la_sm + a.n * 32; float* sv = la_sm + 2 * a.n * 32;
  __shared__ float pctx[4][32][33];
  __shared__ float kred[8][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  const float* base = a.qkv + (long long)b * a.ldq + h * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int idx = tid; idx < a.n * 8; idx += 256) {
    const int i = idx >> 3, c4 = (idx & 7) * 4;
    const float* r = base + i * pstride + c4;
    reinterpret_cast<float4*>(sq)[idx] = *reinterpret_cast<const float4*>(r);
    reinterpret_cast<float4*>(sk)[idx] = *reinterpret_cast<const float4*>(r + hd);
    reinterpret_cast<float4*>(sv)[idx] = *reinterpret_cast<const float4*>(r + 2 * hd);
  }
  __syncthreads();
  // ---- k: softmax over the pixels, per channel kd; thread group kg takes pixels kg, kg + 8, ...
  const int kd = tid & 31, kg = tid >> 5;
  float m = -INFINITY;
  for (int i = kg; i < a.n; i += 8) m = fmaxf(m, sk[i * 32 + kd]);
  kred[kg][kd] = m;
  __syncthreads();
  float kmax = kred[0][kd];
#pragma unroll
  for (int g = 1; g < 8; ++g) kmax = fmaxf(kmax, kred[g][kd]);
  __syncthreads();
  float ps = 0.f;
  for (int i = kg; i < a.n; i += 8) { const float e = expf(sk[i * 32 + kd] - kmax); sk[i * 32 + kd] = e; ps += e; }
  kred[kg][kd] = ps;
  // ---- q: softmax over the channels, per pixel; 4 threads per pixel, 8 channels each; scaled
  for (int qi = tid >> 2; qi < ((a.n + 63) & ~63); qi += 64) {
    const int part = tid & 3;
    if (qi < a.n) {
      float* qp = sq + qi * 32 + part * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = qp[j];
      float qm = v[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) qm = fmaxf(qm, v[j]);
      qm = fmaxf(qm, __shfl_xor(qm, 1)); qm = fmaxf(qm, __shfl_xor(qm, 2));
      float qsum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = expf(v[j] - qm); qsum += v[j]; }
      qsum += __shfl_xor(qsum, 1); qsum += __shfl_xor(qsum, 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) qp[j] = v[j] / qsum * a.scale;
    }
  }
  __syncthreads();
  float ksum = kred[0][d];
#pragma unroll
  for (int g = 1; g < 8; ++g) ksum += kred[g][d];
  const float kinv = 1.0f / ksum;
  // ---- context over this wave's quarter of the pixels
  const int per = (a.n + 3) / 4, i_lo = wave * per, i_hi = min(a.n, i_lo + per);
  float c[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  for (int i = i_lo; i < i_hi; ++i) {
    const float kv = sk[i * 32 + d] * kinv;
    const float* vrow = sv + i * 32 + half * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = fmaf(kv, vrow[e], c[e]);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) pctx[wave][d][half * 16 + e] = c[e];
  __syncthreads();
  float col[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) col[dd] = ((pctx[0][dd][d] + pctx[1][dd][d]) + pctx[2][dd][d]) + pctx[3][dd][d];
  // ---- out[i][e = d] = sum_dd ctx[dd][e] * softmax(q)[i][dd] * scale for this wave's pixels; the halves take alternate pixels
  for (int i = i_lo + half; i < i_hi; i += 2) {
    const float* qrow = sq + i * 32;
    float o = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) o = fmaf(col[dd], qrow[dd], o);
    a.out[((long long)i * a.B + b) * a.ldo + h * 32 + d] = o;
  }
}

// to_qkv (1x1 conv with the GroupNorm in front folded in, 64 input channels) + the LinearAttention core for one (sample, head),
// n = 64 pixels: the head's q, k, v (64 px x 96) are produced by MFMA straight into the LDS tiles the core works on, instead
// of a 25 MB round trip through memory and a launch of their own.  Same formulas as seg_gemm's folded GroupNorm epilogue and
// linattn_lds4_kernel.
struct QkvAttnArgs {
  AttnArgs a;                     // out, B, n, ldo, heads, scale (qkv / ldq unused)
  const float* A; int lda;        // block input h: [pixel][B][lda], 64 channels
  const float* W;                 // packed to_qkv weights [Npad][Kp = 64] (gamma folded)
  const float* bias;              // or null
  const float2* gn_part; int gn_nparts; float gn_inv;
  const float* gn_s; const float* gn_b1; int Npad;
};

__global__ __launch_bounds__(256) void linattn_qkv_kernel(const QkvAttnArgs qa) {
  const AttnArgs& a = qa.a;
  constexpr int QLD = 33;                                         // q rows are padded: the output MFMA reads q[pixel = lane][k] (a column)
  __shared__ __attribute__((aligned(16))) float sq[64 * QLD], sk[64 * 32], sv[64 * 32];     // [pixel][d]
  __shared__ __attribute__((aligned(16))) float ht[64 * 68];                                 // block input tile [pixel][64 + 4]
  float (*pctx)[32][33] = reinterpret_cast<float (*)[32][33]>(ht);                            // [4][32][33] partial contexts: the input tile is dead by then
  static_assert(4 * 32 * 33 <= 64 * 68, "partial contexts must fit the input tile they alias");
  __shared__ float kred[8][32];
  __shared__ double sred[2][4];
  __shared__ float2 s_mr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int hd = a.heads * 32;
  // ---- GroupNorm statistics of sample b from the producer's partials
  {
    double s = 0.0, q = 0.0;
    for (int i = tid; i < qa.gn_nparts; i += 256) { const float2 v = qa.gn_part[(long long)i * a.B + b]; s += (double)v.x; q += (double)v.y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane == 0) { sred[0][wave] = s; sred[1][wave] = q; }
  }
  // ---- block input tile: 64 pixels x 64 channels, 16 float4 per pixel row
  for (int idx = tid; idx < 64 * 16; idx += 256) {
    const int p = idx >> 4, c4 = (idx & 15) * 4;
    *reinterpret_cast<float4*>(ht + p * 68 + c4) = *reinterpret_cast<const float4*>(qa.A + ((long long)p * a.B + b) * qa.lda + c4);
  }
  __syncthreads();
  if (tid == 0) {
    const double ss = ((sred[0][0] + sred[0][1]) + sred[0][2]) + sred[0][3], qq = ((sred[1][0] + sred[1][1]) + sred[1][2]) + sred[1][3];
    const double mean = ss * (double)qa.gn_inv;
    double var = qq * (double)qa.gn_inv - mean * mean; var = var < 0.0 ? 0.0 : var;
    s_mr = make_float2((float)mean, rsqrtf((float)var + 1e-5f));
  }
  __syncthreads();
  const float mean = s_mr.x, rstd = s_mr.y;
  // ---- q | k | v = folded-GroupNorm 1x1 conv: 6 tiles of 32 pixels x 32 channels, K = 64; wave w takes tiles w and w + 4
  for (int t = wave; t < 6; t += 4) {
    const int nt = t % 3, mt = t / 3;                            // nt: 0 q, 1 k, 2 v
    const int nrow = nt * hd + h * 32 + d;                       // this lane's weight row (output channel)
    typedef float f32x16l __attribute__((ext_vector_type(16)));
    f32x16l acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      float av[16], wv[16];
      const float* ap = ht + (mt * 32 + d) * 68 + kc * 32 + 16 * half;
      const float* wp = qa.W + (long long)nrow * 64 + kc * 32 + 16 * half;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 x = *reinterpret_cast<const float4*>(ap + 4 * j), w = *reinterpret_cast<const float4*>(wp + 4 * j);
        av[4 * j] = x.x; av[4 * j + 1] = x.y; av[4 * j + 2] = x.z; av[4 * j + 3] = x.w;
        wv[4 * j] = w.x; wv[4 * j + 1] = w.y; wv[4 * j + 2] = w.z; wv[4 * j + 3] = w.w;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wv[j], acc, 0, 0, 0);
    }
    float* dst = nt == 0 ? sq : (nt == 1 ? sk : sv);
    const float bv = qa.bias ? qa.bias[nrow] : 0.f;
    const float gs = qa.gn_s[nrow], gb = qa.gn_b1[nrow];         // a 1x1 conv: the folded-GroupNorm tables are the same for every pixel
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] * rstd;
      v += bv;
      v += gb - (mean * rstd) * gs;
      dst[p * (nt == 0 ? QLD : 32) + d] = v;
    }
  }
  __syncthreads();
  // ---- from here on: linattn_lds4_kernel
  const int kd = tid & 31, kg = tid >> 5;
  float m = -INFINITY;
  for (int i = kg; i < a.n; i += 8) m = fmaxf(m, sk[i * 32 + kd]);
  kred[kg][kd] = m;
  __syncthreads();
  float kmax = kred[0][kd];
#pragma unroll
  for (int g = 1; g < 8; ++g) kmax = fmaxf(kmax, kred[g][kd]);
  __syncthreads();
  float ps = 0.f;
  for (int i = kg; i < a.n; i += 8) { const float e = expf(sk[i * 32 + kd] - kmax); sk[i * 32 + kd] = e; ps += e; }
  kred[kg][kd] = ps;
  {
    const int qi = tid >> 2, part = tid & 3;
    if (qi < a.n) {
      float* qp = sq + qi * QLD + part * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = qp[j];
      float qm = v[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) qm = fmaxf(qm, v[j]);
      qm = fmaxf(qm, __shfl_xor(qm, 1)); qm = fmaxf(qm, __shfl_xor(qm, 2));
      float qsum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = expf(v[j] - qm); qsum += v[j]; }
      qsum += __shfl_xor(qsum, 1); qsum += __shfl_xor(qsum, 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) qp[j] = v[j] / qsum * a.scale;
    }
  }
  __syncthreads();
  // ---- context[d][e] = sum_px softmax_k[px][d] v[px][e] and out[px][e] = sum_d softmax_q[px][d] context[d][e] on the matrix pipe
  //      (v_mfma_f32_32x32x2_f32; lane (i = lane & 31, half): A[i][k = 2 s + half], B[k = 2 s + half][j = lane & 31]).  Each wave
  //      contracts 16 of the 64 pixels (8 MFMAs); the four partial contexts meet in LDS, scaled by 1 / sum_k.
  typedef float f32x16l __attribute__((ext_vector_type(16)));
  {
    f32x16l cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const int px = wave * 16 + 2 * s2 + half;
      cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[px * 32 + d], sv[px * 32 + d], cacc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pctx[wave][(r & 3) + 8 * (r >> 2) + 4 * half][d] = cacc[r];
  }
  __syncthreads();
  float* ctxs = sk;                                              // [d][32]: k is dead once every wave has its partial context
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int dd = idx >> 5, e = idx & 31;
    float ks = kred[0][dd];
#pragma unroll
    for (int g = 1; g < 8; ++g) ks += kred[g][dd];
    ctxs[idx] = (((pctx[0][dd][e] + pctx[1][dd][e]) + pctx[2][dd][e]) + pctx[3][dd][e]) * (1.0f / ks);
  }
  __syncthreads();
  if (wave < 2) {                                                // two 32-pixel tiles of the output, K = 32
    f32x16l oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    const float* qa_ = sq + (wave * 32 + d) * QLD + half;
    const float* cb = ctxs + half * 32 + d;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa_[2 * s2], cb[2 * s2 * 32], oacc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      a.out[((long long)px * a.B + b) * a.ldo + h * 32 + d] = oacc[r];
    }
  }
}

// The whole 8x8 attention block of a ConvNeXt score net for ONE sample per workgroup (unet_model.py:21-27, 150-175: Residual(PreNorm(LinearAttention))),
// 1024 threads = 4 heads x 4 waves: folded-GroupNorm to_qkv, the LinearAttention core of the four heads side by side, to_out.0, GroupNorm(1, C),
// the residual.  Round 4: as (sample, head) workgroups (linattn_qkv_kernel) the block cost a 27 us launch -- 1024 workgroups of 42 KB LDS are 1.33
// rounds of the chip -- plus 13 us of staging, a 1x1 GEMM pass and a GroupNorm op at the head of the pixel kernel's second launch.  Here every
// sample is one workgroup (one round), the input tile and its statistics are read once instead of four times, and the attention output never
// leaves LDS.  All MFMA: 32x32x2 for to_qkv (as before), 16x16x4 for the context, the output and to_out.0 (no partial tiles to merge).
struct AttnBlockArgs {
  QkvAttnArgs q;                          // to_qkv + core, as linattn_qkv_kernel takes them (q.a.out unused)
  const float* Wo; const float* bo;       // to_out.0: [64][128] (k contiguous), [64]
  const float* gamma; const float* beta;  // to_out.1: GroupNorm(1, 64) affine
  float* out; int ldo; float inv_count;   // block output [pixel][B][ldo]; 1 / (64 channels x 64 pixels)
  unsigned long long* dbg;                // SBM_ATTNBLOCK_STAMPS=1: s_memtime of workgroup 0 at the phase boundaries
};
#define AB_STAMP(i) do { if (ab.dbg && blockIdx.x == 0 && threadIdx.x == 0) ab.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
// a workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the vector-memory queue, i.e. it would wait for
// to_out.0's weights at the first barrier behind their request instead of five barriers later
__device__ __forceinline__ void ab_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int AB_QLD = 33, AB_HT = 68;
constexpr int AB_HEAD_F = 64 * AB_QLD + 64 * 32 + 64 * 32 + 32 * 32 + 8 * 32;      // sq | sk | sv | ctx | kred
constexpr int AB_LDS_F = 64 * AB_HT + 4 * AB_HEAD_F + 64;
constexpr int AB_LDS_BYTES = AB_LDS_F * 4 + 2 * 16 * 8 + 16;

__global__ __launch_bounds__(1024) void attnblock_kernel(const AttnBlockArgs ab) {
  const QkvAttnArgs& qa = ab.q; const AttnArgs& a = qa.a;
  extern __shared__ __attribute__((aligned(16))) float ab_lds[];
  AB_STAMP(0);
  float* ht = ab_lds;                                             // block input tile [pixel][64 + 4] (= the residual)
  float* hb = ht + 64 * AB_HT;
  float* wstat = hb + 4 * AB_HEAD_F;                              // [16 waves][2] (sum, sumsq) of to_out.0's output, then {mean, rstd} at [32], [33]
  double* sred = reinterpret_cast<double*>(wstat + 64);           // [2][16]
  // wave -> (head h = wave & 3, quarter wq = wave >> 2): waves w, w + 4, w + 8, w + 12 share a SIMD, so every SIMD holds one head's four quarters --
  // 2 + 2 + 1 + 1 of the 24 to_qkv tiles (with the quarters of one head on four SIMDs, two SIMDs had 8 tiles and two had 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = wave & 3, wq = wave >> 2, t = wq * 64 + lane, d = lane & 31, half = lane >> 5;
  const int b = blockIdx.x, hd = 128;
  // to_qkv: quarter wq < 3 computes q (0), k (1) or v (2) of its head for BOTH halves of the pixels -- two tiles with the same 32 weight rows, requested
  // before anything else: lane (d, half) holds W[row][32 kc + 16 half .. + 15].  (Quarter 3 idles here: the SIMD's matrix pipe is the limit, and its
  // four quarters together have the head's six tiles whichever way they are dealt.)
  float4 wq0[8];
  if (wq < 3) {
    const float* w0 = qa.W + ((long long)(h * 3 + wq) * 8 * 64 + lane) * 4;      // fragment order (pack_attn_block_fragments): 1 KB per wave and load
#pragma unroll
    for (int j = 0; j < 8; ++j) wq0[j] = *reinterpret_cast<const float4*>(w0 + j * 256);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) wq0[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float* sq = hb + h * AB_HEAD_F; float* sk = sq + 64 * AB_QLD; float* sv = sk + 64 * 32; float* ctx = sv + 64 * 32; float* kred = ctx + 32 * 32;
  typedef float f32x16l __attribute__((ext_vector_type(16)));
  typedef float f32x4l __attribute__((ext_vector_type(4)));
  // ---- GroupNorm statistics of the block input from the producer's partials; the input tile: one float4 per thread
  {
    double s = 0.0, q = 0.0;
    for (int i = tid; i < qa.gn_nparts; i += 1024) { const float2 v = qa.gn_part[(long long)i * a.B + b]; s += (double)v.x; q += (double)v.y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane == 0) { sred[wave] = s; sred[16 + wave] = q; }
    const int p = tid >> 4, c4 = (tid & 15) * 4;
    *reinterpret_cast<float4*>(ht + p * AB_HT + c4) = *reinterpret_cast<const float4*>(qa.A + ((long long)p * a.B + b) * qa.lda + c4);
  }
  __syncthreads();
  AB_STAMP(1);
  if (tid == 0) {
    double ss = 0.0, qq = 0.0;
    for (int w = 0; w < 16; ++w) { ss += sred[w]; qq += sred[16 + w]; }
    const double mean = ss * (double)qa.gn_inv;
    double var = qq * (double)qa.gn_inv - mean * mean; var = var < 0.0 ? 0.0 : var;
    wstat[32] = (float)mean; wstat[33] = rsqrtf((float)var + 1e-5f);
  }
  __syncthreads();
  AB_STAMP(2);
  {
    const float mean = wstat[32], rstd = wstat[33];
    // ---- q | k | v of head h: 6 tiles of 32 pixels x 32 channels, K = 64
    if (wq < 3)
    for (int mt = 0; mt < 2; ++mt) {
      const int nt = wq;                                           // nt: 0 q, 1 k, 2 v
      const int nrow = nt * hd + h * 32 + d;
      f32x16l acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        float av[16], wv[16];
        const float* ap = ht + (mt * 32 + d) * AB_HT + kc * 32 + 16 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 x = *reinterpret_cast<const float4*>(ap + 4 * j), w = wq0[kc * 4 + j];
          av[4 * j] = x.x; av[4 * j + 1] = x.y; av[4 * j + 2] = x.z; av[4 * j + 3] = x.w;
          wv[4 * j] = w.x; wv[4 * j + 1] = w.y; wv[4 * j + 2] = w.z; wv[4 * j + 3] = w.w;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wv[j], acc, 0, 0, 0);
      }
      float* dst = nt == 0 ? sq : (nt == 1 ? sk : sv);
      const float bv = qa.bias ? qa.bias[nrow] : 0.f;
      const float gs = qa.gn_s[nrow], gb = qa.gn_b1[nrow];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r] * rstd;
        v += bv;
        v += gb - (mean * rstd) * gs;
        dst[p * (nt == 0 ? AB_QLD : 32) + d] = v;
      }
    }
  }
  __syncthreads();
  AB_STAMP(3);
  const int l15 = lane & 15, kq = lane >> 4;
  // to_out.0's weights of this wave's column tile, requested here, behind to_qkv (used five barriers on): lane (n, kq) holds Wo[n][32 kq .. 32 kq + 31] -- the MFMA's k
  // runs over {32 kq + s}, i.e. lane group kq contracts head kq
  const int e_pt = wave >> 2, e_nt = wave & 3;
  float4 wo[8];
  {
    const float* wp = ab.Wo + ((long long)e_nt * 8 * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) wo[j] = *reinterpret_cast<const float4*>(wp + j * 256);
  }
  // ---- softmax of k over the 64 pixels (kept as exp, the sum divides the context), of q over the 32 channels (x scale): linattn_lds4_kernel's form
  const int kd = t & 31, kg = t >> 5;
  float m = -INFINITY;
  for (int i = kg; i < 64; i += 8) m = fmaxf(m, sk[i * 32 + kd]);
  kred[kg * 32 + kd] = m;
  ab_sync();
  AB_STAMP(4);
  float kmax = kred[kd];
#pragma unroll
  for (int g = 1; g < 8; ++g) kmax = fmaxf(kmax, kred[g * 32 + kd]);
  ab_sync();
  AB_STAMP(5);
  float ps = 0.f;
  for (int i = kg; i < 64; i += 8) { const float e = expf(sk[i * 32 + kd] - kmax); sk[i * 32 + kd] = e; ps += e; }
  kred[kg * 32 + kd] = ps;
  {
    const int qi = t >> 2, part = t & 3;
    float* qp = sq + qi * AB_QLD + part * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = qp[j];
    float qm = v[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) qm = fmaxf(qm, v[j]);
    qm = fmaxf(qm, __shfl_xor(qm, 1)); qm = fmaxf(qm, __shfl_xor(qm, 2));
    float qsum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = expf(v[j] - qm); qsum += v[j]; }
    qsum += __shfl_xor(qsum, 1); qsum += __shfl_xor(qsum, 2);
    const float qf = a.scale / qsum;
#pragma unroll
    for (int j = 0; j < 8; ++j) qp[j] = v[j] * qf;
  }
  ab_sync();
  AB_STAMP(6);
  // ---- context[d][e] = sum_px k~[px][d] v[px][e] / sum_px k~[px][d]: wave wq owns the 16 x 16 tile (di, ej), K = 64 pixels, v_mfma_f32_16x16x4_f32
  //      (lane (i = lane & 15, kq = lane >> 4): A[i][k = 4 s + kq], B[k][j = lane & 15]; result rows 4 kq + r)
  {
    const int di = wq >> 1, ej = wq & 1;
    f32x4l acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int px = 4 * s + kq; acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sk[px * 32 + di * 16 + l15], sv[px * 32 + ej * 16 + l15], acc, 0, 0, 0); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dd = di * 16 + 4 * kq + r;
      float ks = kred[dd];
#pragma unroll
      for (int g = 1; g < 8; ++g) ks += kred[g * 32 + dd];
      ctx[dd * 32 + ej * 16 + l15] = acc[r] * (1.0f / ks);
    }
  }
  ab_sync();
  AB_STAMP(7);
  // ---- out[px][e] = sum_d q~[px][d] context[d][e]: wave wq owns the 16 pixels of tile wq, both column tiles; the result replaces q~ (rows nobody else reads)
  {
    f32x4l o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float qv = sq[(wq * 16 + l15) * AB_QLD + 4 * s + kq];
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, ctx[(4 * s + kq) * 32 + l15], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, ctx[(4 * s + kq) * 32 + 16 + l15], o1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { float* row = sq + (wq * 16 + 4 * kq + r) * AB_QLD; row[l15] = o0[r]; row[16 + l15] = o1[r]; }
  }
  ab_sync();
  AB_STAMP(8);
  // ---- u = to_out.0(att) + bias: wave (e_pt, e_nt) owns a 16 x 16 tile of the 64 x 64 output, K = 128 = 4 heads x 32
  f32x4l u = {0.f, 0.f, 0.f, 0.f};
  {
    const float* arow = hb + kq * AB_HEAD_F + (e_pt * 16 + l15) * AB_QLD;      // att[px][32 kq + s] = head kq's tile
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      u = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j], wo[j].x, u, 0, 0, 0);
      u = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 1], wo[j].y, u, 0, 0, 0);
      u = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 2], wo[j].z, u, 0, 0, 0);
      u = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 3], wo[j].w, u, 0, 0, 0);
    }
  }
  const int n = e_nt * 16 + l15;
  {
    const float bv = ab.bo ? ab.bo[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { u[r] += bv; s1 += u[r]; s2 += u[r] * u[r]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) { wstat[2 * wave] = s1; wstat[2 * wave + 1] = s2; }
  }
  const float ga = ab.gamma[n], be = ab.beta[n];
  __syncthreads();
  AB_STAMP(9);
  if (tid == 0) {                                                // (fp64 division and square root: one thread, not 1024)
    double ss = 0.0, qq = 0.0;
    for (int w = 0; w < 16; ++w) { ss += (double)wstat[2 * w]; qq += (double)wstat[2 * w + 1]; }
    const double mean_d = ss * (double)ab.inv_count;
    double var = qq * (double)ab.inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
    wstat[32] = (float)mean_d; wstat[33] = (float)(1.0 / sqrt(var + 1e-5));
  }
  __syncthreads();
  AB_STAMP(10);
  {
    const float mean = wstat[32], rstd = wstat[33];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int px = e_pt * 16 + 4 * kq + r;
      ab.out[((long long)px * a.B + b) * ab.ldo + n] = ((u[r] - mean) * rstd) * ga + be + ht[px * AB_HT + n];
    }
  }
  AB_STAMP(15);
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
// y = GroupNorm(1,C)(u) * gamma + beta + x   -- one workgroup per (sample, pixel slice); every slice repeats the (cheap) reduction of
// the producer's partial sums in the same order, so the result does not depend on the slicing
// ------------------------------------------------------------------------------------------------

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
  for (int idx = tid + 256 * blockIdx.y; idx < g.P * c4n; idx += 256 * gridDim.y) {
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
__global__ void sinus_kernel(const float* t, const float* freq, float* out, int B, int dim, int ld, int cos_first) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * ld) return;
  const int b = idx / ld, j = idx - b * ld, half = dim / 2;
  float v = 0.f;
  if (j < 2 * half) {
    const bool first = j < half;
    const float a = t[b] * freq[first ? j : j - half];
    v = (first != (cos_first != 0)) ? sinf(a) : cosf(a);     // unet_model: [sin, cos]; unet_openai: [cos, sin]
  }
  out[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(G, C) (+ SiLU), materialised (unet_openai normalization()): one workgroup per sample; every thread
// always touches the same channel quads, hence the same groups, so the group sums reduce in a fixed order.
// ------------------------------------------------------------------------------------------------
struct Gn32Args { const float* in; float* out; const float* gamma; const float* beta; double2* partial; int B, P, C, ldi, ldo, groups, silu, slices, vslices, gpw; };

__global__ __launch_bounds__(256) void gn32_kernel(const Gn32Args a) {
  __shared__ float2 part[256 * 2];
  __shared__ float2 stat[64];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c4n = a.C / 4, cg = a.C / a.groups;
  const int npt = (c4n + 255) / 256;                        // channel quads per thread (<= 2 for C <= 2048)
  const int rows_per_iter = c4n >= 256 ? 1 : 256 / c4n;     // pixels covered per sweep when C/4 < 256
  const int q0 = c4n >= 256 ? tid : tid % c4n, prow = c4n >= 256 ? 0 : tid / c4n;
  const bool active = c4n >= 256 || tid < rows_per_iter * c4n;
  for (int k = 0; k < 2; ++k) {
    float s = 0.f, q = 0.f;
    const int c4 = q0 + k * 256;
    if (k < npt && c4 < c4n && active)
      for (int p = prow; p < a.P; p += rows_per_iter) {
        const float4 v = *reinterpret_cast<const float4*>(a.in + ((long long)p * a.B + b) * a.ldi + c4 * 4);
        s += (v.x + v.y) + (v.z + v.w); q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    part[k * 256 + tid] = make_float2(s, q);
  }
  __syncthreads();
  if (tid < a.groups) {
    // entries of group g: threads whose quad lies in [g*cg/4, (g+1)*cg/4)
    double s = 0.0, q = 0.0;
    const int lo = tid * cg / 4, hi = (tid + 1) * cg / 4;
    for (int k = 0; k < npt; ++k)
      for (int c4 = lo; c4 < hi; ++c4) {
        if (c4 < k * 256 || c4 >= (k + 1) * 256) continue;
        const int qq = c4 - k * 256;
        if (c4n >= 256) { const float2 v = part[k * 256 + qq]; s += v.x; q += v.y; }
        else for (int r = 0; r < rows_per_iter; ++r) { const float2 v = part[r * c4n + qq]; s += v.x; q += v.y; }
      }
    const double n = (double)a.P * cg, mean = s / n;
    double var = q / n - mean * mean; var = var < 0.0 ? 0.0 : var;
    stat[tid] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
  }
  __syncthreads();
  // With few samples one workgroup per sample leaves most of the chip idle (B = 64: 64 of 256 CUs, each streaming the sample three
  // times): blockIdx.y cuts the pixels into `slices`; every slice's workgroup repeats the statistics pass over the whole sample
  // (L2-resident) and normalises only its own pixels -- same sums in the same order, so the result does not depend on the cut.
  const int slice = blockIdx.y, stride = rows_per_iter * a.slices;
  for (int k = 0; k < npt; ++k) {
    const int c4 = q0 + k * 256;
    if (c4 >= c4n || !active) continue;
    const float2 st = stat[(c4 * 4) / cg];
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c4 * 4), be = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
    for (int p = prow + slice * rows_per_iter; p < a.P; p += stride) {
      float4 v = *reinterpret_cast<const float4*>(a.in + ((long long)p * a.B + b) * a.ldi + c4 * 4);
      v.x = ((v.x - st.x) * st.y) * ga.x + be.x; v.y = ((v.y - st.x) * st.y) * ga.y + be.y;
      v.z = ((v.z - st.x) * st.y) * ga.z + be.z; v.w = ((v.w - st.x) * st.y) * ga.w + be.w;
      if (a.silu) { v.x = act_apply(ACT_SILU, v.x); v.y = act_apply(ACT_SILU, v.y); v.z = act_apply(ACT_SILU, v.z); v.w = act_apply(ACT_SILU, v.w); }
      *reinterpret_cast<float4*>(a.out + ((long long)p * a.B + b) * a.ldo + c4 * 4) = v;
    }
  }
}

// Large tensors (>= 96 KB per sample): statistics and normalisation as TWO launches, so that every element is read twice instead of
// `slices` + 1 times.  The statistics grid is (sample, vslices) with vslices fixed by the tensor shape alone (NOT by the batch), the
// partials are fp64 and every consumer adds them in slice order: a sample's result depends neither on scheduling nor on the batch it
// is computed in (shards of a batch reproduce the unsharded rows bit for bit).
__global__ __launch_bounds__(256) void gn32_stats_kernel(const Gn32Args a) {
  __shared__ float2 part[256 * 2];
  const int tid = threadIdx.x, b = blockIdx.x, slice = blockIdx.y;
  const int c4n = a.C / 4, cg = a.C / a.groups;
  const int npt = (c4n + 255) / 256;
  const int rows_per_iter = c4n >= 256 ? 1 : 256 / c4n;
  const int q0 = c4n >= 256 ? tid : tid % c4n, prow = c4n >= 256 ? 0 : tid / c4n;
  const bool active = c4n >= 256 || tid < rows_per_iter * c4n;
  const int stride = rows_per_iter * a.vslices;
  for (int k = 0; k < 2; ++k) {
    float s = 0.f, q = 0.f;
    const int c4 = q0 + k * 256;
    if (k < npt && c4 < c4n && active) {
      int p = prow + slice * rows_per_iter;
      for (; p + 3 * stride < a.P; p += 4 * stride) {          // four independent loads in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(a.in + ((long long)(p + u * stride) * a.B + b) * a.ldi + c4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (v[u].x + v[u].y) + (v[u].z + v[u].w); q += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w); }
      }
      for (; p < a.P; p += stride) {
        const float4 v = *reinterpret_cast<const float4*>(a.in + ((long long)p * a.B + b) * a.ldi + c4 * 4);
        s += (v.x + v.y) + (v.z + v.w); q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
    part[k * 256 + tid] = make_float2(s, q);
  }
  __syncthreads();
  if (tid < a.groups) {
    double s = 0.0, q = 0.0;
    const int lo = tid * cg / 4, hi = (tid + 1) * cg / 4;
    for (int k = 0; k < npt; ++k)
      for (int c4 = lo; c4 < hi; ++c4) {
        if (c4 < k * 256 || c4 >= (k + 1) * 256) continue;
        const int qq = c4 - k * 256;
        if (c4n >= 256) { const float2 v = part[k * 256 + qq]; s += v.x; q += v.y; }
        else for (int r = 0; r < rows_per_iter; ++r) { const float2 v = part[r * c4n + qq]; s += v.x; q += v.y; }
      }
    a.partial[((long long)b * a.vslices + slice) * a.groups + tid] = make_double2(s, q);
  }
}

__global__ __launch_bounds__(256) void gn32_apply_kernel(const Gn32Args a) {
  __shared__ float2 stat[64];
  const int tid = threadIdx.x, b = blockIdx.x, slice = blockIdx.y;
  const int c4n = a.C / 4, cg = a.C / a.groups;
  const int npt = (c4n + 255) / 256;
  const int rows_per_iter = c4n >= 256 ? 1 : 256 / c4n;
  const int q0 = c4n >= 256 ? tid : tid % c4n, prow = c4n >= 256 ? 0 : tid / c4n;
  const bool active = c4n >= 256 || tid < rows_per_iter * c4n;
  if (tid < a.groups) {
    double s = 0.0, q = 0.0;
    for (int sl = 0; sl < a.vslices; ++sl) { const double2 v = a.partial[((long long)b * a.vslices + sl) * a.groups + tid]; s += v.x; q += v.y; }
    const double n = (double)a.P * cg, mean = s / n;
    double var = q / n - mean * mean; var = var < 0.0 ? 0.0 : var;
    stat[tid] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
  }
  __syncthreads();
  const int stride = rows_per_iter * a.slices;
  for (int k = 0; k < npt; ++k) {
    const int c4 = q0 + k * 256;
    if (c4 >= c4n || !active) continue;
    const float2 st = stat[(c4 * 4) / cg];
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c4 * 4), be = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
    auto norm = [&](float4 v) {
      v.x = ((v.x - st.x) * st.y) * ga.x + be.x; v.y = ((v.y - st.x) * st.y) * ga.y + be.y;
      v.z = ((v.z - st.x) * st.y) * ga.z + be.z; v.w = ((v.w - st.x) * st.y) * ga.w + be.w;
      if (a.silu) { v.x = act_apply(ACT_SILU, v.x); v.y = act_apply(ACT_SILU, v.y); v.z = act_apply(ACT_SILU, v.z); v.w = act_apply(ACT_SILU, v.w); }
      return v;
    };
    int p = prow + slice * rows_per_iter;
    for (; p + 3 * stride < a.P; p += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(a.in + ((long long)(p + u * stride) * a.B + b) * a.ldi + c4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(a.out + ((long long)(p + u * stride) * a.B + b) * a.ldo + c4 * 4) = norm(v[u]);
    }
    for (; p < a.P; p += stride)
      *reinterpret_cast<float4*>(a.out + ((long long)p * a.B + b) * a.ldo + c4 * 4) = norm(*reinterpret_cast<const float4*>(a.in + ((long long)p * a.B + b) * a.ldi + c4 * 4));
  }
}

// Tensors of up to 16 K floats per (sample, group bundle) -- every GroupNorm(32) of configs 4/5: the 16x16 ... 2x2 levels of UNetModel --
// in ONE launch with ONE read: a workgroup owns sample b and `gpw` neighbouring groups (>= 128 contiguous bytes per pixel, so the reads
// are whole cache lines), keeps its elements in registers (<= 16 float4 per thread), takes the mean, then the centred sum of squares from
// the registers (no E[x^2] - mean^2 cancellation), normalises, applies SiLU and writes.  gpw depends on the tensor shape only, the sums
// are reduced in a fixed order: a sample's result depends neither on scheduling nor on the batch it is computed in.
__global__ __launch_bounds__(256) void gn32_reg_kernel(const Gn32Args a) {
  __shared__ float part[256];
  __shared__ float2 stat[8];
  const int tid = threadIdx.x, b = blockIdx.x, bun = blockIdx.y;
  const int cg = a.C / a.groups, cq = cg >> 2, W4 = a.gpw * cq, R = 256 / W4;
  const int q = tid % W4, r = tid / W4;
  const bool active = r < R;
  const int c0 = bun * a.gpw * cg + q * 4, gl = q / cq;
  const int E = cq * R;                                     // partial sums per group
  float4 v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = r + i * R;
    v[i] = (active && p < a.P) ? *reinterpret_cast<const float4*>(a.in + ((long long)p * a.B + b) * a.ldi + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const double inv_n = 1.0 / ((double)a.P * cg);
  auto group_sum = [&](float mine) -> double {            // sum of the `E` partials of group (tid >> 4), valid in lanes with (tid & 15) == 0
    part[tid] = active ? mine : 0.f;
    __syncthreads();
    double acc = 0.0;
    const int g = tid >> 4, l = tid & 15;
    if (g < a.gpw)
      for (int e = l; e < E; e += 16) acc += (double)part[(e / cq) * W4 + g * cq + e % cq];
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    return acc;
  };
  double t = group_sum(s);
  if ((tid & 15) == 0 && (tid >> 4) < a.gpw) stat[tid >> 4].x = (float)(t * inv_n);
  __syncthreads();
  const float mean = stat[gl].x;
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (active && r + i * R < a.P) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  __syncthreads();                                          // everyone has read its partial sums' successor slot
  t = group_sum(qq);
  if ((tid & 15) == 0 && (tid >> 4) < a.gpw) stat[tid >> 4].y = (float)(1.0 / sqrt(t * inv_n + 1e-5));
  __syncthreads();
  if (!active) return;
  const float rstd = stat[gl].y;
  const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c0), be = *reinterpret_cast<const float4*>(a.beta + c0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = r + i * R;
    if (p < a.P) {
      float4 y;
      y.x = ((v[i].x - mean) * rstd) * ga.x + be.x; y.y = ((v[i].y - mean) * rstd) * ga.y + be.y;
      y.z = ((v[i].z - mean) * rstd) * ga.z + be.z; y.w = ((v[i].w - mean) * rstd) * ga.w + be.w;
      if (a.silu) { y.x = act_apply(ACT_SILU, y.x); y.y = act_apply(ACT_SILU, y.y); y.z = act_apply(ACT_SILU, y.z); y.w = act_apply(ACT_SILU, y.w); }
      *reinterpret_cast<float4*>(a.out + ((long long)p * a.B + b) * a.ldo + c0) = y;
    }
  }
}

// groups narrower than 4 channels (toy widths only): one thread per (sample, group), serial over its elements
__global__ void gn32_small_kernel(const Gn32Args a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.groups) return;
  const int b = idx / a.groups, g = idx - b * a.groups, cg = a.C / a.groups;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < a.P; ++p) for (int c = g * cg; c < (g + 1) * cg; ++c) { const double v = a.in[((long long)p * a.B + b) * a.ldi + c]; s += v; q += v * v; }
  const double n = (double)a.P * cg, mean = s / n;
  double var = q / n - mean * mean; var = var < 0.0 ? 0.0 : var;
  const float m = (float)mean, r = (float)(1.0 / sqrt(var + 1e-5));
  for (int p = 0; p < a.P; ++p) for (int c = g * cg; c < (g + 1) * cg; ++c) {
    float y = ((a.in[((long long)p * a.B + b) * a.ldi + c] - m) * r) * a.gamma[c] + a.beta[c];
    if (a.silu) y = act_apply(ACT_SILU, y);
    a.out[((long long)p * a.B + b) * a.ldo + c] = y;
  }
}

// unet_openai QKVAttention core (unet_openai.py:346-358): one workgroup per (sample, head), T <= 64 tokens.
struct AttnQkvArgs { const float* qkv; float* out; int B, T, ldq, ldo, heads, chh; float scale; };

// Thread (t = tid >> 2, part = tid & 3) owns 16 of row t's 64 logits, then 16 of every 64 output channels of token t.  q, k (and later v)
// pass through LDS in 64-channel chunks (coalesced 256-byte token rows); the softmax of a row lives in the registers of its four
// threads.  Tokens / channels beyond T / chh are staged as zeros and masked, so any T <= 64 and any head width work.
__global__ __launch_bounds__(256) void attn_qkv_kernel(const AttnQkvArgs a) {
  constexpr int LD = 65;
  __shared__ float w[64 * LD];               // probabilities [t][u]
  __shared__ float qs[64 * LD], ks[64 * LD];  // one 64-channel chunk of q and of k (then of v): [token][channel]
  const int tid = threadIdx.x, tt = tid >> 2, part = tid & 3;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const float* base = a.qkv + (long long)b * a.ldq + h * 3 * a.chh;
  const long long ts = (long long)a.B * a.ldq;
  const float s2 = a.scale * a.scale;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < a.chh; c0 += 64) {
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int t = idx >> 6, c = idx & 63;
      const bool ok = t < a.T && c0 + c < a.chh;
      qs[t * LD + c] = ok ? base[t * ts + c0 + c] : 0.f;
      ks[t * LD + c] = ok ? base[t * ts + a.chh + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int c = 0; c < 64; ++c) {
      const float qv = qs[tt * LD + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(qv, ks[(part * 16 + j) * LD + c], acc[j]);
    }
    __syncthreads();
  }
  {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = part * 16 + j < a.T ? acc[j] * s2 : -INFINITY; m = fmaxf(m, acc[j]); }
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = expf(acc[j] - m); z += acc[j]; }
    z += __shfl_xor(z, 1); z += __shfl_xor(z, 2);
    const float inv = 1.0f / z;
#pragma unroll
    for (int j = 0; j < 16; ++j) w[tt * LD + part * 16 + j] = acc[j] * inv;
  }
  for (int c0 = 0; c0 < a.chh; c0 += 64) {
    __syncthreads();                           // w complete / previous chunk of v consumed
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int u = idx >> 6, c = idx & 63;
      ks[u * LD + c] = (u < a.T && c0 + c < a.chh) ? base[u * ts + 2 * a.chh + c0 + c] : 0.f;
    }
    __syncthreads();
    float o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
    for (int u = 0; u < 64; ++u) {
      const float p = w[tt * LD + u];
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = fmaf(p, ks[u * LD + part * 16 + j], o[j]);
    }
    if (tt < a.T) {
      float* dst = a.out + ((long long)tt * a.B + b) * a.ldo + h * a.chh + c0 + part * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) if (c0 + part * 16 + j < a.chh) dst[j] = o[j];
    }
  }
}

// Few tokens (T <= 8: the 2x2 / 1x1 middle block with one wide head): one block reduction over the channels per logit, T*T logits.
__global__ __launch_bounds__(256) void attn_qkv_small_kernel(const AttnQkvArgs a) {
  __shared__ float w[64 * 64];
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const float* base = a.qkv + (long long)b * a.ldq + h * 3 * a.chh;
  const long long ts = (long long)a.B * a.ldq;
  const float s2 = a.scale * a.scale;
  for (int pair = 0; pair < a.T * a.T; ++pair) {
    const int t = pair / a.T, u = pair - t * a.T;
    float acc = 0.f;
    for (int c = tid; c < a.chh; c += 256) acc = fmaf(base[t * ts + c], base[u * ts + a.chh + c], acc);
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) w[pair] = red[0] * s2;
    __syncthreads();
  }
  if (tid < a.T) {
    float m = -INFINITY;
    for (int u = 0; u < a.T; ++u) m = fmaxf(m, w[tid * a.T + u]);
    float z = 0.f;
    for (int u = 0; u < a.T; ++u) { const float e = expf(w[tid * a.T + u] - m); w[tid * a.T + u] = e; z += e; }
    const float inv = 1.0f / z;
    for (int u = 0; u < a.T; ++u) w[tid * a.T + u] *= inv;
  }
  __syncthreads();
  for (int c = tid; c < a.chh; c += 256)
    for (int t = 0; t < a.T; ++t) {
      float acc = 0.f;
      for (int u = 0; u < a.T; ++u) acc = fmaf(w[t * a.T + u], base[u * ts + 2 * a.chh + c], acc);
      a.out[((long long)t * a.B + b) * a.ldo + h * a.chh + c] = acc;
    }
}

// rowpack: element (y, x, c) goes to row y, channel x*Cv + c of a [Ho][B][ld] tensor (init_conv's row-packed input)
__global__ void nchw_to_pbc_kernel(const float* in, float* out, int B, int C, int Hn, int Wn, int pad_y, int pad_x, int Ho, int Wo, int ld, int Cv, int rowpack) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ho * Wo * B * Cv;
  if (idx >= total) return;
  const int c = idx % Cv; long long r = idx / Cv; const int b = r % B; const int p = r / B;
  const int y = p / Wo - pad_y, x = p % Wo - pad_x;
  float v = 0.f;
  if (c < C && y >= 0 && y < Hn && x >= 0 && x < Wn) v = in[(((long long)b * C + c) * Hn + y) * Wn + x];
  if (rowpack) out[((long long)(p / Wo) * B + b) * ld + (p % Wo) * Cv + c] = v;
  else out[((long long)p * B + b) * ld + c] = v;
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

// AvgPool2d(k), PBC -> PBC (encoders, h_vae_model_copy.py:52, 31): the k*k inputs are summed in row-major window order
__global__ void avgpool_kernel(const BilArgs o) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = o.C / 4, k = o.scale;
  const long long total = (long long)o.Ho * o.Wo * o.B * c4n;
  if (idx >= total) return;
  const int c = (idx % c4n) * 4; long long r = idx / c4n; const int b = r % o.B; const int p = r / o.B;
  const int oy = p / o.Wo, ox = p - oy * o.Wo;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(o.in + ((long long)((oy * k + i) * o.Wi + ox * k + j) * o.B + b) * o.ldi + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  const float d = (float)(k * k);
  *reinterpret_cast<float4*>(o.out + ((long long)p * o.B + b) * o.ldo + c) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
unsigned long long* g_gemm_dbg = nullptr;   // set by sbm_debug_gemm_stamps (diagnostics only)

// Resolve one GEMM op into its kernel arguments (done once per model, not per launch).
const char* make_gemm_args(const Plan& plan, const GemmOp& o, const float* prm, float* ws, const GemmCold* cold_dev, GemmHot* hot, GemmCold* cold) {
  if ((o.a.ld & 3) || (o.has2 && (o.a2.ld & 3))) return "seg_gemm: row pitches must be multiples of 4 floats";
  if (o.Kp <= 0 || (o.Kp & 31) || (o.has2 && (o.Kp2 <= 0 || (o.Kp2 & 31)))) return "seg_gemm: weight K stride must be a positive multiple of 32";
  if (o.a_ext && o.a.C != o.Kp) return "seg_gemm: an external input matrix must have a multiple of 32 columns";
  if ((long long)o.B * (o.a.ld > o.a2.ld ? o.a.ld : o.a2.ld) >= (1LL << 30) || (long long)o.Npad * (o.Kp > o.Kp2 ? o.Kp : o.Kp2) >= (1LL << 30))
    return "seg_gemm: operand plane too large for 32-bit lane offsets";
  if (o.hdr_off < 0 || o.desc_off < 0 || plan.ident_off < 0) return "seg_gemm: chunk tables missing";
  if (!((o.BM == 64 && o.BN == 64) || (o.BM == 64 && o.BN == 128) || (o.BM == 128 && o.BN == 64) || (o.BM == 128 && o.BN == 128) || (o.BM == 32 && o.BN == 32))) return "seg_gemm: unsupported tile";
  GemmHot h{};
  h.A = o.a_ext ? nullptr : ws + o.a.off;            // external input patched at launch
  h.A2 = o.has2 ? ws + o.a2.off : nullptr;
  h.W = prm + o.w_off; h.W2 = o.has2 ? prm + o.w2_off : nullptr;
  h.desc = reinterpret_cast<const int4*>(prm + o.desc_off); h.hdr = reinterpret_cast<const int4*>(prm + o.hdr_off);
  h.cold = cold_dev;
  h.out = o.out_nchw ? nullptr : ws + o.out.off;     // external output patched at launch
  h.B = o.B; h.lda = o.a.ld; h.lda2 = o.a2.ld; h.Kp = o.Kp; h.Kp2 = o.Kp2; h.Npad = o.Npad;
  const int nstore = o.out_nchw ? o.N : o.out.C;
  h.mtiles = ceil_div(o.B, o.BM); h.ntiles = ceil_div(nstore, o.BN);
  h.split = o.split > 1 ? o.split : 1;
  if (h.split > 1) {
    if (plan.splitk_part_off < 0 || plan.splitk_cnt_off < 0) return "seg_gemm: split-K scratch missing";
    h.part = ws + plan.splitk_part_off; h.cnt = reinterpret_cast<unsigned*>(ws + plan.splitk_cnt_off);
  }
  h.dbg = nullptr;
  GemmCold c{};
  c.bias = o.bias_off >= 0 ? prm + o.bias_off : nullptr;
  c.R = o.has_res ? ws + o.r.off : nullptr;
  c.gn_part = o.gn ? reinterpret_cast<const float2*>(ws + o.gn_part_off) : nullptr;
  c.gn_s = o.gn ? prm + o.gn_s_off : nullptr; c.gn_b1 = o.gn ? prm + o.gn_b1_off : nullptr;
  c.stat_out = o.stat_off >= 0 ? reinterpret_cast<float2*>(ws + o.stat_off) : nullptr;
  c.bias_pix_stride = o.bias_pix_stride; c.gn_nparts = o.gn_nparts; c.gn_inv_count = o.gn_inv_count;
  c.ldr = o.r.ld; c.act = o.act; c.ldo = o.out.ld; c.nstore = nstore; c.out_nchw = o.out_nchw; c.N = o.N;
  c.crop_y = o.crop_y; c.crop_x = o.crop_x; c.Hn = o.Hn; c.Wn = o.Wn; c.Wout = o.Wout;
  *hot = h; *cold = c;
  return nullptr;
}

DwArgs make_dw_args(const Plan& plan, const DwOp& o, const float* prm, float* ws, const StepCtl* ctl) {
  DwArgs d{};
  d.ctl = (o.has_cond && plan.cond_off >= 0) ? ctl : nullptr; d.cond_col = o.has_cond && plan.cond_off >= 0 ? (int)(o.cond_off - plan.cond_off) : 0;
  d.in = ws + o.in.off; d.out = ws + o.out.off; d.w = prm + o.w_off; d.bias = prm + o.bias_off;
  d.cond = o.has_cond ? ws + o.cond_off : nullptr; d.stat = o.stat_off >= 0 ? reinterpret_cast<float2*>(ws + o.stat_off) : nullptr;
  d.B = o.B; d.H = o.in.H; d.W = o.in.W; d.C = o.in.C; d.ldi = o.in.ld; d.ldo = o.out.ld; d.cond_ld = o.cond_ld;
  d.lanes = o.in.C / 4 < 256 ? o.in.C / 4 : 256; d.spb = 256 / d.lanes; d.slabs = o.stat_slabs;
  return d;
}
AttnArgs make_attn_args(const AttnOp& o, float* ws) {
  AttnArgs a{};
  a.qkv = ws + o.qkv.off; a.out = ws + o.out.off; a.B = o.B; a.n = o.qkv.H * o.qkv.W; a.ldq = o.qkv.ld; a.ldo = o.out.ld; a.heads = o.heads; a.scale = o.scale;
  return a;
}
GnResArgs make_gnres_args(const GnResOp& o, const float* prm, float* ws) {
  GnResArgs g{};
  g.u = ws + o.u.off; g.x = ws + o.x.off; g.out = ws + o.out.off; g.part = reinterpret_cast<const float2*>(ws + o.part_off);
  g.gamma = prm + o.gamma_off; g.beta = prm + o.beta_off; g.B = o.B; g.P = o.u.H * o.u.W; g.C = o.u.C;
  g.ldu = o.u.ld; g.ldx = o.x.ld; g.ldo = o.out.ld; g.nparts = o.nparts; g.inv_count = o.inv_count;
  return g;
}

static const char* launch_gemm(const GemmOp& o, GemmHot g, const float* ext_in, float* ext_out, hipStream_t st) {
  if (o.a_ext) g.A = ext_in + o.a.off;
  if (o.out_nchw) g.out = ext_out;
  g.dbg = g_gemm_dbg;
  const long long blocks = (long long)o.Hout * o.Wout * g.mtiles * g.ntiles * g.split;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return "seg_gemm: bad grid";
  dim3 grid((unsigned)blocks), block(256);
  const bool gn = o.gn != 0;
#define SBM_LAUNCH(BM_, BN_, NST_) \
  do { if (gn) SBM_KLAUNCH((seg_gemm<BM_, BN_, NST_, true>), grid, block, 0, st, g); \
       else SBM_KLAUNCH((seg_gemm<BM_, BN_, NST_, false>), grid, block, 0, st, g); } while (0)
  if (o.BM == 64 && o.BN == 64) SBM_LAUNCH(64, 64, 4);
  else if (o.BM == 64 && o.BN == 128) SBM_LAUNCH(64, 128, 3);
  else if (o.BM == 128 && o.BN == 64) SBM_LAUNCH(128, 64, 4);
  else SBM_LAUNCH(128, 128, 3);
#undef SBM_LAUNCH
  return nullptr;
}

// Dynamic-LDS limits of the per-layer kernels that need more than the default 64 KB window: set at model creation (per device, outside any
// stream capture), never at launch time.
void prepare_layer_kernels() {
  ensure_dynamic_lds(reinterpret_cast<const void*>(dwconv7_t8_kernel), 64 * 1024);
  ensure_dynamic_lds(reinterpret_cast<const void*>(linattn_lds4_kernel), 3 * 320 * 128);     // + 18 KB static: within the 160 KB of a CU
  ensure_dynamic_lds(reinterpret_cast<const void*>(attnblock_kernel), AB_LDS_BYTES);
}

const char* launch_op(const DevPlan& dp, int idx, const float* ext_in, float* ext_out, hipStream_t st) {
  const Plan& plan = *dp.plan; const Op& op = plan.ops[idx]; const float* prm = dp.prm; float* ws = dp.ws;
  if (op.chain) return launch_chain(dp, idx, 1, st, ext_out);
  if (op.row) return launch_rowrun(dp, idx, 1, st);
  switch (op.kind) {
    case OP_GEMM: return launch_gemm(op.g, dp.hot[idx], ext_in, ext_out, st);
    case OP_DWCONV7: {
      const DwOp& o = op.dw; const DwArgs d = make_dw_args(plan, o, prm, ws, dp.ctl);
      if (o.stat_slabs > 1) {                       // plan-time choice (16 x 16, channels a multiple of 32): the statistics layout depends on it
        if (o.in.W != 16 || o.in.H != 16 || (o.in.C + 31) / 32 != o.stat_slabs || (o.in.C & 3)) return "dwconv7: slab statistics need a 16x16 tensor in slabs of 32 channels";
        SBM_KLAUNCH(dwconv7_t16_kernel, dim3(o.B, o.stat_slabs), dim3(256), 0, st, d);
      } else if (o.in.W == 8 && o.in.H == 8 && o.in.C / 4 <= 32 && !getenv("SBM_NO_DWT8")) {
        SBM_KLAUNCH(dwconv7_t8_kernel, dim3(o.B), dim3(256), (size_t)(64 + 49) * (o.in.C / 4) * 16, st, d);
      } else if (o.in.W == 8 && o.in.C / 4 <= 64 && o.in.C / 4 >= 4 && (256 % (o.in.C / 4)) == 0 && !getenv("SBM_NO_DW8")) {
        SBM_KLAUNCH(dwconv7_w8_kernel, dim3(o.in.H * 2, ceil_div(o.B, d.spb)), dim3(256), 0, st, d);
      } else {
        dim3 grid(o.in.H * o.in.W, ceil_div(o.B, d.spb));
        SBM_KLAUNCH(dwconv7_kernel, grid, dim3(256), 0, st, d);
      }
    } break;
    case OP_LINATTN: case OP_ATTN: {
      const AttnOp& o = op.at; const AttnArgs a = make_attn_args(o, ws);
      if (o.dh != 32) return "attention: dim_head must be 32";
      if (op.kind == OP_ATTN && a.n > 64) return "softmax attention: more than 64 positions";
      if (op.kind == OP_LINATTN && a.n > 16 && a.n <= 320 && (a.ldq & 3) == 0 && (a.n <= 64 || !getenv("SBM_NO_LINATTN_BIG"))) {
        SBM_KLAUNCH(linattn_lds4_kernel, dim3(o.B * o.heads), dim3(256), (size_t)3 * a.n * 128, st, a);
      }
      else if (op.kind == OP_LINATTN && a.n <= 64 && (a.ldq & 3) == 0) SBM_KLAUNCH(linattn_lds_kernel, dim3(o.B * o.heads), dim3(64), 0, st, a);
      else if (op.kind == OP_LINATTN) SBM_KLAUNCH(linattn_kernel, dim3(o.B * o.heads), dim3(64), 0, st, a);
      else SBM_KLAUNCH(attn_kernel, dim3(o.B * o.heads), dim3(64), 0, st, a);
    } break;
    case OP_GN_RES: {
      const GnResOp& o = op.gr; GnResArgs g = make_gnres_args(o, prm, ws);
      if (o.out_ext) g.out = ext_out;
      // few samples: cut the pixels into slices so that about 512 workgroups run (B = 64: one workgroup per sample left 3/4 of the chip idle)
      int slices = o.B >= 192 ? 1 : ceil_div(512, o.B);
      const int sweeps = ceil_div(g.P * (g.C / 4), 256);
      slices = slices > sweeps ? sweeps : slices; slices = slices > 16 ? 16 : (slices < 1 ? 1 : slices);
      SBM_KLAUNCH(gn_res_kernel, dim3(o.B, slices), dim3(256), 0, st, g);
    } break;
    case OP_SINUS: {
      const SinusOp& o = op.si; const int n = o.B * o.ld;
      SBM_KLAUNCH(sinus_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, ws + plan.time_off, prm + o.freq_off, ws + o.out_off, o.B, o.dim, o.ld, o.cos_first);
    } break;
    case OP_NCHW_TO_PBC: {
      const CvtOp& o = op.cv;
      const int Ho = o.out.H, Wo = o.rowpack ? o.Wo : o.out.W, Cv = o.rowpack ? o.Cpad : o.out.C;
      const long long n = (long long)Ho * Wo * o.B * Cv;
      SBM_KLAUNCH(nchw_to_pbc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ext_in, ws + o.out.off, o.B, o.C, o.Hn, o.Wn, o.pad_y, o.pad_x, Ho, Wo, o.out.ld, Cv, o.rowpack);
    } break;
    case OP_BILINEAR: {
      const BilinearOp& o = op.bl; BilArgs a{};
      a.in = ws + o.in.off; a.out = ws + o.out.off; a.B = o.B; a.Hi = o.in.H; a.Wi = o.in.W; a.Ho = o.out.H; a.Wo = o.out.W; a.C = o.in.C; a.ldi = o.in.ld; a.ldo = o.out.ld; a.scale = o.scale;
      const long long n = (long long)a.Ho * a.Wo * a.B * (a.C / 4);
      SBM_KLAUNCH(bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    } break;
    case OP_AVGPOOL: {
      const BilinearOp& o = op.bl; BilArgs a{};
      a.in = ws + o.in.off; a.out = ws + o.out.off; a.B = o.B; a.Hi = o.in.H; a.Wi = o.in.W; a.Ho = o.out.H; a.Wo = o.out.W; a.C = o.in.C; a.ldi = o.in.ld; a.ldo = o.out.ld; a.scale = o.scale;
      const long long n = (long long)a.Ho * a.Wo * a.B * (a.C / 4);
      SBM_KLAUNCH(avgpool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    } break;
    case OP_GN32: {
      const Gn32Op& o = op.gn; Gn32Args a{};
      a.in = ws + o.in.off; a.out = ws + o.out.off; a.gamma = prm + o.gamma_off; a.beta = prm + o.beta_off;
      a.B = o.B; a.P = o.in.H * o.in.W; a.C = o.in.C; a.ldi = o.in.ld; a.ldo = o.out.ld; a.groups = o.groups; a.silu = o.silu;
      if (o.in.C % o.groups || o.in.C > 2048 || o.groups > 64) return "gn32: C must be a multiple of the group count and <= 2048";
      {
        const int c4n = a.C / 4, rpi = c4n >= 256 ? 1 : 256 / c4n, sweeps = ceil_div(a.P, rpi);     // pixel sweeps of one workgroup
        int sl = ceil_div(512, o.B);                                                                  // aim at >= 512 workgroups
        a.slices = sl < 1 ? 1 : (sl > sweeps ? sweeps : (sl > 16 ? 16 : sl));
      }
      const bool two_pass = (long long)a.P * a.C * 4 >= 96 * 1024 && plan.gn32_part_off >= 0 && !getenv("SBM_GN32_ONE_PASS");
      a.partial = two_pass ? reinterpret_cast<double2*>(ws + plan.gn32_part_off) : nullptr;
      {
        const int c4n = a.C / 4, rpi = c4n >= 256 ? 1 : 256 / c4n, sweeps = ceil_div(a.P, rpi);
        a.vslices = sweeps < 16 ? sweeps : 16;                  // a function of the tensor shape only
      }
      // register-resident single pass (gn32_reg_kernel): a bundle of gpw groups = the fewest whole groups that cover a 128-byte line
      a.gpw = 0;
      if (o.in.C % (4 * o.groups) == 0) {
        static const bool reg_on = [] { const char* e = getenv("SBM_GN32_REG"); return !e || atoi(e) != 0; }();
        const int cg = a.C / o.groups;
        int gpw = 1; while (gpw * cg < 32 && gpw < 8 && o.groups % (gpw * 2) == 0) gpw *= 2;
        const int W4 = gpw * cg / 4, R = W4 <= 256 ? 256 / W4 : 0;
        if (reg_on && R > 0 && ceil_div(a.P, R) <= 16 && o.groups % gpw == 0) a.gpw = gpw;
      }
      if (o.in.C % (4 * o.groups)) SBM_KLAUNCH(gn32_small_kernel, dim3(ceil_div(o.B * o.groups, 256)), dim3(256), 0, st, a);
      else if (a.gpw) SBM_KLAUNCH(gn32_reg_kernel, dim3(o.B, o.groups / a.gpw), dim3(256), 0, st, a);
      else if (two_pass) {
        if (g_prof_start) {      // profiling: one event pair spans both launches
          hipExtLaunchKernelGGL(gn32_stats_kernel, dim3(o.B, a.vslices), dim3(256), 0, st, g_prof_start, nullptr, 0, a);
          hipExtLaunchKernelGGL(gn32_apply_kernel, dim3(o.B, a.slices), dim3(256), 0, st, nullptr, g_prof_stop, 0, a);
        } else {
          hipLaunchKernelGGL(gn32_stats_kernel, dim3(o.B, a.vslices), dim3(256), 0, st, a);
          hipLaunchKernelGGL(gn32_apply_kernel, dim3(o.B, a.slices), dim3(256), 0, st, a);
        }
      } else SBM_KLAUNCH(gn32_kernel, dim3(o.B, a.slices), dim3(256), 0, st, a);
    } break;
    case OP_ATTN_QKV: {
      const AttnQkvOp& o = op.aq; AttnQkvArgs a{};
      a.qkv = ws + o.qkv.off; a.out = ws + o.out.off; a.B = o.B; a.T = o.qkv.H * o.qkv.W; a.ldq = o.qkv.ld; a.ldo = o.out.ld; a.heads = o.heads; a.chh = o.chh; a.scale = o.scale;
      if (a.T > 64) return "attention: more than 64 positions";
      if (a.T <= 8) SBM_KLAUNCH(attn_qkv_small_kernel, dim3(o.B * o.heads), dim3(256), 0, st, a);
      else SBM_KLAUNCH(attn_qkv_kernel, dim3(o.B * o.heads), dim3(256), 0, st, a);
    } break;
    default: return "unknown op";
  }
  return nullptr;
}

bool qkv_linattn_fusable(const Plan& plan, int i) {
  if (getenv("SBM_NO_QKV_FUSE") || i + 1 >= (int)plan.ops.size()) return false;
  const Op& a = plan.ops[i]; const Op& b = plan.ops[i + 1];
  if (a.kind != OP_GEMM || b.kind != OP_LINATTN || a.chain || b.chain) return false;
  const GemmOp& g = a.g; const AttnOp& t = b.at;
  if (g.Hout * g.Wout != 64 || g.KH != 1 || g.KW != 1 || g.Kp != 64 || !g.gn || g.has2 || g.has_res || g.act != ACT_NONE || g.a_ext || g.out_nchw) return false;
  if (g.up != 1 || g.stride != 1 || g.w_pix_stride || g.bias_pix_stride || g.stat_off >= 0) return false;
  if (t.dh != 32 || t.qkv.off != g.out.off || t.qkv.H * t.qkv.W != 64 || g.out.C != 3 * t.heads * 32 || (g.a.ld & 3)) return false;
  return true;
}

const char* launch_qkv_linattn(const DevPlan& dp, int i, hipStream_t st) {
  const Plan& plan = *dp.plan; const GemmOp& g = plan.ops[i].g; const AttnOp& t = plan.ops[i + 1].at;
  QkvAttnArgs q{};
  q.a = make_attn_args(t, dp.ws);
  q.A = dp.ws + g.a.off; q.lda = g.a.ld; q.W = dp.prm + g.w_off; q.bias = g.bias_off >= 0 ? dp.prm + g.bias_off : nullptr;
  q.gn_part = reinterpret_cast<const float2*>(dp.ws + g.gn_part_off); q.gn_nparts = g.gn_nparts; q.gn_inv = g.gn_inv_count;
  q.gn_s = dp.prm + g.gn_s_off; q.gn_b1 = dp.prm + g.gn_b1_off; q.Npad = g.Npad;
  SBM_KLAUNCH(linattn_qkv_kernel, dim3(t.B * t.heads), dim3(256), 0, st, q);
  return nullptr;
}

// ops i .. i + 3 = to_qkv, LinearAttention core, to_out.0, GroupNorm + residual of an 8x8 attention block with 64 channels and 4 heads:
// one attnblock_kernel launch (the plan keeps the last two out of the pixel kernel's run for it: plan.cpp attn_block_on)
bool attn_block_fusable(const Plan& plan, int i) {
  if (!attn_block_on() || !qkv_linattn_fusable(plan, i) || i + 3 >= (int)plan.ops.size()) return false;
  const Op& o2 = plan.ops[i + 2]; const Op& o3 = plan.ops[i + 3];
  if (o2.kind != OP_GEMM || o3.kind != OP_GN_RES || o2.chain || o2.row || o3.chain || o3.row) return false;
  const GemmOp& q = plan.ops[i].g; const AttnOp& t = plan.ops[i + 1].at; const GemmOp& g = o2.g; const GnResOp& r = o3.gr;
  if (t.heads != 4 || q.Npad != 384 || q.a.C != 64) return false;
  if (g.Hout * g.Wout != 64 || g.KH != 1 || g.KW != 1 || g.Kp != 128 || g.Npad != 64 || g.N > 64 || g.gn || g.has2 || g.has_res || g.act != ACT_NONE || g.a_ext || g.out_nchw) return false;
  if (g.up != 1 || g.stride != 1 || g.w_pix_stride || g.bias_pix_stride || g.stat_off < 0 || g.a.off != t.out.off || g.a.H * g.a.W != 64) return false;
  if (r.out_ext || r.u.off != g.out.off || r.part_off != g.stat_off || r.x.off != q.a.off || r.x.ld != q.a.ld || r.u.C != 64 || r.u.H * r.u.W != 64) return false;
  return true;
}

// The block's weights in the order its lanes read them, so that every wave-level load is 1 KB contiguous (as packed [n][k] matrices a wave's
// float4 load touched 32 rows 256 B apart: the first barrier of the kernel came 6.4 us after its start, mostly waiting for the TA).
//   to_qkv : [head h][nt: q | k | v][j = 0 .. 7][lane = 32 half + d][4] = W[128 nt + 32 h + d][32 (j >> 2) + 16 half + 4 (j & 3) ..]
//   to_out.0: [column tile nt][j = 0 .. 7][lane = 16 kq + l][4] = Wo[16 nt + l][32 kq + 4 j ..]
void pack_attn_block_fragments(Plan* plan, int i) {
  GemmOp& g = plan->ops[i].g; GemmOp& go = plan->ops[i + 2].g;
  std::vector<float> fq((size_t)4 * 3 * 8 * 64 * 4, 0.f), fo((size_t)4 * 8 * 64 * 4, 0.f);
  const float* W = plan->params.data() + g.w_off; const float* Wo = plan->params.data() + go.w_off;
  for (int h = 0; h < 4; ++h) for (int nt = 0; nt < 3; ++nt)
    for (int j = 0; j < 8; ++j) for (int lane = 0; lane < 64; ++lane) {
      const int d = lane & 31, half = lane >> 5, nrow = nt * 128 + h * 32 + d, k = 32 * (j >> 2) + 16 * half + 4 * (j & 3);
      memcpy(&fq[((((size_t)(h * 3 + nt)) * 8 + j) * 64 + lane) * 4], W + (size_t)nrow * 64 + k, 16);
    }
  for (int nt = 0; nt < 4; ++nt) for (int j = 0; j < 8; ++j) for (int lane = 0; lane < 64; ++lane) {
    const int l = lane & 15, kq = lane >> 4;
    memcpy(&fo[(((size_t)nt * 8 + j) * 64 + lane) * 4], Wo + (size_t)(16 * nt + l) * 128 + 32 * kq + 4 * j, 16);
  }
  auto append = [&](const std::vector<float>& v) { while (plan->params.size() & 3) plan->params.push_back(0.f); const int64_t off = (int64_t)plan->params.size(); plan->params.insert(plan->params.end(), v.begin(), v.end()); return off; };
  g.wt_off = append(fq); go.wt_off = append(fo);
}

const char* launch_attn_block(const DevPlan& dp, int i, hipStream_t st) {
  const Plan& plan = *dp.plan; const GemmOp& g = plan.ops[i].g; const AttnOp& t = plan.ops[i + 1].at; const GemmOp& go = plan.ops[i + 2].g; const GnResOp& r = plan.ops[i + 3].gr;
  AttnBlockArgs ab{};
  QkvAttnArgs& q = ab.q;
  q.a = make_attn_args(t, dp.ws);
  q.A = dp.ws + g.a.off; q.lda = g.a.ld; q.W = dp.prm + g.w_off; q.bias = g.bias_off >= 0 ? dp.prm + g.bias_off : nullptr;
  q.gn_part = reinterpret_cast<const float2*>(dp.ws + g.gn_part_off); q.gn_nparts = g.gn_nparts; q.gn_inv = g.gn_inv_count;
  q.gn_s = dp.prm + g.gn_s_off; q.gn_b1 = dp.prm + g.gn_b1_off; q.Npad = g.Npad;
  if (g.wt_off < 0 || go.wt_off < 0) return "attention block: no fragment-order weights (model creation packs them)";
  q.W = dp.prm + g.wt_off;
  ab.Wo = dp.prm + go.wt_off; ab.bo = go.bias_off >= 0 ? dp.prm + go.bias_off : nullptr;
  ab.gamma = dp.prm + r.gamma_off; ab.beta = dp.prm + r.beta_off;
  ab.out = dp.ws + r.out.off; ab.ldo = r.out.ld; ab.inv_count = r.inv_count;
  static unsigned long long* d_dbg = [] { unsigned long long* p = nullptr; if (std::getenv("SBM_ATTNBLOCK_STAMPS")) (void)hipMalloc(&p, 16 * sizeof(unsigned long long)); return p; }();
  ab.dbg = d_dbg;
  SBM_KLAUNCH(attnblock_kernel, dim3(t.B), dim3(1024), AB_LDS_BYTES, st, ab);
  if (d_dbg) {                                                   // diagnostics only (synchronises: not under stream capture)
    unsigned long long hbuf[16];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hbuf, d_dbg, sizeof(hbuf), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "attnblock stamps (s_memtime at ~2.1 GHz):");
      for (int i = 1; i < 16; ++i) if (hbuf[i]) fprintf(stderr, " [%d] %.2f us", i, (double)(hbuf[i] - hbuf[0]) / 2100.0);
      fprintf(stderr, "\n");
    }
  }
  return nullptr;
}

const char* run_plan(const DevPlan& dp, const float* ext_in, const float* t_dev, float* ext_out, hipStream_t st, int time_ops, const StepArgs* em) {
  const Plan& plan = *dp.plan;
  if (plan.time_off >= 0 && t_dev && t_dev != dp.ws + plan.time_off)
    if (hipMemcpyAsync(dp.ws + plan.time_off, t_dev, sizeof(float) * plan.B, hipMemcpyDeviceToDevice, st) != hipSuccess) return "time copy failed";
  if (plan.aux_off >= 0 && dp.aux)
    if (hipMemcpyAsync(dp.ws + plan.aux_off, dp.aux, sizeof(float) * plan.B * plan.aux_C, hipMemcpyDeviceToDevice, st) != hipSuccess) return "aux copy failed";
  bool em_done = false;
  const bool skip_cvt = (time_ops & 4) != 0;        // the caller keeps the converted input current itself (sampler loop)
  time_ops &= 3;
  auto skipped = [&](int i) { return (time_ops == 1 && plan.ops[i].time_path) || (time_ops == 2 && !plan.ops[i].time_path) || (skip_cvt && plan.ops[i].kind == OP_NCHW_TO_PBC); };
  for (int i = 0; i < (int)plan.ops.size();) {
    int n = 1;
    const char* e;
    if (skipped(i)) { ++i; continue; }
    if (plan.ops[i].fuse_next == 2) {               // a whole 8x8 attention block (four ops) as one kernel
      e = launch_attn_block(dp, i, st);
      if (e) return e;
      i += 4;
      continue;
    }
    if (plan.ops[i].fuse_next == 1) {               // to_qkv + LinearAttention core as one kernel
      e = launch_qkv_linattn(dp, i, st);
      if (e) return e;
      i += 2;
      continue;
    }
    if (plan.ops[i].chain) {                        // a maximal run of chain ops is ONE launch
      while (i + n < (int)plan.ops.size() && plan.ops[i + n].chain && !skipped(i + n)) ++n;
      const bool ends = i + n == (int)plan.ops.size();
      if (ends) em_done = em != nullptr;
      e = launch_chain(dp, i, n, st, ext_out, ends ? em : nullptr);
    } else if (plan.ops[i].row) {                   // ... and so is a run of row-kernel ops (the one-position level)
      while (i + n < (int)plan.ops.size() && plan.ops[i + n].row && !skipped(i + n)) ++n;
      e = launch_rowrun(dp, i, n, st);
    } else {
      e = launch_op(dp, i, ext_in, ext_out, st);
    }
    if (e) return e;
    i += n;
  }
  if (em && !em_done) return "run_plan: the sampler's update was to ride on the last launch, which is not a pixel-kernel run";
  if (hipGetLastError() != hipSuccess) return "kernel launch failed";
  return nullptr;
}

}  // namespace sbm
