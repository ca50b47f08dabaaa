// chain.hip -- the XCD-resident chain kernel: a run of consecutive low-resolution ops of the score net (the 2x2 and 1x1
// levels: ~50 of the 95 ops, each far too small to fill the chip) executed by ONE launch.
//
// Why it is possible: nothing in the network mixes samples, so a tile of 32 samples can be carried through any number of
// layers by a fixed set of workgroups with no grid-wide synchronisation.  Why it is per XCD: an MI355X has 8 XCDs with
// private L2s that are coherent only at kernel boundaries; data stored by one CU is visible to another CU of the SAME XCD
// as soon as it has reached that L2 (measured: tools/xcd_sync_probe.hip -- plain stores + `s_waitcnt vmcnt(0)` on the
// producer, `sc1` loads that bypass the consumer's L1; across XCDs the same sequence reads stale data).  So:
//
//   * every workgroup reads its XCC_ID and registers with the cluster of that XCD (membership is decided by where the
//     hardware actually placed it, never assumed from blockIdx);
//   * cluster c carries sample tiles c, c + nclusters, ... (B = 256 -> 8 tiles, one per XCD) through all ops of the run;
//   * after each op the cluster's workgroups meet at a barrier on a per-XCD counter (~0.8 us, vs ~5 us for a kernel
//     boundary plus the pipeline fill/drain of a tiny launch);
//   * activations written inside the launch are read with sc1 loads; weights and tables with ordinary cached loads.
//
// A barrier that does not complete within ~0.2 s sets ChainSync::abort_flag and every workgroup leaves: the grid always drains.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "device.h"

namespace sbm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DEVINL __device__ __forceinline__

namespace {

DEVINL float chain_act(int act, float v) {
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

DEVINL unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

// Pointers reach the kernel through descriptor structs, i.e. as generic pointers: loads through them would be FLAT
// instructions, which count on vmcnt AND lgkmcnt and force the compiler to drain everything (`s_waitcnt vmcnt(0)
// lgkmcnt(0)`) before each use -- no prefetch would ever overlap.  Casting to the global address space gives
// global_load (vmcnt only, returned in order), so the waits become counted.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4n gfloat4;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4 const_i32x4;
DEVINL float4 ldw4(const float* p) { const f32x4n v = *(const gfloat4*)(unsigned long long)p; return make_float4(v.x, v.y, v.z, v.w); }   // cached global load (weights, tables)
DEVINL void st4(float* p, float4 v) { f32x4n t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(gfloat4*)(unsigned long long)p = t; }
DEVINL int4 sload16(const int4* p) { const i32x4 v = *(const_i32x4*)(unsigned long long)p; return make_int4(v.x, v.y, v.z, v.w); }   // uniform: s_load_dwordx4

// L1-bypassing loads (global_load ... sc1): data another CU of this XCD stored earlier in the same launch
DEVINL float ld1(const float* p) { return __hip_atomic_load((gfloat*)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL float2 ld2(const float* p) {
  const unsigned long long v = __hip_atomic_load((gu64*)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
DEVINL float4 ld4(const float* p) { const float2 a = ld2(p), b = ld2(p + 2); return make_float4(a.x, a.y, b.x, b.y); }

DEVINL unsigned ctr_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL unsigned ctr_load(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL void ctr_zero(unsigned* p) { (void)__hip_atomic_exchange(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int A_LD = 36;                       // LDS row stride (floats) of a wave's 32x32 A tile

// Op descriptors live in LDS (staged once per launch); copying one through readfirstlane puts it in SGPRs.
template <class T> DEVINL T uniform_copy(const T& src) {
  static_assert(sizeof(T) % 4 == 0, "word-sized struct");
  struct Words { int w[sizeof(T) / 4]; };
  Words in, out;
  __builtin_memcpy(&in, &src, sizeof(T));                      // (memcpy, not pointer casts: no aliasing assumptions)
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) out.w[i] = __builtin_amdgcn_readfirstlane(in.w[i]);
  T dst;
  __builtin_memcpy(&dst, &out, sizeof(T));
  return dst;
}

template <class T> DEVINL T* uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}


// NT threads per workgroup, GROUPS independent sample groups per XCD.  GROUPS = 1: one 512-thread workgroup per CU carries a
// 32-sample tile per XCD.  GROUPS = 2: two 256-thread workgroups per CU; the XCD's workgroups split into two clusters that
// carry 16-sample tiles independently of each other, so that one cluster's barrier waits / L2 round trips / reductions are
// covered by the other's arithmetic (the chain is latency-bound: matrix pipe busy 12 %, waves parked 68 % with GROUPS = 1).
template <int NT, int GROUPS>
struct ChainImpl {
static constexpr int NW = NT / 64;             // waves per workgroup
static constexpr int TROWS = 32 / GROUPS;      // samples per tile (the MFMA tile stays 32 rows; rows >= TROWS are never stored)
static constexpr int SMEM_F = NW * 32 * A_LD;  // floats: per-wave A tiles are the largest user (linattn: NW*1120, GEMM reduce: NW*32*33)

struct Ctx {
  int tid, lane, wave;
  int slot, cs;          // this workgroup's index within its cluster, cluster size
  int b0, B;             // first sample of the tile being carried, batch size
  float* smem; float2* mr; unsigned* flag;
  unsigned* bar; unsigned gen; ChainSync* sync;
  unsigned spin_limit; unsigned* abort_host; bool skip_arrive; bool no_prefetch;
  unsigned long long* tr;   // diagnostics: per-op trace slots of the stamping workgroup (null otherwise)
};
#define TR(i) do { if (c.tr && c.tid == 0) c.tr[i] = __builtin_amdgcn_s_memtime(); } while (0)

// Weights (and small tables) of the NEXT op, touched once by the cluster's threads while they wait at the barrier: every XCD
// reads every weight exactly once per forward, always from beyond its own L2, so without this each op would start with a
// serial chain of cold misses (descriptor -> weights).
// The loaded words are folded into a value the caller consumes after the barrier, so the loads are ordinary tracked loads
// (a load whose destination register the compiler believes dead would land in whatever lives there by then).
// a barrier timed out: flag it for every workgroup (device copy) and for the host (host-mapped copy, read by the API without a
// synchronisation at the entry of the next call on this model)
static DEVINL void raise_abort(const Ctx& c) {
  (void)__hip_atomic_exchange(&c.sync->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (c.abort_host) __hip_atomic_store(c.abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static DEVINL unsigned prefetch_next(const ChainOp& nx, const Ctx& c) {
  unsigned sink = 0;
  if (nx.kind != OP_GEMM) return sink;
  const ChainGemm& m = nx.gm;
  const float* W = uniform_ptr(m.g.W);
  const float* W2 = uniform_ptr(m.g.W2);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)m.tapmask);
  const int tapf = __builtin_amdgcn_readfirstlane(m.tap_floats), w2f = __builtin_amdgcn_readfirstlane(m.w2_floats);
  const int lpt = tapf >> 5, ntap = __builtin_popcount(mask);          // 128-byte lines per tap block
  const int total = ntap * lpt + (w2f >> 5);
  const int stride = c.cs * NT;
  for (int L = c.slot * NT + c.tid; L < total; L += stride) {
    if (L < ntap * lpt) {
      int k = L / lpt; const int r = L - k * lpt;
      unsigned mm = mask; int tap = 0;
      while (true) { const int b = __builtin_ctz(mm); if (k == 0) { tap = b; break; } mm &= mm - 1; --k; }
      sink ^= __float_as_uint(ld1(W + (long long)tap * tapf + (long long)r * 32));
    } else {
      sink ^= __float_as_uint(ld1(W2 + (long long)(L - ntap * lpt) * 32));
    }
  }
  return sink;
}

// Every wave first waits until its own stores have been acknowledged by L2 -- explicitly: __syncthreads() alone is a
// workgroup-scope fence, which on this target does NOT wait for vmcnt (the waves of a workgroup share one L1) -- then one
// thread announces the workgroup and polls the XCD's counter, while all threads touch the next op's weights.
static DEVINL bool cluster_barrier(Ctx& c, const ChainOp* next) {
  c.gen += (unsigned)c.cs;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (c.tid == 0 && !c.skip_arrive) ctr_add(c.bar, 1u);
  const unsigned sink = (next && !c.no_prefetch) ? prefetch_next(*next, c) : 0u;
  if (c.tid == 0) {
    unsigned spins = 0, ok = 1;
    while (ctr_load(c.bar) < c.gen) {
      if (++spins > c.spin_limit) { raise_abort(c); ok = 0; break; }
      if ((spins & 1023u) == 0 && ctr_load(&c.sync->abort_flag)) { ok = 0; break; }
    }
    *c.flag = ok;
  }
  __syncthreads();
  if (sink == 0x9e3779b9u) c.flag[-1] = 1;                    // consumes the prefetched words (never true in practice; harmless if it is)
  return *c.flag != 0;
}

// ------------------------------------------------------------------------------------------------
// GEMM: unit = (output pixel, 32-column block) of the 32-sample tile; the 4 waves split the chunk list (K), operands go
// straight from L2 to registers (lane (i, h) of v_mfma_f32_32x32x2_f32 takes k = 16 h + j for the j-th MFMA of a chunk, on
// both operands), partial accumulators meet in LDS.  Same chunk tables, folded GroupNorm and epilogue as seg_gemm.
// ------------------------------------------------------------------------------------------------
// One K chunk (32 wide) of a wave: the A tile (32 samples x 32 k) is fetched COALESCED -- lane l takes 8 bytes of row
// (l >> 4) + 4 i, sixteen lanes per 128-byte row, so a wave-instruction asks L2 for 4 whole lines (L1 is bypassed: a
// per-lane-row pattern would ask for the same line once per instruction, and the request rate, not the bandwidth, was the
// limit) -- and is turned into MFMA fragments through a wave-private LDS tile.  B (weights) goes straight to registers:
// its lines stay in L1 between the four instructions that walk them.
static constexpr int NA8 = TROWS / 4;           // 8-byte A loads per lane and chunk (16 lanes per row, 4 rows per instruction)
static constexpr int NA16 = TROWS / 8;          // 16-byte A loads per lane and chunk (8 lanes per row, 8 rows per instruction)
struct Frag { float2 araw[NA8]; float w[16]; };

static DEVINL void load_frag(Frag& f, const GemmHot& g, const int4 d, int b0, int B, int lane, int brow, int fh) {
  const bool s2 = (d.w & 1) != 0;
  const int lda = s2 ? g.lda2 : g.lda;
  const float* ap = (s2 ? g.A2 : g.A) + d.x + (lane & 15) * 2;
  const float* wp = (s2 ? g.W2 : g.W) + d.y + (long long)brow * (s2 ? g.Kp2 : g.Kp) + 16 * fh;
#pragma unroll
  for (int i = 0; i < NA8; ++i) f.araw[i] = ld2(ap + (long long)min(b0 + (lane >> 4) + 4 * i, B - 1) * lda);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 v = ldw4(wp + 4 * j); f.w[4 * j] = v.x; f.w[4 * j + 1] = v.y; f.w[4 * j + 2] = v.z; f.w[4 * j + 3] = v.w; }
}

// araw -> this wave's LDS tile -> a[16] = A[row fi][16 fh + j]
static DEVINL void stage_frag(const Frag& f, float* tile, int lane, float* a) {
#pragma unroll
  for (int i = 0; i < NA8; ++i) *reinterpret_cast<float2*>(tile + ((lane >> 4) + 4 * i) * A_LD + (lane & 15) * 2) = f.araw[i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave: LDS executes in order; keep the compiler from reordering
  const float* rp = tile + (lane & 31) * A_LD + 16 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 v = *reinterpret_cast<const float4*>(rp + 4 * j); a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the tile may be overwritten by the next chunk
}

// Hand-counted variant for the K-split loop: B first (4 cached 16-byte loads), then A (4 L1-bypassing 16-byte loads, eight
// lanes per 128-byte row -- 8-byte sc1 loads run at 0.54-0.70x the 16-byte rate), all issued by inline asm so that the wait in
// front of a chunk is exactly "everything but the next chunk's eight loads" instead of whatever the compiler's scoreboard
// makes of a loop with conditional refills.  The fragments stay in their slots (never copied) until wait_fragm names them.
struct FragM { f32x4n a[NA16]; f32x4n w[4]; };

static DEVINL void load_fragm(FragM& f, const GemmHot& g, const int4 d, int b0, int B, int lane, int brow, int fh) {
  const bool s2 = (d.w & 1) != 0;
  const int lda = s2 ? g.lda2 : g.lda;
  const float* ap = (s2 ? g.A2 : g.A) + d.x + (lane & 7) * 4;
  const float* wp = (s2 ? g.W2 : g.W) + d.y + (long long)brow * (s2 ? g.Kp2 : g.Kp) + 16 * fh;
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.w[j]) : "v"(wp + 4 * j) : "memory");
#pragma unroll
  for (int i = 0; i < NA16; ++i) {
    const float* p = ap + (long long)min(b0 + (lane >> 3) + 8 * i, B - 1) * lda;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(f.a[i]) : "v"(p) : "memory");
  }
}
// younger: the next chunk's eight loads were issued after this chunk's and may stay in flight
// younger: how many chunks issued after this one (8 loads each) may stay in flight
static DEVINL void wait_fragm(FragM& f, int younger) {
  if constexpr (NA16 == 4) {
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(24)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
  } else {                                                    // 16-row tiles: six loads per chunk
    if (younger) asm volatile("s_waitcnt vmcnt(6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
  }
}
static DEVINL void stage_fragm(const FragM& f, float* tile, int lane, float* a) {
#pragma unroll
  for (int i = 0; i < NA16; ++i) *reinterpret_cast<f32x4n*>(tile + ((lane >> 3) + 8 * i) * A_LD + (lane & 7) * 4) = f.a[i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* rp = tile + (lane & 31) * A_LD + 16 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 v = *reinterpret_cast<const float4*>(rp + 4 * j); a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Short-K GEMMs (1x1 convolutions, the 1x1 level): one unit per WAVE, the whole K walked by that wave -- no K split, no
// cross-wave reduction, no workgroup barrier inside the unit loop, and up to 8 x 32 units in flight per cluster instead of 32.
static DEVINL void chain_gemm_wpu(const GemmHot& g, const GemmCold& cold, int P, bool gn, Ctx& c) {
  const int fi = c.lane & 31, fh = c.lane >> 5;
  const int nblk = (cold.nstore + 31) >> 5;
  const int units = P * nblk;
  float* tile = c.smem + c.wave * 32 * A_LD;                   // wave-private: A staging, then the output transpose
  float rr[16];
  if (gn) {
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) rr[r2] = c.mr[(r2 & 3) + 8 * (r2 >> 2) + 4 * fh].y;
  }
  constexpr int PF = 2;
  for (int u = c.slot * NW + c.wave; u < units; u += c.cs * NW) {
    const int p = u / nblk, nb = u - p * nblk, n0 = nb * 32;
    const int4 hd = sload16(g.hdr + p);
    const int4* __restrict__ dsc = g.desc + hd.x;
    const int total = hd.y, n1 = gn ? hd.z : 0;
    const int brow = min(n0 + fi, g.Npad - 1);
    // epilogue operands: lane owns 4 columns (c4) of rows er, er + 8, er + 16, er + 24
    const int er = c.lane >> 3, c4 = (c.lane & 7) * 4, n = n0 + c4;
    const bool ncol = n < cold.nstore;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gs = bv, gb = bv, rv[4] = {bv, bv, bv, bv};
    if (ncol) {
      if (cold.bias) bv = ldw4(cold.bias + (long long)p * cold.bias_pix_stride + n);
      if (gn) { gs = ldw4(cold.gn_s + (long long)p * g.Npad + n); gb = ldw4(cold.gn_b1 + (long long)p * g.Npad + n); }
      if (cold.R) {
#pragma unroll
        for (int t = 0; t < NA16; ++t) { const int bb = c.b0 + er + 8 * t; if (bb < c.B) rv[t] = ld4(cold.R + ((long long)hd.w * c.B + bb) * cold.ldr + n); }
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto run = [&](int lo, int hi) {
      if (lo >= hi) return;
      const int nk = hi - lo;
      Frag f[PF];
#pragma unroll
      for (int sI = 0; sI < PF; ++sI) if (sI < nk) load_frag(f[sI], g, sload16(dsc + lo + sI), c.b0, c.B, c.lane, brow, fh);
      for (int k = 0; k < nk; k += PF) {
#pragma unroll
        for (int sI = 0; sI < PF; ++sI) {
          if (k + sI < nk) {
            float av[16];
            stage_frag(f[sI], tile, c.lane, av);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], f[sI].w[j], acc, 0, 0, 0);
            if (k + sI + PF < nk) load_frag(f[sI], g, sload16(dsc + lo + k + sI + PF), c.b0, c.B, c.lane, brow, fh);
          }
        }
      }
    };
    if (gn) {
      run(0, n1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= rr[r];
      run(n1, total);
    } else {
      run(0, total);
    }
    // transpose through the wave's LDS tile (same wave: LDS is in order; the asm waits also stop compiler reordering)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * fh) * A_LD + fi] = acc[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < NA16; ++t) {
      const int row = er + 8 * t, bb = c.b0 + row;
      const float4 tv = *reinterpret_cast<const float4*>(tile + row * A_LD + c4);
      float v[4] = {tv.x, tv.y, tv.z, tv.w};
      float s = 0.f, q = 0.f;
      if (ncol && bb < c.B) {
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        if (gn) {
          const float2 mrv = c.mr[row]; const float mrs = mrv.x * mrv.y;
          v[0] += gb.x - mrs * gs.x; v[1] += gb.y - mrs * gs.y; v[2] += gb.z - mrs * gs.z; v[3] += gb.w - mrs * gs.w;
        }
        v[0] += rv[t].x; v[1] += rv[t].y; v[2] += rv[t].z; v[3] += rv[t].w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = chain_act(cold.act, v[e]);
        st4(g.out + ((long long)p * c.B + bb) * cold.ldo + n, make_float4(v[0], v[1], v[2], v[3]));
        s = (v[0] + v[1]) + (v[2] + v[3]); q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
      if (cold.stat_out) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if ((c.lane & 7) == 0 && bb < c.B) cold.stat_out[((long long)p * nblk + nb) * c.B + bb] = make_float2(s, q);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // tile reads done before the next unit stages into it
  }
}

static DEVINL void chain_gemm(const ChainGemm& m, int P, Ctx& c) {
  TR(0);
  const GemmHot g = uniform_copy(m.g);
  const GemmCold cold = uniform_copy(m.cold);
  const int fi = c.lane & 31, fh = c.lane >> 5;
  const int nblk = (cold.nstore + 31) >> 5;
  const bool gn = cold.gn_part != nullptr;
  float rr[16];
  if (gn) {
    const int r = c.tid & 31, pl = c.tid >> 5;                 // NT/32 = 16 threads per tile row
    const int bb = min(c.b0 + r, c.B - 1);
    double s = 0.0, q = 0.0;
    for (int i = pl; i < cold.gn_nparts; i += NT / 32) { const float2 v = ld2(reinterpret_cast<const float*>(cold.gn_part + (long long)i * c.B + bb)); s += (double)v.x; q += (double)v.y; }
    double* red = reinterpret_cast<double*>(c.smem);           // [NT/32][32][2]
    red[(pl * 32 + r) * 2] = s; red[(pl * 32 + r) * 2 + 1] = q;
    __syncthreads();
    if (c.tid < 32) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int k = 0; k < NT / 32; ++k) { ss += red[(k * 32 + c.tid) * 2]; qq += red[(k * 32 + c.tid) * 2 + 1]; }
      const double inv = (double)cold.gn_inv_count, mean = ss * inv;
      double var = qq * inv - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      c.mr[c.tid] = make_float2((float)mean, rsqrtf((float)var + 1e-5f));
    }
    __syncthreads();
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) rr[r2] = c.mr[(r2 & 3) + 8 * (r2 >> 2) + 4 * fh].y;
  }
  TR(1);
  if (__builtin_amdgcn_readfirstlane(m.wpu)) { TR(1); chain_gemm_wpu(g, cold, P, gn, c); TR(4); return; }
  float* atile = c.smem + c.wave * 32 * A_LD;
  const int units = P * nblk;
  const int row = (c.tid >> 3) & 31, c4 = (c.tid & 7) * 4;       // epilogue: threads 0..255 own 4 columns of one row
  const int* __restrict__ sched = uniform_ptr(m.sched);
  const int rounds = __builtin_amdgcn_readfirstlane(m.rounds);
  const bool use_sched = sched != nullptr && c.cs == 32;        // the schedule was built for 32 workgroups per cluster
  const int trips = use_sched ? rounds : (units - c.slot + c.cs - 1) / c.cs;
  for (int it = 0; it < trips; ++it) {
    const int u = use_sched ? sched[it * 32 + c.slot] : c.slot + it * c.cs;
    if (u < 0 || u >= units) continue;
    const int p = u / nblk, nb = u - p * nblk, n0 = nb * 32;
    const int4 hd = sload16(g.hdr + p);
    const int4* __restrict__ dsc = g.desc + hd.x;
    const int total = hd.y, n1 = gn ? hd.z : 0;
    const int brow = min(n0 + fi, g.Npad - 1);
    // epilogue operands do not depend on the product: ask for them first
    const int n = n0 + c4, bb = c.b0 + row;
    const bool ok = c.tid < 256 && row < TROWS && n < cold.nstore && bb < c.B;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gs = bv, gb = bv, rv = bv;
    if (ok) {
      if (cold.bias) bv = ldw4(cold.bias + (long long)p * cold.bias_pix_stride + n);
      if (gn) {
        gs = ldw4(cold.gn_s + (long long)p * g.Npad + n);
        gb = ldw4(cold.gn_b1 + (long long)p * g.Npad + n);
      }
      if (cold.R) rv = ld4(cold.R + ((long long)hd.w * c.B + bb) * cold.ldr + n);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // this wave's chunks of [lo, hi): lo + wave, lo + wave + NW, ...; PF of them in flight (4 waves per workgroup have 512 registers
    // each: four chunks deep; 8 waves: two)
    constexpr int PF = 2;
    auto run = [&](int lo, int hi) {
      const int first = lo + c.wave;
      if (first >= hi) return;
      const int nk = (hi - first + NW - 1) / NW;
      FragM f[PF];
#pragma unroll
      for (int sI = 0; sI < PF; ++sI) if (sI < nk) load_fragm(f[sI], g, sload16(dsc + first + sI * NW), c.b0, c.B, c.lane, brow, fh);
      for (int k = 0; k < nk; k += PF) {
#pragma unroll
        for (int sI = 0; sI < PF; ++sI) {
          if (k + sI < nk) {
            float av[16];
            { const int rem = nk - 1 - (k + sI); wait_fragm(f[sI], rem < PF - 1 ? rem : PF - 1); }
            stage_fragm(f[sI], atile, c.lane, av);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], f[sI].w[j >> 2][j & 3], acc, 0, 0, 0);
            if (k + sI + PF < nk) load_fragm(f[sI], g, sload16(dsc + first + (k + sI + PF) * NW), c.b0, c.B, c.lane, brow, fh);
          }
        }
      }
    };
    TR(2);
    if (gn) {
      run(0, n1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= rr[r];
      run(n1, total);
    } else {
      run(0, total);
    }
    TR(3);
    __syncthreads();                                           // every wave is done with its A tile: the area becomes the reduction buffer
    float* red = c.smem;                                       // [NW waves][32 rows][33]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(c.wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 33 + fi] = acc[r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = red[row * 33 + c4 + e];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[(w * 32 + row) * 33 + c4 + e];
      v[e] = t;
    }
    float s = 0.f, q = 0.f;
    if (ok) {
      v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      if (gn) {
        const float2 mrv = c.mr[row]; const float mrs = mrv.x * mrv.y;
        v[0] += gb.x - mrs * gs.x; v[1] += gb.y - mrs * gs.y; v[2] += gb.z - mrs * gs.z; v[3] += gb.w - mrs * gs.w;
      }
      v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = chain_act(cold.act, v[e]);
      st4(g.out + ((long long)p * c.B + bb) * cold.ldo + n, make_float4(v[0], v[1], v[2], v[3]));
      s = (v[0] + v[1]) + (v[2] + v[3]); q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if (cold.stat_out && c.tid < 256) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      if ((c.tid & 7) == 0 && row < TROWS && bb < c.B) cold.stat_out[((long long)p * nblk + nb) * c.B + bb] = make_float2(s, q);
    }
    __syncthreads();                                           // red is reused by the next unit
    TR(4);
  }
}

// depthwise 7x7 + bias + time condition + GroupNorm partial per (pixel, sample); see dwconv7_kernel
static DEVINL void chain_dw(const DwArgs& dsrc, Ctx& c) {
  const DwArgs d = uniform_copy(dsrc);
  const int P = d.H * d.W, npairs = P * TROWS;
  const int lanes = d.C / 4 < NT ? d.C / 4 : NT, spb = NT / lanes;   // threads per (pixel, sample) pair, pairs per pass
  const int sl = c.tid / lanes, cl = c.tid - sl * lanes;
  float2* red = reinterpret_cast<float2*>(c.smem);
  for (int base = 0; base < npairs; base += c.cs * spb) {
    const int qi = base + c.slot * spb + sl;
    const int p = qi / TROWS, b = c.b0 + (qi % TROWS);
    const bool active = sl < spb && qi < npairs && b < c.B;
    float s = 0.f, q = 0.f;
    if (active) {
      const int oy = p / d.W, ox = p - oy * d.W;
      const int ky0 = max(0, 3 - oy), ky1 = min(7, d.H + 3 - oy), kx0 = max(0, 3 - ox), kx1 = min(7, d.W + 3 - ox);
      for (int c4 = cl; c4 * 4 < d.C; c4 += lanes) {
        const int ch = c4 * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.H <= 4 && d.W <= 4) {
          // the window covers the whole image: walk the source pixels in raster order (= the tap order), four loads in flight
          for (int q0 = 0; q0 < P; q0 += 4) {
            float4 x[4], w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int qp = min(q0 + j, P - 1), qy = qp / d.W, qx = qp - qy * d.W;
              x[j] = ld4(d.in + ((long long)qp * d.B + b) * d.ldi + ch);
              w[j] = *reinterpret_cast<const float4*>(d.w + ((qy - oy + 3) * 7 + (qx - ox + 3)) * d.C + ch);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (q0 + j < P) { acc.x = fmaf(x[j].x, w[j].x, acc.x); acc.y = fmaf(x[j].y, w[j].y, acc.y); acc.z = fmaf(x[j].z, w[j].z, acc.z); acc.w = fmaf(x[j].w, w[j].w, acc.w); }
          }
        } else {
          for (int ky = ky0; ky < ky1; ++ky)
            for (int kx = kx0; kx < kx1; ++kx) {
              const int qp = (oy - 3 + ky) * d.W + (ox - 3 + kx);
              const float4 x = ld4(d.in + ((long long)qp * d.B + b) * d.ldi + ch);
              const float4 w = *reinterpret_cast<const float4*>(d.w + (ky * 7 + kx) * d.C + ch);
              acc.x = fmaf(x.x, w.x, acc.x); acc.y = fmaf(x.y, w.y, acc.y); acc.z = fmaf(x.z, w.z, acc.z); acc.w = fmaf(x.w, w.w, acc.w);
            }
        }
        const float4 bi = *reinterpret_cast<const float4*>(d.bias + ch);
        acc.x += bi.x; acc.y += bi.y; acc.z += bi.z; acc.w += bi.w;
        if (d.cond) {
          const float4 cv = *reinterpret_cast<const float4*>(dw_cond_row(d, b) + ch);
          acc.x += cv.x; acc.y += cv.y; acc.z += cv.z; acc.w += cv.w;
        }
        *reinterpret_cast<float4*>(d.out + ((long long)p * d.B + b) * d.ldo + ch) = acc;
        s += acc.x + acc.y + acc.z + acc.w;
        q += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
      }
    }
    if (d.stat) {
      red[c.tid] = make_float2(s, q);
      __syncthreads();
      if (active && cl == 0) {
        float ss = 0.f, qq = 0.f;
        for (int i = 0; i < lanes; ++i) { ss += red[c.tid + i].x; qq += red[c.tid + i].y; }
        d.stat[(long long)p * d.B + b] = make_float2(ss, qq);
      }
      __syncthreads();
    }
  }
}

// LinearAttention core, one wave per (sample, head); see linattn_kernel
// n <= 16: every global load of a task (k for all pixels, this half's q, the v tile) is issued before the first use; the
// arithmetic and its order are those of linattn_kernel.
static DEVINL void chain_linattn16(const AttnArgs a, Ctx& c) {
  float* my = c.smem + c.wave * 1120;
  float (*ctx)[33] = reinterpret_cast<float (*)[33]>(my);      // overlays the v tile: v is dead once ctx is formed
  float* sv = my;                                              // [pixel][32]
  float (*qs)[32] = reinterpret_cast<float (*)[32]>(my + 32 * 33);
  const int d = c.lane & 31, half = c.lane >> 5;
  const int ntask = TROWS * a.heads, hd = a.heads * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int base = 0; base < ntask; base += c.cs * NW) {
    const int t = base + c.slot * NW + c.wave;
    const int bi = c.b0 + t / a.heads, h = t % a.heads;
    const bool active = t < ntask && bi < c.B;
    const int b = min(bi, c.B - 1);
    const float* qb = a.qkv + (long long)b * a.ldq + h * 32;
    float kk[16], qq[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) kk[i] = ld1(qb + (long long)min(i, a.n - 1) * pstride + hd + d);
#pragma unroll
    for (int j = 0; j < 8; ++j) qq[j] = ld1(qb + (long long)min(half + 2 * j, a.n - 1) * pstride + d);
    float2 vv[4];                                              // v tile: n x 32 floats, 16 lanes per pixel row
#pragma unroll
    for (int j = 0; j < 4; ++j) vv[j] = ld2(qb + (long long)min((c.lane >> 4) + 4 * j, a.n - 1) * pstride + 2 * hd + (c.lane & 15) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float2*>(sv + ((c.lane >> 4) + 4 * j) * 32 + (c.lane & 15) * 2) = vv[j];
    float kmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) if ((i & 1) == half && i < a.n) kmax = fmaxf(kmax, kk[i]);
    kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
    float ksum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) if ((i & 1) == half && i < a.n) ksum += expf(kk[i] - kmax);
    ksum += __shfl_xor(ksum, 32);
    const float kinv = 1.0f / ksum;
    float cc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cc[e] = 0.f;
    __syncthreads();                                           // v tile written
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < a.n) {
        const float kv = expf(kk[i] - kmax) * kinv;
        const float* vrow = sv + i * 32 + half * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) cc[e] = fmaf(kv, vrow[e], cc[e]);
      }
    }
    __syncthreads();                                           // everyone is done reading v: ctx may overlay it
#pragma unroll
    for (int e = 0; e < 16; ++e) ctx[d][half * 16 + e] = cc[e];
    __syncthreads();
    float col[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) col[dd] = ctx[dd][d];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 2 * j + half;
      if (2 * j < a.n) {                                       // uniform
        const bool ok = i < a.n;
        const float qv = ok ? qq[j] : 0.f;
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
        if (ok && active) a.out[((long long)i * a.B + b) * a.ldo + h * 32 + d] = o;
        __syncthreads();
      }
    }
  }
}

static DEVINL void chain_linattn(const AttnArgs& asrc, Ctx& c) {
  const AttnArgs a = uniform_copy(asrc);
  if (a.n <= 16) { chain_linattn16(a, c); return; }
  float* my = c.smem + c.wave * 1120;
  float (*ctx)[33] = reinterpret_cast<float (*)[33]>(my);
  float (*qs)[32] = reinterpret_cast<float (*)[32]>(my + 32 * 33);
  const int d = c.lane & 31, half = c.lane >> 5;
  const int ntask = TROWS * a.heads, hd = a.heads * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int base = 0; base < ntask; base += c.cs * NW) {
    const int t = base + c.slot * NW + c.wave;
    const int bi = c.b0 + t / a.heads, h = t % a.heads;
    const bool active = t < ntask && bi < c.B;
    const int b = min(bi, c.B - 1);
    const float* qb = a.qkv + (long long)b * a.ldq + h * 32;
    float kmax = -INFINITY;
    for (int i = half; i < a.n; i += 2) kmax = fmaxf(kmax, ld1(qb + i * pstride + hd + d));
    kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
    float ksum = 0.f;
    for (int i = half; i < a.n; i += 2) ksum += expf(ld1(qb + i * pstride + hd + d) - kmax);
    ksum += __shfl_xor(ksum, 32);
    const float kinv = 1.0f / ksum;
    float cc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cc[e] = 0.f;
    for (int i = 0; i < a.n; ++i) {
      const float kv = expf(ld1(qb + i * pstride + hd + d) - kmax) * kinv;
      const float* vrow = qb + i * pstride + 2 * hd + half * 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float2 vv = ld2(vrow + 2 * e); cc[2 * e] = fmaf(kv, vv.x, cc[2 * e]); cc[2 * e + 1] = fmaf(kv, vv.y, cc[2 * e + 1]); }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) ctx[d][half * 16 + e] = cc[e];
    __syncthreads();
    float col[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) col[dd] = ctx[dd][d];
    for (int i0 = 0; i0 < a.n; i0 += 2) {
      const int i = i0 + half;
      const bool ok = i < a.n;
      const float qv = ok ? ld1(qb + i * pstride + d) : 0.f;
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
      if (ok && active) a.out[((long long)i * a.B + b) * a.ldo + h * 32 + d] = o;
      __syncthreads();
    }
  }
}

// softmax attention core, n <= 16 inside the chain; one wave per (sample, head); see attn_kernel
static DEVINL void chain_attn(const AttnArgs& asrc, Ctx& c) {
  const AttnArgs a = uniform_copy(asrc);
  float* my = c.smem + c.wave * 1088;
  float (*sk)[17] = reinterpret_cast<float (*)[17]>(my);
  float (*sv)[17] = reinterpret_cast<float (*)[17]>(my + 32 * 17);
  const int ntask = TROWS * a.heads, hd = a.heads * 32;
  const long long pstride = (long long)a.B * a.ldq;
  for (int base = 0; base < ntask; base += c.cs * NW) {
    const int t = base + c.slot * NW + c.wave;
    const int bi = c.b0 + t / a.heads, h = t % a.heads;
    const bool active = t < ntask && bi < c.B;
    const int b = min(bi, c.B - 1);
    const float* qb = a.qkv + (long long)b * a.ldq + h * 32;
    for (int idx = c.lane; idx < 32 * a.n; idx += 64) {
      const int j = idx >> 5, dd = idx & 31;
      sk[dd][j] = ld1(qb + j * pstride + hd + dd); sv[dd][j] = ld1(qb + j * pstride + 2 * hd + dd);
    }
    __syncthreads();
    if (c.lane < a.n) {
      float q[32];
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) q[dd] = ld1(qb + c.lane * pstride + dd) * a.scale;
      float m = -INFINITY;
      for (int j = 0; j < a.n; ++j) { float tt = 0.f; for (int dd = 0; dd < 32; ++dd) tt = fmaf(q[dd], sk[dd][j], tt); m = fmaxf(m, tt); }
      float s = 0.f, o[32];
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) o[dd] = 0.f;
      for (int j = 0; j < a.n; ++j) {
        float tt = 0.f; for (int dd = 0; dd < 32; ++dd) tt = fmaf(q[dd], sk[dd][j], tt);
        const float e = expf(tt - m); s += e;
#pragma unroll
        for (int dd = 0; dd < 32; ++dd) o[dd] = fmaf(e, sv[dd][j], o[dd]);
      }
      const float inv = 1.0f / s;
      if (active)
#pragma unroll
        for (int dd = 0; dd < 32; ++dd) a.out[((long long)c.lane * a.B + b) * a.ldo + h * 32 + dd] = o[dd] * inv;
    }
    __syncthreads();
  }
}

// y = GroupNorm(1,C)(u) * gamma + beta + x, one workgroup per sample; see gn_res_kernel
static DEVINL void chain_gn_res(const GnResArgs& gsrc, Ctx& c) {
  const GnResArgs g = uniform_copy(gsrc);
  double* rs = reinterpret_cast<double*>(c.smem);              // [NT] sums, [NT] squares
  double* rq = rs + NT;
  for (int r = c.slot; r < TROWS; r += c.cs) {
    const int b = c.b0 + r;
    if (b >= c.B) break;
    double s = 0.0, q = 0.0;
    for (int i = c.tid; i < g.nparts; i += NT) { const float2 v = ld2(reinterpret_cast<const float*>(g.part + (long long)i * g.B + b)); s += v.x; q += v.y; }
    rs[c.tid] = s; rq[c.tid] = q;
    __syncthreads();
    for (int o = NT / 2; o >= 1; o >>= 1) { if (c.tid < o) { rs[c.tid] += rs[c.tid + o]; rq[c.tid] += rq[c.tid + o]; } __syncthreads(); }
    const double mean_d = rs[0] * (double)g.inv_count;
    double var = rq[0] * (double)g.inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + 1e-5));
    const int c4n = g.C / 4;
    for (int idx = c.tid; idx < g.P * c4n; idx += NT) {
      const int p = idx / c4n, ch = (idx - p * c4n) * 4;
      const long long row = (long long)p * g.B + b;
      const float4 u = ld4(g.u + row * g.ldu + ch);
      const float4 x = ld4(g.x + row * g.ldx + ch);
      const float4 ga = *reinterpret_cast<const float4*>(g.gamma + ch);
      const float4 be = *reinterpret_cast<const float4*>(g.beta + ch);
      float4 y;
      y.x = ((u.x - mean) * rstd) * ga.x + be.x + x.x; y.y = ((u.y - mean) * rstd) * ga.y + be.y + x.y;
      y.z = ((u.z - mean) * rstd) * ga.z + be.z + x.z; y.w = ((u.w - mean) * rstd) * ga.w + be.w + x.w;
      *reinterpret_cast<float4*>(g.out + row * g.ldo + ch) = y;
    }
    __syncthreads();
  }
}

};  // struct ChainImpl

}  // namespace

template <int NT, int GROUPS>
__global__ __launch_bounds__(NT, 2) void chain_kernel(const ChainOp* __restrict__ ops, int count, ChainSync* sync, int B, unsigned long long* dbg, const ChainCtl ctl) {
  using K = ChainImpl<NT, GROUPS>;
  using Ctx = typename K::Ctx;
  __shared__ __attribute__((aligned(16))) float smem[K::SMEM_F];
  __shared__ __attribute__((aligned(16))) ChainOp sops[CHAIN_MAX_OPS];
  __shared__ float2 mr[32];
  __shared__ unsigned info[8];
  Ctx c;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.smem = smem; c.mr = mr; c.flag = &info[7]; c.sync = sync; c.B = B; c.gen = 0; c.tr = nullptr;
  c.spin_limit = ctl.spin_limit ? ctl.spin_limit : SPIN_LIMIT; c.abort_host = ctl.abort_host;
  {                                                            // stage the op descriptors
    const int nw = count * (int)(sizeof(ChainOp) / 4);
    const int* src = reinterpret_cast<const int*>(ops);
    int* dst = reinterpret_cast<int*>(sops);
    for (int i = c.tid; i < nw; i += NT) dst[i] = src[i];
  }
  const unsigned xcc = xcc_id();
  // ---- registration: which XCD am I on, who else is here.  A launch that finds the abort flag of an EARLIER launch still
  //      set (the host has not reset the state yet) does nothing at all: it must not touch counters that may be inconsistent.
  if (c.tid == 0) {
    unsigned ok = ctr_load(&sync->abort_flag) ? 0u : 1u, slot = 0;
    info[6] = ok;                                              // 0: stale abort, leave without registering
    if (ok) {
      slot = ctr_add(&sync->count[xcc], 1u);
      ctr_add(&sync->arrived, 1u + (slot >> 31));              // issued only after `slot` has come back
      unsigned spins = 0;
      while (ctr_load(&sync->arrived) < gridDim.x) {
        if (++spins > c.spin_limit) { K::raise_abort(c); ok = 0; break; }
        if ((spins & 255u) == 0 && ctr_load(&sync->abort_flag)) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // clusters: the workgroups of an XCD are dealt round-robin (by arrival) to GROUPS groups; cluster = (XCD, group)
    const unsigned grp = slot % GROUPS;
    unsigned nact = 0, ci = 0, mine = 0;
    for (unsigned i = 0; i < 8; ++i) {
      const unsigned n = ctr_load(&sync->count[i]);
      for (unsigned g = 0; g < GROUPS; ++g) {
        const unsigned sz = (n + GROUPS - 1 - g) / GROUPS;
        if (i == xcc && g == grp) mine = sz;
        if (sz) { if (i < xcc || (i == xcc && g < grp)) ++ci; ++nact; }
      }
    }
    info[0] = slot / GROUPS; info[1] = mine; info[2] = ci; info[3] = nact; info[4] = ok; info[5] = grp;
  }
  __syncthreads();
  if (!info[6]) return;
  c.slot = (int)info[0]; c.cs = (int)info[1];
  const int ci = (int)info[2], nact = (int)info[3];
  bool alive = info[4] != 0;
  const unsigned cl = xcc * GROUPS + info[5];                  // cluster id
  c.bar = &sync->bar[cl * 16];
  c.skip_arrive = (ctl.debug_skip & 1u) && ci == 0 && c.slot == 0;     // test knob: this workgroup never announces itself at a barrier
  c.no_prefetch = (ctl.debug_skip & 2u) != 0;                          // experiment knob: no weight touches during the barriers
  const int tiles = (B + K::TROWS - 1) / K::TROWS;
  const bool stamp = dbg && ci == 0 && c.slot == 0 && c.tid == 0;
  if (stamp) dbg[0] = __builtin_amdgcn_s_memtime();
  for (int tile = ci; alive && tile < tiles; tile += nact) {
    c.b0 = tile * K::TROWS;
    for (int k = 0; alive && k < count; ++k) {
      c.tr = stamp && tile == ci ? dbg + 1 + 2 * count + 8 * k : nullptr;
      const ChainOp& op = sops[k];
      const int kind = __builtin_amdgcn_readfirstlane(op.kind);
      switch (kind) {
        case OP_GEMM: K::chain_gemm(op.gm, __builtin_amdgcn_readfirstlane(op.P), c); break;
        case OP_DWCONV7: K::chain_dw(op.d, c); break;
        case OP_LINATTN: K::chain_linattn(op.a, c); break;
        case OP_ATTN: K::chain_attn(op.a, c); break;
        case OP_GN_RES: K::chain_gn_res(op.r, c); break;
        default: break;
      }
      if (stamp && tile == ci) dbg[1 + 2 * k] = __builtin_amdgcn_s_memtime();        // this workgroup finished op k
      alive = K::cluster_barrier(c, k + 1 < count ? &sops[k + 1] : nullptr);
      if (stamp && tile == ci) dbg[2 + 2 * k] = __builtin_amdgcn_s_memtime();        // the whole cluster has
    }
  }
  // ---- leave: the last workgroup of the cluster clears its barrier counter, the last of the grid the registration.
  //      After an abort nothing is cleared: the host resets the whole block before the next launch (api.hip). ----
  if (c.tid == 0 && !ctr_load(&sync->abort_flag)) {
    if (ctr_add(&sync->left[cl * 16], 1u) == (unsigned)c.cs - 1u) { ctr_zero(&sync->bar[cl * 16]); ctr_zero(&sync->left[cl * 16]); }
    if (ctr_add(&sync->done, 1u) == gridDim.x - 1u) {
      for (int i = 0; i < 8; ++i) ctr_zero(&sync->count[i]);
      ctr_zero(&sync->arrived); ctr_zero(&sync->done);
    }
  }
}

unsigned long long* g_chain_dbg = nullptr;   // diagnostics only (sbm_debug_chain_stamps)

int chain_variant_groups() { return 1; }   // two interleaved 16-sample clusters per XCD (GROUPS = 2, 256-thread workgroups) measured SLOWER: DESIGN section 8

// The chain kernel's barriers need every workgroup of the grid resident at once.  Checked against the occupancy calculator at
// model creation (api.hip): grid = workgroups per CU (1 x 512 threads, or 2 x 256) x CUs, never more than fits.
const char* chain_launch_geometry(int cus, int* grid, int* block) {
  const int groups = chain_variant_groups();
  int per_cu = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_kernel<512, 1>, 512, 0);
  if (e != hipSuccess) return "chain kernel: occupancy query failed";
  if (per_cu < groups) return "chain kernel: the device cannot keep the whole grid resident (occupancy below the cluster design)";
  *grid = groups * cus; *block = 512;
  return nullptr;
}

const char* launch_chain(const DevPlan& dp, int first, int count, hipStream_t st, float* ext_out, const StepArgs* em) {
  const Plan& plan = *dp.plan;
  if (dp.use_pix) return launch_pixrun(dp, first, count, st, ext_out, em);
  if (em) return "chain kernel: the sampler's update cannot ride on this launch";     // pixel mode: the marked ops are sample-owned workgroups (pixrun.hip); no chain tables exist
  if (!dp.chain_ops || !dp.chain_sync) return "chain kernel: no op table";
  for (int i = first; i < first + count; ++i) {
    const Op& op = plan.ops[i];
    if (!op.chain) return "chain kernel: op is not marked for the chain";
    if (op.kind == OP_GEMM && (op.g.a_ext || op.g.out_nchw)) return "chain kernel: external operands are not supported";
    if ((op.kind == OP_LINATTN || op.kind == OP_ATTN) && op.at.dh != 32) return "attention: dim_head must be 32";
    if (op.kind == OP_ATTN && op.at.qkv.H * op.at.qkv.W > 16) return "chain kernel: softmax attention over more than 16 positions";
  }
  static const bool no_pf = std::getenv("SBM_CHAIN_NO_PREFETCH") != nullptr;
  ChainCtl ctl = dp.chain_ctl;
  if (no_pf) ctl.debug_skip |= 2u;
  for (int done = 0; done < count; done += CHAIN_MAX_OPS) {
    const ChainOp* ops = dp.chain_ops + first + done;
    const int n = count - done < CHAIN_MAX_OPS ? count - done : CHAIN_MAX_OPS;
    if (g_prof_start) hipExtLaunchKernelGGL((chain_kernel<512, 1>), dim3(dp.chain_grid), dim3(512), 0, st, g_prof_start, g_prof_stop, 0, ops, n, dp.chain_sync, plan.B, g_chain_dbg, ctl);
    else hipLaunchKernelGGL((chain_kernel<512, 1>), dim3(dp.chain_grid), dim3(512), 0, st, ops, n, dp.chain_sync, plan.B, g_chain_dbg, ctl);
  }
  return nullptr;
}

}  // namespace sbm
