// Probe for the round-3 rewrite of the sample-owned low-resolution kernel (pixrun.hip).  Three parts, one binary:
//   1. MFMA issue rates with real (distinct) operands: v_mfma_f32_16x16x4_f32 / 32x32x2 / 4x4x1 for 1, 2, 4 accumulators per wave
//      and 1, 2, 4 waves per SIMD -- is one dependent chain per wave (many waves) faster than several accumulators per wave?
//   2. L2 weight-stream rates when EVERY workgroup walks the same bytes: contiguous 1 KB per wave instruction vs the [n][k]
//      pattern (16 rows x 64 B, two instructions per 128-byte line), 8 / 16 waves, 4 / 8 loads in flight per lane.
//   3. The planned GEMM core: a chain of 3x3 convolutions at 4x4 (16 pixel rows on the MFMA M axis, activations in LDS, weights
//      [tap][n][k] streamed to registers, one 16x16 accumulator per wave, K split over wave groups), with and without the ring
//      prefetch of the NEXT op issued before the epilogue; checked against a host reference.
// build: hipcc --offload-arch=gfx950 -O3 -o pixgemm_probe pixgemm_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;
typedef __attribute__((address_space(1))) float gf32;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ part 1: MFMA issue
template <int NACC>
__global__ __launch_bounds__(1024) void mf16(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0, 0, 0, 0};
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = in[threadIdx.x + 64 * j]; b[j] = in[threadIdx.x + 64 * j + 1024]; }
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(j + k) & 7], acc[k], 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(1024) void mf32(const float* in, float* out, int iters) {
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = in[threadIdx.x + 64 * j]; b[j] = in[threadIdx.x + 64 * j + 1024]; }
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + k) & 7], acc[k], 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(1024) void mf4(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0, 0, 0, 0};
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = in[threadIdx.x + 64 * j]; b[j] = in[threadIdx.x + 64 * j + 1024]; }
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[(j + k) & 7], acc[k], 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------ part 2: weight stream
// MODE 0: a wave instruction reads 1 KB contiguous; the waves of a workgroup take consecutive KBs.
// MODE 1: the [n][k] pattern of the GEMM core: wave w owns 16 rows of `rowf` floats; a step reads 32 floats of each row
//         (two dwordx4 per lane: lane (n = l & 15, g = l >> 4) at row n, k0 + 4 g and k0 + 16 + 4 g).
#define GLOADP(r0, r1, p, OFF) asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:" #OFF : "=&v"(r0), "=&v"(r1) : "v"(p) : "memory")
#define VMWAITP(N, r0, r1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N) : "memory")
template <int MODE, int D>
__global__ __launch_bounds__(1024) void wstream(const float* __restrict__ w, long long floats_per_wg, int rowf, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x4 ring[D][2];
  f32x4 sum = {0, 0, 0, 0};
  if (MODE == 0) {
    const long long steps = floats_per_wg / (512 * nw) / D * D;      // a step = 2 KB per wave
    const float* p = w + (long long)wave * 512 + lane * 4;
    const long long stride = 512LL * nw;
#pragma unroll
    for (int r = 0; r < D; ++r) GLOADP(ring[r][0], ring[r][1], p + r * stride, 1024);
    for (long long i = 0; i < steps; i += D) {
#pragma unroll
      for (int r = 0; r < D; ++r) {
        VMWAITP(2 * D - 2, ring[r][0], ring[r][1]);
        sum += ring[r][0] + ring[r][1];
        const long long j = i + r + D < steps ? i + r + D : r;        // past the end: re-read the head (keeps the count exact)
        GLOADP(ring[r][0], ring[r][1], p + j * stride, 1024);
      }
    }
  } else {
    // the arena is a sequence of blocks of nw * 16 rows x rowf floats; wave w walks rows 16 w .. 16 w + 15 of every block
    const int n = lane & 15, g = lane >> 4;
    const long long blockf = (long long)nw * 16 * rowf;
    const long long nblocks = floats_per_wg / blockf;
    const int spb = rowf / 32;                                // steps per block
    const long long steps = nblocks * spb / D * D;
    auto addr = [&](long long s) { const long long bl = s / spb; const int st = (int)(s - bl * spb); return w + bl * blockf + (long long)(wave * 16 + n) * rowf + st * 32 + 4 * g; };
#pragma unroll
    for (int r = 0; r < D; ++r) GLOADP(ring[r][0], ring[r][1], addr(r), 64);
    for (long long i = 0; i < steps; i += D) {
#pragma unroll
      for (int r = 0; r < D; ++r) {
        VMWAITP(2 * D - 2, ring[r][0], ring[r][1]);
        sum += ring[r][0] + ring[r][1];
        const long long j = i + r + D < steps ? i + r + D : r;
        GLOADP(ring[r][0], ring[r][1], addr(j), 64);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < D; ++r) VMWAITP(0, ring[r][0], ring[r][1]);
  out[blockIdx.x * 1024 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

// ------------------------------------------------------------------------------------------------ part 3: GEMM core
// Activations: LDS slabs T[chunk][17 rows][16 k] (row 16 stays zero: the source of padding taps), chunk stride CS = 272 floats.
// Work list per wave: items = (op, pass of <= NW column tiles); an item's K steps (one step = one tap x 32 k = two 16-k chunks) are a
// host-built table of {weight float offset, A chunk offset | row-table index}, dealt to ksplit wave groups and padded to a multiple
// of D with no-op steps (all rows -> the zero row), so that the loader runs exactly D steps ahead of the consumer ACROSS item
// boundaries: the next op's first weights are in flight under this op's reduction and epilogue.
// The weight ring is loaded with inline-asm loads that hipcc does not count; every consumer waits with a hand-counted vmcnt that
// names the slot's registers (cdna_hip_programming.md 5.7 item 1, form ii).
constexpr int CS = 272;
// Weights are host-packed in MFMA FRAGMENT order, so that every wave-load is 1 KB contiguous (the probe's part 2: 118 GB/s per
// workgroup against 37 GB/s for 64-byte row segments): block (column PAIR tp, step f) = 4 KB =
//   [half h = 0, 1 (16 k each)][tile t = 0, 1 of the pair][lane][4 floats] = W[n = 32 tp + 16 t + (lane & 15)][k = 32 dc + 16 h + 4 (lane >> 4) + i]
// and a wave owns a column PAIR: two independent accumulators (one chain per wave does not reach the MFMA rate, part 1).
struct PStep { unsigned w_off; unsigned a_sel; };      // w_off: float offset of the step's block for pair 0; a_sel = LDS float offset of the step's first chunk | row-table index << 16
struct PItem {
  int step_base, nkp, tshift, tile0, pair_stride, kshift, out_off, pad0;   // tshift: log2(column pairs of the pass); pair_stride: floats between the blocks of consecutive pairs
  const float* bias; float oscale; int pad;
};
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678f)); }

// The weight ring lives in AGPRs with hard-coded indices: slot r = a[16 r .. 16 r + 15] = {half 0 tile 0, half 0 tile 1, half 1 tile 0,
// half 1 tile 1} x 4 floats.  hipcc never allocates AGPRs in this kernel (no spills at this register count: audited in the ISA), so no
// compiler copy can read a slot before its load has landed -- which is what went wrong with asm loads into compiler-allocated VGPRs
// (a v_mov of a ring register in a peeled loop iteration, ahead of the hand-placed s_waitcnt).  Loads and MFMAs are inline asm; the
// MFMA takes its B operand straight from the AGPR.  vmcnt(N): N = 4 (D - 1) loads are always issued behind a slot's own four before it
// is consumed again (the loader never skips a step), and anything else in flight only makes the wait stricter.
#define ACL16(b) "a" #b
#define RING_CLOBBER_0 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"
#define RING_CLOBBER_1 "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"
#define RING_CLOBBER_2 "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"
#define RING_CLOBBER_3 "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
#define RING_LOAD(voff, base, A0, A3, A4, A7, A8, A11, A12, A15, CLOB) \
  asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, %1\n\t" \
               "global_load_dwordx4 a[" #A4 ":" #A7 "], %0, %1 offset:1024\n\t" \
               "global_load_dwordx4 a[" #A8 ":" #A11 "], %0, %1 offset:2048\n\t" \
               "global_load_dwordx4 a[" #A12 ":" #A15 "], %0, %1 offset:3072" \
               :: "v"(voff), "s"(base) : CLOB)
#define MFA(ACC, A, B) "v_mfma_f32_16x16x4_f32 " ACC ", " A ", a" #B ", " ACC "\n\t"
// 16 MFMAs of one step: acc0 += A0[i] x (half 0, tile 0)[i], acc1 += A0[i] x (half 0, tile 1)[i], then half 1 with A1
#define RING_MFMA(acc0, acc1, a0, a1, WAITN, B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, B10, B11, B12, B13, B14, B15) \
  asm volatile("s_waitcnt vmcnt(" #WAITN ")\n\t" \
               MFA("%0", "%2", B0) MFA("%1", "%2", B4) MFA("%0", "%3", B1) MFA("%1", "%3", B5) MFA("%0", "%4", B2) MFA("%1", "%4", B6) MFA("%0", "%5", B3) MFA("%1", "%5", B7) \
               MFA("%0", "%6", B8) MFA("%1", "%6", B12) MFA("%0", "%7", B9) MFA("%1", "%7", B13) MFA("%0", "%8", B10) MFA("%1", "%8", B14) MFA("%0", "%9", B11) "v_mfma_f32_16x16x4_f32 %1, %9, a" #B15 ", %1" \
               : "+v"(acc0), "+v"(acc1) : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]))

constexpr int D = 4;
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_chain(const PItem* __restrict__ items, int nitems, const PStep* __restrict__ steps, const unsigned char* __restrict__ rowtab, int nrowtab,
                                                      const float* __restrict__ arena, const float* __restrict__ x0, int C0, int in0_off, float* __restrict__ out, int Cl, int outl_off,
                                                      int act_floats, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + act_floats;           // [NW][16 rows][32 cols]
  unsigned char* rtab = reinterpret_cast<unsigned char*>(red + NW * 512);   // [nrowtab][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  for (int i = tid; i < act_floats; i += NW * 64) lds[i] = 0.f;
  for (int i = tid; i < nrowtab * 16; i += NW * 64) rtab[i] = rowtab[i];
  __syncthreads();
  {
    const float sc = 1.0f + (float)(b & 7) * 0.125f;
    for (int i = tid; i < 16 * C0; i += NW * 64) { const int r = i / C0, k = i - r * C0; lds[in0_off + (k >> 4) * CS + r * 16 + (k & 15)] = x0[i] * sc; }
  }
  __syncthreads();
  unsigned asel[D];
  auto lane_off = [&](const PItem& it) { return (unsigned)(((wave & ((1 << it.tshift) - 1)) * it.pair_stride + lane * 4) * 4); };   // bytes
  auto slice_steps = [&](const PItem& it) { return steps + it.step_base + (wave >> it.tshift) * it.nkp; };
#define LOAD_SLOT(r, e, off) do { const float* base_ = arena + (e).w_off; \
    if (r == 0) RING_LOAD(off, base_, 0, 3, 4, 7, 8, 11, 12, 15, RING_CLOBBER_0); \
    else if (r == 1) RING_LOAD(off, base_, 16, 19, 20, 23, 24, 27, 28, 31, RING_CLOBBER_1); \
    else if (r == 2) RING_LOAD(off, base_, 32, 35, 36, 39, 40, 43, 44, 47, RING_CLOBBER_2); \
    else RING_LOAD(off, base_, 48, 51, 52, 55, 56, 59, 60, 63, RING_CLOBBER_3); } while (0)
#define MFMA_SLOT(r, a0, a1) do { \
    if (r == 0) RING_MFMA(acc0, acc1, a0, a1, 12, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15); \
    else if (r == 1) RING_MFMA(acc0, acc1, a0, a1, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31); \
    else if (r == 2) RING_MFMA(acc0, acc1, a0, a1, 12, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47); \
    else RING_MFMA(acc0, acc1, a0, a1, 12, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63); } while (0)
  PItem cur = items[0];
  unsigned off_c = lane_off(cur);
  const PStep* st_c = slice_steps(cur);
#pragma unroll
  for (int r = 0; r < D; ++r) {
    const PStep e = st_c[r];
    asel[r] = e.a_sel;
    LOAD_SLOT(r, e, off_c);
  }
  // A operands of step 0 and the row selector of step 1
  int rs = rtab[(asel[0] >> 16) * 16 + n];
  const float* ap0 = lds + (asel[0] & 0xffff) + rs * 16 + 4 * g;
  f32x4 a0 = *reinterpret_cast<const f32x4*>(ap0), a1 = *reinterpret_cast<const f32x4*>(ap0 + CS);
  rs = rtab[(asel[1] >> 16) * 16 + n];
  for (int ii = 0; ii < nitems; ++ii) {
    const PItem nxt = items[ii + 1 < nitems ? ii + 1 : 0];
    const unsigned off_n = lane_off(nxt);
    const PStep* st_n = slice_steps(nxt);
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 0] = __builtin_amdgcn_s_memtime();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const int nu = cur.nkp / D;
    for (int u = 0; u < nu; ++u) {
      const bool last = u == nu - 1;
      const PStep* lst = last ? st_n : st_c + (u + 1) * D;
      const unsigned loff = last ? off_n : off_c;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        // A operands of the NEXT step (its row selector was read one step ago), row selector of the step after it
        const float* apn = lds + (asel[(r + 1) % D] & 0xffff) + rs * 16 + 4 * g;
        const f32x4 n0 = *reinterpret_cast<const f32x4*>(apn), n1 = *reinterpret_cast<const f32x4*>(apn + CS);
        const PStep e = lst[r];
        MFMA_SLOT(r, a0, a1);
        asel[r] = e.a_sel;
        rs = rtab[(asel[(r + 2) % D] >> 16) * 16 + n];
        LOAD_SLOT(r, e, loff);
        a0 = n0; a1 = n1;
      }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));      // MFMA results -> VALU / LDS readers (hipcc pads nothing behind an asm MFMA)
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 1] = __builtin_amdgcn_s_memtime();
    // partial tiles -> LDS: red[wave][row][32 cols]
#pragma unroll
    for (int v = 0; v < 4; ++v) { red[(wave * 16 + 4 * g + v) * 32 + n] = acc0[v]; red[(wave * 16 + 4 * g + v) * 32 + 16 + n] = acc1[v]; }
    __syncthreads();
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 2] = __builtin_amdgcn_s_memtime();
    // epilogue: thread = (row, 4 columns)
    const int pairs = 1 << cur.tshift, ksplit = 1 << cur.kshift, quads = pairs * 8;
    for (int idx = tid; idx < 16 * quads; idx += NW * 64) {
      const int row = idx / quads, c4 = idx - row * quads, pr = c4 >> 3, sub = c4 & 7;
      f32x4 s = {0, 0, 0, 0};
      for (int sl = 0; sl < ksplit; ++sl) s += *reinterpret_cast<const f32x4*>(red + ((sl * pairs + pr) * 16 + row) * 32 + 4 * sub);
      const int col = (cur.tile0 + 2 * pr) * 16 + 4 * sub;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(cur.bias + col);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = gelu(s[e] + bi[e]) * cur.oscale;
      *reinterpret_cast<f32x4*>(lds + cur.out_off + (col >> 4) * CS + row * 16 + (col & 15)) = y;
    }
    __syncthreads();
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 3] = __builtin_amdgcn_s_memtime();
    // the A operands prefetched for the next item's first step were read BEFORE this epilogue wrote its output: read them again
    {
      int rs0 = rtab[(asel[0] >> 16) * 16 + n];
      const float* apn = lds + (asel[0] & 0xffff) + rs0 * 16 + 4 * g;
      a0 = *reinterpret_cast<const f32x4*>(apn); a1 = *reinterpret_cast<const f32x4*>(apn + CS);
    }
    cur = nxt; off_c = off_n; st_c = st_n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: RING_CLOBBER_0, RING_CLOBBER_1, RING_CLOBBER_2, RING_CLOBBER_3);
  for (int i = tid; i < 16 * Cl; i += NW * 64) { const int r = i / Cl, k = i - r * Cl; out[(long long)b * 16 * Cl + i] = lds[outl_off + (k >> 4) * CS + r * 16 + (k & 15)]; }
}

// ---- the same chain for at most FOUR output rows (the 2x2 and 1x1 levels): v_mfma_f32_4x4x1_16b_f32, 16 blocks = 16 column quads
// (64 columns per wave) sharing one k; a step = 16 k x 64 columns = 4 KB: load j holds W[n = 64 tile + lane][k = 16 dc + 4 j + i].
// Lane (b, i) = (lane >> 2, lane & 3) supplies A[row i][k]; lane l receives D[row v][column l] in register v (checked numerically here).
#define MF4(ACC, A, B) "v_mfma_f32_4x4x1_16b_f32 " ACC ", " A ", a" #B ", " ACC "\n\t"
#define RING_MFMA4(acc0, acc1, q0, q1, q2, q3, WAITN, B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, B10, B11, B12, B13, B14, B15) \
  asm volatile("s_waitcnt vmcnt(" #WAITN ")\n\t" \
               MF4("%0", "%2", B0) MF4("%1", "%3", B1) MF4("%0", "%4", B2) MF4("%1", "%5", B3) MF4("%0", "%6", B4) MF4("%1", "%7", B5) MF4("%0", "%8", B6) MF4("%1", "%9", B7) \
               MF4("%0", "%10", B8) MF4("%1", "%11", B9) MF4("%0", "%12", B10) MF4("%1", "%13", B11) MF4("%0", "%14", B12) MF4("%1", "%15", B13) MF4("%0", "%16", B14) "v_mfma_f32_4x4x1_16b_f32 %1, %17, a" #B15 ", %1" \
               : "+v"(acc0), "+v"(acc1) : "v"(q0[0]), "v"(q0[1]), "v"(q0[2]), "v"(q0[3]), "v"(q1[0]), "v"(q1[1]), "v"(q1[2]), "v"(q1[3]), \
                 "v"(q2[0]), "v"(q2[1]), "v"(q2[2]), "v"(q2[3]), "v"(q3[0]), "v"(q3[1]), "v"(q3[2]), "v"(q3[3]))
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_chain_small(const PItem* __restrict__ items, int nitems, const PStep* __restrict__ steps, const unsigned char* __restrict__ rowtab, int nrowtab,
                                                            const float* __restrict__ arena, const float* __restrict__ x0, int C0, int in0_off, float* __restrict__ out, int Cl, int outl_off,
                                                            int act_floats, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + act_floats;           // [NW][4 rows][64 cols]
  unsigned char* rtab = reinterpret_cast<unsigned char*>(red + NW * 512);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 3;
  const int b = blockIdx.x;
  for (int i = tid; i < act_floats; i += NW * 64) lds[i] = 0.f;
  for (int i = tid; i < nrowtab * 16; i += NW * 64) rtab[i] = rowtab[i];
  __syncthreads();
  {
    const float sc = 1.0f + (float)(b & 7) * 0.125f;
    for (int i = tid; i < 16 * C0; i += NW * 64) { const int r = i / C0, k = i - r * C0; lds[in0_off + (k >> 4) * CS + r * 16 + (k & 15)] = x0[i] * sc; }
  }
  __syncthreads();
  unsigned asel[D];
  auto lane_off = [&](const PItem& it) { return (unsigned)(((wave & ((1 << it.tshift) - 1)) * it.pair_stride + lane * 4) * 4); };
  auto slice_steps = [&](const PItem& it) { return steps + it.step_base + (wave >> it.tshift) * it.nkp; };
#define MFMA4_SLOT(r) do { \
    if (r == 0) RING_MFMA4(acc0, acc1, q0, q1, q2, q3, 12, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15); \
    else if (r == 1) RING_MFMA4(acc0, acc1, q0, q1, q2, q3, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31); \
    else if (r == 2) RING_MFMA4(acc0, acc1, q0, q1, q2, q3, 12, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47); \
    else RING_MFMA4(acc0, acc1, q0, q1, q2, q3, 12, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63); } while (0)
  PItem cur = items[0];
  unsigned off_c = lane_off(cur);
  const PStep* st_c = slice_steps(cur);
#pragma unroll
  for (int r = 0; r < D; ++r) {
    const PStep e = st_c[r];
    asel[r] = e.a_sel;
    LOAD_SLOT(r, e, off_c);
  }
  int rs = rtab[(asel[0] >> 16) * 16 + ri];
  const float* ap0 = lds + (asel[0] & 0xffff) + rs * 16;
  f32x4 q0 = *reinterpret_cast<const f32x4*>(ap0), q1 = *reinterpret_cast<const f32x4*>(ap0 + 4), q2 = *reinterpret_cast<const f32x4*>(ap0 + 8), q3 = *reinterpret_cast<const f32x4*>(ap0 + 12);
  rs = rtab[(asel[1] >> 16) * 16 + ri];
  for (int ii = 0; ii < nitems; ++ii) {
    const PItem nxt = items[ii + 1 < nitems ? ii + 1 : 0];
    const unsigned off_n = lane_off(nxt);
    const PStep* st_n = slice_steps(nxt);
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 0] = __builtin_amdgcn_s_memtime();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const int nu = cur.nkp / D;
    for (int u = 0; u < nu; ++u) {
      const bool last = u == nu - 1;
      const PStep* lst = last ? st_n : st_c + (u + 1) * D;
      const unsigned loff = last ? off_n : off_c;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        const float* apn = lds + (asel[(r + 1) % D] & 0xffff) + rs * 16;
        const f32x4 n0 = *reinterpret_cast<const f32x4*>(apn), n1 = *reinterpret_cast<const f32x4*>(apn + 4), n2 = *reinterpret_cast<const f32x4*>(apn + 8), n3 = *reinterpret_cast<const f32x4*>(apn + 12);
        const PStep e = lst[r];
        MFMA4_SLOT(r);
        asel[r] = e.a_sel;
        rs = rtab[(asel[(r + 2) % D] >> 16) * 16 + ri];
        LOAD_SLOT(r, e, loff);
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
      }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 1] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int v = 0; v < 4; ++v) red[(wave * 4 + v) * 64 + lane] = acc0[v] + acc1[v];
    __syncthreads();
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 2] = __builtin_amdgcn_s_memtime();
    const int tiles = 1 << cur.tshift, ksplit = 1 << cur.kshift, quads = tiles * 16;
    for (int idx = tid; idx < 4 * quads; idx += NW * 64) {
      const int row = idx / quads, c4 = idx - row * quads, tl = c4 >> 4, sub = c4 & 15;
      f32x4 s = {0, 0, 0, 0};
      for (int sl = 0; sl < ksplit; ++sl) s += *reinterpret_cast<const f32x4*>(red + ((sl * tiles + tl) * 4 + row) * 64 + 4 * sub);
      const int col = (cur.tile0 + tl) * 64 + 4 * sub;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(cur.bias + col);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = gelu(s[e] + bi[e]) * cur.oscale;
      *reinterpret_cast<f32x4*>(lds + cur.out_off + (col >> 4) * CS + row * 16 + (col & 15)) = y;
    }
    __syncthreads();
    if (stamps && b == 0 && tid == 0) stamps[ii * 4 + 3] = __builtin_amdgcn_s_memtime();
    {
      int rs0 = rtab[(asel[0] >> 16) * 16 + ri];
      const float* apn = lds + (asel[0] & 0xffff) + rs0 * 16;
      q0 = *reinterpret_cast<const f32x4*>(apn); q1 = *reinterpret_cast<const f32x4*>(apn + 4); q2 = *reinterpret_cast<const f32x4*>(apn + 8); q3 = *reinterpret_cast<const f32x4*>(apn + 12);
    }
    cur = nxt; off_c = off_n; st_c = st_n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: RING_CLOBBER_0, RING_CLOBBER_1, RING_CLOBBER_2, RING_CLOBBER_3);
  for (int i = tid; i < 16 * Cl; i += NW * 64) { const int r = i / Cl, k = i - r * Cl; out[(long long)b * 16 * Cl + i] = lds[outl_off + (k >> 4) * CS + r * 16 + (k & 15)]; }
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

struct HostOp { int N, Kp, ntap, in_off, out_off; long long w_off, b_off, wf_off; float oscale; signed char src[16][16]; };
struct Tables { std::vector<PItem> items; std::vector<PStep> steps; std::vector<unsigned char> rowtab; };
// fragment-order copy of an op's weights: [pair][step f = tap * ndc + dc][half][tile of the pair][lane][4]
static void pack_fragments(const std::vector<float>& arena_in, const HostOp& o, std::vector<float>* out) {
  const int ndc = o.Kp / 32, F = o.ntap * ndc, pairs = o.N / 32;
  out->assign((size_t)pairs * F * 1024, 0.f);
  for (int tp = 0; tp < pairs; ++tp) for (int f = 0; f < F; ++f) {
    const int tap = f / ndc, dc = f % ndc;
    float* blk = out->data() + ((size_t)tp * F + f) * 1024;
    for (int h = 0; h < 2; ++h) for (int t = 0; t < 2; ++t) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
      const int nn = 32 * tp + 16 * t + (lane & 15), k = 32 * dc + 16 * h + 4 * (lane >> 4) + i;
      blk[((h * 2 + t) * 64 + lane) * 4 + i] = arena_in[o.w_off + ((size_t)tap * o.N + nn) * o.Kp + k];
    }
  }
}
// small path: [tile of 64 columns][step f = tap * (Kp / 16) + dc][k group j][lane][4]
static void pack_fragments_small(const std::vector<float>& arena_in, const HostOp& o, std::vector<float>* out) {
  const int ndc = o.Kp / 16, F = o.ntap * ndc, tiles = o.N / 64;
  out->assign((size_t)tiles * F * 1024, 0.f);
  for (int tl = 0; tl < tiles; ++tl) for (int f = 0; f < F; ++f) {
    const int tap = f / ndc, dc = f % ndc;
    float* blk = out->data() + ((size_t)tl * F + f) * 1024;
    for (int j = 0; j < 4; ++j) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i)
      blk[(j * 64 + lane) * 4 + i] = arena_in[o.w_off + ((size_t)tap * o.N + 64 * tl + lane) * o.Kp + 16 * dc + 4 * j + i];
  }
}
static Tables build_tables_small(const std::vector<HostOp>& ops, int NW, int D, const float* d_arena) {
  Tables t;
  t.rowtab.assign(16, 16);
  for (const HostOp& o : ops) {
    const int first_tab = (int)t.rowtab.size() / 16;
    for (int tp = 0; tp < o.ntap; ++tp) for (int r = 0; r < 16; ++r) t.rowtab.push_back(o.src[tp][r] < 0 ? 16 : (unsigned char)o.src[tp][r]);
    const int T = o.N / 64, ndc = o.Kp / 16, F = o.ntap * ndc;
    for (int t0 = 0; t0 < T;) {
      int tiles = 1; while (tiles * 2 <= NW && tiles * 2 <= T - t0) tiles *= 2;
      const int ksplit = NW / tiles;
      const int nk = (F + ksplit - 1) / ksplit, nkp = (nk + D - 1) / D * D;
      PItem it{};
      it.step_base = (int)t.steps.size(); it.nkp = nkp; it.tile0 = t0; it.pair_stride = F * 1024; it.out_off = o.out_off; it.oscale = o.oscale;
      it.bias = d_arena + o.b_off;
      int sh = 0; while ((1 << sh) < tiles) ++sh; it.tshift = sh;
      sh = 0; while ((1 << sh) < ksplit) ++sh; it.kshift = sh;
      for (int sl = 0; sl < ksplit; ++sl)
        for (int j = 0; j < nkp; ++j) {
          const int f = sl + j * ksplit;
          PStep e;
          if (f < F) { const int tp = f / ndc, dc = f % ndc; e.w_off = (unsigned)(o.wf_off + ((long long)t0 * F + f) * 1024); e.a_sel = (unsigned)(o.in_off + dc * CS) | ((unsigned)(first_tab + tp) << 16); }
          else { e.w_off = (unsigned)(o.wf_off + (long long)t0 * F * 1024); e.a_sel = (unsigned)o.in_off; }
          t.steps.push_back(e);
        }
      t.items.push_back(it);
      t0 += tiles;
    }
  }
  return t;
}
// rowtab entry 0 is the no-op table (all rows -> the zero row 16)
static Tables build_tables(const std::vector<HostOp>& ops, int NW, int D, const float* d_arena) {
  Tables t;
  t.rowtab.assign(16, 16);
  for (const HostOp& o : ops) {
    const int first_tab = (int)t.rowtab.size() / 16;
    for (int tp = 0; tp < o.ntap; ++tp) for (int r = 0; r < 16; ++r) t.rowtab.push_back(o.src[tp][r] < 0 ? 16 : (unsigned char)o.src[tp][r]);
    const int T = o.N / 32, ndc = o.Kp / 32, F = o.ntap * ndc;
    for (int t0 = 0; t0 < T;) {
      int pairs = 1; while (pairs * 2 <= NW && pairs * 2 <= T - t0) pairs *= 2;
      const int ksplit = NW / pairs;
      const int nk = (F + ksplit - 1) / ksplit, nkp = (nk + D - 1) / D * D;
      PItem it{};
      it.step_base = (int)t.steps.size(); it.nkp = nkp; it.tile0 = 2 * t0; it.pair_stride = F * 1024; it.out_off = o.out_off; it.oscale = o.oscale;
      it.bias = d_arena + o.b_off;
      int sh = 0; while ((1 << sh) < pairs) ++sh; it.tshift = sh;
      sh = 0; while ((1 << sh) < ksplit) ++sh; it.kshift = sh;
      for (int sl = 0; sl < ksplit; ++sl)
        for (int j = 0; j < nkp; ++j) {
          const int f = sl + j * ksplit;
          PStep e;
          if (f < F) { const int tp = f / ndc, dc = f % ndc; e.w_off = (unsigned)(o.wf_off + ((long long)t0 * F + f) * 1024); e.a_sel = (unsigned)(o.in_off + 2 * dc * CS) | ((unsigned)(first_tab + tp) << 16); }
          else { e.w_off = (unsigned)(o.wf_off + (long long)t0 * F * 1024); e.a_sel = (unsigned)o.in_off; }
          t.steps.push_back(e);
        }
      t.items.push_back(it);
      t0 += pairs;
    }
  }
  return t;
}

int main(int argc, char** argv) {
  const int part = argc > 1 ? atoi(argv[1]) : 7;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* d_in; float* d_o;
  CK(hipMalloc(&d_in, 4096 * 4)); CK(hipMalloc(&d_o, 1024 * 1024 * 4));
  { std::vector<float> h(4096); unsigned s = 7; for (auto& v : h) v = frand(s); CK(hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice)); }
  if (part & 1) {
    const int iters = 4096;
#define MF(K, NAME, NACC, FLOP) \
    for (int nw : {4, 8, 16}) { \
      float best = 1e9f; \
      for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((K<NACC>), dim3(256), dim3(nw * 64), 0, 0, d_in, d_o, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
      printf("%s nacc=%d waves/SIMD=%d: %8.1f us  %6.1f TFLOP/s  (%.1f ns per MFMA per wave)\n", NAME, NACC, nw / 4, best * 1e3, 256.0 * nw * iters * 8 * NACC * FLOP / (best * 1e-3) / 1e12, best * 1e6 / (iters * 8.0 * NACC)); \
    }
    MF(mf16, "16x16x4", 1, 2048.0) MF(mf16, "16x16x4", 2, 2048.0) MF(mf16, "16x16x4", 4, 2048.0)
    MF(mf32, "32x32x2", 1, 4096.0) MF(mf32, "32x32x2", 2, 4096.0)
    MF(mf4, "4x4x1  ", 1, 512.0) MF(mf4, "4x4x1  ", 2, 512.0) MF(mf4, "4x4x1  ", 4, 512.0)
  }
  if (part & 2) {
    const double mb = 9.437184;
    const long long floats = (long long)(mb * 1e6 / 4);
    float* w; CK(hipMalloc(&w, floats * 4 + (1 << 20))); CK(hipMemset(w, 0, floats * 4 + (1 << 20)));
#define WS(MODE, DD, NWV, ROWF) { \
      float best = 1e9f; \
      for (int rep = 0; rep < 4; ++rep) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((wstream<MODE, DD>), dim3(256), dim3(NWV * 64), 0, 0, w, floats, ROWF, d_o); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms; } \
      printf("stream mode %d depth %d waves %2d rowf %4d: %7.1f us  %6.1f GB/s per WG  %5.2f TB/s aggregate\n", MODE, DD, NWV, ROWF, best * 1e3, mb * 1e6 / (best * 1e-3) / 1e9, 256 * mb * 1e6 / (best * 1e-3) / 1e12); }
    WS(0, 2, 8, 0) WS(0, 4, 8, 0) WS(0, 2, 16, 0) WS(0, 4, 16, 0) WS(0, 8, 16, 0)
    WS(1, 2, 8, 128) WS(1, 4, 8, 128) WS(1, 2, 16, 128) WS(1, 4, 16, 128) WS(1, 8, 16, 128)
    WS(1, 4, 16, 256) WS(1, 4, 16, 64) WS(1, 4, 16, 32)
    CK(hipFree(w));
  }
  if (part & 4) {
    // eight 3x3 convolutions at 4x4: 128 -> 256 -> 128 -> ... (1.18 MB of weights each); `small` = the same at 2x2 (rows 4..15 idle)
    for (int small = 0; small < 2; ++small) {
    const int nops = 8;
    std::vector<HostOp> ops(nops);
    std::vector<float> arena;
    unsigned seed = 12345;
    const int slabA = 0, slabB = 17 * CS;                          // two ping-pong slabs of 256 channels (+ one chunk of slack each)
    const int act_floats = 2 * 17 * CS;
    double total_mb = 0, flop = 0;
    for (int k = 0; k < nops; ++k) {
      const int cin = (k & 1) ? 256 : 128, cout = (k & 1) ? 128 : 256;
      HostOp& o = ops[k]; o.N = cout; o.Kp = cin; o.ntap = 9;
      o.in_off = (k & 1) ? slabB : slabA; o.out_off = (k & 1) ? slabA : slabB; o.oscale = 1.0f;
      const int S = small ? 2 : 4;
      for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) {
        const int y = r / S + t / 3 - 1, x = r % S + t % 3 - 1;
        o.src[t][r] = (r >= S * S || y < 0 || y >= S || x < 0 || x >= S) ? -1 : (signed char)(y * S + x);
      }
      o.w_off = (long long)arena.size();
      const float ws = 1.5f / std::sqrt(9.0f * cin);
      for (size_t i = 0; i < (size_t)9 * cout * cin; ++i) arena.push_back(frand(seed) * ws);
      o.b_off = (long long)arena.size();
      for (int i = 0; i < cout; ++i) arena.push_back(frand(seed) * 0.1f);
      total_mb += 9.0 * cout * cin * 4 / 1e6; flop += 2.0 * 16 * 9 * cin * cout;
    }
    std::vector<long long> wf_big(nops), wf_small(nops);
    for (int k = 0; k < nops; ++k) {                             // fragment-order copies behind the reference layout
      std::vector<float> fr; pack_fragments(arena, ops[k], &fr);
      wf_big[k] = (long long)arena.size();
      arena.insert(arena.end(), fr.begin(), fr.end());
      pack_fragments_small(arena, ops[k], &fr);
      wf_small[k] = (long long)arena.size();
      arena.insert(arena.end(), fr.begin(), fr.end());
      ops[k].wf_off = wf_big[k];
    }
    const int C0 = 128, Cl = 128;
    std::vector<float> x0(16 * C0); for (auto& v : x0) v = frand(seed);
    if (small) for (int r = 4; r < 16; ++r) for (int c = 0; c < C0; ++c) x0[r * C0 + c] = 0.f;
    std::vector<float> ref((size_t)8 * 16 * Cl);
    for (int v = 0; v < 8; ++v) {
      std::vector<float> cur(16 * 256, 0.f), nxt(16 * 256, 0.f);
      const float sc = 1.0f + v * 0.125f;
      for (int r = 0; r < 16; ++r) for (int c = 0; c < C0; ++c) cur[r * 256 + c] = x0[r * C0 + c] * sc;
      for (int k = 0; k < nops; ++k) {
        const HostOp& o = ops[k];
        for (int r = 0; r < 16; ++r) for (int nn = 0; nn < o.N; ++nn) {
          double sacc = 0;
          for (int t = 0; t < 9; ++t) { const int sr = o.src[t][r]; if (sr < 0) continue; const float* wr = &arena[o.w_off + ((size_t)t * o.N + nn) * o.Kp]; for (int c = 0; c < o.Kp; ++c) sacc += (double)cur[sr * 256 + c] * wr[c]; }
          const double u = sacc + arena[o.b_off + nn];
          nxt[r * 256 + nn] = (float)(0.5 * u * (1.0 + std::erf(u * 0.70710678118654752)));
        }
        cur.swap(nxt);
      }
      for (int r = 0; r < 16; ++r) for (int c = 0; c < Cl; ++c) ref[((size_t)v * 16 + r) * Cl + c] = cur[r * 256 + c];
    }
    float* d_arena; CK(hipMalloc(&d_arena, arena.size() * 4)); CK(hipMemcpy(d_arena, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    float* d_x0; CK(hipMalloc(&d_x0, x0.size() * 4)); CK(hipMemcpy(d_x0, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
    float* d_out; CK(hipMalloc(&d_out, (size_t)256 * 16 * Cl * 4));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 4 * 8));
    printf("GEMM chain (%s): %d convolutions 3x3, %.2f MB of weights and %.1f MFLOP (16-row) per workgroup, 256 workgroups\n", small ? "2x2" : "4x4", nops, total_mb, flop / 1e6);
    auto run = [&](auto kern, int NW, int D, const char* tag, bool small_path = false) {
      for (int k = 0; k < nops; ++k) ops[k].wf_off = small_path ? wf_small[k] : wf_big[k];
      Tables t = small_path ? build_tables_small(ops, NW, D, d_arena) : build_tables(ops, NW, D, d_arena);
      PItem* d_items; PStep* d_steps; unsigned char* d_rt;
      CK(hipMalloc(&d_items, t.items.size() * sizeof(PItem))); CK(hipMemcpy(d_items, t.items.data(), t.items.size() * sizeof(PItem), hipMemcpyHostToDevice));
      CK(hipMalloc(&d_steps, t.steps.size() * sizeof(PStep))); CK(hipMemcpy(d_steps, t.steps.data(), t.steps.size() * sizeof(PStep), hipMemcpyHostToDevice));
      CK(hipMalloc(&d_rt, t.rowtab.size())); CK(hipMemcpy(d_rt, t.rowtab.data(), t.rowtab.size(), hipMemcpyHostToDevice));
      const size_t ldsb = sizeof(float) * (act_floats + NW * 512) + t.rowtab.size() + 64;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      const int nitems = (int)t.items.size();
      CK(hipMemset(d_st, 0, 64 * 4 * 8));
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), ldsb, 0, d_items, nitems, d_steps, d_rt, (int)t.rowtab.size() / 16, d_arena, d_x0, C0, ops[0].in_off, d_out, Cl, ops[nops - 1].out_off, act_floats,
                           rep == 5 ? d_st : nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 1 && rep < 5 && ms < best) best = ms;
      }
      std::vector<float> got((size_t)256 * 16 * Cl);
      CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
      double maxerr = 0, maxref = 0;
      for (int bb : {0, 3, 255}) for (int i = 0; i < (small_path ? 4 : 16) * Cl; ++i) {      // the 4-row path never writes the idle rows
        const double r = ref[(size_t)(bb & 7) * 16 * Cl + i];
        maxerr = std::max(maxerr, std::fabs((double)got[(size_t)bb * 16 * Cl + i] - r)); maxref = std::max(maxref, std::fabs(r));
      }
      std::vector<unsigned long long> st(64 * 4);
      CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
      printf("%-26s %7.1f us  %5.1f GB/s per WG  %5.1f TFLOP/s (16-row)  max err %.2e (max |ref| %.2e)  steps %zu\n", tag, best * 1e3, total_mb * 1e6 / (best * 1e-3) / 1e9,
             256 * flop / (best * 1e-3) / 1e12, maxerr, maxref, t.steps.size());
      printf("    per item (WG 0, cycles: kloop / partials+sync / epilogue+sync):");
      for (int k = 0; k < nitems && k < 16; ++k) { const unsigned long long* q = &st[k * 4]; printf("  [%llu/%llu/%llu]", q[1] - q[0], q[2] - q[1], q[3] - q[2]); }
      printf("\n");
      CK(hipFree(d_items)); CK(hipFree(d_steps)); CK(hipFree(d_rt));
    };
    run(gemm_chain<8>, 8, 4, "8 waves, depth 4");
    if (small) {
      run(gemm_chain_small<8>, 8, 4, "8 waves, 4x4x1 path", true);
      run(gemm_chain_small<16>, 16, 4, "16 waves, 4x4x1 path", true);
      run(gemm_chain_small<4>, 4, 4, "4 waves, 4x4x1 path", true);
    }
    CK(hipFree(d_arena)); CK(hipFree(d_x0)); CK(hipFree(d_out)); CK(hipFree(d_st));
    }
  }
  return 0;
}
