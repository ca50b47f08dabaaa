// pixrun.hip -- every level of an 8x8 score net (8x8, 4x4, 2x2, 1x1) as RUNS of ops executed by SAMPLE-OWNED workgroups: for the PolyMNIST nets two
// launches per forward (the 8x8 attention block between them is kernels.hip attnblock_kernel), which in a sampling loop also hold init_conv (an im2col
// op + a 1x1 conv), the conv that writes the NCHW score, and the EM update (kernel tail) -- DESIGN.md 3b', 3b'', 3b'''.
//
// Nothing in the network mixes samples, so a workgroup takes ONE sample through a whole run of ops alone: no barrier
// ever leaves the workgroup, the sample's activations stay in LDS from the first op to the last, and the run's weights are streamed
// from L2 (all workgroups of an XCD walk the same bytes in the same order).  Round 3 rewrite, round 4 additions; what the measurements behind it
// say (tools/pixgemm_probe.hip, profiles/r03_pixgemm_probe.txt):
//
//   weights   : streamed in MFMA FRAGMENT order, packed on the host at model creation, so that every wave-load is 1 KB contiguous:
//               118 GB/s per workgroup (30 TB/s out of the L2s) against 37 GB/s for 64-byte row segments of a [n][k] matrix.
//   ring      : a wave keeps D = 4 steps of 4 KB in flight in AGPRs with hard-coded indices, loaded and consumed by inline asm (the MFMA
//               takes its B operand straight from the AGPR) behind hand-counted s_waitcnt vmcnt.  hipcc cannot be asked for this: its
//               own loads are clustered at the end of the unrolled body and waited for with vmcnt(0), and asm loads into registers IT
//               allocates get copied (v_mov in a peeled iteration) before the hand-placed wait -- silently wrong values.
//   steps     : the K loop of every GEMM is a host-built table of steps {weight block, LDS chunk | row table}, dealt to the wave groups
//               that split K and padded to a multiple of D with no-op steps; the loader simply runs D steps ahead of the consumer ACROSS
//               op boundaries, so the next GEMM's first weights are in flight under this one's reduction and epilogue.
//   MFMA      : >= 2 independent accumulators per wave (one dependent chain per wave does NOT reach the issue rate at 2 waves per SIMD:
//               98 vs 153 TFLOP/s).  16 output rows (4x4): v_mfma_f32_16x16x4_f32, a wave owns 32 columns (2 accumulators); the K loop
//               then runs at the matrix pipe's issue limit.  <= 4 output rows (2x2, 1x1): v_mfma_f32_4x4x1_16b_f32, 16 blocks = 16
//               column quads sharing one k, a wave owns 64 columns, even / odd k in two accumulators -- a quarter of the issue time
//               per weight byte, which makes these levels L2-stream bound instead of matrix-pipe bound.
//   rows      : activations live in LDS slabs T[k >> 4][17 rows][16 k]; row 16 of every chunk is never written and stays zero: the
//               source of padding taps and idle rows, so a tap is a plain row gather (row table per tap, built on the host) with no
//               branch.  Padding, stride, transposed convolution, nearest upsampling and the fused 1x1 residual conv are all tables.
//
// Everything belongs to the model handle (PixState, owned by DevPlan): no process-global maps.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "device.h"
#include "sampler_dev.h"

namespace sbm {

namespace {

typedef float f32x4p __attribute__((ext_vector_type(4)));

// conditions that are false in production launches (stamps, debug sub-ranges): their blocks go to the end of the function, out of the hot lines
#ifdef PIX_EXP_GOTO1
#define PIX_COLD(c) (c)
#else
#define PIX_COLD(c) __builtin_expect(!!(c), 0)
#endif
constexpr int PNT = 512, PNW = 8;         // threads / waves per workgroup (2 waves per SIMD)
constexpr int PD = 4;                     // ring depth in steps (4 KB per wave and step)
constexpr int PCS = 272;                  // floats per LDS chunk: 17 rows x 16 k
constexpr int PIX_STAGE = 101;
constexpr int PIX_VEC = 102;              // a GEMM with ONE output pixel as a vector-ALU mat-vec (no MFMA, no AGPR ring): see the kernel
constexpr int PIX_DW1 = 103, PIX_GNR1 = 104;   // the depth-wise op / GroupNorm + residual at ONE position, inside a run of PIX_VEC ops
constexpr int P_RED_F = PNW * 512, P_STAT_SLOTS = 64, P_SCR_F = 4352;

struct PixStep { unsigned w_off; unsigned a_sel; };   // w_off: float offset (fragment arena) of the step's block for column group 0 of the pass;
                                                      // a_sel = LDS float offset of the step's first chunk | row table << 16 | source 2 << 31
enum : int { PIF_SMALL = 1, PIF_GN = 2, PIF_SCALE2 = 4, PIF_FIRST = 8, PIF_LAST = 16, PIF_NCHW = 32 };      // PIF_NCHW: the output is the plan's external (B, N, H, W) tensor
struct PixItem {                          // one pass of a GEMM: up to 8 column groups (32 columns, or 64 on the 4-row path) -- and EVERYTHING the pass needs,
                                          // so that the kernel reads nothing but this table (scalar loads) on the GEMM path
  int step_base, nkp;                     // step lists of the K slices: step_base + slice * nkp, nkp a multiple of PD
  int gshift, kshift;                     // log2(column groups of the pass), log2(K slices) -- groups * slices = 8 waves
  int grp0, grp_stride;                   // first column group; floats between the blocks of consecutive groups
  int flags, P_out;                       // PIF_*: 4-row path, folded GroupNorm on source 1, source 2 enters unnormalised, first / last pass of its op
  int Npad, nstore, act, ldo;
  int gn_stat, stat_out, out_lds, res_lds; // LDS statistics slots consumed / produced (-1 none), output / residual slabs (-1 none)
  float gn_inv; int rowstride;            // 1 / (channels x pixels) of the GroupNorm; Npad when the folded-GroupNorm tables are per output pixel, else 0
  int res_row[4];                         // 16 bytes: residual row of output row r (outputs of at most 16 rows; 64-row outputs read their own row)
  int rshift, rg0;                        // 64-row outputs (8x8): log2(row groups of 16 in the pass) and its first row group; 0, 0 otherwise
  const float* bias; const float* gn_s; const float* gn_b1;     // never null: a zero vector stands in
  float* out_g; float2* stat_g;           // workspace copies of the output / its statistics, when something outside the launch reads them
  int out_sub, res_sub;                   // floats between the four 16-row sub-slabs of a 64-row output / residual tensor
  int rg_stride, perm_tab;                // phase-grouped transposed convolutions: steps between the row groups' OWN step lists (0: shared list); row table
                                          // (16-entry units) that maps a pass row to its output pixel, -1 none
};
static_assert(sizeof(PixItem) == 152, "PixItem layout");

struct PixOpD {
  int kind;                              // OP_GEMM, OP_DWCONV7, OP_GN_RES, OP_LINATTN, PIX_STAGE
  int nitems;                            // GEMM: its passes (consecutive entries of the item table)
  int gn_stat, stat_out;                 // LDS statistics slots consumed / produced, -1 none
  int in_lds, out_lds, x_lds, pad;
  int in_sub, out_sub, x_sub, st_sub;    // floats between the 16-row sub-slabs of 64-row tensors (8x8)
  float* out_g; float2* stat_g;          // non-null: the result / its statistics also go to the workspace
  DwArgs dw; GnResArgs gr; AttnArgs at;
  const float* st_src; const float2* st_part; int st_ld, st_C, st_P, st_lds, st_stat, st_nparts;
  // PIX_VEC: out[n] = sum_k a[k] W[k][n] over 512 threads = N columns x S K-slices (N S = 512); steps of 8 KB: thread (column, slice) takes one float4 of weights
  // and the matching four activations of its slice.  Source 1 (v_n1 steps; GroupNorm folded as in the MFMA path) and source 2 (v_n2 steps, raw).
  int v_run, v_stream, v_slots, v_recs;   // on the FIRST op of a run: program ops in the run, step index where the run's slot table begins, its slots,
                                          // step index of the run's 64-byte records (PixVecRec words, one record per program op of the run)
  int v_step_base, v_n1, v_n2, v_sshift;  // v_n1: slots (four steps each) of this op, a multiple of 4
  int v_Npad, v_nstore, v_act, v_ldo, v_gn, v_res_lds;     // v_res_lds: float offset of the residual ROW (-1 none)
  float v_gn_inv; int v_pad;
  const float* v_bias; const float* v_gn_s; const float* v_gn_b1;
};

// The step / item / op tables are read through the CONSTANT address space: with a uniform address that is always a scalar load (s_load).
// Read as ordinary global memory they turn into vector loads -- the kernel stores through pointers it has loaded (op.out_g, ...), so hipcc
// cannot prove the tables unclobbered -- and every vector load drags a compiler-placed s_waitcnt vmcnt(0) into the K loop, which drains
// the weight ring.
typedef int i32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ PixStep ldc_step(const PixStep* p) {
  const i32x2p v = *(const __attribute__((address_space(4))) i32x2p*)(unsigned long long)p;
  return PixStep{(unsigned)v[0], (unsigned)v[1]};
}
// A 64-byte record as sixteen separate scalar words.  (NOT as one ext_vector_type(16) value: with the record in s[16:31] hipcc 7.2 read lane 4 --
// the GroupNorm 1 / count -- out of the register that held lane 0, and in the loop-carried form kept the loop counter in the tuple's lane 4:
// wrong moments, no diagnostic.  Separate loads from one base are merged into wide s_loads by the compiler itself, correctly.)
struct Rec16 { int w[16]; __device__ __forceinline__ int operator[](int i) const { return w[i]; } };
__device__ __forceinline__ Rec16 ldc_x16(const void* p) {
  Rec16 r;
  const __attribute__((address_space(4))) int* q = (const __attribute__((address_space(4))) int*)(unsigned long long)p;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = q[i];
  return r;
}
typedef Rec16 i32x16p;
// eight table words (separate scalar words, for the same reason).  The step groups of the one-position runs are stored with the weight offsets
// first and the four activation selectors behind them.
struct Rec8 { int w[8]; __device__ __forceinline__ int operator[](int i) const { return w[i]; } };
__device__ __forceinline__ Rec8 ldc_x8(const void* p) {
  Rec8 r;
  const __attribute__((address_space(4))) int* q = (const __attribute__((address_space(4))) int*)(unsigned long long)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = q[i];
  return r;
}
typedef Rec8 i32x8p;
template <typename T> __device__ __forceinline__ T ldc(const T* p) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "ldc: 4- or 8-byte fields");
  if constexpr (sizeof(T) == 4) { const int v = *(const __attribute__((address_space(4))) int*)(unsigned long long)p; T r; __builtin_memcpy(&r, &v, 4); return r; }
  else { const i32x2p v = *(const __attribute__((address_space(4))) i32x2p*)(unsigned long long)p; T r; __builtin_memcpy(&r, &v, 8); return r; }
}
// Sum over the 64 lanes without LDS traffic (__shfl_xor is a ds_bpermute per step): quad swaps, half-row and row mirrors by DPP give every lane
// its 16-lane row's sum; the four row sums are read with v_readlane and added in a fixed order.  Every lane returns the same total.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
// float offset of channel k of row r inside a slab
__device__ __forceinline__ int pix_at(int k, int r) { return (k >> 4) * PCS + r * 16 + (k & 15); }
// ... of a tensor of up to 64 rows: four 17-row sub-slabs (rows 16 s .. 16 s + 15 in sub-slab s), `sub` floats apart -- so every chunk keeps
// its own zero row and the GEMM's row gather stays one table look-up whatever the resolution
__device__ __forceinline__ int pix_at2(int k, int r, int sub) { return (r >> 4) * sub + (k >> 4) * PCS + (r & 15) * 16 + (k & 15); }
// a load through a pointer that was itself read from a table: the compiler only knows a generic pointer (flat_load); this says "global"
__device__ __forceinline__ float ldg(const float* p, int i) { return ((const __attribute__((address_space(1))) float*)(unsigned long long)p)[i]; }
__device__ __forceinline__ int sbyte(const int* w, int r) { return (int)(signed char)((w[r >> 2] >> (8 * (r & 3))) & 0xff); }

// ---- the weight ring: AGPRs.  With any AGPR in use hipcc splits the 256 registers of a 2-waves-per-SIMD kernel 128 / 128 (no attribute of
// this LLVM moves that line), and past 128 VGPRs it SPILLS INTO AGPRs -- lowest free index first.  Between two asm statements nothing tells it
// that the ring is live (clobber lists only protect values that live ACROSS a statement), and a first version with the ring at a0.. had slot 0
// overwritten by spills inside the depth-wise op: wrong first K steps, no fault, no message.  So the ring sits at the TOP of the AGPR half
// (a40 .. a127: slot r = a[40 + 16 r ..], epilogue operands a[104 .. 127]), a0 .. a39 are left to the compiler, and the build audit
// (tests/test_host_api.py::test_pixel_kernel_register_audit) fails when a compiler-generated instruction names a40 or above.
#define PIX_CLOB_0 "a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55"
#define PIX_CLOB_1 "a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71"
#define PIX_CLOB_2 "a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87"
#define PIX_CLOB_3 "a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103"
#define PIX_CLOB_E "a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
#define PIX_RING_LOAD(voff, base, A0, A3, A4, A7, A8, A11, A12, A15, CLOB) \
  asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, %1\n\t" \
               "global_load_dwordx4 a[" #A4 ":" #A7 "], %0, %1 offset:1024\n\t" \
               "global_load_dwordx4 a[" #A8 ":" #A11 "], %0, %1 offset:2048\n\t" \
               "global_load_dwordx4 a[" #A12 ":" #A15 "], %0, %1 offset:3072" \
               :: "v"(voff), "s"(base) : CLOB)
#define PIX_LOAD_SLOT(r, base_, off) do { \
    if ((r) == 0) PIX_RING_LOAD(off, base_, 40, 43, 44, 47, 48, 51, 52, 55, PIX_CLOB_0); \
    else if ((r) == 1) PIX_RING_LOAD(off, base_, 56, 59, 60, 63, 64, 67, 68, 71, PIX_CLOB_1); \
    else if ((r) == 2) PIX_RING_LOAD(off, base_, 72, 75, 76, 79, 80, 83, 84, 87, PIX_CLOB_2); \
    else PIX_RING_LOAD(off, base_, 88, 91, 92, 95, 96, 99, 100, 103, PIX_CLOB_3); } while (0)
// vmcnt(12): the three other slots' loads (12) are always issued behind a slot's own four before it is consumed again -- the loader never
// skips a step -- and anything else in flight (epilogue operands, stores, the compiler's own loads) only makes the wait stricter.
#define PIX_MF16(ACC, A, B) "v_mfma_f32_16x16x4_f32 " ACC ", " A ", a" #B ", " ACC "\n\t"
#define PIX_MFMA16(acc0, acc1, a0, a1, B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, B10, B11, B12, B13, B14, B15) \
  asm volatile("s_waitcnt vmcnt(12)\n\t" \
               PIX_MF16("%0", "%2", B0) PIX_MF16("%1", "%2", B4) PIX_MF16("%0", "%3", B1) PIX_MF16("%1", "%3", B5) PIX_MF16("%0", "%4", B2) PIX_MF16("%1", "%4", B6) PIX_MF16("%0", "%5", B3) PIX_MF16("%1", "%5", B7) \
               PIX_MF16("%0", "%6", B8) PIX_MF16("%1", "%6", B12) PIX_MF16("%0", "%7", B9) PIX_MF16("%1", "%7", B13) PIX_MF16("%0", "%8", B10) PIX_MF16("%1", "%8", B14) PIX_MF16("%0", "%9", B11) "v_mfma_f32_16x16x4_f32 %1, %9, a" #B15 ", %1" \
               : "+v"(acc0), "+v"(acc1) : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]))
#define PIX_MFMA16_SLOT(r, a0, a1) do { \
    if ((r) == 0) PIX_MFMA16(acc0, acc1, a0, a1, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55); \
    else if ((r) == 1) PIX_MFMA16(acc0, acc1, a0, a1, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71); \
    else if ((r) == 2) PIX_MFMA16(acc0, acc1, a0, a1, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87); \
    else PIX_MFMA16(acc0, acc1, a0, a1, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97, 98, 99, 100, 101, 102, 103); } while (0)
#define PIX_MF4(ACC, A, B) "v_mfma_f32_4x4x1_16b_f32 " ACC ", " A ", a" #B ", " ACC "\n\t"
#define PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, B10, B11, B12, B13, B14, B15) \
  asm volatile("s_waitcnt vmcnt(12)\n\t" \
               PIX_MF4("%0", "%2", B0) PIX_MF4("%1", "%3", B1) PIX_MF4("%0", "%4", B2) PIX_MF4("%1", "%5", B3) PIX_MF4("%0", "%6", B4) PIX_MF4("%1", "%7", B5) PIX_MF4("%0", "%8", B6) PIX_MF4("%1", "%9", B7) \
               PIX_MF4("%0", "%10", B8) PIX_MF4("%1", "%11", B9) PIX_MF4("%0", "%12", B10) PIX_MF4("%1", "%13", B11) PIX_MF4("%0", "%14", B12) PIX_MF4("%1", "%15", B13) PIX_MF4("%0", "%16", B14) "v_mfma_f32_4x4x1_16b_f32 %1, %17, a" #B15 ", %1" \
               : "+v"(acc0), "+v"(acc1) : "v"(q0[0]), "v"(q0[1]), "v"(q0[2]), "v"(q0[3]), "v"(q1[0]), "v"(q1[1]), "v"(q1[2]), "v"(q1[3]), \
                 "v"(q2[0]), "v"(q2[1]), "v"(q2[2]), "v"(q2[3]), "v"(q3[0]), "v"(q3[1]), "v"(q3[2]), "v"(q3[3]))
#define PIX_MFMA4_SLOT(r) do { \
    if ((r) == 0) PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55); \
    else if ((r) == 1) PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71); \
    else if ((r) == 2) PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87); \
    else PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97, 98, 99, 100, 101, 102, 103); } while (0)
// Timing-only experiment builds (tools/build_variant.sh; the results are WRONG): -DPIX_EXP_NOLOAD drops the K loops' weight loads,
// -DPIX_EXP_NOMFMA their matrix instructions -- what each side costs alone.
#ifdef PIX_EXP_NOMFMA
#undef PIX_MFMA16
#undef PIX_MFMA4
#define PIX_MFMA16(acc0, acc1, a0, a1, ...) asm volatile("s_waitcnt vmcnt(12)" : "+v"(acc0), "+v"(acc1) : "v"(a0[0]), "v"(a1[0]))
#define PIX_MFMA4(acc0, acc1, q0, q1, q2, q3, ...) asm volatile("s_waitcnt vmcnt(12)" : "+v"(acc0), "+v"(acc1) : "v"(q0[0]), "v"(q1[0]), "v"(q2[0]), "v"(q3[0]))
#endif
#ifdef PIX_EXP_NOLOAD
#define PIX_KLOAD_SLOT(r, base_, off) do { asm volatile("" :: "s"(base_), "v"(off)); } while (0)
#else
#define PIX_KLOAD_SLOT(r, base_, off) PIX_LOAD_SLOT(r, base_, off)
#endif
// epilogue operands of one output quad: bias, folded-GroupNorm S and B1 (three dwordx4) into a[BASE .. BASE + 11]
#define PIX_ELOAD(pb, ps, pg, A0, A3, A4, A7, A8, A11) \
  asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, off\n\t" \
               "global_load_dwordx4 a[" #A4 ":" #A7 "], %1, off\n\t" \
               "global_load_dwordx4 a[" #A8 ":" #A11 "], %2, off" \
               :: "v"(pb), "v"(ps), "v"(pg) : PIX_CLOB_E)
#define PIX_EREAD4(dst, A0, A1, A2, A3) \
  asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\tv_accvgpr_read_b32 %3, a" #A3 \
               : "=v"(dst[0]), "=v"(dst[1]), "=v"(dst[2]), "=v"(dst[3]))

// The one-position runs (PIX_VEC) stream their weights through the SAME AGPR ring: a slot = four 8 KB steps = one dwordx4 per thread and step, the
// four of them 8 KB apart (four scalar bases; a mat-vec's steps are contiguous in the arena); consumed through v_accvgpr_read into four float4.
#define PIX_VRING_LOAD(v0, b0, b1, b2, b3, A0, A3, A4, A7, A8, A11, A12, A15, CLOB) \
  asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, %1\n\t" \
               "global_load_dwordx4 a[" #A4 ":" #A7 "], %0, %2\n\t" \
               "global_load_dwordx4 a[" #A8 ":" #A11 "], %0, %3\n\t" \
               "global_load_dwordx4 a[" #A12 ":" #A15 "], %0, %4" \
               :: "v"(v0), "s"(b0), "s"(b1), "s"(b2), "s"(b3) : CLOB)
#ifdef PIX_EXP_NOLOAD_VEC      // timing-only build (tools/build_variant.sh): the one-position runs without their weight stream
#define PIX_VLOAD_SLOT(r, base_) do { asm volatile("" :: "s"(base_), "v"(vo0)); } while (0)
#else
#define PIX_VLOAD_SLOT(r, base_) do { \
    const float* b1_ = (base_) + 2048; const float* b2_ = (base_) + 4096; const float* b3_ = (base_) + 6144; \
    if ((r) == 0) PIX_VRING_LOAD(vo0, base_, b1_, b2_, b3_, 40, 43, 44, 47, 48, 51, 52, 55, PIX_CLOB_0); \
    else if ((r) == 1) PIX_VRING_LOAD(vo0, base_, b1_, b2_, b3_, 56, 59, 60, 63, 64, 67, 68, 71, PIX_CLOB_1); \
    else if ((r) == 2) PIX_VRING_LOAD(vo0, base_, b1_, b2_, b3_, 72, 75, 76, 79, 80, 83, 84, 87, PIX_CLOB_2); \
    else PIX_VRING_LOAD(vo0, base_, b1_, b2_, b3_, 88, 91, 92, 95, 96, 99, 100, 103, PIX_CLOB_3); } while (0)
#endif
// One step: four ring registers out of the AGPRs and straight into the accumulator, in ONE asm statement -- given the values as C++ objects hipcc
// fetched a whole slot first and, short of VGPRs, parked the surplus in a104 / a105: the pending pass's epilogue operands (the build audit
// caught it).  The first step of a slot waits for the slot's loads.
#define PIX_VFMA1(WAIT, B0, B1, B2, B3) \
  asm volatile(WAIT "v_accvgpr_read_b32 %4, a" #B0 "\n\tv_accvgpr_read_b32 %5, a" #B1 "\n\tv_accvgpr_read_b32 %6, a" #B2 "\n\tv_accvgpr_read_b32 %7, a" #B3 "\n\t" \
               "v_fmac_f32 %0, %4, %8\n\tv_fmac_f32 %1, %5, %9\n\tv_fmac_f32 %2, %6, %10\n\tv_fmac_f32 %3, %7, %11" \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(vt0), "=&v"(vt1), "=&v"(vt2), "=&v"(vt3) \
               : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]))
#define PIX_VSTEP(i, B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, B10, B11, B12, B13, B14, B15) do { \
    if ((i) == 0) PIX_VFMA1("s_waitcnt vmcnt(12)\n\t", B0, B1, B2, B3); else if ((i) == 1) PIX_VFMA1("", B4, B5, B6, B7); \
    else if ((i) == 2) PIX_VFMA1("", B8, B9, B10, B11); else PIX_VFMA1("", B12, B13, B14, B15); } while (0)
#define PIX_VFMA_STEP(r, i) do { \
    if ((r) == 0) PIX_VSTEP(i, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55); \
    else if ((r) == 1) PIX_VSTEP(i, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71); \
    else if ((r) == 2) PIX_VSTEP(i, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87); \
    else PIX_VSTEP(i, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97, 98, 99, 100, 101, 102, 103); } while (0)

// LinearAttention core (unet_model.py:169-175) of one sample with the position count N a compile-time constant: every loop over positions
// or head channels is fully unrolled, so its LDS reads are in flight together (with run-time bounds each iteration exposed a full LDS
// round trip: 9.6 us at 4x4).  q | k | v in one slab of 3 * hd channels; contexts in `scr` ([hd][33]).
template <int N>
__device__ __forceinline__ void pix_linattn(float* lds, float* scr, int tid, int in_lds, int out_lds, int heads, float scale, float* aout, int aldo, int B, int b) {
  // Register budget: this kernel must stay under 128 VGPRs -- past that hipcc spills into AGPRs, which hold the weight ring (see the
  // build audit in tests/test_host_api.py) -- so the unrolling below is bounded (groups of 4 positions / 16 channels in flight).
  const int hd = heads * 32;
  if (tid < hd) {                                        // k: softmax over the positions, per channel
    float v[N];
#pragma unroll
    for (int p = 0; p < N; ++p) v[p] = lds[in_lds + pix_at(hd + tid, p)];
    float m = v[0];
#pragma unroll
    for (int p = 1; p < N; ++p) m = fmaxf(m, v[p]);
    float z = 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) { v[p] = expf(v[p] - m); z += v[p]; }
    const float inv = 1.0f / z;
#pragma unroll
    for (int p = 0; p < N; ++p) lds[in_lds + pix_at(hd + tid, p)] = v[p] * inv;
  } else if (tid >= 256 && tid < 256 + N * heads) {      // q: softmax over the head's 32 channels, per position; scaled (two halves of 16 in registers)
    const int t = tid - 256, p = t / heads, h = t - p * heads;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 x = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p));
      m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
    }
    float z = 0.f;
#pragma unroll 2
    for (int c = 0; c < 8; ++c) {
      float4 x = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p));
      x.x = expf(x.x - m); x.y = expf(x.y - m); x.z = expf(x.z - m); x.w = expf(x.w - m);
      z += (x.x + x.y) + (x.z + x.w);
      *reinterpret_cast<float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p)) = x;
    }
    const float inv = scale / z;
#pragma unroll 2
    for (int c = 0; c < 8; ++c) {
      float4 x = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p));
      x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
      *reinterpret_cast<float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p)) = x;
    }
  }
  __syncthreads();
  // context[h][d][e] = sum_p k[p][h, d] v[p][h, e]: thread = (h, d, 8 consecutive e)
  for (int t = tid; t < heads * 32 * 4; t += PNT) {
    const int h = t >> 7, dd = (t >> 2) & 31, e0 = (t & 3) * 8;
    float cx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cx[e] = 0.f;
#pragma unroll 4
    for (int p = 0; p < N; ++p) {
      const float kv = lds[in_lds + pix_at(hd + h * 32 + dd, p)];
      const float4 v0 = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(2 * hd + h * 32 + e0, p)), v1 = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(2 * hd + h * 32 + e0 + 4, p));
      cx[0] = fmaf(kv, v0.x, cx[0]); cx[1] = fmaf(kv, v0.y, cx[1]); cx[2] = fmaf(kv, v0.z, cx[2]); cx[3] = fmaf(kv, v0.w, cx[3]);
      cx[4] = fmaf(kv, v1.x, cx[4]); cx[5] = fmaf(kv, v1.y, cx[5]); cx[6] = fmaf(kv, v1.z, cx[6]); cx[7] = fmaf(kv, v1.w, cx[7]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) scr[(h * 32 + dd) * 33 + e0 + e] = cx[e];
  }
  __syncthreads();
  // out[p][h, e] = sum_d q[p][h, d] context[h][d][e]: thread = (p, channel); the q row is read as float4s (broadcast within the head)
  for (int t = tid; t < N * hd; t += PNT) {
    const int ch = t % hd, p = t / hd, h = ch >> 5, e = ch & 31;
    float o = 0.f;
#pragma unroll 4
    for (int c = 0; c < 8; ++c) {
      const float4 qv = *reinterpret_cast<const float4*>(lds + in_lds + pix_at(h * 32 + 4 * c, p));
      o = fmaf(qv.x, scr[(h * 32 + 4 * c) * 33 + e], o); o = fmaf(qv.y, scr[(h * 32 + 4 * c + 1) * 33 + e], o);
      o = fmaf(qv.z, scr[(h * 32 + 4 * c + 2) * 33 + e], o); o = fmaf(qv.w, scr[(h * 32 + 4 * c + 3) * 33 + e], o);
    }
    lds[out_lds + pix_at(ch, p)] = o;
    if (aout) aout[((long long)p * B + b) * aldo + ch] = o;
  }
  __syncthreads();
}

}  // namespace

__global__ __launch_bounds__(PNT, 1) void pixrun_kernel(const PixOpD* __restrict__ ops, int count, const PixItem* __restrict__ items, int nitems, const PixStep* __restrict__ steps,
                                                        const unsigned short* __restrict__ rowtab, int nrowtab, const float* __restrict__ frag, const StepCtl* __restrict__ ctl, int B,
                                                        int act_floats, unsigned long long* dbg, float* __restrict__ ext_out, const int em_lds, const StepArgs em) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + act_floats;                         // [8 waves][512]: 16 rows x 32 columns, or 4 rows x 64 columns
  float* stat = red + P_RED_F;                           // [slots][8 waves][2]: per-wave partial (sum, sumsq); readers add the eight in wave order
  float* scr = stat + P_STAT_SLOTS * PNW * 2;            // scratch (LinearAttention contexts)
  unsigned short* rtab = reinterpret_cast<unsigned short*>(scr + P_SCR_F);   // [nrowtab][16]: source row of output row r in 16-float units from the slab origin (16 = the zero row)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tid0 = tid;
  const int b = blockIdx.x;
  // a run that begins with init_conv's im2col: the sample's row-packed image (written by the previous step's update kernel, so a trip to memory)
  // is requested before the LDS is cleared and lands in scr under it
  const bool stage0 = ldc(&ops[0].kind) == OP_IM2COL;
  float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (stage0) {
    const int ld = ldc(&ops[0].st_ld);
    if (tid < 2 * ld) pre0 = *reinterpret_cast<const float4*>(ldc(&ops[0].st_src) + ((long long)(tid / (ld >> 2)) * B + b) * ld + 4 * (tid % (ld >> 2)));
  }
  for (int i = tid; i < act_floats + P_RED_F; i += PNT) lds[i] = 0.f;
  for (int i = tid; i < nrowtab * 16; i += PNT) rtab[i] = rowtab[i];
  if (stage0 && tid < 2 * ldc(&ops[0].st_ld)) *reinterpret_cast<float4*>(scr + 4 * tid) = pre0;
  // the time conditions of the depth-wise ops: the step's row of the sampler's table, resolved once per launch
  const float* cond_tab = nullptr;
  if (ctl && ctl->table_mode) cond_tab = ctl->table + (long long)ctl->step * ctl->tab_ld;
  __syncthreads();
  // the eight waves' partial (sum, sumsq) of a slot are 64 contiguous bytes: four 16-byte reads and a fixed three-level tree (sixteen 4-byte reads and
  // two chains of eight dependent adds before: every reader of statistics -- each mat-vec, each folded GroupNorm -- waits for this)
  auto stat_sum = [&](int slot, float* su, float* sq) {
    const float4* sp = reinterpret_cast<const float4*>(stat + slot * PNW * 2);
    const float4 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
    *su = ((v0.x + v0.z) + (v1.x + v1.z)) + ((v2.x + v2.z) + (v3.x + v3.z));
    *sq = ((v0.y + v0.w) + (v1.y + v1.w)) + ((v2.y + v2.w) + (v3.y + v3.w));
  };
  // ---- loader state (per wave; survives the non-GEMM ops between two GEMMs) ----
  unsigned asel[PD];
  auto lane_off = [&](const PixItem* it) { return (unsigned)(((wave & ((1 << ldc(&it->gshift)) - 1)) * ldc(&it->grp_stride) + lane * 4) * 4); };   // bytes
  auto slice_steps = [&](const PixItem* it) {
    const int gs = ldc(&it->gshift), rs = ldc(&it->rshift);
    return steps + ldc(&it->step_base) + (wave >> (gs + rs)) * ldc(&it->nkp) + ((wave >> gs) & ((1 << rs) - 1)) * ldc(&it->rg_stride);
  };
  // output quads of a pass owned by this thread: (row, 4 columns) x 2
  struct Quads { int row[2], prow[2], col[2]; bool ok[2]; int total; };      // total: output quads of the pass (wave-uniform)
  auto quads_of = [&](const PixItem* it) {
    Quads q;
    const bool small = (ldc(&it->flags) & PIF_SMALL) != 0;
    const int gw = small ? 64 : 32, qshift = ldc(&it->gshift) + (small ? 4 : 3), rshift = ldc(&it->rshift);
    const int prow = ldc(&it->P_out) < (16 << rshift) ? ldc(&it->P_out) : (16 << rshift);      // rows of the pass
    const int total = prow << qshift;
    q.total = total;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + j * PNT, idc = idx < total ? idx : total - 1;
      q.ok[j] = idx < total;
      q.prow[j] = idc >> qshift;                           // row inside the pass: row group (prow >> 4), row of the unit (prow & 15)
      q.row[j] = ldc(&it->rg0) * 16 + q.prow[j];
      q.col[j] = ldc(&it->grp0) * gw + ((idc & ((1 << qshift) - 1)) << 2);
    }
    return q;
  };
  // request the epilogue operands (bias, folded-GroupNorm S and B1) of a pass into a[104 .. 127]
  auto request_epilogue_operands = [&](const PixItem* it) -> Quads {
    const Quads q = quads_of(it);
    const int npad = ldc(&it->Npad), rst = ldc(&it->rowstride);
    const float* bias = ldc(&it->bias); const float* gs = ldc(&it->gn_s); const float* gb = ldc(&it->gn_b1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (wave * 64 + j * PNT >= q.total) continue;       // no lane of this wave owns a quad j of this pass (wave-uniform: nothing to fetch)
      const int colc = q.col[j] < npad - 4 ? q.col[j] : npad - 4;
      const float* pb = bias + colc; const float* ps = gs + (long long)q.row[j] * rst + colc; const float* pg = gb + (long long)q.row[j] * rst + colc;
      if (j == 0) PIX_ELOAD(pb, ps, pg, 104, 107, 108, 111, 112, 115); else PIX_ELOAD(pb, ps, pg, 116, 119, 120, 123, 124, 127);
    }
    return q;                                             // the pass's own epilogue reuses the mapping (the index arithmetic is VALU time)
  };
  int ci = 0;
  const PixItem* pc = items;
  unsigned off_c = lane_off(pc);
  const PixStep* st_c = slice_steps(pc);
#pragma unroll
  for (int r = 0; r < PD; ++r) {
    const PixStep e = ldc_step(st_c + r);
    asel[r] = e.a_sel;
    const float* base_ = frag + e.w_off;
    PIX_LOAD_SLOT(r, base_, off_c);
  }
  Quads q_cur = request_epilogue_operands(pc);
  // {kind, passes} of every program op, packed eight ops to a scalar-cache line at the end of the step array (ops[0].pad): read from the 416-byte
  // descriptors, the first word of every op was a miss of its own at the top of the op
  const PixStep* opk = steps + ldc(&ops[0].pad);
  for (int k = 0; k < count; ++k) {
    const PixOpD& op = ops[k];
#ifdef PIX_EXP_NOOPK
    const PixStep kk = PixStep{(unsigned)ldc(&op.kind), (unsigned)ldc(&op.nitems)};
#else
    const PixStep kk = ldc_step(opk + k);
#endif
    const int kind = (int)kk.w_off;
    if (PIX_COLD(dbg) && b == 0 && tid == 0) dbg[k] = __builtin_amdgcn_s_memtime();
    // A GEMM op -- half of the program's ops -- goes straight to its code.  Through the chain of `if (kind == ...)` below it walked seven compare blocks
    // that sit KILOBYTES apart (each behind the body of the kind before it): seven instruction-cache lines nobody keeps warm, 0.43 us from the top of
    // every GEMM op to its first pass (-DPIX_STAMP_PRE, round 4).
#ifndef PIX_EXP_NOGOTO
    if (kind == OP_GEMM) goto gemm_op;
#ifndef PIX_EXP_GOTO1
    if (kind == OP_DWCONV7) goto dw_op;                   // (the other kinds likewise: all compares here, in one or two lines, one far jump each)
    if (kind == OP_LINATTN) goto attn_op;
    if (kind == OP_GN_RES) goto gnres_op;
    if (kind != PIX_STAGE && kind != OP_IM2COL) goto vec_run;
#endif
#endif
    if (kind == PIX_STAGE) {
      // (an opaque copy of the thread index: what is derived from it here stays here.  hipcc hoists every cheap tid expression of every branch
      //  in front of the op loop and keeps it for the whole kernel -- forty of them in a0 .. a39, and the ones that did not fit in scratch)
      int tid = tid0; asm volatile("" : "+v"(tid));
      const int C4 = ldc(&op.st_C) >> 2, st_P = ldc(&op.st_P), st_ld = ldc(&op.st_ld), st_lds = ldc(&op.st_lds), st_stat = ldc(&op.st_stat), st_sub = ldc(&op.st_sub);
      const float* src = ldc(&op.st_src);
      for (int idx = tid; idx < st_P * C4; idx += PNT) {
        const int c4 = idx % C4, p = idx / C4;
        const float4 v = *reinterpret_cast<const float4*>(src + ((long long)p * B + b) * st_ld + 4 * c4);
        *reinterpret_cast<float4*>(lds + st_lds + pix_at2(4 * c4, p, st_sub)) = v;
      }
      if (st_stat >= 0 && lane == 0) {
        double s = 0.0, qq = 0.0;
        if (wave == 0) { const float2* part = ldc(&op.st_part); const int np = ldc(&op.st_nparts); for (int i = 0; i < np; ++i) { const float2 v = part[(long long)i * B + b]; s += (double)v.x; qq += (double)v.y; } }
        stat[(st_stat * PNW + wave) * 2] = (float)s; stat[(st_stat * PNW + wave) * 2 + 1] = (float)qq;
      }
      __syncthreads();
      continue;
    }
    if (kind == OP_IM2COL) {
      // init_conv's input matrix: the sample's row-packed image (8 rows of st_ld floats, element (y, x, c) at x * Cpad + c) goes through scr,
      // then thread (k, pixel parity) writes column k = c * 49 + ky * 7 + kx of every other pixel: source (oy + ky - 3, ox + kx - 3) or zero
      int tid = tid0; asm volatile("" : "+v"(tid));
      const int ld = ldc(&op.st_ld), C = ldc(&op.st_C), Cpad = ldc(&op.st_P), Kld = ldc(&op.st_nparts), out_lds = ldc(&op.out_lds), out_sub = ldc(&op.out_sub);
      const float* src = ldc(&op.st_src);
      float* gout = ldc(&op.out_g) ? const_cast<float*>(reinterpret_cast<const float*>(ldc(&op.st_part))) : nullptr;
      if (k != 0 || !stage0) {                              // (the first op of a run: staged by the kernel's prologue; needs 2 ld <= 512 threads)
        for (int i = tid; i < 2 * ld; i += PNT) *reinterpret_cast<float4*>(scr + 4 * i) = *reinterpret_cast<const float4*>(src + ((long long)(i / (ld >> 2)) * B + b) * ld + 4 * (i % (ld >> 2)));
        __syncthreads();
      }
      // thread = (four consecutive columns k, output column ox = its wave): the (c, ky, kx) of its columns and their horizontal validity are fixed,
      // the loop runs over the eight output rows -- four LDS reads and one 16-byte store per pixel
      const int k4 = (tid & 63) * 4, ox = tid >> 6;
      int sofs[4], kyo[4]; bool cok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k4 + j, c = kk / 49, r = kk - 49 * c, ky = r / 7, kx = r - 7 * ky, ix = ox + kx - 3;
        cok[j] = c < C && (unsigned)ix < 8u; kyo[j] = ky - 3; sofs[j] = (ky - 3) * ld + ix * Cpad + c;
      }
      float* dp_ = lds + out_lds + (k4 >> 4) * PCS + (k4 & 15);
#pragma unroll
      for (int oy = 0; oy < 8; ++oy) {
        const int p = oy * 8 + ox;
        float4 v;
        v.x = (cok[0] && (unsigned)(oy + kyo[0]) < 8u) ? scr[sofs[0] + oy * ld] : 0.f;
        v.y = (cok[1] && (unsigned)(oy + kyo[1]) < 8u) ? scr[sofs[1] + oy * ld] : 0.f;
        v.z = (cok[2] && (unsigned)(oy + kyo[2]) < 8u) ? scr[sofs[2] + oy * ld] : 0.f;
        v.w = (cok[3] && (unsigned)(oy + kyo[3]) < 8u) ? scr[sofs[3] + oy * ld] : 0.f;
        *reinterpret_cast<float4*>(dp_ + (p >> 4) * out_sub + (p & 15) * 16) = v;
        if (gout) *reinterpret_cast<float4*>(gout + ((long long)p * B + b) * Kld + k4) = v;
      }
      __syncthreads();
      continue;
    }
  dw_op:
    if (kind == OP_DWCONV7) {
      int tid = tid0; asm volatile("" : "+v"(tid));
      const int lane = tid & 63;
      // depth-wise 7x7 over an image of at most 4x4: the window covers the whole image.  Task = (channel, output row py): its W outputs
      // need the H x (2 W - 1) weights w[qy - py + 3][d + 3], each loaded once (coalesced over the channels, all issued before the first
      // is used: dead taps read a clamped address and are masked afterwards), and the channel's H W inputs.
      const int C = ldc(&op.dw.C), H = ldc(&op.dw.H), W = ldc(&op.dw.W), ldo = ldc(&op.dw.ldo), in_lds = ldc(&op.in_lds), out_lds = ldc(&op.out_lds);
      const int in_sub = ldc(&op.in_sub), out_sub = ldc(&op.out_sub);
      const float* dw_w = ldc(&op.dw.w); const float* dw_bias = ldc(&op.dw.bias); const float* dw_cond = ldc(&op.dw.cond);
      float* dw_out = ldc(&op.out_g) ? ldc(&op.dw.out) : nullptr;
      const float* cond = dw_cond ? (cond_tab && ldc(&op.dw.ctl) ? cond_tab + ldc(&op.dw.cond_col) : dw_cond + (long long)b * ldc(&op.dw.cond_ld)) : nullptr;
      float s = 0.f, qq = 0.f;
      if (H == 8 && W == 8) {
        // 8x8: the 49 x C weights go through LDS (coalesced float4 loads, a handful of registers) -- held in registers, 49 weights per thread
        // pushed the kernel past 128 VGPRs, and hipcc then SPILLS INTO AGPRs, the weight ring's included (silent corruption; the build
        // audit in tests/test_host_api.py counts v_accvgpr instructions).  Task = (channel, output row py): 8 outputs from up to 7 rows.
        for (int i = tid; i < (49 * C) >> 2; i += PNT) *reinterpret_cast<float4*>(scr + 4 * i) = *reinterpret_cast<const float4*>(dw_w + 4 * i);
        __syncthreads();
        // a thread owns TWO neighbouring channels of an output row: inputs and weights are float2, every tap is one v_pk_fma_f32 (this
        // op is VALU-issue-bound: 392 FMAs per channel and row)
        for (int task = tid; task < (C >> 1) * 8; task += PNT) {
          const int c = (task % (C >> 1)) * 2, py = task / (C >> 1);
          sbm_f32x2 acc[8];
          sbm_f32x2 bi = *reinterpret_cast<const sbm_f32x2*>(dw_bias + c);
          if (cond) bi += *reinterpret_cast<const sbm_f32x2*>(cond + c);
#pragma unroll
          for (int px = 0; px < 8; ++px) acc[px] = (sbm_f32x2){0.f, 0.f};
#pragma unroll 1                                           // one window row at a time: unrolled, the seven rows' operands push the kernel past 128 VGPRs (AGPR spills)
          for (int ky = 0; ky < 7; ++ky) {
            const int qy = py + ky - 3;
            if (qy < 0 || qy >= 8) continue;                 // a window row outside the image (21 % of them): a wave holds two output rows, so most of these skips are wave-uniform
            sbm_f32x2 xr[8], wr[7];
#pragma unroll
            for (int qx = 0; qx < 8; ++qx) xr[qx] = *reinterpret_cast<const sbm_f32x2*>(lds + in_lds + pix_at2(c, qy * 8 + qx, in_sub));
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) wr[kx] = *reinterpret_cast<const sbm_f32x2*>(scr + (ky * 7 + kx) * C + c);
#pragma unroll
            for (int px = 0; px < 8; ++px)
#pragma unroll
              for (int kx = 0; kx < 7; ++kx) { const int qx = px + kx - 3; if (qx >= 0 && qx < 8) acc[px] = __builtin_elementwise_fma(xr[qx], wr[kx], acc[px]); }
          }
#pragma unroll
          for (int px = 0; px < 8; ++px) {
            const int p = py * 8 + px;
            const sbm_f32x2 v = acc[px] + bi;
            *reinterpret_cast<sbm_f32x2*>(lds + out_lds + pix_at2(c, p, out_sub)) = v;
            if (dw_out) *reinterpret_cast<sbm_f32x2*>(dw_out + ((long long)p * B + b) * ldo + c) = v;
            s += v.x; s += v.y; qq += v.x * v.x; qq += v.y * v.y;
          }
        }
      } else if (H == 1 && W == 1) {
        // one position: the centre tap (unet_model.py:115 at 1x1; SURVEY appendix A "degenerate sizes")
        for (int c = tid; c < C; c += PNT) {
          const float a = fmaf(lds[in_lds + pix_at(c, 0)], dw_w[24 * C + c], 0.f);
          const float v = a + (dw_bias[c] + (cond ? cond[c] : 0.f));
          lds[out_lds + pix_at(c, 0)] = v;
          if (dw_out) dw_out[(long long)b * ldo + c] = v;
          s += v; qq += v * v;
        }
      } else {
      for (int task = tid; task < C * H; task += PNT) {
        const int c = task % C, py = task / C;
        float wv[4][7], xv[4][4], acc[4];
        float bi = dw_bias[c];
        const float cv = cond ? cond[c] : 0.f;
#pragma unroll
        for (int qy = 0; qy < 4; ++qy)
#pragma unroll
          for (int dx = 0; dx < 7; ++dx) {
            const int qyc = qy < H ? qy : H - 1;
            wv[qy][dx] = dw_w[((qyc - py + 3) * 7 + dx) * C + c];
          }
#pragma unroll
        for (int qy = 0; qy < 4; ++qy)
#pragma unroll
          for (int qx = 0; qx < 4; ++qx) xv[qy][qx] = (qy < H && qx < W) ? lds[in_lds + pix_at(c, qy * W + qx)] : 0.f;      // dead inputs are zeros: their weights do not matter
        bi += cv;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          float a = 0.f;
#pragma unroll
          for (int qy = 0; qy < 4; ++qy)
#pragma unroll
            for (int qx = 0; qx < 4; ++qx) a = fmaf(xv[qy][qx], wv[qy][qx - px + 3], a);
          acc[px] = a + bi;
        }
#pragma unroll
        for (int px = 0; px < 4; ++px)
          if (px < W) {
            const int p = py * W + px;
            lds[out_lds + pix_at(c, p)] = acc[px];
            if (dw_out) dw_out[((long long)p * B + b) * ldo + c] = acc[px];
            s += acc[px]; qq += acc[px] * acc[px];
          }
      }
      }
      s = wave_sum(s); qq = wave_sum(qq);
      const int so = ldc(&op.stat_out);
      if (lane == 0 && so >= 0) { stat[(so * PNW + wave) * 2] = s; stat[(so * PNW + wave) * 2 + 1] = qq; }
      if (PIX_COLD(ldc(&op.stat_g))) {                              // the plan's layout is one partial per pixel: the whole sum goes to pixel 0 (debug sub-ranges only)
        if (lane == 0) { red[wave] = s; red[PNW + wave] = qq; }      // (not scr: at 8x8 it holds the staged weights other waves may still be reading)
        __syncthreads();
        if (tid == 0) {
          float ss = 0.f, q2 = 0.f;
          for (int w = 0; w < PNW; ++w) { ss += red[w]; q2 += red[PNW + w]; }
          float2* dst = ldc(&op.dw.stat);
          dst[b] = make_float2(ss, q2);
          for (int p = 1; p < H * W; ++p) dst[(long long)p * B + b] = make_float2(0.f, 0.f);
        }
      }
      __syncthreads();
      continue;
    }
  gnres_op:
    if (kind == OP_GN_RES) {
      int tid = tid0; asm volatile("" : "+v"(tid));
      float su, sq;
      stat_sum(ldc(&op.gn_stat), &su, &sq);
      const float inv_count = ldc(&op.gr.inv_count);
      const double mean_d = (double)su * (double)inv_count;
      double var = (double)sq * (double)inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
      const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + 1e-5));
      const int C4 = ldc(&op.gr.C) >> 2, P = ldc(&op.gr.P), ldo = ldc(&op.gr.ldo), in_lds = ldc(&op.in_lds), x_lds = ldc(&op.x_lds), out_lds = ldc(&op.out_lds);
      const int in_sub = ldc(&op.in_sub), x_sub = ldc(&op.x_sub), out_sub = ldc(&op.out_sub);
      const float* gamma = ldc(&op.gr.gamma); const float* beta = ldc(&op.gr.beta);
      float* gout = ldc(&op.out_g) ? ldc(&op.gr.out) : nullptr;
      for (int idx = tid; idx < P * C4; idx += PNT) {
        const int c4 = idx % C4, p = idx / C4;
        const float4 u = *reinterpret_cast<const float4*>(lds + in_lds + pix_at2(4 * c4, p, in_sub)), x = *reinterpret_cast<const float4*>(lds + x_lds + pix_at2(4 * c4, p, x_sub));
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * c4), be = *reinterpret_cast<const float4*>(beta + 4 * c4);
        float4 y;
        y.x = ((u.x - mean) * rstd) * ga.x + be.x + x.x; y.y = ((u.y - mean) * rstd) * ga.y + be.y + x.y;
        y.z = ((u.z - mean) * rstd) * ga.z + be.z + x.z; y.w = ((u.w - mean) * rstd) * ga.w + be.w + x.w;
        *reinterpret_cast<float4*>(lds + out_lds + pix_at2(4 * c4, p, out_sub)) = y;
        if (gout) *reinterpret_cast<float4*>(gout + ((long long)p * B + b) * ldo + 4 * c4) = y;
      }
      __syncthreads();
      continue;
    }
  attn_op:
    if (kind == OP_LINATTN) {
      int tid = tid0; asm volatile("" : "+v"(tid));
      // LinearAttention core (unet_model.py:169-175) of one sample: n <= 16 positions, heads x 32 channels; q | k | v in one slab
      const int n = ldc(&op.at.n), heads = ldc(&op.at.heads), hd = heads * 32, in_lds = ldc(&op.in_lds), out_lds = ldc(&op.out_lds), aldo = ldc(&op.at.ldo);
      const float scale = ldc(&op.at.scale);
      float* aout = ldc(&op.out_g) ? ldc(&op.at.out) : nullptr;
      if (n == 16) pix_linattn<16>(lds, scr, tid, in_lds, out_lds, heads, scale, aout, aldo, B, b);
      else pix_linattn<4>(lds, scr, tid, in_lds, out_lds, heads, scale, aout, aldo, B, b);
      continue;
    }
  vec_run:
    if (kind == PIX_VEC || kind == PIX_DW1 || kind == PIX_GNR1) {
      // A RUN of one-position ops (the 1x1 level and the strided convolution that enters it), v_run program ops long.  On the MFMA path such an
      // op costs 4.4 us around a 1.3 us K loop -- partial tiles through LDS, two barriers, a 512-thread epilogue for 128 .. 256 outputs.  Two
      // things make ops this small slow, and the run is built around both (round 4, tools/vec_stamps.py):
      //  * a global load nobody requested ahead costs 1 .. 3 us (every workgroup asks the L2 for the same lines at the same moment).  So the
      //    run's weights are ONE STREAM -- the 8 KB steps of all its mat-vecs back to back, each op padded to 16 steps -- pulled through the
      //    AGPR ring 16 steps ahead ACROSS op boundaries, and every op's per-thread operands (bias / folded-GroupNorm tables, depth-wise centre
      //    taps, GroupNorm affine) are requested while the op before it runs;
      //  * an op's descriptor read when the op starts is a chain of scalar-cache misses (kind -> branch -> shapes -> pointers: ~1.5 us of
      //    the 4 us such an op took).  So every op of a run has a 64-byte RECORD with all it needs (PixVecRec, host-built, contiguous), read
      //    with one scalar load a whole op ahead.
      // A mat-vec: thread = (column, K slice), 4 FMAs per 16-byte weight load, K slices are adjacent lanes and meet by DPP, the owner applies
      // the epilogue and stores -- one barrier per op.
      const int run = ldc(&op.v_run);
      int tid = tid0; asm volatile("" : "+v"(tid));
      const int lane = tid & 63;
      // weight stream of the run: slots of four steps, table entries of 8 words {weight offset, weight offset of the slot 4 further on, -, -,
      // four activation selectors}
      int nload = ldc(&op.v_slots);                         // slots still to be requested
      const int* sc = reinterpret_cast<const int*>(steps + ldc(&op.v_stream));     // next slot to consume
      const int* recs = reinterpret_cast<const int*>(steps + ldc(&op.v_recs));
      const unsigned vo0 = (unsigned)tid * 16u;
      if (nload > 0) {                                      // (a run without a mat-vec leaves the ring -- the next MFMA pass's first steps -- alone)
#pragma unroll
        for (int r = 0; r < PD; ++r) { const float* base_ = frag + (unsigned)ldc(sc + 8 * r); PIX_VLOAD_SLOT(r, base_); }
        nload -= PD;
      }
      float pe0 = 0.f, pe1 = 0.f, pe2 = 0.f;
      // the per-thread operands of the op whose record is R_, requested one op ahead (pointers and shapes come out of the record: no chasing)
#define PIX_RPTR(R_, i_) reinterpret_cast<const float*>(((unsigned long long)(unsigned)(R_)[i_]) | ((unsigned long long)(unsigned)(R_)[(i_) + 1] << 32))
#define PIX_REQUEST_SMALL(R_) do { \
        const int kd__ = (R_)[0] & 0xff; \
        if (kd__ == (PIX_VEC & 0xff)) { \
          const int c0__ = tid >> (((R_)[0] >> 8) & 0xf), c__ = c0__ < (R_)[9] ? c0__ : (R_)[9] - 1; \
          pe0 = ldg(PIX_RPTR(R_, 10), c__); pe1 = ldg(PIX_RPTR(R_, 12), c__); pe2 = ldg(PIX_RPTR(R_, 14), c__); \
        } else if (kd__ == (PIX_DW1 & 0xff)) { \
          const int c__ = tid < (R_)[8] ? tid : (R_)[8] - 1; \
          const float* dwc__ = PIX_RPTR(R_, 14); \
          const float* cond__ = dwc__ ? (cond_tab && (((R_)[0] >> 19) & 1) ? cond_tab + (R_)[9] : dwc__ + (long long)b * (R_)[1]) : PIX_RPTR(R_, 12); \
          pe0 = ldg(PIX_RPTR(R_, 10), c__); pe1 = ldg(PIX_RPTR(R_, 12), c__); pe2 = ldg(cond__, c__); \
          if (!dwc__) pe2 = 0.f; \
        } else { \
          const int c__ = tid < (R_)[8] ? tid : (R_)[8] - 1; \
          pe0 = ldg(PIX_RPTR(R_, 10), c__); pe1 = ldg(PIX_RPTR(R_, 12), c__); \
        } } while (0)
      // (The records are NOT carried from one op to the next as 16-register vectors: hipcc 7.2 kept such a loop-carried tuple in s[16:31] and
      // the loop counter in s20 -- lane 4, the GroupNorm 1 / count -- at the same time.  Each op reads its own record, and TOUCHES the next
      // one's line a whole op ahead, so that read is a scalar-cache hit.)
      { const i32x16p r0 = ldc_x16(recs); PIX_REQUEST_SMALL(r0); }
      int touch = 0;
      for (int j = 0; j < run; ++j) {
        const i32x16p rc = ldc_x16(recs + 16 * j);
        if (touch == 0x7f5a5a5a) asm volatile("s_nop 0");     // (consumes the previous op's touch: never true)
        touch = ldc(recs + 16 * (j + 1 < run ? j + 1 : j));      // first touch of the NEXT op's record line, a whole op ahead
        const int kd = rc[0] & 0xff;
        if (PIX_COLD(dbg) && b == 0 && tid == 0 && j > 0) dbg[k + j] = __builtin_amdgcn_s_memtime();
#ifdef PIX_STAMP_VEC
#define VSTAMP(ph) do { if (PIX_COLD(dbg) && b == 0 && tid == 0 && k + j < 96) dbg[128 + (k + j) * 4 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VSTAMP(ph) do { } while (0)
#endif
        const int out_lds = rc[5], so = rc[3];
        const bool slow = ((rc[0] >> 18) & 1) != 0;          // results / statistics also leave the launch: those pointers stay in the big descriptor
        float ov = 0.f;                                     // this thread's stored output (0 when it owns none): the op's statistics
        // The op's per-thread operands (requested while the previous op ran) are taken HERE, before this op's slot loop issues ring loads the
        // compiler cannot see: vector-memory loads complete in order, and a compiler-placed wait for these three values behind sixteen newer
        // ring loads waited for the NEXT op's weights (1 us per op in the first version, tools/vec_stamps.py).
        float qe0 = pe0, qe1 = pe1, qe2 = pe2;
#ifndef PIX_EXP_NOPIN
        asm volatile("" : "+v"(qe0), "+v"(qe1), "+v"(qe2));
#endif
        VSTAMP(0);
        if (kd == (PIX_VEC & 0xff)) {
          const int sshift = (rc[0] >> 8) & 0xf, S = 1 << sshift, slice = tid & (S - 1), col = tid >> sshift;
          const int aoff = ((slice * 4) >> 4) * PCS + ((slice * 4) & 15);
          const bool gn = ((rc[0] >> 16) & 1) != 0;
          float mean = 0.f, rstd = 1.f, inv_r = 1.f;
          if (gn) {
            float su, sq;
            stat_sum(rc[2], &su, &sq);
            const float ginv = __builtin_bit_cast(float, rc[4]);
            const double mean_d = (double)su * (double)ginv;
            double var = (double)sq * (double)ginv - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
            mean = (float)mean_d; rstd = rsqrtf((float)var + 1e-5f); inv_r = sqrtf((float)var + 1e-5f);
          }
          f32x4p acc = {0.f, 0.f, 0.f, 0.f};
          // one slot: four steps out of the ring, then the slot's refill four slots further down the stream -- or, once the run's stream is
          // exhausted, with the next MFMA pass's first steps, which the ring held when the run began.  The activations are requested in front
          // of the wait.  Source-2 steps (bit 31: the fused 1x1 residual conv, which enters unnormalised) are scaled by 1 / rstd so that the
          // common GroupNorm scaling cancels, as on the MFMA path.
          VSTAMP(1);
          i32x8p e_cur = ldc_x8(sc);
          auto slots4 = [&](auto sc_tag) __attribute__((always_inline)) {
            constexpr bool SC = decltype(sc_tag)::value;
#pragma unroll
            for (int r = 0; r < PD; ++r) {
              const i32x8p e = e_cur;
              e_cur = ldc_x8(sc + 8);                        // the NEXT slot's entry: a table read asked for when it is needed costs a scalar-cache miss per slot
              f32x4p a[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4p*>(__builtin_assume_aligned(lds + ((unsigned)e[4 + i] & 0x7fffffffu) + aoff, 16));
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                f32x4p av = a[i];
                if (SC) { const float f = e[4 + i] < 0 ? inv_r : 1.0f; av *= f; }
                float vt0, vt1, vt2, vt3;
                PIX_VFMA_STEP(r, i);
              }
              sc += 8;
              if (nload > 0) { const float* base_ = frag + (unsigned)e[1]; PIX_VLOAD_SLOT(r, base_); --nload; }
              else { const PixStep pe_ = ldc_step(st_c + r); const float* base_ = frag + pe_.w_off; PIX_LOAD_SLOT(r, base_, off_c); }
            }
          };
          const int ns = rc[1];                             // slots of this op, a multiple of 4
#if defined(PIX_EXP_RECDBG)
          if (false) for (int si = 0; si < ns; si += PD) slots4(std::true_type{});
#elif defined(PIX_EXP_RECDBG2)
          inv_r = 1.0f;
          if (gn && ((rc[0] >> 17) & 1)) for (int si = 0; si < ns; si += PD) slots4(std::true_type{});
#else
          if (gn && ((rc[0] >> 17) & 1)) for (int si = 0; si < ns; si += PD) slots4(std::true_type{});
#endif
          else for (int si = 0; si < ns; si += PD) slots4(std::false_type{});
          VSTAMP(2);
          float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
          if (S >= 2) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
          if (S >= 4) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
          if (S >= 8) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, true));   // row_half_mirror
          const int nstore = rc[8], res_lds = rc[6];
#ifdef PIX_EXP_DUMP
          if (dbg && b == 0 && tid == 0 && gn && dbg[439] == 0) {
            dbg[439] = 1 + k + j;
            float su_, sq_; stat_sum(rc[2], &su_, &sq_);
            const float vals[8] = {su_, sq_, mean, rstd, inv_r, t, qe0, qe1};
            for (int q = 0; q < 8; ++q) dbg[440 + q] = (unsigned long long)__builtin_bit_cast(unsigned, vals[q]);
            dbg[448] = (unsigned long long)(unsigned)rc[2]; dbg[449] = (unsigned long long)(unsigned)rc[4]; dbg[450] = (unsigned long long)(unsigned)qe2;
          }
#endif
          if (gn) t = t * rstd + (qe2 - (mean * rstd) * qe1);
          t += qe0;
          const bool own = slice == 0 && col < nstore;
          const int colc = col < nstore ? col : 0;
          if (res_lds >= 0) t += lds[res_lds + (colc >> 4) * PCS + (colc & 15)];
          const float ot = ((rc[0] >> 12) & 0xf) == ACT_GELU ? 0.5f * t * (1.0f + sbm_erf(t * 0.70710678118654752440f)) : t;      // (eligibility: GELU or none)
          if (own) {
            lds[out_lds + pix_at(col, 0)] = ot;
            ov = ot;
          }
          if (PIX_COLD(slow) && own) { const PixOpD& o = ops[k + j]; if (ldc(&o.out_g)) ldc(&o.out_g)[(long long)b * ldc(&o.v_ldo) + col] = ot; }
        } else if (kd == (PIX_DW1 & 0xff)) {
          // depth-wise 7x7 at one position = its centre tap (unet_model.py:115 at 1x1; SURVEY appendix A "degenerate sizes")
          if (tid < rc[8]) {
            const float a = fmaf(lds[rc[6] + pix_at(tid, 0)], qe0, 0.f);
            ov = a + (qe1 + qe2);
            lds[out_lds + pix_at(tid, 0)] = ov;
            if (PIX_COLD(slow)) { const PixOpD& o = ops[k + j]; if (ldc(&o.out_g)) ldc(&o.dw.out)[(long long)b * ldc(&o.dw.ldo) + tid] = ov; }
          }
        } else {
          // GroupNorm(1, C)(u) + x at one position (LinearAttention tail + Residual, unet_model.py:173-175, 21-27)
          float su, sq;
          stat_sum(rc[2], &su, &sq);
          const float inv_count = __builtin_bit_cast(float, rc[4]);
          const double mean_d = (double)su * (double)inv_count;
          double var = (double)sq * (double)inv_count - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
          const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + 1e-5));
          if (tid < rc[8]) {
            const float u = lds[rc[6] + pix_at(tid, 0)], x = lds[rc[7] + pix_at(tid, 0)];
            const float y = ((u - mean) * rstd) * qe0 + qe1 + x;
            lds[out_lds + pix_at(tid, 0)] = y;
            if (PIX_COLD(slow)) { const PixOpD& o = ops[k + j]; if (ldc(&o.out_g)) ldc(&o.gr.out)[(long long)b * ldc(&o.gr.ldo) + tid] = y; }
          }
        }
        if (j + 1 < run) { const i32x16p rn = ldc_x16(recs + 16 * (j + 1)); PIX_REQUEST_SMALL(rn); }
        VSTAMP(3);
        if (so >= 0) {
          const float sacc = wave_sum(ov), qacc = wave_sum(ov * ov);
          if (lane == 0) { stat[(so * PNW + wave) * 2] = sacc; stat[(so * PNW + wave) * 2 + 1] = qacc; }
        }
        __syncthreads();
        if (PIX_COLD(slow)) {                                         // the plan's partial layouts (debug sub-ranges only): one (sum, sumsq) per 32-column block of a
          const PixOpD& o = ops[k + j];                     // GEMM's output, one per pixel for the depth-wise op
          if (ldc(&o.stat_g) && tid == 0) {
            float su, sq;
            stat_sum(so, &su, &sq);
            float2* dst = kd == (PIX_VEC & 0xff) ? ldc(&o.stat_g) : ldc(&o.dw.stat);
            if (kd == (PIX_VEC & 0xff)) {
              const int nstore = rc[8], nblk = (nstore + 31) >> 5;
              for (int bk = 0; bk < nblk; ++bk) {
                float s32 = 0.f, q32 = 0.f;
                for (int c = bk * 32; c < bk * 32 + 32 && c < nstore; ++c) { const float v = lds[out_lds + pix_at(c, 0)]; s32 += v; q32 += v * v; }
                dst[(long long)bk * B + b] = make_float2(s32, q32);
              }
            } else {
              dst[b] = make_float2(su, sq);
            }
          }
        }
      }
      k += run - 1;
      continue;
    }
    // ---------------------------------------------------------------- GEMM: everything comes from the item table
  gemm_op:
#define PSTAMP_AT(ph) do { if (PIX_COLD(dbg) && b == 0 && tid == 0 && ci < 96) dbg[128 + ci * 4 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
    // default: four phase stamps per pass (K loop | partial tiles + barrier | epilogue + barrier).  -DPIX_STAMP_EPI / -DPIX_STAMP_PRE (diagnostic
    // builds, tools/build_variant.sh) move the four stamps INSIDE the epilogue / the pass prologue instead: PSTAMP_X(id)
#if defined(PIX_STAMP_EPI)
#define PSTAMP(ph) do { if ((ph) == 2) PSTAMP_AT(0); } while (0)
#define PSTAMP_X(id) do { if ((id) >= 10 && (id) <= 12) PSTAMP_AT((id) - 9); } while (0)
#elif defined(PIX_STAMP_PRE)
#define PSTAMP(ph) do { if ((ph) == 0) PSTAMP_AT(3); } while (0)
#define PSTAMP_X(id) do { if ((id) >= 0 && (id) <= 2) PSTAMP_AT(id); } while (0)
#else
#define PSTAMP(ph) PSTAMP_AT(ph)
#define PSTAMP_X(id) do { } while (0)
#endif
    const int g_nitems = (int)kk.a_sel;
    for (int it = 0; it < g_nitems; ++it) {
      PSTAMP_X(0);
      const int ni = ci + 1 < nitems ? ci + 1 : 0;
      const PixItem* pn = items + ni;
      const unsigned off_n = lane_off(pn);
      const PixStep* st_n = slice_steps(pn);
      PSTAMP_X(1);
      const int flags = ldc(&pc->flags);
      const bool small = (flags & PIF_SMALL) != 0, gn = (flags & PIF_GN) != 0, scale2 = (flags & PIF_SCALE2) != 0;
      // The folded GroupNorm's mean / rstd are needed in the EPILOGUE only (the K loop runs on raw activations) -- except by the few ops with a
      // fused residual conv, whose source 2 is scaled by 1 / rstd inside the loop.  Computed here by all 512 threads they cost every pass
      // 0.3 us of dependent LDS reads and fp64 arithmetic in front of the K loop (round 4, -DPIX_STAMP_PRE); wave 0 -- the oldest wave, which
      // wins the matrix pipe from its SIMD partner and reaches the barrier behind the K loop first -- computes them there instead, while it
      // would wait, and hands them over through LDS.
      float mean = 0.f, rstd = 1.f, inv_r = 1.f;
      auto gn_moments = [&]() {
        float su, sq;
        stat_sum(ldc(&pc->gn_stat), &su, &sq);
        const float ginv = ldc(&pc->gn_inv);
        const double mean_d = (double)su * (double)ginv;
        double var = (double)sq * (double)ginv - mean_d * mean_d; var = var < 0.0 ? 0.0 : var;
        mean = (float)mean_d; rstd = rsqrtf((float)var + 1e-5f); inv_r = sqrtf((float)var + 1e-5f);
      };
      if (gn && scale2) gn_moments();
      PSTAMP_X(2);
      const int g_stat_out = ldc(&pc->stat_out);
      if ((flags & PIF_FIRST) && g_stat_out >= 0 && lane == 0) { stat[(g_stat_out * PNW + wave) * 2] = 0.f; stat[(g_stat_out * PNW + wave) * 2 + 1] = 0.f; }
      PSTAMP(0);
      f32x4p acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int nu = ldc(&pc->nkp) / PD;
      if (!small) {
        // Both waves of a SIMD run this loop and share its matrix pipe; they fall into lock-step (same code, fair arbitration), so whatever a
        // wave does BETWEEN two MFMA blocks is pipe idle time for both (round 4: the loop ran at 80 % of the MFMA issue rate with or without its
        // weight loads, tools/build_variant.sh noload).  Hence every LDS read is REQUESTED in front of a block and CONSUMED behind it -- the A
        // operands of step i + 1, the row selector of step i + 2 -- and the gap holds only the ring loads and a few address operations; the
        // source-2 scaling (GroupNorm-folded ops with a fused residual conv) is a separate instantiation.
        const int g = lane >> 4;
        const int n = (((wave >> ldc(&pc->gshift)) & ((1 << ldc(&pc->rshift)) - 1)) + ldc(&pc->rg0)) * 16 + (lane & 15);     // this lane's output row: row group of the wave's unit x 16 + lane
        auto kloop16 = [&](auto sc_tag) {
          constexpr bool SC = decltype(sc_tag)::value;
          int rs = rtab[((asel[0] >> 16) & 0x7fff) * 16 + n];
          const float* ap0 = lds + (asel[0] & 0xffff) + rs * 16 + 4 * g;
          f32x4p a0 = *reinterpret_cast<const f32x4p*>(ap0), a1 = *reinterpret_cast<const f32x4p*>(ap0 + PCS);
          if (SC && (asel[0] >> 31)) { a0 *= inv_r; a1 *= inv_r; }
          rs = rtab[((asel[1] >> 16) & 0x7fff) * 16 + n];
          const float* apn = lds + (asel[1] & 0xffff) + rs * 16 + 4 * g;      // A operands of the NEXT step
          for (int u = 0; u < nu; ++u) {
            const bool last = u == nu - 1;
            const PixStep* lst = last ? st_n : st_c + (u + 1) * PD;
            const unsigned loff = last ? off_n : off_c;
#pragma unroll
            for (int r = 0; r < PD; ++r) {
              // requested here, used behind the MFMA block: next step's A operands, the row selector of the step after it, this slot's next table entry
              const f32x4p n0 = *reinterpret_cast<const f32x4p*>(apn), n1 = *reinterpret_cast<const f32x4p*>(apn + PCS);
              const unsigned an = asel[(r + 1) % PD], an2 = asel[(r + 2) % PD];
              const int rs2 = rtab[((an2 >> 16) & 0x7fff) * 16 + n];
              const PixStep e = ldc_step(lst + r);
              PIX_MFMA16_SLOT(r, a0, a1);
              asel[r] = e.a_sel;
              const float* base_ = frag + e.w_off;
              PIX_KLOAD_SLOT(r, base_, loff);
              apn = lds + (an2 & 0xffff) + rs2 * 16 + 4 * g;
              if (SC) { const float f = (an >> 31) ? inv_r : 1.0f; a0 = n0 * f; a1 = n1 * f; }
              else { a0 = n0; a1 = n1; }
            }
          }
        };
        if (scale2) kloop16(std::true_type{}); else kloop16(std::false_type{});
      } else {
        const int ri = lane & 3;
        int rs = rtab[((asel[0] >> 16) & 0x7fff) * 16 + ri];
        const float* ap0 = lds + (asel[0] & 0xffff) + rs * 16;
        f32x4p q0 = *reinterpret_cast<const f32x4p*>(ap0), q1 = *reinterpret_cast<const f32x4p*>(ap0 + 4), q2 = *reinterpret_cast<const f32x4p*>(ap0 + 8), q3 = *reinterpret_cast<const f32x4p*>(ap0 + 12);
        if (scale2 && (asel[0] >> 31)) { q0 *= inv_r; q1 *= inv_r; q2 *= inv_r; q3 *= inv_r; }
        rs = rtab[((asel[1] >> 16) & 0x7fff) * 16 + ri];
        for (int u = 0; u < nu; ++u) {
          const bool last = u == nu - 1;
          const PixStep* lst = last ? st_n : st_c + (u + 1) * PD;
          const unsigned loff = last ? off_n : off_c;
#pragma unroll
          for (int r = 0; r < PD; ++r) {
            const unsigned an = asel[(r + 1) % PD];
            const float* apn = lds + (an & 0xffff) + rs * 16;
            f32x4p n0 = *reinterpret_cast<const f32x4p*>(apn), n1 = *reinterpret_cast<const f32x4p*>(apn + 4), n2 = *reinterpret_cast<const f32x4p*>(apn + 8), n3 = *reinterpret_cast<const f32x4p*>(apn + 12);
            if (scale2 && (an >> 31)) { n0 *= inv_r; n1 *= inv_r; n2 *= inv_r; n3 *= inv_r; }
            const PixStep e = ldc_step(lst + r);
            PIX_MFMA4_SLOT(r);
            asel[r] = e.a_sel;
            rs = rtab[((asel[(r + 2) % PD] >> 16) & 0x7fff) * 16 + ri];
            const float* base_ = frag + e.w_off;
            PIX_KLOAD_SLOT(r, base_, loff);
            q0 = n0; q1 = n1; q2 = n2; q3 = n3;
          }
        }
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));      // MFMA results -> VALU / LDS readers (hipcc pads nothing behind an asm MFMA)
      PSTAMP(1);
      // partial tiles -> LDS.  16-row path: lane (n, g) holds rows 4 g + v of columns n (acc0) and 16 + n (acc1); 4-row path: row v of column `lane`
      if (!small) {
        const int n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) { red[wave * 512 + (4 * g + v) * 32 + n] = acc0[v]; red[wave * 512 + (4 * g + v) * 32 + 16 + n] = acc1[v]; }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave * 512 + v * 64 + lane] = acc0[v] + acc1[v];
      }
      if (gn && !scale2 && wave == 0) { gn_moments(); if (lane == 0) { scr[0] = mean; scr[1] = rstd; } }      // (scr: nothing else uses it during a GEMM pass)
      __syncthreads();
      PSTAMP(2);
      if (gn && !scale2) { mean = scr[0]; rstd = scr[1]; }
      // epilogue: operands out of the AGPRs (everything but the 16 loads just issued for the next pass has landed)
      {
        f32x4p eb[2], es[2], eg[2];
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        PIX_EREAD4(eb[0], 104, 105, 106, 107); PIX_EREAD4(es[0], 108, 109, 110, 111); PIX_EREAD4(eg[0], 112, 113, 114, 115);
        PIX_EREAD4(eb[1], 116, 117, 118, 119); PIX_EREAD4(es[1], 120, 121, 122, 123); PIX_EREAD4(eg[1], 124, 125, 126, 127);
        const Quads q = q_cur;
        q_cur = request_epilogue_operands(pn);             // a[104 .. 127] are free again: the next pass's operands fly under this epilogue and its K loop
        PSTAMP_X(10);
        const int gw = small ? 64 : 32, gshift = ldc(&pc->gshift), ushift = gshift + ldc(&pc->rshift), ksplit = 1 << ldc(&pc->kshift), grp0 = ldc(&pc->grp0);
        const int out_sub = ldc(&pc->out_sub), res_sub = ldc(&pc->res_sub), P_out = ldc(&pc->P_out);
        const int nstore = ldc(&pc->nstore), act = ldc(&pc->act), out_lds = ldc(&pc->out_lds), res_lds = ldc(&pc->res_lds), ldo = ldc(&pc->ldo);
        const int res_row[4] = {ldc(&pc->res_row[0]), ldc(&pc->res_row[1]), ldc(&pc->res_row[2]), ldc(&pc->res_row[3])};
        float* out_g = ldc(&pc->out_g); float2* stat_g = ldc(&pc->stat_g);
        float sacc = 0.f, qacc = 0.f;
        const int perm_tab = ldc(&pc->perm_tab);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // the epilogue is VALU-issue-bound (two waves per SIMD, four cycles per instruction): a wave none of whose lanes owns a quad j of
          // this pass -- every pass of fewer than 1024 / 512 output quads -- skips the whole block instead of executing it masked
          if (wave * 64 + j * PNT >= q.total) continue;
          const int row = perm_tab >= 0 ? (int)rtab[perm_tab * 16 + q.row[j]] : q.row[j];      // (phase-grouped passes: pass row -> output pixel)
          const int col = q.col[j];
          const int cw = col - grp0 * gw, grp = small ? cw >> 6 : cw >> 5, sub = cw & (gw - 1);
          const int unit = ((q.prow[j] >> 4) << gshift) | grp;       // the wave (of K slice 0) that owns this output: [column group | row group]
          f32x4p t = {0.f, 0.f, 0.f, 0.f};
          for (int sl = 0; sl < ksplit; ++sl) t += *reinterpret_cast<const f32x4p*>(red + ((sl << ushift) | unit) * 512 + (q.prow[j] & 15) * gw + sub);
          const bool ok = q.ok[j] && col < nstore;
          f32x4p o = {0.f, 0.f, 0.f, 0.f};
          if (ok) {
            if (gn) t = t * rstd + (eg[j] - (mean * rstd) * es[j]);
            t += eb[j];
            if (res_lds >= 0) t += *reinterpret_cast<const f32x4p*>(lds + res_lds + pix_at2(col, P_out > 16 ? row : sbyte(res_row, row & 15), res_sub));
            if (act == ACT_GELU) {                        // the one activation of the ConvNeXt blocks: packed, the switch outside the element loop
              const sbm_f32x2 g0 = sbm_gelu2((sbm_f32x2){t[0], t[1]}), g1 = sbm_gelu2((sbm_f32x2){t[2], t[3]});
              o[0] = g0.x; o[1] = g0.y; o[2] = g1.x; o[3] = g1.y;
            } else {                                       // ACT_NONE: the only other activation a Unet op has (pix_gemm_supported)
              o = t;
            }
            if (out_lds >= 0) *reinterpret_cast<f32x4p*>(lds + out_lds + pix_at2(col, row, out_sub)) = o;
            if (out_g) *reinterpret_cast<f32x4p*>(out_g + ((long long)row * B + b) * ldo + col) = o;
            if (flags & PIF_NCHW) {                       // the plan's external output, (B, nstore, P_out): the last conv of a score net
#pragma unroll
              for (int i = 0; i < 4; ++i) if (col + i < nstore) ext_out[((long long)b * nstore + col + i) * P_out + row] = o[i];
            }
            sacc += o[0] + o[1] + o[2] + o[3]; qacc += o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
          }
          if (PIX_COLD(stat_g)) {                                    // the plan's partial layout: one (sum, sumsq) per (pixel, 32-column block) = 8 consecutive threads
            float s32 = o[0] + o[1] + o[2] + o[3], q32 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { s32 += __shfl_xor(s32, m); q32 += __shfl_xor(q32, m); }
            const int nbk = col >> 5, nblk = (nstore + 31) >> 5;
            if ((tid & 7) == 0 && q.ok[j] && nbk < nblk) stat_g[((long long)row * nblk + nbk) * B + b] = make_float2(s32, q32);
          }
        }
        PSTAMP_X(11);
        if (g_stat_out >= 0) {                             // this wave's share of the output statistics, added to its own partial (passes of one op accumulate)
          sacc = wave_sum(sacc); qacc = wave_sum(qacc);
          if (lane == 0) { stat[(g_stat_out * PNW + wave) * 2] += sacc; stat[(g_stat_out * PNW + wave) * 2 + 1] += qacc; }
        }
      }
      PSTAMP_X(12);
      __syncthreads();                                    // red is reused; the output slab and the statistics are complete
      PSTAMP(3);
      pc = pn; off_c = off_n; st_c = st_n; ci = ni;
    }
  }
  // the loads issued for the wrapped-around "next" pass must land before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: PIX_CLOB_0, PIX_CLOB_1, PIX_CLOB_2, PIX_CLOB_3, PIX_CLOB_E);
  if (em_lds >= 0) {
    // The sampling loop's EM predictor (sampler.hip em_step_kernel, the same device functions) applied by the kernel that ends the score net: the
    // last conv left the score of this sample in an LDS slab ([64 pixels][16]); thread q < CHW / 4 takes element quad q = (channel, row, four
    // columns) of the NCHW state, as a thread of the update kernel would.  The last workgroup to finish moves the step counter on.
    int tid = tid0; asm volatile("" : "+v"(tid));
    const int stepi = em.step_counter ? *em.step_counter : em.step_base;
    if (tid < (em.CHW >> 2)) {
      const int e = tid * 4, c = e >> 6, r = e & 63;
      const float t = em.t ? em.t[b] : em.timesteps[stepi];
      const Coef cf = sde_coef(em.sde, t);
      const float4 x = *reinterpret_cast<const float4*>(em.x + (long long)b * em.CHW + e);
      const float* sp = lds + em_lds + (r >> 4) * (2 * PCS) + (c >> 4) * PCS + (r & 15) * 16 + (c & 15);      // pixels r .. r + 3: one 16-row group
      const float4 s = make_float4(sp[0], sp[16], sp[32], sp[48]);
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!em.sde.probability_flow) z = load_noise4(em, b, e, (unsigned)stepi);
      float4 m, o;
      em_update4(em, cf, x, s, z, &m, &o);
      *reinterpret_cast<float4*>(em.mean_out + (long long)b * em.CHW + e) = m;
      *reinterpret_cast<float4*>(em.x_out + (long long)b * em.CHW + e) = o;
      if (em.pbc_out) store_pbc(em, b, e, o);
    }
    if (em.adv_counter) advance_by_last_workgroup(em, stepi);
  }
  if (PIX_COLD(dbg) && b == 0 && tid == 0) dbg[count] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------------------------------------
// host: fragment-order weights (once per model), programs of op runs (per run, on demand), launch
// ------------------------------------------------------------------------------------------------
namespace {

struct PSlot { long long base; int ld; int P; int lds; int t0, t1; };      // tensors of 64 pixels are four 16-row sub-slabs, chunks_of(ld) chunks each

void set_sbyte(int* w, int r, int v) { w[r >> 2] = (w[r >> 2] & ~(0xff << (8 * (r & 3)))) | ((v & 0xff) << (8 * (r & 3))); }

struct PFragOp {                          // fragment-order copy of one GEMM op's weights
  bool small = false;
  long long off = 0;                      // float offset in the fragment arena
  int F = 0;                              // steps per column group: live taps x K steps (+ source 2)
  int groups = 0;                         // column groups (32 or 64 columns)
  std::vector<int> taps;                  // live taps (ky * KW + kx), in step order
  int ndc = 0, ndc2 = 0;                  // K steps per tap / of source 2
  bool vec = false; int sshift = 0;       // PIX_VEC packing: [step][thread = column << sshift | slice][4], F = steps of 8 KB (ndc per tap, ndc2 of source 2)
};

struct PProgram { PixOpD* d_ops = nullptr; PixItem* d_items = nullptr; PixStep* d_steps = nullptr; unsigned short* d_rowtab = nullptr;
                  int n_ops = 0, n_items = 0, n_rowtab = 0, act_floats = 0; bool needs_ext_out = false; int em_lds = -1; std::vector<int> kinds; };

}  // namespace

struct PixState {
  float* d_frag = nullptr; size_t frag_floats = 0;
  std::map<int, PFragOp> fops;            // by plan op index
  std::map<long long, PProgram> programs; // by first * 4096 + count
  std::mutex mu;                          // programs are built on demand (debug sub-ranges); one handle may be driven from several host threads in the tests
  const float* zero4 = nullptr;           // zeros, as many as the widest op has columns (tail of the fragment arena)
  long long zero_blk = 0;                 // float offset of an 8 KB block of zeros (padding steps of the one-position runs' weight stream)
};

namespace {

bool pix_tap_live(const GemmOp& g, int ky, int kx) {
  const int P_out = g.Hout * g.Wout;
  for (int r = 0; r < P_out; ++r)
    if (tap_source_pixel(r / g.Wout, r % g.Wout, ky, kx, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W) >= 0) return true;
  return false;
}

const char* pix_gemm_supported(const GemmOp& g) {
  const int P_out = g.Hout * g.Wout, P_in = g.a.H * g.a.W;
  if ((P_out > 16 && P_out != 64) || (P_in > 16 && P_in != 64) || g.a_ext || (g.out_nchw && (g.crop_y || g.crop_x || g.Hn != g.Hout || g.Wn != g.Wout || g.stat_off >= 0)) || g.w_pix_stride || g.bias_pix_stride || g.a_bcast || g.a_rows) return "pixel kernel: unsupported GEMM op";
  if (P_out == 64 && (g.up2 != 1 || (g.has_res && g.r.H * g.r.W != 64) || (g.has2 && g.a2.H * g.a2.W != 64))) return "pixel kernel: a 64-pixel output takes its residual / source 2 at its own resolution only";
  if ((g.Kp & 31) || (g.has2 && (g.Kp2 & 31)) || (g.Npad & 31) || (!g.out_nchw && (g.out.C & 3))) return "pixel kernel: K and N must be multiples of 32";
  if (g.act != ACT_NONE && g.act != ACT_GELU) return "pixel kernel: activations other than GELU are not built in (no Unet op has one)";
  if (g.has2 && g.a2.H * g.a2.W > 16 && g.a2.H * g.a2.W != 64) return "pixel kernel: source 2 has an unsupported pixel count";
  return nullptr;
}

// [tap][Npad][Kp] (+ [Npad][Kp2]) -> fragment order.  16-row path: block (pair tp, step f) = [half h][tile t][lane][4] =
// W[n = 32 tp + 16 t + (lane & 15)][k = 32 dc + 16 h + 4 (lane >> 4) + i]; 4-row path: block (tile tl, step f) = [k group j][lane][4] =
// W[n = 64 tl + lane][k = 16 dc + 4 j + i].  Only taps that reach at least one output pixel are packed.
void pack_fragments(const float* prm, const GemmOp& g, PFragOp* fo, std::vector<float>* arena) {
  const int P_out = g.Hout * g.Wout;
  fo->small = P_out <= 4;
  const int kstep = fo->small ? 16 : 32, gw = fo->small ? 64 : 32;
  fo->ndc = g.Kp / kstep; fo->ndc2 = g.has2 ? g.Kp2 / kstep : 0;
  fo->taps.clear();
  for (int ky = 0; ky < g.KH; ++ky) for (int kx = 0; kx < g.KW; ++kx) if (pix_tap_live(g, ky, kx)) fo->taps.push_back(ky * g.KW + kx);
  fo->F = (int)fo->taps.size() * fo->ndc + fo->ndc2;
  fo->groups = (g.Npad + gw - 1) / gw;
  fo->off = (long long)arena->size();
  arena->resize(arena->size() + (size_t)fo->groups * fo->F * 1024, 0.f);
  float* dst = arena->data() + fo->off;
  for (int gi = 0; gi < fo->groups; ++gi)
    for (int f = 0; f < fo->F; ++f) {
      const bool s2 = f >= (int)fo->taps.size() * fo->ndc;
      const int tap = s2 ? 0 : fo->taps[f / fo->ndc], dc = s2 ? f - (int)fo->taps.size() * fo->ndc : f % fo->ndc;
      const float* W = s2 ? prm + g.w2_off : prm + g.w_off + (long long)tap * g.Npad * g.Kp;
      const int Kp = s2 ? g.Kp2 : g.Kp;
      float* blk = dst + ((size_t)gi * fo->F + f) * 1024;
      if (!fo->small) {
        for (int h = 0; h < 2; ++h) for (int t = 0; t < 2; ++t) for (int lane = 0; lane < 64; ++lane) {
          const int n = 32 * gi + 16 * t + (lane & 15), k = 32 * dc + 16 * h + 4 * (lane >> 4);
          if (n < g.Npad) std::memcpy(blk + ((h * 2 + t) * 64 + lane) * 4, W + (long long)n * Kp + k, 16);
        }
      } else {
        for (int j = 0; j < 4; ++j) for (int lane = 0; lane < 64; ++lane) {
          const int n = 64 * gi + lane, k = 16 * dc + 4 * j;
          if (n < g.Npad) std::memcpy(blk + (j * 64 + lane) * 4, W + (long long)n * Kp + k, 16);
        }
      }
    }
}

// One output pixel, a power-of-two width that fills the 512 threads with at most 8 K slices per column: the vector-ALU mat-vec (PIX_VEC).
bool pix_phase_on() { const char* e = std::getenv("SBM_PIX_PHASE"); return !(e && std::atoi(e) == 0); }   // phase-grouped transposed convolutions (read at model creation)
bool pix_vec_on() { const char* e = std::getenv("SBM_PIX_VEC"); return !(e && std::atoi(e) == 0); }      // read at every model creation (the tests switch it)
bool pix_vec_eligible(const GemmOp& g) {
  if (!pix_vec_on() || g.Hout * g.Wout != 1 || g.a.H * g.a.W > 16 || g.out_nchw) return false;
  if (g.Npad != 64 && g.Npad != 128 && g.Npad != 256 && g.Npad != 512) return false;
  if (g.act != ACT_NONE && g.act != ACT_GELU) return false;
  if (g.has2 && g.a2.H * g.a2.W != 1) return false;
  if (g.has_res && g.r.H * g.r.W != 1) return false;
  return true;
}
// [tap][Npad][Kp] (+ [Npad][Kp2]) -> [step][thread][4]: step = (live tap, 4 S consecutive k), thread = (column n, slice s) holds W[n][k0 + 4 s .. + 3]
void pack_vec(const float* prm, const GemmOp& g, PFragOp* fo, std::vector<float>* arena) {
  fo->vec = true; fo->small = true;
  int sh = 0; while ((g.Npad << sh) < PNT) ++sh;
  fo->sshift = sh;
  const int S = 1 << sh, kstep = 4 * S;
  fo->ndc = g.Kp / kstep; fo->ndc2 = g.has2 ? g.Kp2 / kstep : 0;
  fo->taps.clear();
  for (int ky = 0; ky < g.KH; ++ky) for (int kx = 0; kx < g.KW; ++kx) if (pix_tap_live(g, ky, kx)) fo->taps.push_back(ky * g.KW + kx);
  fo->F = (int)fo->taps.size() * fo->ndc + fo->ndc2;
  fo->groups = 1;
  fo->off = (long long)arena->size();
  arena->resize(arena->size() + (size_t)((fo->F + 15) / 16 * 16) * PNT * 4, 0.f);      // padded with zero steps to whole rounds of the ring (4 slots of 4 steps)
  float* dst = arena->data() + fo->off;
  for (int f = 0; f < fo->F; ++f) {
    const bool s2 = f >= (int)fo->taps.size() * fo->ndc;
    const int tap = s2 ? 0 : fo->taps[f / fo->ndc], dc = s2 ? f - (int)fo->taps.size() * fo->ndc : f % fo->ndc;
    const float* W = s2 ? prm + g.w2_off : prm + g.w_off + (long long)tap * g.Npad * g.Kp;
    const int Kp = s2 ? g.Kp2 : g.Kp;
    for (int t = 0; t < PNT; ++t) {
      const int n = t >> sh, k = dc * kstep + 4 * (t & (S - 1));
      std::memcpy(dst + ((size_t)f * PNT + t) * 4, W + (long long)n * Kp + k, 16);
    }
  }
}

struct PBuilder {
  const Plan& plan; const DevPlan& dp; const PixState& st; int first, count;
  std::vector<PSlot> slots; int act_chunks = 0;
  std::vector<PixOpD> ops; std::vector<PixItem> items; std::vector<PixStep> steps; std::vector<unsigned short> rowtab;
  std::map<long long, int> stat_of;
  std::map<int, int> produced;
  const std::vector<PSlot>* preset = nullptr;
  std::vector<int> touched;
  const char* err = nullptr;
  bool needs_ext_out = false; int em_lds = -1;
  int run_first = -1, run_stream = -1, run_slots = 0;   // one-position run being assembled: its first program op, the step index of its slot table, its slots

  static bool run_kind(int kind) { return kind == PIX_VEC || kind == PIX_DW1 || kind == PIX_GNR1; }
  void flush_run() {
    if (run_first < 0) return;
    ops[run_first].v_stream = run_stream >= 0 ? run_stream : 0;
    ops[run_first].v_slots = run_slots;
    // word 1 of a slot's entry: the weight offset of the slot FOUR further down the stream -- what the consumer of this slot requests next
    for (int sidx = 0; sidx + PD < run_slots; ++sidx) steps[run_stream + 4 * sidx].a_sel = steps[run_stream + 4 * (sidx + PD)].w_off;
    // the run's records: 16 words per program op, everything the op's hot path needs (the kernel reads record j + 1 while op j runs)
    ops[run_first].v_recs = (int)steps.size();
    const int nrun = ops[run_first].v_run;
    for (int j = 0; j < nrun; ++j) {
      const PixOpD& d = ops[run_first + j];
      unsigned w[16] = {0};
      auto put_ptr = [&](int i, const void* ptr) { const unsigned long long v = (unsigned long long)ptr; w[i] = (unsigned)v; w[i + 1] = (unsigned)(v >> 32); };
      auto fbits = [](float f) { unsigned u; std::memcpy(&u, &f, 4); return u; };
      const unsigned slow = (d.out_g || d.stat_g) ? 1u : 0u;
      w[2] = (unsigned)d.gn_stat; w[3] = (unsigned)d.stat_out; w[5] = (unsigned)d.out_lds;
      if (d.kind == PIX_VEC) {
        w[0] = (unsigned)(PIX_VEC & 0xff) | ((unsigned)d.v_sshift << 8) | ((unsigned)d.v_act << 12) | ((unsigned)(d.v_gn ? 1 : 0) << 16) | ((unsigned)(d.v_n2 > 0 ? 1 : 0) << 17) | (slow << 18);
        w[1] = (unsigned)d.v_n1; w[4] = fbits(d.v_gn_inv); w[6] = (unsigned)d.v_res_lds; w[8] = (unsigned)d.v_nstore; w[9] = (unsigned)d.v_Npad;
        put_ptr(10, d.v_bias); put_ptr(12, d.v_gn_s); put_ptr(14, d.v_gn_b1);
      } else if (d.kind == PIX_DW1) {
        w[0] = (unsigned)(PIX_DW1 & 0xff) | (slow << 18) | ((unsigned)(d.dw.ctl ? 1 : 0) << 19);
        w[1] = (unsigned)d.dw.cond_ld; w[6] = (unsigned)d.in_lds; w[8] = (unsigned)d.dw.C; w[9] = (unsigned)d.dw.cond_col;
        put_ptr(10, d.dw.w + 24 * (long long)d.dw.C); put_ptr(12, d.dw.bias); put_ptr(14, d.dw.cond);
      } else {
        w[0] = (unsigned)(PIX_GNR1 & 0xff) | (slow << 18);
        w[4] = fbits(d.gr.inv_count); w[6] = (unsigned)d.in_lds; w[7] = (unsigned)d.x_lds; w[8] = (unsigned)d.gr.C;
        put_ptr(10, d.gr.gamma); put_ptr(12, d.gr.beta);
      }
      if (preset && std::getenv("SBM_PIX_DUMP_RECS"))
        fprintf(stderr, "rec run@%d op %d kind %d: w0 %08x slots/ld %d gn_stat %d stat_out %d ginv %08x out_lds %d w6 %d w7 %d w8 %d w9 %d | v_n1 %d v_n2 %d step_base %d\n", run_first, j, d.kind, w[0], (int)w[1],
                (int)w[2], (int)w[3], w[4], (int)w[5], (int)w[6], (int)w[7], (int)w[8], (int)w[9], d.v_n1, d.v_n2, d.v_step_base);
      for (int q = 0; q < 8; ++q) steps.push_back(PixStep{w[2 * q], w[2 * q + 1]});
    }
    run_first = -1; run_stream = -1; run_slots = 0;
  }
  void push_op(const PixOpD& d) {
    if (run_kind(d.kind)) {
      if (run_first < 0) { run_first = (int)ops.size(); ops.push_back(d); ops.back().v_run = 1; }
      else { ops.push_back(d); ops[run_first].v_run += 1; }
      if (d.kind == PIX_VEC) { if (run_stream < 0) run_stream = d.v_step_base; run_slots += d.v_n1; }
    } else {
      flush_run();
      ops.push_back(d);
    }
  }

  static int chunks_of(int ld) { return (ld + 15) / 16 + 1; }     // + one chunk of slack: K is walked to the weights' padded depth
  static int nsub_of(int P) { return P > 16 ? 4 : 1; }
  static int slab_chunks(const PSlot& s) { return nsub_of(s.P) * chunks_of(s.ld); }
  static int sub_floats(int ld) { return chunks_of(ld) * PCS; }     // floats between the sub-slabs of a 64-pixel tensor
  // LDS float offset of (row 0, channel 0 of the view)
  int slab(const View& v, int* slot_index) {
    const int P = v.H * v.W;
    for (size_t i = 0; i < slots.size(); ++i) {
      PSlot& s = slots[i];
      if (s.ld == v.ld && s.P == P && v.off >= s.base && v.off < s.base + s.ld) {
        if (slot_index) *slot_index = (int)i;
        touched.push_back((int)i);
        if ((v.off - s.base) & 15) err = "pixel kernel: a view does not start on a 16-channel boundary";
        return s.lds + (int)((v.off - s.base) >> 4) * PCS;
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
      else if (o.kind == OP_IM2COL) { look(o.ic.out); }
    }
    PSlot s{base, v.ld, P, act_chunks * PCS, (int)ops.size(), (int)ops.size()};
    if (v.ld & 15) err = "pixel kernel: a tensor's channel pitch is not a multiple of 16";
    if (P > 16 && P != 64) err = "pixel kernel: tensors of more than 16 pixels must be 8x8";
    if (preset) {
      for (const PSlot& ps : *preset) if (ps.base == base && ps.ld == v.ld && ps.P == P) s.lds = ps.lds;
    } else {
      act_chunks += slab_chunks(s);
    }
    slots.push_back(s);
    touched.push_back((int)slots.size() - 1);
    if (slot_index) *slot_index = (int)slots.size() - 1;
    if ((v.off - s.base) & 15) err = "pixel kernel: a view does not start on a 16-channel boundary";
    return s.lds + (int)((v.off - s.base) >> 4) * PCS;
  }
  PixOpD stage_op(const PSlot& s) {
    PixOpD d{}; d.kind = PIX_STAGE;
    d.st_src = dp.ws + s.base; d.st_ld = s.ld; d.st_C = s.ld; d.st_P = s.P; d.st_lds = s.lds; d.st_stat = -1; d.st_sub = sub_floats(s.ld);
    d.gn_stat = d.stat_out = -1;
    return d;
  }
  // An op that writes only PART of a buffer (one half of a skip concatenation) while the rest was produced outside the run: the whole
  // buffer is staged first, so that later readers of the full width find the other half.
  // The views op `o` reads / writes, for rest_filled_in_run
  static void op_views(const Op& o, std::vector<const View*>* rd, std::vector<const View*>* wr) {
    if (o.kind == OP_GEMM) { rd->push_back(&o.g.a); if (o.g.has2) rd->push_back(&o.g.a2); if (o.g.has_res) rd->push_back(&o.g.r); wr->push_back(&o.g.out); }
    else if (o.kind == OP_DWCONV7) { rd->push_back(&o.dw.in); wr->push_back(&o.dw.out); }
    else if (o.kind == OP_GN_RES) { rd->push_back(&o.gr.u); rd->push_back(&o.gr.x); wr->push_back(&o.gr.out); }
    else if (o.kind == OP_LINATTN || o.kind == OP_ATTN) { rd->push_back(&o.at.qkv); wr->push_back(&o.at.out); }
    else if (o.kind == OP_IM2COL) { wr->push_back(&o.ic.out); }      // (its input is the row-packed tensor in memory: never a slab)
  }
  // Op `cur` writes the channels [v.off, v.off + v.C) of slot si and nothing of the slot has been produced yet.  True when every channel of
  // the slot OUTSIDE that range is written by a later op of this run before any op of the run reads it (the two halves of a skip
  // concatenation: the down path stores its half, the up path's producer fills the other half before the concatenation is read) -- then
  // the buffer need not be staged from the workspace first.
  bool rest_filled_in_run(int si, const View& v, int cur) const {
    const PSlot& sl = slots[si];
    std::vector<char> have(sl.ld, 0);
    auto inside = [&](const View& w) { return w.ld == sl.ld && w.H * w.W == sl.P && w.off >= sl.base && w.off < sl.base + sl.ld; };
    for (int c = 0; c < v.C; ++c) have[(int)(v.off - sl.base) + c] = 1;
    for (int j = cur + 1; j < first + count; ++j) {
      std::vector<const View*> rd, wr;
      op_views(plan.ops[j], &rd, &wr);
      for (const View* r : rd) if (inside(*r)) for (int c = 0; c < r->C; ++c) if (!have[(int)(r->off - sl.base) + c]) return false;
      for (const View* w : wr) if (inside(*w)) for (int c = 0; c < w->C; ++c) have[(int)(w->off - sl.base) + c] = 1;
    }
    return true;     // nothing in the run reads the rest before it is written (readers outside the run take the workspace copy, which every writer keeps current)
  }
  void mark_produced(const View& v, int cur = -1) {
    int si; slab(v, &si);
    if (!produced.count(si) && (v.off != slots[si].base || v.C < slots[si].ld) && !(cur >= 0 && rest_filled_in_run(si, v, cur))) push_op(stage_op(slots[si]));
    produced[si] = 1;
  }
  int stat_slot(long long off) {
    auto it = stat_of.find(off);
    if (it != stat_of.end()) return it->second;
    const int s = (int)stat_of.size();
    stat_of[off] = s;
    return s;
  }
  int need(const View& v, long long stat_off = -1, int nparts = 0) {
    int si; const int off = slab(v, &si);
    const bool have_stat = stat_off < 0 || stat_of.count(stat_off);
    if (produced.count(si) && have_stat) return off;
    PixOpD d = stage_op(slots[si]);
    if (produced.count(si)) d.st_P = 0;
    if (stat_off >= 0 && !stat_of.count(stat_off)) {
      d.st_stat = stat_slot(stat_off); d.st_part = reinterpret_cast<const float2*>(dp.ws + stat_off); d.st_nparts = nparts;
    }
    push_op(d);
    produced[si] = 1;
    return off;
  }
  // a row table: source pixel q of output row r (-1: none) -> its row in 16-float units from the source slab's origin (sub-slab q >> 4, row
  // q & 15 of a 64-pixel source); 16 = the zero row.  `count` = 16 or 64 output rows; returns the table's index in units of 16 entries.
  int add_rowtab(const int* rows, int count, int src_ld) {
    const int idx = (int)rowtab.size() / 16, units = chunks_of(src_ld) * 17;
    for (int r = 0; r < count; ++r) rowtab.push_back((unsigned short)(rows[r] < 0 ? 16 : (rows[r] >> 4) * units + (rows[r] & 15)));
    return idx;
  }
};

size_t pix_lds_bytes(int act_floats, int nrowtab) { return sizeof(float) * ((size_t)act_floats + P_RED_F + P_STAT_SLOTS * PNW * 2 + P_SCR_F) + (size_t)nrowtab * 32 + 16; }

const char* pixrun_build(const DevPlan& dp, PixState& st, int first, int count, PProgram* out) {
  const Plan& plan = *dp.plan;
  auto fill = [&](PBuilder& b) -> const char* {
    b.rowtab.assign(64, 16);                              // table 0 (64 rows): every row -> the zero row (no-op steps)
    for (int i = first; i < first + count; ++i) {
      const Op& op = plan.ops[i];
      b.touched.clear();
      PixOpD d{};
      d.kind = op.kind; d.gn_stat = -1; d.stat_out = -1; d.out_lds = -1; d.in_lds = -1;
      size_t item_first = b.items.size();
      if (op.kind == OP_DWCONV7) {
        const DwOp& o = op.dw;
        if (((o.in.H > 4 || o.in.W > 4) && !(o.in.H == 8 && o.in.W == 8 && 49 * o.in.C <= P_SCR_F && (o.in.C & 3) == 0)) || o.stat_slabs > 1) return "pixel kernel: depth-wise op outside the supported shapes";
        d.in_lds = b.need(o.in); d.in_sub = PBuilder::sub_floats(o.in.ld);
        d.out_lds = b.slab(o.out, nullptr); b.mark_produced(o.out, i); d.out_sub = PBuilder::sub_floats(o.out.ld);
        d.dw = make_dw_args(plan, o, dp.prm, dp.ws, dp.ctl);
        if (o.stat_off >= 0) d.stat_out = b.stat_slot(o.stat_off);
        if (pix_vec_on() && o.in.H == 1 && o.in.W == 1 && o.in.C <= PNT) d.kind = PIX_DW1;
      } else if (op.kind == OP_GN_RES) {
        const GnResOp& o = op.gr;
        if ((o.u.H * o.u.W > 16 && o.u.H * o.u.W != 64) || o.out_ext || (o.u.C & 3)) return "pixel kernel: unsupported GroupNorm+residual op";
        d.in_lds = b.need(o.u, o.part_off, o.nparts); d.in_sub = PBuilder::sub_floats(o.u.ld);
        d.x_lds = b.need(o.x); d.x_sub = PBuilder::sub_floats(o.x.ld);
        d.gn_stat = b.stat_slot(o.part_off);
        d.out_lds = b.slab(o.out, nullptr); b.mark_produced(o.out, i); d.out_sub = PBuilder::sub_floats(o.out.ld);
        d.gr = make_gnres_args(o, dp.prm, dp.ws);
        if (pix_vec_on() && o.u.H * o.u.W == 1 && o.u.C <= PNT) d.kind = PIX_GNR1;
      } else if (op.kind == OP_LINATTN) {
        const AttnOp& o = op.at;
        { const int np = o.qkv.H * o.qkv.W;
          if ((np != 16 && np != 4) || o.dh != 32 || o.heads * 32 > 256 || o.heads * 32 * 33 > P_SCR_F || np * o.heads > 256) return "pixel kernel: unsupported attention op"; }
        d.in_lds = b.need(o.qkv);
        d.out_lds = b.slab(o.out, nullptr); b.mark_produced(o.out, i);
        d.at = make_attn_args(o, dp.ws);
      } else if (op.kind == OP_IM2COL) {
        const Im2colOp& o = op.ic;
        if (o.H != 8 || o.W != 8 || o.KH != 7 || o.KW != 7 || o.pad != 3 || o.H * o.in.ld > P_SCR_F || 2 * o.in.ld > PNT || (o.in.ld & 3) || o.C * 49 > o.out.ld || o.out.ld != 256) return "pixel kernel: im2col outside the supported shapes";
        d.st_src = dp.ws + o.in.off; d.st_ld = o.in.ld; d.st_C = o.C; d.st_P = o.Cpad; d.st_nparts = o.out.ld;
        d.out_lds = b.slab(o.out, nullptr); b.mark_produced(o.out, i); d.out_sub = PBuilder::sub_floats(o.out.ld);
        d.st_lds = d.out_lds; d.st_stat = o.out.ld;            // (st_stat: the pitch of the workspace copy)
        d.st_part = reinterpret_cast<const float2*>(dp.ws + o.out.off);      // (st_part: the workspace copy itself, written when out_g is set)
      } else if (op.kind == OP_GEMM) {
        const GemmOp& g = op.g;
        if (const char* e = pix_gemm_supported(g)) return e;
        auto fit = st.fops.find(i);
        if (fit == st.fops.end()) return "pixel kernel: no fragment-order weights for this op";
        const PFragOp& fo = fit->second;
        const int P_out = g.Hout * g.Wout;
        const int a_lds = b.need(g.a, g.gn ? g.gn_part_off : -1, g.gn_nparts);
        int a2_lds = -1;
        if (g.has2) a2_lds = b.need(g.a2);
        if (fo.vec) {                                        // one output pixel: the vector-ALU mat-vec
          d.kind = PIX_VEC;
          d.v_res_lds = -1;
          if (g.has_res) d.v_res_lds = b.need(g.r);
          d.v_sshift = fo.sshift; d.v_Npad = g.Npad; d.v_nstore = g.out.C; d.v_act = g.act; d.v_ldo = g.out.ld;
          d.v_bias = g.bias_off >= 0 ? dp.prm + g.bias_off : st.zero4; d.v_gn_s = d.v_gn_b1 = st.zero4;
          if (g.gn) { d.v_gn = 1; d.gn_stat = b.stat_slot(g.gn_part_off); d.v_gn_inv = g.gn_inv_count; d.v_gn_s = dp.prm + g.gn_s_off; d.v_gn_b1 = dp.prm + g.gn_b1_off; }
          d.out_g = dp.ws + g.out.off;
          d.out_lds = b.slab(g.out, nullptr); b.mark_produced(g.out, i);
          if (g.stat_off >= 0) { d.stat_out = b.stat_slot(g.stat_off); d.stat_g = reinterpret_cast<float2*>(dp.ws + g.stat_off); }
          const int S = 1 << fo.sshift, kstep = 4 * S, n1 = (int)fo.taps.size() * fo.ndc;
          d.v_step_base = (int)b.steps.size(); d.v_n2 = fo.ndc2;
          std::vector<unsigned> asels;                       // the op's activation selectors in step order, padded to whole rounds of the ring
          for (int f = 0; f < fo.F; ++f) {
            const bool s2 = f >= n1;
            const int ti = s2 ? 0 : f / fo.ndc, dc = s2 ? f - n1 : f % fo.ndc, k0 = dc * kstep;
            int q = 0;
            if (!s2) { const int tp = fo.taps[ti]; q = tap_source_pixel(0, 0, tp / g.KW, tp % g.KW, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W); if (q < 0) return "pixel kernel: a packed tap of a one-pixel op is dead"; }
            asels.push_back((unsigned)((s2 ? a2_lds : a_lds) + q * 16 + (k0 >> 4) * PCS + (k0 & 15)) | (s2 && g.gn ? 0x80000000u : 0u));
          }
          const int padded = (fo.F + 15) / 16 * 16;
          while ((int)asels.size() < padded) asels.push_back((unsigned)a_lds);       // zero weights: any readable activation
          for (int s0 = 0; s0 < padded; s0 += 4) {           // one slot = 8 words = 4 table entries: {weight offset, -} {-, -} {sel 0, sel 1} {sel 2, sel 3}
            b.steps.push_back(PixStep{(unsigned)(fo.off + (long long)s0 * PNT * 4), 0u});
            b.steps.push_back(PixStep{0u, 0u});
            b.steps.push_back(PixStep{asels[s0], asels[s0 + 1]});
            b.steps.push_back(PixStep{asels[s0 + 2], asels[s0 + 3]});
          }
          d.v_n1 = padded / 4;
          item_first = b.items.size();
        } else {
        PixItem proto{};                                    // the fields every pass of this op shares
        proto.res_lds = -1;
        if (g.has_res) { proto.res_lds = b.need(g.r); proto.res_sub = PBuilder::sub_floats(g.r.ld); }
        proto.out_sub = PBuilder::sub_floats(g.out.ld);
        proto.P_out = P_out; proto.Npad = g.Npad; proto.nstore = g.out.C; proto.act = g.act; proto.ldo = g.out.ld;
        proto.bias = g.bias_off >= 0 ? dp.prm + g.bias_off : st.zero4;
        proto.gn_s = proto.gn_b1 = st.zero4; proto.gn_stat = -1; proto.stat_out = -1;
        if (g.gn) { proto.gn_stat = b.stat_slot(g.gn_part_off); proto.gn_inv = g.gn_inv_count; proto.gn_s = dp.prm + g.gn_s_off; proto.gn_b1 = dp.prm + g.gn_b1_off; proto.rowstride = g.Npad; proto.flags |= PIF_GN; }
        if (g.gn && g.has2) proto.flags |= PIF_SCALE2;
        if (fo.small) proto.flags |= PIF_SMALL;
        if (g.out_nchw) {                                   // no slab, no workspace copy: the pass stores into the external tensor (kernel argument)
          proto.flags |= PIF_NCHW; proto.nstore = g.N; proto.out_g = nullptr; proto.out_lds = -1; proto.ldo = 0; b.needs_ext_out = true;
          if (P_out == 64 && g.N <= 16 && i == (int)plan.ops.size() - 1) {      // ... and, for the sampler's fused EM update (kernel tail), a [64][16] slab of its own
            View sv; sv.off = (1LL << 40) + i; sv.ld = 16; sv.C = 16; sv.H = g.Hout; sv.W = g.Wout;
            proto.out_lds = b.slab(sv, nullptr); proto.out_sub = PBuilder::sub_floats(16); b.em_lds = proto.out_lds;
          }
        } else {
        proto.out_g = dp.ws + g.out.off;
        proto.out_lds = b.slab(g.out, nullptr); b.mark_produced(g.out, i);
        }
        d.out_g = proto.out_g;
        if (g.stat_off >= 0) { proto.stat_out = b.stat_slot(g.stat_off); proto.stat_g = reinterpret_cast<float2*>(dp.ws + g.stat_off); d.stat_g = proto.stat_g; }
        // row tables: one per live tap (+ source 2), 16 or 64 output rows
        const int w2 = g.has2 ? g.a2.W : (g.has_res ? g.r.W : 1);
        const int nrows = P_out > 16 ? 64 : 16;
        // A stride-2 transposed convolution onto 8x8: an output pixel of phase (oy & 1, ox & 1) has 4 of the 16 taps, so with rows in raster order
        // every tap multiplies 48 zero rows of 64 (ups.2.3: 21 us for 2.1 MFLOP of live work).  PHASE-GROUPED: the pass's four row groups are the
        // four phases (16 pixels each), every row group walks its OWN step list of 4 taps, and the epilogue maps pass rows back to pixels.
        const bool phased = pix_phase_on() && g.transposed && g.stride == 2 && P_out == 64 && (g.Wout & 1) == 0 && !g.has2 && !g.has_res && !g.gn && fo.groups * 4 == PNW;
        int perm[64];
        for (int r = 0; r < 64; ++r) perm[r] = r;
        if (phased) {
          const int hw = g.Wout / 2;
          for (int r = 0; r < 64; ++r) { const int ph = r >> 4, i = r & 15; perm[r] = (2 * (i / hw) + (ph >> 1)) * g.Wout + 2 * (i % hw) + (ph & 1); }
        }
        std::vector<int> tab_of;
        for (int tp : fo.taps) {
          int rows[64];
          for (int r = 0; r < nrows; ++r) { const int px = perm[r]; rows[r] = px < P_out ? tap_source_pixel(px / g.Wout, px % g.Wout, tp / g.KW, tp % g.KW, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W) : -1; }
          tab_of.push_back(b.add_rowtab(rows, nrows, g.a.ld));
        }
        proto.perm_tab = -1;
        if (phased) { proto.perm_tab = (int)b.rowtab.size() / 16; for (int r = 0; r < 64; ++r) b.rowtab.push_back((unsigned short)perm[r]); }
        int tab2 = 0;
        {
          int rows[64];
          for (int r = 0; r < nrows; ++r) rows[r] = r < P_out ? ((r / g.Wout) / g.up2) * w2 + ((r % g.Wout) / g.up2) : -1;
          if (g.has2) tab2 = b.add_rowtab(rows, nrows, g.a2.ld);
          for (int r = 0; r < 16; ++r) set_sbyte(proto.res_row, r, rows[r] < 0 ? 0 : (rows[r] & 0x7f));
        }
        if (b.rowtab.size() / 16 > 0x7fff) return "pixel kernel: too many row tables";
        // passes: up to 8 units = column groups x row groups of 16 (64-row outputs: the row groups of a column group share its weight stream),
        // the rest of the 8 waves split K
        const int kchunk = fo.small ? 1 : 2;               // LDS chunks per K step
        const int nrg = P_out > 16 ? 4 : 1;
        item_first = b.items.size();
        if (phased) {
          // one pass: fo.groups column groups x 4 phases = 8 waves, no K split; phase ph walks the taps that reach one of its 16 pixels
          std::vector<std::vector<int>> ptaps(4);
          for (int ph = 0; ph < 4; ++ph)
            for (size_t ti = 0; ti < fo.taps.size(); ++ti) {
              bool live = false;
              for (int i = 0; i < 16 && !live; ++i) { const int px = perm[ph * 16 + i]; live = tap_source_pixel(px / g.Wout, px % g.Wout, fo.taps[ti] / g.KW, fo.taps[ti] % g.KW, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W) >= 0; }
              if (live) ptaps[ph].push_back((int)ti);
            }
          size_t mx = 0; for (auto& v : ptaps) mx = std::max(mx, v.size());
          const int nk = (int)mx * fo.ndc, nkp = (nk + PD - 1) / PD * PD;
          PixItem it = proto;
          it.step_base = (int)b.steps.size(); it.nkp = nkp; it.grp0 = 0; it.grp_stride = fo.F * 1024; it.rg0 = 0; it.rg_stride = nkp;
          int sh = 0; while ((1 << sh) < fo.groups) ++sh; it.gshift = sh;
          it.rshift = 2; it.kshift = 0;
          for (int ph = 0; ph < 4; ++ph)
            for (int j = 0; j < nkp; ++j) {
              PixStep e;
              if (j < (int)ptaps[ph].size() * fo.ndc) {
                const int ti = ptaps[ph][j / fo.ndc], dc = j % fo.ndc;
                const long long aoff = a_lds + (long long)dc * kchunk * PCS;
                if (b.preset && aoff > 0xffff) return "pixel kernel: LDS offset out of the step table's range";
                e.w_off = (unsigned)(fo.off + ((long long)ti * fo.ndc + dc) * 1024);
                e.a_sel = (unsigned)(aoff & 0xffff) | ((unsigned)tab_of[ti] << 16);
              } else {
                e.w_off = (unsigned)fo.off;
                e.a_sel = (unsigned)(a_lds & 0xffff);       // row table 0: every row reads zeros
              }
              b.steps.push_back(e);
            }
          b.items.push_back(it);
        } else
        for (int g0 = 0; g0 < fo.groups;) {
          int grp = 1; while (grp * 2 <= PNW && grp * 2 <= fo.groups - g0) grp *= 2;
          int rgs = 1; while (rgs * 2 <= nrg && grp * rgs * 2 <= PNW) rgs *= 2;
          const int ksplit = PNW / (grp * rgs);
          const int nk = (fo.F + ksplit - 1) / ksplit, nkp = (nk + PD - 1) / PD * PD;
          for (int rg0 = 0; rg0 < nrg; rg0 += rgs) {
            PixItem it = proto;
            it.step_base = (int)b.steps.size(); it.nkp = nkp; it.grp0 = g0; it.grp_stride = fo.F * 1024; it.rg0 = rg0;
            int sh = 0; while ((1 << sh) < grp) ++sh; it.gshift = sh;
            sh = 0; while ((1 << sh) < rgs) ++sh; it.rshift = sh;
            sh = 0; while ((1 << sh) < ksplit) ++sh; it.kshift = sh;
            for (int sl = 0; sl < ksplit; ++sl)
              for (int j = 0; j < nkp; ++j) {
                const int f = sl + j * ksplit;
                PixStep e;
                if (f < fo.F) {
                  const bool s2 = f >= (int)fo.taps.size() * fo.ndc;
                  const int ti = s2 ? 0 : f / fo.ndc, dc = s2 ? f - (int)fo.taps.size() * fo.ndc : f % fo.ndc;
                  const long long aoff = (s2 ? a2_lds : a_lds) + (long long)dc * kchunk * PCS;
                  if (b.preset && aoff > 0xffff) return "pixel kernel: LDS offset out of the step table's range";      // (the first pass lays the slabs out without reuse)
                  e.w_off = (unsigned)(fo.off + ((long long)g0 * fo.F + f) * 1024);
                  e.a_sel = (unsigned)(aoff & 0xffff) | ((unsigned)(s2 ? tab2 : tab_of[ti]) << 16) | (s2 ? 0x80000000u : 0u);
                } else {
                  e.w_off = (unsigned)(fo.off + (long long)g0 * fo.F * 1024);
                  e.a_sel = (unsigned)(a_lds & 0xffff);       // row table 0: every row reads zeros
                }
                b.steps.push_back(e);
              }
            b.items.push_back(it);
          }
          g0 += grp;
        }
        d.nitems = (int)b.items.size() - (int)item_first;
        b.items[item_first].flags |= PIF_FIRST; b.items.back().flags |= PIF_LAST;
        }
      } else {
        return "pixel kernel: op kind not supported";
      }
      // Results leave the launch (workspace store) only when something OUTSIDE the run reads them: a later op of the plan, or nothing
      // in this run at all (a debug sub-range).  Statistics likewise.
      {
        auto reads = [&](const Op& o, const View& v) {
          auto hit = [&](const View& w) { return w.ld == v.ld && w.H * w.W == v.H * v.W && w.off - v.off < v.ld && v.off - w.off < v.ld; };
          if (o.kind == OP_GEMM) return hit(o.g.a) || (o.g.has2 && hit(o.g.a2)) || (o.g.has_res && hit(o.g.r));
          if (o.kind == OP_DWCONV7) return hit(o.dw.in);
          if (o.kind == OP_GN_RES) return hit(o.gr.u) || hit(o.gr.x);
          if (o.kind == OP_LINATTN || o.kind == OP_ATTN) return hit(o.at.qkv);
          if (o.kind == OP_IM2COL) return hit(o.ic.in);
          return true;                                   // any other kind: assume it reads everything
        };
        const View* outv = op.kind == OP_GEMM ? &op.g.out : (op.kind == OP_DWCONV7 ? &op.dw.out : (op.kind == OP_GN_RES ? &op.gr.out : (op.kind == OP_IM2COL ? &op.ic.out : &op.at.out)));
        bool outside = i == (int)plan.ops.size() - 1;
        for (int j = 0; j < (int)plan.ops.size() && !outside; ++j) if ((j < first || j >= first + count) && j > i && reads(plan.ops[j], *outv)) outside = true;
        long long soff = op.kind == OP_GEMM ? op.g.stat_off : (op.kind == OP_DWCONV7 ? op.dw.stat_off : -1);
        bool stat_outside = false;
        for (int j = 0; j < (int)plan.ops.size() && soff >= 0; ++j)
          if ((j < first || j >= first + count) && ((plan.ops[j].kind == OP_GEMM && plan.ops[j].g.gn && plan.ops[j].g.gn_part_off == soff) || (plan.ops[j].kind == OP_GN_RES && plan.ops[j].gr.part_off == soff))) stat_outside = true;
        if (count == 1) { outside = true; stat_outside = true; }
        static float dummy_out;                            // non-null marker for the element-wise ops (they store through their own argument blocks)
        if (op.kind == OP_GEMM) {
          for (size_t ii = item_first; ii < b.items.size(); ++ii) { if (!outside) b.items[ii].out_g = nullptr; if (!stat_outside) b.items[ii].stat_g = nullptr; }
          if (!outside) d.out_g = nullptr;
          if (!stat_outside) d.stat_g = nullptr;
        } else {
          d.out_g = outside ? &dummy_out : nullptr;
          d.stat_g = (stat_outside && op.kind == OP_DWCONV7) ? reinterpret_cast<float2*>(&dummy_out) : nullptr;
        }
      }
      if (b.err) return b.err;
      for (int si : b.touched) b.slots[si].t1 = std::max(b.slots[si].t1, (int)b.ops.size());
      b.push_op(d);
    }
    b.flush_run();
    return nullptr;
  };
  PBuilder b0{plan, dp, st, first, count};
  if (const char* e = fill(b0)) return e;
  // pack the slabs by live interval, in whole chunks: every slab starts on a chunk boundary, so row 16 of every chunk position is the
  // zero row of whichever tensor lives there
  std::vector<PSlot> packed = b0.slots;
  int top_chunks = 0;
  {
    std::vector<size_t> order(packed.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return packed[x].t0 < packed[y].t0; });
    std::vector<size_t> placed;
    for (size_t oi : order) {
      PSlot& sl = packed[oi];
      const int size = PBuilder::slab_chunks(sl);
      int off = 0;
      bool moved = true;
      while (moved) {
        moved = false;
        for (size_t pj : placed) {
          const PSlot& o = packed[pj];
          const int osize = PBuilder::slab_chunks(o), ooff = o.lds / PCS;
          const bool live = !(o.t1 < sl.t0 || sl.t1 < o.t0);
          if (live && off < ooff + osize && ooff < off + size) { off = ooff + osize; moved = true; }
        }
      }
      sl.lds = off * PCS;
      placed.push_back(oi);
      top_chunks = std::max(top_chunks, off + size);
    }
  }
  PBuilder b{plan, dp, st, first, count};
  b.preset = &packed;
  if (const char* e = fill(b)) return e;
  const int act_floats = top_chunks * PCS;
  if (b.stat_of.size() > (size_t)P_STAT_SLOTS) return "pixel kernel: too many statistics slots";
  if (act_floats > 0xffff) return "pixel kernel: the activations exceed the step table's LDS range";
  if (pix_lds_bytes(act_floats, (int)b.rowtab.size() / 16) > 160 * 1024) return "pixel kernel: the activations do not fit the LDS";
  if (b.items.empty()) {                                   // a run without a GEMM still needs one (never consumed) item for the loader's prologue
    PixItem it{}; it.step_base = (int)b.steps.size(); it.nkp = PD; it.grp_stride = 0; it.P_out = 1; it.Npad = 32; it.perm_tab = -1; it.bias = it.gn_s = it.gn_b1 = st.zero4;
    // (step_base is not 0: a one-position run may have put its slot table at the front of the step array)
    b.items.push_back(it);
    for (int j = 0; j < PNW * PD; ++j) b.steps.push_back(PixStep{0u, 0u});
  }
  b.ops[0].pad = (int)b.steps.size();                      // the kernel's {kind, passes} table, one entry per program op
  for (const PixOpD& d : b.ops) b.steps.push_back(PixStep{(unsigned)d.kind, (unsigned)(d.kind == OP_GEMM ? d.nitems : 0)});
  PProgram p;
  p.n_ops = (int)b.ops.size(); p.n_items = (int)b.items.size(); p.n_rowtab = (int)b.rowtab.size() / 16; p.act_floats = act_floats; p.needs_ext_out = b.needs_ext_out; p.em_lds = b.em_lds;
  for (const PixOpD& d : b.ops) p.kinds.push_back(d.kind * 1000 + (d.kind == OP_GEMM ? d.nitems : 0));
  auto up = [&](void** dst, const void* src, size_t bytes) -> bool {
    if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up(reinterpret_cast<void**>(&p.d_ops), b.ops.data(), sizeof(PixOpD) * b.ops.size()) ||
      !up(reinterpret_cast<void**>(&p.d_items), b.items.data(), sizeof(PixItem) * b.items.size()) ||
      !up(reinterpret_cast<void**>(&p.d_steps), b.steps.data(), sizeof(PixStep) * b.steps.size()) ||
      !up(reinterpret_cast<void**>(&p.d_rowtab), b.rowtab.data(), b.rowtab.size() * sizeof(unsigned short))) {
    (void)hipFree(p.d_ops); (void)hipFree(p.d_items); (void)hipFree(p.d_steps); (void)hipFree(p.d_rowtab);
    return "pixel kernel: device allocation of a program failed";
  }
  // the GEMM ops of a run started mid-plan (debug sub-ranges) have no items of their own program before them: nothing to fix up --
  // item indices are relative to this program's item array
  *out = p;
  return nullptr;
}

void free_program(PProgram& p) { (void)hipFree(p.d_ops); (void)hipFree(p.d_items); (void)hipFree(p.d_steps); (void)hipFree(p.d_rowtab); p = PProgram{}; }

}  // namespace

unsigned long long* g_pix_dbg = nullptr;

// Model creation: fragment-order copies of the weights of every chain-level GEMM op (host params still available), and the programs of
// every maximal run of such ops.  Any failure leaves the model without a pixel state (the caller falls back to the legacy plan).
const char* pixrun_prepare(DevPlan& dp, const float* host_params, double* frag_mb) {
  const Plan& plan = *dp.plan;
  if (plan.ident_off < 0) return "pixel kernel: the plan has no zero vector";
  PixState* st = new (std::nothrow) PixState();
  if (!st) return "pixel kernel: out of host memory";
  std::vector<float> arena;
  int max_npad = 32;
  for (int i = 0; i < (int)plan.ops.size(); ++i) {
    const Op& op = plan.ops[i];
    if (!op.chain || op.time_path || op.kind != OP_GEMM) continue;
    if (const char* e = pix_gemm_supported(op.g)) { delete st; return e; }
    if (pix_vec_eligible(op.g)) pack_vec(host_params, op.g, &st->fops[i], &arena);
    else pack_fragments(host_params, op.g, &st->fops[i], &arena);
    max_npad = std::max(max_npad, op.g.Npad);
  }
  st->zero_blk = (long long)arena.size();                  // 8 KB of zeros: the padding steps of the one-position runs
  arena.resize(arena.size() + (size_t)PNT * 4, 0.f);
  const size_t zero_off = arena.size();                    // a zero VECTOR as long as the widest op: absent bias / GroupNorm tables are read at any column
  arena.resize(arena.size() + (size_t)max_npad + 64, 0.f);
  if (frag_mb) *frag_mb = arena.size() * 4.0 / 1e6;
  if (arena.size() > 0xffffffffull) { delete st; return "pixel kernel: fragment arena too large"; }
  static const double max_mb = std::getenv("SBM_PIX_MAX_MB") ? std::atof(std::getenv("SBM_PIX_MAX_MB")) : 48.0;
  // every workgroup streams ALL of these bytes once per forward (about 110 GB/s per CU): beyond a few tens of MB the per-layer kernels,
  // which share a weight tile among the samples of a tile, are faster (Unet(256): 300 MB)
  if (arena.size() * 4.0 / 1e6 > max_mb) { delete st; return "pixel kernel: the low-resolution weights exceed SBM_PIX_MAX_MB (sample-owned streaming would be slower than the per-layer kernels)"; }
  st->frag_floats = arena.size();
  if (hipMalloc(&st->d_frag, arena.size() * sizeof(float) + 4096) != hipSuccess) { delete st; return "pixel kernel: hipMalloc failed"; }
  if (hipMemcpy(st->d_frag, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(st->d_frag); delete st; return "pixel kernel: hipMemcpy failed"; }
  st->zero4 = st->d_frag + zero_off;
  ensure_dynamic_lds(reinterpret_cast<const void*>(pixrun_kernel), 160 * 1024);      // here, at model creation: per device and outside any stream capture
  dp.pix = st;
  for (int i = 0; i < (int)plan.ops.size();) {
    if (!plan.ops[i].chain || plan.ops[i].time_path) { ++i; continue; }
    int n = 1;
    while (i + n < (int)plan.ops.size() && plan.ops[i + n].chain && !plan.ops[i + n].time_path) ++n;
    PProgram p;
    if (const char* e = pixrun_build(dp, *st, i, n, &p)) { pixrun_release(dp); return e; }
    st->programs[(long long)i * 4096 + n] = p;
    i += n;
  }
  return nullptr;
}

void pixrun_release(DevPlan& dp) {
  PixState* st = dp.pix;
  if (!st) return;
  for (auto& kv : st->programs) free_program(kv.second);
  (void)hipFree(st->d_frag);
  delete st;
  dp.pix = nullptr;
}

static const PProgram* pix_program(const DevPlan& dp, int first, int count, const char** err) {
  PixState* st = dp.pix;
  std::lock_guard<std::mutex> lock(st->mu);
  const long long key = (long long)first * 4096 + count;
  auto it = st->programs.find(key);
  if (it == st->programs.end()) {                         // a sub-range (debug tools, op-by-op checks): built on demand, not capturable
    PProgram p;
    if (const char* e = pixrun_build(dp, *st, first, count, &p)) { *err = e; return nullptr; }
    it = st->programs.emplace(key, p).first;
  }
  return &it->second;
}

// true when the program that ends the plan can apply the sampler's EM update itself (launch_pixrun's `em`)
bool pixrun_can_fuse_em(const DevPlan& dp) {
  if (!dp.pix || !dp.use_pix) return false;
  const int nops = (int)dp.plan->ops.size();
  std::lock_guard<std::mutex> lock(dp.pix->mu);
  for (const auto& kv : dp.pix->programs) if ((int)(kv.first / 4096) + (int)(kv.first % 4096) == nops && kv.second.em_lds >= 0) return true;
  return false;
}

const char* launch_pixrun(const DevPlan& dp, int first, int count, hipStream_t st, float* ext_out, const StepArgs* em) {
  if (!dp.pix) return "pixel kernel: the model has no pixel state";
  const Plan& plan = *dp.plan;
  const char* err = nullptr;
  const PProgram* p = pix_program(dp, first, count, &err);
  if (!p) return err;
  if (p->needs_ext_out && !ext_out) return "pixel kernel: the run ends in the plan's external output, which this call does not have";
  if (em && (p->em_lds < 0 || em->B != plan.B || em->CHW > 4 * PNT || (em->CHW & 63))) return "pixel kernel: this run cannot apply the sampler's update";
  const int em_lds = em ? p->em_lds : -1;
  const StepArgs em_args = em ? *em : StepArgs{};
  const dim3 grid(plan.B), block(PNT);
  const size_t ldsb = pix_lds_bytes(p->act_floats, p->n_rowtab);
  if (g_prof_start) hipExtLaunchKernelGGL(pixrun_kernel, grid, block, ldsb, st, g_prof_start, g_prof_stop, 0, p->d_ops, p->n_ops, p->d_items, p->n_items, p->d_steps, p->d_rowtab, p->n_rowtab,
                                          dp.pix->d_frag, dp.ctl, plan.B, p->act_floats, g_pix_dbg, ext_out, em_lds, em_args);
  else hipLaunchKernelGGL(pixrun_kernel, grid, block, ldsb, st, p->d_ops, p->n_ops, p->d_items, p->n_items, p->d_steps, p->d_rowtab, p->n_rowtab, dp.pix->d_frag, dp.ctl, plan.B, p->act_floats, g_pix_dbg, ext_out, em_lds, em_args);
  return nullptr;
}

int pixrun_program_ops(const DevPlan& dp, int first, int count, int* kinds, int cap) {
  if (!dp.pix) return -1;
  const char* err = nullptr;
  const PProgram* p = pix_program(dp, first, count, &err);
  if (!p) return -1;
  for (int i = 0; i < (int)p->kinds.size() && i < cap; ++i) kinds[i] = p->kinds[i];
  return (int)p->kinds.size();
}

}  // namespace sbm
