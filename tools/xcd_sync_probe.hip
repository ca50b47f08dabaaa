// xcd_sync_probe.hip -- hardware facts the XCD-resident chain kernel depends on (gfx950):
//   1. XCC_ID is readable and workgroups are spread round-robin over the 8 XCDs;
//   2. latency of a barrier among the workgroups of ONE XCD, with workgroup-scope (L2-local) vs agent-scope atomics;
//   3. which load flavour sees data another CU of the same XCD has just stored (stale-L1 test).
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_sync_probe tools/xcd_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Sync { unsigned count[8]; unsigned arrived; unsigned abort_flag; unsigned pad[6]; unsigned bar[8 * 16]; };

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int SCOPE> __device__ __forceinline__ unsigned rmw_add(unsigned* p, unsigned v) {
  if (SCOPE == 0) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false on timeout
template <int SCOPE> __device__ __forceinline__ bool cluster_barrier(unsigned* ctr, unsigned target, unsigned* abort_flag) {
  __syncthreads();                                   // every wave: s_waitcnt vmcnt(0) -> its stores reached L2
  bool ok = true;
  if (threadIdx.x == 0) {
    rmw_add<SCOPE>(ctr, 1u);
    unsigned spins = 0;
    while (true) {
      const unsigned v = SCOPE == 0 ? rmw_add<0>(ctr, 0u) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

__device__ __forceinline__ float4 load_sc1(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float4 load_sc0sc1(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// out[block] = {xcc, slot, cluster size, barrier cycles (100 MHz clock ticks) wg-scope, agent-scope, mismatches per load mode x4}
template <int BSCOPE, int MIX>
__global__ __launch_bounds__(256) void probe(Sync* s, float* buf, unsigned* out, int iters) {
  __shared__ unsigned sh[4];
  const unsigned x = MIX ? blockIdx.x / 32 : xcc_id();
  if (threadIdx.x == 0) {
    sh[0] = __hip_atomic_fetch_add(&s->count[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&s->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&s->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    sh[1] = __hip_atomic_load(&s->count[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[2] = spins >= (1u << 22);
  }
  __syncthreads();
  const unsigned slot = sh[0], cs = sh[1];
  if (sh[2]) { if (threadIdx.x == 0) out[blockIdx.x * 16] = 0xdead; return; }
  unsigned* bar = &s->bar[x * 16];
  unsigned gen = 0;
  // ---- barrier latency
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { gen += cs; if (!cluster_barrier<BSCOPE>(bar, gen, &s->abort_flag)) return; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { gen += cs; if (!cluster_barrier<BSCOPE>(bar, gen, &s->abort_flag)) return; asm volatile("buffer_inv sc1" ::: "memory"); }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(8);
  unsigned long long m3 = __builtin_amdgcn_s_memtime();
  unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 16 + 9] = (unsigned)(t2 - t1); out[blockIdx.x * 16 + 10] = (unsigned)(m3 - t2); out[blockIdx.x * 16 + 11] = (unsigned)(w1 - w0); (void)r0; }
  // ---- visibility: write my 4 KiB chunk, barrier, read the neighbour's
  float* mine = buf + ((size_t)x * 64 + slot) * 1024;
  const float* theirs = buf + ((size_t)x * 64 + (slot + 1) % cs) * 1024;
  unsigned bad[5] = {0, 0, 0, 0, 0};
  for (int mode = 0; mode < 5; ++mode) {
    for (int i = 0; i < iters; ++i) {
      const float val = (float)(i * 7 + mode * 1000 + 1);
      reinterpret_cast<float4*>(mine)[threadIdx.x] = make_float4(val + slot, val, val, val);
      gen += cs; if (!cluster_barrier<BSCOPE>(bar, gen, &s->abort_flag)) return;
      float4 v;
      if (mode == 0) v = reinterpret_cast<const float4*>(theirs)[threadIdx.x];
      else if (mode == 1) v = load_sc1(theirs + threadIdx.x * 4);
      else if (mode == 2) v = load_sc0sc1(theirs + threadIdx.x * 4);
      else if (mode == 3) { asm volatile("buffer_inv sc1" ::: "memory"); v = reinterpret_cast<const float4*>(theirs)[threadIdx.x]; }
      else { asm volatile("buffer_inv sc0" ::: "memory"); v = reinterpret_cast<const float4*>(theirs)[threadIdx.x]; }
      if (v.x != val + (float)((slot + 1) % cs) || v.y != val) bad[mode]++;
      gen += cs; if (!cluster_barrier<BSCOPE>(bar, gen, &s->abort_flag)) return;
    }
  }
  for (int mode = 0; mode < 5; ++mode) {
    unsigned b = bad[mode];
    for (int o = 32; o >= 1; o >>= 1) b += __shfl_xor(b, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x * 16 + 4 + mode], b);
  }
  if (threadIdx.x == 0) { out[blockIdx.x * 16] = xcc_id(); out[blockIdx.x * 16 + 1] = slot; out[blockIdx.x * 16 + 2] = cs; out[blockIdx.x * 16 + 3] = (unsigned)(t1 - t0); }
}

int main() {
  const int grid = 256, iters = 2000;
  Sync* s; float* buf; unsigned* out;
  CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&buf, 8 * 64 * 1024 * sizeof(float))); CK(hipMalloc(&out, grid * 16 * sizeof(unsigned)));
  std::vector<unsigned> h(grid * 16);
  for (int scope = 1; scope < 3; ++scope) {
    CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(out, 0, grid * 16 * sizeof(unsigned))); CK(hipMemset(buf, 0, 8 * 64 * 1024 * sizeof(float)));
    if (scope == 1) hipLaunchKernelGGL((probe<1, 0>), dim3(grid), dim3(256), 0, 0, s, buf, out, iters);
    else hipLaunchKernelGGL((probe<1, 1>), dim3(grid), dim3(256), 0, 0, s, buf, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned abort_flag = 0; CK(hipMemcpy(&abort_flag, &s->abort_flag, 4, hipMemcpyDeviceToHost));
    int hist[16] = {0}, rr = 0; unsigned long long cyc = 0; unsigned long long bad[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < grid; ++b) {
      hist[h[b * 16] & 15]++; rr += (h[b * 16] == (unsigned)(b % 8)); cyc += h[b * 16 + 3];
      for (int m = 0; m < 5; ++m) bad[m] += h[b * 16 + 4 + m];
    }
    printf("barrier atomics: %s scope | abort=%u | xcc histogram:", scope == 1 ? "clusters = XCDs" : "clusters span XCDs (blockIdx/32)", abort_flag);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf(" | blockIdx%%8==xcc for %d/%d | barrier %.3f us (avg over %d)\n", rr, grid, (double)cyc / grid / iters / 100.0, iters);
    printf("   barrier+buffer_inv sc1: %.1f memtime ticks; plain barrier %.1f ticks; calibration: %u memtime ticks per %u wall_clock64 ticks (100 MHz) => %.1f MHz\n",
           (double)h[9] / iters, (double)h[3] / iters, h[10], h[11], 100.0 * h[10] / h[11]);
    printf("   mismatching lanes: plain %llu | sc1 %llu | sc0sc1 %llu | buffer_inv sc1 + plain %llu | buffer_inv sc0 + plain %llu  (of %llu)\n", bad[0], bad[1], bad[2], bad[3],
           bad[4], (unsigned long long)grid * 256 * iters);
  }
  return 0;
}
