// fid.hip -- the step right after the sampling path: statistics of the evaluation features and the Frechet
// distance between two Gaussians, in fp64 on the device.
//
// Reference (pytorch_fid/fid_score.py): the Inception pool3 activations of all images are collected into a float64 array
// (:127-153, :163-184), mu = np.mean(act, 0), sigma = np.cov(act, rowvar=False) (:248-250, :273-279), then
// calculate_frechet_distance (:187-240): d^2 = |mu1 - mu2|^2 + Tr(S1) + Tr(S2) - 2 Tr(sqrtm(S1 S2)), sqrtm by scipy (Schur).
//
// Here: features stay on the GPU; every rank accumulates sum(x) and the Gram matrix X^T X of its shard in fp64 (one
// v_mfma_f64_16x16x4_f64 GEMM per batch of features), the two are all-reduced once (host side: RCCL, 2048 + 2048^2 doubles),
// and the matrix square root is a coupled Newton-Schulz iteration made of the same fp64 MFMA GEMM.  Tr sqrtm(S1 S2) is computed
// as Tr sqrtm(S1^1/2 S2 S1^1/2): the same eigenvalues, but every iterate is a polynomial in a SYMMETRIC matrix, so the
// iteration never needs a transpose and the GEMM has one operand form (A^T B with both operands k-major).
//
// The Inception network itself is not built (no weights offline); the entry points take features.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "../../include/sbmae.h"

namespace sbm {
int api_fail(const std::string& m);                            // api.hip: records the message for sbm_last_error
int api_fail_hip(const char* what, hipError_t e);

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int DG_T = 128, DG_BK = 16, DG_LD = 144;             // tile edge, K chunk, LDS row stride in doubles (144 = 128 + 16: the two
                                                               // 16-lane halves of a ds_read_b64 group land on disjoint bank halves)

template <typename T> __device__ __forceinline__ void load8(const T* p, double* r);
template <> __device__ __forceinline__ void load8<double>(const double* p, double* r) {
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double2 v = q[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, double* r) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1];
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
}

// C[M x N] = alpha * op(A) * B + beta * C + diag * I with B [K][N] row-major and the first operand either k-major (AT: it is
// At [K][M], the product is At^T B -- the Gram matrix X^T X reads X for both operands) or row-major A [M][K] (the Newton-Schulz
// products: the iteration must be made of TRUE products; with A^T in place of A the rounding-level asymmetry of the iterates is
// an unstable mode and the iteration blows up after ~17 steps on a singular matrix).  fp64 accumulation on v_mfma_f64_16x16x4_f64.  One workgroup = 4 waves (2 x 2), a 128 x 128 tile of C; a wave owns 64 x 64 = 4 x 4
// MFMA tiles (lane l of the MFMA: A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15], D[row (l >> 4) + 4 r][col l & 15]).
// K chunks of 16 go global -> registers -> LDS, the next chunk's loads in flight under the current chunk's 64 MFMAs per wave
// (an fp64 MFMA holds the pipe 64 cycles: the kernel is matrix-pipe bound at 8 bytes per clock per CU of operand traffic).
// TN = 64: a 128 x 64 tile (wave: 4 x 2 MFMA tiles, 64 accumulator registers), 57 KB of LDS and <= 256 registers per wave, so that
// TWO workgroups share a CU -- the fp64 MFMA issues once per ~140 cycles per wave (tools/mfma_clock_probe), and only a second
// wave on the SIMD fills the gaps.
template <typename TIn, bool AT, int TN>
__global__ __launch_bounds__(256, TN == 64 ? 2 : 1) void dgemm_kernel(const TIn* __restrict__ At, long lda, const TIn* __restrict__ Bm, long ldb,
                                                       double* __restrict__ C, long ldc, int M, int N, int K, double alpha, double beta, double diag) {
  __shared__ __attribute__((aligned(16))) double As[2][DG_BK * DG_LD];
  constexpr int LDB = TN + 16, EB = TN / 16, NU = TN / 32;      // B tile row stride, columns per loader thread, MFMA tiles per wave in N
  __shared__ __attribute__((aligned(16))) double Bs[2][DG_BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * DG_T, n0 = blockIdx.x * TN;
  const int kr = tid >> 4, c8 = (tid & 15) * 8, cb = (tid & 15) * EB;   // loader: row kr of the chunk, 8 (A) / EB (B) consecutive columns
  const int am = tid >> 1, ak = (tid & 1) * 8;                 // !AT loader: row am of the tile, 8 consecutive k
  const bool vec_a = (lda % (16 / (long)sizeof(TIn))) == 0 && (reinterpret_cast<uintptr_t>(At) & 15) == 0 && (AT ? m0 + DG_T <= M : m0 + am < M);
  const bool vec_b = EB == 8 && (ldb % (16 / (long)sizeof(TIn))) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0 && n0 + TN <= N;
  double ra[8], rb[8];
  auto gload = [&](int k0) {
    const int k = k0 + kr;
    if constexpr (AT) {
      if (k < K && vec_a) load8<TIn>(At + (long)k * lda + m0 + c8, ra);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e] = (k < K && m0 + c8 + e < M) ? (double)At[(long)k * lda + m0 + c8 + e] : 0.0;
      }
    } else {
      if (vec_a && k0 + ak + 8 <= K) load8<TIn>(At + (long)(m0 + am) * lda + k0 + ak, ra);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e] = (m0 + am < M && k0 + ak + e < K) ? (double)At[(long)(m0 + am) * lda + k0 + ak + e] : 0.0;
      }
    }
    if (k < K && vec_b) load8<TIn>(Bm + (long)k * ldb + n0 + c8, rb);
    else {
#pragma unroll
      for (int e = 0; e < EB; ++e) rb[e] = (k < K && n0 + cb + e < N) ? (double)Bm[(long)k * ldb + n0 + cb + e] : 0.0;
    }
  };
  auto lstore = [&](int buf) {
    double2* pb = reinterpret_cast<double2*>(&Bs[buf][kr * LDB + cb]);
#pragma unroll
    for (int i = 0; i < EB / 2; ++i) pb[i] = make_double2(rb[2 * i], rb[2 * i + 1]);
    if constexpr (AT) {
      double2* pa = reinterpret_cast<double2*>(&As[buf][kr * DG_LD + c8]);
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[i] = make_double2(ra[2 * i], ra[2 * i + 1]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) As[buf][(ak + e) * DG_LD + am] = ra[e];        // transposed into the k-major tile
    }
  };
  f64x4 acc[4][NU];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;
  for (int k0 = 0, it = 0; k0 < K; k0 += DG_BK, ++it) {
    const int buf = it & 1;
    const bool more = k0 + DG_BK < K;
    if (more) gload(k0 + DG_BK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double* ap = &As[buf][(4 * kk + fk) * DG_LD + wm * 64 + fr];
      const double* bp = &Bs[buf][(4 * kk + fk) * LDB + wn * (TN / 2) + fr];
      double a[4], b[NU];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = ap[16 * t];
#pragma unroll
      for (int u = 0; u < NU; ++u) b[u] = bp[16 * u];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[u], acc[t][u], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + t * 16 + fk + 4 * r, col = n0 + wn * (TN / 2) + u * 16 + fr;
        if (row < M && col < N) {
          double v = alpha * acc[t][u][r];
          if (beta != 0.0) v += beta * C[(long)row * ldc + col];
          if (row == col) v += diag;
          C[(long)row * ldc + col] = v;
        }
      }
}

template <typename TIn>
void dgemm_tn(const TIn* At, long lda, const TIn* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha, double beta, double diag, hipStream_t st) {
  static const int tn = std::getenv("SBM_DGEMM_TN") ? std::atoi(std::getenv("SBM_DGEMM_TN")) : 64;
  dim3 grid((N + tn - 1) / tn, (M + DG_T - 1) / DG_T);
  if (tn == 64) hipLaunchKernelGGL((dgemm_kernel<TIn, true, 64>), grid, dim3(256), 0, st, At, lda, B, ldb, C, ldc, M, N, K, alpha, beta, diag);
  else hipLaunchKernelGGL((dgemm_kernel<TIn, true, 128>), grid, dim3(256), 0, st, At, lda, B, ldb, C, ldc, M, N, K, alpha, beta, diag);
}
void dgemm_nn(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha, double beta, double diag, hipStream_t st) {
  static const int tn = std::getenv("SBM_DGEMM_TN") ? std::atoi(std::getenv("SBM_DGEMM_TN")) : 64;
  dim3 grid((N + tn - 1) / tn, (M + DG_T - 1) / DG_T);
  if (tn == 64) hipLaunchKernelGGL((dgemm_kernel<double, false, 64>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, diag);
  else hipLaunchKernelGGL((dgemm_kernel<double, false, 128>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, diag);
}

// A <- (A + A^T) / 2 (square, in place: thread (i, j), i < j, owns the pair)
__global__ void symmetrize_kernel(double* __restrict__ A, int n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (long)i * n);
  if (i < j) { const double v = 0.5 * (A[idx] + A[(long)j * n + i]); A[idx] = v; A[(long)j * n + i] = v; }
}

// sum[d] += sum over rows of x[., d] (fp64)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, long n, int dims, double* __restrict__ sum) {
  const int d = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ double red[4][64];
  double s = 0.0;
  if (d < dims)
    for (long r = (long)blockIdx.y * 4 + part; r < n; r += (long)gridDim.y * 4) s += (double)x[r * ld + d];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && d < dims) atomicAdd(&sum[d], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// mu = sum / n, sigma = (gram - n mu mu^T) / (n - 1)      (np.mean / np.cov(rowvar=False): ddof = 1)
__global__ void finalize_stats_kernel(const double* __restrict__ sum, const double* __restrict__ gram, double n, int dims, double* __restrict__ mu, double* __restrict__ sigma) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)dims * dims) return;
  const int i = (int)(idx / dims), j = (int)(idx - (long)i * dims);
  const double mi = sum[i] / n, mj = sum[j] / n;
  sigma[idx] = (gram[idx] - n * mi * mj) / (n - 1.0);
  if (j == 0) mu[i] = mi;
}

// out[0] += sum of squares (Frobenius^2), out[1] += trace
__global__ __launch_bounds__(256) void fro_trace_kernel(const double* __restrict__ A, int n, double* __restrict__ out) {
  __shared__ double rs[256], rt[256];
  double s = 0.0, t = 0.0;
  const long total = (long)n * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const double v = A[i];
    s += v * v;
    if (i / n == i % n) t += v;
  }
  rs[threadIdx.x] = s; rt[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rt[threadIdx.x] += rt[threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) { atomicAdd(&out[0], rs[0]); atomicAdd(&out[1], rt[0]); }
}

// Y = A * s (+ shift on the diagonal), Z = I
__global__ void ns_init_kernel(const double* __restrict__ A, double s, double shift, int n, double* __restrict__ Y, double* __restrict__ Z) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const bool dg = idx / n == idx % n;
  Y[idx] = (A[idx] + (dg ? shift : 0.0)) * s;
  if (Z) Z[idx] = dg ? 1.0 : 0.0;
}

// out[0] = |mu1 - mu2|^2
__global__ __launch_bounds__(256) void diff2_kernel(const double* __restrict__ a, const double* __restrict__ b, int n, double* __restrict__ out) {
  __shared__ double rs[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { const double d = a[i] - b[i]; s += d * d; }
  rs[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if ((int)threadIdx.x < o) rs[threadIdx.x] += rs[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = rs[0];
}

struct NsResult { double trace; int iters; bool finite; };

// Principal square root of a symmetric positive semi-definite A (n x n) by the coupled Newton-Schulz iteration
//   Y0 = A / |A|_F, Z0 = I;  T = (3 I - Z Y) / 2;  Y <- Y T, Z <- T Z;   Y -> (A / |A|_F)^1/2
// stopped when Tr(Y) no longer moves (relative 1e-13, or no longer improving below 1e-9) -- on a singular A the null-space noise
// grows by 1.5 per step after that, so running on would destroy the result -- or after max_iters.  Work: Y, Z, T, and two scratch matrices (5 n^2 doubles); the root
// is left in *root (one of the work matrices).  scal is 2 doubles of device scratch, hscal 2 doubles of pinned host memory.
int ns_sqrt(const double* A, int n, double shift, int max_iters, double* work, double* scal, double* hscal, double** root, NsResult* res, hipStream_t st) {
  const long nn = (long)n * n;
  double* Y = work; double* Z = work + nn; double* T = work + 2 * nn; double* Y2 = work + 3 * nn; double* Z2 = work + 4 * nn;
  const int eb = (int)((nn + 255) / 256);
  hipError_t e;
  if ((e = hipMemsetAsync(scal, 0, 2 * sizeof(double), st)) != hipSuccess) return api_fail_hip("hipMemsetAsync", e);
  hipLaunchKernelGGL(ns_init_kernel, dim3(eb), dim3(256), 0, st, A, 1.0, shift, n, Y, (double*)nullptr);     // Y = A + shift I (to take its norm)
  hipLaunchKernelGGL(fro_trace_kernel, dim3(256), dim3(256), 0, st, Y, n, scal);
  if ((e = hipMemcpyAsync(hscal, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) return api_fail_hip("hipMemcpyAsync", e);
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return api_fail_hip("hipStreamSynchronize", e);
  const double nf = std::sqrt(hscal[0]);
  res->finite = std::isfinite(nf);
  res->iters = 0; res->trace = 0.0; *root = Y;
  if (!res->finite) return 0;
  if (nf == 0.0) { hipLaunchKernelGGL(ns_init_kernel, dim3(eb), dim3(256), 0, st, A, 0.0, 0.0, n, Y, (double*)nullptr); return 0; }
  hipLaunchKernelGGL(ns_init_kernel, dim3(eb), dim3(256), 0, st, A, 1.0 / nf, shift, n, Y, Z);
  double tr_prev = hscal[1] / nf, delta_prev = 1.0;
  bool use_prev = false;
  for (int k = 0; k < max_iters; ++k) {
    dgemm_nn(Z, n, Y, n, T, n, n, n, n, -0.5, 0.0, 1.5, st);                  // T = 1.5 I - 0.5 Z Y
    dgemm_nn(Y, n, T, n, Y2, n, n, n, n, 1.0, 0.0, 0.0, st);                  // Y' = Y T
    dgemm_nn(T, n, Z, n, Z2, n, n, n, n, 1.0, 0.0, 0.0, st);                  // Z' = T Z
    double* t0 = Y; Y = Y2; Y2 = t0; t0 = Z; Z = Z2; Z2 = t0;
    if ((e = hipMemsetAsync(scal, 0, 2 * sizeof(double), st)) != hipSuccess) return api_fail_hip("hipMemsetAsync", e);
    hipLaunchKernelGGL(fro_trace_kernel, dim3(256), dim3(256), 0, st, Y, n, scal);
    if ((e = hipMemcpyAsync(hscal, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) return api_fail_hip("hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return api_fail_hip("hipStreamSynchronize", e);
    const double tr = hscal[1];
    res->iters = k + 1;
    if (!std::isfinite(tr) || !std::isfinite(hscal[0])) { res->finite = false; break; }
    const double delta = std::fabs(tr - tr_prev) / std::fabs(tr);
    if (delta <= 1e-13) { tr_prev = tr; break; }
    // the trace has stopped improving at rounding level: from here on the steps only amplify null-space noise (by 1.5 each) --
    // keep the previous iterate, which the buffer swap left in Y2
    if (k >= 3 && delta_prev < 1e-9 && delta >= delta_prev) { use_prev = true; break; }
    tr_prev = tr; delta_prev = delta;
  }
  // Y ~ (A / nf)^1/2  ->  A^1/2 = Y sqrt(nf)
  if (use_prev) { double* t0 = Y; Y = Y2; Y2 = t0; }
  hipLaunchKernelGGL(ns_init_kernel, dim3(eb), dim3(256), 0, st, Y, std::sqrt(nf), 0.0, n, Y2, (double*)nullptr);
  *root = Y2;
  res->trace = tr_prev * std::sqrt(nf);
  return 0;
}

}  // namespace
}  // namespace sbm

using namespace sbm;

extern "C" {

int sbm_fid_accumulate(const float* feats, long n, int dims, long ld, double* sum, double* gram, void* stream) {
  if (!feats || !sum || !gram || n < 0 || dims <= 0 || ld < dims) return api_fail("sbm_fid_accumulate: bad arguments");
  if (n == 0) return 0;
  if (n > 0x7fffffffL) return api_fail("sbm_fid_accumulate: more than 2^31 rows per call");      // before the first launch: no partial update
  hipStream_t st = (hipStream_t)stream;
  const int yb = (int)(n < 4096 ? (n + 63) / 64 : 64);
  hipLaunchKernelGGL(colsum_kernel, dim3((dims + 63) / 64, yb > 0 ? yb : 1), dim3(256), 0, st, feats, ld, n, dims, sum);
  dgemm_tn<float>(feats, ld, feats, ld, gram, dims, dims, dims, (int)n, 1.0, 1.0, 0.0, st);       // gram += X^T X
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : api_fail_hip("sbm_fid_accumulate", e);
}

int sbm_fid_finalize_stats(const double* sum, const double* gram, long n, int dims, double* mu, double* sigma, void* stream) {
  if (!sum || !gram || !mu || !sigma || n < 2 || dims <= 0) return api_fail("sbm_fid_finalize_stats: bad arguments (need n >= 2)");
  const long nn = (long)dims * dims;
  hipLaunchKernelGGL(finalize_stats_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum, gram, (double)n, dims, mu, sigma);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : api_fail_hip("sbm_fid_finalize_stats", e);
}

long sbm_fid_work_doubles(int dims) { return 7L * dims * dims + 16; }

int sbm_fid_frechet(const double* mu1, const double* sigma1, const double* mu2, const double* sigma2, int dims, double eps, int max_iters,
                    double* work, double* out, void* stream) {
  if (!mu1 || !sigma1 || !mu2 || !sigma2 || !work || !out || dims <= 0) return api_fail("sbm_fid_frechet: bad arguments");
  if (max_iters <= 0) max_iters = 60;
  hipStream_t st = (hipStream_t)stream;
  const int n = dims;
  const long nn = (long)n * n;
  double* M = work + 5 * nn; double* U = work + 6 * nn; double* scal = work + 7 * nn;     // scal: 16 doubles
  double* hscal = nullptr;
  hipError_t e;
  if ((e = hipHostMalloc(&hscal, 8 * sizeof(double))) != hipSuccess) return api_fail_hip("hipHostMalloc", e);
  int rc = 0;
  double tr_cov = 0.0; int it1 = 0, it2 = 0, used_eps = 0;
  for (int attempt = 0; attempt < 2 && rc == 0; ++attempt) {
    const double shift = attempt ? eps : 0.0;                  // fid_score.py:224-229: on a non-finite root, add eps to both diagonals and retry
    double* S = nullptr; NsResult r1{}, r2{};
    if ((rc = ns_sqrt(sigma1, n, shift, max_iters, work, scal, hscal, &S, &r1, st)) != 0) break;
    it1 = r1.iters;
    bool ok = r1.finite;
    if (ok) {
      hipLaunchKernelGGL(ns_init_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, sigma2, 1.0, shift, n, M, (double*)nullptr);   // M = S2 (+ eps I)
      dgemm_nn(M, n, S, n, U, n, n, n, n, 1.0, 0.0, 0.0, st);                // U = S2 S
      dgemm_nn(S, n, U, n, M, n, n, n, n, 1.0, 0.0, 0.0, st);                // M = S S2 S   (same eigenvalues as S1 S2)
      hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, M, n);
      double* R = nullptr;
      if ((rc = ns_sqrt(M, n, 0.0, max_iters, work, scal, hscal, &R, &r2, st)) != 0) break;
      it2 = r2.iters; ok = r2.finite; tr_cov = r2.trace;
    }
    used_eps = attempt;
    if (ok) break;
    if (attempt == 1) { rc = api_fail("sbm_fid_frechet: the matrix square root is not finite even with eps on the diagonals"); }
  }
  if (rc == 0) {
    // |mu1 - mu2|^2, Tr(S1), Tr(S2)
    do {
      if ((e = hipMemsetAsync(scal, 0, 8 * sizeof(double), st)) != hipSuccess) { rc = api_fail_hip("hipMemsetAsync", e); break; }
      hipLaunchKernelGGL(fro_trace_kernel, dim3(256), dim3(256), 0, st, sigma1, n, scal);
      hipLaunchKernelGGL(fro_trace_kernel, dim3(256), dim3(256), 0, st, sigma2, n, scal + 2);
      hipLaunchKernelGGL(diff2_kernel, dim3(1), dim3(256), 0, st, mu1, mu2, n, scal + 4);
      if ((e = hipMemcpyAsync(hscal, scal, 5 * sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) { rc = api_fail_hip("hipMemcpyAsync", e); break; }
      if ((e = hipStreamSynchronize(st)) != hipSuccess) { rc = api_fail_hip("hipStreamSynchronize", e); break; }
      out[0] = hscal[4] + hscal[1] + hscal[3] - 2.0 * tr_cov;
      out[1] = tr_cov; out[2] = (double)it1; out[3] = (double)it2; out[4] = (double)used_eps;
    } while (0);
  }
  (void)hipHostFree(hscal);
  return rc;
}

int sbm_debug_dgemm_tn(const double* At, const double* B, double* C, int M, int N, int K, double alpha, double beta, double diag, void* stream) {
  if (!At || !B || !C || M <= 0 || N <= 0 || K == 0) return api_fail("sbm_debug_dgemm_tn: bad arguments");
  if (K < 0) dgemm_nn(At, -K, B, N, C, N, M, N, -K, alpha, beta, diag, (hipStream_t)stream);      // K < 0: the first operand is A [M][-K] row-major
  else dgemm_tn<double>(At, M, B, N, C, N, M, N, K, alpha, beta, diag, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : api_fail_hip("sbm_debug_dgemm_tn", e);
}

}  // extern "C"
