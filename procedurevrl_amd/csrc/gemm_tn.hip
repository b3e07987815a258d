// bf16 MFMA GEMM, "TN" form (weight gradients): the C-ABI entry points.  The kernels live in gemm_tn_core.h.
#include <algorithm>
#include <cstdlib>
#include "gemm_tn8_core.h"

namespace {

// Kernel choice: the register-transposed 8-wave kernel (256x256 tiles; it stages half tiles, N or K = 128 mod 256, with
// zero columns) whenever N and K are multiples of 128 and dW has at least one full tile; otherwise the 128x128
// transposing-read kernel.  (Measured-and-rejected alternatives live under tools/probe/, outside this library.)
bool tn_use_rt(int64_t N, int64_t K) { return (N % 128 == 0) && (K % 128 == 0) && N * K >= 256 * 256; }
// The ping-pong LDS-DMA kernel (gemm_tn8_core.h) takes the shapes made of whole 256x256 tiles -- every Linear of the ViT-B encoder;
// PVRL_TN8=0 sends them back to the register-transposed kernel (A/B runs; read once).  Results are bit-identical either way.
bool tn8_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_TN8");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on != 0;
}
// (its buffer descriptors and per-lane offsets are 32-bit: a slice's rows x the row pitch must stay below 2 GiB)
bool tn_use_tn8(int64_t N, int64_t K, int64_t slice_rows, int64_t ldp, int64_t ldq) {
  return tn8_enabled() && (N % 256 == 0) && (K % 256 == 0) && (slice_rows + 64) * std::max(ldp, ldq) * 2 < (int64_t(1) << 31) - 4096;
}

}  // namespace

extern "C" int64_t pvrl_gemm_tn_plan_splits(int64_t M, int64_t N, int64_t K) {
  if (N <= 0 || K <= 0 || (N % 128) || (K % 128)) return PVRL_EINVAL;
  if (tn_use_rt(N, K)) {
    // one workgroup per CU and ONE round: as many (slice, tile) pairs as fit the 256 CUs, slices of >= 64 rows
    const int64_t tiles = cdiv(N, 256) * cdiv(K, 256);
    int64_t s = 8 * pvrl_compute_cus_per_xcd() / tiles;
    if (tiles == 1) s = 128;          // a single 256x256 tile: 128 slices measured ahead of 256 (200,736 x 256 x 256: 89 vs 109 us)
    const int64_t smax = M / 64;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
  }
  // 128x128 kernels: a multiple of 8 (slice s lives on XCD s % 8), enough (n, k) tiles x slices to fill
  // 8 XCDs x 64 resident workgroups about twice, but at least ~256 rows per slice
  const int64_t tiles = (N / 128) * (K / 128);
  int64_t per_xcd = cdiv(128, tiles);
  if (per_xcd < 1) per_xcd = 1;
  int64_t s = 8 * per_xcd;
  const int64_t minrows = tiles == 1 ? 2048 : 256;     // one 128x128 tile: 802,848 x 128 x 128 at 392 slices 108 us, at 1,024 150 us
  while (s > 8 && M / s < minrows) s -= 8;
  return s;
}

extern "C" int64_t pvrl_gemm_tn_workspace_bytes(int64_t N, int64_t K, int64_t splits) {
  return splits * (N * K + N) * (int64_t)sizeof(float) + 256;   // + a zero page for out-of-range rows
}

namespace {
int gemm_tn_impl(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K, int64_t splits,
                 float beta, float* dW, int64_t ldw, int64_t n_valid, int64_t k_valid, float* dbias, float beta_bias,
                 void* workspace, int64_t workspace_bytes, const float* gscale, float* nonfinite, void* stream);
}

extern "C" int pvrl_gemm_tn_bf16(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N,
                                 int64_t K, int64_t splits, float beta, float* dW, float* dbias, void* workspace,
                                 int64_t workspace_bytes, const float* gscale, float* nonfinite, void* stream) {
  return gemm_tn_impl(P, ldp, Q, ldq, M, N, K, splits, beta, dW, K, N, K, dbias, beta, workspace, workspace_bytes, gscale,
                      nonfinite, stream);
}

extern "C" int pvrl_gemm_tn_into_bf16(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N,
                                      int64_t K, int64_t splits, float beta, float* dW, int64_t ldw, int64_t n_valid,
                                      int64_t k_valid, float* dbias, float beta_bias, void* workspace,
                                      int64_t workspace_bytes, const float* gscale, float* nonfinite, void* stream) {
  if (n_valid < 1 || n_valid > N || k_valid < 1 || k_valid > K || ldw < k_valid) return PVRL_EINVAL;
  return gemm_tn_impl(P, ldp, Q, ldq, M, N, K, splits, beta, dW, ldw, n_valid, k_valid, dbias, beta_bias, workspace,
                      workspace_bytes, gscale, nonfinite, stream);
}

namespace {
int gemm_tn_impl(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K, int64_t splits,
                 float beta, float* dW, int64_t ldw, int64_t n_valid, int64_t k_valid, float* dbias, float beta_bias,
                 void* workspace, int64_t workspace_bytes, const float* gscale, float* nonfinite, void* stream) {
  if (!P || !Q || !dW || !workspace || N <= 0 || K <= 0 || (N % 128) || (K % 128) || splits < 1 || M < 0)
    return PVRL_EINVAL;
  const bool use_rt = tn_use_rt(N, K);
  if (!use_rt && (splits < 8 || (splits % 8))) return PVRL_EINVAL;   // slice s lives on XCD s % 8 in those kernels
  if ((ldp % 8) || (ldq % 8) || ((uintptr_t)P % 16) || ((uintptr_t)Q % 16)) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_gemm_tn_workspace_bytes(N, K, splits)) return PVRL_EINVAL;
  GemmTN p;
  p.P = (const op_t*)P; p.ldp = ldp; p.Q = (const op_t*)Q; p.ldq = ldq;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  int ms = cdiv(M > 0 ? M : 1, splits);
  p.Ms = cdiv(ms, TM) * TM;
  p.part = (float*)workspace;
  p.cpart = dbias ? p.part + splits * N * K : nullptr;
  hipStream_t s = (hipStream_t)stream;
  char* zp = (char*)workspace + splits * (N * K + N) * (int64_t)sizeof(float);
  p.zero_page = (const op_t*)zp;
  if (use_rt) {
    p.tiles_k = (int)cdiv(K, 256);
    p.tiles_nk = (int)cdiv(N, 256) * p.tiles_k;
    p.npairs = (int)splits * p.tiles_nk;
    p.Ms_pairs = cdiv(p.npairs, 8);
    if (tn_use_tn8(N, K, p.Ms, ldp, ldq)) hipLaunchKernelGGL(gemm_tn8_kernel, dim3((unsigned)(8 * p.Ms_pairs)), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(gemm_tn_rt8_kernel, dim3((unsigned)(8 * p.Ms_pairs)), dim3(512), 0, s, p);
  } else {
    p.tiles_k = (int)(K / 128);
    p.tiles_nk = (int)(N / 128) * p.tiles_k;
    hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), dim3((unsigned)(splits * p.tiles_nk)), dim3(256), 0, s, p);
  }
  PVRL_LAUNCH_CHECK();
  const long NK = N * K;
  const long nthreads = (NK >> 2) + (dbias ? N : 0);
  const bool into = ldw != K || n_valid != N || k_valid != K || beta_bias != beta;
  if (into) {
    if (beta_bias != beta && dbias) {       // one beta per launch: the bias gets its own pass of the small kernel
      hipLaunchKernelGGL(tn_reduce_into_kernel, dim3((unsigned)cdiv(NK >> 2, 64)), dim3(256), 0, s, p.part, p.cpart,
                         (int)splits, NK, (int)N, (int)K, beta, dW, (long)ldw, (int)n_valid, (int)k_valid, (float*)nullptr, gscale, nonfinite);
      hipLaunchKernelGGL(tn_reduce_into_kernel, dim3((unsigned)cdiv(N, 64)), dim3(256), 0, s, p.part, p.cpart,
                         (int)splits, 0L, (int)N, (int)K, beta_bias, dW, (long)ldw, (int)n_valid, (int)k_valid, dbias, gscale, nonfinite);
    } else {
      hipLaunchKernelGGL(tn_reduce_into_kernel, dim3((unsigned)cdiv(nthreads, 64)), dim3(256), 0, s, p.part, p.cpart,
                         (int)splits, NK, (int)N, (int)K, beta, dW, (long)ldw, (int)n_valid, (int)k_valid, dbias, gscale, nonfinite);
    }
  } else if (nthreads < 64 * 256 && splits >= 8)
    hipLaunchKernelGGL(tn_reduce_small_kernel, dim3((unsigned)cdiv(nthreads, 64)), dim3(256), 0, s, p.part, p.cpart,
                       (int)splits, NK, (int)N, beta, dW, dbias, gscale, nonfinite);
  else
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)cdiv(nthreads, 256)), dim3(256), 0, s, p.part, p.cpart,
                       (int)splits, NK, (int)N, beta, dW, dbias, gscale, nonfinite);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// grouped weight gradients
// ---------------------------------------------------------------------------------------------------------
namespace {
bool tn_group_ok(int nprob, const pvrl_tn_problem* pr) {
  if (nprob < 1 || nprob > TN_GROUP_MAX || !pr) return false;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = pr[i];
    if (!q.P || !q.Q || !q.dW || q.M < 1 || q.N <= 0 || q.K <= 0 || (q.N % 128) || (q.K % 128)) return false;
    if ((q.ldp % 8) || (q.ldq % 8) || ((uintptr_t)q.P % 16) || ((uintptr_t)q.Q % 16)) return false;
  }
  return true;
}
int64_t tn_group_tiles(int nprob, const pvrl_tn_problem* pr) {
  int64_t t = 0;
  for (int i = 0; i < nprob; ++i) t += cdiv(pr[i].N, 256) * cdiv(pr[i].K, 256);
  return t;
}
}  // namespace

extern "C" int64_t pvrl_gemm_tn_grouped_plan_splits(int nprob, const pvrl_tn_problem* problems) {
  if (!tn_group_ok(nprob, problems)) return PVRL_EINVAL;
  const int64_t T = tn_group_tiles(nprob, problems);
  int64_t smax = 32;
  for (int i = 0; i < nprob; ++i) smax = std::min<int64_t>(smax, std::max<int64_t>(1, problems[i].M / 64));
  // the smallest slice count whose T*s equal work items fill whole rounds of the (256) CUs to >= 97 %, else the best one
  int64_t best = 1;
  double best_eff = 0.0;
  const int64_t ncu = 8 * pvrl_compute_cus_per_xcd();
  for (int64_t s = 1; s <= smax; ++s) {
    const int64_t items = T * s;
    const double eff = (double)items / (double)(ncu * cdiv(items, ncu));
    if (eff >= 0.97) return s;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
  }
  return best;
}

extern "C" int64_t pvrl_gemm_tn_grouped_workspace_bytes(int nprob, const pvrl_tn_problem* problems, int64_t splits) {
  if (!tn_group_ok(nprob, problems) || splits < 1) return PVRL_EINVAL;
  int64_t b = 0;
  for (int i = 0; i < nprob; ++i) b += splits * (problems[i].N * problems[i].K + problems[i].N) * (int64_t)sizeof(float);
  return b;
}

extern "C" int pvrl_gemm_tn_grouped_bf16(int nprob, const pvrl_tn_problem* problems, int64_t splits, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (!tn_group_ok(nprob, problems) || splits < 1 || !workspace) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_gemm_tn_grouped_workspace_bytes(nprob, problems, splits)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  TnGroup g = {};
  g.nprob = nprob;
  float* w = (float*)workspace;
  int first = 0;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = problems[i];
    GemmTN& p = g.prob[i];
    p.P = (const op_t*)q.P; p.ldp = q.ldp; p.Q = (const op_t*)q.Q; p.ldq = q.ldq;
    p.M = (int)q.M; p.N = (int)q.N; p.K = (int)q.K;
    p.Ms = cdiv(cdiv(q.M, splits), TM) * TM;
    p.part = w;
    w += splits * q.N * q.K;
    p.cpart = q.dbias ? w : nullptr;
    w += splits * q.N;
    p.zero_page = nullptr;
    p.tiles_k = (int)cdiv(q.K, 256);
    p.tiles_nk = (int)cdiv(q.N, 256) * p.tiles_k;
    p.npairs = (int)splits * p.tiles_nk;
    p.Ms_pairs = 0;
    g.first[i] = first;
    first += p.npairs;
  }
  g.first[nprob] = first;
  g.total = first;
  g.per_xcd = cdiv(first, 8);
  bool all8 = true;
  for (int i = 0; i < nprob; ++i) all8 = all8 && tn_use_tn8(problems[i].N, problems[i].K, g.prob[i].Ms, problems[i].ldp, problems[i].ldq);
  if (all8) hipLaunchKernelGGL(gemm_tn8_grouped_kernel, dim3((unsigned)(8 * g.per_xcd)), dim3(512), 0, s, g);
  else hipLaunchKernelGGL(gemm_tn_rt8_grouped_kernel, dim3((unsigned)(8 * g.per_xcd)), dim3(512), 0, s, g);
  PVRL_LAUNCH_CHECK();
  static_assert(TN_RED_MAX >= TN_GROUP_MAX, "reduce table too small");
  TnReduceGroup r = {};
  r.nprob = nprob; r.splits = (int)splits;
  int blocks = 0;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = problems[i];
    r.part[i] = g.prob[i].part; r.cpart[i] = g.prob[i].cpart; r.out[i] = q.dW; r.bias_out[i] = q.dbias;
    r.NK[i] = q.N * q.K; r.N[i] = (int)q.N; r.beta[i] = q.beta; r.gscale[i] = q.gscale; r.nonfinite[i] = q.nonfinite;
    r.first[i] = blocks;
    blocks += (int)cdiv((r.NK[i] >> 2) + (q.dbias ? q.N : 0), 256);
  }
  r.first[nprob] = blocks;
  hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, r);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
