// bf16 MFMA GEMM, "NT" form: the C-ABI entry points.  The kernel, its epilogues and the tile launcher live in
// gemm_nt_core.h (shared with the measured-and-rejected variants kept under tools/probe/, which are NOT part of this library).
#include <cstdlib>
#include "gemm_nt_core.h"
#include "gemm_nt8_core.h"
#include "gemm_nt_skinny.h"

// cls_chain.hip: the cls rows' fp32 MFMA kernel, also used for the K % 128 == 0 shapes of pvrl_gemm_nt_f32_small
bool pvrl_cls_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float alpha, float* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, hipStream_t s, int* status);

namespace {

// ---------------------------------------------------------------------------
// Small fp32 GEMM for the projection head / step-logit path, where M is a few
// dozen rows and the reference keeps fp32 (lib/models/vit.py:299-307):
//   C[M,N] = alpha * (A[M,K] . B[N,K]^T) + bias[N]        (all fp32)
// HBM-bound on B (label_emb: 9871 x 512 fp32 = 20 MB): each workgroup streams 16
// rows of B once, coalesced, against up to 64 rows of A held in LDS.
// ---------------------------------------------------------------------------
// 64x64 output tile, 32-deep K chunks, 4x4 outputs per thread from k-major LDS tiles (float4 reads); optional split-K
// over grid.z (the logits backward has M = 32, N = 512, K = 9871: 8 output tiles only) into fp32 partials that a second
// kernel sums in a fixed order -- no atomics, so two runs of a training step are bit-identical.
constexpr int SG_T = 64, SG_K = 32, SG_LD = 68;
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             const float* __restrict__ bias, float alpha,
                                                             float* __restrict__ C, long ldc, int M, int N, int K,
                                                             int kchunk, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float sa[SG_K][SG_LD];
  __shared__ __attribute__((aligned(16))) float sb[SG_K][SG_LD];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * SG_T, m0 = blockIdx.y * SG_T;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const int tx = tid & 15, ty = tid >> 4;
  const int lrow = tid >> 2, lk = (tid & 3) * 8;      // loader: one tile row, 8 consecutive k
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += SG_K) {
    // sixteen loads per thread, UNCONDITIONAL from clamped indices and masked by selects: inside `cond ? load : 0` each one is
    // followed by s_waitcnt vmcnt(0) (16 serial round trips per 32-deep K chunk, round 3 audit)
    float va[8], vb[8];
    const float* ar = A + (long)min(m0 + lrow, M - 1) * lda;
    const float* br = B + (long)min(n0 + lrow, N - 1) * ldb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = min(k0 + lk + e, K - 1);
      va[e] = ar[k];
      vb[e] = br[k];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool kin = k0 + lk + e < kend;
      va[e] = (m0 + lrow < M && kin) ? va[e] : 0.f;
      vb[e] = (n0 + lrow < N && kin) ? vb[e] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[lk + e][lrow] = va[e]; sb[lk + e][lrow] = vb[e]; }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SG_K; ++k) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sa[k][ty * 4]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb[k][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      if (split) part[((long)blockIdx.z * M + m) * N + n] = acc[i][j];      // deterministic: partials + ordered reduce
      else C[(long)m * ldc + n] = alpha * acc[i][j] + (bias ? bias[n] : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void f32_small_reduce_kernel(const float* __restrict__ part, int splits, long MN, int N,
                                                               const float* __restrict__ bias, float alpha,
                                                               float* __restrict__ C, long ldc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
    float a = 0.f;
    for (int z = 0; z < splits; ++z) a += part[(long)z * MN + i];
    const long m = i / N;
    const int n = (int)(i - m * N);
    C[m * ldc + n] = alpha * a + (bias ? bias[n] : 0.f);
  }
}

static int f32_small_plan(int64_t M, int64_t N, int64_t K, int* kchunk_out) {
  const int tiles = cdiv(N, SG_T) * cdiv(M, SG_T);
  int splits = 1;
  if (tiles < 128 && K >= 512) {           // few output tiles and a long reduction: split K over grid.z
    splits = 256 / tiles;
    const int maxs = (int)(K / 128);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  const int kchunk = cdiv(cdiv(K, splits), SG_K) * SG_K;
  if (kchunk_out) *kchunk_out = kchunk;
  return cdiv(K, kchunk);
}

// Tile selection: 256x256 / 16 waves for the encoder's 50k-row GEMMs (M >= 4096 and N a multiple of 256), 256x128 for
// mid-sized M, 128x128 / 4 waves (two workgroups per CU) for the small-M stacks (order transformer, CLIP text) and for
// every N that is not a multiple of 256 (MViT's 128 / 384 / 640 / 1152-wide layers: 128x128 measured 5-16 % ahead of
// 256x128 on 12 of the 14 such shapes of an MViTv2-S step, tools/probe/mvit_gemm_tiles.py; the two-output GELU
// epilogue is the exception).
// (Cutting the ragged last wave of 256x256 tiles off into a 128x128-tile launch was measured 12 % SLOWER: the second
//  launch serialises behind the first; one launch with a partly idle last wave wins.)
// CUs per XCD of the current device (MI355X: 256 / 8 = 32) -- sizes the last-round split (gemm_nt_core.h nt_tail_plan)
int nt_cus_per_xcd() {
  static int cus = 0;
  if (cus == 0) {
    cus = pvrl_compute_cus_per_xcd();           // (common.h: all CUs of an XCD, or PVRL_COMPUTE_CUS of them)
    const char* e = getenv("PVRL_NT_CUS");      // probe runs: fewer persistent workgroups per XCD for this kernel family only
    if (e && atoi(e) > 0 && atoi(e) < cus) cus = atoi(e);
  }
  return cus;
}
#ifndef PVRL_NT_TAILS_DEFAULT
#define PVRL_NT_TAILS_DEFAULT 1
#endif
// PVRL_NT_TAILS=0 switches the sub-tiling of the ragged last round off (A/B runs; read once)
int nt_tails_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_NT_TAILS");
    on = e ? (e[0] == '0' ? 0 : 1) : PVRL_NT_TAILS_DEFAULT;
  }
  return on;
}

int nt_wide_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_NT_WIDE");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

// PVRL_NT8=0 sends the 256x256 shapes back to the 16-wave one-tile kernel (A/B runs, tools/bench_kernels.py); read once
int nt8_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_NT8");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

// rows from which the mid-sized shapes take 256 x 128 tiles instead of 128 x 128 (PVRL_NT_MID_M: A/B runs; read once)
int nt_mid_m() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("PVRL_NT_MID_M");
    m = e && atoi(e) > 0 ? atoi(e) : 2048;
  }
  return m;
}

// PVRL_NT_TILE=22|42|44|26|25 forces a tile shape where it is legal for the problem (shape sweeps: tools/probe/mvit_gemm_times.py); read once
int nt_forced_tile() {
  static int t = -1;
  if (t < 0) {
    const char* e = getenv("PVRL_NT_TILE");
    t = e ? atoi(e) : 0;
  }
  return t;
}

// PVRL_NT_SKINNY=0 sends the few-row problems (M <= 192: the pre-training head's stack) back to the 128 x 128 tile (A/B runs; read once)
int nt_skinny_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_NT_SKINNY");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

template <int EPI>
int launch_nt(const GemmNT& p, hipStream_t s) {
  constexpr bool two_out = EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU;
  if (nt_skinny_ok(p) && nt_forced_tile() == 0 && nt_skinny_enabled()) return launch_nt_skinny<EPI>(p, s);
  switch (nt_forced_tile()) {
    case 22: return launch_tile<EPI, 2, 2>(p, s);
    case 42: return launch_tile<EPI, 4, 2>(p, s);
    case 44: if (p.N % 256 == 0) return launch_tile<EPI, 4, 4>(p, s); break;
    case 26: if (p.N % 384 == 0) return launch_tile<EPI, 2, 6>(p, s); break;
    case 25: if (p.N % 320 == 0) return launch_tile<EPI, 2, 5>(p, s); break;
    default: break;
  }
  if (nt_wide_enabled() && p.N == 768 && p.M >= 4096 && p.M < 20000) return launch_tile<EPI, 2, 6>(p, s);   // MViT stage 4 (M = 12,576): 99 tiles of 128 x 384 x 2 fill 198 CUs; 256 x 256 tiles 150 (-10 %)
  if (p.M >= 4096 && p.N % 256 == 0) {
    // persistent 8-wave ping-pong kernel (gemm_nt8_core.h); its load stream runs two K-tiles ahead, so K >= 128
    // (the fp32-table form of PVRL_EPI_RESID_16 -- the embedding prologue, one launch per step -- lives in the one-tile kernel only)
    if (nt8_enabled() && p.K >= 2 * BK && !(EPI == PVRL_EPI_RESID_16 && p.aux_rowmod != 0)) return launch_nt8<EPI>(p, s);
    return launch_tile<EPI, 4, 4>(p, s);
  }
  // N = 384 / 1152 and 640 at M >= 100k rows (MViTv2-S stages 1-2): one 128 x 384 / 128 x 320 tile row instead of three / five 128 x 128
  // column tiles -- the A panel is fetched once and a workgroup's fixed costs cover 3x / 2.5x the output (8-23 % per shape; at
  // M = 50,208 the 128 x 128 tiles' two workgroups per CU win by 5-9 %: gpurun_out/r3_w_shapes_*.txt).  PVRL_NT_WIDE=0: A/B runs
  if (nt_wide_enabled() && p.M >= 4096 && p.N % 256 != 0) {
    if (p.N % 384 == 0 && p.M >= 100000) return launch_tile<EPI, 2, 6>(p, s);
    if (p.N % 320 == 0) return launch_tile<EPI, 2, 5>(p, s);
  }
  if (p.M >= nt_mid_m() && (p.N % 256 == 0 || two_out)) return launch_tile<EPI, 4, 2>(p, s);
  return launch_tile<EPI, 2, 2>(p, s);
}

}  // namespace

extern "C" int pvrl_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
                                 int64_t K, int epilogue, const float* bias, const float* rowscale,
                                 const void* aux, int64_t aux_ld, int64_t aux_rowmod, void* out0, int64_t ld0,
                                 void* out1, int64_t ld1, const float* bias2, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (bias2 && epilogue != PVRL_EPI_RESID_F32 && epilogue != PVRL_EPI_RESID_16) return PVRL_EINVAL;
  if (!A || !W || !out0 || N <= 0 || K <= 0 || (N % 128) || (K % BK)) return PVRL_EINVAL;
  if ((lda % 8) || (ldw % 8) || (ld0 % 8)) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_GELU || epilogue == PVRL_EPI_QGELU) && (!out1 || (ld1 % 8))) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_RESID_F32 || epilogue == PVRL_EPI_RESID_16 || epilogue == PVRL_EPI_DGELU || epilogue == PVRL_EPI_DQGELU) &&
      (!aux || (aux_ld % 8)))
    return PVRL_EINVAL;
  GemmNT p;
  p.A = (const op_t*)A; p.lda = lda; p.W = (const op_t*)W; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = bias; p.bias2 = bias2; p.rowscale = rowscale; p.aux = aux; p.aux_ld = aux_ld; p.aux_rowmod = (int)aux_rowmod;
  p.out0 = out0; p.ld0 = ld0; p.out1 = out1; p.ld1 = ld1; p.m_off = 0; p.gm = 0;      // 0: launch_tile picks the rasterisation group height for the shape
  p.cus = nt_cus_per_xcd(); p.tails = nt_tails_enabled();
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case PVRL_EPI_BF16: return launch_nt<PVRL_EPI_BF16>(p, s);
    case PVRL_EPI_GELU: return launch_nt<PVRL_EPI_GELU>(p, s);
    case PVRL_EPI_QGELU: return launch_nt<PVRL_EPI_QGELU>(p, s);
    case PVRL_EPI_RESID_F32: return launch_nt<PVRL_EPI_RESID_F32>(p, s);
    case PVRL_EPI_F32: return launch_nt<PVRL_EPI_F32>(p, s);
    case PVRL_EPI_DGELU: return launch_nt<PVRL_EPI_DGELU>(p, s);
    case PVRL_EPI_DQGELU: return launch_nt<PVRL_EPI_DQGELU>(p, s);
    case PVRL_EPI_RESID_16: return launch_nt<PVRL_EPI_RESID_16>(p, s);
    default: return PVRL_EINVAL;
  }
}

extern "C" int pvrl_gemm_nt_batched_bf16(int nprob, const pvrl_nt_problem* problems, int epilogue, void* stream) {
  if (nprob <= 0) return PVRL_OK;
  if (!problems || (epilogue != PVRL_EPI_BF16 && epilogue != PVRL_EPI_F32 && epilogue != PVRL_EPI_RESID_F32)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int i0 = 0; i0 < nprob; i0 += NT_BATCH_MAX) {
    GemmNTBatch g = {};
    g.n = nprob - i0 < NT_BATCH_MAX ? nprob - i0 : NT_BATCH_MAX;
    int blocks = 0;
    for (int i = 0; i < g.n; ++i) {
      const pvrl_nt_problem& q = problems[i0 + i];
      if (!q.A || !q.W || !q.out0 || q.M <= 0 || q.N <= 0 || q.K <= 0 || (q.N % 128) || (q.K % BK)) return PVRL_EINVAL;
      if ((q.lda % 8) || (q.ldw % 8) || (q.ld0 % 8) || (epilogue == PVRL_EPI_RESID_F32 && (!q.aux || (q.aux_ld % 8)))) return PVRL_EINVAL;
      GemmNT& p = g.prob[i];
      p.A = (const op_t*)q.A; p.lda = q.lda; p.W = (const op_t*)q.W; p.ldw = q.ldw;
      p.M = (int)q.M; p.N = (int)q.N; p.K = (int)q.K;
      p.bias = q.bias; p.bias2 = nullptr; p.rowscale = q.rowscale; p.aux = q.aux; p.aux_ld = q.aux_ld; p.aux_rowmod = 0;
      p.out0 = q.out0; p.ld0 = q.ld0; p.out1 = nullptr; p.ld1 = 0; p.m_off = 0;
      p.cus = nt_cus_per_xcd(); p.tails = 0;
      p.tiles_n = p.N / 128; p.tiles_m = cdiv(p.M, 128); p.gm = nt_gm_for(p.tiles_n);
      p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
      g.first[i] = blocks;
      blocks += p.nwg;
    }
    g.first[g.n] = blocks;
    switch (epilogue) {
      case PVRL_EPI_BF16: hipLaunchKernelGGL(gemm_nt_batched_kernel<PVRL_EPI_BF16>, dim3(blocks), dim3(256), 0, s, g); break;
      case PVRL_EPI_F32: hipLaunchKernelGGL(gemm_nt_batched_kernel<PVRL_EPI_F32>, dim3(blocks), dim3(256), 0, s, g); break;
      default: hipLaunchKernelGGL(gemm_nt_batched_kernel<PVRL_EPI_RESID_F32>, dim3(blocks), dim3(256), 0, s, g); break;
    }
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

// PVRL_F32_SMALL_MFMA=0 keeps every shape on the 64 x 64-tile FMA kernel (A/B runs; read once)
static int f32_small_mfma_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_F32_SMALL_MFMA");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

extern "C" int64_t pvrl_gemm_nt_f32_small_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int splits = f32_small_plan(M, N, K, nullptr);
  return splits > 1 ? (int64_t)splits * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int pvrl_gemm_nt_f32_small(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                      float alpha, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0) return PVRL_OK;
  if (!A || !B || !C || K <= 0) return PVRL_EINVAL;
  if (f32_small_mfma_enabled()) {       // K % 128 == 0: one workgroup per 16 columns of B, fp32 MFMA (cls_chain.hip)
    int st = PVRL_OK;
    if (pvrl_cls_gemm_f32(A, lda, B, ldb, bias, alpha, C, ldc, M, N, K, (hipStream_t)stream, &st)) return st;
  }
  int kchunk;
  const int splits = f32_small_plan(M, N, K, &kchunk);
  if (splits > 1 && (!workspace || workspace_bytes < pvrl_gemm_nt_f32_small_workspace_bytes(M, N, K))) return PVRL_EINVAL;
  dim3 grid(cdiv(N, SG_T), cdiv(M, SG_T), splits);
  hipLaunchKernelGGL(gemm_f32_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, (long)lda, B, (long)ldb,
                     bias, alpha, C, (long)ldc, (int)M, (int)N, (int)K, kchunk, (float*)workspace);
  PVRL_LAUNCH_CHECK();
  if (splits > 1) {
    const long MN = (long)M * N;
    hipLaunchKernelGGL(f32_small_reduce_kernel, dim3((unsigned)cdiv(MN, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, splits, MN, (int)N, bias, alpha, C, (long)ldc);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}
