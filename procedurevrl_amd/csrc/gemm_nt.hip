// bf16 MFMA GEMM, "NT" form:  C[M,N] = epilogue( A[M,K] . W[N,K]^T ), fp32 accumulate.
//
// Replaces the nn.Linear / F.linear calls of the reference hot path
// (lib/models/vit.py:54-60 Mlp, :75-92 Attention qkv/proj, :133 temporal_fc,
// :174-180 PatchEmbed conv-as-GEMM) and, fed with the transposed weight copy,
// their data-gradients.  One kernel family, fused epilogues:
//   bias, per-row scale (DropPath), exact-erf GELU / QuickGELU (+ pre-activation
//   kept for backward), fp32 residual add, GELU-derivative for the MLP backward.
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles), BK = 64, operands staged with
// 16-byte global_load_lds (LDS-DMA) into a double-buffered 2 x 32 KiB LDS ring.
// LDS tiles are [128 rows][64 bf16] (128-byte rows); the 16-byte chunk index is XOR-
// swizzled on the *global source* side (LDS-DMA writes lane-linear) and on the
// ds_read_b128 side with the same involution so that every ds_read_b128 lane group
// hits 16 distinct 16-byte slots.  Operands are swapped in the MFMA (a = W rows,
// b = A rows) and the W rows feeding tile nt are permuted (n = 16*q + 4*nt + r) so a
// lane ends up owning 16 consecutive output columns of one output row: the
// epilogue streams 16-/32-/64-byte contiguous pieces per lane.
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct GemmNT {
  const bf16* A; long lda;
  const bf16* W; long ldw;
  int M, N, K;
  const float* bias;      // [N] or null
  const float* rowscale;  // [M] or null
  const void* aux;        // fp32 residual [*, aux_ld] or bf16 pre-activation [M, aux_ld]
  long aux_ld; int aux_rowmod;
  void* out0; long ld0;
  void* out1; long ld1;
  int tiles_n, nwg;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int swz_x(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, p.nwg);
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses (per lane: 4 X chunks + 4 W chunks per K-step) ----
  const bf16* gx[4];
  const bf16* gw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int it = wave * 4 + j;
    const int row = it * 8 + (lane >> 3);
    const int pc = lane & 7;
    int grow = m0 + row;
    grow = grow < p.M ? grow : p.M - 1;
    gx[j] = p.A + (long)grow * p.lda + ((pc ^ swz_x(row)) << 3);
    gw[j] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w(row)) << 3);
  }
  auto stage = [&](int buf, int k0) {
    char* bx = smem + buf * 2 * TILE_BYTES;
    char* bw = bx + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = wave * 4 + j;
      glds16(gx[j] + k0, bx + it * 1024);
      glds16(gw[j] + k0, bw + it * 1024);
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile), ks = 0; ks = 1 is ^64 ----
  int xoff[4], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rx = wm * 64 + t * 16 + i;
      xoff[t] = rx * 128 + ((q ^ swz_x(rx)) << 4);
      const int rw = wn * 64 + 16 * (i >> 2) + 4 * t + (i & 3);
      woff[t] = rw * 128 + ((q ^ swz_w(rw)) << 4);
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and fences the previous compute
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* bx = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* bw = bx + TILE_BYTES;
    // all 16 fragment reads of the K-step are issued up front; the MFMAs of the first half overlap the
    // LDS latency of the second half (the compiler otherwise serialises read -> wait(0) -> 8 MFMAs)
    bf16x8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf0[t] = *reinterpret_cast<const bf16x8*>(bw + woff[t]);
      xf0[t] = *reinterpret_cast<const bf16x8*>(bx + xoff[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf1[t] = *reinterpret_cast<const bf16x8*>(bw + (woff[t] ^ 64));
      xf1[t] = *reinterpret_cast<const bf16x8*>(bx + (xoff[t] ^ 64));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
    // schedule: 8 reads, then one read per two MFMAs while the first half computes, then the rest of the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  }

  // ---- epilogue: lane owns row m, 16 consecutive columns nb..nb+15 (e = 4*nt + reg) ----
  const int q = lane >> 4, i = lane & 15;
  const int nb = n0 + wn * 64 + 16 * q;
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[nb + e] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + i;
    if (m >= p.M) continue;
    const float rs = p.rowscale ? p.rowscale[m] : 1.f;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[mt][nt][r] + bv[nt * 4 + r];

    if constexpr (EPI == PVRL_EPI_BF16) {
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o0[e] = (bf16)(rs * v[e]); o1[e] = (bf16)(rs * v[8 + e]); }
      bf16* o = (bf16*)p.out0 + (long)m * p.ld0 + nb;
      *reinterpret_cast<bf16x8*>(o) = o0;
      *reinterpret_cast<bf16x8*>(o + 8) = o1;
    } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
      bf16x8 u0, u1, g0, g1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = v[e], b = v[8 + e];
        u0[e] = (bf16)a; u1[e] = (bf16)b;
        g0[e] = (bf16)(EPI == PVRL_EPI_GELU ? gelu_erf(a) : quick_gelu(a));
        g1[e] = (bf16)(EPI == PVRL_EPI_GELU ? gelu_erf(b) : quick_gelu(b));
      }
      bf16* ou = (bf16*)p.out0 + (long)m * p.ld0 + nb;
      bf16* og = (bf16*)p.out1 + (long)m * p.ld1 + nb;
      *reinterpret_cast<bf16x8*>(ou) = u0; *reinterpret_cast<bf16x8*>(ou + 8) = u1;
      *reinterpret_cast<bf16x8*>(og) = g0; *reinterpret_cast<bf16x8*>(og + 8) = g1;
    } else if constexpr (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32) {
      float* o = (float*)p.out0 + (long)m * p.ld0 + nb;
      if constexpr (EPI == PVRL_EPI_RESID_F32) {
        const int mr = p.aux_rowmod ? (m % p.aux_rowmod) : m;
        const float* r = (const float*)p.aux + (long)mr * p.aux_ld + nb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 rv = *reinterpret_cast<const f32x4*>(r + 4 * c);
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = rv[e] + rs * v[4 * c + e];
          *reinterpret_cast<f32x4*>(o + 4 * c) = ov;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = rs * v[4 * c + e];
          *reinterpret_cast<f32x4*>(o + 4 * c) = ov;
        }
      }
    } else {  // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
      const bf16* u = (const bf16*)p.aux + (long)m * p.aux_ld + nb;
      const bf16x8 ua = *reinterpret_cast<const bf16x8*>(u);
      const bf16x8 ub = *reinterpret_cast<const bf16x8*>(u + 8);
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float da = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
        const float db = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ub[e]) : quick_gelu_grad((float)ub[e]);
        o0[e] = (bf16)(rs * v[e] * da);
        o1[e] = (bf16)(rs * v[8 + e] * db);
      }
      bf16* o = (bf16*)p.out0 + (long)m * p.ld0 + nb;
      *reinterpret_cast<bf16x8*>(o) = o0;
      *reinterpret_cast<bf16x8*>(o + 8) = o1;
    }
  }
}

// ---------------------------------------------------------------------------
// Small fp32 GEMM for the projection head / step-logit path, where M is a few
// dozen rows and the reference keeps fp32 (lib/models/vit.py:299-307):
//   C[M,N] = alpha * (A[M,K] . B[N,K]^T) + bias[N]        (all fp32)
// HBM-bound on B (label_emb: 9871 x 512 fp32 = 20 MB): each workgroup streams 16
// rows of B once, coalesced, against up to 64 rows of A held in LDS.
// ---------------------------------------------------------------------------
constexpr int SG_NB = 16, SG_MB = 64, SG_KB = 128;
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             const float* __restrict__ bias, float alpha,
                                                             float* __restrict__ C, long ldc, int M, int N, int K) {
  __shared__ float sa[SG_MB][SG_KB + 1];
  __shared__ float sb[SG_NB][SG_KB + 1];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * SG_NB, m0 = blockIdx.y * SG_MB;
  const int tn = tid & 15, tmr = tid >> 4;  // thread: column n0+tn, rows m0 + tmr + 16*r
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += SG_KB) {
    for (int idx = tid; idx < SG_MB * SG_KB; idx += 256) {
      const int r = idx / SG_KB, c = idx - r * SG_KB;
      const int m = m0 + r, k = k0 + c;
      sa[r][c] = (m < M && k < K) ? A[(long)m * lda + k] : 0.f;
    }
    for (int idx = tid; idx < SG_NB * SG_KB; idx += 256) {
      const int r = idx / SG_KB, c = idx - r * SG_KB;
      const int n = n0 + r, k = k0 + c;
      sb[r][c] = (n < N && k < K) ? B[(long)n * ldb + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < SG_KB; ++c) {
      const float b = sb[tn][c];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(sa[tmr + 16 * r][c], b, acc[r]);
    }
    __syncthreads();
  }
  const int n = n0 + tn;
  if (n < N) {
    const float bb = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + tmr + 16 * r;
      if (m < M) C[(long)m * ldc + n] = alpha * acc[r] + bb;
    }
  }
}

template <int EPI>
int launch_nt(const GemmNT& p, hipStream_t s) {
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(p.nwg), dim3(256), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace

extern "C" int pvrl_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
                                 int64_t K, int epilogue, const float* bias, const float* rowscale,
                                 const void* aux, int64_t aux_ld, int64_t aux_rowmod, void* out0, int64_t ld0,
                                 void* out1, int64_t ld1, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!A || !W || !out0 || N <= 0 || K <= 0 || (N % BN) || (K % BK)) return PVRL_EINVAL;
  if ((lda % 8) || (ldw % 8) || (ld0 % 8)) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_GELU || epilogue == PVRL_EPI_QGELU) && (!out1 || (ld1 % 8))) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_RESID_F32 || epilogue == PVRL_EPI_DGELU || epilogue == PVRL_EPI_DQGELU) &&
      (!aux || (aux_ld % 8)))
    return PVRL_EINVAL;
  GemmNT p;
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = bias; p.rowscale = rowscale; p.aux = aux; p.aux_ld = aux_ld; p.aux_rowmod = (int)aux_rowmod;
  p.out0 = out0; p.ld0 = ld0; p.out1 = out1; p.ld1 = ld1;
  p.tiles_n = (int)(N / BN);
  p.nwg = cdiv(M, BM) * p.tiles_n;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case PVRL_EPI_BF16: return launch_nt<PVRL_EPI_BF16>(p, s);
    case PVRL_EPI_GELU: return launch_nt<PVRL_EPI_GELU>(p, s);
    case PVRL_EPI_QGELU: return launch_nt<PVRL_EPI_QGELU>(p, s);
    case PVRL_EPI_RESID_F32: return launch_nt<PVRL_EPI_RESID_F32>(p, s);
    case PVRL_EPI_F32: return launch_nt<PVRL_EPI_F32>(p, s);
    case PVRL_EPI_DGELU: return launch_nt<PVRL_EPI_DGELU>(p, s);
    case PVRL_EPI_DQGELU: return launch_nt<PVRL_EPI_DQGELU>(p, s);
    default: return PVRL_EINVAL;
  }
}

extern "C" int pvrl_gemm_nt_f32_small(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                      float alpha, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      void* stream) {
  if (M <= 0 || N <= 0) return PVRL_OK;
  if (!A || !B || !C || K <= 0) return PVRL_EINVAL;
  dim3 grid(cdiv(N, SG_NB), cdiv(M, SG_MB));
  hipLaunchKernelGGL(gemm_f32_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, (long)lda, B, (long)ldb,
                     bias, alpha, C, (long)ldc, (int)M, (int)N, (int)K);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
