// 16-bit MFMA GEMM, "NT" form, FEW rows:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )  with M <= 192.
//
// The order / diffusion transformer of the pre-training head (lib/models/tfm_model.py:129-204) runs its four-layer stack on the 9
// clip embeddings of 4 videos -- 36 rows per denoise level, 144 in the batched backward.  A 128x128-tile workgroup then walks the
// whole reduction alone (K = 2048: 32 stages) and a launch takes 12-20 us whatever the 0.04-0.6 GFLOP inside it: 76 such launches
// were 1.0 ms of the full pre-training step (profiles/r5_timeline_full.txt).  Here one workgroup owns 16 output columns, its EIGHT
// waves split the reduction (K / 8 each, straight from global memory / L2: no LDS staging, the 16 x K slice of W is read once per
// 48-row pass), the partial tiles are summed through LDS in wave order (deterministic) and every epilogue of gemm_nt_core.h is
// applied element by element: N / 16 workgroups of 2-8 MFMA steps each, ~5 us.
// Same arithmetic as gemm_nt_kernel up to the order of the fp32 sums over k.
#pragma once
#include "gemm_nt_core.h"

namespace {

constexpr int SK_MT = 3;          // 16-row tiles per pass (48 rows); more rows: blockIdx.y passes
constexpr int SK_NW = 8;
constexpr int SK_MAX_M = 192;

template <int EPI>
__global__ __launch_bounds__(64 * SK_NW) void gemm_nt_skinny_kernel(GemmNT p) {
  __shared__ float part[SK_NW][SK_MT][4][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * SK_MT);
  const int mt_n = min(SK_MT, (p.M - m0 + 15) >> 4);
  const int kw = p.K / SK_NW, kbeg = wave * kw;
  const op_t* wrow = p.W + (long)(n0 + r) * p.ldw + kbeg + q * 8;
  const op_t* xrow[SK_MT];
#pragma unroll
  for (int t = 0; t < SK_MT; ++t) xrow[t] = p.A + (long)min(m0 + t * 16 + r, p.M - 1) * p.lda + kbeg + q * 8;
  f32x4 acc[SK_MT];
#pragma unroll
  for (int t = 0; t < SK_MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // lane (r, q) holds k = kk + 8 q .. + 7 of row r of both operands (rows clamped: rows >= M are computed and never stored)
#pragma unroll 4
  for (int kk = 0; kk < kw; kk += 32) {
    const opx8 wf = *reinterpret_cast<const opx8*>(wrow + kk);
    opx8 xf[SK_MT];
#pragma unroll
    for (int t = 0; t < SK_MT; ++t) xf[t] = *reinterpret_cast<const opx8*>(xrow[t] + kk);
#pragma unroll
    for (int t = 0; t < SK_MT; ++t) acc[t] = MFMA_16x16x32(xf[t], wf, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < SK_MT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) part[wave][t][i][lane] = acc[t][i];
  __syncthreads();
  // element e = (tile t, i, lane l): row m0 + 16 t + 4 (l / 16) + i, column n0 + l % 16
  for (int e = tid; e < mt_n * 256; e += 64 * SK_NW) {
    const int t = e >> 8, i = (e >> 6) & 3, l = e & 63;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SK_NW; ++w) s += part[w][t][i][l];
    const int m = m0 + t * 16 + 4 * (l >> 4) + i, n = n0 + (l & 15);
    if (m >= p.M) continue;
    const float rs = p.rowscale ? p.rowscale[m] : 1.f;
    float b = p.bias ? p.bias[n] : 0.f;
    if constexpr (EPI == PVRL_EPI_RESID_F32) {
      if (p.bias2 && !p.rowscale) b += p.bias2[n];
      const long ar = p.aux_rowmod ? (long)((m + p.m_off) % p.aux_rowmod) : (long)m;
      float v = rs * (s + b) + reinterpret_cast<const float*>(p.aux)[ar * p.aux_ld + n];
      if (p.bias2 && p.rowscale) v += p.bias2[n];
      reinterpret_cast<float*>(p.out0)[(long)m * p.ld0 + n] = v;
    } else if constexpr (EPI == PVRL_EPI_RESID_16) {
      if (p.bias2 && !p.rowscale) b += p.bias2[n];
      const float a = p.aux_rowmod ? reinterpret_cast<const float*>(p.aux)[(long)((m + p.m_off) % p.aux_rowmod) * p.aux_ld + n]
                                   : (float)reinterpret_cast<const op_t*>(p.aux)[(long)m * p.aux_ld + n];
      float v = rs * (s + b) + a;
      if (p.bias2 && p.rowscale) v += p.bias2[n];
      reinterpret_cast<op_t*>(p.out0)[(long)m * p.ld0 + n] = (op_t)v;
    } else if constexpr (EPI == PVRL_EPI_F32) {
      reinterpret_cast<float*>(p.out0)[(long)m * p.ld0 + n] = rs * (s + b);
    } else if constexpr (EPI == PVRL_EPI_BF16) {
      reinterpret_cast<op_t*>(p.out0)[(long)m * p.ld0 + n] = (op_t)(rs * (s + b));
    } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
      const float u = s + b;
      reinterpret_cast<op_t*>(p.out0)[(long)m * p.ld0 + n] = (op_t)u;
      reinterpret_cast<op_t*>(p.out1)[(long)m * p.ld1 + n] =
          (op_t)(EPI == PVRL_EPI_GELU ? gelu_erf2((f32x2_t){u, u})[0] : quick_gelu(u));
    } else {     // PVRL_EPI_DGELU / PVRL_EPI_DQGELU
      const float ua = (float)reinterpret_cast<const op_t*>(p.aux)[(long)m * p.aux_ld + n];
      const float d = EPI == PVRL_EPI_DGELU ? gelu_erf_grad(ua) : quick_gelu_grad(ua);
      reinterpret_cast<op_t*>(p.out0)[(long)m * p.ld0 + n] = (op_t)(rs * (s + b) * d);
    }
  }
}

inline bool nt_skinny_ok(const GemmNT& p) { return p.M <= SK_MAX_M && p.K % (32 * SK_NW) == 0 && p.N % 16 == 0; }

template <int EPI>
int launch_nt_skinny(const GemmNT& p, hipStream_t s) {
  hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI>), dim3((unsigned)(p.N / 16), (unsigned)cdiv(p.M, 16 * SK_MT)), dim3(64 * SK_NW), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace
