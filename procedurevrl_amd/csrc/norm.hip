// LayerNorm forward / backward over fp32 residual-stream rows, one 64-lane wave per row.
// Reference: nn.LayerNorm(eps=1e-6) in the TimeSformer blocks (lib/models/vit.py:104,109,116,
// 228, ctor :488) and the fp32 LayerNorm subclass of the order / CLIP transformer
// (lib/models/tfm_model.py:18-24, eps=1e-5).
//
// HBM-bound.  Forward reads fp32 x once (16 B per lane per load) and writes the bf16 GEMM
// operand (or fp32 for the projection head) plus per-row mean / rstd.  Backward fuses the
// residual-gradient add (dx_out = dx_in + dLN) and accumulates dgamma / dbeta in registers
// across a grid-stride loop; per-block partials are summed by a second tiny kernel
// (deterministic, no atomics).
#include "common.h"
#include "../../include/pvrl.h"

namespace {

template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4IO<op_t> {
  static __device__ __forceinline__ f32x4 load(const op_t* p) {
    const opx4 b = *reinterpret_cast<const opx4*>(p);
    return (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  static __device__ __forceinline__ void store(op_t* p, f32x4 v) {
    opx4 b;
    b[0] = (op_t)v[0]; b[1] = (op_t)v[1]; b[2] = (op_t)v[2]; b[3] = (op_t)v[3];
    *reinterpret_cast<opx4*>(p) = b;
  }
};

template <int C, typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, TOut* __restrict__ y,
                                                     long ldy, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int M) {
  constexpr int NV = C / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = *reinterpret_cast<const f32x4*>(xr + 4 * lane + 256 * j);
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mu = wave_sum(s) * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; ss += d * d; }
  const float rs = rsqrtf(wave_sum(ss) * (1.0f / C) + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  TOut* yr = y + (long)row * ldy;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * lane + 256 * j;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * g[e] + b[e];
    Vec4IO<TOut>::store(yr + c, o);
  }
}

template <int C, typename TDy>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                     long ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ dx_in, long ldi, float* __restrict__ dx_out,
                                                     long ldo, float* __restrict__ part, int M, op_t* __restrict__ dxs,
                                                     long ldxs, const float* __restrict__ dxs_scale, int dxs_rows,
                                                     int want_sum) {
  constexpr int NV = C / 256;
  __shared__ float red[4][2][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pstride = (2 + want_sum) * C;       // floats per workgroup in `part`: dgamma | dbeta | (column sums of dx_out)
  f32x4 g[NV], dg[NV], db[NV], ds[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    g[j] = *reinterpret_cast<const f32x4*>(gamma + 4 * lane + 256 * j);
    dg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ds[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NV], gy[NV], din[NV];
    float c1 = 0.f, c2 = 0.f;
    // the incoming residual gradient is not needed before the row reductions, but its load goes out WITH x and dy: one
    // memory round trip per row instead of two
#pragma unroll
    for (int j = 0; j < NV; ++j)
      din[j] = dx_in ? *reinterpret_cast<const f32x4*>(dx_in + (long)row * ldi + 4 * lane + 256 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * lane + 256 * j;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + c);
      const f32x4 d = Vec4IO<TDy>::load(dy + (long)row * lddy + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[j][e] = (xv[e] - mu) * rs;
        gy[j][e] = d[e] * g[j][e];
        c1 += gy[j][e];
        c2 += gy[j][e] * xh[j][e];
        dg[j][e] += d[e] * xh[j][e];
        db[j][e] += d[e];
      }
    }
    c1 = wave_sum(c1) * (1.0f / C);
    c2 = wave_sum(c2) * (1.0f / C);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * lane + 256 * j;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (gy[j][e] - c1 - xh[j][e] * c2);
      o += din[j];
      *reinterpret_cast<f32x4*>(dx_out + (long)row * ldo + c) = o;
      if (want_sum && row < dxs_rows) ds[j] += o;      // unscaled column sums of the rows that feed the next stage
      if (dxs && row < dxs_rows) {   // bf16 (optionally DropPath-scaled) copy: the GEMM operand of the next backward stage
        const float sc = dxs_scale ? dxs_scale[row] : 1.f;
        Vec4IO<op_t>::store(dxs + (long)row * ldxs + c, sc * o);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * lane + 256 * j;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][0][c + e] = dg[j][e];
      red[wave][1][c + e] = db[j][e];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
    const int w = idx / C, c = idx - w * C;
    part[(long)blockIdx.x * pstride + idx] = red[0][w][c] + red[1][w][c] + red[2][w][c] + red[3][w][c];
  }
  if (want_sum) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * lane + 256 * j;
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][0][c + e] = ds[j][e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      part[(long)blockIdx.x * pstride + 2 * C + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
  }
}

// out[j] = beta*out[j] + gscale * sum_b part[b][j]   (j over 2*C: dgamma then dbeta); gscale / nonfinite as in gemm_tn_core.h: the
// scale of the fp16 flavour's S-scaled backward is taken out where the parameter gradient is written, and a non-finite value
// written raises the optimiser's skip flag
// 16 columns x 16 row groups per block (64-byte row segments), LDS tree at the end: 96 blocks for C = 768.
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, int nblk, int n, float beta,
                                                             float* __restrict__ out0, float* __restrict__ out1,
                                                             int half, int pstride, const float* __restrict__ gscale,
                                                             float* __restrict__ nonfinite) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float a0 = 0.f, a1 = 0.f;
  if (j < n) {
    int b = g;
    for (; b + 16 < nblk; b += 32) { a0 += part[(long)b * pstride + j]; a1 += part[(long)(b + 16) * pstride + j]; }
    if (b < nblk) a0 += part[(long)b * pstride + j];
  }
  red[g][c] = a0 + a1;
  __syncthreads();
  if (g == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    float* o = (j < half) ? out0 + j : out1 + (j - half);
    if (gscale) t *= *gscale;
    const float v = (beta != 0.f ? beta * *o : 0.f) + t;
    *o = v;
    if (nonfinite && (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u) *nonfinite = 1.f;
  }
}

// The same for MANY LayerNorms in one launch: a backward of the encoder has 37 of them, each followed by this 7-us reduce of its
// per-workgroup partials (dgamma | dbeta | optional column sums) -- 49 launches of a few microseconds on an otherwise idle chip.
// When nobody needs a block's gradients before the end of the backward, pvrl_layernorm_bwd leaves the partials in their own
// workspaces (dgamma = null) and ONE launch reduces them all.
constexpr int LN_RED_MAX = 40;
struct LnReduceItem { const float* part; int nblk, n, C, pstride; float beta, beta_sum; float* dgamma; float* dbeta; float* dxsum; int first; };
struct LnReduceBatch { int n; LnReduceItem it[LN_RED_MAX]; const float* gscale; float* nonfinite; };
__global__ __launch_bounds__(256) void colpart_reduce_batched_kernel(LnReduceBatch g) {
  __shared__ float red[16][17];
  int q = 0;
  for (int t = 1; t < g.n; ++t)
    if ((int)blockIdx.x >= g.it[t].first) q = t;
  const LnReduceItem w = g.it[q];
  const int c = threadIdx.x & 15, gr = threadIdx.x >> 4;
  const int j = ((int)blockIdx.x - w.first) * 16 + c;
  float a0 = 0.f, a1 = 0.f;
  if (j < w.n) {
    int b = gr;
    for (; b + 16 < w.nblk; b += 32) { a0 += w.part[(long)b * w.pstride + j]; a1 += w.part[(long)(b + 16) * w.pstride + j]; }
    if (b < w.nblk) a0 += w.part[(long)b * w.pstride + j];
  }
  red[gr][c] = a0 + a1;
  __syncthreads();
  if (gr == 0 && j < w.n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    float* o = j < w.C ? w.dgamma + j : (j < 2 * w.C ? w.dbeta + (j - w.C) : w.dxsum + (j - 2 * w.C));
    const float beta = j < 2 * w.C ? w.beta : w.beta_sum;
    if (g.gscale) t *= *g.gscale;
    const float v = (beta != 0.f ? beta * *o : 0.f) + t;
    *o = v;
    if (g.nonfinite && (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u) *g.nonfinite = 1.f;
  }
}

// measured on MI355X (M = 50,208): alone, 256 workgroups are fastest (94 us; 512: 99, 1024: 111, 2048: 117 -- fewer partial
// sums to write and reduce); inside the training step, next to the weight-gradient GEMMs of the side stream, 512 win (572 vs
// 561 clips/s): one workgroup per CU is starved by the co-running kernel.
constexpr int LN_BWD_MAX_BLOCKS = 512;

}  // namespace

extern "C" int pvrl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                  void* y, int64_t ldy, int out_is_f32, float* mean, float* rstd, int64_t M,
                                  int64_t C, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!x || !gamma || !beta || !y || (ldx % 4) || (ldy % 4)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(M, 4)), blk(256);
#define LN_FWD(CC)                                                                                                  \
  if (out_is_f32)                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_kernel<CC, float>), grid, blk, 0, s, x, (long)ldx, gamma, beta, eps, (float*)y,       \
                       (long)ldy, mean, rstd, (int)M);                                                              \
  else                                                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<CC, op_t>), grid, blk, 0, s, x, (long)ldx, gamma, beta, eps, (op_t*)y,         \
                       (long)ldy, mean, rstd, (int)M);
  if (C == 768) { LN_FWD(768) } else if (C == 512) { LN_FWD(512) } else return PVRL_EINVAL;
#undef LN_FWD
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int64_t pvrl_layernorm_bwd_workspace_bytes(int64_t M, int64_t C) {
  const int64_t nblk = cdiv(M > 0 ? M : 1, 4) < LN_BWD_MAX_BLOCKS ? cdiv(M > 0 ? M : 1, 4) : LN_BWD_MAX_BLOCKS;
  return nblk * 3 * C * (int64_t)sizeof(float);   // dgamma | dbeta | optional column sums of dx_out
}

extern "C" int pvrl_layernorm_bwd(const void* dy, int64_t lddy, int dy_is_f32, const float* x, int64_t ldx,
                                  const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                                  int64_t ldi, float* dx_out, int64_t ldo, float beta_acc, float* dgamma, float* dbeta,
                                  void* workspace, int64_t workspace_bytes, int64_t M, int64_t C, void* dxs_bf16,
                                  int64_t ldxs, const float* dxs_scale, int64_t dxs_rows, float* dxsum, const float* gscale,
                                  float* nonfinite, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!dy || !x || !mean || !rstd || !gamma || !dx_out || (!dgamma != !dbeta) || !workspace) return PVRL_EINVAL;
  if ((ldx % 4) || (lddy % 4) || (ldo % 4) || (dx_in && (ldi % 4))) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_layernorm_bwd_workspace_bytes(M, C)) return PVRL_EINVAL;
  if (dxs_bf16 && (ldxs % 4)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = cdiv(M, 4) < LN_BWD_MAX_BLOCKS ? cdiv(M, 4) : LN_BWD_MAX_BLOCKS;
  float* part = (float*)workspace;
  const int want_sum = dxsum ? 1 : 0;
  const int pstride = (2 + want_sum) * (int)C;
#define LN_BWD(CC)                                                                                                   \
  if (dy_is_f32)                                                                                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<CC, float>), dim3(nblk), dim3(256), 0, s, (const float*)dy, (long)lddy, x,      \
                       (long)ldx, mean, rstd, gamma, dx_in, (long)ldi, dx_out, (long)ldo, part, (int)M,              \
                       (op_t*)dxs_bf16, (long)ldxs, dxs_scale, (int)dxs_rows, want_sum);                               \
  else                                                                                                               \
    hipLaunchKernelGGL((ln_bwd_kernel<CC, op_t>), dim3(nblk), dim3(256), 0, s, (const op_t*)dy, (long)lddy, x,        \
                       (long)ldx, mean, rstd, gamma, dx_in, (long)ldi, dx_out, (long)ldo, part, (int)M,              \
                       (op_t*)dxs_bf16, (long)ldxs, dxs_scale, (int)dxs_rows, want_sum);
  if (C == 768) { LN_BWD(768) } else if (C == 512) { LN_BWD(512) } else return PVRL_EINVAL;
#undef LN_BWD
  PVRL_LAUNCH_CHECK();
  if (!dgamma) return PVRL_OK;       // deferred: the partials stay in `workspace` for pvrl_layernorm_bwd_reduce_batched
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((unsigned)cdiv(2 * C, 16)), dim3(256), 0, s, part, nblk,
                     (int)(2 * C), beta_acc, dgamma, dbeta, (int)C, pstride, gscale, nonfinite);
  PVRL_LAUNCH_CHECK();
  if (dxsum) {
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, s, part + 2 * C, nblk, (int)C,
                       beta_acc, dxsum, dxsum, (int)C, pstride, gscale, nonfinite);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

extern "C" int pvrl_layernorm_bwd_reduce_batched(int n, const pvrl_ln_reduce* items, const float* gscale, float* nonfinite,
                                                 void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!items) return PVRL_EINVAL;
  for (int i0 = 0; i0 < n; i0 += LN_RED_MAX) {
    LnReduceBatch g = {};
    g.n = n - i0 < LN_RED_MAX ? n - i0 : LN_RED_MAX;
    g.gscale = gscale; g.nonfinite = nonfinite;
    int blocks = 0;
    for (int i = 0; i < g.n; ++i) {
      const pvrl_ln_reduce& q = items[i0 + i];
      if (!q.part || !q.dgamma || !q.dbeta || q.M <= 0 || (q.C != 768 && q.C != 512) || (q.want_sum && !q.dxsum)) return PVRL_EINVAL;
      LnReduceItem& w = g.it[i];
      w.part = q.part; w.C = (int)q.C;
      w.nblk = cdiv(q.M, 4) < LN_BWD_MAX_BLOCKS ? cdiv(q.M, 4) : LN_BWD_MAX_BLOCKS;
      w.pstride = (2 + (q.want_sum ? 1 : 0)) * (int)q.C;
      w.n = w.pstride;
      w.beta = q.beta; w.beta_sum = q.beta_sum; w.dgamma = q.dgamma; w.dbeta = q.dbeta; w.dxsum = q.dxsum;
      w.first = blocks;
      blocks += cdiv(w.n, 16);
    }
    hipLaunchKernelGGL(colpart_reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}
