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
//
// Round 6: SPLIT residual stream.  The TimeSformer engine keeps the PATCH rows of the residual stream (and of its gradient) in the
// 16-bit operand type and the few cls rows in fp32 (the rounding-model oracle prices it: logits 2.8e-4 ->
// 3.0e-4, profiles/r6_resid16_rounding.txt).  A row-major matrix is then described by TWO pointers (`Rows`): rows [0, rows16) live
// in a 16-bit matrix, rows [rows16, M) in an fp32 one -- per LayerNorm pass over the 50k token rows: 77 instead of 154 MB for x,
// for the incoming and for the outgoing residual gradient.  rows16 = 0 is the all-fp32 matrix of rounds 1-5.
#include <cstdlib>
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

// rows [0, rows16) of a matrix in the 16-bit matrix `lo`, rows [rows16, ...) in the fp32 matrix `hi` (whose row 0 is row rows16).
// The kernels pick the part per wave (forward) / per workgroup (backward: the first workgroups walk the 16-bit rows, the last ones
// the fp32 rows), so every load of a row is unconditional: a load behind a per-row branch is a serial round trip.
struct Rows {
  op_t* lo; long ldlo;
  float* hi; long ldhi;
  int rows16;
};
template <bool LO> struct Part;
template <> struct Part<true> {
  typedef op_t T;
  static __device__ __forceinline__ T* row(const Rows& r, int m) { return r.lo ? r.lo + (long)m * r.ldlo : nullptr; }
};
template <> struct Part<false> {
  typedef float T;
  static __device__ __forceinline__ T* row(const Rows& r, int m) { return r.hi ? r.hi + (long)(m - r.rows16) * r.ldhi : nullptr; }
};

// RPW rows per wave, all their loads in flight together: a 16-bit row is 1.5 KB, and one of them per wave (8 B per lane and load
// instruction) does not keep enough bytes in flight to reach the HBM rate -- 33.6 us = 4.6 TB/s against 5.9 for fp32 rows (round 6).
template <int C, typename TOut, bool LO, int RPW>
__device__ __forceinline__ void ln_fwd_rows(const Rows& x, int row, int nrow, int lane, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float eps, TOut* __restrict__ y, long ldy,
                                            float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int NV = C / 256;
  typedef typename Part<LO>::T TX;
  f32x4 v[RPW][NV];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const TX* xr = Part<LO>::row(x, row + (r < nrow ? r : nrow - 1));      // (a row behind the end: the last row again, not stored)
#pragma unroll
    for (int j = 0; j < NV; ++j) v[r][j] = Vec4IO<TX>::load(xr + 4 * lane + 256 * j);
  }
  f32x4 g[NV], b[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    g[j] = *reinterpret_cast<const f32x4*>(gamma + 4 * lane + 256 * j);
    b[j] = *reinterpret_cast<const f32x4*>(beta + 4 * lane + 256 * j);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[r][j][0] + v[r][j][1]) + (v[r][j][2] + v[r][j][3]);
    const float mu = wave_sum(s) * (1.0f / C);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[r][j][e] - mu; ss += d * d; }
    const float rs = rsqrtf(wave_sum(ss) * (1.0f / C) + eps);
    if (r < nrow) {
      if (lane == 0) {
        if (mean) mean[row + r] = mu;
        if (rstd) rstd[row + r] = rs;
      }
      TOut* yr = y + (long)(row + r) * ldy;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[r][j][e] - mu) * rs * g[j][e] + b[j][e];
        Vec4IO<TOut>::store(yr + 4 * lane + 256 * j, o);
      }
    }
  }
}

constexpr int LN_FWD_RPW16 = 2;      // rows per wave on the 16-bit part
// the first nb_lo workgroups take the 16-bit rows (4 waves x LN_FWD_RPW16 rows), the others the fp32 rows (one row per wave)
template <int C, typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const Rows x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, TOut* __restrict__ y,
                                                     long ldy, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int M, int nb_lo) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((int)blockIdx.x < nb_lo) {
    const int row = ((int)blockIdx.x * 4 + wave) * LN_FWD_RPW16;
    if (row >= x.rows16) return;
    ln_fwd_rows<C, TOut, true, LN_FWD_RPW16>(x, row, min(LN_FWD_RPW16, x.rows16 - row), lane, gamma, beta, eps, y, ldy, mean, rstd);
  } else {
    const int row = x.rows16 + ((int)blockIdx.x - nb_lo) * 4 + wave;
    if (row >= M) return;
    ln_fwd_rows<C, TOut, false, 1>(x, row, 1, lane, gamma, beta, eps, y, ldy, mean, rstd);
  }
}

// workgroups of a backward launch over M rows: one more than the rows need, so that a split matrix always has a workgroup for each part
constexpr int LN_BWD_MAX_BLOCKS = 512;
inline int ln_bwd_max_blocks() {        // PVRL_LN_BWD_BLOCKS: probe runs (read once)
  static int n = 0;
  if (n == 0) {
    const char* e = getenv("PVRL_LN_BWD_BLOCKS");
    n = e && atoi(e) >= 8 ? atoi(e) : LN_BWD_MAX_BLOCKS;
  }
  return n;
}
inline int ln_bwd_nblk(long M) {
  const long n = (M + 3) / 4 + 1;
  return (int)(n < ln_bwd_max_blocks() ? n : ln_bwd_max_blocks());
}
// ... of which the LAST nb_hi walk the fp32 rows [rows16, M)
inline int ln_bwd_nblk_hi(long M, long rows16) {
  if (rows16 <= 0) return ln_bwd_nblk(M);
  if (rows16 >= M) return 0;
  const int nblk = ln_bwd_nblk(M), want = (int)((M - rows16 + 3) / 4), cap = nblk / 8 > 1 ? nblk / 8 : 1;
  return want < cap ? want : cap;
}

// rows row0 + 4 bid + wave, then every 4 nb-th, up to row1: the per-row work of the backward on one part of the split matrices
// HAS_IN: this part of dx_in is there (false: it reads as zeros and no load is issued -- a template parameter, not a branch around
// the loads: behind `ptr ? load : 0` every load waits for its own round trip).
// RPW rows per wave and iteration (rows row, row + 4 nb, ...), every load of all of them issued before the first wait: two on the
// 16-bit part, where one row's nine 8-byte loads per lane keep too few bytes in flight (one row: 83 us = 4.6 TB/s at M = 50k).
template <int C, typename TDy, bool LO, bool HAS_IN, int RPW>
__device__ __forceinline__ void ln_bwd_rows(int row0, int row1, int bid, int nb, int lane, int wave, const TDy* __restrict__ dy,
                                            long lddy, const Rows& x, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const Rows& dx_in, const Rows& dx_out,
                                            op_t* __restrict__ dxs, long ldxs, const float* __restrict__ dxs_scale, int dxs_rows,
                                            int want_sum, const f32x4 (&g)[C / 256], f32x4 (&dg)[C / 256], f32x4 (&db)[C / 256],
                                            f32x4 (&ds)[C / 256]) {
  constexpr int NV = C / 256;
  typedef typename Part<LO>::T TX;
  const int step = nb * 4;
  for (int base = row0 + bid * 4 + wave; base < row1; base += step * RPW) {
    int rowv[RPW];
    float mu[RPW], rs[RPW], sc[RPW];
    f32x4 xv[RPW][NV], dv[RPW][NV], din[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = base + r * step < row1 ? base + r * step : base;      // (behind the end: the first row again, results dropped)
      rowv[r] = row;
      mu[r] = mean[row];
      rs[r] = rstd[row];
      // the DropPath factor of the emitted copy: an UNCONDITIONAL load (of a harmless address when there is no factor) with the others
      sc[r] = *(dxs_scale ? dxs_scale + row : mean + row);
      // the incoming residual gradient is not needed before the row reductions, but its load goes out WITH x and dy: one
      // memory round trip per row instead of two
      const TX* dir = Part<LO>::row(dx_in, row);
      const TX* xr = Part<LO>::row(x, row);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if constexpr (HAS_IN) din[r][j] = Vec4IO<TX>::load(dir + 4 * lane + 256 * j);
        else din[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        xv[r][j] = Vec4IO<TX>::load(xr + 4 * lane + 256 * j);
        dv[r][j] = Vec4IO<TDy>::load(dy + (long)row * lddy + 4 * lane + 256 * j);
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (r > 0 && base + r * step >= row1) break;
      const int row = rowv[r];
      const float scr = dxs_scale ? sc[r] : 1.f;
      f32x4 xh[NV], gy[NV];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = dv[r][j][e];
          xh[j][e] = (xv[r][j][e] - mu[r]) * rs[r];
          gy[j][e] = d * g[j][e];
          c1 += gy[j][e];
          c2 += gy[j][e] * xh[j][e];
          dg[j][e] += d * xh[j][e];
          db[j][e] += d;
        }
      }
      c1 = wave_sum(c1) * (1.0f / C);
      c2 = wave_sum(c2) * (1.0f / C);
      TX* dor = Part<LO>::row(dx_out, row);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = 4 * lane + 256 * j;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs[r] * (gy[j][e] - c1 - xh[j][e] * c2);
        o += din[r][j];
        Vec4IO<TX>::store(dor + c, o);
        if (want_sum && row < dxs_rows) ds[j] += o;      // unscaled column sums of the rows that feed the next stage
        if (dxs && row < dxs_rows)     // bf16 (optionally DropPath-scaled) copy: the GEMM operand of the next backward stage
          Vec4IO<op_t>::store(dxs + (long)row * ldxs + c, scr * o);
      }
    }
  }
}

constexpr int LN_BWD_RPW16 = 2;      // rows per wave and iteration on the 16-bit part

template <int C, typename TDy>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, long lddy, const Rows x,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const Rows dx_in, const Rows dx_out,
                                                     float* __restrict__ part, int M, op_t* __restrict__ dxs,
                                                     long ldxs, const float* __restrict__ dxs_scale, int dxs_rows,
                                                     int want_sum, int nb_hi) {
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
  const int nb_lo = (int)gridDim.x - nb_hi;
#define LN_ROWS(LO_, IN_, r0, r1, b, n)                                                                                          \
  ln_bwd_rows<C, TDy, LO_, IN_, (LO_ ? LN_BWD_RPW16 : 1)>(r0, r1, b, n, lane, wave, dy, lddy, x, mean, rstd, dx_in, dx_out, dxs, ldxs, \
                                                          dxs_scale, dxs_rows, want_sum, g, dg, db, ds)
  if ((int)blockIdx.x < nb_lo) {
    if (dx_in.lo) LN_ROWS(true, true, 0, x.rows16, (int)blockIdx.x, nb_lo);
    else LN_ROWS(true, false, 0, x.rows16, (int)blockIdx.x, nb_lo);
  } else {
    if (dx_in.hi) LN_ROWS(false, true, x.rows16, M, (int)blockIdx.x - nb_lo, nb_hi);
    else LN_ROWS(false, false, x.rows16, M, (int)blockIdx.x - nb_lo, nb_hi);
  }
#undef LN_ROWS
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

// LN_BWD_MAX_BLOCKS, measured on MI355X (M = 50,208): alone, 256 workgroups are fastest (94 us; 512: 99, 1024: 111, 2048: 117 -- fewer
// partial sums to write and reduce); inside the training step, next to the weight-gradient GEMMs of the side stream, 512 win (572 vs
// 561 clips/s): one workgroup per CU is starved by the co-running kernel.

}  // namespace

extern "C" int pvrl_layernorm_fwd_split(const void* x16, int64_t ldx16, int64_t rows16, const float* x, int64_t ldx,
                                        const float* gamma, const float* beta, float eps, void* y, int64_t ldy, int out_is_f32,
                                        float* mean, float* rstd, int64_t M, int64_t C, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (rows16 < 0 || rows16 > M || (rows16 > 0 && (!x16 || (ldx16 % 4))) || (rows16 < M && (!x || (ldx % 4)))) return PVRL_EINVAL;
  if (!gamma || !beta || !y || (ldy % 4)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nb_lo = (int)cdiv(rows16, 4 * LN_FWD_RPW16);
  const dim3 grid((unsigned)(nb_lo + cdiv(M - rows16, 4))), blk(256);
  const Rows xr = {(op_t*)x16, (long)ldx16, (float*)x, (long)ldx, (int)rows16};
#define LN_FWD(CC)                                                                                                  \
  if (out_is_f32)                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_kernel<CC, float>), grid, blk, 0, s, xr, gamma, beta, eps, (float*)y,                 \
                       (long)ldy, mean, rstd, (int)M, nb_lo);                                                       \
  else                                                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<CC, op_t>), grid, blk, 0, s, xr, gamma, beta, eps, (op_t*)y,                   \
                       (long)ldy, mean, rstd, (int)M, nb_lo);
  if (C == 768) { LN_FWD(768) } else if (C == 512) { LN_FWD(512) } else return PVRL_EINVAL;
#undef LN_FWD
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                  void* y, int64_t ldy, int out_is_f32, float* mean, float* rstd, int64_t M,
                                  int64_t C, void* stream) {
  if (M > 0 && !x) return PVRL_EINVAL;
  return pvrl_layernorm_fwd_split(nullptr, 0, 0, x, ldx, gamma, beta, eps, y, ldy, out_is_f32, mean, rstd, M, C, stream);
}

extern "C" int64_t pvrl_layernorm_bwd_workspace_bytes(int64_t M, int64_t C) {
  return (int64_t)ln_bwd_nblk(M > 0 ? M : 1) * 3 * C * (int64_t)sizeof(float);   // dgamma | dbeta | optional column sums of dx_out
}

extern "C" int pvrl_layernorm_bwd_split(const void* dy, int64_t lddy, int dy_is_f32, const pvrl_rows* x, const float* mean,
                                        const float* rstd, const float* gamma, const pvrl_rows* dx_in, const pvrl_rows* dx_out,
                                        float beta_acc, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                        int64_t M, int64_t C, void* dxs_bf16, int64_t ldxs, const float* dxs_scale,
                                        int64_t dxs_rows, float* dxsum, const float* gscale, float* nonfinite, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!dy || !x || !mean || !rstd || !gamma || !dx_out || (!dgamma != !dbeta) || !workspace || (lddy % 4)) return PVRL_EINVAL;
  auto bad = [&](const pvrl_rows* r, bool need) {       // every part that holds rows must be there (x) / may be null (dx_in: zeros)
    if (r->rows16 < 0 || r->rows16 > M || (r->lo && (r->ldlo % 4)) || (r->hi && (r->ldhi % 4))) return true;
    return need && ((r->rows16 > 0 && !r->lo) || (r->rows16 < M && !r->hi));
  };
  if (bad(x, true) || bad(dx_out, true) || (dx_in && bad(dx_in, false))) return PVRL_EINVAL;
  if (dx_out->rows16 != x->rows16 || (dx_in && dx_in->rows16 != x->rows16)) return PVRL_EINVAL;     // one split for the three matrices
  if (workspace_bytes < pvrl_layernorm_bwd_workspace_bytes(M, C)) return PVRL_EINVAL;
  if (dxs_bf16 && (ldxs % 4)) return PVRL_EINVAL;
  auto rows_of = [](const pvrl_rows* r) {
    Rows o = {nullptr, 0, nullptr, 0, 0};
    if (r) o = Rows{(op_t*)r->lo, (long)r->ldlo, (float*)r->hi, (long)r->ldhi, (int)r->rows16};
    return o;
  };
  const Rows xr = rows_of(x), dxi = rows_of(dx_in), dxo = rows_of(dx_out);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = ln_bwd_nblk(M), nb_hi = ln_bwd_nblk_hi(M, xr.rows16);
  float* part = (float*)workspace;
  const int want_sum = dxsum ? 1 : 0;
  const int pstride = (2 + want_sum) * (int)C;
#define LN_BWD(CC)                                                                                                   \
  if (dy_is_f32)                                                                                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<CC, float>), dim3(nblk), dim3(256), 0, s, (const float*)dy, (long)lddy, xr,     \
                       mean, rstd, gamma, dxi, dxo, part, (int)M,                                                    \
                       (op_t*)dxs_bf16, (long)ldxs, dxs_scale, (int)dxs_rows, want_sum, nb_hi);                        \
  else                                                                                                               \
    hipLaunchKernelGGL((ln_bwd_kernel<CC, op_t>), dim3(nblk), dim3(256), 0, s, (const op_t*)dy, (long)lddy, xr,       \
                       mean, rstd, gamma, dxi, dxo, part, (int)M,                                                    \
                       (op_t*)dxs_bf16, (long)ldxs, dxs_scale, (int)dxs_rows, want_sum, nb_hi);
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

extern "C" int pvrl_layernorm_bwd(const void* dy, int64_t lddy, int dy_is_f32, const float* x, int64_t ldx,
                                  const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                                  int64_t ldi, float* dx_out, int64_t ldo, float beta_acc, float* dgamma, float* dbeta,
                                  void* workspace, int64_t workspace_bytes, int64_t M, int64_t C, void* dxs_bf16,
                                  int64_t ldxs, const float* dxs_scale, int64_t dxs_rows, float* dxsum, const float* gscale,
                                  float* nonfinite, void* stream) {
  if (M > 0 && (!x || !dx_out)) return PVRL_EINVAL;
  const pvrl_rows xr = {nullptr, 0, (void*)x, ldx, 0}, dxi = {nullptr, 0, (void*)dx_in, ldi, 0}, dxo = {nullptr, 0, dx_out, ldo, 0};
  return pvrl_layernorm_bwd_split(dy, lddy, dy_is_f32, &xr, mean, rstd, gamma, dx_in ? &dxi : nullptr, &dxo, beta_acc, dgamma, dbeta,
                                  workspace, workspace_bytes, M, C, dxs_bf16, ldxs, dxs_scale, dxs_rows, dxsum, gscale, nonfinite,
                                  stream);
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
      w.nblk = ln_bwd_nblk(q.M);
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
