// HBM-bound layout / cast kernels around the GEMMs of the TimeSformer path.
//  * patchify: frames -> bf16 im2col rows in (b, n, t) token order   (PatchEmbed.forward,
//    lib/models/vit.py:174-180, and the '(b t) n m -> (b n) t m' regroup of vit.py:396)
//  * embed table / batch-sum: pos_embed + time_embed prologue and its gradient (vit.py:370-407)
//  * cast / scale / transpose helpers (bf16 GEMM operands from fp32 masters and fp32 gradients;
//    DropPath row scaling, lib/models/vit_utils.py:140-155)
//  * cls-token group mean / broadcast (vit.py:139-141,147-149)
#include "common.h"
#include "../../include/pvrl.h"

namespace {

// frames fp32 [B][3][T][HI][WI]  ->  out bf16 [(b, n, t)][c*256 + py*16 + px]
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ frames, op_t* __restrict__ out, int B,
                                                       int T, int HI, int WI, long ldo) {
  const int xg = WI >> 3;  // groups of 8 pixels per image row
  const long total = (long)B * 3 * T * HI * xg;
  const int PW = WI >> 4, PH = HI >> 4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    long r = idx;
    const int x8 = (int)(r % xg); r /= xg;
    const int y = (int)(r % HI); r /= HI;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    const float* src = frames + idx * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 d = *reinterpret_cast<const f32x4*>(src + 4);
    opx8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (op_t)a[e]; o[4 + e] = (op_t)d[e]; }
    const int n = (y >> 4) * PW + (x8 >> 1);
    const long row = ((long)b * PH * PW + n) * T + t;
    const int col = c * 256 + (y & 15) * 16 + (x8 & 1) * 8;
    *reinterpret_cast<opx8*>(out + row * ldo + col) = o;
  }
}

// GPU-side input pipeline fused into the patch-embed im2col (reference: CPU workers run tensor_normalize ->
// permute -> random_short_side_scale_jitter (bilinear, align_corners=False) -> crop -> horizontal flip,
// lib/datasets/howto100m.py:437-452, lib/datasets/utils.py:110-160,309-326, lib/datasets/transform.py:8-147).
// frames uint8 [B][T][H0][W0][3] (decoder order) ; prm int32 [B][5] = {new_h, new_w, y_off, x_off, flip}
//   ->  out bf16 [(b, n, t)][c*256 + py*16 + px] of the (crop x crop) clip, normalised (v/255 - mean)/std.
// One thread = 8 consecutive output pixels of one row, all 3 channels (interleaved source bytes are read once).
__global__ __launch_bounds__(256) void frames_u8_patchify_kernel(const unsigned char* __restrict__ frames,
                                                                 const int* __restrict__ prm, op_t* __restrict__ out,
                                                                 int B, int T, int H0, int W0, int crop, float m0,
                                                                 float m1, float m2, float s0, float s1, float s2,
                                                                 long ldo) {
  const int xg = crop >> 3, PW = crop >> 4;
  const long total = (long)B * T * crop * xg;
  const float mean[3] = {m0, m1, m2}, istd[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    long r = idx;
    const int x8 = (int)(r % xg); r /= xg;
    const int y = (int)(r % crop); r /= crop;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const int* q = prm + b * 5;
    const int nh = q[0], nw = q[1], yo = q[2], xo = q[3], flip = q[4];
    const float sy = (float)H0 / (float)nh, sx = (float)W0 / (float)nw;
    float fy = sy * ((float)(y + yo) + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy, y1 = y0 + (y0 < H0 - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    const unsigned char* f0 = frames + (((long)b * T + t) * H0 + y0) * W0 * 3;
    const unsigned char* f1 = frames + (((long)b * T + t) * H0 + y1) * W0 * 3;
    opx8 o[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int x = x8 * 8 + e;
      const int X = (flip ? crop - 1 - x : x) + xo;
      float fx = sx * ((float)X + 0.5f) - 0.5f;
      fx = fx < 0.f ? 0.f : fx;
      const int x0 = (int)fx, x1 = x0 + (x0 < W0 - 1 ? 1 : 0);
      const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = ly0 * (lx0 * (float)f0[x0 * 3 + c] + lx1 * (float)f0[x1 * 3 + c]) +
                        ly1 * (lx0 * (float)f1[x0 * 3 + c] + lx1 * (float)f1[x1 * 3 + c]);
        o[c][e] = (op_t)((v / 255.0f - mean[c]) * istd[c]);
      }
    }
    const int n = (y >> 4) * PW + (x8 >> 1);
    const long row = ((long)b * (crop >> 4) * PW + n) * T + t;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<opx8*>(out + row * ldo + c * 256 + (y & 15) * 16 + (x8 & 1) * 8) = o[c];
  }
}

// Same input pipeline, producing the fp32 clip tensor [B][3][T][crop][crop] the reference's loader would have shipped
// (for consumers that do their own im2col, e.g. the MViT stem).  One thread = 4 consecutive output pixels of one channel row.
__global__ __launch_bounds__(256) void frames_u8_to_f32_kernel(const unsigned char* __restrict__ frames,
                                                               const int* __restrict__ prm, float* __restrict__ out, int B,
                                                               int T, int H0, int W0, int crop, float m0, float m1,
                                                               float m2, float s0, float s1, float s2) {
  const int xg = crop >> 2;
  const long total = (long)B * 3 * T * crop * xg;
  const float mean[3] = {m0, m1, m2}, istd[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    long r = idx;
    const int x4 = (int)(r % xg); r /= xg;
    const int y = (int)(r % crop); r /= crop;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    const int* q = prm + b * 5;
    const int nh = q[0], nw = q[1], yo = q[2], xo = q[3], flip = q[4];
    const float sy = (float)H0 / (float)nh, sx = (float)W0 / (float)nw;
    float fy = sy * ((float)(y + yo) + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy, y1 = y0 + (y0 < H0 - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    const unsigned char* f0 = frames + (((long)b * T + t) * H0 + y0) * W0 * 3 + c;
    const unsigned char* f1 = frames + (((long)b * T + t) * H0 + y1) * W0 * 3 + c;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x4 * 4 + e;
      const int X = (flip ? crop - 1 - x : x) + xo;
      float fx = sx * ((float)X + 0.5f) - 0.5f;
      fx = fx < 0.f ? 0.f : fx;
      const int x0 = (int)fx, x1 = x0 + (x0 < W0 - 1 ? 1 : 0);
      const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
      const float v = ly0 * (lx0 * (float)f0[x0 * 3] + lx1 * (float)f0[x1 * 3]) + ly1 * (lx0 * (float)f1[x0 * 3] + lx1 * (float)f1[x1 * 3]);
      o[e] = (v / 255.0f - mean[c]) * istd[c];
    }
    *reinterpret_cast<f32x4*>(out + idx * 4) = o;
  }
}

// E[n*T + t][c] = bias[c] + pos[1 + n][c] + time[t][c]
__global__ __launch_bounds__(256) void embed_table_kernel(const float* __restrict__ pos, const float* __restrict__ time,
                                                          const float* __restrict__ bias, float* __restrict__ E, int N,
                                                          int T, int C) {
  const long total = (long)N * T * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long nt = idx / C;
    const int t = (int)(nt % T), n = (int)(nt / T);
    E[idx] = (bias ? bias[c] : 0.f) + pos[(long)(1 + n) * C + c] + time[(long)t * C + c];
  }
}

// G[r][c] = sum_b dx[b*rows + r][c]      (gradient of the broadcast embedding table)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ dx, long ld, int B, int rows, int C,
                                                        float* __restrict__ G) {
  const int c4n = C >> 2;
  const long total = (long)rows * c4n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c4 = (int)(idx % c4n);
    const long r = idx / c4n;
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) a += *reinterpret_cast<const f32x4*>(dx + ((long)b * rows + r) * ld + c4 * 4);
    *reinterpret_cast<f32x4*>(G + r * C + c4 * 4) = a;
  }
}

// the same over 16-bit rows (the patch rows of the split residual gradient stream, round 6), fp32 sums
__global__ __launch_bounds__(256) void batch_sum16_kernel(const op_t* __restrict__ dx, long ld, int B, int rows, int C,
                                                          float* __restrict__ G) {
  const int c8n = C >> 3;
  const long total = (long)rows * c8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long r = idx / c8n;
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, d = a;
    for (int b = 0; b < B; ++b) {
      const opx8 v = *reinterpret_cast<const opx8*>(dx + ((long)b * rows + r) * ld + c8 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] += (float)v[e]; d[e] += (float)v[4 + e]; }
    }
    *reinterpret_cast<f32x4*>(G + r * C + c8 * 8) = a;
    *reinterpret_cast<f32x4*>(G + r * C + c8 * 8 + 4) = d;
  }
}

// out bf16[m][c] = rowscale[m] * in fp32[m][c]
__global__ __launch_bounds__(256) void cast_scale_kernel(const float* __restrict__ in, long ldi,
                                                         const float* __restrict__ rowscale, op_t* __restrict__ out,
                                                         long ldo, long M, int C) {
  const int c8n = C >> 3;
  const long total = M * c8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long m = idx / c8n;
    const float rs = rowscale ? rowscale[m] : 1.f;
    const float* src = in + m * ldi + c8 * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 d = *reinterpret_cast<const f32x4*>(src + 4);
    opx8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (op_t)(rs * a[e]); o[4 + e] = (op_t)(rs * d[e]); }
    *reinterpret_cast<opx8*>(out + m * ldo + c8 * 8) = o;
  }
}

// W fp32 [R][C] -> Wt bf16 [C][R]   (32x32 LDS tiles)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, op_t* __restrict__ out, int R,
                                                             int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < R && c < C) out[(long)c * R + r] = (op_t)tile[tx][ty + 8 * k];
  }
}

// W fp32 [R][C] -> Wb bf16 [R][C] and Wt bf16 [C][R] in one pass (both GEMM operand copies of a weight matrix)
// (ldo / ldt: leading dimensions of the two copies -- larger than C / R when the copies live in zero-padded buffers)
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ in, op_t* __restrict__ out,
                                                          op_t* __restrict__ out_t, int R, int C, long ldo, long ldt,
                                                          const float* __restrict__ bias, float* __restrict__ bias_out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (bias && blockIdx.x == 0 && threadIdx.x < 32 && r0 + threadIdx.x < R)     // the layer's bias rides along: fp32 copy into its padded buffer
    bias_out[r0 + threadIdx.x] = bias[r0 + threadIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    const float v = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    tile[ty + 8 * k][tx] = v;
    if (r < R && c < C) out[(long)r * ldo + c] = (op_t)v;
  }
  __syncthreads();
  if (out_t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < R && c < C) out_t[(long)c * ldt + r] = (op_t)tile[tx][ty + 8 * k];
    }
  }
}

// y[r] = beta * y[r] + sum_c W[r][c] * x[c]: one wave per row of a small dense matrix (fp32 or bf16), fp32 x / y.
// (W_fc b_proj and W_fc^T db_e of the fused temporal branch: 768 x 768, a few microseconds)
template <typename TW>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const TW* __restrict__ W, long ld, int R, int C,
                                                        const float* __restrict__ x, float beta, float* __restrict__ y,
                                                        const float* __restrict__ gscale) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float a = 0.f;
  for (int c = lane; c < C; c += 64) a += (float)W[(long)r * ld + c] * x[c];
  a = wave_sum(a);
  if (gscale) a *= *gscale;
  if (lane == 0) y[r] = (beta != 0.f ? beta * y[r] : 0.f) + a;
}

// The same for up to 96 weight matrices in ONE launch (a ViT-B encoder has 85: one 6-us launch each otherwise).
constexpr int CAST_MULTI_MAX = 96;
struct CastItem { const float* in; op_t* out; op_t* out_t; int R, C, first, tiles_c; };
struct CastMulti { int n; CastItem it[CAST_MULTI_MAX]; };
__global__ __launch_bounds__(256) void cast_weight_multi_kernel(CastMulti g) {
  __shared__ float tile[32][33];
  int lo = 0, hi = g.n - 1;                       // last item whose first block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= g.it[mid].first) lo = mid; else hi = mid - 1;
  }
  const CastItem w = g.it[lo];
  const int b = (int)blockIdx.x - w.first;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (b % w.tiles_c) * 32, r0 = (b / w.tiles_c) * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    const float v = (r < w.R && c < w.C) ? w.in[(long)r * w.C + c] : 0.f;
    tile[ty + 8 * k][tx] = v;
    if (r < w.R && c < w.C) w.out[(long)r * w.C + c] = (op_t)v;
  }
  __syncthreads();
  if (w.out_t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < w.R && c < w.C) w.out_t[(long)c * w.R + r] = (op_t)tile[tx][ty + 8 * k];
    }
  }
}

// The same with 64 x 64 tiles and 16-byte loads / 8-byte stores, for matrices whose sides are multiples of 64 (every Linear of the encoder):
// the 32 x 32 form above reads 4 bytes and writes 2 bytes per thread and instruction and sat at 3.4 TB/s for its 1.45 GB.
__global__ __launch_bounds__(256) void cast_weight_multi64_kernel(CastMulti g) {
  __shared__ op_t tile[64][68];
  int lo = 0, hi = g.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= g.it[mid].first) lo = mid; else hi = mid - 1;
  }
  const CastItem w = g.it[lo];
  const int b = (int)blockIdx.x - w.first;
  const int c0 = (b % w.tiles_c) * 64, r0 = (b / w.tiles_c) * 64;
  const int q = threadIdx.x & 15, y = threadIdx.x >> 4;
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(w.in + (long)(r0 + y + 16 * k) * w.C + c0 + 4 * q);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    opx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (op_t)v[k][e];
    *reinterpret_cast<opx4*>(w.out + (long)(r0 + y + 16 * k) * w.C + c0 + 4 * q) = o;
    *reinterpret_cast<opx4*>(&tile[y + 16 * k][4 * q]) = o;
  }
  if (!w.out_t) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = y + 16 * k;
    opx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = tile[4 * q + e][c];
    *reinterpret_cast<opx4*>(w.out_t + (long)(c0 + c) * w.R + r0 + 4 * q) = o;
  }
}

// out[g][c] = resid[g][c] + alpha * sum_t scale[g*G + t] * in[g*G + t][c]
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void group_reduce_kernel(const TIn* __restrict__ in, long ldi, int groups, int G, int C,
                                                           const float* __restrict__ scale, float alpha,
                                                           const float* __restrict__ resid, long ldr,
                                                           TOut* __restrict__ out, long ldo) {
  const long total = (long)groups * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long g = idx / C;
    float a = 0.f;
    for (int t = 0; t < G; ++t) {
      const long r = g * G + t;
      a += (scale ? scale[r] : 1.f) * (float)in[r * ldi + c];
    }
    a *= alpha;
    if (resid) a += resid[g * ldr + c];
    out[g * ldo + c] = (TOut)a;
  }
}

// out bf16[g*G + t][c] = alpha * scale[g*G + t] * in fp32[g][c]
__global__ __launch_bounds__(256) void group_bcast_kernel(const float* __restrict__ in, long ldi, int groups, int G, int C,
                                                          const float* __restrict__ scale, float alpha,
                                                          op_t* __restrict__ out, long ldo) {
  const long total = (long)groups * G * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const long g = r / G;
    out[r * ldo + c] = (op_t)(alpha * (scale ? scale[r] : 1.f) * in[g * ldi + c]);
  }
}

inline unsigned grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int pvrl_patchify(const float* frames, int64_t B, int64_t T, int64_t HI, int64_t WI, void* out, int64_t ldo,
                             void* stream) {
  if (B <= 0) return PVRL_OK;
  if (!frames || !out || (HI % 16) || (WI % 16) || (ldo % 8) || ldo < 768) return PVRL_EINVAL;
  const long total = B * 3 * T * HI * (WI >> 3);
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames, (op_t*)out,
                     (int)B, (int)T, (int)HI, (int)WI, (long)ldo);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_frames_u8_patchify(const void* frames, const int32_t* params, int64_t B, int64_t T, int64_t H0,
                                       int64_t W0, int64_t crop, const float* mean3, const float* std3, void* out,
                                       int64_t ldo, void* stream) {
  if (B <= 0) return PVRL_OK;
  if (!frames || !params || !mean3 || !std3 || !out || crop <= 0 || (crop % 16) || (ldo % 8) || ldo < 768 || H0 <= 0 ||
      W0 <= 0)
    return PVRL_EINVAL;
  if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return PVRL_EINVAL;
  const long total = B * T * crop * (crop >> 3);
  hipLaunchKernelGGL(frames_u8_patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)frames, (const int*)params, (op_t*)out, (int)B, (int)T, (int)H0, (int)W0,
                     (int)crop, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (long)ldo);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_frames_u8_to_f32(const void* frames, const int32_t* params, int64_t B, int64_t T, int64_t H0, int64_t W0,
                                     int64_t crop, const float* mean3, const float* std3, float* out, void* stream) {
  if (B <= 0) return PVRL_OK;
  if (!frames || !params || !mean3 || !std3 || !out || crop <= 0 || (crop % 4) || H0 <= 0 || W0 <= 0) return PVRL_EINVAL;
  if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return PVRL_EINVAL;
  const long total = B * 3 * T * crop * (crop >> 2);
  hipLaunchKernelGGL(frames_u8_to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)frames, (const int*)params, out, (int)B, (int)T, (int)H0, (int)W0, (int)crop,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_embed_table(const float* pos, const float* time, const float* bias, float* E, int64_t N, int64_t T,
                                int64_t C, void* stream) {
  if (!pos || !time || !E) return PVRL_EINVAL;
  hipLaunchKernelGGL(embed_table_kernel, dim3(grid_for(N * T * C)), dim3(256), 0, (hipStream_t)stream, pos, time, bias,
                     E, (int)N, (int)T, (int)C);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_batch_sum(const float* dx, int64_t ld, int64_t B, int64_t rows, int64_t C, float* G, void* stream) {
  if (!dx || !G || (C % 4) || (ld % 4)) return PVRL_EINVAL;
  hipLaunchKernelGGL(batch_sum_kernel, dim3(grid_for(rows * (C >> 2))), dim3(256), 0, (hipStream_t)stream, dx,
                     (long)ld, (int)B, (int)rows, (int)C, G);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_batch_sum_bf16(const void* dx, int64_t ld, int64_t B, int64_t rows, int64_t C, float* G, void* stream) {
  if (!dx || !G || (C % 8) || (ld % 8)) return PVRL_EINVAL;
  hipLaunchKernelGGL(batch_sum16_kernel, dim3(grid_for(rows * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const op_t*)dx,
                     (long)ld, (int)B, (int)rows, (int)C, G);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_cast_scale_bf16(const float* in, int64_t ldi, const float* rowscale, void* out, int64_t ldo,
                                    int64_t M, int64_t C, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!in || !out || (C % 8) || (ldi % 4) || (ldo % 8)) return PVRL_EINVAL;
  hipLaunchKernelGGL(cast_scale_kernel, dim3(grid_for(M * (C >> 3))), dim3(256), 0, (hipStream_t)stream, in, (long)ldi,
                     rowscale, (op_t*)out, (long)ldo, (long)M, (int)C);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_cast_transpose_bf16(const float* in, void* out, int64_t R, int64_t C, void* stream) {
  if (!in || !out || R <= 0 || C <= 0) return PVRL_EINVAL;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0,
                     (hipStream_t)stream, in, (op_t*)out, (int)R, (int)C);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_cast_weight_bf16(const float* in, void* out, void* out_t, int64_t R, int64_t C, void* stream) {
  if (!in || !out || R <= 0 || C <= 0) return PVRL_EINVAL;
  hipLaunchKernelGGL(cast_weight_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0,
                     (hipStream_t)stream, in, (op_t*)out, (op_t*)out_t, (int)R, (int)C, (long)C, (long)R,
                     (const float*)nullptr, (float*)nullptr);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_gemv_rows_f32(const void* W, int w_is_bf16, int64_t ld, int64_t R, int64_t C, const float* x, float beta,
                                  float* y, const float* gscale, void* stream) {
  if (R <= 0) return PVRL_OK;
  if (!W || !x || !y || C <= 0 || ld < C) return PVRL_EINVAL;
  const dim3 grid((unsigned)cdiv(R, 4)), blk(256);
  if (w_is_bf16)
    hipLaunchKernelGGL(gemv_rows_kernel<op_t>, grid, blk, 0, (hipStream_t)stream, (const op_t*)W, (long)ld, (int)R, (int)C,
                       x, beta, y, gscale);
  else
    hipLaunchKernelGGL(gemv_rows_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)W, (long)ld, (int)R, (int)C,
                       x, beta, y, gscale);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_cast_weights_multi_bf16(int n, const pvrl_cast_problem* problems, void* stream) {
  if (n < 0 || (n > 0 && !problems)) return PVRL_EINVAL;
  for (int i0 = 0; i0 < n; i0 += CAST_MULTI_MAX) {
    CastMulti g = {};
    g.n = n - i0 < CAST_MULTI_MAX ? n - i0 : CAST_MULTI_MAX;
    int blocks = 0;
    bool all64 = true;      // sides multiples of 64, 16-byte aligned: the 64 x 64-tile kernel
    for (int i = 0; i < g.n; ++i) {
      const pvrl_cast_problem& q = problems[i0 + i];
      if (!q.in || !q.out || q.R <= 0 || q.C <= 0) return PVRL_EINVAL;
      all64 = all64 && (q.R % 64 == 0) && (q.C % 64 == 0) && ((uintptr_t)q.in % 16 == 0) && ((uintptr_t)q.out % 8 == 0) &&
              (!q.out_t || (uintptr_t)q.out_t % 8 == 0);
    }
    const int ts = all64 ? 64 : 32;
    for (int i = 0; i < g.n; ++i) {
      const pvrl_cast_problem& q = problems[i0 + i];
      CastItem& w = g.it[i];
      w.in = q.in; w.out = (op_t*)q.out; w.out_t = (op_t*)q.out_t; w.R = (int)q.R; w.C = (int)q.C;
      w.first = blocks; w.tiles_c = (int)cdiv(q.C, ts);
      blocks += w.tiles_c * (int)cdiv(q.R, ts);
    }
    if (all64) hipLaunchKernelGGL(cast_weight_multi64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(cast_weight_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

extern "C" int pvrl_cast_weight_pad_bf16(const float* in, void* out, int64_t ldo, void* out_t, int64_t ldt, int64_t R,
                                         int64_t C, const float* bias, float* bias_out, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || ldo < C || (out_t && ldt < R) || (bias && !bias_out)) return PVRL_EINVAL;
  hipLaunchKernelGGL(cast_weight_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0,
                     (hipStream_t)stream, in, (op_t*)out, (op_t*)out_t, (int)R, (int)C, (long)ldo, (long)ldt, bias, bias_out);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_group_reduce(const void* in, int in_is_f32, int64_t ldi, int64_t groups, int64_t G, int64_t C,
                                 const float* scale, float alpha, const float* resid, int64_t ldr, void* out,
                                 int out_is_f32, int64_t ldo, void* stream) {
  if (groups <= 0) return PVRL_OK;
  if (!in || !out) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(grid_for(groups * C)), blk(256);
#define GR(TI, TO)                                                                                                 \
  hipLaunchKernelGGL((group_reduce_kernel<TI, TO>), grid, blk, 0, s, (const TI*)in, (long)ldi, (int)groups, (int)G, \
                     (int)C, scale, alpha, resid, (long)ldr, (TO*)out, (long)ldo)
  if (in_is_f32 && out_is_f32) GR(float, float);
  else if (in_is_f32) GR(float, op_t);
  else if (out_is_f32) GR(op_t, float);
  else GR(op_t, op_t);
#undef GR
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_group_bcast_bf16(const float* in, int64_t ldi, int64_t groups, int64_t G, int64_t C,
                                     const float* scale, float alpha, void* out, int64_t ldo, void* stream) {
  if (groups <= 0) return PVRL_OK;
  if (!in || !out) return PVRL_EINVAL;
  hipLaunchKernelGGL(group_bcast_kernel, dim3(grid_for(groups * G * C)), dim3(256), 0, (hipStream_t)stream, in,
                     (long)ldi, (int)groups, (int)G, (int)C, scale, alpha, (op_t*)out, (long)ldo);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

namespace {
// out[r][c] += a[r] * b[c]   (torch.addr_ spends 54 us on this 768 x 768 update: a broadcast iterator, 4 bytes per thread)
__global__ __launch_bounds__(256) void rank1_add_kernel(float* __restrict__ out, long ld, const float* __restrict__ a,
                                                        const float* __restrict__ b, int R, int C4,
                                                        const float* __restrict__ gscale) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)R * C4) return;
  const int r = (int)(idx / C4), c = (int)(idx - (long)r * C4) * 4;
  const float av = gscale ? a[r] * *gscale : a[r];
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
  f32x4* o = reinterpret_cast<f32x4*>(out + (long)r * ld + c);
  *o = *o + av * bv;
}
}  // namespace

extern "C" int pvrl_rank1_add_f32(float* out, int64_t ld, const float* a, const float* b, int64_t R, int64_t C,
                                  const float* gscale, void* stream) {
  if (R <= 0 || C <= 0) return PVRL_OK;
  if (!out || !a || !b || (C % 4) || (ld % 4)) return PVRL_EINVAL;
  const long n = (long)R * (C / 4);
  hipLaunchKernelGGL(rank1_add_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, (long)ld, a, b, (int)R,
                     (int)(C / 4), gscale);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

// The bias products of the fused temporal branch for MANY blocks in one launch each (twelve 5-7 us launches of each kind per pass otherwise):
//   gemv:  y_i[r] = beta_i * y_i[r] + gscale * W_i[r, :] . x_i      rank-1:  out_i[r][c] += gscale * a_i[r] * b_i[c]
namespace {
constexpr int SMALL_BATCH_MAX = 16;
struct GemvBatch { int n, R, C, w16; long ld; const void* W[SMALL_BATCH_MAX]; const float* x[SMALL_BATCH_MAX]; float* y[SMALL_BATCH_MAX]; float beta[SMALL_BATCH_MAX]; const float* gscale; };
__global__ __launch_bounds__(256) void gemv_rows_batched_kernel(GemvBatch g) {
  const int q = blockIdx.y, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= g.R) return;
  const float* x = g.x[q];
  float a = 0.f;
  if (g.w16) {
    const op_t* W = (const op_t*)g.W[q];
    for (int c = lane; c < g.C; c += 64) a += (float)W[(long)r * g.ld + c] * x[c];
  } else {
    const float* W = (const float*)g.W[q];
    for (int c = lane; c < g.C; c += 64) a += W[(long)r * g.ld + c] * x[c];
  }
  a = wave_sum(a);
  if (g.gscale) a *= *g.gscale;
  if (lane == 0) g.y[q][r] = (g.beta[q] != 0.f ? g.beta[q] * g.y[q][r] : 0.f) + a;
}
struct Rank1Batch { int n, R, C4; long ld; float* out[SMALL_BATCH_MAX]; const float* a[SMALL_BATCH_MAX]; const float* b[SMALL_BATCH_MAX]; const float* gscale; };
__global__ __launch_bounds__(256) void rank1_add_batched_kernel(Rank1Batch g) {
  const int q = blockIdx.y;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)g.R * g.C4) return;
  const int r = (int)(idx / g.C4), c = (int)(idx - (long)r * g.C4) * 4;
  const float av = g.gscale ? g.a[q][r] * *g.gscale : g.a[q][r];
  const f32x4 bv = *reinterpret_cast<const f32x4*>(g.b[q] + c);
  f32x4* o = reinterpret_cast<f32x4*>(g.out[q] + (long)r * g.ld + c);
  *o = *o + av * bv;
}
}  // namespace

extern "C" int pvrl_gemv_rows_batched_f32(int n, const void** W, int w_is_bf16, int64_t ld, int64_t R, int64_t C, const float** x,
                                          const float* beta, float** y, const float* gscale, void* stream) {
  if (n <= 0 || R <= 0) return PVRL_OK;
  if (!W || !x || !y || !beta || C <= 0 || ld < C) return PVRL_EINVAL;
  for (int i0 = 0; i0 < n; i0 += SMALL_BATCH_MAX) {
    GemvBatch g = {};
    g.n = n - i0 < SMALL_BATCH_MAX ? n - i0 : SMALL_BATCH_MAX;
    g.R = (int)R; g.C = (int)C; g.w16 = w_is_bf16; g.ld = ld; g.gscale = gscale;
    for (int i = 0; i < g.n; ++i) {
      if (!W[i0 + i] || !x[i0 + i] || !y[i0 + i]) return PVRL_EINVAL;
      g.W[i] = W[i0 + i]; g.x[i] = x[i0 + i]; g.y[i] = y[i0 + i]; g.beta[i] = beta[i0 + i];
    }
    hipLaunchKernelGGL(gemv_rows_batched_kernel, dim3((unsigned)cdiv(R, 4), (unsigned)g.n), dim3(256), 0, (hipStream_t)stream, g);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

extern "C" int pvrl_rank1_add_batched_f32(int n, float** out, int64_t ld, const float** a, const float** b, int64_t R, int64_t C,
                                          const float* gscale, void* stream) {
  if (n <= 0 || R <= 0 || C <= 0) return PVRL_OK;
  if (!out || !a || !b || (C % 4) || (ld % 4)) return PVRL_EINVAL;
  for (int i0 = 0; i0 < n; i0 += SMALL_BATCH_MAX) {
    Rank1Batch g = {};
    g.n = n - i0 < SMALL_BATCH_MAX ? n - i0 : SMALL_BATCH_MAX;
    g.R = (int)R; g.C4 = (int)(C / 4); g.ld = ld; g.gscale = gscale;
    for (int i = 0; i < g.n; ++i) {
      if (!out[i0 + i] || !a[i0 + i] || !b[i0 + i]) return PVRL_EINVAL;
      g.out[i] = out[i0 + i]; g.a[i] = a[i0 + i]; g.b[i] = b[i0 + i];
    }
    hipLaunchKernelGGL(rank1_add_batched_kernel, dim3((unsigned)cdiv((long)R * g.C4, 256), (unsigned)g.n), dim3(256), 0, (hipStream_t)stream, g);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

extern "C" int pvrl_operand_dtype(void) { return PVRL_OPERAND_CODE; }
