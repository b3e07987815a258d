// HBM-bound kernels of the MViTv2 encoder path (reference lib/models/slowfast_mvit/):
//  * im2col of the 3-D patch-embed convolution (stem_helper.py:290-321, Conv3d k(3,7,7) s(2,4,4) p(1,3,3))
//  * LayerNorm for any width <= 768 with a padded leading dimension (widths 96 / 192 / 384 / 768, attention.py:502,524)
//  * attention_pool: depthwise 3x3x3 Conv3d + LayerNorm(head_dim) on q / k / v, cls token bypassing the conv
//    (attention.py:14-48,239-290), forward, data gradient, weight gradient
//  * max-pool skip connection (attention.py:537-552), forward and backward
//  * decomposed relative-position terms rel[q][j] = q . R_j (attention.py:67-159) and their gradients
// Token layout everywhere: rows [0, B*L) are the patch tokens ordered (b, t, h, w); rows [B*L, B*L+B) the cls tokens.
// Channel widths are padded to multiples of 128 (zero columns) so the bf16 MFMA GEMMs of gemm_nt.hip / gemm_tn.hip apply.
#include "common.h"
#include "../../include/pvrl.h"

namespace {

constexpr int HD = 96;   // head_dim of every MViTv2 block (96/1, 192/2, 384/4, 768/8)

inline unsigned grid_for(long total, int per_block = 256) {
  long b = (total + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 65535L * 16) b = 65535L * 16;
  return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------- im2col 3-D
struct Conv3dGeom {
  int B, Cin, T, H, W, kt, kh, kw, st, sh, sw, pt, ph, pw, To, Ho, Wo, K;   // K = Cin*kt*kh*kw
};

// out bf16 [(b, to, ho, wo)][ldo]; column k = ((c*kt + a)*kh + y)*kw + x (the flatten order of Conv3d.weight), zero for k >= K.
// Round 3.  The first form (1.0 ms for the stem of a 32-clip step) decomposed k with three run-time integer divisions per ELEMENT
// and fetched each tap inside `if (inside)`: eight guarded loads per thread = eight serial memory round trips.
// Generic kernel below: the (c, a, y, x) of every column come from an LDS table built once per workgroup, the eight taps of a
// thread are loaded unconditionally from clamped coordinates and masked by selects, 32-bit index arithmetic: 713 us -- now bound
// by its access pattern (every lane of a load touches a different input row: 64 cache lines per wave instruction).
constexpr int IM2COL_MAXK = 1024;
__global__ __launch_bounds__(256) void im2col3d_kernel(const float* __restrict__ frames, op_t* __restrict__ out,
                                                       Conv3dGeom g, long ldo) {
  __shared__ __attribute__((aligned(16))) unsigned lut[IM2COL_MAXK];          // column k -> c << 24 | a << 16 | y << 8 | x   (0xffffffff: k >= K)
  const int chunks = (int)(ldo >> 3);
  const long rows = (long)g.B * g.To * g.Ho * g.Wo;
  const long total = rows * chunks;
  const int khw = g.kh * g.kw, kvol = g.kt * khw;
  for (int k = threadIdx.x; k < (int)ldo; k += 256) {
    unsigned e = 0xffffffffu;
    if (k < g.K) {
      const int c = k / kvol;
      int rem = k - c * kvol;
      const int a = rem / khw;
      rem -= a * khw;
      const int y = rem / g.kw, x = rem - y * g.kw;
      e = ((unsigned)c << 24) | ((unsigned)a << 16) | ((unsigned)y << 8) | (unsigned)x;
    }
    lut[k] = e;
  }
  __syncthreads();
  const long plane = (long)g.H * g.W;
  // (32-bit index arithmetic: the host checks rows * chunks < 2^31 -- 64-bit divisions cost ~100 VALU instructions each)
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * 256u) {
    const unsigned row = idx / (unsigned)chunks;
    const int ch = (int)(idx - row * (unsigned)chunks);
    unsigned r = row;
    const unsigned r1 = r / (unsigned)g.Wo; const int wo = (int)(r - r1 * (unsigned)g.Wo); r = r1;
    const unsigned r2 = r / (unsigned)g.Ho; const int ho = (int)(r - r2 * (unsigned)g.Ho); r = r2;
    const unsigned r3 = r / (unsigned)g.To; const int to = (int)(r - r3 * (unsigned)g.To);
    const int b = (int)r3;
    const int t0 = to * g.st - g.pt, y0 = ho * g.sh - g.ph, x0 = wo * g.sw - g.pw;
    const float* fb = frames + (long)b * g.Cin * g.T * plane;
    const u32x4 l0 = *reinterpret_cast<const u32x4*>(&lut[ch * 8]);
    const u32x4 l1 = *reinterpret_cast<const u32x4*>(&lut[ch * 8 + 4]);
    float v[8];
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned q = e < 4 ? l0[e] : l1[e - 4];
      const int c = (int)(q >> 24) & 0x7f, ti = t0 + (int)((q >> 16) & 0xff), yi = y0 + (int)((q >> 8) & 0xff), xi = x0 + (int)(q & 0xff);
      ok[e] = q != 0xffffffffu && ti >= 0 && ti < g.T && yi >= 0 && yi < g.H && xi >= 0 && xi < g.W;
      const int tc = min(max(ti, 0), g.T - 1), yc = min(max(yi, 0), g.H - 1), xc = min(max(xi, 0), g.W - 1);
      v[e] = fb[((long)min(c, g.Cin - 1) * g.T + tc) * plane + (long)yc * g.W + xc];       // unconditional: in flight together
    }
    opx8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (op_t)(ok[e] ? v[e] : 0.f);
    *reinterpret_cast<opx8*>(out + (long)row * ldo + ch * 8) = o;
  }
}

// The stem's geometry (W % 4 == 0, W <= 240, Cin*kt*kh <= 64 input lines per output row, pw <= 4, kw <= 8): one workgroup per
// (b, to, ho) = one ROW of Wo output positions.  Phase 1 brings the Cin*kt*kh input rows that row of outputs touches into LDS as
// 16-bit values -- whole rows, float4 loads, fully coalesced, zero rows / zero borders for the padding -- phase 2 assembles the Wo
// output rows from LDS (a column's LDS offset from a table, no bounds checks left) and writes them as full 16-byte chunks.
constexpr int IMR_LINES = 64, IMR_PITCH = 256, IMR_LPAD = 4;
__global__ __launch_bounds__(256) void im2col3d_rows_kernel(const float* __restrict__ frames, op_t* __restrict__ out,
                                                            Conv3dGeom g, long ldo) {
  __shared__ __attribute__((aligned(16))) op_t lines[IMR_LINES][IMR_PITCH];
  __shared__ __attribute__((aligned(16))) unsigned koff[IM2COL_MAXK];          // column k -> byte offset of (line, x) inside `lines`; 0xffffffff: k >= K
  __shared__ unsigned lca[IMR_LINES];             // line -> c << 16 | a << 8 | y
  const int tid = threadIdx.x;
  const int nline = g.Cin * g.kt * g.kh;
  const int khw = g.kh * g.kw;
  for (int k = tid; k < (int)ldo; k += 256) {
    unsigned e = 0xffffffffu;
    if (k < g.K) {
      const int l = k / g.kw, x = k - l * g.kw;
      e = (unsigned)(l * IMR_PITCH + x) * 2u;
    }
    koff[k] = e;
  }
  if (tid < nline) {
    const int y = tid % g.kh, ca = tid / g.kh, a = ca % g.kt, c = ca / g.kt;
    lca[tid] = ((unsigned)c << 16) | ((unsigned)a << 8) | (unsigned)y;
  }
  for (int i = tid; i < IMR_LINES * IMR_PITCH / 2; i += 256) reinterpret_cast<unsigned*>(&lines[0][0])[i] = 0u;   // borders stay zero
  __syncthreads();
  (void)khw;
  const long plane = (long)g.H * g.W;
  const int w4 = g.W >> 2, chunks = (int)(ldo >> 3);
  const int lrow = tid >> 6, j = tid & 63;
  const int groups = g.B * g.To * g.Ho;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int ho = grp % g.Ho, bt = grp / g.Ho, to = bt % g.To, b = bt / g.To;
    // ---- phase 1: input rows -> LDS (lane j = one float4 of a row, four rows per pass) ----
    for (int l = lrow; l < nline; l += 4) {
      const unsigned q = lca[l];
      const int c = (int)(q >> 16), ti = to * g.st - g.pt + (int)((q >> 8) & 0xff), yi = ho * g.sh - g.ph + (int)(q & 0xff);
      const bool ok = ti >= 0 && ti < g.T && yi >= 0 && yi < g.H && j < w4;
      const float* src = frames + (((long)b * g.Cin + c) * g.T + min(max(ti, 0), g.T - 1)) * plane +
                         (long)min(max(yi, 0), g.H - 1) * g.W + 4 * min(j, w4 - 1);
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);                     // unconditional
      opx4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (op_t)(ok ? v[e] : 0.f);
      if (j < w4) *reinterpret_cast<opx4*>(&lines[l][IMR_LPAD + 4 * j]) = o;
    }
    __syncthreads();
    // ---- phase 2: Wo output rows x chunks of 8 columns ----
    for (int task = tid; task < g.Wo * chunks; task += 256) {
      const int wo = task / chunks, ch = task - wo * chunks;
      const unsigned base = (unsigned)(wo * g.sw - g.pw + IMR_LPAD) * 2u;
      const u32x4 k0 = *reinterpret_cast<const u32x4*>(&koff[ch * 8]);
      const u32x4 k1 = *reinterpret_cast<const u32x4*>(&koff[ch * 8 + 4]);
      opx8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned off = e < 4 ? k0[e] : k1[e - 4];
        const op_t v = *reinterpret_cast<const op_t*>(reinterpret_cast<const char*>(&lines[0][0]) + (off == 0xffffffffu ? 0u : off + base));
        o[e] = off == 0xffffffffu ? (op_t)0.f : v;
      }
      *reinterpret_cast<opx8*>(out + ((long)grp * g.Wo + wo) * ldo + ch * 8) = o;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------- LayerNorm (any C <= 768)
constexpr int LNG_MAX = 12;   // 64 lanes x 12 = 768

__device__ __forceinline__ void st_val(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_val(op_t* p, float v) { *p = (op_t)v; }
__device__ __forceinline__ float ld_val(const float* p) { return *p; }
__device__ __forceinline__ float ld_val(const op_t* p) { return (float)*p; }

// One wave per row, NJ = ceil(C / 64) channels per lane, RPI independent rows per loop iteration: with one row and twelve
// guarded chunks per iteration (the first version) a wave had two useful loads in flight at C = 96 and walked 440 scalar
// branch instructions per row.  Loads are unconditional (clamped column / row), masked by selects.
template <typename TO, int NJ, int RPI>
__global__ __launch_bounds__(256) void ln_g_fwd_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, TO* __restrict__ y, long ldy, int C, int Cpad,
                                                       float* __restrict__ mean, float* __restrict__ rstd, long M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[NJ], bt[NJ];
  int cc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    cc[j] = min(lane + 64 * j, C - 1);
    gm[j] = gamma[cc[j]]; bt[j] = beta[cc[j]];
  }
  const float invC = 1.f / (float)C;
  const long stride = (long)gridDim.x * 4;
  for (long row0 = (long)blockIdx.x * 4 + wave; row0 < M; row0 += stride * RPI) {
    float v[RPI][NJ];
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const long row = min(row0 + r * stride, M - 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) v[r][j] = x[row * ldx + cc[j]];
    }
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const long row = row0 + r * stride;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) s += (lane + 64 * j < C) ? v[r][j] : 0.f;
      const float mu = wave_sum(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float d = (lane + 64 * j < C) ? v[r][j] - mu : 0.f;
        q += d * d;
      }
      const float rs = rsqrtf(wave_sum(q) * invC + eps);
      if (row < M) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = lane + 64 * j;
          if (c < C) st_val(y + row * ldy + c, (v[r][j] - mu) * rs * gm[j] + bt[j]);
          else if (c < Cpad) st_val(y + row * ldy + c, 0.f);
        }
        if (lane == 0) {
          if (mean) mean[row] = mu;
          if (rstd) rstd[row] = rs;
        }
      }
    }
  }
}

// out[j] += sum_b part[b][j]  (j < half -> out0[j], else out1[j - half]); fixed order: 8 outputs x 32 slices of the
// block list per workgroup (four loads in flight per thread: 1,024 - 2,048 partial rows make this a latency chain),
// slices combined through LDS.  Replaces same-address fp32 atomics (bit-reproducible).
constexpr int PADD_OUT = 8, PADD_SL = 32;
__global__ __launch_bounds__(256) void partials_add_kernel(const float* __restrict__ part, int nblk, int n,
                                                           float* __restrict__ out0, float* __restrict__ out1, int half) {
  __shared__ float red[PADD_SL][PADD_OUT + 1];
  const int o = threadIdx.x & (PADD_OUT - 1), sl = threadIdx.x / PADD_OUT;
  const int j = blockIdx.x * PADD_OUT + o;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (j < n) {
    int b = sl;
    for (; b + 3 * PADD_SL < nblk; b += 4 * PADD_SL) {
      a0 += part[(long)b * n + j]; a1 += part[(long)(b + PADD_SL) * n + j];
      a2 += part[(long)(b + 2 * PADD_SL) * n + j]; a3 += part[(long)(b + 3 * PADD_SL) * n + j];
    }
    for (; b < nblk; b += PADD_SL) a0 += part[(long)b * n + j];
  }
  red[sl][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < PADD_SL; ++k) t += red[k][o];
    float* dst = j < half ? out0 + j : out1 + (j - half);
    *dst += t;
  }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gamma * dy ; dgamma += dy * xhat ; dbeta += dy
// (same row / channel decomposition as the forward: NJ channels per lane, RPI rows in flight, unconditional clamped loads)
template <typename TD, int NJ, int RPI, bool HAS_RES>
__global__ __launch_bounds__(256) void ln_g_bwd_kernel(const TD* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                       long ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ dres, long ldr, float* __restrict__ dx,
                                                       long lddx, op_t* __restrict__ dx16, long lddx16,
                                                       const float* __restrict__ rowscale16, int C, int Cpad,
                                                       float* __restrict__ part, long M) {
  __shared__ float red[2][4][64 * LNG_MAX / 4];   // reduced in four column quarters to stay small
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NJ], ab[NJ], gm[NJ];
  int cc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ag[j] = ab[j] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    cc[j] = min(lane + 64 * j, C - 1);
    gm[j] = gamma[cc[j]];
  }
  const float invC = 1.f / (float)C;
  const long stride = (long)gridDim.x * 4;
  for (long row0 = (long)blockIdx.x * 4 + wave; row0 < M; row0 += stride * RPI) {
    float d[RPI][NJ], xv[RPI][NJ], dr[RPI][NJ], mu[RPI], rs[RPI], r16[RPI];
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const long row = min(row0 + r * stride, M - 1);
      mu[r] = mean[row]; rs[r] = rstd[row];
      r16[r] = (dx16 && rowscale16) ? rowscale16[row] : 1.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        d[r][j] = ld_val(dy + row * lddy + cc[j]);
        xv[r][j] = x[row * ldx + cc[j]];
        dr[r][j] = HAS_RES ? dres[row * ldr + cc[j]] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const long row = row0 + r * stride;
      const bool live = row < M;
      float g[NJ], xh[NJ];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool ok = live && (lane + 64 * j < C);
        const float dj = ok ? d[r][j] : 0.f;
        xh[j] = ok ? (xv[r][j] - mu[r]) * rs[r] : 0.f;
        g[j] = dj * gm[j];
        ag[j] += dj * xh[j];
        ab[j] += dj;
        s1 += g[j];
        s2 += g[j] * xh[j];
      }
      s1 = wave_sum(s1) * invC;
      s2 = wave_sum(s2) * invC;
      if (live) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = lane + 64 * j;
          if (c < C) {
            const float o = rs[r] * (g[j] - s1 - xh[j] * s2) + dr[r][j];
            dx[row * lddx + c] = o;
            if (dx16) dx16[row * lddx16 + c] = (op_t)(o * r16[r]);      // the next GEMM's 16-bit operand: no separate cast pass
          } else if (c < Cpad) {
            dx[row * lddx + c] = 0.f;
            if (dx16) dx16[row * lddx16 + c] = (op_t)0.f;
          }
        }
      }
    }
  }
  // cross-wave reduction of the parameter gradients, three j at a time
#pragma unroll
  for (int j0 = 0; j0 < NJ; j0 += 3) {
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      if (j0 + jj < NJ) {
        red[0][wave][jj * 64 + lane] = ag[j0 + jj];
        red[1][wave][jj * 64 + lane] = ab[j0 + jj];
      }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int c = lane + 64 * (j0 + jj);
        if (j0 + jj < NJ && c < C) {
          const float a = red[0][0][jj * 64 + lane] + red[0][1][jj * 64 + lane] + red[0][2][jj * 64 + lane] + red[0][3][jj * 64 + lane];
          const float b = red[1][0][jj * 64 + lane] + red[1][1][jj * 64 + lane] + red[1][2][jj * 64 + lane] + red[1][3][jj * 64 + lane];
          part[(long)blockIdx.x * 2 * C + c] = a;          // per-workgroup partials, summed by partials_add_kernel
          part[(long)blockIdx.x * 2 * C + C + c] = b;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- attention_pool
struct PoolGeom {
  int B, H, T, Hh, Ww;      // input token grid (per clip) and heads
  int st, sh, sw;           // conv stride (kernel 3x3x3, padding 1)
  int To, Ho, Wo;
  long cls_row0;            // row of clip 0's cls token in the packed activation (= B*T*Hh*Ww)
  long ld;                  // leading dimension of the packed qkv activation
  int col0;                 // first column of this tensor (q / k / v) in the packed activation; head h adds h*96
};

__device__ __forceinline__ void ld6(const op_t* p, float* v) {
  const unsigned* u = reinterpret_cast<const unsigned*>(p);   // 4-byte aligned (6-channel groups of 2-byte values)
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const unsigned w = u[e];
    op_unpack2(w, v[2 * e], v[2 * e + 1]);
  }
}
// the raw 12 bytes / their unpacking, split so that a kernel can issue ALL its neighbour loads before the first use:
// a load guarded by `if (outside) continue;` makes the compiler wait for each one in turn (s_waitcnt vmcnt(0) per
// neighbour: nine serial round trips per frame, measured 77 % of the wave cycles waiting); loading a clamped address
// unconditionally and zeroing the words of an outside neighbour with a select keeps all nine in flight.
struct Raw6 { unsigned w[3]; };
__device__ __forceinline__ Raw6 ld6_raw(const op_t* p) {
  const unsigned* u = reinterpret_cast<const unsigned*>(p);
  Raw6 r;
#pragma unroll
  for (int e = 0; e < 3; ++e) r.w[e] = u[e];
  return r;
}
__device__ __forceinline__ void unpack6(const Raw6& r, bool ok, float* v) {
#pragma unroll
  for (int e = 0; e < 3; ++e) op_unpack2(ok ? r.w[e] : 0u, v[2 * e], v[2 * e + 1]);
}
__device__ __forceinline__ void st6(op_t* p, const float* v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    union { opx2 h; unsigned w; } x;
    x.h[0] = (op_t)v[2 * e]; x.h[1] = (op_t)v[2 * e + 1];
    u[e] = x.w;
  }
}
__device__ __forceinline__ float sum16(float v) {   // over the 16 lanes that share one token
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// y[bh][lo][96] = LayerNorm_96(conv3d_depthwise(x)[lo]) for lo < Lo, = LayerNorm_96(x_cls) for lo = Lo; conv output kept
// (bf16) for the backward.  16 lanes per output token, 6 channels per lane.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const op_t* __restrict__ qkv, PoolGeom g,
                                                       const float* __restrict__ w, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, op_t* __restrict__ y,
                                                       op_t* __restrict__ cbuf) {
  __shared__ float ws[27 * HD];
  for (int i = threadIdx.x; i < 27 * HD; i += 256) ws[i] = w[(i % HD) * 27 + i / HD];   // [tap][c]
  __syncthreads();
  const int sub = threadIdx.x & 15, c0 = sub * 6;
  const int Lo = g.To * g.Ho * g.Wo, L = g.T * g.Hh * g.Ww;
  const unsigned ntok = (unsigned)((long)g.B * g.H * (Lo + 1));     // < 2^27 (checked by the launcher): 32-bit index math
  float gm[6], bt[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; }
  // Workgroup b runs on XCD b % 8: dealt out in launch order, every XCD would touch every part of the clip and its 4 MiB L2
  // would have to hold the +-1 frame halo of ALL tokens in flight (~10 MB at 56 x 56 tokens per frame: measured 502 us for
  // block 0's q pooling, L2-miss-bound).  xcd_remap gives each XCD one contiguous run of tokens: its own three frames.
  const unsigned wg = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  for (unsigned tok = (wg * 256u + threadIdx.x) >> 4; tok < ntok; tok += (gridDim.x * 256u) >> 4) {
    const int lo = (int)(tok % (unsigned)(Lo + 1));
    const unsigned bh = tok / (unsigned)(Lo + 1);
    const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
    const int col = g.col0 + h * HD + c0;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lo == Lo) {
      ld6(qkv + (g.cls_row0 + b) * g.ld + col, acc);
    } else {
      const int xo = lo % g.Wo, yo = (lo / g.Wo) % g.Ho, to = lo / (g.Wo * g.Ho);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int ti = to * g.st - 1 + a;
        if (ti < 0 || ti >= g.T) continue;
#pragma unroll
        for (int yy = 0; yy < 3; ++yy) {
          const int yi = yo * g.sh - 1 + yy;
          if (yi < 0 || yi >= g.Hh) continue;
#pragma unroll
          for (int xx = 0; xx < 3; ++xx) {
            const int xi = xo * g.sw - 1 + xx;
            if (xi < 0 || xi >= g.Ww) continue;
            float v[6];
            ld6(qkv + ((long)b * L + ((long)ti * g.Hh + yi) * g.Ww + xi) * g.ld + col, v);
            const float* wt = ws + ((a * 3 + yy) * 3 + xx) * HD + c0;
#pragma unroll
            for (int e = 0; e < 6; ++e) acc[e] = fmaf(v[e], wt[e], acc[e]);
          }
        }
      }
    }
    st6(cbuf + (long)tok * HD + c0, acc);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) s += acc[e];
    const float mu = sum16(s) * (1.f / HD);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) q += (acc[e] - mu) * (acc[e] - mu);
    const float rs = rsqrtf(sum16(q) * (1.f / HD) + eps);
    float o[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) o[e] = (acc[e] - mu) * rs * gm[e] + bt[e];
    st6(y + (long)tok * HD + c0, o);
  }
}

// The same for temporal stride 1 (every MViTv2 pooling operator: stride (1, s, s)), sliding along t: the kernel above is
// VALU-bound, not memory-bound -- per output 27 x (address arithmetic + 12-byte load + unpack) next to the 162 FMAs (488 us
// for block 0's q pooling against ~100 us of memory time).  Here 16 lanes own one output COLUMN (b, h, yo, xo) and walk the
// T input frames once: the 9 in-plane neighbours of a frame are loaded ONCE and feed three running sums (the outputs of the
// frame before, this frame and the next: taps a = 2, 1, 0); a third of the loads, unpacks and address computations.
__device__ __forceinline__ void pool_ln_store(const float (&acc)[6], const float (&gm)[6], const float (&bt)[6], float eps,
                                              op_t* cdst, op_t* ydst) {
  st6(cdst, acc);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 6; ++e) s += acc[e];
  const float mu = sum16(s) * (1.f / HD);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 6; ++e) q += (acc[e] - mu) * (acc[e] - mu);
  const float rs = rsqrtf(sum16(q) * (1.f / HD) + eps);
  float o[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) o[e] = (acc[e] - mu) * rs * gm[e] + bt[e];
  st6(ydst, o);
}

template <bool DENSE>
__global__ __launch_bounds__(256) void pool_fwd_t_kernel(const op_t* __restrict__ qkv, PoolGeom g,
                                                         const float* __restrict__ w, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, op_t* __restrict__ y,
                                                         op_t* __restrict__ cbuf) {
  __shared__ float ws[27 * HD];
  for (int i = threadIdx.x; i < 27 * HD; i += 256) ws[i] = w[(i % HD) * 27 + i / HD];   // [tap][c]
  __syncthreads();
  const int sub = threadIdx.x & 15, c0 = sub * 6;
  const int HoWo = g.Ho * g.Wo, Lo = g.T * HoWo, L = g.T * g.Hh * g.Ww, plane = g.Hh * g.Ww;
  const unsigned ncol = (unsigned)((long)g.B * g.H * (HoWo + 1));      // + one "column" per (b, h) for the cls token
  float gm[6], bt[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; }
  const unsigned wg = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  for (unsigned colid = (wg * 256u + threadIdx.x) >> 4; colid < ncol; colid += (gridDim.x * 256u) >> 4) {
    const int pos = (int)(colid % (unsigned)(HoWo + 1));
    const unsigned bh = colid / (unsigned)(HoWo + 1);
    const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
    const int col = g.col0 + h * HD + c0;
    op_t* cdst = cbuf + (long)bh * (Lo + 1) * HD + c0;
    op_t* ydst = y + (long)bh * (Lo + 1) * HD + c0;
    if (pos == HoWo) {                                               // cls token: LayerNorm only
      float acc[6];
      ld6(qkv + (g.cls_row0 + b) * g.ld + col, acc);
      pool_ln_store(acc, gm, bt, eps, cdst + (long)Lo * HD, ydst + (long)Lo * HD);
      continue;
    }
    const int xo = pos % g.Wo, yo = pos / g.Wo;
    int noff[9];                                                      // row offset of the 9 in-plane neighbours, -1 = outside
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
      for (int xx = 0; xx < 3; ++xx) {
        const int yi = yo * g.sh - 1 + yy, xi = xo * g.sw - 1 + xx;
        noff[yy * 3 + xx] = (yi >= 0 && yi < g.Hh && xi >= 0 && xi < g.Ww) ? yi * g.Ww + xi : -1;
      }
    const op_t* base = qkv + (long)b * L * g.ld + col;
    float aP[6], aC[6], aN[6];                                        // running sums of outputs t-1, t, t+1
#pragma unroll
    for (int e = 0; e < 6; ++e) { aP[e] = 0.f; aC[e] = 0.f; aN[e] = 0.f; }
    for (int ti = 0; ti < g.T; ++ti) {
      const op_t* pb = base + (long)ti * plane * g.ld;
      // DENSE: all nine neighbour loads issued together from clamped addresses, an outside neighbour zeroed by a select --
      // the form for the strided (k / v) pools, whose few columns cannot hide nine serial round trips (36 -> 23 us);
      // otherwise the loads are guarded (one wait each, but 118 VGPRs = 4 waves / SIMD: ahead on the dense stride-1 planes)
      Raw6 raw[9];
      int wbase = c0;
      if constexpr (DENSE) {
#pragma unroll
        for (int n = 0; n < 9; ++n) raw[n] = ld6_raw(pb + (long)max(noff[n], 0) * g.ld);
        asm volatile("" : "+v"(wbase));           // opaque per frame: the 162 weights stay in LDS instead of 162 hoisted VGPRs
      }
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        float v[6];
        if constexpr (DENSE) {
          unpack6(raw[n], noff[n] >= 0, v);
        } else {
          if (noff[n] < 0) continue;
          ld6(pb + (long)noff[n] * g.ld, v);
        }
        const float* w0 = ws + n * HD + wbase;                         // tap (a, yy, xx) at [(a*9 + n)][c]
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          aN[e] = fmaf(v[e], w0[e], aN[e]);                            // a = 0: this frame is the one BEFORE output ti + 1
          aC[e] = fmaf(v[e], w0[9 * HD + e], aC[e]);                   // a = 1
          aP[e] = fmaf(v[e], w0[18 * HD + e], aP[e]);                  // a = 2: this frame is the one AFTER output ti - 1
        }
      }
      if (ti >= 1) {
        const long o = ((long)(ti - 1) * HoWo + pos) * HD;
        pool_ln_store(aP, gm, bt, eps, cdst + o, ydst + o);
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) { aP[e] = aC[e]; aC[e] = aN[e]; aN[e] = 0.f; }
    }
    const long o = ((long)(g.T - 1) * HoWo + pos) * HD;
    pool_ln_store(aP, gm, bt, eps, cdst + o, ydst + o);
  }
}

// LayerNorm backward of the pooled tensor: dc = dLN(dy | c) (bf16, same layout), dgamma / dbeta accumulated atomically;
// the cls token's dc goes straight to its row of the packed activation gradient (it bypassed the conv).
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(const op_t* __restrict__ dy, const op_t* __restrict__ cbuf,
                                                          PoolGeom g, const float* __restrict__ gamma, float eps,
                                                          op_t* __restrict__ dc, op_t* __restrict__ dqkv,
                                                          float* __restrict__ part) {
  __shared__ float red[2][16][HD];
  const int sub = threadIdx.x & 15, c0 = sub * 6, tl = threadIdx.x >> 4;
  const int Lo = g.To * g.Ho * g.Wo;
  const long ntok = (long)g.B * g.H * (Lo + 1);
  float gm[6], ag[6], ab[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) { gm[e] = gamma[c0 + e]; ag[e] = ab[e] = 0.f; }
  for (long tok = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; tok < ntok; tok += ((long)gridDim.x * 256) >> 4) {
    float c[6], d[6];
    ld6(cbuf + tok * HD + c0, c);
    ld6(dy + tok * HD + c0, d);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) s += c[e];
    const float mu = sum16(s) * (1.f / HD);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) q += (c[e] - mu) * (c[e] - mu);
    const float rs = rsqrtf(sum16(q) * (1.f / HD) + eps);
    float xh[6], gg[6], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      xh[e] = (c[e] - mu) * rs;
      gg[e] = d[e] * gm[e];
      ag[e] += d[e] * xh[e];
      ab[e] += d[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
    }
    s1 = sum16(s1) * (1.f / HD);
    s2 = sum16(s2) * (1.f / HD);
    float o[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) o[e] = rs * (gg[e] - s1 - xh[e] * s2);
    st6(dc + tok * HD + c0, o);
    const int lo = (int)(tok % (Lo + 1));
    if (lo == Lo) {
      const long bh = tok / (Lo + 1);
      const int h = (int)(bh % g.H), b = (int)(bh / g.H);
      st6(dqkv + (g.cls_row0 + b) * g.ld + g.col0 + h * HD + c0, o);
    }
  }
#pragma unroll
  for (int e = 0; e < 6; ++e) { red[0][tl][c0 + e] = ag[e]; red[1][tl][c0 + e] = ab[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * HD) {
    const int which = threadIdx.x / HD, c = threadIdx.x % HD;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) a += red[which][t][c];
    part[(long)blockIdx.x * 2 * HD + which * HD + c] = a;      // summed by partials_add_kernel (dgamma then dbeta)
  }
}

// data gradient of the depthwise conv: dX[b, l_in, h, c] = sum_taps dc[out(l_in, tap)][c] * w[c][tap]
// S > 0: spatial stride S (a power of two) with temporal stride 1 -- every shipped MViT config -- so the 39 stride
// divisions / remainders per token become shifts and masks and the token decomposition is 32-bit; S = 0: any strides.
template <int S>
__global__ __launch_bounds__(256) void pool_dgrad_kernel(const op_t* __restrict__ dc, PoolGeom g,
                                                         const float* __restrict__ w, op_t* __restrict__ dqkv) {
  __shared__ float ws[27 * HD];
  for (int i = threadIdx.x; i < 27 * HD; i += 256) ws[i] = w[(i % HD) * 27 + i / HD];
  __syncthreads();
  const int sub = threadIdx.x & 15, c0 = sub * 6;
  const int Lo = g.To * g.Ho * g.Wo, L = g.T * g.Hh * g.Ww;
  const int st = S ? 1 : g.st, sh = S ? S : g.sh, sw = S ? S : g.sw;
  const unsigned ntok = (unsigned)((long)g.B * g.H * L);          // < 2^31 (checked by the launcher)
  const unsigned wg = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);     // contiguous tokens per XCD (see pool_fwd_kernel)
  for (unsigned tok = (wg * 256u + threadIdx.x) >> 4; tok < ntok; tok += (gridDim.x * 256u) >> 4) {
    const unsigned l = tok % (unsigned)L, bh = tok / (unsigned)L;
    const unsigned h = bh % (unsigned)g.H, b = bh / (unsigned)g.H;
    const int xi = (int)(l % (unsigned)g.Ww), yi = (int)((l / (unsigned)g.Ww) % (unsigned)g.Hh), ti = (int)(l / (unsigned)(g.Ww * g.Hh));
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int tn = ti + 1 - a;
      if (tn < 0 || (S ? 0 : tn % st)) continue;
      const int to = S ? tn : tn / st;
      if (to >= g.To) continue;
#pragma unroll
      for (int yy = 0; yy < 3; ++yy) {
        const int yn = yi + 1 - yy;
        if (yn < 0 || (S ? (yn & (S - 1)) : yn % sh)) continue;
        const int yo = S ? yn / S : yn / sh;
        if (yo >= g.Ho) continue;
#pragma unroll
        for (int xx = 0; xx < 3; ++xx) {
          const int xn = xi + 1 - xx;
          if (xn < 0 || (S ? (xn & (S - 1)) : xn % sw)) continue;
          const int xo = S ? xn / S : xn / sw;
          if (xo >= g.Wo) continue;
          float v[6];
          ld6(dc + ((long)bh * (Lo + 1) + ((long)to * g.Ho + yo) * g.Wo + xo) * HD + c0, v);
          const float* wt = ws + ((a * 3 + yy) * 3 + xx) * HD + c0;
#pragma unroll
          for (int e = 0; e < 6; ++e) acc[e] = fmaf(v[e], wt[e], acc[e]);
        }
      }
    }
    st6(dqkv + ((long)b * L + l) * g.ld + g.col0 + h * HD + c0, acc);
  }
}

// The same for temporal stride 1, sliding along t (see pool_fwd_t_kernel): 16 lanes own one INPUT column (b, h, yi, xi); the
// (at most 9) outputs of a frame that touch it are loaded once and feed the running sums of dX at t - 1, t and t + 1.
// S = the spatial stride as a compile-time constant (1 / 2 / 4 / 8; 0: any): the nine `% stride`, `/ stride` pairs of a column's
// set-up become masks and shifts -- with run-time strides they were ~540 of the ~2,800 VALU instructions per column (issue-bound kernel).
template <bool DENSE, int S>
__global__ __launch_bounds__(256) void pool_dgrad_t_kernel(const op_t* __restrict__ dc, PoolGeom g,
                                                           const float* __restrict__ w, op_t* __restrict__ dqkv) {
  __shared__ float ws[27 * HD];
  for (int i = threadIdx.x; i < 27 * HD; i += 256) ws[i] = w[(i % HD) * 27 + i / HD];
  __syncthreads();
  const int sub = threadIdx.x & 15, c0 = sub * 6;
  const int HoWo = g.Ho * g.Wo, Lo = g.T * HoWo, plane = g.Hh * g.Ww, L = g.T * plane;
  const unsigned ncol = (unsigned)((long)g.B * g.H * plane);
  const unsigned wg = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  for (unsigned colid = (wg * 256u + threadIdx.x) >> 4; colid < ncol; colid += (gridDim.x * 256u) >> 4) {
    const int pos = (int)(colid % (unsigned)plane);
    const unsigned bh = colid / (unsigned)plane;
    const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
    const int xi = pos % g.Ww, yi = pos / g.Ww;
    int noff[9];                                   // output position (yo * Wo + xo) reached through tap (yy, xx), -1 = none
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
      for (int xx = 0; xx < 3; ++xx) {
        const int yn = yi + 1 - yy, xn = xi + 1 - xx;
        const int sh = S ? S : g.sh, sw = S ? S : g.sw;
        const bool ok = yn >= 0 && xn >= 0 && (yn % sh) == 0 && (xn % sw) == 0 && yn / sh < g.Ho && xn / sw < g.Wo;
        noff[yy * 3 + xx] = ok ? (yn / sh) * g.Wo + xn / sw : -1;
      }
    const op_t* base = dc + (long)bh * (Lo + 1) * HD + c0;
    op_t* dst = dqkv + ((long)b * L + pos) * g.ld + g.col0 + h * HD + c0;
    float aP[6], aC[6], aN[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) { aP[e] = 0.f; aC[e] = 0.f; aN[e] = 0.f; }
    for (int to = 0; to < g.T; ++to) {
      const op_t* pb = base + (long)to * HoWo * HD;
      // DENSE (spatial stride 1: all nine taps exist away from the border): see pool_fwd_t_kernel; with stride 2 / 4 / 8 an input meets
      // at most four / one or two outputs and skipping the others (guarded loads) does less work (measured per stride)
      Raw6 raw[9];
      int wbase = c0;
      if constexpr (DENSE) {
#pragma unroll
        for (int n = 0; n < 9; ++n) raw[n] = ld6_raw(pb + (long)max(noff[n], 0) * HD);
        asm volatile("" : "+v"(wbase));
      }
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        float v[6];
        if constexpr (DENSE) {
          unpack6(raw[n], noff[n] >= 0, v);
        } else {
          if (noff[n] < 0) continue;
          ld6(pb + (long)noff[n] * HD, v);
        }
        const float* w0 = ws + n * HD + wbase;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          aP[e] = fmaf(v[e], w0[e], aP[e]);                            // a = 0: output frame `to` reads input frame to - 1
          aC[e] = fmaf(v[e], w0[9 * HD + e], aC[e]);                   // a = 1
          aN[e] = fmaf(v[e], w0[18 * HD + e], aN[e]);                  // a = 2: ... and input frame to + 1
        }
      }
      if (to >= 1) st6(dst + (long)(to - 1) * plane * g.ld, aP);
#pragma unroll
      for (int e = 0; e < 6; ++e) { aP[e] = aC[e]; aC[e] = aN[e]; aN[e] = 0.f; }
    }
    st6(dst + (long)(g.T - 1) * plane * g.ld, aP);
  }
}

// weight gradient of the depthwise conv: dW[c][tap] += sum_{b,h,out} dc[out][c] * x[in(out, tap)][c]
// block = 8 token lanes x 24 channel quads (8-byte loads); each thread keeps 27 x 4 sums in registers (no atomics).  The 27 neighbour
// loads of a token are unconditional (clamped address, 0/1 mask) so they are all in flight together.
constexpr int PW_LANES = 8, PW_CQ = HD / 4;
__device__ __forceinline__ f32x4 ld4bf(const op_t* p) {
  const u32x2 w = *reinterpret_cast<const u32x2*>(p);
  f32x4 r;
  float a, b;
  op_unpack2(w[0], a, b); r[0] = a; r[1] = b;
  op_unpack2(w[1], a, b); r[2] = a; r[3] = b;
  return r;
}
__global__ __launch_bounds__(PW_CQ * PW_LANES) void pool_wgrad_kernel(const op_t* __restrict__ dc,
                                                                      const op_t* __restrict__ qkv, PoolGeom g,
                                                                      float* __restrict__ part) {
  __shared__ float red[27][HD];
  const int cq = threadIdx.x % PW_CQ, tl = threadIdx.x / PW_CQ, c0 = cq * 4;
  const int Lo = g.To * g.Ho * g.Wo, L = g.T * g.Hh * g.Ww;
  const long ntok = (long)g.B * g.H * Lo;
  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 27 * HD; i += PW_CQ * PW_LANES) red[i / HD][i % HD] = 0.f;
  // each workgroup sums ONE contiguous run of tokens (its 8 token lanes interleaved inside it), runs dealt to the XCDs in
  // contiguous chunks: neighbouring outputs share 2/3 of their 27 inputs, and an XCD's L2 sees only its own frames
  const long wgl = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const long chunk = (ntok + gridDim.x - 1) / gridDim.x;
  const long tend = min(ntok, (wgl + 1) * chunk);
  for (long tok = wgl * chunk + tl; tok < tend; tok += PW_LANES) {
    const int lo = (int)(tok % Lo);
    const long bh = tok / Lo;
    const int h = (int)(bh % g.H), b = (int)(bh / g.H);
    const int xo = lo % g.Wo, yo = (lo / g.Wo) % g.Ho, to = lo / (g.Wo * g.Ho);
    const f32x4 d = ld4bf(dc + (bh * (Lo + 1) + lo) * HD + c0);
    const op_t* xb = qkv + (long)b * L * g.ld + g.col0 + h * HD + c0;
    f32x4 xv[27];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ti = to * g.st - 1 + a;
      const int tc = min(max(ti, 0), g.T - 1);
#pragma unroll
      for (int yy = 0; yy < 3; ++yy) {
        const int yi = yo * g.sh - 1 + yy;
        const int yc = min(max(yi, 0), g.Hh - 1);
#pragma unroll
        for (int xx = 0; xx < 3; ++xx) {
          const int xi = xo * g.sw - 1 + xx;
          const int xc = min(max(xi, 0), g.Ww - 1);
          const float ok = (ti == tc && yi == yc && xi == xc) ? 1.f : 0.f;
          xv[(a * 3 + yy) * 3 + xx] = ld4bf(xb + (((long)tc * g.Hh + yc) * g.Ww + xc) * g.ld) * ok;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] += d * xv[t];
  }
  __syncthreads();
  for (int k = 0; k < PW_LANES; ++k) {      // the 8 token lanes add their sums one after the other: a fixed order
    if (tl == k) {
#pragma unroll
      for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[t][c0 + e] += acc[t][e];
    }
    __syncthreads();
  }
  float* mine = part + (long)blockIdx.x * (27 * HD);
  for (int i = threadIdx.x; i < 27 * HD; i += PW_CQ * PW_LANES) mine[i] = red[i / HD][i % HD];
}

// The same for temporal stride 1, sliding along t: a token lane walks the T frames of one output COLUMN (b, h, yo, xo); the 9
// in-plane inputs of frame ti are loaded once and meet the conv outputs' gradients of frames ti + 1, ti, ti - 1 (taps a = 0, 1,
// 2), which stay in registers: 10 loads per output instead of 28, and no 27-vector of inputs to keep.
__global__ __launch_bounds__(PW_CQ * PW_LANES) void pool_wgrad_t_kernel(const op_t* __restrict__ dc,
                                                                        const op_t* __restrict__ qkv, PoolGeom g,
                                                                        float* __restrict__ part) {
  __shared__ float red[27][HD];
  const int cq = threadIdx.x % PW_CQ, tl = threadIdx.x / PW_CQ, c0 = cq * 4;
  const int HoWo = g.Ho * g.Wo, Lo = g.T * HoWo, plane = g.Hh * g.Ww, L = g.T * plane;
  const unsigned ncol = (unsigned)((long)g.B * g.H * HoWo);       // < 2^27 (checked by the launcher): 32-bit column arithmetic --
  f32x4 acc[27];                                                  // four 64-bit divisions per column were ~half of a column's instructions
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 27 * HD; i += PW_CQ * PW_LANES) red[i / HD][i % HD] = 0.f;
  const unsigned wgl = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const unsigned chunk = (ncol + gridDim.x - 1) / gridDim.x;
  const unsigned cend = min(ncol, (wgl + 1) * chunk);
  for (unsigned colid = wgl * chunk + tl; colid < cend; colid += PW_LANES) {
    const int pos = (int)(colid % (unsigned)HoWo);
    const unsigned bh = colid / (unsigned)HoWo;
    const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
    const int xo = pos % g.Wo, yo = pos / g.Wo;
    int noff[9];
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
      for (int xx = 0; xx < 3; ++xx) {
        const int yi = yo * g.sh - 1 + yy, xi = xo * g.sw - 1 + xx;
        noff[yy * 3 + xx] = (yi >= 0 && yi < g.Hh && xi >= 0 && xi < g.Ww) ? yi * g.Ww + xi : -1;
      }
    const op_t* xb = qkv + (long)b * L * g.ld + g.col0 + h * HD + c0;
    const op_t* db = dc + ((long)bh * (Lo + 1) + pos) * HD + c0;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dP = zero, dC = ld4bf(db), dN = g.T > 1 ? ld4bf(db + (long)HoWo * HD) : zero;   // dc of output frames ti-1, ti, ti+1
    // two frames per iteration: 18 neighbour loads (+ the two dc rows that come into reach) in flight instead of 9 -- at 2 waves
    // per SIMD (108 accumulator registers) the kernel is bound by its memory round trips, one per iteration
    for (int ti = 0; ti < g.T; ti += 2) {
      const int t1 = min(ti + 1, g.T - 1), t2 = min(ti + 2, g.T - 1), t3 = min(ti + 3, g.T - 1);
      const bool has1 = ti + 1 < g.T;
      const op_t* pb0 = xb + (long)ti * plane * g.ld;
      const op_t* pb1 = xb + (long)t1 * plane * g.ld;
      u32x2 raw0[9], raw1[9];
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const long o = (long)max(noff[n], 0) * g.ld;
        raw0[n] = *reinterpret_cast<const u32x2*>(pb0 + o);
        raw1[n] = *reinterpret_cast<const u32x2*>(pb1 + o);
      }
      const f32x4 l2 = ld4bf(db + (long)t2 * HoWo * HD), l3 = ld4bf(db + (long)t3 * HoWo * HD);
      const f32x4 d2 = ti + 2 < g.T ? l2 : zero, d3 = ti + 3 < g.T ? l3 : zero;
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const bool ok = noff[n] >= 0;
        f32x4 xv;
        { float a, b2; op_unpack2(ok ? raw0[n][0] : 0u, a, b2); xv[0] = a; xv[1] = b2; op_unpack2(ok ? raw0[n][1] : 0u, a, b2); xv[2] = a; xv[3] = b2; }
        acc[n] += dN * xv;                 // tap a = 0: frame ti is the first input frame of output ti + 1
        acc[9 + n] += dC * xv;             // a = 1
        acc[18 + n] += dP * xv;            // a = 2
      }
      dP = dC; dC = dN; dN = d2;
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const bool ok = has1 && noff[n] >= 0;
        f32x4 xv;
        { float a, b2; op_unpack2(ok ? raw1[n][0] : 0u, a, b2); xv[0] = a; xv[1] = b2; op_unpack2(ok ? raw1[n][1] : 0u, a, b2); xv[2] = a; xv[3] = b2; }
        acc[n] += dN * xv;
        acc[9 + n] += dC * xv;
        acc[18 + n] += dP * xv;
      }
      dP = dC; dC = dN; dN = d3;
    }
  }
  __syncthreads();
  for (int k = 0; k < PW_LANES; ++k) {      // the 8 token lanes add their sums one after the other: a fixed order
    if (tl == k) {
#pragma unroll
      for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[t][c0 + e] += acc[t][e];
    }
    __syncthreads();
  }
  float* mine = part + (long)blockIdx.x * (27 * HD);
  for (int i = threadIdx.x; i < 27 * HD; i += PW_CQ * PW_LANES) mine[i] = red[i / HD][i % HD];
}

// dw[c][tap] += sum over workgroups of part[wg][tap][c] in a fixed order (deterministic): 16 outputs x 16 slices of
// the workgroup list per block, slices combined through LDS
constexpr int PW_MAX_WG = 2048;
__global__ __launch_bounds__(256) void pool_wgrad_reduce_kernel(const float* __restrict__ part, int nwg, float* __restrict__ dw) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + o;            // 27 * 96 = 2592 = 162 * 16
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = sl;
  for (; b + 48 < nwg; b += 64) {
    a0 += part[(long)b * (27 * HD) + i];
    a1 += part[(long)(b + 16) * (27 * HD) + i];
    a2 += part[(long)(b + 32) * (27 * HD) + i];
    a3 += part[(long)(b + 48) * (27 * HD) + i];
  }
  for (; b < nwg; b += 16) a0 += part[(long)b * (27 * HD) + i];
  red[sl][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 16) {
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) a += red[t][threadIdx.x];
    dw[(i % HD) * 27 + i / HD] += a;
  }
}

// ------------------------------------------------------------------------------------------------- max-pool skip
struct MaxPoolGeom {
  int B, T, H, W, k, s, Ho, Wo, C;   // kernel (1,k,k), stride (1,s,s), padding (0,k/2,k/2)
  long ldi, ldo;
};

// out rows: (b, t, ho, wo) then the B cls rows (copied).  4 channels per thread.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, MaxPoolGeom g,
                                                          float* __restrict__ y, unsigned char* __restrict__ amax) {
  const int c4n = g.C >> 2;
  const long Lo = (long)g.T * g.Ho * g.Wo, L = (long)g.T * g.H * g.W;
  const long rows = g.B * Lo + g.B;
  const unsigned total = (unsigned)(rows * c4n);               // < 2^31 (checked by the launcher): 32-bit index arithmetic
  const int pad = g.k / 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const int c4 = (int)(idx % (unsigned)c4n);
    const long row = idx / (unsigned)c4n;
    f32x4 m;
    if (row >= g.B * Lo) {
      m = *reinterpret_cast<const f32x4*>(x + (g.B * L + (row - g.B * Lo)) * g.ldi + c4 * 4);
    } else {
      const unsigned r32 = (unsigned)row;
      const int wo = (int)(r32 % (unsigned)g.Wo), ho = (int)((r32 / (unsigned)g.Wo) % (unsigned)g.Ho);
      const long bt = r32 / (unsigned)(g.Wo * g.Ho);   // b*T + t
      m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int arg[4] = {-1, -1, -1, -1};
      for (int yy = 0; yy < g.k; ++yy) {
        const int yi = ho * g.s - pad + yy;
        if (yi < 0 || yi >= g.H) continue;
        for (int xx = 0; xx < g.k; ++xx) {
          const int xi = wo * g.s - pad + xx;
          if (xi < 0 || xi >= g.W) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((bt * g.H + yi) * g.W + xi) * g.ldi + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (v[e] > m[e] || arg[e] < 0) { m[e] = v[e]; arg[e] = yy * g.k + xx; }   // first maximum in scan order
        }
      }
      if (amax) {      // window-local position of each channel's winner: the backward routes by it instead of re-scanning
        const unsigned w = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
        *reinterpret_cast<unsigned*>(amax + row * g.C + c4 * 4) = w;
      }
    }
    *reinterpret_cast<f32x4*>(y + row * g.ldo + c4 * 4) = m;
  }
}

// dx = routed dy: every output sends its gradient to the FIRST maximum of its window in scan order (torch max_pool3d
// keeps the first `val > maxval`); cls rows are copied.  Written as a GATHER -- each input position re-scans the (at most
// 2 x 2) windows that contain it and adds, in a fixed order, the gradients of those it wins -- so there are no atomics
// (bit-reproducible) and dx needs no zero fill.  4 channels per thread.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          MaxPoolGeom g, float* __restrict__ dx,
                                                          const unsigned char* __restrict__ amax) {
  const int c4n = g.C >> 2;
  const long Lo = (long)g.T * g.Ho * g.Wo, L = (long)g.T * g.H * g.W;
  const long rows = g.B * L + g.B;
  const unsigned total = (unsigned)(rows * c4n);               // < 2^31 (checked by the launcher): 32-bit index arithmetic
  const int pad = g.k / 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const int c = (int)(idx % (unsigned)c4n) * 4;
    const long row = idx / (unsigned)c4n;
    if (row >= g.B * L) {
      *reinterpret_cast<f32x4*>(dx + row * g.ldi + c) =
          *reinterpret_cast<const f32x4*>(dy + (g.B * Lo + (row - g.B * L)) * g.ldo + c);
      continue;
    }
    const unsigned r32 = (unsigned)row;
    const int xi = (int)(r32 % (unsigned)g.W), yi = (int)((r32 / (unsigned)g.W) % (unsigned)g.H);
    const long bt = r32 / (unsigned)(g.W * g.H);
    const int me = yi * g.W + xi;
    int ho0 = (yi + pad - g.k + 1 + g.s - 1) / g.s, ho1 = (yi + pad) / g.s;     // windows with yi in [ho*s-pad, ho*s-pad+k-1]
    int wo0 = (xi + pad - g.k + 1 + g.s - 1) / g.s, wo1 = (xi + pad) / g.s;
    if (yi + pad - g.k + 1 < 0) ho0 = 0;
    if (xi + pad - g.k + 1 < 0) wo0 = 0;
    ho1 = ho1 < g.Ho - 1 ? ho1 : g.Ho - 1;
    wo1 = wo1 < g.Wo - 1 ? wo1 : g.Wo - 1;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (amax) {       // the forward recorded every window's winner (1 byte per channel): 4 bytes + 16 bytes of dy per window
      for (int ho = ho0; ho <= ho1; ++ho)
        for (int wo = wo0; wo <= wo1; ++wo) {
          const long orow = (bt * g.Ho + ho) * g.Wo + wo;
          const unsigned w = *reinterpret_cast<const unsigned*>(amax + orow * g.C + c);
          const unsigned mine = (unsigned)((yi - (ho * g.s - pad)) * g.k + (xi - (wo * g.s - pad)));
          const f32x4 d = *reinterpret_cast<const f32x4*>(dy + orow * g.ldo + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (((w >> (8 * e)) & 255u) == mine) acc[e] += d[e];
        }
      *reinterpret_cast<f32x4*>(dx + row * g.ldi + c) = acc;
      continue;
    }
    for (int ho = ho0; ho <= ho1; ++ho)
      for (int wo = wo0; wo <= wo1; ++wo) {
        f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {-1, -1, -1, -1};
        for (int yy = 0; yy < g.k; ++yy) {
          const int y2 = ho * g.s - pad + yy;
          if (y2 < 0 || y2 >= g.H) continue;
          for (int xx = 0; xx < g.k; ++xx) {
            const int x2 = wo * g.s - pad + xx;
            if (x2 < 0 || x2 >= g.W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((bt * g.H + y2) * g.W + x2) * g.ldi + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (v[e] > m[e] || arg[e] < 0) { m[e] = v[e]; arg[e] = y2 * g.W + x2; }
          }
        }
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((bt * g.Ho + ho) * g.Wo + wo) * g.ldo + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (arg[e] == me) acc[e] += d[e];
      }
    *reinterpret_cast<f32x4*>(dx + row * g.ldi + c) = acc;
  }
}

// ------------------------------------------------------------------------------------------------- misc
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                     long ldo, long R, int C, float beta) {
  const long total = R * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const float v = in[r * ldi + c];
    out[r * ldo + c] = beta != 0.f ? beta * out[r * ldo + c] + v : v;
  }
}

}  // namespace

extern "C" int pvrl_im2col3d_bf16(const float* frames, int64_t B, int64_t Cin, int64_t T, int64_t H, int64_t W,
                                  int64_t kt, int64_t kh, int64_t kw, int64_t st, int64_t sh, int64_t sw, int64_t pt,
                                  int64_t ph, int64_t pw, void* out, int64_t ldo, void* stream) {
  if (B <= 0) return PVRL_OK;
  if (!frames || !out || (ldo % 8) || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 || sh <= 0 || sw <= 0) return PVRL_EINVAL;
  Conv3dGeom g;
  g.B = (int)B; g.Cin = (int)Cin; g.T = (int)T; g.H = (int)H; g.W = (int)W;
  g.kt = (int)kt; g.kh = (int)kh; g.kw = (int)kw; g.st = (int)st; g.sh = (int)sh; g.sw = (int)sw;
  g.pt = (int)pt; g.ph = (int)ph; g.pw = (int)pw;
  g.To = (int)((T + 2 * pt - kt) / st + 1); g.Ho = (int)((H + 2 * ph - kh) / sh + 1); g.Wo = (int)((W + 2 * pw - kw) / sw + 1);
  g.K = (int)(Cin * kt * kh * kw);
  const long nrows = (long)B * g.To * g.Ho * g.Wo;
  const long total = nrows * (ldo >> 3);
  if (ldo < g.K || ldo > IM2COL_MAXK || (ldo % 8) || g.kt > 255 || g.kh > 255 || g.kw > 255 || Cin > 127 || total >= (1L << 31))
    return PVRL_EINVAL;
  const int nline = (int)(Cin * g.kt * g.kh);
  const bool rows_form = (W % 4 == 0) && W + IMR_LPAD + g.kw <= IMR_PITCH && nline <= IMR_LINES && g.pw <= IMR_LPAD && g.kw <= 8 &&
                         (g.Wo - 1) * g.sw - g.pw + g.kw - 1 + IMR_LPAD < IMR_PITCH;
  if (rows_form) {
    long wgs = (long)B * g.To * g.Ho;
    if (wgs > 8192) wgs = 8192;
    hipLaunchKernelGGL(im2col3d_rows_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, frames, (op_t*)out, g, (long)ldo);
  } else {
    hipLaunchKernelGGL(im2col3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames, (op_t*)out, g, (long)ldo);
  }
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_layernorm_g_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                    void* y, int64_t ldy, int y_is_f32, int64_t M, int64_t C, int64_t Cpad, float* mean,
                                    float* rstd, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!x || !gamma || !beta || !y || C <= 0 || C > 64 * LNG_MAX || Cpad < C || Cpad > 64 * LNG_MAX || ldx < C || ldy < Cpad)
    return PVRL_EINVAL;
  // NJ = channels per lane (Cpad <= 64 NJ: the zero padding columns are written too); RPI rows in flight per wave
  const int nj = (int)((Cpad + 63) / 64);
#define LN_FWD(TO, NJ, RPI)                                                                                              \
  do {                                                                                                                   \
    long wgs = (M + 4 * RPI - 1) / (4 * RPI);                                                                            \
    if (wgs > 8192) wgs = 8192;                                                                                          \
    hipLaunchKernelGGL((ln_g_fwd_kernel<TO, NJ, RPI>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, x,       \
                       (long)ldx, gamma, beta, eps, (TO*)y, (long)ldy, (int)C, (int)Cpad, mean, rstd, (long)M);         \
  } while (0)
#define LN_FWD_T(TO)                                                                                                     \
  do {                                                                                                                   \
    if (nj <= 2) LN_FWD(TO, 2, 4); else if (nj <= 4) LN_FWD(TO, 4, 4); else if (nj <= 6) LN_FWD(TO, 6, 2);               \
    else LN_FWD(TO, 12, 1);                                                                                              \
  } while (0)
  if (y_is_f32) LN_FWD_T(float); else LN_FWD_T(op_t);
#undef LN_FWD_T
#undef LN_FWD
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int64_t pvrl_layernorm_g_bwd_workspace_bytes(int64_t M, int64_t C) {
  int64_t blocks = (M + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return blocks * 2 * C * (int64_t)sizeof(float);
}

extern "C" int pvrl_layernorm_g_bwd(const void* dy, int64_t lddy, int dy_is_f32, const float* x, int64_t ldx,
                                    const float* mean, const float* rstd, const float* gamma, const float* dres,
                                    int64_t ldr, float* dx, int64_t lddx, void* dx16, int64_t lddx16,
                                    const float* rowscale16, int64_t M, int64_t C, int64_t Cpad, float* dgamma, float* dbeta,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace || C <= 0 || C > 64 * LNG_MAX ||
      Cpad < C || Cpad > 64 * LNG_MAX || (dx16 && lddx16 < Cpad) ||
      workspace_bytes < pvrl_layernorm_g_bwd_workspace_bytes(M, C))
    return PVRL_EINVAL;
  long blocks = (M + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  float* part = (float*)workspace;
  const int nj = (int)((Cpad + 63) / 64);
#define LN_BWD(TD, NJ, RPI, RES)                                                                                         \
  hipLaunchKernelGGL((ln_g_bwd_kernel<TD, NJ, RPI, RES>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,    \
                     (const TD*)dy, (long)lddy, x, (long)ldx, mean, rstd, gamma, dres, (long)ldr, dx, (long)lddx,        \
                     (op_t*)dx16, (long)lddx16, rowscale16, (int)C, (int)Cpad, part, (long)M)
#define LN_BWD_R(TD, RES)                                                                                                \
  do {                                                                                                                   \
    if (nj <= 2) LN_BWD(TD, 2, 4, RES); else if (nj <= 4) LN_BWD(TD, 4, 2, RES); else if (nj <= 6) LN_BWD(TD, 6, 2, RES); \
    else LN_BWD(TD, 12, 1, RES);                                                                                         \
  } while (0)
  if (dy_is_f32) { if (dres) LN_BWD_R(float, true); else LN_BWD_R(float, false); }
  else { if (dres) LN_BWD_R(op_t, true); else LN_BWD_R(op_t, false); }
#undef LN_BWD_R
#undef LN_BWD
  PVRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(partials_add_kernel, dim3((unsigned)((2 * C + PADD_OUT - 1) / PADD_OUT)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)part, (int)blocks, (int)(2 * C), dgamma, dbeta, (int)C);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

static int pool_geom(PoolGeom& g, int64_t B, int64_t H, int64_t T, int64_t Hh, int64_t Ww, int64_t st, int64_t sh,
                     int64_t sw, int64_t ld, int64_t col0) {
  if (B <= 0 || H <= 0 || T <= 0 || Hh <= 0 || Ww <= 0 || st <= 0 || sh <= 0 || sw <= 0 || (ld % 4) || (col0 % 4))
    return PVRL_EINVAL;
  g.B = (int)B; g.H = (int)H; g.T = (int)T; g.Hh = (int)Hh; g.Ww = (int)Ww;
  g.st = (int)st; g.sh = (int)sh; g.sw = (int)sw;
  g.To = (int)((T + 2 - 3) / st + 1); g.Ho = (int)((Hh + 2 - 3) / sh + 1); g.Wo = (int)((Ww + 2 - 3) / sw + 1);
  g.cls_row0 = (long)B * T * Hh * Ww;
  g.ld = (long)ld; g.col0 = (int)col0;
  return PVRL_OK;
}

extern "C" int pvrl_mvit_pool_fwd(const void* qkv, int64_t ld, int64_t col0, int64_t B, int64_t H, int64_t T, int64_t Hh,
                                  int64_t Ww, int64_t st, int64_t sh, int64_t sw, const float* w, const float* gamma,
                                  const float* beta, float eps, void* y, void* conv_out, void* stream) {
  PoolGeom g;
  if (!qkv || !w || !gamma || !beta || !y || !conv_out || pool_geom(g, B, H, T, Hh, Ww, st, sh, sw, ld, col0)) return PVRL_EINVAL;
  const long ntok = (long)B * H * ((long)g.To * g.Ho * g.Wo + 1);
  if (ntok >= (1L << 27)) return PVRL_EINVAL;         // 16 lanes per token, 32-bit token arithmetic in the kernel
  if (g.st == 1) {      // temporal stride 1 (every MViTv2 pooling operator): one 16-lane group per output column, sliding along t
    const long ncol = (long)B * H * ((long)g.Ho * g.Wo + 1);
    if (g.sh > 1 || g.sw > 1)
      hipLaunchKernelGGL(pool_fwd_t_kernel<true>, dim3(grid_for(ncol * 16)), dim3(256), 0, (hipStream_t)stream, (const op_t*)qkv,
                         g, w, gamma, beta, eps, (op_t*)y, (op_t*)conv_out);
    else
      hipLaunchKernelGGL(pool_fwd_t_kernel<false>, dim3(grid_for(ncol * 16)), dim3(256), 0, (hipStream_t)stream, (const op_t*)qkv,
                         g, w, gamma, beta, eps, (op_t*)y, (op_t*)conv_out);
  } else {
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid_for(ntok * 16)), dim3(256), 0, (hipStream_t)stream, (const op_t*)qkv, g,
                       w, gamma, beta, eps, (op_t*)y, (op_t*)conv_out);
  }
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

constexpr int PLN_MAX_WG = 2048;    // workgroups (= partial rows) of pool_ln_bwd_kernel
extern "C" int64_t pvrl_mvit_pool_bwd_workspace_bytes(void) {
  return ((int64_t)PW_MAX_WG * 27 * HD + (int64_t)PLN_MAX_WG * 2 * HD) * sizeof(float);
}

extern "C" int pvrl_mvit_pool_bwd(const void* dy, const void* conv_out, const void* qkv, void* dqkv, int64_t ld,
                                  int64_t col0, int64_t B, int64_t H, int64_t T, int64_t Hh, int64_t Ww, int64_t st,
                                  int64_t sh, int64_t sw, const float* w, const float* gamma, float eps, void* dc_scratch,
                                  float* dw, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  PoolGeom g;
  if (!dy || !conv_out || !qkv || !dqkv || !w || !gamma || !dc_scratch || !dw || !dgamma || !dbeta || !workspace ||
      workspace_bytes < pvrl_mvit_pool_bwd_workspace_bytes() || pool_geom(g, B, H, T, Hh, Ww, st, sh, sw, ld, col0))
    return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long Lo = (long)g.To * g.Ho * g.Wo;
  const long ntok = (long)B * H * (Lo + 1);
  long blocks = (ntok * 16 + 255) / 256;
  if (blocks > PLN_MAX_WG) blocks = PLN_MAX_WG;
  float* lnpart = (float*)workspace + (long)PW_MAX_WG * 27 * HD;
  hipLaunchKernelGGL(pool_ln_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const op_t*)dy, (const op_t*)conv_out,
                     g, gamma, eps, (op_t*)dc_scratch, (op_t*)dqkv, lnpart);
  PVRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(partials_add_kernel, dim3((2 * HD + PADD_OUT - 1) / PADD_OUT), dim3(256), 0, s, (const float*)lnpart, (int)blocks,
                     2 * HD, dgamma, dbeta, HD);
  PVRL_LAUNCH_CHECK();
  const long nin = (long)B * H * T * Hh * Ww;
  if (nin >= (1L << 27)) return PVRL_EINVAL;          // 16 lanes per token, 32-bit token arithmetic in the kernel
  {
    const dim3 dg(grid_for(nin * 16)), db(256);
    const int S = (st == 1 && sh == sw && (sh == 1 || sh == 2 || sh == 4 || sh == 8)) ? (int)sh : 0;
#define DGRAD(SS) hipLaunchKernelGGL(pool_dgrad_kernel<SS>, dg, db, 0, s, (const op_t*)dc_scratch, g, w, (op_t*)dqkv)
#define DGRAD_T(DD, SS) hipLaunchKernelGGL((pool_dgrad_t_kernel<DD, SS>), dim3(grid_for((long)B * H * Hh * Ww * 16)), db, 0, s, \
                                           (const op_t*)dc_scratch, g, w, (op_t*)dqkv)
#ifndef PVRL_POOL_DGRAD_S
#define PVRL_POOL_DGRAD_S 1                                     // 0: run-time strides everywhere (A/B builds)
#endif
    if (!PVRL_POOL_DGRAD_S && st == 1 && sh == 1 && sw == 1) DGRAD_T(true, 0);
    else if (!PVRL_POOL_DGRAD_S && st == 1) DGRAD_T(false, 0);
    else if (st == 1 && sh == 1 && sw == 1) DGRAD_T(true, 1);   // temporal stride 1: one 16-lane group per input column, sliding along t
    else if (st == 1 && S == 2) DGRAD_T(false, 2);
    else if (st == 1 && S == 4) DGRAD_T(false, 4);
    else if (st == 1 && S == 8) DGRAD_T(false, 8);
    else if (st == 1) DGRAD_T(false, 0);
#undef DGRAD_T
    else if (S == 2) DGRAD(2); else if (S == 4) DGRAD(4); else if (S == 8) DGRAD(8); else DGRAD(0);
#undef DGRAD
  }
  PVRL_LAUNCH_CHECK();
  long wb = (B * H * Lo + PW_LANES * 16 - 1) / (PW_LANES * 16);      // >= 16 tokens per lane: 2,592 global atomics per block
  if (wb > PW_MAX_WG) wb = PW_MAX_WG;
  if (wb < 1) wb = 1;
  // temporal stride 1 (every MViTv2 pooling operator): the t-sliding form.  Before its nine neighbour loads were issued together
  // it lost to the 27-loads-at-once form on the big planes and on strided pooling; now it is ahead on every block of
  // MViTv2-S (tools/probe/mvit_pool_times.py: 56 x 56 stride 1 880 -> 578 us, 28 x 28 stride 2 332 -> 252, 14 x 14 stride 2 116 -> 91)
  if (st == 1)
    hipLaunchKernelGGL(pool_wgrad_t_kernel, dim3((unsigned)wb), dim3(PW_CQ * PW_LANES), 0, s, (const op_t*)dc_scratch,
                       (const op_t*)qkv, g, (float*)workspace);
  else
    hipLaunchKernelGGL(pool_wgrad_kernel, dim3((unsigned)wb), dim3(PW_CQ * PW_LANES), 0, s, (const op_t*)dc_scratch,
                       (const op_t*)qkv, g, (float*)workspace);
  PVRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(pool_wgrad_reduce_kernel, dim3(27 * HD / 16), dim3(256), 0, s, (const float*)workspace,
                     (int)wb, dw);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

static int maxpool_geom(MaxPoolGeom& g, int64_t B, int64_t T, int64_t H, int64_t W, int64_t s, int64_t C, int64_t ldi,
                        int64_t ldo) {
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || s < 2 || C <= 0 || (C % 4) || (ldi % 4) || (ldo % 4) || ldi < C || ldo < C)
    return PVRL_EINVAL;
  g.B = (int)B; g.T = (int)T; g.H = (int)H; g.W = (int)W; g.s = (int)s; g.k = (int)s + 1; g.C = (int)C;
  const int pad = g.k / 2;
  g.Ho = (int)((H + 2 * pad - g.k) / s + 1); g.Wo = (int)((W + 2 * pad - g.k) / s + 1);
  g.ldi = (long)ldi; g.ldo = (long)ldo;
  return PVRL_OK;
}

extern "C" int pvrl_mvit_maxpool_fwd(const float* x, int64_t ldi, int64_t B, int64_t T, int64_t H, int64_t W, int64_t s,
                                     int64_t C, float* y, int64_t ldo, void* argmax, void* stream) {
  MaxPoolGeom g;
  if (!x || !y || maxpool_geom(g, B, T, H, W, s, C, ldi, ldo)) return PVRL_EINVAL;
  const long total = ((long)B * T * g.Ho * g.Wo + B) * (C >> 2);
  if (total >= (1L << 31) - (1L << 24)) return PVRL_EINVAL;          // 32-bit index arithmetic in the kernel
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, g, y,
                     (unsigned char*)argmax);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_mvit_maxpool_bwd(const float* x, int64_t ldi, const float* dy, int64_t ldo, int64_t B, int64_t T,
                                     int64_t H, int64_t W, int64_t s, int64_t C, float* dx, const void* argmax, void* stream) {
  MaxPoolGeom g;
  if (!x || !dy || !dx || maxpool_geom(g, B, T, H, W, s, C, ldi, ldo)) return PVRL_EINVAL;
  // (no zero fill: the gather kernel writes every element.  An earlier scatter version zeroed dx with hipMemsetAsync, whose
  //  memset node in a captured HIP graph did not re-zero the buffer on replay -- ROCm 7.2 -- so gradients accumulated
  //  across replays; nothing on a captured path uses hipMemset* any more.)
  const long total = ((long)B * T * H * W + B) * (C >> 2);
  if (total >= (1L << 31) - (1L << 24)) return PVRL_EINVAL;          // 32-bit index arithmetic in the kernel
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, g, dx,
                     (const unsigned char*)argmax);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_copy2d_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t R, int64_t C, float beta,
                               void* stream) {
  if (R <= 0 || C <= 0) return PVRL_OK;
  if (!in || !out || ldi < C || ldo < C) return PVRL_EINVAL;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(R * C)), dim3(256), 0, (hipStream_t)stream, in, (long)ldi, out,
                     (long)ldo, (long)R, (int)C, beta);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
