// Temporal attention of the divided space-time block for T = 8 frames (head_dim 64).
// Reference: Block.forward temporal branch, lib/models/vit.py:129-135, through
// Attention.forward vit.py:75-92 on '(b h w) t m' sequences of 8 tokens.
//
// In the token layout used here (patch tokens ordered (b, n, t), t innermost) a temporal
// sequence is 8 consecutive rows of the packed QKV activation, so no rearrange is needed.
// The 8x8 problem per (sequence, head) is far below an MFMA tile and the kernel is HBM-bound
// (read qkv once, write o once): one 64-lane wave per (sequence, head), lane (i, j) owns
// score s_ij, softmax reductions are 3 cross-lane steps, and P.V / the gradient mat-vecs are
// done with lane (row, 8-column chunk) ownership and __shfl broadcasts.  Backward recomputes P.
#include "common.h"
#include "../../include/pvrl.h"

namespace {

constexpr int PITCH = 144;  // 128 B row + 16 B pad: conflict-free ds_read_b128 across rows
constexpr int TILE = 8 * PITCH;

// 8-element dot product of 16-bit operands as four v_dot2 (two exact products + the fp32 accumulator per instruction) instead of
// eight conversions pairs + eight FMAs: round 3 found both kernels VALU-bound, not HBM-bound (backward: 821 VALU instructions per
// 7 KB item = 8.5 B/clk/CU = the 4.6 TB/s it ran at)
__device__ __forceinline__ float dot8(const opx8 a, const opx8 b, float s = 0.f) {
#if defined(PVRL_T8_NO_DOT2)        // A/B builds only (tools/build_variant.py)
#pragma unroll
  for (int e = 0; e < 8; ++e) s = fmaf((float)a[e], (float)b[e], s);
#else
#pragma unroll
  for (int e = 0; e < 4; ++e) s = FDOT2_F32((opx2){a[2 * e], a[2 * e + 1]}, (opx2){b[2 * e], b[2 * e + 1]}, s, false);
#endif
  return s;
}
__device__ __forceinline__ opx8 lds8(const char* p) { return *reinterpret_cast<const opx8*>(p); }

template <bool BWD>
__global__ __launch_bounds__(256) void attn_t8_kernel(const op_t* __restrict__ qkv, long ld, int ntask, int H, float scale,
                                                      op_t* __restrict__ o, const op_t* __restrict__ d_o, long ldo,
                                                      op_t* __restrict__ dqkv, long ldd) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 4 * TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* sq = smem + wave * 4 * TILE;
  char* sk = sq + TILE;
  char* sv = sk + TILE;
  char* sd = sv + TILE;
  const int task = blockIdx.x * 4 + wave;
  const bool valid = task < ntask;
  const int seq = valid ? task / H : 0, h = valid ? task - seq * H : 0;
  const int HD = H * 64;
  const int r8 = lane >> 3, c8 = lane & 7;
  const long grow = (long)seq * 8 + r8;
  {
    const op_t* src = qkv + grow * ld + h * 64 + c8 * 8;
    const u32x4 a = *reinterpret_cast<const u32x4*>(src);
    const u32x4 b = *reinterpret_cast<const u32x4*>(src + HD);
    const u32x4 c = *reinterpret_cast<const u32x4*>(src + 2 * HD);
    const int off = r8 * PITCH + c8 * 16;
    *reinterpret_cast<u32x4*>(sq + off) = a;
    *reinterpret_cast<u32x4*>(sk + off) = b;
    *reinterpret_cast<u32x4*>(sv + off) = c;
    if constexpr (BWD) {
      const u32x4 d = *reinterpret_cast<const u32x4*>(d_o + grow * ldo + h * 64 + c8 * 8);
      *reinterpret_cast<u32x4*>(sd + off) = d;
    }
  }
  __syncthreads();
  // scores: lane (i = r8, j = c8)
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) s = dot8(lds8(sq + r8 * PITCH + c * 16), lds8(sk + c8 * PITCH + c * 16), s);
  s *= scale;
  float mx = s;
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
  float pe = __expf(s - mx);
  float sum = pe;
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  sum += __shfl_xor(sum, 4, 64);
  const float pr = pe / sum;  // P[i = r8][j = c8]

  if constexpr (!BWD) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pj = __shfl(pr, (lane & ~7) | j, 64);
      const opx8 v = lds8(sv + j * PITCH + c8 * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)v[e], acc[e]);
    }
    if (valid) {
      opx8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (op_t)acc[e];
      *reinterpret_cast<opx8*>(o + grow * ldo + h * 64 + c8 * 8) = ov;
    }
  } else {
    float dp = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) dp = dot8(lds8(sd + r8 * PITCH + c * 16), lds8(sv + c8 * PITCH + c * 16), dp);
    float dd = pr * dp;
    dd += __shfl_xor(dd, 1, 64);
    dd += __shfl_xor(dd, 2, 64);
    dd += __shfl_xor(dd, 4, 64);
    const float ds = pr * (dp - dd) * scale;  // dS[i = r8][j = c8] (already times scale)
    float aq[8], ak[8], av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { aq[e] = 0.f; ak[e] = 0.f; av[e] = 0.f; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // dq[i = r8][chunk c8] += dS[r8][t] * k[t]
      const float ds_it = __shfl(ds, (lane & ~7) | t, 64);
      const opx8 kv = lds8(sk + t * PITCH + c8 * 16);
      // dk[j = r8][chunk c8] += dS[t][r8] * q[t] ; dv[j = r8] += P[t][r8] * dO[t]
      const float ds_tj = __shfl(ds, t * 8 + r8, 64);
      const float p_tj = __shfl(pr, t * 8 + r8, 64);
      const opx8 qv = lds8(sq + t * PITCH + c8 * 16);
      const opx8 dv = lds8(sd + t * PITCH + c8 * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        aq[e] = fmaf(ds_it, (float)kv[e], aq[e]);
        ak[e] = fmaf(ds_tj, (float)qv[e], ak[e]);
        av[e] = fmaf(p_tj, (float)dv[e], av[e]);
      }
    }
    if (valid) {
      opx8 oq, ok, ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) { oq[e] = (op_t)aq[e]; ok[e] = (op_t)ak[e]; ov[e] = (op_t)av[e]; }
      op_t* dst = dqkv + grow * ldd + h * 64 + c8 * 8;
      *reinterpret_cast<opx8*>(dst) = oq;
      *reinterpret_cast<opx8*>(dst + HD) = ok;
      *reinterpret_cast<opx8*>(dst + 2 * HD) = ov;
    }
  }
}

}  // namespace

extern "C" int pvrl_attn_t8_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t H, float scale, void* o, int64_t ldo,
                                void* stream) {
  if (nseq <= 0) return PVRL_OK;
  if (!qkv || !o || H <= 0 || (ld % 8) || (ldo % 8)) return PVRL_EINVAL;
  const int ntask = (int)(nseq * H);
  hipLaunchKernelGGL((attn_t8_kernel<false>), dim3((unsigned)cdiv(ntask, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const op_t*)qkv, (long)ld, ntask, (int)H, scale, (op_t*)o, (const op_t*)nullptr, (long)ldo,
                     (op_t*)nullptr, 0L);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_attn_t8_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t H, float scale, const void* d_o,
                                int64_t ldo, void* dqkv, int64_t ldd, void* stream) {
  if (nseq <= 0) return PVRL_OK;
  if (!qkv || !d_o || !dqkv || H <= 0 || (ld % 8) || (ldo % 8) || (ldd % 8)) return PVRL_EINVAL;
  const int ntask = (int)(nseq * H);
  hipLaunchKernelGGL((attn_t8_kernel<true>), dim3((unsigned)cdiv(ntask, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const op_t*)qkv, (long)ld, ntask, (int)H, scale, (op_t*)nullptr, (const op_t*)d_o, (long)ldo,
                     (op_t*)dqkv, (long)ldd);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
