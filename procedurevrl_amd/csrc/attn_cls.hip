// Spatial attention of the encoder's LAST block: the cls query only.
//
// Only `x[:, 0]` of the final norm is read (lib/models/vit.py:418-421), so of the last block's spatial attention
// (vit.py:137-151, Attention.forward vit.py:75-92) only the cls token's output is needed: one query per (clip, frame) sequence and
// head against the sequence's 197 keys and values.  Backward likewise: dO is zero for every patch query, so dS has ONE non-zero row
// per (sequence, head) -- dV = p^T dO and dK = dS^T q are rank-1, dQ is non-zero for the cls query alone.
// (The general kernels, attn_mfma.hip / attn_bwd_fused.hip, spend 82 + 200 us on 197 queries per item; these two move the item's
//  keys and values once: HBM-bound.)
//
// One wave per (sequence, head), lanes = 8 keys x 8 sixteen-byte chunks: every load instruction moves eight whole 128-byte rows.
// fp32 throughout, one rounding to the operand type at the stores.  Sequence addressing: attn_common.h mode 1.
// (A first form with lanes = keys -- each lane reading whole rows in eight 16-byte pieces and holding q / dO / a K row in 192
//  registers -- ran 43 / 203 us: 64 lines per load instruction and two waves per SIMD.)
#include "attn_common.h"
#include "../../include/pvrl.h"

namespace {

constexpr int AC_WAVES = 4;               // items per workgroup

struct AttnCls {
  const op_t* qkv; long ld;
  int H, nseq;
  SeqMap mp;
  float scale;
  op_t* o_cls; long ldo;                  // forward: [nseq][H * 64]
  float* lse;                             // [nseq][H][S]: entry 0 of each (sequence, head) is written / read
  const op_t* d_o_cls; const op_t* ofw_cls;
  op_t* dqkv; op_t* dqkv_cls; long ldd;
  int zero_dq;                            // backward: write the (zero) dQ of the patch tokens
};

// lane = (key group kg = lane >> 3, chunk c = lane & 7): one load instruction fetches eight whole 128-byte K (or V) rows, lane (kg, c)
// holding columns 8 c .. 8 c + 7 of key 8 it + kg; a dot product over the head's 64 columns is eight FMAs per lane and three
// xor-shuffles inside the 8-lane group.
__device__ __forceinline__ float dot8(const float (&a)[8], const opx8 b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s = fmaf(a[e], (float)b[e], s);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  return s;
}
__device__ __forceinline__ void load8(const op_t* p, float (&v)[8]) {
  const opx8 x = *reinterpret_cast<const opx8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}

// forward: every key group runs an online softmax over its keys (running maximum m, sum l, un-normalised output chunk acc); the eight
// groups' states are merged at the end
__global__ __launch_bounds__(64 * AC_WAVES) void attn_cls_fwd_kernel(AttnCls p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = min((int)blockIdx.x * AC_WAVES + wave, p.nseq * p.H - 1);     // (a clamped wave repeats the last item: same stores)
  const int seq = item / p.H, h = item - seq * p.H;
  const int S = p.mp.S, HD = p.H * 64;
  const int c = lane & 7, kg = lane >> 3;
  const SeqRows sr = seq_rows(p.mp, seq);
  float q8[8];
  load8(p.qkv + row_of(sr, 0) * p.ld + h * 64 + 8 * c, q8);
  const float cl = p.scale * 1.4426950408889634f;
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int iters = (S + 7) >> 3;
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    const int key = it * 8 + kg;
    const op_t* row = p.qkv + row_of(sr, min(key, S - 1)) * p.ld + h * 64 + 8 * c;
    const opx8 k8 = *reinterpret_cast<const opx8*>(row + HD);
    const opx8 v8 = *reinterpret_cast<const opx8*>(row + 2 * HD);
    const float dq = dot8(q8, k8);                              // (the shuffles run in every lane: uniform control flow)
    const float sc = key < S ? dq : -INFINITY;
    const float mn = fmaxf(m, sc);
    const float f = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * cl);      // exp2(-inf) = 0 on the first key
    const float e0 = (sc == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((sc - mn) * cl);
    l = l * f + e0;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(acc[e], f, e0 * (float)v8[e]);
    m = mn;
  }
  // merge the key groups (lanes 8 apart hold the same chunk): butterfly over lane bits 3, 4, 5
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float f1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * cl);
    const float f2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m2 - mn) * cl);
    l = l * f1 + l2 * f2;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * f1 + __shfl_xor(acc[e], o, 64) * f2;
    m = mn;
  }
  if (kg == 0) {
    const float inv = 1.0f / l;
    opx8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (op_t)(acc[e] * inv);
    *reinterpret_cast<opx8*>(p.o_cls + (long)seq * p.ldo + h * 64 + 8 * c) = o8;
    if (c == 0) p.lse[((long)seq * p.H + h) * S] = m * p.scale + __logf(l);
  }
}

__global__ __launch_bounds__(64 * AC_WAVES) void attn_cls_bwd_kernel(AttnCls p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = min((int)blockIdx.x * AC_WAVES + wave, p.nseq * p.H - 1);
  const int seq = item / p.H, h = item - seq * p.H;
  const int S = p.mp.S, HD = p.H * 64;
  const int c = lane & 7, kg = lane >> 3;
  const SeqRows sr = seq_rows(p.mp, seq);
  float q8[8], do8[8];
  load8(p.qkv + row_of(sr, 0) * p.ld + h * 64 + 8 * c, q8);
  load8(p.d_o_cls + (long)seq * p.ldo + h * 64 + 8 * c, do8);
  const float D = dot8(do8, *reinterpret_cast<const opx8*>(p.ofw_cls + (long)seq * p.ldo + h * 64 + 8 * c));     // rowsum(dO * O)
  const float lse = p.lse[((long)seq * p.H + h) * S];
  float accq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) accq[e] = 0.f;
  const int iters = (S + 7) >> 3;
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    const int key = it * 8 + kg;
    const bool valid = key < S;
    const int kc = min(key, S - 1);
    const op_t* row = p.qkv + row_of(sr, kc) * p.ld + h * 64 + 8 * c;
    const opx8 k8 = *reinterpret_cast<const opx8*>(row + HD);
    const opx8 v8 = *reinterpret_cast<const opx8*>(row + 2 * HD);
    const float a = dot8(q8, k8), dp = dot8(do8, v8);
    const float pk = valid ? __expf(a * p.scale - lse) : 0.f;
    const float ds = pk * (dp - D) * p.scale;
    if (valid) {
      op_t* out = tok_ptr(p.dqkv, p.dqkv_cls, p.ldd, p.mp, sr, seq, kc) + h * 64 + 8 * c;
      opx8 dk, dv;
#pragma unroll
      for (int e = 0; e < 8; ++e) { dk[e] = (op_t)(ds * q8[e]); dv[e] = (op_t)(pk * do8[e]); }
      *reinterpret_cast<opx8*>(out + HD) = dk;
      *reinterpret_cast<opx8*>(out + 2 * HD) = dv;
      if (kc > 0 && p.zero_dq) *reinterpret_cast<opx8*>(out) = (opx8)(op_t)0.f;       // no gradient reaches a patch query
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) accq[e] = fmaf(ds, (float)k8[e], accq[e]);
  }
  // dQ of the cls query: the key groups' partial sums (lanes 8 apart hold the same chunk)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) accq[e] += __shfl_xor(accq[e], o, 64);
  if (kg == 0) {
    opx8 dq;
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = (op_t)accq[e];
    *reinterpret_cast<opx8*>(p.dqkv_cls + (long)seq * p.ldd + h * 64 + 8 * c) = dq;
  }
}

int check(const AttnCls& p) {
  if (!p.qkv || !p.lse || p.H <= 0 || p.nseq < 0 || p.mp.S <= 1 || p.mp.S > 4096) return PVRL_EINVAL;
  if (p.mp.T <= 0 || (p.nseq % p.mp.T) || (p.ld % 8) || ((uintptr_t)p.qkv & 15)) return PVRL_EINVAL;
  return PVRL_OK;
}

}  // namespace

extern "C" int pvrl_attn_cls_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int64_t T, int64_t cls_base,
                                 float scale, void* o_cls, int64_t ldo, float* lse, void* stream) {
  AttnCls p = {};
  p.qkv = (const op_t*)qkv; p.ld = ld; p.H = (int)H; p.nseq = (int)nseq;
  p.mp.mode = 1; p.mp.S = (int)S; p.mp.T = (int)T; p.mp.cls_base = cls_base;
  p.scale = scale; p.o_cls = (op_t*)o_cls; p.ldo = ldo; p.lse = lse;
  if (nseq == 0) return PVRL_OK;
  if (int e = check(p)) return e;
  if (!o_cls || (ldo % 8) || ((uintptr_t)o_cls & 15)) return PVRL_EINVAL;
  hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((unsigned)cdiv(nseq * H, AC_WAVES)), dim3(64 * AC_WAVES), 0, (hipStream_t)stream, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_attn_cls_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int64_t T, int64_t cls_base,
                                 float scale, const void* o_cls, const void* d_o_cls, int64_t ldo, const float* lse, void* dqkv,
                                 void* dqkv_cls, int64_t ldd, int zero_patch_dq, void* stream) {
  AttnCls p = {};
  p.qkv = (const op_t*)qkv; p.ld = ld; p.H = (int)H; p.nseq = (int)nseq;
  p.mp.mode = 1; p.mp.S = (int)S; p.mp.T = (int)T; p.mp.cls_base = cls_base;
  p.scale = scale; p.ofw_cls = (const op_t*)o_cls; p.d_o_cls = (const op_t*)d_o_cls; p.ldo = ldo; p.lse = const_cast<float*>(lse);
  p.dqkv = (op_t*)dqkv; p.dqkv_cls = (op_t*)dqkv_cls; p.ldd = ldd; p.zero_dq = zero_patch_dq;
  if (nseq == 0) return PVRL_OK;
  if (int e = check(p)) return e;
  if (!o_cls || !d_o_cls || !dqkv || !dqkv_cls || (ldd % 8) || ((uintptr_t)dqkv & 15) || ((uintptr_t)dqkv_cls & 15)) return PVRL_EINVAL;
  if ((ldo % 8) || ((uintptr_t)o_cls & 15) || ((uintptr_t)d_o_cls & 15)) return PVRL_EINVAL;
  hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((unsigned)cdiv(nseq * H, AC_WAVES)), dim3(64 * AC_WAVES), 0, (hipStream_t)stream, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
