// Attention backward for SHORT contiguous sequences (16 < S <= 32 tokens, head_dim 64, no masks): one WAVE per (sequence, head),
// dQ, dK and dV from one evaluation of P and dS, no workgroup synchronisation at all.
//
// Reference semantics: autograd of Attention.forward, lib/models/vit.py:75-92, on the TEMPORAL sequences of Block.forward
// (vit.py:129-135) at 32 frames (BASELINE configs[3]: 8 clips x 196 patches x 12 heads = 18,816 sequences of 32 tokens).  The
// two-pass kernels (attn_bwd_q / attn_bwd_kv<2, ...>) read q, k, v, dO twice and evaluate P twice: 101 + 109 us per block
// (profiles/r4_t32_pmc_hbm_mfma.csv) against ~110 us of HBM time for one pass.
//
// A 32-token head is ONE 32x32 MFMA tile: the wave loads its five 4 KB operands (Q, K, V, dO, O rows of the head, 16 bytes per
// lane and column step, already in v_mfma_f32_32x32x16 operand layout), then
//   S, dP      a = Q / dO, b = K / V: 4 + 4 MFMAs; the accumulators START at -lse/scale and -D*scale (D = rowsum(dO * O) from the
//              operand registers), so P = exp2(c * acc), dS = P * acc
//   dV^T, dK^T reduction over the queries = the register dimension of P / dS: they feed the b operand as they are; a = dO^T / Q^T
//              through ds_read_b64_tr_b16 from this wave's own LDS images (written from the operand registers)
//   dQ^T       reduction over the keys = the lane dimension: dS^T goes through a 2 KB wave-private LDS image (attn_stream.h layout),
//              a = K^T fragments
//   outputs    transposed through the LDS images that are no longer needed, stored as whole 128-byte rows.
// 14 KB of LDS and 126 registers per wave: two waves per workgroup, five workgroups per CU, 20 KB of loads in flight per wave.
// `scale` must be a power of two (folded into the V operand and into D, exactly).
#include "attn_stream.h"
#include "../../include/pvrl.h"
#include <stdlib.h>

namespace {

constexpr int S32_WAVE_LDS = 3 * 4096 + 2048;      // Q, dO, K images (32 rows x 128 B) + the dS^T image / D values
#ifndef PVRL_S32_NW
#define PVRL_S32_NW 2                              // waves (= items) per workgroup (the waves never synchronise): 28 KB of LDS, five workgroups per CU; 1 / 3 / 4 measured 150 / 146 / 152 us against 144
#endif
constexpr int S32_NW = PVRL_S32_NW;

__global__ __launch_bounds__(64 * S32_NW) void attn_bwd_s32_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[S32_NW * S32_WAVE_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long item = (long)blockIdx.x * S32_NW + wave;
  if (item >= (long)p.nseq * p.H) return;            // (wave-uniform; nothing below synchronises across waves)
  const int seq = (int)(item / p.H), h = (int)(item - (long)seq * p.H);
  const int S = p.mp.S, HD = p.H * 64;
  const int n = lane & 31, g = lane >> 5;
  const float c = p.scale * 1.4426950408889634f;
  char* Qi = smem + wave * S32_WAVE_LDS;
  char* Di = Qi + 4096;
  char* Ki = Qi + 8192;
  char* Si = Qi + 12288;

  // ---- operands: row min(n, S - 1) of the sequence, columns 16 s + 8 g .. + 7 of the head (rows past the sequence hold copies)
  const long row = (long)seq * S + min(n, S - 1);
  opx8 qf[4], kf[4], vf[4], df[4], of[4];
  {
    const op_t* qp = p.qkv + row * p.ld + h * 64 + 8 * g;
    const op_t* dp = p.d_o + row * p.ldo + h * 64 + 8 * g;
    const op_t* op = p.ofw + row * p.ldo + h * 64 + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const opx8*>(qp + 16 * s);
      kf[s] = *reinterpret_cast<const opx8*>(qp + HD + 16 * s);
      vf[s] = *reinterpret_cast<const opx8*>(qp + 2 * HD + 16 * s);
      df[s] = *reinterpret_cast<const opx8*>(dp + 16 * s);
      of[s] = *reinterpret_cast<const opx8*>(op + 16 * s);
    }
  }
  // start values of the score accumulator: -lse / scale of the 16 query rows this lane holds (rows 4 g + 8 j + r); -inf past the sequence
  f32x16 sacc, dacc;
  {
    const float* lp = p.lse + item * S;
    const float rs = 1.0f / p.scale;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = 4 * g + 8 * (r >> 2) + (r & 3);
      sacc[r] = qrow < S ? -lp[qrow] * rs : -INFINITY;
    }
  }
  // D = rowsum(dO * O): this lane has half of its row's columns, its partner (lane ^ 32) the other half
  float dsum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum = fmaf((float)df[s][e], (float)of[s][e], dsum);
  dsum += __shfl_xor(dsum, 32, 64);
  // images for the transposed fragments, straight from the operand registers; D through LDS into the accumulator layout
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int o = fb_off(n, 16 * s + 8 * g);
    *reinterpret_cast<opx8*>(Qi + o) = qf[s];
    *reinterpret_cast<opx8*>(Di + o) = df[s];
    *reinterpret_cast<opx8*>(Ki + o) = kf[s];
  }
  if (g == 0) reinterpret_cast<float*>(Si)[n] = n < S ? -dsum * p.scale : 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) vf[s][e] = (op_t)((float)vf[s][e] * p.scale);      // exact: scale is a power of two
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Si) + 8 * j + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) dacc[4 * j + r] = b[r];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) sacc = MFMA_32x32x16(qf[s], kf[s], sacc, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) dacc = MFMA_32x32x16(df[s], vf[s], dacc, 0, 0, 0);

  // ---- P, dS (lane = key n, registers = queries); keys past the sequence contribute nothing
  const float keep = n < S ? 1.f : 0.f;
  opx8 pf[2], sf[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pr = __builtin_amdgcn_exp2f(c * sacc[r]) * keep;
    pf[r >> 3][r & 7] = (op_t)pr;
    sf[r >> 3][r & 7] = (op_t)(pr * dacc[r]);
  }
  // ---- dV^T += dO^T P, dK^T += Q^T dS (k-step t = queries 16 t + {4 g + r, 8 + 4 g + r}: the lane's own registers)
  const int i16 = lane & 15, hi = (lane >> 4) & 1;
  const int trb = g * 512 + (hi ^ g) * 128 + (i16 >> 2) * 32;
  const int tr0 = trb + 8 * (i16 & 3), tr1 = trb + 1024 + ((8 * (i16 & 3)) ^ 16);
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dh][r] = 0.f; dv[dh][r] = 0.f; }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const opx8 ad = tr_frag8(Di + t * 2048 + dh * 256, tr0, tr1);
      dv[dh] = MFMA_32x32x16(ad, pf[t], dv[dh], 0, 0, 0);
      const opx8 aq = tr_frag8(Qi + t * 2048 + dh * 256, tr0, tr1);
      dk[dh] = MFMA_32x32x16(aq, sf[t], dk[dh], 0, 0, 0);
    }
  // ---- dS^T -> [key][query] image (4 consecutive queries = 8 bytes per store; the D values that lived here have been read)
  {
    const int kb = n >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int slot = 2 * (j & 1) + g;
      opx4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = sf[j >> 1][4 * (j & 1) + r];
      *reinterpret_cast<opx4*>(Si + kb * 256 + (j >> 1) * 128 + (n & 3) * 32 + ((slot ^ (kb & 3)) * 8)) = w;
    }
  }
  // ---- dQ^T = K^T dS^T over the 32 keys: two k-steps of 16 keys
  f32x16 dq[2];
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dh][r] = 0.f;
  {
    const int inb = (i16 >> 2) * 32 + ((8 * (i16 & 3)) ^ (16 * g)) + g * 1024;
    const int a0 = inb + hi * 128, a1 = inb + 512 + (hi ^ 1) * 128;
    const int b0 = (2 * g) * 256 + hi * 128 + (i16 >> 2) * 32 + (((i16 & 3) ^ (2 * g)) * 8);
    const int b1 = (2 * g + 1) * 256 + hi * 128 + (i16 >> 2) * 32 + (((i16 & 3) ^ (2 * g + 1)) * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const opx8 bf = tr_frag8(Si + ks * 1024, b0, b1);
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {
        const opx8 af = tr_frag8(Ki + ks * 2048 + dh * 256, a0, a1);
        dq[dh] = MFMA_32x32x16(af, bf, dq[dh], 0, 0, 0);
      }
    }
  }
  // ---- outputs: every accumulator holds 4 consecutive columns of one token per register quad; transposed through an image that
  // is no longer read ([token][64 columns], 16-byte chunks swizzled by the token) and stored as whole rows
  auto emit = [&](const f32x16 (&acc)[2], char* st, long col0) {
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        opx4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (op_t)acc[dh][4 * j + r];
        *reinterpret_cast<opx4*>(st + n * 128 + (((4 * dh + j) ^ (n & 7)) * 16) + 8 * g) = ov;
      }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = lane + 64 * t;
      const int r = idx >> 3, ch = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(st + r * 128 + ((ch ^ (r & 7)) * 16));
      if (r < S) *reinterpret_cast<u32x4*>(p.dqkv + ((long)seq * S + r) * p.ldd + col0 + h * 64 + ch * 8) = v;
    }
  };
  emit(dv, Di, 2 * HD);
  emit(dk, Qi, HD);
  emit(dq, Ki, 0);
}

// PVRL_ATTN_BWD_S32=0 sends the short sequences back to the two-pass kernels (A/B runs); read once
int attn_bwd_s32_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_ATTN_BWD_S32");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

}  // namespace

bool pvrl_attn_bwd_s32_ok(const AttnArgs& p) {
  if (!attn_bwd_s32_enabled() || p.causal || p.kpm || p.mp.mode != 0) return false;
  if (p.mp.S <= 16 || p.mp.S > 32) return false;
  if ((p.ldd % 8) || (p.ldo % 8)) return false;                        // 16-byte row accesses
  int e = 0;
  return frexpf(p.scale, &e) == 0.5f;      // power of two
}

int pvrl_attn_bwd_s32_launch(const AttnArgs& p, hipStream_t s) {
  const long items = (long)p.nseq * p.H;
  hipLaunchKernelGGL(attn_bwd_s32_kernel, dim3((unsigned)((items + S32_NW - 1) / S32_NW)), dim3(64 * S32_NW), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
