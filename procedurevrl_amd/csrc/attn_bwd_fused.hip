// Spatial-attention backward as ONE kernel per (sequence, head): dQ, dK and dV from one evaluation of P and dS.
//
// Reference semantics: autograd of Attention.forward, lib/models/vit.py:75-92, on the spatial sequences of
// Block.forward (vit.py:137-151; 197 tokens per (clip, frame)).  The two-pass form (attn_bwd_q_kernel + attn_bwd_kv_kernel,
// attn_mfma.hip) evaluates S = Q K^T, dP = dO V^T, the exponentials and the dS arithmetic twice and reads q, k, v, dO twice;
// here a workgroup loads the Q, dO and K head slices of its sequence once (3 x 28 KB of LDS), keeps the K / V rows of its waves'
// own keys in registers, and walks the queries in blocks of 32:
//
//   key waves (wave w owns keys 32w .. 32w+31; 7 waves at S = 197), per query block
//     S^T, dP^T   32 queries x 32 keys each, 4 + 4 v_mfma_f32_32x32x16 (a = Q / dO rows from LDS, b = K / V from registers);
//                 the accumulators START at -lse/scale and -D*scale, so P = exp2(c * acc) and dS = P * acc need no subtraction
//     dV^T += dO^T P, dK^T += Q^T dS    (reduction over the block's queries = the MFMA's k: P / dS feed the b operand straight
//                 from registers, a = transposed fragments of the same LDS images through ds_read_b64_tr_b16)
//     dS -> LDS   bf16, [key][query] image of the block (14 KB, two buffers)
//   one barrier per block, then
//   dQ^T block  = K^T dS^T over ALL keys (the reduction crosses the key waves, hence the LDS exchange): eight 16 x 16 output tiles,
//                 7 k-steps of v_mfma_f32_16x16x32 each; one tile on each of waves 0-3, four on the wave that owns no keys,
//                 which balances the matrix pipe of the four SIMDs (2 x 512 + 112 cycles vs 512 + 112 + 448 per block).
//
// D = rowsum(dO * O) is computed in the prologue from the dO registers on their way to LDS (the O head slice is read once and
// never stored).  `scale` must be a power of two (it is 1/8 for head_dim 64): it is folded into the V operand and into D, both
// exactly.  Rows past the sequence are zero in LDS and their -lse start is -inf, so P = 0 there without a compare; keys past the
// sequence meet zero K rows in the dQ product and are never stored.  114 KB of LDS: one 8-wave workgroup per CU.
#include "attn_common.h"
#include "../../include/pvrl.h"

namespace {

#ifndef PVRL_FB_ABLATE
#define PVRL_FB_ABLATE 0      // probe builds only (tools/build_variant.py): 1 = no block loop (loads + final stores), 2 = no global loads
#endif
#ifndef PVRL_FB_TRACE
#define PVRL_FB_TRACE 0       // probe builds only: every wave of workgroup 8 stamps the cycle counter at the seams of each block into spare
#endif                        // LDS; dumped through AttnArgs::dvec (tools/probe/attn_bwd_ab.py trace)
constexpr int FB_ROWS = 224;                 // 7 blocks of 32
constexpr int FB_TILE = FB_ROWS * 128;       // one [224][64] head slice
constexpr int FB_DS = FB_ROWS * 64;          // dS^T image of one query block: [224 keys][32 queries]
constexpr int FB_LDS = 3 * FB_TILE + 2 * FB_DS + 2 * FB_ROWS * 4;
// Block barrier: LDS traffic only.  __syncthreads() is a workgroup-scope release: it puts s_waitcnt vmcnt(0) in front of
// s_barrier, i.e. the dQ wave would wait for its global STORES of the previous block to complete (~1,500 cycles, traced).
#define FB_BARRIER()                                     \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#if PVRL_FB_TRACE
#define FB_STAMP(jb, k)                                                                                                   \
  do {                                                                                                                    \
    if (tracing) reinterpret_cast<unsigned long long*>(smem + FB_LDS)[(wave * 8 + (jb)) * 8 + (k)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define FB_STAMP(jb, k) do { } while (0)
#endif

// [4 rows][16 cols] 128-byte blocks (the unit ds_read_b64_tr_b16 transposes), four blocks per 4-row band.  Two swizzles make the
// 32-row a-operand walk of v_mfma_32x32x16 (lane = row, 16 bytes at a fixed column chunk) conflict-free: the block order within a
// band flips with bit 0 of the band index, the two 16-byte halves of a row's 32 bytes flip with bit 1.
__device__ __forceinline__ int fb_off(int row, int col) {
  const int rb = row >> 2;
  return (rb * 4 + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (((col & 15) * 2) ^ ((rb & 2) << 3));
}

// dQ^T of one 32-query block = K^T dS^T over all keys: 2 query tiles x 4 column tiles of 16 x 16, NQB (or `nqb`) k-steps of 32 keys.
// Runs on the one wave that owns no keys, one block behind the key waves.
template <int NQB>
__device__ __forceinline__ void dq_block(const AttnArgs& p, const SeqRows& sr, const char* Kb, const char* dsr, int nqb, int S,
                                         int seq, int h, int jb, int lane) {
  const int q4 = lane >> 4, i = lane & 15;
  f32x4 acc[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // lane parts of the transposed-fragment addresses: keys 32u + 8 q4 + 4 e2 + {0..3}
  const int in_a = (i >> 2) * 32 + ((8 * (i & 3)) ^ (16 * (q4 & 1))) + q4 * 1024;
  const int ks = 2 * (q4 & 1);
  const int in_b0 = (2 * q4) * 256 + (i >> 2) * 32 + (((i & 3) ^ ks) * 8);
  const int in_b1 = (2 * q4 + 1) * 256 + (i >> 2) * 32 + (((i & 3) ^ (ks + 1)) * 8);
  auto step = [&](int u) {
    opx8 bf[2], af[4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) bf[qt] = tr_frag8(dsr + u * 2048 + qt * 128, in_b0, in_b1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) af[dt] = tr_frag8(Kb + u * 4096, in_a + dt * 128, in_a + 512 + (dt ^ 1) * 128);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = MFMA_16x16x32(af[dt], bf[qt], acc[qt][dt], 0, 0, 0);
  };
  if constexpr (NQB > 0) {
#pragma unroll
    for (int u = 0; u < NQB; ++u) step(u);
  } else {
#pragma unroll 1
    for (int u = 0; u < nqb; ++u) step(u);
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * jb + 16 * qt + i;
    if (query < S) {
      op_t* op = tok_ptr(p.dqkv, p.dqkv_cls, p.ldd, p.mp, sr, seq, query) + h * 64 + 4 * q4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        opx4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (op_t)acc[qt][dt][r];
        *reinterpret_cast<opx4*>(op + 16 * dt) = ov;
      }
    }
  }
}

template <int NQB>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[FB_LDS + (PVRL_FB_TRACE ? 8 * 8 * 8 * 8 : 0)];
#if PVRL_FB_TRACE
  const bool tracing = blockIdx.x == 8;
#endif
  char* Qb = smem;
  char* Db = smem + FB_TILE;
  char* Kb = smem + 2 * FB_TILE;
  char* dsb = smem + 3 * FB_TILE;
  float* sinit = reinterpret_cast<float*>(smem + 3 * FB_TILE + 2 * FB_DS);
  float* dinit = sinit + FB_ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the H heads of a sequence run back to back on ONE XCD (same order as the forward kernel)
  const int xj = blockIdx.x >> 3;
  const int seq = (xj / p.H) * 8 + (blockIdx.x & 7), h = xj % p.H;
  if (seq >= p.nseq) return;
  const int S = p.mp.S;
  const int nqb = (S + 31) >> 5;
  const int HD = p.H * 64;
  const SeqRows sr = seq_rows(p.mp, seq);
  const int n = lane & 31, g = lane >> 5;
  const bool keywave = wave < nqb;

  // ---- prologue: every global load of the workgroup is in flight before the first LDS store
  FB_STAMP(7, 0);
  opx8 kf[4], vf[4];
  {
    u32x4 qv[4], kv[4], dv[4], ov[4];
    float lsev[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 512 * it;
      const int row = idx >> 3, c = idx & 7;
      const int rc = min(row, S - 1);
      const op_t* qp = p.qkv + row_of(sr, rc) * p.ld + h * 64 + c * 8;
#if PVRL_FB_ABLATE & 2
      qv[it] = kv[it] = dv[it] = ov[it] = (u32x4){(unsigned)idx, 0u, 0u, 0u}; lsev[it] = 1.f;
      continue;
#endif
      qv[it] = *reinterpret_cast<const u32x4*>(qp);
      kv[it] = *reinterpret_cast<const u32x4*>(qp + HD);
      dv[it] = *reinterpret_cast<const u32x4*>(tok_ptr(p.d_o, p.d_o_cls, p.ldo, p.mp, sr, seq, rc) + h * 64 + c * 8);
      ov[it] = *reinterpret_cast<const u32x4*>(tok_ptr(p.ofw, p.ofw_cls, p.ldo, p.mp, sr, seq, rc) + h * 64 + c * 8);
      lsev[it] = p.lse[((long)seq * p.H + h) * S + rc];
    }
    {
      const int key = min(32 * wave + n, S - 1);
      const op_t* kp = p.qkv + row_of(sr, key) * p.ld + HD + h * 64 + 8 * g;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const opx8*>(kp + 16 * s);
        vf[s] = *reinterpret_cast<const opx8*>(kp + HD + 16 * s);
      }
    }
    const float rscale = 1.0f / p.scale;
    FB_STAMP(7, 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 512 * it;
      const int row = idx >> 3, c = idx & 7;
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a0, a1, b0, b1;
        op_unpack2(dv[it][e], a0, a1);
        op_unpack2(ov[it][e], b0, b1);
        dsum = fmaf(a0, b0, dsum);
        dsum = fmaf(a1, b1, dsum);
      }
      dsum += __shfl_xor(dsum, 1, 64);
      dsum += __shfl_xor(dsum, 2, 64);
      dsum += __shfl_xor(dsum, 4, 64);
      if (row < FB_ROWS) {
        const bool keep = row < S;
        const unsigned m = keep ? 0xffffffffu : 0u;
        const u32x4 mk = (u32x4){m, m, m, m};
        const int off = fb_off(row, c * 8);
        *reinterpret_cast<u32x4*>(Qb + off) = qv[it] & mk;
        *reinterpret_cast<u32x4*>(Db + off) = dv[it] & mk;
        *reinterpret_cast<u32x4*>(Kb + off) = kv[it] & mk;
        if (c == 0) {
          sinit[row] = keep ? -lsev[it] * rscale : -INFINITY;
          dinit[row] = keep ? -dsum * p.scale : 0.f;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) vf[s][e] = (op_t)((float)vf[s][e] * p.scale);     // exact: scale is a power of two
  }
  FB_STAMP(7, 2);
  __syncthreads();
  FB_STAMP(7, 3);

  const float c = p.scale * 1.4426950408889634f;
  constexpr bool NOLOOP = (PVRL_FB_ABLATE & 1) != 0;
  const int nblk = NOLOOP ? 0 : (NQB > 0 ? NQB : nqb);
  if (wave == 7) {
    // ---- the dQ wave: block jb's product while the key waves are already in block jb + 1
#pragma unroll 1
    for (int jb = 0; jb < nblk; ++jb) {
      FB_STAMP(jb, 0);
      FB_BARRIER();
      FB_STAMP(jb, 1);
      dq_block<NQB>(p, sr, Kb, dsb + (jb & 1) * FB_DS, nqb, S, seq, h, jb, lane);
      FB_STAMP(jb, 2);
    }
#if PVRL_FB_TRACE
    FB_STAMP(7, 4);
    if (tracing && p.dvec)
      for (int i = lane; i < 64; i += 64) reinterpret_cast<unsigned long long*>(p.dvec)[wave * 64 + i] = reinterpret_cast<unsigned long long*>(smem + FB_LDS)[wave * 64 + i];
#endif
    return;
  }
  // ---- key waves
  f32x16 dk[2], dvv[2];
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dh][r] = 0.f; dvv[dh][r] = 0.f; }

  // lane parts of the LDS addresses inside one 32-row query block (4096 bytes of a tile)
  const int rbl = n >> 2, b0 = rbl & 1, b1 = (rbl >> 1) & 1;
  const int rowbase = rbl * 512 + (n & 3) * 32 + ((16 * g) ^ (16 * b1));
  const int e0 = rowbase + b0 * 128, e1 = rowbase + (1 - b0) * 128;      // column step s even / odd (+256 for s >= 2)
  const int i16 = lane & 15, hi = (lane >> 4) & 1;
  const int trb = g * 512 + (hi ^ g) * 128 + (i16 >> 2) * 32;
  const int tr0 = trb + 8 * (i16 & 3), tr1 = trb + 1024 + ((8 * (i16 & 3)) ^ 16);
  const int kbl = rbl & 3;
  const int dsw0 = (8 * wave + rbl) * 256 + (n & 3) * 32;

#pragma unroll 1
  for (int jb = 0; jb < nblk; ++jb) {
    char* dsj = dsb + (jb & 1) * FB_DS;
    FB_STAMP(jb, 0);
    if (keywave) {
      const char* Qj = Qb + jb * 4096;
      const char* Dj = Db + jb * 4096;
      f32x16 sacc, dacc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sinit + 32 * jb + 8 * j + 4 * g);
        const f32x4 b = *reinterpret_cast<const f32x4*>(dinit + 32 * jb + 8 * j + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { sacc[4 * j + r] = a[r]; dacc[4 * j + r] = b[r]; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const opx8 qa = *reinterpret_cast<const opx8*>(Qj + ((s & 1) ? e1 : e0) + 256 * (s >> 1));
        sacc = MFMA_32x32x16(qa, kf[s], sacc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const opx8 da = *reinterpret_cast<const opx8*>(Dj + ((s & 1) ? e1 : e0) + 256 * (s >> 1));
        dacc = MFMA_32x32x16(da, vf[s], dacc, 0, 0, 0);
      }
      FB_STAMP(jb, 1);
      opx8 pf[2], sf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(c * sacc[r]);
        pf[r >> 3][r & 7] = (op_t)pr;
        sf[r >> 3][r & 7] = (op_t)(pr * dacc[r]);
      }
      FB_STAMP(jb, 2);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const opx8 ad = tr_frag8(Dj + t * 2048 + dh * 256, tr0, tr1);
          dvv[dh] = MFMA_32x32x16(ad, pf[t], dvv[dh], 0, 0, 0);
          const opx8 aq = tr_frag8(Qj + t * 2048 + dh * 256, tr0, tr1);
          dk[dh] = MFMA_32x32x16(aq, sf[t], dk[dh], 0, 0, 0);
        }
      // dS^T of this wave's keys -> [key][query] image: 4 consecutive queries (8 bytes) per store
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int slot = 2 * (j & 1) + g;
        opx4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = sf[j >> 1][4 * (j & 1) + r];
        *reinterpret_cast<opx4*>(dsj + dsw0 + (j >> 1) * 128 + ((slot ^ kbl) * 8)) = w;
      }
    }
    FB_STAMP(jb, 3);
    FB_BARRIER();
    FB_STAMP(jb, 4);
  }

  if (keywave) {
    const int key = 32 * wave + n;
    if (key < S) {
      op_t* op = tok_ptr(p.dqkv, p.dqkv_cls, p.ldd, p.mp, sr, seq, key) + HD + h * 64 + 4 * g;
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          opx4 ok, ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) { ok[r] = (op_t)dk[dh][4 * j + r]; ov[r] = (op_t)dvv[dh][4 * j + r]; }
          *reinterpret_cast<opx4*>(op + 32 * dh + 8 * j) = ok;
          *reinterpret_cast<opx4*>(op + HD + 32 * dh + 8 * j) = ov;
        }
    }
  }
#if PVRL_FB_TRACE
  FB_STAMP(7, 4);
  if (tracing && p.dvec)
    for (int i = lane; i < 64; i += 64) reinterpret_cast<unsigned long long*>(p.dvec)[wave * 64 + i] = reinterpret_cast<unsigned long long*>(smem + FB_LDS)[wave * 64 + i];
#endif
}

}  // namespace

// true when the fused kernel covers the case (attn_mfma.hip falls back to the two-pass kernels otherwise)
bool pvrl_attn_bwd_fused_ok(const AttnArgs& p) {
  if (p.causal || p.kpm) return false;
  if (p.mp.S <= 80 || p.mp.S > FB_ROWS) return false;
  int e = 0;
  const float m = frexpf(p.scale, &e);
  return m == 0.5f;      // power of two
}

int pvrl_attn_bwd_fused_launch(const AttnArgs& p, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((p.nseq + 7) / 8) * p.H)), blk(512);
  if (p.mp.S > 192) hipLaunchKernelGGL(attn_bwd_fused_kernel<7>, grid, blk, 0, s, p);      // 7 query blocks, loops unrolled
  else hipLaunchKernelGGL(attn_bwd_fused_kernel<0>, grid, blk, 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
