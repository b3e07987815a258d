// bf16 MFMA GEMM, "TN" form (weight gradients), 256x256 tiles of dW with N % 256 == 0 and K % 256 == 0:
//     part[s][N,K] = sum over the slice's rows m of P[m,N]^T . Q[m,K]     (+ column sums of P = bias gradient)
//
// EIGHT waves in PING-PONG, LDS-DMA staging, transposing fragment reads -- the schedule of the persistent NT kernel
// (gemm_nt8_core.h) on the TN operand geometry.  The register-transposed kernel (gemm_tn_core.h, tn_rt8_pair) puts every wave through
// the same serial stage -- wait for its global loads, v_perm + ds_write the next stage, read fragments, 64 MFMAs, barrier -- and its
// ablation (profiles/r4_tn_ablation.txt) shows what that costs: MFMAs + barriers alone 0.57 ms, loads + staging + reads alone 0.66 ms,
// together 1.03 ms.  Here
//   * a 64-row stage of P and Q (4 x 16 KiB half-tiles P0 P1 Q0 Q1 = 128 columns each) goes from global memory to LDS by raw
//     `buffer_load_dwordx4 ... lds` (no registers, no VALU, no ds_write), already in the [4 m][16 column] 128-byte blocks that
//     ds_read_b64_tr_b16 transposes: one instruction = 4 rows x 256 bytes, lane l = (block l >> 3, row (l >> 1) & 3, half l & 1);
//     rows behind the slice's end read as zero through the descriptor's bounds check;
//   * wave (wm, wn) = (wave >> 2, wave & 3) owns n rows [64 wm, +64) of both n halves and k columns [32 wn, +32) of both k halves:
//     8 x 4 accumulators of v_mfma_f32_16x16x32 (128 registers);
//   * waves 0-3 (one per SIMD) run one barrier interval ahead of waves 4-7: a stage is two phases (n half 0 / 1, both m halves), each
//     a MEMORY segment (16-32 transposing reads into registers, the LDS-DMA instructions of the stage two ahead, the counted wait)
//     and a COMPUTE segment (32 MFMAs on registers), so one wave of every SIMD computes while the other reads;
//   * loads run two stages ahead with counted waits (vmcnt(8)), as in the NT kernel -- same hazard argument (gemm_nt8_core.h).
// Measured and dropped (profiles/r4_tn_ablation.txt): a ring of five 32-row stages (128 KiB in flight instead of 96: 3 % slower), one
// descriptor per stage instead of the stage offset in the scalar offset (4 % slower), an L2 touch 3-6 stages ahead by the tile's first
// reader in its XCD (7-10 % slower).
// Every accumulator sums its rows in the same order as tn_rt8_pair (stage by stage, m half 0 then 1): results are bit-identical.
// Replaces the autograd backward of nn.Linear's weight / bias (lib/models/vit.py:54-60, 75-92, 133; tools/train_net.py:176-181).
#pragma once
#include "gemm_tn_core.h"

namespace {

typedef __amdgpu_buffer_rsrc_t tn_rsrc_t;
constexpr int TN8_HALF = 64 * 128 * 2;   // one half-tile: 64 m x 128 columns x 2 B = 16 KiB, [m / 4][column / 16] blocks of 128 B
constexpr int TN8_BUF = 4 * TN8_HALF;    // one stage: P0 P1 Q0 Q1

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one 1 KiB LDS-DMA copy through a buffer descriptor: lane l's 16 bytes at r.base + soff + voff land at LDS byte lds + 16 l; offsets
// past the descriptor's size read as zero.  (s_nop 4: SGPRs written by v_readfirstlane -> vector-memory instruction; s_nop 0: M0.)
// Probe builds only (tools/probe/tn_ab.py; results are garbage, only the time means something): bit 0 = no LDS-DMA, bit 1 = no fragment
// reads, bit 2 = no MFMAs
#ifndef PVRL_TN8_ABLATE
#define PVRL_TN8_ABLATE 0
#endif
#ifndef PVRL_TN8_PH2
#define PVRL_TN8_PH2 1      // 0: four phases of 16 MFMAs per K-tile (A/B builds)
#endif
__device__ __forceinline__ void tn8_dma16(tn_rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  if (PVRL_TN8_ABLATE & 1) return;
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

#define TN8_BARRIER()                       \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
// end of a memory segment: this wave's fragment reads have completed BEFORE the barrier (the slot they came from may be re-staged
// by the other group right behind it)
#define TN8_MEM_END()                                     \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    TN8_BARRIER();                                        \
    __builtin_amdgcn_s_setprio(1);                        \
  } while (0)
#define TN8_CMP_END()                 \
  do {                                \
    __builtin_amdgcn_s_setprio(0);    \
    TN8_BARRIER();                    \
  } while (0)

__device__ __forceinline__ void tn8_pair(const GemmTN& p, const int pair, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int s = pair / p.tiles_nk;
  const int rem = pair - s * p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int mbeg = s * p.Ms;
  const int rows = min(p.M, mbeg + p.Ms) - mbeg;
  const int q = lane >> 4, i = lane & 15;
  float* part = p.part + (long)s * p.N * p.K;
  const bool has_csum = p.cpart != nullptr && tk == 0;
  const bool do_csum = has_csum && wn == wm;               // waves 0 and 5 (two different SIMDs) sum the columns of their P fragments (VALU in the MFMAs' shadow)
  if (rows <= 0) {                                         // empty slice: its partial tile must still be zero
    for (int mh = 0; mh < 2; ++mh)
      for (int t = 0; t < 4; ++t)
        for (int kt = 0; kt < 4; ++kt)
          *reinterpret_cast<f32x4*>(part + (long)(n0 + 128 * mh + 64 * wm + 16 * t + i) * p.K + k0 + 128 * (kt >> 1) + 32 * wn +
                                    16 * (kt & 1) + 4 * q) = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (do_csum && q == 0)
      for (int mh = 0; mh < 2; ++mh)
        for (int t = 0; t < 4; ++t) p.cpart[(long)s * p.N + n0 + 128 * mh + 64 * wm + 16 * t + i] = 0.f;
    return;
  }
  const int nk = (rows + 63) >> 6;

  // ---- LDS-DMA: instruction e (0, 1) of this wave copies m rows 4 (2 wave + e) + ((lane >> 1) & 3) of a half-tile, 16 bytes at
  // column 16 (lane >> 3) + 8 (lane & 1) -- eight consecutive lanes fill one [4 m][16 column] block ----
  unsigned voffP[2][2], voffQ[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned row = 4u * (2u * (unsigned)wave + (unsigned)e) + (((unsigned)lane >> 1) & 3u);
    const unsigned col = 16u * ((unsigned)lane >> 3) + 8u * ((unsigned)lane & 1u);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      voffP[h][e] = row * (unsigned)p.ldp * 2u + (128u * h + col) * 2u;
      voffQ[h][e] = row * (unsigned)p.ldq * 2u + (128u * h + col) * 2u;
    }
  }
  const unsigned stepP = 64u * (unsigned)p.ldp * 2u, stepQ = 64u * (unsigned)p.ldq * 2u;     // bytes per stage
  // descriptors over the slice's rows of the tile's 256 columns; the first byte behind (rows - 1, column 255) is out of range.  The stage
  // offset travels in the instruction's scalar offset, which gfx950 includes in the range check (tests/kernel_checks.py
  // check_gemm_tn_rows_behind_the_end); a descriptor per stage was measured 4 % slower (scalar work in every memory segment).
  const tn_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.P + (long)mbeg * p.ldp + n0), 0, (int)(((long)(rows - 1) * p.ldp + 256) * 2), 0x00020000);
  const tn_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Q + (long)mbeg * p.ldq + k0), 0, (int)(((long)(rows - 1) * p.ldq + 256) * 2), 0x00020000);
  auto issueP1 = [&](int h, int e, unsigned lb, unsigned soff) { tn8_dma16(rP, voffP[h][e], soff, lb + h * TN8_HALF + wave * 2048 + e * 1024); };
  auto issueQ1 = [&](int c, int e, unsigned lb, unsigned soff) { tn8_dma16(rQ, voffQ[c][e], soff, lb + (2 + c) * TN8_HALF + wave * 2048 + e * 1024); };
  auto issueP = [&](int h, unsigned lb, unsigned soff) { issueP1(h, 0, lb, soff); issueP1(h, 1, lb, soff); };
  auto issueQ = [&](int c, unsigned lb, unsigned soff) { issueQ1(c, 0, lb, soff); issueQ1(c, 1, lb, soff); };

  // ---- fragments: lane (i, q) of a 16-column tile at block column cb, m half ks: rows 32 ks + 8 q + (0..7) = blocks (8 ks + 2 q + h)
  // x cb, 8 bytes at 8 i of each; the transposing read hands lane i column i's four rows ----
  const int xb = q * 2048 + wm * 512 + i * 8;              // P: block column 4 wm + t
  const int wb = q * 2048 + wn * 256 + i * 8;              // Q: block column 2 wn + h2
  opx8 ra[4][2], rb0[2][2], rb1[2][2];                     // [n tile][m half], [h2][m half] (rb0 / rb1: k half 0 / 1)
  f32x4 acc[2][4][4];                                      // [n half][n tile][2 c + h2]
  float cacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (PVRL_TN8_ABLATE & 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ra[t][ks] = (opx8)(op_t)0.5f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { rb0[h][ks] = (opx8)(op_t)0.25f; rb1[h][ks] = (opx8)(op_t)0.125f; }
  }
  opx2 ones2;
  ones2[0] = (op_t)1.0f; ones2[1] = (op_t)1.0f;
  auto rdP = [&](const char* buf, int mh, int ks) {        // 8 reads: the n half's 4 tiles of this wave, m half ks
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (PVRL_TN8_ABLATE & 2) asm volatile("" : "+v"(ra[t][ks]));
      else ra[t][ks] = tr_frag(buf + mh * TN8_HALF + ks * 8192 + t * 128, xb, xb + 1024);
    }
  };
  auto rdQ = [&](const char* buf, int ks) {                // 8 reads: both k halves, m half ks (kept for both n halves)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (PVRL_TN8_ABLATE & 2) {
        asm volatile("" : "+v"(rb0[h][ks]), "+v"(rb1[h][ks]));
      } else {
        rb0[h][ks] = tr_frag(buf + 2 * TN8_HALF + ks * 8192 + h * 128, wb, wb + 1024);
        rb1[h][ks] = tr_frag(buf + 3 * TN8_HALF + ks * 8192 + h * 128, wb, wb + 1024);
      }
    }
  };
  auto mmk = [&](f32x4 (&a)[4][4], int mh, int ks) {       // 16 MFMAs on 16 different accumulators
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (PVRL_TN8_ABLATE & 4) {
          asm volatile("" :: "v"(rb0[h][ks]), "v"(rb1[h][ks]), "v"(ra[t][ks]));
        } else {
          a[t][h] = MFMA_16x16x32(rb0[h][ks], ra[t][ks], a[t][h], 0, 0, 0);
          a[t][2 + h] = MFMA_16x16x32(rb1[h][ks], ra[t][ks], a[t][2 + h], 0, 0, 0);
        }
      }
    if (do_csum) {                                         // column sums of P
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < 4; ++d) cacc[mh][t] = FDOT2_F32((opx2){ra[t][ks][2 * d], ra[t][ks][2 * d + 1]}, ones2, cacc[mh][t], false);
    }
  };
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- cold start: stage 0, stage 1 without its P1; wait for stage 0 ----
  {
    const unsigned l0 = sbase, l1 = sbase + TN8_BUF;
    issueQ(0, l0, 0); issueP(0, l0, 0); issueQ(1, l0, 0); issueP(1, l0, 0);
    issueQ(0, l1, stepQ); issueP(0, l1, stepP); issueQ(1, l1, stepQ);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    TN8_BARRIER();
  }
  if (wm == 1) TN8_BARRIER();                              // waves 4-7 run one barrier interval behind waves 0-3
  for (int kt = 0; kt < nk; ++kt) {
    // the load stream, two cursors: P1 of stage kt + 1 (its slot is free behind phase 3 of stage kt - 1) and P0 / Q0 / Q1 of
    // stage kt + 2 (this stage's slot, free behind phase 1)
    const bool on1 = kt + 1 < nk, on2 = kt + 2 < nk;
    const unsigned sP1 = (unsigned)(kt + 1) * stepP, sP2 = (unsigned)(kt + 2) * stepP, sQ2 = (unsigned)(kt + 2) * stepQ;
    const int cur = kt & 1;
    const char* rbuf = smem + cur * TN8_BUF;
    const unsigned lb = sbase + cur * TN8_BUF, lo = sbase + (cur ^ 1) * TN8_BUF;
    // every memory segment issues its fragment reads FIRST and the LDS-DMA behind them (a DMA instruction blocks its wave while the
    // CU's address path takes the 1 KiB; the reads complete underneath)
#if PVRL_TN8_PH2
    // TWO phases per stage (n half, both m halves: 32 MFMAs each): half the barriers of the four-phase form below, same copies and waits; ~1 % faster
    rdP(rbuf, 0, 0); rdQ(rbuf, 0); rdP(rbuf, 0, 1); rdQ(rbuf, 1);
    if (on1) {
      issueP1(1, 0, lo, sP1); issueP1(1, 1, lo, sP1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN8_MEM_END();
    mmk(acc[0], 0, 0);
    mmk(acc[0], 0, 1);
    TN8_CMP_END();
    rdP(rbuf, 1, 0); rdP(rbuf, 1, 1);
    if (on2) {
      issueQ1(0, 0, lb, sQ2); issueQ1(0, 1, lb, sQ2); issueP1(0, 0, lb, sP2);
      issueP1(0, 1, lb, sP2); issueQ1(1, 0, lb, sQ2); issueQ1(1, 1, lb, sQ2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN8_MEM_END();
    mmk(acc[1], 1, 0);
    mmk(acc[1], 1, 1);
    TN8_CMP_END();
#else
    // ---- phase 0: n half 0, m half 0 ----
    rdP(rbuf, 0, 0);
    rdQ(rbuf, 0);
    if (on1) issueP1(1, 0, lo, sP1);
    TN8_MEM_END();
    mmk(acc[0], 0, 0);
    TN8_CMP_END();
    // ---- phase 1: n half 0, m half 1; behind it this slot's P0 / Q0 / Q1 are free.  Waits for this stage's P1 (read in phase 2) ----
    rdP(rbuf, 0, 1);
    rdQ(rbuf, 1);
    if (on1) {
      issueP1(1, 1, lo, sP1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // younger than this stage's P1: 3 + 3 of the previous stage, 1 + 1 of this one
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN8_MEM_END();
    mmk(acc[0], 0, 1);
    TN8_CMP_END();
    // ---- phase 2: n half 1, m half 0 ----
    rdP(rbuf, 1, 0);
    if (on2) { issueQ1(0, 0, lb, sQ2); issueQ1(0, 1, lb, sQ2); issueP1(0, 0, lb, sP2); }
    TN8_MEM_END();
    mmk(acc[1], 1, 0);
    TN8_CMP_END();
    // ---- phase 3: n half 1, m half 1; behind it this slot's P1 is free.  Waits for the next stage's P0 / Q0 / Q1 ----
    rdP(rbuf, 1, 1);
    if (on2) {
      issueP1(0, 1, lb, sP2); issueQ1(1, 0, lb, sQ2); issueQ1(1, 1, lb, sQ2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // younger than the next stage's P0 / Q0 / Q1: 1 + 1 + 3 + 3 of this stage
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN8_MEM_END();
    mmk(acc[1], 1, 1);
    TN8_CMP_END();
  #endif
  }
  if (wm == 0) TN8_BARRIER();                              // both groups leave the loop together

  // lane holds n = n0 + 128 mh + 64 wm + 16 t + i, k = k0 + 128 c + 32 wn + 16 h2 + 4 q + (0..3)
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float* row = part + (long)(n0 + 128 * mh + 64 * wm + 16 * t + i) * p.K + k0 + 32 * wn + 4 * q;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) *reinterpret_cast<f32x4*>(row + 128 * (kt >> 1) + 16 * (kt & 1)) = acc[mh][t][kt];
    }
  if (do_csum) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = cacc[mh][t];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) p.cpart[(long)s * p.N + n0 + 128 * mh + 64 * wm + 16 * t + i] = v;
      }
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(GemmTN p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TN8_BUF];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int pair = xcd * p.Ms_pairs + jj;
  if (jj >= p.Ms_pairs || pair >= p.npairs) return;
  tn8_pair(p, pair, smem);
}

// several weight gradients in one launch (TnGroup, gemm_tn_core.h): every problem of the group has N % 256 == 0 and K % 256 == 0
__global__ __launch_bounds__(512, 2) void gemm_tn8_grouped_kernel(TnGroup g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TN8_BUF];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int gp = xcd * g.per_xcd + jj;
  if (jj >= g.per_xcd || gp >= g.total) return;
  int q = 0;
#pragma unroll
  for (int t = 1; t < TN_GROUP_MAX; ++t)
    if (t < g.nprob && gp >= g.first[t]) q = t;
  const GemmTN p = g.prob[q];
  tn8_pair(p, gp - g.first[q], smem);
}

}  // namespace
