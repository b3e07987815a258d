// bf16 MFMA GEMM, "TN" form (weight gradients):
//     dW[N,K] = sum_m P[m,N]^T . Q[m,K]        (+ optional column sums of P = bias gradient)
// P = upstream gradient dY (bf16, [M,N]), Q = saved layer input (bf16, [M,K]); fp32 out.
// This is the autograd backward of every nn.Linear on the reference hot path
// (lib/models/vit.py:54-60, 75-92, 133, 174-180; tools/train_net.py:176-181 loss.backward()).
//
// gfx950 design: the reduction index m is the *row* index of both operands, so MFMA
// fragments need 8 consecutive m per lane = a column walk of a row-major tile.  Tiles
// are register-staged (global_load_dwordx4 -> ds_write_b128) into LDS as contiguous
// [4 m][16 col] 128-byte blocks and fragments are fetched with the CDNA4 transposing
// LDS read ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of its
// block: 4 consecutive m).  Output tile 128(n) x 128(k), 4 waves 2x2, 32 m per step,
// double-buffered.  M is split into `splits` slices (fills 256 CUs although N*K/128^2 is
// only 36..144 tiles); slices write fp32 partial tiles that a second tiny kernel sums
// (deterministic, no atomics).  Block order is slice-major so the workgroups alive at
// one time stream the same rows of P and Q through L2 / Infinity Cache.
#pragma once
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct GemmTN {
  const op_t* P; long ldp;
  const op_t* Q; long ldq;
  int M, N, K, Ms, tiles_k, tiles_nk;
  float* part;   // [splits][N][K]
  float* cpart;  // [splits][N] or null
  const op_t* zero_page;  // 256 zero bytes (source of out-of-range rows for the LDS-DMA path)
  int npairs, Ms_pairs;   // rt kernel: (slice, tile) pairs in total / per XCD
};

constexpr int TM = 64;   // reduction rows per pipeline stage (two K=32 MFMA steps)

__device__ __forceinline__ opx8 tr_frag(const char* tile, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off1));
  union { struct { s16x4 a, b; } s; opx8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// WN x WK waves per workgroup, each owning a 64(n) x 64(k) block of dW: tile = (64 WN) x (64 WK).
//   <2,2>: 128x128, 4 waves, 64 KiB LDS, 2 workgroups / CU;   <4,4>: 256x256, 16 waves, 128 KiB LDS, 1 / CU
//   (half the L2->LDS bytes per FLOP; every operand row block is shared by 4 waves instead of 2).
template <int WN, int WK>
__global__ __launch_bounds__(64 * WN * WK) void gemm_tn_kernel(GemmTN p) {
  constexpr int NW = WN * WK, NT = 64 * NW;
  constexpr int PB = 4 * WN, QB = 4 * WK;                 // 16-column blocks per row block of the P / Q tile
  constexpr int PBYTES = TM * 64 * WN * 2, QBYTES = TM * 64 * WK * 2, STAGE = PBYTES + QBYTES;
  constexpr int PINST = 16 * WN / 2, QINST = 16 * WK / 2;  // wave-instructions (4 rows x 256 B) per stage
  constexpr int PER = (PINST + QINST) / NW;
  static_assert((PINST + QINST) % NW == 0, "staging must divide evenly over the waves");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / WN, wn = wave % WN;
  // every slice of M lives on ONE XCD (hardware: block b -> XCD b % 8): its rows of P and Q are pulled into that
  // XCD's L2 once and shared by all (n, k) tiles of the slice instead of being re-fetched by all 8 L2s.
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = (j / p.tiles_nk) * 8 + xcd;
  const int rem = j % p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 64 * WN, k0 = tk * 64 * WK;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int nsteps = (mend - mbeg + TM - 1) / TM;

  // staging: wave-instruction `it` copies 4 tile rows x 256 B of P (it < PINST) or Q.  Lanes are assigned
  // (column block, row in block, half) so that 8 consecutive lanes write one whole 128-byte [4][16] block: the 8-lane
  // groups of ds_write_b128 cover 32 distinct banks, and every global row is still read as full 256-byte lines.
  int srow[PER], scol[PER], soff[PER];
  bool isq[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    int it = wave * PER + e;
    isq[e] = it >= PINST;
    if (isq[e]) it -= PINST;
    const int segs = isq[e] ? WK / 2 : WN / 2;             // 256-byte segments per tile row
    const int rg = it / segs, seg = it - rg * segs;
    srow[e] = rg * 4 + ((lane >> 1) & 3);
    const int c8 = seg * 16 + (lane >> 3) * 2 + (lane & 1);  // 16-byte chunk within the tile row
    scol[e] = c8 * 8;
    const int rb = srow[e] >> 2, cb = c8 >> 1;
    const int nb = isq[e] ? QB : PB;
    soff[e] = (isq[e] ? PBYTES : 0) + (rb * nb + (cb ^ ((rb >> 1) & 1))) * 128 + (srow[e] & 3) * 32 + (c8 & 1) * 16;
  }
  u32x4 rg_[PER];
  auto gload = [&](int st) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int m = mbeg + st * TM + srow[e];
      rg_[e] = (u32x4){0u, 0u, 0u, 0u};
      if (m < mend)
        rg_[e] = isq[e] ? *reinterpret_cast<const u32x4*>(p.Q + (long)m * p.ldq + k0 + scol[e])
                        : *reinterpret_cast<const u32x4*>(p.P + (long)m * p.ldp + n0 + scol[e]);
    }
  };
  auto lwrite = [&](int buf) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int e = 0; e < PER; ++e) *reinterpret_cast<u32x4*>(b + soff[e]) = rg_[e];
  };

  // fragment offsets: lane (i, q); rows 8q..8q+3 (h=0) and 8q+4..8q+7 (h=1) of column tile cb
  const int q = lane >> 4, i = lane & 15;
  int poff[4][2], qoff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = 2 * q + h;
      poff[t][h] = (rb * PB + ((wn * 4 + t) ^ (q & 1))) * 128 + i * 8;
      qoff[t][h] = PBYTES + (rb * QB + ((wk * 4 + t) ^ (q & 1))) * 128 + i * 8;
    }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);

  if (nsteps > 0) {
    gload(0);
    lwrite(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) gload(st + 1);
    const char* b = smem + (st & 1) * STAGE;
    if constexpr (NW <= 4) {
      // 2 waves / SIMD: all 32 transposing reads of the stage are scheduled explicitly -- 16 up front, then one read
      // per MFMA while the first K=32 step computes, then the second step.
      opx8 pf0[4], qf0[4], pf1[4], qf1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        qf0[t] = tr_frag(b, qoff[t][0], qoff[t][1]);
        pf0[t] = tr_frag(b, poff[t][0], poff[t][1]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        qf1[t] = tr_frag(b + 8 * QB * 128, qoff[t][0], qoff[t][1]);
        pf1[t] = tr_frag(b + 8 * PB * 128, poff[t][0], poff[t][1]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = MFMA_16x16x32(qf0[kt], pf0[nt], acc[nt][kt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = MFMA_16x16x32(qf1[kt], pf1[nt], acc[nt][kt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      if (do_csum) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[t] += (float)pf0[t][e] + (float)pf1[t][e];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {   // K = 32 MFMA step: row blocks 8 ks .. 8 ks + 7
        opx8 pf[4], qf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          qf[t] = tr_frag(b + ks * 8 * QB * 128, qoff[t][0], qoff[t][1]);
          pf[t] = tr_frag(b + ks * 8 * PB * 128, poff[t][0], poff[t][1]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[nt][kt] = MFMA_16x16x32(qf[kt], pf[nt], acc[nt][kt], 0, 0, 0);
        if (do_csum) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[t] += (float)pf[t][e];
        }
      }
    }
    if (st + 1 < nsteps) lwrite((st + 1) & 1);
    __syncthreads();
  }

  // lane holds n = n0 + wn*64 + nt*16 + i, k = k0 + wk*64 + kt*16 + 4q + reg
  float* part = p.part + (long)s * p.N * p.K;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n0 + wn * 64 + nt * 16 + i;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(part + (long)n * p.K + k) = acc[nt][kt];
    }
  }
  if (do_csum) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = csum[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) p.cpart[(long)s * p.N + n0 + wn * 64 + t * 16 + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register-transposed staging with EIGHT waves (two per SIMD): wave block 128(n) x 64(k) = 8 x 4 blocks of 16x16x32 MFMAs
// (128 accumulator registers), 64-row stages (64 KiB, two slots), every lane still transposes one 8(m) x 8(col) block per
// stage -- half the staging work per MFMA of the four-wave kernels, and a second wave per SIMD to cover LDS / barrier
// latency (the four-wave kernels need ~2,100 cycles per 1,024-cycle MFMA stage).  One register set, one 64-row stage ahead.
// MEASURED (same-process A/B, M = 50,208): wfc1 233 vs 247 us, wfc2 237 vs 248, wqkv 189 vs 190, wproj 76 vs 80 against the
// four-wave 32x32x16 kernel (knob 7); 57.1 vs 57.5 ms per training step -> the default (knob 0 / 8).
// ---------------------------------------------------------------------------------------------------------
constexpr int RT8_TS = 64;
constexpr int RT8_OPB = 8 * 256 * 16;                      // 32 KiB per operand and stage: [m / 8][col][8 m]
constexpr int RT8_STAGE = 2 * RT8_OPB;

__device__ __forceinline__ void tn_rt8_pair(const GemmTN& p, const int pair, char* smem) {
  constexpr int TS = RT8_TS, OPB = RT8_OPB, STAGE = RT8_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = pair / p.tiles_nk;
  const int rem = pair - s * p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  // N and K are multiples of 128, not necessarily of 256: the last tile along either may be half a tile.  Its missing columns
  // are staged as zeros; the waves whose 128 x 64 block lies in the missing half only help with the staging -- no fragment
  // reads, no MFMAs, no stores (MViTv2-S: 96 / 384-wide layers, 11 of every 38 tiles of a stage-3 block are padding).  The wave
  // -> block map is chosen so that those waves are 4..7, ONE PER SIMD (waves w and w + 4 share a SIMD): the other wave of each
  // SIMD then has the matrix pipe to itself.
  const bool nfull = n0 + 256 <= p.N, kfull = k0 + 256 <= p.K;
#ifndef PVRL_TN_SKIP_PAD
#define PVRL_TN_SKIP_PAD 1                                 // 0: A/B builds (tools/build_variant.py) -- MFMAs over the zero half as before
#endif
  const bool remap = PVRL_TN_SKIP_PAD && !nfull;
  const int wn = remap ? (wave >> 2) : (wave & 1);         // 2 x 4 waves: 128 n x 64 k each
  const int wk = remap ? (wave & 3) : (wave >> 1);
  const bool n_ok = nfull || wn == 0, k_ok = kfull || wk < 2;
  const bool compute = !PVRL_TN_SKIP_PAD || (n_ok && k_ok);   // wave-uniform
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int rows = mend - mbeg;
  const int q = lane >> 4, i = lane & 15;
  float* part = p.part + (long)s * p.N * p.K;
  if (rows <= 0) {                                         // empty slice: its partial tile must still be zero
    if (n_ok && k_ok)
      for (int nt = 0; nt < 8; ++nt)
        for (int kt = 0; kt < 4; ++kt)
          *reinterpret_cast<f32x4*>(part + (long)(n0 + wn * 128 + nt * 16 + i) * p.K + k0 + wk * 64 + kt * 16 + 4 * q) =
              (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.cpart && tk == 0 && wk == 0 && q == 0 && n_ok)
      for (int t = 0; t < 8; ++t) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = 0.f;
    return;
  }
  const int nsteps = (rows + TS - 1) / TS;

  // staging role: waves 0-3 bring the P rows of the stage, waves 4-7 the Q rows; lane = one 8-row x 8-column block
  const bool isq = wave >= 4;
  const long ld2 = (isq ? p.ldq : p.ldp) * 2;              // row pitch in bytes
  const int l256 = (wave & 3) * 64 + lane;
  const int g = l256 >> 5, cg = l256 & 31;                 // 8-row block of the stage, 8-column group
  const char* ubase = reinterpret_cast<const char*>(isq ? p.Q + k0 : p.P + n0) + (long)mbeg * ld2;
  const bool colok = (isq ? kfull : nfull) || cg < 16;        // this lane's 8 columns exist
  const unsigned loff = (unsigned)(8 * g * ld2 + (colok ? cg : 0) * 16);   // (a lane of the missing half re-reads column group 0; masked in wait_set)
  const int wr = (isq ? OPB : 0) + g * 4096 + (cg >> 1) * 256 + (cg & 1) * 128 + ((cg & 7) << 4);
  const bool tile_full = isq ? kfull : nfull;                 // wave-uniform (waves 0-3 stage P, 4-7 stage Q)
  const unsigned colkeep = colok ? 0xffffffffu : 0u;
  // Round 3: a half tile's stages used to take the masked path below for EVERY stage -- plain loads whose values the mask consumes
  // at once, i.e. an s_waitcnt vmcnt(0) inside gload and the load latency exposed once per stage (2.45 us per stage at N = 128
  // against 1.9 for full tiles).  They now take the asm loads like everyone else; the mask is applied when the set is waited for.
  // Probe builds only (tools/probe/tn_ab.py; results are garbage, only the time means something): bit 0 = no global loads, bit 1 = no
  // twrites (v_perm + ds_write_b128), bit 2 = no fragment reads, bit 3 = no MFMAs, bit 4 = no barriers
#ifndef PVRL_TN_ABLATE
#define PVRL_TN_ABLATE 0
#endif
  auto gload = [&](u32x4* r, int st) {
    if (PVRL_TN_ABLATE & 1) return;
    if ((st + 1) * TS <= rows) {
      const char* b = ubase + (long)st * TS * ld2;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[e]) : "v"(b + e * ld2 + loff) : "memory");
    } else {                                               // ragged or surplus stage: clamp the row, zero what is outside
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = st * TS + 8 * g + e;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ubase + (long)min(max(row, 0), rows - 1) * ld2 + (colok ? cg : 0) * 16);
        const unsigned keep = (row < rows && colok) ? 0xffffffffu : 0u;
        r[e] = v & (u32x4){keep, keep, keep, keep};
      }
    }
  };
  auto wait_set = [&](u32x4* r) {        // ONE register set, one stage (2,048 MFMA cycles per SIMD) ahead: it has landed
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
    if (!tile_full) {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] &= (u32x4){colkeep, colkeep, colkeep, colkeep};
    }
  };
  auto twrite = [&](const u32x4* r, int j, char* slot) {   // column j of the lane's 8: gather its 8 m, store 16 B
    if (PVRL_TN_ABLATE & 2) return;
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      o[d] = __builtin_amdgcn_perm(r[2 * d + 1][j >> 1], r[2 * d][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    *reinterpret_cast<u32x4*>(slot + (wr ^ (j << 4))) = o;
  };
  // fragment t of a K = 32 step: 16-byte chunk (m-block q, column 16 t + i); the slot swizzle has period 4 in t, and the Q
  // addresses are the P addresses plus a constant, so four offsets serve all twelve fragments
  int frd[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int slot = (i & 8) | ((i & 7) ^ ((2 * t + (i >> 3)) & 7));
    frd[t] = q * 4096 + (slot << 4) + t * 256;
  }
  const int pbase = wn * 2048, qbase = OPB + wk * 1024;
  auto rfrag = [&](const char* slot, int off, int h) {
    if (PVRL_TN_ABLATE & 4) { opx8 z = (opx8)(op_t)0.5f; asm volatile("" : "+v"(z)); return z; }
    return *reinterpret_cast<const opx8*>(slot + off + h * 16384);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  opx2 ones2;
  ones2[0] = (op_t)1.0f; ones2[1] = (op_t)1.0f;
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(PVRL_TN_ABLATE & 16)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  u32x4 ra[8];
  // one stage: two K = 32 MFMA steps from slot `rs`; the 8 transposed columns of register set r go to slot `ws`
  auto step = [&](const char* rs, const u32x4* r, char* ws) {
    if (!compute) {                                        // a wave of the missing half: staging only
#pragma unroll
      for (int j = 0; j < 8; ++j) twrite(r, j, ws);
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      opx8 qf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t] = rfrag(rs + qbase, frd[t], h);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        opx8 pf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) pf[t] = rfrag(rs + pbase + half * 1024, frd[t], h);
        if (do_csum) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int d = 0; d < 4; ++d)
              cacc[4 * half + nt] =
                  FDOT2_F32((opx2){pf[nt][2 * d], pf[nt][2 * d + 1]}, ones2, cacc[4 * half + nt], false);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            if (PVRL_TN_ABLATE & 8) asm volatile("" :: "v"(qf[kt]), "v"(pf[nt]));
            else acc[4 * half + nt][kt] = MFMA_16x16x32(qf[kt], pf[nt], acc[4 * half + nt][kt], 0, 0, 0);
          }
          if (nt & 1) twrite(r, 4 * h + 2 * half + (nt >> 1), ws);
        }
      }
    }
  };

  char* slot0 = smem;
  char* slot1 = smem + STAGE;
  gload(ra, 0);
  wait_set(ra);
#pragma unroll
  for (int j = 0; j < 8; ++j) twrite(ra, j, slot0);
  gload(ra, 1);
  // invariant at the top of stage st: slot st % 2 is completed by the barrier, ra holds stage st + 1 (in flight)
  for (int st = 0; st < nsteps; st += 2) {
    lds_barrier();
    wait_set(ra);
    step(slot0, ra, slot1);                                // compute stage st from slot0, stage st + 1 -> slot1
    gload(ra, st + 2);
    if (st + 1 >= nsteps) break;
    lds_barrier();
    wait_set(ra);
    step(slot1, ra, slot0);                                // compute stage st + 1 from slot1, stage st + 2 -> slot0
    gload(ra, st + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (n_ok && k_ok) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int n = n0 + wn * 128 + nt * 16 + i;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        *reinterpret_cast<f32x4*>(part + (long)n * p.K + k0 + wk * 64 + kt * 16 + 4 * q) = acc[nt][kt];
    }
  }
  if (do_csum && n_ok) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v = cacc[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = v;
    }
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn_rt8_kernel(GemmTN p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RT8_STAGE];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int pair = xcd * p.Ms_pairs + jj;
  if (jj >= p.Ms_pairs || pair >= p.npairs) return;
  tn_rt8_pair(p, pair, smem);
}

// Several weight gradients in ONE launch: the (slice, tile) pairs of up to 8 problems are laid end to end and dealt to the
// XCDs in contiguous chunks.  A Linear's dW has 9-36 tiles of 256x256; alone, each needs 7-28 row slices to fill 256 CUs
// (short reduction loops, 67 MB of fp32 partials per call); a transformer block's seven dW together have 153 tiles, so 5
// slices give 765 equal work items = 2.99 rounds of 256, with 5x longer loops and 2.3x less partial traffic.
constexpr int TN_GROUP_MAX = 8;
struct TnGroup {
  int nprob, total, per_xcd;
  int first[TN_GROUP_MAX + 1];     // first global pair index of each problem
  GemmTN prob[TN_GROUP_MAX];
};
__global__ __launch_bounds__(512, 2) void gemm_tn_rt8_grouped_kernel(TnGroup g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RT8_STAGE];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int gp = xcd * g.per_xcd + jj;
  if (jj >= g.per_xcd || gp >= g.total) return;
  int q = 0;
#pragma unroll
  for (int t = 1; t < TN_GROUP_MAX; ++t)
    if (t < g.nprob && gp >= g.first[t]) q = t;
  const GemmTN p = g.prob[q];
  tn_rt8_pair(p, gp - g.first[q], smem);
}

// Gradient outputs, common tail of every reduce below.  `gscale` (device scalar or null): the sum is multiplied by it -- the fp16-operand
// flavour runs a backward in S-scaled units (engine.GradStore.begin_scaled) and the kernel that writes a parameter gradient takes the
// scale out again, instead of a separate pass over the gradient buffer; what is already in `out` (beta) is in true units.  `nonfinite`
// (device flag or null): raised when a value written is inf / nan -- the optimiser's "skip this step" flag (optim.hip), checked where
// the gradient is produced instead of by a scan of the whole buffer.  The flag is only ever raised here.
__device__ __forceinline__ bool nonfinite1(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ bool nonfinite4(const f32x4& v) { return nonfinite1(v[0]) | nonfinite1(v[1]) | nonfinite1(v[2]) | nonfinite1(v[3]); }
__device__ __forceinline__ void raise_if(bool bad, float* flag) {
  if (flag && __any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;
}

// out[n][k] = beta*out + gscale * sum_s part[s][n][k];  bias_out[n] = beta*bias_out + gscale * sum_s cpart[s][n]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cpart,
                                                        int splits, long NK, int N, float beta,
                                                        float* __restrict__ out, float* __restrict__ bias_out,
                                                        const float* __restrict__ gscale, float* __restrict__ nonfinite) {
  const long idx4 = (long)blockIdx.x * 256 + threadIdx.x;
  const long n4 = NK >> 2;
  const float gsc = gscale ? *gscale : 1.f;
  bool bad = false;
  if (idx4 < n4) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[idx4];
    for (int s = 1; s < splits; ++s) {
      const f32x4 b = reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
      a += b;
    }
    if (gscale) a *= gsc;
    if (beta != 0.f) a += beta * reinterpret_cast<f32x4*>(out)[idx4];
    reinterpret_cast<f32x4*>(out)[idx4] = a;
    bad = nonfinite4(a);
  } else if (bias_out && idx4 - n4 < N) {
    const int n = (int)(idx4 - n4);
    float a = 0.f;
    for (int s = 0; s < splits; ++s) a += cpart[(long)s * N + n];
    if (gscale) a *= gsc;
    if (beta != 0.f) a += beta * bias_out[n];
    bias_out[n] = a;
    bad = nonfinite1(a);
  }
  raise_if(bad, nonfinite);
}

// The same for SMALL outputs (MViT's 96..768-wide layers: a 128 x 128 dW is 4,096 float4s = 16 workgroups above, each
// thread walking up to 256 slices serially): 64 float4s x 4 slice-quarters per workgroup, quarters combined in LDS in a
// fixed order -- four times the workgroups and a quarter of the serial chain.
__global__ __launch_bounds__(256) void tn_reduce_small_kernel(const float* __restrict__ part, const float* __restrict__ cpart,
                                                              int splits, long NK, int N, float beta,
                                                              float* __restrict__ out, float* __restrict__ bias_out,
                                                              const float* __restrict__ gscale, float* __restrict__ nonfinite) {
  __shared__ f32x4 red[3][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long idx4 = (long)blockIdx.x * 64 + o;
  const long n4 = NK >> 2;
  const bool is_w = idx4 < n4, is_b = !is_w && bias_out && idx4 - n4 < N;
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (is_w) {
    for (int s = sl; s < splits; s += 4) a += reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
  } else if (is_b) {
    for (int s = sl; s < splits; s += 4) a[0] += cpart[(long)s * N + (int)(idx4 - n4)];
  }
  if (sl > 0) red[sl - 1][o] = a;
  __syncthreads();
  if (sl == 0) {
    a = (a + red[0][o]) + (red[1][o] + red[2][o]);
    if (gscale) a *= *gscale;
    bool bad = false;
    if (is_w) {
      if (beta != 0.f) a += beta * reinterpret_cast<f32x4*>(out)[idx4];
      reinterpret_cast<f32x4*>(out)[idx4] = a;
      bad = nonfinite4(a);
    } else if (is_b) {
      const int n = (int)(idx4 - n4);
      const float v = beta != 0.f ? a[0] + beta * bias_out[n] : a[0];
      bias_out[n] = v;
      bad = nonfinite1(v);
    }
    raise_if(bad, nonfinite);
  }
}

// The reduction writing straight into an UN-PADDED destination: out[n][k] for n < nv, k < kv with leading dimension ldo
// (any alignment: scalar stores), bias_out[n] for n < nv.  The MViT engine's padded weight gradients (96 -> 128, 441 -> 512
// ...) land in parameter.grad this way instead of through a padded temporary and two 5-us copy kernels per weight.
__global__ __launch_bounds__(256) void tn_reduce_into_kernel(const float* __restrict__ part, const float* __restrict__ cpart,
                                                             int splits, long NK, int N, int K, float beta,
                                                             float* __restrict__ out, long ldo, int nv, int kv,
                                                             float* __restrict__ bias_out,
                                                             const float* __restrict__ gscale, float* __restrict__ nonfinite) {
  __shared__ f32x4 red[3][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long idx4 = (long)blockIdx.x * 64 + o;
  const long n4 = NK >> 2;
  const bool is_w = idx4 < n4, is_b = !is_w && bias_out && idx4 - n4 < N;
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (is_w) {
    for (int s = sl; s < splits; s += 4) a += reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
  } else if (is_b) {
    for (int s = sl; s < splits; s += 4) a[0] += cpart[(long)s * N + (int)(idx4 - n4)];
  }
  if (sl > 0) red[sl - 1][o] = a;
  __syncthreads();
  if (sl == 0) {
    a = (a + red[0][o]) + (red[1][o] + red[2][o]);
    if (gscale) a *= *gscale;
    bool bad = false;
    if (is_w) {
      const int k4 = K >> 2;
      const int n = (int)(idx4 / k4), k = (int)(idx4 - (long)n * k4) * 4;
      if (n < nv) {
        float* dst = out + (long)n * ldo + k;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < kv) {
            const float v = beta != 0.f ? a[e] + beta * dst[e] : a[e];
            dst[e] = v;
            bad |= nonfinite1(v);
          }
      }
    } else if (is_b) {
      const int n = (int)(idx4 - n4);
      if (n < nv) {
        const float v = beta != 0.f ? a[0] + beta * bias_out[n] : a[0];
        bias_out[n] = v;
        bad = nonfinite1(v);
      }
    }
    raise_if(bad, nonfinite);
  }
}

// the same for every problem of a grouped launch in ONE kernel (seven 10-us launches per transformer block otherwise)
constexpr int TN_RED_MAX = 8;
struct TnReduceGroup {
  int nprob, splits;
  int first[TN_RED_MAX + 1];                 // first block of each problem
  const float* part[TN_RED_MAX]; const float* cpart[TN_RED_MAX];
  float* out[TN_RED_MAX]; float* bias_out[TN_RED_MAX];
  long NK[TN_RED_MAX]; int N[TN_RED_MAX]; float beta[TN_RED_MAX];
  const float* gscale[TN_RED_MAX]; float* nonfinite[TN_RED_MAX];
};
__global__ __launch_bounds__(256) void tn_reduce_grouped_kernel(TnReduceGroup g) {
  int q = 0;
#pragma unroll
  for (int t = 1; t < TN_RED_MAX; ++t)
    if (t < g.nprob && (int)blockIdx.x >= g.first[t]) q = t;
  const float* __restrict__ part = g.part[q];
  const float* __restrict__ cpart = g.cpart[q];
  float* __restrict__ out = g.out[q];
  float* __restrict__ bias_out = g.bias_out[q];
  const long NK = g.NK[q];
  const int N = g.N[q];
  const float beta = g.beta[q];
  const float* gscale = g.gscale[q];
  const float gsc = gscale ? *gscale : 1.f;
  const long idx4 = (long)((int)blockIdx.x - g.first[q]) * 256 + threadIdx.x;
  const long n4 = NK >> 2;
  bool bad = false;
  if (idx4 < n4) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[idx4];
    for (int s = 1; s < g.splits; ++s) a += reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
    if (gscale) a *= gsc;
    if (beta != 0.f) a += beta * reinterpret_cast<f32x4*>(out)[idx4];
    reinterpret_cast<f32x4*>(out)[idx4] = a;
    bad = nonfinite4(a);
  } else if (bias_out && idx4 - n4 < N) {
    const int n = (int)(idx4 - n4);
    float a = 0.f;
    for (int s = 0; s < g.splits; ++s) a += cpart[(long)s * N + n];
    if (gscale) a *= gsc;
    if (beta != 0.f) a += beta * bias_out[n];
    bias_out[n] = a;
    bad = nonfinite1(a);
  }
  raise_if(bad, g.nonfinite[q]);
}

}  // namespace
