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
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct GemmTN {
  const bf16* P; long ldp;
  const bf16* Q; long ldq;
  int M, N, K, Ms, tiles_k, tiles_nk;
  float* part;   // [splits][N][K]
  float* cpart;  // [splits][N] or null
  const bf16* zero_page;  // 256 zero bytes (source of out-of-range rows for the LDS-DMA path)
  int npairs, Ms_pairs;   // rt kernel: (slice, tile) pairs in total / per XCD
};

constexpr int TM = 64;   // reduction rows per pipeline stage (two K=32 MFMA steps)

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
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
      bf16x8 pf0[4], qf0[4], pf1[4], qf1[4];
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
          acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf0[kt], pf0[nt], acc[nt][kt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf1[kt], pf1[nt], acc[nt][kt], 0, 0, 0);
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
        bf16x8 pf[4], qf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          qf[t] = tr_frag(b + ks * 8 * QB * 128, qoff[t][0], qoff[t][1]);
          pf[t] = tr_frag(b + ks * 8 * PB * 128, poff[t][0], poff[t][1]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kt], pf[nt], acc[nt][kt], 0, 0, 0);
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

// LDS-DMA variant of the 128x128 tile: the blocked [4][16] LDS image is written directly by global_load_lds (16 B per
// lane, lane-linear destination = exactly one 128-byte block per 8 lanes; the column-block swizzle moves to the source
// address).  Removes the 8 ds_write_b128 + 32 staging VGPRs per thread and stage of the register-staged kernel, whose
// LDS write cycles (~13 clk per wave-instruction) exceeded the MFMA time of a stage.
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(GemmTN p) {
  constexpr int PB = 8, QB = 8;
  constexpr int PBYTES = TM * 128 * 2, STAGE = 2 * PBYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = (j / p.tiles_nk) * 8 + xcd;
  const int rem = j % p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int nsteps = (mend - mbeg + TM - 1) / TM;

  // 8 LDS-DMA instructions per wave and stage: instruction it = wave*8 + e copies 4 tile rows x 256 B of P (it < 16) or Q
  const bf16* src[8];
  long sstep[8];
  int srow[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int it = wave * 8 + e;
    const bool isq = it >= 16;
    const int rg = isq ? it - 16 : it;                 // row block (4 rows)
    const int r = (lane >> 1) & 3, h = lane & 1;
    const int cb = (lane >> 3) ^ ((rg >> 1) & 1);      // source-side swizzle of the 16-column block
    srow[e] = rg * 4 + r;
    const long ld = isq ? p.ldq : p.ldp;
    src[e] = (isq ? p.Q + k0 : p.P + n0) + (long)(mbeg + srow[e]) * ld + cb * 16 + h * 8;
    sstep[e] = (long)TM * ld;
  }
  auto stage = [&](int buf, int st) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int it = wave * 8 + e;
      const bool ok = mbeg + st * TM + srow[e] < mend;
      const bf16* g = ok ? src[e] + (long)st * sstep[e] : p.zero_page + (lane & 7) * 8;
      glds16(g, b + (it >= 16 ? PBYTES : 0) + (it & 15) * 1024);
    }
  };

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

  if (nsteps > 0) stage(0, 0);
  for (int st = 0; st < nsteps; ++st) {
    __syncthreads();   // this wave's LDS-DMA has landed (vmcnt(0)); everyone finished reading the other buffer
    if (st + 1 < nsteps) stage((st + 1) & 1, st + 1);
    const char* b = smem + (st & 1) * STAGE;
    bf16x8 pf0[4], qf0[4], pf1[4], qf1[4];
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
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf0[kt], pf0[nt], acc[nt][kt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf1[kt], pf1[nt], acc[nt][kt], 0, 0, 0);
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
  }

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

// 256(n) x 256(k) tile, 8 waves (2 along n x 4 along k), each wave a 128 x 64 block (8 x 4 MFMA tiles, 128 accumulator
// VGPRs): 24 transposing reads per 32 MFMAs instead of 32 per 32 for the 64x64 wave block, and half the L2->LDS bytes per
// FLOP of the 128x128 tile.  LDS-DMA staging (no staging VGPRs), 2 x 64 KiB stages, one workgroup per CU.
__global__ __launch_bounds__(512, 2) void gemm_tn_w128_kernel(GemmTN p) {
  constexpr int NB = 16;                                   // 16-column blocks per row block (256 columns)
  constexpr int PBYTES = TM * 256 * 2, STAGE = 2 * PBYTES; // 32 KiB + 32 KiB
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = (j / p.tiles_nk) * 8 + xcd;
  const int rem = j % p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int nsteps = (mend - mbeg + TM - 1) / TM;

  const bf16* src[8];
  long sstep[8];
  int srow[8], sdst[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int it = wave * 8 + e;                       // 0..63
    const bool isq = it >= 32;
    const int l = it & 31;
    const int rg = l >> 1, seg = l & 1;                // 4-row group, 256-byte segment of the 512-byte tile row
    const int r = (lane >> 1) & 3, h = lane & 1;
    const int cb = (seg * 8 + (lane >> 3)) ^ ((rg >> 1) & 1);
    srow[e] = rg * 4 + r;
    const long ld = isq ? p.ldq : p.ldp;
    src[e] = (isq ? p.Q + k0 : p.P + n0) + (long)(mbeg + srow[e]) * ld + cb * 16 + h * 8;
    sstep[e] = (long)TM * ld;
    sdst[e] = (isq ? PBYTES : 0) + rg * (NB * 128) + seg * 1024;
  }
  auto stage = [&](int buf, int st) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = mbeg + st * TM + srow[e] < mend;
      const bf16* g = ok ? src[e] + (long)st * sstep[e] : p.zero_page + (lane & 7) * 8;
      glds16(g, b + sdst[e]);
    }
  };

  const int q = lane >> 4, i = lane & 15;
  int poff[8][2], qoff[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rb = 2 * q + h;
#pragma unroll
    for (int t = 0; t < 8; ++t) poff[t][h] = (rb * NB + ((wn * 8 + t) ^ (q & 1))) * 128 + i * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) qoff[t][h] = PBYTES + (rb * NB + ((wk * 4 + t) ^ (q & 1))) * 128 + i * 8;
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);

  if (nsteps > 0) stage(0, 0);
  for (int st = 0; st < nsteps; ++st) {
    __syncthreads();
    if (st + 1 < nsteps) stage((st + 1) & 1, st + 1);
    const char* b = smem + (st & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* bk = b + ks * 8 * NB * 128;
      bf16x8 pf[8], qf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t] = tr_frag(bk, qoff[t][0], qoff[t][1]);
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t] = tr_frag(bk, poff[t][0], poff[t][1]);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kt], pf[nt], acc[nt][kt], 0, 0, 0);
      if (do_csum) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[t] += (float)pf[t][e];
      }
    }
  }

  float* part = p.part + (long)s * p.N * p.K;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int n = n0 + wn * 128 + nt * 16 + i;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(part + (long)n * p.K + k) = acc[nt][kt];
    }
  }
  if (do_csum) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v = csum[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = v;
    }
  }
}

// 256(n) x 256(k) tile, FOUR waves (2 x 2), each wave a 128 x 128 block of dW = 8 x 8 MFMA tiles = 256 accumulator
// registers (one wave per SIMD, 512-register budget).  Per K=32 step a wave issues 32 transposing reads for 64 MFMAs --
// half the LDS read bytes per FLOP of the 64x64 wave block, which is what bounds the kernels above (ds_read_b64_tr_b16
// streams at half the LDS rate).  Staging is a 4-deep ring of 32-row stages (32 KiB each) filled by LDS-DMA three
// stages ahead with counted vmcnt + raw s_barrier; the fragments of step s+1 are read while the MFMAs of step s run.
// Requires M % 64 == 0 (token matrices: 1568 rows per clip, so an even clip count): every slice is an even number of
// whole stages.
__global__ __launch_bounds__(256, 1) void gemm_tn_ring_kernel(GemmTN p) {
  constexpr int TS = 32, NB = 16, NS = 4;
  constexpr int OPB = (TS / 4) * NB * 128;                 // 16 KiB per operand and stage
  constexpr int STAGE = 2 * OPB;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = (j / p.tiles_nk) * 8 + xcd;
  const int rem = j % p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int nsteps = (mend - mbeg) / TS;

  // Staging: waves 0,1 copy P (row blocks 0-3 / 4-7 of the stage), waves 2,3 copy Q; 8 LDS-DMA instructions per wave and
  // stage, instruction e = row block (e>>1) of the wave's four, 256-byte half (e&1).  Address = uniform base (SGPR)
  // + 32-bit lane offset; the column-block swizzle (row blocks 2,3 of every four) only changes the lane offset.
  const bool isq = wave >= 2;
  const long ld = isq ? p.ldq : p.ldp;
  const char* ubase = reinterpret_cast<const char*>(isq ? p.Q + k0 : p.P + n0) + ((long)mbeg + (wave & 1) * 16) * ld * 2;
  unsigned loff[2];
  {
    const int r = (lane >> 1) & 3, h = lane & 1, c = lane >> 3;
    loff[0] = (unsigned)(r * ld * 2 + (c * 16 + h * 8) * 2);
    loff[1] = (unsigned)(r * ld * 2 + ((c ^ 1) * 16 + h * 8) * 2);
  }
  const int dbase = (isq ? OPB : 0) + (wave & 1) * 4 * (NB * 128);
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  auto stage = [&](int st, int buf) {
    const int sc = st < nsteps ? st : nsteps - 1;            // surplus ring slots re-load the last stage (never read)
    const char* g = ubase + (long)sc * TS * ld * 2;
    const unsigned b = smem_base + buf * STAGE + dbase;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      glds16_raw(g + (long)(e >> 1) * 4 * ld * 2 + (e & 1) * 256, loff[(e >> 2) & 1], b + (e >> 1) * (NB * 128) + (e & 1) * 1024);
  };

  // fragment addresses: lane (i, q) reads row blocks 2q (h=0) and 2q+1 (h=1) of column block t ^ (q&1):
  // even t -> base + (q&1)*128 + t*128, odd t -> base - (q&1)*128 + t*128  (t*128 becomes the instruction offset)
  const int q = lane >> 4, i = lane & 15;
  int pb[2][2], qb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rb = 2 * q + h;
    const int sw = (q & 1) * 128;
    pb[h][0] = (rb * NB + wn * 8) * 128 + i * 8 + sw;
    pb[h][1] = (rb * NB + wn * 8) * 128 + i * 8 - sw;
    qb[h][0] = OPB + (rb * NB + wk * 8) * 128 + i * 8 + sw;
    qb[h][1] = OPB + (rb * NB + wk * 8) * 128 + i * 8 - sw;
  }

  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of P: one extra MFMA per P fragment against a fragment of ones (rows of D all equal)
  f32x4 cacc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) cacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // P fragments: ONE set, refreshed in place for step s+1 as soon as their row of MFMAs of step s has issued;
  // Q fragments: two sets (all eight are live for the whole step).
  bf16x8 pf[8], qfa[8], qfb[8];
  auto step = [&](const bf16x8* qc, bf16x8* qn, const char* b) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qc[kt], pf[nt], acc[nt][kt], 0, 0, 0);
      pf[nt] = tr_frag(b + nt * 128, pb[0][nt & 1], pb[1][nt & 1]);
      qn[nt] = tr_frag(b + nt * 128, qb[0][nt & 1], qb[1][nt & 1]);
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
  };
  auto colsum = [&]() {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) cacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[nt], cacc[nt], 0, 0, 0);
  };

  if (nsteps > 0) {     // nsteps is even (M % 64 == 0 and Ms % 64 == 0)
    stage(0, 0); stage(1, 1); stage(2, 2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      pf[t] = tr_frag(smem + t * 128, pb[0][t & 1], pb[1][t & 1]);
      qfa[t] = tr_frag(smem + t * 128, qb[0][t & 1], qb[1][t & 1]);
    }
    // sub-step: wait until the next stage has landed (this wave's loads of the one after stay in flight), barrier (all
    // waves' parts landed; everyone has finished reading the ring slot about to be refilled), refill it, then run the
    // 64 MFMAs of this step while fetching the fragments of the next one.
    for (int st = 0; st < nsteps; st += 2) {
      const int hb = ((st >> 1) & 1) * 2;                     // ring slot of stage st: 0 or 2
      if (do_csum) colsum();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage(st + 3, (hb + 3) & 3);
      step(qfa, qfb, smem + (hb + 1) * STAGE);
      if (do_csum) colsum();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage(st + 4, hb);
      step(qfb, qfa, smem + (hb ^ 2) * STAGE);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  float* part = p.part + (long)s * p.N * p.K;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int n = n0 + wn * 128 + nt * 16 + i;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const int k = k0 + wk * 128 + kt * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(part + (long)n * p.K + k) = acc[nt][kt];
    }
  }
  if (do_csum && q == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = cacc[t][0];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Register-transposed staging ("rt"): measured on MI355X (tools/probe/lds_rate.hip) ds_read_b64_tr_b16 streams at
// 110-170 B/ns/CU against 245-435 B/ns/CU for ds_read_b128, so every kernel above is bound by its transposing reads.
// Here the transpose happens ONCE per element on the way in: each lane loads an 8(m) x 8(col) bf16 block as eight
// 16-byte row segments (32 lanes cover a 512-byte tile row), transposes it inside its own registers with 32
// v_perm_b32, and writes eight 16-byte [col][8 m] chunks; MFMA fragments are then plain ds_read_b128.
// Tile 256(n) x 256(k), four waves of 128 x 128 (256 accumulator registers, one wave per SIMD), 32-row stages,
// two LDS slots (64 KiB), global loads two steps ahead in two 32-register sets, one barrier per step; all of it
// (32 fragment reads, 32 perms, 8 LDS writes, 8 global loads) is interleaved into the step's 64 MFMAs.
// LDS image per operand and stage: chunk (g = m/8, col c) at ((g*16 + c/16)*16 + slot)*16 B with
// slot = (c & 8) | ((c & 7) ^ (c/8 & 7)).  ds_read_b128 is served in lane groups {0-3,12-15,20-27}, ... with 64 banks:
// such a group reads slots {0-7} of one 256-byte window and {8-15} of another -> conflict-free; ds_write_b128 is
// served 8 consecutive lanes at a time with 32 banks: the 8 lanes hold 8 different (c/8 & 7) -> 8 different slots.
// Any M: the last stage of the last slice is loaded row-clamped and zero-filled.
__global__ __launch_bounds__(256, 1) void gemm_tn_rt_kernel(GemmTN p) {
  constexpr int TS = 32;
  constexpr int OPB = 4 * 256 * 16;                        // 16 KiB per operand and stage
  constexpr int STAGE = 2 * OPB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  int s, rem;
  {
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int pair = xcd * p.Ms_pairs + jj;                // (slice, tile) pairs in slice-major order, one chunk per XCD
    if (jj >= p.Ms_pairs || pair >= p.npairs) return;
    s = pair / p.tiles_nk;
    rem = pair - s * p.tiles_nk;
  }
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int rows = mend - mbeg;
  const int nsteps = (((rows + TS - 1) / TS) + 1) & ~1;    // stages, rounded up to even (the surplus one is all zeros)
  if (rows <= 0) {                                         // empty slice: its partial tile must still be zero
    float* part = p.part + (long)s * p.N * p.K;
    const int q = lane >> 4, i = lane & 15;
    for (int nt = 0; nt < 8; ++nt)
      for (int kt = 0; kt < 8; ++kt)
        *reinterpret_cast<f32x4*>(part + (long)(n0 + wn * 128 + nt * 16 + i) * p.K + k0 + wk * 128 + kt * 16 + 4 * q) =
            (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.cpart && tk == 0 && wk == 0 && q == 0)
      for (int t = 0; t < 8; ++t) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = 0.f;
    return;
  }

  // staging role: waves 0,1 bring P rows [16 w, 16 w + 16) of the stage, waves 2,3 the same rows of Q
  const bool isq = wave >= 2;
  const long ld2 = (isq ? p.ldq : p.ldp) * 2;              // row pitch in bytes
  const int rg = lane >> 5, cg = lane & 31;
  const int g = 2 * (wave & 1) + rg;                       // 8-row block of the stage
  const char* ubase = reinterpret_cast<const char*>(isq ? p.Q + k0 : p.P + n0) + (long)mbeg * ld2;
  const unsigned loff = (unsigned)(8 * g * ld2 + cg * 16);
  const int wr = (isq ? OPB : 0) + g * 4096 + (cg >> 1) * 256 + (cg & 1) * 128 + ((cg & 7) << 4);
  auto gload = [&](u32x4* r, int st) {
    if ((st + 1) * TS <= rows) {                           // whole stage (uniform branch; every stage but the last)
      const char* b = ubase + (long)st * TS * ld2;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const u32x4*>(b + e * ld2 + loff);
    } else {                                               // ragged or surplus stage: clamp the row, zero what is outside
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = st * TS + 8 * g + e;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ubase + (long)min(row, rows - 1) * ld2 + cg * 16);
        const unsigned keep = row < rows ? 0xffffffffu : 0u;   // mask, not a branch: keeps the loads unconditional
        r[e] = v & (u32x4){keep, keep, keep, keep};
      }
    }
  };
  auto twrite = [&](const u32x4* r, int j, char* slot) {   // column j of the lane's 8: gather its 8 m, store 16 B
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      o[d] = __builtin_amdgcn_perm(r[2 * d + 1][j >> 1], r[2 * d][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    *reinterpret_cast<u32x4*>(slot + (wr ^ (j << 4))) = o;
  };

  const int q = lane >> 4, i = lane & 15;
  int prd[8], qrd[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int slot = (i & 8) | ((i & 7) ^ ((2 * t + (i >> 3)) & 7));
    prd[t] = wn * 2048 + q * 4096 + (slot << 4);
    qrd[t] = OPB + wk * 2048 + q * 4096 + (slot << 4);
  }
  auto rfrag = [&](const char* slot, int off, int t) {
    return *reinterpret_cast<const bf16x8*>(slot + off + t * 256);
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of P, taken from the P fragments with v_dot2_f32_bf16 against (1, 1)
  float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  bf16x2 ones2;
  ones2[0] = (bf16)1.0f; ones2[1] = (bf16)1.0f;

  // P fragments 0-6 live in ONE register set, refreshed in place for the next stage right after their row of MFMAs;
  // P fragment 7 and all Q fragments are double-buffered, so the last LDS operation of a step is issued after row 6 and
  // row 7 (128 MFMA clocks) covers its latency in front of the barrier.
  bf16x8 pf[7], pa[1], pb[1], qfa[8], qfb[8];
  u32x4 ra[8], rb[8];
  // one step: MFMAs of stage st (qc, pf, pc) | fragments of stage st+1 from `rs` | transpose registers r -> slot `ws`
  auto step = [&](const bf16x8* qc, bf16x8* qn, const bf16x8* pc, bf16x8* pn, const char* rs, const u32x4* r, char* ws) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qc[kt], nt < 7 ? pf[nt] : pc[0], acc[nt][kt], 0, 0, 0);
      if (nt < 4) {
        qn[2 * nt] = rfrag(rs, qrd[2 * nt], 2 * nt);
        qn[2 * nt + 1] = rfrag(rs, qrd[2 * nt + 1], 2 * nt + 1);
        pf[nt] = rfrag(rs, prd[nt], nt);
        twrite(r, nt, ws);
      } else if (nt < 6) {
        pf[nt] = rfrag(rs, prd[nt], nt);
        if (nt == 4) pn[0] = rfrag(rs, prd[7], 7);
        twrite(r, 2 * nt - 4, ws);
        twrite(r, 2 * nt - 3, ws);
      } else if (nt == 6) {
        pf[6] = rfrag(rs, prd[6], 6);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 LDS reads
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 perms
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 LDS write
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  };
  auto colsum = [&](const bf16x8* pc) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const bf16x8 f = nt < 7 ? pf[nt] : pc[0];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        cacc[nt] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){f[2 * d], f[2 * d + 1]}, ones2, cacc[nt], false);
    }
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  char* slot0 = smem;
  char* slot1 = smem + STAGE;
  gload(ra, 0);
  gload(rb, 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) twrite(ra, j, slot0);
  gload(ra, 2);
  lds_barrier();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) pf[t] = rfrag(slot0, prd[t], t);
    else pa[0] = rfrag(slot0, prd[t], t);
    qfa[t] = rfrag(slot0, qrd[t], t);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) twrite(rb, j, slot1);
  gload(rb, 3);
  // invariant at the top of step st (even): slot (st+1)%2 holds stage st+1 (written during step st-1), ra holds stage
  // st+2, rb stage st+3 (both possibly still in flight), pf/qfa hold the fragments of stage st.
  for (int st = 0; st < nsteps; st += 2) {
    if (do_csum) colsum(pa);
    lds_barrier();
    step(qfa, qfb, pa, pb, slot1, ra, slot0);
    gload(ra, st + 4);
    if (do_csum) colsum(pb);
    lds_barrier();
    step(qfb, qfa, pb, pa, slot0, rb, slot1);
    gload(rb, st + 5);
  }

  float* part = p.part + (long)s * p.N * p.K;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int n = n0 + wn * 128 + nt * 16 + i;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const int k = k0 + wk * 128 + kt * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(part + (long)n * p.K + k) = acc[nt][kt];
    }
  }
  if (do_csum) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v = cacc[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = v;
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
  const int wn = wave & 1, wk = wave >> 1;                 // 2 x 4 waves: 128 n x 64 k each
  const int s = pair / p.tiles_nk;
  const int rem = pair - s * p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  // N and K are multiples of 128, not necessarily of 256: the last tile along either may be half a tile.  Its missing columns
  // are staged as zeros and the two (n) or four (k) waves that own them skip their stores.
  const bool nfull = n0 + 256 <= p.N, kfull = k0 + 256 <= p.K;
  const bool n_ok = nfull || wn == 0, k_ok = kfull || wk < 2;
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
  const unsigned loff = (unsigned)(8 * g * ld2 + cg * 16);
  const int wr = (isq ? OPB : 0) + g * 4096 + (cg >> 1) * 256 + (cg & 1) * 128 + ((cg & 7) << 4);
  const bool colok = (isq ? kfull : nfull) || cg < 16;        // this lane's 8 columns exist
  const bool tile_full = isq ? kfull : nfull;                 // wave-uniform (waves 0-3 stage P, 4-7 stage Q)
  auto gload = [&](u32x4* r, int st) {
    if ((st + 1) * TS <= rows && tile_full) {
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
  };
  auto twrite = [&](const u32x4* r, int j, char* slot) {   // column j of the lane's 8: gather its 8 m, store 16 B
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
  auto rfrag = [&](const char* slot, int off, int h) { return *reinterpret_cast<const bf16x8*>(slot + off + h * 16384); };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  bf16x2 ones2;
  ones2[0] = (bf16)1.0f; ones2[1] = (bf16)1.0f;
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  u32x4 ra[8];
  // one stage: two K = 32 MFMA steps from slot `rs`; the 8 transposed columns of register set r go to slot `ws`
  auto step = [&](const char* rs, const u32x4* r, char* ws) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 qf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t] = rfrag(rs + qbase, frd[t], h);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 pf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) pf[t] = rfrag(rs + pbase + half * 1024, frd[t], h);
        if (do_csum) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int d = 0; d < 4; ++d)
              cacc[4 * half + nt] =
                  __builtin_amdgcn_fdot2_f32_bf16((bf16x2){pf[nt][2 * d], pf[nt][2 * d + 1]}, ones2, cacc[4 * half + nt], false);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[4 * half + nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kt], pf[nt], acc[4 * half + nt][kt], 0, 0, 0);
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

constexpr int RT32_TS = 32;
constexpr int RT32_OPB = 4 * 256 * 16;                     // 16 KiB per operand and stage
constexpr int RT32_STAGE = 2 * RT32_OPB;

// One (slice, tile) pair of problem `p`: the body shared by the single-problem and the grouped kernel.
__device__ __forceinline__ void tn_rt32_pair(const GemmTN& p, const int pair, char* smem) {
  constexpr int TS = RT32_TS, OPB = RT32_OPB, STAGE = RT32_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const int s = pair / p.tiles_nk;
  const int rem = pair - s * p.tiles_nk;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int mbeg = s * p.Ms;
  const int mend = min(p.M, mbeg + p.Ms);
  const int rows = mend - mbeg;
  const int nsteps = (((rows + TS - 1) / TS) + 1) & ~1;    // stages, rounded up to even (the surplus one is all zeros)
  if (rows <= 0) {                                         // empty slice: its partial tile must still be zero
    float* part = p.part + (long)s * p.N * p.K;
    const int q = lane >> 4, i = lane & 15;
    for (int nt = 0; nt < 8; ++nt)
      for (int kt = 0; kt < 8; ++kt)
        *reinterpret_cast<f32x4*>(part + (long)(n0 + wn * 128 + nt * 16 + i) * p.K + k0 + wk * 128 + kt * 16 + 4 * q) =
            (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.cpart && tk == 0 && wk == 0 && q == 0)
      for (int t = 0; t < 8; ++t) p.cpart[(long)s * p.N + n0 + wn * 128 + t * 16 + i] = 0.f;
    return;
  }

  // staging role: waves 0,1 bring P rows [16 w, 16 w + 16) of the stage, waves 2,3 the same rows of Q
  const bool isq = wave >= 2;
  const long ld2 = (isq ? p.ldq : p.ldp) * 2;              // row pitch in bytes
  const int rg = lane >> 5, cg = lane & 31;
  const int g = 2 * (wave & 1) + rg;                       // 8-row block of the stage
  const char* ubase = reinterpret_cast<const char*>(isq ? p.Q + k0 : p.P + n0) + (long)mbeg * ld2;
  const unsigned loff = (unsigned)(8 * g * ld2 + cg * 16);
  const int wr = (isq ? OPB : 0) + g * 4096 + (cg >> 1) * 256 + (cg & 1) * 128 + ((cg & 7) << 4);
  auto gload = [&](u32x4* r, int st) {
    if ((st + 1) * TS <= rows) {                           // whole stage (uniform branch; every stage but the last)
      // raw ISA loads: the compiler's own wait for a compiler-visible load here is s_waitcnt vmcnt(6..0) in front of the
      // first perms of the NEXT step, which also drains the set issued one step later (no look-ahead left); issued as
      // asm the two register sets are ordered by `wait_set` below with vmcnt(8): a true two-step look-ahead
      const char* b = ubase + (long)st * TS * ld2;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[e]) : "v"(b + e * ld2 + loff) : "memory");
    } else {                                               // ragged or surplus stage: clamp the row, zero what is outside
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = st * TS + 8 * g + e;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ubase + (long)min(row, rows - 1) * ld2 + cg * 16);
        const unsigned keep = row < rows ? 0xffffffffu : 0u;   // mask, not a branch: keeps the loads unconditional
        r[e] = v & (u32x4){keep, keep, keep, keep};
      }
    }
  };
  auto twrite = [&](const u32x4* r, int j, char* slot) {   // column j of the lane's 8: gather its 8 m, store 16 B
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      o[d] = __builtin_amdgcn_perm(r[2 * d + 1][j >> 1], r[2 * d][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    *reinterpret_cast<u32x4*>(slot + (wr ^ (j << 4))) = o;
  };

  // v_mfma_f32_32x32x16_bf16: 32-cycle MFMAs leave twice the issue slots per MFMA for the fragment reads, perms, LDS
  // writes and global loads that one wave per SIMD has to interleave.  MEASURED (same-process A/B): 204 vs 250 us (wqkv,
  // 873 TFLOP/s), 263 vs 299 (wfc1, 902), 80 vs 94 (wproj) against the 16x16x32 form of the same kernel -> default.  Wave block 128 x 128 = 4 x 4 blocks of 32 x 32.
  // A / B fragment of a block for K = 16 sub-step u: lane (i = lane % 32, kg = lane / 32) holds the 8 m of m-block 2u + kg
  // for column 32 b + i  -> one ds_read_b128 from the same [m/8][col] LDS image (conflict-free: see the layout note above).
  const int i32 = lane & 31, kg = lane >> 5;
  int prd[4], qrd[4];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    const int slot = (i32 & 8) | ((i32 & 7) ^ ((4 * bb + (i32 >> 3)) & 7));
    const int win = bb * 2 + (i32 >> 4);
    prd[bb] = (wn * 8 + win) * 256 + kg * 4096 + (slot << 4);
    qrd[bb] = OPB + (wk * 8 + win) * 256 + kg * 4096 + (slot << 4);
  }
  auto rfrag = [&](const char* slot, int off, int u) { return *reinterpret_cast<const bf16x8*>(slot + off + u * 8192); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a2][b2][e] = 0.f;
  float cacc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  bf16x2 ones2;
  ones2[0] = (bf16)1.0f; ones2[1] = (bf16)1.0f;

  // fragments of the two K = 16 sub-steps of a stage: set A (sub-step 0) and set B (sub-step 1); while sub-step 0 of stage
  // s computes, sub-step 1's fragments are already in registers and the reads of stage s+1 refill the set that just finished
  bf16x8 pa[4], qa[4], pb[4], qb[4];
  u32x4 ra[8], rb[8];
  auto mma = [&](const bf16x8* pf, const bf16x8* qf) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[kb], pf[nb], acc[nb][kb], 0, 0, 0);
  };
  auto colsum = [&](const bf16x8* pf) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        cacc[nb] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){pf[nb][2 * d], pf[nb][2 * d + 1]}, ones2, cacc[nb], false);
  };
  // one stage: 32 MFMAs | 16 fragment reads of the NEXT stage from `rs` | transpose registers r -> slot `ws`
  auto step = [&](const char* rs, const u32x4* r, char* ws) {
    if (do_csum) { colsum(pa); colsum(pb); }
    mma(pa, qa);
#pragma unroll
    for (int t = 0; t < 4; ++t) { pa[t] = rfrag(rs, prd[t], 0); qa[t] = rfrag(rs, qrd[t], 0); }
#pragma unroll
    for (int j = 0; j < 4; ++j) twrite(r, j, ws);
    mma(pb, qb);
#pragma unroll
    for (int t = 0; t < 4; ++t) { pb[t] = rfrag(rs, prd[t], 1); qb[t] = rfrag(rs, qrd[t], 1); }
#pragma unroll
    for (int j = 4; j < 8; ++j) twrite(r, j, ws);
    // 16 MFMAs each half; per MFMA: <= 1 LDS read / 1 perm-group; the LDS writes go with the later MFMAs of a half
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };
  // all 8 registers of the OLDER set have landed (the 8 loads of the newer set may stay in flight); the "+v" operands tie
  // the perms that consume the set to this wait
  auto wait_set = [&](u32x4* r) {
    asm volatile("s_waitcnt vmcnt(8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  char* slot0 = smem;
  char* slot1 = smem + STAGE;
  gload(ra, 0);
  gload(rb, 1);
  wait_set(ra);
#pragma unroll
  for (int j = 0; j < 8; ++j) twrite(ra, j, slot0);
  gload(ra, 2);
  lds_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    pa[t] = rfrag(slot0, prd[t], 0); qa[t] = rfrag(slot0, qrd[t], 0);
    pb[t] = rfrag(slot0, prd[t], 1); qb[t] = rfrag(slot0, qrd[t], 1);
  }
  wait_set(rb);
#pragma unroll
  for (int j = 0; j < 8; ++j) twrite(rb, j, slot1);
  gload(rb, 3);
  for (int st = 0; st < nsteps; st += 2) {
    lds_barrier();
    wait_set(ra);
    step(slot1, ra, slot0);
    gload(ra, st + 4);
    lds_barrier();
    wait_set(rb);
    step(slot0, rb, slot1);
    gload(rb, st + 5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // D layout of 32x32x16: lane (col n = lane % 32, kg): register e holds row k = (e / 4) * 8 + kg * 4 + e % 4
  float* part = p.part + (long)s * p.N * p.K;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + wn * 128 + nb * 32 + i32;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int k = k0 + wk * 128 + kb * 32 + e4 * 8 + kg * 4;
        *reinterpret_cast<f32x4*>(part + (long)n * p.K + k) =
            (f32x4){acc[nb][kb][4 * e4], acc[nb][kb][4 * e4 + 1], acc[nb][kb][4 * e4 + 2], acc[nb][kb][4 * e4 + 3]};
      }
  }
  if (do_csum) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      float v = cacc[nb];
      v += __shfl_xor(v, 32, 64);
      if (kg == 0) p.cpart[(long)s * p.N + n0 + wn * 128 + nb * 32 + i32] = v;
    }
  }
}

__global__ __launch_bounds__(256, 1) void gemm_tn_rt32_kernel(GemmTN p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RT32_STAGE];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int pair = xcd * p.Ms_pairs + jj;                  // (slice, tile) pairs in slice-major order, one chunk per XCD
  if (jj >= p.Ms_pairs || pair >= p.npairs) return;
  tn_rt32_pair(p, pair, smem);
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

__global__ __launch_bounds__(256, 1) void gemm_tn_rt32_grouped_kernel(TnGroup g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RT32_STAGE];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int gp = xcd * g.per_xcd + jj;
  if (jj >= g.per_xcd || gp >= g.total) return;
  int q = 0;
#pragma unroll
  for (int t = 1; t < TN_GROUP_MAX; ++t)
    if (t < g.nprob && gp >= g.first[t]) q = t;
  const GemmTN p = g.prob[q];
  tn_rt32_pair(p, gp - g.first[q], smem);
}

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

// out[n][k] = beta*out + sum_s part[s][n][k];  bias_out[n] = beta*bias_out + sum_s cpart[s][n]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cpart,
                                                        int splits, long NK, int N, float beta,
                                                        float* __restrict__ out, float* __restrict__ bias_out) {
  const long idx4 = (long)blockIdx.x * 256 + threadIdx.x;
  const long n4 = NK >> 2;
  if (idx4 < n4) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[idx4];
    for (int s = 1; s < splits; ++s) {
      const f32x4 b = reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
      a += b;
    }
    if (beta != 0.f) a += beta * reinterpret_cast<f32x4*>(out)[idx4];
    reinterpret_cast<f32x4*>(out)[idx4] = a;
  } else if (bias_out && idx4 - n4 < N) {
    const int n = (int)(idx4 - n4);
    float a = 0.f;
    for (int s = 0; s < splits; ++s) a += cpart[(long)s * N + n];
    if (beta != 0.f) a += beta * bias_out[n];
    bias_out[n] = a;
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
  const long idx4 = (long)((int)blockIdx.x - g.first[q]) * 256 + threadIdx.x;
  const long n4 = NK >> 2;
  if (idx4 < n4) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[idx4];
    for (int s = 1; s < g.splits; ++s) a += reinterpret_cast<const f32x4*>(part + (long)s * NK)[idx4];
    if (beta != 0.f) a += beta * reinterpret_cast<f32x4*>(out)[idx4];
    reinterpret_cast<f32x4*>(out)[idx4] = a;
  } else if (bias_out && idx4 - n4 < N) {
    const int n = (int)(idx4 - n4);
    float a = 0.f;
    for (int s = 0; s < g.splits; ++s) a += cpart[(long)s * N + n];
    if (beta != 0.f) a += beta * bias_out[n];
    bias_out[n] = a;
  }
}

// 0 = heuristic (register-transposed 256x256 kernel when N and K are multiples of 256, else 128x128 register-staged),
// 1 = 128x128 register-staged, 2 = 128x128 LDS-DMA staged, 3 = 256x256 / 16 waves, 4 = 256x256 / 8 waves,
// 5 = 256x256 LDS-DMA ring, 6 = 256x256 register-transposed 4 waves with 16x16x32 MFMAs, 7 = the same with 32x32x16 MFMAs,
// 8 = register-transposed 8 waves x 128x64 (= what the heuristic picks; benchmark / test knob)
int g_tn_tile = 0;

bool tn_use_rt(int64_t N, int64_t K) {
  if ((g_tn_tile == 0 || g_tn_tile == 8) && (N % 128 == 0) && (K % 128 == 0) && N * K >= 256 * 256)
    return true;     // the 8-wave kernel stages half tiles (N or K = 128 mod 256) with zero columns
  return (g_tn_tile == 6 || g_tn_tile == 7) && (N % 256 == 0) && (K % 256 == 0);
}

}  // namespace

extern "C" int pvrl_debug_set_gemm_tn_tile(int tile) { g_tn_tile = tile; return PVRL_OK; }

extern "C" int64_t pvrl_gemm_tn_plan_splits(int64_t M, int64_t N, int64_t K) {
  if (N <= 0 || K <= 0 || (N % 128) || (K % 128)) return PVRL_EINVAL;
  if (tn_use_rt(N, K)) {
    // one workgroup per CU and ONE round: as many (slice, tile) pairs as fit the 256 CUs, slices of >= 64 rows
    const int64_t tiles = cdiv(N, 256) * cdiv(K, 256);
    int64_t s = 256 / tiles;
    const int64_t smax = M / 64;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
  }
  // 128x128 kernels: a multiple of 8 (slice s lives on XCD s % 8), enough (n, k) tiles x slices to fill
  // 8 XCDs x 64 resident workgroups about twice, but at least ~256 rows per slice
  const int64_t tiles = (N / 128) * (K / 128);
  int64_t per_xcd = cdiv(128, tiles);
  if (per_xcd < 1) per_xcd = 1;
  int64_t s = 8 * per_xcd;
  while (s > 8 && M / s < 256) s -= 8;
  return s;
}

extern "C" int64_t pvrl_gemm_tn_workspace_bytes(int64_t N, int64_t K, int64_t splits) {
  return splits * (N * K + N) * (int64_t)sizeof(float) + 256;   // + a zero page for out-of-range rows
}

extern "C" int pvrl_gemm_tn_bf16(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N,
                                 int64_t K, int64_t splits, float beta, float* dW, float* dbias, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (!P || !Q || !dW || !workspace || N <= 0 || K <= 0 || (N % 128) || (K % 128) || splits < 1 || M < 0)
    return PVRL_EINVAL;
  const bool use_rt = tn_use_rt(N, K);
  if (!use_rt && (splits < 8 || (splits % 8))) return PVRL_EINVAL;   // slice s lives on XCD s % 8 in those kernels
  if ((ldp % 8) || (ldq % 8) || ((uintptr_t)P % 16) || ((uintptr_t)Q % 16)) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_gemm_tn_workspace_bytes(N, K, splits)) return PVRL_EINVAL;
  GemmTN p;
  p.P = (const bf16*)P; p.ldp = ldp; p.Q = (const bf16*)Q; p.ldq = ldq;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  int ms = cdiv(M > 0 ? M : 1, splits);
  p.Ms = cdiv(ms, TM) * TM;
  p.part = (float*)workspace;
  p.cpart = dbias ? p.part + splits * N * K : nullptr;
  hipStream_t s = (hipStream_t)stream;
  char* zp = (char*)workspace + splits * (N * K + N) * (int64_t)sizeof(float);
  p.zero_page = (const bf16*)zp;
  // the 256x256 / 16-wave instantiation is register-starved at 128 VGPRs (spills; 2-3x slower on MI355X) and is
  // only reachable through the benchmark knob
  const bool big = g_tn_tile == 3 && (N % 256 == 0) && (K % 256 == 0);
  if (use_rt) {
    p.tiles_k = (int)cdiv(K, 256);
    p.tiles_nk = (int)cdiv(N, 256) * p.tiles_k;
    p.npairs = (int)splits * p.tiles_nk;
    p.Ms_pairs = cdiv(p.npairs, 8);
    if (g_tn_tile == 0 || g_tn_tile == 8) hipLaunchKernelGGL(gemm_tn_rt8_kernel, dim3((unsigned)(8 * p.Ms_pairs)), dim3(512), 0, s, p);
    else if (g_tn_tile != 6) hipLaunchKernelGGL(gemm_tn_rt32_kernel, dim3((unsigned)(8 * p.Ms_pairs)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_tn_rt_kernel, dim3((unsigned)(8 * p.Ms_pairs)), dim3(256), 0, s, p);
  } else if (g_tn_tile == 5 && (N % 256 == 0) && (K % 256 == 0) && (M % 64 == 0)) {
    p.tiles_k = (int)(K / 256);
    p.tiles_nk = (int)(N / 256) * p.tiles_k;
    hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3((unsigned)(splits * p.tiles_nk)), dim3(256), 0, s, p);
  } else if (g_tn_tile == 4 && (N % 256 == 0) && (K % 256 == 0)) {
    p.tiles_k = (int)(K / 256);
    p.tiles_nk = (int)(N / 256) * p.tiles_k;
    hipLaunchKernelGGL(gemm_tn_w128_kernel, dim3((unsigned)(splits * p.tiles_nk)), dim3(512), 0, s, p);
  } else if (big) {
    p.tiles_k = (int)(K / 256);
    p.tiles_nk = (int)(N / 256) * p.tiles_k;
    hipLaunchKernelGGL((gemm_tn_kernel<4, 4>), dim3((unsigned)(splits * p.tiles_nk)), dim3(1024), 0, s, p);
  } else {
    p.tiles_k = (int)(K / 128);
    p.tiles_nk = (int)(N / 128) * p.tiles_k;
    // measured on MI355X (tools/bench_kernels.py, same process A/B): register staging 505-585 TFLOP/s, LDS-DMA
    // staging 485-550: the kernel is bound by the half-rate ds_read_b64_tr_b16 stream (32 per wave and stage),
    // not by the staging path, so the register-staged form stays the default; knob 2 selects the LDS-DMA form.
    if (g_tn_tile == 2) {
      if (hipMemsetAsync(zp, 0, 256, s) != hipSuccess) return PVRL_EHIP;   // LDS-DMA source for out-of-range rows
      hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3((unsigned)(splits * p.tiles_nk)), dim3(256), 0, s, p);
    } else
      hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), dim3((unsigned)(splits * p.tiles_nk)), dim3(256), 0, s, p);
  }
  PVRL_LAUNCH_CHECK();
  const long NK = N * K;
  const long nthreads = (NK >> 2) + (dbias ? N : 0);
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)cdiv(nthreads, 256)), dim3(256), 0, s, p.part, p.cpart,
                     (int)splits, NK, (int)N, beta, dW, dbias);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// grouped weight gradients
// ---------------------------------------------------------------------------------------------------------
namespace {
bool tn_group_ok(int nprob, const pvrl_tn_problem* pr) {
  if (nprob < 1 || nprob > TN_GROUP_MAX || !pr) return false;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = pr[i];
    if (!q.P || !q.Q || !q.dW || q.M < 1 || q.N <= 0 || q.K <= 0 || (q.N % 128) || (q.K % 128)) return false;
    if (((q.N % 256) || (q.K % 256)) && g_tn_tile != 0 && g_tn_tile != 8) return false;   // half tiles: 8-wave kernel only
    if ((q.ldp % 8) || (q.ldq % 8) || ((uintptr_t)q.P % 16) || ((uintptr_t)q.Q % 16)) return false;
  }
  return true;
}
int64_t tn_group_tiles(int nprob, const pvrl_tn_problem* pr) {
  int64_t t = 0;
  for (int i = 0; i < nprob; ++i) t += cdiv(pr[i].N, 256) * cdiv(pr[i].K, 256);
  return t;
}
}  // namespace

extern "C" int64_t pvrl_gemm_tn_grouped_plan_splits(int nprob, const pvrl_tn_problem* problems) {
  if (!tn_group_ok(nprob, problems)) return PVRL_EINVAL;
  const int64_t T = tn_group_tiles(nprob, problems);
  int64_t smax = 32;
  for (int i = 0; i < nprob; ++i) smax = std::min<int64_t>(smax, std::max<int64_t>(1, problems[i].M / 64));
  // the smallest slice count whose T*s equal work items fill whole rounds of the 256 CUs to >= 97 %, else the best one
  int64_t best = 1;
  double best_eff = 0.0;
  for (int64_t s = 1; s <= smax; ++s) {
    const int64_t items = T * s;
    const double eff = (double)items / (double)(256 * cdiv(items, 256));
    if (eff >= 0.97) return s;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
  }
  return best;
}

extern "C" int64_t pvrl_gemm_tn_grouped_workspace_bytes(int nprob, const pvrl_tn_problem* problems, int64_t splits) {
  if (!tn_group_ok(nprob, problems) || splits < 1) return PVRL_EINVAL;
  int64_t b = 0;
  for (int i = 0; i < nprob; ++i) b += splits * (problems[i].N * problems[i].K + problems[i].N) * (int64_t)sizeof(float);
  return b;
}

extern "C" int pvrl_gemm_tn_grouped_bf16(int nprob, const pvrl_tn_problem* problems, int64_t splits, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (!tn_group_ok(nprob, problems) || splits < 1 || !workspace) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_gemm_tn_grouped_workspace_bytes(nprob, problems, splits)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  TnGroup g = {};
  g.nprob = nprob;
  float* w = (float*)workspace;
  int first = 0;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = problems[i];
    GemmTN& p = g.prob[i];
    p.P = (const bf16*)q.P; p.ldp = q.ldp; p.Q = (const bf16*)q.Q; p.ldq = q.ldq;
    p.M = (int)q.M; p.N = (int)q.N; p.K = (int)q.K;
    p.Ms = cdiv(cdiv(q.M, splits), TM) * TM;
    p.part = w;
    w += splits * q.N * q.K;
    p.cpart = q.dbias ? w : nullptr;
    w += splits * q.N;
    p.zero_page = nullptr;
    p.tiles_k = (int)cdiv(q.K, 256);
    p.tiles_nk = (int)cdiv(q.N, 256) * p.tiles_k;
    p.npairs = (int)splits * p.tiles_nk;
    p.Ms_pairs = 0;
    g.first[i] = first;
    first += p.npairs;
  }
  g.first[nprob] = first;
  g.total = first;
  g.per_xcd = cdiv(first, 8);
  if (g_tn_tile == 0 || g_tn_tile == 8) hipLaunchKernelGGL(gemm_tn_rt8_grouped_kernel, dim3((unsigned)(8 * g.per_xcd)), dim3(512), 0, s, g);
  else hipLaunchKernelGGL(gemm_tn_rt32_grouped_kernel, dim3((unsigned)(8 * g.per_xcd)), dim3(256), 0, s, g);
  PVRL_LAUNCH_CHECK();
  static_assert(TN_RED_MAX >= TN_GROUP_MAX, "reduce table too small");
  TnReduceGroup r = {};
  r.nprob = nprob; r.splits = (int)splits;
  int blocks = 0;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = problems[i];
    r.part[i] = g.prob[i].part; r.cpart[i] = g.prob[i].cpart; r.out[i] = q.dW; r.bias_out[i] = q.dbias;
    r.NK[i] = q.N * q.K; r.N[i] = (int)q.N; r.beta[i] = q.beta;
    r.first[i] = blocks;
    blocks += (int)cdiv((r.NK[i] >> 2) + (q.dbias ? q.N : 0), 256);
  }
  r.first[nprob] = blocks;
  hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, r);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
