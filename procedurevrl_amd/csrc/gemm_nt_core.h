// bf16 MFMA GEMM, "NT" form:  C[M,N] = epilogue( A[M,K] . W[N,K]^T ), fp32 accumulate.
//
// Replaces the nn.Linear / F.linear calls of the reference hot path
// (lib/models/vit.py:54-60 Mlp, :75-92 Attention qkv/proj, :133 temporal_fc,
// :174-180 PatchEmbed conv-as-GEMM) and, fed with the transposed weight copy,
// their data-gradients.  One kernel family, fused epilogues:
//   bias, per-row scale (DropPath), exact-erf GELU / QuickGELU (+ pre-activation
//   kept for backward), fp32 residual add, GELU-derivative for the MLP backward.
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles), BK = 64, operands staged with
// 16-byte global_load_lds (LDS-DMA) into a double-buffered 2 x 32 KiB LDS ring.
// LDS tiles are [128 rows][64 bf16] (128-byte rows); the 16-byte chunk index is XOR-
// swizzled on the *global source* side (LDS-DMA writes lane-linear) and on the
// ds_read_b128 side with the same involution so that every ds_read_b128 lane group
// hits 16 distinct 16-byte slots.  Operands are swapped in the MFMA (a = W rows,
// b = A rows) and the W rows feeding tile nt are permuted (n = 16*q + 4*nt + r) so a
// lane ends up owning 16 consecutive output columns of one output row: the
// epilogue streams 16-/32-/64-byte contiguous pieces per lane.
#pragma once
#include <algorithm>
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct GemmNT {
  const op_t* A; long lda;
  const op_t* W; long ldw;
  int M, N, K;
  const float* bias;      // [N] or null
  const float* bias2;     // [N] or null: added AFTER the row scale (fp32-residual epilogue only)
  const float* rowscale;  // [M] or null
  const void* aux;        // fp32 residual [*, aux_ld] or bf16 pre-activation [M, aux_ld]
  long aux_ld; int aux_rowmod;
  void* out0; long ld0;
  void* out1; long ld1;
  int tiles_m, tiles_n, nwg;
  int cus, tails;   // CUs per XCD; 1 = cut the tiles of the ragged last round into sub-tiles (256x256 kernel, nt_tail_plan)
  int m_off;   // global row of local row 0 (a launch may cover a row range of the logical GEMM)
  int gm;      // rasterisation group height in tiles
};

constexpr int BK = 64;
// rasterisation group height (tile rows per group).  The 32 workgroups an XCD runs at a time cover gm A panels x 32 / gm W tiles: the
// distinct lines they ask the XCD's L2 for per K-tile go with gm + 32 / gm, least at gm = sqrt(32) = 5.7.  Measured in the step (round 4,
// two-phase persistent kernel, three interleaved pairs each against "4 for N >= 2048, else 2", the choice of round 3's per-kernel sweep
// of 2 / 3 / 4 / 6 / 8): gm 2 +0.3 %, 3 +-0, 5 +0.9 %, 6 +0.8 ... +1.0 % (twice), 8 +0.3 %, 12 +0.4 % -> 6 for every shape (with three or
// fewer tile columns the group height does not change what runs together).  The gain is in the L2, not in HBM bytes: FETCH_SIZE per
// launch went UP 8-15 % on the wide shapes (profiles/r4_pmc_hbm_mfma.csv) while the kernels got 1-3 % shorter.
#ifndef PVRL_NT_GM
#define PVRL_NT_GM 0
#endif
constexpr int NT_GM = PVRL_NT_GM;       // 0 = the default below, otherwise forced (probe builds)
static inline int nt_gm_for(int tiles_n) { (void)tiles_n; return NT_GM ? NT_GM : 6; }

__device__ __forceinline__ int swz_x(int row) { return (row >> 1) & 7; }
// W rows are read in a permuted order so that the lanes of one epilogue store instruction write contiguous bytes:
//   tile nt = 2c + h holds column 32c + 8q + 4h + r -> a lane owns 8 consecutive columns (tiles 2c, 2c+1) and the 4 lanes
//   q of a row cover 32 consecutive columns: 64 B of bf16 per store instruction, or a whole 128-byte line of fp32 in
//   two back-to-back 16-byte stores per lane (the earlier natural order for fp32 wrote 64-byte half lines).
// Each order has its own chunk swizzle making ds_read_b128 conflict-free (rows that a lane group reads together
// must land on distinct 16-byte slots of the 256-byte bank row).
template <bool F32OUT> __device__ __forceinline__ int w_row(int nt, int i) {
  return 32 * (nt >> 1) + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3);
}
template <bool F32OUT> __device__ __forceinline__ int swz_w(int row) {
  return ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
}

// ---- cache policy of the epilogue's memory traffic (gfx950 `aux` bits of the raw buffer instructions: 1 = sc0, 2 = nt,
// 16 = sc1).  An epilogue GEMM streams hundreds of MB exactly once -- the fp32 residual / stored pre-activation in, the
// outputs out (consumed by a LATER kernel) -- through a 4 MiB XCD L2 that should be holding the W tiles and A panels the
// main loop re-reads: with default-policy stores every output line is kept in L2 and evicts them (round 2: the GELU GEMM
// fetched 480 MB per launch against 82 MB of operands).  `sc1` stores are written through and dropped (MI355X guide,
// "stores of each flavour"); `nt` marks the read-once loads as streaming.  Swept on MI355X: profiles/r3_nt_cache_policy.txt.
#ifndef PVRL_NT_ST_AUX
#define PVRL_NT_ST_AUX 0
#endif
#ifndef PVRL_NT_LD_AUX
#define PVRL_NT_LD_AUX 0
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// descriptor over `rows` rows of a row-major matrix starting at `base` (all arguments wave-uniform: kernel arguments and
// blockIdx-derived tile origins).  Offsets past the last row are dropped (stores) / read as zero (loads) by the hardware
// bounds check: the ragged last M-tile needs no per-row guard.
__device__ __forceinline__ rsrc_t tile_rsrc(const void* base, long row0, long ld, int elem, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + row0 * ld * elem), 0, (int)(rows * ld * elem), 0x00020000);
}
// Probe builds only (tools/probe/nt8_ab.py epi; results are garbage): bit 0 = the epilogue's stores are dropped (values kept alive),
// bit 1 = the activation math is replaced by the identity, bit 2 = the epilogue's loads (residual / pre-activation) are dropped
#ifndef PVRL_NT_EPI_ABLATE
#define PVRL_NT_EPI_ABLATE 0
#endif
template <int AUX> __device__ __forceinline__ void bst16(rsrc_t r, unsigned off, f32x4 v) {
  if (PVRL_NT_EPI_ABLATE & 1) { asm volatile("" :: "v"(v), "v"(off)); return; }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX);
}
template <int AUX> __device__ __forceinline__ void bst16(rsrc_t r, unsigned off, opx8 v) {
  if (PVRL_NT_EPI_ABLATE & 1) { asm volatile("" :: "v"(v), "v"(off)); return; }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX);
}
template <int AUX, typename T> __device__ __forceinline__ T bld16(rsrc_t r, unsigned off) {
  if (PVRL_NT_EPI_ABLATE & 4) { u32x4 z = {off, 0x3c003c00u, 0x3c003c00u, off}; asm volatile("" : "+v"(z)); return __builtin_bit_cast(T, z); }
  return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}

// The epilogue of ONE 64-row x 64-column accumulator block acc[mt][nt] (4 x 4 MFMA tiles of 16 x 16, operands swapped: a lane owns
// row `rb + 16 mt + (lane & 15)` of the tile and, per column group c = nt >> 1, the 8 consecutive columns `cb[c] + 8 (lane >> 4) + ...`).
//   rb      first row of the block inside the tile
//   c0, c1  first column (inside the tile) of the block's two 32-column groups: wn * 64, wn * 64 + 32 for the 16-wave kernel;
//           32 wn and 128 + 32 wn for the 8-wave kernel, whose wave columns are split over the two W half-tiles
//   BATCH   row tiles whose residual / stored pre-activation loads go out together (one memory round trip per batch)
//   TAB     PVRL_EPI_RESID_16 only: aux is the fp32 row-modulo table of the embedding prologue, not 16-bit residual rows (a template
//           parameter chosen once per tile by nt_epilogue_sel: the two forms share no load, and a load behind a run-time branch is
//           followed by a full wait)
template <int EPI, int BATCH = 2, bool TAB = false>
__device__ __forceinline__ void nt_epilogue_at(const GemmNT& p, f32x4 (&acc)[4][4], int m0, int n0, int rb, int c0, int c1, int lane) {
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int ST = PVRL_NT_ST_AUX, LD = PVRL_NT_LD_AUX;
  static_assert(BATCH == 1 || BATCH == 2 || BATCH == 4, "row tiles per load batch");
  const int q = lane >> 4, i = lane & 15;
  const int ncol[2] = {n0 + c0 + 8 * q, n0 + c1 + 8 * q};   // this lane's first column of each 32-column group
  const int rows = __builtin_amdgcn_readfirstlane(max(0, min(1024, p.M - m0)));    // rows of the matrices from the tile's first row on (any bound >= the tile height that keeps the byte count in 31 bits)
  if constexpr (F32OUT) {
    // lane holds, for c = 0,1: columns ncol[c] + (0..7)  (tile 2c -> +0..3, tile 2c+1 -> +4..7)
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bv[nt] = !p.bias ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.bias + ncol[nt >> 1] + 4 * (nt & 1));
    // Per-row factors (DropPath) and the second bias are fetched HERE, before the first store: a load issued between the
    // stores is followed by s_waitcnt vmcnt(0), and on gfx9 that also waits for every earlier store to be acknowledged by
    // L2 -- four (rowscale) to sixteen (bias2) serial store round trips per thread.  bias2 has no 16 spare registers in the
    // 128-VGPR budget of the 16-wave tile: each lane keeps ONE column of the block's 64 and the others come by ds_bpermute.
    float rs4[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rs4[mt] = p.rowscale ? p.rowscale[min(m0 + rb + mt * 16 + i, p.M - 1)] : 1.f;
    float b2lane = 0.f;
    if constexpr (EPI == PVRL_EPI_RESID_F32) {
      if (p.bias2) {
        b2lane = p.bias2[n0 + (lane < 32 ? c0 + lane : c1 + lane - 32)];
        if (!p.rowscale) {                 // no row factor: rs * (acc + bias) + bias2 = acc + (bias + bias2)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nt][e] += __shfl(b2lane, 32 * (nt >> 1) + 8 * q + 4 * (nt & 1) + e, 64);
        }
      }
    }
    const bool b2_late = EPI == PVRL_EPI_RESID_F32 && p.bias2 && p.rowscale;
    const rsrc_t ro = tile_rsrc(p.out0, m0, p.ld0, 4, rows);
    // the residual: rows of the tile (streamed once: LD policy), or rows modulo aux_rowmod of a small table every clip
    // re-reads (pos / time embedding prologue: default policy)
    const bool tab = EPI == PVRL_EPI_RESID_F32 && p.aux_rowmod != 0;
    const rsrc_t ra = EPI == PVRL_EPI_RESID_F32 ? (tab ? tile_rsrc(p.aux, 0, p.aux_ld, 4, p.aux_rowmod) : tile_rsrc(p.aux, m0, p.aux_ld, 4, rows)) : ro;
    // (Measured and rejected: finishing all 16 tiles in place and storing last, so that the second half's residual loads are
    //  not queued behind the first half's stores: -0.4 % on the step -- the early stores start the write traffic sooner.)
    // The residual loads of BATCH row tiles (16 registers each, freed by the MFMA fragments) go out together: 4 / BATCH memory
    // round trips per wave instead of four dependent ones.  With every CU in its epilogue at the same time the loaded latency of
    // a round trip is microseconds.
#pragma unroll
    for (int bt = 0; bt < 4 / BATCH; ++bt) {
      f32x4 rv[BATCH][4];
      if constexpr (EPI == PVRL_EPI_RESID_F32) {
#pragma unroll
        for (int h = 0; h < BATCH; ++h) {
          const int ml = rb + (BATCH * bt + h) * 16 + i;                        // row inside the tile
          const int mr = tab ? ((min(m0 + ml, p.M - 1) + p.m_off) % p.aux_rowmod) : ml;
          const unsigned ab = (unsigned)mr * (unsigned)p.aux_ld * 4u;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const unsigned off = ab + (unsigned)(ncol[nt >> 1] + 4 * (nt & 1)) * 4u;
            rv[h][nt] = tab ? bld16<0, f32x4>(ra, off) : bld16<LD, f32x4>(ra, off);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < BATCH; ++h) {
        const int mt = BATCH * bt + h;
        const float rs = rs4[mt];
        const unsigned ob = (unsigned)(rb + mt * 16 + i) * (unsigned)p.ld0 * 4u;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = rs * (acc[mt][nt][e] + bv[nt][e]);
          if constexpr (EPI == PVRL_EPI_RESID_F32) {
            ov += rv[h][nt];
            if (b2_late) {
#pragma unroll
              for (int e = 0; e < 4; ++e) ov[e] += __shfl(b2lane, 32 * (nt >> 1) + 8 * q + 4 * (nt & 1) + e, 64);
            }
          }
          bst16<ST>(ro, ob + (unsigned)(ncol[nt >> 1] + 4 * (nt & 1)) * 4u, ov);   // rows past M: dropped by the bounds check
        }
      }
    }
  } else {
    // lane holds, for c = 0,1: columns ncol[c] + (0..7)  (tile 2c -> +0..3, tile 2c+1 -> +4..7)
    float bv[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[c][e] = !p.bias ? 0.f : p.bias[ncol[c] + e];
    float rs4[4];                      // before the first store (see above)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rs4[mt] = p.rowscale ? p.rowscale[min(m0 + rb + mt * 16 + i, p.M - 1)] : 1.f;
    constexpr bool TWO = EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU;
    constexpr bool DACT = EPI == PVRL_EPI_DGELU || EPI == PVRL_EPI_DQGELU;
    // PVRL_EPI_RESID_16 (round 6): the residual add on the 16-bit patch rows of the split residual stream -- out = aux + rs * (acc +
    // bias) + bias2, aux and out 16-bit: 16 B per lane and column group each way where the fp32-residual epilogue moves 2 x 32 B
    // (231 MB less per 50k-row launch).  aux_rowmod != 0: aux is the fp32 pos / time table of the embedding prologue instead (one
    // launch per step: its loads go out row tile by row tile).
    constexpr bool RES16 = EPI == PVRL_EPI_RESID_16;
    constexpr bool tab = RES16 && TAB;
    float b2lane = 0.f;      // the second bias: each lane keeps ONE column of the block's 64, the others come by ds_bpermute (see above)
    if constexpr (RES16) {
      if (p.bias2) {
        b2lane = p.bias2[n0 + (lane < 32 ? c0 + lane : c1 + lane - 32)];
        if (!p.rowscale) {                 // no row factor: rs * (acc + bias) + bias2 = acc + (bias + bias2)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[c][e] += __shfl(b2lane, 32 * c + 8 * q + e, 64);
        }
      }
    }
    const bool b2_late = RES16 && p.bias2 && p.rowscale;
    const rsrc_t r0 = tile_rsrc(p.out0, m0, p.ld0, 2, rows);
    const rsrc_t r1 = TWO ? tile_rsrc(p.out1, m0, p.ld1, 2, rows) : r0;
    const rsrc_t ra = (DACT || RES16) ? (tab ? tile_rsrc(p.aux, 0, p.aux_ld, 4, p.aux_rowmod) : tile_rsrc(p.aux, m0, p.aux_ld, 2, rows)) : r0;
#pragma unroll
    for (int bt = 0; bt < 4 / BATCH; ++bt) {
      opx8 uv[BATCH][2];
      if constexpr ((DACT || RES16) && !tab) {   // the stored pre-activations / the 16-bit residual rows of BATCH row tiles: one round trip
#pragma unroll
        for (int h = 0; h < BATCH; ++h) {
          const unsigned ab = (unsigned)(rb + (BATCH * bt + h) * 16 + i) * (unsigned)p.aux_ld * 2u;
#pragma unroll
          for (int c = 0; c < 2; ++c) uv[h][c] = bld16<LD, opx8>(ra, ab + (unsigned)ncol[c] * 2u);
        }
      }
#pragma unroll
      for (int h = 0; h < BATCH; ++h) {
        const int mt = BATCH * bt + h;
        const unsigned ml = (unsigned)(rb + mt * 16 + i);
        const float rs = rs4[mt];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[mt][2 * c][e] + bv[c][e];
            v[4 + e] = acc[mt][2 * c + 1][e] + bv[c][4 + e];
          }
          const unsigned o0off = ml * (unsigned)p.ld0 * 2u + (unsigned)ncol[c] * 2u;
          if constexpr (EPI == PVRL_EPI_BF16) {
            opx8 o0;
#pragma unroll
            for (int e = 0; e < 8; ++e) o0[e] = (op_t)(rs * v[e]);
            bst16<ST>(r0, o0off, o0);
          } else if constexpr (TWO) {
            opx8 u0, g0;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              u0[e] = (op_t)v[e];
              u0[e + 1] = (op_t)v[e + 1];
              if constexpr (EPI == PVRL_EPI_GELU) {
                const f32x2_t gg = (PVRL_NT_EPI_ABLATE & 2) ? (f32x2_t){v[e] + 1.f, v[e + 1] + 1.f} : gelu_erf2((f32x2_t){v[e], v[e + 1]});
                g0[e] = (op_t)gg[0];
                g0[e + 1] = (op_t)gg[1];
              } else {
                g0[e] = (op_t)quick_gelu(v[e]);
                g0[e + 1] = (op_t)quick_gelu(v[e + 1]);
              }
            }
            bst16<ST>(r0, o0off, u0);
            bst16<ST>(r1, ml * (unsigned)p.ld1 * 2u + (unsigned)ncol[c] * 2u, g0);
          } else if constexpr (RES16) {
            float a[8];
            if constexpr (tab) {
              const int mr = (min(m0 + (int)ml, p.M - 1) + p.m_off) % p.aux_rowmod;
              const unsigned ab = (unsigned)mr * (unsigned)p.aux_ld * 4u + (unsigned)ncol[c] * 4u;
              const f32x4 t0 = bld16<0, f32x4>(ra, ab), t1 = bld16<0, f32x4>(ra, ab + 16u);
#pragma unroll
              for (int e = 0; e < 4; ++e) { a[e] = t0[e]; a[4 + e] = t1[e]; }
            } else {
              const opx8 ua = uv[h][c];
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = (float)ua[e];
            }
            if (b2_late) {          // (one branch, its eight ds_bpermute in flight together)
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] += __shfl(b2lane, 32 * c + 8 * q + e, 64);
            }
            opx8 o0;
#pragma unroll
            for (int e = 0; e < 8; ++e) o0[e] = (op_t)(rs * v[e] + a[e]);
            bst16<ST>(r0, o0off, o0);
          } else {  // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
            const opx8 ua = uv[h][c];
            opx8 o0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float d = (PVRL_NT_EPI_ABLATE & 2) ? (float)ua[e] : EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
              o0[e] = (op_t)(rs * v[e] * d);
            }
            bst16<ST>(r0, o0off, o0);
          }
        }
      }
    }
  }
}

// the 16-wave kernels' form: wave (wm, wn) owns rows [64 wm, +64) x columns [64 wn, +64) of its tile
// picks the table form of PVRL_EPI_RESID_16 (a kernel-uniform choice)
template <int EPI, int BATCH>
__device__ __forceinline__ void nt_epilogue_sel(const GemmNT& p, f32x4 (&acc)[4][4], int m0, int n0, int rb, int c0, int c1, int lane) {
  if constexpr (EPI == PVRL_EPI_RESID_16) {
    if (p.aux_rowmod != 0) { nt_epilogue_at<EPI, BATCH, true>(p, acc, m0, n0, rb, c0, c1, lane); return; }
  }
  nt_epilogue_at<EPI, BATCH, false>(p, acc, m0, n0, rb, c0, c1, lane);
}
template <int EPI>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane) {
  nt_epilogue_sel<EPI, 2>(p, acc, m0, n0, wm * 64, wn * 64, wn * 64 + 32, lane);
}

// ---- the ragged last round (256x256 tiles only) --------------------------------------------------------------------
// An XCD owns T = panels x tiles_n tiles and has C CUs (one 16-wave workgroup each): after floor(T / C) full rounds, L = T mod C
// tiles are left for a last round that keeps L of C CUs busy for a whole tile time -- at M = 50,208: N = 768 -> 2 rounds + 11
// tiles, N = 2304 -> 7 rounds + ONE tile, N = 3072 -> 9 rounds + 12.  Those L tiles are cut along M into f = 4 (if 4 L <= C) or 2
// (if 2 L <= C) sub-tiles of 64 / 128 rows x 256 columns, one workgroup each, in the SAME launch (a separate launch for the tail
// serialises behind the first: round 1, -12 %).  A sub-tile is computed by the first 4 / 8 waves of its workgroup with the
// unchanged 64x64 wave block (same fragments, MFMA loop and epilogue); all 16 waves keep staging.  No partial sums, no fix-up.
struct NtTail { int full, L, f, nblk; };
__host__ __device__ inline NtTail nt_tail_plan(int cm, int tiles_n, int cus, int enable) {
  NtTail t;
  const int T = cm * tiles_n;
  t.full = enable ? (T / cus) * cus : T;
  t.L = T - t.full;
  t.f = 1;
  if (t.L > 0) {
    if (4 * t.L <= cus) t.f = 4;
    else if (2 * t.L <= cus) t.f = 2;
  }
  if (t.f == 1) { t.full = T; t.L = 0; }
  t.nblk = t.full + t.f * t.L;
  return t;
}

// WM x WN waves per workgroup, each owning a 64x64 output block: tile = (64*WM) x (64*WN).
//   <2,2>: 128x128, 4 waves, 64 KiB LDS, 2 workgroups / CU   (small M: order transformer, CLIP text)
//   <4,4>: 256x256, 16 waves, 128 KiB LDS, 1 workgroup / CU  (the encoder's 50k-row GEMMs: half the L2->LDS
//          bytes per FLOP of the 128x128 tile, which is what bounds the small tile at ~0.7-0.9 PFLOP/s)
// (the body is a device function of the problem and the workgroup's index inside it, so that one launch can carry several small
//  problems: gemm_nt_batched_kernel below)
template <int EPI, int WM, int WN>
__device__ __forceinline__ void gemm_nt_body(const GemmNT& p, const int bid) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr bool TAILS = WM == 4 && WN == 4;      // sub-tiles of the last round: 256x256 tiles only
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int NINST = (BM + BN) / 8;          // 1 KiB LDS-DMA instructions per stage
  constexpr bool EVEN = NINST % NW == 0;        // wide tiles (WN = 5 / 6: 10 / 12 waves) deal the instructions round-robin, the last ones guarded
  constexpr int PER = (NINST + NW - 1) / NW;    // per wave
  constexpr int SUB_STAGE = 48 * 1024;          // sub-tiles of the last round: (128 + 256) x 128 B per stage, THREE slots
  __shared__ __attribute__((aligned(16))) char smem[TAILS ? 3 * SUB_STAGE : 2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // L2-aware rasterisation.  Hardware places block b on XCD b % 8 (private 4 MiB L2 each).  Every XCD owns a
  // contiguous range of M-panels and walks it in groups of GM panels x all N-tiles, panel index fastest, so the
  // ~64 tiles resident on an XCD share GM activation panels and a few weight tiles instead of sweeping the whole
  // weight matrix per panel.
  const int GM = p.gm;   // tile rows per rasterisation group
  int tm, tn;
  int f = 1, sub = 0;    // this workgroup computes rows [sub * BM / f, (sub + 1) * BM / f) of its tile
  {
    const int xcd = bid & 7;
    int j = bid >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);                 // panels owned by this XCD
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if constexpr (TAILS) {
      const NtTail t = nt_tail_plan(cm, p.tiles_n, p.cus, p.tails);
      if (j >= t.nblk) return;
      if (j >= t.full) {
        const int s = j - t.full;
        f = t.f;
        j = t.full + s / f;
        sub = s - (s / f) * f;
      }
    } else {
      if (j >= cm * p.tiles_n) return;
    }
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int bm = BM / f;                                    // rows of this workgroup's (sub-)tile
  const int m0 = tm * BM + sub * bm, n0 = tn * BN;
  if (m0 >= p.M) return;                                    // a sub-tile past the ragged end of the last panel (block-uniform)
  const bool active = wm < WM / f;                          // waves that own a 64x64 block of the (sub-)tile
  const int xbytes = bm * BK * 2;                           // the W tile follows the X rows in each stage
  // ---- staging: instruction `it` of a stage copies 8 tile rows (X rows first, then W rows) ----
  // full tile: wave w issues instructions 4 w .. 4 w + 3; sub-tile ((bm + 256) / 8 = 48 / 40 instructions): it = 16 j + w
  const op_t* gsrc[PER];
  bool gval[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int it = (f == 1 && EVEN) ? wave * PER + j : j * NW + wave;
    gval[j] = it < (bm + BN) / 8;
    const int pc = lane & 7;
    if (it < bm / 8) {
      const int row = it * 8 + (lane >> 3);
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[j] = p.A + (long)grow * p.lda + ((pc ^ swz_x(row)) << 3);
    } else {
      const int row = ((gval[j] ? it : bm / 8) - bm / 8) * 8 + (lane >> 3);
      gsrc[j] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w<F32OUT>(row)) << 3);
    }
  }
  auto stage = [&](int buf, int k0) {            // full tiles: two slots, the compiler orders the LDS-DMA (vmcnt(0) at the barrier)
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if constexpr (EVEN) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);
      else if (gval[j]) glds16(gsrc[j] + k0, b + (j * NW + wave) * 1024);     // (wave-uniform guard)
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile), ks = 0; ks = 1 is ^64 ----
  int xoff[4], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rx = wm * 64 + t * 16 + i;
      xoff[t] = rx * 128 + ((q ^ swz_x(rx)) << 4);
      const int rw = wn * 64 + w_row<F32OUT>(t, i);
      woff[t] = rw * 128 + ((q ^ swz_w<F32OUT>(rw)) << 4);
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  // one K = 64 step of this wave's 64x64 block from the stage at bx (X rows) / bw (W rows)
  auto kstep = [&](const char* bx, const char* bw) {
    // all 16 fragment reads of the K-step are issued up front; the MFMAs of the first half overlap the
    // LDS latency of the second half (the compiler otherwise serialises read -> wait(0) -> 8 MFMAs)
    opx8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf0[t] = *reinterpret_cast<const opx8*>(bw + woff[t]);
      xf0[t] = *reinterpret_cast<const opx8*>(bx + xoff[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf1[t] = *reinterpret_cast<const opx8*>(bw + (woff[t] ^ 64));
      xf1[t] = *reinterpret_cast<const opx8*>(bx + (xoff[t] ^ 64));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = MFMA_16x16x32(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = MFMA_16x16x32(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
    // schedule: 8 reads, then one read per two MFMAs while the first half computes, then the rest of the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };

  if (!TAILS || f == 1) {
    // (Round 3, measured and dropped: issuing stages 0 AND 1 together before the first barrier.  The barrier's vmcnt(0) then waits
    //  for both, so no latency is hidden -- ViT-B step -2 %, MViT +0.4 % -- gpurun_out/r3_r_*.json.)
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and fences the previous compute
      if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
      const char* bx = smem + (kt & 1) * STAGE;
      kstep(bx, bx + XBYTES);
    }
  } else {
    // Sub-tile: 4 or 8 computing waves = 1 or 2 per SIMD, 512 / 1,024 MFMA cycles per K-step -- less than the latency of the next
    // stage's LDS-DMA, which a two-slot ring exposes every step (measured: a 64-row sub-tile cost ~0.7 of a full tile).  Three
    // slots, stages issued TWO steps ahead as raw LDS-DMA with counted waits: stage kt + 1 stays in flight across the barrier.
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    int nv = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) nv += gval[j] ? 1 : 0;
    nv = __builtin_amdgcn_readfirstlane(nv);
    auto stage3 = [&](int kt) {
      const unsigned b = sbase + (kt % 3) * SUB_STAGE;
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (gval[j]) glds16_raw_v(gsrc[j] + kt * BK, b + (j * NW + wave) * 1024);
    };
    stage3(0);
    if (nk > 1) stage3(1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {              // this wave's nv instructions of stage kt + 1 may stay in flight
        if (nv >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (nv == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // stage kt has landed for every wave; slot (kt + 2) % 3 = (kt - 1) % 3 is no longer read
      asm volatile("" ::: "memory");
      if (kt + 2 < nk) stage3(kt + 2);
      if (active) {
        const char* bx = smem + (kt % 3) * SUB_STAGE;
        kstep(bx, bx + xbytes);      // (its fragment reads are consumed by its MFMAs: done before the wave reaches the next barrier)
      }
    }
  }

  if (!TAILS || active) nt_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

template <int EPI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmNT p) {
  gemm_nt_body<EPI, WM, WN>(p, (int)blockIdx.x);
}

// Several SMALL problems in one launch (128 x 128 tiles): the 768^3 products of the fused temporal branch -- W_e = W_fc W_proj of every block
// at the start of a forward, dW_fc = dW_e W_proj^T and dW_proj = W_fc^T dW_e of every block at the end of a backward -- are 36 tiles each
// on a 256-CU chip, 16.6 us apiece whatever they carry; twelve of them in one launch fill it (432 tiles).  A problem's workgroups keep
// their XCD mapping: every first[] is a multiple of 8.
constexpr int NT_BATCH_MAX = 12;
struct GemmNTBatch { int n; int first[NT_BATCH_MAX + 1]; GemmNT prob[NT_BATCH_MAX]; };
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_batched_kernel(GemmNTBatch g) {
  int q = 0;
#pragma unroll
  for (int t = 1; t < NT_BATCH_MAX; ++t)
    if (t < g.n && (int)blockIdx.x >= g.first[t]) q = t;
  gemm_nt_body<EPI, 2, 2>(g.prob[q], (int)blockIdx.x - g.first[q]);
}


template <int EPI, int WM, int WN>
int launch_tile(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / (64 * WN);
  p.tiles_m = cdiv(p.M, 64 * WM);
  if (p.gm <= 0) p.gm = nt_gm_for(p.tiles_n);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;   // per-XCD tile lists padded to equal length (surplus blocks exit)
  if (WM == 4 && WN == 4) {                     // XCDs own ceil or floor(tiles_m / 8) panels: the longer block list sizes the grid
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    int nb = nt_tail_plan(qm + (rm ? 1 : 0), p.tiles_n, p.cus, p.tails).nblk;
    if (qm > 0) nb = std::max(nb, nt_tail_plan(qm, p.tiles_n, p.cus, p.tails).nblk);
    p.nwg = 8 * nb;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, WM, WN>), dim3(p.nwg), dim3(64 * WM * WN), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace
