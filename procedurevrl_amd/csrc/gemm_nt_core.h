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
  int m_off;   // global row of local row 0 (a launch may cover a row range of the logical GEMM)
  int gm;      // rasterisation group height in tiles
};

constexpr int BK = 64;
// rasterisation group height: measured on MI355X, 2 tile rows per group is 1-3 % ahead of 8-32 (A rows stay hot while W cycles)
constexpr int NT_GM = 2;

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

template <int EPI>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane) {
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  // ---- epilogue ----
  const int q = lane >> 4, i = lane & 15;
  const int nw0 = n0 + wn * 64;
  if constexpr (F32OUT) {
    // lane holds, for c = 0,1: columns nw0 + 32c + 8q + (0..7)  (tile 2c -> +0..3, tile 2c+1 -> +4..7)
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bv[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw0 + 32 * (nt >> 1) + 8 * q + 4 * (nt & 1))
                      : (f32x4){0.f, 0.f, 0.f, 0.f};
    // Per-row factors (DropPath) and the second bias are fetched HERE, before the first store: a load issued between the
    // stores is followed by s_waitcnt vmcnt(0), and on gfx9 that also waits for every earlier store to be acknowledged by
    // L2 -- four (rowscale) to sixteen (bias2) serial store round trips per thread.  bias2 has no 16 spare registers in the
    // 128-VGPR budget of the 16-wave tile: each lane keeps ONE column of the wave's 64 and the others come by ds_bpermute.
    float rs4[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rs4[mt] = p.rowscale ? p.rowscale[min(m0 + wm * 64 + mt * 16 + i, p.M - 1)] : 1.f;
    float b2lane = 0.f;
    if constexpr (EPI == PVRL_EPI_RESID_F32) {
      if (p.bias2) {
        b2lane = p.bias2[nw0 + lane];
        if (!p.rowscale) {                 // no row factor: rs * (acc + bias) + bias2 = acc + (bias + bias2)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nt][e] += __shfl(b2lane, 32 * (nt >> 1) + 8 * q + 4 * (nt & 1) + e, 64);
        }
      }
    }
    const bool b2_late = EPI == PVRL_EPI_RESID_F32 && p.bias2 && p.rowscale;
    // (Measured and rejected: finishing all 16 tiles in place and storing last, so that the second half's residual loads are
    //  not queued behind the first half's stores: -0.4 % on the step -- the early stores start the write traffic sooner.)
    // The residual loads of TWO row tiles (32 registers, freed by the MFMA fragments) go out together, twice: two memory
    // round trips per wave instead of four dependent ones (all 16 at once would spill).  With every CU in its epilogue at
    // the same time the loaded latency of a round trip is microseconds.  Rows past M load a clamped row and store nothing.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 rv[2][4];
      if constexpr (EPI == PVRL_EPI_RESID_F32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = min(m0 + wm * 64 + (2 * half + h) * 16 + i, p.M - 1);
          const int mr = p.aux_rowmod ? ((m + p.m_off) % p.aux_rowmod) : m;
          const float* r = (const float*)p.aux + (long)mr * p.aux_ld + nw0 + 8 * q;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) rv[h][nt] = *reinterpret_cast<const f32x4*>(r + 32 * (nt >> 1) + 4 * (nt & 1));
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mt = 2 * half + h;
        const int m = m0 + wm * 64 + mt * 16 + i;
        const float rs = rs4[mt];
        float* o = (float*)p.out0 + (long)m * p.ld0 + nw0 + 8 * q;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int off = 32 * (nt >> 1) + 4 * (nt & 1);
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = rs * (acc[mt][nt][e] + bv[nt][e]);
          if constexpr (EPI == PVRL_EPI_RESID_F32) {
            ov += rv[h][nt];
            if (b2_late) {
#pragma unroll
              for (int e = 0; e < 4; ++e) ov[e] += __shfl(b2lane, 8 * q + off + e, 64);
            }
          }
          if (m < p.M) *reinterpret_cast<f32x4*>(o + off) = ov;
        }
      }
    }
  } else {
    // lane holds, for c = 0,1: columns nw0 + 32c + 8q + (0..7)  (tile 2c -> +0..3, tile 2c+1 -> +4..7)
    float bv[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[c][e] = p.bias ? p.bias[nw0 + 32 * c + 8 * q + e] : 0.f;
    float rs4[4];                      // before the first store (see above)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rs4[mt] = p.rowscale ? p.rowscale[min(m0 + wm * 64 + mt * 16 + i, p.M - 1)] : 1.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      opx8 uv[2][2];
      if constexpr (EPI == PVRL_EPI_DGELU || EPI == PVRL_EPI_DQGELU) {   // the stored pre-activations of two row tiles: one round trip
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = min(m0 + wm * 64 + (2 * half + h) * 16 + i, p.M - 1);
#pragma unroll
          for (int c = 0; c < 2; ++c)
            uv[h][c] = *reinterpret_cast<const opx8*>((const op_t*)p.aux + (long)m * p.aux_ld + nw0 + 32 * c + 8 * q);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mt = 2 * half + h;
        const int m = m0 + wm * 64 + mt * 16 + i;
        if (m >= p.M) continue;
        const float rs = rs4[mt];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[mt][2 * c][e] + bv[c][e];
            v[4 + e] = acc[mt][2 * c + 1][e] + bv[c][4 + e];
          }
          const long col = nw0 + 32 * c + 8 * q;
          if constexpr (EPI == PVRL_EPI_BF16) {
            opx8 o0;
#pragma unroll
            for (int e = 0; e < 8; ++e) o0[e] = (op_t)(rs * v[e]);
            *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + col) = o0;
          } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
            opx8 u0, g0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              u0[e] = (op_t)v[e];
              g0[e] = (op_t)(EPI == PVRL_EPI_GELU ? gelu_erf(v[e]) : quick_gelu(v[e]));
            }
            *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + col) = u0;
            *reinterpret_cast<opx8*>((op_t*)p.out1 + (long)m * p.ld1 + col) = g0;
          } else {  // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
            const opx8 ua = uv[h][c];
            opx8 o0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float d = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
              o0[e] = (op_t)(rs * v[e] * d);
            }
            *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + col) = o0;
          }
        }
      }
    }
  }
}

// WM x WN waves per workgroup, each owning a 64x64 output block: tile = (64*WM) x (64*WN).
//   <2,2>: 128x128, 4 waves, 64 KiB LDS, 2 workgroups / CU   (small M: order transformer, CLIP text)
//   <4,4>: 256x256, 16 waves, 128 KiB LDS, 1 workgroup / CU  (the encoder's 50k-row GEMMs: half the L2->LDS
//          bytes per FLOP of the 128x128 tile, which is what bounds the small tile at ~0.7-0.9 PFLOP/s)
template <int EPI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmNT p) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int NINST = (BM + BN) / 8;          // 1 KiB LDS-DMA instructions per stage
  constexpr int PER = NINST / NW;               // per wave
  static_assert(NINST % NW == 0, "stage instructions must divide evenly over the waves");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // L2-aware rasterisation.  Hardware places block b on XCD b % 8 (private 4 MiB L2 each).  Every XCD owns a
  // contiguous range of M-panels and walks it in groups of GM panels x all N-tiles, panel index fastest, so the
  // ~64 tiles resident on an XCD share GM activation panels and a few weight tiles instead of sweeping the whole
  // weight matrix per panel.
  const int GM = p.gm;   // tile rows per rasterisation group (benchmark knob, default 8)
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);                 // panels owned by this XCD
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  // ---- staging: instruction `it` of a stage copies 8 tile rows (X rows first, then W rows) ----
  const op_t* gsrc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int it = wave * PER + j;
    const int pc = lane & 7;
    if (it < BM / 8) {
      const int row = it * 8 + (lane >> 3);
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[j] = p.A + (long)grow * p.lda + ((pc ^ swz_x(row)) << 3);
    } else {
      const int row = (it - BM / 8) * 8 + (lane >> 3);
      gsrc[j] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w<F32OUT>(row)) << 3);
    }
  }
  auto stage = [&](int buf, int k0) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);   // W tile follows the X tile
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
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and fences the previous compute
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* bx = smem + (kt & 1) * STAGE;
    const char* bw = bx + XBYTES;
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
  }

  nt_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

template <int EPI, int WM, int WN>
int launch_tile(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / (64 * WN);
  p.tiles_m = cdiv(p.M, 64 * WM);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;   // per-XCD tile lists padded to equal length (surplus blocks exit)
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, WM, WN>), dim3(p.nwg), dim3(64 * WM * WN), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace
