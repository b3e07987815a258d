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
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct GemmNT {
  const bf16* A; long lda;
  const bf16* W; long ldw;
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
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + i;
      if (m >= p.M) continue;
      const float rs = p.rowscale ? p.rowscale[m] : 1.f;
      float* o = (float*)p.out0 + (long)m * p.ld0 + nw0 + 8 * q;
      const float* r = nullptr;
      if constexpr (EPI == PVRL_EPI_RESID_F32) {
        const int mr = p.aux_rowmod ? ((m + p.m_off) % p.aux_rowmod) : m;
        r = (const float*)p.aux + (long)mr * p.aux_ld + nw0 + 8 * q;
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int off = 32 * (nt >> 1) + 4 * (nt & 1);
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = rs * (acc[mt][nt][e] + bv[nt][e]);
        if constexpr (EPI == PVRL_EPI_RESID_F32) {
          ov += *reinterpret_cast<const f32x4*>(r + off);
          if (p.bias2) ov += *reinterpret_cast<const f32x4*>(p.bias2 + nw0 + 8 * q + off);
        }
        *reinterpret_cast<f32x4*>(o + off) = ov;
      }
    }
  } else {
    // lane holds, for c = 0,1: columns nw0 + 32c + 8q + (0..7)  (tile 2c -> +0..3, tile 2c+1 -> +4..7)
    float bv[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[c][e] = p.bias ? p.bias[nw0 + 32 * c + 8 * q + e] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + i;
      if (m >= p.M) continue;
      const float rs = p.rowscale ? p.rowscale[m] : 1.f;
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
          bf16x8 o0;
#pragma unroll
          for (int e = 0; e < 8; ++e) o0[e] = (bf16)(rs * v[e]);
          *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + col) = o0;
        } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
          bf16x8 u0, g0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            u0[e] = (bf16)v[e];
            g0[e] = (bf16)(EPI == PVRL_EPI_GELU ? gelu_erf(v[e]) : quick_gelu(v[e]));
          }
          *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + col) = u0;
          *reinterpret_cast<bf16x8*>((bf16*)p.out1 + (long)m * p.ld1 + col) = g0;
        } else {  // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
          const bf16x8 ua = *reinterpret_cast<const bf16x8*>((const bf16*)p.aux + (long)m * p.aux_ld + col);
          bf16x8 o0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
            o0[e] = (bf16)(rs * v[e] * d);
          }
          *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + col) = o0;
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
  const bf16* gsrc[PER];
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
    bf16x8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf0[t] = *reinterpret_cast<const bf16x8*>(bw + woff[t]);
      xf0[t] = *reinterpret_cast<const bf16x8*>(bx + xoff[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf1[t] = *reinterpret_cast<const bf16x8*>(bw + (woff[t] ^ 64));
      xf1[t] = *reinterpret_cast<const bf16x8*>(bx + (xoff[t] ^ 64));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
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

// ---------------------------------------------------------------------------------------------------------
// 256x256 tile with EIGHT waves, each owning a 128(m) x 64(n) block (8 x 4 MFMA tiles, 128 accumulator VGPRs): 24
// ds_read_b128 per 64 MFMAs instead of 32 per 64 for two 64x64 wave blocks, half as many waves meeting at each barrier.
// Same LDS image, swizzles, staging (8 LDS-DMA instructions per wave and stage) and epilogue as gemm_nt_kernel.
// MEASURED (same-process A/B, 50k-row shapes): within +-8 % of the 16-wave kernel (faster on the HBM-bound fp32-residual
// epilogue, 450 vs 432 TFLOP/s; slower on K = 3072, 955 vs 1040) -- no net win, kept behind benchmark knob 5.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_w128_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, NW = 8;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int PER = (BM + BN) / 8 / NW;   // 8
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int GM = 8;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16* gsrc[PER];
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
    for (int j = 0; j < PER; ++j) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);
  };

  int xoff[8], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int rx = wm * 128 + t * 16 + i;
      xoff[t] = rx * 128 + ((q ^ swz_x(rx)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rw = wn * 64 + w_row<F32OUT>(t, i);
      woff[t] = XBYTES + rw * 128 + ((q ^ swz_w<F32OUT>(rw)) << 4);
    }
  }

  f32x4 acc[2][4][4];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[hh][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* b = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[8], wf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(b + (woff[t] ^ (ks << 6)));
#pragma unroll
      for (int t = 0; t < 8; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(b + (xoff[t] ^ (ks << 6)));
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt >> 2][mt & 3][nt] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt >> 2][mt & 3][nt], 0, 0, 0);
    }
  }
  nt_epilogue<EPI>(p, acc[0], m0, n0, 2 * wm, wn, lane);
  nt_epilogue<EPI>(p, acc[1], m0, n0, 2 * wm + 1, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Deep-pipelined variant for the 256x256 tile: BK = 32 stages (32 KiB each) in a 4-deep LDS ring, LDS-DMA issued
// THREE stages ahead, counted `s_waitcnt vmcnt(N)` + raw `s_barrier` so that loads stay in flight across barriers.
// (`__syncthreads()` drains vmcnt(0) whenever an LDS-DMA is pending.)  MEASURED on MI355X (tools/bench_kernels.py, same-
// process A/B, 50k-row shapes): 0-12 % SLOWER than the 2-stage BK = 64 kernel above (e.g. 892 vs 1028 TFLOP/s at
// N=768,K=2304) -- twice the barriers per MFMA cost more than the hidden latency buys at 4 waves/SIMD.  Kept behind
// the benchmark knob (tile 4) as a tested reference point; the heuristic never selects it.
// LDS tiles are [256 rows][32 bf16] = 64-byte rows; 16-byte chunk c of row r lives at chunk c ^ g(a(r)), g(a) = (4-a)&3,
// a(r) = (r>>2)&3 for naturally ordered rows and (r>>3)&3 for the bf16-output W row order: every ds_read_b128 lane
// group then touches 16 distinct 16-byte slots.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int g64(int a) { return (4 - a) & 3; }
template <bool F32OUT> __device__ __forceinline__ int swz_w64(int row) { return g64((row >> 3) & 3); }
__device__ __forceinline__ int swz_x64(int row) { return g64((row >> 2) & 3); }

template <int EPI>
__global__ __launch_bounds__(1024) void gemm_nt_pipe_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, NW = 16, BKS = 32, NS = 4;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BKS * 2, STAGE = 2 * XBYTES;   // 16 KiB + 16 KiB
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int GM = 8;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: 32 LDS-DMA instructions per stage (16 rows x 64 B each); wave w issues X instruction w and W instruction w
  const bf16* gx;
  const bf16* gw;
  {
    const int row = wave * 16 + (lane >> 2);
    const int pc = lane & 3;
    int grow = m0 + row;
    grow = grow < p.M ? grow : p.M - 1;
    gx = p.A + (long)grow * p.lda + ((pc ^ swz_x64(row)) << 3);
    gw = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w64<F32OUT>(row)) << 3);
  }
  auto stage = [&](int kt) {
    char* b = smem + (kt & (NS - 1)) * STAGE;
    glds16(gx + kt * BKS, b + wave * 1024);
    glds16(gw + kt * BKS, b + XBYTES + wave * 1024);
  };

  int xoff[4], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rx = wm * 64 + t * 16 + i;
      xoff[t] = rx * 64 + ((q ^ swz_x64(rx)) << 4);
      const int rw = wn * 64 + w_row<F32OUT>(t, i);
      woff[t] = XBYTES + rw * 64 + ((q ^ swz_w64<F32OUT>(rw)) << 4);
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BKS;
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed: this wave has issued 2 loads per stage for stages .. min(kt+2, nk-1)
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave's part of stage kt is in LDS; buffer (kt-1)%4 is free
    if (kt + 3 < nk) stage(kt + 3);
    const char* b = smem + (kt & (NS - 1)) * STAGE;
    bf16x8 xf[4], wf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = *reinterpret_cast<const bf16x8*>(b + woff[t]);
      xf[t] = *reinterpret_cast<const bf16x8*>(b + xoff[t]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
  }
  nt_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Generalised BK = 32 ring kernel: tile (64 WM) x (64 WN), WM*WN waves, NS LDS stages, LDS-DMA NS-1 stages ahead with
// counted vmcnt.  Smaller tiles / fewer stages leave room for TWO workgroups per CU (e.g. 256x128, 3 stages = 72 KiB), so
// one workgroup's barrier / DMA wait is covered by the other's MFMAs.  MEASURED (same-process A/B, 50k-row shapes):
// 256x128 / 3 stages reaches 87-90 % of the 16-wave 256x256 kernel (qkv 819 vs 943, dfc1 915 vs 1022 TFLOP/s), 256x128 /
// 2 stages 81 %, 128x128 / 4 stages 71-74 %: the smaller tiles' extra L2 traffic and halved MFMAs per barrier cost more
// than the second workgroup hides.  Benchmark knobs 7-9.
// ---------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}

template <int EPI, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void gemm_nt_ring_kernel(GemmNT p) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN, BKS = 32;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BKS * 2, WBYTES = BN * BKS * 2, STAGE = XBYTES + WBYTES;
  constexpr int PER = (BM + BN) / 16 / NW;
  static_assert((BM + BN) / 16 % NW == 0, "staging must divide over the waves");
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16* gsrc[PER];
  int gdst[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int it = wave * PER + e;
    const int pc = lane & 3;
    if (it < BM / 16) {
      const int row = it * 16 + (lane >> 2);
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[e] = p.A + (long)grow * p.lda + ((pc ^ swz_x64(row)) << 3);
      gdst[e] = it * 1024;
    } else {
      const int row = (it - BM / 16) * 16 + (lane >> 2);
      gsrc[e] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w64<F32OUT>(row)) << 3);
      gdst[e] = XBYTES + (it - BM / 16) * 1024;
    }
  }
  auto stage = [&](int kt) {
    char* b = smem + (kt % NS) * STAGE;
#pragma unroll
    for (int e = 0; e < PER; ++e) glds16(gsrc[e] + kt * BKS, b + gdst[e]);
  };

  int xoff[4], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rx = wm * 64 + t * 16 + i;
      xoff[t] = rx * 64 + ((q ^ swz_x64(rx)) << 4);
      const int rw = wn * 64 + w_row<F32OUT>(t, i);
      woff[t] = XBYTES + rw * 64 + ((q ^ swz_w64<F32OUT>(rw)) << 4);
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BKS;
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) stage(st);
  for (int kt = 0; kt < nk; ++kt) {
    // stages issued so far: .. min(kt + NS - 2, nk - 1); those after kt may stay in flight
    const int ahead = min(kt + NS - 2, nk - 1) - kt;
    if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * PER>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nk) stage(kt + NS - 1);
    const char* b = smem + (kt % NS) * STAGE;
    bf16x8 xf[4], wf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = *reinterpret_cast<const bf16x8*>(b + woff[t]);
      xf[t] = *reinterpret_cast<const bf16x8*>(b + xoff[t]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
  }
  nt_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------
// 256x256 tile with FOUR waves (2 x 2), each owning a 128 x 128 block = 8 x 8 MFMA tiles = 256 accumulator AGPRs (one
// wave per SIMD, 512-register budget): 16 ds_read_b128 per 64 MFMAs, half the LDS read bytes per FLOP of the 64x64 wave
// block.  BK = 32 stages (32 KiB) in a 4-deep LDS ring filled by raw-ISA LDS-DMA three stages ahead (counted vmcnt + raw
// s_barrier, one barrier per 64 MFMAs); the X fragments of step s+1 replace those of step s in place right after their
// row of MFMAs, the W fragments are double-buffered.  Same [rows][32] LDS image / swizzles as the pipe kernel, same
// epilogue as every other NT kernel.  MEASURED (same-process A/B, 50k-row shapes): 25-30 % SLOWER than the 16-wave
// kernel (qkv 702 vs 908, dfc1 762 vs 1032, fc1+GELU 527 vs 682 TFLOP/s): with no transposition work to hide, four
// waves per SIMD cover LDS / MFMA latencies better than one software-pipelined wave.  Kept behind benchmark knob 6.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, BKS = 32, NS = 4;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BKS * 2, STAGE = 2 * XBYTES;   // 16 KiB + 16 KiB
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int GM = 8;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: 32 LDS-DMA instructions per stage (16 rows x 64 B each): waves 0,1 bring X (rows 128 w ..), waves 2,3 bring W
  const char* gsrc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = (wave & 1) * 128 + e * 16 + (lane >> 2);
    const int pc = lane & 3;
    if (wave < 2) {
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[e] = reinterpret_cast<const char*>(p.A + (long)grow * p.lda + ((pc ^ swz_x64(row)) << 3));
    } else {
      gsrc[e] = reinterpret_cast<const char*>(p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w64<F32OUT>(row)) << 3));
    }
  }
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const unsigned dbase = (wave < 2 ? 0 : XBYTES) + (wave & 1) * 128 * 64;
  auto stage = [&](int kt, int nk) {
    const int kc = kt < nk ? kt : nk - 1;                       // surplus ring slots re-load the last stage (never read)
    const unsigned b = smem_base + (kt & (NS - 1)) * STAGE + dbase;
#pragma unroll
    for (int e = 0; e < 8; ++e) glds16_raw_v(gsrc[e] + kc * (BKS * 2), b + e * 1024);
  };

  // fragment addresses: X block a (0,1) tile mt -> row wm*128 + a*64 + mt*16 + i ; W block b tile nt -> wn*128 + b*64 + w_row
  int xoff[8], woff[8];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int rx = wm * 128 + (t >> 2) * 64 + (t & 3) * 16 + i;
      xoff[t] = rx * 64 + ((q ^ swz_x64(rx)) << 4);
      const int rw = wn * 128 + (t >> 2) * 64 + w_row<F32OUT>(t & 3, i);
      woff[t] = XBYTES + rw * 64 + ((q ^ swz_w64<F32OUT>(rw)) << 4);
    }
  }

  f32x4 acc[2][2][4][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 xf[7], xa[1], xb[1], wfa[8], wfb[8];
  auto rd = [&](const char* b, int off) { return *reinterpret_cast<const bf16x8*>(b + off); };
  // one K = 32 step: 8 rows (X tile r) of 8 MFMAs; LDS reads of the next stage are issued after rows 0..6 only
  auto step = [&](const bf16x8* wc, bf16x8* wnx, const bf16x8* xc7, bf16x8* xn7, const char* nb) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bf16x8 xr = r < 7 ? xf[r] : xc7[0];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[r >> 2][t >> 2][r & 3][t & 3] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xr, acc[r >> 2][t >> 2][r & 3][t & 3], 0, 0, 0);
      if (r < 4) {
        wnx[2 * r] = rd(nb, woff[2 * r]);
        wnx[2 * r + 1] = rd(nb, woff[2 * r + 1]);
        xf[r] = rd(nb, xoff[r]);
      } else if (r < 6) {
        xf[r] = rd(nb, xoff[r]);
        if (r == 4) xn7[0] = rd(nb, xoff[7]);
      } else if (r == 6) {
        xf[6] = rd(nb, xoff[6]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  };

  const int nk = p.K / BKS;     // even (K % 64 == 0)
  stage(0, nk); stage(1, nk); stage(2, nk);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) xf[t] = rd(smem, xoff[t]);
    else xa[0] = rd(smem, xoff[t]);
    wfa[t] = rd(smem, woff[t]);
  }
  for (int kt = 0; kt < nk; kt += 2) {
    const int hb = ((kt >> 1) & 1) * 2;                      // ring slot of stage kt: 0 or 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // stage kt+1 landed (this wave's part; kt+2 stays in flight)
    __builtin_amdgcn_s_barrier();                            // ... everyone's part; slot (kt+3)%4 is free
    stage(kt + 3, nk);
    step(wfa, wfb, xa, xb, smem + (hb + 1) * STAGE);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(kt + 4, nk);
    step(wfb, wfa, xb, xa, smem + (hb ^ 2) * STAGE);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) nt_epilogue<EPI>(p, acc[a][b], m0, n0, 2 * wm + a, 2 * wn + b, lane);
}

// ---------------------------------------------------------------------------------------------------------
// The 4-wave 128x128-per-wave kernel with v_mfma_f32_32x32x16_bf16: 32-cycle MFMAs leave a one-wave-per-SIMD kernel
// twice the issue slots per MFMA for its LDS reads / DMA issue (the 16x16x32 form above loses 25-30 % to the 16-wave
// kernel; for the TN kernel the same switch was worth 13-23 %).  Wave block = 4 x 4 blocks of 32 x 32; the W row feeding
// MFMA row rho of a block is n = 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3), so that lane (m = lane % 32, kg = lane / 32) ends up
// with the 16 consecutive output columns 16 kg .. 16 kg + 15 of the block: 32-byte (bf16) / 64-byte (fp32) pieces per lane,
// two lanes = one 128-byte line of fp32.  Natural-order LDS swizzle for both operands.  MEASURED: no better than the
// 16x16x32 form (qkv 689, dfc1 782 TFLOP/s vs 963 / 1036 for the 16-wave kernel): for NT the one-wave-per-SIMD structure
// itself loses (three 1024-cycle stages of DMA look-ahead, every stall exposed), not the MFMA shape.  Benchmark knob 10.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epi_row16(const GemmNT& p, const f32x16& a, int m, int n) {
  // 16 consecutive output columns n .. n+15 of output row m (m < M)
  const float rs = p.rowscale ? p.rowscale[m] : 1.f;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = a[e] + (p.bias ? p.bias[n + e] : 0.f);
  if constexpr (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32) {
    float* o = (float*)p.out0 + (long)m * p.ld0 + n;
    const float* r = nullptr;
    if constexpr (EPI == PVRL_EPI_RESID_F32) {
      const int mr = p.aux_rowmod ? ((m + p.m_off) % p.aux_rowmod) : m;
      r = (const float*)p.aux + (long)mr * p.aux_ld + n;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 ov = (f32x4){rs * v[4 * c], rs * v[4 * c + 1], rs * v[4 * c + 2], rs * v[4 * c + 3]};
      if constexpr (EPI == PVRL_EPI_RESID_F32) {
        ov += *reinterpret_cast<const f32x4*>(r + 4 * c);
        if (p.bias2) ov += *reinterpret_cast<const f32x4*>(p.bias2 + n + 4 * c);
      }
      *reinterpret_cast<f32x4*>(o + 4 * c) = ov;
    }
  } else if constexpr (EPI == PVRL_EPI_BF16) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 o0;
#pragma unroll
      for (int e = 0; e < 8; ++e) o0[e] = (bf16)(rs * v[8 * c + e]);
      *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + n + 8 * c) = o0;
    }
  } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 u0, g0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u0[e] = (bf16)v[8 * c + e];
        g0[e] = (bf16)(EPI == PVRL_EPI_GELU ? gelu_erf(v[8 * c + e]) : quick_gelu(v[8 * c + e]));
      }
      *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + n + 8 * c) = u0;
      *reinterpret_cast<bf16x8*>((bf16*)p.out1 + (long)m * p.ld1 + n + 8 * c) = g0;
    }
  } else {   // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 ua = *reinterpret_cast<const bf16x8*>((const bf16*)p.aux + (long)m * p.aux_ld + n + 8 * c);
      bf16x8 o0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
        o0[e] = (bf16)(rs * v[8 * c + e] * d);
      }
      *reinterpret_cast<bf16x8*>((bf16*)p.out0 + (long)m * p.ld0 + n + 8 * c) = o0;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4x32_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, BKS = 32, NS = 4;
  constexpr int XBYTES = BM * BKS * 2, STAGE = 2 * XBYTES;   // 16 KiB + 16 KiB
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: waves 0,1 bring X rows 128 w .., waves 2,3 W rows; 8 LDS-DMA instructions (16 rows x 64 B) per wave and stage
  const char* gsrc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = (wave & 1) * 128 + e * 16 + (lane >> 2);
    const int pc = lane & 3;
    if (wave < 2) {
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[e] = reinterpret_cast<const char*>(p.A + (long)grow * p.lda + ((pc ^ swz_x64(row)) << 3));
    } else {
      gsrc[e] = reinterpret_cast<const char*>(p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_x64(row)) << 3));
    }
  }
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const unsigned dbase = (wave < 2 ? 0 : XBYTES) + (wave & 1) * 128 * 64;
  auto stage = [&](int kt, int nk) {
    const int kc = kt < nk ? kt : nk - 1;
    const unsigned b = smem_base + (kt & (NS - 1)) * STAGE + dbase;
#pragma unroll
    for (int e = 0; e < 8; ++e) glds16_raw_v(gsrc[e] + kc * (BKS * 2), b + e * 1024);
  };

  // fragment addresses of block b, K = 16 sub-step u: row r, 16-byte chunk 2u + kg
  const int i32 = lane & 31, kg = lane >> 5;
  int xoff[4], woff[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int rx = wm * 128 + b * 32 + i32;
    xoff[b] = rx * 64 + ((kg ^ swz_x64(rx)) << 4);
    const int rw = wn * 128 + b * 32 + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
    woff[b] = XBYTES + rw * 64 + ((kg ^ swz_x64(rw)) << 4);
  }
  // chunk 2u + kg: (2u + kg) ^ s = (kg ^ s) ^ 2u  -> sub-step 1 is the address ^ 32
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const bf16x8*>(b + (off ^ (u << 5))); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  bf16x8 xa[4], wa[4], xb[4], wb[4];
  auto mma = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
  };
  auto step = [&](const char* nb) {            // one K = 32 stage; fragments of the next stage replace the set just used
    mma(xa, wa);
#pragma unroll
    for (int t = 0; t < 4; ++t) { xa[t] = rd(nb, xoff[t], 0); wa[t] = rd(nb, woff[t], 0); }
    mma(xb, wb);
#pragma unroll
    for (int t = 0; t < 4; ++t) { xb[t] = rd(nb, xoff[t], 1); wb[t] = rd(nb, woff[t], 1); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
  };

  const int nk = p.K / BKS;     // even (K % 64 == 0)
  stage(0, nk); stage(1, nk); stage(2, nk);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xa[t] = rd(smem, xoff[t], 0); wa[t] = rd(smem, woff[t], 0);
    xb[t] = rd(smem, xoff[t], 1); wb[t] = rd(smem, woff[t], 1);
  }
  for (int kt = 0; kt < nk; kt += 2) {
    const int hb = ((kt >> 1) & 1) * 2;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(kt + 3, nk);
    step(smem + (hb + 1) * STAGE);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(kt + 4, nk);
    step(smem + (hb ^ 2) * STAGE);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // D layout: lane (col m = lane % 32, kg): register e <-> MFMA row rho = (e/4)*8 + kg*4 + e%4 <-> column 16 kg + e
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 128 + mb * 32 + i32;
    if (m >= p.M) continue;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) epi_row16<EPI>(p, acc[mb][nb], m, n0 + wn * 128 + nb * 32 + 16 * kg);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The 4-wave 32x32x16 kernel above with the staging discipline of gemm_tn_rt32: operands come in through registers (eight
// raw-ISA 16-byte global loads per lane and stage, two stages ahead, ONE counted vmcnt per stage) and go to LDS with
// ds_write_b128 -- an LDS-DMA piece costs 60-185 issue cycles next to MFMAs (MI355X_MICROARCH.md), eight of them a
// stage's whole MFMA time; a global load + a ds_write_b128 cost a fraction of that.  Two LDS slots of 32 KiB, operand
// image in rotated 16-byte-chunk planes (conflict-free reads and writes).  Knob 13.  MEASURED (MI355X, M = 50,208): 698-814
// TFLOP/s against 969-1069 for the 16-wave default (qkv 254 vs 183 us, dfc1 292 vs 222 us) -- the same as the LDS-DMA form
// (knob 10), so neither the DMA issue cost nor bank conflicts were what held the 4-wave NT kernels back: its stage takes
// ~2,100 cycles for 1,024 cycles of MFMA, exactly like gemm_tn_rt32's; with one wave per SIMD every LDS / barrier latency is
// exposed, with four (the default) it is not.  The clean ISA (one vmcnt(8) per stage, no scratch in the loop) rules out a
// scheduling accident.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_rt32_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, BKS = 32, NS = 2;
  constexpr int XBYTES = BM * BKS * 2, STAGE = 2 * XBYTES;   // 16 KiB + 16 KiB
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: waves 0,1 bring X rows 128 w .., waves 2,3 W rows; 8 LDS-DMA instructions (16 rows x 64 B) per wave and stage
  const char* gsrc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = (wave & 1) * 128 + e * 16 + (lane >> 2);
    const int pc = lane & 3;
    if (wave < 2) {
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[e] = reinterpret_cast<const char*>(p.A + (long)grow * p.lda + (pc << 3));
    } else {
      gsrc[e] = reinterpret_cast<const char*>(p.W + (long)(n0 + row) * p.ldw + (pc << 3));
    }
  }
  // LDS image of an operand stage: four 4 KiB planes, plane c = the 16-byte k-chunk c of all 256 rows, rotated by 64 c bytes:
  // a fragment read (32 consecutive rows of one chunk) is 512 contiguous bytes, and the four chunks of a row -- written by
  // four neighbouring lanes -- land 64 bytes apart in the bank row instead of on the same banks
  auto lds_off = [&](int row, int c) { return c * 4096 + ((row * 16 + c * 64) & 4095); };
  const int opbase = wave < 2 ? 0 : XBYTES;
  // raw-ISA loads (the compiler's own waits would drain the younger register set, see gemm_tn_rt32): set r holds the 8
  // 16-byte pieces this lane contributes to one stage; `wait_set` = all 8 of the OLDER set have landed
  auto gload = [&](u32x4* r, int kt, int nk) {
    const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[e]) : "v"(gsrc[e] + kc * (BKS * 2)) : "memory");
  };
  auto wait_set = [&](u32x4* r) {
    asm volatile("s_waitcnt vmcnt(8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
  };
  int woffs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) woffs[e] = opbase + lds_off((wave & 1) * 128 + e * 16 + (lane >> 2), lane & 3);
  auto lwrite = [&](const u32x4* r, int e, char* slot) { *reinterpret_cast<u32x4*>(slot + woffs[e]) = r[e]; };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // fragment addresses of block b, K = 16 sub-step u: row r, 16-byte chunk 2u + kg
  const int i32 = lane & 31, kg = lane >> 5;
  int xoff[2][4], woff[2][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int rx = wm * 128 + b * 32 + i32;
    const int rw = wn * 128 + b * 32 + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      xoff[u][b] = lds_off(rx, 2 * u + kg);
      woff[u][b] = XBYTES + lds_off(rw, 2 * u + kg);
    }
  }
  auto rd = [&](const char* b, int off) { return *reinterpret_cast<const bf16x8*>(b + off); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  bf16x8 xa[4], wa[4], xb[4], wb[4];
  auto mma = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
  };
  u32x4 ra[8], rb[8];
  // one K = 32 stage: 32 MFMAs | 16 fragment reads of the NEXT stage from `rs` | the 8 staged pieces of set r -> slot `ws`
  auto step = [&](const char* rs, const u32x4* r, char* ws) {
    mma(xa, wa);
#pragma unroll
    for (int t = 0; t < 4; ++t) { xa[t] = rd(rs, xoff[0][t]); wa[t] = rd(rs, woff[0][t]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) lwrite(r, e, ws);
    mma(xb, wb);
#pragma unroll
    for (int t = 0; t < 4; ++t) { xb[t] = rd(rs, xoff[1][t]); wb[t] = rd(rs, woff[1][t]); }
#pragma unroll
    for (int e = 4; e < 8; ++e) lwrite(r, e, ws);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };

  const int nk = p.K / BKS;     // even (K % 64 == 0)
  char* slot0 = smem;
  char* slot1 = smem + STAGE;
  gload(ra, 0, nk);
  gload(rb, 1, nk);
  wait_set(ra);
#pragma unroll
  for (int e = 0; e < 8; ++e) lwrite(ra, e, slot0);
  gload(ra, 2, nk);
  lds_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xa[t] = rd(slot0, xoff[0][t]); wa[t] = rd(slot0, woff[0][t]);
    xb[t] = rd(slot0, xoff[1][t]); wb[t] = rd(slot0, woff[1][t]);
  }
  wait_set(rb);
#pragma unroll
  for (int e = 0; e < 8; ++e) lwrite(rb, e, slot1);
  gload(rb, 3, nk);
  for (int kt = 0; kt < nk; kt += 2) {
    lds_barrier();
    wait_set(ra);
    step(slot1, ra, slot0);
    gload(ra, kt + 4, nk);
    lds_barrier();
    wait_set(rb);
    step(slot0, rb, slot1);
    gload(rb, kt + 5, nk);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // D layout: lane (col m = lane % 32, kg): register e <-> MFMA row rho = (e/4)*8 + kg*4 + e%4 <-> column 16 kg + e
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 128 + mb * 32 + i32;
    if (m >= p.M) continue;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) epi_row16<EPI>(p, acc[mb][nb], m, n0 + wn * 128 + nb * 32 + 16 * kg);
  }
}

// ---------------------------------------------------------------------------------------------------------
// 8 waves (2 per SIMD), each a 128(m) x 64(n) block as 4 x 2 blocks of v_mfma_f32_32x32x16_bf16; BK = 64, 2-stage LDS-DMA,
// natural-order swizzle (swz_x) for both operands, W rows permuted as in the 4-wave 32x32 kernel, generic 16-column row
// epilogue.  Benchmark knob 11.  MEASURED (MI355X, M=50208): 15-22% slower than the 16-wave default on every NT shape
// of the step (qkv 232 vs 191 us, fc1 432 vs 355 us): two waves per SIMD hide less of the LDS-read latency than four.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_w8x32_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, NW = 8;
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int PER = (BM + BN) / 8 / NW;   // 8
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16* gsrc[PER];
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
      gsrc[j] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_x(row)) << 3);
    }
  }
  auto stage = [&](int buf, int k0) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);
  };

  // fragment of block b for K = 16 sub-step u (0..3): row r, 16-byte chunk 2u + kg of the 128-byte row
  const int i32 = lane & 31, kg = lane >> 5;
  int xoff[4], woff[2];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int rx = wm * 128 + b * 32 + i32;
    xoff[b] = rx * 128 + ((kg ^ swz_x(rx)) << 4);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int rw = wn * 64 + b * 32 + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
    woff[b] = XBYTES + rw * 128 + ((kg ^ swz_x(rw)) << 4);
  }
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const bf16x8*>(b + (off ^ (u << 5))); };

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* b = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bf16x8 xf[4], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = rd(b, woff[t], u);
#pragma unroll
      for (int t = 0; t < 4; ++t) xf[t] = rd(b, xoff[t], u);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 128 + mb * 32 + i32;
    if (m >= p.M) continue;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) epi_row16<EPI>(p, acc[mb][nb], m, n0 + wn * 64 + nb * 32 + 16 * kg);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The default 16-wave 256x256 tile with v_mfma_f32_32x32x16_bf16 (wave block 64 x 64 = 2 x 2 blocks): the same LDS bytes
// per FLOP, half the MFMA instructions and half the operand-register reads per FLOP.  Benchmark knob 12.  MEASURED
// (MI355X, M=50208): 12-20 % slower than the 16x16x32 form on every shape (qkv 228 vs 191 us, dfc1 264 vs 228 us): with
// only 2 x 2 accumulator blocks a wave has 4 independent 64-cycle MFMAs in flight instead of 16 32-cycle ones.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(1024) void gemm_nt_w16x32_kernel(GemmNT p) {
  constexpr int BM = 256, BN = 256, NW = 16;
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int PER = (BM + BN) / 8 / NW;   // 8
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 4 x 4 waves of 64 x 64
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16* gsrc[PER];
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
      gsrc[j] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_x(row)) << 3);
    }
  }
  auto stage = [&](int buf, int k0) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);
  };

  // fragment of block b for K = 16 sub-step u (0..3): row r, 16-byte chunk 2u + kg of the 128-byte row
  const int i32 = lane & 31, kg = lane >> 5;
  int xoff[2], woff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int rx = wm * 64 + b * 32 + i32;
    xoff[b] = rx * 128 + ((kg ^ swz_x(rx)) << 4);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int rw = wn * 64 + b * 32 + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
    woff[b] = XBYTES + rw * 128 + ((kg ^ swz_x(rw)) << 4);
  }
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const bf16x8*>(b + (off ^ (u << 5))); };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* b = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bf16x8 xf[2], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = rd(b, woff[t], u);
#pragma unroll
      for (int t = 0; t < 2; ++t) xf[t] = rd(b, xoff[t], u);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = m0 + wm * 64 + mb * 32 + i32;
    if (m >= p.M) continue;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) epi_row16<EPI>(p, acc[mb][nb], m, n0 + wn * 64 + nb * 32 + 16 * kg);
  }
}

// ---------------------------------------------------------------------------
// Small fp32 GEMM for the projection head / step-logit path, where M is a few
// dozen rows and the reference keeps fp32 (lib/models/vit.py:299-307):
//   C[M,N] = alpha * (A[M,K] . B[N,K]^T) + bias[N]        (all fp32)
// HBM-bound on B (label_emb: 9871 x 512 fp32 = 20 MB): each workgroup streams 16
// rows of B once, coalesced, against up to 64 rows of A held in LDS.
// ---------------------------------------------------------------------------
// 64x64 output tile, 32-deep K chunks, 4x4 outputs per thread from k-major LDS tiles (float4 reads); optional split-K
// over grid.z (the logits backward has M = 32, N = 512, K = 9871: 8 output tiles only) into fp32 partials that a second
// kernel sums in a fixed order -- no atomics, so two runs of a training step are bit-identical.
constexpr int SG_T = 64, SG_K = 32, SG_LD = 68;
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             const float* __restrict__ bias, float alpha,
                                                             float* __restrict__ C, long ldc, int M, int N, int K,
                                                             int kchunk, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float sa[SG_K][SG_LD];
  __shared__ __attribute__((aligned(16))) float sb[SG_K][SG_LD];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * SG_T, m0 = blockIdx.y * SG_T;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const int tx = tid & 15, ty = tid >> 4;
  const int lrow = tid >> 2, lk = (tid & 3) * 8;      // loader: one tile row, 8 consecutive k
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += SG_K) {
    float va[8], vb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + lk + e;
      va[e] = (m0 + lrow < M && k < kend) ? A[(long)(m0 + lrow) * lda + k] : 0.f;
      vb[e] = (n0 + lrow < N && k < kend) ? B[(long)(n0 + lrow) * ldb + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[lk + e][lrow] = va[e]; sb[lk + e][lrow] = vb[e]; }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SG_K; ++k) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sa[k][ty * 4]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb[k][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      if (split) part[((long)blockIdx.z * M + m) * N + n] = acc[i][j];      // deterministic: partials + ordered reduce
      else C[(long)m * ldc + n] = alpha * acc[i][j] + (bias ? bias[n] : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void f32_small_reduce_kernel(const float* __restrict__ part, int splits, long MN, int N,
                                                               const float* __restrict__ bias, float alpha,
                                                               float* __restrict__ C, long ldc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
    float a = 0.f;
    for (int z = 0; z < splits; ++z) a += part[(long)z * MN + i];
    const long m = i / N;
    const int n = (int)(i - m * N);
    C[m * ldc + n] = alpha * a + (bias ? bias[n] : 0.f);
  }
}

static int f32_small_plan(int64_t M, int64_t N, int64_t K, int* kchunk_out) {
  const int tiles = cdiv(N, SG_T) * cdiv(M, SG_T);
  int splits = 1;
  if (tiles < 128 && K >= 512) {           // few output tiles and a long reduction: split K over grid.z
    splits = 256 / tiles;
    const int maxs = (int)(K / 128);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  const int kchunk = cdiv(cdiv(K, splits), SG_K) * SG_K;
  if (kchunk_out) *kchunk_out = kchunk;
  return cdiv(K, kchunk);
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

int g_nt_gm = 2;   // measured on MI355X: 2 tile rows per group is 1-3 % ahead of 8-32 (A rows stay hot while W cycles)
int g_force_tile = 0;   // 0 = heuristic, 1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x256 deep pipeline (knob)

template <int EPI>
int launch_pipe(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / 256;
  p.tiles_m = cdiv(p.M, 256);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
  hipLaunchKernelGGL((gemm_nt_pipe_kernel<EPI>), dim3(p.nwg), dim3(1024), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int EPI>
int launch_w128(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / 256;
  p.tiles_m = cdiv(p.M, 256);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
  hipLaunchKernelGGL((gemm_nt_w128_kernel<EPI>), dim3(p.nwg), dim3(512), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int EPI>
int launch_w4(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / 256;
  p.tiles_m = cdiv(p.M, 256);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
  hipLaunchKernelGGL((gemm_nt_w4_kernel<EPI>), dim3(p.nwg), dim3(256), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int EPI, int WM, int WN, int NS>
int launch_ring(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / (64 * WN);
  p.tiles_m = cdiv(p.M, 64 * WM);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<EPI, WM, WN, NS>), dim3(p.nwg), dim3(64 * WM * WN), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int EPI>
int launch_w4x32(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / 256;
  p.tiles_m = cdiv(p.M, 256);
  p.nwg = 8 * cdiv(p.tiles_m, 8) * p.tiles_n;
  hipLaunchKernelGGL((gemm_nt_w4x32_kernel<EPI>), dim3(p.nwg), dim3(256), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int EPI>
int launch_nt(const GemmNT& p, hipStream_t s) {
  int t = g_force_tile;
  if (t == 10 && p.N % 256 == 0) return launch_w4x32<EPI>(p, s);
  if (t == 11 && p.N % 256 == 0) {
    GemmNT q = p;
    q.tiles_n = q.N / 256;
    q.tiles_m = cdiv(q.M, 256);
    q.nwg = 8 * cdiv(q.tiles_m, 8) * q.tiles_n;
    hipLaunchKernelGGL((gemm_nt_w8x32_kernel<EPI>), dim3(q.nwg), dim3(512), 0, s, q);
    PVRL_LAUNCH_CHECK();
    return PVRL_OK;
  }
  if (t == 13 && p.N % 256 == 0 && p.K % 64 == 0) {
    GemmNT q = p;
    q.tiles_n = q.N / 256;
    q.tiles_m = cdiv(q.M, 256);
    q.nwg = 8 * cdiv(q.tiles_m, 8) * q.tiles_n;
    hipLaunchKernelGGL((gemm_nt_rt32_kernel<EPI>), dim3(q.nwg), dim3(256), 0, s, q);
    PVRL_LAUNCH_CHECK();
    return PVRL_OK;
  }
  if (t == 12 && p.N % 256 == 0) {
    GemmNT q = p;
    q.tiles_n = q.N / 256;
    q.tiles_m = cdiv(q.M, 256);
    q.nwg = 8 * cdiv(q.tiles_m, 8) * q.tiles_n;
    hipLaunchKernelGGL((gemm_nt_w16x32_kernel<EPI>), dim3(q.nwg), dim3(1024), 0, s, q);
    PVRL_LAUNCH_CHECK();
    return PVRL_OK;
  }
  if (t == 7) return launch_ring<EPI, 4, 2, 2>(p, s);      // 256x128, 2 stages (48 KiB): 2-3 workgroups / CU
  if (t == 8) return launch_ring<EPI, 4, 2, 3>(p, s);      // 256x128, 3 stages (72 KiB): 2 workgroups / CU
  if (t == 9) return launch_ring<EPI, 2, 2, 4>(p, s);      // 128x128, 4 stages (64 KiB): 2 workgroups / CU
  if (t == 6 && p.N % 256 == 0) return launch_w4<EPI>(p, s);
  if (t == 4 && p.N % 256 == 0) return launch_pipe<EPI>(p, s);
  if (t == 5 && p.N % 256 == 0) return launch_w128<EPI>(p, s);
  if (t == 0) t = (p.M >= 4096 && p.N % 256 == 0) ? 3 : (p.M >= 2048 ? 2 : 1);
  if (t == 3 && p.N % 256) t = 2;
  // (cutting the ragged last wave of 256x256 tiles off into a 128x128-tile launch was measured 12 % SLOWER:
  //  the second launch serialises behind the first; one launch with a partly idle last wave wins)
  if (t == 3) return launch_tile<EPI, 4, 4>(p, s);
  if (t == 2) return launch_tile<EPI, 4, 2>(p, s);
  return launch_tile<EPI, 2, 2>(p, s);
}

}  // namespace

extern "C" int pvrl_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
                                 int64_t K, int epilogue, const float* bias, const float* rowscale,
                                 const void* aux, int64_t aux_ld, int64_t aux_rowmod, void* out0, int64_t ld0,
                                 void* out1, int64_t ld1, const float* bias2, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (bias2 && epilogue != PVRL_EPI_RESID_F32) return PVRL_EINVAL;
  if (!A || !W || !out0 || N <= 0 || K <= 0 || (N % 128) || (K % BK)) return PVRL_EINVAL;
  if ((lda % 8) || (ldw % 8) || (ld0 % 8)) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_GELU || epilogue == PVRL_EPI_QGELU) && (!out1 || (ld1 % 8))) return PVRL_EINVAL;
  if ((epilogue == PVRL_EPI_RESID_F32 || epilogue == PVRL_EPI_DGELU || epilogue == PVRL_EPI_DQGELU) &&
      (!aux || (aux_ld % 8)))
    return PVRL_EINVAL;
  GemmNT p;
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = bias; p.bias2 = bias2; p.rowscale = rowscale; p.aux = aux; p.aux_ld = aux_ld; p.aux_rowmod = (int)aux_rowmod;
  p.out0 = out0; p.ld0 = ld0; p.out1 = out1; p.ld1 = ld1; p.m_off = 0; p.gm = g_nt_gm;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case PVRL_EPI_BF16: return launch_nt<PVRL_EPI_BF16>(p, s);
    case PVRL_EPI_GELU: return launch_nt<PVRL_EPI_GELU>(p, s);
    case PVRL_EPI_QGELU: return launch_nt<PVRL_EPI_QGELU>(p, s);
    case PVRL_EPI_RESID_F32: return launch_nt<PVRL_EPI_RESID_F32>(p, s);
    case PVRL_EPI_F32: return launch_nt<PVRL_EPI_F32>(p, s);
    case PVRL_EPI_DGELU: return launch_nt<PVRL_EPI_DGELU>(p, s);
    case PVRL_EPI_DQGELU: return launch_nt<PVRL_EPI_DQGELU>(p, s);
    default: return PVRL_EINVAL;
  }
}

extern "C" int pvrl_debug_set_gemm_gm(int gm) {
  if (gm < 1) return PVRL_EINVAL;
  g_nt_gm = gm;
  return PVRL_OK;
}

extern "C" int pvrl_debug_set_gemm_tile(int tile) {
  g_force_tile = tile;
  return PVRL_OK;
}

extern "C" int64_t pvrl_gemm_nt_f32_small_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int splits = f32_small_plan(M, N, K, nullptr);
  return splits > 1 ? (int64_t)splits * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int pvrl_gemm_nt_f32_small(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                      float alpha, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0) return PVRL_OK;
  if (!A || !B || !C || K <= 0) return PVRL_EINVAL;
  int kchunk;
  const int splits = f32_small_plan(M, N, K, &kchunk);
  if (splits > 1 && (!workspace || workspace_bytes < pvrl_gemm_nt_f32_small_workspace_bytes(M, N, K))) return PVRL_EINVAL;
  dim3 grid(cdiv(N, SG_T), cdiv(M, SG_T), splits);
  hipLaunchKernelGGL(gemm_f32_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, (long)lda, B, (long)ldb,
                     bias, alpha, C, (long)ldc, (int)M, (int)N, (int)K, kchunk, (float*)workspace);
  PVRL_LAUNCH_CHECK();
  if (splits > 1) {
    const long MN = (long)M * N;
    hipLaunchKernelGGL(f32_small_reduce_kernel, dim3((unsigned)cdiv(MN, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, splits, MN, (int)N, bias, alpha, C, (long)ldc);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}
