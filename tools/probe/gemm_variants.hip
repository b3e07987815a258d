// MEASURED-AND-REJECTED GEMM variants -- benchmark probes, NOT part of libpvrl_hip.so.
//
// Every kernel here was A/B-timed against the product kernels on MI355X (numbers in the comments above each one and in
// DESIGN.md section 3) and lost or tied; they are kept, compiled into tools/probe/libpvrl_probe.so by
// tools/probe/build_probe.py, so the measurements can be repeated (tools/bench_kernels.py) and their results checked
// (tools/probe/check_variants.py).  They share the operand layouts, epilogues and reduce kernels of the product through
// procedurevrl_amd/csrc/gemm_nt_core.h / gemm_tn_core.h.  The kernel is chosen by an explicit `tile` argument: there is
// no process-global state.
//   NT tile: 0 = product heuristic, 1 = 128x128, 2 = 256x128, 3 = 256x256 (the three product tiles), 4 = 256x256 deep
//            pipeline, 5 = 8 waves x 128x64, 6 = 4 waves x 128x128, 7-9 = BK=32 rings, 10-13 = 32x32x16-MFMA forms.
//   TN tile: 0 / 8 = product (register-transposed, 8 waves), 1 = 128x128 transposing reads, 2 = the same with LDS-DMA
//            staging, 3 = 256x256 / 16 waves, 4 = 256x256 / 8 waves, 5 = LDS-DMA ring, 6 / 7 = register-transposed 4 waves
//            with 16x16x32 / 32x32x16 MFMAs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "../../procedurevrl_amd/csrc/gemm_nt_core.h"
#include "../../procedurevrl_amd/csrc/gemm_tn_core.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// PERSISTENT form of gemm_nt_kernel (probe tile 14; MEASURED round 2: 171 vs 128 us on the fp32-residual K = 768 shape,
// 452 vs 351 us dGELU -- the loop-carried state pushes the 128-VGPR kernel into scratch spills): one workgroup per CU walks its XCD's tile list with stride (workgroups per XCD)
// instead of one workgroup per tile.  Same LDS image, staging, K loop and epilogue; what changes is the seam between two
// tiles.  A one-tile workgroup pays, per tile: launch + LDS-DMA pipeline fill (first stage's HBM latency, nothing to
// overlap it with) at the front, and at the back its epilogue's loads / stores must DRAIN before the wave slots and the
// 128 KiB of LDS pass to the next workgroup.  Here the LDS-DMA of the NEXT tile's first stage is issued at the top of the
// LAST K step of the current tile (the other ring slot is free by then), so it is in flight under that step's MFMAs and
// under the whole epilogue, and the epilogue's stores drain under the next tile's first K steps.
// ---------------------------------------------------------------------------------------------------------
template <int EPI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_persist_kernel(GemmNT p) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int NINST = (BM + BN) / 8;
  constexpr int PER = NINST / NW;
  static_assert(NINST % NW == 0, "stage instructions must divide evenly over the waves");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int GM = p.gm;
  // this XCD's tile list (same order as gemm_nt_kernel): entry j -> (tm, tn)
  const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3;
  const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
  const int cm = qm + (xcd < rm ? 1 : 0);
  const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
  const int nlist = cm * p.tiles_n;
  const int gsz = GM * p.tiles_n;
  auto tile_of = [&](int j, int& tm, int& tn) {
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  };
  int j = blockIdx.x >> 3;
  if (j >= nlist) return;
  int tm, tn;
  tile_of(j, tm, tn);

  const op_t* gsrc[PER];
  auto set_src = [&](int m0, int n0) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int it = wave * PER + e;
      const int pc = lane & 7;
      if (it < BM / 8) {
        const int row = it * 8 + (lane >> 3);
        int grow = m0 + row;
        grow = grow < p.M ? grow : p.M - 1;
        gsrc[e] = p.A + (long)grow * p.lda + ((pc ^ swz_x(row)) << 3);
      } else {
        const int row = (it - BM / 8) * 8 + (lane >> 3);
        gsrc[e] = p.W + (long)(n0 + row) * p.ldw + ((pc ^ swz_w<F32OUT>(row)) << 3);
      }
    }
  };
  auto stage = [&](int buf, int k0) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int e = 0; e < PER; ++e) glds16(gsrc[e] + k0, b + (wave * PER + e) * 1024);
  };
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
  const int nk = p.K / BK;
  int slot = 0;                       // ring slot holding the stage the next K step consumes
  set_src(tm * BM, tn * BN);
  stage(0, 0);
  while (true) {
    const int m0 = tm * BM, n0 = tn * BN;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool more = j + nper < nlist;          // workgroup-uniform
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and fences the previous compute
      if (kt + 1 < nk) {
        stage(slot ^ 1, (kt + 1) * BK);
      } else if (more) {                         // the seam: the next tile's first stage goes out under this step + the epilogue
        tile_of(j + nper, tm, tn);
        set_src(tm * BM, tn * BN);
        stage(slot ^ 1, 0);
      }
      const char* bx = smem + slot * STAGE;
      const char* bw = bx + XBYTES;
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
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      slot ^= 1;
    }
    nt_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
    if (!more) break;
    j += nper;
  }
}

template <int EPI, int WM, int WN, int WPC = 1>
int launch_tile_persist(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / (64 * WN);
  p.tiles_m = cdiv(p.M, 64 * WM);
  const int per_xcd = cdiv(p.tiles_m, 8) * p.tiles_n;          // longest per-XCD tile list
  const int cus_per_xcd = 32 * WPC;                            // MI355X: 256 CUs in 8 XCDs, WPC resident workgroups per CU
  p.nwg = 8 * (per_xcd < cus_per_xcd ? per_xcd : cus_per_xcd);
  hipLaunchKernelGGL((gemm_nt_persist_kernel<EPI, WM, WN>), dim3(p.nwg), dim3(64 * WM * WN), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
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
      opx8 xf[8], wf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const opx8*>(b + (woff[t] ^ (ks << 6)));
#pragma unroll
      for (int t = 0; t < 8; ++t) xf[t] = *reinterpret_cast<const opx8*>(b + (xoff[t] ^ (ks << 6)));
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt >> 2][mt & 3][nt] =
              MFMA_16x16x32(wf[nt], xf[mt], acc[mt >> 2][mt & 3][nt], 0, 0, 0);
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
  const op_t* gx;
  const op_t* gw;
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
    opx8 xf[4], wf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = *reinterpret_cast<const opx8*>(b + woff[t]);
      xf[t] = *reinterpret_cast<const opx8*>(b + xoff[t]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = MFMA_16x16x32(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
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

  const op_t* gsrc[PER];
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
    opx8 xf[4], wf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = *reinterpret_cast<const opx8*>(b + woff[t]);
      xf[t] = *reinterpret_cast<const opx8*>(b + xoff[t]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = MFMA_16x16x32(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
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

  opx8 xf[7], xa[1], xb[1], wfa[8], wfb[8];
  auto rd = [&](const char* b, int off) { return *reinterpret_cast<const opx8*>(b + off); };
  // one K = 32 step: 8 rows (X tile r) of 8 MFMAs; LDS reads of the next stage are issued after rows 0..6 only
  auto step = [&](const opx8* wc, opx8* wnx, const opx8* xc7, opx8* xn7, const char* nb) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const opx8 xr = r < 7 ? xf[r] : xc7[0];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[r >> 2][t >> 2][r & 3][t & 3] =
            MFMA_16x16x32(wc[t], xr, acc[r >> 2][t >> 2][r & 3][t & 3], 0, 0, 0);
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
      opx8 o0;
#pragma unroll
      for (int e = 0; e < 8; ++e) o0[e] = (op_t)(rs * v[8 * c + e]);
      *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + n + 8 * c) = o0;
    }
  } else if constexpr (EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      opx8 u0, g0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u0[e] = (op_t)v[8 * c + e];
        g0[e] = (op_t)(EPI == PVRL_EPI_GELU ? gelu_erf(v[8 * c + e]) : quick_gelu(v[8 * c + e]));
      }
      *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + n + 8 * c) = u0;
      *reinterpret_cast<opx8*>((op_t*)p.out1 + (long)m * p.ld1 + n + 8 * c) = g0;
    }
  } else {   // PVRL_EPI_DGELU / PVRL_EPI_DQGELU : out = rs * acc * act'(u)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const opx8 ua = *reinterpret_cast<const opx8*>((const op_t*)p.aux + (long)m * p.aux_ld + n + 8 * c);
      opx8 o0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = EPI == PVRL_EPI_DGELU ? gelu_erf_grad((float)ua[e]) : quick_gelu_grad((float)ua[e]);
        o0[e] = (op_t)(rs * v[8 * c + e] * d);
      }
      *reinterpret_cast<opx8*>((op_t*)p.out0 + (long)m * p.ld0 + n + 8 * c) = o0;
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
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const opx8*>(b + (off ^ (u << 5))); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  opx8 xa[4], wa[4], xb[4], wb[4];
  auto mma = [&](const opx8* xf, const opx8* wf) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[mb][nb] = MFMA_32x32x16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
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
  auto rd = [&](const char* b, int off) { return *reinterpret_cast<const opx8*>(b + off); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  opx8 xa[4], wa[4], xb[4], wb[4];
  auto mma = [&](const opx8* xf, const opx8* wf) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[mb][nb] = MFMA_32x32x16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
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
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const opx8*>(b + (off ^ (u << 5))); };

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
      opx8 xf[4], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = rd(b, woff[t], u);
#pragma unroll
      for (int t = 0; t < 4; ++t) xf[t] = rd(b, xoff[t], u);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = MFMA_32x32x16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
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
  auto rd = [&](const char* b, int off, int u) { return *reinterpret_cast<const opx8*>(b + (off ^ (u << 5))); };

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
      opx8 xf[2], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[t] = rd(b, woff[t], u);
#pragma unroll
      for (int t = 0; t < 2; ++t) xf[t] = rd(b, xoff[t], u);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = MFMA_32x32x16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
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
int launch_nt(const GemmNT& p, hipStream_t s, int t) {
  if (t == 14 && p.N % 256 == 0) return launch_tile_persist<EPI, 4, 4>(p, s);      // persistent 256x256 (gemm_nt_core.h)
  if (t == 15) return launch_tile_persist<EPI, 2, 2, 2>(p, s);                       // persistent 128x128, two workgroups per CU
  if (t == 16) return launch_tile_persist<EPI, 2, 2, 4>(p, s);                       // the same, 4 lists per CU (2 resident at a time)
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

extern "C" int pvrl_probe_gemm_nt_bf16(int tile, int gm, const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
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
  p.A = (const op_t*)A; p.lda = lda; p.W = (const op_t*)W; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = bias; p.bias2 = bias2; p.rowscale = rowscale; p.aux = aux; p.aux_ld = aux_ld; p.aux_rowmod = (int)aux_rowmod;
  p.out0 = out0; p.ld0 = ld0; p.out1 = out1; p.ld1 = ld1; p.m_off = 0; p.gm = gm < 1 ? 2 : gm; p.cus = 32; p.tails = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case PVRL_EPI_BF16: return launch_nt<PVRL_EPI_BF16>(p, s, tile);
    case PVRL_EPI_GELU: return launch_nt<PVRL_EPI_GELU>(p, s, tile);
    case PVRL_EPI_QGELU: return launch_nt<PVRL_EPI_QGELU>(p, s, tile);
    case PVRL_EPI_RESID_F32: return launch_nt<PVRL_EPI_RESID_F32>(p, s, tile);
    case PVRL_EPI_F32: return launch_nt<PVRL_EPI_F32>(p, s, tile);
    case PVRL_EPI_DGELU: return launch_nt<PVRL_EPI_DGELU>(p, s, tile);
    case PVRL_EPI_DQGELU: return launch_nt<PVRL_EPI_DQGELU>(p, s, tile);
    default: return PVRL_EINVAL;
  }
}

// =========================================================================================================
// TN (weight-gradient) variants
// =========================================================================================================
namespace {

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
  const op_t* src[8];
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
      const op_t* g = ok ? src[e] + (long)st * sstep[e] : p.zero_page + (lane & 7) * 8;
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

  const op_t* src[8];
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
      const op_t* g = ok ? src[e] + (long)st * sstep[e] : p.zero_page + (lane & 7) * 8;
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
      opx8 pf[8], qf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t] = tr_frag(bk, qoff[t][0], qoff[t][1]);
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t] = tr_frag(bk, poff[t][0], poff[t][1]);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = MFMA_16x16x32(qf[kt], pf[nt], acc[nt][kt], 0, 0, 0);
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
  opx8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (op_t)1.0f;

  // P fragments: ONE set, refreshed in place for step s+1 as soon as their row of MFMAs of step s has issued;
  // Q fragments: two sets (all eight are live for the whole step).
  opx8 pf[8], qfa[8], qfb[8];
  auto step = [&](const opx8* qc, opx8* qn, const char* b) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        acc[nt][kt] = MFMA_16x16x32(qc[kt], pf[nt], acc[nt][kt], 0, 0, 0);
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
    for (int nt = 0; nt < 8; ++nt) cacc[nt] = MFMA_16x16x32(ones, pf[nt], cacc[nt], 0, 0, 0);
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
    return *reinterpret_cast<const opx8*>(slot + off + t * 256);
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of P, taken from the P fragments with v_dot2_f32_bf16 against (1, 1)
  float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  opx2 ones2;
  ones2[0] = (op_t)1.0f; ones2[1] = (op_t)1.0f;

  // P fragments 0-6 live in ONE register set, refreshed in place for the next stage right after their row of MFMAs;
  // P fragment 7 and all Q fragments are double-buffered, so the last LDS operation of a step is issued after row 6 and
  // row 7 (128 MFMA clocks) covers its latency in front of the barrier.
  opx8 pf[7], pa[1], pb[1], qfa[8], qfb[8];
  u32x4 ra[8], rb[8];
  // one step: MFMAs of stage st (qc, pf, pc) | fragments of stage st+1 from `rs` | transpose registers r -> slot `ws`
  auto step = [&](const opx8* qc, opx8* qn, const opx8* pc, opx8* pn, const char* rs, const u32x4* r, char* ws) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        acc[nt][kt] = MFMA_16x16x32(qc[kt], nt < 7 ? pf[nt] : pc[0], acc[nt][kt], 0, 0, 0);
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
  auto colsum = [&](const opx8* pc) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const opx8 f = nt < 7 ? pf[nt] : pc[0];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        cacc[nt] = FDOT2_F32((opx2){f[2 * d], f[2 * d + 1]}, ones2, cacc[nt], false);
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
  auto rfrag = [&](const char* slot, int off, int u) { return *reinterpret_cast<const opx8*>(slot + off + u * 8192); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a2][b2][e] = 0.f;
  float cacc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_csum = (p.cpart != nullptr) && (tk == 0) && (wk == 0);
  opx2 ones2;
  ones2[0] = (op_t)1.0f; ones2[1] = (op_t)1.0f;

  // fragments of the two K = 16 sub-steps of a stage: set A (sub-step 0) and set B (sub-step 1); while sub-step 0 of stage
  // s computes, sub-step 1's fragments are already in registers and the reads of stage s+1 refill the set that just finished
  opx8 pa[4], qa[4], pb[4], qb[4];
  u32x4 ra[8], rb[8];
  auto mma = [&](const opx8* pf, const opx8* qf) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        acc[nb][kb] = MFMA_32x32x16(qf[kb], pf[nb], acc[nb][kb], 0, 0, 0);
  };
  auto colsum = [&](const opx8* pf) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        cacc[nb] = FDOT2_F32((opx2){pf[nb][2 * d], pf[nb][2 * d + 1]}, ones2, cacc[nb], false);
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


// 0 = heuristic (register-transposed 256x256 kernel when N and K are multiples of 256, else 128x128 register-staged),
// 1 = 128x128 register-staged, 2 = 128x128 LDS-DMA staged, 3 = 256x256 / 16 waves, 4 = 256x256 / 8 waves,
// 5 = 256x256 LDS-DMA ring, 6 = 256x256 register-transposed 4 waves with 16x16x32 MFMAs, 7 = the same with 32x32x16 MFMAs,
// 8 = register-transposed 8 waves x 128x64 (= what the heuristic picks; benchmark / test knob)

bool tn_use_rt(int g_tn_tile, int64_t N, int64_t K) {
  if ((g_tn_tile == 0 || g_tn_tile == 8) && (N % 128 == 0) && (K % 128 == 0) && N * K >= 256 * 256)
    return true;     // the 8-wave kernel stages half tiles (N or K = 128 mod 256) with zero columns
  return (g_tn_tile == 6 || g_tn_tile == 7) && (N % 256 == 0) && (K % 256 == 0);
}

}  // namespace

extern "C" int64_t pvrl_probe_gemm_tn_plan_splits(int g_tn_tile, int64_t M, int64_t N, int64_t K) {
  if (N <= 0 || K <= 0 || (N % 128) || (K % 128)) return PVRL_EINVAL;
  if (tn_use_rt(g_tn_tile, N, K)) {
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

extern "C" int64_t pvrl_probe_gemm_tn_workspace_bytes(int64_t N, int64_t K, int64_t splits) {
  return splits * (N * K + N) * (int64_t)sizeof(float) + 256;   // + a zero page for out-of-range rows
}

extern "C" int pvrl_probe_gemm_tn_bf16(int g_tn_tile, const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N,
                                 int64_t K, int64_t splits, float beta, float* dW, float* dbias, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (!P || !Q || !dW || !workspace || N <= 0 || K <= 0 || (N % 128) || (K % 128) || splits < 1 || M < 0)
    return PVRL_EINVAL;
  const bool use_rt = tn_use_rt(g_tn_tile, N, K);
  if (!use_rt && (splits < 8 || (splits % 8))) return PVRL_EINVAL;   // slice s lives on XCD s % 8 in those kernels
  if ((ldp % 8) || (ldq % 8) || ((uintptr_t)P % 16) || ((uintptr_t)Q % 16)) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_probe_gemm_tn_workspace_bytes(N, K, splits)) return PVRL_EINVAL;
  GemmTN p;
  p.P = (const op_t*)P; p.ldp = ldp; p.Q = (const op_t*)Q; p.ldq = ldq;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  int ms = cdiv(M > 0 ? M : 1, splits);
  p.Ms = cdiv(ms, TM) * TM;
  p.part = (float*)workspace;
  p.cpart = dbias ? p.part + splits * N * K : nullptr;
  hipStream_t s = (hipStream_t)stream;
  char* zp = (char*)workspace + splits * (N * K + N) * (int64_t)sizeof(float);
  p.zero_page = (const op_t*)zp;
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
                     (int)splits, NK, (int)N, beta, dW, dbias, (const float*)nullptr, (float*)nullptr);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// grouped weight gradients
// ---------------------------------------------------------------------------------------------------------
namespace {
bool tn_group_ok(int g_tn_tile, int nprob, const pvrl_tn_problem* pr) {
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

extern "C" int64_t pvrl_probe_gemm_tn_grouped_plan_splits(int g_tn_tile, int nprob, const pvrl_tn_problem* problems) {
  if (!tn_group_ok(g_tn_tile, nprob, problems)) return PVRL_EINVAL;
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

extern "C" int64_t pvrl_probe_gemm_tn_grouped_workspace_bytes(int g_tn_tile, int nprob, const pvrl_tn_problem* problems, int64_t splits) {
  if (!tn_group_ok(g_tn_tile, nprob, problems) || splits < 1) return PVRL_EINVAL;
  int64_t b = 0;
  for (int i = 0; i < nprob; ++i) b += splits * (problems[i].N * problems[i].K + problems[i].N) * (int64_t)sizeof(float);
  return b;
}

extern "C" int pvrl_probe_gemm_tn_grouped_bf16(int g_tn_tile, int nprob, const pvrl_tn_problem* problems, int64_t splits, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (!tn_group_ok(g_tn_tile, nprob, problems) || splits < 1 || !workspace) return PVRL_EINVAL;
  if (workspace_bytes < pvrl_probe_gemm_tn_grouped_workspace_bytes(g_tn_tile, nprob, problems, splits)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  TnGroup g = {};
  g.nprob = nprob;
  float* w = (float*)workspace;
  int first = 0;
  for (int i = 0; i < nprob; ++i) {
    const pvrl_tn_problem& q = problems[i];
    GemmTN& p = g.prob[i];
    p.P = (const op_t*)q.P; p.ldp = q.ldp; p.Q = (const op_t*)q.Q; p.ldq = q.ldq;
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
