// bf16 MFMA GEMM, "NT" form, the encoder's 50k-row shapes:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )  with N % 256 == 0.
//
// PERSISTENT 256x256x64 kernel, EIGHT waves (two per SIMD), 256 registers per lane.  Same math, operand order, LDS swizzles and
// epilogues as gemm_nt_kernel<EPI, 4, 4> (gemm_nt_core.h) -- results are bit-identical -- with a different schedule:
//
//   * wave (wm, wn) = (wave >> 2, wave & 3) owns 128 rows x 64 columns of the tile = 8 x 4 v_mfma_f32_16x16x32 accumulators
//     (128 registers): the 8 waves read 192 KB of fragments per K = 64 step from LDS instead of the 256 KB of sixteen 64x64
//     waves.  Rows: [64 wm, +64) of the tile's upper half and [128 + 64 wm, +64) of its lower half; columns [32 wn, +32) and
//     [128 + 32 wn, +32): every wave touches all four 16 KiB half-tiles (A0 A1 B0 B1: 128 rows x 64 k) of a K-tile, one after
//     the other, so a half-tile's LDS is free -- and refilled -- a quarter of a K-tile after its first use.
//   * the two waves of a SIMD run PING-PONG: waves 0-3 (one per SIMD) are one barrier interval ahead of waves 4-7.  A K-tile is
//     two phases (the wave's upper / lower 64 rows, both k halves: 32 MFMAs each; four phases of 16 until late in round 4, same
//     results, 2.5-4.5 % slower); a phase is a MEMORY segment (fragment reads into registers, two or six 1 KiB LDS-DMA
//     instructions for the K-tile two ahead, the counted wait) and a COMPUTE segment (32 back-to-back MFMAs on registers only,
//     s_setprio 1), separated by workgroup barriers.  While one wave of a SIMD computes,
//     its partner is in its memory segment: the matrix pipe never waits for an LDS read.
//   * LDS-DMA is raw ISA (buffer_load_dwordx4 ... lds) with COUNTED waits: the loads of K-tile u + 2 go out during K-tile u
//     (into the slots K-tile u has just finished reading) and are waited for at the end of K-tile u + 1 with vmcnt(8), i.e. up to
//     64 KB per CU stay in flight across every barrier, each half-tile has at least two phases (~1 us) to land.
//   * the workgroup is PERSISTENT: one per CU walks its XCD's tile list (the order the one-tile kernel is dispatched in).  The
//     load stream runs two K-tiles ahead ACROSS tile seams, so a tile's first two K-tiles arrive under the previous tile's last
//     two and under its epilogue, and the epilogue's stores drain under the next tile's K loop (counted waits never ask for
//     them).  The one-tile kernel pays fill + drain per tile: ~8 us of a 29 us round at K = 768 (profiles/r3_nt_tail_subtiles.txt).
//   * the ragged last round (L = tiles mod CUs per XCD): when 2 L <= CUs, those tiles are cut into two 128-row halves, one
//     workgroup each: a half item is the upper half-tile A0 only, phases 0 and 1 of every K-tile, both waves of every SIMD busy.
//
// Hazards (wave groups G0 = waves 0-3, G1 = waves 4-7, G1 one barrier behind; every phase has a barrier after the memory and
// after the compute segment):
//   RAW  a half-tile is read in the phase AFTER the one whose memory segment waited for it (vmcnt) -- by then both groups have
//        passed a barrier behind their wait;
//   WAR  a half-tile is re-staged in the phase AFTER its last read, and every memory segment ends with lgkmcnt(0) BEFORE its
//        barrier: when G0 issues the DMA, G1's reads of the previous phase have completed behind that barrier.
// Replaces lib/models/vit.py:54-60,75-92,133 (nn.Linear forward + data gradient at the 50k-row shapes).
#pragma once
#include "gemm_nt_core.h"

namespace {

constexpr int NT8_HALF = 128 * BK * 2;   // one half-tile: 128 rows x 64 k x 2 B = 16 KiB
constexpr int NT8_BUF = 4 * NT8_HALF;    // one K-tile: A0 A1 B0 B1

// per-XCD work list: `full` whole tiles, then the L tiles of the ragged last round as 2 L half items (when they fit one round)
struct Nt8Plan { int full, L, nblk; };
__host__ __device__ inline Nt8Plan nt8_plan(int cm, int tiles_n, int cus, int enable) {
  Nt8Plan t;
  const int T = cm * tiles_n;
  t.full = T;
  t.L = 0;
  if (enable) {
    const int full = (T / cus) * cus, L = T - full;
    if (L > 0 && 2 * L <= cus) { t.full = full; t.L = L; }
  }
  t.nblk = t.full + 2 * t.L;
  return t;
}

// Probe builds only (-DPVRL_NT8_TRACE=1, tools/probe/nt8_ab.py trace): waves 0 and 4 of workgroup 8 stamp s_memtime at the four
// seams of every phase of their first tile's first NT8_TRACE_KT K-tiles into spare LDS and dump it through GemmNT::bias2.
#ifndef PVRL_NT8_TRACE
#define PVRL_NT8_TRACE 0
#endif
// Probe builds only (tools/probe/nt8_ab.py ablate; results are garbage, only the time means something): bit 0 = no LDS-DMA,
// bit 1 = no fragment reads, bit 2 = no MFMAs, bit 3 = no barrier behind the compute segments, bit 4 = no s_setprio, bit 5 = every K-tile
// loads K offset 0 (the same 64 KB per tile: L2-hot), bit 6 = every tile loads tile (0, 0) as well (one 64 KB image for the whole chip)
#ifndef PVRL_NT8_ABLATE
#define PVRL_NT8_ABLATE 0
#endif
#ifndef PVRL_NT8_PH2
#define PVRL_NT8_PH2 1      // 0: four phases of 16 MFMAs per K-tile (A/B builds)
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// One 1 KiB LDS-DMA copy through a buffer descriptor: lane l's 16 bytes at r.base + soff + voff land at LDS byte lds + 16 l.
// Offsets past the descriptor's size read as zero (the rows behind M of the ragged last panel).  Raw ISA: the compiler neither
// counts it nor waits for it.  (s_nop 4: a descriptor / offset register produced by v_readfirstlane needs five wait states before
// a vector-memory instruction reads it; s_nop 0: M0 write -> LDS-DMA.  Nothing inside an asm string is padded by hipcc.)
__device__ __forceinline__ void bdma16(rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  if (PVRL_NT8_ABLATE & 1) return;
  if (PVRL_NT8_ABLATE & 32) soff = 0;
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// row tiles whose residual / pre-activation loads go out together in the epilogue: four everywhere (two round trips per wave and
// tile) except the fp32-residual epilogue, whose 64 residual registers next to 128 accumulators, 16 bias values and the second bias
// do not fit 256 (measured by the compiler: 2 registers in scratch)
#ifndef PVRL_NT8_EB
#define PVRL_NT8_EB(EPI) ((EPI) == PVRL_EPI_RESID_F32 ? 2 : 4)
#endif

constexpr int NT8_TRACE_KT = 10;
#if PVRL_NT8_TRACE
#define NT8_STAMP(ph, k)                                                                                         \
  do {                                                                                                           \
    if (tracing && kt < NT8_TRACE_KT)                                                                            \
      reinterpret_cast<unsigned long long*>(smem + 2 * NT8_BUF)[(wm * NT8_TRACE_KT + kt) * 16 + (ph) * 4 + (k)] = \
          __builtin_readcyclecounter();                                                                          \
  } while (0)
#else
#define NT8_STAMP(ph, k) do { } while (0)
#endif

#define NT8_BARRIER()                       \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
// end of a memory segment: this wave's fragment reads have completed BEFORE the barrier (WAR rule above)
#define NT8_MEM_END(ph)                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    NT8_STAMP(ph, 1);                                     \
    NT8_BARRIER();                                        \
    NT8_STAMP(ph, 2);                                     \
    if (!(PVRL_NT8_ABLATE & 16)) __builtin_amdgcn_s_setprio(1); \
  } while (0)
#define NT8_MEM_END_Q()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    NT8_BARRIER();                                        \
    __builtin_amdgcn_s_setprio(1);                        \
  } while (0)
#define NT8_CMP_END_Q()               \
  do {                                \
    __builtin_amdgcn_s_setprio(0);    \
    NT8_BARRIER();                    \
  } while (0)
#define NT8_CMP_END(ph)                                          \
  do {                                                           \
    if (!(PVRL_NT8_ABLATE & 16)) __builtin_amdgcn_s_setprio(0);  \
    NT8_STAMP(ph, 3);                                            \
    if (!(PVRL_NT8_ABLATE & 8)) NT8_BARRIER();                   \
    else __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(GemmNT p) {
  constexpr bool F32OUT = (EPI == PVRL_EPI_RESID_F32 || EPI == PVRL_EPI_F32);
  constexpr int EB = PVRL_NT8_EB(EPI);     // row tiles per epilogue load batch
  // store instructions per wave of one whole tile's epilogue (bounds-checked buffer stores, no branches: every lane of every wave
  // issues all of them; tests/test_nt8_isa.py counts them in the built library)
  constexpr int NST = (F32OUT || EPI == PVRL_EPI_GELU || EPI == PVRL_EPI_QGELU) ? 32 : 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * NT8_BUF + (PVRL_NT8_TRACE ? 2 * NT8_TRACE_KT * 16 * 8 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int nk = p.K / BK;

  // ---- this workgroup's work list: entries j, j + stride, ... of its XCD's list (gemm_nt_kernel's rasterisation) ----
  const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
  const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
  const int cm = qm + (xcd < rm ? 1 : 0);                   // panels owned by this XCD
  const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
  const Nt8Plan plan = nt8_plan(cm, p.tiles_n, p.cus, p.tails);
  const int GM = p.gm, gsz = GM * p.tiles_n;
  auto decode = [&](int e, int& tm, int& tn) {
    const int g = e / gsz, r = e - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
    tn = __builtin_amdgcn_readfirstlane(tn);                // (uniform by construction; said explicitly so that every descriptor built
    tm = __builtin_amdgcn_readfirstlane(tm);                //  from them stays in scalar registers: no waterfall loops around buffer ops)
  };
  int j = blockIdx.x >> 3;
  if (j >= plan.nblk) return;

  // ---- per-lane constants ----
  // LDS-DMA instruction e (0, 1) of this wave copies rows 16 wave + 8 e + (lane >> 3) of a half-tile, chunk lane & 7 of the
  // LDS row holding source chunk (lane & 7) ^ swizzle(row)
  unsigned voffA[2][2], voffW[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 16 * wave + 8 * e + (lane >> 3), pc = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) voffA[h][e] = (unsigned)(128 * h + row) * (unsigned)p.lda * 2u + ((unsigned)(pc ^ swz_x(row)) << 4);
    voffW[e] = (unsigned)row * (unsigned)p.ldw * 2u + ((unsigned)(pc ^ swz_w<F32OUT>(row)) << 4);
  }
  const unsigned wstep = 128u * (unsigned)p.ldw * 2u;       // W half-tile 1 starts 128 rows further
  // fragment byte offsets inside a half-tile (k-half 0; k-half 1 is ^ 64): row r at 128 r, chunk ^ swizzle(r); both swizzles are
  // periodic over the rows one lane reads (x: r + 16 t; w: r + 4 h), so one register per operand
  const int q = lane >> 4, i = lane & 15;
  const int rx = 64 * wm + i, rw = 32 * wn + 8 * (i >> 2) + (i & 3);
  const int xb = rx * 128 + ((q ^ swz_x(rx)) << 4), wb = rw * 128 + ((q ^ swz_w<F32OUT>(rw)) << 4);

  auto srdA = [&](int m0_, int maxrows) {                   // rows [m0_, m0_ + maxrows) of A, clipped at M
    const int rows = min(maxrows, p.M - m0_);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)m0_ * p.lda), 0, rows * (int)p.lda * 2, 0x00020000);
  };
  auto srdW = [&](int n0_) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long)n0_ * p.ldw), 0, 256 * (int)p.ldw * 2, 0x00020000);
  };
  auto issueA1 = [&](int h, int e, unsigned lb, rsrc_t r, unsigned soff) {      // instruction e of this wave's two of A half-tile h
    bdma16(r, voffA[h][e], soff, lb + h * NT8_HALF + wave * 2048 + e * 1024);
  };
  auto issueW1 = [&](int c, int e, unsigned lb, rsrc_t r, unsigned soff) {
    bdma16(r, voffW[e], soff + c * wstep, lb + (2 + c) * NT8_HALF + wave * 2048 + e * 1024);
  };
  auto issueA = [&](int h, unsigned lb, rsrc_t r, unsigned soff) {
    bdma16(r, voffA[h][0], soff, lb + h * NT8_HALF + wave * 2048);
    bdma16(r, voffA[h][1], soff, lb + h * NT8_HALF + wave * 2048 + 1024);
  };
  auto issueW = [&](int c, unsigned lb, rsrc_t r, unsigned soff) {
    bdma16(r, voffW[0], soff + c * wstep, lb + (2 + c) * NT8_HALF + wave * 2048);
    bdma16(r, voffW[1], soff + c * wstep, lb + (2 + c) * NT8_HALF + wave * 2048 + 1024);
  };

  opx8 ra[4][2], rb0[2][2], rb1[2][2];                      // fragments: [row tile][k half], [h][k half] (rb0 / rb1: column group 0 / 1)
  f32x4 acc[2][4][4];                                       // [row half][row tile][nt = 2 c + h]
  // ---- whole tiles: a phase is (row half mh, k half ks) = 4 x 4 accumulators x K = 32 ----
  auto rdAk = [&](const char* buf, int mh, int ks) {        // 4 reads: the row half's 4 row tiles, k half ks
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (PVRL_NT8_ABLATE & 2) asm volatile("" : "+v"(ra[t][ks]));
      else ra[t][ks] = *reinterpret_cast<const opx8*>(buf + mh * NT8_HALF + t * 2048 + (xb ^ (ks * 64)));
    }
  };
  auto rdBk = [&](const char* buf, int ks) {                // 4 reads: both column groups, k half ks (kept for both row halves)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (PVRL_NT8_ABLATE & 2) {
        asm volatile("" : "+v"(rb0[h][ks]), "+v"(rb1[h][ks]));
      } else {
        rb0[h][ks] = *reinterpret_cast<const opx8*>(buf + 2 * NT8_HALF + h * 512 + (wb ^ (ks * 64)));
        rb1[h][ks] = *reinterpret_cast<const opx8*>(buf + 3 * NT8_HALF + h * 512 + (wb ^ (ks * 64)));
      }
    }
  };
  auto mmk = [&](f32x4 (&a)[4][4], int ks) {                // 16 MFMAs on 16 different accumulators
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (PVRL_NT8_ABLATE & 4) {
          asm volatile("" :: "v"(rb0[h][ks]), "v"(rb1[h][ks]), "v"(ra[t][ks]));
        } else {
          a[t][h] = MFMA_16x16x32(rb0[h][ks], ra[t][ks], a[t][h], 0, 0, 0);
          a[t][2 + h] = MFMA_16x16x32(rb1[h][ks], ra[t][ks], a[t][2 + h], 0, 0, 0);
        }
      }
  };
  // ---- half items: a phase is (column group c) x K = 64 of the upper row half ----
  auto rdA = [&](const char* buf, int mh) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ra[t][ks] = *reinterpret_cast<const opx8*>(buf + mh * NT8_HALF + t * 2048 + (xb ^ (ks * 64)));
  };
  auto rdB = [&](const char* buf, int c, opx8 (&rb)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) rb[h][ks] = *reinterpret_cast<const opx8*>(buf + (2 + c) * NT8_HALF + h * 512 + (wb ^ (ks * 64)));
  };
  // one accumulator quadrant x K = 64: 16 MFMAs, every accumulator's k order as in gemm_nt_kernel (k half 0, then 1)
  auto quad = [&](f32x4 (&a)[4][4], int c, const opx8 (&rb)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) a[t][2 * c + h] = MFMA_16x16x32(rb[h][ks], ra[t][ks], a[t][2 * c + h], 0, 0, 0);
  };
  if (PVRL_NT8_ABLATE & 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ra[t][ks] = (opx8)(op_t)0.5f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { rb0[h][ks] = (opx8)(op_t)0.25f; rb1[h][ks] = (opx8)(op_t)0.125f; }
  }
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

#if PVRL_NT8_TRACE
  bool tracing = blockIdx.x == 8 && (wave & 3) == 0;
#endif
  int par = 0;            // ring slot of the current item's K-tile 0
  bool primed = false;    // K-tiles 0 and 1 of the current item are already in the ring (issued under the previous tile; 0 has landed)

  // =========================================================== whole tiles
  while (j < plan.full) {
    int tm, tn;
    decode(j, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int jn = j + stride;
    const bool nxt = jn < plan.full;                        // the load stream continues into another whole tile
    int tmn = 0, tnn = 0;
    if (nxt) decode(jn, tmn, tnn);
    const rsrc_t cA = srdA((PVRL_NT8_ABLATE & 64) ? 0 : m0, 256), cW = srdW((PVRL_NT8_ABLATE & 64) ? 0 : n0);
    const rsrc_t nA = nxt ? srdA(tmn * 256, 256) : cA, nW = nxt ? srdW(tnn * 256) : cW;
    if (!primed) {                                          // cold start: K-tile 0, K-tile 1 without its A1; wait for K-tile 0
      const unsigned l0 = sbase + par * NT8_BUF, l1 = sbase + (par ^ 1) * NT8_BUF;
      issueW(0, l0, cW, 0); issueA(0, l0, cA, 0); issueW(1, l0, cW, 0); issueA(1, l0, cA, 0);
      issueW(0, l1, cW, 128); issueA(0, l1, cA, 128); issueW(1, l1, cW, 128);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      NT8_BARRIER();
    }
    zero_acc();
    if (wm == 1) NT8_BARRIER();                             // G1 runs one barrier interval behind G0
#if PVRL_NT8_TRACE
    unsigned long long tr_c0 = 0, tr_r0 = 0;
    if (tracing) { tr_c0 = __builtin_readcyclecounter(); tr_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int kt = 0; kt < nk; ++kt) {
      // The load stream, two cursors: the lower-row half-tile A1 of K-tile kt + 1 (its slot is free behind phase 3 of K-tile kt - 1)
      // and A0 / B0 / B1 of K-tile kt + 2 (this K-tile's slot, free behind phase 1) -- of this tile or, past its end, of the next.
      const int k1 = kt + 1, k2 = kt + 2;
      const bool in1 = k1 < nk, in2 = k2 < nk;
      const bool on1 = in1 || nxt, on2 = in2 || nxt;
      const rsrc_t a1 = in1 ? cA : nA, a2 = in2 ? cA : nA, w2 = in2 ? cW : nW;
      const unsigned so1 = (unsigned)(in1 ? k1 : k1 - nk) * 128u, so2 = (unsigned)(in2 ? k2 : k2 - nk) * 128u;
      const int cur = (par + kt) & 1;
      const char* rbuf = smem + cur * NT8_BUF;
      const unsigned lb = sbase + cur * NT8_BUF, lo = sbase + (cur ^ 1) * NT8_BUF;
      const bool seam = primed && kt == 0;                  // the previous epilogue's NST stores sit in the queue behind the prefetch
      // Every memory segment issues its fragment reads FIRST and the LDS-DMA behind them: a DMA instruction blocks its wave while the
      // CU's address path takes the 1 KiB (16 cycles each, the group's four waves queue up), and the reads complete underneath.
#if PVRL_NT8_PH2
      // TWO phases per K-tile (row half, both k halves: 32 MFMAs each): half the barriers of the four-phase form below, the same copies,
      // waits and hazards (a half-tile is re-staged in the phase after its last read, read in the phase after its wait); 2.5-4.5 % faster on
      // every shape, bit-identical (profiles/r4_nt8_ablation.txt section 7)
      rdAk(rbuf, 0, 0); rdBk(rbuf, 0); rdAk(rbuf, 0, 1); rdBk(rbuf, 1);
      if (on1) {
        issueA1(1, 0, lo, a1, so1); issueA1(1, 1, lo, a1, so1);
        if (seam) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      NT8_MEM_END(1);
      mmk(acc[0], 0);
      mmk(acc[0], 1);
      NT8_CMP_END(1);
      rdAk(rbuf, 1, 0); rdAk(rbuf, 1, 1);
      if (on2) {
        issueW1(0, 0, lb, w2, so2); issueW1(0, 1, lb, w2, so2); issueA1(0, 0, lb, a2, so2);
        issueA1(0, 1, lb, a2, so2); issueW1(1, 0, lb, w2, so2); issueW1(1, 1, lb, w2, so2);
        if (seam) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      NT8_MEM_END(3);
      mmk(acc[1], 0);
      mmk(acc[1], 1);
      NT8_CMP_END(3);
#else
      // ---- phase 0: rows 0, k half 0 ----
      NT8_STAMP(0, 0);
      rdAk(rbuf, 0, 0);
      rdBk(rbuf, 0);
      if (on1) issueA1(1, 0, lo, a1, so1);
      NT8_MEM_END(0);
      mmk(acc[0], 0);
      NT8_CMP_END(0);
      // ---- phase 1: rows 0, k half 1; behind it this slot's A0 / B0 / B1 are free.  Waits for this K-tile's A1 (read in phase 2) ----
      NT8_STAMP(1, 0);
      rdAk(rbuf, 0, 1);
      rdBk(rbuf, 1);
      if (on1) {
        issueA1(1, 1, lo, a1, so1);
        // younger than this K-tile's A1: 3 + 3 instructions of phases 2 / 3 of the previous K-tile, 1 + 1 of this one
        if (seam) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      NT8_MEM_END(1);
      mmk(acc[0], 1);
      NT8_CMP_END(1);
      // ---- phase 2: rows 1, k half 0 ----
      NT8_STAMP(2, 0);
      rdAk(rbuf, 1, 0);
      if (on2) { issueW1(0, 0, lb, w2, so2); issueW1(0, 1, lb, w2, so2); issueA1(0, 0, lb, a2, so2); }
      NT8_MEM_END(2);
      mmk(acc[1], 0);
      NT8_CMP_END(2);
      // ---- phase 3: rows 1, k half 1; behind it this slot's A1 is free.  Waits for the next K-tile's A0 / B0 / B1 ----
      NT8_STAMP(3, 0);
      rdAk(rbuf, 1, 1);
      if (on2) {
        issueA1(0, 1, lb, a2, so2); issueW1(1, 0, lb, w2, so2); issueW1(1, 1, lb, w2, so2);
        // younger than the next K-tile's A0 / B0 / B1: 1 + 1 + 3 + 3 instructions of this K-tile
        if (seam) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      NT8_MEM_END(3);
      mmk(acc[1], 1);
      NT8_CMP_END(3);
    #endif
    }
    if (wm == 0) NT8_BARRIER();                             // G0 waits for G1's last compute segment: both groups run the epilogue together
#if PVRL_NT8_TRACE
    if (tracing) {
      unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias2));
      if (lane == 0) {       // shader cycles and 100 MHz ticks across the K loop: the clock the kernel ran at
        out[2 * NT8_TRACE_KT * 16 + 2 * wm] = __builtin_readcyclecounter() - tr_c0;
        out[2 * NT8_TRACE_KT * 16 + 2 * wm + 1] = __builtin_amdgcn_s_memrealtime() - tr_r0;
      }
      for (int e = lane; e < NT8_TRACE_KT * 16; e += 64)
        out[wm * NT8_TRACE_KT * 16 + e] = reinterpret_cast<unsigned long long*>(smem + 2 * NT8_BUF)[wm * NT8_TRACE_KT * 16 + e];
      tracing = false;
    }
#endif
    nt_epilogue_at<EPI, EB>(p, acc[0], m0, n0, 64 * wm, 32 * wn, 128 + 32 * wn, lane);
    nt_epilogue_at<EPI, EB>(p, acc[1], m0, n0, 128 + 64 * wm, 32 * wn, 128 + 32 * wn, lane);
    par = (par + nk) & 1;
    primed = nxt;
    j = jn;
  }

  // =========================================================== one half item of the ragged last round (128 rows x 256 columns)
  if (j < plan.nblk) {
    const int s = j - plan.full;
    int tm, tn;
    decode(plan.full + (s >> 1), tm, tn);
    const int m0 = tm * 256 + (s & 1) * 128, n0 = tn * 256;
    if (m0 >= p.M) return;                                  // (workgroup-uniform: the half behind the ragged end of the last panel)
    const rsrc_t lsA = srdA(m0, 128), lsW = srdW(n0);
    // Only the upper half-tile A0 and phases 0 / 1.  Two slots: K-tile u + 2's B0 / A0 go out in phase 1 of K-tile u (last read in
    // phase 0), its B1 in phase 0 of K-tile u + 1 (last read in phase 1 of u); phase 1 waits for all of K-tile u + 1.
    {
      const unsigned l0 = sbase + par * NT8_BUF, l1 = sbase + (par ^ 1) * NT8_BUF;
      issueW(0, l0, lsW, 0); issueA(0, l0, lsA, 0); issueW(1, l0, lsW, 0);
      if (nk > 1) { issueW(0, l1, lsW, 128); issueA(0, l1, lsA, 128); }
      if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NT8_BARRIER();
    }
    zero_acc();
    if (wm == 1) NT8_BARRIER();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = (par + kt) & 1;
      const char* rbuf = smem + cur * NT8_BUF;
      const unsigned lb = sbase + cur * NT8_BUF, lo = sbase + (cur ^ 1) * NT8_BUF;
      // ---- phase 0 ----
      if (kt + 1 < nk) issueW(1, lo, lsW, (unsigned)(kt + 1) * 128u);        // B1 of the other slot: last read in phase 1 of K-tile - 1
      rdA(rbuf, 0);
      rdB(rbuf, 0, rb0);
      NT8_MEM_END_Q();
      quad(acc[0], 0, rb0);
      NT8_CMP_END_Q();
      // ---- phase 1 ----
      if (kt + 2 < nk) {
        issueW(0, lb, lsW, (unsigned)(kt + 2) * 128u);
        issueA(0, lb, lsA, (unsigned)(kt + 2) * 128u);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // everything but K-tile + 2's four instructions
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      rdB(rbuf, 1, rb1);
      NT8_MEM_END_Q();
      quad(acc[0], 1, rb1);
      NT8_CMP_END_Q();
    }
    if (wm == 0) NT8_BARRIER();
    nt_epilogue_at<EPI, EB>(p, acc[0], m0, n0, 64 * wm, 32 * wn, 128 + 32 * wn, lane);
  }
}

template <int EPI>
int launch_nt8(GemmNT p, hipStream_t s) {
  p.tiles_n = p.N / 256;
  p.tiles_m = cdiv(p.M, 256);
  if (p.gm <= 0) p.gm = nt_gm_for(p.tiles_n);
  const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
  const int nb = nt8_plan(qm + (rm ? 1 : 0), p.tiles_n, p.cus, p.tails).nblk;      // the longest per-XCD list
  p.nwg = 8 * std::min(nb, p.cus);                                                  // one persistent workgroup per CU
  hipLaunchKernelGGL((gemm_nt8_kernel<EPI>), dim3(p.nwg), dim3(512), 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace
