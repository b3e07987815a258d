// Spatial-attention backward as ONE persistent kernel: dQ, dK and dV of a (sequence, head) from one evaluation of P and dS,
// the next item's operands already on their way while the current one computes.
//
// Reference semantics: autograd of Attention.forward, lib/models/vit.py:75-92, on the spatial sequences of
// Block.forward (vit.py:137-151; 197 tokens per (clip, frame)).  The two-pass form (attn_bwd_q_kernel + attn_bwd_kv_kernel,
// attn_mfma.hip) evaluates S = Q K^T, dP = dO V^T, the exponentials and the dS arithmetic twice and reads q, k, v, dO twice.
// Here one 8-wave workgroup per CU walks its items; per item the Q, dO and K head slices sit in LDS (3 x 28 KB), the K / V rows
// of a wave's own keys in registers, and the queries are walked in blocks of 32:
//
//   key waves (wave w owns keys 32w .. 32w+31; 7 waves at S = 197), per query block
//     S^T, dP^T   32 queries x 32 keys each, 4 + 4 v_mfma_f32_32x32x16 (a = Q / dO rows from LDS, b = K / V from registers);
//                 the accumulators START at -lse/scale and -D*scale, so P = exp2(c * acc) and dS = P * acc need no subtraction
//     dV^T += dO^T P, dK^T += Q^T dS    (reduction over the block's queries = the MFMA's k: P / dS feed the b operand straight
//                 from registers, a = transposed fragments of the same LDS images through ds_read_b64_tr_b16)
//     dS -> LDS   bf16, [key][query] image of the block (14 KB, two buffers)
//   one barrier per block (LDS traffic only -- no vmcnt drain), then
//   the dQ wave (the wave that owns no keys), one block behind the key waves:
//     dQ^T block = K^T dS^T over ALL keys (the reduction crosses the key waves, hence the LDS exchange): two 32 x 32 output
//                 tiles x 14 k-steps of v_mfma_f32_32x32x16 (half the fragment reads of the 16 x 16 form: the kernel is bound by
//                 LDS read bandwidth, ~2k cycles per block traced), fragments of step u + 1 in flight under the MFMAs of step u.
//
// Hiding the loads (traced: a workgroup that loads, then computes, spends 17k of its 44k cycles in the load burst, and the
// one-workgroup-per-CU turn-around costs another ~4 us per item): the workgroup is persistent and streams item i + 1 in while item i
// computes, by LDS-DMA (no registers).  A block's 32 rows of the Q and dO images are dead once every key wave has passed that
// block's barrier, so from inside block jb + 1 (under its first MFMAs) waves 0-4 send item i + 1's rows 32jb .. 32jb+31 of Q and dO
// into the freed slots, the same rows of O into a three-slot ring and their lse next to it (13 copies of 1 KB: three on each of
// waves 0-3, the lse on wave 4).  Each sender waits `vmcnt(<its copies per slot>)` before a barrier -- all but its newest slot's
// copies have landed -- so a slot is complete two barriers after the block it replaces, and at the top of the block after that
// waves 0-3 (a quarter slot each) read dO and O back, compute D = rowsum(dO * O) and write the start values -lse/scale and
// -D*scale of those rows (the last three slots of an item are finished at the seam and during its own first two blocks).  The K
// image of item i + 1 goes into the second K buffer (four pieces per key wave at the top of item i), its V rows into 16 registers per
// lane.  Outputs leave as whole 128-byte rows: dQ through the block's own dS^T buffer (the dQ wave is its only reader), dK / dV
// through each wave's 2 KB of the buffer the last block did not use.  `scale` must be a power of two (1/8 for head_dim 64): it is
// folded into the V operand and into D, both exactly.  The DMA cannot zero padding rows (they hold copies of the last row): rows
// past the sequence start at -inf, so P = 0 there without a compare, and the wave that owns keys past the sequence zeroes their
// dS.  The first item of a workgroup has nothing to hide behind: all eight waves
// compute its D straight from global memory.  156 KB of LDS, one workgroup per CU.
#include "attn_stream.h"
#include "../../include/pvrl.h"

namespace {

#ifndef PVRL_FB_TRACE
#define PVRL_FB_TRACE 0       // probe builds only: every wave of workgroup 8 stamps the cycle counter at the seams of each block of its
#endif                        // SECOND item into spare LDS; dumped through AttnArgs::dvec (tools/probe/attn_bwd_ab.py trace)
#ifndef PVRL_FB_PRIO7
#define PVRL_FB_PRIO7 3
#endif
#ifndef PVRL_FB_PRIO46
#define PVRL_FB_PRIO46 1
#endif
#ifndef PVRL_FB_ABLATE
#define PVRL_FB_ABLATE 0      // probe builds only (results are garbage, only the dQ wave's time means something): 1 = no start values of streamed
#endif                        // slots, 2 = no dQ staging / stores, 4 = no dQ MFMAs, 8 = no dQ fragment reads
constexpr int FB_DS = FB_ROWS * 64;          // dS^T image of one query block: [224 keys][32 queries]
constexpr int FB_Q = 0, FB_D = FB_TILE, FB_K = 2 * FB_TILE, FB_S = 4 * FB_TILE, FB_INIT = 4 * FB_TILE + 2 * FB_DS;
constexpr int FB_O = FB_INIT + 4 * FB_ROWS * 4;         // {-lse/scale, -D*scale} x two items, then the O ring: 3 slots of 32 rows
constexpr int FB_LSE = FB_O + 3 * 4096;                 // lse of the ring's rows: 3 x 64 floats
constexpr int FB_LDS = FB_LSE + 3 * 256;

#if PVRL_FB_TRACE
#define FB_STAMP(jb, k)                                                                                                   \
  do {                                                                                                                    \
    if (tracing) reinterpret_cast<unsigned*>(smem + FB_LDS)[(wave * 8 + (jb)) * 8 + (k)] = (unsigned)__builtin_readcyclecounter(); \
  } while (0)
#define FB_TRACE_DUMP()                                                                                                    \
  do {                                                                                                                     \
    if (tracing && p.dvec) {                                                                                               \
      FB_STAMP(7, 4);                                                                                                      \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
      reinterpret_cast<unsigned*>(p.dvec)[wave * 64 + lane] = reinterpret_cast<unsigned*>(smem + FB_LDS)[wave * 64 + lane]; \
    }                                                                                                                      \
  } while (0)
#else
#define FB_STAMP(jb, k) do { } while (0)
#define FB_TRACE_DUMP() do { } while (0)
#endif

// One of the 13 copies that bring rows 32jb .. 32jb+31 of the next item in: o = 0..3 Q pieces, 4..7 dO pieces (both into the freed
// slots of the images), 8..11 O pieces and 12 the rows' lse (ring slot rs).  `o` is wave-uniform and fixed per wave, so everything
// that does not depend on the block is computed once (FbCopy): r0 = the lane's row within the block, col its column chunk.
struct FbCopy { int o, r0, col; };
__device__ __forceinline__ FbCopy fb_copy_init(int o, int lane) {
  FbCopy cp;
  cp.o = o;
  const int j = o & 3;                                   // piece 4 jb + j: band index rb = 8 jb + 2 j + (lane >> 5)
  const int band = lane >> 5;
  cp.r0 = 8 * j + 4 * band + ((lane >> 1) & 3);
  const int cb = ((lane >> 3) & 3) ^ band, half = (lane & 1) ^ (j & 1);
  cp.col = 16 * cb + 8 * half;
  if (o >= 12) { cp.r0 = lane & 31; cp.col = 0; }
  return cp;
}
__device__ __forceinline__ void fb_slot_op(const AttnArgs& p, const FbItem& it, int jb, const FbCopy& cp, int rs, int S, unsigned ldsbase) {
  const int rc = min(32 * jb + cp.r0, S - 1);
  // row index in 32 bits (checked on the host), one v_mad_u64_u32 per address
  const unsigned rowi = rc == 0 ? (unsigned)it.sr.base0 : (unsigned)it.sr.base1 + (unsigned)(rc - 1) * (unsigned)it.sr.stride;
  if (cp.o >= 12) {
    const float* lp = p.lse + ((long)it.seq * p.H + it.h) * S + rc;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" :: "v"(lp), "s"(ldsbase + FB_LSE + rs * 256) : "memory", "m0");
    return;
  }
  const int j = cp.o & 3, typ = cp.o >> 2;
  const int hc = it.h * 64 + cp.col;
  if (typ == 0) {
    glds16_raw_v(p.qkv + (unsigned long long)rowi * (unsigned)p.ld + hc, ldsbase + FB_Q + (4 * jb + j) * 1024);
  } else {
    const op_t* tok = typ == 1 ? p.d_o : p.ofw;
    const op_t* cls = typ == 1 ? p.d_o_cls : p.ofw_cls;
    const op_t* a = tok + (unsigned long long)rowi * (unsigned)p.ldo;
    const op_t* b = cls + (long)it.seq * p.ldo;
    const op_t* src = (p.mp.mode == 1 && rc == 0) ? b : a;
    glds16_raw_v(src + hc, ldsbase + (typ == 1 ? FB_D + (4 * jb + j) * 1024 : FB_O + (4 * rs + j) * 1024));
  }
}
// ... and two barriers later: D of those rows from the two images, start values into that item's arrays.
// Done by waves 0-3, a quarter slot (8 rows, one chunk pair per lane) each, at the top of the block that follows the barrier: they
// reach the barriers ~1,000 cycles before the dQ wave does (traced), which is where this work used to sit.
__device__ __forceinline__ void fb_finalize_quarter(const char* smem, int jb, int rs, int qtr, int S, float scale, int lane, float* si) {
  const int pidx = 64 * qtr + lane;
  const int r = pidx >> 3, c8 = (pidx & 7) * 8, row = 32 * jb + r;
  const u32x4 d = *reinterpret_cast<const u32x4*>(smem + FB_D + fb_off(row, c8));
  const u32x4 o = *reinterpret_cast<const u32x4*>(smem + FB_O + rs * 4096 + fb_off(r, c8));
  const float l = reinterpret_cast<const float*>(smem + FB_LSE)[rs * 64 + r];
  float dsum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a0, a1, b0, b1;
    op_unpack2(d[e], a0, a1);
    op_unpack2(o[e], b0, b1);
    dsum = fmaf(a0, b0, dsum);
    dsum = fmaf(a1, b1, dsum);
  }
  dsum = fb_sum8(dsum);
  if ((pidx & 7) == 0) {
    const bool keep = row < S;
    si[row] = keep ? -l * (1.0f / scale) : -INFINITY;      // 1 / scale is exact: a power of two
    si[FB_ROWS + row] = keep ? -dsum * scale : 0.f;
  }
}
// the first item of a workgroup: D and the start values of chunk pair `pidx` straight from global memory
__device__ __forceinline__ void fb_prime_pair(const AttnArgs& p, const FbItem& it, int pidx, int S, float* si) {
  const int row = pidx >> 3, c8 = (pidx & 7) * 8;
  const int rc = min(row, S - 1);
  const u32x4 d = *reinterpret_cast<const u32x4*>(fb_tok(p.d_o, p.d_o_cls, p.ldo, p, it, rc) + it.h * 64 + c8);
  const u32x4 o = *reinterpret_cast<const u32x4*>(fb_tok(p.ofw, p.ofw_cls, p.ldo, p, it, rc) + it.h * 64 + c8);
  const float l = p.lse[((long)it.seq * p.H + it.h) * S + rc];
  float dsum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a0, a1, b0, b1;
    op_unpack2(d[e], a0, a1);
    op_unpack2(o[e], b0, b1);
    dsum = fmaf(a0, b0, dsum);
    dsum = fmaf(a1, b1, dsum);
  }
  dsum = fb_sum8(dsum);
  if ((pidx & 7) == 0 && row < FB_ROWS) {
    const bool keep = row < S;
    si[row] = keep ? -l / p.scale : -INFINITY;
    si[FB_ROWS + row] = keep ? -dsum * p.scale : 0.f;
  }
}

// dQ^T of one 32-query block = K^T dS^T over all keys: two 32 (columns) x 32 (queries) tiles, 2 NQB (or 2 nqb) k-steps of 16 keys.
template <int NQB>
__device__ __forceinline__ void dq_block(const AttnArgs& p, const FbItem& it, const char* Kb, const char* dsr, int nqb, int S,
                                         int jb, int lane) {
  const int g = lane >> 5, hi = (lane >> 4) & 1, i = lane & 15;
  f32x16 acc[2];
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dh][r] = 0.f;
  // lane parts of the transposed-fragment addresses: keys 16 ks + 8 g + 4 e2 + {0..3}
  const int inb = (i >> 2) * 32 + ((8 * (i & 3)) ^ (16 * g)) + g * 1024;
  const int a0 = inb + hi * 128, a1 = inb + 512 + (hi ^ 1) * 128;                         // K image, e2 = 0 / 1 (+256 for dh = 1)
  const int b0 = (2 * g) * 256 + hi * 128 + (i >> 2) * 32 + (((i & 3) ^ (2 * g)) * 8);      // dS^T image
  const int b1 = (2 * g + 1) * 256 + hi * 128 + (i >> 2) * 32 + (((i & 3) ^ (2 * g + 1)) * 8);
  auto load = [&](int ks, opx8& bf, opx8* af) {
    if (PVRL_FB_ABLATE & 8) { bf = (opx8){}; af[0] = af[1] = (opx8){}; return; }
    bf = tr_frag8(dsr + ks * 1024, b0, b1);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) af[dh] = tr_frag8(Kb + ks * 2048 + dh * 256, a0, a1);
  };
  if constexpr (NQB > 0) {
    // fragments run TWO steps ahead of the MFMAs (three register sets): a read queues behind the key waves' traffic for longer than one
    // step's two MFMAs last
    opx8 bf[3], af[3][2];
    load(0, bf[0], af[0]);
    load(1, bf[1], af[1]);
#pragma unroll
    for (int ks = 0; ks < 2 * NQB; ++ks) {
      if (ks + 2 < 2 * NQB) load(ks + 2, bf[(ks + 2) % 3], af[(ks + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {
        if (PVRL_FB_ABLATE & 4) { acc[dh][0] += (float)af[ks % 3][dh][0] + (float)bf[ks % 3][1]; continue; }
        acc[dh] = MFMA_32x32x16(af[ks % 3][dh], bf[ks % 3], acc[dh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll 1
    for (int ks = 0; ks < 2 * nqb; ++ks) {
      opx8 bf, af[2];
      load(ks, bf, af);
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) acc[dh] = MFMA_32x32x16(af[dh], bf, acc[dh], 0, 0, 0);
    }
  }
  // The accumulators hold 4 consecutive columns of one query per register quad: stored from here, one instruction would touch 64
  // different 128-byte lines (traced: ~300 cycles of issue each, and the address pipe clogs for every other wave).  This wave is the
  // only reader of the block's dS^T buffer and is done with it: transpose through it ([query][64 columns], 16-byte chunks swizzled by
  // the query) and store whole 128-byte rows, 8 rows per instruction.
  char* st = const_cast<char*>(dsr);
  if (PVRL_FB_ABLATE & 2) { if (acc[0][0] + acc[1][0] == 12345.f) st[lane] = 1; return; }
  {
    const int q = lane & 31;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        opx4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (op_t)acc[dh][4 * j + r];
        *reinterpret_cast<opx4*>(st + q * 128 + (((4 * dh + j) ^ (q & 7)) * 16) + 8 * g) = ov;
      }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = lane + 64 * t;
    const int r = idx >> 3, ch = idx & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(st + r * 128 + ((ch ^ (r & 7)) * 16));
    const int query = 32 * jb + r;
    if (query < S) *reinterpret_cast<u32x4*>(fb_tok(p.dqkv, p.dqkv_cls, p.ldd, p, it, query) + it.h * 64 + ch * 8) = v;
  }
}

template <int NQB>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnArgs p, int nvb) {
  __shared__ __attribute__((aligned(16))) char smem[FB_LDS + (PVRL_FB_TRACE ? 8 * 8 * 8 * 4 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.mp.S;
  const int nqb = NQB > 0 ? NQB : (S + 31) >> 5;
  const int HD = p.H * 64;
  const int n = lane & 31, g = lane >> 5;
  const bool keywave = wave < nqb;
  const float c = p.scale * 1.4426950408889634f;
  const unsigned ldsbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
#if PVRL_FB_TRACE
  bool tracing = false;
#endif

  FbItem cur, nxt;
  int vcur = -1;
  int vnxt = fb_next(p, blockIdx.x, gridDim.x, nvb, nxt);
  int par = 0;                      // K buffer / start-value arrays of the CURRENT item; the next item's are par ^ 1
  int nit = -1;                     // index of the current item in this workgroup's list (-1: priming)
  if (vnxt < 0) return;

  // ---- priming, every wave: D and the start values of the first item straight from global memory (its images travel by DMA below)
  {
    float* si = reinterpret_cast<float*>(smem + FB_INIT) + (par ^ 1) * 2 * FB_ROWS;
#pragma unroll
    for (int t = 0; t < 4; ++t) fb_prime_pair(p, nxt, tid + 512 * t, S, si);
  }

  if (wave == 7) {
    // =========================================================== the dQ wave (and the start values of the streamed slots)
    // It is the youngest wave of its SIMD and the critical path of a block: without priority its VALU work (start values, store
    // addresses) only gets the issue slots its partner leaves (MI355X_MICROARCH: arbitration by priority, then age).
    __builtin_amdgcn_s_setprio(PVRL_FB_PRIO7);
    while (true) {
      if (vcur >= 0) {
#pragma unroll 1
        for (int jb = 0; jb < nqb; ++jb) {
          FB_STAMP(jb, 0);
          FB_BARRIER();
          FB_STAMP(jb, 1);
          FB_STAMP(jb, 2);
          dq_block<NQB>(p, cur, smem + FB_K + par * FB_TILE, smem + FB_S + (jb & 1) * FB_DS, nqb, S, jb, lane);
          FB_STAMP(jb, 3);
        }
      }
      FB_TRACE_DUMP();
      if (vnxt < 0) break;
      FB_BARRIER();                  // item seam
      par ^= 1; ++nit;
      cur = nxt; vcur = vnxt;
      vnxt = fb_next(p, vcur + gridDim.x, gridDim.x, nvb, nxt);
#if PVRL_FB_TRACE
      tracing = blockIdx.x == 8 && nit == 1;
      FB_STAMP(7, 0);
#endif
    }
    return;
  }

  // =========================================================== key waves (waves >= nqb own no keys: they only move data)
  // lane parts of the LDS addresses inside one 32-row query block (4096 bytes of a tile)
  const int rbl = n >> 2, b0 = rbl & 1, b1 = (rbl >> 1) & 1;
  const int rowbase = rbl * 512 + (n & 3) * 32 + ((16 * g) ^ (16 * b1));
  const int e0 = rowbase + b0 * 128, e1 = rowbase + (1 - b0) * 128;      // column step s even / odd (+256 for s >= 2)
  const int i16 = lane & 15, hi = (lane >> 4) & 1;
  const int trb = g * 512 + (hi ^ g) * 128 + (i16 >> 2) * 32;
  const int tr0 = trb + 8 * (i16 & 3), tr1 = trb + 1024 + ((8 * (i16 & 3)) ^ 16);
  const int kbl = rbl & 3;
  const int dsw0 = (8 * wave + rbl) * 256 + (n & 3) * 32;
  const bool lastkw = wave == nqb - 1 && (S & 31) != 0;       // this wave owns keys past the sequence
  const float keepf = (32 * wave + n) < S ? 1.f : 0.f;
  // The 13 copies of a slot: three on each of waves 0-3 (a Q, a dO and an O piece), the lse on wave 4.  Other splits were traced
  // (two per key wave; Q pieces on the dQ wave and the rest on waves 4-6): the block period stays within 3 % -- it is the sum of the
  // waves' LDS / address-pipe work that bounds it, not who issues it.  `ncp` is also the vmcnt a wave allows in flight at a barrier.
  const int ncp = wave < 4 ? 3 : (wave == 4 ? 1 : 0);
  const FbCopy cp1 = fb_copy_init(wave < 4 ? wave : 12, lane), cp2 = fb_copy_init(4 + (wave & 3), lane), cp3 = fb_copy_init(8 + (wave & 3), lane);
  if (wave >= 4) __builtin_amdgcn_s_setprio(PVRL_FB_PRIO46);      // the younger wave of each SIMD pair
  opx8 kf[4], vf[4];
  f32x16 dk[2], dvv[2];
  while (true) {
    // ---- the next item's K image by LDS-DMA (this wave's 4 of the 28 pieces), the V rows of this wave's keys into registers
    opx8 vfn[4];
    if (vnxt >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = wave + 7 * j;
        int rc, col;
        fb_piece_src(k, lane, S, rc, col);
        glds16_raw_v(p.qkv + row_of(nxt.sr, rc) * p.ld + HD + nxt.h * 64 + col, ldsbase + FB_K + (par ^ 1) * FB_TILE + k * 1024);
      }
      const int key = min(32 * wave + n, S - 1);
      const op_t* vp = p.qkv + row_of(nxt.sr, key) * p.ld + 2 * HD + nxt.h * 64 + 8 * g;
#pragma unroll
      for (int s = 0; s < 4; ++s) vfn[s] = *reinterpret_cast<const opx8*>(vp + 16 * s);
    }

    if (vcur >= 0) {
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dh][r] = 0.f; dvv[dh][r] = 0.f; }
      const float* sinit = reinterpret_cast<const float*>(smem + FB_INIT) + par * 2 * FB_ROWS;
      const float* dinit = sinit + FB_ROWS;
#pragma unroll 1
      for (int jb = 0; jb < nqb; ++jb) {
        char* dsj = smem + FB_S + (jb & 1) * FB_DS;
        FB_STAMP(jb, 0);
        // start values of a streamed slot: the one sent three blocks ago has landed (every sender waited for it before the last barrier)
        if (wave < 4 && !(PVRL_FB_ABLATE & 1)) {
          if (jb >= 3) {
            if (vnxt >= 0)
              fb_finalize_quarter(smem, jb - 3, (nqb * nit + jb - 3) % 3, wave, S, p.scale, lane,
                                  reinterpret_cast<float*>(smem + FB_INIT) + (par ^ 1) * 2 * FB_ROWS);
          } else if (jb >= 1 && nit >= 1) {      // the last two slots of THIS item, sent during the previous one
            fb_finalize_quarter(smem, nqb - 3 + jb, (nqb * (nit - 1) + nqb - 3 + jb) % 3, wave, S, p.scale, lane,
                                reinterpret_cast<float*>(smem + FB_INIT) + par * 2 * FB_ROWS);
          }
        }
        if (keywave) {
          const char* Qj = smem + FB_Q + jb * 4096;
          const char* Dj = smem + FB_D + jb * 4096;
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
          // rows of block jb - 1 are dead since the last barrier: the next item's rows move in, addressed and issued under the MFMAs above
          if (vnxt >= 0 && jb >= 1) {
            const int rs = (nqb * nit + jb - 1) % 3;
            if (ncp >= 1) fb_slot_op(p, nxt, jb - 1, cp1, rs, S, ldsbase);
            if (ncp >= 3) { fb_slot_op(p, nxt, jb - 1, cp2, rs, S, ldsbase); fb_slot_op(p, nxt, jb - 1, cp3, rs, S, ldsbase); }
          }
          opx8 pf[2], sf[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(c * sacc[r]);
            float dsv = pr * dacc[r];
            if (lastkw) dsv *= keepf;
            pf[r >> 3][r & 7] = (op_t)pr;
            sf[r >> 3][r & 7] = (op_t)dsv;
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
        if (!keywave && vnxt >= 0 && jb >= 1) {
          const int rs = (nqb * nit + jb - 1) % 3;
          if (ncp >= 1) fb_slot_op(p, nxt, jb - 1, cp1, rs, S, ldsbase);
          if (ncp >= 3) { fb_slot_op(p, nxt, jb - 1, cp2, rs, S, ldsbase); fb_slot_op(p, nxt, jb - 1, cp3, rs, S, ldsbase); }
        }
        FB_STAMP(jb, 3);
        // this wave's copies of the block before the last have landed (at most the newest slot's are in flight)
        if (ncp >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (ncp >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FB_BARRIER();
        FB_STAMP(jb, 4);
      }
      // start values of slot nqb - 3 (the next item's block loop finishes nqb - 2 and nqb - 1)
      if (wave < 4 && vnxt >= 0 && !(PVRL_FB_ABLATE & 1))
        fb_finalize_quarter(smem, nqb - 3, (nqb * nit + nqb - 3) % 3, wave, S, p.scale, lane,
                            reinterpret_cast<float*>(smem + FB_INIT) + (par ^ 1) * 2 * FB_ROWS);
      // the last block's rows (the other slots were sent from inside the following block, under its first MFMAs)
      if (vnxt >= 0) {
        const int rs = (nqb * nit + nqb - 1) % 3;
        if (ncp >= 1) fb_slot_op(p, nxt, nqb - 1, cp1, rs, S, ldsbase);
        if (ncp >= 3) { fb_slot_op(p, nxt, nqb - 1, cp2, rs, S, ldsbase); fb_slot_op(p, nxt, nqb - 1, cp3, rs, S, ldsbase); }
      }
    } else {
      // priming: the first item's Q and dO images, all slots at once (8 pieces per slot, 56 in all: 8 per wave)
#pragma unroll 1
      for (int q = 0; q < 8; ++q) fb_slot_op(p, nxt, wave, fb_copy_init(q, lane), 0, S, ldsbase);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- item seam
    if (vcur >= 0 && keywave) {
      // dK^T / dV^T hold 4 consecutive columns of one key per register quad: transposed through this wave's 2 KB of the dS^T buffer
      // the last block did NOT use (the dQ wave reads the other one), 16 keys x 64 columns per round, whole rows out
      char* st = smem + FB_S + (nqb & 1) * FB_DS + wave * 2048;
#pragma unroll
      for (int rnd = 0; rnd < 4; ++rnd) {
        const int half = rnd & 1;
        const f32x16* src = (rnd >> 1) ? dvv : dk;
        if ((n >> 4) == half) {
          const int q = n & 15;
#pragma unroll
          for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              opx4 ov;
#pragma unroll
              for (int r = 0; r < 4; ++r) ov[r] = (op_t)src[dh][4 * j + r];
              *reinterpret_cast<opx4*>(st + q * 128 + (((4 * dh + j) ^ (q & 7)) * 16) + 8 * g) = ov;
            }
        }
        // (hipcc 7.2 otherwise sinks the first read below INTO the predicated block above -- in front of the exec restore -- and the
        //  lanes outside the mask keep the previous round's data: found by the parity test, pinned by this fence)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int idx = lane + 64 * t;
          const int r = idx >> 3, ch = idx & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(st + r * 128 + ((ch ^ (r & 7)) * 16));
          const int key = 32 * wave + 16 * half + r;
          if (key < S)
            *reinterpret_cast<u32x4*>(fb_tok(p.dqkv, p.dqkv_cls, p.ldd, p, cur, key) + ((rnd >> 1) ? 2 * HD : HD) + cur.h * 64 + ch * 8) = v;
        }
      }
    }
    FB_TRACE_DUMP();
    if (vnxt < 0) break;
    FB_BARRIER();                    // item seam (all eight waves)
    par ^= 1; ++nit;
    cur = nxt; vcur = vnxt;
    vnxt = fb_next(p, vcur + gridDim.x, gridDim.x, nvb, nxt);
#if PVRL_FB_TRACE
    tracing = blockIdx.x == 8 && nit == 1;
    FB_STAMP(7, 0);
#endif
    // this wave's K rows from the fresh image (a row fragment per column step), its V rows from the prefetch registers
    {
      const char* Kc = smem + FB_K + par * FB_TILE + wave * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const opx8*>(Kc + ((s & 1) ? e1 : e0) + 256 * (s >> 1));
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[s][e] = (op_t)((float)vfn[s][e] * p.scale);     // exact: scale is a power of two
      }
    }
  }
}

}  // namespace

// true when the fused kernel covers the case (attn_mfma.hip falls back to the two-pass kernels otherwise)
bool pvrl_attn_bwd_fused_ok(const AttnArgs& p) {
  if (p.causal || p.kpm) return false;
  if (p.mp.S <= 96 || p.mp.S > FB_ROWS) return false;      // at least four query blocks (the tail slots of an item are finished during its first two)
  if ((long)p.nseq * p.mp.S >= (1L << 31)) return false;      // row indices are formed in 32 bits
  if ((p.ldd % 8) || (p.ldo % 8)) return false;               // dQ / dK / dV rows leave as 16-byte stores, Q / dO / O rows arrive by 16-byte LDS-DMA
  int e = 0;
  const float m = frexpf(p.scale, &e);
  return m == 0.5f;      // power of two
}

int pvrl_attn_bwd_fused_launch(const AttnArgs& p, hipStream_t s) {
  const int cus = 8 * pvrl_compute_cus_per_xcd();       // one persistent workgroup per CU (common.h: PVRL_COMPUTE_CUS leaves some free)
  const int nvb = 8 * ((p.nseq + 7) / 8) * p.H;
  const int grid = nvb < cus ? nvb : cus;                   // a multiple of 8: a workgroup's items stay on its XCD
  if (p.mp.S > 192) hipLaunchKernelGGL(attn_bwd_fused_kernel<7>, dim3(grid), dim3(512), 0, s, p, nvb);      // 7 query blocks, loops unrolled
  else hipLaunchKernelGGL(attn_bwd_fused_kernel<0>, dim3(grid), dim3(512), 0, s, p, nvb);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
