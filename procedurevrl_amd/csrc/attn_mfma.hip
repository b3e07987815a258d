// Multi-head self-attention (head_dim 64) on MFMA, forward + backward, S <= 416 tokens (tuned for S <= 208; the reference takes any
// crop through its pos-embed resize, lib/models/vit.py:374-386 -- 256^2 gives 257 tokens, 320^2 gives 401).
//
// Reference semantics: Attention.forward, lib/models/vit.py:75-92
//     attn = softmax((q @ k^T) * scale);  x = attn @ v
// used for the spatial branch of Block.forward (vit.py:137-151; 197 tokens per (clip, frame)),
// and nn.MultiheadAttention inside ResidualAttentionBlock (lib/models/tfm_model.py:32-53; key padding
// mask) / the CLIP text tower (causal mask).  Backward = autograd of the same expression.
//
// One workgroup (4 waves) per (sequence, head); every kernel keeps <= 57 KiB of LDS so two workgroups share a CU and
// one's tile loads overlap the other's MFMA work.  The whole K and V head slices of a sequence fit in LDS
// (197 x 64 bf16 = 25 KiB each), so softmax is exact single-pass (no online rescale) and the S x S
// score matrix never exists in memory (the reference materialises 477 MB of it).
// MFMA operands are arranged "swapped" (a = keys, b = queries) so that a lane owns ONE query and
// 4 keys per 16-key tile: the softmax row reduction is register-local plus two cross-lane
// steps, and P feeds the second MFMA straight from registers as its b-operand.  V (and, in the
// backward, K / Q / dO) is read through ds_read_b64_tr_b16 from a [4][16]-blocked LDS image, so
// no transposed copies are ever built.
// Round 2 measured a PERSISTENT form of the forward kernel (one workgroup per CU walking its (sequence, head) items, K / V in two
// LDS slots filled by LDS-DMA one item ahead, counted vmcnt so output stores are not waited for; 8 waves sharing each K / V
// fragment between two query tiles, or 16 waves): 121-127 us and 157 us against 100 us for this kernel at 3,072 items.  The
// kernel is not latency-bound: an item costs ~8 us of CU time whatever hides its loads -- 700 MFMAs (1.4 us), 43 k exp2
// (1.3 us at quarter rate), ~6 VALU operations per score (2 us) and 650 KB of LDS fragment reads (3-5 us) that 8-16 waves
// do not overlap with each other -- 12 items per CU = the 96 us measured (3.3 TB/s is a consequence, not the limit).
// The number of 16-key tiles NKT is a template parameter (1, 2, 3, 5, 13): every loop over keys is
// straight-line code the compiler can software-pipeline (a run-time tile count cost 5x in branches).
#include "attn_common.h"
#include "../../include/pvrl.h"
#include <stdlib.h>

// attn_bwd_fused.hip: the persistent, LDS-DMA-streamed one-kernel backward for 96 < S <= 224 without masks
bool pvrl_attn_bwd_fused_ok(const AttnArgs& p);
int pvrl_attn_bwd_fused_launch(const AttnArgs& p, hipStream_t s);
// attn_bwd_s32.hip: one wave per (sequence, head) for 16 < S <= 32 contiguous tokens without masks (temporal attention at 32 frames)
bool pvrl_attn_bwd_s32_ok(const AttnArgs& p);
int pvrl_attn_bwd_s32_launch(const AttnArgs& p, hipStream_t s);

namespace {

// PVRL_ATTN_BWD_FUSED=0 sends every case back to the two-pass kernels (A/B runs); read once
int attn_bwd_fused_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_ATTN_BWD_FUSED");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

// key-padding bits of the 4*NKT keys a lane owns in the "lane = query" layout (key = 16 kt + 4 q4 + r)
template <int NKT, bool GEN>
__device__ __forceinline__ unsigned long long pad_bits_q(const AttnArgs& p, int seq, int q4) {
  unsigned long long bits = 0ull;
  if constexpr (GEN) {
    if (p.kpm) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * q4 + r;
          if (key < p.mp.S && p.kpm[(long)seq * p.mp.S + key]) bits |= 1ull << (kt * 4 + r);
        }
    }
  }
  return bits;
}

// Cooperative load of one head slice [S rows][64] of a packed activation into the blocked LDS image, in two halves so a
// kernel can put EVERY global load it needs (both tiles + its waves' own fragments) in flight before the first wait.
template <int NT, int ROWS = ATT_ROWS_PAD>
struct TileRegs { u32x4 v[(ROWS * 8 + NT - 1) / NT]; };

template <int NT, int ROWS = ATT_ROWS_PAD>
__device__ __forceinline__ void tile_issue(TileRegs<NT, ROWS>& t, const op_t* base, long ld, int col0, const SeqRows& sr, int S,
                                           const op_t* src0, int rows, int tid) {
  constexpr int ITERS = (ROWS * 8 + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + NT * it;
    const int row = idx >> 3, c = idx & 7;
    // BRANCH-FREE: every lane loads (rows past the sequence re-read its last row) and the padding is zeroed by a select.  A load
    // inside `if (row < S)` is followed by s_waitcnt vmcnt(0) at the join: the two guarded iterations of
    // this loop cost the workgroup two extra serial memory round trips in front of its first barrier.
#if defined(PVRL_ATTN_GUARDED_LOADS)      // A/B builds only: the round-1 form
    t.v[it] = (u32x4){0u, 0u, 0u, 0u};
    if (row < S && row < rows) {
      const op_t* src = (row == 0 && src0) ? src0 : base + row_of(sr, row) * ld;
      t.v[it] = *reinterpret_cast<const u32x4*>(src + col0 + c * 8);
    }
#else
    const int rc = min(row, S - 1);
    const op_t* src = (rc == 0 && src0) ? src0 : base + row_of(sr, rc) * ld;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + col0 + c * 8);
    const unsigned keep = row < S ? 0xffffffffu : 0u;
    t.v[it] = v & (u32x4){keep, keep, keep, keep};
    (void)rows;
#endif
  }
}
template <int NT, int ROWS = ATT_ROWS_PAD>
__device__ __forceinline__ void tile_commit(const TileRegs<NT, ROWS>& t, char* bl, int rows, int tid) {
  constexpr int ITERS = (ROWS * 8 + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + NT * it;
    const int row = idx >> 3, c = idx & 7;
    if (row < rows) *reinterpret_cast<u32x4*>(bl + bl_off(row, c * 8)) = t.v[it];
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NKT, bool GEN, int NW, int SC = 0>
__global__ __launch_bounds__(64 * NW, NKT > 13 ? 2 : (NW + 1) / 2) void attn_fwd_kernel(AttnArgs p) {
  constexpr int NKS2 = (NKT + 1) / 2;
  constexpr int BL = NKS2 * 32 * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * BL];
  char* Kb = smem;
  char* Vb = smem + BL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt round-robin to the 8 XCDs: keep the H heads of a sequence on ONE XCD, back to back, so the
  // 128-byte head slices of a token row are fetched together (same DRAM pages, same L2)
  const int xj = blockIdx.x >> 3;
  const int seq = (xj / p.H) * 8 + (blockIdx.x & 7), h = xj % p.H;
  if (seq >= p.nseq) return;
  const int S = SC ? SC : p.mp.S;     // SC: the sequence length as a compile-time constant (0 = run time)
  const int HD = p.H * 64;
  const SeqRows sr = seq_rows(p.mp, seq);
  constexpr int MAXT = (NKT + NW - 1) / NW;   // query tiles per wave
  const int q4 = lane >> 4, i = lane & 15;
  const int nqt = (S + 15) >> 4;
  // every global load of the workgroup goes out before the first wait: this wave's query fragments, then K and V
  opx8 qf[MAXT][2];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int query = (wave + t * NW) * 16 + i;
    const int qrow = query < S ? query : S - 1;
    const op_t* qp = p.qkv + row_of(sr, qrow) * p.ld + h * 64 + q4 * 8;
    qf[t][0] = *reinterpret_cast<const opx8*>(qp);
    qf[t][1] = *reinterpret_cast<const opx8*>(qp + 32);
  }
  {
    TileRegs<64 * NW, NKS2 * 32> kr, vr;
    tile_issue<64 * NW, NKS2 * 32>(kr, p.qkv, p.ld, HD + h * 64, sr, S, nullptr, NKS2 * 32, tid);
    tile_issue<64 * NW, NKS2 * 32>(vr, p.qkv, p.ld, 2 * HD + h * 64, sr, S, nullptr, NKS2 * 32, tid);
    tile_commit<64 * NW, NKS2 * 32>(kr, Kb, NKS2 * 32, tid);
    tile_commit<64 * NW, NKS2 * 32>(vr, Vb, NKS2 * 32, tid);
  }
  __syncthreads();

  const unsigned long long pbits = pad_bits_q<NKT, GEN>(p, seq, q4);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int qt = wave + t * NW;
    if (qt >= nqt) break;
    const int query = qt * 16 + i;
    const opx8 qf0 = qf[t][0], qf1 = qf[t][1];

    f32x4 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int krow = kt * 16 + i;
      const opx8 k0 = bl_row_frag(Kb, krow, q4);
      const opx8 k1 = bl_row_frag(Kb, krow, 4 + q4);
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
      a = MFMA_16x16x32(k0, qf0, a, 0, 0, 0);
      sc[kt] = MFMA_16x16x32(k1, qf1, a, 0, 0, 0);
      if (NKT > 5 && (kt & 1)) __builtin_amdgcn_sched_barrier(0);   // keep the fragment look-ahead (and VGPRs) bounded
    }
    // softmax over the lane's 4*NKT keys.  exp(scale*s - max) is evaluated as exp2(fma(s, c, -max*c)) with
    // c = scale*log2(e) (one FMA + one v_exp per element); P stays un-normalised (<= 1) and the 16 output values are
    // scaled by 1/sum instead of the 4*NKT probabilities.  Only tiles that can hold masked keys pay for the select.
    const float c = p.scale * 1.4426950408889634f;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * q4 + r;
        if (GEN || kt * 16 + 15 >= S) {
          bool msk = key >= S;
          if constexpr (GEN) msk = msk || ((pbits >> (kt * 4 + r)) & 1ull) || (p.causal && key > query);
          if (msk) sc[kt][r] = -INFINITY;
        }
        mx = fmaxf(mx, sc[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mref = (mx == -INFINITY) ? 0.f : mx;      // raw-score maximum (scale > 0)
    const float mc = mref * c;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c, -mc));
        sc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks2 = 0; ks2 < NKS2; ++ks2) {
      opx8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (op_t)sc[2 * ks2][r];
        if (2 * ks2 + 1 < NKT) pf[4 + r] = (op_t)sc[(2 * ks2 + 1 < NKT) ? 2 * ks2 + 1 : 0][r];
        else pf[4 + r] = (op_t)0.f;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const opx8 vf = bl_frag(Vb, ks2, dt, lane);
        oacc[dt] = MFMA_16x16x32(vf, pf, oacc[dt], 0, 0, 0);
      }
      if (NKT > 5) __builtin_amdgcn_sched_barrier(0);
    }
    if (query < S) {
      op_t* op = tok_ptr(p.o, p.o_cls, p.ldo, p.mp, sr, seq, query) + h * 64 + 4 * q4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        opx4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (op_t)(oacc[dt][r] * inv);
        *reinterpret_cast<opx4*>(op + 16 * dt) = ov;
      }
      if (q4 == 0 && p.lse) p.lse[((long)seq * p.H + h) * S + query] = mref * p.scale + __logf(sum);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, pass 1: dQ (and D = rowsum(dO * O)); waves own query tiles exactly as in the forward.
// ------------------------------------------------------------------------------------------
template <int NKT, bool GEN, int NW, int SC = 0>
__global__ __launch_bounds__(64 * NW, NKT > 13 ? 2 : (NW + 1) / 2) void attn_bwd_q_kernel(AttnArgs p) {
  constexpr int NKS2 = (NKT + 1) / 2;
  constexpr int BL = NKS2 * 32 * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * BL];
  char* Kb = smem;
  char* Vb = smem + BL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt round-robin to the 8 XCDs: keep the H heads of a sequence on ONE XCD, back to back, so the
  // 128-byte head slices of a token row are fetched together (same DRAM pages, same L2)
  const int xj = blockIdx.x >> 3;
  const int seq = (xj / p.H) * 8 + (blockIdx.x & 7), h = xj % p.H;
  if (seq >= p.nseq) return;
  const int S = SC ? SC : p.mp.S;     // SC: the sequence length as a compile-time constant (0 = run time)
  const int HD = p.H * 64;
  const SeqRows sr = seq_rows(p.mp, seq);
  constexpr int MAXT = (NKT + NW - 1) / NW;   // query tiles per wave
  const int q4 = lane >> 4, i = lane & 15;
  const int nqt = (S + 15) >> 4;
  const float c = p.scale * 1.4426950408889634f;
  // every global load of the workgroup goes out before the first wait: q / dO / O / lse of this wave's query tiles,
  // then the K and V head slices
  opx8 qf[MAXT][2], df[MAXT][2];
  float lse2[MAXT], dss[MAXT];
  {
    opx8 of[MAXT][2];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int query = (wave + t * NW) * 16 + i;
      const int qj = query < S ? query : S - 1;
      const op_t* qp = p.qkv + row_of(sr, qj) * p.ld + h * 64 + q4 * 8;
      qf[t][0] = *reinterpret_cast<const opx8*>(qp);
      qf[t][1] = *reinterpret_cast<const opx8*>(qp + 32);
      const op_t* dop = tok_ptr(p.d_o, p.d_o_cls, p.ldo, p.mp, sr, seq, qj) + h * 64 + q4 * 8;
      df[t][0] = *reinterpret_cast<const opx8*>(dop);
      df[t][1] = *reinterpret_cast<const opx8*>(dop + 32);
      const op_t* ofp = tok_ptr(p.ofw, p.ofw_cls, p.ldo, p.mp, sr, seq, qj) + h * 64 + q4 * 8;
      of[t][0] = *reinterpret_cast<const opx8*>(ofp);
      of[t][1] = *reinterpret_cast<const opx8*>(ofp + 32);
      lse2[t] = p.lse[((long)seq * p.H + h) * S + qj] * 1.4426950408889634f;
    }
    TileRegs<64 * NW, NKS2 * 32> kr, vr;
    tile_issue<64 * NW, NKS2 * 32>(kr, p.qkv, p.ld, HD + h * 64, sr, S, nullptr, NKS2 * 32, tid);
    tile_issue<64 * NW, NKS2 * 32>(vr, p.qkv, p.ld, 2 * HD + h * 64, sr, S, nullptr, NKS2 * 32, tid);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += (float)df[t][0][e] * (float)of[t][0][e] + (float)df[t][1][e] * (float)of[t][1][e];
      dsum += __shfl_xor(dsum, 16, 64);
      dsum += __shfl_xor(dsum, 32, 64);
      dss[t] = dsum * p.scale;
      const int query = (wave + t * NW) * 16 + i;
      if (q4 == 0 && query < S) p.dvec[((long)seq * p.H + h) * S + query] = dsum;
    }
    tile_commit<64 * NW, NKS2 * 32>(kr, Kb, NKS2 * 32, tid);
    tile_commit<64 * NW, NKS2 * 32>(vr, Vb, NKS2 * 32, tid);
  }
  __syncthreads();

  const unsigned long long pbits = pad_bits_q<NKT, GEN>(p, seq, q4);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int qt = wave + t * NW;
    if (qt >= nqt) break;
    const int query = qt * 16 + i;
    const opx8 qf0 = qf[t][0], qf1 = qf[t][1], df0 = df[t][0], df1 = df[t][1];
    const float lse2_t = lse2[t], dss_t = dss[t];

    // dS for two 16-key tiles at a time, consumed at once by the dQ MFMAs: nothing but dq[] lives across iterations
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks2 = 0; ks2 < NKS2; ++ks2) {
      opx8 sf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * ks2 + half;
        if (kt >= NKT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sf[4 * half + r] = (op_t)0.f;
          continue;
        }
        const int krow = kt * 16 + i;
        const opx8 k0 = bl_row_frag(Kb, krow, q4);
        const opx8 k1 = bl_row_frag(Kb, krow, 4 + q4);
        const opx8 v0 = bl_row_frag(Vb, krow, q4);
        const opx8 v1 = bl_row_frag(Vb, krow, 4 + q4);
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        s = MFMA_16x16x32(k0, qf0, s, 0, 0, 0);
        s = MFMA_16x16x32(k1, qf1, s, 0, 0, 0);
        f32x4 dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp = MFMA_16x16x32(v0, df0, dp, 0, 0, 0);
        dp = MFMA_16x16x32(v1, df1, dp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * q4 + r;
          float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse2_t));
          // (a padded query, query >= S, needs no mask: its lane's dS only feeds that lane's own dq, which is never stored)
          if (GEN || kt * 16 + 15 >= S) {
            bool msk = key >= S || (GEN && query >= S);
            if constexpr (GEN) msk = msk || ((pbits >> (kt * 4 + r)) & 1ull) || (p.causal && key > query);
            if (msk) pr = 0.f;
          }
          sf[4 * half + r] = (op_t)(pr * fmaf(dp[r], p.scale, -dss_t));
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const opx8 kf = bl_frag(Kb, ks2, dt, lane);
        dq[dt] = MFMA_16x16x32(kf, sf, dq[dt], 0, 0, 0);
      }
      if (NKT > 5) __builtin_amdgcn_sched_barrier(0);
    }
    if (query < S) {
      op_t* op = tok_ptr(p.dqkv, p.dqkv_cls, p.ldd, p.mp, sr, seq, query) + h * 64 + 4 * q4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        opx4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (op_t)dq[dt][r];
        *reinterpret_cast<opx4*>(op + 16 * dt) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, pass 2: dK and dV.  Waves own KEY tiles; the query dimension is the MFMA reduction.
// Needs lse and dvec from the forward / pass 1.
// ------------------------------------------------------------------------------------------
template <int NKT, bool GEN, int NW, int SC = 0>
__global__ __launch_bounds__(64 * NW, NKT > 13 ? 2 : (NW + 1) / 2) void attn_bwd_kv_kernel(AttnArgs p) {
  constexpr int NKS2 = (NKT + 1) / 2;
  constexpr int ROWS = NKS2 * 32;
  constexpr int T = ROWS * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * T + 2 * ROWS * 4];
  char* Qb = smem;
  char* Db = smem + T;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * T);
  float* dv_s = lse_s + ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt round-robin to the 8 XCDs: keep the H heads of a sequence on ONE XCD, back to back, so the
  // 128-byte head slices of a token row are fetched together (same DRAM pages, same L2)
  const int xj = blockIdx.x >> 3;
  const int seq = (xj / p.H) * 8 + (blockIdx.x & 7), h = xj % p.H;
  if (seq >= p.nseq) return;
  const int S = SC ? SC : p.mp.S;     // SC: the sequence length as a compile-time constant (0 = run time)
  const int HD = p.H * 64;
  const SeqRows sr = seq_rows(p.mp, seq);
  constexpr int MAXT = (NKT + NW - 1) / NW;   // key tiles per wave
  const int q4 = lane >> 4, i = lane & 15;
  const float c = p.scale * 1.4426950408889634f;
  const int nkt_rt = (S + 15) >> 4;
  // every global load of the workgroup goes out before the first wait: this wave's K / V fragments, then Q and dO
  opx8 kf[MAXT][2], vf[MAXT][2];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int key = (wave + t * NW) * 16 + i;
    const int kj = key < S ? key : S - 1;
    const op_t* kp = p.qkv + row_of(sr, kj) * p.ld + HD + h * 64 + q4 * 8;
    kf[t][0] = *reinterpret_cast<const opx8*>(kp);
    kf[t][1] = *reinterpret_cast<const opx8*>(kp + 32);
    vf[t][0] = *reinterpret_cast<const opx8*>(kp + HD);
    vf[t][1] = *reinterpret_cast<const opx8*>(kp + HD + 32);
  }
  {
    const op_t* src0 = p.mp.mode == 1 ? p.d_o_cls + (long)seq * p.ldo : nullptr;   // dO of token 0 lives in the side buffer
    TileRegs<64 * NW, ROWS> qr, dr;
    tile_issue<64 * NW, ROWS>(qr, p.qkv, p.ld, h * 64, sr, S, nullptr, ROWS, tid);
    tile_issue<64 * NW, ROWS>(dr, p.d_o, p.ldo, h * 64, sr, S, src0, ROWS, tid);
    for (int idx = tid; idx < ROWS; idx += 64 * NW) {
      const long stat = ((long)seq * p.H + h) * S + idx;
      lse_s[idx] = idx < S ? p.lse[stat] * 1.4426950408889634f : 0.f;
      dv_s[idx] = idx < S ? p.dvec[stat] * p.scale : 0.f;
    }
    tile_commit<64 * NW, ROWS>(qr, Qb, ROWS, tid);
    tile_commit<64 * NW, ROWS>(dr, Db, ROWS, tid);
  }
  __syncthreads();

#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int kt = wave + t * NW;
    if (kt >= nkt_rt) break;
    const int key = kt * 16 + i;
    const int kj = key < S ? key : S - 1;
    const opx8 kf0 = kf[t][0], kf1 = kf[t][1], vf0 = vf[t][0], vf1 = vf[t][1];
    bool kbad = key >= S;
    if constexpr (GEN) kbad = kbad || (p.kpm ? (p.kpm[(long)seq * S + kj] != 0) : false);

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int u = 0; u < NKS2; ++u) {
      opx8 pf, sf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qrow = (2 * u + half) * 16 + i;   // a-operand row: query
        const opx8 a0 = bl_row_frag(Qb, qrow, q4);
        const opx8 a1 = bl_row_frag(Qb, qrow, 4 + q4);
        const opx8 d0 = bl_row_frag(Db, qrow, q4);
        const opx8 d1 = bl_row_frag(Db, qrow, 4 + q4);
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        s = MFMA_16x16x32(a0, kf0, s, 0, 0, 0);
        s = MFMA_16x16x32(a1, kf1, s, 0, 0, 0);
        f32x4 dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp = MFMA_16x16x32(d0, vf0, dp, 0, 0, 0);
        dp = MFMA_16x16x32(d1, vf1, dp, 0, 0, 0);
        // s[r] = S[query = (2u+half)*16 + 4*q4 + r][key]
        const int qb = (2 * u + half) * 16 + 4 * q4;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb);   // lse * log2(e)
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dv_s + qb);    // D * scale
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int query = qb + r;
          bool msk = query >= S || kbad;
          if constexpr (GEN) msk = msk || (p.causal && key > query);
          const float pr = msk ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c, -l4[r]));
          pf[half * 4 + r] = (op_t)pr;
          sf[half * 4 + r] = (op_t)(pr * fmaf(dp[r], p.scale, -d4[r]));
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const opx8 qtf = bl_frag(Qb, u, dt, lane);
        const opx8 dtf = bl_frag(Db, u, dt, lane);
        dk[dt] = MFMA_16x16x32(qtf, sf, dk[dt], 0, 0, 0);
        dv[dt] = MFMA_16x16x32(dtf, pf, dv[dt], 0, 0, 0);
      }
    }
    if (key < S) {
      op_t* op = tok_ptr(p.dqkv, p.dqkv_cls, p.ldd, p.mp, sr, seq, key) + HD + h * 64 + 4 * q4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        opx4 ok, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = (op_t)dk[dt][r]; ov[r] = (op_t)dv[dt][r]; }
        *reinterpret_cast<opx4*>(op + 16 * dt) = ok;
        *reinterpret_cast<opx4*>(op + HD + 16 * dt) = ov;
      }
    }
  }
}

int check_common(const AttnArgs& p) {
  if (!p.qkv || p.H <= 0 || p.nseq < 0 || p.mp.S <= 0 || p.mp.S > ATT_ROWS_LONG) return PVRL_EINVAL;
  if ((p.ld % 8)) return PVRL_EINVAL;
  if (p.mp.mode == 1 && (p.mp.T <= 0 || (p.nseq % p.mp.T))) return PVRL_EINVAL;
  if ((p.causal || p.kpm) && p.mp.S > ATT_ROWS) return PVRL_EINVAL;      // the masked forms keep their key bits in one 64-bit word per 16 tiles
  return PVRL_OK;
}

// The two sequence lengths of the TimeSformer path at its benchmark geometry -- 197 (196 patches + cls, spatial) and 32 (temporal,
// long clips) -- are instantiated with S as a compile-time constant: `key >= S` is then false for every key tile but the last
// (none at S = 32), so 12 of 13 tiles lose their per-score compare + select, and the 52 64-bit lane masks that spilled the
// kernels' scalar registers into v_writelane / v_readlane pairs (SGPR spills 66 / 58 -> 0) disappear with them.
#ifndef PVRL_ATTN_SFIX
#define PVRL_ATTN_SFIX 1      // 0: A/B builds without the compile-time sequence lengths (tools/build_variant.py)
#endif
template <int NKT, int NW>
int launch_fwd(const AttnArgs& p, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((p.nseq + 7) / 8) * p.H)), blk(64 * NW);
  constexpr int SFIX = PVRL_ATTN_SFIX ? (NKT == 13 ? 197 : NKT == 2 ? 32 : 0) : 0;
  if (p.causal || p.kpm) hipLaunchKernelGGL((attn_fwd_kernel<NKT, true, NW>), grid, blk, 0, s, p);
  else if (SFIX && p.mp.S == SFIX) hipLaunchKernelGGL((attn_fwd_kernel<NKT, false, NW, SFIX>), grid, blk, 0, s, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<NKT, false, NW>), grid, blk, 0, s, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

template <int NKT, int NW>
int launch_bwd(const AttnArgs& p, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((p.nseq + 7) / 8) * p.H)), blk(64 * NW);
  constexpr int SFIX = PVRL_ATTN_SFIX ? (NKT == 13 ? 197 : NKT == 2 ? 32 : 0) : 0;
  if (p.causal || p.kpm) {
    hipLaunchKernelGGL((attn_bwd_q_kernel<NKT, true, NW>), grid, blk, 0, s, p);
    PVRL_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kv_kernel<NKT, true, NW>), grid, blk, 0, s, p);
  } else if (SFIX && p.mp.S == SFIX) {
    hipLaunchKernelGGL((attn_bwd_q_kernel<NKT, false, NW, SFIX>), grid, blk, 0, s, p);
    PVRL_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kv_kernel<NKT, false, NW, SFIX>), grid, blk, 0, s, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_q_kernel<NKT, false, NW>), grid, blk, 0, s, p);
    PVRL_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kv_kernel<NKT, false, NW>), grid, blk, 0, s, p);
  }
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

}  // namespace

extern "C" int pvrl_attn_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int mode, int64_t T,
                             int64_t cls_base, float scale, int causal, const void* key_padding_mask, void* o,
                             void* o_cls, int64_t ldo, float* lse, void* stream) {
  AttnArgs p = {};
  p.qkv = (const op_t*)qkv; p.ld = ld; p.H = (int)H; p.nseq = (int)nseq;
  p.mp.mode = mode; p.mp.S = (int)S; p.mp.T = (int)T; p.mp.cls_base = cls_base;
  p.scale = scale; p.causal = causal; p.kpm = (const unsigned char*)key_padding_mask;
  p.o = (op_t*)o; p.o_cls = (op_t*)o_cls; p.ldo = ldo; p.lse = lse;
  if (nseq == 0) return PVRL_OK;
  if (int e = check_common(p)) return e;
  if (!o || (ldo % 4) || (mode == 1 && !o_cls)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 16) return launch_fwd<1, 4>(p, s);
  if (S <= 32) return launch_fwd<2, 4>(p, s);
  if (S <= 48) return launch_fwd<3, 4>(p, s);
  if (S <= 80) return launch_fwd<5, 4>(p, s);
  if (S <= 208) return launch_fwd<13, 8>(p, s);
  // longer crops (256^2 -> 257 tokens, 320^2 -> 401): the same kernels with more key tiles, one 8-wave workgroup per CU
  if (S <= 272) return launch_fwd<17, 8>(p, s);
  return launch_fwd<26, 8>(p, s);
}

extern "C" int pvrl_attn_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int mode, int64_t T,
                             int64_t cls_base, float scale, int causal, const void* key_padding_mask, const void* o,
                             const void* o_cls, const void* d_o, const void* d_o_cls, int64_t ldo, const float* lse,
                             float* dvec, void* dqkv, void* dqkv_cls, int64_t ldd, void* stream) {
  AttnArgs p = {};
  p.qkv = (const op_t*)qkv; p.ld = ld; p.H = (int)H; p.nseq = (int)nseq;
  p.mp.mode = mode; p.mp.S = (int)S; p.mp.T = (int)T; p.mp.cls_base = cls_base;
  p.scale = scale; p.causal = causal; p.kpm = (const unsigned char*)key_padding_mask;
  p.ofw = (const op_t*)o; p.ofw_cls = (const op_t*)o_cls; p.d_o = (const op_t*)d_o; p.d_o_cls = (const op_t*)d_o_cls;
  p.ldo = ldo; p.lse = const_cast<float*>(lse); p.dvec = dvec;
  p.dqkv = (op_t*)dqkv; p.dqkv_cls = (op_t*)dqkv_cls; p.ldd = ldd;
  if (nseq == 0) return PVRL_OK;
  if (int e = check_common(p)) return e;
  if (!o || !d_o || !lse || !dvec || !dqkv || (ldo % 8) || (ldd % 4)) return PVRL_EINVAL;
  if (mode == 1 && (!o_cls || !d_o_cls || !dqkv_cls)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (attn_bwd_fused_enabled() && pvrl_attn_bwd_fused_ok(p)) return pvrl_attn_bwd_fused_launch(p, s);
  if (pvrl_attn_bwd_s32_ok(p)) return pvrl_attn_bwd_s32_launch(p, s);
  if (S <= 16) return launch_bwd<1, 4>(p, s);
  if (S <= 32) return launch_bwd<2, 4>(p, s);
  if (S <= 48) return launch_bwd<3, 4>(p, s);
  if (S <= 80) return launch_bwd<5, 4>(p, s);
  if (S <= 208) return launch_bwd<13, 8>(p, s);
  if (S <= 272) return launch_bwd<17, 8>(p, s);
  return launch_bwd<26, 8>(p, s);
}
