// Spatial-attention forward as a persistent, LDS-DMA-streamed kernel (96 < S <= 224 tokens, no masks).
//
// Reference semantics: Attention.forward, lib/models/vit.py:75-92 -- softmax((q k^T) * scale) v -- on the spatial sequences of
// Block.forward (vit.py:137-151).  The one-workgroup-per-item form (attn_fwd_kernel, attn_mfma.hip) loads K and V, computes, stores:
// 84-91 us for the 3,072 (sequence, head) items of a 32-clip step against ~62 us of HBM time.  Here one 7-wave workgroup per CU
// walks its items; while item i computes, the K and V head slices of item i + 1 travel by LDS-DMA into the second pair of images
// (56 pieces of 1 KB, eight per wave, no registers) and its Q rows into 16 registers per lane.
//
//   wave w owns queries 32w .. 32w+31 (7 waves at S = 197; no cross-wave traffic at all in the forward)
//     S^T tiles   32 keys x 32 queries, swapped operands (a = K rows from LDS, b = Q from registers): a lane owns ONE query and
//                 16 keys per tile, so the softmax row reductions are in-register plus one lane <-> lane + 32 exchange; all
//                 7 tiles (112 registers) are kept: exact single-pass softmax, no online rescale
//     O^T += V^T P  per 16 keys: a = transposed V fragments (ds_read_b64_tr_b16), b = P straight from the score registers
//     O           transposed through this wave's 4 KB of LDS and stored as whole 128-byte rows (8 rows per instruction)
//   one barrier per item.
// 140 KB of LDS (2 x (K + V) images + 7 x 4 KB of staging), <= 256 registers, one workgroup per CU.
// (probe, not part of the library: to measure it again, copy the file next to csrc/attn_stream.h, add the two declarations +
//  `if (pvrl_attn_fwd_stream_ok(p)) return pvrl_attn_fwd_stream_launch(p, s);` in front of pvrl_attn_fwd's size dispatch in
//  csrc/attn_mfma.hip, build; result and trace: profiles/r4_attn_bwd_fused.txt section 7)
#include "attn_stream.h"
#include "../../include/pvrl.h"
#include <stdlib.h>

namespace {

#ifndef PVRL_FS_TRACE
#define PVRL_FS_TRACE 0      // probe builds only: the waves of workgroup 8 stamp the cycle counter at the seams of their third item; dumped over
#endif                       // the tail of the lse array (tools/probe/attn_bwd_ab.py ftrace)
#if PVRL_FS_TRACE
#define FS_STAMP(k) do { if (tracing) stamps[k] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define FS_STAMP(k) do { } while (0)
#endif
constexpr int FS_K = 0, FS_V = 2 * FB_TILE, FS_ST = 4 * FB_TILE;      // K images (2), V images (2), per-wave staging
constexpr int FS_LDS = FS_ST + 7 * 4096;

__global__ __launch_bounds__(448, 2) void attn_fwd_stream_kernel(AttnArgs p, int nvb) {
  __shared__ __attribute__((aligned(16))) char smem[FS_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.mp.S;
  const int nkt = (S + 31) >> 5;
  const int HD = p.H * 64;
  const int n = lane & 31, g = lane >> 5;
  const float c = p.scale * 1.4426950408889634f;
  const unsigned ldsbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const bool qwave = wave < nkt;                 // this wave owns queries of the sequence

  // lane parts of the LDS addresses inside one 32-row block (4096 bytes of an image), as in attn_bwd_fused.hip
  const int rbl = n >> 2, b0 = rbl & 1, b1 = (rbl >> 1) & 1;
  const int rowbase = rbl * 512 + (n & 3) * 32 + ((16 * g) ^ (16 * b1));
  const int e0 = rowbase + b0 * 128, e1 = rowbase + (1 - b0) * 128;      // column step s even / odd (+256 for s >= 2)
  const int i16 = lane & 15, hi = (lane >> 4) & 1;
  const int trb = g * 512 + (hi ^ g) * 128 + (i16 >> 2) * 32;
  const int tr0 = trb + 8 * (i16 & 3), tr1 = trb + 1024 + ((8 * (i16 & 3)) ^ 16);

  FbItem cur, nxt;
  int vcur = -1;
  int vnxt = fb_next(p, blockIdx.x, gridDim.x, nvb, nxt);
  if (vnxt < 0) return;
  int par = 0;

  // K and V images of item `it` into image pair `pr`: 56 pieces of 1 KB, eight per wave; this wave's Q rows into registers
  auto send = [&](const FbItem& it, int pr, opx8* q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = wave * 8 + j;                // 0 .. 55
      const int k = idx % 28, isv = idx / 28;
      int rc, col;
      fb_piece_src(k, lane, S, rc, col);
      glds16_raw_v(p.qkv + row_of(it.sr, rc) * p.ld + (1 + isv) * HD + it.h * 64 + col,
                   ldsbase + (isv ? FS_V : FS_K) + pr * FB_TILE + k * 1024);
    }
    const int query = min(32 * wave + n, S - 1);
    const op_t* qp = p.qkv + row_of(it.sr, query) * p.ld + it.h * 64 + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) q[s] = *reinterpret_cast<const opx8*>(qp + 16 * s);
  };

#if PVRL_FS_TRACE
  unsigned stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int nitem = 0;
  bool tracing = false;
#endif
  opx8 qf[4], qn[4];
  send(nxt, 0, qn);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FB_BARRIER();
  while (true) {
#if PVRL_FS_TRACE
    tracing = blockIdx.x == 8 && ++nitem == 3;
#endif
    FS_STAMP(0);
    cur = nxt; vcur = vnxt;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
    vnxt = fb_next(p, vcur + gridDim.x, gridDim.x, nvb, nxt);
    if (vnxt >= 0 && !qwave) send(nxt, par ^ 1, qn);

    if (qwave) {
      const char* Kc = smem + FS_K + par * FB_TILE;
      const char* Vc = smem + FS_V + par * FB_TILE;
      // ---- scores of all key tiles: S^T[key][query], lane (query n, g) holds keys 32 kt + 4 g + 8 j + r in sc[kt][4 j + r]
      f32x16 sc[7];
#pragma unroll
      for (int kt = 0; kt < 7; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const opx8 ka = *reinterpret_cast<const opx8*>(Kc + kt * 4096 + ((s & 1) ? e1 : e0) + 256 * (s >> 1));
            sc[kt] = MFMA_32x32x16(ka, qf[s], sc[kt], 0, 0, 0);
          }
        }
      }
      FS_STAMP(1);
      // the next item's K / V pieces and Q rows go out here, under the score MFMAs (at the top of the item they sat in front of them)
      if (vnxt >= 0) send(nxt, par ^ 1, qn);
      FS_STAMP(2);
      // ---- softmax over the lane's 16 nkt keys and its partner's (lane ^ 32): exp(scale s - max) = exp2(fma(s, c, -max c))
      // (four independent partial maxima / sums: one running value would be a 112-deep dependent chain per lane)
      float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int kt = 0; kt < 7; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (kt == nkt - 1) {               // only the last tile can hold keys past the sequence (copies of the last row)
              const int key = 32 * kt + 4 * g + 8 * (r >> 2) + (r & 3);
              if (key >= S) sc[kt][r] = -INFINITY;
            }
            mx4[r & 3] = fmaxf(mx4[r & 3], sc[kt][r]);
          }
        }
      }
      float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mc = mx * c;
      float sum4[4] = {0.f, 0.f, 0.f, 0.f};
      f32x16 oacc[2];
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dh][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 7; ++kt) {
        if (kt < nkt) {
          opx8 pf[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c, -mc));
            sum4[r & 3] += e;
            pf[r >> 3][r & 7] = (op_t)e;
          }
          // O^T[d][query] += V^T[d][key] P^T[key][query]: k-step t = keys 32 kt + 16 t + {4 g + r, 8 + 4 g + r} (the lane's own registers)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
              const opx8 av = tr_frag8(Vc + kt * 4096 + t * 2048 + dh * 256, tr0, tr1);
              oacc[dh] = MFMA_32x32x16(av, pf[t], oacc[dh], 0, 0, 0);
            }
        }
      }
      FS_STAMP(3);
      float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      // this wave's prefetch has landed long ago; waiting here (before the stores below) keeps `vmcnt(0)` from waiting for stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FS_STAMP(4);
      // ---- O: transpose through this wave's 4 KB ([query][64 columns], 16-byte chunks swizzled by the query), whole rows out
      char* st = smem + FS_ST + wave * 4096;
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          opx4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (op_t)(oacc[dh][4 * j + r] * inv);
          *reinterpret_cast<opx4*>(st + n * 128 + (((4 * dh + j) ^ (n & 7)) * 16) + 8 * g) = ov;
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int idx = lane + 64 * t;
        const int r = idx >> 3, ch = idx & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(st + r * 128 + ((ch ^ (r & 7)) * 16));
        const int query = 32 * wave + r;
        if (query < S) *reinterpret_cast<u32x4*>(fb_tok(p.o, p.o_cls, p.ldo, p, cur, query) + cur.h * 64 + ch * 8) = v;
      }
      const int query = 32 * wave + n;
      if (g == 0 && query < S && p.lse) p.lse[((long)cur.seq * p.H + cur.h) * S + query] = mx * p.scale + __logf(sum);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FS_STAMP(5);
    if (vnxt < 0) break;
    FB_BARRIER();                  // the next item's images are complete; everyone has left the current ones
    FS_STAMP(6);
#if PVRL_FS_TRACE
    if (tracing && lane == 0 && p.lse)
      for (int k = 0; k < 8; ++k) reinterpret_cast<unsigned*>(p.lse)[((long)p.nseq * p.H * S) + wave * 8 + k] = stamps[k];      // behind the array: the probe allocates 64 words more
#endif
    par ^= 1;
  }
}

// PVRL_ATTN_FWD_STREAM=0 sends every case back to the one-workgroup-per-item kernels (A/B runs); read once
int attn_fwd_stream_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("PVRL_ATTN_FWD_STREAM");
    on = e ? (e[0] == '0' ? 0 : 1) : 1;
  }
  return on;
}

}  // namespace

bool pvrl_attn_fwd_stream_ok(const AttnArgs& p) {
  if (!attn_fwd_stream_enabled() || p.causal || p.kpm) return false;
  if (p.mp.S <= 96 || p.mp.S > FB_ROWS) return false;
  if ((p.ldo % 8) || !p.o) return false;                               // 16-byte row stores
  return true;
}

int pvrl_attn_fwd_stream_launch(const AttnArgs& p, hipStream_t s) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  const int nvb = 8 * ((p.nseq + 7) / 8) * p.H;
  const int grid = nvb < cus ? nvb : cus - (cus & 7);      // a multiple of 8: a workgroup's items stay on its XCD
  hipLaunchKernelGGL(attn_fwd_stream_kernel, dim3(grid), dim3(448), 0, s, p, nvb);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
