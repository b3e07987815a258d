// MEASURED AND NOT SHIPPED (round 3): the MViTv2 conv-pool forward (depthwise 3x3x3 convolution + LayerNorm) on the matrix pipe.
// This file is NOT compiled into the product library; it is the kernel as it ran on MI355X (it replaced the stride-(1, s, s) branch of
// pvrl_mvit_pool_fwd in csrc/mvit.hip behind PVRL_POOL_MFMA=1 and passed tests/test_mvit_gpu.py), kept for the measurements:
//
//   form                                                            block 0 q   block 4 q (x10)   per step (forward pools)
//   VALU, 16 lanes per column, sliding along t (shipped)              300 us       61 us             2.43 ms
//   v1: 3 waves x 32 channels, v_mfma_f32_16x16x16, 252 VGPRs         341 us       93 us             3.3 ms
//   v2 (below): 6 waves x 16 channels, 16x16x32 over tap pairs,       351 us       87 us             3.3 ms
//       160 VGPRs, 12 waves per CU
//
// Why it loses although it removes 190 of the VALU kernel's 264 instructions per frame: a wave walks its 8 frames serially with the
// next frame's loads one frame (~500-1,000 cycles) ahead -- less than a memory round trip -- and the register budget (60 VGPRs of
// diagonal operands) leaves neither deeper prefetch nor more than 12 waves per CU to cover it; the VALU kernel is issue-bound
// (tools/pmc_pool.sh: every wave waits 60 % of its cycles while its SIMD's issue slots are ~95 % used by the 4-5 resident waves)
// but keeps 16-20 waves per CU.  Halving the MFMA time (v1 -> v2) and halving the load segment size changed nothing: neither pipe
// nor the texture path is the limit.  Per frame a wave runs one dependent chain -- loads -> 15 MFMAs (five deep per accumulator) ->
// accumulator read -> two cross-lane reductions -> LDS exchange + barrier -> rsqrt -> convert -> stores -- of ~2,000 cycles with ~500 cycles
// of issue in it; it takes 8+ waves per SIMD (or several tiles per wave) to fill that, and the diagonal operands alone cap the kernel at 3.
// Building the operands on the fly from 8 VGPRs of packed weights (5 VALU per MFMA) frees the registers but costs 75 VALU per frame:
// ~2.1 issue cycles per output against the VALU kernel's 2.75 -- not worth a second kernel family.  Staging the input rows in LDS with
// LDS-DMA several frames ahead would fix the memory side only.

// ---- Round 3: the depthwise convolution on the matrix pipe ------------------------------------------------------------------------
// PMC (tools/pmc_pool.sh, block 4): the VALU forms above are instruction-issue-bound -- 2,590 VALU instructions per wave for 3,072
// outputs, every wave waiting 60 % of its cycles while its SIMD's issue slots are ~95 % taken by the four resident waves.  Of the
// 264 instructions per frame only 81 are the (packed) FMAs; 54 unpack bf16, 55 move operands into pairs, the rest is addresses and
// LayerNorm.  A depthwise tap IS a matrix product with a diagonal weight matrix,
//     out[c][pos] += sum_c' diag(w_tap)[c][c'] * in[c'][pos + tap],
// and v_mfma_f32_16x16x32 does it for 16 channels x 16 positions x TWO taps (K = 2 x 16 channels) in 16 cycles of the matrix pipe:
// the 16-bit activations go in as loaded, nothing is unpacked or moved, and the VALU keeps only the masks and the LayerNorm.
//
//   workgroup = 6 waves = one tile of 16 output columns (b, h, yo, xo); wave w owns channels 16 w .. 16 w + 15 of the head
//   B operand, lane (n = lane & 15, g = lane >> 4): 16 bytes = channels 8 (g & 1) .. + 7 of position n's tap (2 p + (g >> 1)) of pair p
//   A operand (registers, 15 x 4): row m = output channel, nonzero only at k = (its own channel, either tap of the pair)
//   D, lane (n, g): channels 4 g .. 4 g + 3 of position n -- an 8-byte store
//   the walk along t keeps three running sums as pool_fwd_t_kernel does; the load of pair p for frame t + 1 is issued as soon as
//   frame t's MFMAs have consumed the register.  LayerNorm over the head's 96 channels: per-position (sum, sum of squares) of the six
//   waves meet in LDS, one barrier per frame.
union Mop8 { u32x4 q; opx8 h; };
__device__ __forceinline__ unsigned op_bits(float v) {
  union { op_t h; unsigned short u; } x;
  x.h = (op_t)v;
  return x.u;
}
// A operand of (temporal tap a, in-plane pair p) for this lane: row m = lane & 15, columns k = 8 g .. 8 g + 7 = (tap 2 p + (g >> 1),
// channels 8 (g & 1) .. + 7); nonzero only at its own channel.  flip: the transposed convolution of the backward (tap -> 26 - tap).
__device__ __forceinline__ opx8 diag_operand(const float* __restrict__ w, int a, int p, int wv, int lane, bool flip) {
  const int m = lane & 15, g = lane >> 4;
  const int k9 = 2 * p + (g >> 1);
  const int tap = a * 9 + min(k9, 8);
  const bool nz = k9 <= 8 && (g & 1) == (m >> 3);
  const unsigned v = nz ? op_bits(w[(wv * 16 + m) * 27 + (flip ? 26 - tap : tap)]) : 0u;
  const int i = m & 7;
  Mop8 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) o.q[d] = (i >> 1) == d ? v << (16 * (i & 1)) : 0u;
  return o.h;
}

__global__ __launch_bounds__(384) void pool_fwd_mfma_kernel(const op_t* __restrict__ qkv, PoolGeom g,
                                                            const float* __restrict__ w, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, op_t* __restrict__ y,
                                                            op_t* __restrict__ cbuf) {
  __shared__ float red[2][6][16][2];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // 16-channel block of this wave
  const int n = lane & 15, kg = lane >> 4;
  const int HoWo = g.Ho * g.Wo, Lo = g.T * HoWo, L = g.T * g.Hh * g.Ww, plane = g.Hh * g.Ww;
  const unsigned nconv = (unsigned)((long)g.B * g.H * HoWo);              // conv columns; then one column per (b, h) cls token
  const unsigned ncol = nconv + (unsigned)(g.B * g.H);
  const unsigned ntile = (ncol + 15u) >> 4;
  const int cin = wv * 16 + 8 * (kg & 1);                                // first channel of this lane's B operand inside the head
  const int cout = wv * 16 + 4 * kg;                                     // first channel of this lane's 4 results
  opx8 dg[3][5];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int p = 0; p < 5; ++p) dg[a][p] = diag_operand(w, a, p, wv, lane, false);
  float gm[4], bt[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { gm[e] = gamma[cout + e]; bt[e] = beta[cout + e]; }
  const long fstride = (long)plane * g.ld;                               // elements between frames of the input
  int par = 0;
  // LayerNorm + the two stores of one output token per position: v = the lane's 4 channels
  auto ln_store = [&](const f32x4 v, op_t* cdst, op_t* ydst, bool ok) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    float q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    if (kg == 0) { red[par][wv][n][0] = s; red[par][wv][n][1] = q; }
    __syncthreads();
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) { S += red[par][u][n][0]; Q += red[par][u][n][1]; }
    par ^= 1;
    const float mu = S * (1.f / HD);
    const float rs = rsqrtf(fmaxf(Q * (1.f / HD) - mu * mu, 0.f) + eps);
    union { opx4 h; u32x2 u; } c, o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c.h[e] = (op_t)v[e];
      o.h[e] = (op_t)((v[e] - mu) * rs * gm[e] + bt[e]);
    }
    if (ok) {
      *reinterpret_cast<u32x2*>(cdst) = c.u;
      *reinterpret_cast<u32x2*>(ydst) = o.u;
    }
  };
  for (unsigned tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const unsigned colid = tile * 16u + (unsigned)n;
    const bool ok = colid < ncol;
    if (tile * 16u >= nconv) {                                           // a tile of cls tokens: LayerNorm only (nconv % 16 == 0: host)
      const unsigned bh = min(colid, ncol - 1) - nconv;
      const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
      union { opx4 h4; u32x2 u; } x;
      x.u = *reinterpret_cast<const u32x2*>(qkv + (g.cls_row0 + b) * g.ld + g.col0 + h * HD + cout);
      const f32x4 v = {(float)x.h4[0], (float)x.h4[1], (float)x.h4[2], (float)x.h4[3]};
      const long o = ((long)bh * (Lo + 1) + Lo) * HD + cout;
      ln_store(v, cbuf + o, y + o, ok);
      continue;
    }
    const unsigned bh = colid / (unsigned)HoWo;                          // (conv tiles are full)
    const int pos = (int)(colid - bh * (unsigned)HoWo);
    const int h = (int)(bh % (unsigned)g.H), b = (int)(bh / (unsigned)g.H);
    const int xo = pos % g.Wo, yo = pos / g.Wo;
    int noff[5];                                                         // element offset of this lane's tap of each pair, -1 = outside
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const int k9 = min(2 * p + (kg >> 1), 8);
      const int yi = yo * g.sh - 1 + k9 / 3, xi = xo * g.sw - 1 + k9 % 3;
      noff[p] = (yi >= 0 && yi < g.Hh && xi >= 0 && xi < g.Ww) ? (yi * g.Ww + xi) * (int)g.ld : -1;
    }
    const op_t* base = qkv + (long)b * L * g.ld + g.col0 + h * HD + cin;
    op_t* cdst = cbuf + ((long)bh * (Lo + 1) + pos) * HD + cout;
    op_t* ydst = y + ((long)bh * (Lo + 1) + pos) * HD + cout;
    f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aC = aP, aN = aP;                    // running sums of outputs t-1, t, t+1
    Mop8 raw[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) raw[p].q = *reinterpret_cast<const u32x4*>(base + max(noff[p], 0));
    for (int ti = 0; ti < g.T; ++ti) {
      const op_t* nb = base + (long)min(ti + 1, g.T - 1) * fstride;      // (the last frame re-reads itself: value unused)
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        Mop8 r = raw[p];
        if (noff[p] < 0) r.q = (u32x4){0u, 0u, 0u, 0u};
        raw[p].q = *reinterpret_cast<const u32x4*>(nb + max(noff[p], 0));        // frame ti + 1, in flight under the rest of frame ti
        aN = MFMA_16x16x32(dg[0][p], r.h, aN, 0, 0, 0);                   // a = 0: this frame is the one BEFORE output ti + 1
        aC = MFMA_16x16x32(dg[1][p], r.h, aC, 0, 0, 0);                   // a = 1
        aP = MFMA_16x16x32(dg[2][p], r.h, aP, 0, 0, 0);                   // a = 2: this frame is the one AFTER output ti - 1
      }
      if (ti >= 1) {
        const long o = (long)(ti - 1) * HoWo * HD;
        ln_store(aP, cdst + o, ydst + o, true);
      }
      aP = aC; aC = aN; aN = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const long o = (long)(g.T - 1) * HoWo * HD;
    ln_store(aP, cdst + o, ydst + o, true);
  }
}


// ---- launcher branch as it stood in pvrl_mvit_pool_fwd ----
#if 0
  const long nconv = (long)B * H * g.Ho * g.Wo;
  // the matrix-pipe form: 16-column tiles (conv columns first, then the cls tokens), 32-bit element offsets inside one clip
  if (g.st == 1 && pool_mfma_enabled() && nconv % 16 == 0 && (ld % 8) == 0 && (col0 % 8) == 0 &&
      (long)g.Hh * g.Ww * ld < (1L << 31) && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)conv_out % 16) == 0) {
    const long ntile = (nconv + (long)B * H + 15) / 16;
    const unsigned grid = (unsigned)std::min<long>(ntile, 256L * 2);      // two resident workgroups per CU walk the tiles
    hipLaunchKernelGGL(pool_fwd_mfma_kernel, dim3(grid), dim3(384), 0, (hipStream_t)stream, (const op_t*)qkv, g, w, gamma, beta, eps,
                       (op_t*)y, (op_t*)conv_out);
    PVRL_LAUNCH_CHECK();
    return PVRL_OK;
  }
#endif
