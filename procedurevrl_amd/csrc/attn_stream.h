// Shared pieces of the PERSISTENT, LDS-DMA-streamed attention kernels (attn_bwd_fused.hip, attn_fwd_stream.hip): the image layout,
// the (sequence, head) work list of a workgroup, the source-address side of an LDS-DMA piece, the LDS-only barrier.
#pragma once
#include "attn_common.h"

constexpr int FB_ROWS = 224;                 // 7 blocks of 32 rows
constexpr int FB_TILE = FB_ROWS * 128;       // one [224][64] head slice

// Block barrier: LDS traffic only.  __syncthreads() is a workgroup-scope release: it puts s_waitcnt vmcnt(0) in front of
// s_barrier, i.e. a wave would wait for its global STORES (and the prefetch in flight) at every block.
#define FB_BARRIER()                                     \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
// [4 rows][16 cols] 128-byte blocks (the unit ds_read_b64_tr_b16 transposes), four blocks per 4-row band.  Two swizzles make the
// 32-row a-operand walk of v_mfma_32x32x16 (lane = row, 16 bytes at a fixed column chunk) conflict-free: the block order within a
// band flips with bit 0 of the band index, the two 16-byte halves of a row's 32 bytes flip with bit 1.
__device__ __forceinline__ int fb_off(int row, int col) {
  const int rb = row >> 2;
  return (rb * 4 + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (((col & 15) * 2) ^ ((rb & 2) << 3));
}

struct FbItem {
  int seq, h;
  SeqRows sr;
};

__device__ __forceinline__ bool fb_decode(const AttnArgs& p, int vb, FbItem& it) {
  // the H heads of a sequence run back to back on ONE XCD (same order as the forward kernel)
  const int xj = vb >> 3;
  it.seq = (xj / p.H) * 8 + (vb & 7);
  it.h = xj % p.H;
  if (it.seq >= p.nseq) return false;
  it.sr = seq_rows(p.mp, it.seq);
  return true;
}
// first virtual block >= vb (stride `st`) that names a real sequence, or -1
__device__ __forceinline__ int fb_next(const AttnArgs& p, int vb, int st, int nvb, FbItem& it) {
  for (; vb < nvb; vb += st)
    if (fb_decode(p, vb, it)) return vb;
  return -1;
}

// sum over the 8 consecutive lanes of a row's chunks, on the VALU (DPP quad permutes + half-row mirror): __shfl_xor would be three
// ds_bpermute round trips through an LDS pipe that is the kernel's bottleneck
__device__ __forceinline__ float fb_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  return v;
}

// row pointer without a branch (token 0 of a spatial sequence lives in the side buffer)
template <typename T>
__device__ __forceinline__ T* fb_tok(T* tok, T* cls, long ld, const AttnArgs& p, const FbItem& it, int j) {
  T* a = tok + row_of(it.sr, j) * ld;
  T* b = cls + (long)it.seq * ld;
  return (p.mp.mode == 1 && j == 0) ? b : a;
}

// Lane l's 16 bytes of an LDS-DMA piece land at piece + 16 l.  Piece k of an image = rows 8k .. 8k+7 (1 KB); under fb_off that
// slot holds (row, 8-column chunk) = the values below -- the copy itself cannot permute, the source addresses do.
__device__ __forceinline__ void fb_piece_src(int k, int lane, int S, int& rc, int& col) {
  const int rb = 2 * k + (lane >> 5);
  const int row = 4 * rb + ((lane >> 1) & 3);
  const int cb = ((lane >> 3) & 3) ^ (rb & 1), half = (lane & 1) ^ ((rb >> 1) & 1);
  rc = min(row, S - 1);
  col = 16 * cb + 8 * half;
}

