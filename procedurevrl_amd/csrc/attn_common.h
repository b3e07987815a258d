// Shared pieces of the MFMA attention kernels (forward, dQ, dK/dV).
//
// Sequence addressing.  The divided space-time attention of the reference
// (lib/models/vit.py:129-151) gathers its sequences with einops rearranges + torch.cat, each a
// full copy of the token tensor.  Here the packed QKV activations stay where the QKV GEMM wrote
// them and the kernels address rows in place:
//   mode 0 (contiguous): row(seq, j) = seq*S + j           -- temporal attention (b h w) t m,
//                                                              order transformer, CLIP text
//   mode 1 (TimeSformer spatial): seq = b*T + t, token 0 is the clip's cls row,
//                                  token j>=1 is patch n=j-1 of frame t:
//            row(seq, 0) = cls_base + b ; row(seq, j) = b*(S-1)*T + (j-1)*T + t
//          outputs for token 0 go to a separate per-(b,t) buffer (the T cls copies of
//          vit.py:139-141 differ after attention and are averaged later, vit.py:147-149).
#pragma once
#include "common.h"

struct SeqMap {
  int mode, S, T;
  long cls_base;
};

__device__ __forceinline__ long seq_row(const SeqMap& mp, int seq, int j) {
  if (mp.mode == 0) return (long)seq * mp.S + j;
  const int b = seq / mp.T, t = seq - b * mp.T;
  return j == 0 ? mp.cls_base + b : (long)b * (mp.S - 1) * mp.T + (long)(j - 1) * mp.T + t;
}

constexpr int ATT_MAX_TILES = 13;               // S <= 208
constexpr int ATT_ROWS = ATT_MAX_TILES * 16;    // 208
constexpr int ATT_ROWS_PAD = 224;               // rounded to 32 for the K=32 MFMA steps
constexpr int ATT_ROWS_LONG = 26 * 16;          // 416: the long-sequence instantiations of the same kernels (NKT = 17, 26)
constexpr int ATT_RM_BYTES = ATT_ROWS_PAD * 128;  // row-major [224][64] bf16 tile, 28 KiB
constexpr int ATT_BL_BYTES = ATT_ROWS_PAD * 128;  // blocked [56][4][4][16] bf16 tile, 28 KiB

// row-major [rows][64] bf16 tile with 16-byte chunk swizzle (conflict-free ds_read_b128 fragments)
__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// blocked tile for ds_read_b64_tr_b16: contiguous [4 rows][16 cols] 128-byte blocks
__device__ __forceinline__ int bl_off(int row, int col) {
  const int rb = row >> 2;
  return (rb * 4 + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (col & 15) * 2;
}

// 8 consecutive columns (16-byte chunk `chunk` of 8) of one row, from the SAME blocked image: with the (cb ^ (rb & 1))
// block swizzle the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots, so one LDS copy of a tile
// serves both the row-wise (a-operand) and the transposed (ds_read_b64_tr_b16) fragment reads.
__device__ __forceinline__ opx8 bl_row_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const opx8*>(tile + bl_off(row, chunk * 8));
}

__device__ __forceinline__ opx8 tr_frag8(const char* tile, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off1));
  union { struct { s16x4 a, b; } s; opx8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// Transposed fragment for a K=32 MFMA step `ks2` (rows 32*ks2 .. 32*ks2+31 of a blocked tile):
// lane (i = lane&15, q = lane>>4) receives column 16*ct + i at rows {32ks2+4q+0..3, 32ks2+16+4q+0..3}.
__device__ __forceinline__ opx8 bl_frag(const char* tile, int ks2, int ct, int lane) {
  const int q = lane >> 4, i = lane & 15;
  const int rb0 = 8 * ks2 + q, rb1 = rb0 + 4;
  const int o0 = (rb0 * 4 + (ct ^ (rb0 & 1))) * 128 + i * 8;
  const int o1 = (rb1 * 4 + (ct ^ (rb1 & 1))) * 128 + i * 8;
  return tr_frag8(tile, o0, o1);
}

// Per-workgroup view of one sequence: row(j) without per-element integer division.
struct SeqRows {
  long base0;   // row of token 0
  long base1;   // row of token 1
  long stride;  // row distance between tokens j and j+1 (j >= 1)
};
__device__ __forceinline__ SeqRows seq_rows(const SeqMap& mp, int seq) {
  SeqRows r;
  if (mp.mode == 0) {
    r.base0 = (long)seq * mp.S; r.base1 = r.base0 + 1; r.stride = 1;
  } else {
    const int b = seq / mp.T, t = seq - b * mp.T;
    r.base0 = mp.cls_base + b; r.base1 = (long)b * (mp.S - 1) * mp.T + t; r.stride = mp.T;
  }
  return r;
}
__device__ __forceinline__ long row_of(const SeqRows& r, int j) { return j == 0 ? r.base0 : r.base1 + (long)(j - 1) * r.stride; }

// Cooperative load of one head slice [S rows][64] of a packed activation into LDS (256 threads).
// All global loads of the tile are issued before the first LDS store so their latencies overlap
// (7 x 16 B in flight per thread for S = 197).  `src0` (optional) overrides the source row of token 0
// (side buffer of the per-(b,t) cls rows).  rm / bl may each be null.  Rows >= S are zero-filled.
constexpr int ATT_LOAD_ITERS = (ATT_ROWS_PAD * 8) / 256;   // 7
__device__ __forceinline__ void load_head_tile(const op_t* base, long ld, int col0, const SeqRows& sr, int S,
                                               const op_t* src0, char* rm, int rm_rows, char* bl, int bl_rows,
                                               int tid) {
  const int maxrows = rm_rows > bl_rows ? rm_rows : bl_rows;
  u32x4 v[ATT_LOAD_ITERS];
#pragma unroll
  for (int it = 0; it < ATT_LOAD_ITERS; ++it) {
    const int idx = tid + 256 * it;
    const int row = idx >> 3, c = idx & 7;
    v[it] = (u32x4){0u, 0u, 0u, 0u};
    if (row < S && row < maxrows) {
      const op_t* src = (row == 0 && src0) ? src0 : base + row_of(sr, row) * ld;
      v[it] = *reinterpret_cast<const u32x4*>(src + col0 + c * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < ATT_LOAD_ITERS; ++it) {
    const int idx = tid + 256 * it;
    const int row = idx >> 3, c = idx & 7;
    if (rm && row < rm_rows) *reinterpret_cast<u32x4*>(rm + rm_off(row, c)) = v[it];
    if (bl && row < bl_rows) *reinterpret_cast<u32x4*>(bl + bl_off(row, c * 8)) = v[it];
  }
}

// kernel arguments shared by attn_mfma.hip (forward, two-pass backward) and attn_bwd_fused.hip
struct AttnArgs {
  const op_t* qkv; long ld;   // packed [rows][3*H*64]: q | k | v, head h at columns h*64
  int H, nseq;
  SeqMap mp;
  float scale;
  int causal;
  const unsigned char* kpm;   // [nseq][S], 1 = key masked, or null
  // forward
  op_t* o; op_t* o_cls; long ldo;
  float* lse;                 // [nseq][H][S]
  // backward
  const op_t* d_o; const op_t* d_o_cls; const op_t* ofw; const op_t* ofw_cls;
  float* dvec;                // [nseq][H][S]  rowsum(dO * O)
  op_t* dqkv; op_t* dqkv_cls; long ldd;   // dqkv_cls: [nseq][3*H*64] partial rows for token 0 (mode 1)
};

// row pointer helpers for per-token outputs / inputs that keep token 0 in a side buffer (mode 1)
template <typename T>
__device__ __forceinline__ T* tok_ptr(T* tok, T* cls, long ld, const SeqMap& mp, const SeqRows& sr, int seq, int j) {
  if (mp.mode == 1 && j == 0) return cls + (long)seq * ld;
  return tok + row_of(sr, j) * ld;
}

__device__ __forceinline__ unsigned pack_opx2(float a, float b) {
  union { opx2 v; unsigned u; } x;
  x.v[0] = (op_t)a; x.v[1] = (op_t)b;
  return x.u;
}

// ---- head_dim 96 (MViTv2): 32-row tiles of 6 sixteen-column blocks, shared by attn_pool.hip and mvit_rel.hip
constexpr int PB_D = 96, NCB = 6;
// blocked [rows][96] bf16 tile: contiguous [4 rows][16 cols] 128-byte blocks, pairwise block swizzle
__device__ __forceinline__ int pb_off(int row, int col) {
  const int rb = row >> 2;
  return (rb * NCB + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (col & 15) * 2;
}
__device__ __forceinline__ opx8 pb_row_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const opx8*>(tile + pb_off(row, chunk * 8));
}
// transposed fragment over the tile's 32 rows: lane (i, q) receives column 16*ct + i at rows {4q..4q+3, 16+4q..16+4q+3}
__device__ __forceinline__ opx8 pb_tr_frag(const char* tile, int ct, int lane) {
  const int q = lane >> 4, i = lane & 15;
  const int rb0 = q, rb1 = q + 4;
  return tr_frag8(tile, (rb0 * NCB + (ct ^ (rb0 & 1))) * 128 + i * 8, (rb1 * NCB + (ct ^ (rb1 & 1))) * 128 + i * 8);
}
