// Pooling attention of MViTv2 (reference: MultiScaleAttention.forward, lib/models/slowfast_mvit/attention.py:404-442):
//     attn = (q * scale) @ k^T  + rel_h[q, kh(k)] + rel_w[q, kw(k)] + rel_t[q, kt(k)]   (patch queries x patch keys only)
//     out  = softmax(attn) @ v  (+ q for patch queries: residual pooling, attention.py:431-435)
// q / k / v are the pooled, LayerNorm-ed tensors [B*H][L+1][96] (cls token LAST), head_dim 96, up to 25,088 queries and
// 392 or 1,568 keys per (clip, head).  Flash-style: 64 queries per workgroup (16 per wave), keys streamed in tiles of 32
// through LDS, online softmax in the exp2 domain, the score tile never leaves registers.
// MFMA operands are swapped (S^T = K Q^T, O^T = V^T P^T) so that one lane owns one query: the softmax statistics are
// per-lane scalars, P feeds the second MFMA straight from registers, and V^T / K^T fragments come from the row-major LDS
// tile through ds_read_b64_tr_b16.
// The decomposed relative-position bias is part of the score MFMA: with E[key][j] the 0/1 key map (j = h(key),
// kh + w(key), kh + kw + t(key); zero rows for the cls key and the padding) the bias is rel[q][:] . E[key][:], i.e. the
// head dimension grows from 96 to 96 + J.  rel arrives from pvrl_mvit_rel_fwd already divided by `scale` and split into
// a hi + lo 16-bit pair (~16 mantissa bits), so  logits = scale * (K Q^T + E rel_hi^T + E rel_lo^T)  and the per-score
// VALU work is one max, one FMA (scale and running max folded into the exp2 argument), one exp2, one add.
// Backward = two kernels (the contraction over queries needs the un-swapped layout): dQ (+ d rel = dS E by MFMA) per
// query tile, dK / dV per key tile.
#include "attn_common.h"
#include "../../include/pvrl.h"

namespace {

constexpr int D = PB_D;                 // head_dim
constexpr int KT = 32;                  // keys (or queries) per LDS tile
constexpr int TILE_BYTES = KT * D * 2;  // 6 KiB
constexpr int JMAX = 40;                // kh + kw + kt <= 36 (14 + 14 + 8); the d-rel MFMA covers 48 columns
constexpr int MAXKEYS = 1664;           // 8*14*14 + 1 = 1569 keys, rounded
constexpr int ET_BYTES = KT * 64 * 2;   // key-map tile [32 keys][64 j], blocked layout (bl_off): 4 KiB
constexpr float LOG2E = 1.4426950408889634f;

struct PA {
  const op_t* q; const op_t* k; const op_t* v;   // [BH][L+1][96]
  const op_t* relp;                              // [BH][Lq][hi JP | lo JP]: (rel / scale) as a 16-bit pair, zero for j >= J
  const char* keymap;                            // [tiles of 32 keys][ET_BYTES]
  op_t* o; long ldo;                             // token-major [B*Lq + B][ldo], column h*96 + d
  float* lse;                                    // [BH][Lq+1]   (log2 domain)
  const op_t* d_o;                               // same layout as o
  float* delta;                                  // [BH][Lq+1]
  op_t* dq; op_t* dk; op_t* dv;                  // [BH][L+1][96]
  float* drel;                                   // [BH][Lq][J]
  float* kv_part;                                // [nsplit][2][BH][Lk+1][96] fp32 partial dK / dV
  int B, H, Lq, Lk, kt, kh, kw, J, JP;
  float scale;
  int gx, gy, gz;                                // logical grid: tiles x (b, h) x slices, see pattn_block
};

// XCD-aware launch order (round 3).  The workgroups of one (b, h) share an operand -- K / V for the query-tile kernels, Q / dO for the
// dK / dV kernel -- and the hardware deals consecutive workgroup ids round-robin over the 8 XCDs: with the natural (tile fastest) order
// every XCD's L2 fetched every (b, h)'s shared operand (forward: 280 MB read per launch against 58 MB of operands; dK / dV: 995 MB at
// 5.7 TB/s -- fabric-bound on re-reads).  1-D grid, id = 8 j + xcd: XCD x runs the (b, h) with bh % 8 == x, their tiles back to back.
#ifndef PVRL_PATTN_XCD
#define PVRL_PATTN_XCD 1
#endif
struct PBlk { int x, y, z; };
__device__ __forceinline__ bool pattn_block(const PA& p, PBlk& o) {
  const int inner = p.gx * p.gz;
#if PVRL_PATTN_XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int i = j % inner;
  o.y = (j / inner) * 8 + xcd;
#else
  const int i = blockIdx.x % inner;
  o.y = blockIdx.x / inner;
#endif
  o.x = i % p.gx;
  o.z = i / p.gx;
  return o.y < p.gy;
}
static unsigned pattn_grid(PA& p, long gx, long gy, long gz) {
  p.gx = (int)gx; p.gy = (int)gy; p.gz = (int)gz;
#if PVRL_PATTN_XCD
  return (unsigned)(8 * ((gy + 7) / 8) * gx * gz);
#else
  return (unsigned)(gx * gy * gz);
#endif
}

// Staging of a PAIR of 32 x 96 tiles (K | V, or Q | dO): 768 chunks of 16 B, exactly three per thread, no branches.  Chunk
// c = tid + 256 e belongs to the first tile for c < 384.  The (row, column, LDS offset) split is a loop invariant kept
// in a PairMap.  Rows past the end of the tensor are CLAMPED to its last row instead of zero-filled: every consumer
// masks the scores of such keys / queries to exactly zero probability, and zero times a finite operand is zero.
// (Guarding the loads with branches instead made the compiler serialise them: s_waitcnt vmcnt(0) at every join.)
struct PairRegs { u32x4 v[3]; };
struct PairMap {
  int row[3], coff[3], lo[3];
  bool second1;                 // slot 0 is always in the first tile, slot 2 in the second, slot 1 depends on the thread
};
__device__ __forceinline__ PairMap pair_map(int tid) {
  PairMap m;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int c = tid + 256 * e;
    const int sec = c >= 384, cc = c - 384 * sec;
    m.row[e] = cc / 12;
    const int ch = cc - m.row[e] * 12;
    m.coff[e] = ch * 8;
    m.lo[e] = sec * TILE_BYTES + pb_off(m.row[e], ch * 8);
  }
  m.second1 = tid >= 128;
  return m;
}
// both tiles row-major [nrows][D] (K | V)
__device__ __forceinline__ void pair_gload(PairRegs& t, const PairMap& m, const op_t* a, const op_t* b, int row0, int nrows) {
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const op_t* base = e == 0 ? a : (e == 2 ? b : (m.second1 ? b : a));
    const int r = min(row0 + m.row[e], nrows - 1);
    t.v[e] = *reinterpret_cast<const u32x4*>(base + (long)r * D + m.coff[e]);
  }
}
__device__ __forceinline__ void pair_lstore(const PairRegs& t, const PairMap& m, char* tiles) {
#pragma unroll
  for (int e = 0; e < 3; ++e) *reinterpret_cast<u32x4*>(tiles + m.lo[e]) = t.v[e];
}
__device__ __forceinline__ long tok_row(const PA& p, int b, int query) {
  return query < p.Lq ? (long)b * p.Lq + query : (long)p.B * p.Lq + b;
}
// Fragment addresses split into a per-lane part (computed once) and a compile-time part (tile row block u, K step, column
// block): the XOR swizzles of pb_off / bl_off / rm_off only ever flip a lane-dependent bit against an even constant, so
// every ds_read in the key loop is `buffer base + lane part` with an immediate offset instead of ~30 address VALU ops per tile.
struct FragLanes {
  int row96, tr96[2];      // [32][96] tiles (K, V, Q, dO): row fragment; transposed fragment of an even / odd column block
  int row64, tr64[2];      // [32][64] key-map tile
  int rm[2];               // [32][64] row-major rel tile(s) of the dK/dV kernel: 16-byte slot group 0 / 1
};
__device__ __forceinline__ FragLanes frag_lanes(int lane) {
  const int i = lane & 15, q4 = lane >> 4;
  const int b = (i >> 2) & 1, qb = q4 & 1, sw = (i >> 1) & 7;
  FragLanes f;
  f.row96 = ((i >> 2) * NCB + ((q4 >> 1) ^ b)) * 128 + (i & 3) * 32 + (q4 & 1) * 16;
  f.tr96[0] = (q4 * NCB + qb) * 128 + i * 8;
  f.tr96[1] = (q4 * NCB - qb) * 128 + i * 8;
  f.row64 = ((i >> 2) * 4 + ((q4 >> 1) ^ b)) * 128 + (i & 3) * 32 + (q4 & 1) * 16;
  f.tr64[0] = (q4 * 4 + qb) * 128 + i * 8;
  f.tr64[1] = (q4 * 4 - qb) * 128 + i * 8;
  f.rm[0] = i * 128 + ((q4 ^ sw) << 4);
  f.rm[1] = i * 128 + (((4 + q4) ^ sw) << 4);
  return f;
}
// = pb_row_frag(tile, 16 u + i, 4 ks + q4)
__device__ __forceinline__ opx8 row96_frag(const char* tile, const FragLanes& f, int u, int ks) {
  return *reinterpret_cast<const opx8*>(tile + f.row96 + u * (4 * NCB * 128) + ks * 256);
}
// = pb_tr_frag(tile, ct, lane)
__device__ __forceinline__ opx8 tr96_frag(const char* tile, const FragLanes& f, int ct) {
  const int o = f.tr96[ct & 1] + ct * 128;
  return tr_frag8(tile, o, o + 4 * NCB * 128);
}
// = bl_row_frag(tile, 16 u + i, 4 js + q4)
__device__ __forceinline__ opx8 row64_frag(const char* tile, const FragLanes& f, int u, int js) {
  return *reinterpret_cast<const opx8*>(tile + f.row64 + u * 2048 + js * 256);
}
// = bl_frag(tile, 0, ct, lane)
__device__ __forceinline__ opx8 tr64_frag(const char* tile, const FragLanes& f, int ct) {
  const int o = f.tr64[ct & 1] + ct * 128;
  return tr_frag8(tile, o, o + 2048);
}

// the (rel / scale) operand of one query: k slots j = 32 js + 8 q4 .. + 8, hi and lo halves; zero for the cls query
template <int NJS>
__device__ __forceinline__ void rel_frags(const PA& p, int bh, int query, int q4, opx8 (&rh)[NJS], opx8 (&rl)[NJS]) {
#pragma unroll
  for (int js = 0; js < NJS; ++js) {
    union { u32x4 u; opx8 v; } h, l;
    h.u = l.u = (u32x4){0u, 0u, 0u, 0u};
    if (query < p.Lq) {
      const op_t* row = p.relp + ((long)bh * p.Lq + query) * 2 * p.JP + (4 * js + q4) * 8;
      h.u = *reinterpret_cast<const u32x4*>(row);
      l.u = *reinterpret_cast<const u32x4*>(row + p.JP);
    }
    rh[js] = h.v; rl[js] = l.v;
  }
}

// ------------------------------------------------------------------------------------------------- key map
// tile image [32 keys][64 j] in the blocked layout: row-wise fragments (score MFMA) and transposed ones (d rel MFMA)
__global__ __launch_bounds__(256) void pattn_keymap_kernel(int kt, int kh, int kw, int ntiles, op_t* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;        // one 16-byte chunk: (tile, key row, 8 columns)
  if (idx >= ntiles * KT * 8) return;
  const int tile = idx / (KT * 8), r = (idx / 8) % KT, ch = idx % 8;
  const int key = tile * KT + r, Lk = kt * kh * kw;
  int j0 = -1, j1 = -1, j2 = -1;
  if (key < Lk) { j0 = (key / kw) % kh; j1 = kh + key % kw; j2 = kh + kw + key / (kw * kh); }
  opx8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = ch * 8 + e;
    v[e] = (op_t)((j == j0 || j == j1 || j == j2) ? 1.f : 0.f);
  }
  *reinterpret_cast<opx8*>(reinterpret_cast<char*>(out) + (long)tile * ET_BYTES + bl_off(r, ch * 8)) = v;
}

// ------------------------------------------------------------------------------------------------- forward
constexpr int FWD_BUF = 2 * TILE_BYTES + ET_BYTES;      // [K | V | E] per buffer
template <int NJS>
__global__ __launch_bounds__(256, 4) void pattn_fwd_kernel(PA p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * FWD_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q4 = lane >> 4;
  const FragLanes fl = frag_lanes(lane);
  PBlk blk;
  if (!pattn_block(p, blk)) return;                        // (block-uniform; before any barrier)
  const int bh = blk.y, b = bh / p.H, h = bh - b * p.H;
  const int Lq1 = p.Lq + 1, Lk1 = p.Lk + 1;
  const int query = blk.x * 64 + wave * 16 + i;
  const int qc = query < Lq1 ? query : Lq1 - 1;
  const op_t* qrow = p.q + ((long)bh * Lq1 + qc) * D;
  opx8 qf[3], rh[NJS], rl[NJS];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) qf[ks] = *reinterpret_cast<const opx8*>(qrow + ks * 32 + q4 * 8);
  rel_frags<NJS>(p, bh, query, q4, rh, rl);
  const op_t* kb = p.k + (long)bh * Lk1 * D;
  const op_t* vb = p.v + (long)bh * Lk1 * D;
  const int ntiles = (Lk1 + KT - 1) / KT;
  PairRegs rkv;
  const PairMap pm = pair_map(tid);
  u32x4 re;
  pair_gload(rkv, pm, kb, vb, 0, Lk1);
  re = *reinterpret_cast<const u32x4*>(p.keymap + tid * 16);
  pair_lstore(rkv, pm, smem);
  *reinterpret_cast<u32x4*>(smem + 2 * TILE_BYTES + tid * 16) = re;
  __syncthreads();

  const float c = p.scale * LOG2E;
  const bool qpatch = query < p.Lq;
  float m = -INFINITY, l = 0.f;
  f32x4 oacc[6];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < ntiles; ++t) {
    const char* Kb = smem + (t & 1) * FWD_BUF;
    const char* Vb = Kb + TILE_BYTES;
    const char* Eb = Vb + TILE_BYTES;
    const bool more = t + 1 < ntiles;
    if (more) {
      pair_gload(rkv, pm, kb, vb, (t + 1) * KT, Lk1);
      re = *reinterpret_cast<const u32x4*>(p.keymap + (long)(t + 1) * ET_BYTES + tid * 16);
    }
    float val[8];                                      // raw scores (K Q^T + E rel^T): logits / scale
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
        s = MFMA_16x16x32(row96_frag(Kb, fl, u, ks), qf[ks], s, 0, 0, 0);
#pragma unroll
      for (int js = 0; js < NJS; ++js) {
        const opx8 ef = row64_frag(Eb, fl, u, js);
        s = MFMA_16x16x32(ef, rh[js], s, 0, 0, 0);
        s = MFMA_16x16x32(ef, rl[js], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) val[u * 4 + r] = s[r];
    }
    if (!more) {                                       // only the last tile holds padding keys
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = t * KT + (e >> 2) * 16 + 4 * q4 + (e & 3);
        if (key >= Lk1) val[e] = -INFINITY;
      }
    }
    float mx = val[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, val[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx * c);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    const bool grew = mn > m;
    m = mn;
    float ps = 0.f;
    float pr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { pr[e] = __builtin_amdgcn_exp2f(fmaf(val[e], c, -mn)); ps += pr[e]; }
    l = l * alpha + ps;
    union { unsigned u[4]; opx8 v; } pf;
#pragma unroll
    for (int e = 0; e < 4; ++e) pf.u[e] = pack_opx2(pr[2 * e], pr[2 * e + 1]);
    if (__builtin_amdgcn_ballot_w64(grew)) {           // no lane's running max moved: alpha == 1 everywhere
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) oacc[dt] *= alpha;
    }
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) oacc[dt] = MFMA_16x16x32(tr96_frag(Vb, fl, dt), pf.v, oacc[dt], 0, 0, 0);
    if (more) {
      char* nb = smem + ((t + 1) & 1) * FWD_BUF;
      pair_lstore(rkv, pm, nb);
      *reinterpret_cast<u32x4*>(nb + 2 * TILE_BYTES + tid * 16) = re;
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (query < Lq1) {
    const float inv = 1.f / l;
    op_t* op = p.o + tok_row(p, b, query) * p.ldo + h * D + 4 * q4;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      opx4 ov;
      opx4 qv = (opx4){(op_t)0.f, (op_t)0.f, (op_t)0.f, (op_t)0.f};
      if (qpatch) qv = *reinterpret_cast<const opx4*>(qrow + 16 * dt + 4 * q4);
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (op_t)(oacc[dt][r] * inv + (float)qv[r]);
      *reinterpret_cast<opx4*>(op + 16 * dt) = ov;
    }
    if (q4 == 0) p.lse[(long)bh * Lq1 + query] = m + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
  }
}

// ------------------------------------------------------------------------------------------------- backward: dQ, d rel
template <int NJS>
__global__ __launch_bounds__(256, NJS == 1 ? 3 : 2) void pattn_bwd_q_kernel(PA p) {
  constexpr int NJT = NJS == 1 ? 2 : 3;                 // 16-column blocks of d rel (J <= 32 / J <= 40)
  __shared__ __attribute__((aligned(16))) char smem[2 * FWD_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q4 = lane >> 4;
  const FragLanes fl = frag_lanes(lane);
  PBlk blk;
  if (!pattn_block(p, blk)) return;                        // (block-uniform; before any barrier)
  const int bh = blk.y, b = bh / p.H, h = bh - b * p.H;
  const int Lq1 = p.Lq + 1, Lk1 = p.Lk + 1;
  const int query = blk.x * 64 + wave * 16 + i;
  const int qc = query < Lq1 ? query : Lq1 - 1;
  const bool qpatch = query < p.Lq;
  const op_t* qrow = p.q + ((long)bh * Lq1 + qc) * D;
  const long orow = tok_row(p, b, qc) * p.ldo + h * D;
  opx8 qf[3], df[3], rh[NJS], rl[NJS];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    qf[ks] = *reinterpret_cast<const opx8*>(qrow + ks * 32 + q4 * 8);
    df[ks] = *reinterpret_cast<const opx8*>(p.d_o + orow + ks * 32 + q4 * 8);
    const opx8 of = *reinterpret_cast<const opx8*>(p.o + orow + ks * 32 + q4 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)df[ks][e] * ((float)of[e] - (qpatch ? (float)qf[ks][e] : 0.f));
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  rel_frags<NJS>(p, bh, query, q4, rh, rl);
  const float lse2 = p.lse[(long)bh * Lq1 + qc];
  if (query < Lq1 && q4 == 0) p.delta[(long)bh * Lq1 + query] = dl;
  const op_t* kb = p.k + (long)bh * Lk1 * D;
  const op_t* vb = p.v + (long)bh * Lk1 * D;
  const int ntiles = (Lk1 + KT - 1) / KT;
  PairRegs rkv;
  const PairMap pm = pair_map(tid);
  u32x4 re;
  pair_gload(rkv, pm, kb, vb, 0, Lk1);
  re = *reinterpret_cast<const u32x4*>(p.keymap + tid * 16);
  pair_lstore(rkv, pm, smem);
  *reinterpret_cast<u32x4*>(smem + 2 * TILE_BYTES + tid * 16) = re;
  __syncthreads();
  const bool qpatch_any = blk.x * 64 < p.Lq;          // uniform: does this workgroup hold any patch query

  const float c = p.scale * LOG2E;
  f32x4 dq[6];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 dracc[NJT];
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) dracc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < ntiles; ++t) {
    const char* Kb = smem + (t & 1) * FWD_BUF;
    const char* Vb = Kb + TILE_BYTES;
    const char* Eb = Vb + TILE_BYTES;
    const bool more = t + 1 < ntiles;
    if (more) {
      pair_gload(rkv, pm, kb, vb, (t + 1) * KT, Lk1);
      re = *reinterpret_cast<const u32x4*>(p.keymap + (long)(t + 1) * ET_BYTES + tid * 16);
    }
    float ds[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        s = MFMA_16x16x32(row96_frag(Kb, fl, u, ks), qf[ks], s, 0, 0, 0);
        dp = MFMA_16x16x32(row96_frag(Vb, fl, u, ks), df[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int js = 0; js < NJS; ++js) {
        const opx8 ef = row64_frag(Eb, fl, u, js);
        s = MFMA_16x16x32(ef, rh[js], s, 0, 0, 0);
        s = MFMA_16x16x32(ef, rl[js], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse2));
        ds[u * 4 + r] = pr * (dp[r] - dl);
      }
    }
    if (!more) {                                       // padding keys of the last tile
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = t * KT + (e >> 2) * 16 + 4 * q4 + (e & 3);
        if (key >= Lk1) ds[e] = 0.f;
      }
    }
    union { unsigned u[4]; opx8 v; } sf;
#pragma unroll
    for (int e = 0; e < 4; ++e) sf.u[e] = pack_opx2(ds[2 * e], ds[2 * e + 1]);
#pragma unroll
    for (int dt = 0; dt < 6; ++dt)
      dq[dt] = MFMA_16x16x32(tr96_frag(Kb, fl, dt), sf.v, dq[dt], 0, 0, 0);
    if (qpatch_any) {                                  // d rel[query][j] += sum_key E[key][j] dS[key][query]
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) dracc[jt] = MFMA_16x16x32(tr64_frag(Eb, fl, jt), sf.v, dracc[jt], 0, 0, 0);
    }
    if (more) {
      char* nb = smem + ((t + 1) & 1) * FWD_BUF;
      pair_lstore(rkv, pm, nb);
      *reinterpret_cast<u32x4*>(nb + 2 * TILE_BYTES + tid * 16) = re;
    }
    __syncthreads();
  }
  if (query < Lq1) {
    op_t* op = p.dq + ((long)bh * Lq1 + query) * D + 4 * q4;
    const op_t* dop = p.d_o + orow + 4 * q4;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      opx4 ov;
      opx4 dv = (opx4){(op_t)0.f, (op_t)0.f, (op_t)0.f, (op_t)0.f};
      if (qpatch) dv = *reinterpret_cast<const opx4*>(dop + 16 * dt);     // residual pooling: d out / d q = 1
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (op_t)(dq[dt][r] * p.scale + (float)dv[r]);
      *reinterpret_cast<opx4*>(op + 16 * dt) = ov;
    }
  }
  // dracc[jt][r] = d rel[query][16 jt + 4 q4 + r] (cls-query rows hold the un-biased scores' dS: not part of rel)
  if (qpatch) {
    float* dr = p.drel + ((long)bh * p.Lq + query) * p.J;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * q4 + r;
        if (j < p.J) dr[j] = dracc[jt][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------- backward: dK, dV
// one wave = 16 keys (lane i owns key column i), workgroup = 64 keys; queries streamed in tiles of 32 (Q, dO and the
// rel operand rows in LDS); the lane's key-map row E[key][:] is the constant MFMA operand of the bias term
template <int NJS>
__global__ __launch_bounds__(256, NJS == 1 ? 3 : 2) void pattn_bwd_kv_kernel(PA p) {
  constexpr int RT_BYTES = NJS * 4096;                  // rel tile: [32 queries][hi | lo] rows of 128 B (NJS such images)
  constexpr int KV_BUF = 2 * TILE_BYTES + RT_BYTES;     // [Q | dO | R]
  __shared__ __attribute__((aligned(16))) char smem[2 * KV_BUF];
  __shared__ __attribute__((aligned(16))) float lse_s[2][KT], dl_s[2][KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q4 = lane >> 4;
  const FragLanes fl = frag_lanes(lane);
  PBlk blk;
  if (!pattn_block(p, blk)) return;                        // (block-uniform; before any barrier)
  const int bh = blk.y, b = bh / p.H, h = bh - b * p.H;
  const int Lq1 = p.Lq + 1, Lk1 = p.Lk + 1;
  const int key = blk.x * 64 + wave * 16 + i;
  const int kc = key < Lk1 ? key : Lk1 - 1;
  opx8 kf[3], vf[3], ef[NJS];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    kf[ks] = *reinterpret_cast<const opx8*>(p.k + ((long)bh * Lk1 + kc) * D + ks * 32 + q4 * 8);
    vf[ks] = *reinterpret_cast<const opx8*>(p.v + ((long)bh * Lk1 + kc) * D + ks * 32 + q4 * 8);
  }
#pragma unroll
  for (int js = 0; js < NJS; ++js)
    ef[js] = *reinterpret_cast<const opx8*>(p.keymap + (long)(kc / KT) * ET_BYTES + bl_off(kc % KT, (js * 4 + q4) * 8));
  const int ntiles_all = (Lq1 + KT - 1) / KT;
  const int per = (ntiles_all + p.gz - 1) / p.gz;                    // query tiles of this z-slice
  const int tbeg = blk.z * per, ntiles = min(ntiles_all, tbeg + per);
  const op_t* qb = p.q + (long)bh * Lq1 * D;

  // Q | dO pair: the dO rows are gathered from the token-major activation (rows (b, query), the cls row last; columns
  // h*96 ..).  Rows past the last query are clamped to the cls row: their lse is +inf below, so P = dS = 0.
  const PairMap pm = pair_map(tid);
  auto qd_gload = [&](PairRegs& t, int row0) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int r = min(row0 + pm.row[e], Lq1 - 1);
      const op_t* qa = qb + (long)r * D + pm.coff[e];
      const op_t* da = p.d_o + tok_row(p, b, r) * p.ldo + h * D + pm.coff[e];
      const op_t* src = e == 0 ? qa : (e == 2 ? da : (pm.second1 ? da : qa));
      t.v[e] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  // the rel rows / lse / delta of a query tile are one contiguous run in HBM: fetched to registers with the tiles
  // (latency under the MFMAs), stored to LDS after the step.  A row is 8 NJS chunks of 16 B: chunk cid = hi/lo * 4 NJS
  // + 4 js + q4 lives in image cid / 8 at the swizzled slot cid % 8 of its row.  The cls query and the rows past it
  // have no bias: zero (selected after a clamped load, no branch).
  struct SideRegs { u32x4 r[NJS]; float l, d; };
  const op_t* relb = p.relp + (long)bh * p.Lq * 2 * p.JP;
  int side_lo[NJS], side_row[NJS], side_col[NJS];              // loop invariants of this thread's chunks
#pragma unroll
  for (int e = 0; e < NJS; ++e) {
    const int cidx = tid + 256 * e;
    const int row = cidx / (8 * NJS), cid = cidx - row * (8 * NJS);
    side_row[e] = row; side_col[e] = cid * 8;
    side_lo[e] = (cid >> 3) * 4096 + rm_off(row, cid & 7);
  }
  auto side_gload = [&](SideRegs& sr, int row0) {             // loads only: the selects wait in side_lstore, after the MFMAs
#pragma unroll
    for (int e = 0; e < NJS; ++e)
      sr.r[e] = *reinterpret_cast<const u32x4*>(relb + (long)min(row0 + side_row[e], p.Lq - 1) * 2 * p.JP + side_col[e]);
    const long si = (long)bh * Lq1 + min(row0 + (tid & (KT - 1)), Lq1 - 1);
    sr.l = p.lse[si]; sr.d = p.delta[si];
  };
  auto side_lstore = [&](const SideRegs& sr, char* rt, int buf, int row0) {
#pragma unroll
    for (int e = 0; e < NJS; ++e)
      *reinterpret_cast<u32x4*>(rt + side_lo[e]) = row0 + side_row[e] < p.Lq ? sr.r[e] : (u32x4){0u, 0u, 0u, 0u};
    if (tid < KT) {
      const bool ok = row0 + tid < Lq1;                        // rows past the last query: exp2(x - inf) = 0
      lse_s[buf][tid] = ok ? sr.l : INFINITY; dl_s[buf][tid] = ok ? sr.d : 0.f;
    }
  };
  PairRegs rqd;
  SideRegs rs;
  qd_gload(rqd, tbeg * KT);
  side_gload(rs, tbeg * KT);
  {
    char* b0 = smem + (tbeg & 1) * KV_BUF;
    pair_lstore(rqd, pm, b0);
    side_lstore(rs, b0 + 2 * TILE_BYTES, tbeg & 1, tbeg * KT);
  }
  __syncthreads();

  const float c = p.scale * LOG2E;
  f32x4 dk[6], dv[6];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  for (int t = tbeg; t < ntiles; ++t) {
    const int buf = t & 1;
    const char* Qb = smem + buf * KV_BUF;
    const char* Db = Qb + TILE_BYTES;
    const char* Rb = Db + TILE_BYTES;
    if (t + 1 < ntiles) {
      qd_gload(rqd, (t + 1) * KT);
      side_gload(rs, (t + 1) * KT);
    }
    float pr[8], ds[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        s = MFMA_16x16x32(row96_frag(Qb, fl, u, ks), kf[ks], s, 0, 0, 0);
        dp = MFMA_16x16x32(row96_frag(Db, fl, u, ks), vf[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int js = 0; js < NJS; ++js)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          // chunk cid = hl * 4 NJS + 4 js + q4 of row 16 u + i: image cid / 8, swizzled slot cid % 8
          const opx8 rf = *reinterpret_cast<const opx8*>(Rb + fl.rm[NJS == 1 ? hl : js] + u * 2048 + (NJS == 1 ? 0 : hl * 4096));
          s = MFMA_16x16x32(rf, ef[js], s, 0, 0, 0);
        }
      // s[r] = S[query = t*32 + 16u + 4*q4 + r][key] / scale
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(&lse_s[buf][u * 16 + 4 * q4]);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(&dl_s[buf][u * 16 + 4 * q4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -l4[r]));
        pr[u * 4 + r] = pv;
        ds[u * 4 + r] = pv * (dp[r] - d4[r]);
      }
    }
    union { unsigned u[4]; opx8 v; } pf, sf;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pf.u[e] = pack_opx2(pr[2 * e], pr[2 * e + 1]);
      sf.u[e] = pack_opx2(ds[2 * e], ds[2 * e + 1]);
    }
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      dv[dt] = MFMA_16x16x32(tr96_frag(Db, fl, dt), pf.v, dv[dt], 0, 0, 0);
      dk[dt] = MFMA_16x16x32(tr96_frag(Qb, fl, dt), sf.v, dk[dt], 0, 0, 0);
    }
    if (t + 1 < ntiles) {
      char* nb = smem + ((t + 1) & 1) * KV_BUF;
      pair_lstore(rqd, pm, nb);
      side_lstore(rs, nb + 2 * TILE_BYTES, (t + 1) & 1, (t + 1) * KT);
    }
    __syncthreads();
  }
  if (key < Lk1) {
    const long nkv = (long)p.gy * Lk1 * D;
    float* kp = p.kv_part + ((long)blk.z * 2) * nkv + ((long)bh * Lk1 + key) * D + 4 * q4;
    float* vp = kp + nkv;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      *reinterpret_cast<f32x4*>(kp + 16 * dt) = dk[dt];
      *reinterpret_cast<f32x4*>(vp + 16 * dt) = dv[dt];
    }
  }
}

// dK = scale * sum_z partial, dV = sum_z partial  (bf16 out)
__global__ __launch_bounds__(256) void pattn_kv_reduce_kernel(const float* __restrict__ part, int nsplit, long nkv,
                                                              float scale, op_t* __restrict__ dk, op_t* __restrict__ dv) {
  const long n4 = nkv >> 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < 2 * n4; idx += (long)gridDim.x * 256) {
    const int which = idx >= n4;
    const long e = (idx - which * n4) * 4;
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nsplit; ++z) a += *reinterpret_cast<const f32x4*>(part + ((long)z * 2 + which) * nkv + e);
    const float sc = which ? 1.f : scale;
    opx4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (op_t)(a[r] * sc);
    *reinterpret_cast<opx4*>((which ? dv : dk) + e) = o;
  }
}

int keymap_tiles(int64_t kt, int64_t kh, int64_t kw) { return (int)((kt * kh * kw + 1 + KT - 1) / KT); }

int fill(PA& p, const void* q, const void* k, const void* v, const void* relp, const void* keymap, int64_t B, int64_t H,
         int64_t Lq, int64_t kt, int64_t kh, int64_t kw, float scale, int64_t ldo) {
  if (!q || !k || !v || !relp || !keymap || B <= 0 || H <= 0 || Lq <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || (ldo % 8) ||
      ldo < H * D)
    return PVRL_EINVAL;
  if (kh + kw + kt > JMAX || kt * kh * kw + 1 > MAXKEYS) return PVRL_EINVAL;
  p.q = (const op_t*)q; p.k = (const op_t*)k; p.v = (const op_t*)v; p.relp = (const op_t*)relp;
  p.keymap = (const char*)keymap;
  p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)(kt * kh * kw);
  p.kt = (int)kt; p.kh = (int)kh; p.kw = (int)kw; p.J = (int)(kh + kw + kt);
  p.JP = (int)pvrl_mvit_rel_width(kt, kh, kw);
  p.scale = scale; p.ldo = ldo;
  return PVRL_OK;
}

}  // namespace

extern "C" int64_t pvrl_mvit_rel_width(int64_t kt, int64_t kh, int64_t kw) { return kh + kw + kt <= 32 ? 32 : 64; }

extern "C" int64_t pvrl_mvit_attn_keymap_bytes(int64_t kt, int64_t kh, int64_t kw) {
  if (kt <= 0 || kh <= 0 || kw <= 0 || kt * kh * kw + 1 > MAXKEYS) return PVRL_EINVAL;
  return (int64_t)keymap_tiles(kt, kh, kw) * ET_BYTES;
}

extern "C" int pvrl_mvit_attn_keymap(int64_t kt, int64_t kh, int64_t kw, void* keymap, void* stream) {
  if (!keymap || kt <= 0 || kh <= 0 || kw <= 0 || kt * kh * kw + 1 > MAXKEYS || kh + kw + kt > JMAX) return PVRL_EINVAL;
  const int nt = keymap_tiles(kt, kh, kw);
  hipLaunchKernelGGL(pattn_keymap_kernel, dim3((unsigned)cdiv((long)nt * KT * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (int)kt, (int)kh, (int)kw, nt, (op_t*)keymap);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_mvit_attn_fwd(const void* q, const void* k, const void* v, const void* relp, const void* keymap,
                                  int64_t B, int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw, float scale,
                                  void* o, int64_t ldo, float* lse, void* stream) {
  PA p = {};
  if (!o || !lse || fill(p, q, k, v, relp, keymap, B, H, Lq, kt, kh, kw, scale, ldo)) return PVRL_EINVAL;
  p.o = (op_t*)o; p.lse = lse;
  const dim3 grid(pattn_grid(p, cdiv(Lq + 1, 64), B * H, 1));
  if (p.JP == 32) hipLaunchKernelGGL(pattn_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pattn_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

// dK / dV: query tiles are shared out so that a workgroup streams >= ~2048 queries.  (Measured: more, smaller workgroups --
// ~3,000 per launch instead of 896 at Lq = 1568 -- lose 7 %: the extra fp32 partials and their reduction cost more than
// the fuller last wave of workgroups gains.)
static int kv_splits(int64_t BH, int64_t Lq, int64_t Lk) {
  (void)BH; (void)Lk;
  int64_t s = (Lq + 1 + 2047) / 2048;
  return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}

extern "C" int64_t pvrl_mvit_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw) {
  return (int64_t)kv_splits(B * H, Lq, kt * kh * kw) * 2 * B * H * (kt * kh * kw + 1) * D * (int64_t)sizeof(float);
}

extern "C" int pvrl_mvit_attn_bwd(const void* q, const void* k, const void* v, const void* relp, const void* keymap,
                                  int64_t B, int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw, float scale,
                                  const void* o, const void* d_o, int64_t ldo, const float* lse, float* delta, void* dq,
                                  void* dk, void* dv, float* drel, void* workspace, int64_t workspace_bytes, void* stream) {
  PA p = {};
  if (!o || !d_o || !lse || !delta || !dq || !dk || !dv || !drel || !workspace ||
      fill(p, q, k, v, relp, keymap, B, H, Lq, kt, kh, kw, scale, ldo))
    return PVRL_EINVAL;
  if (workspace_bytes < pvrl_mvit_attn_bwd_workspace_bytes(B, H, Lq, kt, kh, kw)) return PVRL_EINVAL;
  p.kv_part = (float*)workspace;
  p.o = (op_t*)o; p.d_o = (const op_t*)d_o; p.lse = (float*)lse; p.delta = delta;
  p.dq = (op_t*)dq; p.dk = (op_t*)dk; p.dv = (op_t*)dv; p.drel = drel;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq(pattn_grid(p, cdiv(Lq + 1, 64), B * H, 1));
  if (p.JP == 32) hipLaunchKernelGGL(pattn_bwd_q_kernel<1>, gq, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(pattn_bwd_q_kernel<2>, gq, dim3(256), 0, s, p);
  PVRL_LAUNCH_CHECK();
  const int ns = kv_splits(B * H, Lq, p.Lk);
  const dim3 gk(pattn_grid(p, cdiv(p.Lk + 1, 64), B * H, ns));
  if (p.JP == 32) hipLaunchKernelGGL(pattn_bwd_kv_kernel<1>, gk, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(pattn_bwd_kv_kernel<2>, gk, dim3(256), 0, s, p);
  PVRL_LAUNCH_CHECK();
  const long nkv = (long)B * H * (p.Lk + 1) * D;
  hipLaunchKernelGGL(pattn_kv_reduce_kernel, dim3((unsigned)cdiv(2 * (nkv >> 2), 256)), dim3(256), 0, s, p.kv_part, ns,
                     nkv, scale, p.dk, p.dv);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
