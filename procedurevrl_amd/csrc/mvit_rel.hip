// Decomposed relative-position terms of MViTv2's pooling attention (reference: cal_rel_pos_spatial / cal_rel_pos_temporal,
// lib/models/slowfast_mvit/attention.py:67-159):
//     rel[bh][q][j] = Q[bh][q] . R_j(q),   R_j(q) = rel_pos_h[idx_h[y(q)][j]]  (j <  kh)
//                                                   rel_pos_w[idx_w[x(q)][j-kh]]  (j <  kh+kw)
//                                                   rel_pos_t[idx_t[t(q)][j-kh-kw]]
// and their gradients.  Along one axis every query with the same coordinate v multiplies the SAME k_n x 96 matrix
// R_a(v), so the work is cut into (axis, v, run of queries) units and each unit is a small MFMA GEMM:
//   forward   D[j][query]  = R_a(v)[j][:] . Q[query][:]        one lane owns one query; R_a(v) stays in registers
//   table     D[j][c]     += drel[query][j] * Q[query][c]      32 queries per MFMA step, Q^T through ds_read_b64_tr_b16
// The fp32 tables (forward) and the fp32 d rel (table gradient) enter the MFMA as hi + lo 16-bit pairs, so the products
// keep ~16 mantissa bits (the bf16 q operand is exact): results agree with fp32 FMA arithmetic to ~1e-6.
// dQ += d rel . R stays a VALU kernel: all three axes must be summed in fp32 before the single rounding into the
// 16-bit dQ, and the 16 queries of an MFMA tile never share R along all three axes.
#include "attn_common.h"
#include "../../include/pvrl.h"

namespace {

constexpr int HD = PB_D;          // 96
constexpr int REL_KMAX = 16;      // k_h, k_w, k_t <= 16: one MFMA row block

inline unsigned grid_for(long total, int per_block = 256) {
  long b = (total + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 65535L * 16) b = 65535L * 16;
  return (unsigned)b;
}

struct RelGeom {
  int BH, qt, qh, qw, kt, kh, kw;   // J = kh + kw + kt columns: [0,kh) height, [kh,kh+kw) width, then time
};

// work list: axis a owns units [first[a], first[a+1]); unit -> (v = coordinate, ck = chunk of `per` queries among the
// BH * Lq / qn[a] queries that have coordinate v on that axis)
struct RelAxes {
  const float* R[3];
  const int* idx[3];
  float* part[3];          // table gradient: [chunk][v][j][96] partial sums per axis
  int qn[3], kn[3], off[3], chunks[3], first[4];
  int per;
};

__device__ __forceinline__ int rel_unit(const RelAxes& ax, int u, int& v, int& ck) {
  const int a = u >= ax.first[2] ? 2 : (u >= ax.first[1] ? 1 : 0);
  u -= ax.first[a];
  v = u / ax.chunks[a];
  ck = u - v * ax.chunks[a];
  return a;
}
// the o-th query (o in [0, Lq / qn)) with coordinate v on axis a
__device__ __forceinline__ int rel_query(const RelGeom& g, int a, int v, int o) {
  if (a == 0) { const int t = o / g.qw, x = o - t * g.qw; return (t * g.qh + v) * g.qw + x; }
  if (a == 1) return o * g.qw + v;                          // o = t*qh + y
  return v * g.qh * g.qw + o;
}
__device__ __forceinline__ void split8(const f32x4 r0, const f32x4 r1, opx8& hi, opx8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (op_t)r0[e]; lo[e] = (op_t)(r0[e] - (float)hi[e]);
    hi[4 + e] = (op_t)r1[e]; lo[4 + e] = (op_t)(r1[e] - (float)hi[4 + e]);
  }
}
__device__ __forceinline__ void wave_lds_sync() {   // LDS traffic of one wave is executed in order: only the compiler needs the fence
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------- forward
// Output = the operand form the attention kernels consume (attn_pool.hip): relp[bh][q][hi JP | lo JP], the 16-bit pair
// hi + lo = out_scale * rel[bh][q][j] (out_scale = 1 / attention scale), zero for J <= j < JP.
__global__ __launch_bounds__(256) void rel_fwd_kernel(const op_t* __restrict__ Q, RelGeom g, RelAxes ax, float out_scale,
                                                      int JP, op_t* __restrict__ relp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q4 = lane >> 4;
  int v, ck;
  const int a = rel_unit(ax, blockIdx.x, v, ck);
  const int kn = ax.kn[a];
  const int J = g.kh + g.kw + g.kt, Lq = g.qt * g.qh * g.qw;
  const int n_other = Lq / ax.qn[a];
  const int n = g.BH * n_other;
  opx8 ah[3], al[3];                               // R_a(v): row j = i, channels 32 ks + 8 q4 .. + 8
  {
    const bool jr = i < kn;
    const float* row = ax.R[a] + (long)(jr ? ax.idx[a][v * kn + i] : 0) * HD + 8 * q4;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = r0;
      if (jr) { r0 = *reinterpret_cast<const f32x4*>(row + 32 * ks); r1 = *reinterpret_cast<const f32x4*>(row + 32 * ks + 4); }
      split8(r0, r1, ah[ks], al[ks]);
    }
  }
  const int i0 = ck * ax.per, i1 = min(n, i0 + ax.per);
  const int off = ax.off[a];
  for (int base = i0 + wave * 16; base < i1; base += 64) {
    const int item = base + i;
    const bool valid = item < i1;
    const int it2 = valid ? item : i1 - 1;
    const int bh = it2 / n_other, o = it2 - bh * n_other;
    const int q = rel_query(g, a, v, o);
    const op_t* qrow = Q + ((long)bh * (Lq + 1) + q) * HD + 8 * q4;
    opx8 qf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) qf[ks] = *reinterpret_cast<const opx8*>(qrow + 32 * ks);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      acc = MFMA_16x16x32(ah[ks], qf[ks], acc, 0, 0, 0);
      acc = MFMA_16x16x32(al[ks], qf[ks], acc, 0, 0, 0);
    }
    if (valid) {                                   // acc[r] = rel[query][off + 4 q4 + r]
      op_t* dst = relp + ((long)bh * Lq + q) * 2 * JP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q4 + r < kn) {
          const float x = acc[r] * out_scale;
          const op_t hi = (op_t)x;
          dst[off + 4 * q4 + r] = hi;
          dst[JP + off + 4 * q4 + r] = (op_t)(x - (float)hi);
        }
      if (a == 2)                                  // the time axis comes last: it also clears the padding columns
        for (int j = J + q4; j < JP; j += 4) { dst[j] = (op_t)0.f; dst[JP + j] = (op_t)0.f; }
    }
  }
}

// ------------------------------------------------------------------------------------------------- dQ
// dQ[bh][q][c] += sum_j drel[bh][q][j] * R_j(q)[c]     (dQ is the 16-bit gradient written by the attention backward)
// one thread per (q, CPT channels): the d rel value and the table index of a j are loaded once per CPT / 4 row chunks
// TAB_LDS (round 3): the three tables (<= 240 rows of 96 floats) and their index maps are copied into LDS once per workgroup of 16 waves
// and every gather reads LDS -- the kernel was bound by its vector-memory instruction count (J x 3 sixteen-byte gathers per lane from
// L1 / L2); what is left in the vector-memory path is the d rel row and the read-modify-write of dQ.
constexpr int RELQ_MAXROWS = 240, RELQ_MAXIDX = 1024;
template <int CPT, bool TAB_LDS>
__global__ __launch_bounds__(TAB_LDS ? 1024 : 256) void rel_bwd_q_kernel(const float* __restrict__ drel, RelGeom g,
                                                        const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                        const float* __restrict__ Rt, const int* __restrict__ ih,
                                                        const int* __restrict__ iw, const int* __restrict__ it,
                                                        int nrh, int nrw, int nrt, op_t* __restrict__ dQ) {
  constexpr int NV = CPT / 4, TPQ = HD / CPT;
  constexpr int NT = TAB_LDS ? 1024 : 256;
  __shared__ __attribute__((aligned(16))) float tab[TAB_LDS ? RELQ_MAXROWS * HD : 4];
  __shared__ int tix[TAB_LDS ? 3 * RELQ_MAXIDX : 4];
  if constexpr (TAB_LDS) {
    const int n4h = nrh * (HD / 4), n4w = nrw * (HD / 4), n4t = nrt * (HD / 4);
    for (int i = threadIdx.x; i < n4h + n4w + n4t; i += NT) {
      const f32x4 v = i < n4h ? reinterpret_cast<const f32x4*>(Rh)[i]
                              : (i < n4h + n4w ? reinterpret_cast<const f32x4*>(Rw)[i - n4h] : reinterpret_cast<const f32x4*>(Rt)[i - n4h - n4w]);
      reinterpret_cast<f32x4*>(tab)[i] = v;
    }
    for (int i = threadIdx.x; i < g.qh * g.kh; i += NT) tix[i] = ih[i];
    for (int i = threadIdx.x; i < g.qw * g.kw; i += NT) tix[RELQ_MAXIDX + i] = iw[i];
    for (int i = threadIdx.x; i < g.qt * g.kt; i += NT) tix[2 * RELQ_MAXIDX + i] = it[i];
    __syncthreads();
  }
  const int J = g.kh + g.kw + g.kt;
  const int Lq = g.qt * g.qh * g.qw;
  const unsigned total = (unsigned)((long)g.BH * Lq * TPQ);          // < 2^31 (checked by the launcher): 32-bit index arithmetic
  for (unsigned idx = blockIdx.x * NT + threadIdx.x; idx < total; idx += gridDim.x * NT) {
    const int c = (int)(idx % TPQ) * CPT;
    const long bq = idx / TPQ;
    const int q = (int)((unsigned)bq % (unsigned)Lq);
    const long bh = (unsigned)bq / (unsigned)Lq;
    const int x = q % g.qw, y = (q / g.qw) % g.qh, t = q / (g.qw * g.qh);
    const float* d = drel + bq * J;
    f32x4 a[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) a[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The kernel is bound by its vector-memory instruction count, not by latency (batching the loads of eight j changed
    // nothing): the TPQ = 8 lanes of a query share each d rel value and table index -- lane k of the group loads entry
    // j0 + k of a chunk of eight, ds_bpermute hands them round -- so a chunk costs 2 + 8 NV loads per lane instead of 16 + 8 NV.
    static_assert(TPQ == 8, "one query = one aligned group of eight lanes");
    const int k8 = threadIdx.x & 7;
    auto axis = [&](const float* R, const int* ix, int kn, const float* dj) {
      for (int j0 = 0; j0 < kn; j0 += 8) {
        const int jk = min(j0 + k8, kn - 1);
        const float wk = j0 + k8 < kn ? dj[jk] : 0.f;
        const int rk = ix[jk];
        constexpr int GB = TAB_LDS ? 4 : 8;          // rows gathered per batch (LDS latency needs less in flight; 128-VGPR budget at 16 waves)
#pragma unroll
        for (int e0 = 0; e0 < 8; e0 += GB) {
          f32x4 rows[GB][NV];
          float w[GB];
#pragma unroll
          for (int e = 0; e < GB; ++e) {
            w[e] = __shfl(wk, e0 + e, 8);
            const f32x4* row = reinterpret_cast<const f32x4*>(R + (long)__shfl(rk, e0 + e, 8) * HD + c);
#pragma unroll
            for (int v = 0; v < NV; ++v) rows[e][v] = row[v];
          }
#pragma unroll
          for (int e = 0; e < GB; ++e)
#pragma unroll
            for (int v = 0; v < NV; ++v) a[v] += w[e] * rows[e][v];
        }
      }
    };
    if constexpr (TAB_LDS) {
      axis(tab, tix + y * g.kh, g.kh, d);
      axis(tab + nrh * HD, tix + RELQ_MAXIDX + x * g.kw, g.kw, d + g.kh);
      axis(tab + (nrh + nrw) * HD, tix + 2 * RELQ_MAXIDX + t * g.kt, g.kt, d + g.kh + g.kw);
    } else {
      axis(Rh, ih + y * g.kh, g.kh, d);
      axis(Rw, iw + x * g.kw, g.kw, d + g.kh);
      axis(Rt, it + t * g.kt, g.kt, d + g.kh + g.kw);
    }
    op_t* p = dQ + (bh * (Lq + 1) + q) * HD + c;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      opx4 v = *reinterpret_cast<opx4*>(p + 4 * e);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (op_t)((float)v[r] + a[e][r]);
      *reinterpret_cast<opx4*>(p + 4 * e) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------- table gradient
// part_a[ck][v][j][c] = sum over the unit's queries of drel[query][off + j] * Q[query][c].  Every wave streams its own
// 32-query tiles (own LDS tile, wave-level synchronisation only); the four accumulators meet in LDS at the end.
constexpr int REL_TILE_BYTES = 32 * HD * 2;       // 6 KiB
__global__ __launch_bounds__(256) void rel_bwd_table_kernel(const float* __restrict__ drel, const op_t* __restrict__ Q,
                                                            RelGeom g, RelAxes ax) {
  __shared__ __attribute__((aligned(16))) char smem[4 * REL_TILE_BYTES];      // = 4 waves x [16][96] fp32 for the final sum
  __shared__ int rowq_s[4][32], rowd_s[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q4 = lane >> 4;
  int v, ck;
  const int a = rel_unit(ax, blockIdx.x, v, ck);
  const int kn = ax.kn[a], off = ax.off[a];
  const int J = g.kh + g.kw + g.kt, Lq = g.qt * g.qh * g.qw;
  const int n_other = Lq / ax.qn[a];
  const int n = g.BH * n_other;
  const int i0 = ck * ax.per, i1 = min(n, i0 + ax.per);
  char* tile = smem + wave * REL_TILE_BYTES;
  f32x4 acc[6];
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool jr = i < kn;

  for (int base = i0 + wave * 32; base < i1; base += 128) {
    if (lane < 32) {
      const int item = base + lane;
      int rq = -1, rd = -1;
      if (item < i1) {
        const int bh = item / n_other, o = item - bh * n_other;
        const int q = rel_query(g, a, v, o);
        rq = bh * (Lq + 1) + q; rd = bh * Lq + q;
      }
      rowq_s[wave][lane] = rq; rowd_s[wave][lane] = rd;
    }
    wave_lds_sync();
    u32x4 qv[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {                                   // 32 rows x 12 chunks of 16 B
      const int c = lane + 64 * e;
      const int row = c / 12, ch = c - row * 12;
      const int rq = rowq_s[wave][row];
      qv[e] = (u32x4){0u, 0u, 0u, 0u};
      if (rq >= 0) qv[e] = *reinterpret_cast<const u32x4*>(Q + (long)rq * HD + ch * 8);
    }
    // A operand: d rel of the tile's queries in the k order of pb_tr_frag (rows 4q4+e | 16+4q4+e), column j = i
    float dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = (e < 4 ? 0 : 12) + 4 * q4 + e;
      const int rd = rowd_s[wave][row];
      dv[e] = (jr && rd >= 0) ? drel[(long)rd * J + off + i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int c = lane + 64 * e;
      const int row = c / 12, ch = c - row * 12;
      *reinterpret_cast<u32x4*>(tile + pb_off(row, ch * 8)) = qv[e];
    }
    opx8 dh, dl;
    split8((f32x4){dv[0], dv[1], dv[2], dv[3]}, (f32x4){dv[4], dv[5], dv[6], dv[7]}, dh, dl);
    wave_lds_sync();
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
      const opx8 b = pb_tr_frag(tile, ct, lane);
      acc[ct] = MFMA_16x16x32(dh, b, acc[ct], 0, 0, 0);
      acc[ct] = MFMA_16x16x32(dl, b, acc[ct], 0, 0, 0);
    }
    wave_lds_sync();                                                 // the next tile overwrites what was just read
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                       // [wave][16][96]; acc[ct][r] = D[j = 4 q4 + r][c = 16 ct + i]
#pragma unroll
  for (int ct = 0; ct < 6; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * q4 + r) * HD + 16 * ct + i] = acc[ct][r];
  __syncthreads();
  float* mine = ax.part[a] + ((long)ck * ax.qn[a] + v) * kn * HD;
  for (int e = tid; e < kn * HD; e += 256)
    mine[e] = (red[e] + red[16 * HD + e]) + (red[32 * HD + e] + red[48 * HD + e]);
}

// dR_a[r][c] += sum over the (v, j) pairs whose table index is r, in (v, j) order, of the sum over chunks of part_a
// (deterministic).  One block per table row of the three tables; 4 chunk slices x 96 channels.
struct RelReduce {
  const float* part[3]; const int* idx[3]; float* dR[3];
  int npairs[3], chunks[3], first[4];          // first: block -> (axis, table row)
};
__global__ __launch_bounds__(4 * HD) void rel_table_reduce_kernel(RelReduce rr) {
  __shared__ float red[4][HD];
  const int b = blockIdx.x;
  const int a = b >= rr.first[2] ? 2 : (b >= rr.first[1] ? 1 : 0);
  const int r = b - rr.first[a];
  const int c = threadIdx.x % HD, sl = threadIdx.x / HD;
  const int np = rr.npairs[a], nch = rr.chunks[a];
  const float* part = rr.part[a];
  const int* idx = rr.idx[a];
  float s = 0.f;
  for (int e = 0; e < np; ++e) {
    if (idx[e] != r) continue;
    float s0 = 0.f, s1 = 0.f;
    int ch = sl;
    for (; ch + 4 < nch; ch += 8) { s0 += part[((long)ch * np + e) * HD + c]; s1 += part[((long)(ch + 4) * np + e) * HD + c]; }
    if (ch < nch) s0 += part[((long)ch * np + e) * HD + c];
    s += s0 + s1;
  }
  red[sl][c] = s;
  __syncthreads();
  if (sl == 0) rr.dR[a][(long)r * HD + c] += (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

int rel_geom(RelGeom& g, int64_t BH, int64_t qt, int64_t qh, int64_t qw, int64_t kt, int64_t kh, int64_t kw) {
  if (BH <= 0 || qt <= 0 || qh <= 0 || qw <= 0 || kt <= 0 || kh <= 0 || kw <= 0) return PVRL_EINVAL;
  if (kh > REL_KMAX || kw > REL_KMAX || kt > REL_KMAX) return PVRL_EINVAL;
  if (BH * qt * qh * qw >= (1LL << 31) / HD) return PVRL_EINVAL;          // row indices are 32-bit in the kernels
  g.BH = (int)BH; g.qt = (int)qt; g.qh = (int)qh; g.qw = (int)qw; g.kt = (int)kt; g.kh = (int)kh; g.kw = (int)kw;
  return PVRL_OK;
}

// units of `per` queries: ~1500 workgroups over the three axes, at least 512 queries each (the R_a(v) fragments and the
// final cross-wave sum are per-unit overheads), `per` a multiple of 128 (4 waves x 32-query tiles)
void rel_axes(RelAxes& ax, const RelGeom& g) {
  const long NQ = (long)g.BH * g.qt * g.qh * g.qw;
  long per = (3 * NQ / 1536 + 127) / 128 * 128;
  if (per < 512) per = 512;
  if (per > 8192) per = 8192;
  ax.per = (int)per;
  const int qn[3] = {g.qh, g.qw, g.qt}, kn[3] = {g.kh, g.kw, g.kt};
  int first = 0, off = 0;
  for (int a = 0; a < 3; ++a) {
    ax.qn[a] = qn[a]; ax.kn[a] = kn[a]; ax.off[a] = off;
    off += kn[a];
    ax.chunks[a] = cdiv(NQ / qn[a], per);
    ax.first[a] = first;
    first += qn[a] * ax.chunks[a];
  }
  ax.first[3] = first;
}

}  // namespace

extern "C" int pvrl_mvit_rel_fwd(const void* Q, int64_t BH, int64_t qt, int64_t qh, int64_t qw, int64_t kt, int64_t kh,
                                 int64_t kw, const float* Rh, const float* Rw, const float* Rt, const int32_t* idx_h,
                                 const int32_t* idx_w, const int32_t* idx_t, float out_scale, void* relp, void* stream) {
  RelGeom g;
  if (!Q || !Rh || !Rw || !Rt || !idx_h || !idx_w || !idx_t || !relp || rel_geom(g, BH, qt, qh, qw, kt, kh, kw)) return PVRL_EINVAL;
  RelAxes ax = {};
  rel_axes(ax, g);
  ax.R[0] = Rh; ax.R[1] = Rw; ax.R[2] = Rt;
  ax.idx[0] = idx_h; ax.idx[1] = idx_w; ax.idx[2] = idx_t;
  hipLaunchKernelGGL(rel_fwd_kernel, dim3((unsigned)ax.first[3]), dim3(256), 0, (hipStream_t)stream, (const op_t*)Q, g, ax,
                     out_scale, (int)pvrl_mvit_rel_width(kt, kh, kw), (op_t*)relp);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int64_t pvrl_mvit_rel_bwd_workspace_bytes(int64_t BH, int64_t qt, int64_t qh, int64_t qw, int64_t kt, int64_t kh,
                                                     int64_t kw) {
  RelGeom g;
  if (rel_geom(g, BH, qt, qh, qw, kt, kh, kw)) return PVRL_EINVAL;
  RelAxes ax = {};
  rel_axes(ax, g);
  int64_t e = 0;
  for (int a = 0; a < 3; ++a) e += (int64_t)ax.chunks[a] * ax.qn[a] * ax.kn[a];
  return e * HD * (int64_t)sizeof(float);
}

extern "C" int pvrl_mvit_rel_bwd(const float* drel, const void* Q, void* dQ, int64_t BH, int64_t qt, int64_t qh,
                                 int64_t qw, int64_t kt, int64_t kh, int64_t kw, const float* Rh, const float* Rw,
                                 const float* Rt, const int32_t* idx_h, const int32_t* idx_w, const int32_t* idx_t,
                                 int64_t nrows_h, int64_t nrows_w, int64_t nrows_t, float* dRh, float* dRw, float* dRt,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  RelGeom g;
  if (!drel || !Q || !dQ || !Rh || !Rw || !Rt || !idx_h || !idx_w || !idx_t || !dRh || !dRw || !dRt || !workspace ||
      nrows_h <= 0 || nrows_w <= 0 || nrows_t <= 0 || rel_geom(g, BH, qt, qh, qw, kt, kh, kw))
    return PVRL_EINVAL;
  if (workspace_bytes < pvrl_mvit_rel_bwd_workspace_bytes(BH, qt, qh, qw, kt, kh, kw)) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // 12 channels per thread = 8 lanes per query (with per-thread d rel / index loads: 4 / 8 / 16 / 32 channels measured
  // 2.4 / 1.9 / 2.8 / 4.2 ms per MViTv2-S step)
  const long nthr = (long)BH * qt * qh * qw * (HD / 12);
  if (nthr >= (1L << 31) - (1L << 20)) return PVRL_EINVAL;          // 32-bit index arithmetic in rel_bwd_q_kernel
  static const int relq_lds = [] { const char* e = getenv("PVRL_RELQ_LDS"); return e ? (e[0] != '0') : 1; }();   // 0: A/B runs
  if (relq_lds && nrows_h + nrows_w + nrows_t <= RELQ_MAXROWS && qh * kh <= RELQ_MAXIDX && qw * kw <= RELQ_MAXIDX && qt * kt <= RELQ_MAXIDX &&
      ((uintptr_t)Rh % 16) == 0 && ((uintptr_t)Rw % 16) == 0 && ((uintptr_t)Rt % 16) == 0) {
    // one 16-wave workgroup per CU (104 KB of tables), walking the queries grid-stride
    const unsigned grid = (unsigned)std::min<long>((nthr + 1023) / 1024, 256);
    hipLaunchKernelGGL((rel_bwd_q_kernel<12, true>), dim3(grid), dim3(1024), 0, s, drel, g, Rh, Rw, Rt, idx_h, idx_w, idx_t,
                       (int)nrows_h, (int)nrows_w, (int)nrows_t, (op_t*)dQ);
  } else {
    hipLaunchKernelGGL((rel_bwd_q_kernel<12, false>), dim3(grid_for(nthr)), dim3(256), 0, s, drel, g, Rh, Rw, Rt, idx_h, idx_w, idx_t,
                       (int)nrows_h, (int)nrows_w, (int)nrows_t, (op_t*)dQ);
  }
  PVRL_LAUNCH_CHECK();
  RelAxes ax = {};
  rel_axes(ax, g);
  ax.idx[0] = idx_h; ax.idx[1] = idx_w; ax.idx[2] = idx_t;
  RelReduce rr = {};
  float* w = (float*)workspace;
  const int64_t nrows[3] = {nrows_h, nrows_w, nrows_t};
  float* dR[3] = {dRh, dRw, dRt};
  int first = 0;
  for (int a = 0; a < 3; ++a) {
    ax.part[a] = w;
    w += (long)ax.chunks[a] * ax.qn[a] * ax.kn[a] * HD;
    rr.part[a] = ax.part[a]; rr.idx[a] = ax.idx[a]; rr.dR[a] = dR[a];
    rr.npairs[a] = ax.qn[a] * ax.kn[a]; rr.chunks[a] = ax.chunks[a];
    rr.first[a] = first;
    first += (int)nrows[a];
  }
  rr.first[3] = first;
  hipLaunchKernelGGL(rel_bwd_table_kernel, dim3((unsigned)ax.first[3]), dim3(256), 0, s, drel, (const op_t*)Q, g, ax);
  PVRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(rel_table_reduce_kernel, dim3((unsigned)first), dim3(4 * HD), 0, s, rr);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
