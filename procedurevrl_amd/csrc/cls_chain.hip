// The cls rows' own chain in fp32.
//
// One cls token per clip rides through every block of the divided space-time encoder (lib/models/vit.py:139-157): it is the
// query of the spatial attention whose output, averaged over the T frames, is projected and added to it, then it passes
// the MLP.  Only these B rows reach the head (vit.py:418-421, 299-307), and their rounding errors do NOT average out the way
// the 1,568 patch tokens' errors do inside the attention: measured with the oracle that carries the datapath's rounding
// points (tests/probe_rounding_budget.py, fp16 operands, 12 blocks): logits error 7.0e-4 with every row on the 16-bit path,
// 2.9e-4 when the cls rows' projection and MLP run in fp32 on the fp32 master weights, 2.7e-4 with everything else of their
// chain in fp32 as well.  So the B cls rows get their own small fp32 kernels; the 50k patch rows stay on the MFMA path.
//
//   Y[M, N] = epilogue(X[M, K] . W[N, K]^T)     X, W, Y fp32;  M = clips (<= a few dozen),  N, K multiples of 16 * waves
//
// HBM-bound on W (2.4 - 9.4 MB of fp32 weights per call): one workgroup per 16 output columns streams its 16 rows of W
// once, 16 bytes per lane, with the reduction index split over the waves of the workgroup (each wave: K / waves), against
// the X rows read through L2; v_mfma_f32_16x16x4_f32 (true fp32 products) does the arithmetic so that neither VALU issue nor
// LDS sits between the loads and the accumulators; the waves' partial tiles are summed through LDS in wave order
// (deterministic).
#include "common.h"
#include "../../include/pvrl.h"

namespace {

struct ClsLin {
  const float* X; long ldx;
  const float* W; long ldw;
  const float* bias;
  const float* rowscale;     // [M] or null: multiplies the product
  const float* biasscale;    // [M] or null: multiplies the bias
  const float* aux; long ld_aux;      // fp32 [M, N] or null: added last (residual)
  float* out; long ldo;
  op_t* o16a; op_t* o16b; long ld16;  // GELU epilogue: 16-bit copies of the pre-activation / the activation (or null)
  float* part;               // ksplit > 1: [ksplit][M][N] partial products of the K slices (cls_epilogue_kernel sums them in slice order)
  float alpha;               // multiplies the product (1 for the cls chain; 1 / temperature for the step logits)
  int M, N, K, ksplit;
};

constexpr int CL_MT = 3;      // at most this many 16-row tiles of X per pass: 48 rows (more rows: grid.y passes, W is streamed again)

// EPI 0: aux + rowscale * acc + biasscale * bias;  1: exact-erf GELU(acc + bias).  KU: 16-wide k steps per unrolled chunk (all of a
// chunk's loads are in flight before its first MFMA)
__device__ __forceinline__ void cls_store(const ClsLin& p, int epi, int m, int n, float s) {
  const float b = p.bias ? p.bias[n] : 0.f;
  s *= p.alpha;
  if (epi == 0) {
    float y = (p.rowscale ? p.rowscale[m] * s : s) + (p.biasscale ? p.biasscale[m] * b : b);
    if (p.aux) y += p.aux[(long)m * p.ld_aux + n];
    p.out[(long)m * p.ldo + n] = y;
  } else {
    const float u = s + b;
    const float g = 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
    p.out[(long)m * p.ldo + n] = g;
    if (p.o16a) p.o16a[(long)m * p.ld16 + n] = (op_t)u;
    if (p.o16b) p.o16b[(long)m * p.ld16 + n] = (op_t)g;
  }
}

// CL_MT template parameter MT: 16-row tiles actually carried (M <= 16 -> 1, <= 32 -> 2: a third tile of clamped rows is a third more
// activation loads).  gridDim.z = slices of K (ksplit): with one workgroup per 16 columns the long reduction of fc2 (N = 768, K = 3072)
// ran on 48 CUs and took 25 us for 9.4 MB -- what ONE CU's vector-memory path moves, not HBM; cut into four slices it runs on 192, the
// slices' products go to a workspace and cls_epilogue_kernel sums them in slice order (no atomics, no device-scope fences: on this
// chip an agent-scope release writes an XCD's L2 back -- a last-arriver reduce inside the kernel was measured 3-6x SLOWER than this).
template <int NW, int EPI, int KU, int MT>
__global__ __launch_bounds__(NW * 64) void cls_linear_kernel(ClsLin p) {
  constexpr int CL_MT = MT;
  __shared__ float part[NW][CL_MT][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * CL_MT);
  const int mt_n = min(CL_MT, (p.M - m0 + 15) >> 4);
  const int kslice = p.K / p.ksplit;
  const int kw = kslice / NW, kbeg = blockIdx.z * kslice + wave * kw;
  const float* wrow = p.W + (long)min(n0 + r, p.N - 1) * p.ldw + kbeg + q * 4;      // (N need not be a multiple of 16: columns >= N are never stored)
  const float* xrow[CL_MT];
#pragma unroll
  for (int t = 0; t < CL_MT; ++t) xrow[t] = p.X + (long)min(m0 + t * 16 + r, p.M - 1) * p.ldx + kbeg + q * 4;
  f32x4 acc[CL_MT];
#pragma unroll
  for (int t = 0; t < CL_MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // lane (r, q) holds k = kk + 4 q + j of row r for j = 0..3; MFMA j consumes element j of both operands, so operand A's and B's k
  // indices agree lane by lane (any bijection of k onto (MFMA, quarter) is a valid order of the sum)
  for (int kc = 0; kc < kw; kc += 16 * KU)
#pragma unroll
  for (int kk = kc; kk < kc + 16 * KU; kk += 16) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(wrow + kk);
    f32x4 a4[CL_MT];
    // (every tile unconditionally, rows clamped: a guarded load is a serial load, and rows >= M are never stored)
#pragma unroll
    for (int t = 0; t < CL_MT; ++t) a4[t] = *reinterpret_cast<const f32x4*>(xrow[t] + kk);
#pragma unroll
    for (int t = 0; t < CL_MT; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t][j], b4[j], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < CL_MT; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) part[wave][t][i][lane] = acc[t][i];
  }
  __syncthreads();
  // element e = (tile t, i, lane): row m0 + 16 t + 4 (lane / 16) + i, column n0 + lane % 16
  for (int e = tid; e < mt_n * 256; e += NW * 64) {
    const int t = e >> 8, i = (e >> 6) & 3, l = e & 63;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += part[w][t][i][l];
    const int m = m0 + t * 16 + 4 * (l >> 4) + i, n = n0 + (l & 15);
    if (m >= p.M || n >= p.N) continue;
    if (p.ksplit > 1) p.part[((long)blockIdx.z * p.M + m) * p.N + n] = s;
    else cls_store(p, EPI, m, n, s);
  }
}

__global__ __launch_bounds__(256) void cls_epilogue_kernel(ClsLin p, int epi) {
  const long MN = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < p.ksplit; ++z) s += p.part[(long)z * MN + i];
    cls_store(p, epi, (int)(i / p.N), (int)(i % p.N), s);
  }
}

template <int NW, int KU, int MT>
int launch_cls_mt(const ClsLin& p, int epi, hipStream_t s) {
  dim3 grid((unsigned)cdiv(p.N, 16), (unsigned)cdiv(p.M, 16 * MT), (unsigned)p.ksplit);
  if (epi == 0) hipLaunchKernelGGL((cls_linear_kernel<NW, 0, KU, MT>), grid, dim3(NW * 64), 0, s, p);
  else hipLaunchKernelGGL((cls_linear_kernel<NW, 1, KU, MT>), grid, dim3(NW * 64), 0, s, p);
  PVRL_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    hipLaunchKernelGGL(cls_epilogue_kernel, dim3((unsigned)cdiv((long)p.M * p.N, 256)), dim3(256), 0, s, p, epi);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}
template <int NW, int KU>
int launch_cls_ku(const ClsLin& p, int epi, hipStream_t s) {
  if (p.M <= 16) return launch_cls_mt<NW, KU, 1>(p, epi, s);
  if (p.M <= 32) return launch_cls_mt<NW, KU, 2>(p, epi, s);
  return launch_cls_mt<NW, KU, 3>(p, epi, s);
}
template <int NW>
int launch_cls(const ClsLin& p, int epi, hipStream_t s) {
  const int kw = p.K / p.ksplit / NW;
  if (kw % 96 == 0) return launch_cls_ku<NW, 6>(p, epi, s);      // ViT-B: 768 / 8, and 3072 in four slices
  if (kw % 32 == 0) return launch_cls_ku<NW, 2>(p, epi, s);
  return launch_cls_ku<NW, 1>(p, epi, s);
}

}  // namespace

// slices of K: the long reductions with few column tiles (fc2: N = 768, K = 3072 -> 4 x 48 workgroups)
static int cls_ksplit(int64_t N, int64_t K) { return (K >= 2048 && K % 1024 == 0 && N / 16 < 128) ? 4 : 1; }

extern "C" int64_t pvrl_cls_linear_f32_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t ks = cls_ksplit(N, K);
  return ks > 1 ? ks * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int pvrl_cls_linear_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t M,
                                   int64_t N, int64_t K, int epilogue, const float* rowscale, const float* biasscale,
                                   const float* aux, int64_t ld_aux, float* out, int64_t ldo, void* out16_pre,
                                   void* out16_act, int64_t ld16, void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!X || !W || !out || N <= 0 || K <= 0 || (N % 16) || (K % 128) || (epilogue != 0 && epilogue != 1)) return PVRL_EINVAL;
  if ((ldx % 4) || (ldw % 4) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15)) return PVRL_EINVAL;
  if (epilogue == 1 && (rowscale || biasscale || aux)) return PVRL_EINVAL;
  ClsLin p;
  p.X = X; p.ldx = ldx; p.W = W; p.ldw = ldw; p.bias = bias; p.rowscale = rowscale; p.biasscale = biasscale;
  p.aux = aux; p.ld_aux = ld_aux; p.out = out; p.ldo = ldo;
  p.o16a = (op_t*)out16_pre; p.o16b = (op_t*)out16_act; p.ld16 = ld16;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ksplit = cls_ksplit(N, K); p.alpha = 1.f;
  p.part = (float*)workspace;
  if (p.ksplit > 1 && (!workspace || workspace_bytes < pvrl_cls_linear_f32_workspace_bytes(M, N, K))) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // eight waves per workgroup = eight slices of the workgroup's K range; K / ksplit / 8 stays a multiple of 16
  return launch_cls<8>(p, epilogue, s);
}

// The same kernel behind pvrl_gemm_nt_f32_small (gemm_nt.hip) for its K % 128 == 0 shapes -- C = alpha * (A B^T) + bias with a few dozen
// rows: projection head 768 -> 512 and the step logits against the 9871 x 512 fp32 label table (lib/models/vit.py:299-307), where the
// 64 x 64-tile FMA kernel took 99 us for the 20 MB it streams.  false: not a shape for this kernel (the caller keeps its own).
bool pvrl_cls_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float alpha, float* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, hipStream_t s, int* status) {
  if ((K % 128) || (lda % 4) || (ldb % 4) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || cls_ksplit(N, K) != 1) return false;
  ClsLin p = {};
  p.X = A; p.ldx = lda; p.W = B; p.ldw = ldb; p.bias = bias; p.out = C; p.ldo = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ksplit = 1; p.alpha = alpha;
  *status = launch_cls<8>(p, 0, s);
  return true;
}
