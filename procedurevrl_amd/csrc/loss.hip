// Step-matching loss head of the pre-training loop, fp32 throughout (the logits are
// unit-norm dot products divided by tau = 0.02, so the reference's fp32 is kept here).
//   * L2 normalisation of embeddings                  lib/models/vit.py:300-303,331,339,431
//   * teacher = softmax(teacher logits); keep the entries equal to one of the top-5 values
//     (ties kept, duplicated top values counted as often as they occur); renormalise;
//     loss1 = KLDivLoss(batchmean)(log_softmax(pred), teacher)     tools/train_net.py:152-160
//   * loss2 = MSELoss(mean)(mse[0], mse[1]), gradient to BOTH operands  tools/train_net.py:161
// One workgroup per logit row (K = 9871 step candidates): the row is streamed from HBM twice
// (teacher, student), everything else (softmax, top-5 selection, KL, gradient) is fused.
#include "common.h"
#include "../../include/pvrl.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y,
                                                         long ldy, float* __restrict__ inv_norm, int D) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) { const float v = x[row * ldx + c]; s += v * v; }
  s = block_reduce(s, red, false);
  const float inv = 1.0f / sqrtf(s);
  if (threadIdx.x == 0 && inv_norm) inv_norm[row] = inv;
  for (int c = threadIdx.x; c < D; c += 256) y[row * ldy + c] = x[row * ldx + c] * inv;
}

// dx = inv * (dy - y * (y . dy))
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, long lddy,
                                                         const float* __restrict__ y, long ldy,
                                                         const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                         long ldx, int D) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) s += dy[row * lddy + c] * y[row * ldy + c];
  s = block_reduce(s, red, false);
  const float inv = inv_norm[row];
  for (int c = threadIdx.x; c < D; c += 256) dx[row * ldx + c] = inv * (dy[row * lddy + c] - y[row * ldy + c] * s);
}

constexpr int KL_TOPK_MAX = 8;
constexpr int KL_STAGE_MAX = 12288;     // logits per row kept in LDS (2 x 48 KB): K = 9871 step candidates fits

// STAGE: the row's teacher probabilities and student logits are computed / loaded ONCE and kept in LDS for the eight passes below (each
// thread only ever touches its own k = tid mod 256, so no barrier is needed between the passes); the unstaged form re-reads the rows
// through L2 and recomputes the exponential in every pass (113 us at K = 9871; same expressions, same order of every sum: identical bits)
template <bool STAGE>
__global__ __launch_bounds__(256) void kl_topk_kernel(const float* __restrict__ pred, long ldp,
                                                      const float* __restrict__ teacher, long ldt, int K, int topk,
                                                      float grad_scale, float* __restrict__ row_loss,
                                                      float* __restrict__ dpred, long ldd, float* __restrict__ target_out,
                                                      long ldto) {
  __shared__ float red[4];
  __shared__ float cand_v[256 * KL_TOPK_MAX];
  __shared__ int cand_i[256 * KL_TOPK_MAX];
  __shared__ float top_v[KL_TOPK_MAX];
  __shared__ float stage[STAGE ? 2 * KL_STAGE_MAX : 1];
  float* const tp = stage;
  float* const ps = stage + (STAGE ? KL_STAGE_MAX : 0);
  const long row = blockIdx.x;
  const float* t = teacher + row * ldt;
  const float* p = pred + row * ldp;
  const int tid = threadIdx.x;

  // ---- teacher softmax statistics ----
  float mx = -INFINITY;
  for (int k = tid; k < K; k += 256) {
    const float tv = t[k];
    if (STAGE) { tp[k] = tv; ps[k] = p[k]; }
    mx = fmaxf(mx, tv);
  }
  mx = block_reduce(mx, red, true);
  float se = 0.f;
  for (int k = tid; k < K; k += 256) se += expf((STAGE ? tp[k] : t[k]) - mx);
  se = block_reduce(se, red, false);
  const float tinv = 1.0f / se;
  if (STAGE)
    for (int k = tid; k < K; k += 256) tp[k] = expf(tp[k] - mx) * tinv;
  auto prob = [&](int k) -> float { return STAGE ? tp[k] : expf(t[k] - mx) * tinv; };
  auto logit = [&](int k) -> float { return STAGE ? ps[k] : p[k]; };

  float tsum = 1.0f;
  if (topk > 0) {
    // ---- per-thread top-k of the probabilities (value, index), then block merge ----
    float lv[KL_TOPK_MAX];
    int li[KL_TOPK_MAX];
#pragma unroll
    for (int j = 0; j < KL_TOPK_MAX; ++j) { lv[j] = -1.f; li[j] = -1; }
    for (int k = tid; k < K; k += 256) {
      float v = prob(k);
      int vi = k;
#pragma unroll
      for (int j = 0; j < KL_TOPK_MAX; ++j) {
        if (j < topk && v > lv[j]) {
          const float tv = lv[j]; const int ti = li[j];
          lv[j] = v; li[j] = vi; v = tv; vi = ti;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KL_TOPK_MAX; ++j) { cand_v[tid * KL_TOPK_MAX + j] = lv[j]; cand_i[tid * KL_TOPK_MAX + j] = li[j]; }
    __syncthreads();
    if (tid < 64) {
      const int ncand = 256 * KL_TOPK_MAX;
      for (int r = 0; r < topk; ++r) {
        float bv = -2.f; int bpos = -1;
        for (int c = tid; c < ncand; c += 64) {
          const float v = cand_v[c];
          if (v > bv) { bv = v; bpos = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int op = __shfl_xor(bpos, o, 64);
          if (ov > bv || (ov == bv && op >= 0 && (bpos < 0 || op < bpos))) { bv = ov; bpos = op; }
        }
        if (tid == 0) { top_v[r] = bv; if (bpos >= 0) cand_v[bpos] = -3.f; }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      }
    }
    __syncthreads();
    // ---- renormaliser over kept entries (multiplicity = number of top slots with that value) ----
    float ks = 0.f;
    for (int k = tid; k < K; k += 256) {
      const float v = prob(k);
      int mult = 0;
      for (int j = 0; j < topk; ++j) mult += (top_v[j] == v) ? 1 : 0;
      ks += v * (float)mult;
    }
    tsum = block_reduce(ks, red, false);
  }

  // ---- student log-softmax ----
  float pm = -INFINITY;
  for (int k = tid; k < K; k += 256) pm = fmaxf(pm, logit(k));
  pm = block_reduce(pm, red, true);
  float pse = 0.f;
  for (int k = tid; k < K; k += 256) pse += expf(logit(k) - pm);
  pse = block_reduce(pse, red, false);
  const float plog = logf(pse);
  const float pinv = 1.0f / pse;

  // the renormalised target of entry k (the teacher probability where it equals one of the top values, as often as it occurs there)
  auto target = [&](int k) -> float {
    float tg = prob(k);
    if (topk > 0) {
      int mult = 0;
      for (int j = 0; j < topk; ++j) mult += (top_v[j] == tg) ? 1 : 0;
      tg = tg * (float)mult / tsum;
    }
    return tg;
  };
  float loss = 0.f, tot = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float tg = target(k);
    if (STAGE) tp[k] = tg;
    tot += tg;
    if (target_out) target_out[row * ldto + k] = tg;
    const float logp = logit(k) - pm - plog;
    if (tg > 0.f) loss += tg * (logf(tg) - logp);
  }
  loss = block_reduce(loss, red, false);
  tot = block_reduce(tot, red, false);
  if (tid == 0 && row_loss) row_loss[row] = loss;
  if (dpred) {
    for (int k = tid; k < K; k += 256) {
      const float tg = STAGE ? tp[k] : target(k);
      const float sp = expf(logit(k) - pm) * pinv;
      dpred[row * ldd + k] = grad_scale * (sp * tot - tg);
    }
  }
}

// loss[0] = mean((a-b)^2);  da = gscale * 2 (a-b) / n ; db = -da      (single workgroup; n is small)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                  float gscale, float* __restrict__ loss, float* __restrict__ da,
                                                  float* __restrict__ db) {
  __shared__ float red[4];
  float s = 0.f;
  const float gs = gscale * 2.0f / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float d = a[i] - b[i];
    s += d * d;
    if (da) da[i] = gs * d;
    if (db) db[i] = -gs * d;
  }
  s = block_reduce(s, red, false);
  if (threadIdx.x == 0 && loss) loss[0] = s / (float)n;
}

// MIL-NCE (lib/models/losses.py:15-23) on x[n][n][C] = video . text^T:
//   nom_i = logsumexp_c x[i][i][c];   den_i = logsumexp over row i and column i (all j, c);   loss = mean(den - nom)
// one workgroup per i computes (nom_i, den_i); the gradient kernel is elementwise over x.
__global__ __launch_bounds__(256) void milnce_stats_kernel(const float* __restrict__ x, int n, int C,
                                                           float* __restrict__ nom, float* __restrict__ den) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const long row = (long)n * C;
  float mx = -INFINITY;
  for (int e = tid; e < n * C; e += 256) {
    const int j = e / C, c = e - j * C;
    mx = fmaxf(mx, fmaxf(x[i * row + e], x[j * row + (long)i * C + c]));
  }
  mx = block_reduce(mx, red, true);
  float se = 0.f;
  for (int e = tid; e < n * C; e += 256) {
    const int j = e / C, c = e - j * C;
    se += expf(x[i * row + e] - mx) + expf(x[j * row + (long)i * C + c] - mx);
  }
  se = block_reduce(se, red, false);
  float nm = -INFINITY;
  for (int c = tid; c < C; c += 256) nm = fmaxf(nm, x[i * row + (long)i * C + c]);
  nm = block_reduce(nm, red, true);
  float ns = 0.f;
  for (int c = tid; c < C; c += 256) ns += expf(x[i * row + (long)i * C + c] - nm);
  ns = block_reduce(ns, red, false);
  if (tid == 0) {
    den[i] = mx + logf(se);
    nom[i] = nm + logf(ns);
  }
}

__global__ __launch_bounds__(256) void milnce_grad_kernel(const float* __restrict__ x, const float* __restrict__ nom,
                                                          const float* __restrict__ den, int n, int C, float gscale,
                                                          float* __restrict__ dx) {
  const long total = (long)n * n * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int a = (int)(idx / ((long)n * C));
    const int b = (int)((idx / C) % n);
    const float v = x[idx];
    float g = expf(v - den[a]) + expf(v - den[b]);
    if (a == b) g -= expf(v - nom[a]);
    dx[idx] = gscale * g;
  }
}


// row softmax over fp32 logits (eval-mode output of the wrappers, lib/models/vit.py:355-356): one workgroup per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y,
                                                           long ldy, int N) {
  __shared__ float red[8];
  const float* xr = x + (long)blockIdx.x * ldx;
  float* yr = y + (long)blockIdx.x * ldy;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < N; c += 256) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.f;
  for (int c = threadIdx.x; c < N; c += 256) sm += __expf(xr[c] - mx);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sm;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = threadIdx.x; c < N; c += 256) yr[c] = __expf(xr[c] - mx) * inv;
}

// exact-erf GELU on a small fp32 tensor (time_mlp of the order transformer, lib/models/tfm_model.py:89-94) and its derivative
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    out[i] = dy ? dy[i] * (0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v))
                : 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  }
}

}  // namespace

extern "C" int pvrl_milnce(const float* x, int64_t n, int64_t C, float grad_scale, float* nom, float* den, float* dx,
                           void* stream) {
  if (n <= 0 || C <= 0 || !x || !nom || !den) return PVRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(milnce_stats_kernel, dim3((unsigned)n), dim3(256), 0, s, x, (int)n, (int)C, nom, den);
  PVRL_LAUNCH_CHECK();
  if (dx) {
    long blocks = (n * n * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(milnce_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, nom, den, (int)n, (int)C,
                       grad_scale, dx);
    PVRL_LAUNCH_CHECK();
  }
  return PVRL_OK;
}

extern "C" int pvrl_l2norm_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, float* inv_norm, int64_t M, int64_t D,
                               void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!x || !y) return PVRL_EINVAL;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y, (long)ldy,
                     inv_norm, (int)D);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm,
                               float* dx, int64_t ldx, int64_t M, int64_t D, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!dy || !y || !inv_norm || !dx) return PVRL_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, dy, (long)lddy, y,
                     (long)ldy, inv_norm, dx, (long)ldx, (int)D);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_kl_topk(const float* pred, int64_t ldp, const float* teacher, int64_t ldt, int64_t rows, int64_t K,
                            int64_t topk, float grad_scale, float* row_loss, float* dpred, int64_t ldd,
                            float* target_out, int64_t ldto, void* stream) {
  if (rows <= 0) return PVRL_OK;
  if (!pred || !teacher || K <= 0 || topk < 0 || topk > KL_TOPK_MAX) return PVRL_EINVAL;
  if (K <= KL_STAGE_MAX)
    hipLaunchKernelGGL(kl_topk_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, pred, (long)ldp, teacher,
                       (long)ldt, (int)K, (int)topk, grad_scale, row_loss, dpred, (long)ldd, target_out, (long)ldto);
  else
    hipLaunchKernelGGL(kl_topk_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, pred, (long)ldp, teacher,
                       (long)ldt, (int)K, (int)topk, grad_scale, row_loss, dpred, (long)ldd, target_out, (long)ldto);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_mse(const float* a, const float* b, int64_t n, float grad_scale, float* loss, float* da, float* db,
                        void* stream) {
  if (n <= 0 || !a || !b) return PVRL_EINVAL;
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, grad_scale, loss, da, db);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, void* stream) {
  if (M <= 0) return PVRL_OK;
  if (!x || !y || N <= 0) return PVRL_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y, (long)ldy,
                     (int)N);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_gelu_f32(const float* x, const float* dy, float* out, int64_t n, void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!x || !out) return PVRL_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, out, (long)n);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
