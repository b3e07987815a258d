// Fused optimiser steps over the flat fp32 parameter / gradient buffers.
// Reference: torch.optim.{SGD(nesterov), Adam, AdamW} as constructed by
// lib/models/optimizer.py:93-118 and stepped in tools/train_net.py:176-192 (with the
// gradient-accumulation division `p.grad /= num_iters` folded in as `gscale`).
// One HBM pass: read p, g, state; write p, state (the foreach/unfused path of the reference
// launches ~10 elementwise kernels per parameter tensor, ~300 tensors).
#include "common.h"
#include "../../include/pvrl.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float gscale, int decoupled) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gg = gv[e] * gscale;
        if (decoupled) pv[e] *= (1.0f - lr * wd); else gg += wd * pv[e];
        mv[e] = b1 * mv[e] + (1.0f - b1) * gg;
        vv[e] = b2 * vv[e] + (1.0f - b2) * gg * gg;
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pv[e] -= (lr / bc1) * (mv[e] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long j = i; j < n; ++j) {
        float gg = g[j] * gscale, pj = p[j];
        if (decoupled) pj *= (1.0f - lr * wd); else gg += wd * pj;
        const float mj = b1 * m[j] + (1.0f - b1) * gg;
        const float vj = b2 * v[j] + (1.0f - b2) * gg * gg;
        m[j] = mj; v[j] = vj;
        p[j] = pj - (lr / bc1) * (mj / (sqrtf(vj) / bc2_sqrt + eps));
      }
    }
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long n, float lr, float momentum,
                                                  float dampening, float wd, int nesterov, int first, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gg = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? gg : momentum * buf[i] + (1.0f - dampening) * gg;
      buf[i] = b;
      gg = nesterov ? gg + momentum * b : b;
    }
    p[i] -= lr * gg;
  }
}

}  // namespace

extern "C" int pvrl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, float gscale, int decoupled,
                              void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!p || !g || !m || !v || step < 1) return PVRL_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), gscale, decoupled);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                             float dampening, float weight_decay, int nesterov, int first_step, float gscale,
                             void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!p || !g || (momentum != 0.f && !buf)) return PVRL_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long)n, lr,
                     momentum, dampening, weight_decay, nesterov, first_step, gscale);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
