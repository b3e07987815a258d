// Fused optimiser steps over the flat fp32 parameter / gradient buffers.
// Reference: torch.optim.{SGD(nesterov), Adam, AdamW} as constructed by
// lib/models/optimizer.py:93-118 and stepped in tools/train_net.py:176-192 (with the
// gradient-accumulation division `p.grad /= num_iters` folded in as `gscale`).
// One HBM pass: read p, g, state; write p, state (the foreach/unfused path of the reference
// launches ~10 elementwise kernels per parameter tensor, ~300 tensors).
#include "common.h"
#include "../../include/pvrl.h"

namespace {

// The scalar constants are formed on the host in double and rounded to fp32 once, which is what torch.optim does (Python
// floats 1 - lr*wd, 1 - beta1, 1 - beta2, lr / bias_correction1, sqrt(bias_correction2) handed to fp32 tensor ops):
//   p *= decay (AdamW) | g += wd * p (Adam);  m = m + w1 * (g - m)  [Tensor.lerp_];  v = b2 * v + w2 * g * g;
//   p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)
// Two places where torch itself is not one arithmetic: `sqrt(v) / bc2_sqrt` is a true division in torch's CPU kernels (what
// tests/test_optimizer_gpu.py pins against, <= 1e-6) and a multiply by the fp32 reciprocal in its CUDA/HIP kernels -- <= 1 ulp
// apart, the division is kept; Tensor.lerp_ switches formula at weight 0.5, mirrored below (beta1 < 0.5 takes the other one).
struct AdamConst { float decay, wd, w1, b2, w2, step_size, bc2_sqrt, eps, gscale; int decoupled; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamConst& c) {
  float gg = g * c.gscale;
  if (c.decoupled) p *= c.decay; else gg += c.wd * p;
  m = c.w1 < 0.5f ? m + c.w1 * (gg - m) : gg - (gg - m) * (1.f - c.w1);
  v = c.b2 * v + c.w2 * gg * gg;
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p -= c.step_size * (m / denom);
}

// `skip` (device flag, may be null): non-zero = this step must not be applied (a non-finite loss or gradient was seen on the
// device -- tools/train_net.py:174 `misc.check_nan_losses` raises BEFORE optimizer.step(); the host here reads its metrics only every
// LOG_PERIOD iterations, so the bad step is dropped on the device and reported at the next log point).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, AdamConst c,
                                                   const float* __restrict__ skip) {
  if (skip && *skip != 0.f) return;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        adam_one(pe, gv[e], me, ve, c);
        pv[e] = pe; mv[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long j = i; j < n; ++j) adam_one(p[j], g[j], m[j], v[j], c);
    }
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long n, float lr, float momentum,
                                                  float one_minus_damp, float wd, int nesterov, int first, float gscale,
                                                  const float* __restrict__ skip) {
  if (skip && *skip != 0.f) return;
  // torch.optim.SGD: g += wd * p;  buf = g (first update) | momentum * buf + (1 - dampening) * g;
  //                  g = g + momentum * buf (nesterov) | buf;  p -= lr * g
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gg = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? gg : momentum * buf[i] + one_minus_damp * gg;
      buf[i] = b;
      gg = nesterov ? gg + momentum * b : b;
    }
    p[i] -= lr * gg;
  }
}

// flag = 1 when any element of x is inf / nan (exponent all ones); the flag is only ever raised, never cleared
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ x, long n, float* __restrict__ flag) {
  bool bad = false;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(x + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= (w[e] & 0x7f800000u) == 0x7f800000u;
    } else {
      for (long j = i; j < n; ++j) bad |= (__float_as_uint(x[j]) & 0x7f800000u) == 0x7f800000u;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;
}

// The fp16 flavour's gradient scale of one engine backward (engine.GradStore.begin_scaled) in ONE launch: S = 2^floor(log2(target / max|g|))
// chosen on the device, out = g * S, scale[0] = S, inv[0 .. ninv) = 1 / S (the `gscale` scalar of the gradient-writing kernels, repeated as
// the per-row scale of a GEMM epilogue).  A non-finite max (inf / nan in g) must not turn S into 0 or 1 / S into inf: S stays a finite
// power of two and the non-finite values flow on to the loss / gradient checks.  One workgroup: g is an engine's incoming gradient
// (a few 10^4 elements); torch's abs / max / nan_to_num / clamp / log2 / floor / exp2 / reciprocal / mul chain was 14 launches.
__global__ __launch_bounds__(1024) void grad_scale_begin_kernel(const float* __restrict__ g, long n, float target, float* __restrict__ out,
                                                                float* __restrict__ scale, float* __restrict__ inv, int ninv) {
  __shared__ float red[16];
  __shared__ float s_scale;
  float m = 0.f;
  bool bad = false;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float v = g[i];
    bad |= (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
    m = fmaxf(m, fabsf(v));                         // (fmaxf drops a NaN: tracked separately)
  }
  m = wave_max(m);
  const unsigned long long anybad = __ballot(bad);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = anybad ? __uint_as_float(0x7f800000u) : m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    bool inf = false;
    for (int w = 0; w < 16; ++w) { inf |= (__float_as_uint(red[w]) & 0x7f800000u) == 0x7f800000u; a = fmaxf(a, red[w]); }
    if (inf) a = 3e38f;                             // torch path: nan_to_num(nan = 1, posinf = 3e38); either way S stays finite
    a = fminf(fmaxf(a, 1e-30f), 3e38f);
    float e = floorf(log2f(target / a));
    e = fminf(fmaxf(e, -100.f), 100.f);
    s_scale = exp2f(e);
  }
  __syncthreads();
  const float S = s_scale, iS = 1.0f / S;
  for (long i = threadIdx.x; i < n; i += 1024) out[i] = g[i] * S;
  if (threadIdx.x == 0) scale[0] = S;
  for (int i = threadIdx.x; i < ninv; i += 1024) inv[i] = iS;
}

// end of an optimiser step: count it when it was dropped and re-arm the flag for the next one
__global__ void flag_roll_kernel(float* __restrict__ flag, float* __restrict__ total) {
  if (*flag != 0.f && total) *total += 1.f;
  *flag = 0.f;
}

}  // namespace

extern "C" int pvrl_grad_scale_begin(const float* g, int64_t n, float target, float* out, float* scale, float* inv, int64_t ninv,
                                     void* stream) {
  if (!g || !out || !scale || !inv || n <= 0 || ninv < 1 || !(target > 0.f)) return PVRL_EINVAL;
  hipLaunchKernelGGL(grad_scale_begin_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, (long)n, target, out, scale, inv, (int)ninv);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_flag_roll(float* flag, float* total, void* stream) {
  if (!flag) return PVRL_EINVAL;
  hipLaunchKernelGGL(flag_roll_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, total);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_nonfinite_flag_f32(const float* x, int64_t n, float* flag, void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!x || !flag || ((uintptr_t)x & 15)) return PVRL_EINVAL;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, flag);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int64_t step, double gscale, int decoupled,
                              const float* skip, void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!p || !g || !m || !v || step < 1) return PVRL_EINVAL;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  AdamConst c;
  c.decay = (float)(1.0 - lr * weight_decay); c.wd = (float)weight_decay;
  c.w1 = (float)(1.0 - beta1); c.b2 = (float)beta2; c.w2 = (float)(1.0 - beta2);
  c.step_size = (float)(lr / bc1); c.bc2_sqrt = (float)sqrt(bc2); c.eps = (float)eps; c.gscale = (float)gscale;
  c.decoupled = decoupled;
  // one float4 of each stream per thread, no grid-stride loop: measured on 134.6 M parameters (tools/probe/adam_rate.hip) 695 us against
  // 850-930 us for 2,048-16,384 looping workgroups (5.4 vs 4.1-4.5 TB/s of the 28 bytes per parameter)
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 0x7fffffffL) blocks = 0x7fffffffL;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, c, skip);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}

extern "C" int pvrl_sgd_step(float* p, const float* g, float* buf, int64_t n, double lr, double momentum,
                             double dampening, double weight_decay, int nesterov, int first_step, double gscale,
                             const float* skip, void* stream) {
  if (n <= 0) return PVRL_OK;
  if (!p || !g || (momentum != 0.0 && !buf)) return PVRL_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long)n, (float)lr,
                     (float)momentum, (float)(1.0 - dampening), (float)weight_decay, nesterov, first_step, (float)gscale, skip);
  PVRL_LAUNCH_CHECK();
  return PVRL_OK;
}
