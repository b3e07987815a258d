// Probe: what limits the AdamW pass (7 streams over 134.6 M fp32 parameters: p, g, m, v in; p, m, v out = 3.77 GB)?
// Variants of the launch geometry / memory instructions of csrc/optim.hip's adam_kernel, event-timed over 20 launches each.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/adam_rate.hip -o /tmp/adam_rate && /tmp/adam_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct C { float b1, b2, lr_wd, step, bc2s, eps, gs; };
__device__ __forceinline__ void one(float& p, float g, float& m, float& v, const C& c) {
  g *= c.gs;
  p *= c.lr_wd;
  m = c.b1 * m + (1.f - c.b1) * g;
  v = c.b2 * v + (1.f - c.b2) * g * g;
  p -= c.step * m / (sqrtf(v) / c.bc2s + c.eps);
}
template <int NT, int UNROLL, int NTEMP>
__global__ __launch_bounds__(NT) void adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, long n, C c) {
  const long stride = (long)gridDim.x * NT * 4;
  for (long i0 = ((long)blockIdx.x * NT + threadIdx.x) * 4; i0 < n; i0 += stride * UNROLL) {
    f32x4 pv[UNROLL], gv[UNROLL], mv[UNROLL], vv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long i = i0 + u * stride;
      if (i + 4 <= n) {
        if (NTEMP & 1) {
          pv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + i));
          gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
          mv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i));
          vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
        } else {
          pv[u] = *reinterpret_cast<f32x4*>(p + i); gv[u] = *reinterpret_cast<const f32x4*>(g + i);
          mv[u] = *reinterpret_cast<f32x4*>(m + i); vv[u] = *reinterpret_cast<f32x4*>(v + i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long i = i0 + u * stride;
      if (i + 4 <= n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float a = pv[u][e], b = mv[u][e], d = vv[u][e]; one(a, gv[u][e], b, d, c); pv[u][e] = a; mv[u][e] = b; vv[u][e] = d; }
        if (NTEMP & 2) {
          __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + i));
          __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + i));
          __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + i));
        } else {
          *reinterpret_cast<f32x4*>(p + i) = pv[u]; *reinterpret_cast<f32x4*>(m + i) = mv[u]; *reinterpret_cast<f32x4*>(v + i) = vv[u];
        }
      }
    }
  }
}
template <int NT, int UNROLL, int NTEMP>
void run(const char* name, long blocks, float* p, float* g, float* m, float* v, long n) {
  C c{0.9f, 0.999f, 0.9999f, 1e-4f, 0.5f, 1e-8f, 1.0f};
  if (blocks <= 0) blocks = (n / 4 + NT - 1) / NT / UNROLL;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((adam<NT, UNROLL, NTEMP>), dim3((unsigned)blocks), dim3(NT), 0, 0, p, g, m, v, n, c);
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a, 0);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((adam<NT, UNROLL, NTEMP>), dim3((unsigned)blocks), dim3(NT), 0, 0, p, g, m, v, n, c);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 4; sum += ms; if (ms < best) best = ms;
  }
  printf("%-44s blocks %7ld: mean %7.1f us  min %7.1f us  = %.2f TB/s of the 28 n bytes\n", name, blocks, sum / 5 * 1e3, best * 1e3,
         28.0 * n / (best * 1e-3) / 1e12);
}
int main() {
  const long n = 134617088;      // ~ the step's parameter count, a multiple of 4
  float *p, *g, *m, *v;
  for (int shift = 0; shift < 2; ++shift) {
    // shift 1: the four arrays start at different offsets modulo 64 KB / 1 MB (are equal-phase streams a channel problem?)
    const long pad = shift ? 1 : 0;
    float* base[4];
    for (int k = 0; k < 4; ++k) { hipMalloc(&base[k], (n + 4 * 1048576) * 4); hipMemset(base[k], 0, (n + 4 * 1048576) * 4); }
    p = base[0]; g = base[1] + pad * (16384 + 1024 * 33); m = base[2] + pad * (2 * 16384 + 1024 * 71); v = base[3] + pad * (3 * 16384 + 1024 * 113);
    printf("---- arrays %s\n", shift ? "at staggered offsets (68 / 135 / 203 KB ...)" : "as hipMalloc returns them");
    run<256, 1, 0>("256 thr, grid-stride (shipped)", 4096, p, g, m, v, n);
    run<256, 1, 0>("256 thr, grid-stride", 2048, p, g, m, v, n);
    run<256, 1, 0>("256 thr, grid-stride", 8192, p, g, m, v, n);
    run<256, 1, 0>("256 thr, grid-stride", 16384, p, g, m, v, n);
    run<256, 1, 0>("256 thr, one pass (no loop)", 0, p, g, m, v, n);
    run<256, 2, 0>("256 thr, 2 x float4 in flight", 4096, p, g, m, v, n);
    run<256, 2, 0>("256 thr, 2 x float4 in flight", 2048, p, g, m, v, n);
    run<256, 4, 0>("256 thr, 4 x float4 in flight", 2048, p, g, m, v, n);
    run<512, 1, 0>("512 thr, grid-stride", 2048, p, g, m, v, n);
    run<1024, 1, 0>("1024 thr, grid-stride", 1024, p, g, m, v, n);
    run<256, 1, 2>("256 thr, nontemporal stores", 4096, p, g, m, v, n);
    run<256, 1, 3>("256 thr, nontemporal loads + stores", 4096, p, g, m, v, n);
    run<256, 2, 3>("256 thr, 2 x float4, nontemporal both", 4096, p, g, m, v, n);
    run<256, 2, 3>("256 thr, 2 x float4, nontemporal both", 2048, p, g, m, v, n);
    for (int k = 0; k < 4; ++k) hipFree(base[k]);
  }
  return 0;
}
