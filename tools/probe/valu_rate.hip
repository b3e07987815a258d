// VALU issue cost on gfx950, cycles per wave64 instruction with 1 / 2 waves per SIMD resident: v_fma_f32, v_pk_fma_f32, v_exp_f32,
// v_rcp_f32, v_cvt_pk_bf16_f32 -- the numbers behind the GELU epilogue's and the attention forward's VALU budgets.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probe/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters, float seed) {
  float a[8];
  f32x2 b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = (f32x2){a[i], a[i] + 0.5f}; }
  const f32x2 c2 = {seed * 0.999f, seed * 0.999f};
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(c2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (KIND == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
      if (KIND == 7) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[i]) : "v"(seed));
    }
  }
  long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + b[i][0] + b[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {       // 1 or 2 waves per SIMD on the CU
    probe<KIND><<<1, threads>>>(out, cyc, iters, 1.0001f);
    hipDeviceSynchronize();
    probe<KIND><<<1, threads>>>(out, cyc, iters, 1.0001f);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %d waves/SIMD: %6.2f cycles per instruction and wave (%6.2f per SIMD issue slot)\n", name, threads / 256,
           (double)h / (iters * 8.0), (double)h / (iters * 8.0) / (threads / 256));
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", out, cyc);
  run<7>("v_fmaak_f32 (literal)", out, cyc);
  run<1>("v_pk_fma_f32", out, cyc);
  run<6>("v_pk_mul_f32", out, cyc);
  run<2>("v_exp_f32", out, cyc);
  run<3>("v_rcp_f32", out, cyc);
  run<4>("v_cvt_pk_bf16_f32", out, cyc);
  run<5>("v_max_f32", out, cyc);
  return 0;
}
