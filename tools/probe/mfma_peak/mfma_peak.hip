// Probe: what the matrix pipe sustains on this MI355X with NO memory traffic -- v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 back to back,
// NACC independent accumulators per wave, 1 / 2 / 4 waves per SIMD, every CU busy.  Prints TFLOP/s and the implied clock.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_peak/mfma_peak.hip -o tools/probe/mfma_peak/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters) {
  bx8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters) {
  bx8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f16v acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double run(F launch, double flops_per_launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flops_per_launch * reps / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 4);
  const int cus = 256, iters = 4000;
  for (int wps = 1; wps <= 4; wps *= 2) {            // waves per SIMD
    const int threads = 256, blocks = cus * wps;     // one 4-wave workgroup per CU per wave-per-SIMD
    {
      const double fl = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
      double t = run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(threads), 0, 0, out, iters); }, fl, 5);
      printf("16x16x32  8 acc  %d wave(s)/SIMD: %7.0f TFLOP/s (implied clock at 1024 flop/clk/SIMD... %4.2f GHz)\n", wps, t, t * 1e12 / (1024.0 * 1024) / 1e9);
    }
    {
      const double fl = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
      double t = run([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(threads), 0, 0, out, iters); }, fl, 5);
      printf("16x16x32 16 acc  %d wave(s)/SIMD: %7.0f TFLOP/s (%4.2f GHz)\n", wps, t, t * 1e12 / (1024.0 * 1024) / 1e9);
    }
    {
      const double fl = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 16);
      double t = run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); }, fl, 5);
      printf("32x32x16  4 acc  %d wave(s)/SIMD: %7.0f TFLOP/s (%4.2f GHz)\n", wps, t, t * 1e12 / (1024.0 * 1024) / 1e9);
    }
  }
  // long run: does the clock sag under sustained load?
  {
    const int blocks = cus * 4;
    const double fl = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
    for (int r = 0; r < 4; ++r) {
      double t = run([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters); }, fl, 200);
      printf("sustained (200 launches): %7.0f TFLOP/s\n", t);
    }
  }
  return 0;
}
