// Probe: does the ACCESS PATTERN of a GEMM epilogue cost HBM throughput?  The product NT kernel's epilogue touches, per
// wave instruction, 16 rows x 64 B (bf16) or 16 rows x (4 x 16 B of a 128-B line) (fp32): partial cache lines, many rows.
// An LDS-staged epilogue would touch 1 KiB contiguous per instruction.  This probe moves the same bytes both ways over a
// [M x N] matrix in 256x256 tiles, one 1024-thread workgroup per tile (the GEMM's geometry), no MFMA work:
//   mode 0: bf16 store, epilogue pattern      mode 1: bf16 store, row-contiguous pattern
//   mode 2: fp32 load + store, epilogue pattern   mode 3: fp32 load + store, row-contiguous pattern
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern tools/probe/store_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void pat(void* out, const void* in, int M, int N, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int wm = wave >> 2, wn = wave & 3, q = lane >> 4, i = lane & 15;
  if (MODE == 0) {            // 8 instructions per wave: rows mt*16+i, 16 B at column nw0 + 32c + 8q
    unsigned short* o = (unsigned short*)out;
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + i;
      if (m >= M) continue;
      for (int c = 0; c < 2; ++c) *reinterpret_cast<u32x4*>(o + (long)m * N + n0 + wn * 64 + 32 * c + 8 * q) = v;
    }
  } else if (MODE == 1) {     // the same 128 KiB per tile, each instruction = 2 tile rows of 512 B
    unsigned short* o = (unsigned short*)out;
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (int it = 0; it < 8; ++it) {
      const int r = (wave * 8 + it) * 2 + (lane >> 5);
      const int m = m0 + r;
      if (m >= M) continue;
      *reinterpret_cast<u32x4*>(o + (long)m * N + n0 + (lane & 31) * 8) = v;
    }
  } else if (MODE == 2) {     // fp32: lane holds 8 consecutive floats as two 16-B pieces per (mt, c)
    const float* r = (const float*)in;
    float* o = (float*)out;
    for (int half = 0; half < 2; ++half) {
      f32x4 rv[2][4];
      for (int h = 0; h < 2; ++h) {
        const int m = min(m0 + wm * 64 + (2 * half + h) * 16 + i, M - 1);
        for (int nt = 0; nt < 4; ++nt)
          rv[h][nt] = *reinterpret_cast<const f32x4*>(r + (long)m * N + n0 + wn * 64 + 8 * q + 32 * (nt >> 1) + 4 * (nt & 1));
      }
      for (int h = 0; h < 2; ++h) {
        const int m = m0 + wm * 64 + (2 * half + h) * 16 + i;
        if (m >= M) continue;
        for (int nt = 0; nt < 4; ++nt)
          *reinterpret_cast<f32x4*>(o + (long)m * N + n0 + wn * 64 + 8 * q + 32 * (nt >> 1) + 4 * (nt & 1)) = rv[h][nt] + 1.0f;
      }
    }
  } else {                    // fp32 row-contiguous: each instruction = 1 tile row of 1 KiB; 16 rows per wave
    const float* r = (const float*)in;
    float* o = (float*)out;
    for (int half = 0; half < 2; ++half) {
      f32x4 rv[8];
      for (int it = 0; it < 8; ++it) {
        const int m = min(m0 + wave * 16 + half * 8 + it, M - 1);
        rv[it] = *reinterpret_cast<const f32x4*>(r + (long)m * N + n0 + lane * 4);
      }
      for (int it = 0; it < 8; ++it) {
        const int m = m0 + wave * 16 + half * 8 + it;
        if (m >= M) continue;
        *reinterpret_cast<f32x4*>(o + (long)m * N + n0 + lane * 4) = rv[it] + 1.0f;
      }
    }
  }
}

template <int MODE>
void run(const char* name, int M, int N, double bytes) {
  void *a, *b;
  hipMalloc(&a, (size_t)M * N * 4);
  hipMalloc(&b, (size_t)M * N * 4);
  hipMemset(a, 0, (size_t)M * N * 4);
  const int tiles_n = N / 256, tiles = ((M + 255) / 256) * tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(pat<MODE>, dim3(tiles), dim3(1024), 0, 0, b, a, M, N, tiles_n);
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(pat<MODE>, dim3(tiles), dim3(1024), 0, 0, b, a, M, N, tiles_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s M=%d N=%d  %7.1f us  %6.2f TB/s\n", name, M, N, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
  hipFree(a); hipFree(b);
}

int main() {
  const int M = 50208;
  run<0>("bf16 store, epilogue pattern", M, 3072, (double)M * 3072 * 2);
  run<1>("bf16 store, row-contiguous", M, 3072, (double)M * 3072 * 2);
  run<0>("bf16 store, epilogue pattern", M, 768, (double)M * 768 * 2);
  run<1>("bf16 store, row-contiguous", M, 768, (double)M * 768 * 2);
  run<2>("fp32 load+store, epilogue pattern", M, 768, (double)M * 768 * 8);
  run<3>("fp32 load+store, row-contiguous", M, 768, (double)M * 768 * 8);
  return 0;
}
