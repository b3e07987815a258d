// Hardware-semantics probe for ds_read_b64_tr_b16 and the MFMA C/D layout on gfx950.
// Prints what every lane receives so the kernels' layout assumptions can be checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_kernel(int mode, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  int byte_off;
  if (mode == 0) byte_off = lane * 8;                                   // canonical: 128 B contiguous per 16 lanes
  else { const int g = lane >> 4, i = lane & 15; byte_off = g * 2048 + (i >> 2) * 256 + (i & 3) * 8; }  // rows 256 B apart
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + byte_off));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}

// D = A.B with A[i][k] = (i == k'), check which (row, col) each accumulator register holds.
__global__ void mfma_kernel(float* out) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  // a: rows i = lane&15, k = 8*(lane>>4)+e ; A[i][k] = i*100 + k      b: B[k][j] = (k == 3) ? j + 1 : 0
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (lane >> 4) + e;
    a[e] = (__bf16)(float)((lane & 15) == 5 && k == 3 ? 1.0f : 0.0f);   // A = e_{5,3}
    b[e] = (__bf16)(float)(k == 3 ? (float)((lane & 15) + 1) : 0.0f);   // B[3][j] = j+1
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];   // expect D[5][j] = j+1 : lane with row 5 => lane>>4 == 1, reg 1, value (lane&15)+1
}

int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("tr mode %d (element indices received per lane):\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  float* f; hipMalloc(&f, 256 * 4); float hf[256];
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, f);
  hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
  printf("mfma nonzero accumulators (lane, reg, value):\n");
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hf[l * 4 + r] != 0.f) printf("  lane %2d reg %d = %g\n", l, r, hf[l * 4 + r]);
  return 0;
}
