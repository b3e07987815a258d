// LDS read-rate probe for gfx950: bytes/clk/CU of ds_read_b128, ds_read_b64 and ds_read_b64_tr_b16 with 4..16 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate tools/probe/lds_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void probe(unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  // conflict-free lane addressing: contiguous per lane
  const int base = (tid >> 6) * 4096;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int off = base + ((it & 1) * 2048) ;
      if (MODE == 0) {
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((off + u * 1024 + lane * 16) & 65535));
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
      } else if (MODE == 1) {
        u32x2 v = *reinterpret_cast<const u32x2*>(smem + ((off + u * 512 + lane * 8) & 65535));
        acc += v[0] ^ v[1];
      } else {
        // tr read: 16-lane group reads a [4][16] bf16 block (128 B): lane i -> row i/4, chunk i%4
        const int q = lane >> 4, i = lane & 15;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + ((off + u * 512 + q * 128 + i * 8) & 65535)));
        acc += (unsigned)v[0] ^ (unsigned)v[1] ^ (unsigned)v[2] ^ (unsigned)v[3];
      }
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves, int bytes_per_lane) {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024 * 4); hipMalloc(&cyc, 8 * 256);
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<256, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<256, waves * 64>>>(out, cyc, iters);
  hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double bytes = (double)iters * 16 * 64 * bytes_per_lane * waves;   // per CU
  printf("%-22s waves/CU=%2d  %.1f B/clk/CU (s_memtime clk=%lld)  wall %.3f ms -> %.1f B/ns/CU\n", name, waves,
         bytes / (double)h[0], h[0], ms, bytes / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {4, 8, 16}) {
    run<0>("ds_read_b128", w, 16);
    run<1>("ds_read_b64", w, 8);
    run<2>("ds_read_b64_tr_b16", w, 8);
  }
  return 0;
}
