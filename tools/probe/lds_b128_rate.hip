// ds_read_b128 / ds_write_b128 throughput on one CU (8 waves), cycles per wave-instruction, for
//   linear      lane l -> 16 l                                   (conflict-free by construction)
//   nt8 A frag  lane (i, q) -> row i of a 128-byte-row tile, chunk q ^ swz(row)   (gemm_nt8_core.h rdAk, swz_x)
//   tn frag     lane (i, q) -> q * 4096 + slot(i, t) * 16 + t * 256               (gemm_tn_core.h frd[])
// and the same with 4 / 8 reads in flight before the wait.  The LDS port moves 128 B per cycle at its peak: 8 cycles per 1 KiB instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_b128_rate tools/probe/lds_b128_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WRITE>
__global__ __launch_bounds__(512) void probe(const int* offs, unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  const int tid = threadIdx.x;
  for (int i = tid; i < 131072 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
  unsigned o[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) o[u] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + offs[u * 512 + tid];
  __syncthreads();
  u32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = (u32x4){(unsigned)tid, 1u, 2u, 3u};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (WRITE) asm volatile("ds_write_b128 %0, %1" :: "v"(o[u]), "v"(v[u]) : "memory");
      else asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(o[u]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = clock64();
  unsigned acc = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) acc += v[u][0] ^ v[u][3];
  out[tid] = acc;
  if (tid == 0) cyc[0] = t1 - t0;
}

static int *d_offs; static unsigned* d_out; static long long* d_cyc;
template <int WRITE> double run(const std::vector<int>& offs, int threads) {
  hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int r = 0; r < 2; ++r) { probe<WRITE><<<1, threads>>>(d_offs, d_out, d_cyc, iters); hipDeviceSynchronize(); }
  long long h; hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost);
  return (double)h / ((double)iters * 8 * (threads / 64));
}

int main() {
  hipMalloc(&d_offs, 8 * 512 * 4); hipMalloc(&d_out, 512 * 4); hipMalloc(&d_cyc, 8);
  std::vector<int> lin(8 * 512), nt8(8 * 512), tn(8 * 512);
  for (int u = 0; u < 8; ++u) for (int t = 0; t < 512; ++t) {
    const int wave = t >> 6, lane = t & 63, i = lane & 15, q = lane >> 4;
    lin[u * 512 + t] = ((u * 8 + wave) * 1024 + lane * 16) % 131072;
    { const int wm = wave >> 2, row = 64 * wm + 16 * (u & 3) + i, ks = u >> 2;       // rdAk: 4 row tiles, 2 k halves
      nt8[u * 512 + t] = (row * 128 + ((q ^ ((row >> 1) & 7)) << 4)) ^ (ks * 64); }
    { const int tt = u & 3, h = u >> 2, slot = (i & 8) | ((i & 7) ^ ((2 * tt + (i >> 3)) & 7));
      tn[u * 512 + t] = q * 4096 + (slot << 4) + tt * 256 + h * 16384 + (wave & 1) * 2048 + (wave >> 1) * 1024 + 32768 * (wave & 1); }
  }
  for (int threads : {256, 512}) {
    printf("%d waves: read  linear %5.2f | nt8 A fragments %5.2f | tn fragments %5.2f   cycles per ds_read_b128 (1 KiB)\n", threads / 64,
           run<0>(lin, threads), run<0>(nt8, threads), run<0>(tn, threads));
    printf("%d waves: write linear %5.2f                                            cycles per ds_write_b128\n", threads / 64, run<1>(lin, threads));
  }
  return 0;
}
