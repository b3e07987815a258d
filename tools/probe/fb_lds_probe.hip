// LDS access-pattern probe for the fused attention backward (csrc/attn_bwd_fused.hip): cycles per wave-instruction of each
// of the kernel's LDS address patterns (row fragments, transposed fragments, the dS^T exchange image), 8 waves on one CU, next
// to the patterns of the two-pass kernels' layout (attn_common.h bl_off).
// build: hipcc --offload-arch=gfx950 -O3 -o fb_lds_probe tools/probe/fb_lds_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int NU = 8;     // independent accesses per round (different offset tables)

// kind: 0 ds_read_b128, 1 ds_read_b64_tr_b16, 2 ds_write_b64, 3 ds_write_b128, 4 ds_read_b64
template <int KIND>
__global__ __launch_bounds__(512) void probe(const int* offs, unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 131072 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
  int o[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) o[u] = offs[(wave * NU + u) * 64 + lane];
  __syncthreads();
  unsigned acc = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      char* a = smem + o[u];
      if (KIND == 0) { u32x4 v = *reinterpret_cast<const u32x4*>(a); acc += v[0] ^ v[3]; }
      else if (KIND == 4) { u32x2 v = *reinterpret_cast<const u32x2*>(a); acc += v[0] ^ v[1]; }
      else if (KIND == 1) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
        acc += (unsigned)v[0] ^ (unsigned)v[3];
      } else if (KIND == 2) { *reinterpret_cast<u32x2*>(a) = (u32x2){acc, (unsigned)it}; }
      else { *reinterpret_cast<u32x4*>(a) = (u32x4){acc, (unsigned)it, 1u, 2u}; }
    }
    if (KIND >= 2 && KIND != 4) __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static int* d_offs; static unsigned* d_out; static long long* d_cyc;

template <int KIND>
double run(const std::vector<int>& offs, int nblk = 1) {
  hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  probe<KIND><<<nblk, 512>>>(d_offs, d_out, d_cyc, iters);
  hipDeviceSynchronize();
  probe<KIND><<<nblk, 512>>>(d_offs, d_out, d_cyc, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost);
  return (double)h / ((double)iters * NU * 8);      // cycles per wave-instruction (8 waves share the LDS)
}

// ---- layouts
static int bl_off(int row, int col) { int rb = row >> 2; return (rb * 4 + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (col & 15) * 2; }
static int fb_off(int row, int col) { int rb = row >> 2; return (rb * 4 + ((col >> 4) ^ (rb & 1))) * 128 + (row & 3) * 32 + (((col & 15) * 2) ^ ((rb & 2) << 3)); }
// candidate: no 16-byte flip, block order rotated by (rb & 3)
static int fc_off(int row, int col) { int rb = row >> 2; return (rb * 4 + ((col >> 4) ^ (rb & 3))) * 128 + (row & 3) * 32 + (col & 15) * 2; }

typedef int (*offfn)(int, int);

// a-operand of v_mfma_32x32x16: lane (n = l&31, g = l>>5) reads row 32 jb + n, 16 bytes at column 16 s + 8 g
static std::vector<int> pat_row32(offfn f) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l)
    v[(w * NU + u) * 64 + l] = f(32 * ((w + u) % 7) + (l & 31), 16 * (u & 3) + 8 * (l >> 5));
  return v;
}
// a-operand of v_mfma_16x16x32: lane (i, q) row 16 t + i, chunk q (or 4 + q)
static std::vector<int> pat_row16(offfn f) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l)
    v[(w * NU + u) * 64 + l] = f(16 * ((w + u) % 13) + (l & 15), 8 * ((l >> 4) + 4 * (u & 1)));
  return v;
}
// transposed a-operand of v_mfma_32x32x16 (dK / dV): lane l: m = l&31 -> column block 2 dh + ((l>>4)&1), rows 32 jb + 16 t + 8 e2 + 4 g
static std::vector<int> pat_tr32(offfn f) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) {
    const int g = l >> 5, hi = (l >> 4) & 1, i = l & 15, t = u & 1, e2 = (u >> 1) & 1, dh = (u >> 2) & 1, jb = (w + u) % 7;
    v[(w * NU + u) * 64 + l] = f(32 * jb + 16 * t + 8 * e2 + 4 * g + (i >> 2), 16 * (2 * dh + hi) + 4 * (i & 3));
  }
  return v;
}
// transposed a-operand of v_mfma_16x16x32 (K^T in the dQ product; also the two-pass kernels' bl_frag): lane (i, q): rows 32 u + 8 q + 4 e2, column block dt
static std::vector<int> pat_tr16(offfn f) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, i = l & 15, e2 = u & 1, dt = (u >> 1) & 3, uu = (w + u) % 7;
    v[(w * NU + u) * 64 + l] = f(32 * uu + 8 * q + 4 * e2 + (i >> 2), 16 * dt + 4 * (i & 3));
  }
  return v;
}
// the two-pass kernels' bl_frag: rows 32 ks2 + 4 q (+16), column block ct
static std::vector<int> pat_tr16_old(offfn f) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, i = l & 15, e2 = u & 1, ct = (u >> 1) & 3, ks2 = (w + u) % 7;
    v[(w * NU + u) * 64 + l] = f(32 * ks2 + 16 * e2 + 4 * q + (i >> 2), 16 * ct + 4 * (i & 3));
  }
  return v;
}
// dS^T image [224 keys][32 queries], [4 keys][16 queries] blocks, 8-byte slot swizzle by (kb & 3) (swz = 1) or none
static int ds_off(int key, int query, int swz) {
  const int kb = key >> 2;
  return (kb * 2 + (query >> 4)) * 128 + (key & 3) * 32 + ((((query & 15) >> 2) ^ (swz ? (kb & 3) : 0)) * 8) + (query & 3) * 2;
}
static std::vector<int> pat_ds_write(int swz) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) {
    const int n = l & 31, g = l >> 5, j = u & 3;
    v[(w * NU + u) * 64 + l] = ds_off(32 * (w % 7) + n, 8 * j + 4 * g, swz) + (u >> 2) * 14336;
  }
  return v;
}
static std::vector<int> pat_ds_tr(int swz) {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, i = l & 15, e2 = u & 1, qt = (u >> 1) & 1, uu = (w + u) % 7;
    v[(w * NU + u) * 64 + l] = ds_off(32 * uu + 8 * q + 4 * e2 + (i >> 2), 16 * qt + 4 * (i & 3), swz);
  }
  return v;
}
// broadcast b128 of the per-row start values: lane (., g) reads floats 32 jb + 8 j + 4 g
static std::vector<int> pat_init() {
  std::vector<int> v(8 * NU * 64);
  for (int w = 0; w < 8; ++w) for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l)
    v[(w * NU + u) * 64 + l] = 4 * (32 * ((w + u) % 7) + 8 * (u & 3) + 4 * (l >> 5)) + 100000;
  return v;
}

int main() {
  hipMalloc(&d_offs, 8 * NU * 64 * 4); hipMalloc(&d_out, 512 * 4); hipMalloc(&d_cyc, 8);
  struct { const char* n; offfn f; } L[] = {{"bl (two-pass layout)", bl_off}, {"fb (fused, 16-B flip)", fb_off}, {"fc (block order ^ rb&3)", fc_off}};
  for (auto& l : L) {
    printf("%-26s row32 b128 %6.2f | row16 b128 %6.2f | tr32 %6.2f | tr16(new rows) %6.2f | tr16(bl_frag rows) %6.2f   cycles / wave-instruction\n", l.n,
           run<0>(pat_row32(l.f)), run<0>(pat_row16(l.f)), run<1>(pat_tr32(l.f)), run<1>(pat_tr16(l.f)), run<1>(pat_tr16_old(l.f)));
  }
  for (int swz = 0; swz < 2; ++swz)
    printf("dS image swz=%d: write b64 %6.2f | tr read %6.2f\n", swz, run<2>(pat_ds_write(swz)), run<1>(pat_ds_tr(swz)));
  printf("init broadcast b128 %6.2f\n", run<0>(pat_init()));
  return 0;
}
