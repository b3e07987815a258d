// Ablation probe for the main loop of the default NT GEMM kernel (procedurevrl_amd/csrc/gemm_nt.hip, gemm_nt_kernel
// <EPI, 4, 4>): the same 256x256 tile / 16 waves / BK = 64 / 2-stage LDS-DMA loop, with one ingredient removed per
// variant, to see what the ~44 % MFMA-busy figure is made of.  Results of variants > 0 are numerically meaningless.
//   0 full loop           1 no LDS-DMA in the loop        2 no fragment reads in the loop (registers reused)
//   3 no MFMA (xor-fold)  4 no barrier (vmcnt wait kept)  5 MFMA only (no DMA, no reads, no barrier)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o nt_ablate nt_ablate.hip ; run on the GPU box.
#include "../../procedurevrl_amd/csrc/common.h"
#include <cstdio>
#include <vector>

constexpr int BK = 64;
__device__ __forceinline__ int swz_x(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int w_row(int nt, int i) { return 32 * (nt >> 1) + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3); }
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

struct P { const bf16* A; long lda; const bf16* W; long ldw; int M, N, K; bf16* out; long ldo; int tiles_m, tiles_n, gm; float* ws; int splits; };

template <int ABL>
__global__ __launch_bounds__(1024) void k(P p) {
  constexpr int WM = 4, WN = 4, NW = 16, BM = 256, BN = 256;
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  constexpr int PER = (BM + BN) / 8 / NW;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int GM = p.gm;
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    const int cm = qm + (xcd < rm ? 1 : 0);
    const int mbase = xcd * qm + (xcd < rm ? xcd : rm);
    if (j >= cm * p.tiles_n) return;
    const int gsz = GM * p.tiles_n;
    const int g = j / gsz, r = j - g * gsz;
    const int gm = min(GM, cm - g * GM);
    tn = r / gm;
    tm = mbase + g * GM + (r - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = (ABL == 6) ? (int)blockIdx.y * (p.K / p.splits) : 0;
  const bf16* gsrc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int it = wave * PER + j;
    const int pc = lane & 7;
    if (it < BM / 8) {
      const int row = it * 8 + (lane >> 3);
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      gsrc[j] = p.A + (long)grow * p.lda + kbeg + ((pc ^ swz_x(row)) << 3);
    } else {
      const int row = (it - BM / 8) * 8 + (lane >> 3);
      gsrc[j] = p.W + (long)(n0 + row) * p.ldw + kbeg + ((pc ^ swz_w(row)) << 3);
    }
  }
  auto stage = [&](int buf, int k0) {
    char* b = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(gsrc[j] + k0, b + (wave * PER + j) * 1024);
  };
  int xoff[4], woff[4];
  {
    const int q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rx = wm * 64 + t * 16 + i;
      xoff[t] = rx * 128 + ((q ^ swz_x(rx)) << 4);
      const int rw = wn * 64 + w_row(t, i);
      woff[t] = rw * 128 + ((q ^ swz_w(rw)) << 4);
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = (ABL == 6 ? p.K / p.splits : p.K) / BK;
  stage(0, 0);
  if (ABL == 1 || ABL == 5) stage(1, BK);
  bf16x8 xf0[4], wf0[4], xf1[4], wf1[4];
  if (ABL == 2 || ABL == 5) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf0[t] = *reinterpret_cast<const bf16x8*>(smem + XBYTES + woff[t]);
      xf0[t] = *reinterpret_cast<const bf16x8*>(smem + xoff[t]);
      wf1[t] = *reinterpret_cast<const bf16x8*>(smem + XBYTES + (woff[t] ^ 64));
      xf1[t] = *reinterpret_cast<const bf16x8*>(smem + (xoff[t] ^ 64));
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (ABL == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (ABL != 5) __syncthreads();
    if (ABL != 1 && ABL != 5 && kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* bx = smem + (kt & 1) * STAGE;
    const char* bw = bx + XBYTES;
    if (ABL != 2 && ABL != 5) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf0[t] = *reinterpret_cast<const bf16x8*>(bw + woff[t]);
        xf0[t] = *reinterpret_cast<const bf16x8*>(bx + xoff[t]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf1[t] = *reinterpret_cast<const bf16x8*>(bw + (woff[t] ^ 64));
        xf1[t] = *reinterpret_cast<const bf16x8*>(bx + (xoff[t] ^ 64));
      }
    }
    if (ABL == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { bf16x8 b; f32x4 f; } u0, u1, u2, u3;
        u0.b = wf0[t]; u1.b = xf0[t]; u2.b = wf1[t]; u3.b = xf1[t];
        acc[t][0] += u0.f; acc[t][1] += u1.f; acc[t][2] += u2.f; acc[t][3] += u3.f;
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
      if (ABL != 2 && ABL != 5) {
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
    }
  }
  if (ABL == 6) {   // fp32 partial in register order: [tile][split][wave][reg][lane] x f32x4
    float* w = p.ws + ((((long)(tm * p.tiles_n + tn) * p.splits + blockIdx.y) * 16 + wave) * 16) * 256;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(w + ((mt * 4 + nt) * 64 + lane) * 4) = acc[mt][nt];
    return;
  }
  // plain bf16 epilogue (natural order is irrelevant for timing: 8 consecutive columns per lane)
  const int q = lane >> 4, i = lane & 15;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16)acc[mt][2 * c][e]; o[4 + e] = (bf16)acc[mt][2 * c + 1][e]; }
      *reinterpret_cast<bf16x8*>(p.out + (long)m * p.ldo + n0 + wn * 64 + 32 * c + 8 * q) = o;
    }
  }
}

__global__ __launch_bounds__(1024) void fixup(P p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int y = 0; y < p.splits; ++y) {
    const float* w = p.ws + ((((long)blockIdx.x * p.splits + y) * 16 + wave) * 16) * 256;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] += *reinterpret_cast<const f32x4*>(w + ((mt * 4 + nt) * 64 + lane) * 4);
  }
  const int m0 = tm * 256, n0 = tn * 256;
  const int q = lane >> 4, i = lane & 15;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16)acc[mt][2 * c][e]; o[4 + e] = (bf16)acc[mt][2 * c + 1][e]; }
      *reinterpret_cast<bf16x8*>(p.out + (long)m * p.ldo + n0 + wn * 64 + 32 * c + 8 * q) = o;
    }
  }
}

// main rows as full tiles (whole rounds of 256 CUs) + the ragged last round cut into `splits` K-ranges + fix-up
float run_split(P p, int reps, int splits) {
  const int T = p.tiles_m * p.tiles_n, full = T / 256;
  const int tm_main = full * 256 / p.tiles_n;
  P a = p; a.M = tm_main * 256; a.tiles_m = tm_main;
  P b = p; b.A = p.A + (long)tm_main * 256 * p.lda; b.out = p.out + (long)tm_main * 256 * p.ldo; b.M = p.M - tm_main * 256;
  b.tiles_m = p.tiles_m - tm_main; b.splits = splits;
  const int nwa = 8 * ((a.tiles_m + 7) / 8) * a.tiles_n, nwb = 8 * ((b.tiles_m + 7) / 8) * b.tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto once = [&]() {
    hipLaunchKernelGGL(k<0>, dim3(nwa), dim3(1024), 0, 0, a);
    hipLaunchKernelGGL(k<6>, dim3(nwb, splits), dim3(1024), 0, 0, b);
    hipLaunchKernelGGL(fixup, dim3(b.tiles_m * b.tiles_n), dim3(1024), 0, 0, b);
  };
  for (int i = 0; i < 3; ++i) once();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) once();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("   split: main %d tiles, tail %d tiles x %d -> ", a.tiles_m * a.tiles_n, b.tiles_m * b.tiles_n, splits);
  return ms * 1e3f / reps;
}

template <int ABL>
float run(P p, int reps) {
  const int nwg = 8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<ABL>, dim3(nwg), dim3(1024), 0, 0, p);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<ABL>, dim3(nwg), dim3(1024), 0, 0, p);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const int M = 50208;
  const int shapes[3][2] = {{2304, 768}, {768, 2304}, {768, 3072}};
  for (auto& sh : shapes) {
    const int N = sh[0], K = sh[1];
    std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 31) << 15)); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd();
    bf16 *A, *W, *O;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&O, (size_t)M * N * 2);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    float* WS; hipMalloc(&WS, (size_t)256 * 4 * 256 * 256 * 4);
    P p{A, K, W, K, M, N, K, O, N, (M + 255) / 256, N / 256, 2, WS, 1};
    const double fl = 2.0 * M * N * K;
    const float t0 = run<0>(p, 20), t1 = run<1>(p, 20), t2 = run<2>(p, 20), t3 = run<3>(p, 20), t4 = run<4>(p, 20),
                t5 = run<5>(p, 20);
    printf("N=%d K=%d  full %.1f us (%.0f TF/s) | no-DMA %.1f | no-LDS-read %.1f | no-MFMA %.1f | no-barrier %.1f | MFMA-only %.1f (%.0f TF/s)\n",
           N, K, t0, fl / t0 / 1e6, t1, t2, t3, t4, t5, fl / t5 / 1e6);
    if (N == 768) {   // how much does the ragged third round cost?  510 tiles (two full rounds of 256 CUs) against 591
      P pm = p; pm.M = 170 * 256; pm.tiles_m = 170;
      const float tm = run<0>(pm, 20);
      printf("   M = 43,520 (510 tiles = 1.99 rounds): %.1f us -> %.3f us per tile; M = 50,208 (591 tiles = 2.31 rounds): %.1f us -> %.3f us per tile\n",
             tm, tm / 510, t0, t0 / 591);
    }
    if (N == 768) { for (int sp = 2; sp <= 4; ++sp) { if ((K / 64) % sp) continue; const float ts = run_split(p, 20, sp); printf("%.1f us (full %.1f)\n", ts, t0); } }
    hipFree(A); hipFree(W); hipFree(O); hipFree(WS);
  }
  return 0;
}
