// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the
// ProcedureVRL video-narration training hot path.  Wave size is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// The 16-bit OPERAND TYPE of the datapath (GEMM / attention operands and the activations stored between kernels; everything
// else -- accumulators, residual stream, statistics, losses, parameter gradients -- is fp32).  Fixed at build time:
//   default            bf16   (libpvrl_hip.so)      8 exponent bits: gradients need no scaling
//   -DPVRL_OPERAND_F16 fp16   (libpvrl_hip_f16.so)  3 more mantissa bits at the same MFMA rate (v_mfma_f32_*_f16 = *_bf16
//                              on gfx950): 8x smaller rounding error -- the flavour that meets the 1e-3 parity bar; its
//                              gradients are scaled inside each engine's backward (engine.GradStore.begin_scaled; 5 exponent bits)
// `pvrl_operand_dtype()` reports which one a library was built with.
#if defined(PVRL_OPERAND_F16)
typedef _Float16 op_t;
#define PVRL_OPERAND_CODE 1
#define MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define FDOT2_F32 __builtin_amdgcn_fdot2
#else
typedef __bf16 op_t;
#define PVRL_OPERAND_CODE 0
#define MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define FDOT2_F32 __builtin_amdgcn_fdot2_f32_bf16
#endif
typedef op_t opx2 __attribute__((ext_vector_type(2)));
typedef op_t opx4 __attribute__((ext_vector_type(4)));
typedef op_t opx8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// two packed operands (one dword) -> fp32
__device__ __forceinline__ void op_unpack2(unsigned w, float& lo, float& hi) {
#if defined(PVRL_OPERAND_F16)
  union { unsigned u; opx2 v; } x;
  x.u = w;
  lo = (float)x.v[0]; hi = (float)x.v[1];
#else
  lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);     // bf16 = the upper half of an fp32
#endif
}

#define PVRL_OK 0
#define PVRL_EINVAL (-1)
#define PVRL_EHIP (-2)

#define PVRL_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return PVRL_EHIP - (int)e__ * 16;  \
  } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte async global->LDS copy.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// The same copy issued as raw ISA: wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset, LDS base in M0.
// The compiler does not see an LDS write here, so it does not drain vmcnt(0) in front of later LDS reads: callers
// order the copy against their reads themselves (counted `s_waitcnt vmcnt(N)` + `s_barrier`).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_raw(const void* uniform_base, unsigned lane_off, unsigned lds_wave_base) {
  // (s_nop 4: the base may have been produced by v_readfirstlane right in front of the statement -- a VALU-written SGPR needs five
  //  wait states before a vector-memory instruction reads it as an address; hipcc pads nothing inside an asm string)
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(lane_off), "s"(uniform_base), "s"(lds_wave_base) : "memory", "m0");
}
// per-lane 64-bit source address form of the same raw LDS-DMA copy
__device__ __forceinline__ void glds16_raw_v(const void* lane_ptr, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(lane_ptr), "s"(lds_wave_base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact-erf GELU (nn.GELU(), lib/models/vit.py:29-41) and its derivative for the GEMM epilogues, whose outputs are 16-bit: ONE
// transcendental per element and no reciprocal.  With x = min(|u|, 6) and h(x) = Phi(-x) (the normal CDF's lower tail):
//     gelu(u)  = max(u, 0) - x h(x)                       (u >= 0: u (1 - h);  u < 0: u h)
//     gelu'(u) = 1/2 + copysign(1/2 - G(x), u),   G(x) = Phi(-x) - x phi(x)     (gelu'(-x) = G, gelu'(x) = 1 - G)
//     h(x) = 2^-(1 + x R(x)),            R of degree 5:  |error| <= 2.8e-7 absolute and <= 5.7e-4 relative down to h(6) = 1e-9
//     G(x) = 2^(-x^2 log2(e) / 2) N(x),  N of degree 10, N(0) = 1/2:  |error| <= 2e-7
// Minimax fits over [0, 6] and the check against the exact functions in emulated fp32: tools/probe/gelu_fit.py -- |gelu error| <=
// 5.4e-7, |gelu' error| <= 1.9e-7, the class of the Abramowitz-Stegun 7.1.26 form used before (4.6e-7 / 3.0e-7; v_rcp + v_exp + 17
// VALU instructions per element).  Measured on MI355X (profiles/r4_nt_epilogue.txt): v_exp_f32 / v_rcp_f32 issue in 6 cycles per wave,
// v_fma_f32 in 4.3, v_pk_fma_f32 in 5 (two elements); the GELU GEMM 313 -> 305 us, the dGELU GEMM and the step unchanged within noise (the
// epilogues are bound by their store traffic, the math runs under it).  Beyond |u| = 6 both saturate (gelu -> u or -6e-9, gelu' -> 1 or 0); a NaN goes
// through the select and comes out as NaN.
#ifndef PVRL_GELU_FORM
#define PVRL_GELU_FORM 1      // 0: the Abramowitz-Stegun form (A/B builds, tools/build_variant.py)
#endif
#ifndef PVRL_DGELU_FORM
#define PVRL_DGELU_FORM PVRL_GELU_FORM
#endif
__device__ __forceinline__ float gelu_clamp6(float u) {
  const float a = fabsf(u);
  return a > 6.0f ? 6.0f : a;       // (not fminf: v_min_f32 drops a NaN)
}
// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute): v_rcp_f32 (1 ulp, not the 10-instruction IEEE division), v_exp
// and five FMAs; `e` returns exp(-x^2) for reuse by the derivative.  Kept for A/B builds (PVRL_GELU_FORM=0).
__device__ __forceinline__ float erf_as(float x, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  e = __expf(-ax * ax);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * e;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float u) {
#if PVRL_GELU_FORM == 0
  float e;
  return 0.5f * u * (1.0f + erf_as(u * 0.70710678118654752440f, e));
#else
  const float x = gelu_clamp6(u);
  float p = fmaf(-2.9932052711956203e-05f, x, 0.0007281892467290163f);
  p = fmaf(p, x, -0.007909782230854034f);
  p = fmaf(p, x, 0.053108397871255875f);
  p = fmaf(p, x, 0.45901164412498474f);
  p = fmaf(p, x, 1.1511247158050537f);
  p = fmaf(p, x, 1.0f);
  const float h = __builtin_amdgcn_exp2f(-p);
  return fmaf(-x, h, fmaxf(u, 0.0f));
#endif
}
// two elements at a time: packed-fp32 FMAs (v_pk_fma_f32, two lanes' worth per issue slot) -- written out because under the GELU
// GEMM's register pressure the compiler otherwise falls back to one v_fmaak_f32 per element and coefficient
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t u) {
#if PVRL_GELU_FORM == 0
  return (f32x2_t){gelu_erf(u[0]), gelu_erf(u[1])};
#else
  const f32x2_t x = {gelu_clamp6(u[0]), gelu_clamp6(u[1])};
  f32x2_t p = __builtin_elementwise_fma((f32x2_t)(-2.9932052711956203e-05f), x, (f32x2_t)(0.0007281892467290163f));
  p = __builtin_elementwise_fma(p, x, (f32x2_t)(-0.007909782230854034f));
  p = __builtin_elementwise_fma(p, x, (f32x2_t)(0.053108397871255875f));
  p = __builtin_elementwise_fma(p, x, (f32x2_t)(0.45901164412498474f));
  p = __builtin_elementwise_fma(p, x, (f32x2_t)(1.1511247158050537f));
  p = __builtin_elementwise_fma(p, x, (f32x2_t)(1.0f));
  const f32x2_t h = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
  const f32x2_t r = {fmaxf(u[0], 0.0f), fmaxf(u[1], 0.0f)};
  return __builtin_elementwise_fma(-x, h, r);
#endif
}
__device__ __forceinline__ float gelu_erf_grad(float u) {
#if PVRL_DGELU_FORM == 0
  float e;  // = exp(-u^2 / 2)
  const float cdf = 0.5f * (1.0f + erf_as(u * 0.70710678118654752440f, e));
  return fmaf(u * 0.39894228040143267794f, e, cdf);
#else
  const float x = gelu_clamp6(u);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
  float n = fmaf(1.678248281677952e-06f, x, -3.60403792001307e-05f);
  n = fmaf(n, x, 0.00034598674392327666f);
  n = fmaf(n, x, -0.002000307897105813f);
  n = fmaf(n, x, 0.008002669550478458f);
  n = fmaf(n, x, -0.024410134181380272f);
  n = fmaf(n, x, 0.061244383454322815f);
  n = fmaf(n, x, -0.13256259262561798f);
  n = fmaf(n, x, 0.24993102252483368f);
  n = fmaf(n, x, -0.7978806495666504f);
  n = fmaf(n, x, 0.5f);
  return 0.5f + copysignf(0.5f - e * n, u);
#endif
}
__device__ __forceinline__ float quick_gelu(float u) {
  return u * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * u));
}
__device__ __forceinline__ float quick_gelu_grad(float u) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * u));
  return s * (1.0f + 1.702f * u * (1.0f - s));
}

// XCD-aware bijective block remap: hardware places block b on XCD b % 8; give
// each XCD a contiguous run of logical tile ids so neighbours share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// CUs per XCD that the PERSISTENT / one-round kernels of this library size their grids for (gemm_nt8, gemm_tn8 and its grouped
// launch, attn_bwd_fused: one 512-thread workgroup owns a whole CU -- 256 VGPRs per lane, 128 KB of LDS).  Default: all of them
// (MI355X: 256 CUs / 8 XCDs = 32).  PVRL_COMPUTE_CUS=<n> (read once per process, like the PVRL_NT* A/B switches: the one piece
// of process-wide launch configuration in the library) leaves 32 - n CUs per XCD to kernels of OTHER streams: in a data-parallel job
// RCCL's channel kernels take CUs at a kernel seam and then hold them for the length of a collective; a persistent grid sized for
// CUs it cannot get pays a second wave of workgroups -- up to 2x on that launch (distributed.reserve_comm_cus sets both this and
// RCCL's channel count before the first launch; measured cost of the reservation on one GPU: DESIGN.md section 6).
static inline int pvrl_compute_cus_per_xcd() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      n = 256;
    int c = n / 8;
    const char* e = getenv("PVRL_COMPUTE_CUS");
    if (e && atoi(e) > 0 && atoi(e) < c) c = atoi(e);
    cus = c;
  }
  return cus;
}
