// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the
// ProcedureVRL video-narration training hot path.  Wave size is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. fp32 round-off class): one v_exp, one v_rcp
// and five FMAs instead of the ~40-instruction ocml erff; `e` returns exp(-x^2) for reuse by the derivative.
__device__ __forceinline__ float erf_as(float x, float& e) {
  const float ax = fabsf(x);
  // v_rcp_f32 (1 ulp), NOT __frcp_rn / a division: those expand to the IEEE sequence (2 v_div_scale, v_rcp, 4 FMA, v_div_fmas,
  // v_div_fixup = 10 of the GELU epilogue's 21 VALU instructions per element -- round 3: the two-output GELU and the dGELU GEMMs spent
  // ~10 us of VALU time per 256x256 tile there, what had been booked as store time)
#if defined(PVRL_GELU_IEEE_DIV)      // A/B builds only (tools/build_variant.py)
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
#else
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
#endif
  e = __expf(-ax * ax);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * e;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float u) {
  float e;
  return 0.5f * u * (1.0f + erf_as(u * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float u) {
  float e;  // = exp(-u^2 / 2)
  const float cdf = 0.5f * (1.0f + erf_as(u * 0.70710678118654752440f, e));
  return fmaf(u * 0.39894228040143267794f, e, cdf);
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
