// Shared pieces of the MFMA GEMM / implicit-GEMM convolution family for gfx950 (csrc/gemm.hip): vector types, the LDS-DMA primitive,
// MFMA wrappers, 16-bit store conversions, GELU forms, the XCD-aware tile order, diagnostics macros.
// Split out of gemm.hip in round 6 together with gemm_tile.h (tile geometry + operand address generators),
// gemm_mainloop_ring.h / gemm_mainloop_8phase.h (the main loops) and gemm_epilogue.h; every kernel's machine code is unchanged
// (tools/isa_fingerprint.py, profiles/r06_gemm_split_isa_fingerprint.txt).
#pragma once
#include "kernels.h"
#include <type_traits>

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

static constexpr int BK = 64;                  // halves per K-tile -> 128-byte LDS rows
static constexpr uint32_t OOB = 0x80000000u;   // any offset >= num_records reads as zero

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, 0, 0, 0);
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF: the 16-byte fragments hold bf16 (MMDiT path of a bf16 model); same MFMA rate, same register layout
// one K = 128 step on fp8 (e4m3) operands: a = [a0 | a1], b = [b0 | b1] (16 bytes each), unit e8m0 block scales (127 = 2^0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_mx8(const f16x8 a0, const f16x8 a1, const f16x8 b0, const f16x8 b1, const f32x4 c) {
  const i32x4 x0 = __builtin_bit_cast(i32x4, a0), x1 = __builtin_bit_cast(i32x4, a1);
  const i32x4 y0 = __builtin_bit_cast(i32x4, b0), y1 = __builtin_bit_cast(i32x4, b1);
  const i32x8 a = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  const i32x8 b = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, /*A fp8 e4m3*/ 0, /*B fp8 e4m3*/ 0, 0, 127, 0, 127);
}
template <bool BF>
__device__ __forceinline__ f32x4 mfma16(const f16x8 a, const f16x8 b, const f32x4 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// 16-bit store conversions of the epilogue.  UNet kernels (DIT = false): plain fp16 casts, unchanged code.  MMDiT kernels:
// activations (out16) are bf16 or SATURATING fp16, hook copies (aux16) always saturating fp16 (the reference's hooks are fp16,
// feature_extractor.py:59-60, and real FLUX.1-dev activations leave the fp16 range).
template <bool DIT, bool BF>
__device__ __forceinline__ _Float16 act16(float v) {
  if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)v);
  else if constexpr (DIT) return f32_to_f16_sat(v);
  else return (_Float16)v;
}
template <bool DIT>
__device__ __forceinline__ _Float16 hook16(float v) {
  if constexpr (DIT) return f32_to_f16_sat(v);
  else return (_Float16)v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// exact (erf) GELU, two values at a time (the GEGLU epilogue evaluates 64 per lane; at 2 waves per SIMD the VALU time of the
// round-1 form — Abramowitz-Stegun 7.1.26 with one v_rcp_f32 + one v_exp_f32 and ~15 scalar-float ops — was 5.4 us of a 37-us
// tile, tools/trace_gemm.py).  With u = |x| and q(u) = 1 - Phi(u) = erfc(u / sqrt 2) / 2:
//     gelu(x) = x Phi(x) = max(x, 0) - u q(u),        q(u) = 2^P(u),  P = degree-7 fit of log2 q on [0, 5.5], P(0) = -1
// ONE transcendental, and the Horner chain + the final ops run as packed fp32 (v_pk_fma_f32: two lanes' worth per issue).
// |gelu - exact| <= 6.5e-7 absolute and <= 5.4e-6 relative on x > -4.5 (fp16 output rounding: 4.9e-4); u is clamped at 5.5,
// beyond which q < 2e-8 (coefficients: Lawson-weighted least squares on 4000 Chebyshev nodes, checked on 400k points in fp32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 u = {fminf(fabsf(x[0]), 5.5f), fminf(fabsf(x[1]), 5.5f)};
  f32x2 P = u * -1.735116371e-06f + 5.974406668e-05f;
  P = P * u + -9.168680408e-04f;
  P = P * u + 8.457269520e-03f;
  P = P * u + -5.386104062e-02f;
  P = P * u + -4.585619271e-01f;
  P = P * u + -1.151209950e+00f;
  P = P * u + -1.0f;
  const f32x2 q = {__builtin_amdgcn_exp2f(P[0]), __builtin_amdgcn_exp2f(P[1])};
  const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return r - u * q;
}

// XCD-aware bijective remap: consecutive tiles (which share the same A row-block) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// tanh-approximate GELU (activations.GELU(approximate="tanh"), Flux FeedForward / proj_mlp):
// 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp_f32 + one v_rcp_f32
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);               // 2u
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u2));
}

}  // namespace gdf
