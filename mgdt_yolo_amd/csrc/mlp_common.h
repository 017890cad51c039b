// Device helpers shared by the ConvNeXtV2 kernels (mlp_chain.hip: the two-pass MLP; cnx_block.hip: the whole block in one launch).
#pragma once
#include "common.h"

__device__ __forceinline__ float row_sum16(float v) {   // sum over the 16 lanes of a row; valid in lane 15 of the row
  int x;
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true); v += __builtin_bit_cast(float, x);
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true); v += __builtin_bit_cast(float, x);
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true); v += __builtin_bit_cast(float, x);
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true); v += __builtin_bit_cast(float, x);
  return v;
}

// GELU = v * Phi(v) = max(v, 0) - |v| * Phi(-|v|), with the normal tail through ONE transcendental:
// log2 Phi(-a) is smooth, a degree-7 polynomial (least-squares fit on [0, 6.5], beyond that |v| * tail < 3e-10) reproduces the exact
// erf GELU to 6e-7 absolute - fp32 rounding level, far below the bf16 resolution the hidden map is rounded to.  libm's erff costs
// ~4x more VALU time and the two passes were bound by it (2 * 4C GELUs per pixel).
__device__ __forceinline__ float gelu_fast(float v) {
  const float a = fminf(fabsf(v), 6.5f);
  float p = fmaf(a, -1.80876783e-06f, 6.10729982e-05f);
  p = fmaf(a, p, -9.26397088e-04f);
  p = fmaf(a, p, 8.49198863e-03f);
  p = fmaf(a, p, -5.39291965e-02f);
  p = fmaf(a, p, -4.58491793e-01f);
  p = fmaf(a, p, -1.15124323e+00f);
  p = fmaf(a, p, -9.99995048e-01f);
  return fmaf(-fabsf(v), __builtin_amdgcn_exp2f(p), fmaxf(v, 0.f));
}


// four values at once with the polynomial on the packed-fp32 pipe.  The coefficients come in as a run-time array (kernel argument ->
// SGPRs): given literals the compiler prefers one v_fmaak_f32 per value, with register operands it emits v_pk_fma_f32, two values per
// issue slot (the GELU polynomial is what bounds pwconv1's epilogue).
typedef __attribute__((ext_vector_type(2))) float f32x2;
struct GeluCoef { float c[8]; float cap; };
static inline GeluCoef gelu_coef() {
  return GeluCoef{{-1.80876783e-06f, 6.10729982e-05f, -9.26397088e-04f, 8.49198863e-03f, -5.39291965e-02f, -4.58491793e-01f, -1.15124323e+00f, -9.99995048e-01f}, 6.5f};
}
__device__ __forceinline__ f32x4 gelu_fast4(f32x4 v, const GeluCoef& k) {
  f32x2 a[2], p[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a[h] = f32x2{fminf(fabsf(v[2 * h]), k.cap), fminf(fabsf(v[2 * h + 1]), k.cap)};
    p[h] = __builtin_elementwise_fma(a[h], f32x2{k.c[0], k.c[0]}, f32x2{k.c[1], k.c[1]});
#pragma unroll
    for (int t = 2; t < 8; ++t) p[h] = __builtin_elementwise_fma(a[h], p[h], f32x2{k.c[t], k.c[t]});
  }
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = fmaf(-fabsf(v[i]), __builtin_amdgcn_exp2f(p[i >> 1][i & 1]), fmaxf(v[i], 0.f));
  return o;
}
