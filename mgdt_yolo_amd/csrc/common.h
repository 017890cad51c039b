// Shared device/host helpers for libmgdt_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mgdt.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MGDT_WAVE 64

// ---- error plumbing (thread-local text, int status across the C boundary) ----
void mgdt_set_error(const char* fmt, ...);
#define MGDT_FAIL(code, ...)        \
  do {                              \
    mgdt_set_error(__VA_ARGS__);    \
    return (code);                  \
  } while (0)
#define MGDT_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline bool view_ok(const mgdt_view* v) { return v && v->p && v->n > 0 && v->h > 0 && v->w > 0 && v->c > 0; }
static inline bool view_nhwc(const mgdt_view* v) { return v->sc == 1; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t dtype_size(int dt) { return dt == MGDT_BF16 ? 2 : 4; }

// ---- scalar load/store with conversion ----
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16>(bf16* p, float v) { *p = (bf16)v; }

template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
  *(bf16x4*)p = o;
}
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  bf16x4 o = *(const bf16x4*)p;
  return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}


// sigmoid through the hardware exp2 / rcp units (~1 ulp each): the SiLU epilogue must not turn HBM-bound convs VALU-bound
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// fp8 (OCP e4m3fn on gfx950) operands: 8 values per lane in one 64-bit register, same (row, 8g..8g+7) K layout as the bf16 16x16x32 form,
// same accumulator layout, the bf16 form's rate (MI355X_MICROARCH.md: non-scaled fp8 = bf16 cycles).
__device__ __forceinline__ f32x4 mma_q8(long w, long p, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w, p, acc, 0, 0, 0);
}
// 8 bf16 activations -> 8 e4m3 bytes: x * xq, clamped to the finite e4m3 range (the conversion itself does not saturate), round-to-nearest-even
__device__ __forceinline__ long quant8(bf16x8 v, float xq) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  typedef __attribute__((ext_vector_type(2))) int i32x2;
  const u32x4 d = __builtin_bit_cast(u32x4, v);
  auto cv = [&](unsigned int w, float& lo, float& hi) __attribute__((always_inline)) {
    lo = __builtin_amdgcn_fmed3f(__uint_as_float(w << 16) * xq, -448.f, 448.f);
    hi = __builtin_amdgcn_fmed3f(__uint_as_float(w & 0xffff0000u) * xq, -448.f, 448.f);
  };
  float l0, h0, l1, h1, l2, h2, l3, h3;
  cv(d[0], l0, h0); cv(d[1], l1, h1); cv(d[2], l2, h2); cv(d[3], l3, h3);
  i32x2 o = {0, 0};
  o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(l0, h0, o[0], false);
  o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(l1, h1, o[0], true);
  o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(l2, h2, o[1], false);
  o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(l3, h3, o[1], true);
  return __builtin_bit_cast(long, o);
}

// exact unsigned division by a runtime constant: q = mulhi(n, mul) >> sh  (n < 2^31), host-side magic numbers
struct FastDiv { uint32_t mul, sh, d; };
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d;
  if (d == 1) { f.mul = 0; f.sh = 0; return f; }
  uint32_t l = 0; while ((1u << l) < d) ++l;            // ceil(log2 d)
  uint64_t m = ((uint64_t)1 << (32 + l - 1)) / d + 1;     // round-up magic for 31-bit n
  f.mul = (uint32_t)m; f.sh = l - 1;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case MGDT_ACT_SILU: return v * fast_sigmoid(v);
    case MGDT_ACT_RELU: return fmaxf(v, 0.0f);
    case MGDT_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    default: return v;
  }
}

#define MGDT_DISPATCH_DTYPE(dt, ...)                                    \
  do {                                                                  \
    if ((dt) == MGDT_F32) { using T = float; __VA_ARGS__; }             \
    else if ((dt) == MGDT_BF16) { using T = bf16; __VA_ARGS__; }        \
    else MGDT_FAIL(MGDT_BAD_DTYPE, "unsupported dtype %d", (int)(dt));  \
  } while (0)
