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
