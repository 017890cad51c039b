// Vectorised training-mode BatchNorm kernels (statistics, normalise + activation, backward sums, backward apply) for pixel-linear NHWC
// views: 4 channels (8 / 16 bytes) per thread, per-thread fp32 partial sums widened to fp64 only when threads are combined, the activation a
// template constant (a run-time switch inside the element loop serialises four exp -> rcp chains).  Same math and the same partial-sum
// format as the scalar kernels in train.hip, which stay as the fallback for spatially strided views.
// Reference: nn.BatchNorm2d in training mode + the activation of Conv (nn/modules/conv.py:25-42) and their autograd.
#include "common.h"

#define BNF_SPLITS 512          // == RED_SPLITS of train.hip (the final kernels there sum this many partial rows)

template <int ACT> __device__ __forceinline__ float bnf_act(float u) {
  if (ACT == MGDT_ACT_SILU) return u * fast_sigmoid(u);
  if (ACT == MGDT_ACT_RELU) return fmaxf(u, 0.f);
  if (ACT == MGDT_ACT_GELU) return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
  return u;
}
template <int ACT> __device__ __forceinline__ float bnf_grad(float u) {
  if (ACT == MGDT_ACT_SILU) { const float s = fast_sigmoid(u); return s * (1.f + u * (1.f - s)); }
  if (ACT == MGDT_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (ACT == MGDT_ACT_GELU) return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
  return 1.f;
}

struct BnfView { const char* p; long pix; };      // pixel-linear view: element (pixel, c) at p + (pixel * pix + c) * sizeof(T)

// per-channel partial sums of (f0, f1) over this workgroup's pixel range -> partial[split][c][2] (double); F: (pixel, c0) -> two f32x4
template <typename F>
__device__ __forceinline__ void bnf_reduce(long npix, int C, double* partial, F f) {
  const int Q = C >> 2;
  const int PL = 256 / (Q < 256 ? Q : 256);                      // pixel lanes per pass (Q <= 256: C <= 1024)
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int split = blockIdx.x;
  const long p0 = split * npix / BNF_SPLITS, p1 = (split + 1) * npix / BNF_SPLITS;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    long p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) f(p + u * PL, q * 4, a[u], b[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += a[u]; s1 += b[u]; }
    }
    for (; p < p1; p += PL) { f32x4 a, b; f(p, q * 4, a, b); s0 += a; s1 += b; }
  }
  __shared__ float red[2][4][256];
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][j][threadIdx.x] = s0[j]; red[1][j][threadIdx.x] = s1[j]; }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += 256) {                   // channel o = quad o/4, element o%4: sum its PL pixel lanes in order
    const int qq = o >> 2, j = o & 3;
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < PL; ++k) { t0 += (double)red[0][j][k * Q + qq]; t1 += (double)red[1][j][k * Q + qq]; }
    partial[((long)split * C + o) * 2] = t0;
    partial[((long)split * C + o) * 2 + 1] = t1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bnf_stats_kernel(BnfView y, long npix, int C, double* partial) {
  bnf_reduce(npix, C, partial, [&](long p, int c0, f32x4& a, f32x4& b) {
    const f32x4 v = load4<T>((const T*)y.p + p * y.pix + c0);
    a = v; b = v * v;
  });
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bnf_fwd_kernel(BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, BnfView r1, BnfView r2, BnfView z, long nquads, int Q) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nquads; i += (long)gridDim.x * 256) {
    const long p = i / Q;
    const int c0 = (int)(i - p * Q) * 4;
    const f32x4 v = load4<T>((const T*)y.p + p * y.pix + c0);
    f32x4 o;
    if (mean) {
      const f32x4 m = *(const f32x4*)(mean + c0), rs = *(const f32x4*)(rstd + c0), g = *(const f32x4*)(gamma + c0), b = *(const f32x4*)(beta + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bnf_act<ACT>(g[j] * ((v[j] - m[j]) * rs[j]) + b[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bnf_act<ACT>(v[j] + (beta ? beta[c0 + j] : 0.f));
    }
    if (r1.p) o += load4<T>((const T*)r1.p + p * r1.pix + c0);
    if (r2.p) o += load4<T>((const T*)r2.p + p * r2.pix + c0);
    store4<T>((T*)z.p + p * z.pix + c0, o);
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bnf_bwd_partial_kernel(BnfView gz, BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, long npix, int C, double* partial) {
  // this thread's four channels are fixed (bnf_reduce: quad = threadIdx.x % Q): their parameters are loaded once, not per pixel
  const int cq = (threadIdx.x % (C >> 2)) * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 m = mean ? *(const f32x4*)(mean + cq) : zero, rs = mean ? *(const f32x4*)(rstd + cq) : zero;
  const f32x4 ga = mean ? *(const f32x4*)(gamma + cq) : zero, be = beta ? *(const f32x4*)(beta + cq) : zero;
  const bool bn = mean != nullptr;
  bnf_reduce(npix, C, partial, [&](long p, int c0, f32x4& a, f32x4& b) {
    const f32x4 v = load4<T>((const T*)y.p + p * y.pix + c0), gv = load4<T>((const T*)gz.p + p * gz.pix + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = bn ? (v[j] - m[j]) * rs[j] : v[j];
      const float u = bn ? ga[j] * xh + be[j] : v[j] + be[j];
      const float g = gv[j] * bnf_grad<ACT>(u);
      a[j] = g; b[j] = g * xh;
    }
  });
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bnf_bwd_apply_kernel(BnfView gz, BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ coef,
                                                            BnfView dy, long nquads, int Q) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nquads; i += (long)gridDim.x * 256) {
    const long p = i / Q;
    const int c0 = (int)(i - p * Q) * 4;
    const f32x4 v = load4<T>((const T*)y.p + p * y.pix + c0), gv = load4<T>((const T*)gz.p + p * gz.pix + c0);
    f32x4 o;
    if (mean) {
      const f32x4 m = *(const f32x4*)(mean + c0), rs = *(const f32x4*)(rstd + c0), ga = *(const f32x4*)(gamma + c0), be = *(const f32x4*)(beta + c0);
      const f32x4 k0 = *(const f32x4*)(coef + 2 * c0), k1 = *(const f32x4*)(coef + 2 * c0 + 4);     // (mean g, mean g*xhat) pairs of 4 channels
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - m[j]) * rs[j];
        const float g = gv[j] * bnf_grad<ACT>(ga[j] * xh + be[j]);
        const float mg = j < 2 ? k0[2 * j] : k1[2 * j - 4], mgx = j < 2 ? k0[2 * j + 1] : k1[2 * j - 3];
        o[j] = ga[j] * rs[j] * (g - mg - xh * mgx);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gv[j] * bnf_grad<ACT>(v[j] + (beta ? beta[c0 + j] : 0.f));
    }
    store4<T>((T*)dy.p + p * dy.pix + c0, o);
  }
}

// ---- host: a view qualifies when its pixels are equally spaced (sh == W*sw, sn == H*sh), channels contiguous and 4-aligned
static bool bnf_linear(const mgdt_view* v, int dtype, BnfView* out) {
  if (!v || !v->p) { out->p = nullptr; out->pix = 0; return true; }
  if (v->sc != 1 || v->c % 4 || v->sw % 4 || v->sh != (long)v->w * v->sw || v->sn != (long)v->h * v->sh || (uintptr_t)v->p % (4 * dtype_size(dtype))) return false;
  out->p = (const char*)v->p; out->pix = v->sw;
  return true;
}
static inline int bnf_grid(long n) { return (int)std::min<long>((n + 255) / 256, 8192); }

#define BNF_ACT_DISPATCH(act, ...)                                       \
  switch (act) {                                                         \
    case MGDT_ACT_SILU: { constexpr int ACT = MGDT_ACT_SILU; __VA_ARGS__; } break; \
    case MGDT_ACT_RELU: { constexpr int ACT = MGDT_ACT_RELU; __VA_ARGS__; } break; \
    case MGDT_ACT_GELU: { constexpr int ACT = MGDT_ACT_GELU; __VA_ARGS__; } break; \
    default: { constexpr int ACT = MGDT_ACT_NONE; __VA_ARGS__; } break;  \
  }

// each returns false when the views do not qualify (the caller runs the scalar kernels)
bool mgdt_bnf_stats(const mgdt_view* y, double* partial, int dtype, hipStream_t st) {
  BnfView v;
  if (y->c > 1024 || !bnf_linear(y, dtype, &v)) return false;
  const long npix = (long)y->n * y->h * y->w;
  if (dtype == MGDT_F32) bnf_stats_kernel<float><<<BNF_SPLITS, 256, 0, st>>>(v, npix, y->c, partial);
  else bnf_stats_kernel<bf16><<<BNF_SPLITS, 256, 0, st>>>(v, npix, y->c, partial);
  return true;
}
bool mgdt_bnf_fwd(const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, const mgdt_view* r1,
                  const mgdt_view* r2, const mgdt_view* z, int dtype, hipStream_t st) {
  BnfView vy, v1, v2, vz;
  if (!bnf_linear(y, dtype, &vy) || !bnf_linear(r1, dtype, &v1) || !bnf_linear(r2, dtype, &v2) || !bnf_linear(z, dtype, &vz)) return false;
  const int Q = y->c / 4;
  const long nq = (long)y->n * y->h * y->w * Q;
  BNF_ACT_DISPATCH(act, {
    if (dtype == MGDT_F32) bnf_fwd_kernel<float, ACT><<<bnf_grid(nq), 256, 0, st>>>(vy, mean, rstd, gamma, beta, v1, v2, vz, nq, Q);
    else bnf_fwd_kernel<bf16, ACT><<<bnf_grid(nq), 256, 0, st>>>(vy, mean, rstd, gamma, beta, v1, v2, vz, nq, Q);
  });
  return true;
}
bool mgdt_bnf_bwd_partial(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                          double* partial, int dtype, hipStream_t st) {
  BnfView vg, vy;
  if (y->c > 1024 || !bnf_linear(gz, dtype, &vg) || !bnf_linear(y, dtype, &vy)) return false;
  const long npix = (long)y->n * y->h * y->w;
  BNF_ACT_DISPATCH(act, {
    if (dtype == MGDT_F32) bnf_bwd_partial_kernel<float, ACT><<<BNF_SPLITS, 256, 0, st>>>(vg, vy, mean, rstd, gamma, beta, npix, y->c, partial);
    else bnf_bwd_partial_kernel<bf16, ACT><<<BNF_SPLITS, 256, 0, st>>>(vg, vy, mean, rstd, gamma, beta, npix, y->c, partial);
  });
  return true;
}
bool mgdt_bnf_bwd_apply(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                        const float* coef, const mgdt_view* dy, int dtype, hipStream_t st) {
  BnfView vg, vy, vd;
  if (!bnf_linear(gz, dtype, &vg) || !bnf_linear(y, dtype, &vy) || !bnf_linear(dy, dtype, &vd)) return false;
  const int Q = y->c / 4;
  const long nq = (long)y->n * y->h * y->w * Q;
  BNF_ACT_DISPATCH(act, {
    if (dtype == MGDT_F32) bnf_bwd_apply_kernel<float, ACT><<<bnf_grid(nq), 256, 0, st>>>(vg, vy, mean, rstd, gamma, beta, coef, vd, nq, Q);
    else bnf_bwd_apply_kernel<bf16, ACT><<<bnf_grid(nq), 256, 0, st>>>(vg, vy, mean, rstd, gamma, beta, coef, vd, nq, Q);
  });
  return true;
}
