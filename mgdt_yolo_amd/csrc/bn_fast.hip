// Vectorised training-mode BatchNorm kernels (statistics, normalise + activation, backward sums, backward apply) for pixel-linear NHWC
// views: 4 channels (8 / 16 bytes) per thread, per-thread fp32 partial sums widened to fp64 only when threads are combined, the activation a
// template constant (a run-time switch inside the element loop serialises four exp -> rcp chains).  Same math and the same partial-sum
// format as the scalar kernels in train.hip, which stay as the fallback for spatially strided views.
// Reference: nn.BatchNorm2d in training mode + the activation of Conv (nn/modules/conv.py:25-42) and their autograd.
#include "common.h"

#define BNF_SPLITS 512          // == RED_SPLITS of train.hip (the final kernels there sum this many partial rows)

// exact-GELU pieces with ONE hardware exp2 + one rcp: e = exp(-u^2/2) serves both the normal density and Abramowitz-Stegun 7.1.26
// erf(x) = 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1/(1 + p x), x = |u|/sqrt(2)  (|error| < 1.5e-7: below fp32 round-off of the sums that consume it;
// libm's erff + expf cost ~4x as many instructions and made the GELU passes of the ConvNeXt blocks VALU-bound)
__device__ __forceinline__ void bnf_gelu_parts(float u, float& cdf, float& pdf) {
  const float ax = fabsf(u) * 0.70710678118654752f;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);            // exp(-u^2 / 2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erfa = 1.f - poly * e;                                                  // erf(|u| / sqrt 2)
  cdf = 0.5f * (1.f + (u < 0.f ? -erfa : erfa));
  pdf = 0.3989422804014327f * e;
}
template <int ACT> __device__ __forceinline__ float bnf_act(float u) {
  if (ACT == MGDT_ACT_SILU) return u * fast_sigmoid(u);
  if (ACT == MGDT_ACT_RELU) return fmaxf(u, 0.f);
  if (ACT == MGDT_ACT_GELU) { float cdf, pdf; bnf_gelu_parts(u, cdf, pdf); return u * cdf; }
  return u;
}
template <int ACT> __device__ __forceinline__ float bnf_grad(float u) {
  if (ACT == MGDT_ACT_SILU) { const float s = fast_sigmoid(u); return s * (1.f + u * (1.f - s)); }
  if (ACT == MGDT_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (ACT == MGDT_ACT_GELU) { float cdf, pdf; bnf_gelu_parts(u, cdf, pdf); return cdf + u * pdf; }
  return 1.f;
}

struct BnfView { const char* p; long pix; };      // pixel-linear view: element (pixel, c) at p + (pixel * pix + c) * sizeof(T)

// V consecutive channels of one pixel as floats: 16 bytes per access for bf16 with V = 8 and for fp32 with V = 4
template <typename T, int V> __device__ __forceinline__ void ldv(const T* p, float (&o)[V]);
template <> __device__ __forceinline__ void ldv<float, 4>(const float* p, float (&o)[4]) { const f32x4 t = *(const f32x4*)p; o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
template <> __device__ __forceinline__ void ldv<bf16, 4>(const bf16* p, float (&o)[4]) { const bf16x4 t = *(const bf16x4*)p; o[0] = (float)t[0]; o[1] = (float)t[1]; o[2] = (float)t[2]; o[3] = (float)t[3]; }
template <> __device__ __forceinline__ void ldv<bf16, 8>(const bf16* p, float (&o)[8]) {
  const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)t[j];
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const float (&o)[V]);
template <> __device__ __forceinline__ void stv<float, 4>(float* p, const float (&o)[4]) { *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]}; }
template <> __device__ __forceinline__ void stv<bf16, 4>(bf16* p, const float (&o)[4]) {
  bf16x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = (bf16)o[j];
  *(bf16x4*)p = t;
}
template <> __device__ __forceinline__ void stv<bf16, 8>(bf16* p, const float (&o)[8]) {
  bf16x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = (bf16)o[j];
  *(bf16x8*)p = t;
}
template <int V> __device__ __forceinline__ void ldp(const float* p, float (&o)[V]) {      // V per-channel parameters (16-byte aligned: c0 % 4 == 0)
#pragma unroll
  for (int j = 0; j < V; j += 4) { const f32x4 t = *(const f32x4*)(p + j); o[j] = t[0]; o[j + 1] = t[1]; o[j + 2] = t[2]; o[j + 3] = t[3]; }
}

// per-channel partial sums of (f0, f1) over this workgroup's pixel range -> partial[split][c][2] (double); F: (pixel, c0) -> two float[V]
template <int V, typename F>
__device__ __forceinline__ void bnf_reduce(long npix, int C, double* partial, F f) {
  const int Q = C / V;
  const int PL = 256 / (Q < 256 ? Q : 256);                      // pixel lanes per pass (Q <= 256)
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int split = blockIdx.x;
  const long p0 = split * npix / gridDim.x, p1 = (split + 1) * npix / gridDim.x;     // gridDim.x = number of pixel splits (a multiple of 16, <= BNF_SPLITS)
  float s0[V], s1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (pl < PL) {
    long p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      float a[4][V], b[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) f(p + u * PL, q * V, a[u], b[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) { s0[j] += a[u][j]; s1[j] += b[u][j]; }
    }
    for (; p < p1; p += PL) {
      float a[V], b[V];
      f(p, q * V, a, b);
#pragma unroll
      for (int j = 0; j < V; ++j) { s0[j] += a[j]; s1[j] += b[j]; }
    }
  }
  __shared__ float red[2][V][256];
#pragma unroll
  for (int j = 0; j < V; ++j) { red[0][j][threadIdx.x] = s0[j]; red[1][j][threadIdx.x] = s1[j]; }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += 256) {                   // channel o = vector o / V, element o % V: sum its PL pixel lanes in order
    const int qq = o / V, j = o % V;
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < PL; ++k) { t0 += (double)red[0][j][k * Q + qq]; t1 += (double)red[1][j][k * Q + qq]; }
    partial[((long)split * C + o) * 2] = t0;
    partial[((long)split * C + o) * 2 + 1] = t1;
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bnf_stats_kernel(BnfView y, long npix, int C, double* partial) {
  bnf_reduce<V>(npix, C, partial, [&](long p, int c0, float (&a)[V], float (&b)[V]) {
    ldv<T, V>((const T*)y.p + p * y.pix + c0, a);
#pragma unroll
    for (int j = 0; j < V; ++j) b[j] = a[j] * a[j];
  });
}

// element-wise kernels: one channel vector per thread and iteration, 32-bit index math (host guarantees < 2^31 vectors)
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bnf_fwd_kernel(BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, BnfView r1, BnfView r2, BnfView z, uint32_t nvec, uint32_t Q) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += gridDim.x * 256u) {
    const uint32_t pq = i / Q;
    const long p = pq;
    const int c0 = (int)(i - pq * Q) * V;
    float v[V], o[V];
    ldv<T, V>((const T*)y.p + p * y.pix + c0, v);
    if (mean) {
      float m[V], rs[V], g[V], b[V];
      ldp<V>(mean + c0, m); ldp<V>(rstd + c0, rs); ldp<V>(gamma + c0, g); ldp<V>(beta + c0, b);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = bnf_act<ACT>(g[j] * ((v[j] - m[j]) * rs[j]) + b[j]);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = bnf_act<ACT>(v[j] + (beta ? beta[c0 + j] : 0.f));
    }
    if (r1.p) {
      float r[V];
      ldv<T, V>((const T*)r1.p + p * r1.pix + c0, r);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] += r[j];
    }
    if (r2.p) {
      float r[V];
      ldv<T, V>((const T*)r2.p + p * r2.pix + c0, r);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] += r[j];
    }
    stv<T, V>((T*)z.p + p * z.pix + c0, o);
  }
}

template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bnf_bwd_partial_kernel(BnfView gz, BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, long npix, int C, double* partial) {
  // this thread's V channels are fixed (bnf_reduce: vector = threadIdx.x % Q): their parameters are loaded once, not per pixel
  const int cq = (threadIdx.x % (C / V)) * V;
  float m[V], rs[V], ga[V], be[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    m[j] = mean ? mean[cq + j] : 0.f; rs[j] = mean ? rstd[cq + j] : 0.f;
    ga[j] = mean ? gamma[cq + j] : 0.f; be[j] = beta ? beta[cq + j] : 0.f;
  }
  const bool bn = mean != nullptr;
  bnf_reduce<V>(npix, C, partial, [&](long p, int c0, float (&a)[V], float (&b)[V]) {
    float v[V], gv[V];
    ldv<T, V>((const T*)y.p + p * y.pix + c0, v);
    ldv<T, V>((const T*)gz.p + p * gz.pix + c0, gv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float xh = bn ? (v[j] - m[j]) * rs[j] : v[j];
      const float u = bn ? ga[j] * xh + be[j] : v[j] + be[j];
      const float g = gv[j] * bnf_grad<ACT>(u);
      a[j] = g; b[j] = g * xh;
    }
  });
}

template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bnf_bwd_apply_kernel(BnfView gz, BnfView y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ coef,
                                                            BnfView dy, uint32_t nvec, uint32_t Q) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += gridDim.x * 256u) {
    const uint32_t pq = i / Q;
    const long p = pq;
    const int c0 = (int)(i - pq * Q) * V;
    float v[V], gv[V], o[V];
    ldv<T, V>((const T*)y.p + p * y.pix + c0, v);
    ldv<T, V>((const T*)gz.p + p * gz.pix + c0, gv);
    if (mean) {
      float m[V], rs[V], ga[V], be[V], k[2 * V];
      ldp<V>(mean + c0, m); ldp<V>(rstd + c0, rs); ldp<V>(gamma + c0, ga); ldp<V>(beta + c0, be);
      ldp<2 * V>(coef + 2 * c0, k);                         // (mean g, mean g*xhat) pairs
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (v[j] - m[j]) * rs[j];
        const float g = gv[j] * bnf_grad<ACT>(ga[j] * xh + be[j]);
        o[j] = ga[j] * rs[j] * (g - k[2 * j] - xh * k[2 * j + 1]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = gv[j] * bnf_grad<ACT>(v[j] + (beta ? beta[c0 + j] : 0.f));
    }
    stv<T, V>((T*)dy.p + p * dy.pix + c0, o);
  }
}

// ---- host: a view qualifies when its pixels are equally spaced (sh == W*sw, sn == H*sh), channels contiguous and V-aligned
static bool bnf_linear(const mgdt_view* v, int dtype, int V, BnfView* out) {
  if (!v || !v->p) { out->p = nullptr; out->pix = 0; return true; }
  if (v->sc != 1 || v->c % V || v->sw % V || v->sh != (long)v->w * v->sw || v->sn != (long)v->h * v->sh || (uintptr_t)v->p % (V * dtype_size(dtype))) return false;
  out->p = (const char*)v->p; out->pix = v->sw;
  return true;
}
// widest vector all views allow: 8 channels (16 bytes) for bf16, else 4
static int bnf_pick(int dtype, std::initializer_list<const mgdt_view*> views, BnfView* outs) {
  static const int vmax = getenv("MGDT_BN_V") ? atoi(getenv("MGDT_BN_V")) : 4;      // experiment knob
  for (int V : {dtype == MGDT_BF16 && vmax >= 8 ? 8 : 4, 4}) {
    bool ok = true;
    int k = 0;
    for (const mgdt_view* v : views) ok = bnf_linear(v, dtype, V, &outs[k++]) && ok;
    if (ok) return V;
  }
  return 0;
}
// pixel splits of a reduction: ~128 pixels each, at least 64, a multiple of 16 (the final kernels add 16 sub-sums), at most BNF_SPLITS.  Few splits on small
// maps keep the final kernel's dependent load chain short (it was the larger half of a 16 us statistics pass on a 20x20 map).
static inline int bnf_splits(long npix) { return (int)std::min<long>(BNF_SPLITS, std::max<long>(64, (npix / 128 + 15) / 16 * 16)); }
static inline int bnf_grid(long n) { return (int)std::min<long>((n + 255) / 256, 8192); }

#define BNF_ACT_DISPATCH(act, ...)                                       \
  switch (act) {                                                         \
    case MGDT_ACT_SILU: { constexpr int ACT = MGDT_ACT_SILU; __VA_ARGS__; } break; \
    case MGDT_ACT_RELU: { constexpr int ACT = MGDT_ACT_RELU; __VA_ARGS__; } break; \
    case MGDT_ACT_GELU: { constexpr int ACT = MGDT_ACT_GELU; __VA_ARGS__; } break; \
    default: { constexpr int ACT = MGDT_ACT_NONE; __VA_ARGS__; } break;  \
  }
// K is a kernel template name taking <T, V, ...>; fp32 always runs V = 4
#define BNF_TV(V, F32CALL, BF4CALL, BF8CALL) \
  if (dtype == MGDT_F32) { F32CALL; } else if (V == 8) { BF8CALL; } else { BF4CALL; }

// each returns false when the views do not qualify (the caller runs the scalar kernels)
int mgdt_bnf_stats(const mgdt_view* y, double* partial, int dtype, hipStream_t st) {
  BnfView v[1];
  const int V = bnf_pick(dtype, {y}, v);
  if (!V || y->c / V > 256) return 0;
  const long npix = (long)y->n * y->h * y->w;
  const int ns = bnf_splits(npix);
  BNF_TV(V, (bnf_stats_kernel<float, 4><<<ns, 256, 0, st>>>(v[0], npix, y->c, partial)),
         (bnf_stats_kernel<bf16, 4><<<ns, 256, 0, st>>>(v[0], npix, y->c, partial)),
         (bnf_stats_kernel<bf16, 8><<<ns, 256, 0, st>>>(v[0], npix, y->c, partial)));
  return ns;
}
bool mgdt_bnf_fwd(const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, const mgdt_view* r1,
                  const mgdt_view* r2, const mgdt_view* z, int dtype, hipStream_t st) {
  BnfView v[4];
  const int V = bnf_pick(dtype, {y, r1, r2, z}, v);
  if (!V) return false;
  const int Q = y->c / V;
  const long nq = (long)y->n * y->h * y->w * Q;
  if (nq >= 0x7fffffffL) return false;
  BNF_ACT_DISPATCH(act, {
    BNF_TV(V, (bnf_fwd_kernel<float, 4, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], mean, rstd, gamma, beta, v[1], v[2], v[3], (uint32_t)nq, (uint32_t)Q)),
           (bnf_fwd_kernel<bf16, 4, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], mean, rstd, gamma, beta, v[1], v[2], v[3], (uint32_t)nq, (uint32_t)Q)),
           (bnf_fwd_kernel<bf16, 8, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], mean, rstd, gamma, beta, v[1], v[2], v[3], (uint32_t)nq, (uint32_t)Q)));
  });
  return true;
}
int mgdt_bnf_bwd_partial(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                          double* partial, int dtype, hipStream_t st) {
  BnfView v[2];
  const int V = bnf_pick(dtype, {gz, y}, v);
  if (!V || y->c / V > 256) return 0;
  const long npix = (long)y->n * y->h * y->w;
  const int ns = bnf_splits(npix);
  BNF_ACT_DISPATCH(act, {
    BNF_TV(V, (bnf_bwd_partial_kernel<float, 4, ACT><<<ns, 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, npix, y->c, partial)),
           (bnf_bwd_partial_kernel<bf16, 4, ACT><<<ns, 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, npix, y->c, partial)),
           (bnf_bwd_partial_kernel<bf16, 8, ACT><<<ns, 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, npix, y->c, partial)));
  });
  return ns;
}
bool mgdt_bnf_bwd_apply(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                        const float* coef, const mgdt_view* dy, int dtype, hipStream_t st) {
  BnfView v[3];
  const int V = bnf_pick(dtype, {gz, y, dy}, v);
  if (!V) return false;
  const int Q = y->c / V;
  const long nq = (long)y->n * y->h * y->w * Q;
  if (nq >= 0x7fffffffL) return false;
  BNF_ACT_DISPATCH(act, {
    BNF_TV(V, (bnf_bwd_apply_kernel<float, 4, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, coef, v[2], (uint32_t)nq, (uint32_t)Q)),
           (bnf_bwd_apply_kernel<bf16, 4, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, coef, v[2], (uint32_t)nq, (uint32_t)Q)),
           (bnf_bwd_apply_kernel<bf16, 8, ACT><<<bnf_grid(nq), 256, 0, st>>>(v[0], v[1], mean, rstd, gamma, beta, coef, v[2], (uint32_t)nq, (uint32_t)Q)));
  });
  return true;
}
