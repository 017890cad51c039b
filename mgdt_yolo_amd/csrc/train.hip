// Training-side kernels of the detection path: BatchNorm2d in training mode (batch statistics, running-stat update,
// fused activation and residual adds), its backward, convolution dgrad / wgrad / bias-grad, and the adjoints of the pooling /
// resampling ops.  First-cut versions: correct, deterministic (fixed-order reductions, no float atomics; max-pool
// routing), channel-vectorised; the MFMA treatment the forward convolution got is the next step for dgrad(stride 2) / wgrad.
// Reference semantics: nn.BatchNorm2d(eps 1e-3, momentum 0.03) as set by initialize_weights (yolo/utils/torch_utils.py:254-256):
// normalise with the biased batch variance, update running_var with the unbiased one.
#include <stdlib.h>

#include "common.h"

#define RED_SPLITS 512

// channel-vectorised forms (train_vec.hip); false -> run the scalar kernel
bool mgdt_wgrad_bf16_launch(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* partial, int nsplit, hipStream_t st);
bool mgdt_v4_add(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int dtype, hipStream_t st);
bool mgdt_v4_maxpool5_bwd(const mgdt_view* x, const mgdt_view* gy, float* gx_f32, int dtype, hipStream_t st);
// vectorised versions for pixel-linear views (bn_fast.hip); each returns false when a view does not qualify
int mgdt_bnf_stats(const mgdt_view* y, double* partial, int dtype, hipStream_t st);      // returns the number of pixel splits written (0: not handled)
bool mgdt_bnf_fwd(const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, const mgdt_view* r1,
                  const mgdt_view* r2, const mgdt_view* z, int dtype, hipStream_t st);
int mgdt_bnf_bwd_partial(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                          double* partial, int dtype, hipStream_t st);
bool mgdt_bnf_bwd_apply(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                        const float* coef, const mgdt_view* dy, int dtype, hipStream_t st);
// channel lanes of a reduction block: the whole channel range when it is narrow, so that all 256 threads have pixels to walk
static __host__ __device__ inline int red_cw(int c) { return c > 32 ? 64 : c > 16 ? 32 : c > 8 ? 16 : c > 4 ? 8 : 4; }

__device__ __forceinline__ float act_fwd(float u, int act) {
  switch (act) {
    case MGDT_ACT_SILU: return u * fast_sigmoid(u);     // hardware exp2 / rcp (~1 ulp each), as the inference epilogues
    case MGDT_ACT_RELU: return fmaxf(u, 0.f);
    case MGDT_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
    default: return u;
  }
}
__device__ __forceinline__ float act_grad(float u, int act) {
  switch (act) {
    case MGDT_ACT_SILU: { float s = fast_sigmoid(u); return s * (1.f + u * (1.f - s)); }
    case MGDT_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case MGDT_ACT_GELU: return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
    default: return 1.f;
  }
}

// ------------------------------------------------------------------------------------------------ per-channel reductions
// partial[split][c][2] (double) = sum over the split's pixels of (f0, f1); pixel ranges are fixed -> deterministic.
template <typename T, typename F>
__device__ __forceinline__ void channel_reduce(const mgdt_view v, double* partial, F f) {
  // grid: (cdiv(C, CW), RED_SPLITS); block 256 = CW channel lanes x 256/CW pixel lanes; four pixels per thread are read before they are
  // accumulated (in order) so that a thread keeps several requests in flight
  const int CW = red_cw(v.c), PL = 256 / CW;
  const int cl = threadIdx.x % CW, pl = threadIdx.x / CW;
  const int c = blockIdx.x * CW + cl, split = blockIdx.y;
  const long npix = (long)v.n * v.h * v.w;
  const int p0 = (int)(split * npix / RED_SPLITS), p1 = (int)((split + 1) * npix / RED_SPLITS);
  double s0 = 0.0, s1 = 0.0;
  if (c < v.c) {
    const int HW = v.h * v.w;
    auto at = [&](int p, float& a0, float& a1) __attribute__((always_inline)) {
      const int n = p / HW, rem = p - n * HW;
      const int yy = rem / v.w, xx = rem - yy * v.w;
      f((long)n * v.sn + (long)yy * v.sh + (long)xx * v.sw + c, (long)n, (long)yy, (long)xx, c, a0, a1);
    };
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      float a0[4], a1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) at(p + u * PL, a0[u], a1[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += a0[u]; s1 += a1[u]; }
    }
    for (; p < p1; p += PL) { float a0, a1; at(p, a0, a1); s0 += a0; s1 += a1; }
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x < CW && c < v.c) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < PL; ++k) { t0 += red[0][k * CW + threadIdx.x]; t1 += red[1][k * CW + threadIdx.x]; }
    partial[((long)split * v.c + c) * 2] = t0;
    partial[((long)split * v.c + c) * 2 + 1] = t1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const mgdt_view y, double* partial) {
  const T* p = (const T*)y.p;
  channel_reduce<T>(y, partial, [&](long off, long, long, long, int, float& a0, float& a1) { float v = (float)p[off]; a0 = v; a1 = v * v; });
}

// sum of the RED_SPLITS partial rows of 16 channels per workgroup: 16 threads per channel each add 32 rows (loads independent, in flight
// together), thread 0 of the channel adds the 16 sub-sums in order - a fixed association, so the result does not depend on scheduling.
// (One thread per channel walking all 512 rows was a ~100 us dependent chain per BN layer.)
#define FIN_CW 16
__device__ __forceinline__ bool final_sum(const double* __restrict__ partial, int C, int nsplit, int& c, double& s0, double& s1) {
  __shared__ double red[2][16][FIN_CW];
  const int cl = threadIdx.x % FIN_CW, part = threadIdx.x / FIN_CW;
  c = blockIdx.x * FIN_CW + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    const int ROWS = nsplit / 16;                          // nsplit: a multiple of 16
#pragma unroll 8
    for (int k = 0; k < ROWS; ++k) {
      const double2 v = *(const double2*)(partial + ((long)(part * ROWS + k) * C + c) * 2);
      a0 += v.x; a1 += v.y;
    }
  }
  red[0][part][cl] = a0; red[1][part][cl] = a1;
  __syncthreads();
  if (part != 0 || c >= C) return false;
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { s0 += red[0][k][cl]; s1 += red[1][k][cl]; }
  return true;
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* partial, int C, int nsplit, double count, float eps, float momentum, float* mean, float* rstd,
                                      float* running_mean, float* running_var) {
  int c;
  double s, ss;
  if (!final_sum(partial, C, nsplit, c, s, ss)) return;
  double m = s / count, var = ss / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

extern "C" size_t mgdt_reduce_workspace_bytes(int c) { return ((size_t)RED_SPLITS * c * 2 + (size_t)c * 2) * sizeof(double) + (size_t)c * 2 * sizeof(float); }   // partials + per-channel sums + their means

extern "C" int mgdt_bn_stats_fwd(const mgdt_view* y, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                 float* running_var, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(y) || !mean || !rstd || !ws) MGDT_FAIL(MGDT_BAD_ARG, "bn_stats: null/empty argument");
  if (y->sc != 1) MGDT_FAIL(MGDT_BAD_SHAPE, "bn_stats: NHWC view required");
  int ns = mgdt_bnf_stats(y, (double*)ws, dtype, (hipStream_t)s);
  if (!ns) {
    dim3 grid(cdiv(y->c, red_cw(y->c)), RED_SPLITS);
    MGDT_DISPATCH_DTYPE(dtype, (bn_stats_partial_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>(*y, (double*)ws)));
    ns = RED_SPLITS;
  }
  bn_stats_final_kernel<<<cdiv(y->c, FIN_CW), 256, 0, (hipStream_t)s>>>((const double*)ws, y->c, ns, (double)y->n * y->h * y->w, eps, momentum, mean, rstd,
                                                                     running_mean, running_var);
  MGDT_CHECK_LAUNCH("bn_stats_fwd");
  return MGDT_OK;
}

// z = act(gamma * (y - mean) * rstd + beta) [+ r1] [+ r2]   (gamma/beta/mean/rstd may be NULL -> plain act(y + bias?))
template <typename T>
__global__ void bn_act_fwd_kernel(const mgdt_view y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                  const mgdt_view r1, const mgdt_view r2, const mgdt_view z) {
  long total = (long)y.n * y.h * y.w * y.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % y.c);
    long t = i / y.c;
    int w = (int)(t % y.w);
    t /= y.w;
    int h = (int)(t % y.h);
    long n = t / y.h;
    float v = (float)((const T*)y.p)[n * y.sn + h * y.sh + w * y.sw + c];
    float u = mean ? gamma[c] * ((v - mean[c]) * rstd[c]) + beta[c] : v + (beta ? beta[c] : 0.f);
    float o = act_fwd(u, act);
    if (r1.p) o += (float)((const T*)r1.p)[n * r1.sn + h * r1.sh + w * r1.sw + c];
    if (r2.p) o += (float)((const T*)r2.p)[n * r2.sn + h * r2.sh + w * r2.sw + c];
    ((T*)z.p)[n * z.sn + h * z.sh + w * z.sw + c] = (T)o;
  }
}

static mgdt_view null_view() { mgdt_view v; memset(&v, 0, sizeof(v)); return v; }
static inline int ew_grid(long total) { return (int)std::min<long>((total + 255) / 256, 16384); }

extern "C" int mgdt_bn_act_fwd(const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                               const mgdt_view* r1, const mgdt_view* r2, const mgdt_view* z, int dtype, mgdt_stream s) {
  if (!view_ok(y) || !view_ok(z)) MGDT_FAIL(MGDT_BAD_ARG, "bn_act: null/empty view");
  if (y->sc != 1 || z->sc != 1 || y->n != z->n || y->h != z->h || y->w != z->w || y->c != z->c) MGDT_FAIL(MGDT_BAD_SHAPE, "bn_act: matching NHWC views required");
  if ((mean != nullptr) != (rstd != nullptr) || (mean && (!gamma || !beta))) MGDT_FAIL(MGDT_BAD_ARG, "bn_act: mean/rstd/gamma/beta must come together");
  long total = (long)y->n * y->h * y->w * y->c;
  mgdt_view a = (r1 && r1->p) ? *r1 : null_view(), b = (r2 && r2->p) ? *r2 : null_view();
  if (!mgdt_bnf_fwd(y, mean, rstd, gamma, beta, act, r1, r2, z, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (bn_act_fwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*y, mean, rstd, gamma, beta, act, a, b, *z)));
  MGDT_CHECK_LAUNCH("bn_act_fwd");
  return MGDT_OK;
}

// backward of z = act(u), u = gamma*xhat + beta, xhat = (y-mean)*rstd:
//   g = gz * act'(u); dbeta = sum g; dgamma = sum g*xhat; dy = gamma*rstd*(g - dbeta/m - xhat*dgamma/m)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const mgdt_view gz, const mgdt_view y, const float* mean, const float* rstd,
                                                             const float* gamma, const float* beta, int act, double* partial) {
  const T* gp = (const T*)gz.p;
  const T* yp = (const T*)y.p;
  channel_reduce<T>(gz, partial, [&](long off, long n, long yy, long xx, int c, float& a0, float& a1) {
    float v = (float)yp[n * y.sn + yy * y.sh + xx * y.sw + c];
    float xh = mean ? (v - mean[c]) * rstd[c] : v;
    float u = mean ? gamma[c] * xh + beta[c] : v + (beta ? beta[c] : 0.f);
    float g = (float)gp[off] * act_grad(u, act);
    a0 = g;
    a1 = g * xh;
  });
}

// per-channel totals of the two BN-backward sums (+ the parameter gradients), once, instead of RED_SPLITS loads per element
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* partial, int C, int nsplit, double count, double* sums, float* coef, float* dgamma, float* dbeta) {
  int c;
  double sg, sgx;
  if (!final_sum(partial, C, nsplit, c, sg, sgx)) return;
  sums[2 * c] = sg; sums[2 * c + 1] = sgx;
  coef[2 * c] = (float)(sg / count); coef[2 * c + 1] = (float)(sgx / count);     // what the apply pass subtracts, divided once
  if (dbeta) dbeta[c] = (float)sg;
  if (dgamma) dgamma[c] = (float)sgx;
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const mgdt_view gz, const mgdt_view y, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, int act, const double* sums, double count, const mgdt_view dy) {
  long total = (long)y.n * y.h * y.w * y.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % y.c);
    long t = i / y.c;
    int w = (int)(t % y.w);
    t /= y.w;
    int h = (int)(t % y.h);
    long n = t / y.h;
    const double sg = sums[2 * c], sgx = sums[2 * c + 1];
    float v = (float)((const T*)y.p)[n * y.sn + h * y.sh + w * y.sw + c];
    float gzv = (float)((const T*)gz.p)[n * gz.sn + h * gz.sh + w * gz.sw + c];
    float o;
    if (mean) {
      float xh = (v - mean[c]) * rstd[c];
      float g = gzv * act_grad(gamma[c] * xh + beta[c], act);
      o = gamma[c] * rstd[c] * (g - (float)(sg / count) - xh * (float)(sgx / count));
    } else {
      o = gzv * act_grad(v + (beta ? beta[c] : 0.f), act);
    }
    ((T*)dy.p)[n * dy.sn + h * dy.sh + w * dy.sw + c] = (T)o;
  }
}

extern "C" int mgdt_bn_act_bwd(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, int act, float* dgamma, float* dbeta, const mgdt_view* dy, void* ws, int dtype,
                               mgdt_stream s) {
  if (!view_ok(gz) || !view_ok(y) || !view_ok(dy) || !ws) MGDT_FAIL(MGDT_BAD_ARG, "bn_act_bwd: null/empty argument");
  if (gz->sc != 1 || y->sc != 1 || dy->sc != 1 || gz->c != y->c || dy->c != y->c || gz->n != y->n || gz->h != y->h || gz->w != y->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "bn_act_bwd: matching NHWC views required");
  dim3 grid(cdiv(y->c, red_cw(y->c)), RED_SPLITS);
  long total = (long)y->n * y->h * y->w * y->c;
  double* sums = (double*)ws + (size_t)RED_SPLITS * y->c * 2;
  int ns = mgdt_bnf_bwd_partial(gz, y, mean, rstd, gamma, beta, act, (double*)ws, dtype, (hipStream_t)s);
  if (!ns) {
    MGDT_DISPATCH_DTYPE(dtype, (bn_bwd_partial_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>(*gz, *y, mean, rstd, gamma, beta, act, (double*)ws)));
    ns = RED_SPLITS;
  }
  float* coef = (float*)(sums + (size_t)y->c * 2);
  bn_bwd_final_kernel<<<cdiv(y->c, FIN_CW), 256, 0, (hipStream_t)s>>>((const double*)ws, y->c, ns, (double)y->n * y->h * y->w, sums, coef, dgamma, dbeta);
  if (!mgdt_bnf_bwd_apply(gz, y, mean, rstd, gamma, beta, act, coef, dy, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (bn_bwd_apply_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*gz, *y, mean, rstd, gamma, beta, act, sums,
                                                                                                 (double)y->n * y->h * y->w, *dy)));
  MGDT_CHECK_LAUNCH("bn_act_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ conv dgrad (direct, any stride)
// dx[n,iy,ix,ci] (+)= sum_{ky,kx,co} dy[n,(iy+pad-ky)/s,(ix+pad-kx)/s,co] * w[co][ci][ky][kx]   (only where divisible by s)
template <typename T>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const mgdt_view dy, const float* __restrict__ w, int KS, int stride, int Cin_tot,
                                                         const mgdt_view dx, int accumulate) {
  // thread = (input pixel, 4 input channels)
  const int Q = dx.c / 4;
  long total = (long)dx.n * dx.h * dx.w * Q;
  const int pad = KS / 2, Cout = dy.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int ix = (int)(t % dx.w);
    t /= dx.w;
    int iy = (int)(t % dx.h);
    long n = t / dx.h;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < KS; ++ky) {
      int ty = iy + pad - ky;
      if (ty < 0 || ty % stride) continue;
      int oy = ty / stride;
      if (oy >= dy.h) continue;
      for (int kx = 0; kx < KS; ++kx) {
        int tx = ix + pad - kx;
        if (tx < 0 || tx % stride) continue;
        int ox = tx / stride;
        if (ox >= dy.w) continue;
        const T* gp = (const T*)dy.p + n * dy.sn + oy * dy.sh + ox * dy.sw;
        for (int co = 0; co < Cout; ++co) {
          float g = (float)gp[co];
          const float* wp = w + (((long)co * Cin_tot + q * 4) * KS + ky) * KS + kx;
          const long cs = (long)KS * KS;
          acc[0] = fmaf(g, wp[0], acc[0]);
          acc[1] = fmaf(g, wp[cs], acc[1]);
          acc[2] = fmaf(g, wp[2 * cs], acc[2]);
          acc[3] = fmaf(g, wp[3 * cs], acc[3]);
        }
      }
    }
    T* op = (T*)dx.p + n * dx.sn + iy * dx.sh + ix * dx.sw + q * 4;
    if (accumulate) acc += load4<T>(op);
    store4<T>(op, acc);
  }
}

extern "C" int mgdt_conv_dgrad(const mgdt_view* dy, const float* w_oihw, int k, int stride, const mgdt_view* dx, int accumulate, int dtype,
                               mgdt_stream s) {
  if (!view_ok(dy) || !view_ok(dx) || !w_oihw) MGDT_FAIL(MGDT_BAD_ARG, "conv_dgrad: null/empty argument");
  if (dy->sc != 1 || dx->sc != 1 || dx->c % 4 || dx->sw % 4 || dx->sh % 4 || dx->sn % 4 || dy->n != dx->n) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_dgrad: NHWC views, cin%%4==0");
  const int pad = k / 2;
  if ((dx->h + 2 * pad - k) / stride + 1 != dy->h || (dx->w + 2 * pad - k) / stride + 1 != dy->w) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_dgrad: dy/dx geometry mismatch");
  long total = (long)dx->n * dx->h * dx->w * (dx->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (conv_dgrad_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*dy, w_oihw, k, stride, dx->c, *dx, accumulate)));
  MGDT_CHECK_LAUNCH("conv_dgrad");
  return MGDT_OK;
}

bool mgdt_wgrad_mfma_launch(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* partial, int nsplit, int dtype,
                            hipStream_t st);   // wgrad_mfma.hip

// ------------------------------------------------------------------------------------------------ conv wgrad (+ bias grad)
// dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n,oy,ox,co] * (x [+ x2])[n, oy*s-pad+ky, ox*s-pad+kx, ci]
// block = one tap x 16 couts x 16 cins x one pixel split; 256 threads = 16 (co quad, ci quad) pairs x 16 pixel lanes.
#define WG_SPLITS 16
// pixel splits of the NHWC path: more of them for small weight tensors (few (co, ci, tap) blocks) so that the grid still fills the chip;
// the partial buffer stays <= max(16 splits, 16 MiB)
int mgdt_wgrad_bf16_groups(int cin, int cout, int k);   // wgrad_bf16.hip
static inline int wgrad_splits(int cin, int cout, int k) {
  const long nel = (long)cin * cout * k * k;
  static const long cap = getenv("MGDT_WGRAD_SPLITS") ? atol(getenv("MGDT_WGRAD_SPLITS")) : 512;     // experiment knob
  static const long budget = getenv("MGDT_WGRAD_PARTIAL_ELEMS") ? atol(getenv("MGDT_WGRAD_PARTIAL_ELEMS")) : (4L << 20);     // experiment knob
  long ns = std::max<long>(WG_SPLITS, std::min<long>(cap, budget / std::max<long>(nel, 1)));
  // big channel tiles leave few (cout, cin) groups: up to twice the splits so that groups x splits reaches ~2 workgroups per CU (the 64->96 3x3 head
  // convolution ran 216 workgroups of 4 waves on 256 CUs)
  static const long fill = getenv("MGDT_WGRAD_FILL") ? atol(getenv("MGDT_WGRAD_FILL")) : 448;     // experiment knob, 0 = off
  const int g = fill ? mgdt_wgrad_bf16_groups(cin, cout, k) : 0;
  if (g > 0 && g * ns < fill) ns = std::min<long>({cap, 2 * ns, (fill + g - 1) / g});
  return (int)ns;
}
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_partial_kernel(const mgdt_view x, const mgdt_view x2, const mgdt_view dy, int KS, int stride,
                                                                 float* __restrict__ partial, int nsplit) {
  const int tap = blockIdx.z % (KS * KS), split = blockIdx.z / (KS * KS);
  const int ky = tap / KS, kx = tap % KS, pad = KS / 2;
  const int co0 = blockIdx.x * 16, ci0 = blockIdx.y * 16;
  const int pair = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int coq = co0 + (pair >> 2) * 4, ciq = ci0 + (pair & 3) * 4;
  const int M = dy.n * dy.h * dy.w, HW = dy.h * dy.w;
  const int p0 = (int)((long)split * M / nsplit), p1 = (int)((long)(split + 1) * M / nsplit);
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  const bool live = coq < dy.c && ciq < x.c;
  if (live) {
    // four pixels per round: their loads are issued together (zero weight for taps outside the image instead of a branch), then the
    // 16 FMAs per pixel in pixel order
    auto fetch = [&](int p, f32x4& g, f32x4& v) __attribute__((always_inline)) {
      const bool inr = p < p1;
      const int pp = inr ? p : p0;
      const int n = pp / HW, rem = pp - n * HW;
      const int oy = rem / dy.w, ox = rem - oy * dy.w;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      const bool ok = inr && (unsigned)iy < (unsigned)x.h && (unsigned)ix < (unsigned)x.w;
      const int iyc = ok ? iy : 0, ixc = ok ? ix : 0;
      g = load4<T>((const T*)dy.p + (long)n * dy.sn + (long)oy * dy.sh + (long)ox * dy.sw + coq);
      v = load4<T>((const T*)x.p + (long)n * x.sn + (long)iyc * x.sh + (long)ixc * x.sw + ciq);
      if (x2.p) v += load4<T>((const T*)x2.p + (long)n * x2.sn + (long)iyc * x2.sh + (long)ixc * x2.sw + ciq);
      if (!ok) g = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    for (int p = p0 + pl; p < p1; p += 64) {
      f32x4 g[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) fetch(p + u * 16, g[u], v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(g[u][a], v[u][b], acc[a][b]);
    }
  }
  __shared__ float red[16][16][17];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) red[pl][pair][a * 4 + b] = acc[a][b];
  __syncthreads();
  // thread t sums element (pair = t / 16, e = t % 16) over the 16 pixel lanes
  {
    int pr = threadIdx.x >> 4, e = threadIdx.x & 15;
    float sum = 0.f;
    for (int k = 0; k < 16; ++k) sum += red[k][pr][e];
    int co = co0 + (pr >> 2) * 4 + (e >> 2), ci = ci0 + (pr & 3) * 4 + (e & 3);
    if (co < dy.c && ci < x.c) partial[(((long)split * dy.c + co) * x.c + ci) * KS * KS + tap] = sum;
  }
}

// 64 weight elements per workgroup, 4 threads per element: thread `part` adds rows part, part+4, ... (independent loads, 8 in flight), then the
// four sub-sums are added in order.  Fixed association; a single thread walking up to 256 rows was a dependent-latency chain of ~50 us.
__global__ __launch_bounds__(256) void wgrad_final_kernel(const float* __restrict__ partial, long n, float* dw, int accumulate, int nsplit) {
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, part = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + el;
  float a = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int k = part; k < nsplit; k += 4) a += partial[(long)k * n + i];
  }
  red[part][el] = a;
  __syncthreads();
  if (part == 0 && i < n) {
    const float t = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const mgdt_view dy, double* partial) {
  const T* p = (const T*)dy.p;
  channel_reduce<T>(dy, partial, [&](long off, long, long, long, int, float& a0, float& a1) { a0 = (float)p[off]; a1 = 0.f; });
}
__global__ __launch_bounds__(256) void bias_grad_final_kernel(const double* partial, int C, float* db, int accumulate) {
  int c;
  double s, unused;
  if (!final_sum(partial, C, RED_SPLITS, c, s, unused)) return;
  db[c] = accumulate ? db[c] + (float)s : (float)s;
}

// generic-stride variant (the 3-channel stem reads the NCHW image): one block per weight element, fixed-order tree reduction
template <typename TX, typename T>
__global__ __launch_bounds__(256) void conv_wgrad_generic_kernel(const mgdt_view x, const mgdt_view dy, int KS, int stride, float* __restrict__ partial,
                                                                 long nel) {
  const int e = blockIdx.x, split = blockIdx.y;   // e = ((co * Cin + ci) * KS + ky) * KS + kx; WG_SPLITS pixel ranges per element
  const int kx = e % KS, ky = (e / KS) % KS, ci = (e / (KS * KS)) % x.c, co = e / (KS * KS * x.c);
  const int pad = KS / 2;
  const long M = (long)dy.n * dy.h * dy.w, HW = (long)dy.h * dy.w;
  double acc = 0.0;
  const long q0 = split * M / WG_SPLITS, q1 = (split + 1) * M / WG_SPLITS;
  for (long p = q0 + threadIdx.x; p < q1; p += 256) {
    long n = p / HW, rem = p - n * HW;
    int oy = (int)(rem / dy.w), ox = (int)(rem - (long)oy * dy.w);
    int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if ((unsigned)iy >= (unsigned)x.h || (unsigned)ix >= (unsigned)x.w) continue;
    acc += (double)((float)((const T*)dy.p)[n * dy.sn + oy * dy.sh + ox * dy.sw + co * dy.sc] *
                    (float)((const TX*)x.p)[n * x.sn + iy * x.sh + ix * x.sw + ci * x.sc]);
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long)split * nel + e] = (float)red[0];
}

extern "C" size_t mgdt_conv_wgrad_workspace_bytes(int cin, int cout, int k) {
  return std::max((size_t)wgrad_splits(cin, cout, k) * cin * cout * k * k * sizeof(float), mgdt_reduce_workspace_bytes(cout));
}

extern "C" int mgdt_conv_wgrad(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* dw_oihw, float* dbias,
                               int accumulate, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(dy) || !ws) MGDT_FAIL(MGDT_BAD_ARG, "conv_wgrad: null/empty argument");
  // dw_oihw == NULL: partial sums only (ws keeps [mgdt_conv_wgrad_splits][cout][cin][k*k] fp32); mgdt_wgrad_final_batch adds them up later for many
  // convolutions in one launch.  Not for the generic (non-NHWC / channels % 4) path and not together with dbias (which reuses ws).
  const bool partial_only = dw_oihw == nullptr;
  if (partial_only && (dbias || x->sc != 1 || x->c % 4 || dy->c % 4)) MGDT_FAIL(MGDT_BAD_ARG, "conv_wgrad: partial-only mode needs NHWC views with channels %% 4 == 0 and no dbias");
  hipStream_t st = (hipStream_t)s;
  if (dy->sc != 1 || x->n != dy->n) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_wgrad: dy must be an NHWC view with the batch of x");
  if (x->sc != 1 || x->c % 4 || dy->c % 4) {   // generic path: fp32 input of any layout (the image), no fused x2
    if (x2 && x2->p) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_wgrad: x2 needs the NHWC path");
    int nel = dy->c * x->c * k * k;
    if (dtype == MGDT_F32) conv_wgrad_generic_kernel<float, float><<<dim3(nel, WG_SPLITS), 256, 0, st>>>(*x, *dy, k, stride, (float*)ws, nel);
    else conv_wgrad_generic_kernel<float, bf16><<<dim3(nel, WG_SPLITS), 256, 0, st>>>(*x, *dy, k, stride, (float*)ws, nel);
    wgrad_final_kernel<<<cdiv(nel, 64), 256, 0, st>>>((const float*)ws, nel, dw_oihw, accumulate, WG_SPLITS);
    if (dbias) {
      dim3 g2(cdiv(dy->c, red_cw(dy->c)), RED_SPLITS);
      MGDT_DISPATCH_DTYPE(dtype, (bias_grad_partial_kernel<T><<<g2, 256, 0, st>>>(*dy, (double*)ws)));
      bias_grad_final_kernel<<<cdiv(dy->c, FIN_CW), 256, 0, st>>>((const double*)ws, dy->c, dbias, accumulate);
    }
    MGDT_CHECK_LAUNCH("conv_wgrad(generic)");
    return MGDT_OK;
  }
  mgdt_view b = (x2 && x2->p) ? *x2 : null_view();
  const int nsplit = wgrad_splits(x->c, dy->c, k);
  if ((long)dy->n * dy->h * dy->w >= 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_wgrad: too many pixels");
  static const bool no_mfma = getenv("MGDT_WGRAD_VALU") != nullptr;     // experiment knob: the VALU outer-product kernel
  static const bool no_bf16 = getenv("MGDT_WGRAD_F32MFMA") != nullptr;   // experiment knob: fp32 MFMA for bf16 tensors too
  if (!no_mfma && !no_bf16 && dtype == MGDT_BF16 && mgdt_wgrad_bf16_launch(x, x2, dy, k, stride, (float*)ws, nsplit, st)) {
  } else if (no_mfma || !mgdt_wgrad_mfma_launch(x, x2, dy, k, stride, (float*)ws, nsplit, dtype, st)) {
    dim3 grid(cdiv(dy->c, 16), cdiv(x->c, 16), k * k * nsplit);
    MGDT_DISPATCH_DTYPE(dtype, (conv_wgrad_partial_kernel<T><<<grid, 256, 0, st>>>(*x, b, *dy, k, stride, (float*)ws, nsplit)));
  }
  long n = (long)dy->c * x->c * k * k;
  if (!partial_only) wgrad_final_kernel<<<cdiv(n, 64), 256, 0, st>>>((const float*)ws, n, dw_oihw, accumulate, nsplit);
  if (dbias) {
    dim3 g2(cdiv(dy->c, red_cw(dy->c)), RED_SPLITS);
    MGDT_DISPATCH_DTYPE(dtype, (bias_grad_partial_kernel<T><<<g2, 256, 0, st>>>(*dy, (double*)ws)));
    bias_grad_final_kernel<<<cdiv(dy->c, FIN_CW), 256, 0, st>>>((const double*)ws, dy->c, dbias, accumulate);
  }
  MGDT_CHECK_LAUNCH("conv_wgrad");
  return MGDT_OK;
}

extern "C" int mgdt_conv_wgrad_splits(int cin, int cout, int k) { return wgrad_splits(cin, cout, k); }

// The final sums of many weight gradients in one launch (a training step has ~60 of them, each a ~7 us launch of its own otherwise): job j adds the
// nsplit partial rows of its convolution in the fixed order of wgrad_final_kernel (4 interleaved sub-sums, then their sum).
struct WgFinalJob { const float* partial; float* dw; long n; int nsplit, accumulate; };
#define WGF_BATCH 32
struct WgFinalJobs { WgFinalJob j[WGF_BATCH]; };
__global__ __launch_bounds__(256) void wgrad_final_batch_kernel(const WgFinalJobs jobs) {
  const WgFinalJob J = jobs.j[blockIdx.y];
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, part = threadIdx.x >> 6;
  for (long base = blockIdx.x * 64L; base < J.n; base += gridDim.x * 64L) {
    const long i = base + el;
    float a = 0.f;
    if (i < J.n) {
#pragma unroll 8
      for (int k = part; k < J.nsplit; k += 4) a += J.partial[(long)k * J.n + i];
    }
    __syncthreads();
    red[part][el] = a;
    __syncthreads();
    if (part == 0 && i < J.n) {
      const float t = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
      J.dw[i] = J.accumulate ? J.dw[i] + t : t;
    }
  }
}
extern "C" int mgdt_wgrad_final_batch(const mgdt_wgrad_final_desc* d, int n, mgdt_stream s) {
  if (!d || n < 0) MGDT_FAIL(MGDT_BAD_ARG, "wgrad_final_batch: null descriptor array");
  hipStream_t st = (hipStream_t)s;
  for (int i0 = 0; i0 < n; i0 += WGF_BATCH) {
    WgFinalJobs jobs;
    const int m = std::min(WGF_BATCH, n - i0);
    for (int q = 0; q < WGF_BATCH; ++q) {
      const mgdt_wgrad_final_desc& e = d[i0 + (q < m ? q : 0)];
      if (!e.partial || !e.dw || e.n <= 0 || e.nsplit < 1) MGDT_FAIL(MGDT_BAD_ARG, "wgrad_final_batch: descriptor %d", i0 + q);
      jobs.j[q] = WgFinalJob{e.partial, e.dw, q < m ? e.n : 0, e.nsplit, e.accumulate};
    }
    long nmax = 0;
    for (int q = 0; q < m; ++q) nmax = std::max(nmax, jobs.j[q].n);
    wgrad_final_batch_kernel<<<dim3((unsigned)std::min<long>(cdiv(nmax, 64), 1024), m), 256, 0, st>>>(jobs);       // workgroups past a job's end leave at once
  }
  MGDT_CHECK_LAUNCH("wgrad_final_batch");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise add / adjoints
template <typename T>
__global__ void add_kernel(const mgdt_view a, const mgdt_view b, const mgdt_view o) {
  long total = (long)o.n * o.h * o.w * o.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % o.c);
    long t = i / o.c;
    int w = (int)(t % o.w);
    t /= o.w;
    int h = (int)(t % o.h);
    long n = t / o.h;
    float v = (float)((const T*)a.p)[n * a.sn + h * a.sh + w * a.sw + c * a.sc] + (float)((const T*)b.p)[n * b.sn + h * b.sh + w * b.sw + c * b.sc];
    ((T*)o.p)[n * o.sn + h * o.sh + w * o.sw + c * o.sc] = (T)v;
  }
}

extern "C" int mgdt_add_fwd(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int dtype, mgdt_stream s) {
  if (!view_ok(a) || !view_ok(b) || !view_ok(o)) MGDT_FAIL(MGDT_BAD_ARG, "add: null/empty view");
  if (a->n != o->n || a->h != o->h || a->w != o->w || a->c != o->c || b->n != o->n || b->h != o->h || b->w != o->w || b->c != o->c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "add: shape mismatch");
  long total = (long)o->n * o->h * o->w * o->c;
  if (!mgdt_v4_add(a, b, o, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (add_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*a, *b, *o)));
  MGDT_CHECK_LAUNCH("add_fwd");
  return MGDT_OK;
}

// MaxPool2d(5,1,2) backward as a GATHER (deterministic, no float atomics): gx[i] = sum of gy[o] over the outputs o whose window's
// argmax is i; argmax = ATen's scan: first maximum in (ky, kx) order, a NaN replaces whatever was found before it.  Every window that
// contains i is re-scanned exactly as the forward scanned it (25 x 25 cached loads per element; the P5 map is tiny), and the
// contributions are added in (oy, ox) order, so two runs give the same bits.
template <typename T>
__global__ void maxpool5_bwd_kernel(const mgdt_view x, const mgdt_view gy, float* gx_f32) {
  long total = (long)x.n * x.h * x.w * x.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % x.c);
    long t = i / x.c;
    int w = (int)(t % x.w);
    t /= x.w;
    int h = (int)(t % x.h);
    long n = t / x.h;
    const T* xb = (const T*)x.p + n * x.sn + c;
    const T* gb = (const T*)gy.p + n * gy.sn + c;
    const float mine = (float)xb[h * x.sh + w * x.sw];
    float acc = 0.f;
    for (int oy = h - 2; oy <= h + 2; ++oy) {
      if ((unsigned)oy >= (unsigned)x.h) continue;
      for (int ox = w - 2; ox <= w + 2; ++ox) {
        if ((unsigned)ox >= (unsigned)x.w) continue;
        float best = -INFINITY;
        int by = oy, bx = ox;
        for (int dy = -2; dy <= 2; ++dy) {
          int yy = oy + dy;
          if ((unsigned)yy >= (unsigned)x.h) continue;
          for (int dx = -2; dx <= 2; ++dx) {
            int xx = ox + dx;
            if ((unsigned)xx >= (unsigned)x.w) continue;
            float v = (yy == h && xx == w) ? mine : (float)xb[yy * x.sh + xx * x.sw];
            if (v > best || isnan(v)) { best = v; by = yy; bx = xx; }
          }
        }
        if (by == h && bx == w) acc += (float)gb[oy * gy.sh + ox * gy.sw];
      }
    }
    gx_f32[((n * x.h + h) * x.w + w) * x.c + c] = acc;
  }
}

// gx_f32: dense fp32 [n][h][w][c], every element written (no need to zero it).
extern "C" int mgdt_maxpool5_bwd(const mgdt_view* x, const mgdt_view* gy, float* gx_f32, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(gy) || !gx_f32) MGDT_FAIL(MGDT_BAD_ARG, "maxpool5_bwd: null/empty argument");
  if (x->sc != 1 || gy->sc != 1 || x->n != gy->n || x->h != gy->h || x->w != gy->w || x->c != gy->c) MGDT_FAIL(MGDT_BAD_SHAPE, "maxpool5_bwd: matching NHWC views");
  long total = (long)x->n * x->h * x->w * x->c;
  if (!mgdt_v4_maxpool5_bwd(x, gy, gx_f32, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (maxpool5_bwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*x, *gy, gx_f32)));
  MGDT_CHECK_LAUNCH("maxpool5_bwd");
  return MGDT_OK;
}

// nearest x(scale) up-sampling adjoint: gx[iy,ix] = sum of the gy pixels that read it
template <typename T>
__global__ void nearest_bwd_kernel(const mgdt_view gy, const mgdt_view gx) {
  long total = (long)gx.n * gx.h * gx.w * gx.c;
  const float sy = (float)gx.h / (float)gy.h, sx = (float)gx.w / (float)gy.w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % gx.c);
    long t = i / gx.c;
    int w = (int)(t % gx.w);
    t /= gx.w;
    int h = (int)(t % gx.h);
    long n = t / gx.h;
    // candidate outputs: a small bracket around [h/scale, (h+1)/scale), each verified with the forward's index rule
    const int oy_lo = max(0, (int)floorf((float)h / sy) - 1), oy_hi = min(gy.h - 1, (int)ceilf((float)(h + 1) / sy) + 1);
    const int ox_lo = max(0, (int)floorf((float)w / sx) - 1), ox_hi = min(gy.w - 1, (int)ceilf((float)(w + 1) / sx) + 1);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      if (min((int)floorf((float)oy * sy), gx.h - 1) != h) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        if (min((int)floorf((float)ox * sx), gx.w - 1) != w) continue;
        acc += (float)((const T*)gy.p)[n * gy.sn + oy * gy.sh + ox * gy.sw + c];
      }
    }
    ((T*)gx.p)[n * gx.sn + h * gx.sh + w * gx.sw + c] = (T)acc;
  }
}

extern "C" int mgdt_nearest_bwd(const mgdt_view* gy, const mgdt_view* gx, int dtype, mgdt_stream s) {
  if (!view_ok(gy) || !view_ok(gx)) MGDT_FAIL(MGDT_BAD_ARG, "nearest_bwd: null/empty view");
  if (gy->sc != 1 || gx->sc != 1 || gy->n != gx->n || gy->c != gx->c) MGDT_FAIL(MGDT_BAD_SHAPE, "nearest_bwd: NHWC views, same n/c");
  long total = (long)gx->n * gx->h * gx->w * gx->c;
  MGDT_DISPATCH_DTYPE(dtype, (nearest_bwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*gy, *gx)));
  MGDT_CHECK_LAUNCH("nearest_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ grouped / depth-wise convolution (DWConv, conv.py:82-86)
// Not on any target YAML: plain VALU kernels, one thread per element, so that training a model that holds a DWConv is served instead of refused.
// w: [cout][cin/groups][k][k] fp32 (nn.Conv2d's layout for groups > 1).
#define GC_SPLITS 64
template <typename T>
__global__ __launch_bounds__(256) void gconv_dgrad_kernel(const mgdt_view dy, const float* __restrict__ w, int KS, int stride, int groups, const mgdt_view dx,
                                                          int accumulate) {
  const int cig = dx.c / groups, cog = dy.c / groups, pad = KS / 2;
  const long total = (long)dx.n * dx.h * dx.w * dx.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % dx.c);
    long t = i / dx.c;
    const int ix = (int)(t % dx.w);
    t /= dx.w;
    const int iy = (int)(t % dx.h);
    const long n = t / dx.h;
    const int g = ci / cig, cl = ci - g * cig;
    float acc = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
      const int ty = iy + pad - ky;
      if (ty < 0 || ty % stride || ty / stride >= dy.h) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int tx = ix + pad - kx;
        if (tx < 0 || tx % stride || tx / stride >= dy.w) continue;
        const T* gp = (const T*)dy.p + n * dy.sn + (long)(ty / stride) * dy.sh + (long)(tx / stride) * dy.sw + g * cog;
        for (int co = 0; co < cog; ++co) acc = fmaf(ldf<T>(gp + co), w[(((long)(g * cog + co) * cig + cl) * KS + ky) * KS + kx], acc);
      }
    }
    T* op = (T*)dx.p + n * dx.sn + (long)iy * dx.sh + (long)ix * dx.sw + ci;
    if (accumulate) acc += ldf<T>(op);
    stf<T>(op, acc);
  }
}

// thread = one weight element x one pixel split; partial[split][cout * cin/groups * k * k]
template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_partial_kernel(const mgdt_view x, const mgdt_view dy, int KS, int stride, int groups, float* __restrict__ partial,
                                                                  int nel) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nel) return;
  const int cig = x.c / groups, cog = dy.c / groups, pad = KS / 2;
  const int kx = e % KS, ky = (e / KS) % KS, cl = (e / (KS * KS)) % cig, co = e / (KS * KS * cig);
  const int ci = (co / cog) * cig + cl;
  const long npix = (long)dy.n * dy.h * dy.w, per = (npix + GC_SPLITS - 1) / GC_SPLITS;
  const long p0 = blockIdx.y * per, p1 = p0 + per < npix ? p0 + per : npix;
  float acc = 0.f;
  for (long p = p0; p < p1; ++p) {
    const int ox = (int)(p % dy.w);
    const long t = p / dy.w;
    const int oy = (int)(t % dy.h);
    const long n = t / dy.h;
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if (iy < 0 || iy >= x.h || ix < 0 || ix >= x.w) continue;
    acc = fmaf(ldf<T>((const T*)dy.p + n * dy.sn + (long)oy * dy.sh + (long)ox * dy.sw + co),
               ldf<T>((const T*)x.p + n * x.sn + (long)iy * x.sh + (long)ix * x.sw + ci), acc);
  }
  partial[(long)blockIdx.y * nel + e] = acc;
}

extern "C" size_t mgdt_gconv_wgrad_workspace_bytes(int cin, int cout, int k, int groups) {
  return groups > 0 ? (size_t)GC_SPLITS * cout * (cin / groups) * k * k * sizeof(float) : 0;
}

extern "C" int mgdt_gconv_dgrad(const mgdt_view* dy, const float* w, int k, int stride, int groups, const mgdt_view* dx, int accumulate, int dtype, mgdt_stream s) {
  if (!view_ok(dy) || !view_ok(dx) || !w) MGDT_FAIL(MGDT_BAD_ARG, "gconv_dgrad: null/empty argument");
  if (groups < 1 || dx->c % groups || dy->c % groups || dy->sc != 1 || dx->sc != 1 || dy->n != dx->n) MGDT_FAIL(MGDT_BAD_SHAPE, "gconv_dgrad: NHWC views, channels %% groups == 0");
  const int pad = k / 2;
  if ((dx->h + 2 * pad - k) / stride + 1 != dy->h || (dx->w + 2 * pad - k) / stride + 1 != dy->w) MGDT_FAIL(MGDT_BAD_SHAPE, "gconv_dgrad: dy/dx geometry mismatch");
  const long total = (long)dx->n * dx->h * dx->w * dx->c;
  MGDT_DISPATCH_DTYPE(dtype, (gconv_dgrad_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*dy, w, k, stride, groups, *dx, accumulate)));
  MGDT_CHECK_LAUNCH("gconv_dgrad");
  return MGDT_OK;
}

extern "C" int mgdt_gconv_wgrad(const mgdt_view* x, const mgdt_view* dy, int k, int stride, int groups, float* dw, int accumulate, void* ws, int dtype,
                                mgdt_stream s) {
  if (!view_ok(x) || !view_ok(dy) || !dw || !ws) MGDT_FAIL(MGDT_BAD_ARG, "gconv_wgrad: null/empty argument");
  if (groups < 1 || x->c % groups || dy->c % groups || dy->sc != 1 || x->sc != 1 || dy->n != x->n) MGDT_FAIL(MGDT_BAD_SHAPE, "gconv_wgrad: NHWC views, channels %% groups == 0");
  const int pad = k / 2;
  if ((x->h + 2 * pad - k) / stride + 1 != dy->h || (x->w + 2 * pad - k) / stride + 1 != dy->w) MGDT_FAIL(MGDT_BAD_SHAPE, "gconv_wgrad: x/dy geometry mismatch");
  const int nel = dy->c * (x->c / groups) * k * k;
  hipStream_t st = (hipStream_t)s;
  MGDT_DISPATCH_DTYPE(dtype, (gconv_wgrad_partial_kernel<T><<<dim3(cdiv(nel, 256), GC_SPLITS), 256, 0, st>>>(*x, *dy, k, stride, groups, (float*)ws, nel)));
  wgrad_final_kernel<<<cdiv(nel, 64), 256, 0, st>>>((const float*)ws, nel, dw, accumulate, GC_SPLITS);
  MGDT_CHECK_LAUNCH("gconv_wgrad");
  return MGDT_OK;
}
