// Direct (VALU) convolution: the 3-channel stem (reads the caller's NCHW fp32 image in place, writes NHWC),
// grouped / depthwise convs from the registry (DWConv) and any shape the MFMA kernel does not take.
// One thread = one output pixel x 16 output channels; weights [K][Cout] fp32 are wave-uniform loads.
#include <type_traits>

#include "common.h"

__global__ void pack_direct_kernel(const float* __restrict__ w, const float* cb, const float* g, const float* b,
                                   const float* mu, const float* var, float eps, int cin_g, int cout, int k,
                                   float* __restrict__ wout, float* __restrict__ bias_out) {
  // wout[(tap*cin_g + ci)*cout + co] = w[co][ci][ky][kx] * scale[co]
  long total = (long)k * k * cin_g * cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int co = (int)(i % cout);
    long t = i / cout;
    int ci = (int)(t % cin_g);
    int tap = (int)(t / cin_g);
    float s = g ? g[co] / sqrtf(eps + var[co]) : 1.f;
    wout[i] = w[(((long)co * cin_g + ci) * k + tap / k) * k + tap % k] * s;
  }
  int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co < cout) {
    float bo = 0.f;
    if (g) {
      bo = b[co] - g[co] * mu[co] / sqrtf(var[co] + eps);
      if (cb) bo += g[co] / sqrtf(eps + var[co]) * cb[co];
    } else if (cb) {
      bo = cb[co];
    }
    bias_out[co] = bo;
  }
}

extern "C" int mgdt_conv_pack_direct(const float* w, const float* cb, const float* g, const float* b, const float* mu,
                                     const float* var, float eps, int cin_g, int cout, int k, float* wout,
                                     float* bias_out, mgdt_stream s) {
  if (!w || !wout || !bias_out) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_direct: null pointer");
  long total = (long)k * k * cin_g * cout;
  int grid = (int)std::max<long>(std::min<long>((total + 255) / 256, 4096), cdiv(cout, 256));
  pack_direct_kernel<<<grid, 256, 0, (hipStream_t)s>>>(w, cb, g, b, mu, var, eps, cin_g, cout, k, wout, bias_out);
  MGDT_CHECK_LAUNCH("conv_pack_direct");
  return MGDT_OK;
}

template <typename TX, typename TY, int COB>
__global__ __launch_bounds__(256) void conv_direct_kernel(const TX* __restrict__ x, long xsn, long xsh, long xsw, long xsc,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          TY* __restrict__ y, long ysn, long ysh, long ysw, long ysc, int N,
                                                          int H, int W, int Cin, int Ho, int Wo, int Cout, int KS, int stride,
                                                          int groups, int act) {
  long m = blockIdx.x * (long)blockDim.x + threadIdx.x;
  long M = (long)N * Ho * Wo;
  if (m >= M) return;
  const int co0 = blockIdx.y * COB;
  const int cin_g = Cin / groups, cout_g = Cout / groups;
  int n = (int)(m / ((long)Ho * Wo));
  int rem = (int)(m - (long)n * Ho * Wo);
  int oy = rem / Wo, ox = rem - oy * Wo;
  const int pad = KS / 2;
  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;
  if (groups == 1) {
    for (int ky = 0; ky < KS; ++ky) {
      int iy = oy * stride - pad + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        int ix = ox * stride - pad + kx;
        if (ix < 0 || ix >= W) continue;
        const TX* xp = x + n * xsn + iy * xsh + ix * xsw;
        const float* wp = w + (long)((ky * KS + kx) * Cin) * Cout + co0;
        for (int ci = 0; ci < Cin; ++ci) {
          float xv = (float)xp[ci * xsc];
#pragma unroll
          for (int j = 0; j < COB; ++j)
            if (co0 + j < Cout) acc[j] = fmaf(xv, wp[(long)ci * Cout + j], acc[j]);
        }
      }
    }
  } else {
    for (int j = 0; j < COB; ++j) {
      int co = co0 + j;
      if (co >= Cout) break;
      int gi = co / cout_g;
      for (int ky = 0; ky < KS; ++ky) {
        int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < KS; ++kx) {
          int ix = ox * stride - pad + kx;
          if (ix < 0 || ix >= W) continue;
          const TX* xp = x + n * xsn + iy * xsh + ix * xsw + (long)gi * cin_g * xsc;
          const float* wp = w + (long)((ky * KS + kx) * cin_g) * Cout + co;
          for (int ci = 0; ci < cin_g; ++ci) acc[j] = fmaf((float)xp[ci * xsc], wp[(long)ci * Cout], acc[j]);
        }
      }
    }
  }
  TY* yp = y + n * ysn + oy * ysh + ox * ysw;
#pragma unroll
  for (int j = 0; j < COB; ++j)
    if (co0 + j < Cout) yp[(co0 + j) * ysc] = (TY)act_apply(acc[j] + bias[co0 + j], act);
}

// Stem specialisation (cin <= 4, k = 3): the caller's NCHW fp32 image is read in place, weights sit in LDS (broadcast reads),
// each thread produces 16 consecutive NHWC output channels of one pixel and stores them as 16-byte vectors.
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void conv_stem_kernel(const TX* __restrict__ x, long xsn, long xsh, long xsw, long xsc,
                                                        const float* __restrict__ w, const float* __restrict__ bias, TY* __restrict__ y,
                                                        long ysn, long ysh, long ysw, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                                                        int stride, int act) {
  __shared__ __attribute__((aligned(16))) float wl[9 * 4 * 16 + 16];
  __shared__ float lut[std::is_same<TX, uint8_t>::value ? 256 : 1];
  if (std::is_same<TX, uint8_t>::value)   // uint8 image: pixel / 255 with the fp32 division of the reference's preprocess (predictor.py:129,
    lut[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.f);              // val.py:34, train.py:64), kept in fp32 for the multiply-adds
  const int co0 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < 9 * Cin * 16; i += 256) wl[i] = w[(long)(i / 16) * Cout + co0 + (i % 16)];
  if (threadIdx.x < 16) wl[9 * 4 * 16 + threadIdx.x] = bias[co0 + threadIdx.x];
  __syncthreads();
  long m = blockIdx.x * 256L + threadIdx.x;
  long M = (long)N * Ho * Wo;
  if (m >= M) return;
  int n = (int)(m / ((long)Ho * Wo));
  int rem = (int)(m - (long)n * Ho * Wo);
  int oy = rem / Wo, ox = rem - oy * Wo;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = wl[9 * 4 * 16 + j];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int iy = oy * stride - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int ix = ox * stride - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const TX* xp = x + n * xsn + iy * xsh + ix * xsw;
      for (int ci = 0; ci < Cin; ++ci) {
        float xv;
        if constexpr (std::is_same<TX, uint8_t>::value) xv = lut[xp[ci * xsc]];
        else xv = (float)xp[ci * xsc];
        const f32x4* wv = (const f32x4*)(wl + ((ky * 3 + kx) * Cin + ci) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t = wv[q];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q * 4 + j] = fmaf(xv, t[j], acc[q * 4 + j]);
        }
      }
    }
  }
  TY* yp = y + n * ysn + oy * ysh + ox * ysw + co0;
  if constexpr (sizeof(TY) == 2) {      // 16 bf16 = two 16-byte stores (four 8-byte ones touch every cache line of the wave four times)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)act_apply(acc[h * 8 + j], act);
      *(bf16x8*)(yp + h * 8) = o;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = act_apply(acc[q * 4 + j], act);
      store4<TY>(yp + q * 4, v);
    }
  }
}

extern "C" int mgdt_conv2d_direct_fwd(const mgdt_view* x, int x_dtype, const float* w, const float* bias, int k, int stride,
                                      int groups, int act, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !w || !bias) MGDT_FAIL(MGDT_BAD_ARG, "conv2d_direct: null/empty argument");
  if (k < 1 || k > 7 || !(k & 1) || stride < 1 || groups < 1 || x->c % groups || y->c % groups)
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d_direct: k=%d stride=%d groups=%d cin=%d cout=%d", k, stride, groups, x->c, y->c);
  const int pad = k / 2;
  const int Ho = (x->h + 2 * pad - k) / stride + 1, Wo = (x->w + 2 * pad - k) / stride + 1;
  if (y->n != x->n || y->h != Ho || y->w != Wo) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d_direct: y is %dx%dx%d, expected %dx%dx%d", y->n, y->h, y->w, x->n, Ho, Wo);
  long M = (long)x->n * Ho * Wo;
  hipStream_t st0 = (hipStream_t)s;
  if (k == 3 && groups == 1 && x->c <= 4 && y->c % 16 == 0 && y->sc == 1 && y->sw % 4 == 0 && y->sh % 4 == 0 && y->sn % 4 == 0 &&
      (uintptr_t)y->p % 16 == 0 && (x_dtype == MGDT_F32 || x_dtype == MGDT_U8 || (x_dtype == MGDT_BF16 && dtype == MGDT_BF16))) {   // stem fast path
    dim3 sg(cdiv(M, 256), y->c / 16);
#define STEM(TX, TY) conv_stem_kernel<TX, TY><<<sg, 256, 0, st0>>>((const TX*)x->p, x->sn, x->sh, x->sw, x->sc, w, bias, (TY*)y->p, y->sn, y->sh, y->sw, x->n, \
                                                              x->h, x->w, x->c, Ho, Wo, y->c, stride, act)
    if (x_dtype == MGDT_U8) { if (dtype == MGDT_BF16) STEM(uint8_t, bf16); else STEM(uint8_t, float); }
    else if (x_dtype == MGDT_BF16) STEM(bf16, bf16);
    else if (dtype == MGDT_BF16) STEM(float, bf16);
    else STEM(float, float);
#undef STEM
    MGDT_CHECK_LAUNCH("conv2d_direct_fwd(stem)");
    return MGDT_OK;
  }
  if (x_dtype == MGDT_U8) MGDT_FAIL(MGDT_BAD_DTYPE, "conv2d_direct: uint8 input is taken by the stem path only (k=3, cin<=4, cout%%16==0)");
  constexpr int COB = 16;
  dim3 grid(cdiv(M, 256), cdiv(y->c, COB));
  hipStream_t st = (hipStream_t)s;
#define LAUNCH(TX, TY)                                                                                              \
  conv_direct_kernel<TX, TY, COB><<<grid, 256, 0, st>>>((const TX*)x->p, x->sn, x->sh, x->sw, x->sc, w, bias, (TY*)y->p, \
                                                        y->sn, y->sh, y->sw, y->sc, x->n, x->h, x->w, x->c, Ho, Wo, y->c, k, \
                                                        stride, groups, act)
  if (x_dtype == MGDT_F32 && dtype == MGDT_F32) LAUNCH(float, float);
  else if (x_dtype == MGDT_F32 && dtype == MGDT_BF16) LAUNCH(float, bf16);
  else if (x_dtype == MGDT_BF16 && dtype == MGDT_BF16) LAUNCH(bf16, bf16);
  else if (x_dtype == MGDT_BF16 && dtype == MGDT_F32) LAUNCH(bf16, float);
  else MGDT_FAIL(MGDT_BAD_DTYPE, "conv2d_direct: dtypes %d -> %d", x_dtype, dtype);
#undef LAUNCH
  MGDT_CHECK_LAUNCH("conv2d_direct_fwd");
  return MGDT_OK;
}
