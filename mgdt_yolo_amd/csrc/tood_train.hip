// Training side of the TOODHead (reference nn/modules/head.py:466-572, block.py:401-432, mmcv ModulatedDeformConv2d): the adjoints that the
// conv / reduction kernels of train.hip do not already cover.
//   GroupNorm(16) + act backward   : GN is a per-(image, channel) affine u = y*A + B of the conv output; the backward needs the sums
//                                    S1 = sum_hw g_u, S2 = sum_hw g_u*y (mgdt_nc_reduce) and then dy = g_u*P + y*Q + R with per-(image, channel)
//                                    coefficients (mgdt_gn_affine, mgdt_nc_affine_act_bwd, mgdt_gn_bwd_coef, mgdt_nc_axpby)
//   layer attention backward       : the tiny GAP -> 1x1 -> ReLU -> 1x1 -> sigmoid chain per image (mgdt_tood_layer_attn_bwd)
//   probability gate backward      : d(x * sigmoid(l)) (mgdt_pixel_gate_bwd)
//   DCNv2 training                 : sampled-and-modulated columns materialised once (mgdt_dcn_im2col) so that the weight / column gradients are
//                                    ordinary 1x1 conv wgrad / dgrad; the column gradient is scattered back to the input and reduced to the
//                                    offset / mask-logit gradients by mgdt_dcn_col2im_bwd (mmcv modulated_deformable_col2im(_coord)).
// Parity: unpinned like the forward (mmcv is absent); tests compare with torch.autograd of oracle/tood.py.
#include "common.h"

#define TT_AT(T, v, n, h, w, c) ((T*)(v).p + ((long)(n) * (v).sn + (long)(h) * (v).sh + (long)(w) * (v).sw + (c)))
static inline int tt_grid(long n) { return (int)std::min<long>((n + 255) / 256, 16384); }
static bool tt_v4(const mgdt_view* v, int dtype) {
  return v && v->p && v->sc == 1 && v->c % 4 == 0 && v->sw % 4 == 0 && v->sh % 4 == 0 && v->sn % 4 == 0 && (uintptr_t)v->p % (4 * dtype_size(dtype)) == 0;
}
static bool tt_same(const mgdt_view* a, const mgdt_view* b) { return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c; }

__device__ __forceinline__ float tt_act_grad(float u, int act) {
  if (act == MGDT_ACT_SILU) { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }
  if (act == MGDT_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  return 1.f;
}

// ------------------------------------------------------------------------------------------------ GroupNorm as a per-(image, channel) affine
// Sy = sum_hw y, Syy = sum_hw y^2 (fp32 [n][c]).  mean / rstd per (image, group) (biased variance E[y^2] - mean^2, the forward kernel's form);
// A = gamma * rstd, B = beta - mean * rstd * gamma.
__global__ __launch_bounds__(256) void gn_affine_kernel(const float* __restrict__ Sy, const float* __restrict__ Syy, int C, int hw, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int G, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                        float* __restrict__ A, float* __restrict__ B) {
  __shared__ float sm[2][256];
  const int n = blockIdx.x, cpg = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    double s = 0.0, ss = 0.0;
    for (int j = 0; j < cpg; ++j) { s += Sy[(long)n * C + g * cpg + j]; ss += Syy[(long)n * C + g * cpg + j]; }
    const double cnt = (double)cpg * hw, m = s / cnt;
    double var = ss / cnt - m * m;
    if (var < 0.0) var = 0.0;
    sm[0][g] = (float)m; sm[1][g] = (float)(1.0 / sqrt(var + (double)eps));
    mean[(long)n * G + g] = sm[0][g]; rstd[(long)n * G + g] = sm[1][g];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float m = sm[0][c / cpg], r = sm[1][c / cpg];
    A[(long)n * C + c] = gamma[c] * r;
    B[(long)n * C + c] = beta[c] - m * r * gamma[c];
  }
}
extern "C" int mgdt_gn_affine(const float* Sy, const float* Syy, int n, int c, int hw, const float* gamma, const float* beta, int groups, float eps,
                              float* mean_ng, float* rstd_ng, float* A, float* B, mgdt_stream s) {
  if (!Sy || !Syy || !gamma || !beta || !mean_ng || !rstd_ng || !A || !B) MGDT_FAIL(MGDT_BAD_ARG, "gn_affine: null pointer");
  if (n < 1 || c < 1 || hw < 1 || groups < 1 || groups > 256 || c % groups) MGDT_FAIL(MGDT_BAD_SHAPE, "gn_affine: c=%d groups=%d", c, groups);
  gn_affine_kernel<<<n, 256, 0, (hipStream_t)s>>>(Sy, Syy, c, hw, gamma, beta, groups, eps, mean_ng, rstd_ng, A, B);
  MGDT_CHECK_LAUNCH("gn_affine");
  return MGDT_OK;
}

// gu = g * act'(y * A[n][c] + B[n][c])   (A, B NULL: u = y - with act = ReLU and y the layer's OUTPUT this is the ReLU mask)
template <typename T>
__global__ __launch_bounds__(256) void nc_affine_act_bwd_kernel(const mgdt_view g, const mgdt_view y, const float* __restrict__ A, const float* __restrict__ B, int act,
                                                                const mgdt_view gu) {
  const int Q = y.c >> 2;
  const long total = (long)y.n * y.h * y.w * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    long t = i / Q;
    const int w = (int)(t % y.w);
    t /= y.w;
    const int h = (int)(t % y.h);
    const long n = t / y.h;
    const f32x4 gv = load4<T>(TT_AT(const T, g, n, h, w, c)), yv = load4<T>(TT_AT(const T, y, n, h, w, c));
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = A ? yv[j] * A[n * y.c + c + j] + B[n * y.c + c + j] : yv[j];
      o[j] = gv[j] * tt_act_grad(u, act);
    }
    store4<T>(TT_AT(T, gu, n, h, w, c), o);
  }
}
extern "C" int mgdt_nc_affine_act_bwd(const mgdt_view* g, const mgdt_view* y, const float* A, const float* B, int act, const mgdt_view* gu, int dtype, mgdt_stream s) {
  if (!view_ok(g) || !view_ok(y) || !view_ok(gu) || (A != nullptr) != (B != nullptr)) MGDT_FAIL(MGDT_BAD_ARG, "nc_affine_act_bwd: null/empty argument");
  if (!tt_v4(g, dtype) || !tt_v4(y, dtype) || !tt_v4(gu, dtype) || !tt_same(g, y) || !tt_same(gu, y)) MGDT_FAIL(MGDT_BAD_SHAPE, "nc_affine_act_bwd: matching 4-aligned NHWC views");
  const long nq = (long)y->n * y->h * y->w * (y->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (nc_affine_act_bwd_kernel<T><<<tt_grid(nq), 256, 0, (hipStream_t)s>>>(*g, *y, A, B, act, *gu)));
  MGDT_CHECK_LAUNCH("nc_affine_act_bwd");
  return MGDT_OK;
}

// S1 = sum_hw g_u, S2 = sum_hw g_u * y.  With xhat = (y - mean) * rstd:  sum_hw g_u*xhat = rstd * (S2 - mean*S1).
//   dgamma[c] = sum_n rstd*(S2 - mean*S1),  dbeta[c] = sum_n S1
//   M1[n,g] = mean over the group of g_u*gamma,  M2[n,g] = mean over the group of g_u*gamma*xhat
//   dy = g_u * P + y * Q + R,  P = gamma*rstd,  Q = -rstd^2 * M2,  R = -rstd*M1 + mean*rstd^2*M2
__global__ __launch_bounds__(256) void gn_bwd_coef_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma, int C, int hw, int G,
                                                          float* __restrict__ P, float* __restrict__ Q, float* __restrict__ R, float* __restrict__ pimg) {
  __shared__ float sm[2][256];
  const int n = blockIdx.x, cpg = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    const float m = mean[(long)n * G + g], r = rstd[(long)n * G + g];
    double a = 0.0, b = 0.0;
    for (int j = 0; j < cpg; ++j) {
      const int c = g * cpg + j;
      const float s1 = S1[(long)n * C + c], s2 = S2[(long)n * C + c];
      a += (double)(gamma[c] * s1);
      b += (double)(gamma[c] * r * (s2 - m * s1));
    }
    const double cnt = (double)cpg * hw;
    sm[0][g] = (float)(a / cnt); sm[1][g] = (float)(b / cnt);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float m = mean[(long)n * G + g], r = rstd[(long)n * G + g], M1 = sm[0][g], M2 = sm[1][g];
    const float s1 = S1[(long)n * C + c], s2 = S2[(long)n * C + c];
    P[(long)n * C + c] = gamma[c] * r;
    Q[(long)n * C + c] = -r * r * M2;
    R[(long)n * C + c] = -r * M1 + m * r * r * M2;
    pimg[((long)n * 2) * C + c] = r * (s2 - m * s1);
    pimg[((long)n * 2 + 1) * C + c] = s1;
  }
}
// out[i] (+)= sum_n pimg[n * stride + i], i < count
__global__ void tt_sum_rows_kernel(const float* __restrict__ pimg, int N, long stride, int count, float* __restrict__ out, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += pimg[(long)n * stride + i];
  out[i] = accumulate ? out[i] + s : s;
}
extern "C" size_t mgdt_gn_bwd_workspace_bytes(int n, int c) { return (size_t)n * 2 * c * sizeof(float); }
extern "C" int mgdt_gn_bwd_coef(const float* S1, const float* S2, const float* mean_ng, const float* rstd_ng, const float* gamma, int n, int c, int hw, int groups,
                                float* P, float* Q, float* R, float* dgamma, float* dbeta, int accumulate, void* ws, mgdt_stream s) {
  if (!S1 || !S2 || !mean_ng || !rstd_ng || !gamma || !P || !Q || !R || !dgamma || !dbeta || !ws) MGDT_FAIL(MGDT_BAD_ARG, "gn_bwd_coef: null pointer");
  if (n < 1 || c < 1 || hw < 1 || groups < 1 || groups > 256 || c % groups) MGDT_FAIL(MGDT_BAD_SHAPE, "gn_bwd_coef: c=%d groups=%d", c, groups);
  hipStream_t st = (hipStream_t)s;
  gn_bwd_coef_kernel<<<n, 256, 0, st>>>(S1, S2, mean_ng, rstd_ng, gamma, c, hw, groups, P, Q, R, (float*)ws);
  // pimg rows are [n][2][c]: sum over n of the two halves; the halves are contiguous per image, so treat each image's row as 2c values
  tt_sum_rows_kernel<<<cdiv(c, 256), 256, 0, st>>>((const float*)ws, n, 2L * c, c, dgamma, accumulate);          // first c entries of every row
  tt_sum_rows_kernel<<<cdiv(c, 256), 256, 0, st>>>((const float*)ws + c, n, 2L * c, c, dbeta, accumulate);       // second c entries
  MGDT_CHECK_LAUNCH("gn_bwd_coef");
  return MGDT_OK;
}

// out = a * sa[n][c] (+ b * sb[n][c]) (+ shift[n][c]);  sa NULL: 1
template <typename T>
__global__ __launch_bounds__(256) void nc_axpby_kernel(const mgdt_view a, const float* __restrict__ sa, const mgdt_view b, const float* __restrict__ sb,
                                                       const float* __restrict__ shift, const mgdt_view o) {
  const int Q = o.c >> 2;
  const long total = (long)o.n * o.h * o.w * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    long t = i / Q;
    const int w = (int)(t % o.w);
    t /= o.w;
    const int h = (int)(t % o.h);
    const long n = t / o.h;
    f32x4 v = load4<T>(TT_AT(const T, a, n, h, w, c));
    if (sa) v = v * *(const f32x4*)(sa + n * o.c + c);
    if (b.p) v += load4<T>(TT_AT(const T, b, n, h, w, c)) * *(const f32x4*)(sb + n * o.c + c);
    if (shift) v += *(const f32x4*)(shift + n * o.c + c);
    store4<T>(TT_AT(T, o, n, h, w, c), v);
  }
}
extern "C" int mgdt_nc_axpby(const mgdt_view* a, const float* sa, const mgdt_view* b, const float* sb, const float* shift, const mgdt_view* out, int dtype,
                             mgdt_stream s) {
  if (!view_ok(a) || !view_ok(out)) MGDT_FAIL(MGDT_BAD_ARG, "nc_axpby: null/empty view");
  const bool hb = b && b->p;
  if (hb && !sb) MGDT_FAIL(MGDT_BAD_ARG, "nc_axpby: b needs sb");
  if (!tt_v4(a, dtype) || !tt_v4(out, dtype) || !tt_same(a, out) || (hb && (!tt_v4(b, dtype) || !tt_same(b, out)))) MGDT_FAIL(MGDT_BAD_SHAPE, "nc_axpby: matching 4-aligned NHWC views");
  mgdt_view bb;
  memset(&bb, 0, sizeof(bb));
  if (hb) bb = *b;
  const long nq = (long)out->n * out->h * out->w * (out->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (nc_axpby_kernel<T><<<tt_grid(nq), 256, 0, (hipStream_t)s>>>(*a, sa, bb, sb, shift, *out)));
  MGDT_CHECK_LAUNCH("nc_axpby");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ probability gate backward
// forward (mgdt_pixel_gate_fwd): out = x * sigmoid(l), l one logit per pixel.  gx = g * s;  gl = s*(1-s) * sum_c g*x.   16 lanes per pixel.
template <typename T>
__global__ __launch_bounds__(256) void pixel_gate_bwd_kernel(const mgdt_view g, const mgdt_view x, const mgdt_view l, const mgdt_view gx, const mgdt_view gl) {
  const int Q = x.c >> 2, lane = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const long M = (long)x.n * x.h * x.w, HW = (long)x.h * x.w;
  for (long m0 = blockIdx.x * 16L; m0 < M; m0 += gridDim.x * 16L) {         // uniform trip count: the shuffles need every lane
    const long m = m0 + pl;
    const bool ok = m < M;
    const long mm = ok ? m : 0;
    const long n = mm / HW, rem = mm - n * HW;
    const int h = (int)(rem / x.w), w = (int)(rem - (long)h * x.w);
    const float lv = ldf<T>(TT_AT(const T, l, n, h, w, 0));
    const float sg = 1.f / (1.f + expf(-lv));
    float dot = 0.f;
    for (int q = lane; q < Q; q += 16) {
      if (!ok) continue;
      const f32x4 gv = load4<T>(TT_AT(const T, g, n, h, w, 4 * q)), xv = load4<T>(TT_AT(const T, x, n, h, w, 4 * q));
      store4<T>(TT_AT(T, gx, n, h, w, 4 * q), gv * sg);
      dot += (gv[0] * xv[0] + gv[1] * xv[1]) + (gv[2] * xv[2] + gv[3] * xv[3]);
    }
    for (int o = 1; o < 16; o <<= 1) dot += __shfl_xor(dot, o);
    if (ok && lane == 0) stf<T>(TT_AT(T, gl, n, h, w, 0), dot * sg * (1.f - sg));
  }
}
extern "C" int mgdt_pixel_gate_bwd(const mgdt_view* g, const mgdt_view* x, const mgdt_view* logit, const mgdt_view* gx, const mgdt_view* glogit, int dtype, mgdt_stream s) {
  if (!view_ok(g) || !view_ok(x) || !view_ok(logit) || !view_ok(gx) || !view_ok(glogit)) MGDT_FAIL(MGDT_BAD_ARG, "pixel_gate_bwd: null/empty view");
  if (!tt_v4(g, dtype) || !tt_v4(x, dtype) || !tt_v4(gx, dtype) || !tt_same(g, x) || !tt_same(gx, x) || logit->c != 1 || glogit->c != 1 || logit->n != x->n ||
      logit->h != x->h || logit->w != x->w || glogit->n != x->n || glogit->h != x->h || glogit->w != x->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "pixel_gate_bwd: g / x / gx matching 4-aligned NHWC views, one-channel logit maps");
  const long M = (long)x->n * x->h * x->w;
  MGDT_DISPATCH_DTYPE(dtype, (pixel_gate_bwd_kernel<T><<<(int)std::min<long>((M + 15) / 16, 8192), 256, 0, (hipStream_t)s>>>(*g, *x, *logit, *gx, *glogit)));
  MGDT_CHECK_LAUNCH("pixel_gate_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ layer attention backward (TaskDecomposition, head.py:107-131)
// forward (tood_layer_attn_kernel): avg = sums/hw; z1 = W1 avg + b1; hd = relu(z1); z2 = W2 hd + b2; wg = sigmoid(z2); scale[n][k*feat + j] = wg[k].
// Given dscale[n][C]: dsums[n][C] and this image's parameter gradients pimg[n][P], P = hid*C + hid + S*hid + S laid out [dW1 | db1 | dW2 | db2].
__global__ __launch_bounds__(256) void tood_layer_attn_bwd_kernel(const float* __restrict__ sums, const float* __restrict__ dscale, float inv_hw,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, int C, int hid, int S, float* __restrict__ dsums,
                                                                  float* __restrict__ pimg) {
  extern __shared__ float sm[];   // avg[C] | z1[hid] | wg[S] | dz2[S] | dz1[hid]
  float* avg = sm; float* z1 = sm + C; float* wg = z1 + hid; float* dz2 = wg + S; float* dz1 = dz2 + S;
  const int n = blockIdx.x, feat = C / S;
  const long P = (long)hid * C + hid + (long)S * hid + S;
  float* dW1 = pimg + (long)n * P; float* db1 = dW1 + (long)hid * C; float* dW2 = db1 + hid; float* db2 = dW2 + (long)S * hid;
  for (int c = threadIdx.x; c < C; c += 256) avg[c] = sums[(long)n * C + c] * inv_hw;
  __syncthreads();
  for (int j = threadIdx.x; j < hid; j += 256) {
    float a = b1[j];
    for (int c = 0; c < C; ++c) a = fmaf(w1[(long)j * C + c], avg[c], a);
    z1[j] = a;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < S; k += 256) {
    float a = b2[k];
    for (int j = 0; j < hid; ++j) a = fmaf(w2[(long)k * hid + j], fmaxf(z1[j], 0.f), a);
    const float sg = 1.f / (1.f + expf(-a));
    float dw = 0.f;
    for (int j = 0; j < feat; ++j) dw += dscale[(long)n * C + k * feat + j];
    wg[k] = sg;
    dz2[k] = dw * sg * (1.f - sg);
    db2[k] = dz2[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * hid; i += 256) dW2[i] = dz2[i / hid] * fmaxf(z1[i % hid], 0.f);
  for (int j = threadIdx.x; j < hid; j += 256) {
    float a = 0.f;
    for (int k = 0; k < S; ++k) a = fmaf(w2[(long)k * hid + j], dz2[k], a);
    dz1[j] = z1[j] > 0.f ? a : 0.f;
    db1[j] = dz1[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hid * C; i += 256) dW1[i] = dz1[i / C] * avg[i % C];
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int j = 0; j < hid; ++j) a = fmaf(w1[(long)j * C + c], dz1[j], a);
    dsums[(long)n * C + c] = a * inv_hw;
  }
}
extern "C" size_t mgdt_tood_layer_attn_bwd_workspace_bytes(int n, int c, int hid, int stacked) {
  return (size_t)n * ((size_t)hid * c + hid + (size_t)stacked * hid + stacked) * sizeof(float);
}
// dW1 (hid*c), db1 (hid), dW2 (stacked*hid), db2 (stacked): written or accumulated
extern "C" int mgdt_tood_layer_attn_bwd(const float* sums, const float* dscale, int n, int c, int hw, const float* w1, const float* b1, const float* w2,
                                        const float* b2, int hid, int stacked, float* dsums, float* dw1, float* db1, float* dw2, float* db2, int accumulate,
                                        void* ws, mgdt_stream s) {
  if (!sums || !dscale || !w1 || !b1 || !w2 || !b2 || !dsums || !dw1 || !db1 || !dw2 || !db2 || !ws) MGDT_FAIL(MGDT_BAD_ARG, "tood_layer_attn_bwd: null pointer");
  if (n < 1 || c < 1 || hid < 1 || stacked < 1 || c % stacked || hw < 1 || c > 8192) MGDT_FAIL(MGDT_BAD_SHAPE, "tood_layer_attn_bwd: c=%d hid=%d stacked=%d", c, hid, stacked);
  hipStream_t st = (hipStream_t)s;
  const long P = (long)hid * c + hid + (long)stacked * hid + stacked;
  tood_layer_attn_bwd_kernel<<<n, 256, (size_t)(c + 2 * hid + 2 * stacked) * sizeof(float), st>>>(sums, dscale, 1.f / (float)hw, w1, b1, w2, b2, c, hid, stacked, dsums,
                                                                                                  (float*)ws);
  const float* w_ = (const float*)ws;
  tt_sum_rows_kernel<<<cdiv(hid * c, 256), 256, 0, st>>>(w_, n, P, hid * c, dw1, accumulate);
  tt_sum_rows_kernel<<<1, 256, 0, st>>>(w_ + (long)hid * c, n, P, hid, db1, accumulate);
  tt_sum_rows_kernel<<<cdiv(stacked * hid, 256), 256, 0, st>>>(w_ + (long)hid * c + hid, n, P, stacked * hid, dw2, accumulate);
  tt_sum_rows_kernel<<<1, 256, 0, st>>>(w_ + (long)hid * c + hid + (long)stacked * hid, n, P, stacked, db2, accumulate);
  MGDT_CHECK_LAUNCH("tood_layer_attn_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ DCNv2 columns and their adjoint
struct DcnTap { float c1, c2, c3, c4, mk; int h0, w0; bool v1, v2, v3, v4, inside; float lh, lw; };
template <typename T>
__device__ __forceinline__ DcnTap dcn_tap(const T* op, int oy, int ox, int tap, int H, int W) {
  DcnTap t;
  const float hy = (float)(oy - 1 + tap / 3) + (float)op[2 * tap], wx = (float)(ox - 1 + tap % 3) + (float)op[2 * tap + 1];
  t.mk = 1.f / (1.f + expf(-(float)op[18 + tap]));
  t.inside = hy > -1.f && wx > -1.f && hy < (float)H && wx < (float)W;
  t.h0 = (int)floorf(hy); t.w0 = (int)floorf(wx);
  t.lh = hy - (float)t.h0; t.lw = wx - (float)t.w0;
  const int h1 = t.h0 + 1, w1 = t.w0 + 1;
  t.v1 = t.inside && t.h0 >= 0 && t.w0 >= 0; t.v2 = t.inside && t.h0 >= 0 && w1 <= W - 1;
  t.v3 = t.inside && h1 <= H - 1 && t.w0 >= 0; t.v4 = t.inside && h1 <= H - 1 && w1 <= W - 1;
  t.c1 = (1.f - t.lh) * (1.f - t.lw); t.c2 = (1.f - t.lh) * t.lw; t.c3 = t.lh * (1.f - t.lw); t.c4 = t.lh * t.lw;
  return t;
}
// col[n, y, x, c*9 + tap] = mask * bilinear(x; p + offset)      thread = (pixel, tap, channel quad)
template <typename T>
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const mgdt_view x, const mgdt_view om, const mgdt_view col) {
  const int Q = x.c >> 2, H = x.h, W = x.w;
  const long total = (long)x.n * H * W * 9 * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % Q);
    long t_ = i / Q;
    const int tap = (int)(t_ % 9);
    t_ /= 9;
    const int ox = (int)(t_ % W);
    t_ /= W;
    const int oy = (int)(t_ % H);
    const long n = t_ / H;
    const DcnTap t = dcn_tap<T>(TT_AT(const T, om, n, oy, ox, 0), oy, ox, tap, H, W);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t.v1) v += load4<T>(TT_AT(const T, x, n, t.h0, t.w0, 4 * q)) * t.c1;
    if (t.v2) v += load4<T>(TT_AT(const T, x, n, t.h0, t.w0 + 1, 4 * q)) * t.c2;
    if (t.v3) v += load4<T>(TT_AT(const T, x, n, t.h0 + 1, t.w0, 4 * q)) * t.c3;
    if (t.v4) v += load4<T>(TT_AT(const T, x, n, t.h0 + 1, t.w0 + 1, 4 * q)) * t.c4;
    T* cp = TT_AT(T, col, n, oy, ox, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) stf<T>(cp + (4 * q + j) * 9 + tap, v[j] * t.mk);     // channel-major, tap-minor: the 1x1 weight is weight.view(cout, cin*9) as it is
  }
}
extern "C" int mgdt_dcn_im2col(const mgdt_view* x, const mgdt_view* om, const mgdt_view* col, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(om) || !view_ok(col)) MGDT_FAIL(MGDT_BAD_ARG, "dcn_im2col: null/empty view");
  if (!tt_v4(x, dtype) || !tt_v4(col, dtype) || om->sc != 1 || om->c < 27 || col->c != 9 * x->c || om->n != x->n || om->h != x->h || om->w != x->w || col->n != x->n ||
      col->h != x->h || col->w != x->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dcn_im2col: x (c %% 4 == 0), offset/mask map with >= 27 channels, col with 9*c channels");
  const long nq = (long)x->n * x->h * x->w * 9 * (x->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (dcn_im2col_kernel<T><<<tt_grid(nq), 256, 0, (hipStream_t)s>>>(*x, *om, *col)));
  MGDT_CHECK_LAUNCH("dcn_im2col");
  return MGDT_OK;
}

// Adjoint of dcn_im2col for one (pixel, tap): 8 lanes share the channels.  With gc = d loss / d col[c*9 + tap]:
//   gx[corner][c] += gc * mask * coef_corner               (float atomics into the fp32 buffer gx_f32: the only order-dependent sum, as in mmcv)
//   d mask        = sum_c gc * bilinear(x)  ->  d logit = d mask * m * (1 - m)
//   d offset_h    = sum_c gc * mask * sum_corners (d coef / d h) * x[corner],  d offset_w likewise      (mmcv dmcn_get_coordinate_weight)
template <typename T>
__global__ __launch_bounds__(256) void dcn_col2im_bwd_kernel(const mgdt_view gcol, const mgdt_view x, const mgdt_view om, float* __restrict__ gx, const mgdt_view gom) {
  const int Q = x.c >> 2, H = x.h, W = x.w, C = x.c, lane = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const long M = (long)x.n * H * W * 9;
  for (long m0 = blockIdx.x * 32L; m0 < M; m0 += gridDim.x * 32L) {          // uniform trip count per wave (shuffles)
    const long m = m0 + grp;
    const bool ok = m < M;
    const long mm = ok ? m : 0;
    const int tap = (int)(mm % 9);
    long t_ = mm / 9;
    const int ox = (int)(t_ % W);
    t_ /= W;
    const int oy = (int)(t_ % H);
    const long n = t_ / H;
    const DcnTap t = dcn_tap<T>(TT_AT(const T, om, n, oy, ox, 0), oy, ox, tap, H, W);
    float dm = 0.f, dh = 0.f, dw = 0.f;
    if (ok && t.inside)
      for (int q = lane; q < Q; q += 8) {
        const T* gp = TT_AT(const T, gcol, n, oy, ox, 0);
        const f32x4 gc = {ldf<T>(gp + (4 * q) * 9 + tap), ldf<T>(gp + (4 * q + 1) * 9 + tap), ldf<T>(gp + (4 * q + 2) * 9 + tap), ldf<T>(gp + (4 * q + 3) * 9 + tap)};
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 x1 = t.v1 ? load4<T>(TT_AT(const T, x, n, t.h0, t.w0, 4 * q)) : z, x2 = t.v2 ? load4<T>(TT_AT(const T, x, n, t.h0, t.w0 + 1, 4 * q)) : z;
        const f32x4 x3 = t.v3 ? load4<T>(TT_AT(const T, x, n, t.h0 + 1, t.w0, 4 * q)) : z, x4 = t.v4 ? load4<T>(TT_AT(const T, x, n, t.h0 + 1, t.w0 + 1, 4 * q)) : z;
        const f32x4 val = x1 * t.c1 + x2 * t.c2 + x3 * t.c3 + x4 * t.c4;
        const f32x4 ddh = (x3 - x1) * (1.f - t.lw) + (x4 - x2) * t.lw;      // d val / d h
        const f32x4 ddw = (x2 - x1) * (1.f - t.lh) + (x4 - x3) * t.lh;      // d val / d w
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dm += gc[j] * val[j];
          dh += gc[j] * ddh[j];
          dw += gc[j] * ddw[j];
          const float gm = gc[j] * t.mk;
          float* base = gx + ((n * H) * W) * C + 4 * q + j;
          if (t.v1) atomicAdd(base + ((long)t.h0 * W + t.w0) * C, gm * t.c1);
          if (t.v2) atomicAdd(base + ((long)t.h0 * W + t.w0 + 1) * C, gm * t.c2);
          if (t.v3) atomicAdd(base + ((long)(t.h0 + 1) * W + t.w0) * C, gm * t.c3);
          if (t.v4) atomicAdd(base + ((long)(t.h0 + 1) * W + t.w0 + 1) * C, gm * t.c4);
        }
      }
    for (int o = 1; o < 8; o <<= 1) { dm += __shfl_xor(dm, o); dh += __shfl_xor(dh, o); dw += __shfl_xor(dw, o); }
    if (ok && lane == 0) {
      T* go = TT_AT(T, gom, n, oy, ox, 0);
      stf<T>(go + 2 * tap, dh * t.mk);
      stf<T>(go + 2 * tap + 1, dw * t.mk);
      stf<T>(go + 18 + tap, dm * t.mk * (1.f - t.mk));
      if (tap == 0)
        for (int c = 27; c < gom.c; ++c) stf<T>(go + c, 0.f);                 // padding channels of the offset/mask conv
    }
  }
}
// gx_f32: dense fp32 [n][h][w][c], must be ZERO on entry (the scatter accumulates).  gom: NHWC map with >= 27 channels (all written).
extern "C" int mgdt_dcn_col2im_bwd(const mgdt_view* gcol, const mgdt_view* x, const mgdt_view* om, float* gx_f32, const mgdt_view* gom, int dtype, mgdt_stream s) {
  if (!view_ok(gcol) || !view_ok(x) || !view_ok(om) || !gx_f32 || !view_ok(gom)) MGDT_FAIL(MGDT_BAD_ARG, "dcn_col2im_bwd: null/empty argument");
  if (!tt_v4(x, dtype) || !tt_v4(gcol, dtype) || om->sc != 1 || gom->sc != 1 || om->c < 27 || gom->c < 27 || gcol->c != 9 * x->c || om->n != x->n || om->h != x->h ||
      om->w != x->w || gom->n != x->n || gom->h != x->h || gom->w != x->w || gcol->n != x->n || gcol->h != x->h || gcol->w != x->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dcn_col2im_bwd: x (c %% 4 == 0), column gradient with 9*c channels, offset/mask maps with >= 27 channels");
  const long M = (long)x->n * x->h * x->w * 9;
  MGDT_DISPATCH_DTYPE(dtype, (dcn_col2im_bwd_kernel<T><<<(int)std::min<long>((M + 31) / 32, 16384), 256, 0, (hipStream_t)s>>>(*gcol, *x, *om, gx_f32, *gom)));
  MGDT_CHECK_LAUNCH("dcn_col2im_bwd");
  return MGDT_OK;
}
