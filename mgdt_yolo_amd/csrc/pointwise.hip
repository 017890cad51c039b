// HBM-bound companions of the convolutions: pooling-attention (SPR), SPPF max-pools, GD-neck resamplers,
// ConvNeXtV2 depthwise+LayerNorm, GRN statistics, the Injection tail and the Detect decode.
// All work on NHWC views, 4 channels (16 B fp32 / 8 B bf16) per lane, channel-fastest so that a wave touches
// whole contiguous lines.
#include "common.h"

#define SPR_SPLITS 16

static inline int grid_for(long work, int block = 256, int cap = 8192) { return (int)std::min<long>((work + block - 1) / block, cap); }

static inline bool vec4_ok(const mgdt_view* v, int dtype) {
  return v->sc == 1 && v->c % 4 == 0 && v->sw % 4 == 0 && v->sh % 4 == 0 && v->sn % 4 == 0 &&
         ((uintptr_t)v->p % (4 * dtype_size(dtype))) == 0;
}

// adaptive_avg_pool2d bin [start, end) for output index o of `osz` bins over `isz` inputs (ATen rule)
__device__ __forceinline__ int bin_start(int o, int isz, int osz) { return (int)(((long)o * isz) / osz); }
__device__ __forceinline__ int bin_end(int o, int isz, int osz) { return (int)(((long)(o + 1) * isz + osz - 1) / osz); }

// ------------------------------------------------------------------------------------------------ copy / cast / layout
template <typename TX, typename TY>
__global__ void copy_kernel(const TX* __restrict__ x, long xsn, long xsh, long xsw, long xsc, TY* __restrict__ y, long ysn,
                            long ysh, long ysw, long ysc, int N, int H, int W, int C) {
  long total = (long)N * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long t = i / C;
    int w = (int)(t % W);
    t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    y[n * ysn + h * ysh + w * ysw + c * ysc] = (TY)(float)x[n * xsn + h * xsh + w * xsw + c * xsc];
  }
}

extern "C" int mgdt_copy_fwd(const mgdt_view* x, int xdt, const mgdt_view* y, int ydt, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "copy: null/empty view");
  if (x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "copy: shape mismatch");
  long total = (long)x->n * x->h * x->w * x->c;
  int g = grid_for(total);
  hipStream_t st = (hipStream_t)s;
#define L(TX, TY) copy_kernel<TX, TY><<<g, 256, 0, st>>>((const TX*)x->p, x->sn, x->sh, x->sw, x->sc, (TY*)y->p, y->sn, y->sh, y->sw, y->sc, x->n, x->h, x->w, x->c)
  if (xdt == MGDT_F32 && ydt == MGDT_F32) L(float, float);
  else if (xdt == MGDT_F32 && ydt == MGDT_BF16) L(float, bf16);
  else if (xdt == MGDT_BF16 && ydt == MGDT_F32) L(bf16, float);
  else if (xdt == MGDT_BF16 && ydt == MGDT_BF16) L(bf16, bf16);
  else MGDT_FAIL(MGDT_BAD_DTYPE, "copy: dtypes %d -> %d", xdt, ydt);
#undef L
  MGDT_CHECK_LAUNCH("copy_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ SPR pooling
// partial[n][split][c][5] = sums over the row band of {all, bin00, bin01, bin10, bin11}; bins follow adaptive_avg_pool2d(2).
template <typename T>
__global__ __launch_bounds__(256) void spr_pool_kernel(const T* __restrict__ x, long sn, long sh, long sw, int H, int W, int C,
                                                       float* __restrict__ partial) {
  const int n = blockIdx.x, split = blockIdx.y;
  const int Q = C / 4;                     // channel quads
  const int q = threadIdx.x % Q, prow = threadIdx.x / Q, PR = 256 / Q;
  const int r0 = (int)((long)split * H / SPR_SPLITS), r1 = (int)((long)(split + 1) * H / SPR_SPLITS);
  const int hs1 = bin_start(1, H, 2), he0 = bin_end(0, H, 2), ws1 = bin_start(1, W, 2), we0 = bin_end(0, W, 2);
  f32x4 acc[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (prow < PR) {
    const long npix = (long)(r1 - r0) * W;
    for (long p = prow; p < npix; p += PR) {
      int yy = r0 + (int)(p / W), xx = (int)(p % W);
      f32x4 v = load4<T>(x + n * sn + yy * sh + xx * sw + q * 4);
      bool t0 = yy < he0, t1 = yy >= hs1, l0 = xx < we0, l1 = xx >= ws1;
      acc[0] += v;
      if (t0 && l0) acc[1] += v;
      if (t0 && l1) acc[2] += v;
      if (t1 && l0) acc[3] += v;
      if (t1 && l1) acc[4] += v;
    }
  }
  __shared__ float red[256 * 20];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(k * 4 + j) * 256 + threadIdx.x] = acc[k][j];
  __syncthreads();
  // thread t < Q*20 reduces one (q, k, j) over the PR pixel rows (fixed order -> deterministic)
  for (int o = threadIdx.x; o < Q * 20; o += 256) {
    int qq = o % Q, kj = o / Q;
    float sum = 0.f;
    for (int pr = 0; pr < PR; ++pr) sum += red[kj * 256 + pr * Q + qq];
    int k = kj / 4, j = kj % 4;
    partial[(((long)n * SPR_SPLITS + split) * C + qq * 4 + j) * 5 + k] = sum;
  }
}

extern "C" int mgdt_spr_pool_fwd(const mgdt_view* x, float* pooled, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !pooled) MGDT_FAIL(MGDT_BAD_ARG, "spr_pool: null/empty argument");
  if (!vec4_ok(x, dtype) || x->c / 4 > 256) MGDT_FAIL(MGDT_BAD_SHAPE, "spr_pool: need NHWC view, c%%4==0, c<=1024 (c=%d)", x->c);
  dim3 grid(x->n, SPR_SPLITS);
  MGDT_DISPATCH_DTYPE(dtype, (spr_pool_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w, x->c, pooled)));
  MGDT_CHECK_LAUNCH("spr_pool_fwd");
  return MGDT_OK;
}

// one block per image: finish the means, run the shared SPR MLP on each of the `groups` channel groups, softmax over groups
__global__ __launch_bounds__(256) void spr_attn_kernel(const float* __restrict__ partial, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, int C, int G, int H, int W,
                                                       float* __restrict__ attn) {
  extern __shared__ float sm[];
  const int n = blockIdx.x, cw = C / G, hid = cw / 4;
  float* pooled = sm;               // [C][5] means
  float* hbuf = sm + C * 5;         // [G][hid]
  float* obuf = hbuf + G * hid;     // [C] sigmoid outputs
  const int hs1 = bin_start(1, H, 2), he0 = bin_end(0, H, 2), ws1 = bin_start(1, W, 2), we0 = bin_end(0, W, 2);
  const float cnt[5] = {(float)H * W, (float)he0 * we0, (float)he0 * (W - ws1), (float)(H - hs1) * we0, (float)(H - hs1) * (W - ws1)};
  for (int i = threadIdx.x; i < C * 5; i += 256) {
    float sum = 0.f;
    for (int sp = 0; sp < SPR_SPLITS; ++sp) sum += partial[((long)n * SPR_SPLITS + sp) * C * 5 + i];
    pooled[i] = sum / cnt[i % 5];
  }
  __syncthreads();
  // fc1 + relu: input vector of group gi = [pool1(c) for c in group] ++ [pool2(c, bin) c-major]   (spr_module.py:21-23)
  for (int o = threadIdx.x; o < G * hid; o += 256) {
    int gi = o / hid, hj = o % hid;
    const float* wr = w1 + (long)hj * 5 * cw;
    float acc = b1[hj];
    for (int c = 0; c < cw; ++c) acc = fmaf(wr[c], pooled[(gi * cw + c) * 5 + 0], acc);
    for (int c = 0; c < cw; ++c)
      for (int bn = 0; bn < 4; ++bn) acc = fmaf(wr[cw + c * 4 + bn], pooled[(gi * cw + c) * 5 + 1 + bn], acc);
    hbuf[o] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += 256) {
    int gi = o / cw, c = o % cw;
    float acc = b2[c];
    for (int hj = 0; hj < hid; ++hj) acc = fmaf(w2[(long)c * hid + hj], hbuf[gi * hid + hj], acc);
    obuf[o] = 1.f / (1.f + expf(-acc));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cw; c += 256) {   // softmax over the G groups (block.py:278)
    float mx = -INFINITY;
    for (int gi = 0; gi < G; ++gi) mx = fmaxf(mx, obuf[gi * cw + c]);
    float den = 0.f;
    for (int gi = 0; gi < G; ++gi) den += expf(obuf[gi * cw + c] - mx);
    for (int gi = 0; gi < G; ++gi) attn[(long)n * C + gi * cw + c] = expf(obuf[gi * cw + c] - mx) / den;
  }
}

extern "C" int mgdt_spr_attn_fwd(const float* pooled, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                 const float* fc2_b, int n, int c, int groups, int h, int w, float* attn, mgdt_stream s) {
  if (!pooled || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !attn) MGDT_FAIL(MGDT_BAD_ARG, "spr_attn: null pointer");
  if (groups < 1 || c % groups || (c / groups) % 4 || c > 4096 || h < 1 || w < 1) MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn: c=%d groups=%d", c, groups);
  int cw = c / groups, hid = cw / 4;
  size_t lds = (size_t)(c * 5 + groups * hid + c) * sizeof(float);
  spr_attn_kernel<<<n, 256, lds, (hipStream_t)s>>>(pooled, fc1_w, fc1_b, fc2_w, fc2_b, c, groups, h, w, attn);
  MGDT_CHECK_LAUNCH("spr_attn_fwd");
  return MGDT_OK;
}

template <typename T>
__global__ void scale_channels_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, const float* __restrict__ attn,
                                      T* __restrict__ y, long ysn, long ysh, long ysw, int N, int H, int W, int C) {
  const int Q = C / 4;
  long total = (long)N * H * W * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int w = (int)(t % W);
    t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    f32x4 v = load4<T>(x + n * xsn + h * xsh + w * xsw + q * 4);
    f32x4 a = *(const f32x4*)(attn + (long)n * C + q * 4);
    store4<T>(y + n * ysn + h * ysh + w * ysw + q * 4, v * a);
  }
}

extern "C" int mgdt_scale_channels_fwd(const mgdt_view* x, const float* attn, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !attn) MGDT_FAIL(MGDT_BAD_ARG, "scale_channels: null/empty argument");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "scale_channels: views must be matching NHWC, c%%4==0");
  long total = (long)x->n * x->h * x->w * (x->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (scale_channels_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>(
                                 (const T*)x->p, x->sn, x->sh, x->sw, attn, (T*)y->p, y->sn, y->sh, y->sw, x->n, x->h, x->w, x->c)));
  MGDT_CHECK_LAUNCH("scale_channels_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ SPPF pools
// maxpool5 applied 1x/2x/3x == max over (5,9,13)-windows with -inf padding; one pass over x writes all three.
template <typename T>
__global__ void sppf_pool_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, T* __restrict__ y1, long s1n, long s1h,
                                 long s1w, T* __restrict__ y2, long s2n, long s2h, long s2w, T* __restrict__ y3, long s3n,
                                 long s3h, long s3w, int N, int H, int W, int C) {
  const int Q = C / 4;
  long total = (long)N * H * W * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int w = (int)(t % W);
    t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    f32x4 m5 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}, m9 = m5, m13 = m5;
    for (int dy = -6; dy <= 6; ++dy) {
      int yy = h + dy;
      if (yy < 0 || yy >= H) continue;
      int ady = dy < 0 ? -dy : dy;
      for (int dx = -6; dx <= 6; ++dx) {
        int xx = w + dx;
        if (xx < 0 || xx >= W) continue;
        int adx = dx < 0 ? -dx : dx;
        int rad = ady > adx ? ady : adx;
        f32x4 v = load4<T>(x + n * xsn + yy * xsh + xx * xsw + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m13[j] = fmaxf(m13[j], v[j]);
          if (rad <= 4) m9[j] = fmaxf(m9[j], v[j]);
          if (rad <= 2) m5[j] = fmaxf(m5[j], v[j]);
        }
      }
    }
    store4<T>(y1 + n * s1n + h * s1h + w * s1w + q * 4, m5);
    store4<T>(y2 + n * s2n + h * s2h + w * s2w + q * 4, m9);
    store4<T>(y3 + n * s3n + h * s3h + w * s3w + q * 4, m13);
  }
}

extern "C" int mgdt_sppf_pool_fwd(const mgdt_view* x, const mgdt_view* y1, const mgdt_view* y2, const mgdt_view* y3, int dtype,
                                  mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y1) || !view_ok(y2) || !view_ok(y3)) MGDT_FAIL(MGDT_BAD_ARG, "sppf_pool: null/empty view");
  for (const mgdt_view* v : {x, y1, y2, y3})
    if (!vec4_ok(v, dtype) || v->n != x->n || v->h != x->h || v->w != x->w || v->c != x->c)
      MGDT_FAIL(MGDT_BAD_SHAPE, "sppf_pool: views must be matching NHWC, c%%4==0");
  long total = (long)x->n * x->h * x->w * (x->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (sppf_pool_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>(
                                 (const T*)x->p, x->sn, x->sh, x->sw, (T*)y1->p, y1->sn, y1->sh, y1->sw, (T*)y2->p, y2->sn, y2->sh,
                                 y2->sw, (T*)y3->p, y3->sn, y3->sh, y3->sw, x->n, x->h, x->w, x->c)));
  MGDT_CHECK_LAUNCH("sppf_pool_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ resamplers
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                               long ysh, long ysw, int N, int Ho, int Wo, int C) {
  const int Q = C / 4;
  long total = (long)N * Ho * Wo * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int ox = (int)(t % Wo);
    t /= Wo;
    int oy = (int)(t % Ho);
    int n = (int)(t / Ho);
    int y0 = bin_start(oy, H, Ho), y1 = bin_end(oy, H, Ho), x0 = bin_start(ox, W, Wo), x1 = bin_end(ox, W, Wo);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) acc += load4<T>(x + n * xsn + yy * xsh + xx * xsw + q * 4);
    float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, acc * inv);
  }
}

extern "C" int mgdt_adaptive_avgpool_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "avgpool: null/empty view");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "avgpool: NHWC views, c%%4==0, same n/c");
  long total = (long)y->n * y->h * y->w * (y->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (avgpool_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w,
                                                                                        (T*)y->p, y->sn, y->sh, y->sw, y->n, y->h, y->w, y->c)));
  MGDT_CHECK_LAUNCH("adaptive_avgpool_fwd");
  return MGDT_OK;
}

// F.interpolate(bilinear, align_corners=False): src = max(0, (dst+0.5)*in/out - 0.5), upper neighbour clamped
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int o, int isz, int osz) {
  float scale = (float)isz / (float)osz;
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > isz - 1) i0 = isz - 1;
  int i1 = i0 + (i0 < isz - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  return Lerp{i0, i1, 1.f - l1, l1};
}

template <typename T>
__device__ __forceinline__ f32x4 bilerp(const T* x, long sh, long sw, Lerp ly, Lerp lx) {
  f32x4 v00 = load4<T>(x + ly.i0 * sh + lx.i0 * sw), v01 = load4<T>(x + ly.i0 * sh + lx.i1 * sw);
  f32x4 v10 = load4<T>(x + ly.i1 * sh + lx.i0 * sw), v11 = load4<T>(x + ly.i1 * sh + lx.i1 * sw);
  return (v00 * lx.l0 + v01 * lx.l1) * ly.l0 + (v10 * lx.l0 + v11 * lx.l1) * ly.l1;
}

template <typename T>
__global__ void bilinear_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                                long ysh, long ysw, int N, int Ho, int Wo, int C) {
  const int Q = C / 4;
  long total = (long)N * Ho * Wo * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int ox = (int)(t % Wo);
    t /= Wo;
    int oy = (int)(t % Ho);
    int n = (int)(t / Ho);
    store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, bilerp<T>(x + n * xsn + q * 4, xsh, xsw, lerp_of(oy, H, Ho), lerp_of(ox, W, Wo)));
  }
}

extern "C" int mgdt_bilinear_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "bilinear: null/empty view");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "bilinear: NHWC views, c%%4==0, same n/c");
  long total = (long)y->n * y->h * y->w * (y->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (bilinear_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w,
                                                                                         (T*)y->p, y->sn, y->sh, y->sw, y->n, y->h, y->w, y->c)));
  MGDT_CHECK_LAUNCH("bilinear_fwd");
  return MGDT_OK;
}

template <typename T>
__global__ void nearest_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                               long ysh, long ysw, int N, int Ho, int Wo, int C) {
  const int Q = C / 4;
  long total = (long)N * Ho * Wo * Q;
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int ox = (int)(t % Wo);
    t /= Wo;
    int oy = (int)(t % Ho);
    int n = (int)(t / Ho);
    int iy = min((int)floorf((float)oy * sy), H - 1), ix = min((int)floorf((float)ox * sx), W - 1);
    store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, load4<T>(x + n * xsn + iy * xsh + ix * xsw + q * 4));
  }
}

extern "C" int mgdt_nearest_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "nearest: null/empty view");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "nearest: NHWC views, c%%4==0, same n/c");
  long total = (long)y->n * y->h * y->w * (y->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (nearest_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w,
                                                                                        (T*)y->p, y->sn, y->sh, y->sw, y->n, y->h, y->w, y->c)));
  MGDT_CHECK_LAUNCH("nearest_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ Injection tail
template <typename T>
__global__ void inject_kernel(const T* __restrict__ loc, long lsn, long lsh, long lsw, const T* __restrict__ ga, long asn, long ash,
                              long asw, const T* __restrict__ gf, long fsn, long fsh, long fsw, int Hg, int Wg, T* __restrict__ y,
                              long ysn, long ysh, long ysw, int N, int H, int W, int C, int use_pool) {
  const int Q = C / 4;
  long total = (long)N * H * W * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q = (int)(i % Q);
    long t = i / Q;
    int ox = (int)(t % W);
    t /= W;
    int oy = (int)(t % H);
    int n = (int)(t / H);
    f32x4 l = load4<T>(loc + n * lsn + oy * lsh + ox * lsw + q * 4);
    f32x4 sig, feat;
    if (use_pool) {
      int y0 = bin_start(oy, Hg, H), y1 = bin_end(oy, Hg, H), x0 = bin_start(ox, Wg, W), x1 = bin_end(ox, Wg, W);
      sig = f32x4{0.f, 0.f, 0.f, 0.f};
      feat = sig;
      for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
          sig += load4<T>(ga + n * asn + yy * ash + xx * asw + q * 4);
          feat += load4<T>(gf + n * fsn + yy * fsh + xx * fsw + q * 4);
        }
      float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
      sig *= inv;
      feat *= inv;
    } else {
      Lerp ly = lerp_of(oy, Hg, H), lx = lerp_of(ox, Wg, W);
      const T* a = ga + n * asn + q * 4;
      f32x4 h[4] = {load4<T>(a + ly.i0 * ash + lx.i0 * asw), load4<T>(a + ly.i0 * ash + lx.i1 * asw),
                    load4<T>(a + ly.i1 * ash + lx.i0 * asw), load4<T>(a + ly.i1 * ash + lx.i1 * asw)};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          h[k][j] = fminf(fmaxf(h[k][j] + 3.f, 0.f), 6.f) / 6.f;   // h_sigmoid BEFORE the interpolation (block.py:393)
      sig = (h[0] * lx.l0 + h[1] * lx.l1) * ly.l0 + (h[2] * lx.l0 + h[3] * lx.l1) * ly.l1;
      feat = bilerp<T>(gf + n * fsn + q * 4, fsh, fsw, ly, lx);
    }
    store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, l * sig + feat);
  }
}

extern "C" int mgdt_inject_fwd(const mgdt_view* local, const mgdt_view* ga, const mgdt_view* gf, const mgdt_view* y, int dtype,
                               mgdt_stream s) {
  if (!view_ok(local) || !view_ok(ga) || !view_ok(gf) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "inject: null/empty view");
  for (const mgdt_view* v : {local, ga, gf, y})
    if (!vec4_ok(v, dtype) || v->n != y->n || v->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "inject: NHWC views, c%%4==0, same n/c");
  if (local->h != y->h || local->w != y->w || ga->h != gf->h || ga->w != gf->w) MGDT_FAIL(MGDT_BAD_SHAPE, "inject: spatial mismatch");
  int use_pool = local->h < ga->h;
  long total = (long)y->n * y->h * y->w * (y->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (inject_kernel<T><<<grid_for(total), 256, 0, (hipStream_t)s>>>(
                                 (const T*)local->p, local->sn, local->sh, local->sw, (const T*)ga->p, ga->sn, ga->sh, ga->sw, (const T*)gf->p,
                                 gf->sn, gf->sh, gf->sw, ga->h, ga->w, (T*)y->p, y->sn, y->sh, y->sw, y->n, y->h, y->w, y->c, use_pool)));
  MGDT_CHECK_LAUNCH("inject_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ ConvNeXtV2: dw7x7 + LayerNorm
// block = PPB pixels x Q channel-quads; LayerNorm over the pixel's channels through LDS (two-pass mean / variance).
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw,
                                                         const float* __restrict__ dw, const float* __restrict__ db,
                                                         const float* __restrict__ lw, const float* __restrict__ lb, float eps,
                                                         T* __restrict__ y, long ysn, long ysh, long ysw, int N, int H, int W, int C) {
  const int Q = C / 4, PPB = 256 / Q;
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const long M = (long)N * H * W;
  const long m = blockIdx.x * (long)PPB + pl;
  const bool active = pl < PPB && m < M;
  __shared__ float red[256];
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int n = 0, oy = 0, ox = 0;
  if (active) {
    n = (int)(m / ((long)H * W));
    int rem = (int)(m - (long)n * H * W);
    oy = rem / W;
    ox = rem - oy * W;
    acc = *(const f32x4*)(db + q * 4);
    for (int ky = 0; ky < 7; ++ky) {
      int iy = oy + ky - 3;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 7; ++kx) {
        int ix = ox + kx - 3;
        if (ix < 0 || ix >= W) continue;
        f32x4 v = load4<T>(x + n * xsn + iy * xsh + ix * xsw + q * 4);
        f32x4 wv = *(const f32x4*)(dw + (long)(ky * 7 + kx) * C + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], wv[j], acc[j]);
      }
    }
  }
  red[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  __syncthreads();
  float mean = 0.f;
  if (pl < PPB) {
    for (int k = 0; k < Q; ++k) mean += red[pl * Q + k];
    mean /= (float)C;
  }
  __syncthreads();
  f32x4 d = acc - mean;
  red[threadIdx.x] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  __syncthreads();
  float var = 0.f;
  if (pl < PPB) {
    for (int k = 0; k < Q; ++k) var += red[pl * Q + k];
    var /= (float)C;
  }
  if (active) {
    float rstd = 1.f / sqrtf(var + eps);
    f32x4 g = *(const f32x4*)(lw + q * 4), b = *(const f32x4*)(lb + q * 4);
    store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, d * rstd * g + b);
  }
}

extern "C" int mgdt_dwconv7_ln_fwd(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                                   float eps, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !dw_w || !dw_b || !ln_w || !ln_b) MGDT_FAIL(MGDT_BAD_ARG, "dwconv7_ln: null/empty argument");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c || x->c / 4 > 256)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dwconv7_ln: matching NHWC views, c%%4==0, c<=1024");
  int Q = x->c / 4, PPB = 256 / Q;
  long M = (long)x->n * x->h * x->w;
  MGDT_DISPATCH_DTYPE(dtype, (dwconv7_ln_kernel<T><<<cdiv(M, PPB), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, dw_w, dw_b, ln_w, ln_b, eps,
                                                                                       (T*)y->p, y->sn, y->sh, y->sw, x->n, x->h, x->w, x->c)));
  MGDT_CHECK_LAUNCH("dwconv7_ln_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ GRN statistics
// ws[n][c] = sum_{h,w} t^2 ; then scale[n][c] = gamma[c] * sqrt(ws) / (mean_c sqrt(ws) + 1e-6) + 1
template <typename T>
__global__ __launch_bounds__(256) void grn_sumsq_kernel(const T* __restrict__ t, long sn, long sh, long sw, int H, int W, int C,
                                                        float* __restrict__ ws) {
  const int n = blockIdx.x, q0 = blockIdx.y * 16;   // 16 quads = 64 channels per block
  const int ql = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int q = q0 + ql;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q * 4 < C) {
    const long npix = (long)H * W;
    for (long p = pr; p < npix; p += 16) {
      f32x4 v = load4<T>(t + n * sn + (p / W) * sh + (p % W) * sw + q * 4);
      acc += v * v;
    }
  }
  __shared__ float red[4][256];
#pragma unroll
  for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    int qq = threadIdx.x >> 2, j = threadIdx.x & 3;
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += red[j][r * 16 + qq];
    int c = (q0 + qq) * 4 + j;
    if (c < C) ws[(long)n * C + c] = sum;
  }
}

__global__ __launch_bounds__(256) void grn_scale_kernel(const float* __restrict__ ws, const float* __restrict__ gamma, int C,
                                                        float* __restrict__ scale) {
  const int n = blockIdx.x;
  __shared__ float red[256];
  float part = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) part += sqrtf(ws[(long)n * C + c]);
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  float mean = red[0] / (float)C;
  for (int c = threadIdx.x; c < C; c += 256) scale[(long)n * C + c] = gamma[c] * (sqrtf(ws[(long)n * C + c]) / (mean + 1e-6f)) + 1.f;
}

extern "C" int mgdt_grn_stats_fwd(const mgdt_view* t, const float* gamma, float* ws, float* scale, int dtype, mgdt_stream s) {
  if (!view_ok(t) || !gamma || !ws || !scale) MGDT_FAIL(MGDT_BAD_ARG, "grn_stats: null/empty argument");
  if (!vec4_ok(t, dtype)) MGDT_FAIL(MGDT_BAD_SHAPE, "grn_stats: NHWC view, c%%4==0");
  dim3 grid(t->n, cdiv(t->c, 64));
  MGDT_DISPATCH_DTYPE(dtype, (grn_sumsq_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>((const T*)t->p, t->sn, t->sh, t->sw, t->h, t->w, t->c, ws)));
  grn_scale_kernel<<<t->n, 256, 0, (hipStream_t)s>>>(ws, gamma, t->c, scale);
  MGDT_CHECK_LAUNCH("grn_stats_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ Detect decode
// One thread per (image, anchor).  y[n][ch][a] is anchor-contiguous, so a wave's stores per channel are coalesced.
template <typename T>
__global__ void detect_decode_kernel(const T* __restrict__ f, long sn, long sh, long sw, int N, int H, int W, int R, int nc,
                                     float stride, int a_off, int a_total, float* __restrict__ y) {
  long total = (long)N * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int a = (int)(i % ((long)H * W));
    int n = (int)(i / ((long)H * W));
    int oy = a / W, ox = a - oy * W;
    const T* p = f + n * sn + oy * sh + ox * sw;
    float d[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
      float mx = -INFINITY;
      for (int k = 0; k < R; ++k) mx = fmaxf(mx, (float)p[side * R + k]);
      float den = 0.f, num = 0.f;
      for (int k = 0; k < R; ++k) {
        float e = expf((float)p[side * R + k] - mx);
        den += e;
        num += e * (float)k;
      }
      d[side] = num / den;
    }
    float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
    float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    float* yo = y + (long)n * (4 + nc) * a_total + a_off + a;
    yo[0] = (x1 + x2) / 2.f * stride;
    yo[(long)a_total] = (y1 + y2) / 2.f * stride;
    yo[2L * a_total] = (x2 - x1) * stride;
    yo[3L * a_total] = (y2 - y1) * stride;
    for (int c = 0; c < nc; ++c) yo[(long)(4 + c) * a_total] = 1.f / (1.f + expf(-(float)p[4 * R + c]));
  }
}

extern "C" int mgdt_detect_decode_fwd(const mgdt_view* feat, int reg_max, int nc, float stride, int a_off, int a_total, float* y,
                                      int dtype, mgdt_stream s) {
  if (!view_ok(feat) || !y) MGDT_FAIL(MGDT_BAD_ARG, "detect_decode: null/empty argument");
  if (feat->sc != 1 || feat->c != 4 * reg_max + nc || reg_max < 1 || a_off < 0 || a_off + feat->h * feat->w > a_total)
    MGDT_FAIL(MGDT_BAD_SHAPE, "detect_decode: c=%d reg_max=%d nc=%d a_off=%d a_total=%d", feat->c, reg_max, nc, a_off, a_total);
  long total = (long)feat->n * feat->h * feat->w;
  MGDT_DISPATCH_DTYPE(dtype, (detect_decode_kernel<T><<<grid_for(total, 64), 64, 0, (hipStream_t)s>>>((const T*)feat->p, feat->sn, feat->sh, feat->sw, feat->n,
                                                                                                feat->h, feat->w, reg_max, nc, stride, a_off, a_total, y)));
  MGDT_CHECK_LAUNCH("detect_decode_fwd");
  return MGDT_OK;
}
