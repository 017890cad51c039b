// Adjoints of the MSPA pooling-attention and Gather-Distribute neck ops (training side of nn/modules/block.py:209-399,
// spr_module.py, convnextv2.py:48-77, utils.py:145-182).  First-cut, deterministic (fixed-order reductions), NHWC.
#include "common.h"

#define NC_SPLITS 64

// channel-vectorised forms (train_vec.hip); false -> run the scalar kernel
bool mgdt_v4_ew_binary(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int mode, int dtype, hipStream_t st);
bool mgdt_v4_channel_affine(const mgdt_view* x, const float* scale, const float* shift, const mgdt_view* y, int dtype, hipStream_t st);
bool mgdt_v4_avgpool_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, hipStream_t st);
bool mgdt_v4_bilinear_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, hipStream_t st);
bool mgdt_v4_grn_bwd_apply(const mgdt_view* g, const mgdt_view* t, const float* scale, const float* coef, const mgdt_view* dt, int dtype, hipStream_t st);
bool mgdt_v4_spr_out_bwd(const mgdt_view* gy, const float* attn, const float* dpooled, const mgdt_view* gx, int dtype, hipStream_t st);
bool mgdt_v4_nc_reduce_partial(const mgdt_view* a, const mgdt_view* b, float* partial, int nsplit, int dtype, hipStream_t st);
static inline int ew_grid(long total) { return (int)std::min<long>((total + 255) / 256, 16384); }
__device__ __forceinline__ int bin_start(int o, int isz, int osz) { return (int)(((long)o * isz) / osz); }
__device__ __forceinline__ int bin_end(int o, int isz, int osz) { return (int)(((long)(o + 1) * isz + osz - 1) / osz); }

#define DECODE_NHWC(i, v, n, h, w, c) \
  int c = (int)((i) % (v).c);         \
  long t_ = (i) / (v).c;              \
  int w = (int)(t_ % (v).w);          \
  t_ /= (v).w;                        \
  int h = (int)(t_ % (v).h);          \
  long n = t_ / (v).h;
#define AT(T, v, n, h, w, c) (((T*)(v).p)[(n) * (v).sn + (h) * (v).sh + (w) * (v).sw + (c)])

// ------------------------------------------------------------------------------------------------ elementwise helpers
// mode 0: o = a*b    mode 1: o = a * hsigmoid'(b)  (1/6 inside (-3,3))    mode 2: o = a * hsigmoid(b)    mode 3: o = hsigmoid(b)
template <typename T>
__global__ void ew_binary_kernel(const mgdt_view a, const mgdt_view b, const mgdt_view o, int mode) {
  long total = (long)o.n * o.h * o.w * o.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, o, n, h, w, c)
    float x = (float)AT(const T, a, n, h, w, c), y = (float)AT(const T, b, n, h, w, c), r;
    if (mode == 0) r = x * y;
    else if (mode == 1) r = (y > -3.f && y < 3.f) ? x * (1.f / 6.f) : 0.f;
    else if (mode == 2) r = x * (fminf(fmaxf(y + 3.f, 0.f), 6.f) / 6.f);
    else r = fminf(fmaxf(y + 3.f, 0.f), 6.f) / 6.f;   // mode 3: hsigmoid(b)
    AT(T, o, n, h, w, c) = (T)r;
  }
}

extern "C" int mgdt_ew_binary(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int mode, int dtype, mgdt_stream s) {
  if (!view_ok(a) || !view_ok(b) || !view_ok(o)) MGDT_FAIL(MGDT_BAD_ARG, "ew_binary: null/empty view");
  if (a->sc != 1 || b->sc != 1 || o->sc != 1 || a->n != o->n || a->h != o->h || a->w != o->w || a->c != o->c || b->n != o->n || b->h != o->h || b->w != o->w || b->c != o->c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "ew_binary: matching NHWC views");
  long total = (long)o->n * o->h * o->w * o->c;
  if (!mgdt_v4_ew_binary(a, b, o, mode, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (ew_binary_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*a, *b, *o, mode)));
  MGDT_CHECK_LAUNCH("ew_binary");
  return MGDT_OK;
}

// y = x * scale[n][c] + shift[c]   (GRN applied to a materialised tensor: the training path keeps pwconv2's true input)
template <typename T>
__global__ void channel_affine_kernel(const mgdt_view x, const float* scale, const float* shift, const mgdt_view y) {
  long total = (long)x.n * x.h * x.w * x.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, x, n, h, w, c)
    float v = (float)AT(const T, x, n, h, w, c);
    if (scale) v *= scale[n * x.c + c];
    if (shift) v += shift[c];
    AT(T, y, n, h, w, c) = (T)v;
  }
}
extern "C" int mgdt_channel_affine(const mgdt_view* x, const float* scale, const float* shift, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "channel_affine: null/empty view");
  if (x->sc != 1 || y->sc != 1 || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "channel_affine: matching NHWC views");
  long total = (long)x->n * x->h * x->w * x->c;
  if (!mgdt_v4_channel_affine(x, scale, shift, y, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (channel_affine_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*x, scale, shift, *y)));
  MGDT_CHECK_LAUNCH("channel_affine");
  return MGDT_OK;
}

// out[n][c] = sum_{h,w} a*b   (b may be absent -> sum a); partial[n][split][c] then fixed-order final
template <typename T>
__global__ __launch_bounds__(256) void nc_reduce_partial_kernel(const mgdt_view a, const mgdt_view b, float* partial) {
  const int n = blockIdx.x, split = blockIdx.y, cb = blockIdx.z * 64;
  const int c = cb + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int npix = a.h * a.w;
  const int p0 = (int)((long)split * npix / NC_SPLITS), p1 = (int)((long)(split + 1) * npix / NC_SPLITS);
  float acc = 0.f;
  if (c < a.c)
    for (int p = p0 + pl; p < p1; p += 4) {
      int yy = p / a.w, xx = p - yy * a.w;
      float v = (float)AT(const T, a, (long)n, yy, xx, c);
      if (b.p) v *= (float)AT(const T, b, (long)n, yy, xx, c);
      acc += v;
    }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && c < a.c)
    partial[((long)n * NC_SPLITS + split) * a.c + c] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
}
__global__ void nc_reduce_final_kernel(const float* partial, int N, int C, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  int n = i / C, c = i - n * C;
  float s = 0.f;
  for (int k = 0; k < NC_SPLITS; ++k) s += partial[((long)n * NC_SPLITS + k) * C + c];
  out[i] = s;
}

extern "C" size_t mgdt_nc_reduce_workspace_bytes(int n, int c) { return (size_t)n * NC_SPLITS * c * sizeof(float); }
extern "C" int mgdt_nc_reduce(const mgdt_view* a, const mgdt_view* b, float* out, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(a) || !out || !ws) MGDT_FAIL(MGDT_BAD_ARG, "nc_reduce: null/empty argument");
  if (a->sc != 1 || (b && b->p && (b->sc != 1 || b->n != a->n || b->h != a->h || b->w != a->w || b->c != a->c))) MGDT_FAIL(MGDT_BAD_SHAPE, "nc_reduce: matching NHWC views");
  mgdt_view bb;
  memset(&bb, 0, sizeof(bb));
  if (b && b->p) bb = *b;
  dim3 grid(a->n, NC_SPLITS, cdiv(a->c, 64));
  if (!mgdt_v4_nc_reduce_partial(a, b, (float*)ws, NC_SPLITS, dtype, (hipStream_t)s))
    MGDT_DISPATCH_DTYPE(dtype, (nc_reduce_partial_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>(*a, bb, (float*)ws)));
  nc_reduce_final_kernel<<<cdiv((long)a->n * a->c, 256), 256, 0, (hipStream_t)s>>>((const float*)ws, a->n, a->c, out);
  MGDT_CHECK_LAUNCH("nc_reduce");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ resampler adjoints (gather form)
template <typename T>
__global__ void avgpool_bwd_kernel(const mgdt_view gy, const mgdt_view gx, int accumulate) {
  long total = (long)gx.n * gx.h * gx.w * gx.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, gx, n, h, w, c)
    float acc = 0.f;
    int oy_lo = max(0, (int)(((long)h * gy.h) / gx.h) - 1), oy_hi = min(gy.h - 1, (int)(((long)(h + 1) * gy.h) / gx.h) + 1);
    int ox_lo = max(0, (int)(((long)w * gy.w) / gx.w) - 1), ox_hi = min(gy.w - 1, (int)(((long)(w + 1) * gy.w) / gx.w) + 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0 = bin_start(oy, gx.h, gy.h), y1 = bin_end(oy, gx.h, gy.h);
      if (h < y0 || h >= y1) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0 = bin_start(ox, gx.w, gy.w), x1 = bin_end(ox, gx.w, gy.w);
        if (w < x0 || w >= x1) continue;
        acc += (float)AT(const T, gy, n, oy, ox, c) / (float)((y1 - y0) * (x1 - x0));
      }
    }
    if (accumulate) acc += (float)AT(const T, gx, n, h, w, c);
    AT(T, gx, n, h, w, c) = (T)acc;
  }
}

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
__global__ void bilinear_bwd_kernel(const mgdt_view gy, const mgdt_view gx, int accumulate) {
  long total = (long)gx.n * gx.h * gx.w * gx.c;
  const float ry = (float)gy.h / (float)gx.h, rx = (float)gy.w / (float)gx.w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, gx, n, h, w, c)
    // outputs that can touch input row h: src in (h-1, h+1)  ->  o in ((h-0.5)*r - 0.5 - r, (h+1.5)*r + r)
    int oy_lo = max(0, (int)floorf(((float)h - 1.f) * ry) - 1), oy_hi = min(gy.h - 1, (int)ceilf(((float)h + 2.f) * ry) + 1);
    int ox_lo = max(0, (int)floorf(((float)w - 1.f) * rx) - 1), ox_hi = min(gy.w - 1, (int)ceilf(((float)w + 2.f) * rx) + 1);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      Lerp ly = lerp_of(oy, gx.h, gy.h);
      float wy = (ly.i0 == h ? ly.l0 : 0.f) + (ly.i1 == h ? ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        Lerp lx = lerp_of(ox, gx.w, gy.w);
        float wx = (lx.i0 == w ? lx.l0 : 0.f) + (lx.i1 == w ? lx.l1 : 0.f);
        if (wx == 0.f) continue;
        acc += wy * wx * (float)AT(const T, gy, n, oy, ox, c);
      }
    }
    if (accumulate) acc += (float)AT(const T, gx, n, h, w, c);
    AT(T, gx, n, h, w, c) = (T)acc;
  }
}

#define RESAMPLE_BWD(fname, kern, vkern, label)                                                                                  \
  extern "C" int fname(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, mgdt_stream s) {              \
    if (!view_ok(gy) || !view_ok(gx)) MGDT_FAIL(MGDT_BAD_ARG, label ": null/empty view");                                 \
    if (gy->sc != 1 || gx->sc != 1 || gy->n != gx->n || gy->c != gx->c) MGDT_FAIL(MGDT_BAD_SHAPE, label ": NHWC views, same n/c"); \
    long total = (long)gx->n * gx->h * gx->w * gx->c;                                                                     \
    if (!vkern(gy, gx, accumulate, dtype, (hipStream_t)s))                                                               \
      MGDT_DISPATCH_DTYPE(dtype, (kern<T><<<ew_grid(total), 256, 0, (hipStream_t)s>>>(*gy, *gx, accumulate)));            \
    MGDT_CHECK_LAUNCH(label);                                                                                             \
    return MGDT_OK;                                                                                                       \
  }
RESAMPLE_BWD(mgdt_adaptive_avgpool_bwd, avgpool_bwd_kernel, mgdt_v4_avgpool_bwd, "adaptive_avgpool_bwd")
RESAMPLE_BWD(mgdt_bilinear_bwd, bilinear_bwd_kernel, mgdt_v4_bilinear_bwd, "bilinear_bwd")

// ------------------------------------------------------------------------------------------------ SPR attention backward
// forward (per image, per group gi): v = [pool1(c)] ++ [pool2(c,bin)]; hdn = relu(W1 v + b1); o = sigmoid(W2 hdn + b2);
// attn = softmax over groups.  Given dattn[n][C]: dpooled[n][C][5] and this image's contribution to dW1,db1,dW2,db2
// (pimg[n][P], P = hid*5cw + hid + cw*hid + cw) - summed over images by spr_param_reduce_kernel.
__global__ __launch_bounds__(256) void spr_attn_bwd_kernel(const float* __restrict__ partial, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, int C, int G, int H, int W, int splits,
                                                           const float* __restrict__ dattn, float* __restrict__ dpooled,
                                                           float* __restrict__ pimg) {
  extern __shared__ float sm[];
  const int n = blockIdx.x, cw = C / G, hid = cw / 4;
  float* pooled = sm;                 // [C][5]
  float* hbuf = pooled + C * 5;       // [G][hid] post-relu
  float* obuf = hbuf + G * hid;       // [C] sigmoid outputs
  float* abuf = obuf + C;             // [C] softmax outputs
  float* dz2 = abuf + C;              // [C] grad wrt fc2 pre-activation
  float* dz1 = dz2 + C;               // [G][hid] grad wrt fc1 pre-activation
  const int hs1 = bin_start(1, H, 2), he0 = bin_end(0, H, 2), ws1 = bin_start(1, W, 2), we0 = bin_end(0, W, 2);
  const float cnt[5] = {(float)H * W, (float)he0 * we0, (float)he0 * (W - ws1), (float)(H - hs1) * we0, (float)(H - hs1) * (W - ws1)};
  for (int i = threadIdx.x; i < C * 5; i += 256) {
    float sum = 0.f;
    for (int sp = 0; sp < splits; ++sp) sum += partial[((long)n * splits + sp) * C * 5 + i];
    pooled[i] = sum / cnt[i % 5];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < G * hid; o += 256) {
    int gi = o / hid, hj = o % hid;
    const float* wr = w1 + (long)hj * 5 * cw;
    float acc = b1[hj];
    for (int c = 0; c < cw; ++c) acc = fmaf(wr[c], pooled[(gi * cw + c) * 5], acc);
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
  for (int c = threadIdx.x; c < cw; c += 256) {
    float mx = -INFINITY, den = 0.f;
    for (int gi = 0; gi < G; ++gi) mx = fmaxf(mx, obuf[gi * cw + c]);
    for (int gi = 0; gi < G; ++gi) den += expf(obuf[gi * cw + c] - mx);
    for (int gi = 0; gi < G; ++gi) abuf[gi * cw + c] = expf(obuf[gi * cw + c] - mx) / den;
    // softmax backward over the groups, then sigmoid backward
    float dot = 0.f;
    for (int gi = 0; gi < G; ++gi) dot += dattn[(long)n * C + gi * cw + c] * abuf[gi * cw + c];
    for (int gi = 0; gi < G; ++gi) {
      float a = abuf[gi * cw + c], dob = a * (dattn[(long)n * C + gi * cw + c] - dot), o = obuf[gi * cw + c];
      dz2[gi * cw + c] = dob * o * (1.f - o);
    }
  }
  __syncthreads();
  const int P1 = hid * 5 * cw;
  float* pW1 = pimg + (long)n * (P1 + hid + cw * hid + cw);
  float* pb1 = pW1 + P1;
  float* pW2 = pb1 + hid;
  float* pb2 = pW2 + cw * hid;
  for (int o = threadIdx.x; o < G * hid; o += 256) {      // d hidden (pre-relu)
    int gi = o / hid, hj = o % hid;
    float acc = 0.f;
    for (int c = 0; c < cw; ++c) acc = fmaf(w2[(long)c * hid + hj], dz2[gi * cw + c], acc);
    dz1[o] = hbuf[o] > 0.f ? acc : 0.f;
  }
  for (int o = threadIdx.x; o < cw * hid; o += 256) {     // dW2[c][hj] = sum_g dz2[g,c] * hbuf[g,hj]
    int c = o / hid, hj = o % hid;
    float acc = 0.f;
    for (int gi = 0; gi < G; ++gi) acc = fmaf(dz2[gi * cw + c], hbuf[gi * hid + hj], acc);
    pW2[o] = acc;
  }
  for (int c = threadIdx.x; c < cw; c += 256) {
    float acc = 0.f;
    for (int gi = 0; gi < G; ++gi) acc += dz2[gi * cw + c];
    pb2[c] = acc;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < P1; o += 256) {           // dW1[hj][k] = sum_g dz1[g,hj] * v_g[k]
    int hj = o / (5 * cw), k = o % (5 * cw);
    float acc = 0.f;
    for (int gi = 0; gi < G; ++gi) {
      float v = k < cw ? pooled[(gi * cw + k) * 5] : pooled[(gi * cw + (k - cw) / 4) * 5 + 1 + (k - cw) % 4];
      acc = fmaf(dz1[gi * hid + hj], v, acc);
    }
    pW1[o] = acc;
  }
  for (int hj = threadIdx.x; hj < hid; hj += 256) {
    float acc = 0.f;
    for (int gi = 0; gi < G; ++gi) acc += dz1[gi * hid + hj];
    pb1[hj] = acc;
  }
  for (int i = threadIdx.x; i < C * 5; i += 256) {        // d pooled mean (c, k): sum_hj W1[hj][idx(k)] * dz1[g,hj]
    int cfull = i / 5, k = i % 5, gi = cfull / cw, c = cfull % cw;
    int col = k == 0 ? c : cw + c * 4 + (k - 1);
    float acc = 0.f;
    for (int hj = 0; hj < hid; ++hj) acc = fmaf(w1[(long)hj * 5 * cw + col], dz1[gi * hid + hj], acc);
    dpooled[(long)n * C * 5 + i] = acc / cnt[k];          // per-pixel share of the mean
  }
}

__global__ void spr_param_reduce_kernel(const float* pimg, int N, int P, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += pimg[(long)n * P + i];
  out[i] = s;
}

// gx[n,h,w,c] = gy*attn[n,c] + dpooled share of every pooling bin the pixel belongs to
template <typename T>
__global__ void spr_out_bwd_kernel(const mgdt_view gy, const float* attn, const float* dpooled, const mgdt_view gx) {
  long total = (long)gx.n * gx.h * gx.w * gx.c;
  const int H = gx.h, W = gx.w;
  const int hs1 = bin_start(1, H, 2), he0 = bin_end(0, H, 2), ws1 = bin_start(1, W, 2), we0 = bin_end(0, W, 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, gx, n, h, w, c)
    const float* dp = dpooled + ((long)n * gx.c + c) * 5;
    bool t0 = h < he0, t1 = h >= hs1, l0 = w < we0, l1 = w >= ws1;
    float v = (float)AT(const T, gy, n, h, w, c) * attn[(long)n * gx.c + c] + dp[0];
    if (t0 && l0) v += dp[1];
    if (t0 && l1) v += dp[2];
    if (t1 && l0) v += dp[3];
    if (t1 && l1) v += dp[4];
    AT(T, gx, n, h, w, c) = (T)v;
  }
}

// P = number of SPR parameters = hid*5cw + hid + cw*hid + cw; param_grads laid out [dW1 | db1 | dW2 | db2]
extern "C" size_t mgdt_spr_bwd_workspace_bytes(int n, int c, int groups) {
  int cw = c / groups, hid = cw / 4;
  size_t P = (size_t)hid * 5 * cw + hid + (size_t)cw * hid + cw;
  return ((size_t)n * c * 5 + (size_t)n * P) * sizeof(float);
}
extern "C" int mgdt_spr_bwd(const mgdt_view* gy, const float* pooled_partial, int splits, const float* attn, const float* dattn,
                            const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int groups,
                            const mgdt_view* gx, float* param_grads, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(gy) || !view_ok(gx) || !pooled_partial || !attn || !dattn || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !param_grads || !ws)
    MGDT_FAIL(MGDT_BAD_ARG, "spr_bwd: null/empty argument");
  const int C = gx->c, N = gx->n;
  if (gy->sc != 1 || gx->sc != 1 || gy->c != C || gy->n != N || gy->h != gx->h || gy->w != gx->w || groups < 1 || C % groups || (C / groups) % 4)
    MGDT_FAIL(MGDT_BAD_SHAPE, "spr_bwd: matching NHWC views, c %% (4*groups) == 0");
  const int cw = C / groups, hid = cw / 4;
  const int P = hid * 5 * cw + hid + cw * hid + cw;
  float* dpooled = (float*)ws;
  float* pimg = dpooled + (size_t)N * C * 5;
  size_t lds = (size_t)(C * 5 + groups * hid + 3 * C + groups * hid) * sizeof(float);
  hipStream_t st = (hipStream_t)s;
  spr_attn_bwd_kernel<<<N, 256, lds, st>>>(pooled_partial, fc1_w, fc1_b, fc2_w, fc2_b, C, groups, gx->h, gx->w, splits, dattn, dpooled, pimg);
  spr_param_reduce_kernel<<<cdiv(P, 256), 256, 0, st>>>(pimg, N, P, param_grads);
  long total = (long)N * gx->h * gx->w * C;
  if (!mgdt_v4_spr_out_bwd(gy, attn, dpooled, gx, dtype, st))
    MGDT_DISPATCH_DTYPE(dtype, (spr_out_bwd_kernel<T><<<ew_grid(total), 256, 0, st>>>(*gy, attn, dpooled, *gx)));
  MGDT_CHECK_LAUNCH("spr_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ ConvNeXtV2: LayerNorm + dw7x7 backward
// y = LN(u) * lw + lb, u = dwconv7(x) + db.  Pass 1 (per pixel): du = LN backward, also per-block partials of dlw, dlb.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const mgdt_view u, const mgdt_view gy, const float* lw, float eps, const mgdt_view du,
                                                     float* part /* [nblk][2][C] */) {
  // one wave (64 lanes) per pixel, lanes stride the channels; 4 pixels per block
  const int C = u.c, lane = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const long M = (long)u.n * u.h * u.w;
  extern __shared__ float sm[];      // [4][2][C] per-pixel-lane partial dlw/dlb
  float* mylw = sm + (pl * 2) * C;
  float* mylb = mylw + C;
  for (int c = lane; c < C; c += 64) { mylw[c] = 0.f; mylb[c] = 0.f; }
  for (long m = blockIdx.x * 4L + pl; m < M; m += gridDim.x * 4L) {
    long n = m / ((long)u.h * u.w), rem = m - n * (long)u.h * u.w;
    int h = (int)(rem / u.w), w = (int)(rem - (long)h * u.w);
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < C; c += 64) { float v = (float)AT(const T, u, n, h, w, c); s += v; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float mean = s / (float)C;
    for (int c = lane; c < C; c += 64) { float d = (float)AT(const T, u, n, h, w, c) - mean; ss += d * d; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    float rstd = 1.f / sqrtf(ss / (float)C + eps);
    float a = 0.f, b = 0.f;   // sum g*lw, sum g*lw*xhat
    for (int c = lane; c < C; c += 64) {
      float xh = ((float)AT(const T, u, n, h, w, c) - mean) * rstd, g = (float)AT(const T, gy, n, h, w, c);
      mylw[c] += g * xh;
      mylb[c] += g;
      a += g * lw[c];
      b += g * lw[c] * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    for (int c = lane; c < C; c += 64) {
      float xh = ((float)AT(const T, u, n, h, w, c) - mean) * rstd, g = (float)AT(const T, gy, n, h, w, c);
      AT(T, du, n, h, w, c) = (T)(rstd * (g * lw[c] - a / (float)C - xh * b / (float)C));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    int k = i / C, c = i % C;
    float t = 0.f;
    for (int p = 0; p < 4; ++p) t += sm[(p * 2 + k) * C + c];
    part[((long)blockIdx.x * 2 + k) * C + c] = t;
  }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int rows, int cols, float* out0, float* out1) {
  // part[rows][2][cols] -> out0[c] = sum rows part[r][0][c], out1[c] = sum rows part[r][1][c].  16 columns per workgroup, 16 threads per column
  // adding rows part, part+16, ... (independent loads), then the 16 sub-sums in order: fixed association, no 1024-deep dependent chain.
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, pt = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  const int k = i / cols, c = i - k * cols;
  float a = 0.f;
  if (i < 2 * cols) {
#pragma unroll 8
    for (int r = pt; r < rows; r += 16) a += part[((long)r * 2 + k) * cols + c];
  }
  red[pt][cl] = a;
  __syncthreads();
  if (pt == 0 && i < 2 * cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    (k == 0 ? out0 : out1)[c] = t;
  }
}

// depthwise 7x7: dx = corr(du, w flipped) ; dw[tap][c] = sum_pix du[p] * x[p+tap] ; db[c] = sum du
template <typename T>
__global__ void dwconv7_dgrad_kernel(const mgdt_view du, const float* __restrict__ w49c, const mgdt_view dx, int accumulate) {
  long total = (long)dx.n * dx.h * dx.w * dx.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, dx, n, h, w, c)
    float acc = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
      int oy = h + 3 - ky;
      if ((unsigned)oy >= (unsigned)du.h) continue;
      for (int kx = 0; kx < 7; ++kx) {
        int ox = w + 3 - kx;
        if ((unsigned)ox >= (unsigned)du.w) continue;
        acc = fmaf((float)AT(const T, du, n, oy, ox, c), w49c[(ky * 7 + kx) * dx.c + c], acc);
      }
    }
    if (accumulate) acc += (float)AT(const T, dx, n, h, w, c);
    AT(T, dx, n, h, w, c) = (T)acc;
  }
}
#define DW_SPLITS 32
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const mgdt_view x, const mgdt_view du, float* part /* [split][50][C] */) {
  // block = (tap 0..48 or 49 = bias, split); 256 threads = 64 channels x 4 pixel lanes; grid.z = channel blocks
  const int tap = blockIdx.x, split = blockIdx.y, c = blockIdx.z * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const long M = (long)du.n * du.h * du.w, HW = (long)du.h * du.w;
  const long p0 = split * M / DW_SPLITS, p1 = (split + 1) * M / DW_SPLITS;
  const int ky = tap / 7, kx = tap % 7;
  float acc = 0.f;
  if (c < du.c)
    for (long p = p0 + pl; p < p1; p += 4) {
      long n = p / HW, rem = p - n * HW;
      int oy = (int)(rem / du.w), ox = (int)(rem - (long)oy * du.w);
      float g = (float)AT(const T, du, n, oy, ox, c);
      if (tap == 49) { acc += g; continue; }
      int iy = oy + ky - 3, ix = ox + kx - 3;
      if ((unsigned)iy >= (unsigned)x.h || (unsigned)ix >= (unsigned)x.w) continue;
      acc = fmaf(g, (float)AT(const T, x, n, iy, ix, c), acc);
    }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && c < du.c)
    part[((long)split * 50 + tap) * du.c + c] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
}
__global__ void dw_final_kernel(const float* part, int C, float* dw_c49, float* db) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 50 * C) return;
  int tap = i / C, c = i % C;
  float s = 0.f;
  for (int k = 0; k < DW_SPLITS; ++k) s += part[((long)k * 50 + tap) * C + c];
  if (tap == 49) db[c] = s;
  else dw_c49[(long)c * 49 + tap] = s;      // nn.Conv2d depthwise weight layout (C,1,7,7)
}


// ---- channel-vectorised versions (C % 4 == 0, 8/16-byte aligned views): 4 channels per thread, 8-/16-byte accesses ----
#define AT4(T, v, n, h, w, c) ((T*)(v).p + ((n) * (v).sn + (h) * (v).sh + (w) * (v).sw + (c)))
// LayerNorm backward: 8 lanes per pixel, lane l owns channel quads l, l+8, ... (NQ of them, values kept in registers: one read of u and gy);
// the dlw / dlb partial sums stay in registers across the thread's pixels and are combined once per workgroup.
template <typename T, int NQ>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const mgdt_view u, const mgdt_view gy, const float* __restrict__ lw, float eps, const mgdt_view du,
                                                         float* part /* [nblk][2][C] */) {
  const int C = u.c, Q = C >> 2, l = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const long M = (long)u.n * u.h * u.w, HW = (long)u.h * u.w;
  f32x4 wq[NQ], alw[NQ], alb[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int q = l + 8 * k;
    wq[k] = q < Q ? *(const f32x4*)(lw + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    alw[k] = alb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invC = 1.f / (float)C;
  for (long m0 = blockIdx.x * 32L; m0 < M; m0 += gridDim.x * 32L) {        // uniform trip count per wave: shuffles below need all lanes
    const long m = m0 + pl;
    const bool ok = m < M;
    const long mm = ok ? m : 0;
    const long n = mm / HW, rem = mm - n * HW;
    const int h = (int)(rem / u.w), w = (int)(rem - (long)h * u.w);
    f32x4 uv[NQ], gv[NQ];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int q = l + 8 * k;
      const bool on = ok && q < Q;
      uv[k] = on ? load4<T>(AT4(const T, u, n, h, w, 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
      gv[k] = on ? load4<T>(AT4(const T, gy, n, h, w, 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (uv[k][0] + uv[k][1]) + (uv[k][2] + uv[k][3]);
    }
    for (int o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * invC;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (l + 8 * k < Q)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = uv[k][j] - mean; ss += d * d; }
    for (int o = 1; o < 8; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.f / sqrtf(ss * invC + eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (l + 8 * k < Q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (uv[k][j] - mean) * rstd, g = gv[k][j];
          uv[k][j] = xh;
          alw[k][j] += g * xh;
          alb[k][j] += g;
          a += g * wq[k][j];
          b += g * wq[k][j] * xh;
        }
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int q = l + 8 * k;
      if (ok && q < Q) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (gv[k][j] * wq[k][j] - a * invC - uv[k][j] * b * invC);
        store4<T>(AT4(T, du, n, h, w, 4 * q), o);
      }
    }
  }
  // combine the 32 pixel lanes: LDS [32][2][C] would be 24 KB at C = 96; go through it in two halves of the k index instead
  extern __shared__ float sm[];      // [32][C]
  for (int kk = 0; kk < 2; ++kk) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int q = l + 8 * k;
      if (q < Q) *(f32x4*)(sm + (long)pl * C + 4 * q) = kk == 0 ? alw[k] : alb[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int p = 0; p < 32; ++p) t += sm[p * C + c];
      part[((long)blockIdx.x * 2 + kk) * C + c] = t;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv7_dgrad_vec_kernel(const mgdt_view du, const float* __restrict__ w49c, const mgdt_view dx, int accumulate) {
  const int Q = dx.c >> 2;
  const long total = (long)dx.n * dx.h * dx.w * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % Q);
    long t = i / Q;
    const int w = (int)(t % dx.w);
    t /= dx.w;
    const int h = (int)(t % dx.h);
    const long n = t / dx.h;
    // branch-free taps: out-of-range positions read a clamped (valid) address and are multiplied by 0 - with `continue` around the loads every load
    // waited for the previous bounds test and the 49 taps ran as one dependent chain
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int oy = h + 3 - ky;
      const bool vy = (unsigned)oy < (unsigned)du.h;
      const int oyc = min(max(oy, 0), du.h - 1);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int ox = w + 3 - kx;
        const float m = (vy && (unsigned)ox < (unsigned)du.w) ? 1.f : 0.f;
        const int oxc = min(max(ox, 0), du.w - 1);
        const f32x4 g = load4<T>(AT4(const T, du, n, oyc, oxc, 4 * q));
        const f32x4 wv = *(const f32x4*)(w49c + (ky * 7 + kx) * dx.c + 4 * q);
        acc += g * wv * m;
      }
    }
    if (accumulate) acc += load4<T>(AT4(const T, dx, n, h, w, 4 * q));
    store4<T>(AT4(T, dx, n, h, w, 4 * q), acc);
  }
}

// depthwise weight gradient: block = (ky, split); thread = (channel quad, pixel lane) holds the 7 kx taps x 4 channels of its row ky in
// registers (du read once per 7 taps), bias sum rides with ky == 0.  part[split][50][C].
#define DWV_SPLITS 320
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_vec_kernel(const mgdt_view x, const mgdt_view du, float* part) {
  const int C = du.c, Q = C >> 2, ky = blockIdx.x, split = blockIdx.y;
  const int QB = Q < 64 ? Q : 64;                       // quads per pass over the channels (grid.z covers the rest)
  const int q = blockIdx.z * QB + threadIdx.x % QB, pl = threadIdx.x / QB, PL = 256 / QB;
  const long M = (long)du.n * du.h * du.w, HW = (long)du.h * du.w;
  const long p0 = split * M / DWV_SPLITS, p1 = (split + 1) * M / DWV_SPLITS;
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q < Q && pl < PL)
    for (long p = p0 + pl; p < p1; p += PL) {
      const long n = p / HW, rem = p - n * HW;
      const int oy = (int)(rem / du.w), ox = (int)(rem - (long)oy * du.w);
      const f32x4 g = load4<T>(AT4(const T, du, n, oy, ox, 4 * q));
      acc[7] += g;
      const int iy = oy + ky - 3;
      const bool vy = (unsigned)iy < (unsigned)x.h;
      const int iyc = min(max(iy, 0), x.h - 1);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {                   // branch-free: clamped address, zero weight outside the image
        const int ix = ox + kx - 3;
        const float m = (vy && (unsigned)ix < (unsigned)x.w) ? 1.f : 0.f;
        const int ixc = min(max(ix, 0), x.w - 1);
        acc[kx] += g * load4<T>(AT4(const T, x, n, iyc, ixc, 4 * q)) * m;
      }
    }
  __shared__ float red[8][4][256];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[k][j][threadIdx.x] = acc[k][j];
  __syncthreads();
  // (tap k, channel) pairs of this block: 8 x 4*QB
  for (int o = threadIdx.x; o < 8 * 4 * QB; o += 256) {
    const int k = o / (4 * QB), cc = o - k * 4 * QB, ql = cc >> 2, j = cc & 3;
    const int c = (blockIdx.z * QB + ql) * 4 + j;
    if (c >= C || (k == 7 && ky != 0)) continue;
    float t = 0.f;
    for (int z = 0; z < PL; ++z) t += red[k][j][z * QB + ql];
    const int tap = k == 7 ? 49 : ky * 7 + k;
    part[((long)split * 50 + tap) * C + c] = t;
  }
}
__global__ __launch_bounds__(256) void dw_final_vec_kernel(const float* __restrict__ part, int C, int nsplit, float* dw_c49, float* db) {
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, pt = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  const int tap = i / C, c = i - tap * C;
  float a = 0.f;
  if (i < 50 * C) {
#pragma unroll 10
    for (int k = pt; k < nsplit; k += 16) a += part[((long)k * 50 + tap) * C + c];
  }
  red[pt][cl] = a;
  __syncthreads();
  if (pt == 0 && i < 50 * C) {
    float t = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) t += red[z][cl];
    if (tap == 49) db[c] = t;
    else dw_c49[(long)c * 49 + tap] = t;
  }
}
static bool vec4_ok(const mgdt_view* v, int dtype) {
  return v->sc == 1 && v->c % 4 == 0 && v->sw % 4 == 0 && v->sh % 4 == 0 && v->sn % 4 == 0 && (uintptr_t)v->p % (4 * dtype_size(dtype)) == 0;
}

extern "C" size_t mgdt_dwconv7_ln_bwd_workspace_bytes(int c) { return (size_t)(1024 * 2 * c + DWV_SPLITS * 50 * c) * sizeof(float); }
// x: block input, u: dwconv output (pre-LN, saved by the forward), gy: grad of the LN output.  Produces dx (written or
// accumulated), d dw weight (C,1,7,7), d dw bias, d ln weight/bias.  du_tmp: caller-provided NHWC scratch like u.
extern "C" int mgdt_dwconv7_ln_bwd(const mgdt_view* x, const mgdt_view* u, const mgdt_view* gy, const float* dw_w49c, const float* ln_w, float eps,
                                   const mgdt_view* du_tmp, const mgdt_view* dx, int accumulate_dx, float* d_dw_w, float* d_dw_b,
                                   float* d_ln_w, float* d_ln_b, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(u) || !view_ok(gy) || !view_ok(du_tmp) || !view_ok(dx) || !dw_w49c || !ln_w || !d_dw_w || !d_dw_b || !d_ln_w || !d_ln_b || !ws)
    MGDT_FAIL(MGDT_BAD_ARG, "dwconv7_ln_bwd: null/empty argument");
  const int C = x->c;
  for (const mgdt_view* v : {x, u, gy, du_tmp, dx})
    if (v->sc != 1 || v->c != C || v->n != x->n || v->h != x->h || v->w != x->w) MGDT_FAIL(MGDT_BAD_SHAPE, "dwconv7_ln_bwd: matching NHWC views");
  hipStream_t st = (hipStream_t)s;
  const long M = (long)x->n * x->h * x->w;
  const int nblk = (int)std::min<long>((M + 3) / 4, 1024);
  float* part = (float*)ws;
  float* part2 = part + (size_t)1024 * 2 * C;
  bool vec = C <= 128;
  for (const mgdt_view* v : {x, u, gy, du_tmp, dx}) vec = vec && vec4_ok(v, dtype);
  long total = M * C;
  if (vec) {
    const int nb2 = (int)std::min<long>((M + 31) / 32, 1024);
    const int nq = cdiv(C / 4, 8);
#define LN_VEC(NQ) MGDT_DISPATCH_DTYPE(dtype, (ln_bwd_vec_kernel<T, NQ><<<nb2, 256, (size_t)32 * C * sizeof(float), st>>>(*u, *gy, ln_w, eps, *du_tmp, part)))
    if (nq == 1) LN_VEC(1); else if (nq == 2) LN_VEC(2); else if (nq == 3) LN_VEC(3); else LN_VEC(4);
#undef LN_VEC
    colsum_kernel<<<cdiv(2 * C, 16), 256, 0, st>>>(part, nb2, C, d_ln_w, d_ln_b);
    MGDT_DISPATCH_DTYPE(dtype, (dwconv7_dgrad_vec_kernel<T><<<ew_grid(total / 4), 256, 0, st>>>(*du_tmp, dw_w49c, *dx, accumulate_dx)));
    const int qb = std::min(C / 4, 64);
    dim3 g(7, DWV_SPLITS, cdiv(C / 4, qb));
    MGDT_DISPATCH_DTYPE(dtype, (dwconv7_wgrad_vec_kernel<T><<<g, 256, 0, st>>>(*x, *du_tmp, part2)));
    dw_final_vec_kernel<<<cdiv(50 * C, 16), 256, 0, st>>>(part2, C, DWV_SPLITS, d_dw_w, d_dw_b);
    MGDT_CHECK_LAUNCH("dwconv7_ln_bwd");
    return MGDT_OK;
  }
  MGDT_DISPATCH_DTYPE(dtype, (ln_bwd_kernel<T><<<nblk, 256, (size_t)8 * C * sizeof(float), st>>>(*u, *gy, ln_w, eps, *du_tmp, part)));
  colsum_kernel<<<cdiv(2 * C, 16), 256, 0, st>>>(part, nblk, C, d_ln_w, d_ln_b);
  MGDT_DISPATCH_DTYPE(dtype, (dwconv7_dgrad_kernel<T><<<ew_grid(total), 256, 0, st>>>(*du_tmp, dw_w49c, *dx, accumulate_dx)));
  dim3 g(50, DW_SPLITS, cdiv(C, 64));
  MGDT_DISPATCH_DTYPE(dtype, (dwconv7_wgrad_kernel<T><<<g, 256, 0, st>>>(*x, *du_tmp, part2)));
  dw_final_kernel<<<cdiv(50 * C, 256), 256, 0, st>>>(part2, C, d_dw_w, d_dw_b);
  MGDT_CHECK_LAUNCH("dwconv7_ln_bwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ GRN backward
// out = t*(gamma*Nx + 1) + beta,  Nx[n,c] = Gx/(mean_c Gx + 1e-6),  Gx = sqrt(sum_hw t^2).
// A[n,c] = sum_hw g*t (from mgdt_nc_reduce), S[n,c] = sum_hw t^2 (the forward's statistic), B[n,c] = sum_hw g.
//   dgamma[c] = sum_n Nx*A ; dbeta[c] = sum_n B ; coef[n,c] = dGx/Gx with dGx = dNx/(m+eps) - (1/C) sum_c' dNx*Gx/(m+eps)^2, dNx = gamma*A
//   dt = g*(gamma*Nx+1) + coef*t
__global__ __launch_bounds__(256) void grn_bwd_small_kernel(const float* __restrict__ S, const float* __restrict__ A, const float* __restrict__ gamma,
                                                            int C, float* __restrict__ scale, float* __restrict__ coef, float* __restrict__ nx_a) {
  const int n = blockIdx.x;
  __shared__ float r0[256], r1[256];
  float p0 = 0.f, p1 = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float gx = sqrtf(S[(long)n * C + c]);
    p0 += gx;
  }
  r0[threadIdx.x] = p0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) r0[threadIdx.x] += r0[threadIdx.x + o]; __syncthreads(); }
  const float m = r0[0] / (float)C, den = m + 1e-6f;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float gx = sqrtf(S[(long)n * C + c]);
    p1 += gamma[c] * A[(long)n * C + c] * gx;
  }
  r1[threadIdx.x] = p1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) r1[threadIdx.x] += r1[threadIdx.x + o]; __syncthreads(); }
  const float corr = r1[0] / ((float)C * den * den);
  for (int c = threadIdx.x; c < C; c += 256) {
    float gx = sqrtf(S[(long)n * C + c]), nx = gx / den;
    float dnx = gamma[c] * A[(long)n * C + c];
    float dgx = dnx / den - corr;
    scale[(long)n * C + c] = gamma[c] * nx + 1.f;
    coef[(long)n * C + c] = gx > 0.f ? dgx / gx : 0.f;
    nx_a[(long)n * C + c] = nx * A[(long)n * C + c];
  }
}
__global__ void rowsum2_kernel(const float* a, const float* b, int N, int C, float* oa, float* ob) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sa = 0.f, sb = 0.f;
  for (int n = 0; n < N; ++n) { sa += a[(long)n * C + c]; sb += b[(long)n * C + c]; }
  oa[c] = sa;
  ob[c] = sb;
}
template <typename T>
__global__ void grn_bwd_apply_kernel(const mgdt_view g, const mgdt_view t, const float* scale, const float* coef, const mgdt_view dt) {
  long total = (long)t.n * t.h * t.w * t.c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    DECODE_NHWC(i, t, n, h, w, c)
    float v = (float)AT(const T, g, n, h, w, c) * scale[n * t.c + c] + coef[n * t.c + c] * (float)AT(const T, t, n, h, w, c);
    AT(T, dt, n, h, w, c) = (T)v;
  }
}

// ws: 3*n*c floats.  S, A, B: fp32 [n][c] from mgdt_nc_reduce (t*t, g*t, g).
extern "C" int mgdt_grn_bwd(const mgdt_view* g, const mgdt_view* t, const float* S, const float* A, const float* B, const float* gamma,
                            const mgdt_view* dt, float* dgamma, float* dbeta, void* ws, int dtype, mgdt_stream s) {
  if (!view_ok(g) || !view_ok(t) || !view_ok(dt) || !S || !A || !B || !gamma || !dgamma || !dbeta || !ws) MGDT_FAIL(MGDT_BAD_ARG, "grn_bwd: null/empty argument");
  const int C = t->c, N = t->n;
  for (const mgdt_view* v : {g, t, dt})
    if (v->sc != 1 || v->c != C || v->n != N || v->h != t->h || v->w != t->w) MGDT_FAIL(MGDT_BAD_SHAPE, "grn_bwd: matching NHWC views");
  hipStream_t st = (hipStream_t)s;
  float* scale = (float*)ws;
  float* coef = scale + (size_t)N * C;
  float* nxa = coef + (size_t)N * C;
  grn_bwd_small_kernel<<<N, 256, 0, st>>>(S, A, gamma, C, scale, coef, nxa);
  rowsum2_kernel<<<cdiv(C, 256), 256, 0, st>>>(nxa, B, N, C, dgamma, dbeta);
  long total = (long)N * t->h * t->w * C;
  if (!mgdt_v4_grn_bwd_apply(g, t, scale, coef, dt, dtype, st))
    MGDT_DISPATCH_DTYPE(dtype, (grn_bwd_apply_kernel<T><<<ew_grid(total), 256, 0, st>>>(*g, *t, scale, coef, *dt)));
  MGDT_CHECK_LAUNCH("grn_bwd");
  return MGDT_OK;
}
