// HBM-bound companions of the convolutions: pooling-attention (SPR), SPPF max-pools, GD-neck resamplers,
// ConvNeXtV2 depthwise+LayerNorm, GRN statistics, the Injection tail and the Detect decode.
// All work on NHWC views, 4 channels (16 B fp32 / 8 B bf16) per lane, channel-fastest so that a wave touches
// whole contiguous lines.
#include <algorithm>
#include <vector>

#include <type_traits>
#include "common.h"

#define SPR_SPLITS 16

static inline int grid_for(long work, int block = 256, int cap = 8192) { return (int)std::min<long>((work + block - 1) / block, cap); }

// flat index -> (n, h, w, channel-vector q) with exact magic-number division (64-bit '/' and '%' cost ~100 instructions each)
struct PixIdx { FastDiv fq, fw, fh; int Q, W, H; };
static inline PixIdx make_pixidx(int H, int W, int Q) { return PixIdx{make_fastdiv((uint32_t)Q), make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)H), Q, W, H}; }
__device__ __forceinline__ void decode_idx(uint32_t i, const PixIdx& d, int& n, int& h, int& w, int& q) {
  uint32_t t = fdiv(i, d.fq);
  q = (int)(i - t * d.Q);
  uint32_t t2 = fdiv(t, d.fw);
  w = (int)(t - t2 * d.W);
  uint32_t t3 = fdiv(t2, d.fh);
  h = (int)(t2 - t3 * d.H);
  n = (int)t3;
}
// V consecutive channels <-> V floats (V = 4: 8/16 bytes, V = 8: 16/32 bytes per lane)
template <typename T, int V> __device__ __forceinline__ void ldv(const T* p, float (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; k += 4) {
    f32x4 t = load4<T>(p + k);
    v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3];
  }
}
template <> __device__ __forceinline__ void ldv<bf16, 8>(const bf16* p, float (&v)[8]) {
  bf16x8 t = *(const bf16x8*)p;
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; k += 4) store4<T>(p + k, f32x4{v[k], v[k + 1], v[k + 2], v[k + 3]});
}
template <> __device__ __forceinline__ void stv<bf16, 8>(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = (bf16)v[k];
  *(bf16x8*)p = t;
}
static inline bool vecN_ok(const mgdt_view* v, int dtype, int V) {
  return v->sc == 1 && v->c % V == 0 && v->sw % V == 0 && v->sh % V == 0 && v->sn % V == 0 && ((uintptr_t)v->p % (V * dtype_size(dtype))) == 0;
}
// dispatch on dtype and vector width: bf16 views that allow it move 16 bytes per lane (V = 8), everything else V = 4
#define MGDT_DISPATCH_TV(dtype, v8ok, ...)                                             \
  do {                                                                                 \
    if ((dtype) == MGDT_F32) { using T = float; constexpr int V = 4; __VA_ARGS__; }    \
    else if ((dtype) == MGDT_BF16 && (v8ok)) { using T = bf16; constexpr int V = 8; __VA_ARGS__; } \
    else if ((dtype) == MGDT_BF16) { using T = bf16; constexpr int V = 4; __VA_ARGS__; } \
    else MGDT_FAIL(MGDT_BAD_DTYPE, "unsupported dtype %d", (int)(dtype));              \
  } while (0)

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
    float v;
    if constexpr (std::is_same<TX, uint8_t>::value) v = (float)x[n * xsn + h * xsh + w * xsw + c * xsc] / 255.0f;   // the preprocess division, exact
    else v = (float)x[n * xsn + h * xsh + w * xsw + c * xsc];
    y[n * ysn + h * ysh + w * ysw + c * ysc] = (TY)v;
  }
}

template <typename TX, typename TY, int V>
__global__ void copy_vec_kernel(const TX* __restrict__ x, long xsn, long xsh, long xsw, TY* __restrict__ y, long ysn, long ysh, long ysw,
                                uint32_t total, PixIdx d) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, h, w, q;
    decode_idx(i, d, n, h, w, q);
    float v[V];
    ldv<TX, V>(x + n * xsn + h * xsh + w * xsw + q * V, v);
    stv<TY, V>(y + n * ysn + h * ysh + w * ysw + q * V, v);
  }
}

// The 3-channel image (any layout: NCHW planes as the dataloader hands it; uint8 is divided by 255 as detect/train.py:64 does) as a 4-channel NHWC map with
// a zero 4th channel: what the weight-gradient kernels of the stem read.  One thread per pixel: three coalesced plane reads, one 8 / 16-byte store.
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void image_pad4_kernel(const TX* __restrict__ x, long xsn, long xsh, long xsw, long xsc, TY* __restrict__ y, long ysn, long ysh,
                                                         long ysw, int N, int H, int W, int C) {
  const long total = (long)N * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % W);
    long t = i / W;
    const int h = (int)(t % H);
    const long n = t / H;
    const TX* xp = x + n * xsn + h * xsh + w * xsw;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < C) v[c] = std::is_same<TX, uint8_t>::value ? (float)xp[c * xsc] / 255.f : (float)xp[c * xsc];
    store4<TY>(y + n * ysn + h * ysh + w * ysw, v);
  }
}
extern "C" int mgdt_image_pad4_fwd(const mgdt_view* x, int xdt, const mgdt_view* y, int ydt, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "image_pad4: null/empty view");
  if (x->c < 1 || x->c > 3 || y->c != 4 || x->n != y->n || x->h != y->h || x->w != y->w || y->sc != 1 || y->sw % 4 || y->sh % 4 || y->sn % 4 ||
      (uintptr_t)y->p % (4 * dtype_size(ydt)))
    MGDT_FAIL(MGDT_BAD_SHAPE, "image_pad4: x with 1-3 channels, y a 4-channel NHWC view of the same size");
  const long total = (long)x->n * x->h * x->w;
  const int g = (int)std::min<long>((total + 255) / 256, 16384);
  hipStream_t st = (hipStream_t)s;
#define LP(TX, TY) image_pad4_kernel<TX, TY><<<g, 256, 0, st>>>((const TX*)x->p, x->sn, x->sh, x->sw, x->sc, (TY*)y->p, y->sn, y->sh, y->sw, x->n, x->h, x->w, x->c)
  if (xdt == MGDT_U8 && ydt == MGDT_BF16) LP(uint8_t, bf16);
  else if (xdt == MGDT_U8 && ydt == MGDT_F32) LP(uint8_t, float);
  else if (xdt == MGDT_F32 && ydt == MGDT_F32) LP(float, float);
  else if (xdt == MGDT_F32 && ydt == MGDT_BF16) LP(float, bf16);
  else if (xdt == MGDT_BF16 && ydt == MGDT_BF16) LP(bf16, bf16);
  else MGDT_FAIL(MGDT_BAD_DTYPE, "image_pad4: dtypes %d -> %d", xdt, ydt);
#undef LP
  MGDT_CHECK_LAUNCH("image_pad4_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_copy_fwd(const mgdt_view* x, int xdt, const mgdt_view* y, int ydt, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "copy: null/empty view");
  if (x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "copy: shape mismatch");
  long total = (long)x->n * x->h * x->w * x->c;
  hipStream_t st = (hipStream_t)s;
  if (total < 0x7fffffffL && xdt != MGDT_U8 && vecN_ok(x, xdt, 8) && vecN_ok(y, ydt, 8)) {   // NHWC both sides: 8 channels per lane
    PixIdx d = make_pixidx(x->h, x->w, x->c / 8);
    uint32_t tv = (uint32_t)(total / 8);
    int g = grid_for(tv);
#define LV(TX, TY) copy_vec_kernel<TX, TY, 8><<<g, 256, 0, st>>>((const TX*)x->p, x->sn, x->sh, x->sw, (TY*)y->p, y->sn, y->sh, y->sw, tv, d)
    if (xdt == MGDT_F32 && ydt == MGDT_F32) LV(float, float);
    else if (xdt == MGDT_F32 && ydt == MGDT_BF16) LV(float, bf16);
    else if (xdt == MGDT_BF16 && ydt == MGDT_F32) LV(bf16, float);
    else if (xdt == MGDT_BF16 && ydt == MGDT_BF16) LV(bf16, bf16);
    else MGDT_FAIL(MGDT_BAD_DTYPE, "copy: dtypes %d -> %d", xdt, ydt);
#undef LV
    MGDT_CHECK_LAUNCH("copy_fwd");
    return MGDT_OK;
  }
  int g = grid_for(total);
#define L(TX, TY) copy_kernel<TX, TY><<<g, 256, 0, st>>>((const TX*)x->p, x->sn, x->sh, x->sw, x->sc, (TY*)y->p, y->sn, y->sh, y->sw, y->sc, x->n, x->h, x->w, x->c)
  if (xdt == MGDT_F32 && ydt == MGDT_F32) L(float, float);
  else if (xdt == MGDT_F32 && ydt == MGDT_BF16) L(float, bf16);
  else if (xdt == MGDT_BF16 && ydt == MGDT_F32) L(bf16, float);
  else if (xdt == MGDT_BF16 && ydt == MGDT_BF16) L(bf16, bf16);
  else if (xdt == MGDT_U8 && ydt == MGDT_F32) L(uint8_t, float);       // uint8 image / 255 (detect/train.py:64) into a float map
  else if (xdt == MGDT_U8 && ydt == MGDT_BF16) L(uint8_t, bf16);
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
    // pixels prow, prow + PR, ... of the band in raster order; 4 loads are issued before the (in-order) accumulation so that a thread
    // keeps 4 requests in flight instead of one, and (yy, xx) advance incrementally instead of by 64-bit divisions
    const int npix = (r1 - r0) * W;
    int yy = r0 + prow / W, xx = prow % W;
    const int dy = PR / W, dx = PR % W;
    auto advance = [&](int& y_, int& x_) __attribute__((always_inline)) { y_ += dy; x_ += dx; if (x_ >= W) { x_ -= W; ++y_; } };
    auto accum = [&](f32x4 v, int y_, int x_) __attribute__((always_inline)) {
      const bool t0 = y_ < he0, t1 = y_ >= hs1, l0 = x_ < we0, l1 = x_ >= ws1;
      acc[0] += v;
      if (t0 && l0) acc[1] += v;
      if (t0 && l1) acc[2] += v;
      if (t1 && l0) acc[3] += v;
      if (t1 && l1) acc[4] += v;
    };
    const T* xb = x + n * sn + q * 4;
    int p = prow;
    for (; p + 3 * PR < npix; p += 4 * PR) {
      int ys[4], xs[4];
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { ys[k] = yy; xs[k] = xx; v[k] = load4<T>(xb + yy * sh + xx * sw); advance(yy, xx); }
#pragma unroll
      for (int k = 0; k < 4; ++k) accum(v[k], ys[k], xs[k]);
    }
    for (; p < npix; p += PR) { accum(load4<T>(xb + yy * sh + xx * sw), yy, xx); advance(yy, xx); }
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
                                                       int do_softmax, float* __restrict__ attn) {
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
  if (!do_softmax) {                               // the module's standalone form: sigmoid weights (spr_module.py:27-31)
    for (int o = threadIdx.x; o < C; o += 256) attn[(long)n * C + o] = obuf[o];
    return;
  }
  for (int c = threadIdx.x; c < cw; c += 256) {   // softmax over the G groups (block.py:278)
    float mx = -INFINITY;
    for (int gi = 0; gi < G; ++gi) mx = fmaxf(mx, obuf[gi * cw + c]);
    float den = 0.f;
    for (int gi = 0; gi < G; ++gi) den += expf(obuf[gi * cw + c] - mx);
    for (int gi = 0; gi < G; ++gi) attn[(long)n * C + gi * cw + c] = expf(obuf[gi * cw + c] - mx) / den;
  }
}

extern "C" int mgdt_spr_attn_fwd(const float* pooled, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                 const float* fc2_b, int n, int c, int groups, int h, int w, int softmax, float* attn, mgdt_stream s) {
  if (!pooled || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !attn) MGDT_FAIL(MGDT_BAD_ARG, "spr_attn: null pointer");
  if (groups < 1 || c % groups || (c / groups) % 4 || c > 4096 || h < 1 || w < 1) MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn: c=%d groups=%d", c, groups);
  int cw = c / groups, hid = cw / 4;
  size_t lds = (size_t)(c * 5 + groups * hid + c) * sizeof(float);
  spr_attn_kernel<<<n, 256, lds, (hipStream_t)s>>>(pooled, fc1_w, fc1_b, fc2_w, fc2_b, c, groups, h, w, softmax, attn);
  MGDT_CHECK_LAUNCH("spr_attn_fwd");
  return MGDT_OK;
}

// attention MLP + channel scaling in one launch: grid (K, N); every workgroup of image n recomputes the (tiny) SPR MLP of that image
// from the pooled partial sums - identical arithmetic in identical order, so all K copies agree bit for bit - keeps the C softmax
// weights in LDS and scales its 1/K share of the pixels.  Saves a dependent launch per MSPA block.
// optional extra outputs of the scaling pass: F x F average pools of the SCALED map (adaptive_avg_pool2d with H % F == W % F == 0), written
// where the GD neck's SimFusion modules would otherwise launch their own pooling kernels (nn/modules/block.py:289-329)
struct SprPool { void* y; long sn, sh, sw; int F; };

// Scaling pass in FM x FM pixel cells (FM = the largest pooling factor asked for, 2 or 4): a thread owns one cell x V channels, has all its
// FM*FM vectors in flight at once, scales / rounds / stores them in row-major order and accumulates the pooled sums on the way - the FM x FM
// pool of the cell and (FM = 4) its four 2 x 2 pools, each in the row-major order of mgdt_adaptive_avgpool_fwd, so the pooled maps are
// bit-equal to pooling the stored map.  No second pass over the map (the re-reading form cost as much as the pooling launches it replaced).
template <typename T, int V, int FM>
__device__ __forceinline__ void spr_scale_cells(const T* __restrict__ x, long xsn, long xsh, long xsw, T* __restrict__ y, long ysn, long ysh, long ysw,
                                                const float* att, int n, int H, int W, int C, SprPool big, SprPool small) {
  const int Q = C / V, Hc = H / FM, Wc = W / FM, NC = Hc * Wc;
  const int c0 = (int)((long)blockIdx.x * NC / gridDim.x), c1 = (int)((long)(blockIdx.x + 1) * NC / gridDim.x);
  for (int j = threadIdx.x; j < (c1 - c0) * Q; j += 256) {
    const int cl = j / Q, q = j - cl * Q, cell = c0 + cl, cy = cell / Wc, cx = cell - cy * Wc;
    float a[V];
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] = att[q * V + k];
    float v[FM * FM][V];
#pragma unroll
    for (int e = 0; e < FM * FM; ++e) ldv<T, V>(x + n * xsn + (long)(cy * FM + e / FM) * xsh + (long)(cx * FM + e % FM) * xsw + q * V, v[e]);
    float accb[V], accs[4][V];
#pragma unroll
    for (int k = 0; k < V; ++k) { accb[k] = 0.f; accs[0][k] = accs[1][k] = accs[2][k] = accs[3][k] = 0.f; }
#pragma unroll
    for (int e = 0; e < FM * FM; ++e) {
      const int yy = e / FM, xx = e % FM;
#pragma unroll
      for (int k = 0; k < V; ++k) v[e][k] = (float)(T)(v[e][k] * a[k]);              // the value the map stores
      stv<T, V>(y + n * ysn + (long)(cy * FM + yy) * ysh + (long)(cx * FM + xx) * ysw + q * V, v[e]);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        accb[k] += v[e][k];
        if (FM == 4) accs[(yy >> 1) * 2 + (xx >> 1)][k] += v[e][k];
      }
    }
    if (big.F) {
      const float inv = 1.f / (float)(FM * FM);
#pragma unroll
      for (int k = 0; k < V; ++k) accb[k] *= inv;
      stv<T, V>((T*)big.y + n * big.sn + cy * big.sh + cx * big.sw + q * V, accb);
    }
    if (FM == 4 && small.F) {
#pragma unroll
      for (int sc = 0; sc < 4; ++sc) {
#pragma unroll
        for (int k = 0; k < V; ++k) accs[sc][k] *= 0.25f;
        stv<T, V>((T*)small.y + n * small.sn + (long)(cy * 2 + (sc >> 1)) * small.sh + (long)(cx * 2 + (sc & 1)) * small.sw + q * V, accs[sc]);
      }
    }
  }
}

// Latency-lean prologue of spr_attn_scale_kernel (C <= 256, per-tile sums, fc weights small enough for LDS): every workgroup of the image repeats
// it, so its length is a floor under the launch.  The first form walked the fc weights in global memory inside the dot products (one dependent
// load per multiply-add) and reduced the partial sums four loads at a time: 13 us of the 17 us launch at 20x20x256 were prologue (phase
// stamps, MGDT_SPR_DBG).  Here every global load - partial sums, both weight matrices, both biases - is issued before the first one is
// consumed, the matrices go to LDS as padded rows, and fc1 reads them as float4 with `tpo` threads per output.
// LDS: pvec [G][5 cw] (means in fc1's column order: cw whole-map means, then 4 bin means per channel) | hbuf [G hid] | obuf [C] | att [C] |
//      w1s [hid][5 cw + 4] | w2s [cw][hid + 1] | b1s [hid] | b2s [cw]
__device__ __forceinline__ void spr_prologue_fast(const float* __restrict__ partial, const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float* __restrict__ w2, const float* __restrict__ b2, int C, int G, int n, int nsplit, int tiles_x,
                                                  int tiles_y, FastDiv fd_gst, FastDiv fd_tx, FastDiv fd_l4, const float* cnt, float* sm, float* red, float* att, unsigned long long* TS, int& nt) {
#define SPR_STAMP() do { if (TS) TS[nt++] = __builtin_amdgcn_s_memrealtime(); } while (0)
  const int tid = threadIdx.x, cw = C / G, hid = cw / 4, L = 5 * cw, L4 = L / 4, r1 = L + 4, r2 = hid + 1;
  float* pvec = sm;
  float* hbuf = sm + C * 5;
  float* obuf = hbuf + G * hid;
  float* w1s = att + C + ((4 - ((C * 7 + G * hid) & 3)) & 3);        // 16-byte aligned rows
  float* w2s = w1s + hid * r1;
  float* b1s = w2s + cw * r2;
  float* b2s = b1s + hid;
  // ---- every global load of the prologue, issued back to back
  constexpr int PB = 16, WB = 6;
  const int ph = 256 / C, c = tid % C, sp0 = tid / C;                 // threads per channel, this thread's channel and first slot
  const float* pc = partial + (long)n * nsplit * C + c;
  float pv[PB];
#pragma unroll
  for (int u = 0; u < PB; ++u) { const int sp = sp0 + u * ph; pv[u] = (sp0 < ph && sp < nsplit) ? pc[(long)sp * C] : 0.f; }
  const int nw4 = hid * L4;
  float4 wv[WB];
#pragma unroll
  for (int u = 0; u < WB; ++u) { const int i4 = tid + u * 256; wv[u] = i4 < nw4 ? ((const float4*)w1)[i4] : make_float4(0.f, 0.f, 0.f, 0.f); }
  const float w2v = tid < cw * hid ? w2[tid] : 0.f;
  const float b1v = tid < hid ? b1[tid] : 0.f, b2v = tid < cw ? b2[tid] : 0.f;
  SPR_STAMP();
  // ---- weights to LDS
#pragma unroll
  for (int u = 0; u < WB; ++u) {
    const int i4 = tid + u * 256;
    if (i4 < nw4) { const int row = (int)fdiv((uint32_t)i4, fd_l4); *(float4*)(w1s + row * r1 + (i4 - row * L4) * 4) = wv[u]; }
  }
  for (int i4 = tid + WB * 256; i4 < nw4; i4 += 256) { const int row = (int)fdiv((uint32_t)i4, fd_l4); *(float4*)(w1s + row * r1 + (i4 - row * L4) * 4) = ((const float4*)w1)[i4]; }
  if (tid < cw * hid) w2s[(tid / hid) * r2 + tid % hid] = w2v;
  for (int i = tid + 256; i < cw * hid; i += 256) w2s[(i / hid) * r2 + i % hid] = w2[i];
  if (tid < hid) b1s[tid] = b1v;
  if (tid < cw) b2s[tid] = b2v;
  for (int i = tid + 256; i < cw; i += 256) b2s[i] = b2[i];
  __builtin_amdgcn_s_waitcnt(0x0F70); SPR_STAMP();
  // ---- per-tile sums -> 5 pooled means per channel (tile = ty * tiles_x + tx lies inside ONE pooling bin)
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  auto add = [&](float v, int sp) __attribute__((always_inline)) {
    const int tile = (int)fdiv((uint32_t)sp, fd_gst), ty = (int)fdiv((uint32_t)tile, fd_tx), tx = tile - ty * tiles_x;
    const int bin = (2 * ty >= tiles_y ? 2 : 0) + (2 * tx >= tiles_x ? 1 : 0);
    acc[0] += v;
    acc[1] += bin == 0 ? v : 0.f;
    acc[2] += bin == 1 ? v : 0.f;
    acc[3] += bin == 2 ? v : 0.f;
    acc[4] += bin == 3 ? v : 0.f;
  };
#pragma unroll
  for (int u = 0; u < PB; ++u) add(pv[u], sp0 + u * ph);             // slots past the end were loaded as 0
  if (sp0 < ph) {
    for (int sp = sp0 + PB * ph; sp < nsplit; sp += PB * ph) {
#pragma unroll
      for (int u = 0; u < PB; ++u) pv[u] = sp + u * ph < nsplit ? pc[(long)(sp + u * ph) * C] : 0.f;
#pragma unroll
      for (int u = 0; u < PB; ++u) add(pv[u], sp + u * ph);
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k * 256 + tid] = acc[k];
  __syncthreads();
  for (int o = tid; o < C * 5; o += 256) {
    const int cc = o % C, k = o / C, gi = cc / cw, ci = cc - gi * cw;
    float sum = 0.f;
    for (int p = 0; p < ph; ++p) sum += red[k * 256 + p * C + cc];
    pvec[gi * L + (k == 0 ? ci : cw + 4 * ci + (k - 1))] = sum / cnt[k];
  }
  __syncthreads();
  SPR_STAMP();
  // ---- fc1 + relu (spr_module.py:21-23): tpo threads share one output's dot product (float4 pieces, interleaved), summed by an xor butterfly
  const int nout = G * hid;
  int tpo = 64;
  while (tpo > 1 && (256 / tpo < nout || tpo > L4)) tpo >>= 1;
  const int opp = 256 / tpo;                                          // outputs per pass
  for (int base = 0; base < nout; base += opp) {
    const int o = base + tid / tpo, p = tid % tpo;
    float a0 = 0.f, a1 = 0.f;
    if (o < nout) {
      const int gi = o / hid, hj = o - gi * hid;
      const float4* wr = (const float4*)(w1s + hj * r1);
      const float4* pg = (const float4*)(pvec + gi * L);
      int i = p;
      for (; i + tpo < L4; i += 2 * tpo) {
        const float4 wa = wr[i], xa = pg[i], wb = wr[i + tpo], xb = pg[i + tpo];
        a0 = fmaf(wa.w, xa.w, fmaf(wa.z, xa.z, fmaf(wa.y, xa.y, fmaf(wa.x, xa.x, a0))));
        a1 = fmaf(wb.w, xb.w, fmaf(wb.z, xb.z, fmaf(wb.y, xb.y, fmaf(wb.x, xb.x, a1))));
      }
      if (i < L4) { const float4 wa = wr[i], xa = pg[i]; a0 = fmaf(wa.w, xa.w, fmaf(wa.z, xa.z, fmaf(wa.y, xa.y, fmaf(wa.x, xa.x, a0)))); }
    }
    float s = a0 + a1;
    for (int m = tpo >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (o < nout && p == 0) hbuf[o] = fmaxf(s + b1s[o % hid], 0.f);
  }
  __syncthreads();
  SPR_STAMP();
  for (int o = tid; o < C; o += 256) {                                // fc2 + sigmoid
    const int gi = o / cw, ci = o - gi * cw;
    float s = b2s[ci];
#pragma unroll 4
    for (int hj = 0; hj < hid; ++hj) s = fmaf(w2s[ci * r2 + hj], hbuf[gi * hid + hj], s);
    obuf[o] = 1.f / (1.f + expf(-s));
  }
  __syncthreads();
  SPR_STAMP();
  for (int ci = tid; ci < cw; ci += 256) {                            // softmax over the G groups (block.py:278)
    float mx = -INFINITY;
    for (int gi = 0; gi < G; ++gi) mx = fmaxf(mx, obuf[gi * cw + ci]);
    float den = 0.f;
    for (int gi = 0; gi < G; ++gi) { const float e = expf(obuf[gi * cw + ci] - mx); att[gi * cw + ci] = e; den += e; }
    for (int gi = 0; gi < G; ++gi) att[gi * cw + ci] /= den;
  }
  __syncthreads();
}

template <typename T, int V>
__global__ __launch_bounds__(256) void spr_attn_scale_kernel(const float* __restrict__ partial, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2, int C, int G, int H, int W,
                                                             const T* __restrict__ x, long xsn, long xsh, long xsw, T* __restrict__ y, long ysn, long ysh,
                                                             long ysw, FastDiv fd_q, FastDiv fd_w, int nsplit, int tiles_x, int tiles_y, SprPool pa, SprPool pb, FastDiv fd_gst,
                                                             FastDiv fd_tx, FastDiv fd_l4, int fast, unsigned long long* dbg) {
  extern __shared__ float sm[];
  unsigned long long TS[10]; int nt = 0;
  auto stamp = [&]() __attribute__((always_inline)) { if (dbg) TS[nt++] = __builtin_amdgcn_s_memrealtime(); };
  stamp();
  const int n = blockIdx.y, cw = C / G, hid = cw / 4;
  float* pooled = sm;               // [C][5] means
  float* hbuf = sm + C * 5;         // [G][hid]
  float* obuf = hbuf + G * hid;     // [C] sigmoid outputs
  float* att = obuf + C;            // [C] softmax over groups
  const int hs1 = bin_start(1, H, 2), he0 = bin_end(0, H, 2), ws1 = bin_start(1, W, 2), we0 = bin_end(0, W, 2);
  const float cnt[5] = {(float)H * W, (float)he0 * we0, (float)he0 * (W - ws1), (float)(H - hs1) * we0, (float)(H - hs1) * (W - ws1)};
  __shared__ float red[5 * 256];
  if (fast) {
    spr_prologue_fast(partial, w1, b1, w2, b2, C, G, n, nsplit, tiles_x, tiles_y, fd_gst, fd_tx, fd_l4, cnt, sm, red, att, dbg ? TS : nullptr, nt);
  } else {
    if (tiles_x > 0) {
      // per-tile sums written by mgdt_csp_block_fwd: partial[n][slot][c], slot = tile * gst + k, tile = ty * tiles_x + tx lies inside ONE pooling
      // bin.  256 threads read coalesced rows of C floats with many loads in flight (a serial walk over hundreds of slots per value is
      // latency-bound: it cost more than the block kernel itself), then the threads that share a channel are summed in a fixed order.
      const int gst = nsplit / (tiles_x * tiles_y);
      for (int c0 = 0; c0 < C; c0 += 256) {
        const int lanes_c = min(C - c0, 256);                     // channels handled in this pass
        const int ph = 256 / lanes_c;                             // threads per channel (C is a multiple of 4, <= 256 typical)
        const int c = c0 + threadIdx.x % lanes_c, sp0 = threadIdx.x / lanes_c;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (sp0 < ph) {
          for (int sp = sp0; sp < nsplit; sp += ph) {
            const float v = partial[((long)n * nsplit + sp) * C + c];
            const int tile = sp / gst, ty = tile / tiles_x, tx = tile - ty * tiles_x;
            const int bin = (2 * ty >= tiles_y ? 2 : 0) + (2 * tx >= tiles_x ? 1 : 0);
            acc[0] += v;
            acc[1] += bin == 0 ? v : 0.f;
            acc[2] += bin == 1 ? v : 0.f;
            acc[3] += bin == 2 ? v : 0.f;
            acc[4] += bin == 3 ? v : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) red[k * 256 + threadIdx.x] = acc[k];
        __syncthreads();
        for (int o = threadIdx.x; o < lanes_c * 5; o += 256) {
          const int cc = o % lanes_c, k = o / lanes_c;
          float sum = 0.f;
          for (int p = 0; p < ph; ++p) sum += red[k * 256 + p * lanes_c + cc];
          pooled[(c0 + cc) * 5 + k] = sum / cnt[k];
        }
        __syncthreads();
      }
    } else {
      for (int i = threadIdx.x; i < C * 5; i += 256) {
        float sum = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) sum += partial[((long)n * nsplit + sp) * C * 5 + i];
        pooled[i] = sum / cnt[i % 5];
      }
      __syncthreads();
    }
    for (int o = threadIdx.x; o < G * hid; o += 256) {   // fc1 + relu (spr_module.py:21-23), as in spr_attn_kernel
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
      for (int gi = 0; gi < G; ++gi) att[gi * cw + c] = expf(obuf[gi * cw + c] - mx) / den;
    }
    __syncthreads();
  }
  stamp();
  auto fin = [&]() __attribute__((always_inline)) { if (dbg) { __builtin_amdgcn_s_waitcnt(0); stamp(); if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = TS[k]; } };
  {
    // pooled outputs with factors {4}, {2}, {4, 2}: the cell form does everything in one pass
    const int fa = pa.F, fb = pb.F, fm = max(fa, fb);
    const bool cells = fm > 0 && (fm == 2 || fm == 4) && (fa == 0 || fa == fm || (fm == 4 && fa == 2)) && (fb == 0 || fb == fm || (fm == 4 && fb == 2)) && !(fa == fb);
    if (cells) {
      const SprPool none = {nullptr, 0, 0, 0, 0};
      const SprPool big = fa == fm ? pa : (fb == fm ? pb : none), small = (fa && fa < fm) ? pa : ((fb && fb < fm) ? pb : none);
      if (fm == 4) spr_scale_cells<T, V, 4>(x, xsn, xsh, xsw, y, ysn, ysh, ysw, att, n, H, W, C, big, small);
      else spr_scale_cells<T, V, 2>(x, xsn, xsh, xsw, y, ysn, ysh, ysw, att, n, H, W, C, big, small);
      fin();
      return;
    }
  }
  const int Q = C / V, HW = H * W;
  const int p0 = (int)((long)blockIdx.x * HW / gridDim.x), p1 = (int)((long)(blockIdx.x + 1) * HW / gridDim.x);
  const uint32_t total = (uint32_t)(p1 - p0) * Q;
  auto one = [&](uint32_t i, long& xo, long& yo, int& qv) __attribute__((always_inline)) {
    const int pl = (int)fdiv(i, fd_q), q = (int)i - pl * Q;
    const int p = p0 + pl, h = (int)fdiv((uint32_t)p, fd_w), w = p - h * W;
    xo = n * xsn + h * xsh + w * xsw + q * V; yo = n * ysn + h * ysh + w * ysw + q * V; qv = q * V;
  };
  uint32_t i = threadIdx.x;
  for (; i + 3 * 256 < total; i += 4 * 256) {       // 4 independent loads in flight per thread
    float v[4][V];
    long yo[4];
    int qv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { long xo; one(i + u * 256, xo, yo[u], qv[u]); ldv<T, V>(x + xo, v[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[u][k] *= att[qv[u] + k];
      stv<T, V>(y + yo[u], v[u]);
    }
  }
  for (; i < total; i += 256) {
    long xo, yo; int qv; float v[V];
    one(i, xo, yo, qv);
    ldv<T, V>(x + xo, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] *= att[qv + k];
    stv<T, V>(y + yo, v);
  }
  // pooled copies: exactly what mgdt_adaptive_avgpool_fwd computes from the stored map - every scaled value rounded to T first, summed row by
  // row in fp32, times 1 / F^2 - from x again (this workgroup's share of the image was just streamed through L2)
  const SprPool pools[2] = {pa, pb};
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    const SprPool P = pools[pi];
    if (P.F == 0) continue;
    const int F = P.F, Ho = H / F, Wo = W / F, HWo = Ho * Wo;
    const int q0 = (int)((long)blockIdx.x * HWo / gridDim.x), q1 = (int)((long)(blockIdx.x + 1) * HWo / gridDim.x);
    const float inv = 1.f / (float)(F * F);
    for (int j = threadIdx.x; j < (q1 - q0) * Q; j += 256) {
      const int pl = j / Q, q = j - pl * Q, po = q0 + pl, oy = po / Wo, ox = po - oy * Wo;
      float a[V], acc[V];
#pragma unroll
      for (int k = 0; k < V; ++k) { a[k] = att[q * V + k]; acc[k] = 0.f; }
      for (int yy = 0; yy < F; ++yy)
        for (int xx = 0; xx < F; ++xx) {
          float v[V];
          ldv<T, V>(x + n * xsn + (long)(oy * F + yy) * xsh + (long)(ox * F + xx) * xsw + q * V, v);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += (float)(T)(v[k] * a[k]);
        }
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] *= inv;
      stv<T, V>((T*)P.y + n * P.sn + oy * P.sh + ox * P.sw + q * V, acc);
    }
  }
  fin();
}

extern "C" int mgdt_spr_attn_scale_fwd(const float* pooled, int nsplit, int tiles_x, int tiles_y, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int groups,
                                       const mgdt_view* x, const mgdt_view* y, const mgdt_view* pool_a, const mgdt_view* pool_b, int dtype, mgdt_stream s) {
  if (nsplit <= 0) nsplit = SPR_SPLITS;
  if (tiles_x > 0 && (tiles_y <= 0 || nsplit % (tiles_x * tiles_y) || tiles_x % 2 || tiles_y % 2 || x->h % 2 || x->w % 2))
    MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn_scale: per-tile sums need an even tile grid over an even map (nsplit=%d tiles %dx%d)", nsplit, tiles_y, tiles_x);
  if (!pooled || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "spr_attn_scale: null/empty argument");
  const int c = x->c;
  if (groups < 1 || c % groups || (c / groups) % 4 || c > 4096) MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn_scale: c=%d groups=%d", c, groups);
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || y->c != c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn_scale: views must be matching NHWC, c%%4==0");
  const int cw = c / groups, hid = cw / 4;
  size_t lds = (size_t)(c * 5 + groups * hid + 2 * c) * sizeof(float);
  const size_t wbytes = ((size_t)hid * (5 * cw + 4) + (size_t)cw * (hid + 1) + hid + cw + 4) * sizeof(float);
  // the latency-lean prologue: per-tile sums, one pass over the channels, fc weights staged in LDS (cw <= 80); anything else takes the general one
  const int fast = tiles_x > 0 && c <= 256 && lds + wbytes <= 48 * 1024 && ((uintptr_t)fc1_w & 15) == 0;
  if (fast) lds += wbytes;
  SprPool pl[2] = {{nullptr, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0}};
  bool pool8 = true;
  const mgdt_view* pv[2] = {pool_a, pool_b};
  for (int i = 0; i < 2; ++i) {
    const mgdt_view* p = pv[i];
    if (!p || !p->p) continue;
    if (!view_ok(p) || !vec4_ok(p, dtype) || p->n != x->n || p->c != c || p->h < 1 || x->h % p->h || x->w % p->w || x->h / p->h != x->w / p->w)
      MGDT_FAIL(MGDT_BAD_SHAPE, "spr_attn_scale: a pooled output must be an NHWC view of the same n, c whose size divides the map by one factor");
    pl[i].y = p->p; pl[i].sn = p->sn; pl[i].sh = p->sh; pl[i].sw = p->sw; pl[i].F = x->h / p->h;
    pool8 = pool8 && vecN_ok(p, dtype, 8);
  }
  static unsigned long long* dbgbuf = nullptr;      // MGDT_SPR_DBG: per workgroup {start, attention weights ready, end} in 10 ns ticks
  if (getenv("MGDT_SPR_DBG") && !dbgbuf) (void)hipMalloc((void**)&dbgbuf, 4096 * 8 * 8);
  int gK = 0;
  MGDT_DISPATCH_TV(dtype, vecN_ok(x, dtype, 8) && vecN_ok(y, dtype, 8) && pool8, {
    const long vecs = (long)x->h * x->w * (c / V);
    static const long per_wg = getenv("MGDT_SPR_VECS") ? atol(getenv("MGDT_SPR_VECS")) : 2048;   // experiment knob: vectors per workgroup
    // every workgroup repeats the attention prologue (the reduction of the per-tile sums dominates it): ~12 workgroups per image measured best on all
    // four backbone levels (33.5 / 21.0 / 17.2 / 20.6 us vs 41.9 / 23.2 / 17.2 / 20.6 with up to 64)
    static const long kmax = getenv("MGDT_SPR_KMAX") ? atol(getenv("MGDT_SPR_KMAX")) : 12;
    const int K = (int)std::max<long>(1, std::min<long>(kmax, vecs / per_wg));
    spr_attn_scale_kernel<T, V><<<dim3(K, x->n), 256, lds, (hipStream_t)s>>>(pooled, fc1_w, fc1_b, fc2_w, fc2_b, c, groups, x->h, x->w, (const T*)x->p, x->sn,
                                                                              x->sh, x->sw, (T*)y->p, y->sn, y->sh, y->sw, make_fastdiv((uint32_t)(c / V)),
                                                                              make_fastdiv((uint32_t)x->w), nsplit, tiles_x, tiles_y, pl[0], pl[1],
                                                                              make_fastdiv((uint32_t)(tiles_x > 0 ? nsplit / (tiles_x * tiles_y) : 1)),
                                                                              make_fastdiv((uint32_t)std::max(tiles_x, 1)), make_fastdiv((uint32_t)(5 * cw / 4)), fast, dbgbuf);
    gK = K;
  });
  MGDT_CHECK_LAUNCH("spr_attn_scale_fwd");
  if (dbgbuf && gK * x->n <= 4096) {
    (void)hipStreamSynchronize((hipStream_t)s);
    std::vector<unsigned long long> h((size_t)gK * x->n * 8);
    (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0; double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const int ns = fast ? 8 : 3;
    for (int i = 0; i < gK * x->n; ++i) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + ns - 1]); for (int k = 0; k + 1 < ns; ++k) ph[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]); }
    fprintf(stderr, "spr_attn_scale c %d %dx%d: %d wgs, span %.1f us; mean per wg (us):", c, x->h, x->w, gK * x->n, (t1 - t0) * 0.01);
    for (int k = 0; k + 1 < ns; ++k) fprintf(stderr, " %.2f", ph[k] * 0.01 / (gK * x->n));
    fprintf(stderr, fast ? "  (issue | loads land | sums | fc1 | fc2 | softmax | scale)\n" : "  (prologue | scale)\n");
  }
  return MGDT_OK;
}

template <typename T, int V>
__global__ void scale_channels_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, const float* __restrict__ attn,
                                      T* __restrict__ y, long ysn, long ysh, long ysw, uint32_t total, int C, PixIdx d) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, h, w, q;
    decode_idx(i, d, n, h, w, q);
    float v[V], a[V];
    ldv<T, V>(x + n * xsn + h * xsh + w * xsw + q * V, v);
    ldv<float, V>(attn + (long)n * C + q * V, a);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] *= a[k];
    stv<T, V>(y + n * ysn + h * ysh + w * ysw + q * V, v);
  }
}

extern "C" int mgdt_scale_channels_fwd(const mgdt_view* x, const float* attn, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !attn) MGDT_FAIL(MGDT_BAD_ARG, "scale_channels: null/empty argument");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "scale_channels: views must be matching NHWC, c%%4==0");
  long tot = (long)x->n * x->h * x->w * x->c;
  if (tot >= 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "scale_channels: too large");
  MGDT_DISPATCH_TV(dtype, vecN_ok(x, dtype, 8) && vecN_ok(y, dtype, 8), {
    PixIdx d = make_pixidx(x->h, x->w, x->c / V);
    uint32_t total = (uint32_t)(tot / V);
    scale_channels_kernel<T, V><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, attn, (T*)y->p, y->sn, y->sh, y->sw, total, x->c, d);
  });
  MGDT_CHECK_LAUNCH("scale_channels_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ SPPF pools
// One workgroup = one image x 8 channels: the whole HxW plane (20x20 at 640^2) sits in LDS as fp32 with a 2-pixel -inf halo and MaxPool2d(5,1,2)
// is applied three times as separable row / column 5-tap max passes, writing y1, y2, y3 after each.  The halo makes the taps unconditional: with
// bounds tests around the LDS reads every read waited for the previous compare (32 us for a 3 MB map).
template <typename T, int SPPF_CG>
__global__ __launch_bounds__(256) void sppf_pool_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, T* __restrict__ y1, long s1n,
                                                        long s1h, long s1w, T* __restrict__ y2, long s2n, long s2h, long s2w,
                                                        T* __restrict__ y3, long s3n, long s3h, long s3w, int H, int W) {
  extern __shared__ float sm[];
  const int n = blockIdx.x, c0 = blockIdx.y * SPPF_CG, HW = H * W, WP = W + 4, PP = (H + 4) * WP;
  float* A = sm;                  // [PP][SPPF_CG]
  float* Bf = sm + PP * SPPF_CG;
  constexpr int QN = SPPF_CG / 4;
  const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < PP * QN; i += 256) {   // 4 channels per lane
    const int pp = i / QN, qq = (i % QN) * 4;
    const int hh = pp / WP - 2, ww = pp % WP - 2;
    const bool in = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
    *(f32x4*)(A + pp * SPPF_CG + qq) = in ? load4<T>(x + n * xsn + hh * xsh + ww * xsw + c0 + qq) : ninf;
    *(f32x4*)(Bf + pp * SPPF_CG + qq) = ninf;
  }
  __syncthreads();
  T* outs[3] = {y1, y2, y3};
  const long on[3] = {s1n, s2n, s3n}, oh[3] = {s1h, s2h, s3h}, ow[3] = {s1w, s2w, s3w};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = threadIdx.x; i < HW * SPPF_CG; i += 256) {   // rows: A -> Bf
      const int c = i % SPPF_CG, p = i / SPPF_CG, h = p / W, w = p - h * W;
      const int o = ((h + 2) * WP + w + 2) * SPPF_CG + c;
      const float a0 = A[o - 2 * SPPF_CG], a1 = A[o - SPPF_CG], a2 = A[o], a3 = A[o + SPPF_CG], a4 = A[o + 2 * SPPF_CG];
      Bf[o] = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), a4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW * SPPF_CG; i += 256) {   // columns: Bf -> A
      const int c = i % SPPF_CG, p = i / SPPF_CG, h = p / W, w = p - h * W;
      const int o = ((h + 2) * WP + w + 2) * SPPF_CG + c, rs = WP * SPPF_CG;
      const float b0 = Bf[o - 2 * rs], b1 = Bf[o - rs], b2 = Bf[o], b3 = Bf[o + rs], b4 = Bf[o + 2 * rs];
      A[o] = fmaxf(fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)), b4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW * QN; i += 256) {
      const int p = i / QN, qq = (i % QN) * 4, h = p / W, w = p - h * W;
      store4<T>(outs[pass] + n * on[pass] + h * oh[pass] + w * ow[pass] + c0 + qq, *(const f32x4*)(A + ((h + 2) * WP + w + 2) * SPPF_CG + qq));
    }
  }
}

// Separable form in TWO phases: three chained MaxPool2d(5, 1, 2) with -inf padding are the max over the 5x5, 9x9 and 13x13 windows clipped to the map
// (max is associative and exact, so y1, y2, y3 are bit-identical to the chained pools).  Phase 1: per pixel the running maxima over 5 / 9 / 13
// columns of its row; phase 2: the maxima over 5 / 9 / 13 rows of those - two barriers instead of the nine of the chained form (the kernel is
// a latency chain: one small workgroup per (image, channel group), everything in LDS, in the storage type).
template <typename T, int V>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, T* __restrict__ y1, long s1n, long s1h, long s1w,
                                                         T* __restrict__ y2, long s2n, long s2h, long s2w, T* __restrict__ y3, long s3n, long s3h, long s3w, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char sm3[];
  const int n = blockIdx.x, c0 = blockIdx.y * V, WP = W + 12, HP = H + 12;
  T* X = (T*)sm3;                                           // [H][WP][V]: the rows of the map with a 6-column -inf halo
  T* R = X + (size_t)H * WP * V;                            // [3][HP][W][V]: row maxima (5, 9, 13 wide) with a 6-row -inf halo
  float ninf[V];
#pragma unroll
  for (int k = 0; k < V; ++k) ninf[k] = -INFINITY;
  for (int i = threadIdx.x; i < H * WP; i += 256) {
    const int h = i / WP, wp = i - h * WP, w = wp - 6;
    float v[V];
    if ((unsigned)w < (unsigned)W) ldv<T, V>(x + n * xsn + h * xsh + w * xsw + c0, v);
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = -INFINITY;
    }
    stv<T, V>(X + (size_t)i * V, v);
  }
  for (int i = threadIdx.x; i < 3 * 12 * W; i += 256) {    // halo rows of the three row-maximum planes
    const int pl = i / (12 * W), rem = i - pl * 12 * W, hr = rem / W, w = rem - hr * W;
    const int hp = hr < 6 ? hr : H + hr;                   // rows 0..5 and H+6..H+11
    stv<T, V>(R + ((size_t)(pl * HP + hp) * W + w) * V, ninf);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int h = i / W, w = i - h * W;
    const T* row = X + ((size_t)h * WP + w) * V;            // row[j] = column w - 6 + j
    float t[13][V];
#pragma unroll
    for (int j = 0; j < 13; ++j) ldv<T, V>(row + (size_t)j * V, t[j]);
    float m5[V], m9[V], m13[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      m5[k] = fmaxf(fmaxf(fmaxf(t[4][k], t[5][k]), fmaxf(t[6][k], t[7][k])), t[8][k]);
      m9[k] = fmaxf(fmaxf(m5[k], fmaxf(t[2][k], t[3][k])), fmaxf(t[9][k], t[10][k]));
      m13[k] = fmaxf(fmaxf(m9[k], fmaxf(t[0][k], t[1][k])), fmaxf(t[11][k], t[12][k]));
    }
    stv<T, V>(R + ((size_t)(0 * HP + h + 6) * W + w) * V, m5);
    stv<T, V>(R + ((size_t)(1 * HP + h + 6) * W + w) * V, m9);
    stv<T, V>(R + ((size_t)(2 * HP + h + 6) * W + w) * V, m13);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int h = i / W, w = i - h * W;
    float o[V], t[V];
    // y1: 5 rows of the 5-wide maxima
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = -INFINITY;
#pragma unroll
    for (int j = 4; j <= 8; ++j) {
      ldv<T, V>(R + ((size_t)(0 * HP + h + j) * W + w) * V, t);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], t[k]);
    }
    stv<T, V>(y1 + n * s1n + h * s1h + w * s1w + c0, o);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = -INFINITY;
#pragma unroll
    for (int j = 2; j <= 10; ++j) {
      ldv<T, V>(R + ((size_t)(1 * HP + h + j) * W + w) * V, t);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], t[k]);
    }
    stv<T, V>(y2 + n * s2n + h * s2h + w * s2w + c0, o);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = -INFINITY;
#pragma unroll
    for (int j = 0; j <= 12; ++j) {
      ldv<T, V>(R + ((size_t)(2 * HP + h + j) * W + w) * V, t);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], t[k]);
    }
    stv<T, V>(y3 + n * s3n + h * s3h + w * s3w + c0, o);
  }
}

extern "C" int mgdt_sppf_pool_fwd(const mgdt_view* x, const mgdt_view* y1, const mgdt_view* y2, const mgdt_view* y3, int dtype,
                                  mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y1) || !view_ok(y2) || !view_ok(y3)) MGDT_FAIL(MGDT_BAD_ARG, "sppf_pool: null/empty view");
  for (const mgdt_view* v : {x, y1, y2, y3})
    if (!vec4_ok(v, dtype) || v->n != x->n || v->h != x->h || v->w != x->w || v->c != x->c)
      MGDT_FAIL(MGDT_BAD_SHAPE, "sppf_pool: views must be matching NHWC, c%%4==0");
  {
    // two-phase form: LDS = rows + three row-maximum planes in the storage type
    static const bool old_form = getenv("MGDT_SPPF_CHAINED") != nullptr;      // experiment knob: the chained three-pass kernel
    bool all8 = true;
    for (const mgdt_view* v : {x, y1, y2, y3}) all8 = all8 && vecN_ok(v, dtype, 8);
    const int V3 = (dtype == MGDT_BF16 && all8) ? 8 : 4;
    const size_t lds3 = ((size_t)x->h * (x->w + 12) + (size_t)3 * (x->h + 12) * x->w) * V3 * dtype_size(dtype);
    if (!old_form && lds3 <= 64 * 1024) {
      dim3 grid3(x->n, x->c / V3);
      MGDT_DISPATCH_TV(dtype, all8, {
        sppf_pool3_kernel<T, V><<<grid3, 256, lds3, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, (T*)y1->p, y1->sn, y1->sh, y1->sw, (T*)y2->p, y2->sn, y2->sh, y2->sw,
                                                                    (T*)y3->p, y3->sn, y3->sh, y3->sw, x->h, x->w);
      });
      MGDT_CHECK_LAUNCH("sppf_pool_fwd");
      return MGDT_OK;
    }
  }
  const long pp = (long)(x->h + 4) * (x->w + 4);                                  // plane with its 2-pixel halo
  const int cg = (x->c % 8 == 0 && pp * 8 * 2 * 4 <= 64 * 1024) ? 8 : 4;          // channels per workgroup so the planes fit 64 KiB of LDS
  size_t lds = (size_t)pp * cg * 2 * sizeof(float);
  if (lds > 64 * 1024) MGDT_FAIL(MGDT_BAD_SHAPE, "sppf_pool: the %dx%d map does not fit the LDS plane kernel (h*w <= 2048)", x->h, x->w);
  dim3 grid(x->n, x->c / cg);
#define SPPF_L(CG) MGDT_DISPATCH_DTYPE(dtype, (sppf_pool_kernel<T, CG><<<grid, 256, lds, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, (T*)y1->p, y1->sn, y1->sh, y1->sw, \
                                                 (T*)y2->p, y2->sn, y2->sh, y2->sw, (T*)y3->p, y3->sn, y3->sh, y3->sw, x->h, x->w)))
  if (cg == 8) SPPF_L(8); else SPPF_L(4);
#undef SPPF_L
  MGDT_CHECK_LAUNCH("sppf_pool_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ resamplers
template <typename T, int V>
__global__ void avgpool_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                               long ysh, long ysw, uint32_t total, PixIdx d) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, oy, ox, q;
    decode_idx(i, d, n, oy, ox, q);
    int y0 = bin_start(oy, H, d.H), y1 = bin_end(oy, H, d.H), x0 = bin_start(ox, W, d.W), x1 = bin_end(ox, W, d.W);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        float v[V];
        ldv<T, V>(x + n * xsn + yy * xsh + xx * xsw + q * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += v[k];
      }
    float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] *= inv;
    stv<T, V>(y + n * ysn + oy * ysh + ox * ysw + q * V, acc);
  }
}

#define RESAMPLE_ENTRY(fname, kern, label)                                                                                              \
  extern "C" int fname(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s) {                                             \
    if (!view_ok(x) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, label ": null/empty view");                                                \
    if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, label ": NHWC views, c%%4==0, same n/c"); \
    long tot = (long)y->n * y->h * y->w * y->c;                                                                                        \
    if (tot >= 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, label ": too large");                                                            \
    MGDT_DISPATCH_TV(dtype, vecN_ok(x, dtype, 8) && vecN_ok(y, dtype, 8), {                                                            \
      PixIdx d = make_pixidx(y->h, y->w, y->c / V);                                                                                    \
      uint32_t total = (uint32_t)(tot / V);                                                                                            \
      kern<T, V><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w, (T*)y->p, y->sn, y->sh, y->sw, total, d); \
    });                                                                                                                                \
    MGDT_CHECK_LAUNCH(label);                                                                                                          \
    return MGDT_OK;                                                                                                                    \
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

template <typename T, int V, bool HSIG>
__device__ __forceinline__ void bilerp(const T* x, long sh, long sw, Lerp ly, Lerp lx, float (&o)[V]) {
  float v00[V], v01[V], v10[V], v11[V];
  ldv<T, V>(x + ly.i0 * sh + lx.i0 * sw, v00);
  ldv<T, V>(x + ly.i0 * sh + lx.i1 * sw, v01);
  ldv<T, V>(x + ly.i1 * sh + lx.i0 * sw, v10);
  ldv<T, V>(x + ly.i1 * sh + lx.i1 * sw, v11);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    if (HSIG) {   // h_sigmoid BEFORE the interpolation (block.py:393)
      v00[k] = fminf(fmaxf(v00[k] + 3.f, 0.f), 6.f) / 6.f; v01[k] = fminf(fmaxf(v01[k] + 3.f, 0.f), 6.f) / 6.f;
      v10[k] = fminf(fmaxf(v10[k] + 3.f, 0.f), 6.f) / 6.f; v11[k] = fminf(fmaxf(v11[k] + 3.f, 0.f), 6.f) / 6.f;
    }
    o[k] = (v00[k] * lx.l0 + v01[k] * lx.l1) * ly.l0 + (v10[k] * lx.l0 + v11[k] * lx.l1) * ly.l1;
  }
}

template <typename T, int V>
__global__ void bilinear_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                                long ysh, long ysw, uint32_t total, PixIdx d) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, oy, ox, q;
    decode_idx(i, d, n, oy, ox, q);
    float o[V];
    bilerp<T, V, false>(x + n * xsn + q * V, xsh, xsw, lerp_of(oy, H, d.H), lerp_of(ox, W, d.W), o);
    stv<T, V>(y + n * ysn + oy * ysh + ox * ysw + q * V, o);
  }
}

template <typename T, int V>
__global__ void nearest_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, int H, int W, T* __restrict__ y, long ysn,
                               long ysh, long ysw, uint32_t total, PixIdx d) {
  const float sy = (float)H / (float)d.H, sx = (float)W / (float)d.W;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, oy, ox, q;
    decode_idx(i, d, n, oy, ox, q);
    int iy = min((int)floorf((float)oy * sy), H - 1), ix = min((int)floorf((float)ox * sx), W - 1);
    float v[V];
    ldv<T, V>(x + n * xsn + iy * xsh + ix * xsw + q * V, v);
    stv<T, V>(y + n * ysn + oy * ysh + ox * ysw + q * V, v);
  }
}

RESAMPLE_ENTRY(mgdt_adaptive_avgpool_fwd, avgpool_kernel, "adaptive_avgpool")
RESAMPLE_ENTRY(mgdt_bilinear_fwd, bilinear_kernel, "bilinear")
RESAMPLE_ENTRY(mgdt_nearest_fwd, nearest_kernel, "nearest")

// ------------------------------------------------------------------------------------------------ Injection tail
template <typename T, int V>
__global__ void inject_kernel(const T* __restrict__ loc, long lsn, long lsh, long lsw, const T* __restrict__ ga, long asn, long ash,
                              long asw, const T* __restrict__ gf, long fsn, long fsh, long fsw, int Hg, int Wg, T* __restrict__ y,
                              long ysn, long ysh, long ysw, uint32_t total, PixIdx d, int use_pool) {
  const int H = d.H, W = d.W;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int n, oy, ox, q;
    decode_idx(i, d, n, oy, ox, q);
    float l[V], sig[V], feat[V];
    ldv<T, V>(loc + n * lsn + oy * lsh + ox * lsw + q * V, l);
    if (use_pool) {
      int y0 = bin_start(oy, Hg, H), y1 = bin_end(oy, Hg, H), x0 = bin_start(ox, Wg, W), x1 = bin_end(ox, Wg, W);
#pragma unroll
      for (int k = 0; k < V; ++k) sig[k] = feat[k] = 0.f;
      for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
          float a[V], f[V];
          ldv<T, V>(ga + n * asn + yy * ash + xx * asw + q * V, a);
          ldv<T, V>(gf + n * fsn + yy * fsh + xx * fsw + q * V, f);
#pragma unroll
          for (int k = 0; k < V; ++k) { sig[k] += a[k]; feat[k] += f[k]; }
        }
      float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
      for (int k = 0; k < V; ++k) { sig[k] *= inv; feat[k] *= inv; }
    } else {
      Lerp ly = lerp_of(oy, Hg, H), lx = lerp_of(ox, Wg, W);
      bilerp<T, V, true>(ga + n * asn + q * V, ash, asw, ly, lx, sig);
      bilerp<T, V, false>(gf + n * fsn + q * V, fsh, fsw, ly, lx, feat);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) l[k] = l[k] * sig[k] + feat[k];
    stv<T, V>(y + n * ysn + oy * ysh + ox * ysw + q * V, l);
  }
}

extern "C" int mgdt_inject_fwd(const mgdt_view* local, const mgdt_view* ga, const mgdt_view* gf, const mgdt_view* y, int dtype,
                               mgdt_stream s) {
  if (!view_ok(local) || !view_ok(ga) || !view_ok(gf) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "inject: null/empty view");
  bool v8 = true;
  for (const mgdt_view* v : {local, ga, gf, y}) {
    if (!vec4_ok(v, dtype) || v->n != y->n || v->c != y->c) MGDT_FAIL(MGDT_BAD_SHAPE, "inject: NHWC views, c%%4==0, same n/c");
    v8 = v8 && vecN_ok(v, dtype, 8);
  }
  if (local->h != y->h || local->w != y->w || ga->h != gf->h || ga->w != gf->w) MGDT_FAIL(MGDT_BAD_SHAPE, "inject: spatial mismatch");
  int use_pool = local->h < ga->h;
  long tot = (long)y->n * y->h * y->w * y->c;
  if (tot >= 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "inject: too large");
  MGDT_DISPATCH_TV(dtype, v8, {
    PixIdx d = make_pixidx(y->h, y->w, y->c / V);
    uint32_t total = (uint32_t)(tot / V);
    inject_kernel<T, V><<<grid_for(total), 256, 0, (hipStream_t)s>>>((const T*)local->p, local->sn, local->sh, local->sw, (const T*)ga->p, ga->sn, ga->sh, ga->sw,
                                                                     (const T*)gf->p, gf->sn, gf->sh, gf->sw, ga->h, ga->w, (T*)y->p, y->sn, y->sh, y->sw, total, d, use_pool);
  });
  MGDT_CHECK_LAUNCH("inject_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ ConvNeXtV2: dw7x7 + LayerNorm
// block = PPB pixels x Q channel-quads; LayerNorm over the pixel's channels through LDS (two-pass mean / variance).
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw,
                                                         const float* __restrict__ dw, const float* __restrict__ db,
                                                         const float* __restrict__ lw, const float* __restrict__ lb, float eps,
                                                         T* __restrict__ y, long ysn, long ysh, long ysw, int N, int H, int W, int C,
                                                         T* __restrict__ uo = nullptr, long usn = 0, long ush = 0, long usw = 0) {
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
  if (uo && active) store4<T>(uo + n * usn + oy * ush + ox * usw + q * 4, acc);   // training: keep the pre-LayerNorm map
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

// LDS-tiled variant: one workgroup = an 8x8 output tile of one image x all channels.  The (8+6)x(8+6) input halo is staged
// once in LDS (each input pixel is used by up to 49 outputs), a thread owns one channel quad and ~6 pixels, keeps one kernel
// row of weights (7 x float4) in registers at a time; LayerNorm over the pixel's channels goes through LDS partials.
#define DW_TH 8
#define DW_TW 8
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void dwconv7_ln_tiled_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw,
                                                               const float* __restrict__ dw, const float* __restrict__ db,
                                                               const float* __restrict__ lw, const float* __restrict__ lb, float eps,
                                                               T* __restrict__ y, long ysn, long ysh, long ysw, int H, int W, int C,
                                                               T* __restrict__ uo, long usn, long ush, long usw) {
  extern __shared__ __attribute__((aligned(16))) char smem_dw[];
  const int Q = C / 4, PL = 256 / Q;                    // channel quads, pixel lanes
  constexpr int HH = DW_TH + 6, HW_ = DW_TW + 6, NPIX = DW_TH * DW_TW;
  T* halo = (T*)smem_dw;                                // [HH][HW_][C]
  float* wl = (float*)(smem_dw + (((size_t)HH * HW_ * C * sizeof(T) + 15) & ~(size_t)15));   // [49][C]
  float* red = wl + 49 * C;                             // [NPIX][Q] partial sums (reused for mean and variance)
  const int tiles_x = (W + DW_TW - 1) / DW_TW, tiles_y = (H + DW_TH - 1) / DW_TH;
  const int n = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x % (tiles_x * tiles_y);
  const int ty0 = (tr / tiles_x) * DW_TH, tx0 = (tr % tiles_x) * DW_TW;
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const bool live = pl < PL;
  if (live)
    for (int p = pl; p < HH * HW_; p += PL) {            // stage the halo (zero padding outside the image); q, pl fixed per thread
      int hy = p / HW_, hx = p % HW_;
      int iy = ty0 + hy - 3, ix = tx0 + hx - 3;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = load4<T>(x + n * xsn + iy * xsh + ix * xsw + q * 4);
      store4<T>(halo + (long)p * C + q * 4, v);
    }
  for (int i = threadIdx.x; i < 49 * Q; i += 256) *(f32x4*)(wl + i * 4) = *(const f32x4*)(dw + i * 4);
  __syncthreads();
  f32x4 acc[MAXP];                                      // MAXP >= ceil(NPIX / PL) (host-selected instantiation)
  if (live) {
    const f32x4 bq = *(const f32x4*)(db + q * 4);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) acc[k] = bq;
    for (int ky = 0; ky < 7; ++ky) {
      f32x4 wr[7];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) wr[kx] = *(const f32x4*)(wl + (ky * 7 + kx) * C + q * 4);
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        int p = pl + k * PL;
        if (p < NPIX) {
          int py = p / DW_TW, px = p % DW_TW;
          const T* hp = halo + ((long)(py + ky) * HW_ + px) * C + q * 4;
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            f32x4 v = load4<T>(hp + kx * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(v[j], wr[kx][j], acc[k][j]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      int p = pl + k * PL;
      if (p < NPIX) red[p * Q + q] = acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    }
  }
  __syncthreads();
  float mean[MAXP];
  if (live) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      int p = pl + k * PL;
      mean[k] = 0.f;
      if (p < NPIX) {
        for (int j = 0; j < Q; ++j) mean[k] += red[p * Q + j];
        mean[k] /= (float)C;
      }
    }
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      int p = pl + k * PL;
      if (p < NPIX) {
        f32x4 d = acc[k] - mean[k];
        red[p * Q + q] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      }
    }
  }
  __syncthreads();
  if (live) {
    const f32x4 gq = *(const f32x4*)(lw + q * 4), bq2 = *(const f32x4*)(lb + q * 4);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      int p = pl + k * PL;
      if (p >= NPIX) continue;
      int oy = ty0 + p / DW_TW, ox = tx0 + p % DW_TW;
      if (oy >= H || ox >= W) continue;
      float var = 0.f;
      for (int j = 0; j < Q; ++j) var += red[p * Q + j];
      float rstd = 1.f / sqrtf(var / (float)C + eps);
      if (uo) store4<T>(uo + n * usn + oy * ush + ox * usw + q * 4, acc[k]);
      store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, (acc[k] - mean[k]) * rstd * gq + bq2);
    }
  }
}

// Row-sliding variant (C <= 128): a thread owns one channel quad and one ROW of the 8x8 output tile.  Per kernel row it reads the
// 14 halo values of its row once and reuses each for up to 7 of its 8 outputs from registers: 4x fewer LDS reads and bf16->fp32
// conversions than one-pixel-per-iteration, 8 independent accumulators of ILP.  LayerNorm: per-pixel partials -> 64 threads
// reduce over the quads -> broadcast, twice (mean, then centred variance: the two-pass form F.layer_norm uses).
template <typename T, int TS>
__global__ __launch_bounds__(256) void dwconv7_ln_row_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw,
                                                             const float* __restrict__ dw, const float* __restrict__ db,
                                                             const float* __restrict__ lw, const float* __restrict__ lb, float eps,
                                                             T* __restrict__ y, long ysn, long ysh, long ysw, int H, int W, int C,
                                                             T* __restrict__ uo, long usn, long ush, long usw, unsigned long long* dbg) {
  unsigned long long TT[6] = {0, 0, 0, 0, 0, 0};
  const bool dbgw = dbg && threadIdx.x == 0;
  if (dbgw) TT[0] = wall_clock64();
  extern __shared__ __attribute__((aligned(16))) char smem_dw[];
  const int Q = C / 4;                                  // blockDim.x == 8 * Q
  constexpr int HH = TS + 6, HW_ = TS + 6, NPIX = TS * TS;
  T* halo = (T*)smem_dw;                                // [HH][HW_][C]
  float* wl = (float*)(smem_dw + (((size_t)HH * HW_ * C * sizeof(T) + 15) & ~(size_t)15));   // [49][C]
  float* red = wl + 49 * C;                             // [NPIX][Q] partial sums (reused for mean and variance)
  float* stat = red + NPIX * Q;                         // [NPIX] mean, then rstd
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int n = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x % (tiles_x * tiles_y);
  const int ty0 = (tr / tiles_x) * TS, tx0 = (tr % tiles_x) * TS;
  const int q = threadIdx.x % Q, row = threadIdx.x / Q;
  // stage the halo (zero padding outside the image); q, row fixed per thread.  Loads are issued before their LDS
  // stores so that a thread has all its 25 global requests in flight at once (one round trip instead of 25 dependent ones: with two
  // workgroups per CU there is nobody else to hide that latency).
  constexpr int NST = (HH * HW_ + TS - 1) / TS;
  {
    // branch-free: out-of-image pixels get an out-of-range offset of a bounds-checked descriptor (zeros come back), so all NST loads of a
    // thread are issued back to back
    const long ext = ((long)(H - 1) * xsh + (long)(W - 1) * xsw + C) * (long)sizeof(T);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + n * xsn), 0, (int)ext, 0x00020000);
    using raw_t = typename std::conditional<sizeof(T) == 2, bf16x4, f32x4>::type;      // staged as loaded (no fp32 round trip)
    raw_t v[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int p = row + i * TS;
      const int hy = p / HW_, hx = p % HW_;
      const int iy = ty0 + hy - 3, ix = tx0 + hx - 3;
      const bool ok = p < HH * HW_ && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const uint32_t off = ok ? (uint32_t)((iy * xsh + ix * xsw + q * 4) * (long)sizeof(T)) : 0x80000000u;
      if constexpr (sizeof(T) == 2) v[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
      else v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int p = row + i * TS;
      if (p < HH * HW_) *(raw_t*)(halo + (long)p * C + q * 4) = v[i];
    }
  }
  if (dbgw) TT[1] = wall_clock64();
  for (int i = threadIdx.x; i < 49 * Q; i += blockDim.x) *(f32x4*)(wl + i * 4) = *(const f32x4*)(dw + i * 4);
  __syncthreads();
  if (dbgw) TT[2] = wall_clock64();
  f32x4 acc[TS];
  {
    const f32x4 bq = *(const f32x4*)(db + q * 4);
#pragma unroll
    for (int k = 0; k < TS; ++k) acc[k] = bq;
  }
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    const T* hrow = halo + (long)(row + ky) * HW_ * C + q * 4;
    f32x4 in[HW_];
#pragma unroll
    for (int i = 0; i < HW_; ++i) in[i] = load4<T>(hrow + i * C);
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const f32x4 wv = *(const f32x4*)(wl + (ky * 7 + kx) * C + q * 4);
#pragma unroll
      for (int k = 0; k < TS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(in[k + kx][j], wv[j], acc[k][j]);
    }
  }
  if (dbgw) TT[3] = wall_clock64();
#pragma unroll
  for (int k = 0; k < TS; ++k) red[(row * TS + k) * Q + q] = acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  __syncthreads();
  if (threadIdx.x < NPIX) {
    float m = 0.f;
    for (int j = 0; j < Q; ++j) m += red[threadIdx.x * Q + j];
    stat[threadIdx.x] = m / (float)C;
  }
  __syncthreads();
  float mean[TS];
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    mean[k] = stat[row * TS + k];
    const f32x4 d = acc[k] - mean[k];
    red[(row * TS + k) * Q + q] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  __syncthreads();
  if (threadIdx.x < NPIX) {
    float var = 0.f;
    for (int j = 0; j < Q; ++j) var += red[threadIdx.x * Q + j];
    stat[threadIdx.x] = 1.f / sqrtf(var / (float)C + eps);
  }
  __syncthreads();
  if (dbgw) TT[4] = wall_clock64();
  const f32x4 gq = *(const f32x4*)(lw + q * 4), bq2 = *(const f32x4*)(lb + q * 4);
  const int oy = ty0 + row;
  if (oy < H) {
#pragma unroll
    for (int k = 0; k < TS; ++k) {
      const int ox = tx0 + k;
      if (ox >= W) continue;
      const float rstd = stat[row * TS + k];
      if (uo) store4<T>(uo + n * usn + oy * ush + ox * usw + q * 4, acc[k]);
      store4<T>(y + n * ysn + oy * ysh + ox * ysw + q * 4, (acc[k] - mean[k]) * rstd * gq + bq2);
    }
  }
  if (dbgw) { TT[5] = wall_clock64(); for (int i = 0; i < 6; ++i) dbg[blockIdx.x * 6 + i] = TT[i]; }
}

static int dwconv7_ln_impl(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b, float eps,
                           const mgdt_view* y, const mgdt_view* u, int dtype, mgdt_stream s);
extern "C" int mgdt_dwconv7_ln_fwd(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                                   float eps, const mgdt_view* y, int dtype, mgdt_stream s) {
  return dwconv7_ln_impl(x, dw_w, dw_b, ln_w, ln_b, eps, y, nullptr, dtype, s);
}
extern "C" int mgdt_dwconv7_ln_train_fwd(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                                         float eps, const mgdt_view* y, const mgdt_view* u_out, int dtype, mgdt_stream s) {
  if (!view_ok(u_out) || !vec4_ok(u_out, dtype) || u_out->n != x->n || u_out->h != x->h || u_out->w != x->w || u_out->c != x->c)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dwconv7_ln_train: u_out must be an NHWC view like x");
  return dwconv7_ln_impl(x, dw_w, dw_b, ln_w, ln_b, eps, y, u_out, dtype, s);
}
static int dwconv7_ln_impl(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b, float eps,
                           const mgdt_view* y, const mgdt_view* u, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !dw_w || !dw_b || !ln_w || !ln_b) MGDT_FAIL(MGDT_BAD_ARG, "dwconv7_ln: null/empty argument");
  if (!vec4_ok(x, dtype) || !vec4_ok(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c || x->c / 4 > 256)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dwconv7_ln: matching NHWC views, c%%4==0, c<=1024");
  {   // LDS-tiled fast path: one thread per (channel quad, tile row); tile side 8 or 10: whichever needs fewer rounds of workgroups over the
      // chip x pixels per round (a 40x40 map: 25 tiles of 8x8 per image = 800 workgroups = two rounds at 2 per CU; 16 tiles of 10x10 = one round)
    const int Qt = x->c / 4;
    auto lds_of = [&](int ts) {
      const size_t halo_b = ((size_t)(ts + 6) * (ts + 6) * x->c * dtype_size(dtype) + 15) & ~(size_t)15;
      return halo_b + (size_t)49 * x->c * 4 + (size_t)ts * ts * Qt * 4 + (size_t)ts * ts * sizeof(float);
    };
    auto cost_of = [&](int ts) -> long {
      const size_t l = lds_of(ts);
      if (ts * Qt > 256 || l + 256 > (ts == 8 ? 64 : 80) * 1024) return -1;
      const long nwg = (long)x->n * cdiv(x->h, ts) * cdiv(x->w, ts), per_cu = std::max<long>(1, std::min<long>(160 * 1024 / (long)(l + 256), 2048 / (ts * Qt)));
      return cdiv(nwg, 256 * per_cu) * ts * ts;
    };
    static const int force_ts = getenv("MGDT_DW_TS") ? atoi(getenv("MGDT_DW_TS")) : 0;      // experiment knob
    const long c8 = cost_of(8), c10 = cost_of(10);
    const int ts = force_ts == 8 || force_ts == 10 ? force_ts : (c10 >= 0 && (c8 < 0 || c10 < c8) ? 10 : 8);
    if (Qt <= 32 && (ts == 8 ? c8 : c10) >= 0) {
      const int tiles = cdiv(x->h, ts) * cdiv(x->w, ts);
      const size_t lds_t = lds_of(ts);
      static unsigned long long* dbgbuf = nullptr;
      if (getenv("MGDT_DW_DBG") && !dbgbuf) (void)hipMalloc((void**)&dbgbuf, (size_t)x->n * tiles * 6 * 8);
      static std::atomic<bool> attr10_f32{false}, attr10_bf16{false};
      if (ts == 10) {
        MGDT_DISPATCH_DTYPE(dtype, {
          std::atomic<bool>& fl = sizeof(T) == 2 ? attr10_bf16 : attr10_f32;
          if (!fl.load()) {
            (void)hipFuncSetAttribute((const void*)dwconv7_ln_row_kernel<T, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            fl.store(true);
          }
          dwconv7_ln_row_kernel<T, 10><<<x->n * tiles, 10 * Qt, lds_t, (hipStream_t)s>>>(
              (const T*)x->p, x->sn, x->sh, x->sw, dw_w, dw_b, ln_w, ln_b, eps, (T*)y->p, y->sn, y->sh, y->sw, x->h, x->w, x->c,
              u ? (T*)u->p : nullptr, u ? u->sn : 0, u ? u->sh : 0, u ? u->sw : 0, dbgbuf);
        });
      } else {
        MGDT_DISPATCH_DTYPE(dtype, (dwconv7_ln_row_kernel<T, 8><<<x->n * tiles, 8 * Qt, lds_t, (hipStream_t)s>>>(
                                       (const T*)x->p, x->sn, x->sh, x->sw, dw_w, dw_b, ln_w, ln_b, eps, (T*)y->p, y->sn, y->sh, y->sw, x->h, x->w, x->c,
                                       u ? (T*)u->p : nullptr, u ? u->sn : 0, u ? u->sh : 0, u ? u->sw : 0, dbgbuf)));
      }
      MGDT_CHECK_LAUNCH("dwconv7_ln_fwd(row)");
      if (dbgbuf) {
        const int nwg = x->n * tiles;
        std::vector<unsigned long long> h((size_t)nwg * 6);
        (void)hipStreamSynchronize((hipStream_t)s);
        (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t5 = 0; double ph[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 6]); t5 = std::max(t5, h[i * 6 + 5]); for (int j = 0; j < 5; ++j) ph[j] += (double)(h[i * 6 + j + 1] - h[i * 6 + j]); }
        fprintf(stderr, "dwconv row x10ns: span %llu; avg per WG: halo %.0f weights+sync %.0f taps %.0f layernorm %.0f store %.0f\n", t5 - t0, ph[0] / nwg, ph[1] / nwg,
                ph[2] / nwg, ph[3] / nwg, ph[4] / nwg);
      }
      return MGDT_OK;
    }
    const size_t lds_t = (((size_t)(DW_TH + 6) * (DW_TW + 6) * x->c * dtype_size(dtype) + 15) & ~(size_t)15) + (size_t)49 * x->c * 4 + (size_t)DW_TH * DW_TW * Qt * 4;
    if (Qt <= 64 && lds_t <= 64 * 1024) {
      const int tiles = cdiv(x->h, DW_TH) * cdiv(x->w, DW_TW);
      const int need = cdiv(DW_TH * DW_TW, 256 / Qt);      // pixels per thread
#define DWT_L(MP) MGDT_DISPATCH_DTYPE(dtype, (dwconv7_ln_tiled_kernel<T, MP><<<x->n * tiles, 256, lds_t, (hipStream_t)s>>>( \
                                     (const T*)x->p, x->sn, x->sh, x->sw, dw_w, dw_b, ln_w, ln_b, eps, (T*)y->p, y->sn, y->sh, y->sw, x->h, x->w, x->c, \
                                     u ? (T*)u->p : nullptr, u ? u->sn : 0, u ? u->sh : 0, u ? u->sw : 0)))
      if (need <= 4) DWT_L(4); else if (need <= 7) DWT_L(7); else if (need <= 8) DWT_L(8); else DWT_L(16);
#undef DWT_L
      MGDT_CHECK_LAUNCH("dwconv7_ln_fwd(tiled)");
      return MGDT_OK;
    }
  }
  int Q = x->c / 4, PPB = 256 / Q;
  long M = (long)x->n * x->h * x->w;
  MGDT_DISPATCH_DTYPE(dtype, (dwconv7_ln_kernel<T><<<cdiv(M, PPB), 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, dw_w, dw_b, ln_w, ln_b, eps,
                                                                                       (T*)y->p, y->sn, y->sh, y->sw, x->n, x->h, x->w, x->c,
                                                                                       u ? (T*)u->p : nullptr, u ? u->sn : 0, u ? u->sh : 0, u ? u->sw : 0)));
  MGDT_CHECK_LAUNCH("dwconv7_ln_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ GRN statistics
// ws[n][split][c] = partial sum_{h,w in band} t^2 (fixed order -> deterministic); then
// scale[n][c] = gamma[c] * sqrt(sum) / (mean_c sqrt(sum) + 1e-6) + 1
#define GRN_SPLITS MGDT_GRN_SPLITS
template <typename T>
__global__ __launch_bounds__(256) void grn_sumsq_kernel(const T* __restrict__ t, long sn, long sh, long sw, int H, int W, int C,
                                                        float* __restrict__ ws) {
  const int n = blockIdx.x, q0 = blockIdx.y * 16, split = blockIdx.z;   // 16 quads = 64 channels per block
  const int ql = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int q = q0 + ql;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q * 4 < C) {
    const int npix = H * W;
    const int p0 = (int)((long)split * npix / GRN_SPLITS), p1 = (int)((long)(split + 1) * npix / GRN_SPLITS);
    for (int p = p0 + pr; p < p1; p += 16) {
      int yy = p / W, xx = p - yy * W;
      f32x4 v = load4<T>(t + n * sn + yy * sh + xx * sw + q * 4);
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
    if (c < C) ws[((long)n * GRN_SPLITS + split) * C + c] = sum;
  }
}

__global__ __launch_bounds__(256) void grn_scale_kernel(const float* __restrict__ ws, const float* __restrict__ gamma, int C,
                                                        float* __restrict__ scale) {
  const int n = blockIdx.x;
  extern __shared__ float gx[];   // [C]
  __shared__ float red[256];
  float part = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sum = 0.f;
    for (int sp = 0; sp < GRN_SPLITS; ++sp) sum += ws[((long)n * GRN_SPLITS + sp) * C + c];
    gx[c] = sqrtf(sum);
    part += gx[c];
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  float mean = red[0] / (float)C;
  for (int c = threadIdx.x; c < C; c += 256) scale[(long)n * C + c] = gamma[c] * (gx[c] / (mean + 1e-6f)) + 1.f;
}

extern "C" int mgdt_grn_stats_fwd(const mgdt_view* t, const float* gamma, float* ws, float* scale, int dtype, mgdt_stream s) {
  if (!view_ok(t) || !gamma || !ws || !scale) MGDT_FAIL(MGDT_BAD_ARG, "grn_stats: null/empty argument");
  if (!vec4_ok(t, dtype) || t->c > 8192) MGDT_FAIL(MGDT_BAD_SHAPE, "grn_stats: NHWC view, c%%4==0, c<=8192");
  dim3 grid(t->n, cdiv(t->c, 64), GRN_SPLITS);
  MGDT_DISPATCH_DTYPE(dtype, (grn_sumsq_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>((const T*)t->p, t->sn, t->sh, t->sw, t->h, t->w, t->c, ws)));
  grn_scale_kernel<<<t->n, 256, t->c * sizeof(float), (hipStream_t)s>>>(ws, gamma, t->c, scale);
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

// LDS-transposing variant: a block takes 128 consecutive anchors, loads their NHWC rows coalesced into an fp32 tile [128][no+1]
// (odd row stride -> conflict-free column reads), then writes y[n][ch][a] with consecutive lanes = consecutive anchors.
#define DEC_A 128
template <typename T>
__global__ __launch_bounds__(256) void detect_decode_tile_kernel(const T* __restrict__ f, long sn, long sh, long sw, int H, int W, int R,
                                                                 int nc, float stride, int a_off, int a_total, float* __restrict__ y) {
  extern __shared__ float tile[];
  const int no = 4 * R + nc, ld = no + 1, HW = H * W;
  const int n = blockIdx.y, a0 = blockIdx.x * DEC_A;
  const int na = min(DEC_A, HW - a0), Q = no / 4;
  for (int i = threadIdx.x; i < na * Q; i += 256) {
    int al = i / Q, q = i - al * Q, a = a0 + al;
    int oy = a / W, ox = a - oy * W;
    f32x4 v = load4<T>(f + n * sn + oy * sh + ox * sw + q * 4);
    float* d = tile + al * ld + q * 4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  __syncthreads();
  float* yo = y + (long)n * (4 + nc) * a_total + a_off + a0;
  if (threadIdx.x < na) {   // box: DFL softmax-expectation per side, dist2bbox, stride
    const int al = threadIdx.x, a = a0 + al;
    const float* p = tile + al * ld;
    float dd[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
      float mx = -INFINITY;
      for (int k = 0; k < R; ++k) mx = fmaxf(mx, p[side * R + k]);
      float den = 0.f, num = 0.f;
      for (int k = 0; k < R; ++k) {
        float e = expf(p[side * R + k] - mx);
        den += e;
        num += e * (float)k;
      }
      dd[side] = num / den;
    }
    int oy = a / W, ox = a - oy * W;
    float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
    float x1 = ax - dd[0], y1 = ay - dd[1], x2 = ax + dd[2], y2 = ay + dd[3];
    yo[al] = (x1 + x2) / 2.f * stride;
    yo[(long)a_total + al] = (y1 + y2) / 2.f * stride;
    yo[2L * a_total + al] = (x2 - x1) * stride;
    yo[3L * a_total + al] = (y2 - y1) * stride;
  }
  for (int i = threadIdx.x; i < nc * DEC_A; i += 256) {   // cls: sigmoid, lanes = consecutive anchors of one channel plane
    int c = i / DEC_A, al = i - c * DEC_A;
    if (al < na) yo[(long)(4 + c) * a_total + al] = 1.f / (1.f + expf(-tile[al * ld + 4 * R + c]));
  }
}

extern "C" int mgdt_detect_decode_fwd(const mgdt_view* feat, int reg_max, int nc, float stride, int a_off, int a_total, float* y,
                                      int dtype, mgdt_stream s) {
  if (!view_ok(feat) || !y) MGDT_FAIL(MGDT_BAD_ARG, "detect_decode: null/empty argument");
  if (feat->sc != 1 || feat->c != 4 * reg_max + nc || reg_max < 1 || a_off < 0 || a_off + feat->h * feat->w > a_total)
    MGDT_FAIL(MGDT_BAD_SHAPE, "detect_decode: c=%d reg_max=%d nc=%d a_off=%d a_total=%d", feat->c, reg_max, nc, a_off, a_total);
  const int no = feat->c;
  size_t lds = (size_t)DEC_A * (no + 1) * sizeof(float);
  if (vec4_ok(feat, dtype) && lds <= 144 * 1024) {
    if (lds > 64 * 1024) {      // reg_max = 16 heads (TOODHead): the transposed tile needs more than the default dynamic LDS limit
      static std::atomic<bool> attr_set{false};
      if (!attr_set) {
        for (const void* k : {(const void*)detect_decode_tile_kernel<float>, (const void*)detect_decode_tile_kernel<bf16>}) {
          hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
          if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "detect_decode: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        attr_set = true;
      }
    }
    dim3 grid(cdiv((long)feat->h * feat->w, DEC_A), feat->n);
    MGDT_DISPATCH_DTYPE(dtype, (detect_decode_tile_kernel<T><<<grid, 256, lds, (hipStream_t)s>>>((const T*)feat->p, feat->sn, feat->sh, feat->sw, feat->h, feat->w,
                                                                                                 reg_max, nc, stride, a_off, a_total, y)));
    MGDT_CHECK_LAUNCH("detect_decode_fwd");
    return MGDT_OK;
  }
  long total = (long)feat->n * feat->h * feat->w;
  MGDT_DISPATCH_DTYPE(dtype, (detect_decode_kernel<T><<<grid_for(total, 64), 64, 0, (hipStream_t)s>>>((const T*)feat->p, feat->sn, feat->sh, feat->sw, feat->n,
                                                                                                feat->h, feat->w, reg_max, nc, stride, a_off, a_total, y)));
  MGDT_CHECK_LAUNCH("detect_decode_fwd");
  return MGDT_OK;
}
