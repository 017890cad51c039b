// Channel-vectorised forms (4 channels, 8 / 16 bytes per access) of the element-wise and gather kernels of the training path; each
// mgdt_v4_* returns false when a view does not qualify (channels not a multiple of 4, strides or base not 4-element aligned) and the
// caller runs its scalar kernel.  Same arithmetic per element as the scalar kernels in train.hip / train_gd.hip (same order of the
// gather sums), so both give the same bits.
#include "common.h"

static bool v4_ok(const mgdt_view* v, int dtype) {
  return v && v->p && v->sc == 1 && v->c % 4 == 0 && v->sw % 4 == 0 && v->sh % 4 == 0 && v->sn % 4 == 0 && (uintptr_t)v->p % (4 * dtype_size(dtype)) == 0;
}
static inline int v4_grid(long n) { return (int)std::min<long>((n + 255) / 256, 16384); }
__device__ __forceinline__ int v4_bin_start(int o, int isz, int osz) { return (int)(((long)o * isz) / osz); }
__device__ __forceinline__ int v4_bin_end(int o, int isz, int osz) { return (int)(((long)(o + 1) * isz + osz - 1) / osz); }

#define DECODE_Q(i, v, n, h, w, c) \
  const int Q_ = (v).c >> 2;       \
  const int c = (int)((i) % Q_) * 4; \
  long t_ = (i) / Q_;              \
  const int w = (int)(t_ % (v).w); \
  t_ /= (v).w;                     \
  const int h = (int)(t_ % (v).h); \
  const long n = t_ / (v).h;
#define P4(T, v, n, h, w, c) ((T*)(v).p + ((n) * (v).sn + (h) * (v).sh + (w) * (v).sw + (c)))
#define FOR_QUADS(v) \
  const long total = (long)(v).n * (v).h * (v).w * ((v).c >> 2); \
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256)

template <typename T>
__global__ __launch_bounds__(256) void v4_add_kernel(const mgdt_view a, const mgdt_view b, const mgdt_view o) {
  FOR_QUADS(o) {
    DECODE_Q(i, o, n, h, w, c)
    store4<T>(P4(T, o, n, h, w, c), load4<T>(P4(const T, a, n, h, w, c)) + load4<T>(P4(const T, b, n, h, w, c)));
  }
}
bool mgdt_v4_add(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int dtype, hipStream_t st) {
  if (!v4_ok(a, dtype) || !v4_ok(b, dtype) || !v4_ok(o, dtype)) return false;
  const long nq = (long)o->n * o->h * o->w * (o->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_add_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*a, *b, *o)));
  return true;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void v4_ew_binary_kernel(const mgdt_view a, const mgdt_view b, const mgdt_view o) {
  FOR_QUADS(o) {
    DECODE_Q(i, o, n, h, w, c)
    const f32x4 x = load4<T>(P4(const T, a, n, h, w, c)), y = load4<T>(P4(const T, b, n, h, w, c));
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) r[j] = x[j] * y[j];
      else if (MODE == 1) r[j] = (y[j] > -3.f && y[j] < 3.f) ? x[j] * (1.f / 6.f) : 0.f;
      else if (MODE == 2) r[j] = x[j] * (fminf(fmaxf(y[j] + 3.f, 0.f), 6.f) / 6.f);
      else r[j] = fminf(fmaxf(y[j] + 3.f, 0.f), 6.f) / 6.f;
    }
    store4<T>(P4(T, o, n, h, w, c), r);
  }
}
bool mgdt_v4_ew_binary(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int mode, int dtype, hipStream_t st) {
  if (!v4_ok(a, dtype) || !v4_ok(b, dtype) || !v4_ok(o, dtype) || mode < 0 || mode > 3) return false;
  const long nq = (long)o->n * o->h * o->w * (o->c / 4);
#define EWB(M) MGDT_DISPATCH_DTYPE(dtype, (v4_ew_binary_kernel<T, M><<<v4_grid(nq), 256, 0, st>>>(*a, *b, *o)))
  if (mode == 0) EWB(0); else if (mode == 1) EWB(1); else if (mode == 2) EWB(2); else EWB(3);
#undef EWB
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void v4_channel_affine_kernel(const mgdt_view x, const float* __restrict__ scale, const float* __restrict__ shift, const mgdt_view y) {
  FOR_QUADS(x) {
    DECODE_Q(i, x, n, h, w, c)
    f32x4 v = load4<T>(P4(const T, x, n, h, w, c));
    if (scale) v = v * *(const f32x4*)(scale + n * x.c + c);
    if (shift) v += *(const f32x4*)(shift + c);
    store4<T>(P4(T, y, n, h, w, c), v);
  }
}
bool mgdt_v4_channel_affine(const mgdt_view* x, const float* scale, const float* shift, const mgdt_view* y, int dtype, hipStream_t st) {
  if (!v4_ok(x, dtype) || !v4_ok(y, dtype)) return false;
  const long nq = (long)x->n * x->h * x->w * (x->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_channel_affine_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*x, scale, shift, *y)));
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void v4_avgpool_bwd_kernel(const mgdt_view gy, const mgdt_view gx, int accumulate) {
  const int fy = (gx.h % gy.h == 0 && gx.w % gy.w == 0) ? gx.h / gy.h : 0, fx = fy ? gx.w / gy.w : 0;
  FOR_QUADS(gx) {
    DECODE_Q(i, gx, n, h, w, c)
    if (fy > 0) {                                        // integer pooling factors: the pixel lies in exactly one bin of fy x fx pixels
      f32x4 acc = load4<T>(P4(const T, gy, n, h / fy, w / fx, c));
      const float cnt = (float)(fy * fx);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = acc[j] / cnt;
      if (accumulate) acc += load4<T>(P4(const T, gx, n, h, w, c));
      store4<T>(P4(T, gx, n, h, w, c), acc);
      continue;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = max(0, (int)(((long)h * gy.h) / gx.h) - 1), oy_hi = min(gy.h - 1, (int)(((long)(h + 1) * gy.h) / gx.h) + 1);
    const int ox_lo = max(0, (int)(((long)w * gy.w) / gx.w) - 1), ox_hi = min(gy.w - 1, (int)(((long)(w + 1) * gy.w) / gx.w) + 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const int y0 = v4_bin_start(oy, gx.h, gy.h), y1 = v4_bin_end(oy, gx.h, gy.h);
      if (h < y0 || h >= y1) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int x0 = v4_bin_start(ox, gx.w, gy.w), x1 = v4_bin_end(ox, gx.w, gy.w);
        if (w < x0 || w >= x1) continue;
        const f32x4 g = load4<T>(P4(const T, gy, n, oy, ox, c));
        const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += g[j] / cnt;
      }
    }
    if (accumulate) acc += load4<T>(P4(const T, gx, n, h, w, c));
    store4<T>(P4(T, gx, n, h, w, c), acc);
  }
}
bool mgdt_v4_avgpool_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, hipStream_t st) {
  if (!v4_ok(gy, dtype) || !v4_ok(gx, dtype)) return false;
  const long nq = (long)gx->n * gx->h * gx->w * (gx->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_avgpool_bwd_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*gy, *gx, accumulate)));
  return true;
}

struct V4Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ V4Lerp v4_lerp_of(int o, int isz, int osz) {
  float scale = (float)isz / (float)osz;
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > isz - 1) i0 = isz - 1;
  int i1 = i0 + (i0 < isz - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  return V4Lerp{i0, i1, 1.f - l1, l1};
}
template <typename T>
__global__ __launch_bounds__(256) void v4_bilinear_bwd_kernel(const mgdt_view gy, const mgdt_view gx, int accumulate) {
  const float ry = (float)gy.h / (float)gx.h, rx = (float)gy.w / (float)gx.w;
  const bool x2 = gy.h == 2 * gx.h && gy.w == 2 * gx.w;       // exact 2x up-sampling: interior pixels have a fixed 4x4 stencil
  FOR_QUADS(gx) {
    DECODE_Q(i, gx, n, h, w, c)
    if (x2 && h >= 1 && h < gx.h - 1 && w >= 1 && w < gx.w - 1) {
      // src = o/2 - 0.25: outputs 2h-1, 2h, 2h+1, 2h+2 reach input h with weights 0.25, 0.75, 0.75, 0.25 (exactly what lerp_of yields there); same
      // summation order as the generic scan below, so both paths give the same bits
      const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
          const f32x4 g = load4<T>(P4(const T, gy, n, 2 * h - 1 + dy, 2 * w - 1 + dx, c));
          const float ww = wt[dy] * wt[dx];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += ww * g[j];
        }
      if (accumulate) acc += load4<T>(P4(const T, gx, n, h, w, c));
      store4<T>(P4(T, gx, n, h, w, c), acc);
      continue;
    }
    const int oy_lo = max(0, (int)floorf(((float)h - 1.f) * ry) - 1), oy_hi = min(gy.h - 1, (int)ceilf(((float)h + 2.f) * ry) + 1);
    const int ox_lo = max(0, (int)floorf(((float)w - 1.f) * rx) - 1), ox_hi = min(gy.w - 1, (int)ceilf(((float)w + 2.f) * rx) + 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const V4Lerp ly = v4_lerp_of(oy, gx.h, gy.h);
      const float wy = (ly.i0 == h ? ly.l0 : 0.f) + (ly.i1 == h ? ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const V4Lerp lx = v4_lerp_of(ox, gx.w, gy.w);
        const float wx = (lx.i0 == w ? lx.l0 : 0.f) + (lx.i1 == w ? lx.l1 : 0.f);
        if (wx == 0.f) continue;
        const f32x4 g = load4<T>(P4(const T, gy, n, oy, ox, c));
        const float ww = wy * wx;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += ww * g[j];
      }
    }
    if (accumulate) acc += load4<T>(P4(const T, gx, n, h, w, c));
    store4<T>(P4(T, gx, n, h, w, c), acc);
  }
}
bool mgdt_v4_bilinear_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, hipStream_t st) {
  if (!v4_ok(gy, dtype) || !v4_ok(gx, dtype)) return false;
  const long nq = (long)gx->n * gx->h * gx->w * (gx->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_bilinear_bwd_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*gy, *gx, accumulate)));
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void v4_grn_bwd_apply_kernel(const mgdt_view g, const mgdt_view t, const float* __restrict__ scale, const float* __restrict__ coef, const mgdt_view dt) {
  FOR_QUADS(t) {
    DECODE_Q(i, t, n, h, w, c)
    const f32x4 gv = load4<T>(P4(const T, g, n, h, w, c)), tv = load4<T>(P4(const T, t, n, h, w, c));
    const f32x4 sc = *(const f32x4*)(scale + n * t.c + c), cf = *(const f32x4*)(coef + n * t.c + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gv[j] * sc[j] + cf[j] * tv[j];
    store4<T>(P4(T, dt, n, h, w, c), o);
  }
}
bool mgdt_v4_grn_bwd_apply(const mgdt_view* g, const mgdt_view* t, const float* scale, const float* coef, const mgdt_view* dt, int dtype, hipStream_t st) {
  if (!v4_ok(g, dtype) || !v4_ok(t, dtype) || !v4_ok(dt, dtype)) return false;
  const long nq = (long)t->n * t->h * t->w * (t->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_grn_bwd_apply_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*g, *t, scale, coef, *dt)));
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void v4_spr_out_bwd_kernel(const mgdt_view gy, const float* __restrict__ attn, const float* __restrict__ dpooled, const mgdt_view gx) {
  const int H = gx.h, W = gx.w;
  const int hs1 = v4_bin_start(1, H, 2), he0 = v4_bin_end(0, H, 2), ws1 = v4_bin_start(1, W, 2), we0 = v4_bin_end(0, W, 2);
  FOR_QUADS(gx) {
    DECODE_Q(i, gx, n, h, w, c)
    const bool t0 = h < he0, t1 = h >= hs1, l0 = w < we0, l1 = w >= ws1;
    const f32x4 g = load4<T>(P4(const T, gy, n, h, w, c)), at = *(const f32x4*)(attn + n * gx.c + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* dp = dpooled + (n * gx.c + c + j) * 5;
      float v = g[j] * at[j] + dp[0];
      if (t0 && l0) v += dp[1];
      if (t0 && l1) v += dp[2];
      if (t1 && l0) v += dp[3];
      if (t1 && l1) v += dp[4];
      o[j] = v;
    }
    store4<T>(P4(T, gx, n, h, w, c), o);
  }
}
bool mgdt_v4_spr_out_bwd(const mgdt_view* gy, const float* attn, const float* dpooled, const mgdt_view* gx, int dtype, hipStream_t st) {
  if (!v4_ok(gy, dtype) || !v4_ok(gx, dtype)) return false;
  const long nq = (long)gx->n * gx->h * gx->w * (gx->c / 4);
  MGDT_DISPATCH_DTYPE(dtype, (v4_spr_out_bwd_kernel<T><<<v4_grid(nq), 256, 0, st>>>(*gy, attn, dpooled, *gx)));
  return true;
}

// out[n][c] = sum_{h,w} a*b: workgroup (image, split) covers all channels, thread = (channel quad, pixel lane); partial[n][split][c]
template <typename T>
__global__ __launch_bounds__(256) void v4_nc_reduce_partial_kernel(const mgdt_view a, const mgdt_view b, float* __restrict__ partial, int nsplit) {
  const int n = blockIdx.x, split = blockIdx.y, C = a.c, Q = C >> 2;
  const int QB = Q < 256 ? Q : 256, PL = 256 / QB;
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB;
  const int npix = a.h * a.w;
  const int p0 = (int)((long)split * npix / nsplit), p1 = (int)((long)(split + 1) * npix / nsplit);
  __shared__ float red[4][256];
  for (int q0 = 0; q0 < Q; q0 += QB) {
    const int q = q0 + ql;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (q < Q && pl < PL) {
      int p = p0 + pl;
      for (; p + 3 * PL < p1; p += 4 * PL) {                // four pixels in flight: the loop is a chain of dependent-latency round trips otherwise
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = p + u * PL, yy = pp / a.w, xx = pp - yy * a.w;
          v[u] = load4<T>(P4(const T, a, (long)n, yy, xx, 4 * q));
          if (b.p) v[u] = v[u] * load4<T>(P4(const T, b, (long)n, yy, xx, 4 * q));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
      }
      for (; p < p1; p += PL) {
        const int yy = p / a.w, xx = p - yy * a.w;
        f32x4 v = load4<T>(P4(const T, a, (long)n, yy, xx, 4 * q));
        if (b.p) v = v * load4<T>(P4(const T, b, (long)n, yy, xx, 4 * q));
        acc += v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * QB; o += 256) {
      const int qq = o >> 2, j = o & 3;
      if (q0 + qq >= Q) continue;
      float t = 0.f;
      for (int z = 0; z < PL; ++z) t += red[j][z * QB + qq];
      partial[((long)n * nsplit + split) * C + (q0 + qq) * 4 + j] = t;
    }
  }
}
bool mgdt_v4_nc_reduce_partial(const mgdt_view* a, const mgdt_view* b, float* partial, int nsplit, int dtype, hipStream_t st) {
  if (!v4_ok(a, dtype) || (b && b->p && !v4_ok(b, dtype))) return false;
  mgdt_view bb;
  memset(&bb, 0, sizeof(bb));
  if (b && b->p) bb = *b;
  MGDT_DISPATCH_DTYPE(dtype, (v4_nc_reduce_partial_kernel<T><<<dim3(a->n, nsplit), 256, 0, st>>>(*a, bb, partial, nsplit)));
  return true;
}

// MaxPool2d(5,1,2) backward with the image's map in LDS: workgroup = (image, block of CB channels); pass 1 finds every window's argmax
// (ATen's scan order: first maximum in (ky,kx) order, a NaN replaces what was found before it), pass 2 gathers gy over the windows whose
// argmax is the pixel, in (oy, ox) order.  All three LDS maps carry a 2-pixel halo (x: -inf, argmax: a value no pixel has, gy: 0), so the 25-tap
// loops have no bounds tests: their LDS reads are independent and pipeline (with the tests every read waited for the previous compare: 107 us).
template <typename T>
__global__ __launch_bounds__(256) void v4_maxpool5_bwd_kernel(const mgdt_view x, const mgdt_view gy, float* __restrict__ gx_f32, int CB) {
  extern __shared__ float lds[];
  const int H = x.h, W = x.w, HW = H * W, n = blockIdx.x, c0 = blockIdx.y * CB;
  const int HP = H + 4, WP = W + 4, PP = HP * WP;
  float* xs = lds;                                      // [PP][CB]
  float* gs = lds + (size_t)PP * CB;                    // [PP][CB] gy
  unsigned short* am = (unsigned short*)(gs + (size_t)PP * CB);   // [PP][CB] argmax = padded index of the window's maximum
  const int QB = CB >> 2;
  for (int i = threadIdx.x; i < PP * QB; i += 256) {
    const int pp = i / QB, q = i - pp * QB;
    const int yy = pp / WP - 2, xx = pp % WP - 2;
    const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && c0 + 4 * q < x.c;
    *(f32x4*)(xs + (size_t)pp * CB + 4 * q) = in ? load4<T>(P4(const T, x, (long)n, yy, xx, c0 + 4 * q)) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    *(f32x4*)(gs + (size_t)pp * CB + 4 * q) = in ? load4<T>(P4(const T, gy, (long)n, yy, xx, c0 + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = threadIdx.x; i < PP * CB; i += 256) am[i] = 0xFFFFu;
  __syncthreads();
  for (int i = threadIdx.x; i < HW * CB; i += 256) {
    const int p = i / CB, c = i - p * CB;
    const int oy = p / W, ox = p - oy * W;
    const int base = (oy * WP + ox) * CB + c;            // padded (oy, ox) = window's top-left corner
    float v[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) v[t] = xs[base + ((t / 5) * WP + (t % 5)) * CB];
    float best = -INFINITY;
    int bt = 12;                                         // a window whose every value is -inf (cannot happen for in-image data) keeps its centre
#pragma unroll
    for (int t = 0; t < 25; ++t)
      if (v[t] > best || isnan(v[t])) { best = v[t]; bt = t; }      // halo values (-inf) never win: `>` is strict
    am[((oy + 2) * WP + ox + 2) * CB + c] = (unsigned short)((oy + bt / 5) * WP + ox + bt % 5);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW * CB; i += 256) {
    const int p = i / CB, c = i - p * CB;
    if (c0 + c >= x.c) continue;
    const int h = p / W, w = p - h * W;
    const int me = (h + 2) * WP + w + 2;
    const int base = (h * WP + w) * CB + c;              // padded (h, w): the first window (oy = h - 2, ox = w - 2) that contains the pixel
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      const int o = base + ((t / 5) * WP + (t % 5)) * CB;
      acc += am[o] == me ? gs[o] : 0.f;
    }
    gx_f32[((long)n * HW + p) * x.c + c0 + c] = acc;
  }
}
bool mgdt_v4_maxpool5_bwd(const mgdt_view* x, const mgdt_view* gy, float* gx_f32, int dtype, hipStream_t st) {
  if (!v4_ok(x, dtype) || !v4_ok(gy, dtype)) return false;
  const long PP = (long)(x->h + 4) * (x->w + 4);
  if (PP > 65534) return false;
  int CB = 16;
  while (CB > 4 && PP * CB * 10 > 64 * 1024) CB >>= 1;      // x (4 B) + gy (4 B) + argmax (2 B) per padded element within the default 64 KB
  if (PP * CB * 10 > 64 * 1024) return false;
  const size_t lds = (size_t)PP * CB * 10;
  MGDT_DISPATCH_DTYPE(dtype, (v4_maxpool5_bwd_kernel<T><<<dim3(x->n, (x->c + CB - 1) / CB), 256, lds, st>>>(*x, *gy, gx_f32, CB)));
  return true;
}
