// TOODHead pieces that the Detect path does not have (reference nn/modules/head.py:67-131, 466-572, block.py:401-432):
// GroupNorm(+act) on NHWC, the layer-attention MLP of TaskDecomposition, the modulated deformable 3x3 convolution (DCNv2, mmcv's
// ModulatedDeformConv2d - mmcv is not shipped with the reference, so this follows mmcv's published kernel; parity unpinned) and the
// per-pixel sigmoid gate.  Everything else of the head (3x3 / 1x1 convs, the dynamic reduction conv = 1x1 conv with a
// per-(image, channel) input scale, decode) runs on the existing conv / decode kernels.
#include <algorithm>

#include "conv_igemm_kernel.h"

static inline bool nhwc4(const mgdt_view* v, int dtype) {
  return v->sc == 1 && v->c % 4 == 0 && v->sw % 4 == 0 && v->sh % 4 == 0 && v->sn % 4 == 0 && ((uintptr_t)v->p % (4 * dtype_size(dtype))) == 0;
}

// ------------------------------------------------------------------------------------------------ GroupNorm (+ activation)
#define GN_SPLITS 16
// partial[n][split][c][2] = {sum, sum of squares} over a band of rows (fixed order -> deterministic)
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, long sn, long sh, long sw, int H, int W, int C, float* __restrict__ partial) {
  const int n = blockIdx.x, split = blockIdx.y;
  const int Q = C / 4;
  const int r0 = (int)((long)split * H / GN_SPLITS), r1 = (int)((long)(split + 1) * H / GN_SPLITS);
  const long npix = (long)(r1 - r0) * W;
  for (int q = threadIdx.x; q < Q; q += 256) {        // (channel quads are few: C <= 1024 in the heads)
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    for (long p = 0; p < npix; ++p) {
      const int yy = r0 + (int)(p / W), xx = (int)(p % W);
      const f32x4 v = load4<T>(x + n * sn + yy * sh + xx * sw + q * 4);
      s1 += v; s2 += v * v;
    }
    float* o = partial + (((long)n * GN_SPLITS + split) * C + q * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = s1[j]; o[2 * j + 1] = s2[j]; }
  }
}

// the same reduction with pixels spread over the threads of a block (used when there are many pixels per band)
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_wide_kernel(const T* __restrict__ x, long sn, long sh, long sw, int H, int W, int C,
                                                              float* __restrict__ partial) {
  const int n = blockIdx.x, split = blockIdx.y;
  const int Q = C / 4, PR = 256 / Q;                    // Q <= 64: PR >= 4 pixel lanes
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int r0 = (int)((long)split * H / GN_SPLITS), r1 = (int)((long)(split + 1) * H / GN_SPLITS);
  const long npix = (long)(r1 - r0) * W;
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (pl < PR)
    for (long p = pl; p < npix; p += PR) {
      const int yy = r0 + (int)(p / W), xx = (int)(p % W);
      const f32x4 v = load4<T>(x + n * sn + yy * sh + xx * sw + q * 4);
      s1 += v; s2 += v * v;
    }
  __shared__ float red[2][4][256];
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][j][threadIdx.x] = s1[j]; red[1][j][threadIdx.x] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < Q * 4) {
    const int qq = threadIdx.x / 4, j = threadIdx.x % 4;
    float a = 0.f, b = 0.f;
    for (int r = 0; r < PR; ++r) { a += red[0][j][r * Q + qq]; b += red[1][j][r * Q + qq]; }
    float* o = partial + (((long)n * GN_SPLITS + split) * C + qq * 4 + j) * 2;
    o[0] = a; o[1] = b;
  }
}

// A[n][c] = gamma[c] * rstd(group), B[n][c] = beta[c] - mean(group) * A  (biased variance, eps inside the sqrt: nn.GroupNorm)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int C, int G, float count, float eps, float* __restrict__ AB) {
  extern __shared__ float sm[];   // [C][2] channel sums, then [G][2] mean / rstd
  float* cs = sm;
  float* gs = sm + 2 * C;
  const int n = blockIdx.x, cg = C / G;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int sp = 0; sp < GN_SPLITS; ++sp) {
      const float* p = partial + (((long)n * GN_SPLITS + sp) * C + c) * 2;
      a += p[0]; b += p[1];
    }
    cs[2 * c] = a; cs[2 * c + 1] = b;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    float a = 0.f, b = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) { a += cs[2 * c]; b += cs[2 * c + 1]; }
    const float mean = a / (count * cg), var = fmaxf(b / (count * cg) - mean * mean, 0.f);
    gs[2 * g] = mean; gs[2 * g + 1] = 1.f / sqrtf(var + eps);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float a = gamma[c] * gs[2 * g + 1];
    AB[((long)n * C + c) * 2] = a;
    AB[((long)n * C + c) * 2 + 1] = beta[c] - gs[2 * g] * a;
  }
}

template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, const float* __restrict__ AB, int act, T* __restrict__ y, long ysn,
                                long ysh, long ysw, int H, int W, int C, long total) {
  const int Q = C / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q); long t = i / Q;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    f32x4 v = load4<T>(x + n * xsn + h * xsh + w * xsw + q * 4);
    const float* ab = AB + ((long)n * C + q * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_apply(fmaf(v[j], ab[2 * j], ab[2 * j + 1]), act);
    store4<T>(y + n * ysn + h * ysh + w * ysw + q * 4, v);
  }
}

extern "C" size_t mgdt_groupnorm_workspace_bytes(int n, int c) { return ((size_t)n * GN_SPLITS * c * 2 + (size_t)n * c * 2) * sizeof(float); }

extern "C" int mgdt_groupnorm_fwd(const mgdt_view* x, const float* gamma, const float* beta, int groups, float eps, int act, void* ws, const mgdt_view* y,
                                  int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !gamma || !beta || !ws) MGDT_FAIL(MGDT_BAD_ARG, "groupnorm: null/empty argument");
  if (!nhwc4(x, dtype) || !nhwc4(y, dtype) || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c || groups < 1 || x->c % groups || x->c > 4096)
    MGDT_FAIL(MGDT_BAD_SHAPE, "groupnorm: matching NHWC views, c%%4==0, c%%groups==0");
  hipStream_t st = (hipStream_t)s;
  float* partial = (float*)ws;
  float* AB = partial + (size_t)x->n * GN_SPLITS * x->c * 2;
  const int Q = x->c / 4;
  if (Q <= 64) MGDT_DISPATCH_DTYPE(dtype, (gn_partial_wide_kernel<T><<<dim3(x->n, GN_SPLITS), 256, 0, st>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w, x->c, partial)));
  else MGDT_DISPATCH_DTYPE(dtype, (gn_partial_kernel<T><<<dim3(x->n, GN_SPLITS), 256, 0, st>>>((const T*)x->p, x->sn, x->sh, x->sw, x->h, x->w, x->c, partial)));
  gn_finalize_kernel<<<x->n, 256, (size_t)(2 * x->c + 2 * groups) * sizeof(float), st>>>(partial, gamma, beta, x->c, groups, (float)x->h * x->w, eps, AB);
  const long total = (long)x->n * x->h * x->w * Q;
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  MGDT_DISPATCH_DTYPE(dtype, (gn_apply_kernel<T><<<grid, 256, 0, st>>>((const T*)x->p, x->sn, x->sh, x->sw, AB, act, (T*)y->p, y->sn, y->sh, y->sw, x->h, x->w,
                                                                        x->c, total)));
  MGDT_CHECK_LAUNCH("groupnorm_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ TaskDecomposition layer attention
// sums[n][c] = sum_hw feat (from mgdt_nc_reduce).  w = sigmoid(W2 relu(W1 avg + b1) + b2) in R^S; the reduction conv's weight is
// scaled per stacked block k by w[k] (head.py:117-123) == the INPUT channel k*feat + j is scaled by w[k]: scale[n][k*feat + j] = w[k].
__global__ __launch_bounds__(256) void tood_layer_attn_kernel(const float* __restrict__ sums, float inv_hw, const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, int C, int hid, int S,
                                                              float* __restrict__ scale) {
  extern __shared__ float sm[];   // avg[C] | h[hid] | wgt[S]
  float* avg = sm; float* hb = sm + C; float* wg = hb + hid;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) avg[c] = sums[(long)n * C + c] * inv_hw;
  __syncthreads();
  for (int j = threadIdx.x; j < hid; j += 256) {
    float a = b1[j];
    for (int c = 0; c < C; ++c) a = fmaf(w1[(long)j * C + c], avg[c], a);
    hb[j] = fmaxf(a, 0.f);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < S; k += 256) {
    float a = b2[k];
    for (int j = 0; j < hid; ++j) a = fmaf(w2[(long)k * hid + j], hb[j], a);
    wg[k] = 1.f / (1.f + expf(-a));
  }
  __syncthreads();
  const int feat = C / S;
  for (int c = threadIdx.x; c < C; c += 256) scale[(long)n * C + c] = wg[c / feat];
}

extern "C" int mgdt_tood_layer_attn_fwd(const float* sums, int n, int c, int hw, const float* w1, const float* b1, const float* w2, const float* b2, int hid,
                                        int stacked, float* scale, mgdt_stream s) {
  if (!sums || !w1 || !b1 || !w2 || !b2 || !scale) MGDT_FAIL(MGDT_BAD_ARG, "tood_layer_attn: null pointer");
  if (n < 1 || c < 1 || hid < 1 || stacked < 1 || c % stacked || hw < 1 || c > 8192) MGDT_FAIL(MGDT_BAD_SHAPE, "tood_layer_attn: c=%d hid=%d stacked=%d", c, hid, stacked);
  tood_layer_attn_kernel<<<n, 256, (size_t)(c + hid + stacked) * sizeof(float), (hipStream_t)s>>>(sums, 1.f / (float)hw, w1, b1, w2, b2, c, hid, stacked, scale);
  MGDT_CHECK_LAUNCH("tood_layer_attn_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ DCNv2 (modulated deformable conv 3x3, stride 1, pad 1)
// om: per pixel 18 offsets (dy, dx per kernel point, mmcv order) then 9 mask LOGITS (sigmoid applied here, head.py:525).
// One thread = one pixel x 16 output channels; the 9*Cin sampled-and-modulated values are produced on the fly (bilinear, zero outside,
// mmcv dmcn_im2col_bilinear) and contracted with the GEMM-ordered weights w[(tap*Cin + ci)][cout] held in LDS.
template <typename T>
__global__ __launch_bounds__(256) void dcnv2_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, const T* __restrict__ om, long osn, long osh, long osw,
                                                    const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, long ysn, long ysh,
                                                    long ysw, int N, int H, int W, int Cin, int Cout) {
  extern __shared__ float wl[];   // [9*Cin][16]
  const int co0 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < 9 * Cin * 16; i += 256) wl[i] = (co0 + i % 16 < Cout) ? w[(long)(i / 16) * Cout + co0 + (i % 16)] : 0.f;
  __syncthreads();
  const long m = blockIdx.x * 256L + threadIdx.x;
  const long M = (long)N * H * W;
  if (m >= M) return;
  const int n = (int)(m / ((long)H * W));
  const int rem = (int)(m - (long)n * H * W);
  const int oy = rem / W, ox = rem - oy * W;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;
  const T* op = om + n * osn + oy * osh + ox * osw;
  const T* xb = x + n * xsn;
  for (int tap = 0; tap < 9; ++tap) {
    const float hy = (float)(oy - 1 + tap / 3) + (float)op[2 * tap], wx = (float)(ox - 1 + tap % 3) + (float)op[2 * tap + 1];
    const float mk = 1.f / (1.f + expf(-(float)op[18 + tap]));
    if (!(hy > -1.f && wx > -1.f && hy < (float)H && wx < (float)W)) continue;
    const int h0 = (int)floorf(hy), w0 = (int)floorf(wx), h1 = h0 + 1, w1 = w0 + 1;
    const float lh = hy - (float)h0, lw = wx - (float)w0, hh = 1.f - lh, hw = 1.f - lw;
    const bool v1 = h0 >= 0 && w0 >= 0, v2 = h0 >= 0 && w1 <= W - 1, v3 = h1 <= H - 1 && w0 >= 0, v4 = h1 <= H - 1 && w1 <= W - 1;
    const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
    for (int ci = 0; ci < Cin; ++ci) {
      float val = 0.f;
      if (v1) val += c1 * (float)xb[h0 * xsh + w0 * xsw + ci];
      if (v2) val += c2 * (float)xb[h0 * xsh + w1 * xsw + ci];
      if (v3) val += c3 * (float)xb[h1 * xsh + w0 * xsw + ci];
      if (v4) val += c4 * (float)xb[h1 * xsh + w1 * xsw + ci];
      val *= mk;
      const f32x4* wv = (const f32x4*)(wl + (tap * Cin + ci) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = wv[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q * 4 + j] = fmaf(val, t[j], acc[q * 4 + j]);
      }
    }
  }
  T* yp = y + n * ysn + oy * ysh + ox * ysw + co0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (co0 + j < Cout) yp[j] = (T)acc[j];
}

// MFMA variant (bf16, cin % 8 == 0, cout % 16 == 0): DCNv2 is the implicit-GEMM convolution with a different activation gather - the
// K piece (tap, 8 channels) of pixel r is bilinearly sampled at the tap's offset position and multiplied by its mask instead of being
// read at the integer tap position.  Weights: the ordinary MFMA fragment panel of mgdt_conv_pack (k = 3, no BN), staged in LDS.
// Lane (r, g) of chunk kc produces piece kc*4 + g of pixel r: four 16-byte corner loads (clamped coordinates, zero weight outside, mmcv's
// dmcn_im2col_bilinear rules), 8-channel lerp, mask, pack -> B operand; NB MFMAs per chunk.
template <int NB>
__global__ __launch_bounds__(256) void dcnv2_mfma_kernel(const bf16* __restrict__ x, long xsn, long xsh, long xsw, const bf16* __restrict__ om, long osn,
                                                         long osh, long osw, const char* __restrict__ wpk, bf16* __restrict__ y, long ysn, long ysh, long ysw,
                                                         int N, int H, int W, int Cin, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char wlm[];  // [nchunks][NB][64][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  for (int i = tid; i < nchunks * NB * 64; i += 256) ((uint4*)wlm)[i] = ((const uint4*)wpk)[i];
  __syncthreads();
  const long M = (long)N * H * W;
  const int CP = Cin / 8;
  for (long tile = blockIdx.x * 4L + wave; tile * 16 < M; tile += gridDim.x * 4L) {
    const long m = tile * 16 + r;
    const bool pv = m < M;
    const long mm = pv ? m : 0;
    const int n = (int)(mm / ((long)H * W)), rem = (int)(mm - (long)n * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const bf16* op = om + n * osn + oy * osh + ox * osw;
    const bf16* xb = x + n * xsn;
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nchunks; ++kc) {
      const int p = kc * 4 + g, tap = p / CP, c0 = (p - tap * CP) * 8;
      bf16x8 frag;
#pragma unroll
      for (int e = 0; e < 8; ++e) frag[e] = (bf16)0.f;
      if (pv && tap < 9) {
        const float hy = (float)(oy - 1 + tap / 3) + (float)op[2 * tap], wx = (float)(ox - 1 + tap % 3) + (float)op[2 * tap + 1];
        if (hy > -1.f && wx > -1.f && hy < (float)H && wx < (float)W) {
          const float mk = 1.f / (1.f + expf(-(float)op[18 + tap]));
          const int h0 = (int)floorf(hy), w0 = (int)floorf(wx), h1 = h0 + 1, w1 = w0 + 1;
          const float lh = hy - (float)h0, lw = wx - (float)w0, hh = 1.f - lh, hw = 1.f - lw;
          const float c1 = (h0 >= 0 && w0 >= 0) ? hh * hw * mk : 0.f, c2 = (h0 >= 0 && w1 <= W - 1) ? hh * lw * mk : 0.f;
          const float c3 = (h1 <= H - 1 && w0 >= 0) ? lh * hw * mk : 0.f, c4 = (h1 <= H - 1 && w1 <= W - 1) ? lh * lw * mk : 0.f;
          const int h0c = max(h0, 0), w0c = max(w0, 0), h1c = min(h1, H - 1), w1c = min(w1, W - 1);
          const bf16x8 v1 = *(const bf16x8*)(xb + h0c * xsh + w0c * xsw + c0), v2 = *(const bf16x8*)(xb + h0c * xsh + w1c * xsw + c0);
          const bf16x8 v3 = *(const bf16x8*)(xb + h1c * xsh + w0c * xsw + c0), v4 = *(const bf16x8*)(xb + h1c * xsh + w1c * xsw + c0);
#pragma unroll
          for (int e = 0; e < 8; ++e) frag[e] = (bf16)(c1 * (float)v1[e] + c2 * (float)v2[e] + c3 * (float)v3[e] + c4 * (float)v4[e]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mma(*(const bf16x8*)(wlm + ((size_t)(kc * NB + nb) * 64 + lane) * 16), frag, acc[nb]);
    }
    if (pv) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) store4<bf16>(y + n * ysn + oy * ysh + ox * ysw + nb * 16 + 4 * g, acc[nb]);
    }
  }
}

/* DCNv2 on the matrix cores: packed_w from mgdt_conv_pack(w, NULL, NULL bn, cin, cout, k = 3, bf16) (no bias: DyDCNv2 has a norm). */
extern "C" int mgdt_dcnv2_mfma_fwd(const mgdt_view* x, const mgdt_view* offset_mask, const void* packed_w, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(offset_mask) || !view_ok(y) || !packed_w) MGDT_FAIL(MGDT_BAD_ARG, "dcnv2_mfma: null/empty argument");
  if (dtype != MGDT_BF16 || x->c % 8 || y->c % 16 || y->c > 64 || x->sc != 1 || y->sc != 1 || offset_mask->sc != 1 || offset_mask->c < 27 || x->n != y->n ||
      x->h != y->h || x->w != y->w || offset_mask->n != x->n || offset_mask->h != x->h || offset_mask->w != x->w || x->sw % 8 || x->sh % 8 || x->sn % 8 ||
      (uintptr_t)x->p % 16 || y->sw % 4 || y->sh % 4 || y->sn % 4 || (uintptr_t)y->p % 8)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dcnv2_mfma: bf16 NHWC views, cin %% 8 == 0, cout in {16, 32, 48, 64}, offset_mask with >= 27 channels");
  const int nchunks = (9 * (x->c / 8) + 3) / 4, NB = y->c / 16;
  const long M = (long)x->n * x->h * x->w;
  const int grid = (int)std::min<long>(cdiv(M, 64), 2048);
  const size_t lds = (size_t)nchunks * NB * 1024;
  if (lds > 144 * 1024) MGDT_FAIL(MGDT_BAD_SHAPE, "dcnv2_mfma: weight panel %zu B does not fit", lds);
  if (lds > 64 * 1024) {      // 64 -> 64 channels (the scale-s head): 72 KiB of dynamic LDS needs the opt-in (idempotent, same value from every caller)
    const void* ks[4] = {(const void*)dcnv2_mfma_kernel<1>, (const void*)dcnv2_mfma_kernel<2>, (const void*)dcnv2_mfma_kernel<3>, (const void*)dcnv2_mfma_kernel<4>};
    hipError_t e = hipFuncSetAttribute(ks[NB > 4 ? 3 : NB - 1], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "dcnv2_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
#define DCN_L(B) dcnv2_mfma_kernel<B><<<grid, 256, lds, (hipStream_t)s>>>((const bf16*)x->p, x->sn, x->sh, x->sw, (const bf16*)offset_mask->p, offset_mask->sn, \
                                                                         offset_mask->sh, offset_mask->sw, (const char*)packed_w, (bf16*)y->p, y->sn, y->sh, y->sw, \
                                                                         x->n, x->h, x->w, x->c, nchunks)
  switch (NB) { case 1: DCN_L(1); break; case 2: DCN_L(2); break; case 3: DCN_L(3); break; default: DCN_L(4); break; }
#undef DCN_L
  MGDT_CHECK_LAUNCH("dcnv2_mfma_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_dcnv2_fwd(const mgdt_view* x, const mgdt_view* offset_mask, const float* w_gemm, const float* bias, const mgdt_view* y, int dtype,
                              mgdt_stream s) {
  if (!view_ok(x) || !view_ok(offset_mask) || !view_ok(y) || !w_gemm) MGDT_FAIL(MGDT_BAD_ARG, "dcnv2: null/empty argument");
  if (x->sc != 1 || y->sc != 1 || offset_mask->sc != 1 || offset_mask->c < 27 || x->n != y->n || x->h != y->h || x->w != y->w || offset_mask->n != x->n ||
      offset_mask->h != x->h || offset_mask->w != x->w || x->c > 512)
    MGDT_FAIL(MGDT_BAD_SHAPE, "dcnv2: x, y, offset_mask must be NHWC views of one spatial size; offset_mask has >= 27 channels (18 offsets + 9 mask logits)");
  const long M = (long)x->n * x->h * x->w;
  dim3 grid((unsigned)cdiv(M, 256), (unsigned)cdiv(y->c, 16));
  const size_t lds = (size_t)9 * x->c * 16 * sizeof(float);
  MGDT_DISPATCH_DTYPE(dtype, (dcnv2_kernel<T><<<grid, 256, lds, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, (const T*)offset_mask->p, offset_mask->sn,
                                                                                   offset_mask->sh, offset_mask->sw, w_gemm, bias, (T*)y->p, y->sn, y->sh, y->sw,
                                                                                   x->n, x->h, x->w, x->c, y->c)));
  MGDT_CHECK_LAUNCH("dcnv2_fwd");
  return MGDT_OK;
}

// ------------------------------------------------------------------------------------------------ y = x * sigmoid(gate[n,h,w])
template <typename T>
__global__ void pixel_gate_kernel(const T* __restrict__ x, long xsn, long xsh, long xsw, const T* __restrict__ g, long gsn, long gsh, long gsw,
                                  T* __restrict__ y, long ysn, long ysh, long ysw, int H, int W, int C, long total) {
  const int Q = C / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q); long t = i / Q;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    const float gate = 1.f / (1.f + expf(-(float)g[n * gsn + h * gsh + w * gsw]));
    store4<T>(y + n * ysn + h * ysh + w * ysw + q * 4, load4<T>(x + n * xsn + h * xsh + w * xsw + q * 4) * gate);
  }
}

extern "C" int mgdt_pixel_gate_fwd(const mgdt_view* x, const mgdt_view* gate, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(gate) || !view_ok(y)) MGDT_FAIL(MGDT_BAD_ARG, "pixel_gate: null/empty view");
  if (!nhwc4(x, dtype) || !nhwc4(y, dtype) || gate->c != 1 || x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c || gate->n != x->n ||
      gate->h != x->h || gate->w != x->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "pixel_gate: x, y matching NHWC (c%%4==0), gate N x H x W x 1");
  const long total = (long)x->n * x->h * x->w * (x->c / 4);
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  MGDT_DISPATCH_DTYPE(dtype, (pixel_gate_kernel<T><<<grid, 256, 0, (hipStream_t)s>>>((const T*)x->p, x->sn, x->sh, x->sw, (const T*)gate->p, gate->sn, gate->sh,
                                                                                      gate->sw, (T*)y->p, y->sn, y->sh, y->sw, x->h, x->w, x->c, total)));
  MGDT_CHECK_LAUNCH("pixel_gate_fwd");
  return MGDT_OK;
}
