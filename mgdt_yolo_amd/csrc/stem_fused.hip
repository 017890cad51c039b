// The two stride-2 3x3 convolutions at the top of every YOLOv8 graph (layers 0 and 1: 3 -> 16 -> 32 channels at scale n) in ONE launch,
// bf16 inference path.  Reference: nn/modules/conv.py:25-42 (Conv = conv + BN + SiLU), models/v8/*.yaml rows 0-1; the uint8 / 255
// of the predictor's preprocess (yolo/engine/predictor.py:128-129) is folded into the loader.
//
// Why: unfused, layer 0 reads the image with 27 scattered 2-byte loads per output pixel and does 432 VALU FMAs for it (the stem kernel is
// VALU- and load-issue-bound at ~1.5 TB/s), writes 105 MB (B=32, 640^2) that layer 1 immediately reads back.  Here a workgroup owns an
// 8 x 16 tile of layer 1's output: the 35 x 80 input patch (3 planes) comes in with 16-byte row loads, layer 0 runs on MFMA straight out
// of LDS (K = 9 (plane, row) combos x 4 consecutive columns, the first column's weight is zero so that every fragment piece is an aligned
// 4-byte LDS word: no packing instructions), its 17 x 33 x 16 output stays in LDS as the NHWC map layer 1 (an ordinary implicit GEMM,
// K = 144) reads; only layer 1's output goes to HBM.
#include <type_traits>

#include "conv_igemm_kernel.h"

struct StemArgs {
  const void* x; long xsn, xsc, xsh, xsw;     // element strides of the NCHW image
  const char* w0; const float* b0;            // layer 0: [2 chunks][64 lanes][16 B] bf16 fragments (see mgdt_stem2_pack) + bias[16]
  const char* w1; const float* b1;            // layer 1: mgdt_conv_pack(16, 32, 3, bf16) panel + bias[32]
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  int N, H, W, H0, W0, H1, W1, tiles_x, tiles_y, total, per_xcd, fast;
};

constexpr int ST_TH = 8, ST_TW = 16;                 // layer-1 output tile
constexpr int ST_R0H = 2 * ST_TH + 1, ST_R0W = 2 * ST_TW + 1;      // layer-0 region 17 x 33
constexpr int ST_XH = 2 * ST_R0H + 1, ST_XWV = 80, ST_XW = 88;     // input patch 35 rows x 80 columns (row pitch 88 elements)
constexpr int ST_PS = 48;                            // layer-0 map: 16 channels (32 B) + 16 B pad per pixel
constexpr int ST_RP0 = ST_R0H * ST_R0W, ST_RP0A = (ST_RP0 + 15) / 16 * 16;

__device__ __forceinline__ float stem_silu(float v) { return v * fast_sigmoid(v); }

template <typename TX>
__global__ __launch_bounds__(256) void stem2_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 X[3][ST_XH][ST_XW];
  __shared__ __attribute__((aligned(16))) char Y0[ST_RP0A * ST_PS];
  __shared__ float lut[std::is_same<TX, uint8_t>::value ? 256 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int v = blockIdx.x;
  const int tlin = (v & 7) * a.per_xcd + (v >> 3);             // XCD k takes a contiguous range of tiles (halo rows shared in its L2)
  if ((v >> 3) >= a.per_xcd || tlin >= a.total) return;
  const int tpi = a.tiles_x * a.tiles_y;
  const int n = tlin / tpi, trem = tlin - n * tpi;
  const int ty0 = (trem / a.tiles_x) * ST_TH, tx0 = (trem % a.tiles_x) * ST_TW;

  // layer-1 weights: 5 K chunks x 2 cout blocks, requested first, used last
  bf16x8 A1[5][2];
#pragma unroll
  for (int kc = 0; kc < 5; ++kc)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) A1[kc][nb] = *(const bf16x8*)(a.w1 + ((size_t)(kc * 2 + nb) * 64 + lane) * 16);
  const bf16x8 A0a = *(const bf16x8*)(a.w0 + (size_t)lane * 16), A0b = *(const bf16x8*)(a.w0 + (size_t)(64 + lane) * 16);

  if constexpr (std::is_same<TX, uint8_t>::value) lut[tid] = __fdiv_rn((float)tid, 255.f);     // the preprocess division, exact
  // ---- input patch: rows 4*ty0-3 .. +34, columns 4*tx0-8 .. +79 of the three planes, zero outside the image
  const int iy0 = 4 * ty0 - 3, ix0 = 4 * tx0 - 8;
  const TX* xb = (const TX*)a.x + (long)n * a.xsn;
  if (a.fast) {                  // bf16 image, rows contiguous and 16-byte aligned, W % 8 == 0: whole 8-element vectors are inside or outside
    // all of a thread's 16-byte requests are issued before the first LDS store (clamped address, zeroed afterwards when outside the image): with a
    // bounds test around each load the 4-5 requests of a thread ran as dependent round trips
    constexpr int NI = 3 * ST_XH * (ST_XWV / 8), NU = (NI + 255) / 256;
    uint4 val[NU];
    bool inb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = min(tid + u * 256, NI - 1);
      const int pl = i / (ST_XH * (ST_XWV / 8)), rem = i - pl * (ST_XH * (ST_XWV / 8));
      const int row = rem / (ST_XWV / 8), seg = rem - row * (ST_XWV / 8);
      const int iy = iy0 + row, ix = ix0 + seg * 8;
      inb[u] = (unsigned)iy < (unsigned)a.H && ix >= 0 && ix + 8 <= a.W;
      const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 8);
      val[u] = *(const uint4*)(xb + (long)pl * a.xsc + (long)iyc * a.xsh + ixc);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = tid + u * 256;
      if (i < NI) {
        const int pl = i / (ST_XH * (ST_XWV / 8)), rem = i - pl * (ST_XH * (ST_XWV / 8));
        const int row = rem / (ST_XWV / 8), seg = rem - row * (ST_XWV / 8);
        *(uint4*)&X[pl][row][seg * 8] = inb[u] ? val[u] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  } else {
    if constexpr (std::is_same<TX, uint8_t>::value) __syncthreads();      // lut
    for (int i = tid; i < 3 * ST_XH * ST_XWV; i += 256) {
      const int pl = i / (ST_XH * ST_XWV), rem = i - pl * (ST_XH * ST_XWV);
      const int row = rem / ST_XWV, col = rem - row * ST_XWV;
      const int iy = iy0 + row, ix = ix0 + col;
      float val = 0.f;
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const TX e = xb[(long)pl * a.xsc + (long)iy * a.xsh + (long)ix * a.xsw];
        if constexpr (std::is_same<TX, uint8_t>::value) val = lut[e];
        else val = (float)e;
      }
      X[pl][row][col] = (bf16)val;
    }
  }
  __syncthreads();

  // ---- layer 0 on the 17 x 33 region: K = 9 (plane, ky) combos x 4 columns {2x-2 (zero weight), 2x-1, 2x, 2x+1}
  {
    const f32x4 bias0 = *(const f32x4*)(a.b0 + 4 * g);
    const int y00 = 2 * ty0 - 1, x00 = 2 * tx0 - 1;            // image coordinates of the region's first layer-0 pixel
    for (int grp = wave; grp < ST_RP0A / 16; grp += 4) {
      const int q = grp * 16 + r;
      const int qq = q < ST_RP0 ? q : ST_RP0 - 1;
      const int ry = qq / ST_R0W, rx = qq - ry * ST_R0W;
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u4;
      u4 f0, f1 = {0u, 0u, 0u, 0u};
      {
        const int cb = 2 * g;                                   // chunk 0: combos 2g, 2g+1 (< 8)
        const bf16* p0 = &X[cb / 3][2 * ry + cb % 3][4 + 2 * rx];
        const bf16* p1 = &X[(cb + 1) / 3][2 * ry + (cb + 1) % 3][4 + 2 * rx];
        f0[0] = *(const unsigned*)p0; f0[1] = *(const unsigned*)(p0 + 2);
        f0[2] = *(const unsigned*)p1; f0[3] = *(const unsigned*)(p1 + 2);
      }
      if (g == 0) {                                             // chunk 1: combo 8 = (plane 2, ky 2), the rest of K is padding
        const bf16* p0 = &X[2][2 * ry + 2][4 + 2 * rx];
        f1[0] = *(const unsigned*)p0; f1[1] = *(const unsigned*)(p0 + 2);
      }
      f32x4 acc = bias0;
      acc = mma(A0a, __builtin_bit_cast(bf16x8, f0), acc);
      acc = mma(A0b, __builtin_bit_cast(bf16x8, f1), acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = stem_silu(acc[j]);
      const int yy = y00 + ry, xx = x00 + rx;
      if (!((unsigned)yy < (unsigned)a.H0 && (unsigned)xx < (unsigned)a.W0)) acc = f32x4{0.f, 0.f, 0.f, 0.f};   // layer 1's zero padding
      if (q < ST_RP0) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)acc[j];
        *(bf16x4*)(Y0 + q * ST_PS + 8 * g) = o;
      }
    }
  }
  __syncthreads();

  // ---- layer 1: output row oy of the tile = one 16-pixel group (lane r = ox); taps at region pixel (2*oy + ky, 2*r + kx)
  {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
    int boff[5];
#pragma unroll
    for (int kc = 0; kc < 5; ++kc) {
      const int p = kc * 4 + g;
      int tap = p >> 1, cp = p & 1;
      if (tap >= 9) { tap = 4; cp = 0; }                        // padded piece: zero weights
      boff[kc] = ((tap / 3) * ST_R0W + tap % 3) * ST_PS + cp * 16;
    }
    f32x4 bias1[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bias1[nb] = *(const f32x4*)(a.b1 + nb * 16 + 4 * g);
    for (int oy = wave; oy < ST_TH; oy += 4) {
      const char* pin = Y0 + ((2 * oy) * ST_R0W + 2 * r) * ST_PS;
      f32x4 acc[2] = {bias1[0], bias1[1]};
#pragma unroll
      for (int kc = 0; kc < 5; ++kc) {
        const bf16x8 B = *(const bf16x8*)(pin + boff[kc]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mma(A1[kc][nb], B, acc[nb]);
      }
      const int gy = ty0 + oy, gx = tx0 + r;
      const int yo = (gy < a.H1 && gx < a.W1) ? n * a.ysn + gy * a.ysh + gx * a.ysw : MGDT_OOB;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[nb][j] = stem_silu(acc[nb][j]);
        bstore4<bf16>(yrs, (uint32_t)yo + (uint32_t)((nb * 16 + 4 * g) * 2), acc[nb]);
      }
    }
  }
}

// layer-0 weights in fragment order: w_folded fp32 [16][3][3][3] (BN already folded) -> bf16 [2][64][8]
__global__ void stem2_pack_kernel(const float* __restrict__ w, bf16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, c = i >> 9;
  const int k = c * 32 + 8 * (lane >> 4) + j, cout = lane & 15;
  const int cb = k >> 2, e = k & 3;                               // combo = plane * 3 + ky; e = column 0 (zero weight), 1..3 = kx 0..2
  float v = 0.f;
  if (cb < 9 && e >= 1) v = w[((cout * 3 + cb / 3) * 3 + cb % 3) * 3 + (e - 1)];
  out[i] = (bf16)v;
}

extern "C" size_t mgdt_stem2_packed_bytes(void) { return 2 * 64 * 16; }

/* w_folded: fp32 [16][3][3][3] with the BatchNorm scale already multiplied in (fuse_conv_and_bn's W'); packed: mgdt_stem2_packed_bytes() */
extern "C" int mgdt_stem2_pack(const float* w_folded, void* packed, mgdt_stream s) {
  if (!w_folded || !packed) MGDT_FAIL(MGDT_BAD_ARG, "stem2_pack: null pointer");
  stem2_pack_kernel<<<4, 256, 0, (hipStream_t)s>>>(w_folded, (bf16*)packed);
  MGDT_CHECK_LAUNCH("stem2_pack");
  return MGDT_OK;
}

/* y = SiLU(conv1(SiLU(conv0(x)))): x = N x 3 x H x W image (NCHW, any strides; x_dtype MGDT_BF16 / MGDT_F32 / MGDT_U8 (u8: / 255 on the fly)),
 * conv0 = 3x3 s2 3 -> 16 (packed0 from mgdt_stem2_pack + bias0[16]), conv1 = 3x3 s2 16 -> 32 (packed1 = mgdt_conv_pack(16, 32, 3, bf16) +
 * bias1[32]); y = N x H1 x W1 x 32 bf16 NHWC view. */
extern "C" int mgdt_stem2_fwd(const mgdt_view* x, int x_dtype, const void* packed0, const float* bias0, const void* packed1, const float* bias1,
                              const mgdt_view* y, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !packed0 || !bias0 || !packed1 || !bias1) MGDT_FAIL(MGDT_BAD_ARG, "stem2: null/empty argument");
  if (x->c != 3 || y->c != 32 || y->sc != 1 || y->sw % 4 || y->sh % 4 || y->sn % 4 || (uintptr_t)y->p % 8) MGDT_FAIL(MGDT_BAD_SHAPE, "stem2: x must have 3 channels, y 32 (NHWC, 8-byte aligned)");
  const int H0 = (x->h - 1) / 2 + 1, W0 = (x->w - 1) / 2 + 1, H1 = (H0 - 1) / 2 + 1, W1 = (W0 - 1) / 2 + 1;
  if (y->n != x->n || y->h != H1 || y->w != W1) MGDT_FAIL(MGDT_BAD_SHAPE, "stem2: y is %dx%dx%d, expected %dx%dx%d", y->n, y->h, y->w, x->n, H1, W1);
  const long yext = ((long)(y->n - 1) * y->sn + (long)(y->h - 1) * y->sh + (long)(y->w - 1) * y->sw + y->c) * 2;
  if (yext >= 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "stem2: y spans >= 2 GiB");
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x->p; a.xsn = x->sn; a.xsc = x->sc; a.xsh = x->sh; a.xsw = x->sw;
  a.w0 = (const char*)packed0; a.b0 = bias0; a.w1 = (const char*)packed1; a.b1 = bias1;
  a.y = (char*)y->p; a.ysn = (int)(y->sn * 2); a.ysh = (int)(y->sh * 2); a.ysw = (int)(y->sw * 2); a.y_bytes = (uint32_t)yext;
  a.N = x->n; a.H = x->h; a.W = x->w; a.H0 = H0; a.W0 = W0; a.H1 = H1; a.W1 = W1;
  a.tiles_x = cdiv(W1, ST_TW); a.tiles_y = cdiv(H1, ST_TH);
  a.total = a.N * a.tiles_x * a.tiles_y;
  a.per_xcd = cdiv(a.total, 8);
  a.fast = x_dtype == MGDT_BF16 && x->sw == 1 && x->w % 8 == 0 && x->sh % 8 == 0 && x->sc % 8 == 0 && x->sn % 8 == 0 && (uintptr_t)x->p % 16 == 0;
  const int grid = 8 * a.per_xcd;
  hipStream_t st = (hipStream_t)s;
  if (x_dtype == MGDT_BF16) stem2_kernel<bf16><<<grid, 256, 0, st>>>(a);
  else if (x_dtype == MGDT_F32) stem2_kernel<float><<<grid, 256, 0, st>>>(a);
  else if (x_dtype == MGDT_U8) stem2_kernel<uint8_t><<<grid, 256, 0, st>>>(a);
  else MGDT_FAIL(MGDT_BAD_DTYPE, "stem2: image dtype %d", x_dtype);
  MGDT_CHECK_LAUNCH("stem2_fwd");
  return MGDT_OK;
}
