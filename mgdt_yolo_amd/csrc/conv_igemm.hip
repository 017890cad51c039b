// Fused implicit-GEMM convolution on MFMA for gfx950 (fp32 exact path and bf16 path).
//
// GEMM view:  D[cout][pixel] = sum_k  Wp[cout][k] * X[pixel][k],  k = (tap, cin)
//   * the WEIGHTS are the MFMA "A" operand (rows = 16 output channels), the PIXELS the "B" operand (cols = 16
//     output pixels).  The accumulator then holds, per lane, 4 CONSECUTIVE output channels of one pixel, i.e.
//     an 8-byte (bf16) / 16-byte (fp32) contiguous NHWC store - no LDS transpose in the epilogue.
//   * K is cut into 16-byte "pieces" (4 fp32 / 8 bf16 consecutive input channels of one tap); a chunk = 4 pieces
//     = one MFMA fragment column group (lane>>4).  Activations go global -> VGPR straight in fragment order
//     (each lane loads the 16 bytes it will feed to the MFMA; for 1x1 convs a wave-instruction covers whole
//     contiguous NHWC lines), so there is no LDS round trip and no barrier for them.
//   * weights are pre-packed in fragment order (1 KiB per [chunk][16 couts]) and staged ONCE per workgroup in
//     LDS (linear, conflict-free ds_read_b128); when the whole panel fits (<=128 KiB) the workgroup is
//     persistent over pixel tiles and the main loop has no barrier at all.
//   * fused: second input add (MSPA sp+spx), per-(image,channel) input affine (GRN / attention scale),
//     bias (folded BN), activation, up to two residual adds, channel-sliced input/output views (chunk/cat).
#include "common.h"

template <typename T> struct Piece;
template <> struct Piece<float> { static constexpr int PE = 4; typedef f32x4 frag; };
template <> struct Piece<bf16> { static constexpr int PE = 8; typedef bf16x8 frag; };

struct ConvArgs {
  const char* x; long xsn, xsh, xsw;
  const char* x2; long x2sn, x2sh, x2sw;
  const float* in_scale; const float* in_shift;
  const char* wpk; const float* bias;
  char* y; long ysn, ysh, ysw;
  const char* r1; long r1sn, r1sh, r1sw;
  const char* r2; long r2sn, r2sh, r2sw;
  int N, H, W, Cin, Ho, Wo, Cout;
  int KS, stride, pad, CP, nchunks, NTtot, act;
  int M, HoWo, numTiles, seg_chunks, nseg, tab_bytes;
};

// ------------------------------------------------------------------------------------------------ packing
template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cin, int Cout, int KS,
                            int CP, int nchunks, int NTtot, T* __restrict__ out) {
  constexpr int PE = Piece<T>::PE;
  long total = (long)nchunks * NTtot * 64 * PE;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int j = (int)(i % PE);
    long t = i / PE;
    int lane = (int)(t % 64);
    t /= 64;
    int nb = (int)(t % NTtot);
    int kc = (int)(t / NTtot);
    int r = lane & 15, g = lane >> 4;
    int p = kc * 4 + g;
    int tap = p / CP, cp = p % CP;
    int cin = cp * PE + j, cout = nb * 16 + r;
    float v = 0.f;
    if (tap < KS * KS && cin < Cin && cout < Cout) {
      v = w[(((long)cout * Cin + cin) * KS + tap / KS) * KS + tap % KS];
      if (scale) v *= scale[cout];
    }
    out[i] = (T)v;
  }
}

// scale[c] = gamma/sqrt(var+eps); bias_out[c] = beta - gamma*mean/sqrt(var+eps) (+ scale*conv_bias); padded with 0
__global__ void fold_kernel(const float* cb, const float* g, const float* b, const float* mu, const float* var, float eps,
                            int Cout, int Cpad, float* scale, float* bias_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cpad) return;
  float s = 1.f, bo = 0.f;
  if (c < Cout) {
    if (g) {
      s = g[c] / sqrtf(eps + var[c]);
      bo = b[c] - g[c] * mu[c] / sqrtf(var[c] + eps);
      if (cb) bo += s * cb[c];
    } else if (cb) {
      bo = cb[c];
    }
  }
  scale[c] = s;
  bias_out[c] = bo;
}

// ------------------------------------------------------------------------------------------------ device helpers
template <typename T> __device__ __forceinline__ typename Piece<T>::frag zero_frag();
template <> __device__ __forceinline__ f32x4 zero_frag<float>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ bf16x8 zero_frag<bf16>() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

template <typename T>
__device__ __forceinline__ typename Piece<T>::frag frag_add(typename Piece<T>::frag a, typename Piece<T>::frag b) {
  typename Piece<T>::frag o;
#pragma unroll
  for (int i = 0; i < Piece<T>::PE; ++i) o[i] = (T)((float)a[i] + (float)b[i]);
  return o;
}

template <typename T>
__device__ __forceinline__ typename Piece<T>::frag frag_affine(typename Piece<T>::frag a, const float* sc, const float* sh) {
  typename Piece<T>::frag o;
#pragma unroll
  for (int i = 0; i < Piece<T>::PE; ++i) {
    float v = (float)a[i];
    if (sc) v *= sc[i];
    if (sh) v += sh[i];
    o[i] = (T)v;
  }
  return o;
}

__device__ __forceinline__ f32x4 mma(f32x4 w, f32x4 p, f32x4 acc) {
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], p[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x4 mma(bf16x8 w, bf16x8 p, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, p, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ main kernel
template <typename T, int NT, int MT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  typedef typename Piece<T>::frag frag;
  constexpr int PE = Piece<T>::PE;
  constexpr int BM = 4 * MT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* ptab = (uint32_t*)smem;
  char* wlds = smem + a.tab_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int nb0 = blockIdx.y * NT;

  // piece table: (dy, dx, channel offset) per 16-byte K piece; 0xFFFFFFFF = zero padding piece
  for (int p = tid; p < a.nchunks * 4; p += 256) {
    int tap = p / a.CP, cp = p % a.CP;
    uint32_t e = 0xFFFFFFFFu;
    if (tap < a.KS * a.KS) e = (uint32_t)(tap / a.KS) | ((uint32_t)(tap % a.KS) << 8) | ((uint32_t)(cp * PE) << 16);
    ptab[p] = e;
  }

  auto stage = [&](int seg) {
    const int c0 = seg * a.seg_chunks;
    const int nc = min(a.seg_chunks, a.nchunks - c0);
    const int nblk = nc * NT;  // 1 KiB blocks
    for (int i = tid; i < nblk * 64; i += 256) {
      int blk = i >> 6, l = i & 63;
      int kc = blk / NT, nt = blk % NT;
      const uint4* src = (const uint4*)(a.wpk + ((long)(c0 + kc) * a.NTtot + nb0 + nt) * 1024) + l;
      ((uint4*)(wlds + (long)blk * 1024))[l] = *src;
    }
  };
  if (a.nseg == 1) stage(0);
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    int pn[MT], py[MT], px[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = tile * BM + (wave * MT + mt) * 16 + r;
      pv[mt] = m < a.M;
      int mm = pv[mt] ? m : 0;
      int n = mm / a.HoWo, rem = mm - n * a.HoWo;
      int oy = rem / a.Wo;
      pn[mt] = n; py[mt] = oy; px[mt] = rem - oy * a.Wo;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int seg = 0; seg < a.nseg; ++seg) {
      if (a.nseg > 1) {
        __syncthreads();
        stage(seg);
        __syncthreads();
      }
      const int c0 = seg * a.seg_chunks;
      const int nc = min(a.seg_chunks, a.nchunks - c0);
      for (int kc = 0; kc < nc; ++kc) {
        const uint32_t e = ptab[(c0 + kc) * 4 + g];
        const bool pvalid = e != 0xFFFFFFFFu;
        const int dy = e & 0xFF, dx = (e >> 8) & 0xFF, ch = (int)(e >> 16);
        frag P[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int iy = py[mt] * a.stride - a.pad + dy, ix = px[mt] * a.stride - a.pad + dx;
          bool ok = pvalid && pv[mt] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          frag v = zero_frag<T>();
          if (ok) {
            v = *(const frag*)(a.x + (pn[mt] * a.xsn + iy * a.xsh + ix * a.xsw + ch) * (long)sizeof(T));
            if (a.x2) v = frag_add<T>(v, *(const frag*)(a.x2 + (pn[mt] * a.x2sn + iy * a.x2sh + ix * a.x2sw + ch) * (long)sizeof(T)));
            if (a.in_scale || a.in_shift)
              v = frag_affine<T>(v, a.in_scale ? a.in_scale + (long)pn[mt] * a.Cin + ch : nullptr,
                                 a.in_shift ? a.in_shift + ch : nullptr);
          }
          P[mt] = v;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const frag Wf = *(const frag*)(wlds + ((long)(kc * NT + nt) * 64 + lane) * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mma(Wf, P[mt], acc[nt][mt]);
        }
      }
    }

    // epilogue: lane holds couts (nb0+nt)*16 + 4g .. +3 of pixel (mt, r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!pv[mt]) continue;
      const long yo = pn[mt] * a.ysn + py[mt] * a.ysh + px[mt] * a.ysw;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = (nb0 + nt) * 16 + 4 * g;
        if (co >= a.Cout) continue;
        const f32x4 b = *(const f32x4*)(a.bias + co);
        f32x4 v = acc[nt][mt];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j] + b[j], a.act);
        if (a.r1) {
          f32x4 q = load4<T>((const T*)a.r1 + pn[mt] * a.r1sn + py[mt] * a.r1sh + px[mt] * a.r1sw + co);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += q[j];
        }
        if (a.r2) {
          f32x4 q = load4<T>((const T*)a.r2 + pn[mt] * a.r2sn + py[mt] * a.r2sh + px[mt] * a.r2sw + co);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += q[j];
        }
        store4<T>((T*)a.y + yo + co, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline int piece_elems(int dtype) { return dtype == MGDT_BF16 ? 8 : 4; }

static void conv_geometry(int cin, int cout, int k, int dtype, int* CP, int* nchunks, int* NTtot) {
  int pe = piece_elems(dtype);
  *CP = (cin + pe - 1) / pe;
  int pieces = k * k * (*CP);
  *nchunks = (pieces + 3) / 4;
  *NTtot = (cout + 15) / 16;
}

extern "C" size_t mgdt_conv_packed_bytes(int cin, int cout, int k, int dtype) {
  int CP, nchunks, NTtot;
  conv_geometry(cin, cout, k, dtype, &CP, &nchunks, &NTtot);
  // + scratch for the per-channel BN scale (fp32[cout_pad]) appended after the packed panel
  return (size_t)nchunks * NTtot * 1024 + (size_t)NTtot * 16 * sizeof(float);
}

extern "C" int mgdt_conv_pack(const float* w, const float* cb, const float* g, const float* b, const float* mu,
                              const float* var, float eps, int cin, int cout, int k, int dtype, void* packed,
                              float* bias_out, mgdt_stream s) {
  if (!w || !packed || !bias_out) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack: null pointer");
  if ((g != nullptr) != (b != nullptr) || (g != nullptr) != (mu != nullptr) || (g != nullptr) != (var != nullptr))
    MGDT_FAIL(MGDT_BAD_ARG, "conv_pack: BN arguments must be all present or all NULL");
  int CP, nchunks, NTtot;
  conv_geometry(cin, cout, k, dtype, &CP, &nchunks, &NTtot);
  hipStream_t st = (hipStream_t)s;
  float* scale = (float*)((char*)packed + (size_t)nchunks * NTtot * 1024);
  int cpad = NTtot * 16;
  fold_kernel<<<cdiv(cpad, 64), 64, 0, st>>>(cb, g, b, mu, var, eps, cout, cpad, scale, bias_out);
  long total = (long)nchunks * NTtot * 64 * piece_elems(dtype);
  int grid = (int)std::min<long>((total + 255) / 256, 4096);
  MGDT_DISPATCH_DTYPE(dtype, (pack_kernel<T><<<grid, 256, 0, st>>>(w, g ? scale : nullptr, cin, cout, k, CP, nchunks, NTtot, (T*)packed)));
  MGDT_CHECK_LAUNCH("conv_pack");
  return MGDT_OK;
}

template <typename T, int NT, int MT>
static int launch_igemm(const ConvArgs& a, int gx, int gy, size_t lds, hipStream_t st) {
  static bool attr_set = false;  // idempotent; racing setters write the same value
  auto kern = conv_igemm_kernel<T, NT, MT>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  kern<<<dim3(gx, gy), 256, lds, st>>>(a);
  MGDT_CHECK_LAUNCH("conv2d_fwd");
  return MGDT_OK;
}

template <typename T>
static int dispatch_igemm(const ConvArgs& a, int NT, int MT, int gx, int gy, size_t lds, hipStream_t st) {
#define CASE(nt, mt) \
  if (NT == nt && MT == mt) return launch_igemm<T, nt, mt>(a, gx, gy, lds, st);
  CASE(1, 4) CASE(2, 4) CASE(3, 4) CASE(4, 4) CASE(5, 2) CASE(6, 2) CASE(8, 2) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(4, 2)
#undef CASE
  MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: no kernel for NT=%d MT=%d", NT, MT);
}

extern "C" int mgdt_conv2d_fwd(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                               const void* packed_w, const float* bias, int k, int stride, int act, const mgdt_view* r1,
                               const mgdt_view* r2, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !packed_w || !bias) MGDT_FAIL(MGDT_BAD_ARG, "conv2d: null/empty view or weights");
  if (dtype != MGDT_F32 && dtype != MGDT_BF16) MGDT_FAIL(MGDT_BAD_DTYPE, "conv2d: dtype %d", dtype);
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: k=%d stride=%d unsupported", k, stride);
  const int pe = piece_elems(dtype);
  const int pad = k / 2;
  const int Ho = (x->h + 2 * pad - k) / stride + 1, Wo = (x->w + 2 * pad - k) / stride + 1;
  if (y->n != x->n || y->h != Ho || y->w != Wo) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: y is %dx%dx%d, expected %dx%dx%d", y->n, y->h, y->w, x->n, Ho, Wo);
  if (x->c % pe || y->c % 4) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: cin=%d must be a multiple of %d and cout=%d of 4", x->c, pe, y->c);
  auto aligned = [&](const mgdt_view* v, int q) {
    return v->sc == 1 && v->sw % q == 0 && v->sh % q == 0 && v->sn % q == 0 && ((uintptr_t)v->p % (q * dtype_size(dtype))) == 0;
  };
  if (!aligned(x, pe) || !aligned(y, 4)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: x/y must be NHWC views (sc==1) with 16-byte aligned pieces");
  if (x2 && x2->p && (x2->n != x->n || x2->h != x->h || x2->w != x->w || x2->c != x->c || !aligned(x2, pe)))
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: x2 must match x");
  for (const mgdt_view* rr : {r1, r2})
    if (rr && rr->p && (rr->n != y->n || rr->h != Ho || rr->w != Wo || rr->c != y->c || !aligned(rr, 4)))
      MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: residual must match y");

  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const char*)x->p; a.xsn = x->sn; a.xsh = x->sh; a.xsw = x->sw;
  if (x2 && x2->p) { a.x2 = (const char*)x2->p; a.x2sn = x2->sn; a.x2sh = x2->sh; a.x2sw = x2->sw; }
  a.in_scale = in_scale; a.in_shift = in_shift;
  a.wpk = (const char*)packed_w; a.bias = bias;
  a.y = (char*)y->p; a.ysn = y->sn; a.ysh = y->sh; a.ysw = y->sw;
  if (r1 && r1->p) { a.r1 = (const char*)r1->p; a.r1sn = r1->sn; a.r1sh = r1->sh; a.r1sw = r1->sw; }
  if (r2 && r2->p) { a.r2 = (const char*)r2->p; a.r2sn = r2->sn; a.r2sh = r2->sh; a.r2sw = r2->sw; }
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Ho = Ho; a.Wo = Wo; a.Cout = y->c;
  a.KS = k; a.stride = stride; a.pad = pad; a.act = act;
  conv_geometry(a.Cin, a.Cout, k, dtype, &a.CP, &a.nchunks, &a.NTtot);
  long M = (long)a.N * Ho * Wo;
  if (M > 0x7fffffffL || (long)x->n * x->sn > 0x7fffffffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: problem too large");
  a.M = (int)M; a.HoWo = Ho * Wo;

  // tile choice: NT = cout blocks per workgroup (prefer all of them: activations are then read once)
  int NT = 1;
  for (int c : {8, 6, 5, 4, 3, 2, 1})
    if (a.NTtot % c == 0) { NT = c; break; }
  int MT = NT <= 4 ? 4 : 2;
  auto wgs = [&](int nt, int mt) { return (long)cdiv(M, 64 * mt) * (a.NTtot / nt); };
  if (MT == 4 && wgs(NT, 4) < 512) MT = 2;                      // small maps: more, smaller tiles
  while (wgs(NT, MT) < 256 && NT > 1 && NT % 2 == 0) NT /= 2;   // ... and split the couts over workgroups
  a.numTiles = cdiv(M, 64 * MT);
  a.tab_bytes = ((a.nchunks * 16) + 15) & ~15;
  size_t panel = (size_t)a.nchunks * NT * 1024;
  if (panel + a.tab_bytes <= 128 * 1024) { a.seg_chunks = a.nchunks; a.nseg = 1; }
  else { a.seg_chunks = std::max(1, 64 / NT); a.nseg = cdiv(a.nchunks, a.seg_chunks); }
  size_t lds = a.tab_bytes + (size_t)a.seg_chunks * NT * 1024;
  int gx = std::min(a.numTiles, a.nseg == 1 ? 1024 : 4096), gy = a.NTtot / NT;
  hipStream_t st = (hipStream_t)s;
  if (dtype == MGDT_F32) return dispatch_igemm<float>(a, NT, MT, gx, gy, lds, st);
  return dispatch_igemm<bf16>(a, NT, MT, gx, gy, lds, st);
}
