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
#include "conv_igemm_kernel.h"

// ------------------------------------------------------------------------------------------------ packing
// `dgrad`: pack the weights of the data-gradient convolution instead - rows = input channels of the original conv, K = (flipped tap,
// output channel): dx = conv(dy, w'[ci][co][ky][kx] = w[co][ci][KS-1-ky][KS-1-kx]) for stride 1.  Cin / Cout are then those of the
// dgrad conv (Cin = original cout, Cout = original cin).
// The BN fold of fold_kernel runs inline (same expressions, same bits): one launch per convolution instead of two - in training every
// convolution is re-packed after every optimizer step.  FoldArgs.g == nullptr: no BN (scale 1); bias_out == nullptr: the caller folds.
struct FoldArgs { const float *cb, *g, *b, *mu, *var; float eps; int cout_real, cpad; float* bias_out; };
template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cin, int Cout, int KS,
                            int CP, int nchunks, int NTtot, T* __restrict__ out, int dgrad = 0, FoldArgs fa = FoldArgs{}) {
  constexpr int PE = Piece<T>::PE;
  long total = (long)nchunks * NTtot * 64 * PE;
  if (fa.bias_out && blockIdx.x == 0)
    for (int c = threadIdx.x; c < fa.cpad; c += blockDim.x) {
      float bo = 0.f;
      if (c < fa.cout_real) {
        if (fa.g) {
          const float s = fa.g[c] / sqrtf(fa.eps + fa.var[c]);
          bo = fa.b[c] - fa.g[c] * fa.mu[c] / sqrtf(fa.var[c] + fa.eps);
          if (fa.cb) bo += s * fa.cb[c];
        } else if (fa.cb) {
          bo = fa.cb[c];
        }
      }
      fa.bias_out[c] = bo;
    }
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
    if (dgrad >= 2) {                                  // compact K: list index -> the phase's tap (rows {1} / {1,2}, columns likewise)
      const int nc_ = 1 + ((dgrad - 2) & 1), nr_ = 1 + ((dgrad - 2) >> 1);
      tap = tap < nr_ * nc_ ? (1 + tap / nc_) * 3 + (1 + tap % nc_) : 99;
    }
    int cin = cp * PE + j, cout = nb * 16 + r;
    float v = 0.f;
    if (tap < KS * KS && cin < Cin && cout < Cout) {
      if (dgrad == 1) v = w[(((long)cin * Cout + cout) * KS + (KS - 1 - tap / KS)) * KS + (KS - 1 - tap % KS)];   // original layout [cin=co][cout=ci][ky][kx]
      else if (dgrad >= 2) {   // stride-2 phase (py, px) of a 3x3 conv: tap (dy', dx') of the dy grid carries w[ky = py + 1 - 2*dy'][kx = px + 1 - 2*dx'] when that exists
        const int py = (dgrad - 2) >> 1, px = (dgrad - 2) & 1;
        const int ky = py + 1 - 2 * (tap / KS - 1), kx = px + 1 - 2 * (tap % KS - 1);
        v = ((unsigned)ky < 3u && (unsigned)kx < 3u) ? w[(((long)cin * Cout + cout) * 3 + ky) * 3 + kx] : 0.f;
      }
      else v = w[(((long)cout * Cin + cin) * KS + tap / KS) * KS + tap % KS];
      if (scale) v *= scale[cout];
      else if (fa.g) v *= fa.g[cout] / sqrtf(fa.eps + fa.var[cout]);
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

// ------------------------------------------------------------------------------------------------ host side
static inline int piece_elems(int dtype) { return dtype == MGDT_BF16 ? 8 : 4; }

// taps of phase (py, px) of a stride-2 3x3 data gradient (see pack_kernel): row taps {1} for py = 0, {1, 2} for py = 1, likewise the columns
static inline int phase_ntaps(int mode) { return mode < 2 ? 9 : (1 + ((mode - 2) >> 1)) * (1 + ((mode - 2) & 1)); }
static inline unsigned long long phase_taplist(int mode, int k) {
  unsigned long long l = 0;
  if (mode < 2) { for (int t = 0; t < k * k; ++t) l |= (unsigned long long)t << (4 * t); return l; }
  const int nr = 1 + ((mode - 2) >> 1), nc = 1 + ((mode - 2) & 1);
  for (int i = 0; i < nr * nc; ++i) l |= (unsigned long long)((1 + i / nc) * 3 + (1 + i % nc)) << (4 * i);
  return l;
}
static void conv_geometry(int cin, int cout, int k, int dtype, int* CP, int* nchunks, int* NTtot, int ntaps = 0) {
  int pe = piece_elems(dtype);
  *CP = (cin + pe - 1) / pe;
  int pieces = (ntaps > 0 ? ntaps : k * k) * (*CP);
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
  (void)scale;
  long total = (long)nchunks * NTtot * 64 * piece_elems(dtype);
  int grid = (int)std::min<long>((total + 255) / 256, 4096);
  const FoldArgs fa{cb, g, b, mu, var, eps, cout, cpad, bias_out};
  MGDT_DISPATCH_DTYPE(dtype, (pack_kernel<T><<<grid, 256, 0, st>>>(w, nullptr, cin, cout, k, CP, nchunks, NTtot, (T*)packed, 0, fa)));
  MGDT_CHECK_LAUNCH("conv_pack");
  return MGDT_OK;
}

// weights of the stride-1 data-gradient convolution (see pack_kernel): cin / cout are those of the ORIGINAL conv; the packed panel is
// used with mgdt_conv2d_fwd(x = dy, y = dx, k, stride 1); bias_out (cin padded to 16) is zero-filled.
extern "C" int mgdt_conv_pack_dgrad(const float* w, int cin, int cout, int k, int phase, int dtype, void* packed, float* bias_out, mgdt_stream s) {
  if (!w || !packed || !bias_out) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_dgrad: null pointer");
  if (phase < -1 || phase > 3 || (phase >= 0 && k != 3)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_pack_dgrad: phase %d (stride-2 phases need k = 3)", phase);
  int CP, nchunks, NTtot;
  conv_geometry(cout, cin, k, dtype, &CP, &nchunks, &NTtot, phase < 0 ? 0 : phase_ntaps(2 + phase));       // the dgrad conv maps cout -> cin channels; a phase keeps only its own taps
  hipStream_t st = (hipStream_t)s;
  float* scale = (float*)((char*)packed + (size_t)nchunks * NTtot * 1024);
  const int cpad = NTtot * 16;
  (void)scale;
  long total = (long)nchunks * NTtot * 64 * piece_elems(dtype);
  int grid = (int)std::min<long>((total + 255) / 256, 4096);
  const FoldArgs fa{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, cin, cpad, bias_out};      // zero bias
  MGDT_DISPATCH_DTYPE(dtype, (pack_kernel<T><<<grid, 256, 0, st>>>(w, nullptr, cout, cin, k, CP, nchunks, NTtot, (T*)packed, phase < 0 ? 1 : 2 + phase, fa)));
  MGDT_CHECK_LAUNCH("conv_pack_dgrad");
  return MGDT_OK;
}

// ---- fp8 (e4m3) panels - BASELINE configs[4].  Per output channel: BN-folded weights w' = w * gamma / sqrt(var + eps), w_scale = max|w'| / 448,
// panel bytes = e4m3(w' / w_scale) in the bf16 panel's fragment order (8 K values per lane: 512-byte blocks), oscale = w_scale / x_qscale
// (1 for the padding channels), bias_out as mgdt_conv_pack.  One workgroup per output channel computes the scale; the packing pass is elementwise.
__global__ void q8_wscale_kernel(const float* __restrict__ w, FoldArgs fa, int per_cout, float xq, float* __restrict__ wscale, float* __restrict__ oscale) {
  const int c = blockIdx.x;
  __shared__ float red[256];
  float m = 0.f;
  if (c < fa.cout_real)
    for (int i = threadIdx.x; i < per_cout; i += blockDim.x) m = fmaxf(m, fabsf(w[(long)c * per_cout + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float bo = 0.f, ws = 1.f;
    if (c < fa.cout_real) {
      float fold = 1.f;
      if (fa.g) {
        fold = fa.g[c] / sqrtf(fa.eps + fa.var[c]);
        bo = fa.b[c] - fa.g[c] * fa.mu[c] / sqrtf(fa.var[c] + fa.eps);
        if (fa.cb) bo += fold * fa.cb[c];
      } else if (fa.cb) {
        bo = fa.cb[c];
      }
      const float amax = red[0] * fabsf(fold);
      ws = amax > 0.f ? amax / 448.f : 1.f;
    }
    wscale[c] = ws;
    oscale[c] = ws / xq;
    fa.bias_out[c] = bo;
  }
}
__global__ void pack_q8_kernel(const float* __restrict__ w, FoldArgs fa, const float* __restrict__ wscale, int Cin, int Cout, int KS, int CP, int nchunks,
                               int NTtot, long* __restrict__ out) {
  typedef __attribute__((ext_vector_type(2))) int i32x2;
  const long total = (long)nchunks * NTtot * 64;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(t % 64);
    const int nb = (int)((t / 64) % NTtot), kc = (int)(t / 64 / NTtot);
    const int r = lane & 15, g = lane >> 4;
    const int p = kc * 4 + g, tap = p / CP, cp = p % CP, cout = nb * 16 + r;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cin = cp * 8 + j;
      float q = 0.f;
      if (tap < KS * KS && cin < Cin && cout < Cout) {
        q = w[(((long)cout * Cin + cin) * KS + tap / KS) * KS + tap % KS];
        if (fa.g) q *= fa.g[cout] / sqrtf(fa.eps + fa.var[cout]);
        q = __builtin_amdgcn_fmed3f(q / wscale[cout], -448.f, 448.f);
      }
      v[j] = q;
    }
    i32x2 o = {0, 0};
    o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], o[0], false);
    o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], o[0], true);
    o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], o[1], false);
    o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], o[1], true);
    out[t] = __builtin_bit_cast(long, o);
  }
}

extern "C" size_t mgdt_conv_packed_bytes_fp8(int cin, int cout, int k) {
  int CP, nchunks, NTtot;
  conv_geometry(cin, cout, k, MGDT_BF16, &CP, &nchunks, &NTtot);
  return (size_t)nchunks * NTtot * 512 + (size_t)NTtot * 16 * sizeof(float);   // + the per-channel weight scales (scratch of the packing pass)
}

extern "C" int mgdt_conv_pack_fp8(const float* w, const float* cb, const float* g, const float* b, const float* mu, const float* var, float eps,
                                  int cin, int cout, int k, float x_qscale, void* packed, float* bias_out, float* oscale_out, mgdt_stream s) {
  if (!w || !packed || !bias_out || !oscale_out) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_fp8: null pointer");
  if ((g != nullptr) != (b != nullptr) || (g != nullptr) != (mu != nullptr) || (g != nullptr) != (var != nullptr))
    MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_fp8: BN arguments must be all present or all NULL");
  if (!(x_qscale > 0.f) || !(x_qscale < 3.0e38f)) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_fp8: x_qscale must be a positive finite number");
  if (cin % 8 || cout % 4 || (k != 1 && k != 3)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_pack_fp8: cin=%d must be a multiple of 8, cout=%d of 4, k=%d 1 or 3", cin, cout, k);
  int CP, nchunks, NTtot;
  conv_geometry(cin, cout, k, MGDT_BF16, &CP, &nchunks, &NTtot);
  hipStream_t st = (hipStream_t)s;
  float* wscale = (float*)((char*)packed + (size_t)nchunks * NTtot * 512);
  const int cpad = NTtot * 16;
  const FoldArgs fa{cb, g, b, mu, var, eps, cout, cpad, bias_out};
  q8_wscale_kernel<<<cpad, 256, 0, st>>>(w, fa, cin * k * k, x_qscale, wscale, oscale_out);
  const long total = (long)nchunks * NTtot * 64;
  pack_q8_kernel<<<(int)std::min<long>((total + 255) / 256, 4096), 256, 0, st>>>(w, fa, wscale, cin, cout, k, CP, nchunks, NTtot, (long*)packed);
  MGDT_CHECK_LAUNCH("conv_pack_fp8");
  return MGDT_OK;
}

// ---- batched re-pack: after an optimizer step every live packed panel of a training model is refreshed in a handful of launches instead of one (two) per
// convolution (131 launches of ~4.5 us per step for the MSPA-GD n model).  One descriptor = one mgdt_conv_pack / mgdt_conv_pack_dgrad call.
struct PackJob { const float* w; FoldArgs fa; int Cin, Cout, KS, CP, nchunks, NTtot, dgrad; void* out; };
#define PACK_BATCH 16
struct PackJobs { PackJob j[PACK_BATCH]; };
template <typename T>
__global__ void pack_batch_kernel(const PackJobs jobs) {
  const PackJob& J = jobs.j[blockIdx.y];
  constexpr int PE = Piece<T>::PE;
  const long total = (long)J.nchunks * J.NTtot * 64 * PE;
  const FoldArgs fa = J.fa;
  if (fa.bias_out && blockIdx.x == 0)
    for (int c = threadIdx.x; c < fa.cpad; c += blockDim.x) {
      float bo = 0.f;
      if (c < fa.cout_real) {
        if (fa.g) {
          const float s = fa.g[c] / sqrtf(fa.eps + fa.var[c]);
          bo = fa.b[c] - fa.g[c] * fa.mu[c] / sqrtf(fa.var[c] + fa.eps);
          if (fa.cb) bo += s * fa.cb[c];
        } else if (fa.cb) {
          bo = fa.cb[c];
        }
      }
      fa.bias_out[c] = bo;
    }
  const int Cin = J.Cin, Cout = J.Cout, KS = J.KS, CP = J.CP, NTtot = J.NTtot, dgrad = J.dgrad;
  const float* __restrict__ w = J.w;
  T* __restrict__ out = (T*)J.out;
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
    if (dgrad >= 2) {                                  // compact K: list index -> the phase's tap (rows {1} / {1,2}, columns likewise)
      const int nc_ = 1 + ((dgrad - 2) & 1), nr_ = 1 + ((dgrad - 2) >> 1);
      tap = tap < nr_ * nc_ ? (1 + tap / nc_) * 3 + (1 + tap % nc_) : 99;
    }
    int cin = cp * PE + j, cout = nb * 16 + r;
    float v = 0.f;
    if (tap < KS * KS && cin < Cin && cout < Cout) {
      if (dgrad == 1) v = w[(((long)cin * Cout + cout) * KS + (KS - 1 - tap / KS)) * KS + (KS - 1 - tap % KS)];
      else if (dgrad >= 2) {
        const int py = (dgrad - 2) >> 1, px = (dgrad - 2) & 1;
        const int ky = py + 1 - 2 * (tap / KS - 1), kx = px + 1 - 2 * (tap % KS - 1);
        v = ((unsigned)ky < 3u && (unsigned)kx < 3u) ? w[(((long)cin * Cout + cout) * 3 + ky) * 3 + kx] : 0.f;
      }
      else v = w[(((long)cout * Cin + cin) * KS + tap / KS) * KS + tap % KS];
      if (fa.g) v *= fa.g[cout] / sqrtf(fa.eps + fa.var[cout]);
    }
    out[i] = (T)v;
  }
}

extern "C" int mgdt_conv_pack_batch(const mgdt_pack_desc* d, int n, mgdt_stream s) {
  if (!d || n < 0) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_batch: null descriptor array");
  hipStream_t st = (hipStream_t)s;
  for (int dtype : {MGDT_F32, MGDT_BF16}) {
    PackJobs jobs;
    int m = 0;
    auto flush = [&]() {
      if (!m) return;
      for (int q = m; q < PACK_BATCH; ++q) { jobs.j[q] = jobs.j[0]; jobs.j[q].nchunks = 0; jobs.j[q].fa.bias_out = nullptr; }     // unused slots: no work
      if (dtype == MGDT_F32) pack_batch_kernel<float><<<dim3(32, m), 256, 0, st>>>(jobs);
      else pack_batch_kernel<bf16><<<dim3(32, m), 256, 0, st>>>(jobs);
      m = 0;
    };
    for (int i = 0; i < n; ++i) {
      const mgdt_pack_desc& e = d[i];
      if (e.dtype != dtype) continue;
      if (!e.w || !e.packed || !e.bias_out) MGDT_FAIL(MGDT_BAD_ARG, "conv_pack_batch: descriptor %d has a null pointer", i);
      if (e.mode < 0 || e.mode > 5 || (e.mode >= 2 && e.k != 3)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv_pack_batch: descriptor %d mode %d", i, e.mode);
      PackJob& J = jobs.j[m++];
      const int gi = e.mode == 0 ? e.cin : e.cout, go = e.mode == 0 ? e.cout : e.cin;      // the data-gradient conv maps cout -> cin channels
      conv_geometry(gi, go, e.k, dtype, &J.CP, &J.nchunks, &J.NTtot, e.mode >= 2 ? phase_ntaps(e.mode) : 0);
      J.w = e.w; J.Cin = gi; J.Cout = go; J.KS = e.k; J.dgrad = e.mode; J.out = e.packed;
      J.fa = e.mode == 0 ? FoldArgs{e.conv_bias, e.bn_gamma, e.bn_beta, e.bn_mean, e.bn_var, e.bn_eps, go, J.NTtot * 16, e.bias_out}
                         : FoldArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, go, J.NTtot * 16, e.bias_out};
      if (m == PACK_BATCH) flush();
    }
    flush();
  }
  MGDT_CHECK_LAUNCH("conv_pack_batch");
  return MGDT_OK;
}

bool mgdt_conv3x3_lds_launch(const mgdt_view* x, const mgdt_view* y, const void* packed_w, const float* bias, int act, int CP, int nchunks, int NTtot, hipStream_t st,
                             const float* q8_oscale, float q8_xq, int stride);

template <typename T, int NT, int MT, bool Q8>
int launch_igemm(const ConvArgs& a, int gx, int gy, int threads, size_t lds, hipStream_t st);   // defined in conv_igemm_inst_*.hip

template <typename T>
static int dispatch_igemm(const ConvArgs& a, int NT, int MT, int gx, int gy, int threads, size_t lds, hipStream_t st) {
#define CASE(nt, mt) \
  if (NT == nt && MT == mt) return launch_igemm<T, nt, mt, false>(a, gx, gy, threads, lds, st);
  CASE(1, 2) CASE(1, 4) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(5, 2) CASE(6, 2) CASE(8, 2)
#undef CASE
  MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: no kernel for NT=%d MT=%d", NT, MT);
}

static int dispatch_igemm_q8(const ConvArgs& a, int NT, int gx, int gy, int threads, size_t lds, hipStream_t st) {
#define CASE(nt) \
  if (NT == nt) return launch_igemm<bf16, nt, 2, true>(a, gx, gy, threads, lds, st);
  CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(8)
#undef CASE
  MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d_fp8: no kernel for NT=%d", NT);
}

static int conv2d_fwd_impl(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                           const void* packed_w, const float* bias, int k, int stride, int act, const mgdt_view* r1,
                           const mgdt_view* r2, const mgdt_view* y, int dtype, int tapmode, mgdt_stream s,
                           const float* q8_oscale = nullptr, float q8_xq = 1.f);
extern "C" int mgdt_conv2d_fwd(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                               const void* packed_w, const float* bias, int k, int stride, int act, const mgdt_view* r1,
                               const mgdt_view* r2, const mgdt_view* y, int dtype, mgdt_stream s) {
  return conv2d_fwd_impl(x, x2, in_scale, in_shift, packed_w, bias, k, stride, act, r1, r2, y, dtype, 0, s);
}
// The phase convolutions of a stride-2 3x3 data gradient: panels packed by mgdt_conv_pack_dgrad(phase = 0..3) hold only the 1, 2 or 4 taps the
// phase uses (K = taps * channels instead of 9 * channels); `phase` selects the matching tap list.  k = 3, stride 1.
extern "C" int mgdt_conv2d_phase_fwd(const mgdt_view* x, const void* packed_w, const float* bias, int phase, const mgdt_view* r1, const mgdt_view* r2,
                                     const mgdt_view* y, int dtype, mgdt_stream s) {
  if (phase < 0 || phase > 3) MGDT_FAIL(MGDT_BAD_ARG, "conv2d_phase: phase %d", phase);
  return conv2d_fwd_impl(x, nullptr, nullptr, nullptr, packed_w, bias, 3, 1, MGDT_ACT_NONE, r1, r2, y, dtype, 2 + phase, s);
}
// fp8 (e4m3) variant - BASELINE configs[4]: bf16 views, panel from mgdt_conv_pack_fp8 (same x_qscale), fp8 MFMAs, fp32 accumulation
extern "C" int mgdt_conv2d_fp8_fwd(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                                   const void* packed_w, const float* bias, const float* oscale, float x_qscale, int k, int stride, int act,
                                   const mgdt_view* r1, const mgdt_view* r2, const mgdt_view* y, mgdt_stream s) {
  if (!oscale) MGDT_FAIL(MGDT_BAD_ARG, "conv2d_fp8: null oscale");
  if (!(x_qscale > 0.f) || !(x_qscale < 3.0e38f)) MGDT_FAIL(MGDT_BAD_ARG, "conv2d_fp8: x_qscale must be a positive finite number");
  return conv2d_fwd_impl(x, x2, in_scale, in_shift, packed_w, bias, k, stride, act, r1, r2, y, MGDT_BF16, 0, s, oscale, x_qscale);
}
static int conv2d_fwd_impl(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                           const void* packed_w, const float* bias, int k, int stride, int act, const mgdt_view* r1,
                           const mgdt_view* r2, const mgdt_view* y, int dtype, int tapmode, mgdt_stream s,
                           const float* q8_oscale, float q8_xq) {
  const bool q8 = q8_oscale != nullptr;
  const int WB = q8 ? 512 : 1024;
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
  const long sz = (long)dtype_size(dtype);
  auto extent = [&](const mgdt_view* v) { return ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz; };
  // every view is addressed through a 32-bit buffer descriptor: extents < 2 GiB, row / pixel strides < 8 MiB (24-bit multiplies)
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes) {
    if (!v || !v->p) return;
    if (extent(v) >= 0x7fffffffL || v->sh * sz >= (1L << 23) || v->sw * sz >= (1L << 23)) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * sz); *sh = (int)(v->sh * sz); *sw = (int)(v->sw * sz); *bytes = (uint32_t)extent(v);
  };
  const char* yp = nullptr;
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes);
  bind(x2, &a.x2, &a.x2sn, &a.x2sh, &a.x2sw, &a.x2_bytes);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes);
  bind(r1, &a.r1, &a.r1sn, &a.r1sh, &a.r1sw, &a.r1_bytes);
  bind(r2, &a.r2, &a.r2sn, &a.r2sh, &a.r2sw, &a.r2_bytes);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: a view spans >= 2 GiB or has a row stride >= 8 MiB");
  a.y = (char*)yp;
  a.in_scale = in_scale; a.in_shift = in_shift;
  a.wpk = (const char*)packed_w; a.bias = bias;
  a.oscale = q8_oscale; a.xq = q8_xq;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Ho = Ho; a.Wo = Wo; a.Cout = y->c;
  a.KS = k; a.stride = stride; a.pad = pad; a.act = act;
  a.ntaps = tapmode >= 2 ? phase_ntaps(tapmode) : k * k;
  a.taplist = phase_taplist(tapmode, k);
  conv_geometry(a.Cin, a.Cout, k, dtype, &a.CP, &a.nchunks, &a.NTtot, tapmode >= 2 ? a.ntaps : 0);
  long M = (long)a.N * Ho * Wo;
  if (M > 0x7fffffffL - (1 << 20)) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: problem too large");
  a.M = (int)M; a.HoWo = Ho * Wo;
  a.fd_howo = make_fastdiv((uint32_t)a.HoWo); a.fd_wo = make_fastdiv((uint32_t)Wo);
  // plain 3x3 stride-1 bf16 layers with 32-80 input channels on large maps: the LDS-staged kernel (conv3x3_lds.hip)
  if (tapmode == 0 && dtype == MGDT_BF16 && k == 3 && !(x2 && x2->p) && !in_scale && !in_shift && !(r1 && r1->p) && !(r2 && r2->p) &&
      mgdt_conv3x3_lds_launch(x, y, packed_w, bias, act, a.CP, a.nchunks, a.NTtot, (hipStream_t)s, q8_oscale, q8_xq, stride)) {
    MGDT_CHECK_LAUNCH("conv2d(3x3 lds)");
    return MGDT_OK;
  }

  // tile choice: NT = cout blocks per workgroup - as many as divide NTtot and keep the whole weight panel in LDS
  // (activations are then read NTtot/NT times; once when NT == NTtot), fewer when the grid would starve the 256 CUs
  int LDS_PANEL_KIB = 144;   // measured (round 2, tools/knob_sweep.sh, B = 32 bf16): 96 -> 1.377, 128 -> 1.326, 144 / 150 -> 1.312 ms per step (NT 2 -> 4 on the 128 -> 256 stride-2 conv)
  { const char* e = getenv("MGDT_CONV_PANEL_KIB"); if (e) LDS_PANEL_KIB = atoi(e); }   // experiment knob (not part of the ABI)
  int NT = 1;
  for (int c : {8, 6, 5, 4, 3, 2, 1})
    if (a.NTtot % c == 0 && (long)a.nchunks * c * WB <= LDS_PANEL_KIB * 1024L) { NT = c; break; }
  int MT = 2;   // MT=2 with depth-4 prefetch measured faster than MT=4 on every wide layer of the target nets
  if (!q8 && NT == 1 && a.NTtot == 1) { const char* e = getenv("MGDT_CONV_MT1"); MT = e && atoi(e) == 4 ? 4 : 2; }   // (MT = 8 spilled registers: removed)
  int waves = 8;
  auto wgs = [&](int nt, int wv) { return (long)cdiv(M, 16 * wv * MT) * (a.NTtot / nt); };
  // (4-wave workgroups for small maps were tried: 8 waves measured faster on the whole net - fewer, fuller workgroups stage the weight panel less often)
  {
    const char* e = getenv("MGDT_CONV_MINWG");            // experiment knob (not part of the ABI)
    const int minwg = e ? atoi(e) : 128;               // measured: 64 -> 1.800, 128 -> 1.791, 256 -> 1.811, 512 -> 1.876 ms per step
    while (wgs(NT, waves) < minwg && NT > 1 && NT % 2 == 0) NT /= 2;    // ... and split the couts over workgroups
  }
  {   // a grid a few workgroups larger than what is resident at once (~3 eight-wave workgroups per CU) runs a second, almost empty round at the first one's
      // full cost: 32 -> 512 at 40x40 (B = 32) is 800 workgroups on 768 slots.  Halving NT gives twice as many half-size workgroups: two full rounds of half the work.
    // Measured (r02f): that layer 30.6 -> 28.1 us, but the whole step 1.294 -> 1.322 ms (the next kernels start behind a longer tail of small workgroups): off by default.
    static const bool tail_split = getenv("MGDT_CONV_TAIL_SPLIT") != nullptr;     // experiment knob (not part of the ABI)
    const long w0 = wgs(NT, waves);
    if (tail_split && w0 > 768 && w0 <= 960 && NT % 2 == 0) NT /= 2;
  }
  {   // experiment knob (not part of the ABI)
    const char* e;
    if ((e = getenv("MGDT_CONV_WAVES"))) waves = std::min(8, std::max(1, atoi(e)));   // __launch_bounds__(512): at most 8 waves per workgroup
  }
  a.numTiles = cdiv(M, 16 * waves * MT);
  a.T8 = cdiv(a.numTiles, 8);
  a.tab_bytes = (a.nchunks + 3) / 4 * 4 * 4 * 16;   // uint4 per piece, padded to a multiple of 4 chunks (>= nchp in the kernel)
  size_t panel = (size_t)a.nchunks * NT * WB;
  if (panel <= (size_t)LDS_PANEL_KIB * 1024) { a.seg_chunks = a.nchunks; a.nseg = 1; }
  else { a.seg_chunks = 64; a.nseg = cdiv(a.nchunks, a.seg_chunks); }   // NT == 1 here: 64 KiB segments
  size_t lds = a.tab_bytes + (size_t)(q8 ? 2 : 1) * NT * 16 * sizeof(float) + (size_t)a.seg_chunks * NT * WB;
  int gcap = waves == 8 ? 256 : 1024;   // measured with the 144 KiB panels: 256 -> 1.293, 384 -> 1.306, 512 -> 1.312 ms per step (one persistent workgroup per CU and cout group)
  { const char* e = getenv("MGDT_CONV_GCAP"); if (e) gcap = atoi(e); }
  if (lds > 160 * 1024) MGDT_FAIL(MGDT_BAD_SHAPE, "conv2d: %zu bytes of LDS for NT=%d, %d chunks", lds, NT, a.seg_chunks);   // cannot happen with panels <= 150 KiB (table <= 9.4 KiB)
  int gx = std::min(8 * a.T8, gcap), gy = a.NTtot / NT;   // persistent: ~2 (8-wave) / 4 (4-wave) workgroups per CU
  hipStream_t st = (hipStream_t)s;
  if (q8) return dispatch_igemm_q8(a, NT, gx, gy, waves * 64, lds, st);
  if (dtype == MGDT_F32) return dispatch_igemm<float>(a, NT, MT, gx, gy, waves * 64, lds, st);
  return dispatch_igemm<bf16>(a, NT, MT, gx, gy, waves * 64, lds, st);
}
