// Convolution weight gradient on the matrix cores (fp32-in / fp32-accumulate MFMA, exact fp32 products):
//   dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n,oy,ox,co] * (x [+ x2])[n, oy*s - pad + ky, ox*s - pad + kx, ci]
// Reference: the autograd of nn.Conv2d inside every Conv (nn/modules/conv.py:25-42) under yolo/engine/trainer.py:343.
//
// GEMM view: D[co][(tap, ci)] = sum_pixels dyT[co][pixel] * X[pixel][(tap, ci)], K = the B*Ho*Wo output pixels.  v_mfma_f32_16x16x4_f32
// takes ONE fp32 per lane for each operand - lane (r, g) supplies A[row r][k = g] and B[k = g][col r] - so with k = 4 consecutive pixels a
// lane's A value is dy[pixel g][co0 + r] and its B value x[pixel g shifted by the tap][ci0 + r]: both are plain NHWC loads in which the 16
// lanes of a pixel read 16 consecutive channels (64 contiguous bytes).  No transposition, no LDS staging; taps outside the image and
// channels past the tensor are out-of-range buffer offsets that load zeros.  A wave owns NT cout blocks x MT (tap, ci) blocks
// (NT*MT <= 16 accumulator tiles) and walks its share of the pixels four at a time, U quads of loads in flight; the four waves of a
// workgroup take interleaved quads and are summed through LDS; workgroups split the pixel range (fixed order, summed by
// wgrad_final_kernel: deterministic).  bf16 activations are widened to fp32 on load (the fp32 MFMA runs at the fp32 vector rate, 157 TF/s
// peak - an order of magnitude above the VALU outer products it replaces, which re-read both tensors once per tap).
#include "conv_igemm_kernel.h"

struct WgArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  const char* x2; int x2sn, x2sh, x2sw; uint32_t x2_bytes;
  const char* dy; int dsn, dsh, dsw; uint32_t dy_bytes;
  float* partial;
  int N, H, W, Cin, Ho, Wo, Cout, KS, stride, pad, nsplit, M, HoWo, CIB, ncolb, ncob;
  FastDiv fd_howo, fd_wo;
};

template <typename T> __device__ __forceinline__ float wg_load(__amdgpu_buffer_rsrc_t rs, uint32_t off);
template <> __device__ __forceinline__ float wg_load<float>(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
template <> __device__ __forceinline__ float wg_load<bf16>(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0);
  return __builtin_bit_cast(float, (unsigned)u << 16);
}

constexpr int WG_U = 4;          // pixel quads per iteration

template <typename T, int NT, int MT, bool X2>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const WgArgs a) {
  constexpr int SZ = (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) f32x4 red[4][NT * MT][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int cob0 = blockIdx.x * NT, colb0 = blockIdx.y * MT, split = blockIdx.z;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X2 ? a.x2 : a.x), 0, X2 ? a.x2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

  // per lane: channel byte offsets (MGDT_OOB when the channel does not exist) of its NT cout blocks and MT (tap, ci) blocks, tap of each block
  int coff[NT], xoff[MT], tapdy[MT], tapbit[MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (cob0 + nt) * 16 + r;
    coff[nt] = co < a.Cout ? co * SZ : MGDT_OOB;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int cb = colb0 + mt;
    const int tap = cb / a.CIB, cib = cb - tap * a.CIB;
    const int ci = cib * 16 + r;
    const bool ok = cb < a.ncolb && ci < a.Cin;
    const int ky = tap / a.KS, kx = tap - ky * a.KS;
    xoff[mt] = ok ? ky * a.xsh + kx * a.xsw + ci * SZ : MGDT_OOB;
    tapdy[mt] = X2 ? (ok ? ky * a.x2sh + kx * a.x2sw + ci * SZ : MGDT_OOB) : 0;
    tapbit[mt] = tap < 9 ? tap : 31;
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int p0 = (int)((long)split * a.M / a.nsplit), p1 = (int)((long)(split + 1) * a.M / a.nsplit);
  const int ssh = a.stride >> 1;
  // quads of this split: q = 0 .. ceil((p1 - p0) / 4); wave w takes quads w, w + 4, ...; U of them per iteration
  const int nq = (p1 - p0 + 3) >> 2;
  for (int q0 = wave * WG_U; q0 < nq; q0 += 4 * WG_U) {
    float A[WG_U][NT], B[WG_U][MT];
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
      const int p = p0 + (q0 + u) * 4 + g;
      const bool pv = (q0 + u) < nq && p < p1;
      const int pp = pv ? p : p0;
      const int n = (int)fdiv((uint32_t)pp, a.fd_howo), rem = pp - n * a.HoWo;
      const int oy = (int)fdiv((uint32_t)rem, a.fd_wo), ox = rem - oy * a.Wo;
      const int iy0 = (oy << ssh) - a.pad, ix0 = (ox << ssh) - a.pad;
      const int dyo = pv ? n * a.dsn + oy * a.dsh + ox * a.dsw : MGDT_OOB;
      const int xo = n * a.xsn + iy0 * a.xsh + ix0 * a.xsw;
      const int x2o = X2 ? n * a.x2sn + iy0 * a.x2sh + ix0 * a.x2sw : 0;
      uint32_t mask = 1u;
      if (a.KS == 3) {
        const uint32_t rm = (uint32_t)((unsigned)iy0 < (unsigned)a.H) | ((uint32_t)((unsigned)(iy0 + 1) < (unsigned)a.H) << 1) |
                            ((uint32_t)((unsigned)(iy0 + 2) < (unsigned)a.H) << 2);
        const uint32_t cm = (uint32_t)((unsigned)ix0 < (unsigned)a.W) | ((uint32_t)((unsigned)(ix0 + 1) < (unsigned)a.W) << 1) |
                            ((uint32_t)((unsigned)(ix0 + 2) < (unsigned)a.W) << 2);
        mask = ((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 3 : 0u) | ((rm & 4u) ? cm << 6 : 0u);
      } else {
        mask = ((unsigned)iy0 < (unsigned)a.H && (unsigned)ix0 < (unsigned)a.W) ? 1u : 0u;
      }
      if (!pv) mask = 0u;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) A[u][nt] = wg_load<T>(drs, (dyo | coff[nt]) < 0 ? (uint32_t)MGDT_OOB : (uint32_t)(dyo + coff[nt]));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool ok = ((mask >> tapbit[mt]) & 1u) && xoff[mt] >= 0;
        float v = wg_load<T>(xrs, ok ? (uint32_t)(xo + xoff[mt]) : (uint32_t)MGDT_OOB);
        if (X2) v = (float)(T)(v + wg_load<T>(x2rs, ok ? (uint32_t)(x2o + tapdy[mt]) : (uint32_t)MGDT_OOB));   // the forward's rounded pre-add
        B[u][mt] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < WG_U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][nt], B[u][mt], acc[nt][mt], 0, 0, 0);
  }

  // the four waves hold partial sums over disjoint pixels: add them in wave order through LDS, wave w finishes tiles w, w + 4, ...
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave][nt * MT + mt][lane] = acc[nt][mt];
  __syncthreads();
  const int taps = a.KS * a.KS;
  for (int b = wave; b < NT * MT; b += 4) {
    const f32x4 s = ((red[0][b][lane] + red[1][b][lane]) + red[2][b][lane]) + red[3][b][lane];
    const int nt = b / MT, mt = b - nt * MT;
    const int cb = colb0 + mt;
    if (cb >= a.ncolb) continue;
    const int tap = cb / a.CIB, cib = cb - tap * a.CIB;
    const int ci = cib * 16 + r;
    if (ci >= a.Cin) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = (cob0 + nt) * 16 + 4 * g + j;            // D layout: lane (r, g) holds rows 4g .. 4g+3 (cout) of column r (ci)
      if (co < a.Cout) a.partial[(((long)split * a.Cout + co) * a.Cin + ci) * taps + tap] = s[j];
    }
  }
}

template <typename T, int NT, int MT>
static void wg_launch(const WgArgs& a, dim3 grid, hipStream_t st) {
  if (a.x2) conv_wgrad_mfma_kernel<T, NT, MT, true><<<grid, 256, 0, st>>>(a);
  else conv_wgrad_mfma_kernel<T, NT, MT, false><<<grid, 256, 0, st>>>(a);
}

// Called by mgdt_conv_wgrad (train.hip) for NHWC inputs; partial: fp32 [nsplit][cout][cin][k*k].  Returns false when a view cannot be
// addressed through 32-bit descriptors (the caller keeps the VALU kernel).
bool mgdt_wgrad_mfma_launch(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* partial, int nsplit, int dtype, hipStream_t st) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || x->sc != 1 || dy->sc != 1) return false;
  WgArgs a;
  memset(&a, 0, sizeof(a));
  const long sz = (long)dtype_size(dtype);
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz;
    if (ext >= 0x7fffffffL || v->sh * sz >= (1L << 28)) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * sz); *sh = (int)(v->sh * sz); *sw = (int)(v->sw * sz); *bytes = (uint32_t)ext;
  };
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes);
  if (x2 && x2->p) {
    if (x2->sc != 1) return false;
    bind(x2, &a.x2, &a.x2sn, &a.x2sh, &a.x2sw, &a.x2_bytes);
  }
  bind(dy, &a.dy, &a.dsn, &a.dsh, &a.dsw, &a.dy_bytes);
  if (!fits) return false;
  a.partial = partial;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Ho = dy->h; a.Wo = dy->w; a.Cout = dy->c; a.KS = k; a.stride = stride; a.pad = k / 2;
  a.nsplit = nsplit; a.M = dy->n * dy->h * dy->w; a.HoWo = dy->h * dy->w;
  a.CIB = cdiv(a.Cin, 16); a.ncolb = k * k * a.CIB; a.ncob = cdiv(a.Cout, 16);
  a.fd_howo = make_fastdiv((uint32_t)a.HoWo); a.fd_wo = make_fastdiv((uint32_t)a.Wo);
  // tile: NT cout blocks x MT (tap, ci) blocks per wave, NT * MT <= 16 accumulator tiles
  int NT = a.ncob >= 4 ? 4 : (a.ncob >= 2 ? 2 : 1);
  int MT = NT == 4 ? 4 : (NT == 2 ? 8 : 9);
  if (a.ncolb < MT) MT = a.ncolb >= 8 ? 8 : (a.ncolb >= 4 ? 4 : (a.ncolb >= 2 ? 2 : 1));
  dim3 grid(cdiv(a.ncob, NT), cdiv(a.ncolb, MT), nsplit);
#define WG_CASE(nt, mt) if (NT == nt && MT == mt) { if (dtype == MGDT_F32) wg_launch<float, nt, mt>(a, grid, st); else wg_launch<bf16, nt, mt>(a, grid, st); return true; }
  WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 8) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1) WG_CASE(1, 9) WG_CASE(1, 8) WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
#undef WG_CASE
  return false;
}
