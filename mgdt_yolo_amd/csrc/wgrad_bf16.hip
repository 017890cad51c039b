// Convolution weight gradient for the bf16 training path on the bf16 matrix cores (fp32 accumulate):
//   dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n,oy,ox,co] * (x [+ x2])[n, oy*s - pad + ky, ox*s - pad + kx, ci]
// Reference: the autograd of nn.Conv2d inside every Conv (nn/modules/conv.py:25-42) under autocast (yolo/engine/trainer.py:329-343).
//
// GEMM view: D[co][(tap, ci)] = sum over output pixels of dyT[co][pixel] * X[pixel shifted by the tap][ci]; the reduction index is the
// PIXEL, which is the slow index of both NHWC tensors, so both MFMA operands need a transpose.  A workgroup stages an 8x8 tile of dy and the
// matching input tile of x (with its 3x3 halo, zero outside the image; x + x2 rounded to bf16 as the forward rounded it) into LDS as plain
// [pixel][channel] rows and reads the fragments with ds_read_b64_tr_b16 (the hardware transpose read of gfx950): per 16-lane group it takes
// 4 pixel rows x 16 channels and hands lane i channel i of the 4 pixels = 4 consecutive k of v_mfma_f32_16x16x32_bf16.  Because every lane
// supplies its own row address, a tap is a constant added to the row address and stride 2 a doubled row step: one LDS image serves all 9
// taps.  k <-> pixel assignment inside a 32-pixel step is chosen so that the 32 lanes of a half-wave read 8 consecutive pixels (row pitch an
// odd multiple of 32 B -> conflict-free for stride 1).
// Work split: wave (wa, wb) of the 4 owns AT cout blocks x NCW (tap, ci-block) columns; workgroups split (cout group, ci group) x pixel
// tiles; per-split partial sums are added in fixed order by wgrad_final_kernel (train.hip): deterministic.
#include "common.h"

#define MGDT_OOB ((int)0x80000000)   // beyond every view extent: buffer loads return 0
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS3(T, p) ((T __attribute__((address_space(3)))*)(p))

struct WbArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  const char* x2; int x2sn, x2sh, x2sw; uint32_t x2_bytes;
  const char* dy; int dsn, dsh, dsw; uint32_t dy_bytes;
  float* partial;
  int H, W, Cin, Ho, Wo, Cout, KS, stride, pad, nsplit;
  int tiles_x, tiles_per_img, ntiles;
  int WA, WB, WK, BT, ncig, XTW, XTH, XP, DP, x_lds_bytes, ci8, co8, NTS;      // ci8 / co8: 8-channel items per staged row
  FastDiv fd_ci8, fd_co8, fd_xtw, fd_tpi, fd_tx, fd_xitems, fd_ditems;
};

__device__ __forceinline__ u32x4 bf16x8_add(u32x4 a, u32x4 b) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a0 = __builtin_bit_cast(float, a[k] << 16), a1 = __builtin_bit_cast(float, a[k] & 0xffff0000u);
    const float b0 = __builtin_bit_cast(float, b[k] << 16), b1 = __builtin_bit_cast(float, b[k] & 0xffff0000u);
    const bf16 r0 = (bf16)(a0 + b0), r1 = (bf16)(a1 + b1);
    o[k] = (uint32_t)__builtin_bit_cast(unsigned short, r0) | ((uint32_t)__builtin_bit_cast(unsigned short, r1) << 16);
  }
  return o;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, p1));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int AT, int NCW>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WbArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wb_lds[];
  char* xs = wb_lds;
  char* ds = wb_lds + a.NTS * a.x_lds_bytes;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int WB = a.WB, wa = wave % a.WA, wb = (wave / a.WA) % WB, wk = wave / (a.WA * WB);     // waves along cout, columns and pixel tiles
  const int cog = blockIdx.x / a.ncig, cig = blockIdx.x - cog * a.ncig, split = blockIdx.y;
  const int co0 = cog * (AT * a.WA * 16), ci0 = cig * (a.BT * 16);
  const int taps = a.KS * a.KS, ncol = taps * a.BT, s = a.stride;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? a.x2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

  // this wave's columns: col = wb + WB * j -> (tap, ci block); LDS byte offset of the column's fragment relative to a pixel's row address
  int boff[NCW];
#pragma unroll
  for (int j = 0; j < NCW; ++j) {
    const int col = wb + WB * j;
    const int tap = col / a.BT, cb = col - tap * a.BT;
    const int ky = tap / a.KS, kx = tap - ky * a.KS;
    boff[j] = col < ncol ? (ky * a.XTW + kx) * a.XP + cb * 32 : 0;
  }
  // lane's two row addresses of a 32-pixel step (see the header: half-waves read 8 consecutive pixels of a tile row)
  int abase[2], bbase[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int ty = (g >> 1) + 2 * rd, tx = (g & 1) * 4 + (i >> 2);
    abase[rd] = (ty * 8 + tx) * a.DP + (i & 3) * 8 + wa * AT * 32;
    bbase[rd] = ((ty * s) * a.XTW + tx * s) * a.XP + (i & 3) * 8;
  }
  const int astep = 32 * a.DP, bstep = 4 * s * a.XTW * a.XP;       // second 32-pixel step of the tile: 4 tile rows further

  f32x4 acc[AT][NCW];
#pragma unroll
  for (int at = 0; at < AT; ++at)
#pragma unroll
    for (int j = 0; j < NCW; ++j) acc[at][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t0 = (int)((long)split * a.ntiles / a.nsplit), t1 = (int)((long)(split + 1) * a.ntiles / a.nsplit);
  const int xitems = a.XTH * a.XTW * a.ci8, ditems = 64 * a.co8, d_lds_bytes = 64 * a.DP;
  constexpr int U = 4;                                   // 16-byte loads in flight per thread while staging
  for (int t = t0; t < t1; t += a.NTS) {
    const int cnt = min(a.NTS, t1 - t);                  // tiles staged together (small channel counts: many tiles per barrier pair)
    __syncthreads();                                     // the previous stage's fragments have been read
    for (int base = tid; base < cnt * xitems; base += 256 * U) {
      u32x4 v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it0 = base + u * 256;
        const bool in = it0 < cnt * xitems;
        const int sidx = (int)fdiv((uint32_t)it0, a.fd_xitems), it = it0 - sidx * xitems;
        const int tt = t + sidx;
        const int n = (int)fdiv((uint32_t)tt, a.fd_tpi), rt = tt - n * a.tiles_per_img;
        const int tyi = (int)fdiv((uint32_t)rt, a.fd_tx), txi = rt - tyi * a.tiles_x;
        const int pix = (int)fdiv((uint32_t)it, a.fd_ci8), c8 = it - pix * a.ci8;
        const int py = (int)fdiv((uint32_t)pix, a.fd_xtw), px = pix - py * a.XTW;
        const int iy = tyi * 8 * s - a.pad + py, ix = txi * 8 * s - a.pad + px, ci = ci0 + c8 * 8;
        const bool ok = in && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && ci < a.Cin;
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (uint32_t)(n * a.xsn + iy * a.xsh + ix * a.xsw + ci * 2) : (uint32_t)MGDT_OOB, 0, 0));
        if (a.x2) {
          const u32x4 v2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x2rs, ok ? (uint32_t)(n * a.x2sn + iy * a.x2sh + ix * a.x2sw + ci * 2) : (uint32_t)MGDT_OOB, 0, 0));
          v[u] = bf16x8_add(v[u], v2);
        }
        dst[u] = in ? sidx * a.x_lds_bytes + pix * a.XP + c8 * 16 : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) *(u32x4*)(xs + dst[u]) = v[u];
    }
    for (int base = tid; base < cnt * ditems; base += 256 * U) {
      u32x4 v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it0 = base + u * 256;
        const bool in = it0 < cnt * ditems;
        const int sidx = (int)fdiv((uint32_t)it0, a.fd_ditems), it = it0 - sidx * ditems;
        const int tt = t + sidx;
        const int n = (int)fdiv((uint32_t)tt, a.fd_tpi), rt = tt - n * a.tiles_per_img;
        const int tyi = (int)fdiv((uint32_t)rt, a.fd_tx), txi = rt - tyi * a.tiles_x;
        const int pix = (int)fdiv((uint32_t)it, a.fd_co8), c8 = it - pix * a.co8;
        const int oy = tyi * 8 + (pix >> 3), ox = txi * 8 + (pix & 7), co = co0 + c8 * 8;
        const bool ok = in && oy < a.Ho && ox < a.Wo && co < a.Cout;
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, ok ? (uint32_t)(n * a.dsn + oy * a.dsh + ox * a.dsw + co * 2) : (uint32_t)MGDT_OOB, 0, 0));
        dst[u] = in ? sidx * d_lds_bytes + pix * a.DP + c8 * 16 : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) *(u32x4*)(ds + dst[u]) = v[u];
    }
    __syncthreads();
    for (int sidx = wk; sidx < cnt; sidx += a.WK) {
      const char* xt = xs + sidx * a.x_lds_bytes;
      const char* dt = ds + sidx * d_lds_bytes;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 A[AT];
#pragma unroll
        for (int at = 0; at < AT; ++at) A[at] = tr_frag(dt + abase[0] + c * astep + at * 32, dt + abase[1] + c * astep + at * 32);
#pragma unroll
        for (int j = 0; j < NCW; ++j) {
          const bf16x8 B = tr_frag(xt + bbase[0] + c * bstep + boff[j], xt + bbase[1] + c * bstep + boff[j]);
#pragma unroll
          for (int at = 0; at < AT; ++at) acc[at][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[at], B, acc[at][j], 0, 0, 0);
        }
      }
    }
  }

  // waves that split the pixel tiles (WK > 1) hold partial sums of the same outputs: add them in wave order through LDS
  if (a.WK > 1) {
    __syncthreads();
    f32x4* red = (f32x4*)wb_lds;                           // [wave][AT * NCW][64]; wave = wk * (WA * WB) + its (wb, wa) position
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
      for (int j = 0; j < NCW; ++j) red[(wave * AT * NCW + at * NCW + j) * 64 + lane] = acc[at][j];
    __syncthreads();
    if (wk != 0) return;
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
      for (int j = 0; j < NCW; ++j) {
        f32x4 t = red[(wave * AT * NCW + at * NCW + j) * 64 + lane];
        for (int q = 1; q < a.WK; ++q) t += red[((wave + q * a.WA * WB) * AT * NCW + at * NCW + j) * 64 + lane];
        acc[at][j] = t;
      }
  }
  // D layout: lane (i, g) holds rows 4g .. 4g+3 (cout) of column i (ci)
#pragma unroll
  for (int j = 0; j < NCW; ++j) {
    const int col = wb + WB * j;
    if (col >= ncol) continue;
    const int tap = col / a.BT, cb = col - tap * a.BT;
    const int ci = ci0 + cb * 16 + i;
    if (ci >= a.Cin) continue;
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wa * AT + at) * 16 + 4 * g + r;
        if (co < a.Cout) a.partial[(((long)split * a.Cout + co) * a.Cin + ci) * taps + tap] = acc[at][j][r];
      }
  }
}

struct WbCfg { int AT, NCW, WA, WB, WK, BT; double cost; };

// Pick the wave tile (AT cout blocks x NCW columns) and the arrangement of the 4 waves (WA along cout, WB along columns, WK along the
// pixel tiles - for layers with fewer channels than one wave tile), hence the workgroup's channel tile.
static WbCfg wb_pick(int ncob, int ncib, int taps, int stride, int ks) {
  static const int ATS[] = {1, 2, 4}, NCWS[] = {1, 2, 3, 5, 9, 12};
  WbCfg best{0, 0, 0, 0, 0, 0, 1e30};
  for (int AT : ATS)
    for (int NCW : NCWS)
      for (int WK = 1; WK <= 4; WK *= 2)
        for (int WA = 1; WA * WK <= 4; WA *= 2) {
          const int WB = 4 / (WA * WK);
          if (AT * NCW > 24 || (WK > 1 && AT * NCW > 6)) continue;
          const int BT = std::min({NCW * WB / taps, 24, std::max(ncib, 1)});
          if (BT < 1) continue;
          const int cot = AT * WA;
          const int xtl = 7 * stride + ks;
          const long ldsb = (long)xtl * xtl * (BT * 32 + ((BT & 1) ? 0 : 32)) + 64L * (cot * 32 + ((cot & 1) ? 0 : 32));
          if (ldsb > 64 * 1024) continue;
          const long groups = (long)((ncob + cot - 1) / cot) * ((ncib + BT - 1) / BT);
          const double mfma = (double)AT * NCW * 16.0 * 2 / WK;               // clocks per tile (2 steps), waves in parallel on the 4 SIMDs
          const double ldsr = 4.0 * (AT + NCW) * 2 * 4.0 * 2 / WK;            // tr reads share the CU's LDS port
          const double stage = ((double)xtl * xtl * BT * 2 + 64.0 * cot * 2) / 256.0 * 24.0 + 150.0;   // 16-byte items per thread + barriers
          const double cost = groups * (std::max(mfma, ldsr) + stage);
          if (cost < best.cost) best = WbCfg{AT, NCW, WA, WB, WK, BT, cost};
        }
  return best;
}

template <int AT, int NCW>
static void wb_launch(const WbArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  conv_wgrad_bf16_kernel<AT, NCW><<<grid, 256, lds, st>>>(a);
}

// (cout tile, cin tile) groups of the configuration wb_pick chooses at stride 1: train.hip sizes the pixel splits so that groups x splits fills the chip
int mgdt_wgrad_bf16_groups(int cin, int cout, int k) {
  if ((k != 1 && k != 3) || cin % 4 || cout % 8) return 0;
  const int ncob = cdiv(cout, 16), ncib = cdiv(cin, 16);
  const WbCfg cfg = wb_pick(ncob, ncib, k * k, 1, k);
  return cfg.AT ? cdiv(ncob, cfg.AT * cfg.WA) * cdiv(ncib, cfg.BT) : 0;
}

// Called by mgdt_conv_wgrad (train.hip) for bf16 NHWC inputs; partial: fp32 [nsplit][cout][cin][k*k].  false -> the caller keeps the fp32 MFMA kernel.
bool mgdt_wgrad_bf16_launch(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* partial, int nsplit, hipStream_t st) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || (k == 1 && stride != 1) || x->sc != 1 || dy->sc != 1) return false;
  // input channels in multiples of 4: an 8-channel item of a 4-channel-granular tensor also carries the next pixel's first channels; they only
  // reach accumulators of input channels >= Cin, which the epilogue never writes (each (cout, cin) sum is independent)
  if (x->c % 4 || dy->c % 8) return false;
  WbArgs a;
  memset(&a, 0, sizeof(a));
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes, int al) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * 2;
    if (ext >= 0x7fffffffL || (uintptr_t)v->p % (2 * al) || v->sn % al || v->sh % al || v->sw % al) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * 2); *sh = (int)(v->sh * 2); *sw = (int)(v->sw * 2); *bytes = (uint32_t)ext;
  };
  const int xal = x->c % 8 ? 4 : 8;
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes, xal);
  if (x2 && x2->p) {
    if (x2->sc != 1) return false;
    bind(x2, &a.x2, &a.x2sn, &a.x2sh, &a.x2sw, &a.x2_bytes, xal);
  }
  bind(dy, &a.dy, &a.dsn, &a.dsh, &a.dsw, &a.dy_bytes, 8);
  if (!fits) return false;
  a.partial = partial;
  a.H = x->h; a.W = x->w; a.Cin = x->c; a.Ho = dy->h; a.Wo = dy->w; a.Cout = dy->c; a.KS = k; a.stride = stride; a.pad = k / 2; a.nsplit = nsplit;
  a.tiles_x = cdiv(a.Wo, 8);
  a.tiles_per_img = a.tiles_x * cdiv(a.Ho, 8);
  const long ntiles = (long)dy->n * a.tiles_per_img;
  if (ntiles >= 0x7fffffffL) return false;
  a.ntiles = (int)ntiles;
  const int ncob = cdiv(a.Cout, 16), ncib = cdiv(a.Cin, 16);
  const WbCfg cfg = wb_pick(ncob, ncib, k * k, stride, k);
  if (!cfg.AT) return false;
  a.WA = cfg.WA; a.WB = cfg.WB; a.WK = cfg.WK; a.BT = cfg.BT;
  a.ncig = cdiv(ncib, cfg.BT);
  const int ncog = cdiv(ncob, cfg.AT * cfg.WA);
  a.XTW = a.XTH = 7 * stride + k;
  const int cit = cfg.BT * 16, cot = cfg.AT * cfg.WA * 16;
  a.XP = cit * 2 + ((cfg.BT & 1) ? 0 : 32);                 // row pitch: an odd multiple of 32 bytes
  a.DP = cot * 2 + (((cot / 16) & 1) ? 0 : 32);
  a.x_lds_bytes = a.XTH * a.XTW * a.XP;
  a.ci8 = cit / 8; a.co8 = cot / 8;
  a.fd_ci8 = make_fastdiv((uint32_t)a.ci8); a.fd_co8 = make_fastdiv((uint32_t)a.co8); a.fd_xtw = make_fastdiv((uint32_t)a.XTW);
  a.fd_tpi = make_fastdiv((uint32_t)a.tiles_per_img); a.fd_tx = make_fastdiv((uint32_t)a.tiles_x);
  const size_t per_tile = (size_t)a.x_lds_bytes + (size_t)64 * a.DP;
  if (per_tile > 64 * 1024) return false;
  // small channel tiles: stage several pixel tiles per barrier pair (up to ~32 KB, so that a few workgroups still share a CU)
  a.NTS = (int)std::max<long>(1, std::min<long>({8L, (long)(32 * 1024 / per_tile), (ntiles + nsplit - 1) / nsplit}));
  a.fd_xitems = make_fastdiv((uint32_t)(a.XTH * a.XTW * a.ci8)); a.fd_ditems = make_fastdiv((uint32_t)(64 * a.co8));
  const size_t lds = std::max(per_tile * a.NTS, (size_t)(cfg.WK > 1 ? 4 : 0) * cfg.AT * cfg.NCW * 1024);
  dim3 grid(ncog * a.ncig, nsplit);
  static const bool dbg = getenv("MGDT_WGRAD_DBG") != nullptr;
  if (dbg) fprintf(stderr, "wgrad_bf16 cin %d cout %d k %d s %d %dx%d: AT %d NCW %d WA %d WB %d WK %d BT %d NTS %d grid %d x %d lds %zu\n", a.Cin, a.Cout, k, stride, a.Ho, a.Wo, cfg.AT, cfg.NCW, cfg.WA, cfg.WB, cfg.WK, cfg.BT, a.NTS, ncog * a.ncig, nsplit, lds);
#define WB_CASE(at, ncw) if (cfg.AT == at && cfg.NCW == ncw) { wb_launch<at, ncw>(a, grid, lds, st); return true; }
  WB_CASE(1, 1) WB_CASE(1, 2) WB_CASE(1, 3) WB_CASE(1, 5) WB_CASE(1, 9) WB_CASE(1, 12)
  WB_CASE(2, 1) WB_CASE(2, 2) WB_CASE(2, 3) WB_CASE(2, 5) WB_CASE(2, 9) WB_CASE(2, 12)
  WB_CASE(4, 1) WB_CASE(4, 2) WB_CASE(4, 3) WB_CASE(4, 5)
#undef WB_CASE
  return false;
}
