// One launch per CSP block (MSPA_C2f / C2f) of the bf16 inference path: the whole block for one spatial tile of one image stays on the
// CU.  Reference: nn/modules/block.py:209-287 (MSPA_C2f), :187-207 (C2f), :514-526 (Bottleneck).
//
//   front   MSPA: sp0 = cv0(x0), sp1 = cv1(sp0 + x1), sp2 = cv2(sp1 + x2)   (register-chained MFMAs, as mgdt_pw_chain3_fwd)
//           C2f : [y0 | y1] = cv1(x)                                          (1x1 conv, activations global -> VGPR)
//   middle  n bottlenecks, each two 3x3 convs wd -> wd (+ shortcut), on LDS-resident maps of the tile + halo
//   back    1x1 conv over the concat [front outputs | bottleneck outputs] -> Cout, written to HBM; MSPA: per-tile channel sums for
//           the SPR pooling attention (the block output is never re-read for pooling)
//
// Why: at 20x20 .. 80x80 the unfused block is 7-8 dependent launches of 10-40 us with < 5 us of work each (grids far below 256 CUs,
// one fill/drain each).  Here a workgroup (8 waves) owns a TH x TW tile of one image; the bottleneck chain needs a halo of 2n pixels,
// which is recomputed per tile (the recomputed part is cheap: K <= 576, and it is MFMA work that replaces HBM round trips).
//
// Layout on the CU: two NHWC maps P (bottleneck input / output) and T (its hidden map) over the (TH+2h) x (TW+2h) region, pixel stride
// wd*2 + 16 bytes (an odd number of 16-byte slots: 16 consecutive pixels hit 16 different LDS slots), and the concat buffer of the
// tile's own pixels.  Every 3x3 conv walks its output rectangle as one linear pixel range (the wrap-around columns are computed and
// never consumed), so a tap is a constant byte offset: B operand = one ds_read_b128 at (pixel + tap) * stride + piece * 16.
// GEMM orientation as conv_igemm: weights = A operand (rows = 16 output channels), pixels = B operand, so a lane ends with 4 consecutive
// channels of one pixel = one 8-byte LDS / HBM store.  Each wave keeps the weight fragments of ONE cout block of the running conv in
// registers (weights-stationary: 12..72 VGPRs) and streams pixel groups through them; the packed panels are the ordinary
// mgdt_conv_pack layout, read straight from L2.  Pixels outside the image are written as zeros (the next conv's zero padding).
#include "conv_igemm_kernel.h"

struct CspArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const char* front; const float* front_bias;
  const char* mid[4]; const float* mid_bias[4];
  const char* back; const float* back_bias;
  float* pool;
  int N, H, W, Cin, Cout, wd, nbtl, shortcut, act;
  int TH, TW, halo, RH, RW, tiles_x, tiles_y, total_tiles, per_xcd;
  int catC, nchb, nbo;                 // concat channels, K chunks / cout blocks of the back conv
  int front_nch, front_nb;             // C2f front: K chunks (Cin / 32) and cout blocks (2c / 16)
  int RPA, TPA, PS, CS;                // allocated region / tile pixels, pixel strides (bytes) of P/T and of the concat buffer
  int chain_words;                     // MSPA: 16-byte words of the chain blob's weight part
};

constexpr int CSP_THREADS = 512;
constexpr int CSP_NW = CSP_THREADS / 64;
constexpr int CSP_NCHB_MAX = 8;        // back conv: K <= 256 concat channels
constexpr int CSP_FRONT_NCH_MAX = 8;   // C2f front: Cin <= 256

__device__ __forceinline__ float csp_act(float v, int act) { return act == MGDT_ACT_SILU ? v * fast_sigmoid(v) : act_apply(v, act); }

typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int csp_raw2;

__device__ __forceinline__ void lds_store4(char* p, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
  *(bf16x4*)p = o;
}
__device__ __forceinline__ f32x4 lds_load4(const char* p) {
  const bf16x4 o = *(const bf16x4*)p;
  return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}

// MODE 0 = MSPA_C2f, 1 = C2f.  WD = bottleneck width (8, 16, 32, 64).
template <int WD, int MODE>
__global__ __launch_bounds__(CSP_THREADS) void csp_block_kernel(const CspArgs a) {
  constexpr int CP = WD / 8;                         // 16-byte pieces per tap
  constexpr int NB = WD >= 16 ? WD / 16 : 1;         // cout blocks of a wd -> wd conv
  constexpr int NCH = (9 * CP + 3) / 4;              // K chunks of a 3x3 conv
  constexpr int NBK = NB, KC = (NBK + 1) / 2;        // pw chain geometry (mlp_chain.hip)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* goff = (int*)smem;                                        // [RPA] byte offset of the region pixel in the x view, MGDT_OOB outside the image
  int* yoff = goff + a.RPA;                                      // [TPA] byte offset of the tile pixel in the y view
  short* ctab = (short*)(yoff + a.TPA);                          // [RPA] index of the region pixel in the tile, -1 outside
  char* Pb = (char*)(ctab + a.RPA);                              // RPA, TPA are multiples of 16: every array below starts 16-byte aligned
  char* Tb = Pb + (size_t)a.RPA * a.PS;
  char* catb = Tb + (size_t)a.RPA * a.PS;
  char* wl = catb + (size_t)a.TPA * a.CS;                        // MSPA: staged chain weights + bias

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // XCD k (workgroup id % 8) takes the contiguous tile range [k*per_xcd, (k+1)*per_xcd): neighbouring tiles share their halo in that XCD's L2
  const int v = blockIdx.x;
  const int tlin = (v & 7) * a.per_xcd + (v >> 3);
  if ((v >> 3) >= a.per_xcd || tlin >= a.total_tiles) return;    // uniform per workgroup: no barrier is skipped by a part of it
  const int tpi = a.tiles_x * a.tiles_y;
  const int n = tlin / tpi, trem = tlin - n * tpi;
  const int ty0 = (trem / a.tiles_x) * a.TH, tx0 = (trem % a.tiles_x) * a.TW;
  const int RP = a.RH * a.RW, TP = a.TH * a.TW;

  // ---- tables, zero fill, chain weights
  for (int q = tid; q < a.RPA; q += CSP_THREADS) {
    const int ry = q / a.RW, rx = q - ry * a.RW;
    const int iy = ty0 - a.halo + ry, ix = tx0 - a.halo + rx;
    const bool in = q < RP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    goff[q] = in ? n * a.xsn + iy * a.xsh + ix * a.xsw : MGDT_OOB;
    const int cy = ry - a.halo, cx = rx - a.halo;
    ctab[q] = (q < RP && (unsigned)cy < (unsigned)a.TH && (unsigned)cx < (unsigned)a.TW) ? (short)(cy * a.TW + cx) : (short)-1;
  }
  for (int t = tid; t < a.TPA; t += CSP_THREADS) {
    const int cy = t / a.TW, cx = t - cy * a.TW;
    yoff[t] = t < TP ? n * a.ysn + (ty0 + cy) * a.ysh + (tx0 + cx) * a.ysw : MGDT_OOB;
  }
  {
    const int words = (int)(((size_t)2 * a.RPA * a.PS) >> 4);
    uint4* z = (uint4*)Pb;
    for (int i = tid; i < words; i += CSP_THREADS) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (MODE == 0) {
    for (int i = tid; i < a.chain_words; i += CSP_THREADS) ((uint4*)wl)[i] = ((const uint4*)a.front)[i];
    float* bl = (float*)(wl + (size_t)a.chain_words * 16);
    for (int i = tid; i < 3 * NBK * 16; i += CSP_THREADS) bl[i] = ((const float*)(a.front + (size_t)a.chain_words * 16))[i];
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const int ngr = (RP + 15) >> 4;                                // 16-pixel groups of the region

  // ================================================================ front
  if (MODE == 0) {
    const char* const wlane = wl + lane * 16;
    const float* bl = (const float*)(wl + (size_t)a.chain_words * 16);
    for (int grp = wave; grp < ngr; grp += CSP_NW) {
      const int q = grp * 16 + r;
      const int go = goff[q];                                     // q < RPA always (RPA is padded to whole groups)
      const int ct = ctab[q];
      f32x4 X[4][NBK];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
          const int c = blk * 16 + 4 * g;
          const int dead = c >= WD ? MGDT_OOB : 0;
          X[i][blk] = bload4<bf16>(xrs, (uint32_t)(go | dead) + (uint32_t)((i * WD + c) * 2));
        }
      f32x4 prev[NBK];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        bf16x8 Bf[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int blk = kc * 2 + e / 4;
            float vv = 0.f;
            if (blk < NBK) vv = X[i][blk][e % 4] + (i ? prev[blk][e % 4] : 0.f);
            Bf[kc][e] = (bf16)vv;
          }
#pragma unroll
        for (int ob = 0; ob < NBK; ++ob) {
          f32x4 acc = *(const f32x4*)(bl + (i * NBK + ob) * 16 + 4 * g);
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) acc = mma(*(const bf16x8*)(wlane + ((i * KC + kc) * NBK + ob) * 1024), Bf[kc], acc);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = (float)(bf16)csp_act(acc[j], a.act);
          prev[ob] = acc;
          const int c = ob * 16 + 4 * g;
          if (c < WD && ct >= 0) lds_store4(catb + ct * a.CS + (i * WD + c) * 2, acc);
        }
      }
      // bottleneck input = sp2 + x3 (the pending add of block.py:259), zero outside the image
#pragma unroll
      for (int ob = 0; ob < NBK; ++ob) {
        const int c = ob * 16 + 4 * g;
        f32x4 p0 = prev[ob] + X[3][ob];
        if (go == MGDT_OOB) p0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < WD) lds_store4(Pb + q * a.PS + c * 2, p0);
      }
    }
  } else {
    // C2f front: 1x1 conv Cin -> 2c over the region; wave = (cout block, pixel-group phase)
    const int nbv = wave % a.front_nb, gv = wave / a.front_nb, gst = CSP_NW / a.front_nb;
    bf16x8 A[CSP_FRONT_NCH_MAX];
#pragma unroll
    for (int kc = 0; kc < CSP_FRONT_NCH_MAX; ++kc)
      if (kc < a.front_nch) A[kc] = *(const bf16x8*)(a.front + ((size_t)(kc * a.front_nb + nbv) * 64 + lane) * 16);
    const f32x4 bias = *(const f32x4*)(a.front_bias + nbv * 16 + 4 * g);
    const int cc = nbv * 16 + 4 * g;                             // output channel of the 2c-wide cv1 output
    for (int grp = gv; grp < ngr; grp += gst) {
      const int q = grp * 16 + r;
      const int go = goff[q];
      const int ct = ctab[q];
      bf16x8 Bf[CSP_FRONT_NCH_MAX];
#pragma unroll
      for (int kc = 0; kc < CSP_FRONT_NCH_MAX; ++kc)
        if (kc < a.front_nch) Bf[kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, (uint32_t)go + (uint32_t)((kc * 4 + g) * 16), 0, 0));
      f32x4 acc = bias;
#pragma unroll
      for (int kc = 0; kc < CSP_FRONT_NCH_MAX; ++kc)
        if (kc < a.front_nch) acc = mma(A[kc], Bf[kc], acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = csp_act(acc[j], a.act);
      if (go == MGDT_OOB) acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ct >= 0) lds_store4(catb + ct * a.CS + cc * 2, acc);
      if (cc >= WD) lds_store4(Pb + q * a.PS + (cc - WD) * 2, acc);          // second half = the bottleneck chain's input (block.py:201-203)
    }
  }
  __syncthreads();

  // ================================================================ middle: 2 * nbtl 3x3 convs, P -> T -> P ...
  {
    const int nbv = wave % NB, gv = wave / NB, gst = CSP_NW / NB;
    int boff[NCH];                                               // this lane's byte offset of chunk kc's piece relative to its pixel
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const int p = kc * 4 + g;
      int tap = p / CP, cp = p - tap * CP;
      if (tap >= 9) { tap = 4; cp = 0; }                         // padded piece: weights are zero, read something finite
      boff[kc] = ((tap / 3 - 1) * a.RW + (tap % 3 - 1)) * a.PS + cp * 16;
    }
    const int cch = nbv * 16 + 4 * g;                            // this lane's first output channel
    const int slot0 = (MODE == 0 ? 3 : 2) * WD;                  // concat offset of the first bottleneck output
    for (int j = 0; j < 2 * a.nbtl; ++j) {
      const char* in = (j & 1) ? Tb : Pb;
      char* out = (j & 1) ? Pb : Tb;
      bf16x8 A[NCH];
#pragma unroll
      for (int kc = 0; kc < NCH; ++kc) A[kc] = *(const bf16x8*)(a.mid[j] + ((size_t)(kc * NB + nbv) * 64 + lane) * 16);
      const f32x4 bias = *(const f32x4*)(a.mid_bias[j] + nbv * 16 + 4 * g);
      const int lo = (j + 1) * a.RW + (j + 1), hi = (a.RH - j - 2) * a.RW + (a.RW - j - 1);
      const int ng = (hi - lo + 15) >> 4;
      const bool second = j & 1;
      for (int grp = gv; grp < ng; grp += gst) {
        const int q = lo + grp * 16 + r;
        const char* pin = in + q * a.PS;
        f32x4 acc = bias;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) acc = mma(A[kc], *(const bf16x8*)(pin + boff[kc]), acc);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = csp_act(acc[jj], a.act);
        if (q < hi && cch < WD) {
          char* po = out + q * a.PS + cch * 2;
          if (second && a.shortcut) acc += lds_load4(po);       // x + cv2(cv1(x)): `out` still holds the bottleneck's input at this pixel
          if (goff[q] == MGDT_OOB) acc = f32x4{0.f, 0.f, 0.f, 0.f};
          lds_store4(po, acc);
          if (second) {
            const int ct = ctab[q];
            if (ct >= 0) lds_store4(catb + ct * a.CS + (slot0 + (j >> 1) * WD + cch) * 2, acc);
          }
        }
      }
      __syncthreads();
    }
  }

  // ================================================================ back: 1x1 conv over the concat, store, per-tile channel sums
  {
    const int tg = (TP + 15) >> 4;
    int coff[CSP_NCHB_MAX];
#pragma unroll
    for (int kc = 0; kc < CSP_NCHB_MAX; ++kc) {
      const int p = kc * 4 + g;
      coff[kc] = p * 8 < a.catC ? p * 16 : 0;                   // padded piece: zero weights, finite data
    }
    // bins of adaptive_avg_pool2d(2): the tile lies inside one of them (host-checked for MSPA)
    const int bin = (ty0 >= a.H / 2 ? 2 : 0) + (tx0 >= a.W / 2 ? 1 : 0);
    for (int ob = wave; ob < a.nbo; ob += CSP_NW) {
      bf16x8 A[CSP_NCHB_MAX];
#pragma unroll
      for (int kc = 0; kc < CSP_NCHB_MAX; ++kc)
        if (kc < a.nchb) A[kc] = *(const bf16x8*)(a.back + ((size_t)(kc * a.nbo + ob) * 64 + lane) * 16);
      const f32x4 bias = *(const f32x4*)(a.back_bias + ob * 16 + 4 * g);
      const int co = ob * 16 + 4 * g;
      const int dead = co >= a.Cout ? MGDT_OOB : 0;
      f32x4 psum = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int grp = 0; grp < tg; ++grp) {
        const int t = grp * 16 + r;
        const char* pc = catb + t * a.CS;
        f32x4 acc = bias;
#pragma unroll
        for (int kc = 0; kc < CSP_NCHB_MAX; ++kc)
          if (kc < a.nchb) acc = mma(A[kc], *(const bf16x8*)(pc + coff[kc]), acc);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = (float)(bf16)csp_act(acc[jj], a.act);
        bstore4<bf16>(yrs, (uint32_t)(yoff[t] | dead) + (uint32_t)(co * 2), acc);
        if (t < TP) psum += acc;
      }
      if (a.pool) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) psum[jj] += __shfl_xor(psum[jj], m, 64);
        if (r == 0 && co < a.Cout) {
          float* pp = a.pool + (((size_t)n * tpi + trem) * a.Cout + co) * 5;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            pp[jj * 5 + 0] = psum[jj];
#pragma unroll
            for (int b = 0; b < 4; ++b) pp[jj * 5 + 1 + b] = b == bin ? psum[jj] : 0.f;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct CspGeom { int TH, TW, halo, RH, RW, RPA, TPA, PS, CS, tiles_x, tiles_y; size_t lds; };

static bool csp_geometry(int mode, int H, int W, int wd, int nbtl, int catC, size_t extra, int th, int tw, CspGeom* g) {
  g->TH = th; g->TW = tw; g->halo = 2 * nbtl;
  g->RH = th + 2 * g->halo; g->RW = tw + 2 * g->halo;
  const int RP = g->RH * g->RW;
  g->RPA = ((RP + 15) / 16) * 16 + ((g->RW + 1 + 15) / 16) * 16 + 16;      // whole groups + the reach of the last group's taps
  g->TPA = ((th * tw + 15) / 16) * 16;
  g->PS = wd == 8 ? 16 : wd * 2 + 16;
  g->CS = catC * 2 + 16;
  g->tiles_x = W / tw; g->tiles_y = H / th;
  g->lds = (size_t)g->RPA * 4 + (size_t)g->TPA * 4 + (size_t)g->RPA * 2 + 32 + 2 * (size_t)g->RPA * g->PS + (size_t)g->TPA * g->CS + extra;
  (void)mode;
  return g->lds <= 156 * 1024 && RP < 32000;
}

static int csp_chain_nbk(int wd) { return wd >= 16 ? wd / 16 : 1; }

/* supported configurations of the fused block: bf16, wd in {8,16,32,64}, n in {1,2}; MSPA: even H, W; C2f: Cin % 32 == 0, Cin <= 256 */
extern "C" int mgdt_csp_block_supported(int mode, int cin, int cout, int wd, int nbtl, int h, int w, int dtype) {
  if (dtype != MGDT_BF16 || (wd != 8 && wd != 16 && wd != 32 && wd != 64) || nbtl < 1 || nbtl > 2 || cout % 4) return 0;
  const int catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  if (catC > 256 || catC % 8) return 0;
  if (mode == 0) {
    if (cin != 4 * wd || h % 2 || w % 2 || h < 4 || w < 4) return 0;
  } else {
    if (cin % 32 || cin > 256 || wd < 16) return 0;
  }
  return 1;
}

static bool csp_pick_tile(int mode, int H, int W, int N, int wd, int nbtl, int catC, size_t extra, CspGeom* best) {
  const char* e = getenv("MGDT_CSP_TILE");       // experiment knob "th,tw" (not part of the ABI)
  int fth = 0, ftw = 0;
  if (e && sscanf(e, "%d,%d", &fth, &ftw) != 2) fth = ftw = 0;
  const int qh = mode == 0 ? H / 2 : H, qw = mode == 0 ? W / 2 : W;     // MSPA: a tile must lie inside one adaptive_avg_pool2d(2) bin
  double best_cost = 1e30;
  bool found = false;
  for (int th = 2; th <= std::min(qh, 32); ++th) {
    if (qh % th) continue;
    for (int tw = 2; tw <= std::min(qw, 32); ++tw) {
      if (qw % tw) continue;
      if (fth && (th != fth || tw != ftw)) continue;
      CspGeom g;
      if (!csp_geometry(mode, H, W, wd, nbtl, catC, extra, th, tw, &g)) continue;
      const double wgs = (double)N * g.tiles_x * g.tiles_y;
      const double ratio = (double)(g.RH * g.RW) / (th * tw);
      const double fill = (double)(th * tw) / g.TPA;              // lanes of the tile's last pixel group that do work
      double cost = ratio / fill;
      if (wgs < 512) cost *= 512.0 / wgs;                        // fewer workgroups than 2 per CU: the chip is not filled
      if (g.lds > 76 * 1024) cost *= 1.3;                        // one workgroup per CU: nothing overlaps its barriers
      if (cost < best_cost) { best_cost = cost; *best = g; found = true; }
    }
  }
  return found;
}

/* x: N x H x W x Cin view, y: N x H x W x Cout view (bf16 NHWC).  front: MSPA - the blob of mgdt_pw_chain_pack (3 convs); C2f - the
 * mgdt_conv_pack panel of cv1 (+ front_bias).  mid[2*nbtl] / mid_bias: mgdt_conv_pack panels of the bottlenecks' 3x3 convs in
 * execution order.  back / back_bias: the 1x1 conv over the concat.  pool: NULL or fp32 [N][tiles][Cout][5] in the layout of
 * mgdt_spr_pool_fwd with `tiles` splits (query the count with mgdt_csp_block_tiles). */
extern "C" int mgdt_csp_block_tiles(int mode, int n, int cin, int cout, int wd, int nbtl, int h, int w, int* geom6) {
  if (!mgdt_csp_block_supported(mode, cin, cout, wd, nbtl, h, w, MGDT_BF16)) return 0;
  const int catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  const int nbk = csp_chain_nbk(wd);
  const size_t extra = mode == 0 ? (size_t)3 * ((nbk + 1) / 2) * nbk * 1024 + 3 * nbk * 16 * 4 : 0;
  CspGeom g;
  if (!csp_pick_tile(mode, h, w, n, wd, nbtl, catC, extra, &g)) return 0;
  if (geom6) { geom6[0] = g.TH; geom6[1] = g.TW; geom6[2] = g.RH; geom6[3] = g.RW; geom6[4] = (int)g.lds; geom6[5] = n * g.tiles_x * g.tiles_y; }
  return g.tiles_x * g.tiles_y;
}

extern "C" int mgdt_csp_block_fwd(int mode, const mgdt_view* x, const void* front, const float* front_bias, const void* const* mid,
                                  const float* const* mid_bias, int nbtl, int shortcut, const void* back, const float* back_bias, int wd, int act,
                                  const mgdt_view* y, float* pool, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !front || !mid || !mid_bias || !back || !back_bias) MGDT_FAIL(MGDT_BAD_ARG, "csp_block: null/empty argument");
  if (!mgdt_csp_block_supported(mode, x->c, y->c, wd, nbtl, x->h, x->w, dtype))
    MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: mode=%d cin=%d cout=%d wd=%d n=%d %dx%d dtype=%d not covered", mode, x->c, y->c, wd, nbtl, x->h, x->w, dtype);
  if (mode == 1 && !front_bias) MGDT_FAIL(MGDT_BAD_ARG, "csp_block: C2f front needs its bias");
  if (x->n != y->n || x->h != y->h || x->w != y->w) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: x and y must have one spatial size");
  for (int j = 0; j < 2 * nbtl; ++j)
    if (!mid[j] || !mid_bias[j]) MGDT_FAIL(MGDT_BAD_ARG, "csp_block: missing conv panel %d", j);
  CspArgs a;
  memset(&a, 0, sizeof(a));
  const long sz = 2;
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes, int q) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz;
    if (v->sc != 1 || v->sw % q || v->sh % q || v->sn % q || (uintptr_t)v->p % (q * sz) || ext >= 0x7fffffffL) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * sz); *sh = (int)(v->sh * sz); *sw = (int)(v->sw * sz); *bytes = (uint32_t)ext;
  };
  const char* yp = nullptr;
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes, mode == 0 ? 4 : 8);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes, 4);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: views must be aligned NHWC (sc == 1) and span < 2 GiB");
  a.y = (char*)yp;
  a.front = (const char*)front; a.front_bias = front_bias;
  for (int j = 0; j < 2 * nbtl; ++j) { a.mid[j] = (const char*)mid[j]; a.mid_bias[j] = mid_bias[j]; }
  a.back = (const char*)back; a.back_bias = back_bias; a.pool = pool;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = y->c; a.wd = wd; a.nbtl = nbtl; a.shortcut = shortcut; a.act = act;
  a.catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  a.nchb = (a.catC / 8 + 3) / 4; a.nbo = (a.Cout + 15) / 16;
  a.front_nch = x->c / 32; a.front_nb = mode == 1 ? 2 * wd / 16 : 1;
  if (mode == 1 && (CSP_NW % a.front_nb || a.front_nch > CSP_FRONT_NCH_MAX)) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: C2f front %d -> %d not covered", x->c, 2 * wd);
  const int nbk = csp_chain_nbk(wd);
  a.chain_words = mode == 0 ? 3 * ((nbk + 1) / 2) * nbk * 64 : 0;
  const size_t extra = mode == 0 ? (size_t)a.chain_words * 16 + 3 * nbk * 16 * 4 : 0;
  CspGeom g;
  if (!csp_pick_tile(mode, a.H, a.W, a.N, wd, nbtl, a.catC, extra, &g)) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: no tile fits %dx%d wd=%d n=%d", a.H, a.W, wd, nbtl);
  a.TH = g.TH; a.TW = g.TW; a.halo = g.halo; a.RH = g.RH; a.RW = g.RW; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
  a.RPA = g.RPA; a.TPA = g.TPA; a.PS = g.PS; a.CS = g.CS;
  a.total_tiles = a.N * g.tiles_x * g.tiles_y;
  a.per_xcd = cdiv(a.total_tiles, 8);
  const int grid = 8 * a.per_xcd;
  hipStream_t st = (hipStream_t)s;
#define CSP_LAUNCH(WDV, MODEV)                                                                                              \
  do {                                                                                                                      \
    static bool attr = false;                                                                                               \
    if (!attr) {                                                                                                            \
      hipError_t e_ = hipFuncSetAttribute((const void*)csp_block_kernel<WDV, MODEV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e_ != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "csp_block: hipFuncSetAttribute: %s", hipGetErrorString(e_));         \
      attr = true;                                                                                                          \
    }                                                                                                                       \
    csp_block_kernel<WDV, MODEV><<<grid, CSP_THREADS, g.lds, st>>>(a);                                                      \
  } while (0)
  if (mode == 0) {
    switch (wd) { case 8: CSP_LAUNCH(8, 0); break; case 16: CSP_LAUNCH(16, 0); break; case 32: CSP_LAUNCH(32, 0); break; default: CSP_LAUNCH(64, 0); break; }
  } else {
    switch (wd) { case 16: CSP_LAUNCH(16, 1); break; case 32: CSP_LAUNCH(32, 1); break; default: CSP_LAUNCH(64, 1); break; }
  }
#undef CSP_LAUNCH
  MGDT_CHECK_LAUNCH("csp_block_fwd");
  return MGDT_OK;
}
