// One launch per CSP block (MSPA_C2f / C2f) of the bf16 inference path: the whole block for one spatial tile of one image stays on the
// CU.  Reference: nn/modules/block.py:209-287 (MSPA_C2f), :187-207 (C2f), :514-526 (Bottleneck).
//
//   front   MSPA: sp0 = cv0(x0), sp1 = cv1(sp0 + x1), sp2 = cv2(sp1 + x2)   (register-chained MFMAs, as mgdt_pw_chain3_fwd)
//           C2f : [y0 | y1] = cv1(x) comes from the ordinary 1x1 conv launch; the front copies it into the tile buffers
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
#include <algorithm>
#include <vector>

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
  int RPA, TPA, PS, CS;                // allocated region / tile pixels, pixel strides (bytes) of P/T and of the concat buffer
  int chain_words;                     // MSPA: 16-byte words of the chain blob's weight part
  unsigned long long* dbg;             // MGDT_CSP_DBG: 8 wall-clock stamps (10 ns units) per workgroup
  int pool_gst;                        // back phase: waves per cout block = pool slots per tile (1 when there are >= 8 cout blocks)
  FastDiv fd_rw, fd_tw, fd_tx, fd_tpi; // table phase: division by region width / tile width / tiles per row / tiles per image without the ~40-instruction sequence
};

constexpr int CSP_THREADS = 512;
constexpr int CSP_NW = CSP_THREADS / 64;
constexpr int CSP_NCHB_MAX = 8;        // back conv: K <= 256 concat channels (host-side bound)

// SiLU only, branch-free (the host refuses anything else): a run-time activation switch here turned every epilogue into four serial
// branchy exp -> rcp chains; as straight-line code the four values' transcendentals interleave
__device__ __forceinline__ float csp_act(float v, int) { return v * fast_sigmoid(v); }

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
__global__ __launch_bounds__(CSP_THREADS, ((WD <= 16 || (WD == 32 && MODE == 1)) ? 4 : 2)) void csp_block_kernel(const CspArgs a) {   // 4: two workgroups per CU (<= 128 VGPRs) where LDS allows it
  constexpr int CP = WD / 8;                         // 16-byte pieces per tap
  constexpr int NCHB = WD == 8 ? 2 : (WD == 16 ? 3 : (WD == 32 ? 5 : 8));   // K chunks of the back conv at n = 2 (concat <= 256 channels)
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
  const int n = (int)fdiv((uint32_t)tlin, a.fd_tpi), trem = tlin - n * tpi;
  const int tyi = (int)fdiv((uint32_t)trem, a.fd_tx);
  const int ty0 = tyi * a.TH, tx0 = (trem - tyi * a.tiles_x) * a.TW;
  const int RP = a.RH * a.RW, TP = a.TH * a.TW;
  auto stamp = [&](int k) __attribute__((always_inline)) { if (a.dbg && tid == 0) a.dbg[(size_t)tlin * 8 + k] = __builtin_amdgcn_s_memrealtime(); };
  stamp(0);

  // weights-stationary: this wave's cout block of the running 3x3 conv lives in registers.  Every conv's fragments are requested one
  // phase AHEAD (the first ones right here, before the tables and the front), so no phase starts by waiting on L2.
  const int nbv = wave % NB, gvm = wave / NB, gstm = CSP_NW / NB;
  bf16x8 A[NCH];
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) A[kc] = *(const bf16x8*)(a.mid[0] + ((size_t)(kc * NB + nbv) * 64 + lane) * 16);
  f32x4 bias = *(const f32x4*)(a.mid_bias[0] + nbv * 16 + 4 * g);

  // ---- tables, zero fill, chain weights
  for (int q = tid; q < a.RPA; q += CSP_THREADS) {
    const int ry = (int)fdiv((uint32_t)q, a.fd_rw), rx = q - ry * a.RW;
    const int iy = ty0 - a.halo + ry, ix = tx0 - a.halo + rx;
    const bool in = q < RP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    goff[q] = in ? n * a.xsn + iy * a.xsh + ix * a.xsw : MGDT_OOB;
    const int cy = ry - a.halo, cx = rx - a.halo;
    ctab[q] = (q < RP && (unsigned)cy < (unsigned)a.TH && (unsigned)cx < (unsigned)a.TW) ? (short)(cy * a.TW + cx) : (short)-1;
  }
  for (int t = tid; t < a.TPA; t += CSP_THREADS) {
    const int cy = (int)fdiv((uint32_t)t, a.fd_tw), cx = t - cy * a.TW;
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
  stamp(1);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const int ngr = (RP + 15) >> 4;                                // 16-pixel groups of the region

  // ================================================================ front
  if (MODE == 0) {
    // each wave takes a contiguous run of pixel groups and requests the inputs of U groups before it touches the first one: the phase is
    // bound by HBM/L2 latency, not by work (a wave that waits for one group at a time leaves the CU with 16 requests in flight)
    constexpr int U = WD == 64 ? 2 : (WD == 32 ? 3 : 5);
    const char* const wlane = wl + lane * 16;
    const float* bl = (const float*)(wl + (size_t)a.chain_words * 16);
    const int per = ngr / CSP_NW, rem = ngr - per * CSP_NW;   // balanced: the first `rem` waves take one group more
    const int gbeg = wave * per + min(wave, rem), gend = gbeg + per + (wave < rem ? 1 : 0);
    for (int g0 = gbeg; g0 < gend; g0 += U) {
      int go[U], ct[U];
      csp_raw2 XR[U][4][NBK];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool gv_ = g0 + u < gend;
        const int q = (g0 + u) * 16 + r;
        go[u] = gv_ ? goff[q] : MGDT_OOB;
        ct[u] = gv_ ? (int)ctab[q] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int blk = 0; blk < NBK; ++blk) {
            const int c = blk * 16 + 4 * g;
            const int dead = c >= WD ? MGDT_OOB : 0;
            XR[u][i][blk] = __builtin_amdgcn_raw_buffer_load_b64(xrs, (uint32_t)(go[u] | dead) + (uint32_t)((i * WD + c) * 2), 0, 0);
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (g0 + u >= gend) break;                               // wave-uniform
        const int q = (g0 + u) * 16 + r;
        f32x4 X[4][NBK];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int blk = 0; blk < NBK; ++blk) {
            const bf16x4 o = __builtin_bit_cast(bf16x4, XR[u][i][blk]);
            X[i][blk] = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
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
            if (c < WD && ct[u] >= 0) lds_store4(catb + ct[u] * a.CS + (i * WD + c) * 2, acc);
          }
        }
        // bottleneck input = sp2 + x3 (the pending add of block.py:259), zero outside the image
#pragma unroll
        for (int ob = 0; ob < NBK; ++ob) {
          const int c = ob * 16 + 4 * g;
          f32x4 p0 = prev[ob] + X[3][ob];
          if (go[u] == MGDT_OOB) p0 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (c < WD) lds_store4(Pb + q * a.PS + c * 2, p0);
        }
      }
    }
  } else {
    // C2f: x is cv1's output [y0 | y1] (2*WD channels, written by the 1x1 conv launch in front of this one: a wide streaming GEMM that
    // gains nothing from living here); the front copies the tile's own pixels into the concat buffer and y1 (+ halo) into P
    constexpr int NB2 = 2 * WD / 16;
    constexpr int U = WD == 64 ? 2 : 4;
    const int per = ngr / CSP_NW, rem = ngr - per * CSP_NW;
    const int gbeg = wave * per + min(wave, rem), gend = gbeg + per + (wave < rem ? 1 : 0);
    for (int g0 = gbeg; g0 < gend; g0 += U) {
      int go[U], ct[U];
      csp_raw2 XR[U][NB2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool gv_ = g0 + u < gend;
        const int q = (g0 + u) * 16 + r;
        go[u] = gv_ ? goff[q] : MGDT_OOB;
        ct[u] = gv_ ? (int)ctab[q] : -1;
#pragma unroll
        for (int blk = 0; blk < NB2; ++blk)
          XR[u][blk] = __builtin_amdgcn_raw_buffer_load_b64(xrs, (uint32_t)go[u] + (uint32_t)((blk * 16 + 4 * g) * 2), 0, 0);   // zeros outside the image
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (g0 + u >= gend) break;                               // wave-uniform
        const int q = (g0 + u) * 16 + r;
#pragma unroll
        for (int blk = 0; blk < NB2; ++blk) {
          const int cc = blk * 16 + 4 * g;
          if (ct[u] >= 0) *(csp_raw2*)(catb + ct[u] * a.CS + cc * 2) = XR[u][blk];
          if (cc >= WD) *(csp_raw2*)(Pb + q * a.PS + (cc - WD) * 2) = XR[u][blk];      // second half = the bottleneck chain's input (block.py:201-203)
        }
      }
    }
  }
  stamp(2);
  __syncthreads();
  stamp(3);

  // ================================================================ middle: 2 * nbtl 3x3 convs, P -> T -> P ...
  constexpr bool PREB = WD <= 16;                                // prefetch the back conv's first fragments too (register budget)
  bf16x8 AB[NCHB];
  f32x4 biasB = f32x4{0.f, 0.f, 0.f, 0.f};
  const int bgst = a.pool_gst, bgv = wave / (CSP_NW / bgst), obst = CSP_NW / bgst, ob0 = wave % obst;
  {
    const int gv = gvm, gst = gstm;
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
      // narrow convs: the next conv's fragments are requested now and land while this conv computes (wide ones: registers are better
      // spent on two workgroups per CU, their fragments are loaded after the barrier)
      constexpr bool PREN = WD <= 16;
      bf16x8 An[PREN ? NCH : 1];
      f32x4 biasn = bias;
      const bool more = j + 1 < 2 * a.nbtl;                     // uniform
      if (more) {
        if (PREN) {
#pragma unroll
          for (int kc = 0; kc < NCH; ++kc) An[kc] = *(const bf16x8*)(a.mid[j + 1] + ((size_t)(kc * NB + nbv) * 64 + lane) * 16);
          biasn = *(const f32x4*)(a.mid_bias[j + 1] + nbv * 16 + 4 * g);
        }
      } else if (PREB && ob0 < a.nbo) {
#pragma unroll
        for (int kc = 0; kc < NCHB; ++kc)
          if (kc < a.nchb) AB[kc] = *(const bf16x8*)(a.back + ((size_t)(kc * a.nbo + ob0) * 64 + lane) * 16);
        biasB = *(const f32x4*)(a.back_bias + ob0 * 16 + 4 * g);
      }
      const int lo = (j + 1) * a.RW + (j + 1), hi = (a.RH - j - 2) * a.RW + (a.RW - j - 1);
      const int ng = (hi - lo + 15) >> 4;
      const bool second = j & 1;
      auto finish = [&](int q, f32x4 acc) __attribute__((always_inline)) {
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
      };
      for (int grp = gv; grp < ng; grp += 2 * gst) {             // two pixel groups per step: two independent accumulator chains
        const int qa = lo + grp * 16 + r;
        const bool hasb = grp + gst < ng;                        // wave-uniform
        const int qb = hasb ? qa + gst * 16 : qa;
        const char* pa = in + qa * a.PS;
        const char* pb = in + qb * a.PS;
        f32x4 acca = bias, accb = bias;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
          acca = mma(A[kc], *(const bf16x8*)(pa + boff[kc]), acca);
          accb = mma(A[kc], *(const bf16x8*)(pb + boff[kc]), accb);
        }
        finish(qa, acca);
        if (hasb) finish(qb, accb);
      }
      __syncthreads();
      if (j == 0) stamp(4);
      if (more) {
        if (PREN) {
#pragma unroll
          for (int kc = 0; kc < NCH; ++kc) A[kc] = An[kc];
          bias = biasn;
        } else {
#pragma unroll
          for (int kc = 0; kc < NCH; ++kc) A[kc] = *(const bf16x8*)(a.mid[j + 1] + ((size_t)(kc * NB + nbv) * 64 + lane) * 16);
          bias = *(const f32x4*)(a.mid_bias[j + 1] + nbv * 16 + 4 * g);
        }
      }
    }
  }
  stamp(5);

  // ================================================================ back: 1x1 conv over the concat, store, per-tile channel sums
  {
    const int tg = (TP + 15) >> 4;
    int coff[NCHB];
#pragma unroll
    for (int kc = 0; kc < NCHB; ++kc) {
      const int p = kc * 4 + g;
      coff[kc] = p * 8 < a.catC ? p * 16 : 0;                   // padded piece: zero weights, finite data
    }
    // few cout blocks (< 8): the waves that share a block split the tile's pixel groups; their partial sums go to separate pool slots
    const int gst = bgst, gv = bgv;
    for (int ob = ob0; ob < a.nbo; ob += obst) {
      if (!PREB || ob != ob0) {
#pragma unroll
        for (int kc = 0; kc < NCHB; ++kc)
          if (kc < a.nchb) AB[kc] = *(const bf16x8*)(a.back + ((size_t)(kc * a.nbo + ob) * 64 + lane) * 16);
        biasB = *(const f32x4*)(a.back_bias + ob * 16 + 4 * g);
      }
      const int co = ob * 16 + 4 * g;
      const int dead = co >= a.Cout ? MGDT_OOB : 0;
      f32x4 psum = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int grp = gv; grp < tg; grp += gst) {
        const int t = grp * 16 + r;
        const char* pc = catb + t * a.CS;
        f32x4 acc = biasB;
#pragma unroll
        for (int kc = 0; kc < NCHB; ++kc)
          if (kc < a.nchb) acc = mma(AB[kc], *(const bf16x8*)(pc + coff[kc]), acc);
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
          *(f32x4*)(a.pool + (((size_t)n * tpi + trem) * gst + gv) * a.Cout + co) = psum;
        }
      }
    }
  }
  stamp(6);
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
static int csp_pool_gst(int cout) { const int nbo = (cout + 15) / 16; return (nbo < CSP_NW && CSP_NW % nbo == 0) ? CSP_NW / nbo : 1; }

/* supported configurations of the fused block: bf16, wd in {8,16,32,64}, n in {1,2}; MSPA: even H, W; C2f: Cin % 32 == 0, Cin <= 256 */
extern "C" int mgdt_csp_block_supported(int mode, int cin, int cout, int wd, int nbtl, int h, int w, int dtype) {
  if (dtype != MGDT_BF16 || (wd != 8 && wd != 16 && wd != 32 && wd != 64) || nbtl < 1 || nbtl > 2 || cout % 4) return 0;
  const int catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  if (catC > 256 || catC % 8) return 0;
  if (mode == 0) {
    if (cin != 4 * wd || h % 2 || w % 2 || h < 4 || w < 4) return 0;
  } else {
    if (cin != 2 * wd || wd < 16) return 0;
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
      if (wgs < 256) cost *= 256.0 / wgs;                        // fewer workgroups than CUs: the chip is not filled (measured: 128 big tiles beat 512 small ones at 20x20)
      if (g.lds > 76 * 1024) cost *= 1.3;                        // one workgroup per CU: nothing overlaps its barriers
      if (cost < best_cost) { best_cost = cost; *best = g; found = true; }
    }
  }
  return found;
}

/* x: N x H x W x Cin view, y: N x H x W x Cout view (bf16 NHWC).  front: MSPA - the blob of mgdt_pw_chain_pack (3 convs); C2f - the
 * mgdt_conv_pack panel of cv1 (+ front_bias).  mid[2*nbtl] / mid_bias: mgdt_conv_pack panels of the bottlenecks' 3x3 convs in
 * execution order.  back / back_bias: the 1x1 conv over the concat.  pool: NULL or fp32 [N][slots][Cout] per-tile channel sums of y
 * (slots and the tile grid: mgdt_csp_block_tiles). */
extern "C" int mgdt_csp_block_tiles(int mode, int n, int cin, int cout, int wd, int nbtl, int h, int w, int* geom6) {
  if (!mgdt_csp_block_supported(mode, cin, cout, wd, nbtl, h, w, MGDT_BF16)) return 0;
  const int catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  const int nbk = csp_chain_nbk(wd);
  const size_t extra = mode == 0 ? (size_t)3 * ((nbk + 1) / 2) * nbk * 1024 + 3 * nbk * 16 * 4 : 0;
  CspGeom g;
  if (!csp_pick_tile(mode, h, w, n, wd, nbtl, catC, extra, &g)) return 0;
  if (geom6) {
    geom6[0] = g.TH; geom6[1] = g.TW; geom6[2] = g.RH; geom6[3] = g.RW; geom6[4] = (int)g.lds; geom6[5] = n * g.tiles_x * g.tiles_y;
    geom6[6] = g.tiles_x; geom6[7] = g.tiles_y;
  }
  return g.tiles_x * g.tiles_y * csp_pool_gst(cout);      // pool slots per image
}

extern "C" int mgdt_csp_block_fwd(int mode, const mgdt_view* x, const void* front, const float* front_bias, const void* const* mid,
                                  const float* const* mid_bias, int nbtl, int shortcut, const void* back, const float* back_bias, int wd, int act,
                                  const mgdt_view* y, float* pool, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || (mode == 0 && !front) || !mid || !mid_bias || !back || !back_bias) MGDT_FAIL(MGDT_BAD_ARG, "csp_block: null/empty argument");
  if (!mgdt_csp_block_supported(mode, x->c, y->c, wd, nbtl, x->h, x->w, dtype))
    MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: mode=%d cin=%d cout=%d wd=%d n=%d %dx%d dtype=%d not covered", mode, x->c, y->c, wd, nbtl, x->h, x->w, dtype);
  if (x->n != y->n || x->h != y->h || x->w != y->w) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: x and y must have one spatial size");
  if (act != MGDT_ACT_SILU) MGDT_FAIL(MGDT_BAD_ARG, "csp_block: activation %d (only SiLU, the blocks' default, is built)", act);
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
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes, 4);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes, 4);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: views must be aligned NHWC (sc == 1) and span < 2 GiB");
  a.y = (char*)yp;
  a.front = (const char*)front; a.front_bias = front_bias;
  for (int j = 0; j < 2 * nbtl; ++j) { a.mid[j] = (const char*)mid[j]; a.mid_bias[j] = mid_bias[j]; }
  a.back = (const char*)back; a.back_bias = back_bias; a.pool = pool;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = y->c; a.wd = wd; a.nbtl = nbtl; a.shortcut = shortcut; a.act = act;
  a.catC = (mode == 0 ? 3 : 2) * wd + nbtl * wd;
  a.nchb = (a.catC / 8 + 3) / 4; a.nbo = (a.Cout + 15) / 16; a.pool_gst = csp_pool_gst(a.Cout);
  const int nbk = csp_chain_nbk(wd);
  a.chain_words = mode == 0 ? 3 * ((nbk + 1) / 2) * nbk * 64 : 0;
  const size_t extra = mode == 0 ? (size_t)a.chain_words * 16 + 3 * nbk * 16 * 4 : 0;
  CspGeom g;
  if (!csp_pick_tile(mode, a.H, a.W, a.N, wd, nbtl, a.catC, extra, &g)) MGDT_FAIL(MGDT_BAD_SHAPE, "csp_block: no tile fits %dx%d wd=%d n=%d", a.H, a.W, wd, nbtl);
  a.TH = g.TH; a.TW = g.TW; a.halo = g.halo; a.RH = g.RH; a.RW = g.RW; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
  a.RPA = g.RPA; a.TPA = g.TPA; a.PS = g.PS; a.CS = g.CS;
  a.total_tiles = a.N * g.tiles_x * g.tiles_y;
  a.fd_rw = make_fastdiv((uint32_t)g.RW); a.fd_tw = make_fastdiv((uint32_t)g.TW); a.fd_tx = make_fastdiv((uint32_t)g.tiles_x);
  a.fd_tpi = make_fastdiv((uint32_t)(g.tiles_x * g.tiles_y));
  a.per_xcd = cdiv(a.total_tiles, 8);
  const int grid = 8 * a.per_xcd;
  hipStream_t st = (hipStream_t)s;
  static unsigned long long* dbgbuf = nullptr;            // MGDT_CSP_DBG=1: per-workgroup phase stamps, printed after the launch (debug only)
  static size_t dbgcap = 0;
  if (getenv("MGDT_CSP_DBG") && dbgcap < (size_t)a.total_tiles * 8) {
    if (dbgbuf) (void)hipFree(dbgbuf);
    dbgcap = (size_t)a.total_tiles * 8;
    (void)hipMalloc((void**)&dbgbuf, dbgcap * 8);
  }
  a.dbg = getenv("MGDT_CSP_DBG") ? dbgbuf : nullptr;
#define CSP_LAUNCH(WDV, MODEV)                                                                                              \
  do {                                                                                                                      \
    static std::atomic<bool> attr{false};                                                                                            \
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
  if (a.dbg) {
    std::vector<unsigned long long> h((size_t)a.total_tiles * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    double ph[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < a.total_tiles; ++i) {
      t0 = std::min(t0, h[(size_t)i * 8]); t1 = std::max(t1, h[(size_t)i * 8 + 6]);
      for (int k = 0; k < 6; ++k) ph[k] += (double)(h[(size_t)i * 8 + k + 1] - h[(size_t)i * 8 + k]);
    }
    const double nw = a.total_tiles;
    fprintf(stderr, "csp_block mode %d wd %d n %d %dx%d tile %dx%d (%d wgs, lds %zu): span %.1f us; avg per WG (us): tables %.2f front %.2f sync %.2f conv0 %.2f "
                    "other convs %.2f back %.2f\n", mode, wd, nbtl, a.H, a.W, g.TH, g.TW, a.total_tiles, g.lds, (t1 - t0) * 0.01, ph[0] / nw * 0.01,
            ph[1] / nw * 0.01, ph[2] / nw * 0.01, ph[3] / nw * 0.01, ph[4] / nw * 0.01, ph[5] / nw * 0.01);
  }
  return MGDT_OK;
}
