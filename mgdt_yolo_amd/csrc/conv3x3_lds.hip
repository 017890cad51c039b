// 3x3 stride-1 convolution (bf16) with the ACTIVATIONS staged in LDS - for the compute-heavy layers (64-128 input channels on 80x80 / 40x40 maps: the
// Detect branches) where conv_igemm_kernel, which feeds the MFMA B operand straight from global memory, re-requests every input pixel nine times
// (once per tap) through the texture path and ends up address-bound at 330-430 TFLOP/s.
// Reference: Conv.forward_fuse (nn/modules/conv.py:40-42) = act(conv(x)) with BatchNorm folded into the packed weights (mgdt_conv_pack).
//
//   * persistent workgroups (one per CU, 4 waves): a workgroup owns NBW cout blocks; their weight panel (the same fragment-ordered layout conv_igemm uses,
//     1 KB per [K chunk][cout block]) is loaded into LDS ONCE and stays there for all the pixel tiles the workgroup walks;
//   * a tile = 16x16 output pixels; its 18x18 input region (zero outside the image) sits in LDS as [pixel][cin] rows with a 16-byte pad, so the B fragment of
//     lane (r, g) - 8 consecutive channels of pixel r for the tap the K piece belongs to - is one ds_read_b128 at pixel-row address + a per-piece constant
//     (table in LDS); a tap costs nothing but that constant;
//   * wave w owns tile rows 4w .. 4w+3 (MT = 4 pixel groups of 16) and all NBW cout blocks: NBW + 4 fragment reads feed 4*NBW MFMAs per K step; fragments are
//     double-buffered in registers so the LDS reads of step k+1 are in flight under the MFMAs of step k;
//   * the next tile's region is requested into registers before the current tile's MFMAs and written to LDS after them.
#include <type_traits>
#include <vector>

#include "common.h"

#define C3_OOB ((int)0x80000000)
#define C3_TW 16
#define C3_RW 18
#define C3_MAXI 14                     // 16-byte items a thread stages per tile: 324 * (Cin / 8) / 256, Cin <= 80 -> 12.7  (stride 2: 18 = 561 * 8 / 256 for Cin = 64)

struct C3Args {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const char* wpk; const float* bias;
  int N, H, W, Ho, Wo, Cin, Cout, CP, nchunks, NTtot, ncg, tiles_x, tiles_per_img, ntiles, XP, act;
  FastDiv fd_tpi, fd_tx, fd_cp;
  const float* oscale; float xq; // Q8 kernels (fp8 inference): de-quantisation factor per output channel, activation multiplier (mgdt_conv_pack_fp8)
  unsigned long long* dbg;       // MGDT_C3_DBG: per workgroup {start, weights staged, sum(commit), sum(mfma), sum(epilogue), tiles} in 10 ns ticks
};

template <int ACT> __device__ __forceinline__ float c3_act(float v) {
  if (ACT == MGDT_ACT_SILU) return v * fast_sigmoid(v);
  if (ACT == MGDT_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// NCH > 0: the K loop is fully unrolled (no back edge: the compiler can keep the next step's LDS reads in flight under this step's MFMAs; with a run-time
// trip count it waited for every fragment right before its first use and the single wave per SIMD had nothing to hide that latency behind)
// MT = pixel groups (tile rows) per wave: the tile is 4*MT rows x 16 columns (16x16 at MT = 4; 8x16 at MT = 2, for panels of 5 cout blocks x 80 input
// channels that leave no room for the 18x18 region)
// Q8 (BASELINE configs[4]): e4m3 operands - the weight panel holds 512-byte blocks (mgdt_conv_pack_fp8), the region is converted to e4m3 when it is committed
// to LDS (8 bytes per piece), fragments are ds_read_b64.  Half the LDS bytes and footprint: 80 -> 80 layers fit with 16x16 tiles (bf16: igemm kernel).  Measured
// (B = 32, 80x80): 64 -> 96 48 us (bf16 form 48), 80 -> 80 44 us (bf16 igemm 49); MFMA phase 5.2 us per tile for 460 MFMAs per wave vs 5.5 us for 432 in bf16 - the
// phase is not LDS-bandwidth bound, so halving the bytes buys little.
// S = 2 (round 2): the stride-2 down-sampling convolutions (32 -> 64 at 320 -> 160 .. 64 -> 128 at 80 -> 40).  Same structure; the region of a TH x 16 output
// tile is (2 TH + 1) x 33 input pixels, lane r's pixel sits 2 r columns into its region row, and the cout blocks may be split over `ncg` workgroup groups
// (each stages the region itself: 2 re-reads instead of the igemm kernel's 9 taps x cout groups through the texture path).
// WS = 2 (round 3): eight waves - waves 0-3 and 4-7 take the same pixel rows and each HALF of the cout blocks, so two waves per SIMD share the panel and the
// region: one's LDS reads and epilogue run under the other's MFMAs (a single wave per SIMD had nothing to hide them behind), registers per wave halve
template <int NBW, int ACT, int NCH, int MT, bool Q8 = false, int S = 1, int WS = 1>
__global__ __launch_bounds__(256 * WS, 1) void conv3x3_lds_kernel(const C3Args a) {
  constexpr int NTHR = 256 * WS, NBH = (NBW + WS - 1) / WS;                                // threads, cout blocks per wave (an odd count: the second group's last block is
                                                                                            // a phantom - it multiplies whatever follows the panel and is never stored)
  constexpr int RW = 15 * S + 3, TH = 4 * MT, RH = (TH - 1) * S + 3, RPX = RH * RW;     // region width, tile rows, region rows, region pixels
  constexpr int MAXI = ((S == 1 ? C3_MAXI : 18) + WS - 1) / WS;
  constexpr int WB = Q8 ? 512 : 1024, PB = Q8 ? 8 : 16;         // bytes of a weight block / of an 8-channel piece in LDS
  extern __shared__ __attribute__((aligned(16))) char c3_lds[];
  int* tab = (int*)c3_lds;                                          // [nchunks * 4] byte offset of piece p inside the region, relative to the pixel's row
  char* wl = c3_lds + (((size_t)a.nchunks * 16 + 15) & ~(size_t)15);
  char* xs = wl + (size_t)a.nchunks * NBW * WB;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all & 3, wcol = wave_all >> 2;                 // pixel rows of the tile, cout half
  const int cg = blockIdx.x % a.ncg, w0 = blockIdx.x / a.ncg, wstep = gridDim.x / a.ncg;      // cout group, first tile, tile step of this workgroup
  const int nb0 = cg * NBW;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);

  // ---- once per workgroup: piece table and the weight panel of its cout blocks
  for (int p = tid; p < a.nchunks * 4; p += NTHR) {
    const int tap = (int)fdiv((uint32_t)p, a.fd_cp), cp = p - tap * a.CP;
    tab[p] = tap < 9 ? ((tap / 3) * RW + (tap % 3)) * a.XP + cp * PB : 0;       // padding pieces carry zero weights: any in-range address
  }
  constexpr int WV = WB / 16;                                       // 16-byte vectors per weight block
  {
    // eight requests of a thread in flight before its first LDS store (the plain copy loop waited for every load in turn)
    const int total = a.nchunks * NBW * WV;
    for (int i0 = tid; i0 < total; i0 += 8 * NTHR) {
      uint4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u * NTHR, total - 1);
        const int kc = i / (NBW * WV), rem = i - kc * NBW * WV, bw = rem / WV, ln = rem % WV;
        const bool ok = nb0 + bw < a.NTtot;
        t[u] = ok ? *(const uint4*)(a.wpk + ((size_t)(kc * a.NTtot + nb0 + bw) * WV + ln) * 16) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * NTHR < total) *(uint4*)(wl + (size_t)(i0 + u * NTHR) * 16) = t[u];
    }
  }
  const int nbw0 = nb0 + wcol * NBH;                                 // this wave's first cout block
  f32x4 bias[NBH], osc[Q8 ? NBH : 1];
#pragma unroll
  for (int bw = 0; bw < NBH; ++bw) {
    bias[bw] = nbw0 + bw < a.NTtot ? *(const f32x4*)(a.bias + (nbw0 + bw) * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (Q8) {                                              // accumulators start from bias / oscale, the epilogue multiplies by oscale
      osc[bw] = nbw0 + bw < a.NTtot ? *(const f32x4*)(a.oscale + (nbw0 + bw) * 16 + 4 * g) : f32x4{1.f, 1.f, 1.f, 1.f};
      bias[bw] = bias[bw] / osc[bw];
    }
  }

  // ---- staging of one tile's input region: item = (region pixel, 8-channel piece)
  const int nitems = RPX * a.CP;
  uint4 stage[MAXI];
  auto issue = [&](int t) {
    const int n = (int)fdiv((uint32_t)t, a.fd_tpi), rt = t - n * a.tiles_per_img;
    const int tyi = (int)fdiv((uint32_t)rt, a.fd_tx), txi = rt - tyi * a.tiles_x;
    const int iy0 = tyi * TH * S - 1, ix0 = txi * C3_TW * S - 1;
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      const int it = tid + u * NTHR;
      const int pix = (int)fdiv((uint32_t)it, a.fd_cp), c8 = it - pix * a.CP;
      const int py = pix / RW, px = pix - py * RW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = it < nitems && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      stage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (uint32_t)(n * a.xsn + iy * a.xsh + ix * a.xsw + c8 * 16) : (uint32_t)C3_OOB, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      const int it = tid + u * NTHR;
      if (it < nitems) {
        const int pix = (int)fdiv((uint32_t)it, a.fd_cp), c8 = it - pix * a.CP;
        if constexpr (Q8) *(long*)(xs + pix * a.XP + c8 * 8) = quant8(__builtin_bit_cast(bf16x8, stage[u]), a.xq);
        else *(uint4*)(xs + pix * a.XP + c8 * 16) = stage[u];
      }
    }
  };

  const char* const wlane = wl + lane * PB;
  int bbase[MT];                                                     // region row address of this lane's pixel in the wave's pixel groups
#pragma unroll
  for (int m = 0; m < MT; ++m) bbase[m] = ((wave * MT + m) * S * RW + r * S) * a.XP;

  // every global load issued so far (bias, weights) is retired HERE: otherwise the compiler, unable to order them against the prefetches that are
  // pending at the loop's back edge, waits for vmcnt(0) - i.e. for the NEXT tile's prefetch - in front of the first MFMA of every tile
  __builtin_amdgcn_s_waitcnt(0x0F70);
  unsigned long long T0 = 0, Tc = 0, Tm = 0, Te = 0, tl = 0, ntl = 0;
  if (a.dbg) { T0 = __builtin_amdgcn_s_memrealtime(); tl = T0; }
  int t = w0;
  if (t < a.ntiles) issue(t);
  while (t < a.ntiles) {
    __syncthreads();                                                 // the previous tile's fragments have been read (first pass: table + weights written)
    commit();
    __syncthreads();
    if (a.dbg) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); Tc += n_ - tl; tl = n_; }
    const int tn = t + wstep;
    if (tn < a.ntiles) issue(tn);                                    // in flight under this tile's MFMAs

    f32x4 acc[NBH][MT];
#pragma unroll
    for (int bw = 0; bw < NBH; ++bw)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[bw][m] = bias[bw];
    typedef typename std::conditional<Q8, long, bf16x8>::type frag_t;
    frag_t A[2][NBH], B[2][MT];
    auto load_frags = [&](int kc, int buf) {
      const int off = tab[kc * 4 + g];
#pragma unroll
      for (int bw = 0; bw < NBH; ++bw) A[buf][bw] = *(const frag_t*)(wlane + (size_t)(kc * NBW + wcol * NBH + bw) * WB);
#pragma unroll
      for (int m = 0; m < MT; ++m) B[buf][m] = *(const frag_t*)(xs + bbase[m] + off);
    };
    auto mm = [&](frag_t w, frag_t p, f32x4 c) __attribute__((always_inline)) {
      if constexpr (Q8) return mma_q8(w, p, c);
      else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, p, c, 0, 0, 0);
    };
    load_frags(0, 0);
    if (NCH > 0) {
#pragma unroll
      for (int kc = 0; kc < NCH; ++kc) {
        if (kc + 1 < NCH) load_frags(kc + 1, (kc + 1) & 1);
#pragma unroll
        for (int bw = 0; bw < NBH; ++bw)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[bw][m] = mm(A[kc & 1][bw], B[kc & 1][m], acc[bw][m]);
      }
    } else {
      for (int kc = 0; kc < a.nchunks; kc += 2) {
        if (kc + 1 < a.nchunks) load_frags(kc + 1, 1);
#pragma unroll
        for (int bw = 0; bw < NBH; ++bw)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[bw][m] = mm(A[0][bw], B[0][m], acc[bw][m]);
        if (kc + 1 < a.nchunks) {
          if (kc + 2 < a.nchunks) load_frags(kc + 2, 0);
#pragma unroll
          for (int bw = 0; bw < NBH; ++bw)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[bw][m] = mm(A[1][bw], B[1][m], acc[bw][m]);
        }
      }
    }

    if (a.dbg) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); Tm += n_ - tl; tl = n_; }
    // epilogue: lane (r, g) holds couts 4g .. 4g+3 of pixel r: one 8-byte store per (cout block, pixel group)
    {
      const int n = (int)fdiv((uint32_t)t, a.fd_tpi), rt = t - n * a.tiles_per_img;
      const int tyi = (int)fdiv((uint32_t)rt, a.fd_tx), txi = rt - tyi * a.tiles_x;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = tyi * TH + wave * MT + m, ox = txi * C3_TW + r;
        const bool pin = oy < a.Ho && ox < a.Wo;
        const int po = n * a.ysn + oy * a.ysh + ox * a.ysw;
#pragma unroll
        for (int bw = 0; bw < NBH; ++bw) {
          const int co = (nbw0 + bw) * 16 + 4 * g;
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)c3_act<ACT>(Q8 ? acc[bw][m][j] * osc[Q8 ? bw : 0][j] : acc[bw][m][j]);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, o), yrs,
                                                (pin && co < a.Cout) ? (uint32_t)(po + co * 2) : (uint32_t)C3_OOB, 0, 0);
        }
      }
    }
    if (a.dbg) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); Te += n_ - tl; tl = n_; ++ntl; }
    t = tn;
  }
  if (a.dbg && tid == 0) { unsigned long long* d = a.dbg + (size_t)blockIdx.x * 6; d[0] = T0; d[1] = tl; d[2] = Tc; d[3] = Tm; d[4] = Te; d[5] = ntl; }
}

// true when the layer is launched here (the caller returns), false: conv_igemm takes it
bool mgdt_conv3x3_lds_launch(const mgdt_view* x, const mgdt_view* y, const void* packed_w, const float* bias, int act, int CP, int nchunks, int NTtot, hipStream_t st,
                             const float* q8_oscale, float q8_xq, int stride) {
  const bool q8 = q8_oscale != nullptr;                    // e4m3 panel from mgdt_conv_pack_fp8
  static const int mode = getenv("MGDT_CONV3_LDS") ? atoi(getenv("MGDT_CONV3_LDS")) : 1;      // experiment knob: 0 = never
  if (!mode) return false;
  static const bool ws2 = !(getenv("MGDT_C3_WAVES8") && atoi(getenv("MGDT_C3_WAVES8")) == 0);   // experiment knob: 0 = the four-wave form everywhere
  const bool wide = mode >= 3 || ws2;     // the eight-wave forms also win where the four-wave ones lost to the igemm kernel: 64 -> 128 stride 2 (26.9 vs 30.6 us) and 80 -> 80 on 12x16 tiles (48.2 vs 50.9 us)
  const int Cin = x->c, Cout = y->c;
  // stride 2 - instantiated: bf16, 4 cout blocks per workgroup, Cin = 32 / 64.  Measured (B = 32, bench step): 32 -> 64 at 160 -> 80 35.9 us (igemm 38-40): default;
  // 64 -> 128 at 80 -> 40 35.4 us (igemm 34.9): only with MGDT_CONV3_LDS=3.  The 8x16-output tiles carry 72 / 144 MFMAs per wave, so a tile costs mostly its
  // staging latency and two barriers with one workgroup per CU - the LDS route wins much less here than the byte counts suggest.
  if (stride == 2 && (q8_oscale || Cin > 64 || NTtot % 4 || (nchunks != 9 && !(nchunks == 18 && wide)))) return false;
  if (Cin % 8 || Cin < 32 || Cin > 80 || Cout % 4 || Cout < 32 || (act != MGDT_ACT_SILU && act != MGDT_ACT_NONE && act != MGDT_ACT_RELU)) return false;
  const long M = (long)x->n * x->h * x->w;
  if (M < 16 * 1024 || x->h < 16 || x->w < 16) return false;                                    // small maps: the igemm kernel's finer tiles fill the chip better
  const long extx = ((long)(x->n - 1) * x->sn + (long)(x->h - 1) * x->sh + (long)(x->w - 1) * x->sw + x->c) * 2;
  const long exty = ((long)(y->n - 1) * y->sn + (long)(y->h - 1) * y->sh + (long)(y->w - 1) * y->sw + y->c) * 2;
  if (extx >= 0x7fffffffL || exty >= 0x7fffffffL) return false;
  C3Args a;
  memset(&a, 0, sizeof(a));
  a.x = (const char*)x->p; a.xsn = (int)(x->sn * 2); a.xsh = (int)(x->sh * 2); a.xsw = (int)(x->sw * 2); a.x_bytes = (uint32_t)extx;
  a.y = (char*)y->p; a.ysn = (int)(y->sn * 2); a.ysh = (int)(y->sh * 2); a.ysw = (int)(y->sw * 2); a.y_bytes = (uint32_t)exty;
  a.wpk = (const char*)packed_w; a.bias = bias;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Ho = y->h; a.Wo = y->w; a.Cin = Cin; a.Cout = Cout; a.CP = CP; a.nchunks = nchunks; a.NTtot = NTtot; a.act = act;
  a.XP = q8 ? Cin + 16 : Cin * 2 + 16;
  a.oscale = q8_oscale; a.xq = q8_xq;
  // every cout block in ONE workgroup (the input region is then staged once per tile); layers whose whole weight panel does not fit next to the
  // region stay on the igemm kernel: splitting the couts over workgroups re-reads the input per group and measured no faster
  const int NBW = stride == 2 ? 4 : NTtot;
  if (NBW != 2 && NBW != 3 && NBW != 4 && NBW != 5 && NBW != 6) return false;
  // (80 couts in bf16: the 8x16-tile form measured slower than the igemm kernel - 66.9 vs 55 us at 80 -> 80, 80x80, B = 32; the 12x16 form (MGDT_CONV3_LDS=3) 58.9 us; MGDT_CONV3_LDS=2 forces 8x16)
  a.ncg = NTtot / NBW;
  const size_t fixed = (((size_t)nchunks * 16 + 15) & ~(size_t)15) + (size_t)nchunks * NBW * (q8 ? 512 : 1024);
  auto region = [&](int mt) { return stride == 2 ? (size_t)(8 * mt + 1) * 33 * a.XP : (size_t)(4 * mt + 2) * C3_RW * a.XP; };
  int MT = stride == 2 ? 2 : 4;                            // stride 1: 16x16 tiles when the region fits next to the panel, else 12x16 (5 cout blocks), else 8x16; stride 2: 8x16
  if (stride == 1 && fixed + region(MT) > 160 * 1024) MT = NBW == 5 && !q8 && wide && fixed + region(3) <= 160 * 1024 ? 3 : 2;   // 12x16 tiles at 80 -> 80: 58.9 us vs 55 us igemm
  const size_t lds = fixed + region(MT);
  if (lds > 160 * 1024) return false;
  if (NBW == 5 && !q8 && MT == 2 && mode < 2) return false;
  if (stride == 1 && (q8 ? MT != 4 : ((NBW == 5 && MT == 4) || (NBW != 5 && MT != 4)))) return false;   // instantiated: bf16 5 blocks with 12x16 / 8x16 tiles, everything else 16x16
  const int TH = 4 * MT;
  a.tiles_x = cdiv(y->w, C3_TW);
  a.tiles_per_img = a.tiles_x * cdiv(y->h, TH);
  a.ntiles = x->n * a.tiles_per_img;
  if ((stride == 2 ? (2 * TH + 1) * 33 * CP > 256 * 18 : (TH + 2) * C3_RW * CP > 256 * C3_MAXI)) return false;
  a.fd_tpi = make_fastdiv((uint32_t)a.tiles_per_img); a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_cp = make_fastdiv((uint32_t)CP);
  int nwg = 256 / a.ncg * a.ncg;                                                                // one workgroup per CU, a multiple of the cout groups (the e4m3 form's LDS
                                                                                                // footprint would let two share a CU at 64 -> 96, its ~370 registers do not)
  nwg = (int)std::min<long>(nwg, (long)a.ntiles * a.ncg);
  nwg = nwg / a.ncg * a.ncg;
  if (nwg < a.ncg) return false;
 #define C3_LAUNCH_Q8(NB, ACTV)                                                                                      \
  {                                                                                                                  \
    static std::atomic<bool> qattr{false}, qattr18{false}, qattr23{false};                                            \
    if (nchunks == 18) {                                                                                             \
      if (!qattr18) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 18, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); qattr18 = true; } \
      conv3x3_lds_kernel<NB, ACTV, 18, 4, true><<<nwg, 256, lds, st>>>(a);                                           \
    } else if (nchunks == 23) {                                                                                      \
      if (!qattr23) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 23, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); qattr23 = true; } \
      conv3x3_lds_kernel<NB, ACTV, 23, 4, true><<<nwg, 256, lds, st>>>(a);                                           \
    } else {                                                                                                         \
      if (!qattr) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 0, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); qattr = true; } \
      conv3x3_lds_kernel<NB, ACTV, 0, 4, true><<<nwg, 256, lds, st>>>(a);                                            \
    }                                                                                                                \
  }
#define C3_ACT_Q8(NB) \
  if (act == MGDT_ACT_SILU) C3_LAUNCH_Q8(NB, MGDT_ACT_SILU) else if (act == MGDT_ACT_RELU) C3_LAUNCH_Q8(NB, MGDT_ACT_RELU) else C3_LAUNCH_Q8(NB, MGDT_ACT_NONE)
#define C3_LAUNCH(NB, ACTV, MTV, WSV)                                                                                     \
  {                                                                                                                  \
    static std::atomic<bool> attr{false}, attr18{false}, attr23{false};                                                                        \
    if (nchunks == 18) {                                                                                             \
      if (!attr18) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 18, MTV, false, 1, WSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr18 = true; } \
      conv3x3_lds_kernel<NB, ACTV, 18, MTV, false, 1, WSV><<<nwg, 256 * WSV, lds, st>>>(a);                                                    \
    } else if (nchunks == 23) {                                                                                      \
      if (!attr23) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 23, MTV, false, 1, WSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr23 = true; } \
      conv3x3_lds_kernel<NB, ACTV, 23, MTV, false, 1, WSV><<<nwg, 256 * WSV, lds, st>>>(a);                                               \
    } else {                                                                                                         \
      if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<NB, ACTV, 0, MTV, false, 1, WSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
      conv3x3_lds_kernel<NB, ACTV, 0, MTV, false, 1, WSV><<<nwg, 256 * WSV, lds, st>>>(a);                                                     \
    }                                                                                                                \
  }
#define C3_ACT(NB, MTV, WSV) \
  if (act == MGDT_ACT_SILU) C3_LAUNCH(NB, MGDT_ACT_SILU, MTV, WSV) else if (act == MGDT_ACT_RELU) C3_LAUNCH(NB, MGDT_ACT_RELU, MTV, WSV) else C3_LAUNCH(NB, MGDT_ACT_NONE, MTV, WSV)
  static unsigned long long* dbgbuf = nullptr;
  if (getenv("MGDT_C3_DBG") && !dbgbuf) (void)hipMalloc((void**)&dbgbuf, 256 * 6 * 8);   // nwg <= 256
  a.dbg = dbgbuf;
#define C3_LAUNCH_S2(ACTV, NCHV)                                                                                      \
  {                                                                                                                  \
    static std::atomic<bool> sattr{false}, sattr8{false};                                                            \
    if (ws2) {                                                                                                       \
      if (!sattr8) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<4, ACTV, NCHV, 2, false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); sattr8 = true; } \
      conv3x3_lds_kernel<4, ACTV, NCHV, 2, false, 2, 2><<<nwg, 512, lds, st>>>(a);                                   \
    } else {                                                                                                         \
      if (!sattr) { (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<4, ACTV, NCHV, 2, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); sattr = true; } \
      conv3x3_lds_kernel<4, ACTV, NCHV, 2, false, 2><<<nwg, 256, lds, st>>>(a);                                      \
    }                                                                                                                \
  }
#define C3_ACT_S2(NCHV) \
  if (act == MGDT_ACT_SILU) C3_LAUNCH_S2(MGDT_ACT_SILU, NCHV) else if (act == MGDT_ACT_RELU) C3_LAUNCH_S2(MGDT_ACT_RELU, NCHV) else C3_LAUNCH_S2(MGDT_ACT_NONE, NCHV)
  if (stride == 2) {
    if (nchunks == 9) { C3_ACT_S2(9) } else { C3_ACT_S2(18) }
  } else if (q8) {
    if (NBW == 6) { C3_ACT_Q8(6) } else if (NBW == 5) { C3_ACT_Q8(5) } else if (NBW == 4) { C3_ACT_Q8(4) } else if (NBW == 3) { C3_ACT_Q8(3) } else { C3_ACT_Q8(2) }
  } else if (ws2 && (NBW == 6 || NBW == 4 || (NBW == 5 && MT == 3))) {
    // eight waves, two per SIMD, each wave half the cout blocks
    if (NBW == 6) { C3_ACT(6, 4, 2) } else if (NBW == 4) { C3_ACT(4, 4, 2) } else { C3_ACT(5, 3, 2) }
  } else if (NBW == 6) { C3_ACT(6, 4, 1) } else if (NBW == 5 && MT == 3) { C3_ACT(5, 3, 1) } else if (NBW == 5) { C3_ACT(5, 2, 1) } else if (NBW == 4) { C3_ACT(4, 4, 1) } else if (NBW == 3) { C3_ACT(3, 4, 1) } else { C3_ACT(2, 4, 1) }
#undef C3_ACT
#undef C3_ACT_Q8
#undef C3_ACT_S2
#undef C3_LAUNCH_S2
#undef C3_LAUNCH_Q8
#undef C3_LAUNCH
  if (dbgbuf) {
    std::vector<unsigned long long> h(256 * 6);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0; double c = 0, m = 0, e = 0, nt = 0;
    for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 6]); t1 = std::max(t1, h[i * 6 + 1]); c += h[i * 6 + 2]; m += h[i * 6 + 3]; e += h[i * 6 + 4]; nt += h[i * 6 + 5]; }
    fprintf(stderr, "conv3x3_lds cin %d cout %d %dx%d: %d wgs, %d tiles, span %.1f us; per tile (us): wait+commit %.2f mfma %.2f epilogue %.2f\n", Cin, Cout, x->h, x->w, nwg,
            a.ntiles, (t1 - t0) * 0.01, c / nt * 0.01, m / nt * 0.01, e / nt * 0.01);
  }
  return true;
}
