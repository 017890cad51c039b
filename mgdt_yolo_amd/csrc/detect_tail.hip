// Detect head tail in one launch (bf16 inference): the two final 1x1 convolutions with bias (box: c2 -> 4*reg_max, cls: c3 -> nc), the
// raw (B, no, H, W) map the head returns, and the decode - DFL softmax-expectation, dist2bbox, stride, sigmoid - into y (B, 4+nc, A).
// Reference: nn/modules/head.py:150-177 (cv2[i][2], cv3[i][2], torch.cat, DFL, dist2bbox), nn/modules/block.py:36-54, yolo/utils/tal.py:491-500.
// Unfused this was three launches (11 + 26 + 45 us at B=32, 80x80) that wrote the 39 MB map and read it back for the decode.
//
// A wave takes 32 consecutive anchors of one image (two 16-pixel MFMA groups).  Activations go global -> VGPR in fragment order (a lane
// loads the 16 bytes it feeds to the MFMA); both weight panels (mgdt_conv_pack layout) sit in LDS for the whole persistent workgroup.
// reg_max = 4 makes the box epilogue lane-local: lane (r, g) of the 16-output box tile holds exactly the four DFL bins of side g of
// anchor r, so the softmax-expectation needs no cross-lane traffic; the four sides meet through three shuffles for dist2bbox.  Decoded
// values are staged per wave as a [4+nc][32] fp32 tile in LDS and leave as 128-byte runs along the anchor axis of y.
// The decode consumes the bf16-ROUNDED logits (what the separate decode kernel read back from the map), so both paths agree.
#include "conv_igemm_kernel.h"

struct DtArgs {
  const char* tb; int bsn, bsh, bsw; uint32_t tb_bytes;
  const char* tc; int csn, csh, csw; uint32_t tc_bytes;
  char* feat; int fsn, fsh, fsw; uint32_t feat_bytes;
  const char* wb; const float* bb; const char* wc; const float* bc;
  float* y;
  unsigned long long* best;            // optional [N][a_total]: NMS key of every anchor's best class
  const char* wb3; const float* bb3;   // optional: the box branch's SECOND 3x3 Conv+BN+SiLU (c2 -> c2 = 16) evaluated here; tb is then ITS input
  int N, H, W, HW, c2, c3, nc, kch, nbc, units_per_img, units, a_off, a_total;
  float stride;
  FastDiv fd_w;
};

constexpr int DT_THREADS = 256;
constexpr int DT_P = 20;                // row pitch (floats) of a wave's 16-anchor staging tiles: 16-byte aligned rows
constexpr int DT_WT = (2 * 16 + 4) * DT_P;   // floats of LDS per wave

__global__ __launch_bounds__(DT_THREADS) void detect_tail_kernel(const DtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wbl = smem;                                       // box panel: 1 chunk x 1 block
  char* wcl = smem + 1024;                                // cls panel: kch x nbc blocks of 1 KiB
  float* biasl = (float*)(wcl + (size_t)a.kch * a.nbc * 1024);       // [16] box | [nbc*16] cls
  float* ytile = biasl + 16 + a.nbc * 16;                 // [4 waves][DT_WT floats]: per wave two 16 x DT_P class tiles (ping-pong) + one 4 x DT_P box tile
  char* wb3l = (char*)(ytile + (size_t)4 * DT_WT);      // box 3x3 panel: 5 K chunks x 1 block of 1 KiB (+ its bias[16])
  float* bias3l = (float*)(wb3l + 5 * 1024);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  for (int i = tid; i < 64; i += DT_THREADS) ((uint4*)wbl)[i] = ((const uint4*)a.wb)[i];
  for (int i = tid; i < a.kch * a.nbc * 64; i += DT_THREADS) ((uint4*)wcl)[i] = ((const uint4*)a.wc)[i];
  for (int i = tid; i < 16 + a.nbc * 16; i += DT_THREADS) biasl[i] = i < 16 ? a.bb[i] : a.bc[i - 16];
  if (a.wb3) {
    for (int i = tid; i < 5 * 64; i += DT_THREADS) ((uint4*)wb3l)[i] = ((const uint4*)a.wb3)[i];
    if (tid < 16) bias3l[tid] = a.bb3[tid];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.tb, 0, a.tb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)a.tc, 0, a.tc_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, a.feat_bytes, 0x00020000);
  float* yt = ytile + (size_t)wave * DT_WT;               // [0, 16 P) and [16 P, 32 P): class tiles, [32 P, 36 P): box tile
  const char* wlane_b = wbl + lane * 16;
  const char* wlane_c = wcl + lane * 16;
  const int boff = (8 * g < a.c2) ? g * 16 : MGDT_OOB;    // box input: one K chunk, pieces past c2 are zero
  const int r4 = 16;                                      // 4 * reg_max

  // A wave's results leave through small LDS tiles as 16-byte runs along the anchor axis of y (round 3: one 16 classes x 16 anchors tile per MFMA block instead
  // of the whole [4+nc][32] tile: 9 KB of LDS per workgroup instead of 44, four workgroups per CU instead of two); the best class of an anchor is kept as a
  // running (score, class) pair per lane over its own classes and combined over the four lanes of the anchor at the end (first maximal class, ops.py:225-226).
  const bool vec_ok = (a.HW & 3) == 0 && (a.a_off & 3) == 0 && (a.a_total & 3) == 0 && ((uintptr_t)a.y & 15) == 0;
  auto wave_sync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // tile rows [row0, row0 + nrow) of 16 anchors starting at anchor an0 of image n -> y (lane = (row, 4 anchors))
  auto flush = [&](const float* tile, int nrow, int row0, int n, int an0) __attribute__((always_inline)) {
    const int row = lane >> 2, q = (lane & 3) * 4;
    if (row < nrow && row0 + row < 4 + a.nc) {
      float* dst = a.y + ((size_t)n * (4 + a.nc) + row0 + row) * a.a_total + a.a_off + an0 + q;
      const float* src = tile + row * DT_P + q;
      if (vec_ok) { if (an0 + q < a.HW) *(f32x4*)dst = *(const f32x4*)src; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (an0 + q + e < a.HW) dst[e] = src[e];
      }
    }
  };
  int flip = 0;
  for (int unit = blockIdx.x * 4 + wave; unit < a.units; unit += gridDim.x * 4) {
    const int n = unit / a.units_per_img, a0 = (unit - n * a.units_per_img) * 32;
    float bs[2] = {-1.f, -1.f};
    int bcl[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int an = a0 + u * 16 + r;
      const bool pv = an < a.HW;
      const int oy = (int)fdiv((uint32_t)(pv ? an : 0), a.fd_w), ox = (pv ? an : 0) - oy * a.W;
      const int bo = pv ? n * a.bsn + oy * a.bsh + ox * a.bsw : MGDT_OOB;
      const int co = pv ? n * a.csn + oy * a.csh + ox * a.csw : MGDT_OOB;
      const int fo = pv ? n * a.fsn + oy * a.fsh + ox * a.fsw : MGDT_OOB;
      // ---- box branch: 16 outputs = 4 sides x 4 bins; lane (r, g): bins of side g
      bf16x8 Bb;
      if (a.wb3) {
        // the box branch's second conv (3x3, 16 -> 16, BN + SiLU: head.py:150 cv2[i][1]) as an implicit GEMM over tb = its input: K = 9 taps x 2
        // pieces = 18 pieces in 5 chunks; the result (rounded to bf16 like the stored map) is the B operand of the final 1x1 (K slots e < 4)
        bf16x8 B3[5];
#pragma unroll
        for (int kc = 0; kc < 5; ++kc) {
          const int p = kc * 4 + g, tap = p >> 1, cp = p & 1;
          const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
          const bool ok = pv && tap < 9 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          B3[kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, ok ? (uint32_t)(n * a.bsn + iy * a.bsh + ix * a.bsw + cp * 16) : (uint32_t)MGDT_OOB, 0, 0));
        }
        f32x4 acc3 = *(const f32x4*)(bias3l + 4 * g);
#pragma unroll
        for (int kc = 0; kc < 5; ++kc) acc3 = mma(*(const bf16x8*)(wb3l + kc * 1024 + lane * 16), B3[kc], acc3);
#pragma unroll
        for (int e = 0; e < 8; ++e) Bb[e] = e < 4 ? (bf16)(acc3[e] * fast_sigmoid(acc3[e])) : (bf16)0.f;
      } else {
        Bb = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, (bo | boff) < 0 ? (uint32_t)MGDT_OOB : (uint32_t)(bo + boff), 0, 0));
      }
      bf16x8 Bc[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int piece = kc * 4 + g;
        const bool ok = kc < a.kch && piece * 8 < a.c3;
        Bc[kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(crs, (ok && co >= 0) ? (uint32_t)(co + piece * 16) : (uint32_t)MGDT_OOB, 0, 0));
      }
      f32x4 accb = *(const f32x4*)(biasl + 4 * g);
      accb = mma(*(const bf16x8*)wlane_b, Bb, accb);
      float lb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) lb[j] = (float)(bf16)accb[j];                // the stored (rounded) logits are what gets decoded
      bstore4<bf16>(frs, (uint32_t)fo + (uint32_t)(4 * g * 2), f32x4{lb[0], lb[1], lb[2], lb[3]});
      // DFL (block.py:36-54): softmax over the 4 bins, expectation with weights 0..3
      const float mx = fmaxf(fmaxf(lb[0], lb[1]), fmaxf(lb[2], lb[3]));
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = __builtin_amdgcn_exp2f((lb[k] - mx) * 1.4426950408889634f);     // hardware exp2 / rcp (~1 ulp each): libm's expf and the IEEE division
        den += e;                                                                       // were half of this kernel's VALU instructions
        num += e * (float)k;
      }
      const float dd = num * __builtin_amdgcn_rcpf(den);
      const float dl = __shfl(dd, r, 64), dt = __shfl(dd, 16 + r, 64), dr = __shfl(dd, 32 + r, 64), db = __shfl(dd, 48 + r, 64);
      const float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;            // make_anchors offset 0.5 (tal.py:476-488)
      const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;       // dist2bbox (tal.py:491-500), then * stride (head.py:176)
      const float comp = g == 0 ? (x1 + x2) / 2.f * a.stride : g == 1 ? (y1 + y2) / 2.f * a.stride : g == 2 ? (x2 - x1) * a.stride : (y2 - y1) * a.stride;
      {
        float* bt = yt + 32 * DT_P;
        bt[g * DT_P + r] = comp;
        wave_sync();
        flush(bt, 4, 0, n, a0 + u * 16);
      }
      // ---- class branch
      for (int nb = 0; nb < a.nbc; ++nb) {
        f32x4 acc = *(const f32x4*)(biasl + 16 + nb * 16 + 4 * g);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
          if (kc < a.kch) acc = mma(*(const bf16x8*)(wlane_c + (size_t)(kc * a.nbc + nb) * 1024), Bc[kc], acc);
        const int c = nb * 16 + 4 * g;
        float lc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lc[j] = (float)(bf16)acc[j];
        bstore4<bf16>(frs, c < a.nc ? (uint32_t)fo + (uint32_t)((r4 + c) * 2) : (uint32_t)MGDT_OOB, f32x4{lc[0], lc[1], lc[2], lc[3]});
        float* ct = yt + flip * 16 * DT_P;               // ping-pong: the previous block's tile may still be on its way out of the other half
        flip ^= 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sg = fast_sigmoid(lc[j]);
          ct[(4 * g + j) * DT_P + r] = sg;
          if (c + j < a.nc && sg > bs[u]) { bs[u] = sg; bcl[u] = c + j; }       // classes come in ascending order: the first maximal one stays
        }
        wave_sync();
        flush(ct, 16, 4 + nb * 16, n, a0 + u * 16);
      }
    }
    if (a.best) {
      // best class of an anchor = the best of its four lanes (r, g = 0..3): higher score, then lower class; lane (r, 0) writes the NMS key
      // ((~score bits) << 32 | anchor * nc + class), so that mgdt_nms_fwd does not have to scan the 80 score rows of y again
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
          const float os = __shfl_xor(bs[u], off, 64);
          const int oc = __shfl_xor(bcl[u], off, 64);
          if (os > bs[u] || (os == bs[u] && oc < bcl[u])) { bs[u] = os; bcl[u] = oc; }
        }
        if (g == 0 && a0 + u * 16 + r < a.HW) {
          const unsigned an = (unsigned)(a.a_off + a0 + u * 16 + r);
          a.best[(size_t)n * a.a_total + an] = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(bs[u])) << 32) | (unsigned long long)(an * (unsigned)a.nc + (unsigned)bcl[u]);
        }
      }
    }
    wave_sync();                                           // the last tiles have been read before the next unit writes them
  }
}

/* tb: N x H x W x c2 box-branch input, tc: N x H x W x c3 class-branch input (bf16 NHWC views); wb/bb = mgdt_conv_pack(c2, 16, 1, bf16) of
 * cv2[i][2] (4*reg_max = 16 outputs: reg_max must be 4), wc/bc = mgdt_conv_pack(c3, nc, 1, bf16) of cv3[i][2]; feat: N x H x W x (16+nc) raw
 * head map (written); y: fp32 [N][4+nc][a_total], this level's anchors at a_off.  Covered: c2 <= 32, c3 <= 128, nc <= 256; returns 1 from
 * mgdt_detect_tail_supported when so. */
extern "C" int mgdt_detect_tail_supported(int c2, int c3, int nc, int reg_max, int dtype) {
  return dtype == MGDT_BF16 && reg_max == 4 && c2 % 8 == 0 && c2 <= 32 && c3 % 8 == 0 && c3 <= 128 && nc >= 4 && nc <= 256 && nc % 4 == 0;   // nc % 4: 8-byte rows of the raw map
}

extern "C" int mgdt_detect_tail_fwd(const mgdt_view* tb, const mgdt_view* tc, const void* wb, const float* bb, const void* wc, const float* bc, int nc,
                                    float stride, int a_off, int a_total, const mgdt_view* feat, float* y, unsigned long long* best_keys,
                                    const void* wb3, const float* bb3, mgdt_stream s) {
  if (!view_ok(tb) || !view_ok(tc) || !view_ok(feat) || !wb || !bb || !wc || !bc || !y) MGDT_FAIL(MGDT_BAD_ARG, "detect_tail: null/empty argument");
  if (!mgdt_detect_tail_supported(tb->c, tc->c, nc, 4, MGDT_BF16)) MGDT_FAIL(MGDT_BAD_SHAPE, "detect_tail: c2=%d c3=%d nc=%d not covered", tb->c, tc->c, nc);
  if (feat->c != 16 + nc || tb->n != tc->n || tb->h != tc->h || tb->w != tc->w || feat->n != tb->n || feat->h != tb->h || feat->w != tb->w ||
      a_off < 0 || a_off + tb->h * tb->w > a_total)
    MGDT_FAIL(MGDT_BAD_SHAPE, "detect_tail: shapes");
  DtArgs a;
  memset(&a, 0, sizeof(a));
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes, int q) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * 2;
    if (v->sc != 1 || v->sw % q || v->sh % q || v->sn % q || (uintptr_t)v->p % (q * 2) || ext >= 0x7fffffffL) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * 2); *sh = (int)(v->sh * 2); *sw = (int)(v->sw * 2); *bytes = (uint32_t)ext;
  };
  const char* fp = nullptr;
  bind(tb, &a.tb, &a.bsn, &a.bsh, &a.bsw, &a.tb_bytes, 8);
  bind(tc, &a.tc, &a.csn, &a.csh, &a.csw, &a.tc_bytes, 8);
  bind(feat, &fp, &a.fsn, &a.fsh, &a.fsw, &a.feat_bytes, 4);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "detect_tail: views must be 16-byte aligned NHWC (sc == 1) below 2 GiB");
  a.feat = (char*)fp;
  a.wb = (const char*)wb; a.bb = bb; a.wc = (const char*)wc; a.bc = bc; a.y = y; a.best = best_keys;
  a.wb3 = (const char*)wb3; a.bb3 = bb3;
  if (wb3 && (!bb3 || tb->c != 16)) MGDT_FAIL(MGDT_BAD_SHAPE, "detect_tail: the in-launch 3x3 box conv is built for 16 -> 16 channels");
  a.N = tb->n; a.H = tb->h; a.W = tb->w; a.HW = tb->h * tb->w; a.c2 = tb->c; a.c3 = tc->c; a.nc = nc;
  a.kch = cdiv(tc->c, 32); a.nbc = cdiv(nc, 16);
  a.units_per_img = cdiv(a.HW, 32); a.units = a.N * a.units_per_img;
  a.a_off = a_off; a.a_total = a_total; a.stride = stride; a.fd_w = make_fastdiv((uint32_t)tb->w);
  const size_t lds = 1024 + (size_t)a.kch * a.nbc * 1024 + (size_t)(16 + a.nbc * 16) * 4 + (size_t)4 * DT_WT * 4 + 5 * 1024 + 64;
  if (lds > 150 * 1024) MGDT_FAIL(MGDT_BAD_SHAPE, "detect_tail: nc=%d needs %zu B of LDS", nc, lds);
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)detect_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "detect_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = 150 * 1024;
  }
  const int grid = std::min(cdiv(a.units, 4), 1024);
  detect_tail_kernel<<<grid, DT_THREADS, lds, (hipStream_t)s>>>(a);
  MGDT_CHECK_LAUNCH("detect_tail_fwd");
  return MGDT_OK;
}
