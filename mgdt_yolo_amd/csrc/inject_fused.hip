// InjectionMultiSum_Auto_pool (reference nn/modules/block.py:376-399), up-sampling branch, in ONE launch:
//   y = local_embedding(x) * bilinear(h_sigmoid(ga)) + bilinear(gf)
// local_embedding is a 1x1 conv + BN (no activation); ga / gf are the global act / embedding maps at (Hg, Wg) <= (H, W).
// The 256-channel local map never goes to HBM (the unfused pair writes and re-reads it), and every global-map pixel a patch
// needs is fetched once per workgroup into LDS instead of once per output pixel and tap.
//
// Workgroup = 4 waves = a 4 x 16 patch of output pixels; wave w owns output row w (16 pixels = the MFMA N dimension) and all
// output channels: acc[NB] in the usual D layout (lane (r, g): channels nb*16 + 4g .. +3 of pixel r).  The tail reads the four
// bilinear taps of the lane's pixel from the LDS patch (8-byte reads at [src pixel][channel]), applies h_sigmoid to the gate
// taps BEFORE interpolating (block.py:393), and stores 4 consecutive channels.
#include "conv_igemm_kernel.h"

struct InjArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  const char* ga; const char* gf; int gsn, gsh, gsw;   // equally laid out (host-checked)
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const char* wpk; const float* bias;
  int N, H, W, Cin, Cout, Hg, Wg, PH, PW, tiles_x, tiles_y;
};

constexpr int INJ_TH = 4, INJ_TW = 16, INJ_THREADS = 256, INJ_CPAD = 8;   // channel stride of an LDS patch pixel: Cout + 8 (bank spread)

// F.interpolate(bilinear, align_corners=False): src = max(0, (dst+0.5)*in/out - 0.5), upper neighbour clamped
// (no fma contraction: the host sizes the LDS patch with this same function and must get the same integers as the device)
__device__ __host__ inline void inj_lerp(int o, int isz, int osz, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
  const float scale = (float)isz / (float)osz;
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > isz - 1) i0 = isz - 1;
  i1 = i0 + (i0 < isz - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

template <typename T, int KC, int NB>
__global__ __launch_bounds__(INJ_THREADS) void conv1x1_inject_kernel(const InjArgs a) {
  typedef typename Piece<T>::frag frag;
  constexpr int SZ = (int)sizeof(T);
  constexpr int CS = NB * 16 + INJ_CPAD;                 // elements per LDS patch pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                        // [KC][NB][64][16 B]
  T* sg = (T*)(smem + (size_t)KC * NB * 1024);            // gate patch  [PH*PW][CS]
  T* sf = sg + (size_t)a.PH * a.PW * CS;                  // feat patch
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  for (int i = tid; i < KC * NB * 64; i += INJ_THREADS) ((uint4*)wl)[i] = ((const uint4*)a.wpk)[i];   // weights once per (persistent) workgroup
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const char* const wlane = wl + lane * 16;
  const int npatch = a.N * a.tiles_x * a.tiles_y;
#pragma unroll 1
  for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
  int b = patch;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y, n = b / a.tiles_y;
  const int oy = ty * INJ_TH + wave, ox = tx * INJ_TW + r;
  const bool pv = oy < a.H && ox < a.W;

  // activations of my pixel first (they have the longest way to come)
  const int xo = pv ? n * a.xsn + oy * a.xsh + ox * a.xsw : MGDT_OOB;
  frag P[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int cb = (kc * 4 + g) * 16;
    P[kc] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(xrs, (cb < a.Cin * SZ) ? (uint32_t)(xo + cb) : (uint32_t)MGDT_OOB, 0, 0));
  }
  // source patch of this workgroup: rows/cols touched by its first and last output row/col (uniform)
  int py0, py1, px0, px1, t0, t1; float tl;
  inj_lerp(ty * INJ_TH, a.Hg, a.H, py0, t1, tl);
  inj_lerp(min(ty * INJ_TH + INJ_TH - 1, a.H - 1), a.Hg, a.H, t0, py1, tl);
  inj_lerp(tx * INJ_TW, a.Wg, a.W, px0, t1, tl);
  inj_lerp(min(tx * INJ_TW + INJ_TW - 1, a.W - 1), a.Wg, a.W, t0, px1, tl);
  const int ph = py1 - py0 + 1, pw = px1 - px0 + 1;       // <= PH, PW (host bound)
  __syncthreads();                                         // the previous patch's tail is done with the LDS maps
  {
    constexpr int VPP = NB * 16 / 8;                        // 16-byte vectors per pixel
    const int nvec = ph * pw * VPP;
    for (int i = tid; i < nvec; i += INJ_THREADS) {
      const int v = i % VPP, p = i / VPP;
      const int sy = p / pw, sx = p - sy * pw;
      const long go = (long)n * a.gsn + (long)(py0 + sy) * a.gsh + (long)(px0 + sx) * a.gsw + v * 16;
      const uint4 va = *(const uint4*)(a.ga + go), vf = *(const uint4*)(a.gf + go);
      *(uint4*)((char*)sg + ((size_t)p * CS + v * 8) * SZ) = va;
      *(uint4*)((char*)sf + ((size_t)p * CS + v * 8) * SZ) = vf;
    }
  }
  __syncthreads();

  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = *(const f32x4*)(a.bias + nb * 16 + 4 * g);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mma(*(const frag*)(wlane + (kc * NB + nb) * 1024), P[kc], acc[nb]);

  // bilinear taps of my pixel inside the patch
  int y0, y1, x0, x1; float wy1, wx1;
  inj_lerp(min(oy, a.H - 1), a.Hg, a.H, y0, y1, wy1);
  inj_lerp(min(ox, a.W - 1), a.Wg, a.W, x0, x1, wx1);
  const int o00 = ((y0 - py0) * pw + (x0 - px0)) * CS + 4 * g, o01 = ((y0 - py0) * pw + (x1 - px0)) * CS + 4 * g;
  const int o10 = ((y1 - py0) * pw + (x0 - px0)) * CS + 4 * g, o11 = ((y1 - py0) * pw + (x1 - px0)) * CS + 4 * g;
  const int yo = pv ? n * a.ysn + oy * a.ysh + ox * a.ysw : MGDT_OOB;
  auto ld = [&](const T* base, int off) __attribute__((always_inline)) {
    const bf16x4 v = *(const bf16x4*)(base + off);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  };
  auto hs = [](f32x4 v) __attribute__((always_inline)) {   // h_sigmoid = relu6(v + 3) / 6 as one fma + clamp (an IEEE divide costs ~10 VALU ops
    f32x4 o;                                               // per tap; the result differs by <= 1 ulp, far below bf16 resolution)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(fmaf(v[j], 1.f / 6.f, 0.5f), 0.f), 1.f);
    return o;
  };
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c = nb * 16;
    // same association as the stand-alone kernel: (v00*lx0 + v01*lx1)*ly0 + (v10*lx0 + v11*lx1)*ly1
    const f32x4 sig = (hs(ld(sg, o00 + c)) * (1.f - wx1) + hs(ld(sg, o01 + c)) * wx1) * (1.f - wy1) +
                      (hs(ld(sg, o10 + c)) * (1.f - wx1) + hs(ld(sg, o11 + c)) * wx1) * wy1;
    const f32x4 feat = (ld(sf, o00 + c) * (1.f - wx1) + ld(sf, o01 + c) * wx1) * (1.f - wy1) +
                       (ld(sf, o10 + c) * (1.f - wx1) + ld(sf, o11 + c) * wx1) * wy1;
    // the unfused pair rounds the local map to the storage type before the tail reads it back: keep that rounding
    f32x4 loc;
#pragma unroll
    for (int j = 0; j < 4; ++j) loc[j] = (float)(T)acc[nb][j];
    bstore4<T>(yrs, (uint32_t)yo + (uint32_t)((c + 4 * g) * SZ), loc * sig + feat);
  }
  }   // patch loop
}

// ---- the same with the NEXT layer's 1x1 convolution behind it ------------------------------------------------------------------------
// In the GD neck the injection's 256-channel output has ONE consumer: C2f.cv1, a 1x1 Conv + BN + SiLU down to 64 channels
// (models/v8/mspa_c2f_gd_yolov8.yaml head rows 4-5; nn/modules/block.py:199-201).  Written out and read back, that map is 105 MB each way at
// B = 32, 80x80 - more than everything else the two launches move.  Here the injected values go from the first GEMM's accumulators through
// the bilinear tail straight into the second GEMM's B operand (the accumulator layout of one MFMA is the B layout of the next once the
// second panel's K order is permuted to match - the trick of mgdt_cnx_mlp_fwd), rounded to bf16 exactly where the stored map would be:
// x (64 ch) + the two global maps in, 64 channels out.  Workgroup = TH waves = a TH x 16 patch; both panels + both source patches in LDS.
struct InjConvArgs {
  InjArgs a;
  const char* gx; int gxsn, gxsh, gxsw; uint32_t gx_bytes;   // GCONV: the 32-channel global input the two global convs read (ga / gf are then computed here)
  const char* wg; const float* bias_g;                       // GCONV: merged panel [global_act | global_embedding] 32 -> 2 * Cout, mgdt_conv_pack layout
  const char* w2; const float* bias2;      // 1x1 panel of the second conv with its K (= Cout of the injection) in ACCUMULATOR order, BN folded
  char* y2; int y2sn, y2sh, y2sw; uint32_t y2_bytes;
  int C2, act2;
};

template <int KC, int NB, int NB2, int TH, bool GCONV>
__global__ __launch_bounds__(64 * TH) void conv1x1_inject_conv_kernel(const InjConvArgs A) {
  typedef bf16 T;
  typedef bf16x8 frag;
  const InjArgs& a = A.a;
  constexpr int SZ = 2, THREADS = 64 * TH;
  constexpr int CS = NB * 16 + INJ_CPAD;
  constexpr int KC2 = NB / 2;                              // K chunks of the second conv: two 16-row blocks of the first conv's output each
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                         // [KC][NB][64][16 B]
  char* w2l = smem + (size_t)KC * NB * 1024;               // [KC2][NB2][64][16 B]
  T* sg = (T*)(w2l + (size_t)KC2 * NB2 * 1024);
  T* sf = sg + (size_t)a.PH * a.PW * CS;
  // GCONV: the two source patches are kept TRANSPOSED, as MFMA A fragments over K = source pixel (<= 64 = 2 chunks): frag(nb, kc)[lane (r, g)][e] =
  // map[source pixel kc*32 + 8g + e][channel nb*16 + r].  The bilinear interpolation of all 256 channels of 16 output pixels is then two
  // MFMAs per 16-channel block and map (B = the lane's pixel's four tap weights scattered over its 64 source-pixel slots) instead of ~100
  // VALU instructions per block: the tail was VALU-bound (h_sigmoid + four-tap lerps in fp32 for 2 x 256 channels per pixel).  h_sigmoid is
  // applied once per source pixel while the patch is built (the gate is then rounded to bf16 once more: <= 2^-9 relative on the gate).
  char* hgT = (char*)sg;                                   // [NB][2][64 lanes][16 B]
  char* gfT = hgT + (size_t)NB * 2 * 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  {
    // both weight panels: every 16-byte request is issued before the first LDS store (the copy loops waited for each load in turn - eight L2 round
    // trips in front of a persistent workgroup's first patch)
    constexpr int W1 = KC * NB * 64, W2 = KC2 * NB2 * 64, N1 = (W1 + THREADS - 1) / THREADS, N2 = (W2 + THREADS - 1) / THREADS;
    uint4 t1[N1], t2[N2];
#pragma unroll
    for (int u = 0; u < N1; ++u) t1[u] = ((const uint4*)a.wpk)[min(tid + u * THREADS, W1 - 1)];
#pragma unroll
    for (int u = 0; u < N2; ++u) t2[u] = ((const uint4*)A.w2)[min(tid + u * THREADS, W2 - 1)];
#pragma unroll
    for (int u = 0; u < N1; ++u) if (tid + u * THREADS < W1) ((uint4*)wl)[tid + u * THREADS] = t1[u];
#pragma unroll
    for (int u = 0; u < N2; ++u) if (tid + u * THREADS < W2) ((uint4*)w2l)[tid + u * THREADS] = t2[u];
  }
  if (GCONV) {
    for (int i = tid; i < NB * 2 * 2 * 64; i += THREADS) ((uint4*)hgT)[i] = make_uint4(0u, 0u, 0u, 0u);     // unused source slots must stay finite (x 0 weights)
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)A.y2, 0, A.y2_bytes, 0x00020000);
  const char* const wlane = wl + lane * 16;
  const char* const w2lane = w2l + lane * 16;
  // GCONV: global_act and global_embedding (two 1x1 Conv+BN, no activation, 32 -> NB*16 channels each; block.py:379-381) are evaluated here on the
  // patch's source pixels instead of being read back from HBM: this wave owns 2*NB/TH of the 2*NB cout blocks, weights in registers
  static_assert(!GCONV || (2 * NB) % TH == 0, "cout blocks per wave");
  constexpr int GB = GCONV ? (2 * NB) / TH : 1;
  bf16x8 Ag[GB];
  f32x4 bg[GB];
  if (GCONV) {
#pragma unroll
    for (int t = 0; t < GB; ++t) {
      Ag[t] = *(const bf16x8*)(A.wg + ((size_t)(wave * GB + t) * 64 + lane) * 16);
      bg[t] = *(const f32x4*)(A.bias_g + (wave * GB + t) * 16 + 4 * g);
    }
  }
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(GCONV ? A.gx : a.x), 0, GCONV ? A.gx_bytes : 0u, 0x00020000);
  const int npatch = a.N * a.tiles_x * a.tiles_y;
  // Everything a patch reads from HBM - this lane's x fragments and (GCONV) the source pixels of the global convs - is requested ONE PATCH
  // AHEAD, right after the LDS patch of the current one is built: with one 8-wave workgroup per CU nothing else hides that round trip.
  constexpr int NG = 4;                                    // 16-pixel groups of a source patch (host: PH * PW <= 64)
  auto geom = [&](int patch, int& n, int& ty, int& tx, int& py0, int& px0, int& ph, int& pw) __attribute__((always_inline)) {
    int b = patch;
    tx = b % a.tiles_x; b /= a.tiles_x;
    ty = b % a.tiles_y; n = b / a.tiles_y;
    int py1, px1, t0, t1; float tl;
    inj_lerp(ty * TH, a.Hg, a.H, py0, t1, tl);
    inj_lerp(min(ty * TH + TH - 1, a.H - 1), a.Hg, a.H, t0, py1, tl);
    inj_lerp(tx * INJ_TW, a.Wg, a.W, px0, t1, tl);
    inj_lerp(min(tx * INJ_TW + INJ_TW - 1, a.W - 1), a.Wg, a.W, t0, px1, tl);
    ph = py1 - py0 + 1; pw = px1 - px0 + 1;
  };
  auto request = [&](int patch, frag(&Pq)[KC], frag(&Bq)[NG]) __attribute__((always_inline)) {
    int n, ty, tx, py0, px0, ph, pw;
    const bool live = patch < npatch;
    geom(live ? patch : 0, n, ty, tx, py0, px0, ph, pw);
    const int oy = ty * TH + wave, ox = tx * INJ_TW + r;
    const int xo = (live && oy < a.H && ox < a.W) ? n * a.xsn + oy * a.xsh + ox * a.xsw : MGDT_OOB;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int cb = (kc * 4 + g) * 16;
      Pq[kc] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(xrs, (cb < a.Cin * SZ) ? (uint32_t)(xo + cb) : (uint32_t)MGDT_OOB, 0, 0));
    }
    if (GCONV) {
      const int npx = ph * pw;
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        const int p = grp * 16 + r;
        const bool v_ = live && p < npx;
        const int sy = (v_ ? p : 0) / pw, sx = (v_ ? p : 0) - sy * pw;
        const uint32_t go = v_ ? (uint32_t)(n * A.gxsn + (py0 + sy) * A.gxsh + (px0 + sx) * A.gxsw + g * 16) : (uint32_t)MGDT_OOB;
        Bq[grp] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(grs, go, 0, 0));      // 32 input channels = one K chunk
      }
    }
  };
  frag P[KC], Pn[KC], Bgc[NG], Bgn[NG];
  request((int)blockIdx.x, P, Bgc);
#pragma unroll 1
  for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
    int n, ty, tx, py0, px0, ph, pw;
    geom(patch, n, ty, tx, py0, px0, ph, pw);
    const int oy = ty * TH + wave, ox = tx * INJ_TW + r;
    const bool pv = oy < a.H && ox < a.W;
    __syncthreads();                                       // the previous patch's tail is done with the LDS maps
    if (GCONV) {
      const int npx = ph * pw;
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (grp * 16 >= npx) break;                          // uniform
        const int p = grp * 16 + r;
        const bool v_ = p < npx;
        const frag Bg = Bgc[grp];
#pragma unroll
        for (int t = 0; t < GB; ++t) {
          const int nbg = wave * GB + t;                    // uniform
          const f32x4 o = mma(Ag[t], Bg, bg[t]);
          const bool gate = nbg < NB;                        // uniform
          char* base = (gate ? hgT : gfT) + ((size_t)((nbg % NB) * 2 + (p >> 5)) * 64 + ((p & 31) >> 3) * 16 + 4 * g) * 16 + (p & 7) * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = (float)(bf16)o[i];                     // the stored (rounded) global map value
            if (gate) v = fminf(fmaxf(fmaf(v, 1.f / 6.f, 0.5f), 0.f), 1.f);      // h_sigmoid (block.py:344-350) before the interpolation (block.py:393)
            if (v_) *(bf16*)(base + i * 16) = (bf16)v;       // lane slot (r = 4g + i, g' = source pixel / 8), element = source pixel % 8
          }
        }
      }
    } else {
      constexpr int VPP = NB * 16 / 8;
      const int nvec = ph * pw * VPP;
      for (int i = tid; i < nvec; i += THREADS) {
        const int v = i % VPP, p = i / VPP;
        const int sy = p / pw, sx = p - sy * pw;
        const long go = (long)n * a.gsn + (long)(py0 + sy) * a.gsh + (long)(px0 + sx) * a.gsw + v * 16;
        const uint4 va = *(const uint4*)(a.ga + go), vf = *(const uint4*)(a.gf + go);
        *(uint4*)((char*)sg + ((size_t)p * CS + v * 8) * SZ) = va;
        *(uint4*)((char*)sf + ((size_t)p * CS + v * 8) * SZ) = vf;
      }
    }
    request(patch + (int)gridDim.x, Pn, Bgn);              // the next patch's HBM reads fly under this patch's GEMMs and tail
    __syncthreads();
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = *(const f32x4*)(a.bias + nb * 16 + 4 * g);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mma(*(const frag*)(wlane + (kc * NB + nb) * 1024), P[kc], acc[nb]);
    int y0, y1, x0, x1; float wy1, wx1;
    inj_lerp(min(oy, a.H - 1), a.Hg, a.H, y0, y1, wy1);
    inj_lerp(min(ox, a.W - 1), a.Wg, a.W, x0, x1, wx1);
    const int o00 = ((y0 - py0) * pw + (x0 - px0)) * CS + 4 * g, o01 = ((y0 - py0) * pw + (x1 - px0)) * CS + 4 * g;
    const int o10 = ((y1 - py0) * pw + (x0 - px0)) * CS + 4 * g, o11 = ((y1 - py0) * pw + (x1 - px0)) * CS + 4 * g;
    auto ld = [&](const T* base, int off) __attribute__((always_inline)) {
      const bf16x4 v = *(const bf16x4*)(base + off);
      return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    };
    auto hs = [](f32x4 v) __attribute__((always_inline)) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(fmaf(v[j], 1.f / 6.f, 0.5f), 0.f), 1.f);
      return o;
    };
    f32x4 acc2[NB2];
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) acc2[nb] = *(const f32x4*)(A.bias2 + nb * 16 + 4 * g);
    // GCONV: this lane's pixel as an interpolation B operand: its four tap weights at their source-pixel slots (coinciding taps add up), zeros elsewhere
    frag WB[2];
    if (GCONV) {
      const int s00 = (y0 - py0) * pw + (x0 - px0), s01 = (y0 - py0) * pw + (x1 - px0), s10 = (y1 - py0) * pw + (x0 - px0), s11 = (y1 - py0) * pw + (x1 - px0);
      const float w00 = (1.f - wy1) * (1.f - wx1), w01 = (1.f - wy1) * wx1, w10 = wy1 * (1.f - wx1), w11 = wy1 * wx1;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int sidx = kc * 32 + 8 * g + e;
          const float w = (sidx == s00 ? w00 : 0.f) + (sidx == s01 ? w01 : 0.f) + (sidx == s10 ? w10 : 0.f) + (sidx == s11 ? w11 : 0.f);
          WB[kc][e] = (bf16)w;
        }
    }
    const char* const hglane = hgT + lane * 16;
    const char* const gflane = gfT + lane * 16;
#pragma unroll
    for (int j = 0; j < KC2; ++j) {
      frag B2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (GCONV) {
          const int nb = 2 * j + h;
          f32x4 sig = f32x4{0.f, 0.f, 0.f, 0.f}, feat = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            sig = mma(*(const frag*)(hglane + (nb * 2 + kc) * 1024), WB[kc], sig);
            feat = mma(*(const frag*)(gflane + (nb * 2 + kc) * 1024), WB[kc], feat);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) B2[h * 4 + i] = (T)((float)(T)acc[nb][i] * sig[i] + feat[i]);
          continue;
        }
        const int c = (2 * j + h) * 16;
        const f32x4 sig = (hs(ld(sg, o00 + c)) * (1.f - wx1) + hs(ld(sg, o01 + c)) * wx1) * (1.f - wy1) +
                          (hs(ld(sg, o10 + c)) * (1.f - wx1) + hs(ld(sg, o11 + c)) * wx1) * wy1;
        const f32x4 feat = (ld(sf, o00 + c) * (1.f - wx1) + ld(sf, o01 + c) * wx1) * (1.f - wy1) +
                           (ld(sf, o10 + c) * (1.f - wx1) + ld(sf, o11 + c) * wx1) * wy1;
#pragma unroll
        for (int i = 0; i < 4; ++i) B2[h * 4 + i] = (T)((float)(T)acc[2 * j + h][i] * sig[i] + feat[i]);   // both roundings of the unfused pair
      }
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) acc2[nb] = mma(*(const frag*)(w2lane + (j * NB2 + nb) * 1024), B2, acc2[nb]);
    }
    const int yo = pv ? n * A.y2sn + oy * A.y2sh + ox * A.y2sw : MGDT_OOB;
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) {
      const int c = nb * 16 + 4 * g;
      f32x4 v = acc2[nb];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = A.act2 == MGDT_ACT_SILU ? v[i] * fast_sigmoid(v[i]) : act_apply(v[i], A.act2);
      bstore4<T>(yrs, c < A.C2 ? (uint32_t)yo + (uint32_t)(c * SZ) : (uint32_t)MGDT_OOB, v);
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) P[kc] = Pn[kc];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) Bgc[grp] = Bgn[grp];
  }
}

static int inj_kc(int cin) { return (cin + 31) / 32; }   // K chunks of the packed 1x1 panel (mgdt_conv_pack layout)

// largest source patch any workgroup needs (rows x cols)
static void inj_patch(int H, int W, int Hg, int Wg, int* PH, int* PW) {
  int ph = 1, pw = 1, a0, a1, b0, b1; float t;
  for (int ty = 0; ty * INJ_TH < H; ++ty) {
    inj_lerp(ty * INJ_TH, Hg, H, a0, a1, t);
    inj_lerp(std::min(ty * INJ_TH + INJ_TH - 1, H - 1), Hg, H, b0, b1, t);
    ph = std::max(ph, b1 - a0 + 1);
  }
  for (int tx = 0; tx * INJ_TW < W; ++tx) {
    inj_lerp(tx * INJ_TW, Wg, W, a0, a1, t);
    inj_lerp(std::min(tx * INJ_TW + INJ_TW - 1, W - 1), Wg, W, b0, b1, t);
    pw = std::max(pw, b1 - a0 + 1);
  }
  *PH = ph; *PW = pw;
}

static size_t inj_lds(int kc, int nb, int PH, int PW) { return (size_t)kc * nb * 1024 + (size_t)2 * PH * PW * (nb * 16 + INJ_CPAD) * 2; }

/* 1 when mgdt_conv1x1_inject_fwd covers the shapes: bf16, cin <= 128, cout in {128, 256}, up-sampling (h >= hg, w >= wg) */
extern "C" int mgdt_conv1x1_inject_supported(int cin, int cout, int h, int w, int hg, int wg, int dtype) {
  if (dtype != MGDT_BF16 || cin % 8 || cin > 128 || (cout != 128 && cout != 256) || h < hg || w < wg || hg < 1 || wg < 1) return 0;
  int PH, PW;
  inj_patch(h, w, hg, wg, &PH, &PW);
  return inj_lds(inj_kc(cin), cout / 16, PH, PW) <= 80 * 1024;   // two workgroups per CU
}

template <int KC, int NB>
static int inj_launch(const InjArgs& a, size_t lds, hipStream_t st) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1x1_inject_kernel<bf16, KC, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "conv1x1_inject: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  conv1x1_inject_kernel<bf16, KC, NB><<<std::min(a.N * a.tiles_x * a.tiles_y, 512), INJ_THREADS, lds, st>>>(a);   // persistent: 2 per CU
  MGDT_CHECK_LAUNCH("conv1x1_inject_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_conv1x1_inject_fwd(const mgdt_view* x, const void* packed_w, const float* bias, const mgdt_view* ga, const mgdt_view* gf,
                                       const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(ga) || !view_ok(gf) || !view_ok(y) || !packed_w || !bias) MGDT_FAIL(MGDT_BAD_ARG, "conv1x1_inject: null/empty argument");
  if (!mgdt_conv1x1_inject_supported(x->c, y->c, y->h, y->w, ga->h, ga->w, dtype))
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject: shapes/dtype not covered (see mgdt_conv1x1_inject_supported)");
  const long sz = 2;
  if (x->n != y->n || x->h != y->h || x->w != y->w || ga->n != y->n || gf->n != y->n || ga->c != y->c || gf->c != y->c || ga->h != gf->h || ga->w != gf->w ||
      ga->sn != gf->sn || ga->sh != gf->sh || ga->sw != gf->sw)
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject: x/y spatial sizes, ga/gf shapes and layouts must match");
  auto ok = [&](const mgdt_view* v, int q) { return v->sc == 1 && v->sw % q == 0 && v->sh % q == 0 && v->sn % q == 0 && (uintptr_t)v->p % (q * sz) == 0; };
  auto ext = [&](const mgdt_view* v) { return ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz; };
  if (!ok(x, 8) || !ok(ga, 8) || !ok(gf, 8) || !ok(y, 4) || ext(x) >= 0x7fffffffL || ext(y) >= 0x7fffffffL)
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject: views must be aligned NHWC (sc == 1) and < 2 GiB");
  InjArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const char*)x->p; a.xsn = (int)(x->sn * sz); a.xsh = (int)(x->sh * sz); a.xsw = (int)(x->sw * sz); a.x_bytes = (uint32_t)ext(x);
  a.y = (char*)y->p; a.ysn = (int)(y->sn * sz); a.ysh = (int)(y->sh * sz); a.ysw = (int)(y->sw * sz); a.y_bytes = (uint32_t)ext(y);
  a.ga = (const char*)ga->p; a.gf = (const char*)gf->p; a.gsn = (int)(ga->sn * sz); a.gsh = (int)(ga->sh * sz); a.gsw = (int)(ga->sw * sz);
  a.wpk = (const char*)packed_w; a.bias = bias;
  a.N = y->n; a.H = y->h; a.W = y->w; a.Cin = x->c; a.Cout = y->c; a.Hg = ga->h; a.Wg = ga->w;
  inj_patch(a.H, a.W, a.Hg, a.Wg, &a.PH, &a.PW);
  a.tiles_x = cdiv(a.W, INJ_TW); a.tiles_y = cdiv(a.H, INJ_TH);
  const int kc = inj_kc(a.Cin), nb = a.Cout / 16;
  const size_t lds = inj_lds(kc, nb, a.PH, a.PW);
  hipStream_t st = (hipStream_t)s;
#define INJ_CASE(K, B) if (kc == K && nb == B) return inj_launch<K, B>(a, lds, st);
  INJ_CASE(1, 8) INJ_CASE(2, 8) INJ_CASE(3, 8) INJ_CASE(4, 8) INJ_CASE(1, 16) INJ_CASE(2, 16) INJ_CASE(3, 16) INJ_CASE(4, 16)
#undef INJ_CASE
  MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject: no kernel for kc=%d nb=%d", kc, nb);
}

// ---- injection + the following 1x1 conv ------------------------------------------------------------------------------------------------
constexpr int INJ2_TH = 8;
static void inj_patch_th(int H, int W, int Hg, int Wg, int TH, int* PH, int* PW) {
  int ph = 1, pw = 1, a0, a1, b0, b1; float t;
  for (int ty = 0; ty * TH < H; ++ty) {
    inj_lerp(ty * TH, Hg, H, a0, a1, t);
    inj_lerp(std::min(ty * TH + TH - 1, H - 1), Hg, H, b0, b1, t);
    ph = std::max(ph, b1 - a0 + 1);
  }
  for (int tx = 0; tx * INJ_TW < W; ++tx) {
    inj_lerp(tx * INJ_TW, Wg, W, a0, a1, t);
    inj_lerp(std::min(tx * INJ_TW + INJ_TW - 1, W - 1), Wg, W, b0, b1, t);
    pw = std::max(pw, b1 - a0 + 1);
  }
  *PH = ph; *PW = pw;
}
static size_t inj2_lds(int kc, int nb, int nb2, int PH, int PW) {
  // patches: [pixel][channel] rows (maps computed by the caller) or, GCONV, 2 maps x nb blocks x 2 K chunks x 1 KiB fragments - never more than the rows
  return (size_t)kc * nb * 1024 + (size_t)(nb / 2) * nb2 * 1024 + std::max((size_t)2 * PH * PW * (nb * 16 + INJ_CPAD) * 2, (size_t)2 * nb * 2 * 1024);
}

/* 1 when mgdt_conv1x1_inject_conv_fwd covers the shapes: bf16, cin <= 128 (% 8), injection width 256, second conv 256 -> cout2 <= 64 (% 16) */
extern "C" int mgdt_conv1x1_inject_conv_supported(int cin, int cmid, int cout2, int h, int w, int hg, int wg, int dtype) {
  if (dtype != MGDT_BF16 || cin % 8 || cin > 128 || cmid != 256 || cout2 % 16 || cout2 < 16 || cout2 > 64 || h < hg || w < wg || hg < 1 || wg < 1) return 0;
  int PH, PW;
  inj_patch_th(h, w, hg, wg, INJ2_TH, &PH, &PW);
  return PH * PW <= 64 && inj2_lds(inj_kc(cin), cmid / 16, cout2 / 16, PH, PW) <= 156 * 1024;      // source patch within the 64 K slots of the MFMA interpolation
}

template <int KC, int NB2, bool GCONV>
static int inj2_launch(const InjConvArgs& A, size_t lds, hipStream_t st) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1x1_inject_conv_kernel<KC, 16, NB2, INJ2_TH, GCONV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "conv1x1_inject_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  conv1x1_inject_conv_kernel<KC, 16, NB2, INJ2_TH, GCONV><<<std::min(A.a.N * A.a.tiles_x * A.a.tiles_y, 256), 64 * INJ2_TH, lds, st>>>(A);   // persistent: one per CU
  MGDT_CHECK_LAUNCH("conv1x1_inject_conv_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_conv1x1_inject_conv_fwd(const mgdt_view* x, const void* packed_w, const float* bias, const mgdt_view* ga_in, const mgdt_view* gf_in,
                                            const mgdt_view* gsrc, const void* packed_wg, const float* bias_g, int cmid,
                                            const void* packed_w2, const float* bias2, int act2, const mgdt_view* y2, int dtype, mgdt_stream s) {
  // two forms: (ga, gf) = the global act / embedding maps computed by the caller, or (gsrc, packed_wg, bias_g, cmid) = their common 32-channel
  // input + the merged panel [global_act | global_embedding] (mgdt_conv_pack(32, 2 * cmid, 1, bf16)): the maps are then evaluated per tile
  const bool gconv = gsrc && gsrc->p;
  mgdt_view gav, gfv;
  if (gconv) {
    if (!view_ok(gsrc) || !packed_wg || !bias_g || gsrc->c != 32 || cmid != 256) MGDT_FAIL(MGDT_BAD_ARG, "conv1x1_inject_conv: global source must be a 32-channel view with its merged panel (cmid 256)");
    gav = *gsrc; gav.c = cmid; gfv = gav;                   // geometry only (n, h, w): the kernel never dereferences ga / gf in this form
  } else {
    if (!view_ok(ga_in) || !view_ok(gf_in)) MGDT_FAIL(MGDT_BAD_ARG, "conv1x1_inject_conv: null/empty global maps");
    gav = *ga_in; gfv = *gf_in;
  }
  const mgdt_view* ga = &gav; const mgdt_view* gf = &gfv;
  if (!view_ok(x) || !view_ok(y2) || !packed_w || !bias || !packed_w2 || !bias2) MGDT_FAIL(MGDT_BAD_ARG, "conv1x1_inject_conv: null/empty argument");
  if (!mgdt_conv1x1_inject_conv_supported(x->c, ga->c, y2->c, x->h, x->w, ga->h, ga->w, dtype))
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject_conv: shapes/dtype not covered (see mgdt_conv1x1_inject_conv_supported)");
  const long sz = 2;
  if (x->n != y2->n || x->h != y2->h || x->w != y2->w || ga->n != x->n || gf->n != x->n || gf->c != ga->c || ga->h != gf->h || ga->w != gf->w ||
      ga->sn != gf->sn || ga->sh != gf->sh || ga->sw != gf->sw)
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject_conv: x/y2 spatial sizes, ga/gf shapes and layouts must match");
  auto ok = [&](const mgdt_view* v, int q) { return v->sc == 1 && v->sw % q == 0 && v->sh % q == 0 && v->sn % q == 0 && (uintptr_t)v->p % (q * sz) == 0; };
  auto ext = [&](const mgdt_view* v) { return ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz; };
  if (!ok(x, 8) || (!gconv && (!ok(ga, 8) || !ok(gf, 8))) || (gconv && (!ok(gsrc, 8) || ((long)(gsrc->n - 1) * gsrc->sn + (long)(gsrc->h - 1) * gsrc->sh + (long)(gsrc->w - 1) * gsrc->sw + gsrc->c) * sz >= 0x7fffffffL)) ||
      !ok(y2, 4) || ext(x) >= 0x7fffffffL || ext(y2) >= 0x7fffffffL)
    MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject_conv: views must be aligned NHWC (sc == 1) and < 2 GiB");
  InjConvArgs A;
  memset(&A, 0, sizeof(A));
  InjArgs& a = A.a;
  if (gconv) {
    A.gx = (const char*)gsrc->p; A.gxsn = (int)(gsrc->sn * sz); A.gxsh = (int)(gsrc->sh * sz); A.gxsw = (int)(gsrc->sw * sz);
    A.gx_bytes = (uint32_t)(((long)(gsrc->n - 1) * gsrc->sn + (long)(gsrc->h - 1) * gsrc->sh + (long)(gsrc->w - 1) * gsrc->sw + gsrc->c) * sz);
    A.wg = (const char*)packed_wg; A.bias_g = bias_g;
  }
  a.x = (const char*)x->p; a.xsn = (int)(x->sn * sz); a.xsh = (int)(x->sh * sz); a.xsw = (int)(x->sw * sz); a.x_bytes = (uint32_t)ext(x);
  a.ga = (const char*)ga->p; a.gf = (const char*)gf->p; a.gsn = (int)(ga->sn * sz); a.gsh = (int)(ga->sh * sz); a.gsw = (int)(ga->sw * sz);
  a.wpk = (const char*)packed_w; a.bias = bias;
  a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = ga->c; a.Hg = ga->h; a.Wg = ga->w;
  inj_patch_th(a.H, a.W, a.Hg, a.Wg, INJ2_TH, &a.PH, &a.PW);
  a.tiles_x = cdiv(a.W, INJ_TW); a.tiles_y = cdiv(a.H, INJ2_TH);
  A.w2 = (const char*)packed_w2; A.bias2 = bias2; A.act2 = act2; A.C2 = y2->c;
  A.y2 = (char*)y2->p; A.y2sn = (int)(y2->sn * sz); A.y2sh = (int)(y2->sh * sz); A.y2sw = (int)(y2->sw * sz); A.y2_bytes = (uint32_t)ext(y2);
  const int kc = inj_kc(a.Cin), nb2 = cdiv(y2->c, 16);
  const size_t lds = inj2_lds(kc, 16, nb2, a.PH, a.PW);
  hipStream_t st = (hipStream_t)s;
#define INJ2_CASE(K, B) if (kc == K && nb2 == B) return gconv ? inj2_launch<K, B, true>(A, lds, st) : inj2_launch<K, B, false>(A, lds, st);
  INJ2_CASE(1, 1) INJ2_CASE(1, 2) INJ2_CASE(1, 3) INJ2_CASE(1, 4) INJ2_CASE(2, 1) INJ2_CASE(2, 2) INJ2_CASE(2, 3) INJ2_CASE(2, 4)
  INJ2_CASE(3, 1) INJ2_CASE(3, 2) INJ2_CASE(3, 3) INJ2_CASE(3, 4) INJ2_CASE(4, 1) INJ2_CASE(4, 2) INJ2_CASE(4, 3) INJ2_CASE(4, 4)
#undef INJ2_CASE
  MGDT_FAIL(MGDT_BAD_SHAPE, "conv1x1_inject_conv: no kernel for kc=%d nb2=%d", kc, nb2);
}
