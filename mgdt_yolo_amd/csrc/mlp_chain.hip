// ConvNeXtV2 block MLP (reference nn/modules/convnextv2.py:62-77: pwconv1 -> GELU -> GRN -> pwconv2 -> + input) without ever
// writing the 4C-wide hidden map to HBM.
//
// GRN needs sum_hw h^2 per (image, hidden channel) before pwconv2 can start, so the hidden activations are computed TWICE
// (K = C is tiny) instead of stored once and read twice:
//   pass 1 (STATS): h = gelu(W1 t + b1) in MFMA accumulators -> per-lane h^2 sums -> ws[n][split][4C]
//   pass 2 (APPLY): scale[n][j] = gamma[j] * Gx / (mean_j Gx + 1e-6) + 1 with Gx = sqrt(sum) in the prologue, then h again,
//                   v = h * scale + beta, y = W2 v + b2 + res
// The accumulator layout of pwconv1 (lane (r, g) holds hidden rows 4g..4g+3 of pixel r) IS the B-operand layout of the next
// MFMA once pwconv2's K order is permuted to match (done in the pack kernel), so h goes from one GEMM to the next through
// registers only: no LDS round trip, no shuffle.  Both weight matrices (2 * 8 C^2 bf16 = 144 KiB at C = 96) live in LDS.
// HBM traffic per block: t twice + res + y = 4 * N*H*W*C elements, instead of 2 + 2*4 + 4 + 2 = 16 for the unfused chain.
#include <algorithm>

#include "conv_igemm_kernel.h"
#include "mlp_common.h"

struct MlpArgs {
  const char* t; int tsn, tsh, tsw; uint32_t t_bytes;
  const char* res; int rsn, rsh, rsw; uint32_t r_bytes;
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const char* packed;
  const float* gamma; const float* shift;   // APPLY: GRN gamma / beta [4C]
  float* ws;                                // [N][splits][4C] partial sums of h^2: written by STATS, reduced by APPLY
  int N, H, W, C, HW, tiles;                // tiles = wave tiles (MT*16 pixels) per image
  FastDiv fd_w;
  unsigned long long* dbg;                  // MGDT_MLP_DBG: phase timestamps of workgroup (0, 0) (100 MHz ticks)
};

template <typename T> struct MlpGeom {
  static constexpr int PE = Piece<T>::PE, BPC = PE / 4;   // BPC hidden 16-blocks form one K chunk of pwconv2
};
// blob layout: W1 [KC1][NB1][64][PE] | W2 [KC2][NB2][64][PE] | b1 [4C] f32 | b2 [NB2*16] f32
static inline size_t mlp_w1_bytes(int kc1) { return (size_t)kc1 * (kc1 * 8) * 1024; }          // NB1 = 4C/16 = 8*KC1 (C = 32*KC1)
static inline size_t mlp_w2_bytes(int kc1) { return (size_t)(4 * kc1) * (2 * kc1) * 1024; }    // KC2 = 4*KC1, NB2 = 2*KC1

constexpr int MLP_THREADS = 1024;   // 16 waves: the passes are VALU-bound (GELU), so fill all four SIMDs four deep

template <typename T, int KC1, int MT, bool STATS>
__global__ __launch_bounds__(MLP_THREADS) void cnx_mlp_kernel(const MlpArgs a) {
  typedef typename Piece<T>::frag frag;
  constexpr int PE = Piece<T>::PE, BPC = PE / 4, SZ = (int)sizeof(T);
  constexpr int NB1 = 8 * KC1, KC2 = 4 * KC1, NB2 = 2 * KC1, HD = 128 * KC1;   // C = 32*KC1 (bf16)
  static_assert(BPC * KC2 == NB1, "hidden blocks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w1l = smem;
  char* w2l = smem + (size_t)KC1 * NB1 * 1024;
  float* aux = (float*)(STATS ? w2l : w2l + (size_t)KC2 * NB2 * 1024);   // STATS: red[nwaves][HD]; APPLY: scale[HD] | shift[HD]
  float* bl = aux + (STATS ? (MLP_THREADS / 64) * HD : 2 * HD);            // b1[HD] | b2[NB2*16]: read every chunk, so not from global
  constexpr int nthr = MLP_THREADS;   // compile-time trip counts: the staging loads below are issued back to back
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n = blockIdx.y;
  const char* w1g = a.packed;
  const char* w2g = a.packed + (size_t)KC1 * NB1 * 1024;
  const float* b1g = (const float*)(w2g + (size_t)KC2 * NB2 * 1024);
  const float* b1 = bl;
  const float* b2 = bl + HD;

  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.t, 0, a.t_bytes, 0x00020000);
  // pixels of wave tile `tile` of image n; activations of the first tile are requested before the weights are staged
  auto pix = [&](int tile, int mt, int& yo, int& ro) __attribute__((always_inline)) {
    const int p = (tile * MT + mt) * 16 + r;
    const bool pv = tile < a.tiles && p < a.HW;
    const int pp = pv ? p : 0;
    const int oy = (int)fdiv((uint32_t)pp, a.fd_w), ox = pp - oy * a.W;
    yo = pv ? n * a.ysn + __mul24(oy, a.ysh) + __mul24(ox, a.ysw) : MGDT_OOB;
    ro = pv ? n * a.rsn + __mul24(oy, a.rsh) + __mul24(ox, a.rsw) : MGDT_OOB;
    return pv ? n * a.tsn + __mul24(oy, a.tsh) + __mul24(ox, a.tsw) : MGDT_OOB;
  };
  auto load_t = [&](int tile, frag(&P)[KC1][MT], int(&yo)[MT], int(&ro)[MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int xo = pix(tile, mt, yo[mt], ro[mt]);
#pragma unroll
      for (int kc = 0; kc < KC1; ++kc) {
        const int cb = (kc * 4 + g) * 16;                      // piece offset in bytes; pieces past C read zeros
        const uint32_t off = (cb < a.C * SZ) ? (uint32_t)(xo + cb) : (uint32_t)MGDT_OOB;
        P[kc][mt] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(trs, off, 0, 0));
      }
    }
  };
  const bool dbg = a.dbg && tid == 0;
  unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0;
  if (dbg) T0 = wall_clock64();
  const int tile0 = blockIdx.x * nw + wave, tstep = gridDim.x * nw;
  frag P[KC1][MT];
  int yo[MT], ro[MT];
  load_t(tile0, P, yo, ro);

  // all global reads of the prologue are issued before the first LDS store waits on any of them
  constexpr int W1 = KC1 * NB1 * 64, W2 = STATS ? 0 : KC2 * NB2 * 64;   // 16-byte words
  // every workgroup reads the same blob at the same moment: each starts at its own 4 KiB offset (per XCD, ids/8 are the co-resident
  // ones), otherwise all of them queue on the same L2 channels (measured: 10 us for 144 KiB)
  const int wg8 = (int)((blockIdx.y * gridDim.x + blockIdx.x) >> 3);
  const int rot1 = (wg8 * 256) % W1, rot2 = STATS ? 0 : (wg8 * 256) % (W2 > 0 ? W2 : 1);
  auto rw = [](int i, int rot, int w) __attribute__((always_inline)) { const int k = i + rot; return k >= w ? k - w : k; };
  constexpr int F1 = W1 / nthr, R1 = W1 % nthr, F2 = W2 / nthr, R2 = W2 % nthr;   // full rounds + a partial one (kept out of the arrays:
  uint4 tmp1[F1 > 0 ? F1 : 1], tmp2[F2 > 0 ? F2 : 1];                               // a guarded array element would live in scratch)
  uint4 tail1 = make_uint4(0, 0, 0, 0), tail2 = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < F1; ++i) tmp1[i] = ((const uint4*)w1g)[rw(tid + i * nthr, rot1, W1)];
  if (R1 && tid < R1) tail1 = ((const uint4*)w1g)[rw(tid + F1 * nthr, rot1, W1)];
  if (!STATS) {
#pragma unroll
    for (int i = 0; i < F2; ++i) tmp2[i] = ((const uint4*)w2g)[rw(tid + i * nthr, rot2, W2)];
    if (R2 && tid < R2) tail2 = ((const uint4*)w2g)[rw(tid + F2 * nthr, rot2, W2)];
  }
  float gsum = 0.f;
  if (!STATS && tid < HD)       // GRN statistic of my channel: the splits summed in order
    for (int sp = 0; sp < (int)gridDim.x; ++sp) gsum += a.ws[((long)n * gridDim.x + sp) * HD + tid];
  for (int i = tid; i < HD + NB2 * 16; i += nthr) bl[i] = b1g[i];
#pragma unroll
  for (int i = 0; i < F1; ++i) ((uint4*)w1l)[rw(tid + i * nthr, rot1, W1)] = tmp1[i];
  if (R1 && tid < R1) ((uint4*)w1l)[rw(tid + F1 * nthr, rot1, W1)] = tail1;
  if (!STATS) {
    static_assert(HD <= nthr, "one thread per hidden channel in the GRN prologue");
#pragma unroll
    for (int i = 0; i < F2; ++i) ((uint4*)w2l)[rw(tid + i * nthr, rot2, W2)] = tmp2[i];
    if (R2 && tid < R2) ((uint4*)w2l)[rw(tid + F2 * nthr, rot2, W2)] = tail2;
    // GRN: scale[c] = gamma[c] * Gx[c] / (mean_c Gx + 1e-6) + 1 with Gx = sqrt(sum_hw h^2) (nn/modules/utils GRN)
    if (tid < HD) aux[HD + tid] = sqrtf(gsum);
    __syncthreads();
    if (wave == 0) {            // mean over channels: strided partials, then a butterfly - one fixed order
      float part = 0.f;
      for (int c = lane; c < HD; c += 64) part += aux[HD + c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
      if (lane == 0) bl[HD + NB2 * 16] = part / (float)HD;
    }
    __syncthreads();
    const float mean = bl[HD + NB2 * 16];
    if (tid < HD) { aux[tid] = a.gamma[tid] * (sqrtf(gsum) / (mean + 1e-6f)) + 1.f; aux[HD + tid] = a.shift[tid]; }
  }
  __syncthreads();

  if (dbg) T1 = wall_clock64();
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, STATS ? 0u : a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.t), 0, a.res ? a.r_bytes : 0u, 0x00020000);
  const char* const w1lane = w1l + lane * 16;
  const char* const w2lane = w2l + lane * 16;

  float* const red = aux + wave * HD;     // STATS: this wave's h^2 sums, one LDS word per hidden channel (no cross-wave sharing)
  if (STATS) {
    for (int c = lane; c < HD; c += 64) red[c] = 0.f;
  }

  for (int tile = tile0; tile < a.tiles; tile += tstep) {
    f32x4 acc2[STATS ? 1 : NB2][MT];
    if (!STATS) {
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        const f32x4 b = *(const f32x4*)(b2 + nb * 16 + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[nb][mt] = b;
      }
    }
#pragma unroll 1
    for (int j = 0; j < KC2; ++j) {            // one K chunk of pwconv2 = BPC hidden blocks (rolled: the GELU bodies are large)
      if constexpr (STATS) {
        // operands swapped: D[m = pixel 4g+i][n = hidden r], so a lane holds 4 pixels of ONE hidden channel and the sum over
        // pixels is 3 in-register adds + 2 cross-row exchanges instead of a 16-lane reduction per value
        static_assert(MT == 1, "STATS pass is written for MT == 1");
#pragma unroll
        for (int b = 0; b < BPC; ++b) {
          const float bias = b1[(j * BPC + b) * 16 + r];
          f32x4 acc = f32x4{bias, bias, bias, bias};
#pragma unroll
          for (int kc = 0; kc < KC1; ++kc) acc = mma(P[kc][0], *(const frag*)(w1lane + (kc * NB1 + j * BPC + b) * 1024), acc);
          float sq = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float h = (float)(T)gelu_fast(acc[i]);   // rounded as the stored hidden map would be
            sq += (tile * 16 + 4 * g + i < a.HW) ? h * h : 0.f;
          }
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (g == 0) red[(j * BPC + b) * 16 + r] += sq;
        }
      } else {
        f32x4 acc1[BPC][MT];
#pragma unroll
        for (int b = 0; b < BPC; ++b) {
          const f32x4 bias = *(const f32x4*)(b1 + (j * BPC + b) * 16 + 4 * g);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc1[b][mt] = bias;
        }
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc)
#pragma unroll
          for (int b = 0; b < BPC; ++b) {
            const frag Wf = *(const frag*)(w1lane + (kc * NB1 + j * BPC + b) * 1024);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc1[b][mt] = mma(Wf, P[kc][mt], acc1[b][mt]);
          }
        frag B2[MT];
#pragma unroll
        for (int b = 0; b < BPC; ++b) {
          const f32x4 sc = *(const f32x4*)(aux + (j * BPC + b) * 16 + 4 * g), sh = *(const f32x4*)(aux + HD + (j * BPC + b) * 16 + 4 * g);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float h = (float)(T)gelu_fast(acc1[b][mt][i]);   // rounded as the stored hidden map would be
              B2[mt][b * 4 + i] = (T)(h * sc[i] + sh[i]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
          const frag Wf = *(const frag*)(w2lane + (j * NB2 + nb) * 1024);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc2[nb][mt] = mma(Wf, B2[mt], acc2[nb][mt]);
        }
      }
    }
    if (dbg) T2 = wall_clock64();
    int yo_c[MT], ro_c[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { yo_c[mt] = yo[mt]; ro_c[mt] = ro[mt]; }
    if (tile + tstep < a.tiles) load_t(tile + tstep, P, yo, ro);    // uniform per wave
    if (!STATS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
          const int cob = (nb * 16 + 4 * g) * SZ;
          const int dead = cob >= a.C * SZ ? MGDT_OOB : 0;
          f32x4 v = acc2[nb][mt];
          if (a.res) v += bload4<T>(rrs, (uint32_t)(ro_c[mt] | dead) + cob);
          bstore4<T>(yrs, (uint32_t)(yo_c[mt] | dead) + cob, v);
        }
    }
  }

  if (dbg) { T3 = wall_clock64(); unsigned long long* o = a.dbg + ((STATS ? 0 : 1024) + blockIdx.y * gridDim.x + blockIdx.x) * 4; o[0] = T0; o[1] = T1; o[2] = T2; o[3] = T3; }
  if (STATS) {   // waves summed in a fixed order: deterministic
    __syncthreads();
    for (int c = tid; c < HD; c += nthr) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += aux[w * HD + c];
      a.ws[((long)n * gridDim.x + blockIdx.x) * HD + c] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------ packing
template <typename T>
__global__ void cnx_mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1v, const float* __restrict__ w2,
                                    const float* __restrict__ b2v, int C, int KC1, char* __restrict__ out) {
  constexpr int PE = Piece<T>::PE, BPC = PE / 4;
  const int HD = 4 * C, NB1 = 8 * KC1, KC2 = 4 * KC1, NB2 = 2 * KC1;
  T* o1 = (T*)out;
  T* o2 = (T*)(out + (size_t)KC1 * NB1 * 1024);
  float* ob1 = (float*)(out + (size_t)KC1 * NB1 * 1024 + (size_t)KC2 * NB2 * 1024);
  float* ob2 = ob1 + HD;
  const long n1 = (long)KC1 * NB1 * 64 * PE, n2 = (long)KC2 * NB2 * 64 * PE;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n1 + n2 + HD + NB2 * 16; i += (long)gridDim.x * blockDim.x) {
    if (i < n1) {           // pwconv1: rows = hidden channel nb*16 + r, K piece = input channels (kc*4+g)*PE + e
      const int e = (int)(i % PE); long t = i / PE;
      const int lane = (int)(t % 64); t /= 64;
      const int nb = (int)(t % NB1), kc = (int)(t / NB1);
      const int cin = (kc * 4 + (lane >> 4)) * PE + e, hid = nb * 16 + (lane & 15);
      o1[i] = (T)(cin < C ? w1[(long)hid * C + cin] : 0.f);
    } else if (i < n1 + n2) {   // pwconv2: rows = output channel, K slot (j, g, e) = hidden (j*BPC + e/4)*16 + 4g + e%4 (accumulator order)
      const long q = i - n1;
      const int e = (int)(q % PE); long t = q / PE;
      const int lane = (int)(t % 64); t /= 64;
      const int nb = (int)(t % NB2), j = (int)(t / NB2);
      const int hid = (j * BPC + e / 4) * 16 + 4 * (lane >> 4) + e % 4, co = nb * 16 + (lane & 15);
      o2[q] = (T)(co < C ? w2[(long)co * HD + hid] : 0.f);
    } else if (i < n1 + n2 + HD) {
      const int c = (int)(i - n1 - n2);
      ob1[c] = b1v ? b1v[c] : 0.f;
    } else {
      const int c = (int)(i - n1 - n2 - HD);
      ob2[c] = (b2v && c < C) ? b2v[c] : 0.f;
    }
  }
}

static int mlp_kc1(int c, int dtype) {   // 0: shape not covered by the fused kernels (caller keeps the three-launch chain)
  if (dtype != MGDT_BF16 || c <= 0 || c % 32) return 0;
  const int kc1 = c / 32;
  if (kc1 > 3) return 0;                 // both weight panels must fit in LDS
  return kc1;
}

extern "C" size_t mgdt_cnx_mlp_packed_bytes(int c, int dtype) {
  const int kc1 = mlp_kc1(c, dtype);
  if (!kc1) return 0;
  return mlp_w1_bytes(kc1) + mlp_w2_bytes(kc1) + (size_t)(4 * c + 2 * kc1 * 16) * sizeof(float);
}

extern "C" int mgdt_cnx_mlp_pack(const float* w1, const float* b1, const float* w2, const float* b2, int c, void* packed, int dtype, mgdt_stream s) {
  const int kc1 = mlp_kc1(c, dtype);
  if (!kc1) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_mlp_pack: c=%d dtype=%d not covered (bf16, c in {32, 64, 96})", c, dtype);
  if (!w1 || !w2 || !packed) MGDT_FAIL(MGDT_BAD_ARG, "cnx_mlp_pack: null argument");
  cnx_mlp_pack_kernel<bf16><<<256, 256, 0, (hipStream_t)s>>>(w1, b1, w2, b2, c, kc1, (char*)packed);
  MGDT_CHECK_LAUNCH("cnx_mlp_pack");
  return MGDT_OK;
}

constexpr int MLP_MT = 1;
// workgroups per image: one wave tile per wave, but at least ~256 workgroups over the batch when the image has that many tiles
static int mlp_splits(int n, int hw) {
  const int tiles = cdiv(hw, 16 * MLP_MT);
  return std::max(cdiv(tiles, MLP_THREADS / 64), std::min(cdiv(256, n), tiles));
}

extern "C" size_t mgdt_cnx_mlp_workspace_bytes(int n, int h, int w, int c) {
  return (size_t)n * mlp_splits(n, h * w) * 4 * c * sizeof(float);
}

template <int KC1>
static int mlp_launch(MlpArgs& a, const float* gamma, const float* beta, float* ws, hipStream_t st) {
  constexpr int MT = MLP_MT, HD = 128 * KC1;
  const int splits = mlp_splits(a.N, a.HW);
  const size_t w1b = mlp_w1_bytes(KC1), w2b = mlp_w2_bytes(KC1);
  const size_t lds_b = (size_t)(HD + 2 * KC1 * 16 + 4) * sizeof(float);
  const size_t lds_s = w1b + (size_t)(MLP_THREADS / 64) * HD * sizeof(float) + lds_b, lds_a = w1b + w2b + (size_t)2 * HD * sizeof(float) + lds_b;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    for (const void* k : {(const void*)cnx_mlp_kernel<bf16, KC1, MT, true>, (const void*)cnx_mlp_kernel<bf16, KC1, MT, false>}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "cnx_mlp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    attr_set = true;
  }
  a.ws = ws;
  static unsigned long long* dbg_buf = nullptr;
  if (getenv("MGDT_MLP_DBG") && !dbg_buf) (void)hipMalloc((void**)&dbg_buf, 2048 * 32);
  a.dbg = dbg_buf;
  cnx_mlp_kernel<bf16, KC1, MT, true><<<dim3(splits, a.N), MLP_THREADS, lds_s, st>>>(a);
  a.gamma = gamma; a.shift = beta;
  cnx_mlp_kernel<bf16, KC1, MT, false><<<dim3(splits, a.N), MLP_THREADS, lds_a, st>>>(a);
  MGDT_CHECK_LAUNCH("cnx_mlp_fwd");
  if (dbg_buf) {
    static unsigned long long h[2048 * 4];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
    const int nwg = splits * a.N;
    for (int pass = 0; pass < 2; ++pass) {
      const unsigned long long* q = h + pass * 4096;
      unsigned long long t0 = ~0ull, t3 = 0, s0max = 0; double st_sum = 0, lp_sum = 0, st_max = 0, lp_max = 0;
      for (int i = 0; i < nwg && i < 1024; ++i) {
        t0 = std::min(t0, q[i * 4]); t3 = std::max(t3, q[i * 4 + 3]); s0max = std::max(s0max, q[i * 4]);
        st_sum += q[i * 4 + 1] - q[i * 4]; lp_sum += q[i * 4 + 2] - q[i * 4 + 1];
        st_max = std::max(st_max, (double)(q[i * 4 + 1] - q[i * 4])); lp_max = std::max(lp_max, (double)(q[i * 4 + 2] - q[i * 4 + 1]));
      }
      fprintf(stderr, "cnx_mlp %s x10ns: first start -> last end %llu, last start +%llu, stage avg %.0f max %.0f, loop avg %.0f max %.0f\n", pass ? "apply" : "stats",
              t3 - t0, s0max - t0, st_sum / nwg, st_max, lp_sum / nwg, lp_max);
    }
  }
  return MGDT_OK;
}

extern "C" int mgdt_cnx_mlp_fwd(const mgdt_view* t, const mgdt_view* res, const void* packed, const float* gamma, const float* beta, void* ws,
                                const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(t) || !view_ok(y) || !packed || !gamma || !beta || !ws) MGDT_FAIL(MGDT_BAD_ARG, "cnx_mlp: null/empty argument");
  const int kc1 = mlp_kc1(t->c, dtype);
  if (!kc1) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_mlp: c=%d dtype=%d not covered (bf16, c in {32, 64, 96})", t->c, dtype);
  const long sz = (long)dtype_size(dtype);
  auto same = [&](const mgdt_view* v) { return v->n == t->n && v->h == t->h && v->w == t->w && v->c == t->c; };
  if (!same(y) || (res && res->p && !same(res))) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_mlp: t, res and y must have one shape");
  MlpArgs a;
  memset(&a, 0, sizeof(a));
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes, int q) {
    if (!v || !v->p) return;
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz;
    if (v->sc != 1 || v->sw % q || v->sh % q || v->sn % q || (uintptr_t)v->p % (q * sz) || ext >= 0x7fffffffL || v->sh * sz >= (1L << 23) ||
        v->sw * sz >= (1L << 23)) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * sz); *sh = (int)(v->sh * sz); *sw = (int)(v->sw * sz); *bytes = (uint32_t)ext;
  };
  const char* yp = nullptr;
  bind(t, &a.t, &a.tsn, &a.tsh, &a.tsw, &a.t_bytes, 8);
  bind(res, &a.res, &a.rsn, &a.rsh, &a.rsw, &a.r_bytes, 4);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes, 4);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_mlp: views must be 16-byte aligned NHWC (sc == 1), < 2 GiB, row stride < 8 MiB");
  a.y = (char*)yp;
  a.packed = (const char*)packed;
  a.N = t->n; a.H = t->h; a.W = t->w; a.C = t->c; a.HW = t->h * t->w;
  a.tiles = cdiv(a.HW, 16 * MLP_MT);
  a.fd_w = make_fastdiv((uint32_t)t->w);
  hipStream_t st = (hipStream_t)s;
  switch (kc1) {
    case 1: return mlp_launch<1>(a, gamma, beta, (float*)ws, st);
    case 2: return mlp_launch<2>(a, gamma, beta, (float*)ws, st);
    default: return mlp_launch<3>(a, gamma, beta, (float*)ws, st);
  }
}

// ================================================================================================ MSPA point-wise chain
// MSPA_C2f's hierarchical front (reference nn/modules/block.py:250-259):  sp0 = conv0(x0), sp1 = conv1(sp0 + x1),
// sp2 = conv2(sp1 + x2) with x_i / sp_i channel slices of width wd of the block input / concat buffer, conv_i = 1x1 conv + BN +
// SiLU.  Three dependent launches over tiny channel counts become one: each sp_i stays in the MFMA accumulators, is rounded to
// the storage type (what the next launch would have read back), written to its concat slot and - re-interpreted as a B operand,
// the K order of the packed weights is permuted to the accumulator order - fed straight to the next GEMM.
// HBM traffic: 3 wd in + 3 wd out per pixel (the unfused chain reads sp0 / sp1 back: 5 wd in).
struct ChainArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const char* packed;
  int N, H, W, wd, act, M, HW, tiles;
  FastDiv fd_hw, fd_w;
};

constexpr int CHAIN_THREADS = 512;

template <typename T, int NBK, int MT>
__global__ __launch_bounds__(CHAIN_THREADS) void pw_chain3_kernel(const ChainArgs a) {
  typedef typename Piece<T>::frag frag;
  constexpr int PE = Piece<T>::PE, BPC = PE / 4, SZ = (int)sizeof(T);
  constexpr int KC = (NBK + BPC - 1) / BPC;
  constexpr int WWORDS = 3 * KC * NBK * 64;                      // 16-byte words of the three weight panels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  float* bl = (float*)(smem + (size_t)WWORDS * 16);               // bias [3][NBK*16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nw = CHAIN_THREADS / 64;
  const int r = lane & 15, g = lane >> 4;
  for (int i = tid; i < WWORDS; i += CHAIN_THREADS) ((uint4*)wl)[i] = ((const uint4*)a.packed)[i];
  for (int i = tid; i < 3 * NBK * 16; i += CHAIN_THREADS) bl[i] = ((const float*)(a.packed + (size_t)WWORDS * 16))[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const char* const wlane = wl + lane * 16;
  typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int raw2;
  typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int raw4;

  for (int tile = blockIdx.x * nw + wave; tile < a.tiles; tile += gridDim.x * nw) {
    int xo[MT], yo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = (tile * MT + mt) * 16 + r;
      const bool pv = m < a.M;
      const int mm = pv ? m : 0;
      const int n = (int)fdiv((uint32_t)mm, a.fd_hw), rem = mm - n * a.HW;
      const int oy = (int)fdiv((uint32_t)rem, a.fd_w), ox = rem - oy * a.W;
      xo[mt] = pv ? n * a.xsn + __mul24(oy, a.xsh) + __mul24(ox, a.xsw) : MGDT_OOB;
      yo[mt] = pv ? n * a.ysn + __mul24(oy, a.ysh) + __mul24(ox, a.ysw) : MGDT_OOB;
    }
    // every input slice of the tile is requested up front (accumulator order: lane (r, g) <- channels blk*16 + 4g .. +3 of pixel r)
    f32x4 X[3][NBK][MT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int blk = 0; blk < NBK; ++blk) {
        const int c = blk * 16 + 4 * g;
        const int dead = c >= a.wd ? MGDT_OOB : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) X[i][blk][mt] = bload4<T>(xrs, (uint32_t)(xo[mt] | dead) + (uint32_t)((i * a.wd + c) * SZ));
      }
    f32x4 prev[NBK][MT];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      frag Bf[KC][MT];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < PE; ++e) {
            const int blk = kc * BPC + e / 4;
            float v = 0.f;
            if (blk < NBK) v = X[i][blk][mt][e % 4] + (i ? prev[blk][mt][e % 4] : 0.f);
            Bf[kc][mt][e] = (T)v;
          }
#pragma unroll
      for (int ob = 0; ob < NBK; ++ob) {
        const f32x4 bias = *(const f32x4*)(bl + (i * NBK + ob) * 16 + 4 * g);
        const int c = ob * 16 + 4 * g;
        const int dead = c >= a.wd ? MGDT_OOB : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 acc = bias;
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) acc = mma(*(const frag*)(wlane + ((i * KC + kc) * NBK + ob) * 1024), Bf[kc][mt], acc);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = (float)(T)(a.act == MGDT_ACT_SILU ? acc[j] * fast_sigmoid(acc[j]) : act_apply(acc[j], a.act));
          prev[ob][mt] = acc;                                   // already rounded: what the next launch would have read back
          bstore4<T>(yrs, (uint32_t)(yo[mt] | dead) + (uint32_t)((i * a.wd + c) * SZ), acc);
        }
      }
    }
  }
}

// conv `idx` of the chain: BN folded, K in accumulator order, rows = output channels
template <typename T>
__global__ void pw_chain_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, int wd, int NBK, int KC, int idx, T* __restrict__ out) {
  constexpr int PE = Piece<T>::PE, BPC = PE / 4;
  const int total = KC * NBK * 64 * PE;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i % PE; int t = i / PE;
    const int lane = t % 64; t /= 64;
    const int ob = t % NBK, kc = t / NBK;
    const int cin = (kc * BPC + e / 4) * 16 + 4 * (lane >> 4) + e % 4, co = ob * 16 + (lane & 15);
    float v = 0.f;
    if (kc * BPC + e / 4 < NBK && cin < wd && co < wd) v = w[(long)co * wd + cin] * (scale ? scale[co] : 1.f);
    out[(long)idx * total + i] = (T)v;
  }
}

__global__ void fold_kernel(const float* cb, const float* g, const float* b, const float* mu, const float* var, float eps, int Cout, int Cpad,
                            float* scale, float* bias_out);   // conv_igemm.hip

static int chain_nbk(int wd, int dtype) {   // 0: not covered
  if (dtype != MGDT_BF16 || wd < 4 || wd % 4 || wd > 64) return 0;
  const int nbk = (wd + 15) / 16;
  return nbk == 3 ? 4 : nbk;
}
static size_t chain_wbytes(int nbk) { return (size_t)3 * ((nbk + 1) / 2) * nbk * 1024; }

extern "C" size_t mgdt_pw_chain_packed_bytes(int wd, int dtype) {
  const int nbk = chain_nbk(wd, dtype);
  if (!nbk) return 0;
  return chain_wbytes(nbk) + (size_t)(3 + 1) * nbk * 16 * sizeof(float);   // weights | bias[3] | BN scale scratch
}

extern "C" int mgdt_pw_chain_pack(int idx, const float* w, const float* cb, const float* g, const float* b, const float* mu, const float* var, float eps,
                                  int wd, int dtype, void* packed, mgdt_stream s) {
  const int nbk = chain_nbk(wd, dtype);
  if (!nbk || idx < 0 || idx > 2) MGDT_FAIL(MGDT_BAD_SHAPE, "pw_chain_pack: wd=%d dtype=%d idx=%d not covered (bf16, wd %% 4 == 0, wd <= 64)", wd, dtype, idx);
  if (!w || !packed) MGDT_FAIL(MGDT_BAD_ARG, "pw_chain_pack: null pointer");
  if ((g != nullptr) != (b != nullptr) || (g != nullptr) != (mu != nullptr) || (g != nullptr) != (var != nullptr))
    MGDT_FAIL(MGDT_BAD_ARG, "pw_chain_pack: BN arguments must be all present or all NULL");
  hipStream_t st = (hipStream_t)s;
  float* bias = (float*)((char*)packed + chain_wbytes(nbk)) + idx * nbk * 16;
  float* scale = (float*)((char*)packed + chain_wbytes(nbk)) + 3 * nbk * 16;
  fold_kernel<<<cdiv(nbk * 16, 64), 64, 0, st>>>(cb, g, b, mu, var, eps, wd, nbk * 16, scale, bias);
  pw_chain_pack_kernel<bf16><<<16, 256, 0, st>>>(w, g ? scale : nullptr, wd, nbk, (nbk + 1) / 2, idx, (bf16*)packed);
  MGDT_CHECK_LAUNCH("pw_chain_pack");
  return MGDT_OK;
}

template <int NBK, int MT>
static int chain_launch(ChainArgs& a, hipStream_t st) {
  a.tiles = cdiv(a.M, 16 * MT);
  const size_t lds = chain_wbytes(NBK) + (size_t)3 * NBK * 16 * sizeof(float);
  const int grid = std::min(cdiv(a.tiles, CHAIN_THREADS / 64), 2048);
  pw_chain3_kernel<bf16, NBK, MT><<<grid, CHAIN_THREADS, lds, st>>>(a);
  MGDT_CHECK_LAUNCH("pw_chain3_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_pw_chain3_fwd(const mgdt_view* x, const void* packed, int wd, int act, const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !packed) MGDT_FAIL(MGDT_BAD_ARG, "pw_chain3: null/empty argument");
  const int nbk = chain_nbk(wd, dtype);
  if (!nbk) MGDT_FAIL(MGDT_BAD_SHAPE, "pw_chain3: wd=%d dtype=%d not covered (bf16, wd %% 4 == 0, wd <= 64)", wd, dtype);
  if (x->c != 3 * wd || y->c != 3 * wd || x->n != y->n || x->h != y->h || x->w != y->w)
    MGDT_FAIL(MGDT_BAD_SHAPE, "pw_chain3: x and y must be N x H x W x 3*wd views");
  const long sz = (long)dtype_size(dtype);
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * sz;
    if (v->sc != 1 || v->sw % 4 || v->sh % 4 || v->sn % 4 || (uintptr_t)v->p % (4 * sz) || ext >= 0x7fffffffL || v->sh * sz >= (1L << 23) ||
        v->sw * sz >= (1L << 23)) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * sz); *sh = (int)(v->sh * sz); *sw = (int)(v->sw * sz); *bytes = (uint32_t)ext;
  };
  const char* yp = nullptr;
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "pw_chain3: views must be 8-byte aligned NHWC (sc == 1), < 2 GiB, row stride < 8 MiB");
  a.y = (char*)yp;
  a.packed = (const char*)packed;
  const long M = (long)x->n * x->h * x->w;
  if (M > 0x7fffffffL - (1 << 20)) MGDT_FAIL(MGDT_BAD_SHAPE, "pw_chain3: problem too large");
  a.N = x->n; a.H = x->h; a.W = x->w; a.wd = wd; a.act = act; a.M = (int)M; a.HW = x->h * x->w;
  a.fd_hw = make_fastdiv((uint32_t)a.HW); a.fd_w = make_fastdiv((uint32_t)x->w);
  hipStream_t st = (hipStream_t)s;
  switch (nbk) {
    case 1: return chain_launch<1, 4>(a, st);
    case 2: return chain_launch<2, 4>(a, st);
    default: return chain_launch<4, 2>(a, st);
  }
}
