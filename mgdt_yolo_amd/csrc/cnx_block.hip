// A whole ConvNeXtV2 block (reference nn/modules/convnextv2.py:48-77; LayerNorm / GRN nn/modules/utils.py:145-182) in ONE launch, bf16 inference:
//
//   A  dw 7x7 + bias + LayerNorm(eps 1e-6) of a TH x TW tile (+3 halo) from LDS  -> t (bf16) stays in LDS
//   B  pwconv1 (C -> 4C) + GELU on MFMA, the 4C-wide hidden tile stays in REGISTERS (bf16, accumulator order = pwconv2's B-operand order);
//      per-tile sums of h^2 -> global; the workgroups of one image meet at a per-image barrier (GRN needs sum_hw h^2 of the whole image)
//   C  GRN scale from the image's sums, pwconv2 (4C -> C) on MFMA straight from the registers, + residual, store
//
// The three-launch form (mgdt_dwconv7_ln_fwd + the STATS / APPLY passes of mgdt_cnx_mlp_fwd) computed pwconv1 + GELU twice (GELU is what
// bounds those passes), wrote and re-read the normalised map twice, staged the weight panels 2 x 256 times and paid three launch
// fills / drains per block: 71 us per block at B = 32, 40x40, C = 96 (the IFM has three).  Here GELU runs once, nothing but x and y touches
// HBM, and each panel is staged once per workgroup while other work runs.
//
// Why a barrier between workgroups is safe here: the grid never exceeds the number of workgroups the chip holds at once (host side:
// images per launch = 256 / tiles per image, one workgroup per CU by LDS size), workgroups are dispatched in index order, and a waiting
// workgroup waits only for workgroups with the same image index.  The barrier is an arrival counter + a generation word per image and is
// back in its rest state when the kernel ends, so it can be replayed from a hipGraph (and called with other shapes) without resetting
// anything.  What crosses workgroups (the partial sums, the counter, the generation) moves through agent-scope atomics only - no
// cache-wide release / acquire fence (see the exchange below).
//
// LDS (C = 96, 10x20 tile: 143 KiB, one workgroup of 16 waves per CU):
//   U   80 KiB  halo tile (A)  ->  LayerNorm partial sums (A)  ->  pwconv1 panel (B)  ->  pwconv2 panel (C)
//   D   18 KiB  dw weights [49][C] fp32 (A)  ->  b1 | b2 | GRN scale | GRN shift (B, C)
//   T   42 KiB  t tile [pixel][C] bf16, pixel stride 2C + 16 B (A -> B)  ->  per-wave h^2 sums [16][4C] (B)
#include <algorithm>
#include <vector>

#include "conv_igemm_kernel.h"
#include "mlp_common.h"

struct CnxArgs {
  const char* x; int xsn, xsh, xsw; uint32_t x_bytes;   // strides / extents in bytes
  char* y; int ysn, ysh, ysw; uint32_t y_bytes;
  const float* dww; const float* dwb; const float* lnw; const float* lnb; float eps;
  const char* packed;                                   // W1 | W2 | b1 | b2 (mgdt_cnx_mlp_pack)
  const float* gamma; const float* beta;
  float* part;                                          // [N][tiles][4C] per-tile sums of h^2
  unsigned* sync;                                       // [N][2] arrival counter (0 at rest), generation
  int N, H, W, C, TH, TW, RH, RW, SEGS, tiles_x, tiles, n0;
  FastDiv fd_rw, fd_tw;
  GeluCoef gelu;
  const char* w3; const float* b3; int act3, C3;         // TAIL: a 1x1 Conv + BN + act on the block's output (IFM's closing conv), panel with K in accumulator order
  unsigned long long* dbg;
};

constexpr int CNX_THREADS = 1024;
constexpr int CNX_PX = 5;               // output pixels per thread along a row in the depth-wise phase
constexpr int CNX_MAXST = 6;            // 16-byte halo pieces per thread (host-checked)

struct CnxLds { int off_d, off_t, off_s, total; };
static inline CnxLds cnx_lds(int c, int th, int tw) {
  const int kc1 = c / 32, hd = 4 * c;
  const int segs = (tw + CNX_PX - 1) / CNX_PX, rw = segs * CNX_PX + 6, rh = th + 6;
  const int npixa = ((th * tw + 15) / 16) * 16;
  const int u = std::max({rh * rw * c * 2, kc1 * 8 * kc1 * 1024, 4 * kc1 * 2 * kc1 * 1024, th * tw * (c / 4) * 4});
  const int d = std::max(49 * c * 4, (3 * hd + 2 * kc1 * 16 + 16) * 4);
  const int t = std::max(npixa * (2 * c + 16), 16 * hd * 4);
  CnxLds l;
  l.off_d = (u + 15) & ~15; l.off_t = l.off_d + ((d + 15) & ~15); l.off_s = l.off_t + ((t + 15) & ~15);
  l.total = l.off_s + npixa * 4 + 64;
  return l;
}

template <int KC1, bool TAIL>
__global__ __launch_bounds__(CNX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void cnx_block_kernel(const CnxArgs a, const CnxLds L) {
  constexpr int C = 32 * KC1, Q = C / 4, PCS = C / 8, HD = 4 * C;
  constexpr int NB1 = 8 * KC1, KC2 = 4 * KC1, NB2 = 2 * KC1, BPC = 2;
  constexpr int TS_ = 2 * C + 16;                          // pixel stride of the t tile in LDS (bytes): conflict-free ds_read_b128
  constexpr int PX = CNX_PX;
  constexpr int W1V = KC1 * NB1 * 64, W2V = KC2 * NB2 * 64;      // 16-byte words of the two panels
  constexpr int F1 = (W1V + CNX_THREADS - 1) / CNX_THREADS, F2 = (W2V + CNX_THREADS - 1) / CNX_THREADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* U = smem;
  float* D = (float*)(smem + L.off_d);
  char* Tb = smem + L.off_t;
  float* stat = (float*)(smem + L.off_s);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n = a.n0 + (int)blockIdx.x / a.tiles, tile = (int)blockIdx.x % a.tiles;
  const int ty0 = (tile / a.tiles_x) * a.TH, tx0 = (tile % a.tiles_x) * a.TW;
  const int NPIX = a.TH * a.TW, NWT = (NPIX + 15) >> 4;
  const bool dbg = a.dbg && tid == 0;
  unsigned long long TT[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (dbg) TT[0] = wall_clock64();

  // this image's barrier generation, read long before this workgroup arrives at the barrier (it cannot change until every workgroup of the
  // image - this one included - has arrived)
  unsigned gen0 = 0;
  if (tid == 0) gen0 = __hip_atomic_load(a.sync + 2 * n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);

  // ================================================================ A: halo tile -> LDS, depth-wise 7x7, LayerNorm
  {
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int raw4;
    const int total = a.RH * a.RW * PCS;
    raw4 v[CNX_MAXST];
#pragma unroll
    for (int i = 0; i < CNX_MAXST; ++i) {                 // all requests of a thread in flight before the first LDS store waits on one
      const int idx = tid + i * CNX_THREADS;
      const int pix = idx / PCS, pc = idx - pix * PCS;
      const int hy = (int)fdiv((uint32_t)pix, a.fd_rw), hx = pix - hy * a.RW;
      const int iy = ty0 + hy - 3, ix = tx0 + hx - 3;
      const bool ok = idx < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)(n * a.xsn + iy * a.xsh + ix * a.xsw + pc * 16) : (uint32_t)MGDT_OOB;
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);      // zeros outside the image = the conv's zero padding
    }
    for (int i = tid; i < 49 * Q; i += CNX_THREADS) *(f32x4*)(D + i * 4) = *(const f32x4*)(a.dww + i * 4);
#pragma unroll
    for (int i = 0; i < CNX_MAXST; ++i) {
      const int idx = tid + i * CNX_THREADS;
      if (idx < total) ((raw4*)U)[idx] = v[i];
    }
  }
  __syncthreads();
  if (dbg) TT[1] = wall_clock64();

  // thread = (channel quad q, tile row, segment of PX pixels): the 7 x (PX + 6) window slides through registers
  const int q = tid % Q, rs_ = tid / Q, seg = rs_ % a.SEGS, row = rs_ / a.SEGS, px0 = seg * PX;
  const bool actA = row < a.TH;
  f32x4 acc[PX];
  {
    const f32x4 bq = *(const f32x4*)(a.dwb + q * 4);
#pragma unroll
    for (int k = 0; k < PX; ++k) acc[k] = bq;
  }
  if (actA) {
    const bf16* halo = (const bf16*)U;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      const bf16* hrow = halo + ((long)(row + ky) * a.RW + px0) * C + q * 4;
      f32x4 in[PX + 6];
#pragma unroll
      for (int i = 0; i < PX + 6; ++i) in[i] = load4<bf16>(hrow + i * C);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *(const f32x4*)(D + (ky * 7 + kx) * C + q * 4);
#pragma unroll
        for (int k = 0; k < PX; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(in[k + kx][j], wv[j], acc[k][j]);
      }
    }
  }
  // pwconv1's panel is requested now and lands while LayerNorm runs
  uint4 w1r[F1];
#pragma unroll
  for (int i = 0; i < F1; ++i) {
    const int idx = tid + i * CNX_THREADS;
    w1r[i] = idx < W1V ? ((const uint4*)a.packed)[idx] : make_uint4(0, 0, 0, 0);
  }
  __syncthreads();                                          // every thread is done with the halo and the dw weights
  if (dbg) TT[2] = wall_clock64();
  {
    // LayerNorm over the C channels of a pixel (two-pass form of F.layer_norm): per-(pixel, quad) partials in U, 24 of them per pixel
    float* red = (float*)U;
    if (actA) {
#pragma unroll
      for (int k = 0; k < PX; ++k)
        if (px0 + k < a.TW) red[(row * a.TW + px0 + k) * Q + q] = acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    }
    __syncthreads();
    if (tid < NPIX) {
      float m = 0.f;
      for (int j = 0; j < Q; ++j) m += red[tid * Q + j];
      stat[tid] = m / (float)C;
    }
    __syncthreads();
    float mean[PX];
    if (actA) {
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        if (px0 + k < a.TW) {
          mean[k] = stat[row * a.TW + px0 + k];
          const f32x4 d = acc[k] - mean[k];
          red[(row * a.TW + px0 + k) * Q + q] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
      }
    }
    __syncthreads();
    if (tid < NPIX) {
      float var = 0.f;
      for (int j = 0; j < Q; ++j) var += red[tid * Q + j];
      stat[tid] = 1.f / sqrtf(var / (float)C + a.eps);
    }
    __syncthreads();
    if (actA) {
      const f32x4 gq = *(const f32x4*)(a.lnw + q * 4), bq2 = *(const f32x4*)(a.lnb + q * 4);
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        if (px0 + k < a.TW) {
          const int p = row * a.TW + px0 + k;
          const bool inimg = ty0 + row < a.H && tx0 + px0 + k < a.W;
          f32x4 tv = (acc[k] - mean[k]) * stat[p] * gq + bq2;
          if (!inimg) tv = f32x4{0.f, 0.f, 0.f, 0.f};
          store4<bf16>((bf16*)(Tb + p * TS_) + q * 4, tv);
        }
      }
    }
    // rows of the last (partial) 16-pixel group: finite data for the MFMA operand
    for (int i = tid; i < (NWT * 16 - NPIX) * PCS; i += CNX_THREADS)
      *(uint4*)(Tb + (NPIX + i / PCS) * TS_ + (i % PCS) * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();                                          // t complete; U (halo / partials) and D (dw weights) are free
  // ---- stage pwconv1's panel and the biases
#pragma unroll
  for (int i = 0; i < F1; ++i) {
    const int idx = tid + i * CNX_THREADS;
    if (idx < W1V) ((uint4*)U)[idx] = w1r[i];
  }
  float* const bl = D;                                     // b1[HD] | b2[NB2*16]
  float* const aux = D + HD + NB2 * 16;                    // scale[HD] | shift[HD] | mean
  {
    const float* b1g = (const float*)(a.packed + (size_t)(W1V + W2V) * 16);
    for (int i = tid; i < HD + NB2 * 16; i += CNX_THREADS) bl[i] = b1g[i];
  }
  // ================================================================ B: pwconv1 + GELU, hidden tile in registers, sums of h^2
  const int p = wave * 16 + r;                              // this lane's pixel of the wave's 16-pixel group
  const int py = (int)fdiv((uint32_t)(p < NPIX ? p : 0), a.fd_tw), pxx = (p < NPIX ? p : 0) - py * a.TW;
  const bool pv = wave < NWT && p < NPIX && ty0 + py < a.H && tx0 + pxx < a.W;
  bf16x8 P[KC1];
#pragma unroll
  for (int kc = 0; kc < KC1; ++kc) P[kc] = *(const bf16x8*)(Tb + (wave < NWT ? p : 0) * TS_ + (kc * 4 + g) * 16);
  __syncthreads();                                          // panel + biases staged; every wave holds its t fragments: T is free for the sums
  if (dbg) TT[3] = wall_clock64();
  const float pvf = pv ? 1.f : 0.f;
  bf16x8 Hr[KC2];                                           // the hidden tile: lane (r, g) holds rows (j*2 + e/4)*16 + 4g + e%4 of pixel r
  float* const red2 = (float*)Tb + wave * HD;
  if (wave < NWT) {
    const char* const w1lane = U + lane * 16;
    const float* b1 = bl;
#pragma unroll                                              // unrolled: Hr[j] must be a register, not a dynamically indexed (scratch) array
    for (int j = 0; j < KC2; ++j) {
      bf16x8 hv;
      float sq[8];                                          // h^2 of the chunk's 8 hidden rows at this lane's pixel
#pragma unroll
      for (int b = 0; b < BPC; ++b) {                       // one 16-row block at a time: four GELUs in flight keep the register count down
        f32x4 acc1 = *(const f32x4*)(b1 + (j * BPC + b) * 16 + 4 * g);
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc) acc1 = mma(*(const bf16x8*)(w1lane + (kc * NB1 + j * BPC + b) * 1024), P[kc], acc1);
        const f32x4 ge = gelu_fast4(acc1, a.gelu);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16 hb = (bf16)ge[i];                      // rounded as the stored hidden map would be
          hv[b * 4 + i] = hb;
          const float h = (float)hb;
          sq[b * 4 + i] = (h * pvf) * h;                      // pixels outside the map contribute nothing
        }
      }
      // sum over the 16 pixels (lanes r) of each of the 8 values: a reduce-scatter butterfly - at every level a lane hands half of its
      // values to its partner and adds the partner's other half (3 instructions per surviving value instead of a 4-step DPP scan per
      // value).  Partners: r^1, r^2 (DPP quad permutes), r^4, r^8 (ds_swizzle).
      {
        const bool b0 = r & 1, b1_ = r & 2, b2 = r & 4;
        float t4[4], t2[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // level 1: bit 0 = 0 keeps values 0..3, bit 0 = 1 keeps 4..7
          const float send = b0 ? sq[k] : sq[k + 4], mine = b0 ? sq[k + 4] : sq[k];
          t4[k] = mine + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {                       // level 2: bit 1
          const float send = b1_ ? t4[k] : t4[k + 2], mine = b1_ ? t4[k + 2] : t4[k];
          t2[k] = mine + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
        }
        const float send = b2 ? t2[0] : t2[1], mine = b2 ? t2[1] : t2[0];                                                                   // level 3: bit 2
        float t1 = mine + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F));                     // lane ^ 4
        t1 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, t1), 0x201F));                                   // lane ^ 8: both halves hold the total
        // this lane now holds the total of value v = 4*b0 + 2*b1 + b2 (and so does lane r ^ 8)
        const int v = (b0 ? 4 : 0) + (b1_ ? 2 : 0) + (b2 ? 1 : 0);
        if (r < 8) red2[(j * BPC + (v >> 2)) * 16 + 4 * g + (v & 3)] = t1;
      }
      Hr[j] = hv;
    }
  }
  // pwconv2's panel: requested as soon as the hidden tile is done (registers are scarce while pwconv1 runs), in flight across the barrier
  uint4 w2r[F2];
  {
    const uint4* w2g = (const uint4*)(a.packed + (size_t)W1V * 16);
#pragma unroll
    for (int i = 0; i < F2; ++i) {
      const int idx = tid + i * CNX_THREADS;
      w2r[i] = idx < W2V ? w2g[idx] : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();                                          // all pwconv1 MFMAs done: U is free; the per-wave sums are complete
  if (dbg) TT[4] = wall_clock64();
#pragma unroll
  for (int i = 0; i < F2; ++i) {
    const int idx = tid + i * CNX_THREADS;
    if (idx < W2V) ((uint4*)U)[idx] = w2r[i];
  }
  // Inter-workgroup exchange WITHOUT cache-wide fences: an agent-scope release / acquire fence writes back / invalidates the whole L2 of the
  // XCD (measured: 70 us per fence with 256 workgroups doing it).  Instead every access that another workgroup must see is itself an
  // agent-scope atomic, which the hardware performs at the device-coherent level: the partial sums are published with a RETURNING atomic
  // exchange (complete when the value comes back), the workgroup barrier orders them before thread 0's arrival increment, the pollers
  // read the counter and - after their own barrier - the sums with relaxed agent-scope atomic loads.
  if (tid < HD) {
    float s = 0.f;
    for (int w = 0; w < NWT; ++w) s += ((const float*)Tb)[w * HD + tid];      // fixed order: deterministic
    const unsigned prev = __hip_atomic_exchange((unsigned*)(a.part + ((size_t)n * a.tiles + tile) * HD + tid), __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(prev));                           // the returned value is waited for: the exchange has been performed
  }
  __syncthreads();
  if (TAIL) {                                               // the per-wave sums in T have been read: the closing conv's panel goes there
    constexpr int W3V = (NB2 / 2) * NB2 * 64;               // 16-byte words: KC3 = NB2 / 2 chunks x NB2 cout blocks
    for (int i = tid; i < W3V; i += CNX_THREADS) ((uint4*)Tb)[i] = ((const uint4*)a.w3)[i];
  }
  if (tid == 0) {
    // arrival counter + generation: the last of the image's `tiles` workgroups resets the counter and opens the next generation; the others
    // wait for the generation they read BEFORE arriving to change.  The pair is back in its rest state (counter 0) when the kernel ends,
    // whatever the tile count was, so one workspace serves every shape and hipGraph replays need no reset.
    const unsigned old = __hip_atomic_fetch_add(a.sync + 2 * n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)a.tiles - 1u) {
      __hip_atomic_store(a.sync + 2 * n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(a.sync + 2 * n + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(a.sync + 2 * n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (dbg) TT[5] = wall_clock64();
  // the residual (x at this lane's pixel, the channels it will store) is requested now and lands under the GRN prologue
  const int oy_ = ty0 + py, ox_ = tx0 + pxx;
  const int xo = pv ? n * a.xsn + oy_ * a.xsh + ox_ * a.xsw : MGDT_OOB;
  const int yo = pv ? n * a.ysn + oy_ * a.ysh + ox_ * a.ysw : MGDT_OOB;
  typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int raw2;
  raw2 resr[NB2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    const int cob = (nb * 16 + 4 * g) * 2;
    resr[nb] = __builtin_amdgcn_raw_buffer_load_b64(xrs, cob >= C * 2 ? (uint32_t)MGDT_OOB : (uint32_t)xo + cob, 0, 0);
  }
  // GRN (nn/modules/utils GRN): Gx = ||h||_2 over (H, W) per channel, Nx = Gx / (mean_c Gx + 1e-6); y = gamma * (h * Nx) + beta + h
  float gsum = 0.f;
  if (tid < HD) {
    // all tiles' partial sums requested at once, added in tile order (the loop form waited for each device-coherent load in turn: ~1 us apiece)
    const unsigned* pp = (const unsigned*)(a.part + (size_t)n * a.tiles * HD + tid);
    if (a.tiles <= 16) {
      float pv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) pv[t] = t < a.tiles ? __uint_as_float(__hip_atomic_load(pp + (size_t)t * HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) gsum += pv[t];           // + 0.0f for the tiles that do not exist: the same sum, bit for bit
    } else {
      for (int t = 0; t < a.tiles; ++t) gsum += __uint_as_float(__hip_atomic_load(pp + (size_t)t * HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  if (tid < HD) aux[HD + tid] = sqrtf(gsum);
  __syncthreads();
  if (wave == 0) {                                          // mean over channels: strided partials, then a butterfly - one fixed order
    float part = 0.f;
    for (int c = lane; c < HD; c += 64) part += aux[HD + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) aux[2 * HD] = part / (float)HD;
  }
  __syncthreads();
  {
    const float mean = aux[2 * HD];
    __syncthreads();
    if (tid < HD) { aux[tid] = a.gamma[tid] * (sqrtf(gsum) / (mean + 1e-6f)) + 1.f; aux[HD + tid] = a.beta[tid]; }
  }
  __syncthreads();
  // ================================================================ C: pwconv2 from the registers, + residual, store
  if (wave < NWT) {
    const char* const w2lane = U + lane * 16;
    const float* b2 = bl + HD;
    f32x4 acc2[NB2];
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) acc2[nb] = *(const f32x4*)(b2 + nb * 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < KC2; ++j) {
      bf16x8 B2;
#pragma unroll
      for (int b = 0; b < BPC; ++b) {
        const f32x4 sc = *(const f32x4*)(aux + (j * BPC + b) * 16 + 4 * g), sh = *(const f32x4*)(aux + HD + (j * BPC + b) * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) B2[b * 4 + i] = (bf16)((float)Hr[j][b * 4 + i] * sc[i] + sh[i]);
      }
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) acc2[nb] = mma(*(const bf16x8*)(w2lane + (j * NB2 + nb) * 1024), B2, acc2[nb]);
    }
    if (!TAIL) {
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        const int cob = (nb * 16 + 4 * g) * 2;
        const int dead = cob >= C * 2 ? MGDT_OOB : 0;
        const bf16x4 rb = __builtin_bit_cast(bf16x4, resr[nb]);
        const f32x4 v = acc2[nb] + f32x4{(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
        bstore4<bf16>(yrs, (uint32_t)(yo | dead) + cob, v);
      }
    } else {
      // the block's output (rounded to bf16 as the stored map would be) is the next conv's B operand: chunk j3 = output blocks 2*j3, 2*j3 + 1
      constexpr int KC3 = NB2 / 2, NB3 = NB2;
      const char* const w3lane = Tb + lane * 16;
      f32x4 acc3[NB3];
#pragma unroll
      for (int nb = 0; nb < NB3; ++nb) acc3[nb] = *(const f32x4*)(a.b3 + nb * 16 + 4 * g);
#pragma unroll
      for (int j3 = 0; j3 < KC3; ++j3) {
        bf16x8 B3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x4 rb = __builtin_bit_cast(bf16x4, resr[2 * j3 + h]);
#pragma unroll
          for (int i = 0; i < 4; ++i) B3[h * 4 + i] = (bf16)(acc2[2 * j3 + h][i] + (float)rb[i]);
        }
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) acc3[nb] = mma(*(const bf16x8*)(w3lane + (j3 * NB3 + nb) * 1024), B3, acc3[nb]);
      }
#pragma unroll
      for (int nb = 0; nb < NB3; ++nb) {
        const int co = nb * 16 + 4 * g;
        f32x4 v = acc3[nb];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a.act3 == MGDT_ACT_SILU ? v[i] * fast_sigmoid(v[i]) : act_apply(v[i], a.act3);
        bstore4<bf16>(yrs, co < a.C3 ? (uint32_t)yo + (uint32_t)(co * 2) : (uint32_t)MGDT_OOB, v);
      }
    }
  }
  if (dbg) { TT[6] = wall_clock64(); for (int i = 0; i < 7; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = TT[i]; }
}

// ------------------------------------------------------------------------------------------------ host side
static int cnx_kc1(int c, int dtype) { return (dtype == MGDT_BF16 && c > 0 && c % 32 == 0 && c / 32 <= 3) ? c / 32 : 0; }

// tile of the fused kernel for an H x W map of C channels, or false: few tiles per image (every tile is a workgroup that must be resident
// together with all others of its launch), at most 16 pixel groups (one per wave), the depth-wise thread grid within 1024 threads, LDS
static bool cnx_pick_tile(int h, int w, int c, int* th_, int* tw_) {
  const int q = c / 4;
  long best = -1;
  for (int th = 2; th <= 16; ++th)
    for (int tw = CNX_PX; tw <= 40; ++tw) {
      if (th * tw > 256) continue;
      const int segs = (tw + CNX_PX - 1) / CNX_PX, rw = segs * CNX_PX + 6, rh = th + 6;
      if (q * th * segs > CNX_THREADS || (long)rh * rw * (c / 8) > (long)CNX_MAXST * CNX_THREADS) continue;
      if (cnx_lds(c, th, tw).total > 160 * 1024 - 512) continue;
      const long tiles = (long)cdiv(h, th) * cdiv(w, tw);
      const long waste = tiles * th * tw - (long)h * w;                 // pixels computed outside the map
      const long cost = tiles * 100000 + waste * 16 + (256 - th * tw) + (long)rh * rw;      // fewest workgroups, then least overhang, then the smallest halo region
      if (best < 0 || cost < best) { best = cost; *th_ = th; *tw_ = tw; }
    }
  return best >= 0;
}

extern "C" int mgdt_cnx_block_supported(int n, int h, int w, int c, int dtype) {
  int th, tw;
  if (!cnx_kc1(c, dtype) || n < 1 || h < 1 || w < 1 || !cnx_pick_tile(h, w, c, &th, &tw)) return 0;
  return cdiv(h, th) * cdiv(w, tw) <= 256;
}

/* workspace: [512 x {arrival counter, generation}, uint32: MUST BE ZERO at first use and never written by anyone else] [n * tiles * 4c floats] */
extern "C" size_t mgdt_cnx_block_workspace_bytes(int n, int h, int w, int c) {
  int th, tw;
  if (!cnx_pick_tile(h, w, c, &th, &tw)) return 0;
  return 4096 + (size_t)n * cdiv(h, th) * cdiv(w, tw) * 4 * c * sizeof(float);
}

template <int KC1, bool TAIL>
static int cnx_launch(CnxArgs& a, const CnxLds& L, hipStream_t st) {
  static std::atomic<bool> attr{false};
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)cnx_block_kernel<KC1, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "cnx_block: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  // every workgroup of a launch must be resident at once (per-image barrier): one workgroup per CU (LDS), 256 CUs
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "cnx_block: device query failed");
    cus = pr.multiProcessorCount;
  }
  const int per_launch = cus / a.tiles;                    // images per launch
  if (per_launch < 1) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: %d tiles per image exceed the %d compute units", a.tiles, cus);
  const int N = a.N;
  for (int n0 = 0; n0 < N; n0 += per_launch) {
    a.n0 = n0;
    cnx_block_kernel<KC1, TAIL><<<std::min(per_launch, N - n0) * a.tiles, CNX_THREADS, L.total, st>>>(a, L);
  }
  return MGDT_OK;
}

extern "C" int mgdt_cnx_block_fwd(const mgdt_view* x, const float* dw_w49c, const float* dw_b, const float* ln_w, const float* ln_b, float eps, const void* packed,
                                  const float* gamma, const float* beta, const void* tail_w, const float* tail_b, int tail_act, void* ws, size_t ws_bytes,
                                  const mgdt_view* y, int dtype, mgdt_stream s) {
  if (!view_ok(x) || !view_ok(y) || !dw_w49c || !dw_b || !ln_w || !ln_b || !packed || !gamma || !beta || !ws) MGDT_FAIL(MGDT_BAD_ARG, "cnx_block: null/empty argument");
  const int kc1 = cnx_kc1(x->c, dtype);
  if (!kc1 || !mgdt_cnx_block_supported(x->n, x->h, x->w, x->c, dtype)) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: n=%d %dx%d c=%d dtype=%d not covered", x->n, x->h, x->w, x->c, dtype);
  const bool tail = tail_w != nullptr;
  if (y->n != x->n || y->h != x->h || y->w != x->w || (!tail && y->c != x->c) || (tail && (!tail_b || y->c > x->c || y->c % 4)))
    MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: x and y must have one shape (with a closing conv: its output channels <= c, %% 4)");
  if (x->n > 512) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: at most 512 images per call");
  if (ws_bytes < mgdt_cnx_block_workspace_bytes(x->n, x->h, x->w, x->c) || (uintptr_t)ws % 16) MGDT_FAIL(MGDT_WORKSPACE, "cnx_block: workspace too small / unaligned");
  CnxArgs a;
  memset(&a, 0, sizeof(a));
  bool fits = true;
  auto bind = [&](const mgdt_view* v, const char** p, int* sn, int* sh, int* sw, uint32_t* bytes) {
    const long ext = ((long)(v->n - 1) * v->sn + (long)(v->h - 1) * v->sh + (long)(v->w - 1) * v->sw + v->c) * 2;
    if (v->sc != 1 || v->sw % 8 || v->sh % 8 || v->sn % 8 || (uintptr_t)v->p % 16 || ext >= 0x7fffffffL) { fits = false; return; }
    *p = (const char*)v->p; *sn = (int)(v->sn * 2); *sh = (int)(v->sh * 2); *sw = (int)(v->sw * 2); *bytes = (uint32_t)ext;
  };
  const char* yp = nullptr;
  bind(x, &a.x, &a.xsn, &a.xsh, &a.xsw, &a.x_bytes);
  bind(y, &yp, &a.ysn, &a.ysh, &a.ysw, &a.y_bytes);
  if (!fits) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: views must be 16-byte aligned NHWC (sc == 1) below 2 GiB");
  a.y = (char*)yp;
  a.dww = dw_w49c; a.dwb = dw_b; a.lnw = ln_w; a.lnb = ln_b; a.eps = eps; a.packed = (const char*)packed; a.gamma = gamma; a.beta = beta;
  a.sync = (unsigned*)ws; a.part = (float*)((char*)ws + 4096);
  a.w3 = (const char*)tail_w; a.b3 = tail_b; a.act3 = tail_act; a.C3 = y->c;
  a.N = x->n; a.H = x->h; a.W = x->w; a.C = x->c;
  if (!cnx_pick_tile(a.H, a.W, a.C, &a.TH, &a.TW)) MGDT_FAIL(MGDT_BAD_SHAPE, "cnx_block: no tile");
  a.SEGS = cdiv(a.TW, CNX_PX); a.RW = a.SEGS * CNX_PX + 6; a.RH = a.TH + 6;
  a.tiles_x = cdiv(a.W, a.TW); a.tiles = a.tiles_x * cdiv(a.H, a.TH);
  a.fd_rw = make_fastdiv((uint32_t)a.RW); a.fd_tw = make_fastdiv((uint32_t)a.TW);
  a.gelu = gelu_coef();
  const CnxLds L = cnx_lds(a.C, a.TH, a.TW);
  static unsigned long long* dbgbuf = nullptr;
  const bool dbg = getenv("MGDT_CNX_DBG") != nullptr;
  if (dbg && !dbgbuf) (void)hipMalloc((void**)&dbgbuf, (size_t)1024 * 8 * 8);
  a.dbg = dbg ? dbgbuf : nullptr;
  hipStream_t st = (hipStream_t)s;
  int rc;
  switch (kc1) {
    case 1: rc = tail ? cnx_launch<1, true>(a, L, st) : cnx_launch<1, false>(a, L, st); break;
    case 2: rc = tail ? cnx_launch<2, true>(a, L, st) : cnx_launch<2, false>(a, L, st); break;
    default: rc = tail ? cnx_launch<3, true>(a, L, st) : cnx_launch<3, false>(a, L, st); break;
  }
  if (rc != MGDT_OK) return rc;
  MGDT_CHECK_LAUNCH("cnx_block_fwd");
  if (dbg) {
    (void)hipStreamSynchronize(st);
    const int nwg = std::min(1024, std::min(a.N, 256 / a.tiles) * a.tiles);
    std::vector<unsigned long long> h((size_t)nwg * 8);
    (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t6 = 0;
    double ph[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nwg; ++i) {
      t0 = std::min(t0, h[(size_t)i * 8]); t6 = std::max(t6, h[(size_t)i * 8 + 6]);
      for (int k = 0; k < 6; ++k) ph[k] += (double)(h[(size_t)i * 8 + k + 1] - h[(size_t)i * 8 + k]);
    }
    fprintf(stderr, "cnx_block %dx%d c %d tile %dx%d (%d wgs of the last launch, lds %d): span %.1f us; avg per WG (us): stage %.2f dw7x7 %.2f layernorm+panel %.2f "
                    "pwconv1+gelu %.2f sums+image barrier %.2f grn+pwconv2+store %.2f\n", a.H, a.W, a.C, a.TH, a.TW, nwg, L.total, (t6 - t0) * 0.01,
            ph[0] / nwg * 0.01, ph[1] / nwg * 0.01, ph[2] / nwg * 0.01, ph[3] / nwg * 0.01, ph[4] / nwg * 0.01, ph[5] / nwg * 0.01);
  }
  return MGDT_OK;
}
