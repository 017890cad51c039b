// Device side of the implicit-GEMM convolution (see conv_igemm.hip for the design notes); included by the per-dtype
// instantiation units conv_igemm_inst_*.hip so that the 84 template instantiations compile in parallel.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

template <typename T> struct Piece;
template <> struct Piece<float> { static constexpr int PE = 4; typedef f32x4 frag; };
template <> struct Piece<bf16> { static constexpr int PE = 8; typedef bf16x8 frag; };

struct ConvArgs {   // all strides / offsets in BYTES and < 2 GiB (host-checked): the kernel addresses through buffer descriptors
  const char* x; int xsn, xsh, xsw;
  const char* x2; int x2sn, x2sh, x2sw;
  const float* in_scale; const float* in_shift;
  const char* wpk; const float* bias;
  char* y; int ysn, ysh, ysw;
  const char* r1; int r1sn, r1sh, r1sw;
  const char* r2; int r2sn, r2sh, r2sw;
  int N, H, W, Cin, Ho, Wo, Cout;
  int KS, stride, pad, CP, nchunks, NTtot, act;
  int M, HoWo, numTiles, T8, seg_chunks, nseg, tab_bytes;
  FastDiv fd_howo, fd_wo;
  uint32_t x_bytes, x2_bytes, y_bytes, r1_bytes, r2_bytes;   // addressable extents of the views (buffer descriptor ranges)
  int ntaps; unsigned long long taplist;                     // K enumerates these taps only (4 bits each; all KS*KS by default): the phase convs of a stride-2 data gradient use 1, 2 or 4 of the 9
  const float* oscale; float xq;                             // fp8 (Q8) kernels only: per-output-channel de-quantisation factor w_scale[co] / xq, activation multiplier before the e4m3 conversion
};

// ------------------------------------------------------------------------------------------------ device helpers
template <typename T> __device__ __forceinline__ typename Piece<T>::frag zero_frag();
template <> __device__ __forceinline__ f32x4 zero_frag<float>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ bf16x8 zero_frag<bf16>() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

template <typename T>
__device__ __forceinline__ typename Piece<T>::frag frag_add(typename Piece<T>::frag a, typename Piece<T>::frag b) {
  typename Piece<T>::frag o;
#pragma unroll
  for (int i = 0; i < Piece<T>::PE; ++i) o[i] = (T)((float)a[i] + (float)b[i]);
  return o;
}

template <typename T>
__device__ __forceinline__ typename Piece<T>::frag frag_affine(typename Piece<T>::frag a, const float* sc, const float* sh) {
  typename Piece<T>::frag o;
#pragma unroll
  for (int i = 0; i < Piece<T>::PE; ++i) {
    float v = (float)a[i];
    if (sc) v *= sc[i];
    if (sh) v += sh[i];
    o[i] = (T)v;
  }
  return o;
}

// 4 consecutive channels through a bounds-checked descriptor (out-of-range: loads give 0, stores are dropped)
template <typename T> __device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t rs, uint32_t off);
template <> __device__ __forceinline__ f32x4 bload4<float>(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
template <> __device__ __forceinline__ f32x4 bload4<bf16>(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  const bf16x4 o = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
  return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}
template <typename T> __device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t rs, uint32_t off, f32x4 v);
template <> __device__ __forceinline__ void bstore4<float>(__amdgpu_buffer_rsrc_t rs, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, off, 0, 0);
}
template <> __device__ __forceinline__ void bstore4<bf16>(__amdgpu_buffer_rsrc_t rs, uint32_t off, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, o), rs, off, 0, 0);
}

__device__ __forceinline__ f32x4 mma(f32x4 w, f32x4 p, f32x4 acc) {
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], p[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x4 mma(bf16x8 w, bf16x8 p, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, p, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ main kernel
// blockDim.x = 64 * nwaves (4, 8 or 16 waves); a workgroup covers nwaves*MT*16 output pixels x NT*16 output channels per
// tile and is PERSISTENT over tiles blockIdx.x + j*gridDim.x.  Per tile the K-chunks are padded to a multiple of D steps so
// that step j always uses register set j % D (compile-time indices).  At step j the loads of step j+D-1 are issued; in the
// tile's last group they belong to the NEXT tile, so its first loads are in flight while this tile finishes its MFMAs and
// runs its epilogue.  The step body is a handful of integer instructions: one 8-byte table read, per row block a bit test
// on the pixel's tap-validity mask + one add, a bounds-checked buffer load (zero fill for padding), NT ds_read_b128 at
// immediate offsets from a running LDS pointer, NT*MT MFMAs.
struct TileState { int xo; uint32_t vm; int yo, x2o, r1o, r2o, pn; };   // byte offsets; MGDT_OOB = "no such pixel"
#define MGDT_OOB ((int)0x80000000)   // >= every view extent, and stays out of range after a (small) channel offset is added

// Q8 (T = bf16 only): activations stay bf16 in HBM and are converted to e4m3 in registers (quant8), the weight panel holds e4m3 bytes
// (512-byte blocks), the accumulators are multiplied by oscale[cout] in the epilogue - BASELINE configs[4].
template <typename T, int NT, int MT, int D, bool EXTRA, bool MULTI, bool Q8 = false>
__global__ __launch_bounds__(512) void conv_igemm_kernel(const ConvArgs a) {
  typedef typename Piece<T>::frag frag;
  static_assert(!Q8 || std::is_same<T, bf16>::value, "the fp8 kernels read bf16 activations");
  constexpr int WB = Q8 ? 512 : 1024;  // bytes of one [chunk][16 couts] weight block
  constexpr int PE = Piece<T>::PE;
  constexpr int L = D - 1;             // look-ahead in (padded) steps
  constexpr int SZ = (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ptab = (uint4*)smem;          // per 16-byte K piece: {byte offset of (tap, channel) in the x view, tap, byte offset in the x2 view, channel}
  float* blds = (float*)(smem + a.tab_bytes);       // this workgroup's NT*16 bias values (accumulators start from them)
  float* olds = blds + NT * 16;                     // Q8: the matching de-quantisation factors
  char* wlds = smem + a.tab_bytes + (Q8 ? 2 : 1) * NT * 16 * sizeof(float);

  const int nthr = blockDim.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the tile arithmetic below runs on the scalar unit
  const int r = lane & 15, g = lane >> 4;
  const int nb0 = blockIdx.y * NT;
  const int BM = (nthr >> 6) * MT * 16;
  const int nchp = (a.nchunks + D - 1) / D * D;
  const int ssh = a.stride >> 1;       // stride is 1 or 2 (host-checked): multiply by shifting

  for (int p = tid; p < nchp * 4; p += nthr) {   // table padded to nchp chunks: padding pieces carry tap 31 (never valid)
    const int li = p / a.CP, cp = p % a.CP;
    uint4 e = make_uint4(0u, 31u, 0u, 0u);
    if (li < a.ntaps) {
      const int tap = (int)((a.taplist >> (4 * li)) & 15ull);
      const int dy = tap / a.KS, dx = tap % a.KS;
      e = make_uint4((uint32_t)(dy * a.xsh + dx * a.xsw + cp * PE * SZ), (uint32_t)tap, (uint32_t)(dy * a.x2sh + dx * a.x2sw + cp * PE * SZ), (uint32_t)(cp * PE));
    }
    ptab[p] = e;
  }
  if (tid < NT * 16) {
    if constexpr (Q8) {   // accumulators start from bias / oscale so that the epilogue's one multiply restores the bias
      const float os = a.oscale[blockIdx.y * NT * 16 + tid];
      olds[tid] = os;
      blds[tid] = a.bias[blockIdx.y * NT * 16 + tid] / os;
    } else {
      blds[tid] = a.bias[blockIdx.y * NT * 16 + tid];
    }
  }
  __syncthreads();

  auto stage = [&](int seg) __attribute__((always_inline)) {
    const int c0 = seg * a.seg_chunks;
    const int nc = min(a.seg_chunks, a.nchunks - c0);
    const int nblk = nc * NT;  // WB-byte blocks
    constexpr int V = WB / 16;
    // UB requests of a thread are in flight before its first LDS store (a plain copy loop waited for every load in turn: up to 18 L2 round trips in
    // front of a persistent workgroup's first tile for a 144 KiB panel)
    constexpr int UB = MULTI ? 2 : 8;
    const int total = nblk * V;
    for (int i0 = tid; i0 < total; i0 += UB * nthr) {
      uint4 t[UB];
      int dst[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = min(i0 + u * nthr, total - 1);
        const int blk = i / V, l = i % V;
        const int kc = blk / NT, nt = blk % NT;
        t[u] = *((const uint4*)(a.wpk + ((long)(c0 + kc) * a.NTtot + nb0 + nt) * WB) + l);
        dst[u] = i0 + u * nthr < total ? blk * (WB / 16) + l : -1;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (dst[u] >= 0) ((uint4*)wlds)[dst[u]] = t[u];
    }
  };
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? a.x2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.r1 ? a.r1 : a.x), 0, a.r1 ? a.r1_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.r2 ? a.r2 : a.x), 0, a.r2 ? a.r2_bytes : 0u, 0x00020000);

  // v = position in the launch-wide round-robin; workgroups land on XCD (id % 8), so XCD k is given the CONTIGUOUS tile range
  // [k*T8, (k+1)*T8): the halo rows a 3x3 tile shares with its neighbours are then served by that XCD's own L2.
  // The 16 pixels of a row block are consecutive output positions: their first pixel is decomposed on the scalar unit, the
  // lanes add their index and wrap (one row / one image at most when Wo >= 16), so no per-lane division or 32-bit multiply.
  auto setup = [&](int v, TileState(&S)[MT]) __attribute__((always_inline)) {
    const int tile = (v & 7) * a.T8 + (v >> 3);
    const bool tv = (v >> 3) < a.T8 && tile < a.numTiles;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m0 = tv ? tile * BM + (wave * MT + mt) * 16 : 0;   // uniform
      const bool pv = tv && m0 + r < a.M;
      int n, oy, ox, n0 = 0;
      bool bump = false;                                         // fast path: n = n0 (uniform) + bump
      const bool fast = a.Wo >= 16;
      if (fast) {
        n0 = (int)fdiv((uint32_t)m0, a.fd_howo);
        const int rem = m0 - n0 * a.HoWo;
        const int oy0 = (int)fdiv((uint32_t)rem, a.fd_wo), ox0 = rem - oy0 * a.Wo;
        oy = oy0; ox = ox0 + r;
        if (ox >= a.Wo) { ox -= a.Wo; ++oy; }
        if (oy >= a.Ho) { oy = 0; bump = true; }
        n = n0 + (bump ? 1 : 0);
      } else {
        const int mm = pv ? m0 + r : 0;
        n = (int)fdiv((uint32_t)mm, a.fd_howo);
        const int rem = mm - n * a.HoWo;
        oy = (int)fdiv((uint32_t)rem, a.fd_wo); ox = rem - oy * a.Wo;
      }
      auto nmul = [&](int s) __attribute__((always_inline)) { return fast ? n0 * s + (bump ? s : 0) : n * s; };   // n * s without a vector multiply
      TileState t;
      t.pn = n;
      const int iy0 = (oy << ssh) - a.pad, ix0 = (ox << ssh) - a.pad;
      t.xo = nmul(a.xsn) + __mul24(iy0, a.xsh) + __mul24(ix0, a.xsw);
      t.yo = pv ? nmul(a.ysn) + __mul24(oy, a.ysh) + __mul24(ox, a.ysw) : MGDT_OOB;
      t.x2o = t.r1o = t.r2o = 0;
      if (EXTRA && a.x2) t.x2o = nmul(a.x2sn) + __mul24(iy0, a.x2sh) + __mul24(ix0, a.x2sw);
      if (a.r1) t.r1o = pv ? nmul(a.r1sn) + __mul24(oy, a.r1sh) + __mul24(ox, a.r1sw) : MGDT_OOB;
      if (a.r2) t.r2o = pv ? nmul(a.r2sn) + __mul24(oy, a.r2sh) + __mul24(ox, a.r2sw) : MGDT_OOB;
      uint32_t mask = 1u;
      if (a.KS == 3) {
        const uint32_t rm = (uint32_t)((unsigned)iy0 < (unsigned)a.H) | ((uint32_t)((unsigned)(iy0 + 1) < (unsigned)a.H) << 1) |
                            ((uint32_t)((unsigned)(iy0 + 2) < (unsigned)a.H) << 2);
        const uint32_t cm = (uint32_t)((unsigned)ix0 < (unsigned)a.W) | ((uint32_t)((unsigned)(ix0 + 1) < (unsigned)a.W) << 1) |
                            ((uint32_t)((unsigned)(ix0 + 2) < (unsigned)a.W) << 2);
        mask = ((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 3 : 0u) | ((rm & 4u) ? cm << 6 : 0u);
      }
      t.vm = pv ? mask : 0u;   // bit t set <=> tap t of this pixel lies inside the image (0 for rows past M)
      S[mt] = t;
    }
  };

  // activation fragments of one K-chunk (table entry e): each lane fetches the 16 bytes it feeds to the MFMA.  Padding taps /
  // tail rows / padded chunks get an out-of-range offset and the hardware returns zeros: no branch, no select.
  auto load_chunk = [&](const uint4* ep, const TileState(&S)[MT], frag(&P)[MT]) __attribute__((always_inline)) {
    uint32_t ex, ey, ez = 0, ew = 0;
    if (EXTRA) { const uint4 e = *ep; ex = e.x; ey = e.y; ez = e.z; ew = e.w; }
    else { const uint2 e = *(const uint2*)ep; ex = e.x; ey = e.y; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = (S[mt].vm >> ey) & 1u;
      const uint32_t off = ok ? (uint32_t)S[mt].xo + ex : (uint32_t)MGDT_OOB;
      frag v = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
      if (EXTRA) {
        if (a.x2) {
          const uint32_t off2 = ok ? (uint32_t)S[mt].x2o + ez : (uint32_t)MGDT_OOB;
          v = frag_add<T>(v, __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(x2rs, off2, 0, 0)));
        }
        if (ok && (a.in_scale || a.in_shift))
          v = frag_affine<T>(v, a.in_scale ? a.in_scale + (long)S[mt].pn * a.Cin + ew : nullptr, a.in_shift ? a.in_shift + ew : nullptr);
      }
      P[mt] = v;
    }
  };

  f32x4 acc[NT][MT];
  auto init_acc = [&]() __attribute__((always_inline)) {   // accumulators start from the bias of the lane's 4 output channels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 b = *(const f32x4*)(blds + nt * 16 + 4 * g);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b;
    }
  };
  auto compute = [&](const char* wp, const frag(&P)[MT]) __attribute__((always_inline)) {   // wp: this lane's slot of the chunk's weight blocks
    if constexpr (Q8) {
      long Pq[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Pq[mt] = quant8(P[mt], a.xq);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const long Wq = *(const long*)(wp + nt * WB);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mma_q8(Wq, Pq[mt], acc[nt][mt]);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const frag Wf = *(const frag*)(wp + nt * WB);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mma(Wf, P[mt], acc[nt][mt]);
      }
    }
  };
  // lane holds couts (nb0+nt)*16 + 4g .. +3 of pixel (mt, r); stores / residual loads go through bounds-checked descriptors, so
  // tail pixels and padded output channels need no branch.  The activation switch is hoisted out of the loops.
  const bool ragged = (nb0 + NT) * 16 > a.Cout;   // this workgroup's last cout block is partly padding
  // bf16, NT >= 4: the four lanes (r, g = 0..3) of a pixel hold 4 couts (8 bytes) of each cout block; a 4x4 transpose over those lanes (two butterfly
  // stages of ds_bpermute: partner g ^ 1, then g ^ 2) leaves lane g with all 16 couts of block 4q + g, so a pixel's 4 blocks leave as 128 contiguous bytes
  // (two 16-byte stores per lane) instead of sixteen 8-byte pieces in four instructions - full-line writes for the output-heavy 1x1 convolutions.
  constexpr bool WIDE = std::is_same<T, bf16>::value && NT >= 4;
  constexpr int NTW = WIDE ? NT / 4 * 4 : 0;      // cout blocks stored through the transpose; the rest (NT = 5, 6) keep the 8-byte stores
  auto epilogue_act = [&](const TileState(&S)[MT], auto actf) __attribute__((always_inline)) {
    if constexpr (WIDE) {
      if (!ragged) {
        const bool odd = (g & 1) != 0, hi = (g & 2) != 0;
        const int pA = (lane ^ 16) << 2, pB = (lane ^ 32) << 2;
        auto xch = [&](int addr, uint2 v) __attribute__((always_inline)) {
          return make_uint2((unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.x), (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.y));
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int q = 0; q < NTW / 4; ++q) {
            uint2 a4[4], c4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nt = 4 * q + j;
              const int cob = ((nb0 + nt) * 16 + 4 * g) * SZ;
              f32x4 v = acc[nt][mt];
              if constexpr (Q8) v *= *(const f32x4*)(olds + nt * 16 + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = actf(v[e]);
              if (a.r1) v += bload4<T>(r1rs, (uint32_t)S[mt].r1o + cob);
              if (a.r2) v += bload4<T>(r2rs, (uint32_t)S[mt].r2o + cob);
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
              a4[j] = __builtin_bit_cast(uint2, o);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // stage A, partner g ^ 1: c4[2h + (source & 1)] = block 2h + (g & 1) of the two lanes of my half
              const uint2 x = xch(pA, odd ? a4[2 * h] : a4[2 * h + 1]);
              c4[2 * h] = odd ? x : a4[2 * h];
              c4[2 * h + 1] = odd ? a4[2 * h + 1] : x;
            }
            const uint2 x0 = xch(pB, hi ? c4[0] : c4[2]), x1 = xch(pB, hi ? c4[1] : c4[3]);    // stage B, partner g ^ 2
            const uint2 b0 = hi ? x0 : c4[0], b1 = hi ? x1 : c4[1], b2 = hi ? c4[2] : x0, b3 = hi ? c4[3] : x1;   // b[source lane g'] = block 4q + g of source g'
            const uint32_t off = (uint32_t)S[mt].yo + (uint32_t)(((nb0 + 4 * q + g) * 16) * SZ);
            typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4v;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v{b0.x, b0.y, b1.x, b1.y}, yrs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v{b2.x, b2.y, b3.x, b3.y}, yrs, off + 16u, 0, 0);
          }
        }
      }
    }
    const int nt_first = (WIDE && !ragged) ? NTW : 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < nt_first) continue;
        const int cob = ((nb0 + nt) * 16 + 4 * g) * SZ;
        const int dead = (ragged && cob >= a.Cout * SZ) ? MGDT_OOB : 0;
        f32x4 v = acc[nt][mt];
        if constexpr (Q8) v *= *(const f32x4*)(olds + nt * 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = actf(v[j]);
        if (a.r1) v += bload4<T>(r1rs, (uint32_t)(S[mt].r1o | dead) + cob);
        if (a.r2) v += bload4<T>(r2rs, (uint32_t)(S[mt].r2o | dead) + cob);
        bstore4<T>(yrs, (uint32_t)(S[mt].yo | dead) + cob, v);
      }
    }
  };
  auto epilogue = [&](const TileState(&S)[MT]) __attribute__((always_inline)) {
    switch (a.act) {
      case MGDT_ACT_SILU: epilogue_act(S, [](float v) { return v * fast_sigmoid(v); }); break;
      case MGDT_ACT_RELU: epilogue_act(S, [](float v) { return fmaxf(v, 0.f); }); break;
      case MGDT_ACT_GELU: epilogue_act(S, [](float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }); break;
      default: epilogue_act(S, [](float v) { return v; }); break;
    }
  };

  // ---- persistent stream over my tiles ----
  TileState cur[MT], nxt[MT];
  int tile = blockIdx.x;
  setup(tile, cur);
  frag P[D][MT];
  const uint4* tab_g = ptab + g;                 // this lane's column of the piece table
  load_chunk(tab_g, cur, P[0]);                  // first loads in flight BEFORE the weight panel is staged
  if constexpr (L > 1) load_chunk(tab_g + 4, cur, P[1]);
  if constexpr (L > 2) load_chunk(tab_g + 8, cur, P[2]);
  if (!MULTI) { stage(0); __syncthreads(); }
  const char* const wlane = wlds + lane * (WB / 64);
  constexpr bool multi = MULTI;   // weight panel staged in K segments (only when even one cout block does not fit in LDS)

  for (; tile < 8 * a.T8; tile += gridDim.x) {
    init_acc();
    const uint4* tp = tab_g + 4 * L;             // table entry of the chunk that step 0 prefetches
    const char* wp = wlane;                      // weight blocks of the chunk that step 0 computes
    int kl = 0;                                  // chunk index inside the staged weight segment (multi-segment panels only)
    // one step: prefetch (`ahead` = table entry of a later chunk of THIS tile), then the MFMAs of chunk j
    auto step_cur = [&](int j, auto dtag) __attribute__((always_inline)) {
      constexpr int d = decltype(dtag)::value;
      load_chunk(tp, cur, P[(d + L) % D]);
      tp += 4;
      if (multi && kl == 0) { __syncthreads(); stage(j / a.seg_chunks); __syncthreads(); wp = wlane; }
      compute(wp, P[d]);
      wp += NT * WB;
      if (multi && ++kl == a.seg_chunks) kl = 0;
    };
    int j0 = 0;
    for (; j0 < nchp - D; j0 += D) {             // every chunk computed here is real (nchp - D < nchunks)
      step_cur(j0, std::integral_constant<int, 0>{});
      step_cur(j0 + 1, std::integral_constant<int, 1>{});
      if constexpr (D > 2) step_cur(j0 + 2, std::integral_constant<int, 2>{});
      if constexpr (D > 3) step_cur(j0 + 3, std::integral_constant<int, 3>{});
    }
    // last group: step 0 prefetches this tile's final (possibly padded) chunk, steps d >= 1 the NEXT tile's chunk d-1
    setup(tile + gridDim.x, nxt);
    step_cur(j0, std::integral_constant<int, 0>{});
    auto step_last = [&](int j, auto dtag) __attribute__((always_inline)) {
      constexpr int d = decltype(dtag)::value;
      load_chunk(tab_g + 4 * (d - 1), nxt, P[(d + L) % D]);
      if (j < a.nchunks) {
        if (multi && kl == 0) { __syncthreads(); stage(j / a.seg_chunks); __syncthreads(); wp = wlane; }
        compute(wp, P[d]);
        wp += NT * WB;
        if (multi && ++kl == a.seg_chunks) kl = 0;
      }
    };
    step_last(j0 + 1, std::integral_constant<int, 1>{});
    if constexpr (D > 2) step_last(j0 + 2, std::integral_constant<int, 2>{});
    if constexpr (D > 3) step_last(j0 + 3, std::integral_constant<int, 3>{});
    epilogue(cur);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cur[mt] = nxt[mt];
  }
}

template <typename T, int NT, int MT, bool Q8 = false>
int launch_igemm(const ConvArgs& a, int gx, int gy, int threads, size_t lds, hipStream_t st) {
  static std::atomic<bool> attr_set{false};  // idempotent; racing setters write the same value
  constexpr bool M1 = NT == 1;   // the segmented-panel variant exists for NT == 1 only (host never asks for it otherwise)
  const void* ks[8] = {(const void*)conv_igemm_kernel<T, NT, MT, 2, false, false, Q8>, (const void*)conv_igemm_kernel<T, NT, MT, 4, false, false, Q8>,
                       (const void*)conv_igemm_kernel<T, NT, MT, 2, true, false, Q8>,  (const void*)conv_igemm_kernel<T, NT, MT, 4, true, false, Q8>,
                       (const void*)conv_igemm_kernel<T, NT, MT, 2, false, M1, Q8>,    (const void*)conv_igemm_kernel<T, NT, MT, 4, false, M1, Q8>,
                       (const void*)conv_igemm_kernel<T, NT, MT, 2, true, M1, Q8>,     (const void*)conv_igemm_kernel<T, NT, MT, 4, true, M1, Q8>};
  if (!attr_set) {
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    attr_set = true;
  }
  const int D = a.nchunks >= 3 ? 4 : 2;        // look-ahead D-1 <= nchunks
  const bool extra = a.x2 || a.in_scale || a.in_shift;
  void* kargs[] = {(void*)&a};
  hipError_t le = hipLaunchKernel(ks[(a.nseg > 1 ? 4 : 0) + (extra ? 2 : 0) + (D == 4 ? 1 : 0)], dim3(gx, gy), dim3(threads), kargs, lds, st);
  if (le != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "conv2d: launch: %s", hipGetErrorString(le));
  MGDT_CHECK_LAUNCH("conv2d_fwd");
  return MGDT_OK;
}


#define MGDT_IGEMM_INSTANTIATE(T, nt) template int launch_igemm<T, nt, 2>(const ConvArgs&, int, int, int, size_t, hipStream_t);
#define MGDT_IGEMM_INSTANTIATE_Q8(nt) template int launch_igemm<bf16, nt, 2, true>(const ConvArgs&, int, int, int, size_t, hipStream_t);
