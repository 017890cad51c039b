// Batched NMS, one workgroup per image, no host round trip (the reference loops over images in Python and
// calls ~10 small ops + torchvision.ops.nms per image: yolo/utils/ops.py:199-264).
//
//   1. candidate predicate straight from pred[n][4+nc][a] (score > conf, best-class or multi-label, class filter); the best class per
//      anchor comes from the Detect tail kernel when it ran (`best_keys`: it has every score in registers anyway), else from nms_best_kernel
//   2. the candidates are taken in SEGMENTS of descending score (<= 1024 keys first, then <= 4096, ...): a 2048-bin histogram of the
//      scores (one LDS pass + one scan) gives the score edge whose upper side holds at most that many keys - an exact top set, no
//      multi-pass radix select; the MSD radix select on the 64-bit key (descending score, ascending candidate id = anchor*nc + cls ->
//      tie rule "lower candidate index first") remains as the fallback when one bin alone overflows a segment or max_nms cuts inside one
//   3. bitonic sort of the segment's keys (register shuffles for <= 1024 keys, LDS up to 16384, else an L2-resident global buffer);
//      for <= 1024 keys every thread then fetches its candidate's box once and parks it in LDS
//   4. greedy suppression, 64 sorted candidates per step: every lane tests its box against the kept list (LDS broadcast reads),
//      survivors are resolved inside the chunk with ballot + scalar bit operations.
//      The scan stops at max_det kept boxes (identical to nms(...)[:max_det]).
// IoU arithmetic is the torchvision CPU kernel's, in IEEE fp32 with FP contraction off (this file is compiled
// with -ffp-contract=off) on boxes offset by cls*max_wh in fp32 exactly as ops.py:247-248 does, so kept indices
// are bit-exact with the CPU oracle.
#include <stdlib.h>

#include "common.h"

#define NMS_THREADS 1024
#define NMS_LDS_KEYS 16384
#define NMS_BINS 2048
typedef unsigned long long u64;

struct NmsArgs {
  int dbg;
  const float* pred;
  int n, nc, A;
  float conf, iou;
  const int32_t* classes;
  int n_classes, agnostic, multi_label, max_det, max_nms;
  float max_wh;
  float* out;
  int32_t* kept_anchor;
  int32_t* counts;
  const u64* best;  // optional [n][A]: make_key(best score, anchor*nc + best class) of every anchor, unfiltered (written by mgdt_detect_tail_fwd)
  u64* ws;          // per image: sort buffer [cap_pow2] (+ [A] best-class keys when !multi_label)
  long ws_per_image;  // in u64
  int cap_pow2;
};

__device__ __forceinline__ u64 make_key(float score, unsigned cand) {
  return ((u64)(0xFFFFFFFFu - __float_as_uint(score)) << 32) | cand;   // scores are positive: bit order == value order
}
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ bool class_ok(int c, const int32_t* classes, int n_classes) {
  if (!classes) return true;
  for (int i = 0; i < n_classes; ++i)
    if (classes[i] == c) return true;
  return false;
}

// Best class per anchor (first maximal index, ops.py:225-226) for the whole batch at full-chip parallelism; one thread per
// anchor, loads are anchor-contiguous (coalesced) and independent across classes (unrolled, many in flight).
__global__ __launch_bounds__(256) void nms_best_kernel(const NmsArgs a) {
  const int img = blockIdx.y, an = blockIdx.x * 256 + threadIdx.x;
  if (an >= a.A) return;
  const float* P = a.pred + (long)img * (4 + a.nc) * a.A + (long)4 * a.A + an;
  float best = P[0];
  int bc = 0;
  int c = 1;
  for (; c + 8 <= a.nc; c += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = P[(long)(c + k) * a.A];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (v[k] > best) { best = v[k]; bc = c + k; }
  }
  for (; c < a.nc; ++c) {
    float v = P[(long)c * a.A];
    if (v > best) { best = v; bc = c; }
  }
  u64* akeys = a.ws + (long)img * a.ws_per_image + a.cap_pow2;
  akeys[an] = (best > a.conf && class_ok(bc, a.classes, a.n_classes)) ? make_key(best, (unsigned)an * a.nc + bc) : KEY_NONE;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* P = a.pred + (long)img * (4 + a.nc) * a.A;
  u64* gbuf = a.ws + (long)img * a.ws_per_image;
  u64* akeys = gbuf + a.cap_pow2;              // only used when !multi_label
  __shared__ unsigned hist[256];
  __shared__ unsigned s_cnt, s_sel;
  __shared__ u64 s_prefix;

  const long total = a.multi_label ? (long)a.A * a.nc : a.A;
  // candidate key of flat index i (multi: i = c*A + anchor so that reads are anchor-contiguous)
  auto key_at = [&](long i) -> u64 {
    if (a.multi_label) {
      int c = (int)(i / a.A), an = (int)(i - (long)c * a.A);
      float sc = P[(long)(4 + c) * a.A + an];
      if (sc > a.conf && class_ok(c, a.classes, a.n_classes)) return make_key(sc, (unsigned)an * a.nc + c);
      return KEY_NONE;
    }
    if (a.best) {                       // unfiltered best-class keys from the Detect tail: apply the candidate predicate here
      const u64 k = a.best[(long)img * a.A + i];
      const float sc = __uint_as_float(0xFFFFFFFFu - (unsigned)(k >> 32));
      return (sc > a.conf && class_ok((int)((unsigned)k % (unsigned)a.nc), a.classes, a.n_classes)) ? k : KEY_NONE;
    }
    return akeys[i];
  };
  unsigned long long T0 = wall_clock64();
  unsigned long long Tsel = 0, Tsort = 0, Tgreedy = 0;
  long long Cg[6] = {0, 0, 0, 0, 0, 0};      // MGDT_NMS_DBG: cycles of the greedy step's parts (kept list, chunk matrix, barrier, walk, barrier) + steps
  // every pass over the candidates (count, the radix-select passes, the compactions) re-reads the same keys: when they fit, a thread keeps
  // its share (flat indices tid, tid + 1024, ...) in registers for the whole kernel
  constexpr int KPT = 8;
  const bool cached = total <= (long)KPT * NMS_THREADS;
  u64 mykey[KPT];
#pragma unroll
  for (int t = 0; t < KPT; ++t) {
    const long i = (long)t * NMS_THREADS + tid;
    mykey[t] = (cached && i < total) ? key_at(i) : KEY_NONE;
  }
  auto for_each_key = [&](auto fn) __attribute__((always_inline)) {     // fn(key) for every candidate key of this thread (KEY_NONE included)
    if (cached) {
#pragma unroll
      for (int t = 0; t < KPT; ++t) fn(mykey[t]);
    } else {
      for (long i = tid; i < total; i += NMS_THREADS) fn(key_at(i));
    }
  };
  // ---- score histogram: bin = floor(score * NMS_BINS) (exact: a power-of-two scaling), so {bin >= T} == {score >= T / NMS_BINS} is an exact
  // top set of the key order.  shist[b] becomes the number of candidates in bins >= b (shist[NMS_BINS] = 0); their total is the count.
  __shared__ unsigned shist[NMS_BINS + 1];
  __shared__ unsigned s_wsum[NMS_THREADS / 64];
  __shared__ int s_T;
  if (tid == 0) { s_cnt = 0; s_sel = 0; }
  for (int i = tid; i <= NMS_BINS; i += NMS_THREADS) shist[i] = 0;
  __syncthreads();
  for_each_key([&](u64 k) {
    if (k != KEY_NONE) {
      const float sc = __uint_as_float(0xFFFFFFFFu - (unsigned)(k >> 32));
      const int bin = min(NMS_BINS - 1, (int)(sc * (float)NMS_BINS));
      atomicAdd(&shist[bin], 1u);
    }
  });
  __syncthreads();
  {
    static_assert(NMS_BINS == 2 * NMS_THREADS, "two bins per thread");
    const unsigned h0 = shist[2 * tid], h1 = shist[2 * tid + 1], sum = h0 + h1;
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(incl, d, 64);
      if ((tid & 63) >= d) incl += t;
    }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned base = 0, total_c = 0;
    for (int w = 0; w < NMS_THREADS / 64; ++w) { const unsigned v = s_wsum[w]; if (w < (tid >> 6)) base += v; total_c += v; }
    const unsigned below0 = base + incl - sum;            // candidates in bins < 2 * tid
    shist[2 * tid] = total_c - below0;
    shist[2 * tid + 1] = total_c - below0 - h0;
    if (tid == 0) s_cnt = total_c;
  }
  __syncthreads();
  const unsigned ncand = s_cnt;
  const unsigned K = ncand < (unsigned)a.max_nms ? ncand : (unsigned)a.max_nms;

  // R-th smallest key (1-based rank R <= ncand): MSD radix select, 8 bits per pass.  Keys are unique (candidate id in the low word),
  // so exactly R keys are <= the result.
  __shared__ unsigned s_exact;
  auto select_rank = [&](unsigned R) -> u64 {
    u64 prefix = 0;
    unsigned need = R;
    for (int shift = 56; shift >= 0; shift -= 8) {
      for (int i = tid; i < 256; i += NMS_THREADS) hist[i] = 0;
      __syncthreads();
      const u64 himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
      for_each_key([&](u64 k) { if (k != KEY_NONE && (k & himask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u); });
      __syncthreads();
      if (tid < 64) {      // wave 0: first bin b with (keys in bins < b) + hist[b] >= need, by a 64-lane prefix scan over 4 bins per lane
        const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const unsigned sum = h0 + h1 + h2 + h3;
        unsigned incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned t = __shfl_up(incl, d, 64);
          if (tid >= d) incl += t;
        }
        const unsigned excl = incl - sum;
        if (excl < need && need <= incl) {               // exactly one lane
          unsigned run = excl, b = 4 * tid, hb = h0;
          if (run + h0 < need) { run += h0; ++b; hb = h1; if (run + h1 < need) { run += h1; ++b; hb = h2; if (run + h2 < need) { run += h2; ++b; hb = h3; } } }
          s_prefix = prefix | ((u64)b << shift);
          s_sel = need - run;
          s_exact = (need - run == hb) ? 1u : 0u;          // the whole bin is wanted: every lower bit may be 1
        }
      }
      __syncthreads();
      prefix = s_prefix;
      need = s_sel;
      const bool exact = s_exact != 0;
      __syncthreads();
      if (exact) return shift ? (prefix | ((1ull << shift) - 1ull)) : prefix;
    }
    return prefix;
  };

  // bitonic sort ascending (== descending score, ascending candidate id).  Every thread owns disjoint pairs per pass and issues all its
  // loads before the compare-exchanges; the LDS case is a separate instantiation so the compiler emits ds_read/ds_write_b64.
  auto bitonic = [&](auto* buf, unsigned np2) __attribute__((always_inline)) {
    const unsigned npairs = np2 >> 1;
    const unsigned blk = np2 / (NMS_THREADS / 64);          // contiguous elements owned by one wave in the wave-local passes
    const bool local_ok = blk >= 128;
    const unsigned wbase = (tid >> 6) * blk, ln = tid & 63;
    for (unsigned k2 = 2; k2 <= np2; k2 <<= 1) {
      for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
        if (local_ok && j < blk) {
          // both partners lie inside one wave's block: no workgroup barrier (a wave's LDS accesses complete in order)
          for (unsigned p0 = 0; p0 < (blk >> 1); p0 += 256) {
            u64 x[4], y[4];
            unsigned lo[4];
            bool act[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const unsigned p = p0 + t * 64 + ln;
              act[t] = p < (blk >> 1);
              lo[t] = wbase + (((p & ~(j - 1)) << 1) | (p & (j - 1)));
              if (act[t]) { x[t] = buf[lo[t]]; y[t] = buf[lo[t] | j]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (act[t]) {
                const bool up = (lo[t] & k2) == 0;
                if ((x[t] > y[t]) == up) { buf[lo[t]] = y[t]; buf[lo[t] | j] = x[t]; }
              }
            }
          }
          if (j == 1) __syncthreads();                       // stage done: the next stage may start with a cross-wave pass
        } else {
          for (unsigned p0 = 0; p0 < npairs; p0 += NMS_THREADS * 4) {
            u64 x[4], y[4];
            unsigned lo[4];
            bool act[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const unsigned p = p0 + t * NMS_THREADS + tid;
              act[t] = p < npairs;
              lo[t] = ((p & ~(j - 1)) << 1) | (p & (j - 1));     // index with bit log2(j) cleared
              if (act[t]) { x[t] = buf[lo[t]]; y[t] = buf[lo[t] | j]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (act[t]) {
                const bool up = (lo[t] & k2) == 0;
                if ((x[t] > y[t]) == up) { buf[lo[t]] = y[t]; buf[lo[t] | j] = x[t]; }
              }
            }
          }
          __syncthreads();
        }
      }
    }
  };

  // ---- greedy state
  float* cbox = (float*)(smem + (size_t)2 * NMS_THREADS * sizeof(u64));   // segment of <= 1024 keys: the candidates' boxes [5][1024] behind the two sort buffers
  static_assert((size_t)2 * NMS_THREADS * sizeof(u64) + (size_t)5 * NMS_THREADS * sizeof(float) <= (size_t)NMS_LDS_KEYS * sizeof(u64), "candidate boxes fit the key area");
  f32x4* kb4 = (f32x4*)(smem + (size_t)NMS_LDS_KEYS * sizeof(u64));    // kept boxes (class-offset): (x1, y1, x2, y2) - one 16-byte broadcast read per test
  float* karea = (float*)(kb4 + a.max_det);                              // their areas: read only by the rare exact IoU test
  u64* kkey = (u64*)(karea + a.max_det + (a.max_det & 1));              // key of each kept box (8-byte aligned)
  __shared__ int s_nkept;
  __shared__ unsigned s_sup[2];
  __shared__ unsigned s_row[64][2];                          // chunk-local suppression, row form: bit j of row i = the earlier candidate j suppresses i
  if (tid == 0) { s_nkept = 0; s_sup[0] = 0; s_sup[1] = 0; }
  if (tid < 64) { s_row[tid][0] = 0; s_row[tid][1] = 0; }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int md = a.max_det;
  // IoU test of the torchvision CPU kernel in IEEE fp32 (no contraction).  Boxes that do not intersect (almost all pairs) skip the division:
  // then inter == 0 and the quotient is 0, -0 or NaN, none of which exceeds iou_thres >= 0 - the same answer, bit for bit.
  auto overlaps = [&](float kx1, float ky1, float kx2, float ky2, float karea, float bx1, float by1, float bx2, float by2, float area) -> bool {
    const float iw = fminf(kx2, bx2) - fmaxf(kx1, bx1), ih = fminf(ky2, by2) - fmaxf(ky1, by1);
    if (!(iw > 0.f && ih > 0.f)) return false;
    const float inter = iw * ih;
    return inter / (karea + area - inter) > a.iou;
  };
  // Cheap exact-superset pre-test: iw > 0 && ih > 0 implies kx1 < bx2 && bx1 < kx2 && ky1 < by2 && by1 < ky2 (four compares, no arithmetic).
  // Boxes of different classes are max_wh apart, so almost every pair fails it; the IoU arithmetic runs only when some lane of the wave passes.
  auto may_overlap = [&](float kx1, float ky1, float kx2, float ky2, float bx1, float by1, float bx2, float by2) -> bool {
    return kx1 < bx2 && bx1 < kx2 && ky1 < by2 && by1 < ky2;
  };

  // ---- segments: the scan stops at max_det kept boxes, which usually happens within the first few hundred candidates, so only the best
  // `seg` keys are selected and sorted first (1024, then 4x more, ...) instead of all of them
  unsigned done = 0, seg = 1024;
  u64 lo_key = 0;
  bool use_hist = true;
  int bin_hi = NMS_BINS;
  while (done < K) {
    if (s_nkept >= md) break;                                // uniform: read after a barrier (end of the previous segment)
    unsigned long long Ta = wall_clock64();
    // the score edge T whose upper side [T, bin_hi) holds as many keys as fit this segment; exact because bins are score intervals
    unsigned R = 0;
    u64 hi_key = 0;
    bool by_hist = false;
    if (use_hist) {
      const unsigned base = shist[bin_hi], cap = (K - done <= seg + seg / 2) ? seg + seg / 2 : seg;
      if (tid == 0) s_T = bin_hi;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int b = 2 * tid + e;
        if (b < bin_hi && shist[b] - base <= cap && (b == 0 || shist[b - 1] - base > cap)) s_T = b;      // exactly one b (shist is monotone)
      }
      __syncthreads();
      const int T = s_T;
      const unsigned c = shist[T] - base;
      __syncthreads();                                           // s_T is rewritten by the next segment
      if (c >= 1 && done + c <= K) {
        by_hist = true;
        R = done + c;
        hi_key = T == 0 ? KEY_NONE - 1 : (((u64)(0xFFFFFFFFu - __float_as_uint((float)T / (float)NMS_BINS)) << 32) | 0xFFFFFFFFull);
        bin_hi = T;
      }
    }
    if (!by_hist) {                                              // one bin overflows the segment, or max_nms cuts inside a bin: exact rank select
      use_hist = false;
      R = (K - done <= seg + seg / 2) ? K : done + seg;          // do not leave a small tail for another pass
      hi_key = (R == ncand) ? KEY_NONE - 1 : select_rank(R);
    }
    const unsigned cnt = R - done;
    unsigned np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    u64* sbuf = (np2 <= NMS_LDS_KEYS) ? (u64*)smem : gbuf;
    if (tid == 0) s_sel = 0;
    bool preloaded = false;
    __syncthreads();
    auto put = [&](u64 k) __attribute__((always_inline)) {      // wave-aggregated: one LDS atomic per wave, not per key
      const bool sel = k != KEY_NONE && k <= hi_key && (done == 0 || k > lo_key);
      const u64 m = __ballot(sel);
      unsigned wbase = 0;
      if ((tid & 63) == 0 && m) wbase = atomicAdd(&s_sel, (unsigned)__popcll(m));
      wbase = __shfl(wbase, 0);
      if (sel) sbuf[wbase + (unsigned)__popcll(m & ((1ull << (tid & 63)) - 1ull))] = k;
    };
    if (cached) {
#pragma unroll
      for (int t = 0; t < KPT; ++t) put(mykey[t]);
    } else {
      for (long base = 0; base < total; base += NMS_THREADS) put(base + tid < total ? key_at(base + tid) : KEY_NONE);   // uniform trip count (ballots)
    }
    for (unsigned i = cnt + tid; i < (np2 < NMS_THREADS ? (unsigned)NMS_THREADS : np2); i += NMS_THREADS) sbuf[i] = KEY_NONE;
    __syncthreads();
    unsigned long long Tb = wall_clock64();
    if (np2 <= NMS_THREADS) {
      // one key per thread: the 45 passes with partner distance < 64 are register shuffles inside a wave, the 10 cross-wave ones go through
      // two alternating LDS buffers (one barrier each)
      u64* sb2 = (u64*)smem + NMS_THREADS;
      u64 kv = ((u64*)smem)[tid];
      int pb = 0;
      for (unsigned k2 = 2; k2 <= NMS_THREADS; k2 <<= 1) {
        for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
          u64 other;
          if (j < 64) {
            const unsigned lo32 = __shfl_xor((unsigned)kv, (int)j, 64), hi32 = __shfl_xor((unsigned)(kv >> 32), (int)j, 64);
            other = ((u64)hi32 << 32) | lo32;
          } else {
            u64* wb = pb ? sb2 : (u64*)smem;
            wb[tid] = kv;
            __syncthreads();
            other = wb[tid ^ j];
            pb ^= 1;
          }
          const bool up = (tid & k2) == 0, lower = (tid & j) == 0;
          const bool take_min = lower == up;
          kv = take_min ? (kv < other ? kv : other) : (kv > other ? kv : other);
        }
      }
      __syncthreads();                                         // the last cross-wave reads are done before smem[] is rewritten
      ((u64*)smem)[tid] = kv;
      {
        // every thread fetches the box of its (sorted) candidate now - 1024 independent requests in one round trip - and parks it in LDS:
        // the greedy steps below then never wait on a dependent key -> box load (that wait was most of a step)
        const bool valid = (unsigned)tid < cnt;
        const unsigned cand = (unsigned)(kv & 0xFFFFFFFFu);
        const int an = valid ? (int)(cand / (unsigned)a.nc) : 0, cls = valid ? (int)(cand % (unsigned)a.nc) : 0;
        const float cx = P[an], cy = P[(long)a.A + an], w = P[2L * a.A + an], h = P[3L * a.A + an];
        const float x1 = cx - w / 2.f, y1 = cy - h / 2.f, x2 = cx + w / 2.f, y2 = cy + h / 2.f;   // xywh2xyxy, ops.py:372-376
        const float off = a.agnostic ? 0.f : (float)cls * a.max_wh;                                  // ops.py:247
        const float bx1 = x1 + off, by1 = y1 + off, bx2 = x2 + off, by2 = y2 + off;
        cbox[tid] = bx1; cbox[NMS_THREADS + tid] = by1; cbox[2 * NMS_THREADS + tid] = bx2; cbox[3 * NMS_THREADS + tid] = by2;
        cbox[4 * NMS_THREADS + tid] = (bx2 - bx1) * (by2 - by1);
      }
      preloaded = true;
      __syncthreads();
    } else if (np2 <= NMS_LDS_KEYS) bitonic((u64*)smem, np2);
    else bitonic(gbuf, np2);
    unsigned long long Tc = wall_clock64();

    // ---- greedy suppression over this segment, 64 sorted candidates per step, the whole workgroup on each step:
    //   lane = candidate, wave s tests it against the kept boxes k = s, s+16, ... (a kept box is one broadcast LDS read for the wave) and
    //   against the chunk's own candidates 4s .. 4s+3 (bit j of sb[c]: the earlier candidate j suppresses c); the waves' verdicts are
    //   OR-ed in LDS.  Then wave 0 walks the 64 candidates in score order with scalar bit operations only (no IoU arithmetic in the
    //   serial part) and appends the survivors to the kept list.  The next chunk's boxes are loaded while this one is resolved.
    struct Cand { u64 key; float bx1, by1, bx2, by2, area; bool valid; };
    auto load_cand = [&](unsigned idx) -> Cand {
      Cand c;
      c.valid = idx < cnt;
      c.key = c.valid ? sbuf[idx] : KEY_NONE;
      if (preloaded) {                                         // uniform
        const unsigned i = c.valid ? idx : 0u;
        c.bx1 = cbox[i]; c.by1 = cbox[NMS_THREADS + i]; c.bx2 = cbox[2 * NMS_THREADS + i]; c.by2 = cbox[3 * NMS_THREADS + i]; c.area = cbox[4 * NMS_THREADS + i];
        return c;
      }
      const unsigned cand = (unsigned)(c.key & 0xFFFFFFFFu);
      const int an = c.valid ? (int)(cand / (unsigned)a.nc) : 0, cls = c.valid ? (int)(cand % (unsigned)a.nc) : 0;
      const float cx = P[an], cy = P[(long)a.A + an], w = P[2L * a.A + an], h = P[3L * a.A + an];
      const float x1 = cx - w / 2.f, y1 = cy - h / 2.f, x2 = cx + w / 2.f, y2 = cy + h / 2.f;   // xywh2xyxy, ops.py:372-376
      const float off = a.agnostic ? 0.f : (float)cls * a.max_wh;                                  // ops.py:247
      c.bx1 = x1 + off; c.by1 = y1 + off; c.bx2 = x2 + off; c.by2 = y2 + off;
      c.area = (c.bx2 - c.bx1) * (c.by2 - c.by1);
      return c;
    };
    Cand cur = load_cand((unsigned)lane);
    for (unsigned base = 0; base < cnt; base += 64) {
      const int nk = s_nkept;             // uniform: written by wave 0 before the barrier that ended the previous step
      if (nk >= md) break;
      Cand nxt = load_cand(base + 64 + (unsigned)lane);      // in flight during this step
      long long c0 = a.dbg ? clock64() : 0;
      bool sup = false;
      // kept boxes k = wave, wave + 16, ...: four broadcast reads in flight per round (one dependent LDS round trip per box was most of this loop)
      for (int k0 = wave; k0 < nk; k0 += 4 * (NMS_THREADS / 64)) {
        f32x4 q[4];
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * (NMS_THREADS / 64);
          q[u] = kb4[k < nk ? k : k0];
        }
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          hit[u] = k0 + u * (NMS_THREADS / 64) < nk && may_overlap(q[u][0], q[u][1], q[u][2], q[u][3], cur.bx1, cur.by1, cur.bx2, cur.by2);
          any |= hit[u];
        }
        if (__ballot(any)) {                                   // rare
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (hit[u] && overlaps(q[u][0], q[u][1], q[u][2], q[u][3], karea[k0 + u * (NMS_THREADS / 64)], cur.bx1, cur.by1, cur.bx2, cur.by2, cur.area)) sup = true;
        }
      }
      long long c1 = a.dbg ? clock64() : 0;
      // chunk-local matrix, row form: bit j of a lane's row = the earlier candidate j of the chunk suppresses this lane's candidate.  This wave
      // tests the columns j = 4 * wave .. 4 * wave + 3; the (rare) set bits are OR-ed into the lane's row in LDS.
      {
        unsigned rlo = 0, rhi = 0;
        float jb[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = wave * 4 + t;                        // uniform
          jb[t][0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.bx1), j));
          jb[t][1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.by1), j));
          jb[t][2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.bx2), j));
          jb[t][3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.by2), j));
        }
        bool hit[4], any = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          hit[t] = wave * 4 + t < lane && may_overlap(jb[t][0], jb[t][1], jb[t][2], jb[t][3], cur.bx1, cur.by1, cur.bx2, cur.by2);
          any |= hit[t];
        }
        if (__ballot(any)) {                                   // rare
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = wave * 4 + t;
            const float jar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.area), j));
            // kept box = the earlier candidate j, tested box = this lane's: the argument order of the serial scan
            if (hit[t] && overlaps(jb[t][0], jb[t][1], jb[t][2], jb[t][3], jar, cur.bx1, cur.by1, cur.bx2, cur.by2, cur.area)) {
              if (j < 32) rlo |= 1u << j; else rhi |= 1u << (j - 32);
            }
          }
          if (rlo) atomicOr(&s_row[lane][0], rlo);
          if (rhi) atomicOr(&s_row[lane][1], rhi);
        }
      }
      {
        const u64 m = __ballot(sup);
        if (lane == 0 && m) { atomicOr(&s_sup[0], (unsigned)m); atomicOr(&s_sup[1], (unsigned)(m >> 32)); }
      }
      long long c2 = a.dbg ? clock64() : 0;
      __syncthreads();
      long long c3 = a.dbg ? clock64() : 0;
      if (wave == 0) {
        const u64 supm = ((u64)s_sup[1] << 32) | s_sup[0];
        const u64 row = ((u64)s_row[lane][1] << 32) | s_row[lane][0];     // earlier candidates of the chunk that suppress this lane's
        const u64 alive = __ballot(cur.valid) & ~supm;
        // greedy rule inside the chunk: candidate i is kept iff it is alive and no KEPT earlier candidate suppresses it.  That recursion has
        // exactly one solution (fixed by induction over i) and iterating keep <- alive & ~(row hits keep) from keep = alive reaches it: after t
        // rounds the first t candidates are final, and a round that changes nothing is the solution.  Rows are almost always empty, so one
        // or two lane-parallel rounds replace a 64-step scalar walk.
        u64 keep = alive;
        for (int it = 0; it < 64; ++it) {
          const u64 nk_ = __ballot(((alive >> lane) & 1ull) && (row & keep) == 0);
          if (nk_ == keep) break;
          keep = nk_;
        }
        const int room = md - nk;                              // only the first `room` kept candidates (in score order) fit
        if (__popcll(keep) > room) {
          u64 k2 = keep;
          for (int c = 0; c < room; ++c) k2 &= k2 - 1;         // clear the lowest `room` set bits: what remains is the overflow
          keep &= ~k2;
        }
        if ((keep >> lane) & 1ull) {
          const int slot = nk + __popcll(keep & ((1ull << lane) - 1ull));
          kb4[slot] = f32x4{cur.bx1, cur.by1, cur.bx2, cur.by2};
          karea[slot] = cur.area;
          kkey[slot] = cur.key;
        }
        s_row[lane][0] = 0; s_row[lane][1] = 0;
        if (lane == 0) { s_sup[0] = 0; s_sup[1] = 0; s_nkept = nk + (int)__popcll(keep); }
      }
      long long c4 = a.dbg ? clock64() : 0;
      __syncthreads();
      if (a.dbg) { long long c5 = clock64(); Cg[0] += c1 - c0; Cg[1] += c2 - c1; Cg[2] += c3 - c2; Cg[3] += c4 - c3; Cg[4] += c5 - c4; Cg[5] += 1; }
      cur = nxt;
    }
    __syncthreads();
    unsigned long long Td = wall_clock64();
    Tsel += Tb - Ta; Tsort += Tc - Tb; Tgreedy += Td - Tc;
    done = R;
    lo_key = hi_key;
    seg *= 4;
  }
  __syncthreads();
  // output rows (x1,y1,x2,y2,conf,cls) + anchor index of the kept boxes, one thread per row
  for (int t = tid; t < s_nkept; t += NMS_THREADS) {
    const u64 key = kkey[t];
    const unsigned cand = (unsigned)(key & 0xFFFFFFFFu);
    const int an = (int)(cand / (unsigned)a.nc), cls = (int)(cand % (unsigned)a.nc);
    const float cx = P[an], cy = P[(long)a.A + an], w = P[2L * a.A + an], h = P[3L * a.A + an];
    float* o = a.out + ((long)img * md + t) * 6;
    o[0] = cx - w / 2.f; o[1] = cy - h / 2.f; o[2] = cx + w / 2.f; o[3] = cy + h / 2.f;
    o[4] = __uint_as_float(0xFFFFFFFFu - (unsigned)(key >> 32));
    o[5] = (float)cls;
    a.kept_anchor[(long)img * md + t] = an;
  }
  if (tid == 0) a.counts[img] = s_nkept;
  if (tid == 0 && a.dbg) { unsigned long long T4 = wall_clock64(); gbuf[0] = T4 - T0; gbuf[1] = Tsel; gbuf[2] = Tsort; gbuf[3] = Tgreedy; gbuf[4] = K; for (int i = 0; i < 6; ++i) gbuf[5 + i] = (u64)Cg[i]; }
}

static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

extern "C" size_t mgdt_nms_workspace_bytes(int n, int nc, int a, int multi_label, int max_nms) {
  long cand = multi_label ? (long)a * nc : a;
  long cap = next_pow2((int)std::min<long>(cand, max_nms));
  return (size_t)n * (cap + (multi_label ? 0 : a)) * sizeof(u64);
}

extern "C" int mgdt_nms_fwd(const float* pred, int n, int nc, int a, float conf_thres, float iou_thres, const int32_t* classes,
                            int n_classes, int agnostic, int multi_label, int max_det, int max_nms, float max_wh, float* out,
                            int32_t* kept_anchor, int32_t* counts, const unsigned long long* best_keys, void* ws, size_t ws_bytes, mgdt_stream s) {
  if (!pred || !out || !kept_anchor || !counts || !ws) MGDT_FAIL(MGDT_BAD_ARG, "nms: null pointer");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f)) MGDT_FAIL(MGDT_BAD_ARG, "Invalid Confidence threshold %g, valid values are between 0.0 and 1.0", conf_thres);
  if (!(iou_thres >= 0.f && iou_thres <= 1.f)) MGDT_FAIL(MGDT_BAD_ARG, "Invalid IoU %g, valid values are between 0.0 and 1.0", iou_thres);
  if (n < 1 || nc < 1 || a < 1 || max_det < 1 || max_nms < 1 || (long)a * nc > 0x7fffffffL) MGDT_FAIL(MGDT_BAD_SHAPE, "nms: n=%d nc=%d a=%d max_det=%d", n, nc, a, max_det);
  multi_label = multi_label && nc > 1;   // ops.py:196
  if (ws_bytes < mgdt_nms_workspace_bytes(n, nc, a, multi_label, max_nms)) MGDT_FAIL(MGDT_WORKSPACE, "nms: workspace too small");
  size_t lds = (size_t)NMS_LDS_KEYS * sizeof(u64) + (size_t)(5 * max_det + (max_det & 1)) * sizeof(float) + (size_t)max_det * sizeof(u64);   // keys | kept float4 + area | kept keys
  if (lds > 150 * 1024) MGDT_FAIL(MGDT_BAD_SHAPE, "nms: max_det=%d too large for the LDS kept list", max_det);
  NmsArgs g;
  g.dbg = getenv("MGDT_NMS_DBG") != nullptr;
  g.pred = pred; g.n = n; g.nc = nc; g.A = a; g.conf = conf_thres; g.iou = iou_thres; g.classes = n_classes > 0 ? classes : nullptr;
  g.n_classes = n_classes; g.agnostic = agnostic; g.multi_label = multi_label; g.max_det = max_det; g.max_nms = max_nms;
  g.max_wh = max_wh; g.out = out; g.kept_anchor = kept_anchor; g.counts = counts; g.ws = (u64*)ws;
  g.best = multi_label ? nullptr : (const u64*)best_keys;
  long cand = multi_label ? (long)a * nc : a;
  g.cap_pow2 = next_pow2((int)std::min<long>(cand, max_nms));
  g.ws_per_image = g.cap_pow2 + (multi_label ? 0 : a);
  static size_t attr_lds = 0;   // static LDS (histogram) + dynamic must stay <= 160 KiB: ask for what is needed only
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) MGDT_FAIL(MGDT_LAUNCH_FAIL, "nms: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    attr_lds = lds;
  }
  if (!multi_label && !g.best) nms_best_kernel<<<dim3(cdiv(a, 256), n), 256, 0, (hipStream_t)s>>>(g);
  nms_kernel<<<n, NMS_THREADS, lds, (hipStream_t)s>>>(g);
  MGDT_CHECK_LAUNCH("nms_fwd");
  return MGDT_OK;
}

// ================================================================================================ validator matching
// DetectionValidator._process_batch (yolo/v8/detect/val.py:152-175) for a whole batch: one workgroup per image.
//   iou[l][d] = box_iou(labels, detections) (metrics.py:52-72, eps 1e-7 in the union; this file is built with -ffp-contract=off)
//   per IoU level t: best[d] = label with the largest IoU among {iou >= level, same class};  a label keeps the LOWEST-INDEX detection that
//   chose it (np.unique(det) then np.unique(label) without the re-sort, as the fork is written);  correct[d][t] = d is kept.
// Exact IoU ties between two labels of one detection (numpy's unstable argsort decides in the reference) resolve to the lower label index.
#define VM_T 16      // max IoU levels
__global__ __launch_bounds__(256) void val_match_kernel(const float* __restrict__ det, const int32_t* __restrict__ ndet, int max_det,
                                                        const float* __restrict__ lab, const int32_t* __restrict__ nlab, int max_lab,
                                                        const float* __restrict__ iouv, int T, uint8_t* __restrict__ correct) {
  extern __shared__ int winner[];     // [T][max_lab]: lowest detection index that chose the label
  const int img = blockIdx.x, tid = threadIdx.x;
  const int nd = min(ndet[img], max_det), nl = min(nlab[img], max_lab);
  const float* D = det + (long)img * max_det * 6;
  const float* L = lab + (long)img * max_lab * 5;
  uint8_t* C = correct + (long)img * max_det * T;
  for (int i = tid; i < T * max_lab; i += 256) winner[i] = 0x7fffffff;
  __syncthreads();
  for (int d0 = 0; d0 < max_det; d0 += 256) {     // uniform trip count: barriers inside
    const int d = d0 + tid;
    int best[VM_T];
    float bestv[VM_T];
#pragma unroll
    for (int t = 0; t < VM_T; ++t) { best[t] = -1; bestv[t] = -1.f; }
    if (d < nd) {
      const float x1 = D[d * 6], y1 = D[d * 6 + 1], x2 = D[d * 6 + 2], y2 = D[d * 6 + 3], cls = D[d * 6 + 5];
      const float area_d = (x2 - x1) * (y2 - y1);
      for (int l = 0; l < nl; ++l) {
        if (L[l * 5] != cls) continue;
        const float lx1 = L[l * 5 + 1], ly1 = L[l * 5 + 2], lx2 = L[l * 5 + 3], ly2 = L[l * 5 + 4];
        const float iw = fmaxf(fminf(lx2, x2) - fmaxf(lx1, x1), 0.f), ih = fmaxf(fminf(ly2, y2) - fmaxf(ly1, y1), 0.f);
        const float inter = iw * ih;
        const float iou = inter / ((lx2 - lx1) * (ly2 - ly1) + area_d - inter + 1e-7f);
#pragma unroll
        for (int t = 0; t < VM_T; ++t)
          if (t < T && iou >= iouv[t] && iou > bestv[t]) { bestv[t] = iou; best[t] = l; }
      }
#pragma unroll
      for (int t = 0; t < VM_T; ++t)
        if (t < T && best[t] >= 0) atomicMin(&winner[t * max_lab + best[t]], d);
    }
    __syncthreads();
    // detections of later rounds have larger indices: a winner found in this round is final
    if (d < max_det) {
#pragma unroll
      for (int t = 0; t < VM_T; ++t)
        if (t < T) C[d * T + t] = (d < nd && best[t] >= 0 && winner[t * max_lab + best[t]] == d) ? 1 : 0;
    }
    __syncthreads();
  }
}

extern "C" int mgdt_val_match_fwd(const float* det, const int32_t* ndet, int n, int max_det, const float* labels, const int32_t* nlab, int max_lab,
                                  const float* iouv, int n_iou, uint8_t* correct, mgdt_stream s) {
  if (!det || !ndet || !labels || !nlab || !iouv || !correct) MGDT_FAIL(MGDT_BAD_ARG, "val_match: null pointer");
  if (n < 1 || max_det < 1 || max_lab < 1 || n_iou < 1 || n_iou > VM_T || (size_t)n_iou * max_lab * sizeof(int) > 64 * 1024)
    MGDT_FAIL(MGDT_BAD_SHAPE, "val_match: n=%d max_det=%d max_lab=%d n_iou=%d (<= %d levels, levels*max_lab <= 16384)", n, max_det, max_lab, n_iou, VM_T);
  val_match_kernel<<<n, 256, (size_t)n_iou * max_lab * sizeof(int), (hipStream_t)s>>>(det, ndet, max_det, labels, nlab, max_lab, iouv, n_iou, correct);
  MGDT_CHECK_LAUNCH("val_match_fwd");
  return MGDT_OK;
}
