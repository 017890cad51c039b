// v8DetectionLoss on device: fork-specific task-aligned assigner (HeuristicPositiveSampleAssigner_v1 -> TaskAlignedAssigner,
// yolo/utils/tal.py:56-353), BCE / CIoU / DFL loss (yolo/utils/loss.py:56-89,159-208) and the gradient w.r.t. the raw head maps.
// The reference runs ~60 small torch ops with (B, Nmax, A) temporaries, CPU-resident index tensors and host syncs
// (`fg_mask.sum()`, `max(tss, 1)`); here it is 8 launches, no host sync, fixed-order reductions (run-to-run identical).
//
//   prep     (b,a)    : DFL softmax-expectation -> predicted boxes (grid units + pixels), anchor centres
//   metrics  (b,j,a)  : in-box test, CIoU(gt, pred) clamp 0, sigmoid(score[label]), align = s^alpha * ov^beta
//   topk     (b,j)    : 10 best anchors per GT by (align desc, index asc)  -> byte mask
//   resolve  (b,a)    : mask_pos = topk & in_gt & valid; anchors claimed by >1 GT go to the GT with the largest ALIGN METRIC
//                       (fork change, tal.py:222) -> fg mask, target_gt_idx
//   posmax   (b,j)    : max over the GT's positives of align / overlaps (normalisation terms)
//   loss     (b,a)    : target score = align*pos_ov/(pos_align+eps); BCE over classes, (1-CIoU)*w, DFL*w, partial sums per block
//   finalize          : sums -> loss items [box, cls, dfl] * gains, total * B
//   backward (b,a)    : d total / d logits (cls: sigmoid-t, dfl: softmax CE, box: analytic CIoU -> dist -> softmax)
#include "common.h"

#define TOPK 10
#define TAL_EPS 1e-9f
#define LOSS_BLOCK 256

struct LossArgs {
  const void* feat;          // one level: (B, H, W, no) NHWC view of dtype T
  long fsn, fsh, fsw;
  int B, H, W, R, nc, A, a_off;   // A = anchors over all levels, a_off = first anchor of this level
  float stride;
  const float* gt;           // [B][N][5] = cls, x1, y1, x2, y2 (pixels); rows with box sum <= 0 are padding
  int N;
  float alpha, beta;
  const int* call_dev;       // when set: the per-call counter lives on the device (captured training step), alpha is derived from it
  float* pbox;               // [B][A][4] predicted xyxy, grid units
  float* anc;                // [A][3]    anchor x, y (grid units), stride
  float* align;              // [B][N][A]
  float* ov;                 // [B][N][A]
  unsigned char* mtopk;      // [B][N][A]
  unsigned char* fg;         // [B][A]
  int* gt_idx;               // [B][A]
  float* posmax;             // [B][N][2] = max align, max overlaps over the GT's positives
  float* tscore;             // [B][A] normalised target score (0 for background)
  float* partial;            // [nblocks][4] = sum bce, sum (1-ciou)*w, sum dfl*w, sum tscore
  float* out;                // [5] = total*B, box, cls, dfl (gains applied), tss
  void* grad;                // same layout as feat
  long gsn, gsh, gsw;
  float gain_box, gain_cls, gain_dfl;
  int nblocks;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// CIoU of box1 vs box2 (xyxy), eps on h only as in metrics.py:102-103.  With D != nullptr also d CIoU / d box1 (alpha detached).
__device__ __forceinline__ float ciou_xyxy(const float* b1, const float* b2, float* D) {
  const float eps = 1e-7f;
  float w1 = b1[2] - b1[0], h1 = b1[3] - b1[1] + eps;
  float w2 = b2[2] - b2[0], h2 = b2[3] - b2[1] + eps;
  float ix1 = fmaxf(b1[0], b2[0]), ix2 = fminf(b1[2], b2[2]), iy1 = fmaxf(b1[1], b2[1]), iy2 = fminf(b1[3], b2[3]);
  float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
  float inter = iw * ih;
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cx1 = fminf(b1[0], b2[0]), cx2 = fmaxf(b1[2], b2[2]), cy1 = fminf(b1[1], b2[1]), cy2 = fmaxf(b1[3], b2[3]);
  float cw = cx2 - cx1, ch = cy2 - cy1;
  float c2 = cw * cw + ch * ch + eps;
  float dxc = b2[0] + b2[2] - b1[0] - b1[2], dyc = b2[1] + b2[3] - b1[1] - b1[3];
  float rho2 = (dxc * dxc + dyc * dyc) / 4.f;
  const float k = 0.40528473456935109f;   // 4 / pi^2
  float at = atanf(w2 / h2) - atanf(w1 / h1);
  float v = k * at * at;
  float alpha = v / (v - iou + (1.f + eps));
  float ciou = iou - (rho2 / c2 + v * alpha);
  if (D) {
    // partials of the box-1 coordinates q in {x1, y1, x2, y2}
    const bool px = ix2 > ix1, py = iy2 > iy1;            // clamp(0) passes gradient only where positive
    float d_iw[4] = {(px && b1[0] >= b2[0]) ? -1.f : 0.f, 0.f, (px && b1[2] <= b2[2]) ? 1.f : 0.f, 0.f};
    float d_ih[4] = {0.f, (py && b1[1] >= b2[1]) ? -1.f : 0.f, 0.f, (py && b1[3] <= b2[3]) ? 1.f : 0.f};
    float d_w1[4] = {-1.f, 0.f, 1.f, 0.f}, d_h1[4] = {0.f, -1.f, 0.f, 1.f};
    float d_cw[4] = {b1[0] <= b2[0] ? -1.f : 0.f, 0.f, b1[2] >= b2[2] ? 1.f : 0.f, 0.f};
    float d_ch[4] = {0.f, b1[1] <= b2[1] ? -1.f : 0.f, 0.f, b1[3] >= b2[3] ? 1.f : 0.f};
    float d_dxc[4] = {-1.f, 0.f, -1.f, 0.f}, d_dyc[4] = {0.f, -1.f, 0.f, -1.f};
    const float r = w1 / h1;
    const float d_at_dr = -1.f / (1.f + r * r);           // d(atan(w2/h2) - atan(r)) / dr
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float d_inter = d_iw[q] * ih + iw * d_ih[q];
      float d_uni = d_w1[q] * h1 + w1 * d_h1[q] - d_inter;
      float d_iou = (d_inter * uni - inter * d_uni) / (uni * uni);
      float d_c2 = 2.f * cw * d_cw[q] + 2.f * ch * d_ch[q];
      float d_rho2 = (2.f * dxc * d_dxc[q] + 2.f * dyc * d_dyc[q]) / 4.f;
      float d_r = (d_w1[q] * h1 - w1 * d_h1[q]) / (h1 * h1);
      float d_v = k * 2.f * at * d_at_dr * d_r;
      D[q] = d_iou - ((d_rho2 * c2 - rho2 * d_c2) / (c2 * c2) + d_v * alpha);
    }
  }
  return ciou;
}

// ------------------------------------------------------------------------------------------------ prep
template <typename T>
__global__ void loss_prep_kernel(const LossArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int HW = a.H * a.W;
  if (i >= a.B * HW) return;
  int b = i / HW, al = i - b * HW, oy = al / a.W, ox = al - oy * a.W;
  const T* p = (const T*)a.feat + b * a.fsn + oy * a.fsh + ox * a.fsw;
  float d[4];
  for (int s = 0; s < 4; ++s) {
    float mx = -INFINITY;
    for (int k = 0; k < a.R; ++k) mx = fmaxf(mx, (float)p[s * a.R + k]);
    float den = 0.f, num = 0.f;
    for (int k = 0; k < a.R; ++k) {
      float e = expf((float)p[s * a.R + k] - mx);
      den += e;
      num += e * (float)k;
    }
    d[s] = num / den;
  }
  float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
  int ag = a.a_off + al;
  float* pb = a.pbox + ((long)b * a.A + ag) * 4;
  pb[0] = ax - d[0]; pb[1] = ay - d[1]; pb[2] = ax + d[2]; pb[3] = ay + d[3];
  if (b == 0) { a.anc[ag * 3] = ax; a.anc[ag * 3 + 1] = ay; a.anc[ag * 3 + 2] = a.stride; }
}

// ------------------------------------------------------------------------------------------------ metrics (one level)
template <typename T>
__global__ void tal_metrics_kernel(const LossArgs a) {
  int al = blockIdx.x * blockDim.x + threadIdx.x;
  int HW = a.H * a.W;
  if (al >= HW) return;
  const int b = blockIdx.z, j = blockIdx.y;
  const int ag = a.a_off + al, oy = al / a.W, ox = al - oy * a.W;
  const float* g = a.gt + ((long)b * a.N + j) * 5;
  const float gx1 = g[1], gy1 = g[2], gx2 = g[3], gy2 = g[4];
  const bool valid = (gx1 + gy1 + gx2 + gy2) > 0.f;                       // mask_gt, loss.py:184
  const float ax = ((float)ox + 0.5f) * a.stride, ay = ((float)oy + 0.5f) * a.stride;
  const float dmin = fminf(fminf(ax - gx1, ay - gy1), fminf(gx2 - ax, gy2 - ay));
  float al_v = 0.f, ov_v = 0.f;
  if (valid && dmin > TAL_EPS) {
    const float* pbg = a.pbox + ((long)b * a.A + ag) * 4;
    float pb[4] = {pbg[0] * a.stride, pbg[1] * a.stride, pbg[2] * a.stride, pbg[3] * a.stride};
    float gb[4] = {gx1, gy1, gx2, gy2};
    ov_v = fmaxf(ciou_xyxy(gb, pb, nullptr), 0.f);
    int label = (int)g[0];
    const T* p = (const T*)a.feat + b * a.fsn + oy * a.fsh + ox * a.fsw;
    float sc = sigmoidf_((float)p[4 * a.R + label]);
    const float alpha = a.call_dev ? 0.5f * (float)(100 - *a.call_dev / 161) / 100.f : a.alpha;
    al_v = powf(sc, alpha) * powf(ov_v, a.beta);
  }
  long o = ((long)b * a.N + j) * a.A + ag;
  a.align[o] = al_v;
  a.ov[o] = ov_v;
}

// ------------------------------------------------------------------------------------------------ top-k per GT
__global__ __launch_bounds__(256) void tal_topk_kernel(const LossArgs a) {
  const int b = blockIdx.y, j = blockIdx.x;
  const float* g = a.gt + ((long)b * a.N + j) * 5;
  if (!((g[1] + g[2] + g[3] + g[4]) > 0.f)) return;     // padded GT: indices collapse to 0, count>1 -> 0 (tal.py:293,304)
  const float* al = a.align + ((long)b * a.N + j) * a.A;
  unsigned char* mt = a.mtopk + ((long)b * a.N + j) * a.A;
  __shared__ unsigned long long red[256];
  unsigned long long last = ~0ull;   // keys are selected in strictly decreasing order
  for (int t = 0; t < TOPK && t < a.A; ++t) {
    unsigned long long best = 0ull;
    bool any = false;
    for (int i = threadIdx.x; i < a.A; i += 256) {
      unsigned long long key = ((unsigned long long)__float_as_uint(al[i]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
      if (key < last && (!any || key > best)) { best = key; any = true; }
    }
    red[threadIdx.x] = any ? best : 0ull;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o && red[threadIdx.x + o] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + o];
      __syncthreads();
    }
    last = red[0];
    __syncthreads();
    if (threadIdx.x == 0) mt[0xFFFFFFFFu - (unsigned)(last & 0xFFFFFFFFu)] = 1;
  }
}

// ------------------------------------------------------------------------------------------------ resolve multi-claims
__global__ void tal_resolve_kernel(const LossArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.A) return;
  int b = i / a.A, ag = i - b * a.A;
  const float ax = a.anc[ag * 3] * a.anc[ag * 3 + 2], ay = a.anc[ag * 3 + 1] * a.anc[ag * 3 + 2];
  int cnt = 0, first = 0, best_j = 0;
  float best_al = -1.f;
  for (int j = 0; j < a.N; ++j) {
    long o = ((long)b * a.N + j) * a.A + ag;
    const float* g = a.gt + ((long)b * a.N + j) * 5;
    float al = a.align[o];
    if (al > best_al) { best_al = al; best_j = j; }                 // first maximal index over ALL GTs
    bool valid = (g[1] + g[2] + g[3] + g[4]) > 0.f;
    float dmin = fminf(fminf(ax - g[1], ay - g[2]), fminf(g[3] - ax, g[4] - ay));
    if (a.mtopk[o] && valid && dmin > TAL_EPS) {
      if (cnt == 0) first = j;
      ++cnt;
    }
  }
  int idx = cnt > 1 ? best_j : first;
  a.fg[i] = cnt > 0;
  a.gt_idx[i] = cnt > 0 ? idx : 0;
}

// ------------------------------------------------------------------------------------------------ per-GT maxima over its positives
__global__ __launch_bounds__(256) void tal_posmax_kernel(const LossArgs a) {
  const int b = blockIdx.y, j = blockIdx.x;
  const float* al = a.align + ((long)b * a.N + j) * a.A;
  const float* ov = a.ov + ((long)b * a.N + j) * a.A;
  float ma = 0.f, mo = 0.f;
  for (int i = threadIdx.x; i < a.A; i += 256)
    if (a.fg[(long)b * a.A + i] && a.gt_idx[(long)b * a.A + i] == j) { ma = fmaxf(ma, al[i]); mo = fmaxf(mo, ov[i]); }
  __shared__ float ra[256], ro[256];
  ra[threadIdx.x] = ma; ro[threadIdx.x] = mo;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ra[threadIdx.x] = fmaxf(ra[threadIdx.x], ra[threadIdx.x + o]); ro[threadIdx.x] = fmaxf(ro[threadIdx.x], ro[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { a.posmax[((long)b * a.N + j) * 2] = ra[0]; a.posmax[((long)b * a.N + j) * 2 + 1] = ro[0]; }
}

// ------------------------------------------------------------------------------------------------ loss partial sums (one level)
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

template <typename T>
__global__ __launch_bounds__(LOSS_BLOCK) void loss_fwd_kernel(const LossArgs a, int block0) {
  int i = blockIdx.x * LOSS_BLOCK + threadIdx.x;
  int HW = a.H * a.W;
  float s_bce = 0.f, s_box = 0.f, s_dfl = 0.f, s_ts = 0.f;
  if (i < a.B * HW) {
    int b = i / HW, al = i - b * HW, oy = al / a.W, ox = al - oy * a.W, ag = a.a_off + al;
    const T* p = (const T*)a.feat + b * a.fsn + oy * a.fsh + ox * a.fsw;
    const bool fg = a.fg[(long)b * a.A + ag];
    int label = -1;
    float ts = 0.f;
    const float* g = nullptr;
    if (fg) {
      int j = a.gt_idx[(long)b * a.A + ag];
      g = a.gt + ((long)b * a.N + j) * 5;
      label = max((int)g[0], 0);
      long o = ((long)b * a.N + j) * a.A + ag;
      const float* pm = a.posmax + ((long)b * a.N + j) * 2;
      ts = a.align[o] * pm[1] / (pm[0] + TAL_EPS);                  // tal.py:226-231
    }
    a.tscore[(long)b * a.A + ag] = ts;
    s_ts = ts;
    for (int c = 0; c < a.nc; ++c) s_bce += bce_logits((float)p[4 * a.R + c], c == label ? ts : 0.f);
    if (fg) {
      const float* pb = a.pbox + ((long)b * a.A + ag) * 4;
      float tb[4] = {g[1] / a.stride, g[2] / a.stride, g[3] / a.stride, g[4] / a.stride};   // loss.py:200
      s_box = (1.f - ciou_xyxy(pb, tb, nullptr)) * ts;
      const float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
      float tl4[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
      float dfl = 0.f;
      for (int s = 0; s < 4; ++s) {
        float t = fminf(fmaxf(tl4[s], 0.f), (float)(a.R - 1) - 0.01f);       // bbox2dist clamp, tal.py:506
        int tl = (int)t;
        float wl = (float)(tl + 1) - t, wr = 1.f - wl;
        float mx = -INFINITY;
        for (int k = 0; k < a.R; ++k) mx = fmaxf(mx, (float)p[s * a.R + k]);
        float den = 0.f;
        for (int k = 0; k < a.R; ++k) den += expf((float)p[s * a.R + k] - mx);
        float lse = mx + logf(den);
        dfl += (lse - (float)p[s * a.R + tl]) * wl + (lse - (float)p[s * a.R + tl + 1]) * wr;
      }
      s_dfl = dfl * 0.25f * ts;
    }
  }
  __shared__ float red[4][LOSS_BLOCK];
  red[0][threadIdx.x] = s_bce; red[1][threadIdx.x] = s_box; red[2][threadIdx.x] = s_dfl; red[3][threadIdx.x] = s_ts;
  __syncthreads();
  for (int o = LOSS_BLOCK / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) a.partial[(long)(block0 + blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const LossArgs a) {
  __shared__ double red[4][256];
  double s[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < a.nblocks; i += 256)
    for (int k = 0; k < 4; ++k) s[k] += a.partial[(long)i * 4 + k];
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float tss = fmaxf((float)red[3][0], 1.f);                                  // loss.py:191
    float box = (float)red[1][0] / tss * a.gain_box, cls = (float)red[0][0] / tss * a.gain_cls, dfl = (float)red[2][0] / tss * a.gain_dfl;
    a.out[0] = (box + cls + dfl) * (float)a.B;                                 // loss.py:208
    a.out[1] = box; a.out[2] = cls; a.out[3] = dfl; a.out[4] = tss;
  }
}

// ------------------------------------------------------------------------------------------------ backward (one level)
template <typename T>
__global__ void loss_bwd_kernel(const LossArgs a, float gscale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int HW = a.H * a.W;
  if (i >= a.B * HW) return;
  int b = i / HW, al = i - b * HW, oy = al / a.W, ox = al - oy * a.W, ag = a.a_off + al;
  const T* p = (const T*)a.feat + b * a.fsn + oy * a.fsh + ox * a.fsw;
  T* gp = (T*)a.grad + b * a.gsn + oy * a.gsh + ox * a.gsw;
  const float tss = a.out[4];
  const float kB = gscale * (float)a.B / tss;
  const bool fg = a.fg[(long)b * a.A + ag];
  const float ts = a.tscore[(long)b * a.A + ag];
  int label = -1;
  const float* g = nullptr;
  if (fg) {
    g = a.gt + ((long)b * a.N + a.gt_idx[(long)b * a.A + ag]) * 5;
    label = max((int)g[0], 0);
  }
  for (int c = 0; c < a.nc; ++c)
    gp[4 * a.R + c] = (T)((sigmoidf_((float)p[4 * a.R + c]) - (c == label ? ts : 0.f)) * a.gain_cls * kB);
  if (!fg) {
    for (int k = 0; k < 4 * a.R; ++k) gp[k] = (T)0.f;
    return;
  }
  const float* pb = a.pbox + ((long)b * a.A + ag) * 4;
  float tb[4] = {g[1] / a.stride, g[2] / a.stride, g[3] / a.stride, g[4] / a.stride};
  float D[4];
  ciou_xyxy(pb, tb, D);
  // d(1-ciou)/d dist_s : x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3
  const float dd[4] = {D[0], D[1], -D[2], -D[3]};
  const float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
  float tl4[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
  for (int s = 0; s < 4; ++s) {
    float mx = -INFINITY;
    for (int k = 0; k < a.R; ++k) mx = fmaxf(mx, (float)p[s * a.R + k]);
    float den = 0.f, num = 0.f;
    for (int k = 0; k < a.R; ++k) { float e = expf((float)p[s * a.R + k] - mx); den += e; num += e * (float)k; }
    float dist = num / den;
    float t = fminf(fmaxf(tl4[s], 0.f), (float)(a.R - 1) - 0.01f);
    int tl = (int)t;
    float wl = (float)(tl + 1) - t, wr = 1.f - wl;
    for (int k = 0; k < a.R; ++k) {
      float pk = expf((float)p[s * a.R + k] - mx) / den;
      float g_box = dd[s] * pk * ((float)k - dist) * ts * a.gain_box;
      float g_dfl = (pk - (k == tl ? wl : 0.f) - (k == tl + 1 ? wr : 0.f)) * 0.25f * ts * a.gain_dfl;
      gp[s * a.R + k] = (T)((g_box + g_dfl) * kB);
    }
  }
}

// Vector form (nc % 4 == 0, 4-aligned views): thread = (anchor, channel quad).  Class quads are 8 / 16-byte coalesced accesses (the scalar kernel
// above walked 4R + nc channels per thread: neighbouring lanes 192 bytes apart); the thread of quad 0 also does the anchor's 4R box bins.
template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_vec_kernel(const LossArgs a, float gscale) {
  const int HW = a.H * a.W, R4 = 4 * a.R, Q0 = R4 / 4, QC = a.nc / 4, QT = 1 + QC;      // quad 0 = the whole box part, quads 1.. = class quads
  const long total = (long)a.B * HW * QT;
  const float tss = a.out[4];
  const float kB = gscale * (float)a.B / tss;
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int q = (int)(it % QT);
    const long i = it / QT;
    const int b = (int)(i / HW), al = (int)(i - (long)b * HW), oy = al / a.W, ox = al - oy * a.W, ag = a.a_off + al;
    const T* p = (const T*)a.feat + b * a.fsn + oy * a.fsh + ox * a.fsw;
    T* gp = (T*)a.grad + b * a.gsn + oy * a.gsh + ox * a.gsw;
    const bool fg = a.fg[(long)b * a.A + ag];
    const float ts = a.tscore[(long)b * a.A + ag];
    const float* g = fg ? a.gt + ((long)b * a.N + a.gt_idx[(long)b * a.A + ag]) * 5 : nullptr;
    if (q > 0) {
      const int c0 = (q - 1) * 4;
      const int label = fg ? max((int)g[0], 0) : -1;
      const f32x4 v = load4<T>(p + R4 + c0);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (sigmoidf_(v[j]) - (c0 + j == label ? ts : 0.f)) * a.gain_cls * kB;
      store4<T>(gp + R4 + c0, o);
      continue;
    }
    if (!fg) {
      for (int k = 0; k < Q0; ++k) store4<T>(gp + 4 * k, f32x4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    const float* pb = a.pbox + ((long)b * a.A + ag) * 4;
    float tb[4] = {g[1] / a.stride, g[2] / a.stride, g[3] / a.stride, g[4] / a.stride};
    float D[4];
    ciou_xyxy(pb, tb, D);
    const float dd[4] = {D[0], D[1], -D[2], -D[3]};
    const float ax = (float)ox + 0.5f, ay = (float)oy + 0.5f;
    float tl4[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
    for (int s = 0; s < 4; ++s) {
      float mx = -INFINITY;
      for (int k = 0; k < a.R; ++k) mx = fmaxf(mx, (float)p[s * a.R + k]);
      float den = 0.f, num = 0.f;
      for (int k = 0; k < a.R; ++k) { float e = expf((float)p[s * a.R + k] - mx); den += e; num += e * (float)k; }
      float dist = num / den;
      float t = fminf(fmaxf(tl4[s], 0.f), (float)(a.R - 1) - 0.01f);
      int tl = (int)t;
      float wl = (float)(tl + 1) - t, wr = 1.f - wl;
      for (int k = 0; k < a.R; ++k) {
        float pk = expf((float)p[s * a.R + k] - mx) / den;
        float g_box = dd[s] * pk * ((float)k - dist) * ts * a.gain_box;
        float g_dfl = (pk - (k == tl ? wl : 0.f) - (k == tl + 1 ? wr : 0.f)) * 0.25f * ts * a.gain_dfl;
        gp[s * a.R + k] = (T)((g_box + g_dfl) * kB);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
struct LevelDesc { const mgdt_view* v; float stride; int a_off; };

static void fill_level(LossArgs& a, const mgdt_view* v, float stride, int a_off) {
  a.feat = v->p; a.fsn = v->sn; a.fsh = v->sh; a.fsw = v->sw; a.H = v->h; a.W = v->w; a.stride = stride; a.a_off = a_off;
}

extern "C" size_t mgdt_detect_loss_workspace_bytes(int b, int a_total, int n_gt) {
  size_t bn = (size_t)b * std::max(n_gt, 1) * a_total;
  size_t f = (size_t)b * a_total * 4 + (size_t)a_total * 3 + 2 * bn + (size_t)b * std::max(n_gt, 1) * 2 + (size_t)b * a_total /*tscore*/ +
             (size_t)4 * ((size_t)b * a_total / LOSS_BLOCK + 64) + 8;
  size_t i = (size_t)b * a_total;                      // gt_idx
  size_t u = bn + (size_t)b * a_total;                 // mtopk + fg
  return f * 4 + i * 4 + ((u + 15) & ~(size_t)15) + 256;
}

static int detect_loss_fwd_impl(const mgdt_view* const* feats, const float* strides, int n_levels, int reg_max, int nc,
                                const float* gt, int n_gt, int call_count, const int32_t* call_count_dev, float gain_box, float gain_cls, float gain_dfl,
                                float* out5, unsigned char* fg_out, int32_t* gt_idx_out, float* tscore_out, void* ws,
                                size_t ws_bytes, int dtype, mgdt_stream s) {
  if (!feats || !strides || n_levels < 1 || n_levels > 8 || !out5 || !ws) MGDT_FAIL(MGDT_BAD_ARG, "detect_loss: null/empty argument");
  if (n_gt > 0 && !gt) MGDT_FAIL(MGDT_BAD_ARG, "detect_loss: gt is NULL");
  const int B = feats[0]->n, no = 4 * reg_max + nc;
  int A = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (!view_ok(feats[l]) || feats[l]->sc != 1 || feats[l]->c != no || feats[l]->n != B) MGDT_FAIL(MGDT_BAD_SHAPE, "detect_loss: level %d must be an NHWC (B,%d,H,W) view", l, no);
    A += feats[l]->h * feats[l]->w;
  }
  if (ws_bytes < mgdt_detect_loss_workspace_bytes(B, A, n_gt)) MGDT_FAIL(MGDT_WORKSPACE, "detect_loss: workspace too small");
  hipStream_t st = (hipStream_t)s;
  const int N = std::max(n_gt, 1);
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.R = reg_max; a.nc = nc; a.A = A; a.N = n_gt; a.gt = gt;
  // alpha schedule of the fork: coff = call_count // 161, alpha = 0.5 * (100 - coff) / 100 (tal.py:110,266-267); beta = 8 (loss.py:125-126)
  a.alpha = 0.5f * (float)(100 - call_count / 161) / 100.f;
  a.beta = 8.0f;
  a.call_dev = call_count_dev;
  a.gain_box = gain_box; a.gain_cls = gain_cls; a.gain_dfl = gain_dfl;
  float* f = (float*)ws;
  a.pbox = f; f += (size_t)B * A * 4;
  a.anc = f; f += (size_t)A * 3;
  a.align = f; f += (size_t)B * N * A;
  a.ov = f; f += (size_t)B * N * A;
  a.posmax = f; f += (size_t)B * N * 2;
  a.tscore = f; f += (size_t)B * A;
  a.partial = f;
  int nblocks = 0;
  for (int l = 0; l < n_levels; ++l) nblocks += cdiv((long)B * feats[l]->h * feats[l]->w, LOSS_BLOCK);
  a.nblocks = nblocks;
  f += (size_t)4 * nblocks;
  a.out = out5;
  a.gt_idx = (int*)f; f += (size_t)B * A;
  a.mtopk = (unsigned char*)f;
  a.fg = a.mtopk + (size_t)B * N * A;

  int a_off = 0;
  for (int l = 0; l < n_levels; ++l) {   // prep
    fill_level(a, feats[l], strides[l], a_off);
    long tot = (long)B * a.H * a.W;
    MGDT_DISPATCH_DTYPE(dtype, (loss_prep_kernel<T><<<cdiv(tot, 256), 256, 0, st>>>(a)));
    a_off += a.H * a.W;
  }
  if (n_gt > 0) {
    hipMemsetAsync(a.mtopk, 0, (size_t)B * N * A, st);
    a_off = 0;
    for (int l = 0; l < n_levels; ++l) {
      fill_level(a, feats[l], strides[l], a_off);
      dim3 grid(cdiv(a.H * a.W, 256), n_gt, B);
      MGDT_DISPATCH_DTYPE(dtype, (tal_metrics_kernel<T><<<grid, 256, 0, st>>>(a)));
      a_off += a.H * a.W;
    }
    tal_topk_kernel<<<dim3(n_gt, B), 256, 0, st>>>(a);
    tal_resolve_kernel<<<cdiv((long)B * A, 256), 256, 0, st>>>(a);
    tal_posmax_kernel<<<dim3(n_gt, B), 256, 0, st>>>(a);
  } else {   // empty-label batch: all background (the reference raises AttributeError here; upstream behaviour)
    hipMemsetAsync(a.fg, 0, (size_t)B * A, st);
    hipMemsetAsync(a.gt_idx, 0, (size_t)B * A * 4, st);
  }
  a_off = 0;
  int block0 = 0;
  for (int l = 0; l < n_levels; ++l) {
    fill_level(a, feats[l], strides[l], a_off);
    int nb = cdiv((long)B * a.H * a.W, LOSS_BLOCK);
    MGDT_DISPATCH_DTYPE(dtype, (loss_fwd_kernel<T><<<nb, LOSS_BLOCK, 0, st>>>(a, block0)));
    block0 += nb;
    a_off += a.H * a.W;
  }
  loss_finalize_kernel<<<1, 256, 0, st>>>(a);
  if (fg_out) hipMemcpyAsync(fg_out, a.fg, (size_t)B * A, hipMemcpyDeviceToDevice, st);
  if (gt_idx_out) hipMemcpyAsync(gt_idx_out, a.gt_idx, (size_t)B * A * 4, hipMemcpyDeviceToDevice, st);
  if (tscore_out) hipMemcpyAsync(tscore_out, a.tscore, (size_t)B * A * 4, hipMemcpyDeviceToDevice, st);
  MGDT_CHECK_LAUNCH("detect_loss_fwd");
  return MGDT_OK;
}

extern "C" int mgdt_detect_loss_fwd(const mgdt_view* const* feats, const float* strides, int n_levels, int reg_max, int nc,
                                    const float* gt, int n_gt, int call_count, float gain_box, float gain_cls, float gain_dfl,
                                    float* out5, unsigned char* fg_out, int32_t* gt_idx_out, float* tscore_out, void* ws,
                                    size_t ws_bytes, int dtype, mgdt_stream s) {
  return detect_loss_fwd_impl(feats, strides, n_levels, reg_max, nc, gt, n_gt, call_count, nullptr, gain_box, gain_cls, gain_dfl, out5, fg_out, gt_idx_out,
                              tscore_out, ws, ws_bytes, dtype, s);
}
// Same, with the per-call counter of the assigner's alpha schedule read from device memory: a captured (hipGraph) training step replays
// with a counter the host advances between replays.
extern "C" int mgdt_detect_loss_fwd_dev(const mgdt_view* const* feats, const float* strides, int n_levels, int reg_max, int nc,
                                        const float* gt, int n_gt, const int32_t* call_count_dev, float gain_box, float gain_cls, float gain_dfl,
                                        float* out5, unsigned char* fg_out, int32_t* gt_idx_out, float* tscore_out, void* ws,
                                        size_t ws_bytes, int dtype, mgdt_stream s) {
  if (!call_count_dev) MGDT_FAIL(MGDT_BAD_ARG, "detect_loss_fwd_dev: call_count_dev is NULL");
  return detect_loss_fwd_impl(feats, strides, n_levels, reg_max, nc, gt, n_gt, 0, call_count_dev, gain_box, gain_cls, gain_dfl, out5, fg_out, gt_idx_out,
                              tscore_out, ws, ws_bytes, dtype, s);
}

// grad_feats[l] = gscale * d(total)/d feats[l]; must follow mgdt_detect_loss_fwd with the SAME ws / out5 / arguments.
extern "C" int mgdt_detect_loss_bwd(const mgdt_view* const* feats, const mgdt_view* const* grads, const float* strides, int n_levels,
                                    int reg_max, int nc, const float* gt, int n_gt, float gain_box, float gain_cls, float gain_dfl,
                                    float gscale, const float* out5, void* ws, size_t ws_bytes, int dtype, mgdt_stream s) {
  if (!feats || !grads || !strides || n_levels < 1 || n_levels > 8 || !out5 || !ws) MGDT_FAIL(MGDT_BAD_ARG, "detect_loss_bwd: null/empty argument");
  const int B = feats[0]->n, no = 4 * reg_max + nc;
  int A = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (!view_ok(feats[l]) || !view_ok(grads[l]) || feats[l]->sc != 1 || grads[l]->sc != 1 || feats[l]->c != no || grads[l]->c != no ||
        grads[l]->n != B || grads[l]->h != feats[l]->h || grads[l]->w != feats[l]->w)
      MGDT_FAIL(MGDT_BAD_SHAPE, "detect_loss_bwd: level %d views must be matching NHWC (B,%d,H,W)", l, no);
    A += feats[l]->h * feats[l]->w;
  }
  if (ws_bytes < mgdt_detect_loss_workspace_bytes(B, A, n_gt)) MGDT_FAIL(MGDT_WORKSPACE, "detect_loss_bwd: workspace too small");
  hipStream_t st = (hipStream_t)s;
  const int N = std::max(n_gt, 1);
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.R = reg_max; a.nc = nc; a.A = A; a.N = n_gt; a.gt = gt;
  a.gain_box = gain_box; a.gain_cls = gain_cls; a.gain_dfl = gain_dfl;
  float* f = (float*)ws;
  a.pbox = f; f += (size_t)B * A * 4;
  a.anc = f; f += (size_t)A * 3;
  a.align = f; f += (size_t)B * N * A;
  a.ov = f; f += (size_t)B * N * A;
  a.posmax = f; f += (size_t)B * N * 2;
  a.tscore = f; f += (size_t)B * A;
  int nblocks = 0;
  for (int l = 0; l < n_levels; ++l) nblocks += cdiv((long)B * feats[l]->h * feats[l]->w, LOSS_BLOCK);
  f += (size_t)4 * nblocks;
  a.out = (float*)out5;
  a.gt_idx = (int*)f; f += (size_t)B * A;
  a.mtopk = (unsigned char*)f;
  a.fg = a.mtopk + (size_t)B * N * A;
  int a_off = 0;
  for (int l = 0; l < n_levels; ++l) {
    fill_level(a, feats[l], strides[l], a_off);
    a.grad = grads[l]->p; a.gsn = grads[l]->sn; a.gsh = grads[l]->sh; a.gsw = grads[l]->sw;
    long tot = (long)B * a.H * a.W;
    const mgdt_view* fv = feats[l];
    const mgdt_view* gv = grads[l];
    const bool vec = nc % 4 == 0 && fv->sw % 4 == 0 && fv->sh % 4 == 0 && fv->sn % 4 == 0 && gv->sw % 4 == 0 && gv->sh % 4 == 0 && gv->sn % 4 == 0 &&
                     (uintptr_t)fv->p % (4 * dtype_size(dtype)) == 0 && (uintptr_t)gv->p % (4 * dtype_size(dtype)) == 0;
    if (vec) {
      const long items = tot * (1 + nc / 4);
      MGDT_DISPATCH_DTYPE(dtype, (loss_bwd_vec_kernel<T><<<(int)std::min<long>(cdiv(items, 256), 16384), 256, 0, st>>>(a, gscale)));
    } else
      MGDT_DISPATCH_DTYPE(dtype, (loss_bwd_kernel<T><<<cdiv(tot, 256), 256, 0, st>>>(a, gscale)));
    a_off += a.H * a.W;
  }
  MGDT_CHECK_LAUNCH("detect_loss_bwd");
  return MGDT_OK;
}
