// Box helpers, validator reductions and the predictor's preprocess of the detection path on the device.
// Reference: yolo/utils/ops.py (xywh2xyxy :362-377, xyxy2xywh :345-359, scale_boxes :90-117, clip_boxes :269-285),
// yolo/utils/metrics.py (box_iou :52-72, bbox_iou :75-128, compute_ap :377-407, ap_per_class :410-497),
// yolo/engine/predictor.py:115-130 + yolo/data/augment.py:538-593 (LetterBox).  Compiled with -ffp-contract=off: the arithmetic is
// the reference's expression order in IEEE fp32 / fp64 without fused multiply-adds.
#include "common.h"

static inline int bx_grid(long n) { return (int)std::min<long>((n + 255) / 256, 4096); }

// ---------------------------------------------------------------------------------------------- xywh <-> xyxy on rows of `row` floats
__global__ void box_convert_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int row, int mode) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* p = in + i * row;
    float* o = out + i * row;
    const float a = p[0], b = p[1], c = p[2], d = p[3];
    if (mode == 0) { o[0] = a - c / 2; o[1] = b - d / 2; o[2] = a + c / 2; o[3] = b + d / 2; }       // xywh2xyxy
    else { o[0] = (a + c) / 2; o[1] = (b + d) / 2; o[2] = c - a; o[3] = d - b; }                       // xyxy2xywh
    for (int k = 4; k < row; ++k) o[k] = p[k];
  }
}
extern "C" int mgdt_box_convert(const float* in, float* out, long n, int row, int mode, mgdt_stream s) {
  if (!in || !out || n < 0 || row < 4 || (mode != 0 && mode != 1)) MGDT_FAIL(MGDT_BAD_ARG, "box_convert: bad argument");
  if (n == 0) return MGDT_OK;
  box_convert_kernel<<<bx_grid(n), 256, 0, (hipStream_t)s>>>(in, out, n, row, mode);
  MGDT_CHECK_LAUNCH("box_convert");
  return MGDT_OK;
}

// ---------------------------------------------------------------------------------------------- box_iou: (N,4) x (M,4) -> (N,M)
__global__ void box_iou_kernel(const float* __restrict__ b1, int n, const float* __restrict__ b2, int m, float eps, float* __restrict__ out) {
  const long total = (long)n * m;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / m), c = (int)(i - (long)r * m);
    const float ax1 = b1[r * 4], ay1 = b1[r * 4 + 1], ax2 = b1[r * 4 + 2], ay2 = b1[r * 4 + 3];
    const float bx1 = b2[c * 4], by1 = b2[c * 4 + 1], bx2 = b2[c * 4 + 2], by2 = b2[c * 4 + 3];
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
    const float inter = iw * ih;
    out[i] = inter / ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter + eps);
  }
}
extern "C" int mgdt_box_iou(const float* b1, int n, const float* b2, int m, float eps, float* out, mgdt_stream s) {
  if (n < 0 || m < 0 || ((long)n * m > 0 && (!b1 || !b2 || !out))) MGDT_FAIL(MGDT_BAD_ARG, "box_iou: bad argument");
  if ((long)n * m == 0) return MGDT_OK;
  box_iou_kernel<<<bx_grid((long)n * m), 256, 0, (hipStream_t)s>>>(b1, n, b2, m, eps, out);
  MGDT_CHECK_LAUNCH("box_iou");
  return MGDT_OK;
}

// ---------------------------------------------------------------------------------------------- bbox_iou (IoU / GIoU / DIoU / CIoU), row i of box1 (stride s1: 0 broadcasts one box) vs row i of box2
__global__ void bbox_iou_kernel(const float* __restrict__ b1, int s1, const float* __restrict__ b2, int s2, long n, int xywh, int mode, float eps,
                                float* __restrict__ out) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* p = b1 + i * s1;
    const float* q = b2 + i * s2;
    float b1_x1, b1_y1, b1_x2, b1_y2, b2_x1, b2_y1, b2_x2, b2_y2, w1, h1, w2, h2;
    if (xywh) {
      w1 = p[2]; h1 = p[3]; w2 = q[2]; h2 = q[3];
      const float w1_ = w1 / 2, h1_ = h1 / 2, w2_ = w2 / 2, h2_ = h2 / 2;
      b1_x1 = p[0] - w1_; b1_x2 = p[0] + w1_; b1_y1 = p[1] - h1_; b1_y2 = p[1] + h1_;
      b2_x1 = q[0] - w2_; b2_x2 = q[0] + w2_; b2_y1 = q[1] - h2_; b2_y2 = q[1] + h2_;
    } else {
      b1_x1 = p[0]; b1_y1 = p[1]; b1_x2 = p[2]; b1_y2 = p[3];
      b2_x1 = q[0]; b2_y1 = q[1]; b2_x2 = q[2]; b2_y2 = q[3];
      w1 = b1_x2 - b1_x1; h1 = b1_y2 - b1_y1 + eps;
      w2 = b2_x2 - b2_x1; h2 = b2_y2 - b2_y1 + eps;
    }
    const float inter = fmaxf(fminf(b1_x2, b2_x2) - fmaxf(b1_x1, b2_x1), 0.f) * fmaxf(fminf(b1_y2, b2_y2) - fmaxf(b1_y1, b2_y1), 0.f);
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    float r = iou;
    if (mode) {
      const float cw = fmaxf(b1_x2, b2_x2) - fminf(b1_x1, b2_x1), ch = fmaxf(b1_y2, b2_y2) - fminf(b1_y1, b2_y1);
      if (mode >= 2) {                                   // 2 = DIoU, 3 = CIoU
        const float c2 = cw * cw + ch * ch + eps;
        const float dx = b2_x1 + b2_x2 - b1_x1 - b1_x2, dy = b2_y1 + b2_y2 - b1_y1 - b1_y2;
        const float rho2 = (dx * dx + dy * dy) / 4;
        if (mode == 3) {
          const float da = atanf(w2 / h2) - atanf(w1 / h1);
          const float v = (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846)) * (da * da);
          const float alpha = v / (v - iou + (1 + eps));
          r = iou - (rho2 / c2 + v * alpha);
        } else r = iou - rho2 / c2;
      } else {                                           // 1 = GIoU
        const float c_area = cw * ch + eps;
        r = iou - (c_area - uni) / c_area;
      }
    }
    out[i] = r;
  }
}
extern "C" int mgdt_bbox_iou(const float* b1, int stride1, const float* b2, int stride2, long n, int xywh, int mode, float eps, float* out, mgdt_stream s) {
  if (n < 0 || mode < 0 || mode > 3 || (n > 0 && (!b1 || !b2 || !out))) MGDT_FAIL(MGDT_BAD_ARG, "bbox_iou: bad argument");
  if (n == 0) return MGDT_OK;
  bbox_iou_kernel<<<bx_grid(n), 256, 0, (hipStream_t)s>>>(b1, stride1, b2, stride2, n, xywh, mode, eps, out);
  MGDT_CHECK_LAUNCH("bbox_iou");
  return MGDT_OK;
}

// ---------------------------------------------------------------------------------------------- scale_boxes (+ clip_boxes), in place on rows of `row` floats
__global__ void scale_boxes_kernel(float* __restrict__ b, long n, int row, float gain, float padx, float pady, float h0, float w0) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float* p = b + i * row;
    float x1 = (p[0] - padx) / gain, y1 = (p[1] - pady) / gain, x2 = (p[2] - padx) / gain, y2 = (p[3] - pady) / gain;
    p[0] = fminf(fmaxf(x1, 0.f), w0); p[1] = fminf(fmaxf(y1, 0.f), h0); p[2] = fminf(fmaxf(x2, 0.f), w0); p[3] = fminf(fmaxf(y2, 0.f), h0);
  }
}
extern "C" int mgdt_scale_boxes(float* boxes, long n, int row, float gain, float padx, float pady, float h0, float w0, mgdt_stream s) {
  if (n < 0 || row < 4 || (n > 0 && !boxes) || !(gain > 0.f)) MGDT_FAIL(MGDT_BAD_ARG, "scale_boxes: bad argument");
  if (n == 0) return MGDT_OK;
  scale_boxes_kernel<<<bx_grid(n), 256, 0, (hipStream_t)s>>>(boxes, n, row, gain, padx, pady, h0, w0);
  MGDT_CHECK_LAUNCH("scale_boxes");
  return MGDT_OK;
}

// ---------------------------------------------------------------------------------------------- predictor preprocess: LetterBox + BGR->RGB + HWC->CHW
// src: one uint8 H x W x 3 image (BGR, row pitch src_pitch bytes); dst: uint8 [3][dh][dw] plane set of the batch tensor (RGB planes).
// Resize (when the un-padded size differs) follows cv2.resize(INTER_LINEAR) for 8-bit images: source coordinate (d + 0.5) * scale - 0.5,
// clamped taps, 11-bit fixed-point weights, two-pass rounding ((x * wy >> 4) ... + 2 >> 2 as in OpenCV's VResizeLinear) - cv2 is not
// installed here, so that branch is PARITY UNPINNED; the no-resize branch (copy + 114 border) is exact by construction.
__global__ void letterbox_kernel(const uint8_t* __restrict__ src, int sh, int sw, long src_pitch, uint8_t* __restrict__ dst, int dh, int dw, int nh, int nw,
                                 int top, int left, float scale_x, float scale_y, int resize) {
  const long total = (long)dh * dw;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int y = (int)(i / dw), x = (int)(i - (long)y * dw);
    const int uy = y - top, ux = x - left;
    int bgr[3] = {114, 114, 114};
    if ((unsigned)uy < (unsigned)nh && (unsigned)ux < (unsigned)nw) {
      if (!resize) {
        const uint8_t* p = src + (long)uy * src_pitch + (long)ux * 3;
        bgr[0] = p[0]; bgr[1] = p[1]; bgr[2] = p[2];
      } else {
        float fx = (ux + 0.5f) * scale_x - 0.5f, fy = (uy + 0.5f) * scale_y - 0.5f;
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= sx; fy -= sy;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= sh - 1) { fy = 0.f; sy = sh - 1; }
        const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
        const int ax1 = (int)lrintf(fx * 2048.f), ax0 = 2048 - ax1, ay1 = (int)lrintf(fy * 2048.f), ay0 = 2048 - ay1;
        const uint8_t* r0 = src + (long)sy * src_pitch;
        const uint8_t* r1 = src + (long)sy1 * src_pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int t0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1, t1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
          bgr[c] = ((((ay0 * (t0 >> 4)) >> 16) + ((ay1 * (t1 >> 4)) >> 16) + 2) >> 2);
        }
      }
    }
    dst[i] = (uint8_t)bgr[2];                       // R plane  (im[..., ::-1]: BGR -> RGB, predictor.py:124)
    dst[total + i] = (uint8_t)bgr[1];               // G
    dst[2 * total + i] = (uint8_t)bgr[0];           // B
  }
}
/* One image of the batch.  new_h x new_w = the un-padded (resized) size, top/left = the border offsets LetterBox computed. */
extern "C" int mgdt_letterbox_fwd(const void* src_hwc_bgr, int sh, int sw, long src_pitch, void* dst_chw_rgb, int dh, int dw, int new_h, int new_w, int top,
                                  int left, mgdt_stream s) {
  if (!src_hwc_bgr || !dst_chw_rgb || sh < 1 || sw < 1 || dh < 1 || dw < 1 || new_h < 1 || new_w < 1 || top < 0 || left < 0 || top + new_h > dh || left + new_w > dw)
    MGDT_FAIL(MGDT_BAD_ARG, "letterbox: bad geometry %dx%d -> %dx%d in %dx%d at (%d,%d)", sh, sw, new_h, new_w, dh, dw, top, left);
  const int resize = (new_h != sh || new_w != sw);
  letterbox_kernel<<<bx_grid((long)dh * dw), 256, 0, (hipStream_t)s>>>((const uint8_t*)src_hwc_bgr, sh, sw, src_pitch, (uint8_t*)dst_chw_rgb, dh, dw, new_h, new_w, top,
                                                                         left, (float)sw / new_w, (float)sh / new_h, resize);
  MGDT_CHECK_LAUNCH("letterbox_fwd");
  return MGDT_OK;
}

// ---------------------------------------------------------------------------------------------- ap_per_class (the per-class part)
// Detections arrive grouped by class (segment ci = [seg[ci], seg[ci+1])) and, inside a class, by descending confidence.  One workgroup per
// class: per IoU level the cumulative TP / FP counts, recall = tpc / (n_l + eps), precision = tpc / (tpc + fpc) in fp64, the precision
// envelope, the 101-point interpolated AP with numpy's own interp rule (last knot <= x, slope form) and trapz summed in numpy's pairwise
// order (so the numbers equal the reference's bit for bit); at level 0 also the 1000-point recall / precision-vs-confidence curves.
#define AP_THREADS 256
__device__ double np_interp(double x, const double* xp, const double* fp, int n, double left, double right) {
  if (x > xp[n - 1]) return right;
  if (x < xp[0]) return left;
  int lo = 0, hi = n;                                 // last j with xp[j] <= x
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (x >= xp[mid]) lo = mid + 1; else hi = mid; }
  const int j = lo - 1;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (isnan(r)) { r = slope * (x - xp[j + 1]) + fp[j + 1]; if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j]; }
  return r;
}

__global__ __launch_bounds__(AP_THREADS) void ap_class_kernel(const uint8_t* __restrict__ tp, const float* __restrict__ conf, const int* __restrict__ seg,
                                                              const int* __restrict__ nlab, int T, const double* __restrict__ x101, const double* __restrict__ px,
                                                              double eps, double* __restrict__ ws, double* __restrict__ ap, double* __restrict__ pcur,
                                                              double* __restrict__ rcur) {
  const int ci = blockIdx.x, tid = threadIdx.x;
  const int s0 = seg[ci], n = seg[ci + 1] - s0, nl = nlab[ci];
  if (n == 0 || nl == 0) return;                     // ap / curves stay zero (metrics.py:452-453)
  // workspace of this class: mrec[n+2] | mpre[n+2] | xconf[n] (= -conf in fp64)
  double* mrec = ws + (size_t)3 * s0 + (size_t)4 * ci;
  double* mpre = mrec + n + 2;
  double* xc = mpre + n + 2;
  __shared__ int s_run;
  __shared__ int s_part[AP_THREADS];
  __shared__ double s_vals[101];
  for (int j = 0; j < T; ++j) {
    // inclusive scan of tp[:, j] over the segment, AP_THREADS elements per round
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < n; base += AP_THREADS) {
      const int k = base + tid;
      const int v = k < n ? (int)tp[(size_t)(s0 + k) * T + j] : 0;
      s_part[tid] = v;
      __syncthreads();
      for (int d = 1; d < AP_THREADS; d <<= 1) {
        const int t = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
      }
      const int tpc = s_run + s_part[tid];
      if (k < n) {
        const int fpc = (k + 1) - tpc;
        mrec[k + 1] = (double)tpc / ((double)nl + eps);
        mpre[k + 1] = (double)tpc / (double)(tpc + fpc);
        if (j == 0) xc[k] = -(double)conf[s0 + k];
      }
      __syncthreads();
      if (tid == AP_THREADS - 1) s_run += s_part[tid];
      __syncthreads();
    }
    if (j == 0) {
      // recall / precision vs confidence at the first IoU level (metrics.py:460-465): interp(-px, -conf, curve, left)
      for (int q = tid; q < 1000; q += AP_THREADS) {
        rcur[(size_t)ci * 1000 + q] = np_interp(-px[q], xc, mrec + 1, n, 0.0, mrec[n]);
        pcur[(size_t)ci * 1000 + q] = np_interp(-px[q], xc, mpre + 1, n, 1.0, mpre[n]);
      }
      __syncthreads();
    }
    if (tid == 0) {                                   // sentinels + precision envelope (reverse running maximum), serial: n is small per class
      mrec[0] = 0.0; mrec[n + 1] = 1.0; mpre[0] = 1.0; mpre[n + 1] = 0.0;
      double m = 0.0;
      for (int k = n + 1; k >= 0; --k) { m = fmax(m, mpre[k]); mpre[k] = m; }
    }
    __syncthreads();
    if (tid < 101) s_vals[tid] = np_interp(x101[tid], mrec, mpre, n + 2, mpre[0], mpre[n + 1]);
    __syncthreads();
    if (tid == 0) {
      // np.trapz: (d * (y[1:] + y[:-1]) / 2.0).sum() with numpy's pairwise summation of 100 terms (8 running sums, then the tail)
      double t[100];
      for (int i = 0; i < 100; ++i) t[i] = (x101[i + 1] - x101[i]) * (s_vals[i + 1] + s_vals[i]) / 2.0;
      double r[8];
      for (int i = 0; i < 8; ++i) r[i] = t[i];
      int i = 8;
      for (; i < 100 - (100 % 8); i += 8)
        for (int u = 0; u < 8; ++u) r[u] += t[i + u];
      double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; i < 100; ++i) res += t[i];
      ap[(size_t)ci * T + j] = res;
    }
    __syncthreads();
  }
}

extern "C" size_t mgdt_ap_workspace_bytes(int n_det, int n_cls) { return ((size_t)3 * n_det + (size_t)4 * n_cls + 8) * sizeof(double); }
/* tp: uint8 [n_det][T] and conf fp32 [n_det], both already grouped by class (seg int32 [n_cls + 1]) and by descending confidence inside a class;
 * nlab int32 [n_cls] labels per class; x101 = linspace(0,1,101), px = linspace(0,1,1000) (fp64, device); outputs (zero-initialised by the caller):
 * ap fp64 [n_cls][T], pcur / rcur fp64 [n_cls][1000]. */
extern "C" int mgdt_ap_per_class(const void* tp, const float* conf, const int32_t* seg, const int32_t* nlab, int n_det, int n_cls, int T, const double* x101,
                                 const double* px, double eps, void* ws, double* ap, double* pcur, double* rcur, mgdt_stream s) {
  if (n_cls < 0 || T < 1 || n_det < 0) MGDT_FAIL(MGDT_BAD_ARG, "ap_per_class: bad sizes");
  if (n_cls == 0 || n_det == 0) return MGDT_OK;
  if (!tp || !conf || !seg || !nlab || !x101 || !px || !ws || !ap || !pcur || !rcur) MGDT_FAIL(MGDT_BAD_ARG, "ap_per_class: null pointer");
  ap_class_kernel<<<n_cls, AP_THREADS, 0, (hipStream_t)s>>>((const uint8_t*)tp, conf, seg, nlab, T, x101, px, eps, (double*)ws, ap, pcur, rcur);
  MGDT_CHECK_LAUNCH("ap_per_class");
  return MGDT_OK;
}
