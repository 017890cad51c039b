"""TEST INFRASTRUCTURE ONLY - CPU (numpy) restatement of the validator's AP reduction, the box rescaling and the predictor's LetterBox
(SURVEY section 8(f) ranks 1-2).  Pinned by tests/golden/{metrics_ap,boxes2,letterbox}.npz, generated from the reference's own functions
(tests/golden/gen_golden.py: metrics_ap / boxes2 / letterbox_geom).  `resize_linear_u8` restates cv2.resize(INTER_LINEAR) for 8-bit images
from OpenCV's published algorithm; cv2 is not installed here and the reference vendors no copy: PARITY UNPINNED for that one function."""
import numpy as np


def smooth(y, f=0.05):
    """yolo/utils/metrics.py:293-298."""
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode='valid')


def compute_ap(recall, precision):
    """yolo/utils/metrics.py:377-407 (method 'interp')."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    ap = (np.diff(x) * (y[1:] + y[:-1]) / 2.0).sum()           # np.trapz(y, x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """yolo/utils/metrics.py:410-497 without the plotting."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l, n_p = nt[ci], i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + eps)
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + eps)
    i = smooth(f1.mean(0), 0.1).argmax()
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tp = (r * nt).round()
    fp = (tp / (p + eps) - tp).round()
    return tp, fp, p, r, f1, ap, unique_classes.astype(int)


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """yolo/utils/ops.py:90-117 + clip_boxes :269-285 on a float32 (n, 4) array (returns a new array)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    b = np.array(boxes, np.float32)
    b[:, [0, 2]] -= np.float32(pad[0])
    b[:, [1, 3]] -= np.float32(pad[1])
    b[:, :4] /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, img0_shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, img0_shape[0])
    return b


def letterbox_geometry(shape, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, stride=32):
    """yolo/data/augment.py:554-583 -> (out_h, out_w, unpad_h, unpad_w, top, left, resized?)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return (new_unpad[1] + top + bottom, new_unpad[0] + left + right, new_unpad[1], new_unpad[0], top, left, int(tuple(shape[::-1]) != new_unpad))


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC images, from OpenCV's published resize.cpp:
    source coordinate (d + 0.5) * scale - 0.5, taps clamped to the image, weights in 11-bit fixed point (cvRound(f * 2048)), horizontal pass in
    int32, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2.  PARITY UNPINNED (cv2 absent)."""
    h, w = img.shape[:2]
    sx, sy = w / new_w, h / new_h

    def taps(n_dst, n_src, scale):
        f = (np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * np.float32(scale) - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0).astype(np.float32)
        lo = i0 < 0
        f[lo], i0[lo] = 0, 0
        hi = i0 >= n_src - 1
        f[hi], i0[hi] = 0, n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return i0, i1, 2048 - a1, a1
    x0, x1, ax0, ax1 = taps(new_w, w, sx)
    y0, y1, ay0, ay1 = taps(new_h, h, sy)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]            # (h, new_w, c)
    t0, t1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (t0 >> 4)) >> 16) + ((ay1[:, None, None] * (t1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def letterbox(img_bgr, new_shape=(640, 640), auto=False, stride=32):
    """LetterBox + predictor.py:123-125 for one image: uint8 (h, w, 3) BGR -> uint8 (3, H, W) RGB."""
    oh, ow, nh, nw, top, left, rs = letterbox_geometry(img_bgr.shape[:2], new_shape, auto, stride=stride)
    im = resize_linear_u8(img_bgr, nw, nh) if rs else img_bgr
    out = np.full((oh, ow, 3), 114, np.uint8)
    out[top:top + nh, left:left + nw] = im
    return np.ascontiguousarray(out[..., ::-1].transpose(2, 0, 1))
