"""Detection post-processing with the reference's names (reference: yolo/utils/ops.py)."""
import torch

from ... import _lib as L
from ... import ops as hip


def _rows(x):
    """(..., k >= 4) cuda fp32 tensor -> contiguous 2-D view + row width (the reference's helpers index the last axis)."""
    hip._need_gpu(x)
    if x.dtype != torch.float32:
        raise RuntimeError('box helpers compute in float32')
    if x.shape[-1] < 4:
        raise RuntimeError(f'expected boxes with >= 4 columns, got {tuple(x.shape)}')
    return x.contiguous().view(-1, x.shape[-1]), x.shape[-1]


def _convert(x, mode):
    rows, k = _rows(x)
    out = torch.empty_like(rows)
    L.check(L.lib().mgdt_box_convert(hip.ptr(rows), hip.ptr(out), rows.shape[0], k, mode, hip.stream()), 'box_convert')
    return out.view(x.shape)


def xywh2xyxy(x):
    """(x, y, w, h) -> (x1, y1, x2, y2) on the last axis, other columns copied (ops.py:362-377)."""
    return _convert(x, 0)


def xyxy2xywh(x):
    """(x1, y1, x2, y2) -> (x, y, w, h) (ops.py:345-359)."""
    return _convert(x, 1)


def clip_boxes(boxes, shape):
    """In-place clip to the image (h, w) (ops.py:269-285)."""
    return scale_boxes(shape, boxes, shape, ratio_pad=((1.0, 1.0), (0.0, 0.0)))


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Rescale xyxy boxes (in place, like the reference) from the letter-boxed shape img1_shape (h, w) to the original img0_shape and clip
    them (ops.py:90-117).  `boxes` must be a contiguous view whose rows are >= 4 floats apart (a (n, 4) tensor or pred[:, :4] of (n, 6))."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    hip._need_gpu(boxes)
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] < 4 or boxes.stride(1) != 1:
        raise RuntimeError('scale_boxes: expected a 2-D float32 view with unit column stride')
    if boxes.shape[0]:
        L.check(L.lib().mgdt_scale_boxes(hip.ptr(boxes), boxes.shape[0], boxes.stride(0), float(gain), float(pad[0]), float(pad[1]), float(img0_shape[0]),
                                         float(img0_shape[1]), hip.stream()), 'scale_boxes')
    return boxes


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(),
                        max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680):
    """Batched NMS, same signature and return type as the reference (ops.py:136-266): list of (n_i, 6) tensors
    [x1, y1, x2, y2, conf, cls] on prediction.device.  One fused HIP launch for the whole batch + one D2H of the counts.

    Differences, all documented in DESIGN.md: `max_time_img` is accepted and ignored (no wall-clock truncation);
    score ties are ordered by candidate index; `labels` (autolabelling a-priori boxes) and mask channels (nm > 0) are
    not on the detection hot path and raise.
    """
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    if prediction.shape[1] - nc - 4:
        raise RuntimeError('non_max_suppression: mask channels (segmentation) are out of scope of the detection path')
    if labels and any(len(l) for l in labels):
        raise RuntimeError('non_max_suppression: autolabelling `labels` are out of scope of the detection path')
    if classes is not None and len(classes) == 0:
        return [torch.zeros((0, 6), device=prediction.device)] * bs
    pred = prediction if (prediction.dtype == torch.float32 and prediction.is_contiguous()) else prediction.float().contiguous()
    out, _, counts = hip.nms(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, max_nms, max_wh)
    counts = counts.tolist()   # the one host sync: result sizes
    return [out[i, :counts[i]] for i in range(bs)]


def nms_with_index(prediction, **kw):
    """Like non_max_suppression but also returns the kept anchor indices (int32) per image - used by the parity tests."""
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    kw.setdefault('conf_thres', 0.25); kw.setdefault('iou_thres', 0.45)
    out, kept, counts = hip.nms(prediction.float().contiguous(), kw['conf_thres'], kw['iou_thres'], kw.get('classes'), kw.get('agnostic', False),
                                kw.get('multi_label', False), kw.get('max_det', 300), kw.get('max_nms', 30000), kw.get('max_wh', 7680))
    counts = counts.tolist()
    return [out[i, :c] for i, c in enumerate(counts)], [kept[i, :c] for i, c in enumerate(counts)]
