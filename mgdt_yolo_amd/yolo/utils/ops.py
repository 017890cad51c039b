"""Detection post-processing with the reference's names (reference: yolo/utils/ops.py)."""
import torch

from ... import ops as hip


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
