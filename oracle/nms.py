"""NMS oracle: TEST INFRASTRUCTURE (see oracle/__init__.py).

`non_max_suppression` restates yolo/utils/ops.py:136-266 stage by stage in numpy float32 (IEEE, no
FMA), so integer outputs (kept candidate indices / classes) can be compared bit-exactly.
Deliberate, documented divergences from the reference:
  * the wall-clock `time_limit` break (ops.py:198,262-264) is not replicated (non-deterministic);
  * the pre-NMS sort (ops.py:244, `argsort(descending=True)`, not stable) is made deterministic:
    ties in score keep the lower candidate index first;
  * `greedy_nms` restates torchvision.ops.nms (call site ops.py:249): torchvision is absent from the
    reference tree and from this image, so that one call is PARITY UNPINNED; semantics follow the
    published CPU kernel: visit boxes in descending score, suppress j when
    inter / (area_i + area_j - inter) > thr (strict), areas = (x2-x1)*(y2-y1), no eps, no +1.
"""
import numpy as np

F32 = np.float32


def greedy_nms(boxes, iou_thres):
    """boxes (n,4) float32 xyxy ALREADY sorted by descending score. Returns kept positions (ascending)."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2 = (boxes[:, i].astype(F32) for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = F32(iou_thres)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(F32(0), xx2 - xx1)
        h = np.maximum(F32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return np.asarray(keep, np.int64)


_CLIB = None


def greedy_nms_c(boxes, iou_thres, limit=0):
    """oracle/csrc/greedy_nms.c through ctypes: the same arithmetic as `greedy_nms` in compiled code (what the reference's CPU path has
    at this call: torchvision's C++ kernel).  `limit` > 0 stops after that many keepers (a prefix of the full answer).  Used by the
    timed CPU baseline; checked equal to `greedy_nms` in tests/test_oracle_golden.py."""
    global _CLIB
    import ctypes as C
    import os
    if _CLIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'liboracle.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} is missing: run `make -C oracle` (or __graft_entry__.build())')
        _CLIB = C.CDLL(path)
        _CLIB.oracle_greedy_nms.restype = C.c_int64
        _CLIB.oracle_greedy_nms.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int64, C.c_void_p]
    b = np.ascontiguousarray(boxes, F32)
    keep = np.empty(max(len(b), 1), np.int64)
    n = _CLIB.oracle_greedy_nms(b.ctypes.data, len(b), float(F32(iou_thres)), int(limit), keep.ctypes.data)
    assert n >= 0
    return keep[:n].copy()


def nms_candidates(pred_img, conf_thres, multi_label, classes=None, nc=None):
    """Stages ops.py:201-238 for one image.  pred_img: (4+nc, A) float32.

    Returns (boxes_xyxy (n,4), conf (n,), cls (n,) int64, anchor (n,) int64) in the reference's
    candidate order (anchor-major, then class for multi_label).
    """
    p = np.asarray(pred_img, F32)
    nc = nc or p.shape[0] - 4
    scores = p[4:4 + nc]                       # (nc, A)
    xc = scores.max(0) > F32(conf_thres)       # ops.py:191
    idx = np.nonzero(xc)[0]
    x = p[:, idx].T                            # (m, 4+nc)
    half_w, half_h = x[:, 2] / F32(2), x[:, 3] / F32(2)
    box = np.stack([x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h], 1).astype(F32)
    cls_sc = x[:, 4:4 + nc]
    if multi_label and nc > 1:
        i, j = np.nonzero(cls_sc > F32(conf_thres))   # row-major == torch nonzero order
        b, conf, cls, anc = box[i], cls_sc[i, j], j.astype(np.int64), idx[i]
    else:
        j = cls_sc.argmax(1) if len(idx) else np.zeros((0,), np.int64)
        conf = cls_sc[np.arange(len(idx)), j] if len(idx) else np.zeros((0,), F32)
        m = conf > F32(conf_thres)
        b, conf, cls, anc = box[m], conf[m], j[m].astype(np.int64), idx[m]
    if classes is not None:
        m = np.isin(cls, np.asarray(classes))
        b, conf, cls, anc = b[m], conf[m], cls[m], anc[m]
    return b, conf.astype(F32), cls, anc.astype(np.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, nc=0, max_nms=30000, max_wh=7680, return_index=False, compiled=False):
    """yolo/utils/ops.py:136-266. prediction (B, 4+nc, A) -> list of (n_i, 6) float32 arrays.

    With return_index=True also returns, per image, (anchor index, class) int64 arrays of the kept rows.
    """
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    pred = np.asarray(prediction, F32)
    out, kept = [], []
    for p in pred:
        b, conf, cls, anc = nms_candidates(p, conf_thres, multi_label, classes, nc or None)
        if len(conf) == 0:
            out.append(np.zeros((0, 6), F32))
            kept.append((np.zeros((0,), np.int64), np.zeros((0,), np.int64)))
            continue
        order = np.argsort(-conf, kind='stable')[:max_nms]          # ops.py:244 (+ tie rule)
        b, conf, cls, anc = b[order], conf[order], cls[order], anc[order]
        c = cls.astype(F32) * F32(0 if agnostic else max_wh)         # ops.py:247
        shifted = (b + c[:, None]).astype(F32)
        keep = (greedy_nms_c(shifted, iou_thres, max_det) if compiled else greedy_nms(shifted, iou_thres))[:max_det]   # ops.py:248-250
        out.append(np.concatenate([b[keep], conf[keep, None], cls[keep, None].astype(F32)], 1).astype(F32))
        kept.append((anc[keep], cls[keep]))
    return (out, kept) if return_index else out
