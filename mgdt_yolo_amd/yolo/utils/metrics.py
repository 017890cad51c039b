"""Box overlap functions and the validator's AP reduction with the reference's names (reference: yolo/utils/metrics.py).

`box_iou`, `bbox_iou` (IoU / GIoU / DIoU / CIoU, forward values) run as HIP kernels (mgdt_box_iou / mgdt_bbox_iou).  `ap_per_class` keeps
the reference's signature and return tuple: the O(detections) part - per-class cumulative TP / FP, recall / precision curves, compute_ap's
envelope, 101-point interpolation and integration, the 1000-point P / R-vs-confidence curves - runs on the device in fp64 with numpy's own
arithmetic order (mgdt_ap_per_class: the AP matrix equals the reference's bit for bit); grouping the detections by (class, confidence) is a
device sort; the last few lines (F1 smoothing over 1000 points, arg-max, counts) are the reference's numpy expressions on a (nc, 1000) array."""
import numpy as np
import torch

from ... import _lib as L
from ... import ops as hip


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of (N, 4) and (M, 4) xyxy boxes -> (N, M) (metrics.py:52-72)."""
    hip._need_gpu(box1)
    b1, b2 = box1.float().contiguous(), box2.float().contiguous()
    out = torch.empty(b1.shape[0], b2.shape[0], dtype=torch.float32, device=b1.device)
    L.check(L.lib().mgdt_box_iou(hip.ptr(b1), b1.shape[0], hip.ptr(b2), b2.shape[0], float(eps), hip.ptr(out), hip.stream()), 'box_iou')
    return out


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU / GIoU / DIoU / CIoU of box1 (1, 4) or (n, 4) against box2 (n, 4) -> (n, 1), forward values (metrics.py:75-128).
    (The training loss has its own fused CIoU forward + backward inside mgdt_detect_loss_fwd / _bwd.)"""
    hip._need_gpu(box1)
    b1, b2 = box1.float().contiguous().view(-1, 4), box2.float().contiguous().view(-1, 4)
    n = max(b1.shape[0], b2.shape[0])
    if b1.shape[0] not in (1, n) or b2.shape[0] not in (1, n):
        raise RuntimeError(f'bbox_iou: cannot broadcast {tuple(box1.shape)} with {tuple(box2.shape)}')
    mode = 3 if CIoU else 2 if DIoU else 1 if GIoU else 0
    out = torch.empty(n, 1, dtype=torch.float32, device=b1.device)
    L.check(L.lib().mgdt_bbox_iou(hip.ptr(b1), 4 if b1.shape[0] == n and n > 1 or b1.shape[0] == n == 1 else 0, hip.ptr(b2),
                                  4 if b2.shape[0] == n and n > 1 or b2.shape[0] == n == 1 else 0, n, int(bool(xywh)), mode, float(eps), hip.ptr(out),
                                  hip.stream()), 'bbox_iou')
    return out


def smooth(y, f=0.05):
    """Box filter of fraction f (metrics.py:293-298)."""
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode='valid')


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names=(), eps=1e-16, prefix='', device=None):
    """Average precision per class (metrics.py:410-497): same arguments (numpy arrays or tensors) and the same 7-tuple
    (tp, fp, p, r, f1, ap, unique_classes).  Plotting is host tooling outside the path (plot must be False)."""
    if plot:
        raise RuntimeError('ap_per_class: plotting is host-side tooling outside the detection path')
    dev = torch.device(device or (tp.device if torch.is_tensor(tp) and tp.is_cuda else 'cuda:0'))
    t = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev).to(dt)
    tp_d, conf_d, pcls_d = t(tp, torch.uint8), t(conf, torch.float32), t(pred_cls, torch.float32)
    tcls = np.asarray(target_cls.detach().cpu() if torch.is_tensor(target_cls) else target_cls)
    unique_classes, nt = np.unique(tcls, return_counts=True)                       # metrics.py:445
    nc, T = unique_classes.shape[0], tp_d.shape[1]
    n = tp_d.shape[0]
    # group by class (ascending) and, inside a class, by descending confidence: two stable device sorts (np.argsort(-conf) in the reference;
    # equal confidences keep their input order here)
    o1 = torch.sort(conf_d, descending=True, stable=True).indices
    o2 = torch.sort(pcls_d[o1], stable=True).indices
    order = o1[o2]
    tp_s, conf_s, cls_s = tp_d[order].contiguous(), conf_d[order].contiguous(), pcls_d[order]
    uc = torch.from_numpy(unique_classes.astype(np.float32)).to(dev)
    # detections of label class c occupy [lo_c, hi_c) of the sorted arrays; classes without labels are skipped like the reference's loop
    lo, hi = torch.searchsorted(cls_s, uc, right=False), torch.searchsorted(cls_s, uc, right=True)
    counts = hi - lo
    segs = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32)
    idx = (torch.cat([torch.arange(int(a), int(b), device=dev) for a, b in zip(lo.tolist(), hi.tolist())])
           if nc and n else torch.zeros(0, dtype=torch.long, device=dev))
    tp_c, conf_c = tp_s[idx].contiguous(), conf_s[idx].contiguous()
    nd = int(tp_c.shape[0])
    x101 = torch.from_numpy(np.linspace(0, 1, 101)).to(dev)
    px_np = np.linspace(0, 1, 1000)
    px = torch.from_numpy(px_np).to(dev)
    ap = torch.zeros(nc, T, dtype=torch.float64, device=dev)
    p = torch.zeros(nc, 1000, dtype=torch.float64, device=dev)
    r = torch.zeros(nc, 1000, dtype=torch.float64, device=dev)
    if nc and nd:
        ws = torch.empty(L.lib().mgdt_ap_workspace_bytes(nd, nc), dtype=torch.uint8, device=dev)
        nlab = torch.from_numpy(nt.astype(np.int32)).to(dev)
        L.check(L.lib().mgdt_ap_per_class(hip.ptr(tp_c), hip.ptr(conf_c), hip.ptr(segs), hip.ptr(nlab), nd, nc, T, hip.ptr(x101), hip.ptr(px), float(eps),
                                          hip.ptr(ws), hip.ptr(ap), hip.ptr(p), hip.ptr(r), hip.stream()), 'ap_per_class')
    ap, p, r = ap.cpu().numpy(), p.cpu().numpy(), r.cpu().numpy()
    # the tail of the reference function, verbatim numpy on (nc, 1000) arrays (metrics.py:476-497)
    f1 = 2 * p * r / (p + r + eps)
    i = smooth(f1.mean(0), 0.1).argmax() if nc else 0
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tp_out = (r * nt).round()
    fp_out = (tp_out / (p + eps) - tp_out).round()
    return tp_out, fp_out, p, r, f1, ap, unique_classes.astype(int)
