"""Validator-side hot step (reference: yolo/v8/detect/val.py): matching predictions to labels on the device.

Only the per-batch compute of the reference's DetectionValidator is mirrored here - `_process_batch` (val.py:152-175) with the same
signature and return, plus a batched form that consumes the NMS kernel's output directly.  Dataset handling, plotting, JSON export and the
final `ap_per_class` reduction (numpy, once per validation run) are host glue outside the hot path.
"""
import torch

from .... import ops as hip

__all__ = ('DetectionValidator',)


class DetectionValidator:
    def __init__(self, device='cuda:0'):
        self.device = torch.device(device)
        self.iouv = torch.linspace(0.5, 0.95, 10, device=self.device)     # val.py:60: IoU vector for mAP@0.5:0.95
        self.niou = self.iouv.numel()

    def _process_batch(self, detections, labels):
        """detections (N, 6) [x1, y1, x2, y2, conf, cls], labels (M, 5) [cls, x1, y1, x2, y2] -> correct (N, 10) bool on detections.device."""
        n, m = detections.shape[0], labels.shape[0]
        if n == 0 or m == 0:
            return torch.zeros(n, self.niou, dtype=torch.bool, device=detections.device)
        det = detections.float().contiguous()[None]
        lab = labels.float().contiguous()[None]
        cnt = lambda k: torch.full((1,), k, dtype=torch.int32, device=det.device)
        return hip.val_match(det, cnt(n), lab, cnt(m), self.iouv.to(det.device))[0]

    # ---- the per-batch metric update and the final reduction (val.py:73-117, :123-131; metrics.py:410-497) -------------------------------
    def init_metrics(self, nc=80, conf=0.001, iou=0.7, max_det=300):
        self.nc, self.conf, self.iou, self.max_det = nc, conf, iou, max_det     # validator.py:85-86 / default.yaml
        self.seen, self.stats = 0, []

    def postprocess(self, preds):
        """val.py:63-71: NMS with the validator's settings (multi_label)."""
        from ...utils import ops
        return ops.non_max_suppression(preds, self.conf, self.iou, multi_label=True, max_det=self.max_det)

    def update_metrics(self, preds, batch):
        """preds: list of (n_i, 6) tensors from `postprocess`; batch: the dataloader dict (img, cls, bboxes, batch_idx, ori_shape, ratio_pad).
        Everything stays on the device: boxes are rescaled to native space by mgdt_scale_boxes, matched by mgdt_val_match_fwd."""
        from ...utils import ops
        dev = self.device
        bidx = batch['batch_idx'].to(dev)
        height, width = batch['img'].shape[2:]
        whwh = torch.tensor((width, height, width, height), dtype=torch.float32, device=dev)
        for si, pred in enumerate(preds):
            idx = bidx == si
            cls = batch['cls'].to(dev)[idx].float()
            bbox = batch['bboxes'].to(dev)[idx].float()
            nl, npr = cls.shape[0], pred.shape[0]
            shape = batch['ori_shape'][si]
            correct = torch.zeros(npr, self.niou, dtype=torch.bool, device=dev)
            self.seen += 1
            if npr == 0:
                if nl:
                    self.stats.append((correct, *torch.zeros((2, 0), device=dev), cls.squeeze(-1)))
                continue
            predn = pred.clone()
            ops.scale_boxes(batch['img'][si].shape[1:], predn, shape, ratio_pad=batch['ratio_pad'][si])      # native-space pred
            if nl:
                tbox = ops.xywh2xyxy(bbox.contiguous()) * whwh
                ops.scale_boxes(batch['img'][si].shape[1:], tbox, shape, ratio_pad=batch['ratio_pad'][si])   # native-space labels
                labelsn = torch.cat((cls.view(-1, 1), tbox), 1)
                correct = self._process_batch(predn, labelsn)
            self.stats.append((correct, pred[:, 4], pred[:, 5], cls.squeeze(-1)))

    def get_stats(self):
        """val.py:123-131 + DetMetrics.process: (tp, fp, p, r, f1, ap, ap_class) per class and the summary dict."""
        import numpy as np
        from ...utils.metrics import ap_per_class
        if not self.stats:
            return {}
        tp, conf, pcls, tcls = [torch.cat(x, 0) for x in zip(*self.stats)]
        self.nt_per_class = np.bincount(tcls.cpu().numpy().astype(int), minlength=self.nc)
        if not (len(tp) and bool(tp.any())):
            return {'metrics/precision(B)': 0.0, 'metrics/recall(B)': 0.0, 'metrics/mAP50(B)': 0.0, 'metrics/mAP50-95(B)': 0.0}
        _, _, p, r, f1, ap, ap_class = ap_per_class(tp, conf, pcls, tcls, device=self.device)
        self.ap_class_index, self.ap = ap_class, ap
        return {'metrics/precision(B)': float(p.mean()), 'metrics/recall(B)': float(r.mean()), 'metrics/mAP50(B)': float(ap[:, 0].mean()),
                'metrics/mAP50-95(B)': float(ap.mean())}

    def match_batch(self, det, ndet, labels, nlab):
        """Whole batch in one launch: det (B, max_det, 6) + ndet (B,) int32 exactly as `mgdt_yolo_amd.ops.nms` returns them, labels
        (B, max_lab, 5) zero-padded + nlab (B,) int32 -> correct (B, max_det, 10) bool (rows past ndet are False)."""
        return hip.val_match(det, ndet, labels, nlab, self.iouv.to(det.device))
