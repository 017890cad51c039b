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

    def match_batch(self, det, ndet, labels, nlab):
        """Whole batch in one launch: det (B, max_det, 6) + ndet (B,) int32 exactly as `mgdt_yolo_amd.ops.nms` returns them, labels
        (B, max_lab, 5) zero-padded + nlab (B,) int32 -> correct (B, max_det, 10) bool (rows past ndet are False)."""
        return hip.val_match(det, ndet, labels, nlab, self.iouv.to(det.device))
