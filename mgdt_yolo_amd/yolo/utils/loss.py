"""Detection criterion with the reference's name and call surface (reference: yolo/utils/loss.py:108-208).

`v8DetectionLoss(model)(preds, batch)` -> `(loss.sum() * batch_size, loss.detach())` with loss = [box, cls, dfl] (gains applied),
computed by the fused HIP assigner + loss kernels (mgdt_detect_loss_fwd); gradients w.r.t. the raw head maps come from
mgdt_detect_loss_bwd through a torch.autograd.Function.  Only the target densification (`preprocess`, a Python loop over
the batch in the reference too, loss.py:132-148) runs on the host.
"""
import types

import numpy as np
import torch

from ... import ops

DEFAULT_GAINS = types.SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)     # yolo/cfg/default.yaml:89-91


class _DetectLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, crit, gt, call_count, *feats):
        st = ops.detect_loss_fwd([f.detach() for f in feats], crit.stride_list, crit.reg_max, crit.nc, gt, call_count,
                                 (crit.hyp.box, crit.hyp.cls, crit.hyp.dfl))
        ctx.st = st
        ctx.mark_non_differentiable(st.out5)
        return st.out5[0].clone(), st.out5

    @staticmethod
    def backward(ctx, g_total, _g_out5):
        grads = ops.detect_loss_bwd(ctx.st, 1.0)
        return (None, None, None, *[g * g_total for g in grads])


class v8DetectionLoss:
    """Criterion class for computing detection training losses (this fork: reg_max 4, per-call assigner schedule)."""

    def __init__(self, model):
        m = model.model[-1]                         # Detect() module
        self.hyp = model.args if getattr(model, 'args', None) is not None and hasattr(model.args, 'box') else DEFAULT_GAINS
        self.stride = m.stride
        self.stride_list = [float(s) for s in m.stride.tolist()]
        self.nc, self.no, self.reg_max = m.nc, m.no, m.reg_max
        self.device = next(model.parameters()).device
        self.epoch = 0                              # incremented PER CALL (per batch) as in the reference (loss.py:206)
        self.use_dfl = m.reg_max > 1

    def preprocess(self, batch, batch_size, imgsz_hw):
        """batch dict -> dense (B, Nmax, 5) [cls, xyxy px] on the device (loss.py:132-148,177-181)."""
        idx = batch['batch_idx'].detach().cpu().numpy().reshape(-1).astype(np.int64)
        cls = batch['cls'].detach().cpu().numpy().reshape(-1, 1).astype(np.float32)
        box = batch['bboxes'].detach().cpu().numpy().reshape(-1, 4).astype(np.float32)
        if idx.size == 0:
            return torch.zeros(batch_size, 0, 5, device=self.device)
        counts = np.bincount(idx, minlength=batch_size)
        out = np.zeros((batch_size, int(counts.max()), 5), np.float32)
        h, w = imgsz_hw
        scale = np.array([w, h, w, h], np.float32)
        xywh = box * scale
        xyxy = np.concatenate([xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2], 1).astype(np.float32)
        rows = np.concatenate([cls, xyxy], 1)
        for j in range(batch_size):
            r = rows[idx == j]
            out[j, :len(r)] = r
        return torch.from_numpy(out).to(self.device)

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        feats = list(feats)
        b = feats[0].shape[0]
        imgsz = (feats[0].shape[2] * self.stride_list[0], feats[0].shape[3] * self.stride_list[0])
        gt = self.preprocess(batch, b, imgsz)
        if any(f.requires_grad for f in feats):
            total, out5 = _DetectLossFn.apply(self, gt, self.epoch, *feats)
        else:
            st = ops.detect_loss_fwd(feats, self.stride_list, self.reg_max, self.nc, gt, self.epoch, (self.hyp.box, self.hyp.cls, self.hyp.dfl))
            total, out5 = st.out5[0], st.out5
        self.epoch += 1
        return total, out5[1:4].detach()


def loss_and_head_grads(crit, feats, batch, gscale=1.0):
    """(loss*B, items[3], [d(loss*B)/d feats]) without torch.autograd: the entry the explicit backward pass starts from."""
    feats = list(feats)
    b = feats[0].shape[0]
    imgsz = (feats[0].shape[2] * crit.stride_list[0], feats[0].shape[3] * crit.stride_list[0])
    gt = crit.preprocess(batch, b, imgsz)
    st = ops.detect_loss_fwd(feats, crit.stride_list, crit.reg_max, crit.nc, gt, crit.epoch, (crit.hyp.box, crit.hyp.cls, crit.hyp.dfl))
    grads = ops.detect_loss_bwd(st, gscale)
    crit.epoch += 1
    return st.out5[0], st.out5[1:4], grads
