"""v8DetectionLoss oracle: TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates yolo/utils/loss.py:108-208 (+ BboxLoss :56-89, preprocess :132-148, bbox_decode :150-157)
with the default gains box 7.5 / cls 0.5 / dfl 1.5 (yolo/cfg/default.yaml:89-91).
"""
import torch
import torch.nn.functional as F

from .boxes import ciou_xyxy, xywh2xyxy
from .layers import dist2bbox, make_anchors
from .tal import assign

GAINS = (7.5, 0.5, 1.5)


def dense_targets(batch_idx, cls, bboxes, batch_size, scale):
    """loss.py:132-148: (n,) (n,1) (n,4 xywh normalised) -> (B, Nmax, 5) [cls, xyxy px]."""
    if batch_idx.numel() == 0:
        return torch.zeros(batch_size, 0, 5)
    t = torch.cat((batch_idx.view(-1, 1), cls.view(-1, 1), bboxes), 1).float()
    counts = torch.bincount(t[:, 0].long(), minlength=batch_size)
    out = torch.zeros(batch_size, int(counts.max()), 5)
    for j in range(batch_size):
        rows = t[t[:, 0] == j]
        out[j, :rows.shape[0]] = rows[:, 1:]
    out[..., 1:5] = xywh2xyxy(out[..., 1:5] * scale)
    return out


def detection_loss(feats, batch, strides, reg_max, nc, call_count=0, gains=GAINS):
    """feats: list of (B, 4R+nc, H, W).  Returns (loss*B, items[3], aux dict)."""
    B = feats[0].shape[0]
    no = 4 * reg_max + nc
    pred = torch.cat([f.reshape(B, no, -1) for f in feats], 2)
    pred_distri, pred_scores = pred.split((4 * reg_max, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()
    imgsz = torch.tensor(feats[0].shape[2:], dtype=pred.dtype) * strides[0]
    anchor_points, stride_tensor = make_anchors([f.shape[2:] for f in feats], strides, 0.5, pred.dtype)

    targets = dense_targets(batch['batch_idx'], batch['cls'], batch['bboxes'], B, imgsz[[1, 0, 1, 0]])
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).to(pred.dtype)

    A = pred_distri.shape[1]
    proj = torch.arange(reg_max, dtype=pred.dtype)
    dist = pred_distri.view(B, A, 4, reg_max).softmax(3).matmul(proj)
    pred_bboxes = dist2bbox(dist, anchor_points, xywh=False)

    _, t_bboxes, t_scores, fg, gt_idx = assign(
        pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor), anchor_points * stride_tensor,
        gt_labels, gt_bboxes, mask_gt, call_count, nc)
    tss = max(t_scores.sum(), 1)

    loss = torch.zeros(3, dtype=pred.dtype)
    loss[1] = F.binary_cross_entropy_with_logits(pred_scores, t_scores.to(pred.dtype), reduction='none').sum() / tss
    if fg.sum():
        t_b = t_bboxes / stride_tensor
        weight = t_scores.sum(-1)[fg].unsqueeze(-1)
        iou = ciou_xyxy(pred_bboxes[fg], t_b[fg])
        loss[0] = ((1.0 - iou) * weight).sum() / tss
        # DFL, loss.py:56-89 with reg_max-1
        x1y1, x2y2 = t_b.chunk(2, -1)
        ltrb = torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp(0, reg_max - 1 - 0.01)
        pd = pred_distri[fg].view(-1, reg_max)
        tgt = ltrb[fg]
        tl = tgt.long()
        tr = tl + 1
        wl = tr - tgt
        wr = 1 - wl
        dfl = (F.cross_entropy(pd, tl.view(-1), reduction='none').view(tl.shape) * wl +
               F.cross_entropy(pd, tr.view(-1), reduction='none').view(tl.shape) * wr).mean(-1, keepdim=True)
        loss[2] = (dfl * weight).sum() / tss
    loss = loss * torch.tensor(gains, dtype=pred.dtype)
    aux = dict(fg_mask=fg, target_gt_idx=gt_idx, target_scores=t_scores, target_bboxes=t_bboxes)
    return loss.sum() * B, loss.detach(), aux
