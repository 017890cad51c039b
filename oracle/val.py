"""TEST INFRASTRUCTURE ONLY - CPU restatement of the validator's prediction/label matching.

Follows yolo/v8/detect/val.py:152-175 (`DetectionValidator._process_batch`) and yolo/utils/metrics.py:52-72 (`box_iou`).
Pinned by tests/golden/val_match.npz, produced by the reference's own method (tests/golden/gen_golden.py::val_match).
"""
import numpy as np
import torch


def box_iou(box1, box2, eps=1e-7):
    """metrics.py:52-72: pairwise IoU of (N,4) and (M,4) xyxy boxes, eps in the union."""
    (a1, a2), (b1, b2) = box1.unsqueeze(1).chunk(2, 2), box2.unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def process_batch(detections, labels, iouv):
    """val.py:152-175.  detections (N,6) [x1,y1,x2,y2,conf,cls], labels (M,5) [cls,x1,y1,x2,y2] -> correct (N, len(iouv)) bool.

    Per IoU level: candidate (label, detection) pairs with IoU >= level and equal class; sorted by IoU descending; the first pair of
    every detection survives (its best label); of those - now ordered by detection index, which np.unique leaves behind - the first
    pair of every label survives, i.e. the LOWEST-INDEX detection that chose this label (the re-sort by IoU is commented out in the fork)."""
    n = detections.shape[0]
    correct = np.zeros((n, len(iouv)), bool)
    if n == 0 or labels.shape[0] == 0:
        return correct
    iou = box_iou(labels[:, 1:], detections[:, :4])
    correct_class = labels[:, 0:1] == detections[:, 5]
    for i in range(len(iouv)):
        x = torch.where((iou >= iouv[i]) & correct_class)
        if x[0].shape[0]:
            matches = torch.cat((torch.stack(x, 1), iou[x[0], x[1]][:, None]), 1).cpu().numpy()
            if x[0].shape[0] > 1:
                matches = matches[matches[:, 2].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return correct
