"""Anchor helpers with the reference's names (reference: yolo/utils/tal.py:476-506).

These build small CONSTANT tensors (grid centres / strides), cached by their callers; the per-anchor arithmetic
that uses them (DFL decode, dist2bbox) lives in the HIP decode / loss kernels.
"""
import torch


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """(A, 2) cell centres [x, y] in grid units and the (A, 1) stride column for the head maps `feats` (each (B, C, H, W)); anchors run
    level by level, row by row - the order Detect concatenates its levels in (same contract as tal.py:476-488).  Built from the flat
    cell index (column = index mod W, row = index div W) instead of a mesh grid."""
    if not feats:
        raise ValueError('make_anchors: no feature maps')
    dev = feats[0].device
    centres, per_anchor_stride = [], []
    for fmap, s in zip(feats, strides):
        rows, cols = int(fmap.shape[2]), int(fmap.shape[3])
        cell = torch.arange(rows * cols, device=dev)
        xy = torch.stack((cell % cols, torch.div(cell, cols, rounding_mode='floor')), dim=1).to(torch.float32)
        centres.append(xy + grid_cell_offset)
        per_anchor_stride.append(torch.full((rows * cols, 1), float(s), dtype=torch.float32, device=dev))
    return torch.cat(centres, 0), torch.cat(per_anchor_stride, 0)
