"""Anchor helpers with the reference's names (reference: yolo/utils/tal.py:476-506).

These build small CONSTANT tensors (grid centres / strides), cached by their callers; the per-anchor arithmetic
that uses them (DFL decode, dist2bbox) lives in the HIP decode / loss kernels.
"""
import torch


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """Anchor points (A,2) and stride column (A,1), level-major, row-major (tal.py:476-488)."""
    anchor_points, stride_tensor = [], []
    assert feats is not None
    device = feats[0].device
    for i, stride in enumerate(strides):
        _, _, h, w = feats[i].shape
        sx = torch.arange(end=w, device=device, dtype=torch.float32) + grid_cell_offset
        sy = torch.arange(end=h, device=device, dtype=torch.float32) + grid_cell_offset
        sy, sx = torch.meshgrid(sy, sx, indexing='ij')
        anchor_points.append(torch.stack((sx, sy), -1).view(-1, 2))
        stride_tensor.append(torch.full((h * w, 1), float(stride), dtype=torch.float32, device=device))
    return torch.cat(anchor_points), torch.cat(stride_tensor)
