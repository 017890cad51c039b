"""Deterministic, name-keyed synthetic weights (there are no checkpoints offline).

Used by the golden-fixture generator (applied to the REFERENCE model's state_dict), by the parity
tests / smoke / bench (applied to OUR model's state_dict).  Values depend only on (seed, tensor name,
shape), so two state_dicts with the reference's key names get identical weights.
Statistics follow SURVEY.md section 8(d): non-trivial BN running stats, cls-head bias ~N(-2,1) so NMS has work.
"""
import re
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def seeded_tensor(name, shape, seed=0):
    r = _rng(seed, name)
    shape = tuple(shape)
    n = lambda mu, sd: (r.standard_normal(shape, dtype=np.float32) * np.float32(sd) + np.float32(mu))
    u = lambda a, b: r.uniform(a, b, shape).astype(np.float32)
    if name.endswith('num_batches_tracked') or name.endswith('dfl.conv.weight'):
        return None
    if re.search(r'\.(bn|norm|gn)\.weight$', name) or name.endswith('running_var'):
        v = u(0.75, 1.25)
    elif re.search(r'\.(bn|norm|gn)\.bias$', name) or name.endswith('running_mean'):
        v = n(0.0, 0.1)
    elif re.search(r'grn\.(gamma|beta)$', name):
        v = n(0.0, 0.2)
    elif re.search(r'cv3\.\d+\.2\.bias$', name) or re.search(r'\.\d+\.cv3\.bias$', name):      # Detect / TOODHead cls logits bias
        v = n(-3.0, 1.0)
    elif re.search(r'cv2\.\d+\.2\.bias$', name):      # Detect box-distribution bias
        v = n(1.0, 0.5)
    elif name.endswith('.bias'):
        v = n(0.0, 0.1)
    elif name.endswith('.weight') and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = 1.6                                      # keeps SiLU stacks at O(1) activations
        if re.search(r'cv[23]\.\d+\.2\.weight$', name):
            gain *= 0.05                                # Detect's final 1x1: logits O(1), no saturation
        elif re.search(r'\.convs\.3\.conv\.weight$', name):
            gain *= 2.0                                 # MSPA: compensates the 4-way softmax scale
        v = n(0.0, gain / np.sqrt(fan_in))
    else:
        v = n(0.0, 0.1)
    return torch.from_numpy(np.ascontiguousarray(v)).reshape(shape)


@torch.no_grad()
def seed_state_dict_(module, seed=0):
    """In-place fill of every parameter/buffer of `module` (reference-compatible names)."""
    for name, t in module.state_dict().items():
        v = seeded_tensor(name, t.shape, seed)
        if v is not None:
            t.copy_(v.to(t.dtype))
    return module


def seeded_images(b, h, w, seed=0, c=3):
    """(b,c,h,w) fp32 in [0,1) - numpy-generated so the bytes do not depend on the torch build."""
    r = np.random.default_rng([seed, 12345])
    return torch.from_numpy(r.random((b, c, h, w), dtype=np.float32))


def seeded_labels(b, nc, seed=1, max_boxes=20, min_boxes=1):
    """Synthetic label dict in the dataloader wire format (yolo/data/dataset.py:183-199)."""
    r = np.random.default_rng([seed, 54321])
    idx, cls, box = [], [], []
    for i in range(b):
        n = int(r.integers(min_boxes, max_boxes + 1))
        idx.append(np.full((n,), i, np.float32))
        cls.append(r.integers(0, nc, (n, 1)).astype(np.float32))
        cxy = r.uniform(0.2, 0.8, (n, 2))
        wh = r.uniform(0.05, 0.30, (n, 2))
        box.append(np.concatenate([cxy, wh], 1).astype(np.float32))
    return {'batch_idx': torch.from_numpy(np.concatenate(idx)), 'cls': torch.from_numpy(np.concatenate(cls)),
            'bboxes': torch.from_numpy(np.concatenate(box))}
