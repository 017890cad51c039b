"""LetterBox (reference: yolo/data/augment.py:538-593) for the predictor: geometry on the host exactly as the reference computes it, pixels
on the device - resize (cv2.INTER_LINEAR rule for 8-bit images), 114 border, BGR->RGB and HWC->CHW in one kernel per image, written
straight into the batch tensor (yolo/engine/predictor.py:115-130)."""
import numpy as np
import torch

from ... import _lib as L
from ... import ops as hip


class LetterBox:
    """Resize image and padding for detection (same constructor as the reference)."""

    def __init__(self, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, stride=32):
        self.new_shape = new_shape
        self.auto = auto
        self.scaleFill = scaleFill
        self.scaleup = scaleup
        self.stride = stride

    def geometry(self, shape):
        """Source (h, w) -> (out_h, out_w, resized_h, resized_w, top, left, ratio, (pad_w / 2, pad_h / 2)).  The rule is the reference's
        (augment.py:554-583) and its rounding is part of the contract (fixtures tests/golden/letterbox.npz): one scale for both axes unless
        scaleFill, resized side = round(side * scale), the leftover split in halves with the odd pixel going to the bottom / right edge
        (round(half - 0.1) before, round(half + 0.1) after)."""
        src_h, src_w = shape[0], shape[1]
        dst_h, dst_w = (self.new_shape, self.new_shape) if isinstance(self.new_shape, int) else self.new_shape[:2]
        scale = min(dst_h / src_h, dst_w / src_w)
        if not self.scaleup:
            scale = min(scale, 1.0)
        ratio = (scale, scale)
        res_w, res_h = int(round(src_w * scale)), int(round(src_h * scale))
        pad_w, pad_h = dst_w - res_w, dst_h - res_h
        if self.auto:                                  # minimal rectangle: pad only up to the next multiple of the stride
            pad_w, pad_h = np.mod(pad_w, self.stride), np.mod(pad_h, self.stride)
        elif self.scaleFill:                           # stretch to the target, no border
            pad_w, pad_h, res_w, res_h = 0.0, 0.0, dst_w, dst_h
            ratio = (dst_w / src_w, dst_h / src_h)
        half_w, half_h = pad_w / 2, pad_h / 2
        before = lambda half: int(round(half - 0.1))
        after = lambda half: int(round(half + 0.1))
        top, left = before(half_h), before(half_w)
        return res_h + top + after(half_h), res_w + left + after(half_w), res_h, res_w, top, left, ratio, (half_w, half_h)

    def __call__(self, labels=None, image=None, out=None):
        """image: uint8 (h, w, 3) BGR tensor on the device (or numpy array, copied once).  Returns the letter-boxed image as uint8 (3, H, W) RGB
        planes (written into `out` when given) - the reference's LetterBox followed by predictor.py:123-125."""
        if labels:
            raise RuntimeError('LetterBox: label transformation belongs to the training data pipeline (out of scope)')
        img = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(image)).to('cuda:0')
        hip._need_gpu(img)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3:
            raise RuntimeError('LetterBox: expected a uint8 (h, w, 3) image with packed pixels')
        oh, ow, nh, nw, top, left, _, _ = self.geometry(tuple(img.shape[:2]))
        if out is None:
            out = torch.empty(3, oh, ow, dtype=torch.uint8, device=img.device)
        elif tuple(out.shape) != (3, oh, ow) or not out.is_contiguous():
            raise RuntimeError(f'LetterBox: out must be a contiguous (3, {oh}, {ow}) uint8 tensor')
        L.check(L.lib().mgdt_letterbox_fwd(hip.ptr(img), img.shape[0], img.shape[1], img.stride(0), hip.ptr(out), oh, ow, nh, nw, top, left, hip.stream()), 'letterbox')
        return out
