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
        """(h, w) of the source -> (out_h, out_w, new_unpad_h, new_unpad_w, top, left, ratio, (dw, dh)); augment.py:554-583."""
        new_shape = self.new_shape
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = np.mod(dw, self.stride), np.mod(dh, self.stride)
        elif self.scaleFill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        dw /= 2
        dh /= 2
        top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
        left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
        return new_unpad[1] + top + bottom, new_unpad[0] + left + right, new_unpad[1], new_unpad[0], top, left, ratio, (dw, dh)

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
