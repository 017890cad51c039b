"""Predictor counterpart (reference: yolo/engine/predictor.py:115-130, :210-260 and yolo/v8/detect/predict.py:10-29): the three steps
of `stream_inference` that touch the device - preprocess (LetterBox + BGR->RGB + HWC->CHW on the device, /255 folded into the stem),
inference (AutoBackend.forward) and postprocess (batched NMS + scale_boxes back to the original image) - behind the reference's method
names.  Sources, streams, result containers, plotting and saving are host tooling outside the path."""
import types

import numpy as np
import torch

from ...nn.autobackend import AutoBackend
from ..data.augment import LetterBox


class BasePredictor:
    def __init__(self, overrides=None):
        o = dict(conf=0.25, iou=0.7, imgsz=640, half=False, agnostic_nms=False, max_det=300, classes=None, device='cuda:0')   # yolo/cfg/default.yaml + model.py:241
        o.update(overrides or {})
        self.args = types.SimpleNamespace(**o)
        self.imgsz = self.args.imgsz if isinstance(self.args.imgsz, (tuple, list)) else (self.args.imgsz, self.args.imgsz)
        self.model = None
        self.device = torch.device(self.args.device)

    def setup_model(self, model, verbose=False):
        """predictor.py:295-308: AutoBackend(fuse=True), eval, precision from args.half."""
        self.model = AutoBackend(model, device=self.device, fp16=self.args.half, fuse=True, verbose=verbose)
        self.model.eval()
        return self.model

    def pre_transform(self, im):
        """List of uint8 (h, w, 3) BGR images -> list of letter-boxed uint8 (3, H, W) RGB device tensors (predictor.py:132-142)."""
        same_shapes = all(tuple(x.shape) == tuple(im[0].shape) for x in im)
        auto = same_shapes and self.model.pt
        lb = LetterBox(self.imgsz, auto=auto, stride=self.model.stride)
        dev = lambda x: x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        return [lb(image=dev(x)) for x in im]

    def preprocess(self, im):
        """(N, 3, h, w) tensor or list of (h, w, 3) BGR uint8 images -> the model's input batch (predictor.py:115-130).  The batch stays uint8:
        the first conv's loader divides by 255 exactly as `img /= 255` does, so there is no float image in HBM."""
        if not isinstance(im, torch.Tensor):
            planes = self.pre_transform(im)
            if any(p.shape != planes[0].shape for p in planes):
                raise RuntimeError('preprocess: images of one batch must letter-box to one shape (np.stack in the reference fails the same way)')
            im = torch.stack(planes)                     # one device-to-device gather of uint8 planes
        return im.to(self.device)

    def inference(self, im):
        with torch.no_grad():
            return self.model(im)

    def postprocess(self, preds, img, orig_imgs):
        return preds

    def __call__(self, source):
        """source: list of BGR uint8 images (or an already prepared (N, 3, h, w) tensor) -> postprocessed results."""
        if self.model is None:
            raise RuntimeError('call setup_model(model) first')
        im = self.preprocess(source)
        return self.postprocess(self.inference(im), im, source)


class DetectionPredictor(BasePredictor):
    """reference: yolo/v8/detect/predict.py:10-29.  Results are the per-image (n, 6) [x1, y1, x2, y2, conf, cls] tensors in ORIGINAL image
    coordinates (the reference wraps the same tensor in a `Results` container, a host-side convenience class)."""

    def postprocess(self, preds, img, orig_imgs):
        from ..utils import ops
        preds = ops.non_max_suppression(preds, self.args.conf, self.args.iou, agnostic=self.args.agnostic_nms, max_det=self.args.max_det,
                                        classes=self.args.classes)
        results = []
        for i, pred in enumerate(preds):
            orig_img = orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs
            if not isinstance(orig_imgs, torch.Tensor):
                ops.scale_boxes(img.shape[2:], pred, tuple(orig_img.shape))        # in place on pred[:, :4] (rows of 6 floats)
            results.append(pred)
        return results
