"""AutoBackend counterpart for the in-memory / checkpoint PyTorch branch (reference: nn/autobackend.py:92-121, :313-314, :415-427).

The reference class dispatches over a dozen runtimes (TorchScript, ONNX, TensorRT, ...); on this path there is one backend - the HIP kernels
behind DetectionModel - so only the `nn_module` / `pt` branch exists: move to the device, `fuse()`, pick the precision, expose
`stride / names / fp16 / pt`, `forward(im)` and `warmup()`.  `fp16=True` selects the reduced-precision path of this hardware, bfloat16
(`model.half()`); `fp16=False` the exact fp32 path (`model.float()`)."""
import torch
import torch.nn as nn


class AutoBackend(nn.Module):
    def __init__(self, weights, device=torch.device('cuda:0'), dnn=False, data=None, fp16=False, fuse=True, verbose=True):
        super().__init__()
        from .tasks import BaseModel, attempt_load_one_weight
        if dnn:
            raise RuntimeError('AutoBackend: only the PyTorch-module branch exists on the MI355X path (no OpenCV-DNN / ONNX / TensorRT runtimes)')
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('mgdt_yolo_amd runs on MI355X (HIP) only: AutoBackend needs a cuda device (no CPU fallback)')
        if isinstance(weights, BaseModel):
            model = weights.to(device)                          # autobackend.py:93
        elif isinstance(weights, (str, bytes)) or hasattr(weights, '__fspath__'):
            model, _ = attempt_load_one_weight(weights, device=device)      # a reference-format checkpoint, read without unpickling code
        else:
            raise RuntimeError(f'AutoBackend: cannot load {type(weights).__name__}')
        model = model.fuse(verbose=verbose) if fuse else model                # :94
        self.stride = max(int(model.stride.max()), 32)                         # :97
        self.names = model.names                                               # :98
        model.half() if fp16 else model.float()                                # :99
        self.model = model.eval()
        self.pt = self.nn_module = True
        self.fp16, self.device, self.nhwc, self.triton = bool(fp16), device, False, False

    def forward(self, im, augment=False, visualize=False):
        """(y, feats) of the detection model (autobackend.py:313-314).  uint8 images are accepted as they are (the stem divides by 255)."""
        if augment or visualize:
            raise RuntimeError('augment / visualize are host-side tooling outside the hot path')
        if self.fp16 and im.dtype == torch.float32:
            im = im.to(torch.bfloat16)                          # `im.half()` of the reference (:304-305), in this path's half type
        return self.model(im)

    def warmup(self, imgsz=(1, 3, 640, 640)):
        """One forward on a dummy input: packs every weight panel and lets the allocator settle (autobackend.py:415-427)."""
        im = torch.zeros(*imgsz, dtype=torch.bfloat16 if self.fp16 else torch.float32, device=self.device)
        with torch.no_grad():
            self.forward(im)
