"""Import harness for the read-only reference at /root/reference (build container only).

Used ONLY by tests/golden/gen_golden.py to produce the committed fixtures; nothing on the GPU
box imports this (the reference does not travel).  Recipe = SURVEY.md section 8(c):

  1. shell packages `ultralytics` / `ultralytics.yolo` whose __path__ points at the reference,
     so sub-modules resolve to the real files without running the package __init__s that pull in
     hub / SAM / RT-DETR / exporter;
  2. inert stand-ins for import-only third-party names that are absent in this image
     (cv2, torchvision, timm.models.layers, mmcv.*, mmengine.model).  None of them is arithmetic
     on the pinned path: DropPath(0) is never built, torchvision.ops.nms and mmcv DCNv2 are NOT
     emulated (their results stay "parity unpinned");
  3. YOLO_CONFIG_DIR -> scratch dir, no bytecode, no outbound sockets.
"""
import os
import socket
import sys
import tempfile
import types

REF = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if 'ultralytics' in sys.modules:
        return
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True
    os.environ['YOLO_CONFIG_DIR'] = tempfile.mkdtemp(prefix='yolo_cfg_')
    os.environ.setdefault('YOLO_VERBOSE', 'false')

    def _no_net(*a, **k):
        raise OSError('network disabled in golden harness')

    socket.create_connection = _no_net

    u = _mod('ultralytics', __version__='8.0.120')
    u.__path__ = [REF]
    y = _mod('ultralytics.yolo')
    y.__path__ = [os.path.join(REF, 'yolo')]

    # ---- import-only stand-ins (no arithmetic from these is ever compared) ----
    cv2 = _mod('cv2', __version__='0.0.0')
    cv2.setNumThreads = lambda n: None
    cv2.imread = cv2.imwrite = cv2.imshow = lambda *a, **k: None
    cv2.IMREAD_COLOR = 1

    tv = _mod('torchvision', __version__='0.15.0')
    tv.ops = _mod('torchvision.ops')

    def _nms_unavailable(*a, **k):
        raise RuntimeError('torchvision.ops.nms is not available in this image (parity unpinned)')

    tv.ops.nms = _nms_unavailable
    tv.transforms = _mod('torchvision.transforms')

    _mod('timm')
    _mod('timm.models')

    class DropPath(nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()

    _mod('timm.models.layers', DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_)

    def _unavailable(*a, **k):
        raise RuntimeError('mmcv/mmengine are not available in this image (parity unpinned)')

    _mod('mmcv')
    _mod('mmcv.cnn', build_activation_layer=_unavailable, build_norm_layer=_unavailable,
         ConvModule=_unavailable, Scale=_unavailable)
    _mod('mmcv.ops')
    _mod('mmcv.ops.modulated_deform_conv', ModulatedDeformConv2d=_unavailable)
    _mod('mmengine')
    _mod('mmengine.model', normal_init=_unavailable)


def load():
    """Return a namespace with the reference modules the fixtures are generated from."""
    install()
    import importlib
    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module('ultralytics.yolo.utils')
    ns.metrics = importlib.import_module('ultralytics.yolo.utils.metrics')
    ns.tal = importlib.import_module('ultralytics.yolo.utils.tal')
    ns.ops = importlib.import_module('ultralytics.yolo.utils.ops')
    ns.loss = importlib.import_module('ultralytics.yolo.utils.loss')
    ns.torch_utils = importlib.import_module('ultralytics.yolo.utils.torch_utils')
    ns.modules = importlib.import_module('ultralytics.nn.modules')
    ns.tasks = importlib.import_module('ultralytics.nn.tasks')
    return ns
