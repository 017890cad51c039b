"""mgdt_yolo_amd - MI355X-native (gfx950) detection hot path of the MGDT-YOLO Ultralytics-YOLOv8 fork.

Package layout mirrors the part of the reference package the path touches:
    nn.modules   the module registry (Conv, C2f, MSPA_C2f, SPPF, SimFusion_*, IFM, Injection..., Detect)
    nn.tasks     DetectionModel / parse_model / yaml_model_load
    yolo.utils   ops.non_max_suppression, tal.make_anchors, loss.v8DetectionLoss, torch_utils
    csrc/        HIP kernels + the C ABI (include/mgdt.h) -> libmgdt_hip.so
    models       the reference's model graphs as cfg dicts
"""
__version__ = '0.1.0'
