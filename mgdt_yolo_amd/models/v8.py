"""YOLOv8-family model graphs of the MGDT-YOLO fork as cfg dicts.

Same on-disk schema the reference parses (`[from, repeats, module, args]` rows, `scales`, `nc`;
reference: models/v8/{yolov8,mspa_c2f_yolov8,gd_yolov8,mspa_c2f_gd_yolov8}.yaml, consumed by
nn/tasks.py:604-699).  A user YAML in that schema is loaded by `nn.tasks.yaml_model_load`; these
builders exist so the GPU box (which has no copy of the reference) can build the named configs.
"""
from copy import deepcopy

# [depth, width, max_channels]
SCALES = {'n': [0.33, 0.25, 1024], 's': [0.33, 0.50, 1024], 'm': [0.67, 0.75, 768],
          'l': [1.00, 1.00, 512], 'x': [1.00, 1.25, 512]}


def _backbone(block):
    rows = [[-1, 1, 'Conv', [64, 3, 2]], [-1, 1, 'Conv', [128, 3, 2]]]
    for reps, ch in ((3, 128), (6, 256), (6, 512), (3, 1024)):
        rows.append([-1, reps, block, [ch, True]])
        if ch != 1024:
            rows.append([-1, 1, 'Conv', [ch * 2, 3, 2]])
    rows.append([-1, 1, 'SPPF', [1024, 5]])
    return rows


def _pan_head():
    up = [-1, 1, 'nn.Upsample', ['None', 2, 'nearest']]
    return [list(up), [[-1, 6], 1, 'Concat', [1]], [-1, 3, 'C2f', [512]],
            list(up), [[-1, 4], 1, 'Concat', [1]], [-1, 3, 'C2f', [256]],
            [-1, 1, 'Conv', [256, 3, 2]], [[-1, 12], 1, 'Concat', [1]], [-1, 3, 'C2f', [512]],
            [-1, 1, 'Conv', [512, 3, 2]], [[-1, 9], 1, 'Concat', [1]], [-1, 3, 'C2f', [1024]],
            [[15, 18, 21], 1, 'Detect', ['nc']]]


def _gd_head():
    return [[[2, 4, 6, 9], 1, 'SimFusion_4in', []], [-1, 1, 'IFM', [[64, 32]]],
            [6, 1, 'Conv', [256, 1, 1]], [[2, 4, -1], 1, 'SimFusion_3in', [256]],
            [[-1, 11], 1, 'InjectionMultiSum_Auto_pool', [256, [64, 32], 1]], [-1, 3, 'C2f', [256]],
            [[15], 1, 'Detect', ['nc']]]


def _gd_tood_head():
    """models/v8/mspa_c2f_gd_tood_yolov8.yaml: the GD neck with the task-aligned head, `hidc` = 64 unscaled (works at scale n only)."""
    return _gd_head()[:-1] + [[[15], 1, 'TOODHead', ['nc', 64]]]


def _gd_tood_head_s():
    """Our own variant for scale s (SURVEY section 8 row a15 / BASELINE configs[3]): the reference YAML's hidc=64 only fits scale n because
    parse_model leaves it unscaled (tasks.py:664-665 commented out); here hidc = 128 = the s-scale width of the head input."""
    return _gd_head()[:-1] + [[[15], 1, 'TOODHead', ['nc', 128]]]


def _cfg(block, head, nc):
    return {'nc': nc, 'scales': deepcopy(SCALES), 'backbone': _backbone(block), 'head': head()}


CONFIGS = {
    'yolov8': lambda nc=80: _cfg('C2f', _pan_head, nc),
    'mspa_c2f_yolov8': lambda nc=80: _cfg('MSPA_C2f', _pan_head, nc),
    'gd_yolov8': lambda nc=80: _cfg('C2f', _gd_head, nc),
    'mspa_c2f_gd_yolov8': lambda nc=80: _cfg('MSPA_C2f', _gd_head, nc),
    'mspa_c2f_gd_tood_yolov8_hidc128': lambda nc=80: _cfg('MSPA_C2f', _gd_tood_head_s, nc),      # use with scale='s'
    'mspa_c2f_gd_tood_yolov8': lambda nc=2: _cfg('MSPA_C2f', _gd_tood_head, nc),
}


def get_config(name, scale='n', nc=None):
    """cfg dict for `name` in CONFIGS at compound-scale letter `scale` (like 'yolov8n.yaml' file stems)."""
    d = CONFIGS[name]() if nc is None else CONFIGS[name](nc)      # nc=None: the YAML file's own class count
    d['scale'] = scale
    d['yaml_file'] = f'{name}.yaml'
    return d
