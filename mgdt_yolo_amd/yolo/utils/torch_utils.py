"""Host-side helpers with the reference's names (reference: yolo/utils/torch_utils.py)."""
import math

import torch
import torch.nn as nn


def make_divisible(x, divisor):
    """Nearest x divisible by divisor, upwards (torch_utils.py:269-273)."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def initialize_weights(model):
    """BatchNorm eps 1e-3 / momentum 0.03, inplace activations (torch_utils.py:248-258)."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in [nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU]:
            m.inplace = True


def intersect_dicts(da, db, exclude=()):
    """Keys of da present in db with equal shapes (torch_utils.py:294-296)."""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}


@torch.no_grad()
def fuse_conv_and_bn(conv, bn):
    """Parameter rewrite W' = diag(g/sqrt(var+eps)) W, b' = beta - g*mu/sqrt(var+eps) (torch_utils.py:114-135).
    One-time host-side weight preparation for `model.fuse()`; the kernels fold BN at pack time either way."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups,
                      bias=True).requires_grad_(False).to(conv.weight.device)
    scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
    fused.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
    b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
    fused.bias.copy_(scale * b_conv + bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps)))
    return fused
