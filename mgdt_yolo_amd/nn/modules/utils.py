"""LayerNorm / GRN parameter containers (reference nn/modules/utils.py:145-182); compute is fused into
ConvNeXtV2_Block's kernels (dw7x7+LayerNorm, GRN statistics folded into pwconv2's input affine)."""
import torch
import torch.nn as nn

__all__ = ('LayerNorm', 'GRN')


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format='channels_last'):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ('channels_last', 'channels_first'):
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)


class GRN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, dim))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, dim))
