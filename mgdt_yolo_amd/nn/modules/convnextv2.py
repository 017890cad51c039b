"""ConvNeXtV2_Block (reference nn/modules/convnextv2.py:48-77) as three fused HIP steps on NHWC:
   1. dw7x7 + bias + LayerNorm(eps 1e-6)                       (mgdt_dwconv7_ln_fwd)
   2. pwconv1 (Linear C->4C) + exact GELU as a 1x1 MFMA conv   (mgdt_conv2d_fwd)
   3. GRN statistics -> per-(image,channel) scale, folded with beta into pwconv2's input affine, + residual
      (mgdt_grn_stats_fwd, mgdt_conv2d_fwd with in_scale/in_shift/r1)
The reference permutes NCHW->NHWC->NCHW around the Linear layers; NHWC is the native layout here.
"""
import torch.nn as nn

from ... import ops
from .conv import HipModule
from .utils import GRN, LayerNorm

__all__ = ('ConvNeXtV2_Block',)


class ConvNeXtV2_Block(HipModule):
    def __init__(self, dim, drop_path=0.):
        super().__init__()
        if drop_path > 0.:
            raise RuntimeError('drop_path > 0 is never built on the detection path (IFM uses 0, block.py:337)')
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.grn = GRN(4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.drop_path = nn.Identity()

    def forward(self, x):
        dt = x.dtype
        dim = self.dwconv.in_channels
        dw = self._cached('dw', [self.dwconv.weight],
                          lambda: self.dwconv.weight.detach().float().reshape(dim, 49).t().contiguous())      # [49][C]
        pw1 = self._cached(('pw1', dt), [self.pwconv1.weight, self.pwconv1.bias],
                           lambda: ops.PackedConv(self.pwconv1.weight.detach().reshape(4 * dim, dim, 1, 1), self.pwconv1.bias, None, 1, dt))
        pw2 = self._cached(('pw2', dt), [self.pwconv2.weight, self.pwconv2.bias],
                           lambda: ops.PackedConv(self.pwconv2.weight.detach().reshape(dim, 4 * dim, 1, 1), self.pwconv2.bias, None, 1, dt))
        gb = self._cached('grn', [self.grn.gamma, self.grn.beta],
                          lambda: (self.grn.gamma.detach().float().reshape(-1).contiguous(), self.grn.beta.detach().float().reshape(-1).contiguous()))
        t = ops.dwconv7_ln(x, dw, self.dwconv.bias.detach().float(), self.norm.weight.detach().float(), self.norm.bias.detach().float(),
                           self.norm.eps)
        t = ops.conv2d(t, pw1, 1, ops.ACT_GELU)
        scale = ops.grn_scale(t, gb[0])
        return ops.conv2d(t, pw2, 1, ops.ACT_NONE, in_scale=scale, in_shift=gb[1], r1=x)
