"""ConvNeXtV2_Block (reference nn/modules/convnextv2.py:48-77) on NHWC.  bf16 inference with dim in {32, 64, 96}: the whole block in ONE
launch (mgdt_cnx_block_fwd) when its tiles fit the chip, else dw7x7+LN, then the whole MLP with the 4C hidden map kept on chip
(mgdt_cnx_mlp_fwd).  Otherwise three fused HIP steps:
   1. dw7x7 + bias + LayerNorm(eps 1e-6)                       (mgdt_dwconv7_ln_fwd)
   2. pwconv1 (Linear C->4C) + exact GELU as a 1x1 MFMA conv   (mgdt_conv2d_fwd)
   3. GRN statistics -> per-(image,channel) scale, folded with beta into pwconv2's input affine, + residual
      (mgdt_grn_stats_fwd, mgdt_conv2d_fwd with in_scale/in_shift/r1)
The reference permutes NCHW->NHWC->NCHW around the Linear layers; NHWC is the native layout here.
"""
import torch
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

    @staticmethod
    def _tail_panel(conv, dim, dt):
        """PackedConv of a plain 1x1 Conv+BN+act over this block's `dim` channels with its input channels in accumulator order, or None"""
        from .conv import Conv, act_code
        if (not ops.FUSED_CNX_TAIL or not isinstance(conv, Conv) or not conv.plain_affine() or conv.conv.kernel_size != (1, 1) or conv.conv.stride != (1, 1)
                or conv.conv.groups != 1 or conv.conv.in_channels != dim or conv.conv.out_channels > dim or conv.conv.out_channels % 4
                or conv.__dict__.get('_q8') or ops.Q8_CALIB is not None or conv._forward_hooks):
            return None
        try:
            act_code(conv.act)
        except RuntimeError:
            return None
        return conv._cached(('acc_order', dt), conv.affine_tensors(), lambda: ops.PackedConv(
            conv.conv.weight.detach()[:, ops.acc_order_index(dim, conv.conv.weight.device)], conv.conv.bias, conv.bn_tuple(), 1, dt))

    def backward(self, g):
        x, u, t1, y1, t2, t3, S, pw1, pw2, dw, gb = self._ctx.pop()
        dim = self.dwconv.in_channels
        # pwconv2 (+ bias) on the GRN output t3, residual passes g straight through
        ops.grad_buf(self.pwconv2.weight)
        ops.grad_buf(self.pwconv2.bias)
        ops.conv_wgrad(t3, g, 1, 1, self.pwconv2.weight.grad, dbias=self.pwconv2.bias.grad)
        g3 = ops.conv_dgrad(g, self.pwconv2.weight, 1, 1, torch.empty_like(t3))
        # GRN
        ops.grad_buf(self.grn.gamma)
        ops.grad_buf(self.grn.beta)
        g2 = ops.grn_bwd(g3, t2, S, ops.nc_reduce(g3, t2), ops.nc_reduce(g3), gb[0], self.grn.gamma.grad, self.grn.beta.grad)
        # GELU + pwconv1 bias: plain (no-BN) mode of the BN backward kernel, beta = bias
        ops.grad_buf(self.pwconv1.bias)
        gy1 = ops.bn_act_bwd(g2, y1, None, None, None, self.pwconv1.bias, ops.ACT_GELU, None, self.pwconv1.bias.grad)
        ops.grad_buf(self.pwconv1.weight)
        ops.conv_wgrad(t1, gy1, 1, 1, self.pwconv1.weight.grad)
        gt1 = ops.conv_dgrad(gy1, self.pwconv1.weight, 1, 1, torch.empty_like(t1))
        # LayerNorm + depthwise 7x7
        ops.grad_buf(self.dwconv.weight)
        ops.grad_buf(self.dwconv.bias)
        ops.grad_buf(self.norm.weight)
        ops.grad_buf(self.norm.bias)
        gx = ops.dwconv7_ln_bwd(x, u, gt1, dw, self.norm.weight, self.norm.eps, self.dwconv.weight.grad, self.dwconv.bias.grad,
                                self.norm.weight.grad, self.norm.bias.grad)
        return ops.add(gx, g, out=gx)

    def _train_fwd(self, x, dw, pw1_raw, pw2, gb):
        dim = self.dwconv.in_channels
        t1, u = ops.dwconv7_ln_train(x, dw, self.dwconv.bias.detach().float(), self.norm.weight.detach().float(), self.norm.bias.detach().float(), self.norm.eps)
        y1 = ops.conv2d(t1, pw1_raw, 1, ops.ACT_NONE)                                   # raw Linear output WITHOUT bias
        t2 = ops.bn_act(y1, None, None, None, self.pwconv1.bias, ops.ACT_GELU)          # gelu(y1 + b1)
        S = ops.nc_reduce(t2, t2)                                                       # sum_hw t2^2 (GRN statistic)
        scale = ops.grn_scale(t2, gb[0])
        t3 = ops.channel_affine(t2, scale, gb[1])
        out = ops.conv2d(t3, pw2, 1, ops.ACT_NONE, r1=x)
        self._save_ctx((x, u, t1, y1, t2, t3, S, pw1_raw, pw2, dw, gb))
        return out

    def forward(self, x, tail=None):
        """`tail` (IFM, eval only): the 1x1 Conv+BN+act module that consumes this block's output and nothing else - it then runs inside the block's
        launch; the return value is (conv output, True) when that happened, else this block's output as usual."""
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
        if self.training:
            pw1_raw = self._cached(('pw1raw', dt), [self.pwconv1.weight],
                                   lambda: ops.PackedConv(self.pwconv1.weight.detach().reshape(4 * dim, dim, 1, 1), None, None, 1, dt))
            return self._train_fwd(x, dw, pw1_raw, pw2, gb)
        if ops.cnx_mlp_supported(dim, dt) and ops.cnx_block_supported(x, dt):
            # the whole block in one launch: the normalised map stays in LDS, the hidden 4C tile in registers (mgdt_cnx_block_fwd)
            mlp = self._cached(('mlp', dt), [self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias],
                               lambda: ops.PackedCnxMlp(self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias, dt))
            vec = self._cached('dwln', [self.dwconv.bias, self.norm.weight, self.norm.bias],
                               lambda: tuple(t.detach().float().contiguous() for t in (self.dwconv.bias, self.norm.weight, self.norm.bias)))
            if tail is not None:
                pk3 = self._tail_panel(tail, dim, dt)
                if pk3 is not None:
                    from .conv import act_code
                    return ops.cnx_block(x, dw, vec[0], vec[1], vec[2], self.norm.eps, mlp, gb[0], gb[1], tail=pk3, tail_act=act_code(tail.act)), True
            return ops.cnx_block(x, dw, vec[0], vec[1], vec[2], self.norm.eps, mlp, gb[0], gb[1])
        t = ops.dwconv7_ln(x, dw, self.dwconv.bias.detach().float(), self.norm.weight.detach().float(), self.norm.bias.detach().float(),
                           self.norm.eps)
        if ops.cnx_mlp_supported(dim, dt):     # hidden 4C map never leaves the chip
            mlp = self._cached(('mlp', dt), [self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias],
                               lambda: ops.PackedCnxMlp(self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias, dt))
            return ops.cnx_mlp(t, x, mlp, gb[0], gb[1])
        t = ops.conv2d(t, pw1, 1, ops.ACT_GELU)
        scale = ops.grn_scale(t, gb[0])
        return ops.conv2d(t, pw2, 1, ops.ACT_NONE, in_scale=scale, in_shift=gb[1], r1=x)
