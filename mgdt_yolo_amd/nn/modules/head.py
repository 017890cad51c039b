"""Detect head (reference: nn/modules/head.py:133-186), compute on MI355X via libmgdt_hip.so.

Per level the box branch (3x3, 3x3, 1x1+bias -> 4*reg_max) and the class branch (3x3, 3x3, 1x1+bias -> nc) write
straight into the two channel ranges of one (B, no, H, W) NHWC map (the reference's torch.cat at head.py:160).
Eval: one decode kernel per level does DFL softmax-expectation + dist2bbox + stride scaling + sigmoid and
writes y (B, 4+nc, A) fp32.
"""
import math

import torch
import torch.nn as nn

from ... import ops
from .block import DFL
from .conv import Conv, HipModule

__all__ = ('Detect',)


class _HeadConv(HipModule):
    """The head's final `nn.Conv2d(c, out, 1)` with bias: parameter container + packed 1x1 MFMA conv."""

    @staticmethod
    def run(owner, conv, x, out):
        pk = owner._cached((id(conv), out.dtype), [conv.weight, conv.bias],
                           lambda: ops.PackedConv(conv.weight, conv.bias, None, 1, out.dtype))
        return ops.conv2d(x, pk, 1, ops.ACT_NONE, out=out)

    @staticmethod
    def backward(conv, x, g):
        """Plain conv + bias: fills conv.weight.grad / conv.bias.grad, returns dx."""
        ops.grad_buf(conv.weight)
        ops.grad_buf(conv.bias)
        ops.conv_wgrad(x, g, 1, 1, conv.weight.grad, dbias=conv.bias.grad)
        dx = ops.new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.dtype, x.device)
        return ops.conv_dgrad(g, conv.weight, 1, 1, dx)


class Detect(HipModule):
    """YOLOv8 Detect head for detection models (this fork: reg_max = 4, head.py:145)."""
    dynamic = False
    export = False
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 4
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], self.nc)
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    def forward(self, x):
        """Train: list of (B, no, H, W) raw maps.  Eval: (y (B, 4+nc, A), that list).  Mutates the input list in place
        like the reference (head.py:160)."""
        shape = x[0].shape
        r4 = 4 * self.reg_max
        for i in range(self.nl):
            xi = x[i]
            b, _, h, w = xi.shape
            feat = ops.new_act(b, self.no, h, w, self.cv2[i][0].out_dtype(xi), xi.device)
            pk01 = None if self.training else self._merged_first(i, xi, feat.dtype)
            if pk01 is not None:                              # both branches' first 3x3 conv read xi: one launch, cout = c2 + c3
                t01 = ops.conv2d(xi, pk01, 1, ops.ACT_SILU)
                c2 = self.cv2[i][0].conv.out_channels
                tb, tc = self.cv2[i][1](t01[:, :c2]), self.cv3[i][1](t01[:, c2:])
            else:
                tb = self.cv2[i][1](self.cv2[i][0](xi))      # Conv.forward: eval -> fused run, train -> batch-stat BN + ctx
                tc = self.cv3[i][1](self.cv3[i][0](xi))
            _HeadConv.run(self, self.cv2[i][2], tb, feat[:, :r4])
            _HeadConv.run(self, self.cv3[i][2], tc, feat[:, r4:])
            if self.training:
                self.__dict__.setdefault('_ctx', []).append((tb, tc))
            x[i] = feat
        if self.training:
            return x
        strides = self._cached('stride_list', [self.stride], lambda: [float(v) for v in self.stride.tolist()])  # host copy, no sync per call
        if self.dynamic or self.shape != shape:
            from ...yolo.utils.tal import make_anchors
            self.anchors, self.strides = (t.transpose(0, 1) for t in make_anchors(x, strides, 0.5))
            self.shape = shape
        a_total = sum(f.shape[2] * f.shape[3] for f in x)
        y = torch.empty(shape[0], 4 + self.nc, a_total, dtype=torch.float32, device=x[0].device)
        a_off = 0
        for i, f in enumerate(x):
            ops.detect_decode(f, self.reg_max, self.nc, strides[i], a_off, y)
            a_off += f.shape[2] * f.shape[3]
        return y if self.export else (y, x)

    def _merged_first(self, i, xi, dt):
        """PackedConv of cat(cv2[i][0], cv3[i][0]) along cout (BN folded per branch), or None when the pair is not two plain
        3x3 Conv+BN+SiLU layers the MFMA kernel takes."""
        a, b = self.cv2[i][0], self.cv3[i][0]
        ok = all(isinstance(m, Conv) and hasattr(m, 'bn') and isinstance(m.act, nn.SiLU) and m.conv.kernel_size == (3, 3) and m.conv.stride == (1, 1)
                 and m.conv.groups == 1 and m.conv.bias is None for m in (a, b))
        if not ok or a.bn.eps != b.bn.eps or a.conv.out_channels % 8 or not ops.conv_can_mfma(xi, a.conv.in_channels, a.conv.out_channels + b.conv.out_channels,
                                                                                            3, 1, 1, dt):
            return None
        tens = [t for m in (a, b) for t in (m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var)]
        cat = lambda f: torch.cat([f(a).detach().float(), f(b).detach().float()])
        return self._cached(('first01', i, dt), tens, lambda: ops.PackedConv(
            cat(lambda m: m.conv.weight), None,
            (cat(lambda m: m.bn.weight), cat(lambda m: m.bn.bias), cat(lambda m: m.bn.running_mean), cat(lambda m: m.bn.running_var), a.bn.eps), 3, dt))

    def backward(self, grads):
        """grads: list of d loss / d raw head maps (one per level, NHWC).  Returns the list of input gradients."""
        r4 = 4 * self.reg_max
        ctx = [self._ctx.pop() for _ in range(self.nl)][::-1]
        out = []
        for i, g in enumerate(grads):
            tb, tc = ctx[i]
            gb = self.cv2[i][0].backward(self.cv2[i][1].backward(_HeadConv.backward(self.cv2[i][2], tb, g[:, :r4])))
            gc = self.cv3[i][0].backward(self.cv3[i][1].backward(_HeadConv.backward(self.cv3[i][2], tc, g[:, r4:])))
            out.append(ops.add(gb, gc, out=gb))
        return out

    def bias_init(self):
        """Initialize Detect() biases (reference head.py:179-186); requires stride availability."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[:self.nc] = math.log(5 / self.nc / (640 / s) ** 2)
