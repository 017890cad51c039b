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

__all__ = ('Detect', 'TOODHead', 'Conv_GN', 'TaskDecomposition', 'DyDCNv2', 'Scale')


class _HeadConv(HipModule):
    """The head's final `nn.Conv2d(c, out, 1)` with bias: parameter container + packed 1x1 MFMA conv."""

    @staticmethod
    def run(owner, conv, x, out):
        # class counts that are not a multiple of 4 (the fork's own data: nc = 1 or 2) give raw maps whose pixel rows are not 8-byte aligned:
        # those go through the direct kernel (element-wise stores), everything else through the MFMA kernel
        q = 4
        mfma = (conv.out_channels % q == 0 and out.stride(3) % q == 0 and out.stride(2) % q == 0 and out.stride(0) % q == 0 and
                out.data_ptr() % (q * out.element_size()) == 0 and
                ops.conv_can_mfma(x, conv.in_channels, conv.out_channels, 1, 1, 1, out.dtype))
        pk = owner._cached((id(conv), out.dtype, mfma), [conv.weight, conv.bias],
                           lambda: ops.PackedConv(conv.weight, conv.bias, None, 1, out.dtype, direct=not mfma))
        return ops.conv2d(x, pk, 1, ops.ACT_NONE, out=out)

    @staticmethod
    def backward(conv, x, g, accumulate=False):
        """Plain conv + bias: fills (or adds to: a head shared by several levels) conv.weight.grad / conv.bias.grad, returns dx."""
        ops.grad_buf(conv.weight)
        ops.grad_buf(conv.bias)
        ops.conv_wgrad(x, g, 1, 1, conv.weight.grad, dbias=conv.bias.grad, accumulate=accumulate)
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
        y, a_off, strides = None, 0, None
        if not self.training:
            strides = self._cached('stride_list', [self.stride], lambda: [float(v) for v in self.stride.tolist()])  # host copy, no sync per call
            a_total = sum(t.shape[2] * t.shape[3] for t in x)
            y = torch.empty(shape[0], 4 + self.nc, a_total, dtype=torch.float32, device=x[0].device)
            best = torch.empty(shape[0], a_total, dtype=torch.int64, device=x[0].device)      # NMS keys of the anchors' best classes (detect tail)
        decoded = [False] * self.nl
        for i in range(self.nl):
            xi = x[i]
            b, _, h, w = xi.shape
            feat = ops.new_act(b, self.no, h, w, self.cv2[i][0].out_dtype(xi), xi.device)
            pk01 = None if self.training else self._merged_first(i, xi, feat.dtype)
            if pk01 is not None:                              # both branches' first 3x3 conv read xi: one launch, cout = c2 + c3
                xq = self.q8_site(('first01', i), xi) if feat.dtype == torch.bfloat16 else None
                t01 = ops.conv2d(xi, pk01, 1, ops.ACT_SILU) if xq is None else ops.conv2d_fp8(xi, self._merged_first(i, xi, feat.dtype, xq), 1, ops.ACT_SILU)
                c2 = self.cv2[i][0].conv.out_channels
                tc = self.cv3[i][1](t01[:, c2:])
                box3 = self._box3_in_tail(i, t01[:, :c2], tc, feat.dtype)
                if box3 is not None:
                    # the box branch's second 3x3 conv (16 -> 16) runs inside the tail launch: its input is handed over instead of its output
                    pk3, pkb_pad = box3
                    pkc = self._cached((id(self.cv3[i][2]), feat.dtype), [self.cv3[i][2].weight, self.cv3[i][2].bias],
                                       lambda c=self.cv3[i][2]: ops.PackedConv(c.weight, c.bias, None, 1, feat.dtype))
                    ops.detect_tail(t01[:, :c2], tc, pkb_pad, pkc, self.nc, strides[i], a_off, feat, y, best, pk3=pk3)
                    decoded[i] = True
                    a_off += h * w
                    x[i] = feat
                    continue
                tb = self.cv2[i][1](t01[:, :c2])
            else:
                tb = self.cv2[i][1](self.cv2[i][0](xi))      # Conv.forward: eval -> fused run, train -> batch-stat BN + ctx
                tc = self.cv3[i][1](self.cv3[i][0](xi))
            if not self.training and ops.detect_tail_supported(tb, tc, self.nc, self.reg_max, feat.dtype):
                # both final 1x1 convs, the raw map and the decode in one launch (mgdt_detect_tail_fwd)
                pkb = self._cached((id(self.cv2[i][2]), feat.dtype), [self.cv2[i][2].weight, self.cv2[i][2].bias],
                                   lambda c=self.cv2[i][2]: ops.PackedConv(c.weight, c.bias, None, 1, feat.dtype))
                pkc = self._cached((id(self.cv3[i][2]), feat.dtype), [self.cv3[i][2].weight, self.cv3[i][2].bias],
                                   lambda c=self.cv3[i][2]: ops.PackedConv(c.weight, c.bias, None, 1, feat.dtype))
                ops.detect_tail(tb, tc, pkb, pkc, self.nc, strides[i], a_off, feat, y, best)
                decoded[i] = True
            else:
                _HeadConv.run(self, self.cv2[i][2], tb, feat[:, :r4])
                _HeadConv.run(self, self.cv3[i][2], tc, feat[:, r4:])
            if self.training:
                self._save_ctx((tb, tc))
            else:
                a_off += h * w
            x[i] = feat
        if self.training:
            return x
        if self.dynamic or self.shape != shape:
            from ...yolo.utils.tal import make_anchors
            self.anchors, self.strides = (t.transpose(0, 1) for t in make_anchors(x, strides, 0.5))
            self.shape = shape
        a_off = 0
        for i, f in enumerate(x):
            if not decoded[i]:
                ops.detect_decode(f, self.reg_max, self.nc, strides[i], a_off, y)
            a_off += f.shape[2] * f.shape[3]
        if all(decoded):
            ops.attach_best_keys(y, best)        # every level went through the tail kernel: non_max_suppression(y) skips its best-class scan
        return y if self.export else (y, x)

    def _box3_in_tail(self, i, tb_in, tc, dt):
        """(panel of cv2[i][1], zero-padded accumulator-order panel of cv2[i][2]) when the box branch's second conv can run inside the Detect tail
        launch: eval, bf16, 16 -> 16 channels, plain 3x3 Conv + BN/bias + SiLU, reg_max 4; else None."""
        m3, m1 = self.cv2[i][1], self.cv2[i][2]
        if (self.training or not ops.FUSED_DETECT_BOX3 or dt != torch.bfloat16 or not isinstance(m3, Conv) or not m3.plain_affine() or not isinstance(m3.act, nn.SiLU)
                or m3.conv.kernel_size != (3, 3) or m3.conv.stride != (1, 1) or m3.conv.groups != 1 or m3.conv.in_channels != 16 or m3.conv.out_channels != 16
                or m3._forward_hooks or m3.__dict__.get('_q8') or ops.Q8_CALIB is not None or tb_in.shape[1] != 16
                or not ops.detect_tail_supported(tb_in, tc, self.nc, self.reg_max, dt)):
            return None
        pk3 = m3.packed(dt, direct=False)

        def pad_pack():
            w = m1.weight.detach().float().reshape(m1.out_channels, 16)
            wp = torch.zeros(m1.out_channels, 32, device=w.device)
            wp[:, :16] = w                                     # channels 16..31 do not exist: zero weights
            return ops.PackedConv(wp[:, ops.acc_order_index(32, w.device)].reshape(m1.out_channels, 32, 1, 1), m1.bias, None, 1, dt)
        return pk3, self._cached((id(m1), dt, 'acc32'), [m1.weight, m1.bias], pad_pack)

    def _merged_first(self, i, xi, dt, xq=None):
        """PackedConv of cat(cv2[i][0], cv3[i][0]) along cout (BN folded per branch), or None when the pair is not two plain
        3x3 Conv+BN+SiLU layers the MFMA kernel takes."""
        a, b = self.cv2[i][0], self.cv3[i][0]
        ok = all(isinstance(m, Conv) and m.plain_affine() and isinstance(m.act, nn.SiLU) and m.conv.kernel_size == (3, 3) and m.conv.stride == (1, 1)
                 and m.conv.groups == 1 for m in (a, b))
        fused = ok and not hasattr(a, 'bn')
        if (not ok or fused != (not hasattr(b, 'bn')) or (not fused and a.bn.eps != b.bn.eps) or a.conv.out_channels % 8
                or not ops.conv_can_mfma(xi, a.conv.in_channels, a.conv.out_channels + b.conv.out_channels, 3, 1, 1, dt)):
            return None
        tens = [t for m in (a, b) for t in m.affine_tensors()]
        cat = lambda f: torch.cat([f(a).detach().float(), f(b).detach().float()])
        zb = lambda m: m.conv.bias if m.conv.bias is not None else torch.zeros(m.conv.out_channels, device=m.conv.weight.device)
        bn = lambda: None if fused else (cat(lambda m: m.bn.weight), cat(lambda m: m.bn.bias), cat(lambda m: m.bn.running_mean), cat(lambda m: m.bn.running_var), a.bn.eps)
        cb = lambda: cat(zb) if fused else None                # after fuse(): the folded biases ride on the merged conv
        if xq is not None:      # the e4m3 panel of the same merged convolution (quantize_fp8)
            return self._cached(('first01', i, 'fp8', float(xq)), tens, lambda: ops.PackedConvFp8(cat(lambda m: m.conv.weight), cb(), bn(), 3, xq))
        return self._cached(('first01', i, dt), tens, lambda: ops.PackedConv(cat(lambda m: m.conv.weight), cb(), bn(), 3, dt))

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


# ====================================================================================================================
# TOODHead (reference nn/modules/head.py:466-572).  Inference on the HIP kernels; the task-aligned head is not trained here yet.
# Parity is unpinned: the reference needs mmcv (ModulatedDeformConv2d, ConvModule, Scale), which is not shipped with it and not
# installed, so no fixture can be generated - the DCNv2 kernel follows mmcv's published modulated_deform_conv kernel and the whole
# head is checked against `oracle/tood.py` (a restatement, not a reference run).
# ====================================================================================================================
class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar (constructed by the reference head.py:490, unused by its forward)."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        raise RuntimeError('Scale is a parameter container only: TOODHead.forward never applies it (head.py:537)')


class Conv_GN(HipModule):
    """conv (no bias) + GroupNorm(16) + SiLU (reference head.py:67-81): MFMA conv, then the GroupNorm kernel with the activation."""
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if d != 1 or g != 1 or (p is not None and p != k // 2):
            raise RuntimeError('Conv_GN: only dense, dilation-1, same-padding convolutions are built')
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, groups=g, dilation=d, bias=False)
        self.gn = nn.GroupNorm(16, c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    def forward(self, x, out=None):
        from .conv import act_code
        dt = self.out_dtype(x)
        k, st = self.conv.kernel_size[0], self.conv.stride[0]
        mfma = ops.conv_can_mfma(x, self.conv.in_channels, self.conv.out_channels, k, st, 1, dt)
        pk = self._cached(('raw', dt, not mfma), [self.conv.weight], lambda: ops.PackedConv(self.conv.weight, None, None, k, dt, direct=not mfma))
        y = ops.conv2d(x, pk, st, ops.ACT_NONE)
        z = ops.groupnorm(y, self.gn.weight.detach().float(), self.gn.bias.detach().float(), self.gn.num_groups, self.gn.eps, act_code(self.act), out=out)
        if self.training:
            self._save_ctx((x, y))
        return z

    def backward(self, g, accumulate=False):
        """d loss / d x; parameter gradients written, or added when `accumulate` (the head is shared by the levels)."""
        from .conv import act_code
        x, y = self._ctx.pop()
        for p in (self.conv.weight, self.gn.weight, self.gn.bias):
            ops.grad_buf(p)
        dy = ops.gn_act_bwd(g, y, self.gn.weight.detach().float(), self.gn.bias.detach().float(), self.gn.num_groups, self.gn.eps, act_code(self.act),
                            self.gn.weight.grad, self.gn.bias.grad, accumulate)
        k = self.conv.kernel_size[0]
        ops.conv_wgrad(x, dy, k, 1, self.conv.weight.grad, accumulate=accumulate)
        return ops.conv_dgrad(dy, self.conv.weight, k, 1, ops.like(x))


class _ConvModuleBias(nn.Module):
    """mmcv ConvModule(norm_cfg=None): `conv` with bias + ReLU `activate` - parameter container with the reference's key names."""

    def __init__(self, c1, c2):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, 1, bias=True)
        self.activate = nn.ReLU(inplace=True)


class TaskDecomposition(HipModule):
    """Layer attention folded into the 1x1 reduction conv (reference head.py:83-131, norm_cfg=None as TOODHead builds it):
    the per-image weights w[b, k] scale the reduction conv's weight block k == they scale input channel k*feat + j, which is the
    conv kernel's per-(image, channel) input scale."""

    def __init__(self, feat_channels, stacked_convs, la_down_rate=8, conv_cfg=None, norm_cfg=None):
        super().__init__()
        if norm_cfg is not None:
            raise RuntimeError('TaskDecomposition: norm_cfg is None everywhere in the reference head')
        self.feat_channels, self.stacked_convs = feat_channels, stacked_convs
        self.in_channels = feat_channels * stacked_convs
        self.la_conv1 = nn.Conv2d(self.in_channels, self.in_channels // la_down_rate, 1)
        self.relu = nn.ReLU(inplace=True)
        self.la_conv2 = nn.Conv2d(self.in_channels // la_down_rate, stacked_convs, 1, padding=0)
        self.sigmoid = nn.Sigmoid()
        self.reduction_conv = _ConvModuleBias(self.in_channels, feat_channels)

    def forward(self, feat, sums=None):
        """`sums` = sum_hw feat per (image, channel) (ops.nc_reduce), shared by the two decompositions like the reference's avg_feat."""
        f32 = lambda t: t.detach().float().reshape(t.shape[0], -1).contiguous()
        la = self._cached('la', [self.la_conv1.weight, self.la_conv1.bias, self.la_conv2.weight, self.la_conv2.bias],
                          lambda: (f32(self.la_conv1.weight), self.la_conv1.bias.detach().float().contiguous(), f32(self.la_conv2.weight),
                                   self.la_conv2.bias.detach().float().contiguous()))
        scale = ops.tood_layer_attn(feat, la[0], la[1], la[2], la[3], self.stacked_convs, sums=sums)
        rc = self.reduction_conv.conv
        dt = self.out_dtype(feat)
        # as written in the reference the bmm path uses only `.conv.weight`: reduction_conv's bias parameter is never added (head.py:117-127)
        pk = self._cached(('red', dt), [rc.weight], lambda: ops.PackedConv(rc.weight, None, None, 1, dt))
        out = ops.conv2d(feat, pk, 1, ops.ACT_RELU, in_scale=scale)
        if self.training:
            self._save_ctx((feat, scale, sums if sums is not None else ops.nc_reduce(feat), out, la))
        return out

    def backward(self, g, accumulate=False):
        """-> (t, scale, dsums): d loss / d feat = t * scale[n, c] + dsums[n, c] (the GAP branch, already divided by H*W); the caller sums the
        two decompositions in one pass."""
        feat, scale, sums, out, la = self._ctx.pop()
        rc = self.reduction_conv.conv
        for p in (rc.weight, self.la_conv1.weight, self.la_conv1.bias, self.la_conv2.weight, self.la_conv2.bias):
            ops.grad_buf(p)
        gr = ops.relu_mask(g, out)
        ops.conv_wgrad(ops.channel_affine(feat, scale, None), gr, 1, 1, rc.weight.grad, accumulate=accumulate)      # d W = sum (feat * scale) x g
        t = ops.conv_dgrad(gr, rc.weight, 1, 1, ops.like(feat))
        dscale = ops.nc_reduce(t, feat)
        dsums = ops.tood_layer_attn_bwd(sums, dscale, feat.shape[2] * feat.shape[3], la[0], la[1], la[2], la[3], self.stacked_convs,
                                        self.la_conv1.weight.grad, self.la_conv1.bias.grad, self.la_conv2.weight.grad, self.la_conv2.bias.grad, accumulate)
        return t, scale, dsums


class _DCNParams(nn.Module):
    """mmcv ModulatedDeformConv2d(in, out, 3, padding=1, bias=False) as a parameter container (weight (out, in, 3, 3))."""

    def __init__(self, c1, c2, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c2, c1, 3, 3))
        self.bias = nn.Parameter(torch.zeros(c2)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class DyDCNv2(HipModule):
    """Modulated deformable conv 3x3 + GroupNorm(16) (reference block.py:401-432)."""

    def __init__(self, in_channels, out_channels, stride=1, norm_cfg=dict(type='GN', num_groups=16, requires_grad=True)):
        super().__init__()
        if stride != 1:
            raise RuntimeError('DyDCNv2: stride 1 only (the head never uses another)')
        self.with_norm = norm_cfg is not None
        self.conv = _DCNParams(in_channels, out_channels, bias=not self.with_norm)
        if self.with_norm:
            self.norm = nn.GroupNorm(norm_cfg.get('num_groups', 16), out_channels)

    def forward(self, x, offset_mask, act=ops.ACT_NONE):
        """offset_mask: (B, >=27, H, W) = 18 offsets + 9 mask LOGITS (the sigmoid of head.py:525 runs inside the kernel)."""
        cw = self.conv.weight
        if self.training:
            # training: the sampled-and-modulated columns are materialised once; the DCN is then the 1x1 conv of col with weight.view(cout, cin*9)
            if self.conv.bias is not None or not self.with_norm:
                raise NotImplementedError('DyDCNv2 training: only the normalised, bias-free form the head builds')
            dt = x.dtype
            col = ops.dcn_im2col(x, offset_mask)
            pk1 = self._cached(('col1x1', dt), [cw], lambda: ops.PackedConv(cw.detach().reshape(cw.shape[0], cw.shape[1] * 9, 1, 1), None, None, 1, dt))
            d = ops.conv2d(col, pk1, 1, ops.ACT_NONE)
            z = ops.groupnorm(d, self.norm.weight.detach().float(), self.norm.bias.detach().float(), self.norm.num_groups, self.norm.eps, act)
            self._save_ctx((x, offset_mask, col, d, act))
            return z
        wg = self._cached('gemm', [cw], lambda: cw.detach().float().permute(2, 3, 1, 0).reshape(9 * cw.shape[1], cw.shape[0]).contiguous())
        b = None if self.conv.bias is None else self.conv.bias.detach().float()
        if (b is None and x.dtype == torch.bfloat16 and offset_mask.dtype == x.dtype and cw.shape[1] % 8 == 0 and cw.shape[0] % 16 == 0 and cw.shape[0] <= 64
                and ops.is_nhwc(x) and ops.is_nhwc(offset_mask)):
            pk = self._cached('mfma', [cw], lambda: ops.PackedConv(cw, None, None, 3, torch.bfloat16))     # the ordinary fragment panel of a 3x3 conv
            y = ops.dcnv2_mfma(x, offset_mask, pk)
        else:
            y = ops.dcnv2(x, offset_mask, wg, b, cw.shape[0])
        if not self.with_norm:
            return y
        return ops.groupnorm(y, self.norm.weight.detach().float(), self.norm.bias.detach().float(), self.norm.num_groups, self.norm.eps, act, out=y)


    def backward(self, g, accumulate=False):
        """-> (d loss / d x, d loss / d offset_mask map)."""
        x, om, col, d, act = self._ctx.pop()
        cw = self.conv.weight
        for p in (cw, self.norm.weight, self.norm.bias):
            ops.grad_buf(p)
        gd = ops.gn_act_bwd(g, d, self.norm.weight.detach().float(), self.norm.bias.detach().float(), self.norm.num_groups, self.norm.eps, act,
                            self.norm.weight.grad, self.norm.bias.grad, accumulate)
        w1 = cw.detach().reshape(cw.shape[0], cw.shape[1] * 9, 1, 1)
        ops.conv_wgrad(col, gd, 1, 1, cw.grad.view(cw.shape[0], cw.shape[1] * 9, 1, 1), accumulate=accumulate)
        gcol = ops.conv_dgrad(gd, w1, 1, 1, ops.like(col))
        return ops.dcn_col2im_bwd(gcol, x, om)


class TOODHead(Detect):
    """Task-aligned dynamic detection head (reference head.py:466-572): shared Conv_GN stack, task decomposition, DCNv2-aligned box
    branch, probability-gated class branch; reg_max = 16.  Single shared head over all levels."""

    def __init__(self, nc, hidc, ch=()):
        HipModule.__init__(self)
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        if any(c != hidc for c in ch):
            raise RuntimeError(f'TOODHead: every input level must have hidc={hidc} channels, got {list(ch)} (the reference YAML passes hidc unscaled, '
                               'tasks.py:664-665)')
        self.share_conv = nn.Sequential(Conv_GN(hidc, hidc // 2, 3), Conv_GN(hidc // 2, hidc // 2, 3))
        self.cls_decomp = TaskDecomposition(hidc // 2, 2, 16)
        self.reg_decomp = TaskDecomposition(hidc // 2, 2, 16)
        self.DyDCNV2 = DyDCNv2(hidc // 2, hidc // 2)
        self.spatial_conv_offset = nn.Conv2d(hidc, 3 * 3 * 3, 3, padding=1)
        self.offset_dim = 2 * 3 * 3
        self.cls_prob_conv1 = nn.Conv2d(hidc, hidc // 4, 1)
        self.cls_prob_conv2 = nn.Conv2d(hidc // 4, 1, 3, padding=1)
        self.cv2 = nn.Conv2d(hidc // 2, 4 * self.reg_max, 1)
        self.cv3 = nn.Conv2d(hidc // 2, self.nc, 1)
        self.scale = nn.ModuleList(Scale(1.0) for _ in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    def _bias_conv(self, conv, x, act, dt, pad_to=None):
        """nn.Conv2d with bias on the conv kernels; `pad_to` appends zero output channels so that cout % 4 == 0 (MFMA path)."""
        k = conv.kernel_size[0]

        def build():
            w, b = conv.weight.detach().float(), conv.bias.detach().float()
            if pad_to and pad_to > w.shape[0]:
                w = torch.cat([w, w.new_zeros(pad_to - w.shape[0], *w.shape[1:])])
                b = torch.cat([b, b.new_zeros(pad_to - b.shape[0])])
            mf = ops.conv_can_mfma(x, w.shape[1], w.shape[0], k, 1, 1, dt)
            return ops.PackedConv(w, b, None, k, dt, direct=not mf)
        pk = self._cached((id(conv), dt, pad_to), [conv.weight, conv.bias], build)
        return ops.conv2d(x, pk, 1, act)

    def forward(self, x):
        shape = x[0].shape
        r4 = 4 * self.reg_max
        for i in range(self.nl):
            xi = x[i]
            b, _, h, w = xi.shape
            dt = self.share_conv[0].out_dtype(xi)
            half = self.share_conv[0].conv.out_channels
            feat = ops.new_act(b, 2 * half, h, w, dt, xi.device)        # torch.cat(stack_res_list) = two channel slots
            self.share_conv[0](xi, out=feat[:, :half])
            self.share_conv[1](feat[:, :half], out=feat[:, half:])
            sums = ops.nc_reduce(feat)                                  # both decompositions share avg_feat = GAP(feat) (head.py:516)
            cls_feat = self.cls_decomp(feat, sums)
            reg_feat = self.reg_decomp(feat, sums)
            om = self._bias_conv(self.spatial_conv_offset, feat, ops.ACT_NONE, dt, pad_to=28)     # 18 offsets | 9 mask logits | pad
            reg_feat = self.DyDCNV2(reg_feat, om, act=ops.ACT_RELU)      # F.relu(reg_feat) of head.py:537 folded into the GroupNorm pass
            p1 = self._bias_conv(self.cls_prob_conv1, feat, ops.ACT_RELU, dt)
            prob = self._bias_conv(self.cls_prob_conv2, p1, ops.ACT_NONE, dt)
            out = ops.new_act(b, self.no, h, w, dt, xi.device)
            gated = ops.pixel_gate(cls_feat, prob)
            _HeadConv.run(self, self.cv2, reg_feat, out[:, :r4])
            _HeadConv.run(self, self.cv3, gated, out[:, r4:])
            if self.training:
                self._save_ctx(dict(feat=feat, half=half, om=om, reg_feat=reg_feat, cls_feat=cls_feat, p1=p1, prob=prob, gated=gated))
            x[i] = out
        if self.training:
            return x
        strides = self._cached('stride_list', [self.stride], lambda: [float(v) for v in self.stride.tolist()])
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

    def _bias_conv_bwd(self, conv, x, g, pad_to=None, accumulate=False):
        """Backward of _bias_conv (without its activation): parameter gradients (+=), returns dx."""
        ops.grad_buf(conv.weight)
        ops.grad_buf(conv.bias)
        k = conv.kernel_size[0]
        w = conv.weight
        if pad_to and pad_to > w.shape[0]:          # g carries the zero padding channels of the forward
            wp = self._cached(('wpad', id(conv), pad_to), [w], lambda: torch.cat([w.detach().float(), w.new_zeros(pad_to - w.shape[0], *w.shape[1:])]).contiguous())
            dwp = torch.empty_like(wp)
            dbp = torch.empty(pad_to, dtype=torch.float32, device=w.device)
            ops.conv_wgrad(x, g, k, 1, dwp, dbias=dbp)
            (conv.weight.grad.add_ if accumulate else conv.weight.grad.copy_)(dwp[:w.shape[0]])
            (conv.bias.grad.add_ if accumulate else conv.bias.grad.copy_)(dbp[:w.shape[0]])
            return ops.conv_dgrad(g, wp, k, 1, ops.like(x))
        ops.conv_wgrad(x, g, k, 1, conv.weight.grad, dbias=conv.bias.grad, accumulate=accumulate)
        return ops.conv_dgrad(g, w, k, 1, ops.like(x))

    def backward(self, grads):
        """grads: d loss / d raw maps, one per level.  The head is shared: the levels are walked last to first (every sub-module's saved-state
        stack is LIFO) and parameter gradients are written by the first level processed, accumulated by the others.  Parameters the reference's
        forward never touches (`scale.*`, `reduction_conv.conv.bias`) get zero gradients."""
        r4 = 4 * self.reg_max
        out = [None] * self.nl
        for p in [m.scale for m in self.scale] + [self.cls_decomp.reduction_conv.conv.bias, self.reg_decomp.reduction_conv.conv.bias]:
            ops.grad_buf(p).zero_()
        for step, i in enumerate(reversed(range(self.nl))):
            acc = step > 0
            c = self._ctx.pop()
            g = grads[i]
            feat, half = c['feat'], c['half']
            hw = feat.shape[2] * feat.shape[3]
            # box branch: cv2 <- reg_feat = relu(GN(DCN(reg_feat0, om)))   (ReLU folded into the GroupNorm backward)
            g_reg = _HeadConv.backward(self.cv2, c['reg_feat'], g[:, :r4], accumulate=acc)
            g_reg0, g_om = self.DyDCNV2.backward(g_reg, accumulate=acc)
            # class branch: cv3 <- cls_feat * sigmoid(prob)
            g_gated = _HeadConv.backward(self.cv3, c['gated'], g[:, r4:], accumulate=acc)
            # cls_prob_conv2 has ONE output channel: its gradient map is kept as channel 0 of a zero 4-channel map so that the weight-gradient
            # kernel sees 4-aligned channels (pad_to=4 below drops the padding rows again)
            gp4 = ops.new_act(g.shape[0], 4, g.shape[2], g.shape[3], g.dtype, g.device).zero_()
            g_cls, _ = ops.pixel_gate_bwd(g_gated, c['cls_feat'], c['prob'], glogit=gp4[:, :1])
            g_p1 = ops.relu_mask(self._bias_conv_bwd(self.cls_prob_conv2, c['p1'], gp4, pad_to=4, accumulate=acc), c['p1'])
            gf = self._bias_conv_bwd(self.cls_prob_conv1, feat, g_p1, accumulate=acc)                       # d feat, accumulated below
            gf = ops.add(gf, self._bias_conv_bwd(self.spatial_conv_offset, feat, g_om, pad_to=28, accumulate=acc), out=gf)
            # the two task decompositions (processed in reverse order of the forward: reg, then cls)
            t_r, s_r, ds_r = self.reg_decomp.backward(g_reg0, accumulate=acc)
            t_c, s_c, ds_c = self.cls_decomp.backward(g_cls, accumulate=acc)
            gd = ops.nc_axpby(t_c, s_c, t_r, s_r, ops.ew_add_nc(ds_c, ds_r))                                 # t_c*s_c + t_r*s_r + (dsums_c + dsums_r)[n, c]
            gf = ops.add(gf, gd, out=gf)
            # feat = [f0 | f1]: f1 = share_conv[1](f0), f0 = share_conv[0](x)
            g_f0 = self.share_conv[1].backward(gf[:, half:], accumulate=acc)
            g_f0 = ops.add(g_f0, gf[:, :half], out=g_f0)
            out[i] = self.share_conv[0].backward(g_f0, accumulate=acc)
        return out

    def bias_init(self):
        """reference head.py:562-568 (single shared head: the class prior uses stride 16)."""
        self.cv2.bias.data[:] = 1.0
        self.cv3.bias.data[:self.nc] = math.log(5 / self.nc / (640 / 16) ** 2)
