"""Convolution modules of the registry (reference: nn/modules/conv.py), compute on MI355X via libmgdt_hip.so.

Same class names, constructor signatures and parameter names as the reference (`conv.weight`,
`bn.{weight,bias,running_mean,running_var}`), so YAML graphs and state_dicts transfer unchanged.
nn.Conv2d / nn.BatchNorm2d objects are used as PARAMETER CONTAINERS only; their forward is never called.
"""
import math

import torch
import torch.nn as nn

from ... import ops

__all__ = ('Conv', 'DWConv', 'Concat', 'autopad')


def autopad(k, p=None, d=1):
    """'same' padding (reference nn/modules/conv.py:16-22)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def act_code(m):
    if isinstance(m, nn.SiLU):
        return ops.ACT_SILU
    if isinstance(m, nn.ReLU):
        return ops.ACT_RELU
    if isinstance(m, nn.Identity):
        return ops.ACT_NONE
    if isinstance(m, nn.GELU):
        return ops.ACT_GELU
    raise RuntimeError(f'activation {type(m).__name__} has no fused HIP epilogue (SiLU, ReLU, GELU, Identity are built)')


class HipModule(nn.Module):
    """Base: compute dtype bookkeeping + packed-weight cache invalidated by parameter versions."""
    _cdtype = None   # set by DetectionModel.set_compute_dtype(); None -> follow the input

    def out_dtype(self, x):
        return self._cdtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)

    def _cached(self, key, tensors, build):
        cache = self.__dict__.setdefault('_pk', {})
        ver = (ops.PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in tensors if t is not None)
        hit = cache.get(key)
        if hit is not None and hit[0] != ver and hit[0][1:] == ver[1:] and ops.pack_is_current(hit[1]):
            hit = cache[key] = (ver, hit[1])             # the panel was refreshed in place by ops.repack_all() after the optimizer step
        if hit is None or hit[0] != ver:
            hit = (ver, build())
            cache[key] = hit
        return hit[1]

    def q8_site(self, key, x, x2=None):
        """fp8 bookkeeping of one implicit-GEMM convolution site of this module (`key` None: the module's own conv; merged panels name theirs).
        While BaseModel.quantize_fp8() calibrates: records the largest |input| the site sees.  Returns the site's activation multiplier, or None
        when the site runs in bf16."""
        if ops.Q8_CALIB is not None:
            xe = x if x2 is None else x.float() + x2.float()
            if ops.Q8_CALIB_PCT is None:
                amax = float(xe.abs().max())
            else:                                            # a high percentile of |input| instead of its maximum (outliers saturate at +-448)
                flat = xe.detach().abs().float().reshape(-1)
                if flat.numel() > (1 << 20):
                    flat = flat[::flat.numel() // (1 << 20)]
                k = min(flat.numel(), max(1, int(round(ops.Q8_CALIB_PCT / 100.0 * flat.numel()))))
                amax = float(flat.kthvalue(k).values)
            ops.Q8_CALIB[(self, key)] = max(ops.Q8_CALIB.get((self, key), 0.0), amax)
        q = self.__dict__.get('_q8')
        return None if q is None else q.get(key)

    @staticmethod
    def _no_train_bn(m):
        if m.training:
            raise NotImplementedError('this call path folds BatchNorm running statistics into the weights (eval); the training-mode path '
                                      '(batch statistics + backward) is Conv.train_fwd / Conv.backward - call the module in .train() mode '
                                      'through forward(), or .eval() it (raw head maps for the loss: model.model[-1].training = True)')

    def _save_ctx(self, item):
        """Keep what backward() needs.  Nothing is kept when no backward can follow (torch.no_grad() outside the model-level autograd
        Function), and DetectionModel._predict_once drops every stale context at the start of a train-mode forward, so a forward that is
        never followed by a backward cannot pin activations."""
        if ops.ctx_enabled():
            self.__dict__.setdefault('_ctx', []).append(item)


class Conv(HipModule):
    """Conv2d + BatchNorm2d + act (reference nn/modules/conv.py:25-42); BN folded at pack time (torch_utils.py:114-135)."""
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    # -- the conv's affine part, before and after BaseModel.fuse() -----------------------------------
    def plain_affine(self):
        """True for the two forms the block-level kernels take: bias-free conv + BatchNorm (as built), or - after `fuse()` (tasks.py:121-146:
        `bn` deleted, the folded bias on the conv) - conv + bias.  (the fused model AutoBackend / the predictor run took none of the
        block kernels.)"""
        return (hasattr(self, 'bn') and self.conv.bias is None) or not hasattr(self, 'bn')

    def bn_tuple(self):
        """(gamma, beta, mean, var, eps) for the pack routines, or None after fuse() (the conv then carries its bias)"""
        return (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.bn.eps) if hasattr(self, 'bn') else None

    def affine_tensors(self):
        """every tensor a packed panel of this conv depends on (cache keys)"""
        t = [self.conv.weight, self.conv.bias]
        if hasattr(self, 'bn'):
            t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return t

    def folded_bn_like(self):
        """the affine part as a BatchNorm tuple whatever the form: after fuse() the identity transform carrying the bias (gamma 1, mean 0,
        var 1, eps 0: exact)"""
        if hasattr(self, 'bn'):
            return self.bn_tuple()
        w = self.conv.weight
        one, zero = torch.ones(w.shape[0], device=w.device), torch.zeros(w.shape[0], device=w.device)
        return (one, self.conv.bias.detach().float() if self.conv.bias is not None else zero, zero, one, 0.0)

    # -- packed weights ------------------------------------------------------------------------------
    def _geometry(self):
        c = self.conv
        k, s = c.kernel_size[0], c.stride[0]
        if c.kernel_size[0] != c.kernel_size[1] or c.stride[0] != c.stride[1] or c.dilation != (1, 1) or c.padding != (k // 2, k // 2):
            raise RuntimeError(f'Conv geometry {c} is outside the built kernels (square k, same padding, dilation 1)')
        return k, s

    def packed(self, dtype, direct):
        has_bn = hasattr(self, 'bn')
        tens = [self.conv.weight, self.conv.bias] + ([self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var] if has_bn else [])
        k, _ = self._geometry()

        def build():
            bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.bn.eps) if has_bn else None
            return ops.PackedConv(self.conv.weight, self.conv.bias, bn, k, dtype, direct=direct, groups=self.conv.groups)
        return self._cached((dtype, direct), tens, build)

    def packed_fp8(self, xq):
        has_bn = hasattr(self, 'bn')
        tens = [self.conv.weight, self.conv.bias] + ([self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var] if has_bn else [])
        k, _ = self._geometry()

        def build():
            bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.bn.eps) if has_bn else None
            return ops.PackedConvFp8(self.conv.weight, self.conv.bias, bn, k, xq)
        return self._cached(('fp8', float(xq)), tens, build)

    # -- compute ----------------------------------------------------------------------------------------
    def run(self, x, out=None, x2=None, r1=None, r2=None, in_scale=None, in_shift=None):
        if hasattr(self, 'bn'):
            self._no_train_bn(self.bn)
        k, s = self._geometry()
        dt = self.out_dtype(x) if out is None else out.dtype
        mfma = ops.conv_can_mfma(x, self.conv.in_channels, self.conv.out_channels, k, s, self.conv.groups, dt)
        if mfma and dt == torch.bfloat16:
            xq = self.q8_site(None, x, x2)
            if xq is not None:                  # fp8 inference (BASELINE configs[4]): e4m3 operands, per-tensor activation / per-channel weight scales
                return ops.conv2d_fp8(x, self.packed_fp8(xq), s, act_code(self.act), out=out, x2=x2, r1=r1, r2=r2, in_scale=in_scale, in_shift=in_shift)
        pk = self.packed(dt, direct=not mfma)
        return ops.conv2d(x, pk, s, act_code(self.act), out=out, x2=x2, r1=r1, r2=r2, in_scale=in_scale, in_shift=in_shift)

    def forward(self, x):
        if self.training and hasattr(self, 'bn'):
            return self.train_fwd(x)
        return self.run(x)

    # -- training mode: batch-statistics BN, context for backward ---------------------------------------------
    def train_fwd(self, x, out=None, x2=None, r1=None, r2=None):
        """act(bn_batchstats(conv(x [+ x2]))) [+ r1] [+ r2]; keeps (x, x2, raw conv output, mean, rstd) for backward()."""
        k, s = self._geometry()
        dt = self.out_dtype(x) if out is None else out.dtype
        mfma = ops.conv_can_mfma(x, self.conv.in_channels, self.conv.out_channels, k, s, self.conv.groups, dt)
        pk = self._cached(('raw', dt, not mfma), [self.conv.weight],
                          lambda: ops.PackedConv(self.conv.weight, None, None, k, dt, direct=not mfma, groups=self.conv.groups))
        if x2 is not None and not mfma:
            raise RuntimeError('Conv.train_fwd: the fused pre-add needs the MFMA path (NHWC input, cin % 4 == 0)')
        y = ops.conv2d(x, pk, s, ops.ACT_NONE, x2=x2)
        bn = self.bn
        mean, rstd = ops.bn_stats(y, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        z = ops.bn_act(y, mean, rstd, bn.weight, bn.bias, act_code(self.act), out=out, r1=r1, r2=r2)
        self._save_ctx((x, x2, y, mean, rstd, k, s))
        return z

    def backward(self, gz, need_dx=True, dx_out=None, acc=False, r2=None):
        """gz: grad w.r.t. the pre-residual output.  Fills .grad of conv.weight / bn.weight / bn.bias (overwrite), returns dx
        (written into the view `dx_out` when given; `acc`: added to what dx_out holds; `r2`: one more addend, e.g. a shortcut's gradient -
        both fused into the data-gradient convolution's epilogue)."""
        x, x2, y, mean, rstd, k, s = self._ctx.pop()
        if self.conv.groups != 1:
            return self._backward_grouped(gz, x, y, mean, rstd, k, s, need_dx, dx_out, acc, r2)
        bn = self.bn
        ops.grad_buf(bn.weight)
        ops.grad_buf(bn.bias)
        dy = ops.bn_act_bwd(gz, y, mean, rstd, bn.weight, bn.bias, act_code(self.act), bn.weight.grad, bn.bias.grad)
        ops.grad_buf(self.conv.weight)
        ops.conv_wgrad(x, dy, k, s, self.conv.weight.grad, x2=x2)
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else ops.new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], dy.dtype, dy.device)
        return ops.conv_dgrad(dy, self.conv.weight, k, s, dx, accumulate=bool(acc and dx_out is not None), r2=r2)

    def _backward_grouped(self, gz, x, y, mean, rstd, k, s, need_dx, dx_out, acc, r2):
        """DWConv / Conv with g > 1 (conv.py:82-86): the grouped VALU kernels (weight [cout][cin/g][k][k]); not on any target YAML."""
        bn, g = self.bn, self.conv.groups
        dy = ops.bn_act_bwd(gz, y, mean, rstd, bn.weight, bn.bias, act_code(self.act), ops.grad_buf(bn.weight), ops.grad_buf(bn.bias))
        ops.gconv_wgrad(x, dy, k, s, g, ops.grad_buf(self.conv.weight))
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else ops.new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], dy.dtype, dy.device)
        ops.gconv_dgrad(dy, self.conv.weight, k, s, g, dx, accumulate=bool(acc and dx_out is not None))
        return dx if r2 is None else ops.add(dx, r2, out=dx)

    def forward_fuse(self, x):
        """After BaseModel.fuse(): `bn` is gone and `conv` carries the folded weight + bias (conv.py:40-42)."""
        return self.run(x)


class DWConv(Conv):
    """Depth-wise convolution (reference conv.py:82-86): registry entry, direct kernel."""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class Concat(nn.Module):
    """Concatenate along a dimension (reference conv.py:287-297); channel concat of NHWC maps on device."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def backward(self, g):
        """Adjoint of the channel concat: channel-slice views of g (no copies)."""
        out, c0 = [], 0
        for c in self._split:
            out.append(g[:, c0:c0 + c])
            c0 += c
        return out

    def forward(self, x):
        if self.d != 1:
            raise RuntimeError('Concat: only channel concatenation (dimension=1) is on the detection path')
        self._split = [t.shape[1] for t in x]
        b, _, h, w = x[0].shape
        out = ops.new_act(b, sum(t.shape[1] for t in x), h, w, x[0].dtype, x[0].device)
        c0 = 0
        for t in x:
            ops.copy(t, out[:, c0:c0 + t.shape[1]])
            c0 += t.shape[1]
        return out
