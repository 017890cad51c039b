"""Block modules of the registry (reference: nn/modules/block.py), compute on MI355X via libmgdt_hip.so.

The reference builds these from chunk()/cat()/elementwise torch ops; here every block pre-allocates its
concatenated NHWC buffer once and lets the fused conv kernel read/write channel slices of it, so no
concat, split or add is ever materialised.
"""
import torch
import torch.nn as nn

from ... import ops
from .conv import Conv, HipModule, act_code
from .convnextv2 import ConvNeXtV2_Block
from .spr_module import SPRModule

__all__ = ('DFL', 'SPPF', 'C2f', 'MSPA_C2f', 'Bottleneck', 'SimFusion_4in', 'SimFusion_3in', 'IFM', 'h_sigmoid',
           'InjectionMultiSum_Auto_pool', 'Upsample')


def _plain_block(first, last, bottlenecks):
    """True when a CSP block is made of what mgdt_csp_block_fwd computes: 1x1 Conv+BN+act at both ends, bottlenecks of two dense 3x3 Conv+BN
    with the same activation and one shortcut setting."""
    def conv_ok(c, k):
        return (isinstance(c, Conv) and c.plain_affine() and c.conv.kernel_size == (k, k) and c.conv.stride == (1, 1) and c.conv.groups == 1
                and act_code(c.act) == a0)
    try:
        a0 = act_code(first.act)
    except RuntimeError:
        return False
    if a0 != ops.ACT_SILU:            # the fused block kernel is built for the blocks' default activation
        return False
    if not (conv_ok(first, 1) and conv_ok(last, 1)) or len(bottlenecks) == 0:
        return False
    return all(isinstance(m, Bottleneck) and conv_ok(m.cv1, 3) and conv_ok(m.cv2, 3) and m.add == bottlenecks[0].add
               and m.cv1.conv.in_channels == m.cv1.conv.out_channels == m.cv2.conv.out_channels for m in bottlenecks)


class DFL(nn.Module):
    """Integral module of Distribution Focal Loss (reference block.py:36-54): parameter container; the
    softmax-expectation is fused into the Detect decode kernel."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class Bottleneck(HipModule):
    """Standard bottleneck (reference block.py:514-526); the shortcut add is the conv epilogue's residual."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        k0 = k[0][0] if isinstance(k[0], (tuple, list)) else k[0]
        k1 = k[1][0] if isinstance(k[1], (tuple, list)) else k[1]
        self.cv1 = Conv(c1, c_, k0, 1)
        self.cv2 = Conv(c_, c2, k1, 1, g=g)
        self.add = shortcut and c1 == c2

    def run(self, x, out=None, x2=None):
        """y = [x (+x2)] + cv2(cv1(x (+x2))); `x2` is MSPA's pending `sp + spx[i]` add."""
        if self.training and hasattr(self.cv1, 'bn'):
            t = self.cv1.train_fwd(x, x2=x2)
            return self.cv2.train_fwd(t, out=out, r1=x if self.add else None, r2=x2 if self.add else None)
        t = self.cv1.run(x, x2=x2)
        return self.cv2.run(t, out=out, r1=x if self.add else None, r2=x2 if self.add else None)

    def backward(self, gz, acc_into=None):
        """Returns d/d(x [+ x2]) (the same tensor is the gradient of both addends); with `acc_into` the result is ADDED to that view instead
        (the shortcut's gradient and the accumulation ride in the data-gradient convolution's epilogue: no separate add launches)."""
        gmid = self.cv2.backward(gz)
        return self.cv1.backward(gmid, dx_out=acc_into, acc=acc_into is not None, r2=gz if self.add else None)

    def forward(self, x):
        return self.run(x)


class C2f(HipModule):
    """CSP bottleneck with 2 convolutions (reference block.py:187-207)."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x, head=None):
        """`head` (the model's plan, eval only): (injection module, its inputs) when this block's input is that injection's output and
        nobody else reads it - cv1 then runs inside the injection launch (mgdt_conv1x1_inject_conv_fwd) and `x` is None."""
        if head is not None:
            x_l = head[1][0]
            dt0 = self.cv1.out_dtype(x_l)
            ok = (not (self.training and hasattr(self.cv1, 'bn')) and _plain_block(self.cv1, self.cv2, self.m) and ops.csp_block_supported(
                ops.CSP_C2F, ops.new_act(x_l.shape[0], 2 * self.c, x_l.shape[2], x_l.shape[3], dt0, x_l.device), self.cv2.conv.out_channels, self.c, len(self.m), dt0))
            y01 = head[0].forward_into_conv(head[1], self.cv1) if ok else None
            if y01 is None:
                x = head[0](head[1])                         # not covered after all: the injection's own launch
            else:
                mids = [pk for m in self.m for pk in (m.cv1.packed(y01.dtype, False), m.cv2.packed(y01.dtype, False))]
                return ops.csp_block(ops.CSP_C2F, y01, None, None, mids, self.m[0].add, self.cv2.packed(y01.dtype, False), self.c, act_code(self.cv1.act),
                                     self.cv2.conv.out_channels, False)[0]
        b, _, h, w = x.shape
        c, n = self.c, len(self.m)
        train = self.training and hasattr(self.cv1, 'bn')
        dt = self.cv1.out_dtype(x)
        if not train and x.dtype == dt and _plain_block(self.cv1, self.cv2, self.m):
            y01 = ops.new_act(b, 2 * c, h, w, dt, x.device)
            if ops.csp_block_supported(ops.CSP_C2F, y01, self.cv2.conv.out_channels, c, n, dt):
                # two launches: cv1 (a wide streaming 1x1 GEMM), then the bottleneck chain on LDS-resident tiles + cv2 over the concat
                self.cv1.run(x, out=y01)
                mids = [pk for m in self.m for pk in (m.cv1.packed(dt, False), m.cv2.packed(dt, False))]
                return ops.csp_block(ops.CSP_C2F, y01, None, None, mids, self.m[0].add, self.cv2.packed(dt, False), c, act_code(self.cv1.act),
                                     self.cv2.conv.out_channels, False)[0]
        cat = ops.new_act(b, (2 + n) * c, h, w, dt, x.device)
        (self.cv1.train_fwd if train else self.cv1.run)(x, out=cat[:, :2 * c])
        for j, m in enumerate(self.m):
            m.run(cat[:, (1 + j) * c:(2 + j) * c], out=cat[:, (2 + j) * c:(3 + j) * c])
        return (self.cv2.train_fwd if train else self.cv2.run)(cat)

    forward_split = forward

    def backward(self, g):
        c, n = self.c, len(self.m)
        gcat = self.cv2.backward(g)                                   # grad of the concat buffer
        for j in reversed(range(n)):                                  # bottleneck j: slot 1+j -> slot 2+j
            self.m[j].backward(gcat[:, (2 + j) * c:(3 + j) * c], acc_into=gcat[:, (1 + j) * c:(2 + j) * c])
        return self.cv1.backward(gcat[:, :2 * c])


class MSPA_C2f(HipModule):
    """C2f with multi-scale pooling attention (reference block.py:209-287; scale=4, stride=1)."""

    def __init__(self, inplanes, outplanes, n=1, shortcut=False, g=1, e=0.5, scale=4, stride=1, stype='normal'):
        super().__init__()
        self.nums = scale
        self.inwidth = inplanes // self.nums
        self.outwidth = outplanes // self.nums
        self.stride = stride
        assert stype in ['stage', 'normal'], 'One of these is suppported (stage or normal)'
        self.stype = stype
        self.convs = nn.ModuleList([])
        self.btnk_nums = n
        for i in range(self.nums):
            if self.stride == 1 and i != self.nums - 1:
                self.convs.append(Conv(self.inwidth, self.inwidth, 1, 1))
            else:
                self.convs.append(Conv(inplanes + self.outwidth * (n - 1), outplanes, 1, 1))
        self.bottleneck = nn.ModuleList(Bottleneck(self.inwidth, self.inwidth, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))
        self.attention = SPRModule(self.outwidth)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x, deliver=None):
        """`deliver` (set by the model's neck plan, eval only): {'out': view to write the block output into (a slot of a consumer's concat
        buffer) or None, 'pools': views receiving adaptive_avg_pool2d(output)} - both served by the attention-scaling launch."""
        if self.stride != 1:
            raise RuntimeError('MSPA_C2f: only stride=1 is wired by parse_model and built here')
        d_out = deliver.get('out') if deliver else None
        d_pools = deliver.get('pools', ()) if deliver else ()
        b, _, h, w = x.shape
        wd, s, n = self.inwidth, self.nums, self.btnk_nums
        train = self.training and hasattr(self.convs[0], 'bn')
        run = (lambda m: m.train_fwd) if train else (lambda m: m.run)
        dt = self.convs[0].out_dtype(x)
        at = self.attention
        if (not train and s == 4 and self.inwidth == self.outwidth and self._chain_ok() and _plain_block(self.convs[0], self.convs[3], self.bottleneck)
                and ops.csp_block_supported(ops.CSP_MSPA, x, self.convs[3].conv.out_channels, wd, n, dt)):
            # one launch for the block (mgdt_csp_block_fwd) + one for attention MLP and scaling; the pooled sums come out of the block kernel
            mids = [pk for m in self.bottleneck for pk in (m.cv1.packed(dt, False), m.cv2.packed(dt, False))]
            out, pool, slots, tiles = ops.csp_block(ops.CSP_MSPA, x, self._packed_chain(dt).blob, None, mids, self.bottleneck[0].add,
                                                    self.convs[3].packed(dt, False), wd, act_code(self.convs[0].act), self.convs[3].conv.out_channels, True)
            return ops.spr_attention_scale(out, at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias, s, out=d_out, part=pool, nsplit=slots, tiles=tiles, pools=d_pools)
        cat = ops.new_act(b, (s - 1 + n) * wd, h, w, dt, x.device)
        # sp_i = convs[i](sp_{i-1} + spx[i]) written straight into its concat slot (block.py:250-259)
        if not train and s == 4 and ops.pw_chain_supported(wd, cat.dtype) and x.dtype == cat.dtype and self._chain_ok():
            ops.pw_chain3(x[:, :3 * wd], self._packed_chain(cat.dtype), act_code(self.convs[0].act), cat[:, :3 * wd])   # one launch
        else:
            run(self.convs[0])(x[:, :wd], out=cat[:, :wd])
            for i in range(1, s - 1):
                run(self.convs[i])(cat[:, (i - 1) * wd:i * wd], x2=x[:, i * wd:(i + 1) * wd], out=cat[:, i * wd:(i + 1) * wd])
        # last group: chained bottlenecks, each output kept (block.py:260-263)
        src, pending = cat[:, (s - 2) * wd:(s - 1) * wd], x[:, (s - 1) * wd:s * wd]
        for j, m in enumerate(self.bottleneck):
            dst = cat[:, (s - 1 + j) * wd:(s + j) * wd]
            m.run(src, out=dst, x2=pending)
            src, pending = dst, None
        out = run(self.convs[s - 1])(cat)
        if train:
            attn, part = ops.spr_attention_train(out, at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias, s)
            self._save_ctx((out, attn, part, x.shape))
        else:
            return ops.spr_attention_scale(out, at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias, s, out=d_out, pools=d_pools)   # pool, then MLP + scaling in one launch
        return ops.scale_channels(out, attn)

    def _chain_ok(self):
        """the three front convs are plain 1x1 Conv+BN with one activation (what mgdt_pw_chain3_fwd computes)"""
        cs = list(self.convs)[:3]
        return all(c.plain_affine() and c.conv.kernel_size == (1, 1) and c.conv.groups == 1 and c.conv.in_channels == self.inwidth
                   and c.conv.out_channels == self.inwidth and act_code(c.act) == act_code(cs[0].act) for c in cs)

    def _packed_chain(self, dtype):
        cs = list(self.convs)[:3]
        tens = [t for c in cs for t in c.affine_tensors()]
        return self._cached(('chain', dtype), tens, lambda: ops.PackedPwChain([(c.conv.weight, c.conv.bias, c.bn_tuple()) for c in cs], dtype))

    def backward(self, g):
        out, attn, part, xshape = self._ctx.pop()
        wd, s, n = self.inwidth, self.nums, self.btnk_nums
        at = self.attention
        dattn = ops.nc_reduce(g, out)                               # d/d attn[n,c] = sum_hw g*out
        prms = (at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias)
        gb = [ops.grad_buf(p) for p in prms]
        # the trainer's flat gradient buffer holds the four tensors back to back in this order: the kernel writes its [dW1|db1|dW2|db2] there
        chained = all(t.is_contiguous() and t.dtype == torch.float32 for t in gb) and all(
            a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.storage_offset() + a.numel() == b.storage_offset() for a, b in zip(gb, gb[1:]))
        pg_out = gb[0].new_empty(0).set_(gb[0].untyped_storage(), gb[0].storage_offset(), (sum(t.numel() for t in gb),)) if chained else None
        gout, pg = ops.spr_bwd(g, part, attn, dattn, at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias, s, out=pg_out)
        if not chained:
            o = 0
            for prm, t in zip(prms, gb):
                t.copy_(pg[o:o + t.numel()].view_as(prm))
                o += t.numel()
        gcat = self.convs[s - 1].backward(gout)
        gx = ops.new_act(xshape[0], xshape[1], xshape[2], xshape[3], g.dtype, g.device)
        sl = lambda t, i: t[:, i * wd:(i + 1) * wd]
        for j in reversed(range(n)):                                 # bottleneck j: slot s-2+j (+ pending for j == 0) -> slot s-1+j
            if j > 0:
                self.bottleneck[j].backward(sl(gcat, s - 1 + j), acc_into=sl(gcat, s - 2 + j))
                continue
            gin = self.bottleneck[j].backward(sl(gcat, s - 1 + j))
            dst = sl(gcat, s - 2 + j)
            ops.add(dst, gin, out=dst)
            ops.copy(gin, sl(gx, s - 1))                            # the pending addend spx[s-1]
        for i in reversed(range(1, s - 1)):                          # convs[i](slot i-1 + spx[i])
            gi = self.convs[i].backward(sl(gcat, i), dx_out=sl(gx, i))
            dst = sl(gcat, i - 1)
            ops.add(dst, gi, out=dst)
        self.convs[0].backward(sl(gcat, 0), dx_out=sl(gx, 0))
        return gx


class SPPF(HipModule):
    """Spatial Pyramid Pooling - Fast (reference block.py:138-153); 5/9/13 max windows in one pass."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise RuntimeError('SPPF: k=5 is what the YAMLs use and what the fused pooling kernel implements')
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)   # module-tree parity only

    def forward(self, x):
        b, _, h, w = x.shape
        c_ = self.cv1.conv.out_channels
        train = self.training and hasattr(self.cv1, 'bn')
        cat = ops.new_act(b, 4 * c_, h, w, self.cv1.out_dtype(x), x.device)
        (self.cv1.train_fwd if train else self.cv1.run)(x, out=cat[:, :c_])
        ops.sppf_pools(cat[:, :c_], cat[:, c_:2 * c_], cat[:, 2 * c_:3 * c_], cat[:, 3 * c_:])
        if train:
            self._cat = cat if ops.ctx_enabled() else None
        return (self.cv2.train_fwd if train else self.cv2.run)(cat)

    def backward(self, g):
        c_ = self.cv1.conv.out_channels
        cat, self._cat = self._cat, None
        gcat = self.cv2.backward(g)
        sl = lambda t, i: t[:, i * c_:(i + 1) * c_]
        # y3 = mp(y2), y2 = mp(y1), y1 = mp(x0): route the gradients back through the three max-pools
        g2 = ops.add(sl(gcat, 2), ops.maxpool5_bwd(sl(cat, 2), sl(gcat, 3)))
        g1 = ops.add(sl(gcat, 1), ops.maxpool5_bwd(sl(cat, 1), g2))
        g0 = ops.add(sl(gcat, 0), ops.maxpool5_bwd(sl(cat, 0), g1))
        return self.cv1.backward(g0)


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') of the stock YAML (models/v8/yolov8.yaml:30,34) on device."""

    def __init__(self, size=None, scale_factor=None, mode='nearest'):
        super().__init__()
        if mode != 'nearest' or size is not None:
            raise RuntimeError("Upsample: only (None, scale_factor, 'nearest') is on the detection path")
        self.scale_factor, self.mode, self.size = scale_factor, mode, size

    def forward(self, x):
        b, c, h, w = x.shape
        out = ops.new_act(b, c, int(h * self.scale_factor), int(w * self.scale_factor), x.dtype, x.device)
        return ops.nearest(x, out)

    def backward(self, g):
        b, c, h, w = g.shape
        return ops.nearest_bwd(g, ops.new_act(b, c, int(h / self.scale_factor), int(w / self.scale_factor), g.dtype, g.device))


class SimFusion_4in(nn.Module):
    """FAM (reference block.py:289-307): resample 4 levels to the 3rd one's size and concatenate."""

    def forward(self, x, pre=None):
        """`pre` (the model's neck plan, eval only): {'out': the concat buffer, 'done': slots their producers already wrote} - the MSPA blocks'
        attention-scaling launches leave their pooled copies / their own output in place, so those branches launch nothing here."""
        x_l, x_m, x_s, x_n = x
        b, c, h, w = x_s.shape
        cs = [x_l.shape[1], x_m.shape[1], c, x_n.shape[1]]
        self._shapes = [t.shape for t in x]
        out, done = None, ()
        if pre is not None and pre.get('out') is not None and tuple(pre['out'].shape) == (b, sum(cs), h, w) and pre['out'].dtype == x_s.dtype:
            out, done = pre['out'], pre['done']
        if out is None:
            out = ops.new_act(b, sum(cs), h, w, x_s.dtype, x_s.device)
        o = 0
        if 0 not in done:
            ops.adaptive_avgpool(x_l, out[:, o:o + cs[0]])
        o += cs[0]
        if 1 not in done:
            ops.adaptive_avgpool(x_m, out[:, o:o + cs[1]])
        o += cs[1]
        if not (2 in done and x_s.data_ptr() == out[:, o:o + cs[2]].data_ptr()):
            ops.copy(x_s, out[:, o:o + cs[2]])
        o += cs[2]
        ops.bilinear(x_n, out[:, o:o + cs[3]])
        return out

    def backward(self, g):
        sh = self._shapes
        mk = lambda i: ops.new_act(sh[i][0], sh[i][1], sh[i][2], sh[i][3], g.dtype, g.device)
        c0, c1, c2 = sh[0][1], sh[0][1] + sh[1][1], sh[0][1] + sh[1][1] + sh[2][1]
        return [ops.adaptive_avgpool_bwd(g[:, :c0], mk(0)), ops.adaptive_avgpool_bwd(g[:, c0:c1], mk(1)), g[:, c1:c2],
                ops.bilinear_bwd(g[:, c2:], mk(3))]


class SimFusion_3in(HipModule):
    """LAF (reference block.py:309-329)."""

    def __init__(self, in_channel_list, out_channels):
        super().__init__()
        self.cv1 = Conv(in_channel_list[0], out_channels, act=nn.ReLU()) if in_channel_list[0] != out_channels else nn.Identity()
        self.cv2 = Conv(in_channel_list[1], out_channels, act=nn.ReLU()) if in_channel_list[1] != out_channels else nn.Identity()
        self.cv3 = Conv(in_channel_list[2], out_channels, act=nn.ReLU()) if in_channel_list[2] != out_channels else nn.Identity()
        self.cv_fuse = Conv(out_channels * 3, out_channels, act=nn.ReLU())

    def forward(self, x, pre=None):
        """`pre` (the model's neck plan, eval only): {'out': the concat buffer, 'done': slots already in place, 'pooled0': avg-pooled x[0]}."""
        b, _, h, w = x[1].shape
        oc = self.cv_fuse.conv.out_channels
        dt, dev = x[1].dtype, x[1].device
        self._shapes = [t.shape for t in x]
        cat, done, pooled0 = None, (), None
        if pre is not None and pre.get('out') is not None and tuple(pre['out'].shape) == (b, 3 * oc, h, w) and pre['out'].dtype == dt:
            cat, done, pooled0 = pre['out'], pre['done'], pre.get('pooled0')
        if cat is None:
            cat = ops.new_act(b, 3 * oc, h, w, dt, dev)
        # branch 0: adaptive avg-pool then (optional) 1x1 ReLU conv
        if isinstance(self.cv1, nn.Identity):
            if 0 not in done:
                ops.adaptive_avgpool(x[0], cat[:, :oc])
        else:
            if pooled0 is None or tuple(pooled0.shape) != (b, x[0].shape[1], h, w):
                pooled0 = ops.adaptive_avgpool(x[0], ops.new_act(b, x[0].shape[1], h, w, dt, dev))
            (self.cv1.train_fwd if (self.training and hasattr(self.cv1, 'bn')) else self.cv1.run)(pooled0, out=cat[:, :oc])
        if isinstance(self.cv2, nn.Identity):
            if not (1 in done and x[1].data_ptr() == cat[:, oc:2 * oc].data_ptr()):
                ops.copy(x[1], cat[:, oc:2 * oc])
        else:
            (self.cv2.train_fwd if (self.training and hasattr(self.cv2, 'bn')) else self.cv2.run)(x[1], out=cat[:, oc:2 * oc])
        if isinstance(self.cv3, nn.Identity):
            ops.bilinear(x[2], cat[:, 2 * oc:])
        else:
            (self.cv3.train_fwd if (self.training and hasattr(self.cv3, 'bn')) else self.cv3.run)(
                ops.bilinear(x[2], ops.new_act(b, x[2].shape[1], h, w, dt, dev)), out=cat[:, 2 * oc:])
        return self.cv_fuse(cat)

    def backward(self, g):
        oc = self.cv_fuse.conv.out_channels
        sh = self._shapes
        mk = lambda i: ops.new_act(sh[i][0], sh[i][1], sh[i][2], sh[i][3], g.dtype, g.device)
        gcat = self.cv_fuse.backward(g)
        g0 = gcat[:, :oc] if isinstance(self.cv1, nn.Identity) else self.cv1.backward(gcat[:, :oc])
        g1 = gcat[:, oc:2 * oc] if isinstance(self.cv2, nn.Identity) else self.cv2.backward(gcat[:, oc:2 * oc])
        g2 = gcat[:, 2 * oc:] if isinstance(self.cv3, nn.Identity) else self.cv3.backward(gcat[:, 2 * oc:])
        return [ops.adaptive_avgpool_bwd(g0, mk(0)), g1, ops.bilinear_bwd(g2, mk(2))]


class IFM(nn.Module):
    """Information fusion module (reference block.py:331-342)."""

    def __init__(self, inc, ouc, embed_dim_p=96, fuse_block_num=3) -> None:
        super().__init__()
        self.conv = nn.Sequential(Conv(inc, embed_dim_p), *[ConvNeXtV2_Block(embed_dim_p) for _ in range(fuse_block_num)],
                                  Conv(embed_dim_p, sum(ouc)))

    def forward(self, x):
        mods = list(self.conv)
        if (not self.training and len(mods) >= 2 and isinstance(mods[-2], ConvNeXtV2_Block) and not mods[-2]._forward_hooks and not self.conv._forward_hooks):
            # the closing 1x1 Conv runs inside the last ConvNeXt block's launch when that block takes its one-launch path
            for m in mods[:-2]:
                x = m(x)
            r = mods[-2](x, tail=mods[-1])
            return r[0] if isinstance(r, tuple) else mods[-1](r)
        return self.conv(x)

    def backward(self, g):
        for m in reversed(list(self.conv)):
            g = m.backward(g)
        return g


class h_sigmoid(nn.Module):
    """relu6(x+3)/6 (reference block.py:344-350); fused into the Injection kernel."""

    def __init__(self, inplace=True):
        super().__init__()
        self.relu = nn.ReLU6(inplace=inplace)


class InjectionMultiSum_Auto_pool(HipModule):
    """Information injection (reference block.py:352-399)."""

    def __init__(self, inp: int, oup: int, global_inp: list, flag: int) -> None:
        super().__init__()
        self.global_inp = global_inp
        self.flag = flag
        self.local_embedding = Conv(inp, oup, 1, act=False)
        self.global_embedding = Conv(global_inp[self.flag], oup, 1, act=False)
        self.global_act = Conv(global_inp[self.flag], oup, 1, act=False)
        self.act = h_sigmoid()

    def _merged_global(self, g, xq=None):
        """PackedConv of cat(global_act, global_embedding) along cout, or None when they are not two plain 1x1 Conv+BN(no act) layers."""
        a, b = self.global_act, self.global_embedding
        ok = all(m.plain_affine() and act_code(m.act) == ops.ACT_NONE and m.conv.kernel_size == (1, 1) and m.conv.groups == 1 for m in (a, b))
        dt = a.out_dtype(g)
        fused = not hasattr(a, 'bn')
        if (not ok or fused != (not hasattr(b, 'bn')) or (not fused and a.bn.eps != b.bn.eps) or a.conv.out_channels % 8
                or not ops.conv_can_mfma(g, a.conv.in_channels, a.conv.out_channels + b.conv.out_channels, 1, 1, 1, dt)):
            return None
        tens = [t for m in (a, b) for t in m.affine_tensors()]
        cat = lambda fn: torch.cat([fn(a).detach().float(), fn(b).detach().float()])
        zb = lambda m: m.conv.bias if m.conv.bias is not None else torch.zeros(m.conv.out_channels, device=m.conv.weight.device)
        bn = lambda: None if fused else (cat(lambda m: m.bn.weight), cat(lambda m: m.bn.bias), cat(lambda m: m.bn.running_mean), cat(lambda m: m.bn.running_var), a.bn.eps)
        cb = lambda: cat(zb) if fused else None                # after fuse(): the folded biases ride on the merged conv
        if xq is not None:      # the e4m3 panel of the same merged convolution (quantize_fp8)
            return self._cached(('gaf', 'fp8', float(xq)), tens, lambda: ops.PackedConvFp8(cat(lambda m: m.conv.weight), cb(), bn(), 1, xq))
        return self._cached(('gaf', dt), tens, lambda: ops.PackedConv(cat(lambda m: m.conv.weight), cb(), bn(), 1, dt))

    def forward_into_conv(self, x, conv):
        """Injection + the 1x1 Conv+BN+act `conv` that consumes it (C2f.cv1) in one launch; returns conv's output, or None when the pair is
        not covered (the caller then runs the two modules as usual).  Eval, bf16, up-sampling branch only."""
        x_l, x_g = x
        le = self.local_embedding
        if self.training or not isinstance(conv, Conv) or not conv.plain_affine() or not le.plain_affine():
            return None
        if ops.Q8_CALIB is not None or self.__dict__.get('_q8') or conv.__dict__.get('_q8') or le.__dict__.get('_q8'):
            return None                                      # fp8 calibration / fp8 operands: the per-site path
        dt = le.out_dtype(x_l)
        c0 = sum(self.global_inp[:self.flag])
        g = x_g[:, c0:c0 + self.global_inp[self.flag]]
        pk2g = self._merged_global(g)
        if (pk2g is None or dt != torch.bfloat16 or act_code(le.act) != ops.ACT_NONE or le.conv.kernel_size != (1, 1)
                or le.conv.groups != 1 or conv.conv.kernel_size != (1, 1) or conv.conv.stride != (1, 1) or conv.conv.groups != 1
                or conv.conv.in_channels != le.conv.out_channels):
            return None
        oc = self.global_act.conv.out_channels
        b, _, h, w = x_l.shape
        gh, gw = g.shape[2], g.shape[3]
        probe = ops.new_act(1, oc, gh, gw, dt, x_l.device)
        if not ops.conv1x1_inject_conv_supported(x_l, oc, conv.conv.out_channels, probe, dt):
            return None
        in_launch = g.shape[1] == 32 and oc == 256 and ops.FUSED_INJECT_GCONV        # the two global convs run inside the launch too
        if not in_launch:
            gaf = ops.conv2d(g, pk2g, 1, ops.ACT_NONE)
            ga, gf = gaf[:, :oc], gaf[:, oc:]
            if ga.stride() != gf.stride():
                return None
        pk2 = conv._cached(('acc_order', dt), conv.affine_tensors(), lambda: ops.PackedConv(
            conv.conv.weight.detach()[:, ops.acc_order_index(conv.conv.in_channels, conv.conv.weight.device)], conv.conv.bias, conv.bn_tuple(), 1, dt))
        out = ops.new_act(b, conv.conv.out_channels, h, w, dt, x_l.device)
        if in_launch:
            return ops.conv1x1_inject_conv(x_l, le.packed(dt, direct=False), None, None, pk2, act_code(conv.act), out, gsrc=g, pkg=pk2g)
        return ops.conv1x1_inject_conv(x_l, le.packed(dt, direct=False), ga, gf, pk2, act_code(conv.act), out)

    def forward(self, x):
        x_l, x_g = x
        c0 = sum(self.global_inp[:self.flag])
        g = x_g[:, c0:c0 + self.global_inp[self.flag]]      # split(...)[flag] as a channel-slice view
        train = self.training and hasattr(self.local_embedding, 'bn')
        f = (lambda m: m.train_fwd) if train else (lambda m: m.run)
        pk2 = None if train else self._merged_global(g)
        if pk2 is not None:                                 # global_act and global_embedding read the same g: one launch, two channel ranges
            xq = self.q8_site(('gaf',), g) if pk2.dtype == torch.bfloat16 else None
            gaf = ops.conv2d(g, pk2, 1, ops.ACT_NONE) if xq is None else ops.conv2d_fp8(g, self._merged_global(g, xq), 1, ops.ACT_NONE)
            oc = self.global_act.conv.out_channels
            ga, gf = gaf[:, :oc], gaf[:, oc:]
        else:
            ga = f(self.global_act)(g)
            gf = f(self.global_embedding)(g)
        le = self.local_embedding
        dt = le.out_dtype(x_l)
        if (not train and le.plain_affine() and act_code(le.act) == ops.ACT_NONE and le.conv.kernel_size == (1, 1) and le.conv.groups == 1
                and ops.conv1x1_inject_supported(x_l, le.conv.out_channels, ga, dt) and ga.stride() == gf.stride()):
            return ops.conv1x1_inject(x_l, le.packed(dt, direct=False), ga, gf)   # local map never leaves the chip
        local = f(le)(x_l)
        if train:
            self._save_ctx((local, ga, x_g.shape, c0))
        return ops.inject(local, ga, gf)                    # pool vs up-sample branch chosen from the shapes (block.py:369)

    def backward(self, g):
        local, ga, gshape, c0 = self._ctx.pop()
        b, c, h, w = local.shape
        use_pool = h < ga.shape[2]
        small = lambda: ops.new_act(ga.shape[0], ga.shape[1], ga.shape[2], ga.shape[3], g.dtype, g.device)
        g_sig_up = ops.ew(g, local, ops.EW_MUL)                                     # d/d (resampled gate) = g * local
        if use_pool:    # out = local * avgpool(ga) + avgpool(gf)   (no h_sigmoid on this branch, block.py:385-390)
            sig_up = ops.adaptive_avgpool(ga, ops.new_act(b, c, h, w, g.dtype, g.device))
            g_ga = ops.adaptive_avgpool_bwd(g_sig_up, small())
            g_gf = ops.adaptive_avgpool_bwd(g, small())
        else:           # out = local * bilinear(h_sigmoid(ga)) + bilinear(gf)
            sig_up = ops.bilinear(ops.ew(ga, ga, ops.EW_HSIG), ops.new_act(b, c, h, w, g.dtype, g.device))
            g_hs = ops.bilinear_bwd(g_sig_up, small())
            g_ga = ops.ew(g_hs, ga, ops.EW_HSIG_GRAD, out=g_hs)
            g_gf = ops.bilinear_bwd(g, small())
        g_local = ops.ew(g, sig_up, ops.EW_MUL, out=sig_up)
        gx_l = self.local_embedding.backward(g_local)
        gx_g = torch.zeros(gshape, dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
        dst = gx_g[:, c0:c0 + self.global_inp[self.flag]]
        ggf = self.global_embedding.backward(g_gf)           # note: reverse order of the forward pushes is irrelevant (distinct modules)
        gga = self.global_act.backward(g_ga)
        ops.add(ggf, gga, out=dst)
        return [gx_l, gx_g]
