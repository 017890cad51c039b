"""Functional CPU restatement of the per-layer forward compute (eval mode) of the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Everything takes an explicit `sd` (state_dict with
the reference's parameter names) and a name prefix; there are no nn.Modules here.  All tensors are
NCHW fp32 (or fp64 when the caller passes fp64) on the CPU.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # yolo/utils/torch_utils.py:254-256 (initialize_weights overrides BatchNorm2d.eps)
BN_TRAIN = False  # tests flip this to restate Conv.forward in training mode (batch statistics, conv.py:36-38)
BN_MOMENTUM = 0.03  # yolo/utils/torch_utils.py:254-256
BN_RUNNING_OUT = None  # a dict here collects the running statistics a train-mode forward leaves behind (prefix -> (mean, var))


# ----------------------------------------------------------------------------- a1: Conv
def _act(x, act):
    """act in {'silu','relu','none'} - nn/modules/conv.py:27,35 (default SiLU, nn.ReLU(), Identity)."""
    if act == 'silu':
        return x * torch.sigmoid(x)
    if act == 'relu':
        return torch.relu(x)
    if act == 'none':
        return x
    raise ValueError(act)


def fold_bn(sd, p):
    """W', b' exactly as yolo/utils/torch_utils.py:114-135 (fuse_conv_and_bn), bias-free conv."""
    w = sd[p + '.conv.weight']
    g, b = sd[p + '.bn.weight'], sd[p + '.bn.bias']
    mu, var = sd[p + '.bn.running_mean'], sd[p + '.bn.running_var']
    scale = g / torch.sqrt(BN_EPS + var)
    w2 = (torch.diag(scale) @ w.reshape(w.shape[0], -1)).reshape(w.shape)
    b2 = b - g * mu / torch.sqrt(var + BN_EPS)
    return w2, b2


def conv(x, sd, p, s=1, act='silu', fused=False, g=1):
    """`Conv.forward` / `forward_fuse` - nn/modules/conv.py:25-42; pad = k//2 (autopad :16-22)."""
    w = sd[p + '.conv.weight']
    k = w.shape[-1]
    if fused:
        w2, b2 = fold_bn(sd, p)
        return _act(F.conv2d(x, w2.to(x.dtype), b2.to(x.dtype), s, k // 2, 1, g), act)
    y = F.conv2d(x, w.to(x.dtype), None, s, k // 2, 1, g)
    if BN_TRAIN:
        if BN_RUNNING_OUT is not None:
            # nn.BatchNorm2d in training mode: running <- (1 - momentum) * running + momentum * batch, with the UNBIASED batch variance
            with torch.no_grad():
                n = y.numel() // y.shape[1]
                mu, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=True) if n > 1 else y.var((0, 2, 3), unbiased=False)
                BN_RUNNING_OUT[p + '.bn'] = ((1 - BN_MOMENTUM) * sd[p + '.bn.running_mean'].to(y.dtype) + BN_MOMENTUM * mu,
                                             (1 - BN_MOMENTUM) * sd[p + '.bn.running_var'].to(y.dtype) + BN_MOMENTUM * var)
        y = F.batch_norm(y, None, None, sd[p + '.bn.weight'].to(x.dtype), sd[p + '.bn.bias'].to(x.dtype), True, 0.0, BN_EPS)
        return _act(y, act)
    y = F.batch_norm(y, sd[p + '.bn.running_mean'].to(x.dtype), sd[p + '.bn.running_var'].to(x.dtype),
                     sd[p + '.bn.weight'].to(x.dtype), sd[p + '.bn.bias'].to(x.dtype), False, 0.0, BN_EPS)
    return _act(y, act)


# ----------------------------------------------------------------------------- a3/a4: Bottleneck, C2f
def bottleneck(x, sd, p, shortcut, fused=False):
    """nn/modules/block.py:514-526 with k=((3,3),(3,3)), e=1.0 (c1==c2 always on this path)."""
    y = conv(conv(x, sd, p + '.cv1', fused=fused), sd, p + '.cv2', fused=fused)
    return x + y if shortcut else y


def c2f(x, sd, p, n, shortcut, fused=False):
    """nn/modules/block.py:187-207."""
    y = list(conv(x, sd, p + '.cv1', fused=fused).chunk(2, 1))
    for j in range(n):
        y.append(bottleneck(y[-1], sd, f'{p}.m.{j}', shortcut, fused))
    return conv(torch.cat(y, 1), sd, p + '.cv2', fused=fused)


# ----------------------------------------------------------------------------- a5/a6: SPR, MSPA_C2f
def spr(x, sd, p):
    """nn/modules/spr_module.py:8-31. Flatten order of the 2x2 pool is (c,2,2)."""
    b = x.shape[0]
    o1 = F.adaptive_avg_pool2d(x, 1).reshape(b, -1, 1, 1)
    o2 = F.adaptive_avg_pool2d(x, 2).reshape(b, -1, 1, 1)
    o = torch.cat((o1, o2), 1)
    o = torch.relu(F.conv2d(o, sd[p + '.fc1.weight'].to(x.dtype), sd[p + '.fc1.bias'].to(x.dtype)))
    o = F.conv2d(o, sd[p + '.fc2.weight'].to(x.dtype), sd[p + '.fc2.bias'].to(x.dtype))
    return torch.sigmoid(o)


def mspa_c2f(x, sd, p, n, shortcut, fused=False, scale=4):
    """nn/modules/block.py:209-287 (stride=1, stype='normal')."""
    b, c = x.shape[:2]
    w = c // scale
    spx = x.chunk(scale, 1)
    outs = []
    sp = spx[0]
    for i in range(scale - 1):
        if i > 0:
            sp = sp + spx[i]
        sp = conv(sp, sd, f'{p}.convs.{i}', fused=fused)
        outs.append(sp)
    sp = sp + spx[scale - 1]
    for j in range(n):
        sp = bottleneck(sp, sd, f'{p}.bottleneck.{j}', shortcut, fused)
        outs.append(sp)
    out = conv(torch.cat(outs, 1), sd, f'{p}.convs.{scale - 1}', fused=fused)
    attn = torch.cat([spr(t, sd, p + '.attention') for t in out.chunk(scale, 1)], 1)  # (b, 4w,1,1)
    attn = torch.softmax(attn.view(b, scale, w, 1, 1), 1)                              # over the groups
    return (out.view(b, scale, w, *out.shape[2:]) * attn).reshape(out.shape)


# ----------------------------------------------------------------------------- a7: SPPF
def sppf(x, sd, p, fused=False, k=5):
    """nn/modules/block.py:138-153."""
    x = conv(x, sd, p + '.cv1', fused=fused)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv(torch.cat((x, y1, y2, y3), 1), sd, p + '.cv2', fused=fused)


# ----------------------------------------------------------------------------- a8/a10: SimFusion
def bilinear(x, size):
    """F.interpolate(mode='bilinear', align_corners=False) - block.py:304,328,393-394."""
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


def simfusion_4in(xs):
    """nn/modules/block.py:289-307."""
    x_l, x_m, x_s, x_n = xs
    hw = tuple(x_s.shape[2:])
    return torch.cat([F.adaptive_avg_pool2d(x_l, hw), F.adaptive_avg_pool2d(x_m, hw), x_s, bilinear(x_n, hw)], 1)


def simfusion_3in(xs, sd, p, fused=False):
    """nn/modules/block.py:309-329; cvN is Identity when channel counts already match."""
    hw = tuple(xs[1].shape[2:])
    t = [F.adaptive_avg_pool2d(xs[0], hw), xs[1], bilinear(xs[2], hw)]
    for i in range(3):
        if f'{p}.cv{i + 1}.conv.weight' in sd:
            t[i] = conv(t[i], sd, f'{p}.cv{i + 1}', act='relu', fused=fused)
    return conv(torch.cat(t, 1), sd, p + '.cv_fuse', act='relu', fused=fused)


# ----------------------------------------------------------------------------- a9: IFM / ConvNeXtV2
def convnext_block(x, sd, p):
    """nn/modules/convnextv2.py:48-77; LayerNorm/GRN nn/modules/utils.py:145-182."""
    c = x.shape[1]
    t = F.conv2d(x, sd[p + '.dwconv.weight'].to(x.dtype), sd[p + '.dwconv.bias'].to(x.dtype), 1, 3, 1, c)
    t = t.permute(0, 2, 3, 1)
    t = F.layer_norm(t, (c,), sd[p + '.norm.weight'].to(x.dtype), sd[p + '.norm.bias'].to(x.dtype), 1e-6)
    t = F.linear(t, sd[p + '.pwconv1.weight'].to(x.dtype), sd[p + '.pwconv1.bias'].to(x.dtype))
    t = F.gelu(t)  # exact erf form (nn.GELU default)
    gx = torch.norm(t, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    t = sd[p + '.grn.gamma'].to(x.dtype) * (t * nx) + sd[p + '.grn.beta'].to(x.dtype) + t
    t = F.linear(t, sd[p + '.pwconv2.weight'].to(x.dtype), sd[p + '.pwconv2.bias'].to(x.dtype))
    return x + t.permute(0, 3, 1, 2)


def ifm(x, sd, p, nblocks=3, fused=False):
    """nn/modules/block.py:331-342."""
    x = conv(x, sd, p + '.conv.0', fused=fused)
    for i in range(nblocks):
        x = convnext_block(x, sd, f'{p}.conv.{i + 1}')
    return conv(x, sd, f'{p}.conv.{nblocks + 1}', fused=fused)


# ----------------------------------------------------------------------------- a11: Injection
def inject(xs, sd, p, global_inp, flag, fused=False):
    """nn/modules/block.py:352-399 (h_sigmoid :344-350).  The pooled branch skips h_sigmoid as written."""
    x_l, x_g = xs
    hw = tuple(x_l.shape[2:])
    use_pool = x_l.shape[2] < x_g.shape[2]
    g = x_g.split(list(global_inp), 1)[flag]
    local = conv(x_l, sd, p + '.local_embedding', act='none', fused=fused)
    ga = conv(g, sd, p + '.global_act', act='none', fused=fused)
    gf = conv(g, sd, p + '.global_embedding', act='none', fused=fused)
    if use_pool:
        sig, gf = F.adaptive_avg_pool2d(ga, hw), F.adaptive_avg_pool2d(gf, hw)
    else:
        sig, gf = bilinear(F.relu6(ga + 3) / 6, hw), bilinear(gf, hw)
    return local * sig + gf


# ----------------------------------------------------------------------------- a12-a14: DFL, anchors, Detect
def make_anchors(shapes, strides, offset=0.5, dtype=torch.float32):
    """yolo/utils/tal.py:476-488.  shapes = [(h,w),...] per level."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=dtype) + offset
        sy = torch.arange(h, dtype=dtype) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing='ij')
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=dtype))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """yolo/utils/tal.py:491-500."""
    lt, rb = distance.chunk(2, dim)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def dfl(box, reg_max):
    """nn/modules/block.py:36-54: softmax over the R bins, expectation with weights arange(R)."""
    b, _, a = box.shape
    pr = box.view(b, 4, reg_max, a).softmax(2)
    return (pr * torch.arange(reg_max, dtype=box.dtype).view(1, 1, reg_max, 1)).sum(2)


def detect_raw(xs, sd, p, fused=False):
    """Per-level raw head maps (B, 4R+nc, H, W) - nn/modules/head.py:155-164 (training return)."""
    out = []
    for i, x in enumerate(xs):
        t = conv(conv(x, sd, f'{p}.cv2.{i}.0', fused=fused), sd, f'{p}.cv2.{i}.1', fused=fused)
        box = F.conv2d(t, sd[f'{p}.cv2.{i}.2.weight'].to(x.dtype), sd[f'{p}.cv2.{i}.2.bias'].to(x.dtype))
        t = conv(conv(x, sd, f'{p}.cv3.{i}.0', fused=fused), sd, f'{p}.cv3.{i}.1', fused=fused)
        cls = F.conv2d(t, sd[f'{p}.cv3.{i}.2.weight'].to(x.dtype), sd[f'{p}.cv3.{i}.2.bias'].to(x.dtype))
        out.append(torch.cat((box, cls), 1))
    return out


def detect_decode(feats, strides, reg_max, nc):
    """Eval branch of Detect - nn/modules/head.py:165-177 -> y (B, 4+nc, A)."""
    b = feats[0].shape[0]
    no = 4 * reg_max + nc
    anchors, st = make_anchors([f.shape[2:] for f in feats], strides, 0.5, feats[0].dtype)
    x_cat = torch.cat([f.reshape(b, no, -1) for f in feats], 2)
    box, cls = x_cat.split((4 * reg_max, nc), 1)
    dbox = dist2bbox(dfl(box, reg_max), anchors.t().unsqueeze(0), xywh=True, dim=1) * st.t()
    return torch.cat((dbox, cls.sigmoid()), 1)


# ----------------------------------------------------------------------------- a16: graph from YAML dict
def make_divisible(x, d):
    return math.ceil(x / d) * d


_CH_SCALED = ('Conv', 'C2f', 'MSPA_C2f', 'SPPF')
_REPEAT_ARG = ('C2f', 'MSPA_C2f')


def plan_from_yaml(d, ch=3, scale=None):
    """Restates parse_model's arg rewriting - nn/tasks.py:604-699 - for the modules on the path.

    Returns a list of dicts {i, f, type, args, c2} plus the save-list.
    """
    nc = d['nc']
    scales = d.get('scales')
    depth, width, max_ch = 1.0, 1.0, float('inf')
    if scales:
        scale = scale or d.get('scale') or next(iter(scales))
        depth, width, max_ch = scales[scale]
    chs = [ch]
    rows, save = [], []
    for i, (f, n, m, args) in enumerate(d['backbone'] + d['head']):
        args = [nc if a == 'nc' else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        if m in _CH_SCALED:
            c1, c2 = chs[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            args = [c1, c2, *args[1:]]
            if m in _REPEAT_ARG:
                args.insert(2, n)
                n = 1
        elif m in ('Concat', 'SimFusion_4in'):
            c2 = sum(chs[x] for x in f)
        elif m in ('Detect', 'TOODHead'):
            args = [*args, [chs[x] for x in f]]
            c2 = None
        elif m == 'SimFusion_3in':
            c2 = args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            args = [[chs[x] for x in f], c2]
        elif m == 'IFM':
            c2 = sum(args[0])
            args = [chs[f], *args]
        elif m == 'InjectionMultiSum_Auto_pool':
            c2 = args[0]
            args = [chs[f[0]], *args]
        elif m == 'nn.Upsample':
            c2 = chs[f]
        else:
            raise KeyError(m)
        assert n == 1, 'no repeated plain layers on the target YAMLs'
        rows.append(dict(i=i, f=f, type=m, args=args, c2=c2))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chs = []
        chs.append(c2)
    return rows, sorted(save)


def model_forward(d, sd, x, strides, fused=False, scale=None, decode=True, return_layers=False):
    """`BaseModel._predict_once` (nn/tasks.py:65-87) over the rows of `plan_from_yaml`, eval mode.

    Returns (y, feats) like the reference's Detect eval branch; `strides` is the model's stride list
    (the reference probes it with a 640^2 zero image, tasks.py:241-245).
    """
    rows, save = plan_from_yaml(d, x.shape[1], scale)
    ys, layer_out = [], []
    for r in rows:
        f, t, a, p = r['f'], r['type'], r['args'], f"model.{r['i']}"
        if f != -1:
            xin = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        else:
            xin = x
        if t == 'Conv':
            k = a[2] if len(a) > 2 else 1
            s = a[3] if len(a) > 3 else 1
            assert sd[p + '.conv.weight'].shape[-1] == k
            x = conv(xin, sd, p, s=s, fused=fused)
        elif t == 'MSPA_C2f':
            x = mspa_c2f(xin, sd, p, a[2], a[3] if len(a) > 3 else False, fused)
        elif t == 'C2f':
            x = c2f(xin, sd, p, a[2], a[3] if len(a) > 3 else False, fused)
        elif t == 'SPPF':
            x = sppf(xin, sd, p, fused)
        elif t == 'SimFusion_4in':
            x = simfusion_4in(xin)
        elif t == 'SimFusion_3in':
            x = simfusion_3in(xin, sd, p, fused)
        elif t == 'IFM':
            x = ifm(xin, sd, p, 3, fused)
        elif t == 'InjectionMultiSum_Auto_pool':
            x = inject(xin, sd, p, a[2], a[3], fused)
        elif t == 'Concat':
            x = torch.cat(xin, 1)
        elif t == 'nn.Upsample':
            x = F.interpolate(xin, scale_factor=a[1], mode=a[2])
        elif t == 'Detect':
            feats = detect_raw(list(xin), sd, p, fused)
            x = (detect_decode(feats, strides, sd[p + '.dfl.conv.weight'].shape[1], a[0]), feats) if decode else feats
        elif t == 'TOODHead':
            from . import tood
            feats = tood.toodhead_raw(list(xin), sd, p)
            x = (detect_decode(feats, strides, 16, a[0]), feats) if decode else feats
        ys.append(x if r['i'] in save else None)
        if return_layers:
            layer_out.append(x)
    return (x, layer_out) if return_layers else x
