"""TEST INFRASTRUCTURE ONLY - CPU restatement of the TOODHead path.  PARITY UNPINNED.

The reference's TOODHead (nn/modules/head.py:466-572) needs mmcv (`ModulatedDeformConv2d`, `ConvModule`, `Scale`; imports at
head.py:14-20, block.py:12), which is neither vendored nor installed here, so the reference cannot be run and no fixture exists.
What follows restates (a) the head's own Python, line by line, and (b) mmcv's published modulated deformable convolution
(mmcv/ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh: `dmcn_im2col_bilinear`, `modulated_deformable_im2col_gpu_kernel`;
mmcv >= 1.3, the API block.py:401-432 is written against).  The HIP head is tested against this file only.
"""
import torch
import torch.nn.functional as F

from . import layers as L


def conv_gn(x, sd, p):
    """Conv_GN (head.py:67-81): conv (no bias, same padding) -> GroupNorm(16) -> SiLU."""
    w = sd[p + '.conv.weight'].to(x.dtype)
    y = F.conv2d(x, w, None, 1, w.shape[-1] // 2)
    return F.silu(F.group_norm(y, 16, sd[p + '.gn.weight'].to(x.dtype), sd[p + '.gn.bias'].to(x.dtype), 1e-5))


def task_decomposition(feat, avg_feat, sd, p, feat_channels, stacked):
    """TaskDecomposition.forward (head.py:107-131), norm_cfg=None.  As written in the reference the bmm path uses only
    `reduction_conv.conv.weight`: the ConvModule's bias parameter exists in the state_dict but is never added; then `activate` = ReLU."""
    b, c, h, w = feat.shape
    wgt = F.relu(F.conv2d(avg_feat, sd[p + '.la_conv1.weight'].to(feat.dtype), sd[p + '.la_conv1.bias'].to(feat.dtype)))
    wgt = torch.sigmoid(F.conv2d(wgt, sd[p + '.la_conv2.weight'].to(feat.dtype), sd[p + '.la_conv2.bias'].to(feat.dtype)))
    cw = wgt.reshape(b, 1, stacked, 1) * sd[p + '.reduction_conv.conv.weight'].to(feat.dtype).reshape(1, feat_channels, stacked, feat_channels)
    out = torch.bmm(cw.reshape(b, feat_channels, c), feat.reshape(b, c, h * w)).reshape(b, feat_channels, h, w)
    return F.relu(out)


def modulated_deform_conv3x3(x, offset, mask, weight, bias=None):
    """mmcv ModulatedDeformConv2d, kernel 3, stride 1, pad 1, dilation 1, groups 1, deform_groups 1.

    offset (B, 18, H, W): channel 2k = dy, 2k+1 = dx of kernel point k = i*3 + j; mask (B, 9, H, W) already in [0, 1].
    Sampling = dmcn_im2col_bilinear: value 0 unless -1 < h_im < H and -1 < w_im < W; each of the four corners counts only when it
    lies inside the image."""
    b, c, h, w = x.shape
    ys = torch.arange(h, dtype=x.dtype).view(1, h, 1)
    xs = torch.arange(w, dtype=x.dtype).view(1, 1, w)
    cols = []
    xf = x.reshape(b, c, h * w)
    for k in range(9):
        i, j = k // 3, k % 3
        hy = ys - 1 + i + offset[:, 2 * k]
        wx = xs - 1 + j + offset[:, 2 * k + 1]
        inside = (hy > -1) & (wx > -1) & (hy < h) & (wx < w)
        h0, w0 = torch.floor(hy), torch.floor(wx)
        lh, lw = hy - h0, wx - w0
        h0, w0 = h0.long(), w0.long()
        h1, w1 = h0 + 1, w0 + 1
        val = torch.zeros(b, c, h, w, dtype=x.dtype)
        for hh, ww, cf, ok in ((h0, w0, (1 - lh) * (1 - lw), (h0 >= 0) & (w0 >= 0)), (h0, w1, (1 - lh) * lw, (h0 >= 0) & (w1 <= w - 1)),
                               (h1, w0, lh * (1 - lw), (h1 <= h - 1) & (w0 >= 0)), (h1, w1, lh * lw, (h1 <= h - 1) & (w1 <= w - 1))):
            ok = ok & inside
            idx = (hh.clamp(0, h - 1) * w + ww.clamp(0, w - 1)).reshape(b, 1, h * w).expand(b, c, h * w)
            val = val + torch.gather(xf, 2, idx).reshape(b, c, h, w) * (cf * ok).unsqueeze(1)
        cols.append(val * mask[:, k].unsqueeze(1))
    col = torch.stack(cols, 2).reshape(b, c * 9, h * w)                  # (c, i, j) order = weight.view(out, c*9)
    out = torch.matmul(weight.reshape(weight.shape[0], -1).to(x.dtype), col).reshape(b, -1, h, w)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


def toodhead_raw(xs, sd, p):
    """TOODHead.forward up to the per-level raw maps (head.py:500-537)."""
    td = task_decomposition
    out = []
    for x in xs:
        f0 = conv_gn(x, sd, p + '.share_conv.0')
        f1 = conv_gn(f0, sd, p + '.share_conv.1')
        feat = torch.cat([f0, f1], 1)
        half = f0.shape[1]
        avg = F.adaptive_avg_pool2d(feat, (1, 1))
        cls_feat = td(feat, avg, sd, p + '.cls_decomp', half, 2)
        reg_feat = td(feat, avg, sd, p + '.reg_decomp', half, 2)
        om = F.conv2d(feat, sd[p + '.spatial_conv_offset.weight'].to(x.dtype), sd[p + '.spatial_conv_offset.bias'].to(x.dtype), 1, 1)
        offset, mask = om[:, :18], om[:, 18:].sigmoid()
        reg_feat = modulated_deform_conv3x3(reg_feat, offset, mask, sd[p + '.DyDCNV2.conv.weight'])
        reg_feat = F.group_norm(reg_feat, 16, sd[p + '.DyDCNV2.norm.weight'].to(x.dtype), sd[p + '.DyDCNV2.norm.bias'].to(x.dtype), 1e-5)
        prob = F.conv2d(F.relu(F.conv2d(feat, sd[p + '.cls_prob_conv1.weight'].to(x.dtype), sd[p + '.cls_prob_conv1.bias'].to(x.dtype))),
                        sd[p + '.cls_prob_conv2.weight'].to(x.dtype), sd[p + '.cls_prob_conv2.bias'].to(x.dtype), 1, 1).sigmoid()
        box = F.conv2d(F.relu(reg_feat), sd[p + '.cv2.weight'].to(x.dtype), sd[p + '.cv2.bias'].to(x.dtype))
        cls = F.conv2d(cls_feat * prob, sd[p + '.cv3.weight'].to(x.dtype), sd[p + '.cv3.bias'].to(x.dtype))
        out.append(torch.cat((box, cls), 1))
    return out


def toodhead(xs, sd, p, strides, nc):
    feats = toodhead_raw(xs, sd, p)
    return L.detect_decode(feats, strides, 16, nc), feats
