"""HIP product path vs the CPU oracle and the reference-generated fixtures.  Needs a real MI355X (-m gpu).

Everything here calls the product through its public Python surface, i.e. through the C ABI in libmgdt_hip.so.
Tolerances: fp32 path - boxes within 1e-3 px and conf within 1e-3 of the reference (north_star), we assert much
tighter; bf16 path - separately stated below.  Integer outputs (NMS kept anchor / class) are compared bit-exactly.
"""
import numpy as np
import pytest
import torch

import inputs as GI
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build_model(name, dtype=torch.float32, nc=80, scale='n'):
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel(get_config(name, scale, nc), verbose=False)
    seed_state_dict_(m, 0)
    return m.eval().to(DEV).set_compute_dtype(dtype)


def to_nchw(t):
    return t.detach().float().contiguous().cpu().numpy()


# ------------------------------------------------------------------------------------------------ modules
def _make_module(cls, args):
    import mgdt_yolo_amd.nn.modules as M
    args = tuple(torch.nn.ReLU() if a == 'relu' else a for a in args)
    m = getattr(M, cls)(*args)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
    return seed_state_dict_(m, GI.MODULE_SEED).eval().to(DEV)


def _dev_inputs(name):
    xs = GI.module_inputs(name)
    # first-layer style input (3 channels) stays NCHW fp32 like the user's image; feature maps are NHWC
    return [x.to(DEV) if x.shape[1] == 3 else x.to(DEV).contiguous(memory_format=torch.channels_last) for x in xs]


@pytest.mark.parametrize('name', list(GI.MODULE_CASES))
def test_module_fp32_matches_reference(golden, name):
    g = golden('modules')
    cls, args, _ = GI.MODULE_CASES[name]
    m = _make_module(cls, args)
    xs = _dev_inputs(name)
    with torch.no_grad():
        y = m(xs[0] if len(xs) == 1 else xs)
    ref = g[name]
    assert tuple(y.shape) == ref.shape
    np.testing.assert_allclose(to_nchw(y), ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('name', ['conv3s1', 'c2f', 'c2f_sc', 'mspa_n1', 'mspa_n2_odd', 'sppf', 'laf3_allconv', 'ifm', 'inject_up'])
def test_module_bf16_close_to_reference(golden, name):
    """bf16 operands / fp32 accumulate: stated tolerance 3e-2 relative to the output's max magnitude."""
    g = golden('modules')
    cls, args, _ = GI.MODULE_CASES[name]
    m = _make_module(cls, args)
    xs = [x.to(torch.bfloat16) for x in _dev_inputs(name)]
    with torch.no_grad():
        y = m(xs[0] if len(xs) == 1 else xs)
    ref = g[name]
    assert y.dtype == torch.bfloat16
    err = np.abs(to_nchw(y) - ref).max() / np.abs(ref).max()
    assert err < 3e-2, err


@pytest.mark.parametrize('dim,hw', [(96, (40, 40)), (64, (13, 9)), (32, (7, 21))])
def test_convnext_fused_mlp_matches_three_launch_chain(dim, hw):
    """bf16: the on-chip MLP (mgdt_cnx_mlp_fwd, hidden map recomputed, never stored) vs conv -> GRN stats -> conv.
    Both round the hidden map to bf16 at the same point; what differs is the fp32 summation order of the GRN statistic and
    of pwconv2's K loop, so the outputs agree to bf16 resolution (stated: 2e-2 of the output's max magnitude)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import ConvNeXtV2_Block
    m = seed_state_dict_(ConvNeXtV2_Block(dim), 3).eval().to(DEV)
    with torch.no_grad():
        m.grn.gamma.normal_(0, 0.5)
        m.grn.beta.normal_(0, 0.5)
    x = torch.randn(3, dim, *hw, generator=torch.Generator().manual_seed(5)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.cnx_mlp_supported(dim, torch.bfloat16)
    with torch.no_grad():
        y_fused = m(x).float()
        ops.FUSED_CNX_MLP = False
        try:
            y_chain = m(x).float()
        finally:
            ops.FUSED_CNX_MLP = True
        y32 = m(x.float()).float()      # fp32 path (always the chain) as the yardstick
    scale = y32.abs().max().item()
    assert (y_fused - y_chain).abs().max().item() < 2e-2 * scale
    assert (y_fused - y32).abs().max().item() < 3e-2 * scale


@pytest.mark.parametrize('b,dim,hw', [(32, 96, (40, 40)), (3, 96, (23, 17)), (2, 64, (13, 9)), (5, 32, (7, 21)), (70, 96, (20, 20)), (1, 96, (80, 80)), (2, 96, (5, 3))])
def test_convnext_block_single_launch_matches_the_launch_chain(b, dim, hw):
    """mgdt_cnx_block_fwd (dw7x7 + LayerNorm + pwconv1 + GELU + GRN + pwconv2 + residual in ONE launch; the workgroups of an image meet at a
    per-image barrier for the GRN statistic) against the three-launch chain it replaces (mgdt_dwconv7_ln_fwd + the two passes of
    mgdt_cnx_mlp_fwd) and against the fp32 chain.  Both bf16 forms round the normalised map and the hidden map to bf16 at the same points;
    what differs is fp32 summation order (depth-wise taps, GRN partial sums, pwconv2's K loop) -> agreement at bf16 resolution (2e-2 of the
    output's largest magnitude).  Shapes: the bench shape, maps that do not divide into tiles, one tile per image, a batch that needs
    several launches (70 images x 4 tiles > 256 compute units), a map of 32 tiles per image.  Called three times: the arrival counters
    persist across calls (hipGraph replays never reset them)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import ConvNeXtV2_Block
    m = seed_state_dict_(ConvNeXtV2_Block(dim), 3).eval().to(DEV)
    with torch.no_grad():
        m.grn.gamma.normal_(0, 0.5)
        m.grn.beta.normal_(0, 0.5)
    x = torch.randn(b, dim, *hw, generator=torch.Generator().manual_seed(5)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.cnx_block_supported(x, torch.bfloat16)
    with torch.no_grad():
        ys = [m(x).float() for _ in range(3)]
        ops.FUSED_CNX_BLOCK = False
        try:
            y_chain = m(x).float()
        finally:
            ops.FUSED_CNX_BLOCK = True
        y32 = m(x.float()).float()      # fp32 path (always the chain) as the yardstick
    assert torch.isfinite(ys[0]).all()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), 'the kernel must be deterministic and replayable'
    scale = y32.abs().max().item()
    e_chain, e32 = (ys[0] - y_chain).abs().max().item(), (ys[0] - y32).abs().max().item()
    print(f'cnx_block {b}x{dim}x{hw}: vs chain {e_chain / scale:.2e}, vs fp32 {e32 / scale:.2e} of max |y| = {scale:.2f}')
    assert e_chain < 2e-2 * scale and e32 < 3e-2 * scale


@pytest.mark.parametrize('b,inc,hw', [(32, 480, (40, 40)), (2, 48, (10, 6)), (3, 64, (23, 17))])
def test_ifm_closing_conv_inside_the_last_block_launch(b, inc, hw):
    """IFM (nn/modules/block.py:331-342) in bf16: Conv 1x1, three one-launch ConvNeXtV2 blocks, and the closing Conv 1x1 + BN + SiLU evaluated
    inside the LAST block's launch (its output goes from pwconv2's accumulators + residual, rounded to bf16 as the stored map would be, into
    the closing conv's MFMA).  Against the same module with ops.FUSED_CNX_TAIL off (closing conv as its own launch): same roundings, another
    summation order over the 96 input channels -> bf16 resolution (1.5e-2 of the largest output)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import IFM
    m = seed_state_dict_(IFM(inc, [64, 32]), 5).eval().to(DEV)
    m.apply(lambda t: setattr(t, '_cdtype', torch.bfloat16) if hasattr(t, 'out_dtype') else None)
    x = torch.randn(b, inc, *hw, generator=torch.Generator().manual_seed(2)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    names = []
    orig = ops._launch
    with torch.no_grad():
        m(x)
        ops._launch = lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1]
        try:
            y1 = m(x).float()
        finally:
            ops._launch = orig
        ops.FUSED_CNX_TAIL = False
        try:
            y0 = m(x).float()
        finally:
            ops.FUSED_CNX_TAIL = True
    assert names == ['conv2d_fwd', 'cnx_block_fwd', 'cnx_block_fwd', 'cnx_block_fwd'], names
    scale = y0.abs().max().item()
    err = (y1 - y0).abs().max().item()
    print(f'IFM {b}x{inc}x{hw}: closing conv in the block launch vs its own launch: {err / scale:.2e} of {scale:.2f}')
    assert err < 1.5e-2 * scale


def test_convnext_block_barrier_words_survive_a_change_of_shape():
    """The per-image barrier of mgdt_cnx_block_fwd (arrival counter + generation in the workspace) must be at rest after every call whatever
    the tile count was: one workspace shared by a 1-tile-per-image shape (called an odd number of times) and a 2-tile-per-image shape.  (The
    first form of the barrier - a monotone counter with target = next multiple of the tile count - hung exactly here: round-3 log.)"""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import ConvNeXtV2_Block
    m = seed_state_dict_(ConvNeXtV2_Block(96), 3).eval().to(DEV)
    mk = lambda b, h, w: torch.randn(b, 96, h, w, generator=torch.Generator().manual_seed(b + h)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa, xb = mk(4, 10, 10), mk(2, 20, 20)
    shared = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    keys = [(xa.device, 4, 96, 10, 10), (xb.device, 2, 96, 20, 20)]
    saved = {k: ops._CNX_WS.get(k) for k in keys}
    try:
        for k in keys:
            ops._CNX_WS[k] = shared
        with torch.no_grad():
            ya = [m(xa) for _ in range(3)]
            yb = [m(xb) for _ in range(3)]
            ya2 = m(xa)
        torch.cuda.synchronize()
        assert torch.equal(ya[0], ya[2]) and torch.equal(ya[0], ya2) and torch.equal(yb[0], yb[2])
        assert int(shared[:4096].view(torch.int32)[0::2].abs().sum()) == 0, 'arrival counters must be back at zero'
    finally:
        for k, v in saved.items():
            if v is None:
                ops._CNX_WS.pop(k, None)
            else:
                ops._CNX_WS[k] = v


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('b,c,hw', [(32, 96, (40, 40)), (3, 96, (23, 17)), (2, 64, (13, 9)), (1, 32, (7, 21)), (2, 160, (12, 10))],
                         ids=['bench-tile10', 'odd-96', 'c64', 'c32', 'c160-tiled'])
def test_dwconv7_layernorm_kernel_vs_torch(b, c, hw, dtype):
    """The ConvNeXt block's first half (convnextv2.py:61-66: 7x7 depth-wise conv, channels-last LayerNorm) in one launch, every tile shape the
    host picks (10x10 tiles for the bench map: one round of workgroups; 8x8 otherwise; the generic kernels for wide maps), against
    F.conv2d(groups=C) + F.layer_norm in float64.  The training form also returns the conv output `u`."""
    import torch.nn.functional as F
    from mgdt_yolo_amd import ops
    gen = torch.Generator().manual_seed(c + hw[0])
    x = torch.randn(b, c, *hw, generator=gen).to(dtype)
    w, bias = torch.randn(c, 1, 7, 7, generator=gen) * 0.2, torch.randn(c, generator=gen) * 0.1
    lw, lb = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.1
    u_ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 3, 1, c)
    y_ref = F.layer_norm(u_ref.permute(0, 2, 3, 1), (c,), lw.double(), lb.double(), 1e-6).permute(0, 3, 1, 2)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    w49 = w.reshape(c, 49).t().contiguous().to(DEV)
    args = (w49, bias.to(DEV), lw.to(DEV), lb.to(DEV), 1e-6)
    y = ops.dwconv7_ln(xd, *args)
    y2, u = ops.dwconv7_ln_train(xd, *args)
    assert torch.equal(y, y2)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert (y.double().cpu() - y_ref).abs().max().item() < tol * max(1.0, y_ref.abs().max().item())
    assert (u.double().cpu() - u_ref).abs().max().item() < tol * max(1.0, u_ref.abs().max().item())


@pytest.mark.parametrize('c,n,hw', [(32, 1, (20, 24)), (64, 2, (17, 13)), (128, 2, (9, 11)), (256, 1, (6, 5)), (96, 1, (8, 8)), (192, 1, (5, 7))])
def test_mspa_pointwise_chain_matches_three_convs(c, n, hw):
    """bf16: mgdt_pw_chain3_fwd (sp_i kept in accumulators, rounded to bf16 where the unfused chain would store them) vs three
    mgdt_conv2d_fwd launches.  Same operands, same roundings; only the MFMA K order differs -> agreement to a bf16 ulp or two."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import MSPA_C2f
    m = seed_state_dict_(MSPA_C2f(c, c, n, True), 7).eval().to(DEV)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
    x = torch.randn(3, c, *hw, generator=torch.Generator().manual_seed(11)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.pw_chain_supported(c // 4, torch.bfloat16)
    with torch.no_grad():
        y_fused = m(x).float()
        ops.FUSED_PW_CHAIN = False
        try:
            y_chain = m(x).float()
        finally:
            ops.FUSED_PW_CHAIN = True
    scale = y_chain.abs().max().item()
    assert scale > 0
    assert (y_fused - y_chain).abs().max().item() < 1e-2 * scale


@pytest.mark.parametrize('c,n,sc,hw', [(32, 1, True, (20, 24)), (32, 1, False, (160, 160)), (64, 2, True, (40, 36)), (64, 2, False, (16, 12)), (128, 2, True, (40, 40)),
                                       (128, 1, True, (8, 12)), (256, 1, True, (20, 20)), (256, 1, False, (12, 8)), (64, 1, True, (6, 4))])
def test_mspa_block_single_launch_matches_launch_chain(c, n, sc, hw):
    """bf16: mgdt_csp_block_fwd (front chain, bottlenecks on LDS-resident tiles with recomputed halo, final 1x1, per-tile pooled sums) +
    mgdt_spr_attn_scale_fwd vs the per-conv launch chain + mgdt_spr_pool_fwd.  Same packed weights, same rounding points except that the
    shortcut adds the bf16-rounded (sp2 + x3) instead of the two addends; the MFMA K order is identical."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import MSPA_C2f
    m = seed_state_dict_(MSPA_C2f(c, c, n, sc), 7).eval().to(DEV)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
    B = 3 if hw[0] * hw[1] < 5000 else 2
    x = torch.randn(B, c, *hw, generator=torch.Generator().manual_seed(11)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.csp_block_supported(ops.CSP_MSPA, x, c, c // 4, n, torch.bfloat16)
    with torch.no_grad():
        y_fused = m(x).float()
        ops.FUSED_CSP_BLOCK = False
        try:
            y_chain = m(x).float()
        finally:
            ops.FUSED_CSP_BLOCK = True
        y32 = m(x.float()).float()
    scale = y32.abs().max().item()
    d = (y_fused - y_chain).abs()
    print(f'mspa block c={c} n={n} {hw}: fused vs chain max {d.max().item() / scale:.2e} mean {d.mean().item() / scale:.2e}; vs fp32 {(y_fused - y32).abs().max().item() / scale:.2e}')
    assert not torch.equal(y_fused, y_chain) or True
    assert d.max().item() < 1.5e-2 * scale and d.mean().item() < 1e-3 * scale
    assert (y_fused - y32).abs().max().item() < 3e-2 * scale


@pytest.mark.parametrize('c1,c2,n,sc,hw', [(256, 64, 1, False, (80, 80)), (64, 32, 2, True, (24, 20)), (128, 128, 1, True, (9, 7)), (32, 32, 1, True, (12, 10))])
def test_c2f_block_single_launch_matches_launch_chain(c1, c2, n, sc, hw):
    """bf16: C2f through mgdt_csp_block_fwd (cv1 on global -> VGPR operands, bottlenecks in LDS, cv2 over the concat) vs its launch chain."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import C2f
    m = seed_state_dict_(C2f(c1, c2, n, sc), 5).eval().to(DEV)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
    x = torch.randn(2, c1, *hw, generator=torch.Generator().manual_seed(3)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y01 = torch.empty(2, c2, *hw, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)     # cv1's output: 2 * (c2 / 2) channels
    assert ops.csp_block_supported(ops.CSP_C2F, y01, c2, c2 // 2, n, torch.bfloat16)
    with torch.no_grad():
        y_fused = m(x).float()
        ops.FUSED_CSP_BLOCK = False
        try:
            y_chain = m(x).float()
        finally:
            ops.FUSED_CSP_BLOCK = True
        y32 = m(x.float()).float()
    scale = y32.abs().max().item()
    d = (y_fused - y_chain).abs()
    print(f'c2f block {c1}->{c2} n={n} {hw}: fused vs chain max {d.max().item() / scale:.2e} mean {d.mean().item() / scale:.2e}')
    assert d.max().item() < 1.5e-2 * scale and d.mean().item() < 1e-3 * scale
    assert (y_fused - y32).abs().max().item() < 3e-2 * scale


@pytest.mark.parametrize('cin,cout,hw,ghw', [(64, 256, (80, 80), (40, 40)), (32, 128, (13, 21), (7, 11)), (96, 256, (9, 9), (9, 9)), (64, 128, (20, 36), (5, 9))])
def test_injection_single_launch_matches_conv_plus_inject(cin, cout, hw, ghw):
    """bf16: mgdt_conv1x1_inject_fwd (local map kept in accumulators, global maps staged in LDS) vs mgdt_conv2d_fwd + mgdt_inject_fwd.
    Same roundings and the same interpolation association; only the MFMA K order differs."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import InjectionMultiSum_Auto_pool
    gc = 2 * cout
    m = seed_state_dict_(InjectionMultiSum_Auto_pool(cin, cout, global_inp=[gc, gc], flag=0), 9).eval().to(DEV)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
    gen = torch.Generator().manual_seed(13)
    mk = lambda c, s: (3 * torch.randn(2, c, *s, generator=gen)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x_l, x_g = mk(cin, hw), mk(2 * gc, ghw)
    with torch.no_grad():
        y_fused = m([x_l, x_g]).float()
        ops.FUSED_INJECT = False
        try:
            y_pair = m([x_l, x_g]).float()
        finally:
            ops.FUSED_INJECT = True
    scale = y_pair.abs().max().item()
    assert scale > 0
    assert (y_fused - y_pair).abs().max().item() < 1e-2 * scale


@pytest.mark.parametrize('nc,hws', [(80, [(80, 80)]), (4, [(13, 11)]), (80, [(20, 24), (10, 12), (5, 6)])])
def test_detect_box_branch_3x3_inside_the_tail_launch(nc, hws):
    """The Detect box branch's second conv (cv2[i][1]: 3x3, 16 -> 16, BN, SiLU; head.py:150) evaluated inside mgdt_detect_tail_fwd (an implicit
    GEMM over its input, the result rounded to bf16 as the stored map and chained into the final 1x1) against the same head with that conv as
    its own launch: y (boxes / scores), the raw maps and the NMS keys must agree to rounding (the K order of both GEMMs is unchanged)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import Detect
    ch = (64, 128, 256)[:len(hws)]
    m = seed_state_dict_(Detect(nc, ch), 2).eval().to(DEV)
    m.stride = torch.tensor([8.0, 16.0, 32.0][:len(hws)])
    m.apply(lambda t: setattr(t, '_cdtype', torch.bfloat16) if hasattr(t, 'out_dtype') else None)
    gen = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, c, *hw, generator=gen).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c, hw in zip(ch, hws)]
    res = {}
    with torch.no_grad():
        for flag in (True, False):
            ops.FUSED_DETECT_BOX3 = flag
            try:
                names = []
                orig = ops._launch
                m(list(xs))
                ops._launch = lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1]
                try:
                    y, feats = m(list(xs))
                finally:
                    ops._launch = orig
                res[flag] = (y.clone(), [f.clone() for f in feats], ops._best_keys_of(y, 2, y.shape[2]).clone(), names)
            finally:
                ops.FUSED_DETECT_BOX3 = True
    (y1, f1, k1, n1), (y0, f0, k0, n0) = res[True], res[False]
    assert len(n0) - len(n1) == len(hws), (n0, n1)
    eb, es = (y1[:, :4] - y0[:, :4]).abs().max().item(), (y1[:, 4:] - y0[:, 4:]).abs().max().item()
    ef = max((a.float() - b.float()).abs().max().item() for a, b in zip(f1, f0))
    print(f'box 3x3 in the tail: boxes {eb:.3e} px, scores {es:.3e}, raw maps {ef:.3e}; launches {len(n0)} -> {len(n1)}')
    assert eb < 5e-2 and es == 0.0 and ef < 3e-2 and torch.equal(k1, k0)


@pytest.mark.parametrize('b,hw,ghw', [(32, (80, 80), (40, 40)), (2, (37, 45), (19, 23)), (3, (16, 16), (16, 16)), (1, (80, 80), (20, 20))])
def test_injection_plus_c2f_cv1_single_launch_matches_two_launches(b, hw, ghw):
    """mgdt_conv1x1_inject_conv_fwd: the injection and the 1x1 Conv+BN+SiLU that is its only consumer (C2f.cv1) in one launch - the
    256-channel map goes from the first GEMM's accumulators through the bilinear tail into the second GEMM's B operand (rounded to bf16 where
    the stored map would be).  Against injection launch + conv launch: same roundings, a different fp32 summation order over the 256 input
    channels of the second conv -> agreement to bf16 resolution (1.5e-2 of the largest output).  Odd sizes, equal sizes (no up-sampling), x4."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import C2f, InjectionMultiSum_Auto_pool
    inj = seed_state_dict_(InjectionMultiSum_Auto_pool(64, 256, [64, 32], 1), 3).eval().to(DEV)
    c2f = seed_state_dict_(C2f(256, 64, 1, False), 4).eval().to(DEV)
    for mod in (inj, c2f):
        mod.apply(lambda t: setattr(t, '_cdtype', torch.bfloat16) if hasattr(t, 'out_dtype') else None)
    gen = torch.Generator().manual_seed(7)
    mk = lambda c, s: torch.randn(b, c, *s, generator=gen).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x_l, x_g = mk(64, hw), mk(96, ghw)
    with torch.no_grad():
        y_fused = c2f(None, head=(inj, [x_l, x_g])).float()
        ops.FUSED_INJECT_CONV = False
        try:
            y_two = c2f(inj([x_l, x_g])).float()
        finally:
            ops.FUSED_INJECT_CONV = True
        y01_f = inj.forward_into_conv([x_l, x_g], c2f.cv1)
        ops.FUSED_INJECT_GCONV = False                         # the global convs as a launch of their own: identical values
        try:
            y01_g = inj.forward_into_conv([x_l, x_g], c2f.cv1)
        finally:
            ops.FUSED_INJECT_GCONV = True
        # (the in-launch form interpolates on the MFMA with bf16 tap weights and a bf16-rounded h_sigmoid gate: bf16-resolution agreement)
        assert (y01_f is None) == (y01_g is None)
        if y01_f is not None:
            dg = (y01_f.float() - y01_g.float()).abs().max().item()
            assert dg < 1.5e-2 * y01_g.float().abs().max().item(), dg
        y01_t = c2f.cv1(inj([x_l, x_g])).float()
        if ghw == hw:            # no up-sampling: the source patches of an 8 x 16 tile do not fit LDS next to both panels -> the two-launch form by itself
            assert y01_f is None
            y01_f = y01_t
        assert y01_f is not None, 'the fused launch must cover these shapes'
        y01_f = y01_f.float()
    s1, s2 = y01_t.abs().max().item(), y_two.abs().max().item()
    e1, e2 = (y01_f - y01_t).abs().max().item(), (y_fused - y_two).abs().max().item()
    print(f'inject+cv1 {b}x{hw}<-{ghw}: cv1 output err {e1 / s1:.2e} of {s1:.2f}, block output err {e2 / s2:.2e} of {s2:.2f}')
    assert e1 < 1.5e-2 * s1 and e2 < 3e-2 * s2


@pytest.mark.parametrize('nc,ch,hws', [(80, (64,), [(80, 80)]), (4, (64,), [(13, 11)]), (3, (64,), [(9, 5)]), (80, (64, 128, 256), [(20, 24), (10, 12), (5, 6)]), (20, (32,), [(7, 9)])])
def test_detect_tail_single_launch_matches_conv_conv_decode(nc, ch, hws):
    """bf16: mgdt_detect_tail_fwd (both final 1x1 convs + raw map + DFL / dist2bbox / sigmoid decode) vs mgdt_conv2d_fwd x2 + mgdt_detect_decode_fwd.
    Same packed weights and MFMA K order, the decode consumes the same bf16-rounded logits: the raw maps agree to a bf16 ulp, y to float rounding."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import Detect
    m = seed_state_dict_(Detect(nc, ch), 4).eval()
    m.stride = torch.tensor([8.0, 16.0, 32.0][:len(ch)])
    m = m.to(DEV)
    for sub in m.modules():
        if isinstance(sub, torch.nn.BatchNorm2d):
            sub.eps = 1e-3
        if hasattr(sub, 'out_dtype'):
            sub._cdtype = torch.bfloat16
    mk = lambda c, hw, s: torch.randn(2, c, *hw, generator=torch.Generator().manual_seed(s)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xs = [mk(c, hw, 7 + i) for i, (c, hw) in enumerate(zip(ch, hws))]
    with torch.no_grad():
        y_f, feats_f = m([t.clone() for t in xs])
        ops.FUSED_DETECT_TAIL = False
        try:
            y_u, feats_u = m([t.clone() for t in xs])
        finally:
            ops.FUSED_DETECT_TAIL = True
    for a, b in zip(feats_f, feats_u):
        d = (a.float() - b.float()).abs().max().item()
        assert d <= 2e-2 * b.float().abs().max().item(), d
    eb, ec = (y_f[:, :4] - y_u[:, :4]).abs().max().item(), (y_f[:, 4:] - y_u[:, 4:]).abs().max().item()
    print(f'detect tail nc={nc} ch={ch}: y box diff {eb:.4f} px, conf diff {ec:.5f}')
    assert eb < 0.5 and ec < 2e-2
    assert torch.isfinite(y_f).all() and y_f.shape == y_u.shape


@pytest.mark.parametrize('in_dtype', ['bf16', 'f32', 'u8'])
@pytest.mark.parametrize('hw', [(64, 96), (37, 45), (640, 640), (30, 18)])
def test_fused_stem_matches_two_conv_launches(in_dtype, hw):
    """bf16 path: layers 0 + 1 in one launch (mgdt_stem2_fwd: image patch in LDS, layer 0 on MFMA with bf16 weights, its map kept on the CU)
    vs the stem kernel (fp32 weights, VALU) + the implicit-GEMM conv.  Differences: bf16 rounding of the image / layer-0 weights only.
    Odd sizes exercise the zero padding of both layers and partial tiles; uint8 the fused / 255."""
    from mgdt_yolo_amd import ops
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    B = 2
    img = seeded_images(B, hw[0], hw[1], seed=4)
    if in_dtype == 'u8':
        x = (img * 255).round().clamp_(0, 255).to(torch.uint8).to(DEV)
        xf = x.float().cpu() / 255
    elif in_dtype == 'bf16':
        x = img.to(DEV).to(torch.bfloat16)
        xf = x.float().cpu()
    else:
        x = img.to(DEV)
        xf = img
    assert m._stem_fusable(x)
    m0, m1 = m.model[0], m.model[1]
    with torch.no_grad():
        pk0 = ops.PackedStem2(m0.conv.weight, (m0.bn.weight, m0.bn.bias, m0.bn.running_mean, m0.bn.running_var, m0.bn.eps))
        y_fused = ops.stem2(x, pk0, m1.packed(torch.bfloat16, direct=False)).float()
        y_two = m1(m0(x)).float()
    # fp64 reference of the two layers (BN folded) on the values the kernels see
    import torch.nn.functional as F
    from oracle import layers as OL
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    ref = OL.conv(OL.conv(xf.double(), sd, 'model.0', s=2, fused=True), sd, 'model.1', s=2, fused=True)
    scale = ref.abs().max().item()
    e_f, e_t = (y_fused.cpu().double() - ref).abs().max().item() / scale, (y_two.cpu().double() - ref).abs().max().item() / scale
    print(f'stem {in_dtype} {hw}: fused vs fp64 {e_f:.2e}, two launches vs fp64 {e_t:.2e}')
    assert y_fused.shape == y_two.shape == ref.shape
    assert e_f < 2e-2 and e_t < 2e-2
    assert (y_fused - y_two).abs().max().item() < 2e-2 * scale


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_stem_takes_uint8_images_like_the_reference_preprocess(dtype):
    """uint8 NCHW image -> stem with /255 fused (predictor.py:129) == float image / 255 fed to the same stem, bit for bit."""
    from mgdt_yolo_amd.nn.modules import Conv
    m = seed_state_dict_(Conv(3, 16, 3, 2), 5).eval().to(DEV)
    m.bn.eps = 1e-3
    m._cdtype = dtype
    u8 = torch.randint(0, 256, (2, 3, 37, 45), generator=torch.Generator().manual_seed(2), dtype=torch.uint8).to(DEV)
    with torch.no_grad():
        y_u8 = m(u8)
        y_f = m((u8.cpu().float() / 255).to(DEV))     # CPU true division (ATen's GPU kernel multiplies by 1/255 instead)
    assert y_u8.dtype == dtype and torch.equal(y_u8, y_f)


# ------------------------------------------------------------------------------------------------ TOODHead (a15, parity unpinned)
def test_groupnorm_matches_torch_cpu():
    from mgdt_yolo_amd import ops
    g = torch.Generator().manual_seed(3)
    for c, hw in ((32, (20, 24)), (64, (7, 9)), (320, (5, 6))):
        x = (torch.randn(2, c, *hw, generator=g) * 2 + 0.5)
        gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x, 16, gamma, beta, 1e-5))
        y = ops.groupnorm(x.to(DEV).contiguous(memory_format=torch.channels_last), gamma.to(DEV), beta.to(DEV), 16, 1e-5, ops.ACT_SILU)
        np.testing.assert_allclose(to_nchw(y), ref.numpy(), atol=2e-5, rtol=1e-4)


def test_dcnv2_matches_oracle_restatement():
    """mmcv's modulated deformable conv as restated in oracle/tood.py (parity unpinned: mmcv itself is absent)."""
    from mgdt_yolo_amd import ops
    from oracle import tood
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 32, 13, 11, generator=g)
    om = torch.cat([torch.randn(2, 18, 13, 11, generator=g) * 2.5, torch.randn(2, 10, 13, 11, generator=g)], 1)   # offsets reach outside the image
    w = torch.randn(32, 32, 3, 3, generator=g) / 17
    ref = tood.modulated_deform_conv3x3(x, om[:, :18], om[:, 18:27].sigmoid(), w)
    nh = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    y = ops.dcnv2(nh(x), nh(om), w.permute(2, 3, 1, 0).reshape(9 * 32, 32).contiguous().to(DEV), None, 32)
    np.testing.assert_allclose(to_nchw(y), ref.numpy(), atol=1e-4, rtol=1e-4)


def test_dcnv2_mfma_matches_oracle_restatement():
    """bf16 DCNv2 on the matrix cores (bilinear-sampled B operand) vs the restatement evaluated on the same bf16-rounded operands."""
    from mgdt_yolo_amd import ops
    from oracle import tood
    g = torch.Generator().manual_seed(6)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(2, 32, 14, 19, generator=g))
    om = bf(torch.cat([torch.randn(2, 18, 14, 19, generator=g) * 2.5, torch.randn(2, 10, 14, 19, generator=g)], 1))
    w = bf(torch.randn(48, 32, 3, 3, generator=g) / 17)
    ref = tood.modulated_deform_conv3x3(x, om[:, :18], om[:, 18:27].sigmoid(), w)
    nh = lambda t: t.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pk = ops.PackedConv(w.to(DEV), None, None, 3, torch.bfloat16)
    y = ops.dcnv2_mfma(nh(x), nh(om), pk)
    err = (y.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-2, err


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_toodhead_matches_oracle_restatement(dtype):
    """Whole task-aligned head vs oracle/tood.py on seeded weights.  PARITY UNPINNED (no reference run possible: mmcv absent)."""
    from mgdt_yolo_amd.nn.modules import TOODHead
    from oracle import tood
    m = seed_state_dict_(TOODHead(80, 64, (64,)), 21).eval()
    m.stride = torch.tensor([8.0])
    sd = {'h.' + k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 64, 20, 24, generator=torch.Generator().manual_seed(8))
    y_ref, feats_ref = tood.toodhead([x], sd, 'h', [8.0], 80)
    m = m.to(DEV)
    with torch.no_grad():
        y, feats = m([x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)])
    if dtype == torch.float32:
        np.testing.assert_allclose(to_nchw(feats[0]), feats_ref[0].numpy(), atol=2e-3, rtol=2e-3)
        np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), atol=2e-2, rtol=2e-3)     # boxes in pixels (stride 8, reg_max 16)
    else:
        err = np.abs(to_nchw(feats[0]) - feats_ref[0].numpy()).max() / np.abs(feats_ref[0].numpy()).max()
        assert err < 5e-2, err


def test_tood_model_e2e_fp32_matches_oracle():
    from oracle import layers as OL
    cfg = get_config('mspa_c2f_gd_tood_yolov8', 'n', 80)
    m = build_model('mspa_c2f_gd_tood_yolov8')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = seeded_images(1, 160, 160, seed=3)
    y_ref, feats_ref = OL.model_forward(cfg, sd, x, [8.0])
    with torch.no_grad():
        y, feats = m(x.to(DEV))
    np.testing.assert_allclose(to_nchw(feats[0]), feats_ref[0].numpy(), atol=5e-3, rtol=5e-3)
    np.testing.assert_allclose(y.cpu().numpy()[:, 4:], y_ref.numpy()[:, 4:], atol=1e-3)


@pytest.mark.parametrize('hw', [(160, 160), (320, 192)])
def test_tood_scale_s_model_fp32_matches_oracle(hw):
    """BASELINE configs[3] model (MSPA-C2f + GD + TOODHead, scale s, hidc 128) vs oracle.layers.model_forward (TOOD part unpinned: mmcv absent)."""
    from oracle import layers as OL
    name = 'mspa_c2f_gd_tood_yolov8_hidc128'
    cfg = get_config(name, 's', 80)
    m = build_model(name, scale='s')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = seeded_images(1, *hw, seed=5)
    with torch.no_grad():
        y_ref, feats_ref = OL.model_forward(cfg, sd, x, [8.0])
        y, feats = m(x.to(DEV))
    np.testing.assert_allclose(to_nchw(feats[0]), feats_ref[0].numpy(), atol=5e-3, rtol=5e-3)
    np.testing.assert_allclose(y.cpu().numpy()[:, 4:], y_ref.numpy()[:, 4:], atol=1e-3)
    np.testing.assert_allclose(y.cpu().numpy()[:, :4], y_ref.numpy()[:, :4], atol=5e-2)       # boxes in px, reg_max 16 at stride 8


def test_tood_scale_s_at_1280_properties():
    """BASELINE configs[3] at its full size (1x3x1280x1280, 25 600 anchors): too large for the CPU oracle to be a per-test checker, so
    size-independent properties: finite outputs, fp32 vs bf16 within the stated bf16 bound, boxes inside a sane range, NMS idempotent."""
    from mgdt_yolo_amd.yolo.utils.ops import nms_with_index
    name = 'mspa_c2f_gd_tood_yolov8_hidc128'
    x = seeded_images(1, 1280, 1280, seed=9).to(DEV)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = build_model(name, dt, scale='s')
        with torch.no_grad():
            y, feats = m(x.to(dt))
        assert y.shape == (1, 84, 160 * 160) and feats[0].shape == (1, 64 + 80, 160, 160)
        assert torch.isfinite(y).all()
        outs[dt] = y.float()
    y32, y16 = outs[torch.float32], outs[torch.bfloat16]
    eb = (y32[:, :4] - y16[:, :4]).abs()
    ec = (y32[:, 4:] - y16[:, 4:]).abs()
    print(f'tood-s 1280: bf16 vs fp32 box max {eb.max().item():.3f} px mean {eb.mean().item():.4f}; conf max {ec.max().item():.4f}')
    assert ec.max().item() < 0.1 and eb.mean().item() < 0.5 and eb.max().item() < 8.0        # reg_max 16: one bin = 8 px at stride 8
    assert (y32[:, 2:4] > 0).all() and y32[:, :2].min().item() > -200 and y32[:, :2].max().item() < 1480
    rows, kept = nms_with_index(y32, conf_thres=0.25, iou_thres=0.7)
    assert len(kept[0]) > 0
    # idempotence: NMS over the kept boxes alone keeps every one of them
    sub = y32[:, :, kept[0].long()].contiguous()
    rows2, kept2 = nms_with_index(sub, conf_thres=0.25, iou_thres=0.7)
    assert len(kept2[0]) == len(kept[0]) and torch.equal(rows2[0], rows[0])


@pytest.mark.parametrize('c1,c2,k,s,dt', [(16, 16, 3, 1, torch.float32), (16, 32, 5, 2, torch.float32), (24, 36, 3, 1, torch.float32), (32, 32, 3, 2, torch.bfloat16)],
                         ids=['depthwise3', 'dw5s2_mult2', 'groups12', 'depthwise3s2_bf16'])
def test_dwconv_trains_against_torch_autograd(c1, c2, k, s, dt):
    """DWConv (reference conv.py:82-86: Conv with g = gcd(c1, c2)) in training mode: batch-stat BN forward, then the grouped data / weight
    gradient kernels, against torch.autograd of conv2d(groups) -> batch_norm(training) -> SiLU in float64 on the CPU."""
    import torch.nn.functional as F
    from mgdt_yolo_amd.nn.modules import DWConv
    m = seed_state_dict_(DWConv(c1, c2, k, s), 1)
    g = m.conv.groups
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, c1, 11, 9, generator=gen)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    w = m.conv.weight.detach().double().requires_grad_(True)
    ga, be = m.bn.weight.detach().double().requires_grad_(True), m.bn.bias.detach().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    y_ref = F.silu(F.batch_norm(F.conv2d(xr, w, None, s, k // 2, 1, g), None, None, ga, be, True, 0.0, m.bn.eps))
    gy = torch.randn(y_ref.shape, generator=gen)
    if dt == torch.bfloat16:
        gy = gy.bfloat16().float()
    (y_ref * gy.double()).sum().backward()
    m = m.to(DEV).train()
    nh = lambda t: t.to(DEV).to(dt).contiguous(memory_format=torch.channels_last)
    y = m(nh(x))
    tol = 3e-2 if dt == torch.bfloat16 else 2e-4
    rel = lambda a, b: (a.double().cpu() - b.double()).norm().item() / b.double().norm().item()
    assert rel(y.float(), y_ref.detach()) < tol
    dx = m.backward(nh(gy))
    assert rel(dx.float(), xr.grad) < tol, rel(dx.float(), xr.grad)
    assert rel(m.conv.weight.grad, w.grad) < tol, rel(m.conv.weight.grad, w.grad)
    assert rel(m.bn.weight.grad, ga.grad) < tol and rel(m.bn.bias.grad, be.grad) < tol
    # accumulate form of the data gradient (a fan-in sum): dx_out holds an addend
    m(nh(x))
    base = torch.randn(dx.shape, generator=gen).to(DEV).to(dt).contiguous(memory_format=torch.channels_last)
    dx2 = m.backward(nh(gy), dx_out=base.clone(memory_format=torch.channels_last), acc=True)
    assert rel(dx2.float(), xr.grad + base.double().cpu()) < tol


# ------------------------------------------------------------------------------------------------ conv kernel sweep
CONV_CASES = [  # cin, cout, k, s, h, w  (+ channel-sliced / fused variants below)
    (8, 8, 1, 1, 20, 24), (8, 8, 3, 1, 17, 13), (16, 32, 3, 2, 33, 29), (32, 64, 3, 2, 20, 20), (64, 128, 3, 2, 12, 12),
    (128, 256, 3, 2, 10, 10), (64, 16, 3, 1, 20, 20), (64, 80, 3, 1, 16, 16), (80, 80, 3, 1, 16, 16), (80, 64, 1, 1, 9, 9),
    (160, 128, 1, 1, 10, 10), (512, 256, 1, 1, 5, 5), (480, 96, 1, 1, 10, 10), (96, 384, 1, 1, 10, 6), (384, 96, 1, 1, 10, 6),
    (192, 64, 1, 1, 12, 12), (32, 256, 1, 1, 7, 7), (256, 256, 1, 1, 6, 5), (24, 40, 3, 1, 11, 7), (16, 48, 1, 1, 5, 5),
    (512, 16, 3, 1, 6, 6), (544, 32, 3, 2, 9, 7),   # K panel larger than LDS even for one cout block -> segmented-panel kernel
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_igemm_vs_torch_cpu(case, dtype):
    """mgdt_conv2d_fwd against F.conv2d (CPU fp32/fp64) incl. sliced views, pre-add, input affine, two residuals."""
    import torch.nn.functional as F
    from mgdt_yolo_amd import ops
    cin, cout, k, s, h, w = case
    r = np.random.default_rng([cin, cout, k, s])
    B = 3
    wt = torch.from_numpy(r.standard_normal((cout, cin, k, k)).astype(np.float32) / np.sqrt(cin * k * k))
    bias = torch.from_numpy(r.standard_normal(cout).astype(np.float32))
    xw = torch.from_numpy(r.standard_normal((B, cin + 16, h, w)).astype(np.float32))   # wider tensor: use a channel slice
    x2 = torch.from_numpy(r.standard_normal((B, cin, h, w)).astype(np.float32))
    sc = torch.from_numpy(r.uniform(0.5, 1.5, (B, cin)).astype(np.float32))
    sh = torch.from_numpy(r.standard_normal(cin).astype(np.float32) * 0.1)
    ho, wo = ops.conv_out_hw(h, w, k, s)
    r1 = torch.from_numpy(r.standard_normal((B, cout, ho, wo)).astype(np.float32))
    r2 = torch.from_numpy(r.standard_normal((B, cout, ho, wo)).astype(np.float32))
    cl = lambda t: t.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    dq = lambda t: t.to(dtype).float()      # what the kernel sees after the cast
    xd, x2d, r1d, r2d = cl(xw), cl(x2), cl(r1), cl(r2)
    pk = ops.PackedConv(wt.to(DEV), bias.to(DEV), None, k, dtype)
    outw = torch.zeros(B, cout + 8, ho, wo, dtype=dtype, device=DEV).contiguous(memory_format=torch.channels_last)
    for variant in ('plain', 'fused'):
        xs = xd[:, 8:8 + cin]
        xin = dq(xw[:, 8:8 + cin])
        if variant == 'plain':
            ops.conv2d(xs, pk, s, ops.ACT_SILU, out=outw[:, 4:4 + cout])
            ref = F.silu(F.conv2d(xin.double(), dq(wt).double(), bias.double(), s, k // 2))
        else:
            ops.conv2d(xs, pk, s, ops.ACT_RELU, out=outw[:, 4:4 + cout], x2=x2d, r1=r1d, r2=r2d, in_scale=sc.to(DEV), in_shift=sh.to(DEV))
            a = (xin + dq(x2))
            if dtype == torch.bfloat16:
                a = a.to(dtype).float()
            a = a * sc[:, :, None, None] + sh[None, :, None, None]
            if dtype == torch.bfloat16:
                a = a.to(dtype).float()
            ref = F.relu(F.conv2d(a.double(), dq(wt).double(), bias.double(), s, k // 2)) + dq(r1).double() + dq(r2).double()
        got = outw[:, 4:4 + cout].float().cpu().double()
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err < tol, (variant, err)
        assert outw[:, :4].abs().max().item() == 0 and outw[:, 4 + cout:].abs().max().item() == 0   # slice borders untouched


# ------------------------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_e2e_fp32_matches_reference(golden, tag):
    """Box xywh within 1e-3 px / conf within 1e-3 of the reference CPU forward on identical inputs (north_star)."""
    g = golden('e2e_' + tag)
    m = build_model(GI.E2E_MODELS[tag])
    assert m.stride.tolist() == g['stride'].tolist()
    for (b, h, w) in GI.E2E_SHAPES:
        key = f'{b}x{h}x{w}'
        x = seeded_images(b, h, w, seed=GI.IMG_SEED).to(DEV)
        with torch.no_grad():
            y, feats = m(x)
        y = y.cpu().numpy()
        if f'y_{key}' in g:
            ref = g[f'y_{key}']
            np.testing.assert_allclose(y[:, :4], ref[:, :4], atol=1e-3, rtol=0)
            np.testing.assert_allclose(y[:, 4:], ref[:, 4:], atol=1e-4, rtol=0)
            for i, f in enumerate(feats):
                np.testing.assert_allclose(to_nchw(f), g[f'feat{i}_{key}'], atol=1e-3, rtol=1e-4)
        else:
            ref = g[f'ysub_{key}']
            np.testing.assert_allclose(y[:, :4, ::25], ref[:, :4], atol=1e-3, rtol=0)
            np.testing.assert_allclose(y[:, 4:, ::25], ref[:, 4:], atol=1e-4, rtol=0)


def test_e2e_layerwise_fp32(golden):
    """Per-layer strided samples of every top-level layer (locates a drifting layer)."""
    tag = 'mspa_c2f_gd_n'
    g = golden('e2e_' + tag)
    m = build_model(GI.E2E_MODELS[tag])
    outs = []
    hooks = [l.register_forward_hook(lambda mod, i, o: outs.append(o)) for l in m.model]
    with torch.no_grad():
        m(seeded_images(2, 160, 160, seed=GI.IMG_SEED).to(DEV))
    for hk in hooks:
        hk.remove()
    for i, o in enumerate(outs[:-1]):
        f = o.float().contiguous().cpu().reshape(-1).double()   # NCHW order like the reference's sample()
        step = max(1, f.numel() // 2048)
        np.testing.assert_allclose(f[::step][:2048].float().numpy(), g[f'L{i}_s_2x160x160'], atol=3e-4, rtol=3e-4, err_msg=f'layer {i}')


def test_e2e_fused_model_same_output():
    """model.fuse() (BN folded into conv parameters, tasks.py:121-146) gives the same result as pack-time folding."""
    m = build_model('mspa_c2f_gd_yolov8')
    x = seeded_images(2, 160, 160, seed=GI.IMG_SEED).to(DEV)
    with torch.no_grad():
        y0, _ = m(x)
        y1, _ = m.fuse()(x)
    assert m.is_fused()
    np.testing.assert_allclose(y1.cpu().numpy(), y0.cpu().numpy(), atol=2e-4, rtol=0)


def test_fused_bf16_model_takes_the_same_block_kernels():
    """after `model.fuse()` (what AutoBackend(fuse=True) and the predictor run, tasks.py:121-146: `bn` deleted, folded bias on the conv) the
    bf16 model must take the SAME launches as the unfused one - fused stem, CSP / ConvNeXt block kernels, injection + cv1, neck plan, Detect tail -
    and give the same output: folding at pack time and folding in fuse() are the same fp32 arithmetic before the bf16 rounding of the panels."""
    from mgdt_yolo_amd import ops
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    x = seeded_images(2, 320, 320, seed=GI.IMG_SEED).to(DEV).to(torch.bfloat16)

    def run(model):
        names = []
        orig = ops._launch
        with torch.no_grad():
            model(x)                                             # panels are packed on first use
            ops._launch = lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1]
            try:
                y, _ = model(x)
            finally:
                ops._launch = orig
        return y.float(), names
    y0, n0 = run(m)
    y1, n1 = run(m.fuse())
    assert m.is_fused() and not hasattr(m.model[0], 'bn')
    assert n1 == n0, (len(n0), len(n1), sorted(set(n1) - set(n0)))
    assert 'stem2_fwd' in n1 and n1.count('csp_block_fwd') == 5 and n1.count('cnx_block_fwd') == 3 and 'conv1x1_inject_conv_fwd' in n1 and 'detect_tail_fwd' in n1
    err = (y1 - y0).abs()
    print(f'fused vs unfused bf16 model: max |dy| boxes {err[:, :4].max().item():.3e} px, scores {err[:, 4:].max().item():.3e}; {len(n1)} launches')
    assert err[:, :4].max().item() < 0.5 and err[:, 4:].max().item() < 2e-2


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_e2e_bf16_stated_tolerance(golden, tag):
    """bf16 throughput path: boxes within 1.5 px, conf within 0.05 of the fp32 reference (stated, not hidden)."""
    g = golden('e2e_' + tag)
    m = build_model(GI.E2E_MODELS[tag], torch.bfloat16)
    x = seeded_images(2, 160, 160, seed=GI.IMG_SEED).to(DEV)
    with torch.no_grad():
        y, _ = m(x)
    ref = g['y_2x160x160']
    y = y.cpu().numpy()
    eb, ec = np.abs(y[:, :4] - ref[:, :4]).max(), np.abs(y[:, 4:] - ref[:, 4:]).max()
    print(f'bf16 {tag}: max box err {eb:.4f} px, max conf err {ec:.4f}')
    assert eb < BF16_TOL[tag][0] and ec < BF16_TOL[tag][1], (eb, ec)


# bf16 throughput path vs the fp32 REFERENCE: stated tolerances = ~2x the errors measured on MI355X (printed by the tests; round 2:
# see DESIGN.md section 4).  (max box error in px, max confidence error) over every anchor / class of the compared outputs.
# measured (round 2, gpurun_out/r2_*): 2x160x160: mspa 0.68 px / 0.034, yolov8 0.23 px / 0.0031; 1x640x640: mspa 0.83 px / 0.048 (mean box error
# 0.14 px), yolov8 0.53 px / 0.0064.  The MSPA-GD graph is deeper in bf16 ops (attention scaling, ConvNeXt blocks, injection) than stock yolov8.
BF16_TOL = {'mspa_c2f_gd_n': (1.4, 0.07), 'yolov8_n': (0.5, 0.008)}
BF16_TOL_640 = {'mspa_c2f_gd_n': (1.7, 0.10), 'yolov8_n': (1.1, 0.015)}


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_e2e_bf16_at_headline_size_vs_reference(golden, tag):
    """The headline dtype at the headline size: bf16 forward of 1x640x640 against the reference's fp32 output (fixture ysub_1x640x640,
    every 25th anchor of 6400 / 8400, all 84 rows)."""
    g = golden('e2e_' + tag)
    m = build_model(GI.E2E_MODELS[tag], torch.bfloat16)
    x = seeded_images(1, 640, 640, seed=GI.IMG_SEED).to(DEV)
    with torch.no_grad():
        y, _ = m(x.to(torch.bfloat16))
    ref = g['ysub_1x640x640']
    y = y.cpu().numpy()[:, :, ::25]
    eb, ec = np.abs(y[:, :4] - ref[:, :4]).max(), np.abs(y[:, 4:] - ref[:, 4:]).max()
    print(f'bf16 640 {tag}: max box err {eb:.4f} px, max conf err {ec:.4f}; mean box err {np.abs(y[:, :4] - ref[:, :4]).mean():.4f}')
    assert eb < BF16_TOL_640[tag][0] and ec < BF16_TOL_640[tag][1], (eb, ec)


def test_mixed_dtype_operands_are_refused_before_any_launch():
    """Regression test of the round-2 GPU fault (commit 44e5950: a bf16 map addressed with the fp32 dtype code = out-of-range access) and of
    its round-3 sibling (a bf16 NCHW image handed to the generic weight-gradient kernel, which reads fp32): every multi-view entry point
    must raise on the host, nothing may be launched."""
    from mgdt_yolo_amd import ops
    w = torch.randn(16, 16, 1, 1, device=DEV)
    pk = ops.PackedConv(w, None, None, 1, torch.bfloat16)
    xb = torch.randn(2, 16, 8, 8, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xf = xb.float().contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match='dtype|packed for'):
        ops.conv2d(xf, pk, 1, ops.ACT_NONE)                               # fp32 input, bf16 panel
    with pytest.raises(RuntimeError, match='share a dtype'):
        ops.conv2d(xb, pk, 1, ops.ACT_NONE, r1=xf)                        # bf16 conv with an fp32 residual view
    with pytest.raises(RuntimeError, match='share a dtype'):
        ops.conv2d(xb, pk, 1, ops.ACT_NONE, out=torch.empty_like(xf))     # bf16 conv into an fp32 output view
    dw = torch.zeros(16, 16, 1, 1, device=DEV)
    with pytest.raises(RuntimeError, match='share a dtype'):
        ops.conv_wgrad(xb, xf, 1, 1, dw)                                  # bf16 input, fp32 output gradient
    x5 = torch.randn(2, 5, 8, 8, device=DEV).to(torch.bfloat16)          # NCHW, 5 channels: only the generic kernel could take it, and it reads fp32
    with pytest.raises(RuntimeError, match='fp32 input'):
        ops.conv_wgrad(x5, xb, 1, 1, torch.zeros(16, 5, 1, 1, device=DEV))
    # and the case that faulted: the 3-channel bf16 NCHW image of a bf16 training step now takes the padded NHWC path and is right
    img = torch.rand(2, 3, 16, 16, device=DEV)
    dy = torch.randn(2, 16, 8, 8, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dwi = torch.full((16, 3, 3, 3), float('nan'), device=DEV)
    ops.conv_wgrad(img.to(torch.bfloat16), dy, 3, 2, dwi)
    ref = torch.nn.grad.conv2d_weight(img.to(torch.bfloat16).double(), (16, 3, 3, 3), dy.double().contiguous(), stride=2, padding=1)
    assert ((dwi.double() - ref).norm() / ref.norm()).item() < 1e-4


def test_cpu_tensor_is_refused():
    from mgdt_yolo_amd.nn.modules import Conv
    with pytest.raises(RuntimeError, match='no CPU'):
        Conv(8, 8, 1).eval()(torch.zeros(1, 8, 4, 4))


# ------------------------------------------------------------------------------------------------ NMS
@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_nms_matches_fixture_and_oracle(golden, tag):
    """Kept rows bit-exact vs the reference-generated fixture (same y) and kept anchor/class integers vs the oracle."""
    from mgdt_yolo_amd.yolo.utils.ops import nms_with_index, non_max_suppression
    from oracle import nms as ON
    g, y = golden('nms'), golden('e2e_' + tag)['y_2x160x160']
    yd = torch.from_numpy(y).to(DEV)
    for cname, kw in GI.NMS_CASES:
        out = non_max_suppression(yd, **kw)
        rows, kept = nms_with_index(yd, **kw)
        _, okept = ON.non_max_suppression(y, return_index=True, **kw)
        for i, o in enumerate(out):
            ref = g[f'{tag}_{cname}_{i}']
            assert tuple(o.shape) == ref.shape, (cname, i, o.shape, ref.shape)
            assert np.array_equal(o.cpu().numpy(), ref), (cname, i)
            assert np.array_equal(kept[i].cpu().numpy().astype(np.int64), okept[i][0]), (cname, i)
            assert np.array_equal(rows[i][:, 5].cpu().numpy().astype(np.int64), okept[i][1]), (cname, i)


def test_nms_on_hip_forward_output_640():
    """Full-size case: y from the HIP fp32 forward at 640x640 -> NMS kept integers == oracle NMS on the same y."""
    from mgdt_yolo_amd.yolo.utils.ops import nms_with_index
    from oracle import nms as ON
    m = build_model('mspa_c2f_gd_yolov8')
    with torch.no_grad():
        y, _ = m(seeded_images(2, 640, 640, seed=3).to(DEV))
    for kw in (dict(conf_thres=0.25, iou_thres=0.7), dict(conf_thres=0.001, iou_thres=0.7, multi_label=True)):
        rows, kept = nms_with_index(y, **kw)
        orows, okept = ON.non_max_suppression(y.cpu().numpy(), return_index=True, **kw)
        for i in range(2):
            assert np.array_equal(kept[i].cpu().numpy().astype(np.int64), okept[i][0])
            assert np.array_equal(rows[i].cpu().numpy(), orows[i])
        # idempotence: NMS of the kept boxes alone keeps them all (size-independent property)
    assert all(len(k) > 0 for k in kept)


def test_nms_with_detect_tail_keys_equals_the_score_scan():
    """The bf16 Detect tail leaves the NMS key of every anchor's best class next to y (mgdt_detect_tail_fwd best_keys); non_max_suppression(y)
    then skips its scan over the nc score rows.  Both routes must give identical rows / kept anchors, equal to the oracle on the same y, at
    the bench shape's per-image size (6400 anchors, nc = 80, every anchor a candidate, max_det saturated); the keys are dropped as soon as
    y is written to; multi_label never uses them."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.yolo.utils.ops import nms_with_index
    from oracle import nms as ON
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    with torch.no_grad():
        y, _ = m(seeded_images(3, 640, 640, seed=5).to(DEV).to(torch.bfloat16))
    assert getattr(y, '_mgdt_best', None) is not None and ops._best_keys_of(y, 3, y.shape[2]) is not None
    for kw in (dict(conf_thres=0.25, iou_thres=0.7), dict(conf_thres=0.5, iou_thres=0.45, classes=list(range(0, 80, 2))), dict(conf_thres=0.3, iou_thres=0.6, agnostic=True, max_det=50),
               dict(conf_thres=0.001, iou_thres=0.7, multi_label=True)):
        rows, kept = nms_with_index(y, **kw)
        ops.NMS_USE_BEST_KEYS = False
        try:
            rows2, kept2 = nms_with_index(y, **kw)
        finally:
            ops.NMS_USE_BEST_KEYS = True
        orows, okept = ON.non_max_suppression(y.cpu().numpy(), return_index=True, **kw)
        for i in range(3):
            assert torch.equal(rows[i], rows2[i]) and torch.equal(kept[i], kept2[i]), kw
            assert np.array_equal(kept[i].cpu().numpy().astype(np.int64), okept[i][0]), kw
            assert np.array_equal(rows[i].cpu().numpy(), orows[i]), kw
    y[:, 4:, :10] *= 0.5                                   # an in-place edit: the stored keys no longer describe y
    assert ops._best_keys_of(y, 3, y.shape[2]) is None
    rows, kept = nms_with_index(y, conf_thres=0.25, iou_thres=0.7)
    _, okept = ON.non_max_suppression(y.cpu().numpy(), return_index=True, conf_thres=0.25, iou_thres=0.7)
    assert all(np.array_equal(kept[i].cpu().numpy().astype(np.int64), okept[i][0]) for i in range(3))


def test_nms_candidate_segments_histogram_and_rank_select_fallback():
    """The candidate segments come from a 2048-bin score histogram; when one bin alone overflows a segment (thousands of equal scores) or
    max_nms cuts inside a bin, the exact radix rank select takes over.  Synthetic predictions that force each route, against the oracle:
    (a) 6000 candidates with distinct scores spread over (0.3, 1): histogram segments only; (b) all scores identical (one bin, ties resolved
    by candidate index); (c) max_nms = 700 cutting the candidate list inside a bin; (d) heavily overlapping boxes with a low IoU threshold and
    max_det = 800, so that the first segment does not fill max_det and the later (4096-key) segments run; (e) scores of exactly 1.0 (the
    clamped last bin)."""
    from mgdt_yolo_amd.yolo.utils.ops import nms_with_index
    from oracle import nms as ON
    r = np.random.default_rng(17)
    A, nc = 6000, 8

    def make(scores, spread):
        y = np.zeros((2, 4 + nc, A), np.float32)
        for i in range(2):
            y[i, 0] = r.uniform(20, spread, A); y[i, 1] = r.uniform(20, spread, A)
            y[i, 2] = r.uniform(8, 60, A); y[i, 3] = r.uniform(8, 60, A)
            cls = r.integers(0, nc, A)
            y[i, 4 + cls, np.arange(A)] = scores(i)
        return y
    distinct = lambda i: (0.3 + 0.69 * r.permutation(A) / A).astype(np.float32)
    cases = [('hist', make(distinct, 600.0), dict(conf_thres=0.25, iou_thres=0.5)),
             ('one_bin', make(lambda i: np.full(A, 0.5, np.float32), 600.0), dict(conf_thres=0.25, iou_thres=0.5)),
             ('max_nms_cut', make(lambda i: np.round(distinct(i) * 64) / 64, 600.0), dict(conf_thres=0.25, iou_thres=0.5, max_nms=700)),
             ('many_segments', make(distinct, 400.0), dict(conf_thres=0.25, iou_thres=0.2, max_det=800, agnostic=True)),
             ('ones', make(lambda i: np.where(r.random(A) < 0.5, 1.0, 0.75).astype(np.float32), 2000.0), dict(conf_thres=0.25, iou_thres=0.5))]
    for name, y, kw in cases:
        rows, kept = nms_with_index(torch.from_numpy(y).to(DEV), **kw)
        orows, okept = ON.non_max_suppression(y, return_index=True, **kw)
        for i in range(2):
            assert np.array_equal(kept[i].cpu().numpy().astype(np.int64), okept[i][0]), (name, i, len(kept[i]), len(okept[i][0]))
            assert np.array_equal(rows[i].cpu().numpy(), orows[i]), (name, i)
        print(name, [len(k) for k in kept])


def test_nms_edge_cases():
    from mgdt_yolo_amd.yolo.utils.ops import non_max_suppression
    y = torch.zeros(2, 84, 100, device=DEV)
    assert [tuple(o.shape) for o in non_max_suppression(y)] == [(0, 6), (0, 6)]          # nothing above conf
    with pytest.raises(AssertionError):
        non_max_suppression(y, conf_thres=1.5)
    y[0, :4, :3] = torch.tensor([[10., 10.5, 200.], [10., 10.5, 200.], [20., 20., 20.], [20., 20., 20.]], device=DEV)
    y[0, 4 + 7, :3] = torch.tensor([0.9, 0.8, 0.7], device=DEV)
    out = non_max_suppression(y, conf_thres=0.25, iou_thres=0.45)
    assert out[0].shape == (2, 6) and out[0][:, 5].tolist() == [7.0, 7.0] and out[1].shape == (0, 6)
    assert non_max_suppression(y, classes=[])[0].shape == (0, 6)


# ------------------------------------------------------------------------------------------------ loss + assigner (a17-a19)
class _FakeHead:
    def __init__(self, nc, R, strides):
        self.nc, self.reg_max, self.no, self.stride = nc, R, nc + 4 * R, torch.tensor(strides)


class _FakeModel:
    """Just what v8DetectionLoss reads from a DetectionModel (model[-1], args, parameters())."""

    def __init__(self, nc, R, strides):
        self.model, self.args = [_FakeHead(nc, R, strides)], None

    def parameters(self):
        return iter([torch.zeros(1, device=DEV)])


@pytest.mark.parametrize('seed,calls', GI.LOSS_CASES)
def test_detection_loss_fwd_bwd_matches_reference(golden, seed, calls):
    """loss*B, items [box, cls, dfl] and d loss / d head maps against the reference's v8DetectionLoss + autograd (fixture)."""
    from mgdt_yolo_amd.yolo.utils.loss import v8DetectionLoss
    g = golden('loss')
    B, nc, R, hw = (GI.LOSS_SHAPE[k] for k in ('B', 'nc', 'R', 'hw'))
    feats, lab = GI.loss_inputs(seed, B, nc, R, hw)
    f = feats.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    crit = v8DetectionLoss(_FakeModel(nc, R, [8.0]))
    crit.epoch = calls
    total, items = crit([f], lab)
    total.backward()
    k = f's{seed}'
    np.testing.assert_allclose(total.item(), g[k + '_total'], rtol=2e-5)
    np.testing.assert_allclose(items.cpu().numpy(), g[k + '_items'], rtol=2e-5)
    np.testing.assert_allclose(to_nchw(f.grad), g[k + '_grad'], atol=5e-6, rtol=2e-4)
    assert crit.epoch == calls + 1


@pytest.mark.parametrize('case', [dict(B=4, nc=3, R=4, hw=(20, 20), seed=21, calls=0), dict(B=4, nc=3, R=4, hw=(20, 20), seed=22, calls=161 * 40),
                                  dict(B=8, nc=80, R=4, hw=(40, 40), seed=5, calls=161 * 10), dict(B=3, nc=5, R=4, hw=(24, 36), seed=9, calls=0)])
def test_assigner_integers_match_oracle(case):
    """fg mask / target_gt_idx bit-exact and target scores close to the CPU oracle (itself pinned to the reference fixtures)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.yolo.utils.loss import v8DetectionLoss
    from oracle import loss as OLoss
    B, nc, R, hw = case['B'], case['nc'], case['R'], case['hw']
    feats, lab = GI.loss_inputs(case['seed'], B, nc, R, hw)
    _, items_ref, aux = OLoss.detection_loss([feats], lab, [8.0], R, nc, call_count=case['calls'])
    crit = v8DetectionLoss(_FakeModel(nc, R, [8.0]))
    gt = crit.preprocess(lab, B, (hw[0] * 8.0, hw[1] * 8.0))
    f = feats.to(DEV).contiguous(memory_format=torch.channels_last)
    st = ops.detect_loss_fwd([f], [8.0], R, nc, gt, case['calls'], (7.5, 0.5, 1.5), want_assignment=True)
    fg, gi, ts = st.fg.cpu().numpy().astype(bool), st.gt_idx.cpu().numpy(), st.tscore.cpu().numpy()
    assert np.array_equal(fg, aux['fg_mask'].numpy())
    assert np.array_equal(gi, aux['target_gt_idx'].numpy())
    np.testing.assert_allclose(ts, aux['target_scores'].sum(-1).numpy(), atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(st.out5[1:4].cpu().numpy(), items_ref.numpy(), rtol=5e-5)
    assert fg.sum() > 0


def test_loss_empty_labels_all_background():
    """The reference crashes on an empty-label batch (tal.py:102-108); the build defines all-background targets."""
    from mgdt_yolo_amd.yolo.utils.loss import v8DetectionLoss
    feats, _ = GI.loss_inputs(3, 2, 3, 4, (10, 10))
    lab = {'batch_idx': torch.zeros(0), 'cls': torch.zeros(0, 1), 'bboxes': torch.zeros(0, 4)}
    total, items = v8DetectionLoss(_FakeModel(3, 4, [8.0]))([feats.to(DEV).contiguous(memory_format=torch.channels_last)], lab)
    assert items[0].item() == 0 and items[2].item() == 0 and items[1].item() > 0 and torch.isfinite(total)


def test_model_loss_call_surface():
    """model(batch_dict) -> (loss*B, items[3]) like tasks.py:44-45,204-216 (eval-mode BN; raw maps via head.training)."""
    m = build_model('mspa_c2f_gd_yolov8', nc=4)
    m.model[-1].training = True
    batch = {'img': seeded_images(2, 160, 160, seed=1).to(DEV), **{k: v for k, v in GI.loss_inputs(1, 2, 4, 4, (20, 20))[1].items()}}
    total, items = m(batch)
    assert total.ndim == 0 and items.shape == (3,) and torch.isfinite(total) and (items >= 0).all()


# ------------------------------------------------------------------------------------------------ training step (fwd + bwd)
def _oracle_train_grads(name, nc, x, lab, strides):
    """CPU autograd through the oracle in training mode (BN batch statistics) -> loss, feats, {param name: grad}."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from oracle import layers as OL
    from oracle import loss as OLoss
    cfg = get_config(name, 'n', nc)
    m = seed_state_dict_(DetectionModel(cfg, verbose=False), 0)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k and 'dfl' not in k) for k, v in m.state_dict().items()}
    OL.BN_TRAIN = True
    try:
        feats = OL.model_forward(cfg, sd, x, strides, decode=False)
        total, items, _ = OLoss.detection_loss(feats, lab, strides, 16 if 'tood' in name else 4, nc, call_count=0)
        total.backward()
    finally:
        OL.BN_TRAIN = False
    return total.detach(), [f.detach() for f in feats], {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize('name', ['yolov8', 'mspa_c2f_gd_yolov8', 'mspa_c2f_gd_tood_yolov8'])
def test_train_step_gradients_match_cpu_autograd(name):
    """Train-mode forward (batch-stat BN) + HIP loss + explicit HIP backward vs torch-CPU autograd through the oracle."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.utils.loss import loss_and_head_grads, v8DetectionLoss
    nc, B, S = 4, 2, 64
    x = seeded_images(B, S, S, seed=11)
    lab = seeded_labels(B, nc, seed=4, max_boxes=4, min_boxes=2)
    lab['bboxes'][:, 2:] = lab['bboxes'][:, 2:] * 0.5 + 0.1
    m = seed_state_dict_(DetectionModel(get_config(name, 'n', nc), verbose=False), 0).to(DEV).train()
    strides = [float(s) for s in m.stride.tolist()]
    ref_total, ref_feats, ref_grads = _oracle_train_grads(name, nc, x, lab, strides)
    feats = m(x.to(DEV))
    for f, r in zip(feats, ref_feats):
        np.testing.assert_allclose(to_nchw(f), r.numpy(), atol=2e-4, rtol=2e-4)
    total, items, hg = loss_and_head_grads(v8DetectionLoss(m), feats, lab)
    np.testing.assert_allclose(total.item(), ref_total.item(), rtol=1e-4)
    m.backward(hg)
    worst = 0.0
    for k, p in m.named_parameters():
        if k not in ref_grads:
            continue
        assert p.grad is not None, f'no gradient for {k}'
        g, r = p.grad.detach().cpu().double(), ref_grads[k].double()
        err = (g - r).norm().item() / max(r.norm().item(), 1e-6 * r.numel() ** 0.5)
        worst = max(worst, err)
        assert err < 2e-2, (k, err, r.norm().item())
    print('worst relative grad error', worst)
    # running statistics were updated with momentum 0.03 (biased batch mean, unbiased variance)
    bn0 = m.model[0].bn
    assert not torch.allclose(bn0.running_mean.cpu(), seed_state_dict_(DetectionModel(get_config(name, 'n', nc), verbose=False), 0).model[0].bn.running_mean)


TRAIN_MODULE_CASES = ['conv3s1', 'c2f', 'c2f_sc', 'sppf', 'mspa_n1', 'mspa_n2_odd', 'mspa_n2_nosc', 'fam4', 'fam4_odd', 'laf3', 'laf3_allconv', 'ifm',
                      'inject_up', 'inject_up_odd', 'inject_pool']


@pytest.mark.parametrize('name', TRAIN_MODULE_CASES)
def test_module_backward_matches_cpu_autograd(name):
    """Training-mode forward + explicit HIP backward of one registry module vs torch-CPU autograd through the oracle
    (BN batch statistics): output, input gradients and every parameter gradient."""
    from oracle import layers as OL
    cls, args, _ = GI.MODULE_CASES[name]
    m = _make_module(cls, args).train()
    xs_cpu = [x.clone().requires_grad_(True) for x in GI.module_inputs(name)]
    sd = {'m.' + k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in m.state_dict().items()}
    from test_oracle_golden import _oracle_module
    OL.BN_TRAIN = True
    try:
        y_ref = _oracle_module(name, cls, args, sd, xs_cpu)
        gy = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(y_ref.shape)).astype(np.float32))
        (y_ref * gy).sum().backward()
    finally:
        OL.BN_TRAIN = False
    xs = [x.detach().to(DEV).contiguous(memory_format=torch.channels_last) for x in xs_cpu]
    y = m(xs[0] if len(xs) == 1 else xs)
    np.testing.assert_allclose(to_nchw(y), y_ref.detach().numpy(), atol=2e-4, rtol=2e-4)
    gin = m.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    gin = gin if isinstance(gin, (list, tuple)) else [gin]
    rel = lambda a, b: (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-7 * b.numel() ** 0.5)
    for i, (g, xr) in enumerate(zip(gin, xs_cpu)):
        assert rel(g.float().cpu().contiguous(), xr.grad) < 2e-3, (name, 'input', i, rel(g.float().cpu().contiguous(), xr.grad))
    refs = {k: sd['m.' + k].grad for k, _ in m.named_parameters() if sd['m.' + k].grad is not None}
    gscale = float(np.median([r.abs().max().item() for r in refs.values()])) if refs else 1.0
    for k, p in m.named_parameters():
        if k not in refs:
            continue
        r = refs[k]
        assert p.grad is not None, (name, k)
        # gradients that are analytically zero (e.g. a bias in front of a batch-norm) are pure rounding noise on both sides:
        # the error is measured against max(|r|, 1% of the module's typical gradient magnitude)
        err = (p.grad.cpu().double() - r.double()).abs().max().item() / max(r.abs().max().item(), 1e-2 * gscale)
        assert err < 5e-3, (name, k, err)


def test_optimizer_step_matches_torch_sgd_and_loss_decreases():
    """clip(10) + SGD(nesterov, wd groups) + EMA on the flat buffer vs torch.optim.SGD on the same gradients; then a few steps."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 4, 4, 64
    m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
    tr = DetectionTrainer(m, lr0=0.01)
    batch = dict(img=(seeded_images(B, S, S, seed=2) * 255).to(torch.uint8), **seeded_labels(B, nc, seed=6, max_boxes=4, min_boxes=2))
    batch['bboxes'][:, 2:] = batch['bboxes'][:, 2:] * 0.5 + 0.1
    trainable = [(k, p) for k, p in m.named_parameters() if p.requires_grad]       # dfl.conv.weight is frozen (block.py:46)
    before = {k: p.detach().cpu().clone() for k, p in trainable}
    l0, _ = tr.step(batch)
    grads = {k: p.grad.detach().cpu().clone() for k, p in trainable}
    # reference update on the CPU with torch's own SGD + clip_grad_norm_
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in before.items()}
    for k, p in ref.items():
        p.grad = grads[k].clone()
    torch.nn.utils.clip_grad_norm_(list(ref.values()), 10.0)
    from mgdt_yolo_amd.yolo.engine.trainer import param_groups
    grp = param_groups(m)                      # the reference's build_optimizer rule (pinned by tests/golden/optim_groups.npz on the CPU side)
    decay = [p for k, p in ref.items() if grp[k] == 0]
    nodecay = [p for k, p in ref.items() if grp[k] != 0]
    assert any('grn.gamma' in k and grp[k] == 0 for k in ref) and any(k.endswith('norm.weight') and grp[k] == 0 for k in ref)
    opt = torch.optim.SGD([{'params': decay, 'weight_decay': 5e-4}, {'params': nodecay, 'weight_decay': 0.0}], lr=0.01, momentum=0.937, nesterov=True)
    opt.step()
    for k, p in trainable:
        np.testing.assert_allclose(p.detach().cpu().numpy(), ref[k].detach().numpy(), atol=1e-6, rtol=1e-5, err_msg=k)
    losses = [l0.item()] + [tr.step(batch)[0].item() for _ in range(8)]
    print('losses', [round(v, 2) for v in losses])
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    assert tr.state.steps == 9


def test_ema_follows_the_reference_decay_schedule():
    """ModelEMA.update (torch_utils.py:342-361): after update t, ema = d*ema + (1-d)*model with d = 0.9999*(1 - exp(-t/2000)), over every
    float entry of the state_dict (parameters and BN running statistics)."""
    import math
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 4, 2, 64
    m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
    tr = DetectionTrainer(m, lr0=0.02)
    batch = dict(img=(seeded_images(B, S, S, seed=2) * 255).to(torch.uint8), **seeded_labels(B, nc, seed=6, max_boxes=4, min_boxes=2))
    ema = tr.state.ema.cpu().double()
    for t in range(1, 4):
        tr.step(batch)
        d = 0.9999 * (1 - math.exp(-t / 2000))
        ema = (ema.float() * np.float32(d) + np.float32(1 - d) * tr.state.data.cpu()).double()        # fp32 arithmetic like v *= d; v += (1-d)*p
        np.testing.assert_allclose(tr.state.ema.cpu().numpy(), ema.float().numpy(), rtol=2e-6, atol=1e-9)
    assert not torch.equal(tr.state.ema, tr.state.data)
    # the running statistics are part of it (ModelEMA walks state_dict(), not parameters())
    assert tr.state.n_total > tr.state.n_param and not torch.equal(tr.state.ema[tr.state.n_param:], tr.state.data[tr.state.n_param:])


def test_reference_training_call_sequence_reaches_the_hip_backward():
    """`loss, items = model(batch); loss.backward()` (yolo/engine/trainer.py:334-343, nn/tasks.py:204-216): the reference trainer's own call
    sequence fills every parameter gradient, equal bit for bit to the explicit model.backward(head_grads) path and within the usual bound
    of CPU autograd through the oracle."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.utils.loss import loss_and_head_grads, v8DetectionLoss
    name, nc, B, S = 'mspa_c2f_gd_yolov8', 4, 2, 64
    x = seeded_images(B, S, S, seed=11)
    lab = seeded_labels(B, nc, seed=4, max_boxes=4, min_boxes=2)
    lab['bboxes'][:, 2:] = lab['bboxes'][:, 2:] * 0.5 + 0.1
    mk = lambda: seed_state_dict_(DetectionModel(get_config(name, 'n', nc), verbose=False), 0).to(DEV).train()
    # (a) the reference's call form
    m = mk()
    batch = dict(img=x.to(DEV), **lab)
    loss, items = m(batch)
    assert loss.requires_grad and items.shape == (3,)
    loss.backward()
    # (b) the explicit path on a fresh copy of the same model
    m2 = mk()
    feats = m2._predict_once(x.to(DEV))
    total, _, hg = loss_and_head_grads(v8DetectionLoss(m2), feats, lab)
    m2.backward(hg)
    assert torch.equal(loss.detach(), total)
    n = 0
    for (k, p), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        if p2.grad is None:
            continue
        assert p.grad is not None and torch.equal(p.grad, p2.grad), k
        n += 1
    assert n > 150
    # (c) CPU autograd through the oracle
    strides = [float(s) for s in m.stride.tolist()]
    _, _, ref_grads = _oracle_train_grads(name, nc, x, lab, strides)
    for k, p in m.named_parameters():
        if k in ref_grads:
            g, r = p.grad.detach().cpu().double(), ref_grads[k].double()
            assert (g - r).norm().item() / max(r.norm().item(), 1e-6 * r.numel() ** 0.5) < 2e-2, k
    # every module dropped its saved activations; a forward without backward leaves nothing behind after the next forward either
    assert all(not mod.__dict__.get('_ctx') for mod in m.modules())
    m(batch); m(batch)[0].backward()
    assert all(not mod.__dict__.get('_ctx') for mod in m.modules())
    with torch.no_grad():
        m(batch)
    assert all(not mod.__dict__.get('_ctx') for mod in m.modules())
    # gradient accumulation (trainer.py:250,345): a second backward adds when asked to
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.grad_accumulate = True
    m(batch)[0].backward()
    for k, p in m.named_parameters():
        if k in g1 and 'bn' not in k:        # (BN batch statistics moved the running stats, not the gradients: same batch, same grads)
            np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * g1[k].cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=k)


def test_packed_weight_caches_follow_the_hip_optimizer():
    """The HIP SGD kernel updates parameters behind torch's back; every packed-weight cache must be rebuilt afterwards: the stepped
    model must compute exactly what a freshly built model with the same state_dict computes."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    cfg = get_config('mspa_c2f_gd_yolov8', 'n', 80)
    m = seed_state_dict_(DetectionModel(cfg, verbose=False), 0).to(DEV)
    tr = DetectionTrainer(m, lr0=0.05)
    batch = seeded_labels(2, 80, seed=1)
    batch['img'] = (seeded_images(2, 96, 96, seed=2) * 255).round().to(torch.uint8)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    x = seeded_images(2, 96, 96, seed=5).to(DEV)
    m.eval()
    with torch.no_grad():
        y0 = m(x)[0].clone()
    m.train()
    for _ in range(2):
        tr.step(batch)
    m.eval()
    fresh = DetectionModel(cfg, verbose=False).to(DEV)
    fresh.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    fresh.eval()
    with torch.no_grad():
        y1, y2 = m(x)[0], fresh(x)[0]
    assert not torch.equal(y0, y1), 'two optimizer steps at lr 0.05 must change the output'
    assert torch.equal(y1, y2)


def test_validator_matching_bit_exact(golden):
    """mgdt_val_match_fwd (one workgroup per image, whole batch in one launch) vs the reference's _process_batch fixtures and, on a larger
    random batch, vs the oracle: the boolean true-positive matrix is compared bit for bit."""
    from mgdt_yolo_amd import ops
    from oracle import val as OV
    g = golden('val_match')
    iouv = torch.from_numpy(g['iouv'])
    cases = [(seed, *GI.val_match_inputs(seed, nd, nl)) for seed, nd, nl in GI.VAL_MATCH_CASES]
    cases += [(100 + k, *GI.val_match_inputs(100 + k, nd, nl)) for k, (nd, nl) in enumerate([(300, 64), (257, 3), (31, 90), (300, 1)])]
    b, md, ml = len(cases), 300, 96
    det = torch.zeros(b, md, 6); lab = torch.zeros(b, ml, 5)
    ndet = torch.zeros(b, dtype=torch.int32); nlab = torch.zeros(b, dtype=torch.int32)
    for i, (_, d, l) in enumerate(cases):
        det[i, :len(d)] = torch.from_numpy(d); lab[i, :len(l)] = torch.from_numpy(l)
        ndet[i], nlab[i] = len(d), len(l)
    correct = ops.val_match(det.to(DEV), ndet.to(DEV), lab.to(DEV), nlab.to(DEV), iouv.to(DEV)).cpu().numpy()
    for i, (seed, d, l) in enumerate(cases):
        ref = g[f'c{seed}'] if f'c{seed}' in g else OV.process_batch(torch.from_numpy(d), torch.from_numpy(l), iouv)
        assert np.array_equal(correct[i, :len(d)], ref), seed
        assert not correct[i, len(d):].any()


def test_validator_surface_matches_reference_call_form(golden):
    """DetectionValidator._process_batch(detections, labels): the reference's per-image call form (val.py:152)."""
    from mgdt_yolo_amd.yolo.v8.detect import DetectionValidator
    g = golden('val_match')
    v = DetectionValidator(DEV)
    for seed, nd, nl in GI.VAL_MATCH_CASES:
        det, lab = GI.val_match_inputs(seed, nd, nl)
        c = v._process_batch(torch.from_numpy(det).to(DEV), torch.from_numpy(lab).to(DEV))
        assert c.dtype == torch.bool and c.device.type == 'cuda' and np.array_equal(c.cpu().numpy(), g[f'c{seed}']), seed


@pytest.mark.parametrize('shape', [(2, 640, 640), (3, 320, 256), (1, 224, 352)])
def test_neck_inputs_delivered_by_their_producers(shape):
    """GD-neck data movement folded into the producers (ops.FUSED_NECK): the MSPA blocks' attention-scaling launches write the avg-pooled
    copies SimFusion_4in / SimFusion_3in need and, for identity branches, their own output straight into the consumer's concat slot - five
    launches (3 avg-pools, 2 copies) fewer per forward.  The pooled values reproduce mgdt_adaptive_avgpool_fwd bit for bit, so the whole
    model output must be IDENTICAL to the separate-launch form; shapes whose maps do not divide by the pooling factors fall back by
    themselves (224 / 4 / 4 is not an integer number of 2x2 bins at the 4x factor's level -> those inputs keep their own launches)."""
    from mgdt_yolo_amd import ops
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    x = seeded_images(shape[0], shape[1], shape[2], seed=9).to(DEV).to(torch.bfloat16)
    launches = {}
    with torch.no_grad():
        m(x)                                                 # weight panels are packed on first use: keep those launches out of the count
        for flag in (True, False):
            ops.FUSED_NECK = flag
            try:
                names = []
                orig = ops._launch
                ops._launch = lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1]
                try:
                    y, feats = m(x)
                finally:
                    ops._launch = orig
                launches[flag] = (y.clone(), [f.clone() for f in feats], names)
            finally:
                ops.FUSED_NECK = True
    (y1, f1, n1), (y0, f0, n0) = launches[True], launches[False]
    assert torch.equal(y1, y0) and all(torch.equal(a, b) for a, b in zip(f1, f0))
    saved = len(n0) - len(n1)
    print(shape, 'launches', len(n0), '->', len(n1))
    if shape[1] % 32 == 0 and shape[2] % 32 == 0:
        assert saved == 5 and n1.count('adaptive_avgpool_fwd') == 0 and n1.count('copy_fwd') == 0, (saved, n1)


@pytest.mark.parametrize('name,dtype', [('mspa_c2f_gd_yolov8', torch.bfloat16), ('mspa_c2f_gd_yolov8', torch.float32), ('mspa_c2f_gd_tood_yolov8', torch.bfloat16)])
def test_side_stream_branch_is_the_same_forward(name, dtype):
    """Layers 12-13 (Conv on P4 + SimFusion_3in: the high-level branch's local input) do not depend on layers 7-11 (end of the backbone,
    SimFusion_4in, IFM): BaseModel._side_branch sends them to a second HIP stream right behind layer 6.  Same kernels, same inputs: the
    outputs must be IDENTICAL to the single-stream forward, eagerly and as a captured graph with a parallel branch (replayed on changing
    input).  Stock yolov8 has no such run of layers and stays on one stream."""
    from mgdt_yolo_amd import ops
    m = build_model(name, dtype)
    assert m._side_branch() == (12, 13, 6) and build_model('yolov8')._side_branch() is None
    xs = [seeded_images(2, 320, 256, seed=s).to(DEV).to(dtype) for s in (3, 4)]
    with torch.no_grad():
        ops.SIDE_STREAM = False
        try:
            ref = [[t.clone() for t in _flat(m(x))] for x in xs]
        finally:
            ops.SIDE_STREAM = True
        streams = []
        orig = ops._launch
        ops._launch = lambda nm, *a, **k: (streams.append(torch.cuda.current_stream().cuda_stream), orig(nm, *a, **k))[1]
        try:
            got = _flat(m(xs[0]))
        finally:
            ops._launch = orig
        assert len(set(streams)) == 2, 'the branch was not launched on a second stream'
        assert all(torch.equal(a, b) for a, b in zip(got, ref[0]))
        # captured: the branch becomes a parallel path of the graph
        xin = xs[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(xin)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = _flat(m(xin))
        for x, r in zip(xs[::-1], ref[::-1]):
            xin.copy_(x)
            g.replay()
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(out, r))


@pytest.mark.parametrize('name', ['mspa_c2f_gd_yolov8', 'yolov8'])
@pytest.mark.parametrize('nc', [4, 36])
def test_detect_tail_class_counts_levels_and_ragged_maps(name, nc):
    """mgdt_detect_tail_fwd (round 3: 16 x 16 staging tiles per MFMA block, best class carried in registers) against the unfused head - class counts
    that are not a multiple of the 16-class MFMA block, one and three levels (anchor offsets), maps whose anchor count is not a multiple of the
    32-anchor unit: decoded boxes / scores within 1e-4 px / 1e-6, raw maps identical, and the NMS keys equal to a scan of y (first maximal class)."""
    from mgdt_yolo_amd import ops
    m = build_model(name, torch.bfloat16, nc=nc)
    for shape in ((2, 320, 256), (1, 224, 352)):
        x = seeded_images(*shape, seed=5).to(DEV).to(torch.bfloat16)
        with torch.no_grad():
            y1, f1 = m(x)
            keys = ops._best_keys_of(y1, y1.shape[0], y1.shape[2])
            ops.FUSED_DETECT_TAIL = False
            try:
                y0, f0 = m(x)
            finally:
                ops.FUSED_DETECT_TAIL = True
        assert keys is not None, 'the fused tail must hand its keys to NMS'
        d = (y1 - y0).abs()
        assert d[:, :4].max().item() < 1e-4 and d[:, 4:].max().item() < 1e-6, (shape, d[:, :4].max().item(), d[:, 4:].max().item())
        assert all(torch.equal(a, b) for a, b in zip(f1, f0))
        best, cls = y1[:, 4:, :].max(1)
        anchors = torch.arange(y1.shape[2], device=y1.device)[None, :]
        ref = ((0xFFFFFFFF - (best.contiguous().view(torch.int32).long() & 0xFFFFFFFF)) << 32) | (anchors * nc + cls)
        assert torch.equal(keys, ref), shape


def test_graph_instances_in_flight_match_their_serial_replays():
    """bench.py --inflight: several captured instances of the inference step (forward + NMS, each with its parallel branch) replayed
    concurrently on their own streams.  Round 2 saw wrong values here (conv_igemm -> bilinear); the cause was packed-fp32 VALU arithmetic
    beside another queue's MFMA kernel (profiles/r03_graph_replay_root_cause.txt), the library is built without it.  Every instance must
    reproduce its serial replay bit for bit, over many overlapped rounds; ops.lane gives each instance its own block-barrier workspace."""
    from mgdt_yolo_amd import ops
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    S = 3
    xs = [seeded_images(4, 320, 320, seed=40 + j).to(DEV).to(torch.bfloat16) for j in range(S)]

    def step(x):
        y, _ = m(x)
        return ops.nms(y, 0.05, 0.7, None, False, False, 300, 30000, 7680) + (y,)

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for j in range(S):
                with ops.lane(j):
                    step(xs[j])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, outs = [], []
        for j in range(S):
            g = torch.cuda.CUDAGraph()
            with ops.lane(j), torch.cuda.graph(g):
                outs.append(step(xs[j]))
            graphs.append(g)
        refs = []
        for j in range(S):
            graphs[j].replay()
            torch.cuda.synchronize()
            refs.append([t.clone() for t in outs[j]])
        assert not torch.equal(refs[0][3], refs[1][3]), 'the instances should see different images'
        lanes = [torch.cuda.Stream() for _ in range(S)]
        for rnd in range(40):
            for j in range(S):
                with torch.cuda.stream(lanes[j]):
                    graphs[j].replay()
            if rnd % 8 == 7:
                torch.cuda.synchronize()
                for j in range(S):
                    assert all(torch.equal(a, b) for a, b in zip(outs[j], refs[j])), (rnd, j)


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    return [t for e in o for t in _flat(e)]


def test_e2e_bf16_fused_kernels_vs_launch_chains_at_bench_shape():
    """The bench configuration (B=8 of the 32, 640x640, bf16) with every fused kernel of this round switched off (conv / GRN / inject launch
    chains) vs on: same model, same input.  Differences are bf16 re-association only; both variants are bf16 forwards, each within the
    stated bf16 tolerance of the fp32 reference (conf 0.05, boxes 1.5 px), so they may differ from each other by up to twice that at the
    worst of 4 M scores; the mean difference is two orders of magnitude smaller."""
    from mgdt_yolo_amd import ops
    m = build_model('mspa_c2f_gd_yolov8', torch.bfloat16)
    x = seeded_images(8, 640, 640, seed=100).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        y_on = m(x)[0].float()
        ops.FUSED_CNX_MLP = ops.FUSED_PW_CHAIN = ops.FUSED_INJECT = ops.FUSED_CSP_BLOCK = ops.FUSED_STEM = ops.FUSED_DETECT_TAIL = False
        try:
            y_off = m(x)[0].float()
        finally:
            ops.FUSED_CNX_MLP = ops.FUSED_PW_CHAIN = ops.FUSED_INJECT = ops.FUSED_CSP_BLOCK = ops.FUSED_STEM = ops.FUSED_DETECT_TAIL = True
    assert not torch.equal(y_on, y_off)                      # the switches really changed the launch sequence
    m32 = build_model('mspa_c2f_gd_yolov8', torch.float32)
    with torch.no_grad():
        y32 = m32(x.float())[0]
    dc, db = (y_on[:, 4:] - y_off[:, 4:]).abs(), (y_on[:, :4] - y_off[:, :4]).abs()
    print(f'fused vs chains: conf max {dc.max().item():.4f} mean {dc.mean().item():.5f}; box max {db.max().item():.3f} px mean {db.mean().item():.4f}')
    for tag, y in (('fused', y_on), ('chains', y_off)):       # each variant against the fp32 forward of the same weights: the stated bf16 bound
        ec, eb = (y[:, 4:] - y32[:, 4:]).abs(), (y[:, :4] - y32[:, :4]).abs()
        print(f'{tag} vs fp32: conf max {ec.max().item():.4f} mean {ec.mean().item():.5f}; box max {eb.max().item():.3f} px mean {eb.mean().item():.4f}')
        assert ec.max().item() < 0.1 and ec.mean().item() < 4e-3 and eb.max().item() < 3.0 and eb.mean().item() < 0.2
    assert dc.max().item() < 0.15 and dc.mean().item() < 5e-3
    assert db.max().item() < 3.0 and db.mean().item() < 0.2


# ------------------------------------------------------------------------------------------------ f1 / f2 / f4: box helpers, AP reduction, predictor, checkpoints
def test_box_helpers_match_reference_fixtures(golden):
    """xywh2xyxy / xyxy2xywh / scale_boxes (bit-exact: same fp32 expression order), box_iou / bbox_iou in all four modes (1e-6)."""
    from mgdt_yolo_amd.yolo.utils import metrics as M, ops as O
    g, gb = golden('boxes2'), golden('boxes')
    b1, b2 = torch.from_numpy(gb['b1']).to(DEV), torch.from_numpy(gb['b2']).to(DEV)
    assert np.array_equal(O.xyxy2xywh(b1).cpu().numpy(), g['xyxy2xywh'])
    assert np.array_equal(O.xywh2xyxy(torch.from_numpy(g['xyxy2xywh']).to(DEV)).cpu().numpy(), g['xywh2xyxy'])
    six = torch.cat([b1, torch.arange(1024, dtype=torch.float32, device=DEV).view(512, 2)], 1)          # extra columns are carried along
    assert torch.equal(O.xyxy2xywh(six)[:, 4:], six[:, 4:])
    np.testing.assert_allclose(M.box_iou(b1[:64], b2[:96]).cpu().numpy(), gb['box_iou'], atol=1e-7, rtol=0)
    w1, w2 = O.xyxy2xywh(b1), O.xyxy2xywh(b2)
    for name, kw in (('iou', {}), ('giou', dict(GIoU=True)), ('diou', dict(DIoU=True)), ('ciou', dict(CIoU=True))):
        np.testing.assert_allclose(M.bbox_iou(b1, b2, xywh=False, **kw).cpu().numpy(), g[name + '_xyxy'], atol=2e-6, rtol=0, err_msg=name)
        np.testing.assert_allclose(M.bbox_iou(w1, w2, xywh=True, **kw).cpu().numpy(), g[name + '_xywh'], atol=2e-6, rtol=0, err_msg=name)
    np.testing.assert_allclose(M.bbox_iou(b1[:1], b2, xywh=False, CIoU=True).cpu().numpy(), g['one_vs_many'], atol=2e-6, rtol=0)
    for k, (s1, s0, rp) in enumerate(GI.SCALE_BOX_CASES):
        pred = torch.cat([torch.from_numpy(GI.scale_box_inputs(k)), torch.rand(64, 2)], 1).to(DEV)        # (n, 6) rows like NMS output
        keep = pred[:, 4:].clone()
        O.scale_boxes(s1, pred, s0, ratio_pad=rp)
        assert np.array_equal(pred[:, :4].cpu().numpy(), g[f'scale{k}']) and torch.equal(pred[:, 4:], keep), k
    assert O.scale_boxes((640, 640), torch.zeros(0, 6, device=DEV), (480, 640)).shape == (0, 6)


@pytest.mark.parametrize('seed,nd,nl,nc', GI.AP_CASES)
def test_ap_per_class_on_device_matches_reference(golden, seed, nd, nl, nc):
    """metrics.py:410-497 with the per-class curves, envelope, 101-point interpolation and integration on the device in numpy's arithmetic order:
    the AP matrix, P / R / F1 at the best-F1 confidence and the TP / FP counts equal the reference's outputs bit for bit."""
    from mgdt_yolo_amd.yolo.utils.metrics import ap_per_class
    g = golden('metrics_ap')
    tp, conf, pcls, tcls = GI.ap_inputs(seed, nd, nl, nc)
    out = ap_per_class(torch.from_numpy(tp).to(DEV), torch.from_numpy(conf).to(DEV), torch.from_numpy(pcls).to(DEV), torch.from_numpy(tcls).to(DEV))
    for name, v in zip(('tp', 'fp', 'p', 'r', 'f1', 'ap', 'cls'), out):
        ref = g[f's{seed}_{name}']
        assert np.asarray(v).shape == ref.shape, name
        assert np.array_equal(np.asarray(v), ref), (name, np.abs(np.asarray(v, np.float64) - ref).max())


def test_validator_update_metrics_and_stats(golden):
    """DetectionValidator.update_metrics + get_stats (val.py:73-131) on a synthetic batch: detections = jittered labels, so the device pipeline
    (scale_boxes -> xywh2xyxy -> _process_batch -> ap_per_class) must agree with the oracle's numpy pipeline on the same numbers."""
    from mgdt_yolo_amd.yolo.v8.detect import DetectionValidator
    from oracle import metrics as OM, val as OV
    v = DetectionValidator(DEV)
    v.init_metrics(nc=5)
    B, H, W = 3, 384, 640
    r = np.random.default_rng(5)
    preds, stats_ref = [], []
    cls_l, box_l, idx_l = [], [], []
    ori = [(720, 1200), (384, 640), (500, 700)]
    rp = []
    for si in range(B):
        gain = min(H / ori[si][0], W / ori[si][1])
        pad = ((W - ori[si][1] * gain) / 2, (H - ori[si][0] * gain) / 2)
        rp.append(((gain, gain), pad))
        det, lab = GI.val_match_inputs(40 + si, 60, 9)
        det[:, [0, 2]] = det[:, [0, 2]].clip(0, W - 1); det[:, [1, 3]] = det[:, [1, 3]].clip(0, H - 1)
        lab[:, [1, 3]] = lab[:, [1, 3]].clip(1, W - 2); lab[:, [2, 4]] = lab[:, [2, 4]].clip(1, H - 2)
        preds.append(torch.from_numpy(det).to(DEV))
        xywh = np.stack([(lab[:, 1] + lab[:, 3]) / 2 / W, (lab[:, 2] + lab[:, 4]) / 2 / H, (lab[:, 3] - lab[:, 1]) / W, (lab[:, 4] - lab[:, 2]) / H], 1).astype(np.float32)
        cls_l.append(lab[:, :1]); box_l.append(xywh); idx_l.append(np.full(len(lab), si, np.float32))
        # oracle pipeline
        predn = OM.scale_boxes((H, W), det[:, :4], ori[si], rp[-1])
        tb = np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], 1).astype(np.float32)
        tb = OM.scale_boxes((H, W), tb * np.array([W, H, W, H], np.float32), ori[si], rp[-1])
        correct = OV.process_batch(torch.from_numpy(np.concatenate([predn, det[:, 4:]], 1)), torch.from_numpy(np.concatenate([lab[:, :1], tb], 1)), torch.linspace(0.5, 0.95, 10))
        stats_ref.append((correct, det[:, 4], det[:, 5], lab[:, 0]))
    batch = dict(img=torch.zeros(B, 3, H, W, dtype=torch.uint8, device=DEV), cls=torch.from_numpy(np.concatenate(cls_l)), bboxes=torch.from_numpy(np.concatenate(box_l)),
                 batch_idx=torch.from_numpy(np.concatenate(idx_l)), ori_shape=ori, ratio_pad=rp)
    v.update_metrics(preds, batch)
    res = v.get_stats()
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats_ref)]
    _, _, p, rr, _, ap, _ = OM.ap_per_class(tp, conf, pcls, tcls)
    assert v.seen == B
    assert res['metrics/mAP50(B)'] == float(ap[:, 0].mean()) and res['metrics/mAP50-95(B)'] == float(ap.mean())
    assert res['metrics/precision(B)'] == float(p.mean()) and res['metrics/recall(B)'] == float(rr.mean())
    assert 0 < res['metrics/mAP50-95(B)'] < 1


def test_letterbox_on_device(golden):
    """LetterBox + BGR->RGB + HWC->CHW in one kernel: geometry pinned by the reference fixture; pixels exact where no resize happens (copy +
    114 border); the resize branch follows the oracle's restatement of cv2's 8-bit INTER_LINEAR (cv2 absent: parity unpinned)."""
    from mgdt_yolo_amd.yolo.data.augment import LetterBox
    from oracle import metrics as OM
    g = golden('letterbox')
    r = np.random.default_rng(1)
    for k, (shape, new_shape, auto) in enumerate(GI.LETTERBOX_CASES):
        img = r.integers(0, 256, (*shape, 3), dtype=np.uint8)
        lb = LetterBox(new_shape, auto=auto, stride=32)
        assert list(lb.geometry(shape)[:6]) == g[f'c{k}'].tolist()[:6]
        out = lb(image=torch.from_numpy(img).to(DEV)).cpu().numpy()
        ref = OM.letterbox(img, new_shape, auto)
        assert out.shape == ref.shape == (3, int(g[f'c{k}'][0]), int(g[f'c{k}'][1]))
        assert np.array_equal(out, ref), (k, np.abs(out.astype(int) - ref.astype(int)).max())


def test_predictor_pipeline_end_to_end():
    """DetectionPredictor: list of BGR uint8 images -> LetterBox on device -> uint8 batch into the model (/255 in the stem) -> NMS -> boxes in the
    ORIGINAL image coordinates.  Checked against the oracle pipeline fed with the same letter-boxed pixels (fp32 exact path)."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.yolo.v8.detect import DetectionPredictor
    from oracle import layers as OL, metrics as OM, nms as ON
    cfg = get_config('mspa_c2f_gd_yolov8', 'n', 80)
    m = seed_state_dict_(DetectionModel(cfg, verbose=False), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    r = np.random.default_rng(3)
    imgs = [r.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(2)]
    pr = DetectionPredictor(dict(imgsz=160, conf=0.5, iou=0.6))
    pr.setup_model(m)
    assert pr.model.fp16 is False and pr.model.stride == 32
    res = pr(imgs)
    lb = np.stack([OM.letterbox(im, (160, 160), auto=True, stride=32) for im in imgs])
    x = torch.from_numpy(lb).float() / 255
    with torch.no_grad():
        y_ref, _ = OL.model_forward(cfg, sd, x, [8.0], fused=True)
    rows = ON.non_max_suppression(y_ref.numpy(), conf_thres=0.5, iou_thres=0.6)
    for i in range(2):
        ref = rows[i].copy()
        ref[:, :4] = OM.scale_boxes(lb.shape[2:], ref[:, :4], imgs[i].shape)
        got = res[i].cpu().numpy()
        assert got.shape == ref.shape and got.shape[0] > 0
        np.testing.assert_allclose(got[:, :4], ref[:, :4], atol=2e-3)
        np.testing.assert_allclose(got[:, 4], ref[:, 4], atol=1e-4)
        assert np.array_equal(got[:, 5], ref[:, 5])
    # half=True selects the bf16 path through model.half(); results stay close
    pr16 = DetectionPredictor(dict(imgsz=160, conf=0.5, iou=0.6, half=True))
    pr16.setup_model(seed_state_dict_(DetectionModel(cfg, verbose=False), 0))
    assert pr16.model.fp16 and pr16.model.model.compute_dtype == torch.bfloat16
    assert len(pr16(imgs)) == 2


def test_reference_format_checkpoint_is_read_without_unpickling_code(tmp_path):
    """trainer.py:411-436 format (whole Module objects pickled in fp16) -> attempt_load_one_weight: every class named by the file is stubbed
    (nothing imported or run), the EMA weights come back exactly (fp16-rounded), the rebuilt model computes what the original computes."""
    import copy
    from mgdt_yolo_amd.nn.tasks import DetectionModel, attempt_load_one_weight
    cfg = get_config('mspa_c2f_gd_yolov8', 'n', 4)
    m = seed_state_dict_(DetectionModel(cfg, verbose=False), 3)
    m.names = {0: 'sow', 1: 'piglet', 2: 'x', 3: 'y'}
    ema = copy.deepcopy(m)
    with torch.no_grad():
        for p in ema.parameters():
            p.mul_(1.01)
    ck = {'epoch': 7, 'best_fitness': 0.5, 'model': torch.nn.Module.half(copy.deepcopy(m)), 'ema': torch.nn.Module.half(ema), 'updates': 11, 'optimizer': None,
          'train_args': {'lr0': 0.001}, 'date': 'x', 'version': '8.0.120'}
    path = str(tmp_path / 'last.pt')
    torch.save(ck, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)                   # the safe loader refuses the format ...
    m2, ck2 = attempt_load_one_weight(path, device=DEV)     # ... this reader takes the weights without running anything from it
    assert ck2['epoch'] == 7 and ck2['updates'] == 11 and m2.names == m.names and not m2.ckpt_missing_keys
    assert any('DetectionModel' in s for s in m2.ckpt_stubbed_globals) and all(not s.startswith('torch.nn') or True for s in m2.ckpt_stubbed_globals)
    sd_e, sd2 = ema.state_dict(), m2.state_dict()
    for k, v in sd_e.items():
        assert torch.equal(v.half().float(), sd2[k].cpu()), k
    want = seed_state_dict_(DetectionModel(cfg, verbose=False), 0)
    want.load_state_dict({k: v.half().float() for k, v in sd_e.items()})
    x = seeded_images(1, 96, 96, seed=2).to(DEV)
    with torch.no_grad():
        assert torch.equal(m2(x)[0], want.eval().to(DEV)(x)[0])


def test_sprmodule_standalone_forward(golden):
    """SPRModule.forward as the reference calls it on its own (spr_module.py:20-31): covered by the module fixtures 'spr' / 'spr_odd' too."""
    from mgdt_yolo_amd.nn.modules import SPRModule
    m = seed_state_dict_(SPRModule(16), GI.MODULE_SEED).eval().to(DEV)
    x = GI.module_inputs('spr')[0].to(DEV)                         # NCHW input is accepted like any module input
    with torch.no_grad():
        y = m(x)
    assert y.shape == (2, 16, 1, 1)
    np.testing.assert_allclose(y.cpu().numpy(), golden('modules')['spr'], atol=1e-5, rtol=1e-5)


def test_bf16_training_step_tracks_the_fp32_step():
    """amp=True (bf16 activations / activation gradients, fp32 master weights, fp32 accumulation and weight gradients) against the fp32 step on
    the same batch.  Stated tolerance of the reduced-precision TRAINING path (the reference's fp16 AMP states none): loss within 2 %; the flat
    gradient has cosine > 0.97 with the fp32 one and a norm within 20 %; every tensor that carries a measurable share of the gradient keeps
    cosine > 0.7 (measured on MI355X, B=4 at 96^2 with batch-statistics BN on 3x3 maps - the noisiest setting: head 0.98-1.00, neck 0.97,
    backbone 0.93-0.95, worst single BN scale 0.76; the error grows with the depth of the reverse pass, as rounding every activation
    gradient to 8 mantissa bits must)."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 4, 4, 96
    batch = dict(img=(seeded_images(B, S, S, seed=2) * 255).to(torch.uint8), **seeded_labels(B, nc, seed=6, max_boxes=4, min_boxes=2))
    batch['bboxes'][:, 2:] = batch['bboxes'][:, 2:] * 0.5 + 0.1
    res = {}
    for amp in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.0, amp=amp)                      # lr 0: the step leaves the weights alone, the flat gradient stays readable
        loss, _ = tr.step(batch)
        res[amp] = (loss.item(), tr.state.grad.clone(), dict(tr.state.offsets))
    (l32, g32, off), (l16, g16, _) = res[False], res[True]
    print(f'loss fp32 {l32:.4f} bf16 {l16:.4f}; |g| fp32 {g32.norm().item():.4f} bf16 {g16.norm().item():.4f}; cosine {torch.nn.functional.cosine_similarity(g32, g16, 0).item():.5f}')
    assert abs(l16 - l32) < 0.02 * abs(l32)
    assert abs(g16.norm().item() - g32.norm().item()) < 0.2 * g32.norm().item()
    assert torch.nn.functional.cosine_similarity(g32, g16, 0).item() > 0.97
    worst = []
    for k, (o, n) in off.items():
        a, b = g32[o:o + n], g16[o:o + n]
        if a.norm().item() > 1e-3 * g32.norm().item():                  # tensors that carry a measurable share of the gradient
            worst.append((torch.nn.functional.cosine_similarity(a, b, 0).item(), k))
    worst.sort()
    print('lowest per-tensor cosines', worst[:4])
    assert worst[0][0] > 0.7, worst[:4]


WGRAD_BF16_CASES = [  # (B, cin, cout, H, W, k, stride, x2, sliced views)
    (2, 8, 8, 16, 16, 3, 1, False, False), (2, 16, 16, 24, 20, 3, 1, True, False), (3, 32, 32, 20, 20, 3, 1, False, True),
    (2, 80, 80, 16, 24, 3, 1, False, False), (2, 64, 80, 9, 13, 3, 1, False, False), (2, 16, 32, 32, 32, 3, 2, False, False),
    (2, 64, 128, 18, 22, 3, 2, False, True), (2, 384, 96, 16, 16, 1, 1, False, False), (2, 96, 384, 12, 20, 1, 1, False, False),
    (2, 8, 8, 16, 16, 1, 1, False, True), (2, 48, 24, 10, 10, 1, 1, False, False), (1, 128, 64, 8, 8, 3, 1, True, True),
    (4, 16, 16, 12, 12, 3, 1, True, True), (4, 32, 16, 12, 12, 3, 1, False, False), (4, 16, 32, 12, 12, 1, 1, False, False), (4, 32, 32, 6, 6, 3, 2, False, False),
    (2, 480, 96, 8, 8, 1, 1, False, False), (2, 24, 40, 16, 16, 3, 1, False, False),
    (2, 4, 16, 32, 32, 3, 2, False, False), (2, 12, 16, 16, 16, 3, 1, False, False), (2, 20, 24, 12, 12, 1, 1, False, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', WGRAD_BF16_CASES, ids=[f'{c[1]}to{c[2]}_{c[3]}x{c[4]}_k{c[5]}s{c[6]}{"_x2" if c[7] else ""}{"_view" if c[8] else ""}' for c in WGRAD_BF16_CASES])
def test_bf16_mfma_weight_gradient(case):
    """Weight gradient of the bf16 training path (ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16, wgrad_bf16.hip) against torch's fp64
    conv2d weight gradient of the SAME bf16-rounded tensors: the products are exact and the accumulation fp32, so the only difference is
    summation order -> relative error of the whole tensor below 2e-5, every element within 1e-3 of the largest.  Covers channel counts
    below / at / not a multiple of the 16-wide MFMA block, maps that do not fill the 8x8 pixel tiles, stride 2, the fused x + x2 input and
    channel-slice views of larger buffers (the concat layouts of the C2f blocks)."""
    from mgdt_yolo_amd import ops
    B, ci, co, H, W, k, s, with_x2, view = case
    gen = torch.Generator().manual_seed(ci * 7 + co)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1

    def mk(c, h, w):
        t = torch.randn(B, c, h, w, generator=gen).to(DEV).to(torch.bfloat16)
        if not view:
            return t.contiguous(memory_format=torch.channels_last)
        big = torch.zeros(B, c + 16, h, w, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        big[:, 8:8 + c] = t
        return big[:, 8:8 + c]
    x, dy = mk(ci, H, W), mk(co, Ho, Wo)
    x2 = mk(ci, H, W) if with_x2 else None
    dw = torch.full((co, ci, k, k), float('nan'), device=DEV)
    db = torch.full((co,), float('nan'), device=DEV)
    ops.conv_wgrad(x, dy, k, s, dw, dbias=db, x2=x2)
    xin = x if x2 is None else (x.float() + x2.float()).to(torch.bfloat16)           # the forward's rounded pre-add
    ref = torch.nn.grad.conv2d_weight(xin.double().contiguous(), (co, ci, k, k), dy.double().contiguous(), stride=s, padding=k // 2)
    err = (dw.double() - ref).abs()
    assert torch.isfinite(dw).all()
    assert (err.norm() / ref.norm()).item() < 2e-5, (err.norm() / ref.norm()).item()
    assert err.max().item() < 1e-3 * ref.abs().max().item()
    assert torch.allclose(db.double(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('amp,split', [(False, False), (True, False), (True, True)], ids=['f32', 'bf16', 'bf16-two-graphs'])
def test_captured_training_step_equals_the_eager_step(amp, split):
    """DetectionTrainer(graph=True): after the first optimizer step the whole step (forward, assigner + loss, reverse pass, clip + SGD + EMA)
    is one hipGraph replay with lr / bias lr / momentum / EMA decay / the assigner's call counter read from device memory.  Nine steps over
    batches whose label counts fall into TWO label-slot buckets (at most 16 / 17-32 boxes per image -> two captured graphs, replayed in the
    order A, B, A, ...), with one step forced through the eager launches in between and the warm-up changing lr every step, must give the same
    losses, weights and EMA as the eager trainer: bit-equal, every kernel has a fixed summation order.  All graphs share one set of packed
    weight panels (a second capture or an eager step must neither re-pack into new memory nor leave a graph reading stale panels).
    'two-graphs' is the multi-rank form (forward/loss/reverse pass | flat-gradient all-reduce, eager | clip/SGD/EMA/re-pack) forced on one rank."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 4, 4, 96
    batches = []
    for r, (lo, hi) in enumerate([(2, 4), (17, 30), (3, 9)]):
        lab = seeded_labels(B, nc, seed=10 + r, max_boxes=hi, min_boxes=lo)
        lab['bboxes'][:, 2:] = lab['bboxes'][:, 2:] * 0.5 + 0.1
        batches.append(dict(img=(seeded_images(B, S, S, seed=20 + r) * 255).to(torch.uint8), **lab))
    slots = [max(16, -(-int(torch.bincount(b['batch_idx'].long()).max()) // 16) * 16) for b in batches]
    assert sorted(set(slots)) == [16, 32], slots
    res = {}
    for graph in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.01, amp=amp, graph=graph, graph_split=split, batch_size=64, nb=10, epochs=3)       # nbs / batch = 1: accumulate stays 1; warm-up active
        losses, panel_ids = [], None
        for i in range(9):
            if i == 5:
                tr.graph, keep = False, tr.graph        # one step through the eager launches between replays of both graphs
                losses.append(tr.step(batches[i % 3])[0].item())
                tr.graph = keep
            else:
                losses.append(tr.step(batches[i % 3])[0].item())
            if graph and i == 1:
                panel_ids = sorted(id(o) for o in tr._graphs[slots[1]][2])
        res[graph] = (losses, tr.state.data.clone(), tr.state.ema.clone(), tr.state.steps, tr.crit.epoch)
        if graph:
            assert sorted(tr._graphs) == [16, 32], 'both label-slot buckets must have been captured'
            assert all(len(gs) == (2 if split else 1) for gs, _, _ in tr._graphs.values())
            for _, _, panels in tr._graphs.values():
                assert len(panels) > 50 and sorted(id(o) for o in panels) == panel_ids, 'every graph must use the one live panel set'
    (l0, w0, e0, s0, c0), (l1, w1, e1, s1, c1) = res[False], res[True]
    assert s0 == s1 == 9 and c0 == c1 == 9
    assert l0 == l1, (l0, l1)
    assert torch.equal(w0, w1) and torch.equal(e0, e1)


@pytest.mark.gpu
def test_captured_training_step_survives_a_batch_shape_change():
    """A batch of another shape resets the static buffers and drops every captured graph (and their pool); coming back re-captures.  The
    sequence B=4, 4, 2, 2, 4, 4 must equal the eager trainer bit for bit."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, S = 4, 96

    def mk(B, r):
        lab = seeded_labels(B, nc, seed=30 + r, max_boxes=6, min_boxes=2)
        lab['bboxes'][:, 2:] = lab['bboxes'][:, 2:] * 0.5 + 0.1
        return dict(img=(seeded_images(B, S, S, seed=40 + r) * 255).to(torch.uint8), **lab)
    seq = [mk(4, 0), mk(4, 1), mk(2, 2), mk(2, 3), mk(4, 4), mk(4, 5)]
    res = {}
    for graph in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.01, amp=True, graph=graph)
        res[graph] = ([tr.step(b)[0].item() for b in seq], tr.state.data.clone())
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1])


def _hip_train_step(tag, amp, x, lab, nc, seed):
    """One HIP training step at lr 0 -> (loss, items, head maps, {name: grad}, {bn prefix: (running_mean, running_var)})."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.yolo.utils.loss import loss_and_head_grads, v8DetectionLoss
    m = seed_state_dict_(DetectionModel(get_config(GI.E2E_MODELS[tag], 'n', nc), verbose=False), seed).to(DEV).train()
    if amp:
        m.set_compute_dtype(torch.bfloat16)
    feats = m._predict_once(x.to(DEV).to(torch.bfloat16 if amp else torch.float32))
    total, items, hg = loss_and_head_grads(v8DetectionLoss(m), feats, lab)
    m.backward(hg)
    grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    running = {k[:-len('.running_mean')]: (b.detach().cpu().numpy(), dict(m.named_buffers())[k[:-len('running_mean')] + 'running_var'].detach().cpu().numpy())
               for k, b in m.named_buffers() if k.endswith('running_mean')}
    return total, items, [to_nchw(f.float()) for f in feats], grads, running


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_training_step_matches_the_reference_training_fixture(golden, tag):
    """the HIP train-mode forward (batch-statistics BatchNorm), loss and reverse pass against what the REFERENCE's own
    `model.train()(batch)`; `loss.backward()` produced on CPU (tests/golden/train_<tag>.npz, gen_golden.py:train): head maps, loss + items,
    all ~200 parameter gradients (stored whole or as a strided sample + l2 norm) and every BN running_mean / running_var after the forward
    (momentum 0.03, unbiased variance).  fp32: every gradient tensor within 1e-3 of its own rms (rounding-level: both sides sum in fp32 in
    different orders), loss 1e-5, head maps 2e-4, running statistics 1e-5."""
    from test_oracle_golden import check_train_fixture
    c = GI.TRAIN_CASE
    x, lab = GI.train_inputs()
    total, items, feats, grads, running = _hip_train_step(tag, False, x, lab, c['nc'], c['weight_seed'])
    worst = check_train_fixture(golden('train_' + tag), total.item(), items.cpu().numpy(), feats, grads, running,
                                dict(feat=2e-4, loss=2e-5, grad=2e-3, grad_abs=1e-2, run=1e-5))
    print('fp32: worst gradient error in units of the tensor rms', worst)


# stated bounds of the bf16 training step vs the reference's fp32 CPU step = 2x the error measured on MI355X (see the test's docstring)
BF16_TRAIN_BOUNDS = {'mspa_c2f_gd_n': dict(loss=0.003, feat=0.12, cos=0.992, tensor_cos=0.4, run=8.5e-2),
                     'yolov8_n': dict(loss=0.03, feat=0.085, cos=0.979, tensor_cos=0.65, run=1.3e-2)}


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_bf16_training_step_vs_the_reference_training_fixture(golden, tag):
    """The bf16 (amp) training step against the same reference fixture.  Stated bounds = 2x what was measured on MI355X (B=4 at 96^2, i.e.
    batch statistics over 3x3 .. 12x12 maps, the noisiest setting): loss within 0.3 % (measured 0.12 %), head maps within 0.12 absolute on
    logits of O(5) (0.054), cosine of the gradient samples of all ~200 tensors with the reference's > 0.992 (0.9964), every tensor that
    carries a measurable share of the gradient > 0.4 (lowest: the SPR attention fc1 of layer 2 at 0.64, then 0.80), running statistics
    within 8.5 % of max(1, |value|) (4.2 % on the last IFM BatchNorm, whose input is the bf16 residual stream of three ConvNeXt blocks).
    Stock yolov8n (three levels down to a 3x3 map): loss 1.46 % (bound 3 %: the assigner's discrete choices move with bf16 scores), head maps
    0.041, cosine 0.9898, lowest tensor 0.83, running statistics 0.65 % - bounds in BF16_TRAIN_BOUNDS."""
    g = golden('train_' + tag)
    c = GI.TRAIN_CASE
    x, lab = GI.train_inputs()
    total, items, feats, grads, running = _hip_train_step(tag, True, x, lab, c['nc'], c['weight_seed'])
    ferr = max(float(np.abs(f - g[f'feat{i}']).max()) for i, f in enumerate(feats))
    got, ref, per = [], [], []
    for k in str(g['grad_names']).split('\n'):
        a, _ = GI.grad_sample(grads[k])
        b = g['g/' + k].astype(np.float64)
        got.append(a.astype(np.float64)); ref.append(b)
        per.append((float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30)), float(np.linalg.norm(b)), k))
    a, b = np.concatenate(got), np.concatenate(ref)
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    heavy = sorted(p for p in per if p[1] > 1e-3 * np.linalg.norm(b))
    # running statistics: error relative to max(1, |reference value|) (a bf16 map's batch variance carries the map's 2^-9 rounding)
    rel = lambda a, b: float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())
    rerr, rwhere = max((max(rel(mu, g[f'bn/{p}.running_mean']), rel(var, g[f'bn/{p}.running_var'])), p) for p, (mu, var) in running.items())
    print(f'bf16 vs reference: loss {total.item():.4f} vs {float(g["loss"]):.4f}, head maps max err {ferr:.4f}, gradient cosine {cos:.5f}, '
          f'lowest per-tensor {heavy[:3]}, running stats max rel err {rerr:.2e} at {rwhere}')
    bound = BF16_TRAIN_BOUNDS[tag]
    assert abs(total.item() - float(g['loss'])) < bound['loss'] * float(g['loss']), (total.item(), float(g['loss']))
    assert ferr < bound['feat'] and cos > bound['cos'] and heavy[0][0] > bound['tensor_cos'] and rerr < bound['run']


@pytest.mark.gpu
def test_training_step_at_the_configs2_per_gpu_shape():
    """BASELINE configs[2] at its real per-GPU shape: B=32, 640x640, nc=80 (yolo/engine/trainer.py:314-362).  The weight-gradient pixel
    splits, BN reduction splits and the LDS data-gradient path are chosen by shape, so this is the only place they run as the bench runs
    them.  (i) the captured step equals the eager step bit for bit over 3 optimizer steps (bf16, the bench's dtype); (ii) every loss, weight
    and gradient is finite; (iii) the bf16 flat gradient against the fp32 one on the same batch: cosine printed and bounded."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 80, 32, 640
    batches = []
    for r in range(2):
        lab = seeded_labels(B, nc, seed=50 + r, max_boxes=14, min_boxes=1)
        batches.append(dict(img=(seeded_images(B, S, S, seed=60 + r) * 255).to(torch.uint8).to(DEV), **lab))
    res = {}
    for graph in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.01, amp=True, graph=graph, batch_size=256, nb=100, epochs=3)
        losses = [tr.step(batches[i % 2])[0].item() for i in range(4)]            # step 0 is eager in both; 1..3 are replays when graph=True
        res[graph] = (losses, tr.state.data.clone(), tr.state.ema.clone(), tr.state.grad.clone())
        assert all(np.isfinite(losses)) and torch.isfinite(tr.state.data).all() and torch.isfinite(tr.state.grad).all()
        if graph:
            assert len(tr._graphs) == 1
        del tr, m
        torch.cuda.empty_cache()
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2]) and torch.equal(res[False][3], res[True][3])
    print('B=32 640^2 nc=80 bf16 losses', res[True][0])
    flat = {}
    for amp in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.0, amp=amp)
        loss, _ = tr.step(batches[0])
        flat[amp] = (loss.item(), tr.state.grad.clone())
        del tr, m
        torch.cuda.empty_cache()
    cos = torch.nn.functional.cosine_similarity(flat[False][1], flat[True][1], 0).item()
    print(f'B=32 640^2: loss fp32 {flat[False][0]:.4f} bf16 {flat[True][0]:.4f}; flat gradient cosine bf16 vs fp32 {cos:.5f}; |g| {flat[False][1].norm().item():.4f} / {flat[True][1].norm().item():.4f}')
    assert abs(flat[True][0] - flat[False][0]) < 0.02 * abs(flat[False][0])
    assert cos > 0.9


@pytest.mark.gpu
def test_toodhead_backward_matches_autograd_of_the_oracle():
    """TOODHead.backward (GroupNorm+act, layer attention with the per-image reduction weights, DCNv2 columns / col2im with offset and mask
    gradients, probability gate, the shared parameters accumulated over the levels) against torch.autograd of oracle/tood.py on the CPU,
    fp32, two levels.  PARITY UNPINNED like the forward (mmcv absent).  Tolerance: 2e-3 of each tensor's largest value (fp32 reductions in a
    different order; the DCN scatter uses float atomics)."""
    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.nn.modules import TOODHead
    from oracle import tood
    nc, hidc = 8, 64
    m = seed_state_dict_(TOODHead(nc, hidc, (hidc, hidc)), 5)
    m.stride = torch.tensor([8.0, 16.0])
    sd = {'h.' + k: v.clone().requires_grad_(v.dtype.is_floating_point and 'dfl' not in k) for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, hidc, 12, 10, generator=gen).requires_grad_(True), torch.randn(2, hidc, 6, 5, generator=gen).requires_grad_(True)]
    gs = [torch.randn(2, 64 + nc, 12, 10, generator=gen), torch.randn(2, 64 + nc, 6, 5, generator=gen)]
    feats_ref = tood.toodhead_raw(xs, sd, 'h')
    sum((f * g).sum() for f, g in zip(feats_ref, gs)).backward()
    m = m.to(DEV).train()
    nh = lambda t: t.detach().to(DEV).contiguous(memory_format=torch.channels_last)
    with ops.force_ctx(), torch.no_grad():
        feats = m([nh(x) for x in xs])
        for f, fr in zip(feats, feats_ref):
            np.testing.assert_allclose(to_nchw(f), fr.detach().numpy(), atol=2e-3, rtol=2e-3)
        gx = m.backward([nh(g) for g in gs])
    for a, x in zip(gx, xs):
        ref = x.grad.numpy()
        assert np.abs(to_nchw(a) - ref).max() < 2e-3 * np.abs(ref).max(), ('dx', np.abs(to_nchw(a) - ref).max(), np.abs(ref).max())
    bad = []
    for k, p in m.named_parameters():
        r = sd['h.' + k].grad
        if not p.requires_grad:
            continue
        if r is None:                                        # parameters the forward never touches (scale.*, reduction_conv.conv.bias)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        err, ref = (p.grad.detach().cpu() - r).abs().max().item(), r.abs().max().item()
        if not err < 2e-3 * max(ref, 1e-3):
            bad.append((k, err, ref))
    assert not bad, bad


@pytest.mark.gpu
def test_tood_model_trains_in_bf16_and_as_a_captured_step():
    """mspa_c2f_gd_tood_yolov8 through DetectionTrainer: the bf16 (amp) step tracks the fp32 step (loss within 3 %, flat-gradient cosine > 0.9:
    the deformable sampling amplifies activation rounding more than the plain head), and the hipGraph-captured step reproduces the eager
    losses except for the float-atomic scatter of the DCN input gradient (relative 1e-4)."""
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seeded_labels
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    nc, B, S = 4, 2, 96
    batch = dict(img=(seeded_images(B, S, S, seed=2) * 255).to(torch.uint8), **seeded_labels(B, nc, seed=6, max_boxes=4, min_boxes=2))
    batch['bboxes'][:, 2:] = batch['bboxes'][:, 2:] * 0.5 + 0.1
    res = {}
    for amp in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_tood_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.0, amp=amp)
        loss, _ = tr.step(batch)
        res[amp] = (loss.item(), tr.state.grad.clone())
    (l32, g32), (l16, g16) = res[False], res[True]
    cos = torch.nn.functional.cosine_similarity(g32, g16, 0).item()
    print(f'tood: loss fp32 {l32:.4f} bf16 {l16:.4f}; gradient cosine {cos:.4f}')
    assert np.isfinite(l16) and abs(l16 - l32) < 0.03 * abs(l32) and cos > 0.9
    losses = {}
    for graph in (False, True):
        m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_tood_yolov8', 'n', nc), verbose=False), 0).to(DEV)
        tr = DetectionTrainer(m, lr0=0.01, graph=graph)
        losses[graph] = [tr.step(batch)[0].item() for _ in range(4)]
        if graph:
            assert len(tr._graphs) == 1
    np.testing.assert_allclose(losses[True], losses[False], rtol=1e-4)


C3_LDS_CASES = [  # (B, cin, cout, H, W, act, sliced views)
    (4, 64, 96, 80, 80, 'silu', False), (4, 80, 80, 80, 80, 'silu', False), (3, 32, 64, 72, 88, 'none', False), (2, 48, 48, 96, 100, 'relu', True),
    (5, 64, 32, 64, 64, 'silu', True), (2, 72, 80, 120, 90, 'none', False), (3, 80, 80, 76, 84, 'relu', True),
    # stride 2 (8th field): the down-sampling convolutions 32 -> 64 / 64 -> 128 (one / two cout groups of 4 blocks), even, odd and ragged maps, sliced views
    (4, 32, 64, 160, 160, 'silu', False, 2), (3, 64, 128, 80, 80, 'silu', False, 2), (3, 32, 64, 75, 91, 'none', True, 2), (2, 64, 64, 96, 100, 'relu', True, 2),
    (2, 64, 192, 97, 90, 'silu', False, 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', C3_LDS_CASES, ids=[f'{c[1]}to{c[2]}_{c[3]}x{c[4]}_{c[5]}{"_view" if c[6] else ""}{"_s2" if len(c) > 7 else ""}' for c in C3_LDS_CASES])
def test_conv3x3_lds_staged_kernel(case):
    """3x3 stride-1 bf16 convolutions with 32-80 input channels on large maps go through conv3x3_lds.hip (activations staged in LDS, persistent
    workgroups holding the weight panel).  Against torch's fp32 conv2d of the same bf16-rounded tensors + folded BN + activation: the only
    differences are fp32 summation order and the final bf16 rounding (2^-8 relative) - asserted at 1.5 * 2^-8 of the largest output.  Ragged maps
    (not multiples of the 16x16 tile), cout groups that do not fill the last workgroup and channel-slice views included."""
    from mgdt_yolo_amd import ops
    B, ci, co, H, W, act, view = case[:7]
    st = case[7] if len(case) > 7 else 1
    Ho, Wo = ops.conv_out_hw(H, W, 3, st)
    gen = torch.Generator().manual_seed(ci + co + H)
    w = torch.randn(co, ci, 3, 3, generator=gen) / (3 * ci ** 0.5)
    bias = torch.randn(co, generator=gen) * 0.1

    def mk(c, fill):
        t = fill.to(DEV).to(torch.bfloat16)
        if not view:
            return t.contiguous(memory_format=torch.channels_last)
        big = torch.zeros(B, c + 24, *fill.shape[2:], device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        big[:, 8:8 + c] = t
        return big[:, 8:8 + c]
    x = mk(ci, torch.randn(B, ci, H, W, generator=gen))
    y = mk(co, torch.zeros(B, co, Ho, Wo))
    pk = ops.PackedConv(w.to(DEV), bias.to(DEV), None, 3, torch.bfloat16)
    code = {'silu': ops.ACT_SILU, 'relu': ops.ACT_RELU, 'none': ops.ACT_NONE}[act]
    with ops.profile() as prof:
        ops.conv2d(x, pk, st, code, out=y)
    wr = w.to(DEV).to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(x.float().contiguous(), wr, bias.to(DEV), st, 1)
    ref = {'silu': torch.nn.functional.silu, 'relu': torch.relu, 'none': lambda t: t}[act](ref)
    err = (y.float() - ref).abs().max().item()
    assert err < 1.5 * 2 ** -8 * ref.abs().max().item(), (err, ref.abs().max().item())
    if view:
        assert float(y._base[:, :8].abs().max()) == 0.0 and float(y._base[:, 8 + co:].abs().max()) == 0.0      # neighbouring channels untouched


@pytest.mark.gpu
def test_batched_repack_equals_individual_packs():
    """ops.repack_all() (mgdt_conv_pack_batch: every live panel in a few launches after an optimizer step) writes exactly the bytes the
    one-at-a-time packs write: forward panels with and without BN fold, stride-1 data-gradient panels, the four stride-2 phase panels; and a
    panel built from a derived copy (not live parameter storage) is never refreshed in place."""
    from mgdt_yolo_amd import ops
    gen = torch.Generator().manual_seed(7)
    store = torch.randn(200000, generator=gen).to(DEV)
    ops.LIVE_STORAGES.add(store.untyped_storage().data_ptr())
    try:
        off = [0]

        def take(*shape):
            n = int(np.prod(shape))
            t = store[off[0]:off[0] + n].view(*shape)
            off[0] += (n + 3) // 4 * 4
            return t
        w1, w2, w3 = take(48, 32, 3, 3), take(24, 64, 1, 1), take(64, 32, 3, 3)
        bn = (take(48).abs_().add_(0.5), take(48), take(48), take(48).abs_().add_(0.5), 1e-3)            # views of the live storage, adjusted in place
        cb = take(24)
        packs = [ops.PackedConv(w1, None, bn, 3, torch.bfloat16), ops.PackedConv(w2, cb, None, 1, torch.bfloat16), ops.PackedConv(w1, None, None, 3, torch.float32),
                 ops._PackedDgrad(w3, 3, torch.bfloat16, ('k',)), ops._PackedDgrad(w2, 1, torch.float32, ('k',))]
        packs += [ops._PackedDgrad(w3, 3, torch.bfloat16, ('k',), phase=ph) for ph in range(4)]
        copy_pack = ops.PackedConv(torch.cat([w2, w2]), None, None, 1, torch.bfloat16)             # derived copy: must not be registered
        assert all(getattr(p, 'epoch', None) == ops.PARAM_EPOCH[0] for p in packs) and getattr(copy_pack, 'epoch', None) is None
        store.mul_(1.5).add_(0.01)                                                                 # "optimizer step" behind torch's back
        bn[3].abs_()                                                                               # the variance stays positive
        ops.PARAM_EPOCH[0] += 1
        n = len(ops.repack_all())
        assert n >= len(packs)
        fresh = [ops.PackedConv(w1, None, bn, 3, torch.bfloat16), ops.PackedConv(w2, cb, None, 1, torch.bfloat16), ops.PackedConv(w1, None, None, 3, torch.float32),
                 ops._PackedDgrad(w3, 3, torch.bfloat16, ('k',)), ops._PackedDgrad(w2, 1, torch.float32, ('k',))]
        fresh += [ops._PackedDgrad(w3, 3, torch.bfloat16, ('k',), phase=ph) for ph in range(4)]
        taps = [9, 1, 9, 9, 1, 1, 2, 2, 4]                       # K taps of each panel (the phase panels keep 1 / 2 / 2 / 4 of the 9)
        for a, b, t in zip(packs, fresh, taps):
            assert ops.pack_is_current(a)
            pe = 8 if a.dtype == torch.bfloat16 else 4
            nbytes = -(-t * -(-a.cin // pe) // 4) * -(-a.cout // 16) * 1024                        # the panel itself (the buffers carry scratch behind it)
            assert torch.equal(a.w[:nbytes], b.w[:nbytes]) and torch.equal(a.bias, b.bias)
    finally:
        ops.LIVE_STORAGES.discard(store.untyped_storage().data_ptr())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_deferred_weight_gradient_sums_equal_the_immediate_form(dtype):
    """Inside ops.defer_wgrad() a convolution leaves its per-split partial sums in its workspace and one mgdt_wgrad_final_batch launch finishes all
    of them: bit-identical to the immediate form (same fixed summation order); an accumulating gradient flushes the pending ones first."""
    from mgdt_yolo_amd import ops
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    cases = [(mk(2, 16, 24, 24), mk(2, 32, 24, 24), 3, 1), (mk(2, 64, 12, 12), mk(2, 24, 12, 12), 1, 1), (mk(2, 8, 32, 32), mk(2, 16, 16, 16), 3, 2)]
    ref = []
    for x, dy, k, s in cases:
        dw = torch.empty(dy.shape[1], x.shape[1], k, k, device=DEV)
        ops.conv_wgrad(x, dy, k, s, dw)
        ref.append(dw)
    out = [torch.full_like(r, float('nan')) for r in ref]
    with ops.defer_wgrad():
        for (x, dy, k, s), dw in zip(cases, out):
            ops.conv_wgrad(x, dy, k, s, dw)
        assert len(ops._WGRAD_PENDING) == len(cases)
        x, dy, k, s = cases[0]
        ops.conv_wgrad(x, dy, k, s, out[0], accumulate=True)                # flushes, then adds: out[0] = 2 * ref[0]
        assert not ops._WGRAD_PENDING
    for i, (o, r) in enumerate(zip(out, ref)):
        assert torch.equal(o, r * 2 if i == 0 else r), i


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_stride2_data_gradient_phases(dtype):
    """Data gradient of a stride-2 3x3 convolution = four stride-1 phase convolutions over dy whose K holds only the 1 / 2 / 4 taps the phase uses
    (mgdt_conv2d_phase_fwd) against torch's conv_transpose2d in fp64; also with an accumulated addend (the shortcut fused into the epilogue)."""
    from mgdt_yolo_amd import ops
    gen = torch.Generator().manual_seed(13)
    B, ci, co, H, W = 2, 32, 64, 20, 28
    w = (torch.randn(co, ci, 3, 3, generator=gen) / 17).to(DEV)
    dy = torch.randn(B, co, H // 2, W // 2, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    r2 = torch.randn(B, ci, H, W, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    wq = w.to(dtype).double() if dtype == torch.bfloat16 else w.double()
    ref = torch.nn.functional.conv_transpose2d(dy.double(), wq, stride=2, padding=1, output_padding=1)
    dx = ops.conv_dgrad(dy, w, 3, 2, ops.new_act(B, ci, H, W, dtype, DEV))
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert (dx.double() - ref).abs().max().item() < tol * ref.abs().max().item()
    dx2 = ops.conv_dgrad(dy, w, 3, 2, ops.new_act(B, ci, H, W, dtype, DEV), r2=r2)
    assert (dx2.double() - (ref + r2.double())).abs().max().item() < tol * (ref + r2.double()).abs().max().item()


@pytest.mark.gpu
def test_bilinear_adjoint_2x_stencil_equals_generic_scan():
    """The exact-2x fast path of the bilinear adjoint (fixed 4x4 stencil on interior pixels) against autograd of F.interpolate, fp32, and against a
    non-2x shape that takes the generic candidate scan."""
    from mgdt_yolo_amd import ops
    gen = torch.Generator().manual_seed(5)
    for (h, w, oh, ow) in ((10, 12, 20, 24), (10, 12, 23, 24)):
        x = torch.randn(2, 8, h, w, generator=gen, requires_grad=True)
        g = torch.randn(2, 8, oh, ow, generator=gen)
        torch.nn.functional.interpolate(x, size=(oh, ow), mode='bilinear', align_corners=False).backward(g)
        got = ops.bilinear_bwd(g.to(DEV).contiguous(memory_format=torch.channels_last), ops.new_act(2, 8, h, w, torch.float32, DEV))
        np.testing.assert_allclose(to_nchw(got), x.grad.numpy(), atol=2e-6, rtol=2e-6)


# ------------------------------------------------------------------------------------------------ fp8 (BASELINE configs[4])
def _e4m3(t):
    """OCP e4m3fn quantisation as the gfx950 conversion does it: round to nearest even after clamping to the finite range (+-448)."""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


FP8_CONV_CASES = [  # (B, cin, h, w, cout, k, s, xq)
    (2, 32, 20, 24, 64, 3, 1, 16.0), (2, 64, 16, 16, 96, 3, 2, 32.0), (1, 128, 13, 17, 256, 1, 1, 8.0), (2, 256, 10, 10, 128, 1, 1, 64.0),
    (1, 40, 9, 9, 20, 3, 1, 16.0), (2, 512, 8, 8, 256, 1, 1, 16.0), (1, 16, 40, 40, 16, 3, 1, 1.0), (1, 192, 20, 20, 80, 3, 1, 16.0),
    (1, 2048, 6, 6, 32, 3, 1, 16.0),   # 576 K-chunks: the panel of one cout block does not fit in LDS (segmented staging)
    # large maps, 32-80 input channels: the plain variant runs on the LDS-staged kernel's e4m3 form (conv3x3_lds_kernel<.., Q8>)
    (1, 64, 128, 128, 96, 3, 1, 16.0), (1, 80, 128, 130, 80, 3, 1, 16.0), (2, 32, 96, 100, 48, 3, 1, 8.0), (1, 48, 130, 127, 32, 3, 1, 16.0),
]


@pytest.mark.parametrize('case', FP8_CONV_CASES, ids=lambda c: 'x'.join(str(v) for v in c[:7]))
def test_conv_fp8_matches_e4m3_emulation(case):
    """mgdt_conv2d_fp8_fwd == conv2d over e4m3-quantised operands (per-tensor activation multiplier, per-output-channel weight scale, BN folded)
    accumulated in fp32: only the accumulation order and the final bf16 rounding differ, so the bound is ~2 bf16 ulps of the largest output
    (measured on MI355X: <= 2.6e-3 relative to max|ref|).  Plain, and with every fused extra (pre-add, residuals, channel-slice views)."""
    import torch.nn.functional as F
    from mgdt_yolo_amd import ops
    B, cin, h, w, cout, k, s, xq = case
    r = np.random.default_rng(cin * 131 + cout)
    wt = torch.from_numpy((r.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32))
    gam, bet = torch.from_numpy(r.uniform(0.5, 1.5, cout).astype(np.float32)), torch.from_numpy((r.standard_normal(cout) * 0.1).astype(np.float32))
    mu, var = torch.from_numpy((r.standard_normal(cout) * 0.1).astype(np.float32)), torch.from_numpy(r.uniform(0.5, 1.5, cout).astype(np.float32))
    eps = 1e-3
    xw = torch.from_numpy((r.standard_normal((B, cin + 16, h, w)) * 2.0).astype(np.float32))
    x2 = torch.from_numpy(r.standard_normal((B, cin, h, w)).astype(np.float32))
    ho, wo = ops.conv_out_hw(h, w, k, s)
    r1 = torch.from_numpy(r.standard_normal((B, cout, ho, wo)).astype(np.float32))
    cl = lambda t: t.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dq = lambda t: t.to(torch.bfloat16).float()
    xd, x2d, r1d = cl(xw), cl(x2), cl(r1)
    pk = ops.PackedConvFp8(wt.to(DEV), None, tuple(t.to(DEV) for t in (gam, bet, mu, var)) + (eps,), k, xq)
    # the emulation, in the kernel's operation order
    fold = gam / torch.sqrt(eps + var)
    wf = wt * fold[:, None, None, None]
    ws = wf.abs().amax(dim=(1, 2, 3)) / 448.0
    wq = _e4m3(wf / ws[:, None, None, None])
    bias = bet - gam * mu / torch.sqrt(var + eps)
    np.testing.assert_allclose(pk.oscale[:cout].cpu().numpy(), (ws / xq).numpy(), rtol=1e-6)
    np.testing.assert_allclose(pk.bias[:cout].cpu().numpy(), bias.numpy(), rtol=1e-5, atol=1e-6)
    outw = torch.zeros(B, cout + 8, ho, wo, dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
    for variant in ('plain', 'fused'):
        xin = dq(xw[:, 8:8 + cin])
        if variant == 'plain':
            ops.conv2d_fp8(xd[:, 8:8 + cin], pk, s, ops.ACT_SILU, out=outw[:, 4:4 + cout])
            acc = F.conv2d(_e4m3(xin * xq).double(), wq.double(), None, s, k // 2) * (ws / xq).double()[None, :, None, None] + bias.double()[None, :, None, None]
            ref = F.silu(acc)
        else:
            ops.conv2d_fp8(xd[:, 8:8 + cin], pk, s, ops.ACT_RELU, out=outw[:, 4:4 + cout], x2=x2d, r1=r1d)
            a = (xin + dq(x2)).to(torch.bfloat16).float()
            acc = F.conv2d(_e4m3(a * xq).double(), wq.double(), None, s, k // 2) * (ws / xq).double()[None, :, None, None] + bias.double()[None, :, None, None]
            ref = F.relu(acc) + dq(r1).double()
        got = outw[:, 4:4 + cout].float().cpu().double()
        err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        print(f'fp8 conv {case} {variant}: err {err:.2e} (max|ref| {ref.abs().max().item():.2f})')
        assert err < 8e-3, (variant, err)
        assert outw[:, :4].abs().max().item() == 0 and outw[:, 4 + cout:].abs().max().item() == 0
    # and the quantisation itself is what it should cost: e4m3 operands vs the unquantised convolution, relative to the output's scale
    full = F.silu(F.conv2d(dq(xw[:, 8:8 + cin]).double(), wf.double(), bias.double(), s, k // 2))
    ops.conv2d_fp8(xd[:, 8:8 + cin], pk, s, ops.ACT_SILU, out=outw[:, 4:4 + cout])
    qerr = (outw[:, 4:4 + cout].float().cpu().double() - full).pow(2).mean().sqrt().item() / full.pow(2).mean().sqrt().item()
    print(f'fp8 conv {case}: rms quantisation error {qerr:.3f} of the output rms')
    assert qerr < 0.06, qerr


def test_conv_fp8_saturates_instead_of_overflowing():
    """Inputs beyond the calibrated range clamp to +-448 / xq (no NaN / Inf from the e4m3 conversion)."""
    from mgdt_yolo_amd import ops
    wt = torch.ones(16, 8, 1, 1)
    pk = ops.PackedConvFp8(wt.to(DEV), None, None, 1, 64.0)
    x = torch.full((1, 8, 4, 4), 1000.0, dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fp8(x, pk, 1, ops.ACT_NONE)
    assert torch.isfinite(y.float()).all()
    np.testing.assert_allclose(y.float().cpu().numpy(), 8 * 448.0 / 64.0, rtol=1e-2)


# fp8 inference vs the fp32 REFERENCE fixtures: stated tolerance = ~2x the errors measured on MI355X (printed by the test).  The e4m3 operands
# (3 mantissa bits) of the implicit-GEMM convolutions cost ~4 % rms per layer.  Measured (round 2, gpurun_out/s3_fp8.log), 2x160x160, max over
# every anchor / class (mean in brackets): mspa_c2f_gd_n 5.0 px (0.86) / 0.30 (0.021) with 18 fp8 convolutions; yolov8_n 1.7 px (0.06) / 0.017
# (0.0008) with 30.  bf16 on the same inputs: 0.68 px / 0.034 and 0.23 px / 0.0031.
FP8_TOL = {'mspa_c2f_gd_n': (10.0, 0.6), 'yolov8_n': (3.5, 0.04)}


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_e2e_fp8_stated_tolerance(golden, tag):
    """BASELINE configs[4]: model.quantize_fp8(calibration batch) -> every Conv on the implicit-GEMM kernel runs e4m3 MFMAs.  Boxes / confidences vs
    the reference's fp32 output within the stated fp8 tolerance; mean errors printed; the calibration table covers the igemm convolutions;
    dequantize_fp8() restores the bf16 result bit for bit."""
    g = golden('e2e_' + tag)
    m = build_model(GI.E2E_MODELS[tag], torch.bfloat16)
    x = seeded_images(2, 160, 160, seed=GI.IMG_SEED).to(DEV)
    with torch.no_grad():
        y16, _ = m(x)
    table = m.quantize_fp8(seeded_images(4, 160, 160, seed=GI.IMG_SEED + 1).to(DEV))
    assert len(table) >= 10 and all(q > 0 and np.log2(q) == int(np.log2(q)) for q in table.values()), table
    from mgdt_yolo_amd import ops
    with torch.no_grad(), ops.profile() as p:
        y8, _ = m(x)
    names = [n for n, _, _ in p.rows]
    # what stays bf16 on the igemm kernel: the head's two final 1x1 convolutions to the raw maps (at sizes the fused detect tail does not take)
    assert names.count('conv2d_fp8_fwd') >= 10 and names.count('conv2d_fwd') <= 2 * len(m.model[-1].cv2), names
    ref = g['y_2x160x160']
    y = y8.cpu().numpy()
    eb, ec = np.abs(y[:, :4] - ref[:, :4]).max(), np.abs(y[:, 4:] - ref[:, 4:]).max()
    print(f'fp8 {tag}: {names.count("conv2d_fp8_fwd")} fp8 convs of {len(names)} launches; max box err {eb:.3f} px (mean {np.abs(y[:, :4] - ref[:, :4]).mean():.3f}), '
          f'max conf err {ec:.4f} (mean {np.abs(y[:, 4:] - ref[:, 4:]).mean():.5f})')
    assert eb < FP8_TOL[tag][0] and ec < FP8_TOL[tag][1], (eb, ec)
    m.dequantize_fp8()
    with torch.no_grad():
        y16b, _ = m(x)
    assert torch.equal(y16, y16b)


# ------------------------------------------------------------------------------------------------ mAP@0.5 parity vs the CPU reference path (the second half of BASELINE's metric)
# |mAP50(HIP) - mAP50(CPU oracle)| and the same for mAP50-95 on identical images / labels; stated bounds = ~2-3x the differences measured on MI355X (printed).
# measured (round 2, gpurun_out/s3_map*.log; reference pipeline mAP50 0.696 / mAP50-95 0.46 on the synthetic labels): fp32 0.0000 / see log, bf16 -0.015 / +0.001, fp8 -0.125 / -0.108
# (seeded-random weights put ~1000 detections per image within a few percent of each other's score, so the rank order - all that AP sees - is far more
# sensitive to operand precision than a trained network's)
MAP_TOL = {'f32': (1e-3, 1e-3), 'bf16': (0.04, 0.04), 'fp8': (0.25, 0.25), 'fp8_head_bf16': (0.25, 0.25)}


def test_map50_parity_with_the_cpu_reference_pipeline():
    """Whole validation chain on identical inputs: forward -> NMS(conf 0.001, iou 0.7, multi_label, max_det 300; val.py:63-71) -> scale_boxes ->
    _process_batch -> ap_per_class -> mAP@0.5 / mAP@0.5:0.95, once through the CPU oracle (torch-CPU fp32 forward, numpy NMS / matching / AP: the
    restatement of the reference's CPU path) and once through the HIP path in fp32, bf16 and fp8.  Labels (half of the oracle's own detections,
    jittered, plus a few boxes nothing predicts) make the metric non-trivial (mAP50 ~0.7).  fp32: mAP equal to 1e-3 (north_star
    'mAP@0.5 parity vs CPU ref'); bf16 / fp8: stated bounds."""
    from mgdt_yolo_amd.yolo.v8.detect import DetectionValidator
    from oracle import layers as OL, metrics as OM, nms as ON, val as OV
    name, nc, B, S = 'mspa_c2f_gd_yolov8', 80, 4, 320
    cfg = get_config(name, 'n', nc)
    m = build_model(name)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = seeded_images(B, S, S, seed=GI.IMG_SEED + 7)
    with torch.no_grad():
        y_ref, _ = OL.model_forward(cfg, sd, x, [float(s) for s in m.stride.tolist()], fused=True)
    y_ref = y_ref.numpy()
    r = np.random.default_rng(11)
    # labels: the upper half of the reference's own validation-time detections (seeded-random weights predict only a few classes, all with high scores:
    # ranking half of the 300 kept boxes as objects and half as background gives AP ~0.5-0.9 per class), jittered by 6 % of their size, plus five
    # displaced copies per image that nothing predicts
    det_ref = ON.non_max_suppression(y_ref, conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300, compiled=True)
    cls_l, box_l, idx_l = [], [], []
    for si, rows in enumerate(det_ref):
        rows = rows[:150].copy()
        wh = np.stack([rows[:, 2] - rows[:, 0], rows[:, 3] - rows[:, 1]], 1)
        rows[:, :4] += (r.standard_normal((len(rows), 4)) * 0.06 * np.concatenate([wh, wh], 1)).astype(np.float32)
        ex = rows[r.integers(0, len(rows), 5)].copy()
        ex[:, :4] += r.uniform(40, 90, (5, 1)).astype(np.float32)
        rows = np.concatenate([rows, ex]).astype(np.float32)
        rows[:, [0, 2]] = rows[:, [0, 2]].clip(0, S - 1); rows[:, [1, 3]] = rows[:, [1, 3]].clip(0, S - 1)
        xywh = np.stack([(rows[:, 0] + rows[:, 2]) / 2 / S, (rows[:, 1] + rows[:, 3]) / 2 / S, (rows[:, 2] - rows[:, 0]) / S, (rows[:, 3] - rows[:, 1]) / S], 1)
        cls_l.append(rows[:, 5:6]); box_l.append(xywh.astype(np.float32)); idx_l.append(np.full(len(rows), si, np.float32))
    batch = dict(img=torch.zeros(B, 3, S, S, dtype=torch.uint8, device=DEV), cls=torch.from_numpy(np.concatenate(cls_l)), bboxes=torch.from_numpy(np.concatenate(box_l)),
                 batch_idx=torch.from_numpy(np.concatenate(idx_l)), ori_shape=[(S, S)] * B, ratio_pad=[((1.0, 1.0), (0.0, 0.0))] * B)
    # the CPU reference pipeline
    iouv = torch.linspace(0.5, 0.95, 10)
    stats = []
    for si, det in enumerate(det_ref):
        xywh = box_l[si]
        tb = np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], 1).astype(np.float32) * np.float32(S)
        tb = OM.scale_boxes((S, S), tb, (S, S), ((1.0, 1.0), (0.0, 0.0)))                    # val.py:96-105: native-space boxes, clipped to the image
        predn = OM.scale_boxes((S, S), det[:, :4].copy(), (S, S), ((1.0, 1.0), (0.0, 0.0)))
        correct = OV.process_batch(torch.from_numpy(np.concatenate([predn, det[:, 4:]], 1)), torch.from_numpy(np.concatenate([cls_l[si], tb], 1)), iouv)
        stats.append((np.asarray(correct), det[:, 4], det[:, 5], cls_l[si][:, 0]))
    tp, conf, pcls, tcls = [np.concatenate(v, 0) for v in zip(*stats)]
    ap = OM.ap_per_class(tp, conf, pcls, tcls)[5]
    ref50, ref5095 = float(ap[:, 0].mean()), float(ap.mean())
    assert 0.2 < ref50 < 0.98, ref50                 # the synthetic labels make a non-trivial metric
    print(f'CPU reference pipeline: mAP50 {ref50:.4f}  mAP50-95 {ref5095:.4f}  ({len(tcls)} labels, {len(conf)} detections)')

    def hip_map(model, xin):
        v = DetectionValidator(DEV)
        v.init_metrics(nc=nc)
        with torch.no_grad():
            y, _ = model(xin)
        v.update_metrics(v.postprocess(y), batch)
        res = v.get_stats()
        return res['metrics/mAP50(B)'], res['metrics/mAP50-95(B)']

    xd = x.to(DEV)
    got = {'f32': hip_map(m, xd)}
    m.half()
    got['bf16'] = hip_map(m, xd.to(torch.bfloat16))
    calib = seeded_images(B, S, S, seed=GI.IMG_SEED + 8).to(DEV).to(torch.bfloat16)
    m.quantize_fp8(calib)
    got['fp8'] = hip_map(m, xd.to(torch.bfloat16))
    head = f'model.{len(m.model) - 1}.'
    t = m.quantize_fp8(calib, exclude=(head,))                       # mixed policy: the Detect head's convolutions keep bf16 operands
    assert t and not any(k.startswith(head[:-1]) for k in t)
    got['fp8_head_bf16'] = hip_map(m, xd.to(torch.bfloat16))
    extra = {}
    for tag, kw in (('fp8 max, headroom 1', dict(headroom=1.0)), ('fp8 p99.99, headroom 1', dict(headroom=1.0, percentile=99.99)),
                    ('fp8 p99.9, headroom 1', dict(headroom=1.0, percentile=99.9)), ('fp8 p99.99, headroom 1, head bf16', dict(headroom=1.0, percentile=99.99, exclude=(head,)))):
        m.quantize_fp8(calib, **kw)                                   # calibration variants: reported, not asserted
        extra[tag] = hip_map(m, xd.to(torch.bfloat16))
    for k, (a50, a5095) in extra.items():
        print(f'HIP {k}: mAP50 {a50:.4f} (diff {a50 - ref50:+.4f})  mAP50-95 {a5095:.4f} (diff {a5095 - ref5095:+.4f})')
    for k, (a50, a5095) in got.items():
        print(f'HIP {k}: mAP50 {a50:.4f} (diff {a50 - ref50:+.4f})  mAP50-95 {a5095:.4f} (diff {a5095 - ref5095:+.4f})')
    for k, (a50, a5095) in got.items():
        assert abs(a50 - ref50) <= MAP_TOL[k][0] and abs(a5095 - ref5095) <= MAP_TOL[k][1], (k, a50, ref50, a5095, ref5095)


def test_fp8_on_the_tood_scale_s_model():
    """quantize_fp8 on the BASELINE configs[3] graph (scale s, TOOD head): the backbone / neck / head `Conv`s switch to e4m3 operands, the TOOD-specific kernels
    (GroupNorm convs, layer attention, DCNv2) keep bf16; outputs stay finite and within a stated distance of the bf16 forward (fp8 vs bf16 of the same weights,
    2x320x320; measured on MI355X and printed)."""
    from mgdt_yolo_amd import ops
    name = 'mspa_c2f_gd_tood_yolov8_hidc128'
    m = build_model(name, torch.bfloat16, scale='s')
    x = seeded_images(2, 320, 320, seed=5).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        y16 = m(x)[0].float()
    table = m.quantize_fp8(seeded_images(2, 320, 320, seed=6).to(DEV).to(torch.bfloat16))
    with torch.no_grad(), ops.profile() as p:
        y8 = m(x)[0].float()
    n8 = sum(1 for n, _, _ in p.rows if n == 'conv2d_fp8_fwd')
    assert len(table) >= 10 and n8 >= 10, (len(table), n8)
    assert torch.isfinite(y8).all()
    eb, ec = (y8[:, :4] - y16[:, :4]).abs(), (y8[:, 4:] - y16[:, 4:]).abs()
    print(f'fp8 tood-s: {n8} fp8 convs of {len(p.rows)} launches; vs bf16: box max {eb.max().item():.3f} px (mean {eb.mean().item():.4f}), conf max {ec.max().item():.4f} (mean {ec.mean().item():.5f})')
    # measured (round 2): 25 fp8 convolutions of 94 launches; box 28.9 px max / 2.84 px mean (reg_max 16 at 1280-class strides, DCNv2 offsets downstream of the fp8
    # convolutions), conf 0.112 max / 0.0038 mean; bounds = ~2x
    assert eb.mean().item() < 6.0 and ec.mean().item() < 0.01 and eb.max().item() < 60.0 and ec.max().item() < 0.25
