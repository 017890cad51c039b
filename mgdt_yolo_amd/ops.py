"""Thin torch-tensor wrappers over the C ABI (include/mgdt.h).  PyTorch is plumbing here: device memory,
the current HIP stream, nn.Parameter containers.  Every compute step goes through libmgdt_hip.so.

Activations are torch tensors shaped (B, C, H, W) in channels_last memory format (NHWC in memory); channel
slices `t[:, a:b]` are passed as strided views - the reference's chunk()/split()/cat() never copy here.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU, BF16, F32, View  # noqa: F401

U8 = 2    # uint8 image input of the stem only (divided by 255 in the kernel)
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dtype_code(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise RuntimeError(f'mgdt_yolo_amd computes in float32 or bfloat16, not {dt}') from None


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('mgdt_yolo_amd runs on MI355X (HIP) only: got a CPU tensor and there is no CPU/PyTorch fallback')


_PROF = None   # list collecting (name, meta, start_event, end_event) while ops.profile() is active


def _launch(name, symbol, *args, meta=None):
    """Call one C-ABI entry point on the current stream; raise on a non-zero status."""
    fn = getattr(L.lib(), symbol)
    if _PROF is None:
        L.check(fn(*args), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(fn(*args), name)
    e1.record()
    _PROF.append((name, meta or _META.pop(name, None), e0, e1))


_META = {}


class profile:
    """with ops.profile() as p: ...  -> p.rows = [(name, meta, ms)] measured with HIP events on the launch stream."""

    def __enter__(self):
        global _PROF
        _PROF = self._raw = []
        return self

    def __exit__(self, *exc):
        global _PROF
        _PROF = None
        torch.cuda.synchronize()
        self.rows = [(n, m, a.elapsed_time(b)) for n, m, a, b in self._raw]
        return False


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


SIDE_STREAM = os.environ.get('MGDT_SIDE_STREAM', '1') != '0'      # inference: layers that do not depend on their predecessors run on a second HIP stream (BaseModel._side_branch)
_SIDE = {}


def side_stream(device):
    """The second launch stream of `device` (one per process, device and lane - see ops.lane; created outside graph capture on first use)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, _LANE[0])
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


_FORCE_CTX = [0]


def ctx_enabled():
    """True when a train-mode forward should keep activations for backward(): autograd is recording, or the model-level
    autograd Function (which runs its forward under no_grad) asked for it."""
    return _FORCE_CTX[0] > 0 or torch.is_grad_enabled()


class force_ctx:
    def __enter__(self):
        _FORCE_CTX[0] += 1

    def __exit__(self, *exc):
        _FORCE_CTX[0] -= 1
        return False


def view(t):
    """(B,C,H,W) tensor of any strides -> mgdt_view (NHWC-ordered sizes/strides)."""
    _need_gpu(t)
    b, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    return View(t.data_ptr(), b, h, w, c, sn, sh, sw, sc)


def vp(t):
    return C.byref(view(t)) if t is not None else None


def ptr(t):
    if t is None:
        return None
    _need_gpu(t)
    return C.c_void_p(t.data_ptr())


def _same(*ts):
    """Views that one kernel addresses with ONE dtype code must really share it: a bf16 map read as fp32 is an out-of-bounds access on the device,
    so the mismatch is refused here."""
    ts = [t for t in ts if t is not None]
    for t in ts[1:]:
        if t.dtype != ts[0].dtype:
            raise RuntimeError(f'mgdt_yolo_amd: operands of one kernel must share a dtype, got {[str(u.dtype) for u in ts]}')


def new_act(b, c, h, w, dtype, device):
    """NHWC activation buffer presented with NCHW shape semantics."""
    return torch.empty((b, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def is_nhwc(t):
    return t.stride(1) == 1


def grad_buf(p):
    """The gradient tensor of parameter p (allocated on first use; the trainer points it into its flat gradient buffer)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def like(t):
    """Dense NHWC buffer with t's shape/dtype (torch.empty_like would inherit a channel-slice view's odd strides)."""
    return new_act(t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.dtype, t.device)


# ------------------------------------------------------------------ conv
class PackedConv:
    """Caller-owned packed weights of one convolution (BN folded) for one compute dtype."""
    __slots__ = ('w', 'bias', 'k', 'cin', 'cout', 'dtype', 'direct', 'groups', 'src', 'epoch', '__weakref__')

    def __init__(self, weight, conv_bias, bn, k, dtype, direct=False, groups=1):
        """weight: (cout, cin/groups, k, k) fp32 cuda; bn: None or (gamma, beta, mean, var, eps)."""
        lib = L.lib()
        weight = weight.detach().float().contiguous()
        _need_gpu(weight)
        cout, cin_g = weight.shape[0], weight.shape[1]
        dev = weight.device
        g, b, mu, var, eps = (None, None, None, None, 0.0) if bn is None else bn
        f = lambda t: None if t is None else t.detach().float().contiguous()
        g, b, mu, var, cb = f(g), f(b), f(mu), f(var), f(conv_bias)
        self.k, self.cin, self.cout, self.dtype, self.direct, self.groups = k, cin_g * groups, cout, dtype, direct, groups
        if direct:
            self.w = torch.empty(k * k * cin_g * cout, dtype=torch.float32, device=dev)
            self.bias = torch.empty(cout, dtype=torch.float32, device=dev)
            _launch('conv_pack_direct', 'mgdt_conv_pack_direct', ptr(weight), ptr(cb), ptr(g), ptr(b), ptr(mu), ptr(var), eps, cin_g, cout, k,
                                              ptr(self.w), ptr(self.bias), stream())
        else:
            code = dtype_code(dtype)
            nbytes = lib.mgdt_conv_packed_bytes(cin_g, cout, k, code)
            self.w = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.bias = torch.empty((cout + 15) // 16 * 16, dtype=torch.float32, device=dev)
            _launch('conv_pack', 'mgdt_conv_pack', ptr(weight), ptr(cb), ptr(g), ptr(b), ptr(mu), ptr(var), eps, cin_g, cout, k, code,
                                       ptr(self.w), ptr(self.bias), stream())
            if groups == 1:
                _register_pack(self, (weight, cb, g, b, mu, var, float(eps), cin_g, cout, k, code, 0))


class PackedConvFp8:
    """e4m3 panel of one convolution (BN folded, per-output-channel weight scales) for the activation multiplier `xq` - mgdt_conv_pack_fp8."""
    __slots__ = ('w', 'bias', 'oscale', 'xq', 'k', 'cin', 'cout', 'dtype', 'direct', 'groups', '__weakref__')

    def __init__(self, weight, conv_bias, bn, k, xq):
        lib = L.lib()
        weight = weight.detach().float().contiguous()
        _need_gpu(weight)
        cout, cin = weight.shape[0], weight.shape[1]
        dev = weight.device
        g, b, mu, var, eps = (None, None, None, None, 0.0) if bn is None else bn
        f = lambda t: None if t is None else t.detach().float().contiguous()
        g, b, mu, var, cb = f(g), f(b), f(mu), f(var), f(conv_bias)
        self.k, self.cin, self.cout, self.dtype, self.direct, self.groups, self.xq = k, cin, cout, torch.bfloat16, False, 1, float(xq)
        self.w = torch.empty(lib.mgdt_conv_packed_bytes_fp8(cin, cout, k), dtype=torch.uint8, device=dev)
        cpad = (cout + 15) // 16 * 16
        self.bias = torch.empty(cpad, dtype=torch.float32, device=dev)
        self.oscale = torch.empty(cpad, dtype=torch.float32, device=dev)
        _launch('conv_pack_fp8', 'mgdt_conv_pack_fp8', ptr(weight), ptr(cb), ptr(g), ptr(b), ptr(mu), ptr(var), eps, cin, cout, k, self.xq,
                ptr(self.w), ptr(self.bias), ptr(self.oscale), stream())


Q8_CALIB = None    # dict module -> running max|input| while BaseModel.quantize_fp8() runs its calibration batches
Q8_CALIB_PCT = None  # None: the maximum of |input|; a number: that percentile of |input| (per batch, the largest over the batches)


def conv2d_fp8(x, pk, stride, act, out=None, x2=None, r1=None, r2=None, in_scale=None, in_shift=None):
    """mgdt_conv2d_fp8_fwd: bf16 views, e4m3 operands on the MFMA, fp32 accumulation."""
    _need_gpu(x)
    b, _, h, w = x.shape
    ho, wo = conv_out_hw(h, w, pk.k, stride)
    if out is None:
        out = new_act(b, pk.cout, ho, wo, torch.bfloat16, x.device)
    _same(x, x2, r1, r2, out)
    if x.dtype != torch.bfloat16:
        raise RuntimeError(f'conv2d_fp8: activations are bf16 in HBM, the input is {x.dtype}')
    if _PROF is not None:
        _META['conv2d_fp8_fwd'] = dict(shape=(b, pk.cin, h, w, pk.cout, pk.k, stride), flops=2.0 * b * ho * wo * pk.cout * pk.cin * pk.k * pk.k,
                                       bytes=float(b * h * w * pk.cin * 2 + b * ho * wo * pk.cout * 2 + pk.cout * pk.cin * pk.k * pk.k))
    _launch('conv2d_fp8_fwd', 'mgdt_conv2d_fp8_fwd', vp(x), vp(x2), ptr(in_scale), ptr(in_shift), ptr(pk.w), ptr(pk.bias), ptr(pk.oscale), pk.xq,
            pk.k, stride, act, vp(r1), vp(r2), vp(out), stream())
    return out


# ---- batched re-pack after an optimizer step -------------------------------------------------------------------------------------
# Every packed panel built from live parameter storage registers (weakly) what it was built from.  `repack_all()` - called by the trainer
# right after the HIP optimizer moved the weights - refreshes all panels that were valid for the step that just ran with
# mgdt_conv_pack_batch (a handful of launches instead of one per convolution) and stamps them with the new PARAM_EPOCH; the caches that
# hold them (HipModule._cached, the data-gradient cache below) accept a stamped panel instead of rebuilding it.
_PACK_REGISTRY = None
LIVE_STORAGES = set()          # storages that hold the parameters / buffers the HIP optimizer updates in place (FlatState registers its buffer)


def _register_pack(obj, src):
    """Only panels built directly from live parameter storage can be refreshed in place: a panel packed from a derived copy (torch.cat of two
    branches, zero-padded rows) must be rebuilt from a fresh copy, so it is never registered."""
    global _PACK_REGISTRY
    import weakref
    if not all(t is None or t.untyped_storage().data_ptr() in LIVE_STORAGES for t in src[:6]):
        return
    if _PACK_REGISTRY is None:
        _PACK_REGISTRY = weakref.WeakSet()
    obj.src, obj.epoch = src, PARAM_EPOCH[0]
    _PACK_REGISTRY.add(obj)


def repack_all():
    """Refresh every registered panel that was current before the last optimizer step (PARAM_EPOCH - 1); returns them."""
    if not _PACK_REGISTRY:
        return []
    live = [o for o in _PACK_REGISTRY if getattr(o, 'epoch', -2) == PARAM_EPOCH[0] - 1]
    if not live:
        return []
    arr = (L.PackDesc * len(live))()
    for d, o in zip(arr, live):
        w, cb, g, b, mu, var, eps, cin, cout, k, code, mode = o.src
        d.w, d.conv_bias, d.bn_gamma, d.bn_beta, d.bn_mean, d.bn_var = ptr(w), ptr(cb), ptr(g), ptr(b), ptr(mu), ptr(var)
        d.bn_eps, d.cin, d.cout, d.k, d.dtype, d.mode = eps, cin, cout, k, code, mode
        d.packed, d.bias_out = ptr(o.w), ptr(o.bias)
    _launch('conv_pack_batch', 'mgdt_conv_pack_batch', arr, len(live), stream())
    for o in live:
        o.epoch = PARAM_EPOCH[0]
    return live


def pack_is_current(obj):
    return getattr(obj, 'epoch', None) == PARAM_EPOCH[0]


def conv_can_mfma(x, cin, cout, k, s, groups, dtype):
    pe = 8 if dtype == torch.bfloat16 else 4
    return (groups == 1 and k in (1, 3) and s in (1, 2) and cin % pe == 0 and cout % 4 == 0 and x.dtype == dtype
            and is_nhwc(x) and x.stride(3) % pe == 0 and x.stride(2) % pe == 0 and x.stride(0) % pe == 0
            and x.data_ptr() % 16 == 0)


def conv_out_hw(h, w, k, s):
    p = k // 2
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def conv2d(x, pk, stride, act, out=None, x2=None, r1=None, r2=None, in_scale=None, in_shift=None):
    """Fused conv (see mgdt_conv2d_fwd).  Returns `out` (allocated NHWC when None)."""
    _need_gpu(x)
    b, _, h, w = x.shape
    ho, wo = conv_out_hw(h, w, pk.k, stride)
    if out is None:
        out = new_act(b, pk.cout, ho, wo, pk.dtype, x.device)
    if _PROF is not None:
        es = out.element_size()
        _META['conv2d_direct_fwd' if pk.direct else 'conv2d_fwd'] = dict(
            shape=(b, pk.cin, h, w, pk.cout, pk.k, stride), flops=2.0 * b * ho * wo * pk.cout * pk.cin // pk.groups * pk.k * pk.k,
            bytes=float(b * h * w * pk.cin * x.element_size() + b * ho * wo * pk.cout * es + pk.w.numel() * pk.w.element_size()))
    if pk.direct:
        if x2 is not None or r1 is not None or r2 is not None or in_scale is not None or in_shift is not None:
            raise RuntimeError('direct convolution path has no fused extras')
        _launch('conv2d_direct_fwd', 'mgdt_conv2d_direct_fwd', vp(x), U8 if x.dtype == torch.uint8 else dtype_code(x.dtype), ptr(pk.w), ptr(pk.bias), pk.k, stride, pk.groups, act,
                                           vp(out), dtype_code(out.dtype), stream())
    else:
        _same(x, x2, r1, r2, out)
        if x.dtype != pk.dtype:
            raise RuntimeError(f'conv2d: the weights are packed for {pk.dtype}, the input is {x.dtype}')
        _launch('conv2d_fwd', 'mgdt_conv2d_fwd', vp(x), vp(x2), ptr(in_scale), ptr(in_shift), ptr(pk.w), ptr(pk.bias), pk.k, stride, act,
                                    vp(r1), vp(r2), vp(out), dtype_code(pk.dtype), stream())
    return out


# ------------------------------------------------------------------ MSPA attention
def spr_attention(x, fc1_w, fc1_b, fc2_w, fc2_b, groups, softmax=True):
    """softmax_over_groups(SPR(x_group)) -> attn fp32 [B, C] (softmax=False: the bare sigmoid weights)."""
    b, c, h, w = x.shape
    lib = L.lib()
    part = torch.empty(b * L.SPR_SPLITS * c * 5, dtype=torch.float32, device=x.device)
    _launch('spr_pool_fwd', 'mgdt_spr_pool_fwd', vp(x), ptr(part), dtype_code(x.dtype), stream())
    attn = torch.empty(b, c, dtype=torch.float32, device=x.device)
    _launch('spr_attn_fwd', 'mgdt_spr_attn_fwd', ptr(part), ptr(fc1_w), ptr(fc1_b), ptr(fc2_w), ptr(fc2_b), b, c, groups, h, w, int(bool(softmax)),
            ptr(attn), stream())
    return attn


def spr_attention_scale(x, fc1_w, fc1_b, fc2_w, fc2_b, groups, out=None, part=None, nsplit=0, tiles=(0, 0), pools=()):
    """out = x * softmax_over_groups(SPR(x_group)): pooling pass (skipped when the producing kernel already left its per-tile sums in
    `part`, fp32 [b][nsplit][c] over a tiles[0] x tiles[1] (x, y) tile grid), then attention MLP + scaling in one launch.  `pools`: up to two
    NHWC views (b, c, h / F, w / F) that receive adaptive_avg_pool2d(out) in the same launch."""
    b, c, h, w = x.shape
    if part is None:
        part = torch.empty(b * L.SPR_SPLITS * c * 5, dtype=torch.float32, device=x.device)
        _launch('spr_pool_fwd', 'mgdt_spr_pool_fwd', vp(x), ptr(part), dtype_code(x.dtype), stream())
        nsplit, tiles = L.SPR_SPLITS, (0, 0)
    out = like(x) if out is None else out
    pools = list(pools)
    if len(pools) > 2:
        raise RuntimeError('spr_attention_scale: at most two pooled outputs')
    _same(x, out, *pools)
    pa, pb = (pools + [None, None])[:2]
    _launch('spr_attn_scale_fwd', 'mgdt_spr_attn_scale_fwd', ptr(part), int(nsplit), int(tiles[0]), int(tiles[1]), ptr(fc1_w), ptr(fc1_b), ptr(fc2_w), ptr(fc2_b), groups, vp(x), vp(out),
            vp(pa), vp(pb), dtype_code(x.dtype), stream())
    return out


FUSED_NECK = True     # tests flip this: SimFusion inputs delivered by their producers (pooled copies / direct concat-slot writes) vs separate launches


# ------------------------------------------------------------------ layers 0 + 1 in one launch
FUSED_STEM = True        # tests flip this to compare against the two conv launches


class PackedStem2:
    """Layer-0 weights (3 -> 16, k3 s2, BN folded exactly as fuse_conv_and_bn) in the fragment order of mgdt_stem2_fwd + bias."""

    def __init__(self, weight, bn):
        g, b, mu, var, eps = bn
        scale = (g.detach().float() / torch.sqrt(eps + var.detach().float()))
        wf = (weight.detach().float() * scale.view(-1, 1, 1, 1)).contiguous()
        self.bias = (b.detach().float() - g.detach().float() * mu.detach().float() / torch.sqrt(var.detach().float() + eps)).contiguous()
        self.blob = torch.empty(L.lib().mgdt_stem2_packed_bytes(), dtype=torch.uint8, device=weight.device)
        L.check(L.lib().mgdt_stem2_pack(ptr(wf), ptr(self.blob), stream()), 'stem2_pack')


def stem2(x, pk0, pk1):
    """x (B, 3, H, W) NCHW image (bf16 / fp32 / uint8) -> SiLU(conv1(SiLU(conv0(x)))) as (B, 32, H/4, W/4) bf16 NHWC."""
    _need_gpu(x)
    b, _, h, w = x.shape
    h0, w0 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    h1, w1 = (h0 - 1) // 2 + 1, (w0 - 1) // 2 + 1
    y = new_act(b, 32, h1, w1, torch.bfloat16, x.device)
    if _PROF is not None:
        _META['stem2_fwd'] = dict(shape=(b, 3, h, w, 32, 3, 4), flops=2.0 * b * (h0 * w0 * 16 * 27 + h1 * w1 * 32 * 144),
                                  bytes=float(x.numel() * x.element_size() + y.numel() * 2))
    _launch('stem2_fwd', 'mgdt_stem2_fwd', vp(x), U8 if x.dtype == torch.uint8 else dtype_code(x.dtype), ptr(pk0.blob), ptr(pk0.bias), ptr(pk1.w), ptr(pk1.bias),
            vp(y), stream())
    return y


# ------------------------------------------------------------------ whole CSP block (MSPA_C2f / C2f) in one launch
FUSED_CSP_BLOCK = True   # tests flip this to compare against the per-conv launch chain
CSP_MSPA, CSP_C2F = 0, 1


def csp_block_supported(mode, x, cout, wd, nbtl, dtype):
    """channel counts / dtype covered AND a tile decomposition exists for this map (tiles must divide it: none for e.g. a 37-row map)"""
    if not (FUSED_CSP_BLOCK and dtype == torch.bfloat16 and x.dtype == dtype and is_nhwc(x)):
        return False
    lib = L.lib()
    if not lib.mgdt_csp_block_supported(mode, x.shape[1], int(cout), int(wd), int(nbtl), x.shape[2], x.shape[3], dtype_code(dtype)):
        return False
    return lib.mgdt_csp_block_tiles(mode, x.shape[0], x.shape[1], int(cout), int(wd), int(nbtl), x.shape[2], x.shape[3], None) > 0


def csp_block(mode, x, front, front_bias, mids, shortcut, back, wd, act, cout, want_pool):
    """mgdt_csp_block_fwd: `front` = PackedPwChain.blob (MSPA) or PackedConv (C2f); `mids` = PackedConv list of the bottlenecks' 3x3 convs;
    `back` = PackedConv of the 1x1 over the concat.  Returns (y, per-tile channel sums or None, pool slots, (tiles_x, tiles_y))."""
    lib = L.lib()
    b, cin, h, w = x.shape
    nb = len(mids) // 2
    geom = (C.c_int * 8)()
    slots = lib.mgdt_csp_block_tiles(mode, b, cin, cout, wd, nb, h, w, geom)
    if slots <= 0:
        raise RuntimeError('csp_block: configuration not covered')
    y = new_act(b, cout, h, w, x.dtype, x.device)
    pool = torch.empty(b * slots * cout, dtype=torch.float32, device=x.device) if want_pool else None
    marr = (C.c_void_p * len(mids))(*[m.w.data_ptr() for m in mids])
    barr = (C.c_void_p * len(mids))(*[m.bias.data_ptr() for m in mids])
    if _PROF is not None:
        es = x.element_size()
        catc = (3 if mode == CSP_MSPA else 2) * wd + nb * wd
        fl = 2.0 * b * h * w * ((3 * wd * wd if mode == CSP_MSPA else 0) + 2 * nb * 9 * wd * wd + catc * cout)
        _META['csp_block_fwd'] = dict(shape=(b, cin, h, w, cout, wd, nb), flops=fl, bytes=float(b * h * w * (cin + cout) * es))
    _launch('csp_block_fwd', 'mgdt_csp_block_fwd', mode, vp(x), ptr(front), ptr(front_bias), marr, barr, nb, int(bool(shortcut)), ptr(back.w), ptr(back.bias),
            int(wd), act, vp(y), ptr(pool), dtype_code(x.dtype), stream())
    return y, pool, slots, (geom[6], geom[7])


def scale_channels(x, attn, out=None):
    out = like(x) if out is None else out
    _same(x, out)
    _launch('scale_channels_fwd', 'mgdt_scale_channels_fwd', vp(x), ptr(attn), vp(out), dtype_code(x.dtype), stream())
    return out


# ------------------------------------------------------------------ pools / resamplers
def sppf_pools(x, y1, y2, y3):
    _launch('sppf_pool_fwd', 'mgdt_sppf_pool_fwd', vp(x), vp(y1), vp(y2), vp(y3), dtype_code(x.dtype), stream())


def adaptive_avgpool(x, out):
    _same(x, out)
    _launch('adaptive_avgpool_fwd', 'mgdt_adaptive_avgpool_fwd', vp(x), vp(out), dtype_code(x.dtype), stream())
    return out


def bilinear(x, out):
    _same(x, out)
    _launch('bilinear_fwd', 'mgdt_bilinear_fwd', vp(x), vp(out), dtype_code(x.dtype), stream())
    return out


def nearest(x, out):
    _same(x, out)
    _launch('nearest_fwd', 'mgdt_nearest_fwd', vp(x), vp(out), dtype_code(x.dtype), stream())
    return out


def copy(x, out):
    """Strided copy with cast (channel concat, NCHW<->NHWC, fp32<->bf16)."""
    _launch('copy_fwd', 'mgdt_copy_fwd', vp(x), U8 if x.dtype == torch.uint8 else dtype_code(x.dtype), vp(out), dtype_code(out.dtype), stream())
    return out


# ------------------------------------------------------------------ ConvNeXtV2 pieces, Injection, Detect
def dwconv7_ln(x, dw_w49c, dw_b, ln_w, ln_b, eps, out=None):
    out = like(x) if out is None else out
    _launch('dwconv7_ln_fwd', 'mgdt_dwconv7_ln_fwd', vp(x), ptr(dw_w49c), ptr(dw_b), ptr(ln_w), ptr(ln_b), eps, vp(out), dtype_code(x.dtype),
                                        stream())
    return out


def grn_scale(t, gamma):
    """scale[n,c] = gamma[c]*Nx[n,c] + 1 (fp32) for the GRN-folded pwconv2."""
    b, c = t.shape[:2]
    ws = torch.empty(b * L.GRN_SPLITS * c, dtype=torch.float32, device=t.device)
    sc = torch.empty(b, c, dtype=torch.float32, device=t.device)
    _launch('grn_stats_fwd', 'mgdt_grn_stats_fwd', vp(t), ptr(gamma), ptr(ws), ptr(sc), dtype_code(t.dtype), stream())
    return sc


FUSED_PW_CHAIN = True   # tests flip this to compare against three conv launches


def pw_chain_supported(wd, dtype):
    return FUSED_PW_CHAIN and L.lib().mgdt_pw_chain_packed_bytes(int(wd), dtype_code(dtype)) > 0


class PackedPwChain:
    """Three wd->wd 1x1 convs (BN folded) in the fragment order of mgdt_pw_chain3_fwd."""

    def __init__(self, convs, dtype):
        """convs: three (weight (wd, wd, 1, 1), conv_bias or None, bn tuple or None)."""
        wd = convs[0][0].shape[0]
        self.wd, self.dtype = wd, dtype
        self.blob = torch.empty(L.lib().mgdt_pw_chain_packed_bytes(wd, dtype_code(dtype)), dtype=torch.uint8, device=convs[0][0].device)
        f = lambda t: None if t is None else t.detach().float().contiguous()
        for i, (w, cb, bn) in enumerate(convs):
            g, b, mu, var, eps = (None, None, None, None, 0.0) if bn is None else bn
            w, cb, g, b, mu, var = f(w), f(cb), f(g), f(b), f(mu), f(var)
            L.check(L.lib().mgdt_pw_chain_pack(i, ptr(w), ptr(cb), ptr(g), ptr(b), ptr(mu), ptr(var), eps, wd, dtype_code(dtype), ptr(self.blob), stream()),
                    'pw_chain_pack')


def pw_chain3(x, pk, act, out):
    """out[:, i*wd:(i+1)*wd] = sp_i of the MSPA point-wise chain (mgdt_pw_chain3_fwd); x, out: (B, 3*wd, H, W) NHWC views."""
    if _PROF is not None:
        b, c, h, w = x.shape
        _META['pw_chain3_fwd'] = dict(shape=(b, pk.wd, h, w, pk.wd, 1, 1), flops=2.0 * b * h * w * pk.wd * pk.wd * 3, bytes=float(2 * b * h * w * c * x.element_size()))
    _launch('pw_chain3_fwd', 'mgdt_pw_chain3_fwd', vp(x), ptr(pk.blob), pk.wd, act, vp(out), dtype_code(x.dtype), stream())
    return out


FUSED_CNX_MLP = True    # tests flip this to compare against the three-launch chain


def cnx_mlp_supported(c, dtype):
    """True when the fused ConvNeXtV2 MLP kernels cover (c, dtype); otherwise callers keep the conv/GRN chain."""
    return FUSED_CNX_MLP and L.lib().mgdt_cnx_mlp_packed_bytes(int(c), dtype_code(dtype)) > 0


class PackedCnxMlp:
    """pwconv1/pwconv2 weights + biases in the fragment order of the fused MLP kernels (see mgdt_cnx_mlp_pack)."""

    def __init__(self, w1, b1, w2, b2, dtype):
        c = w2.shape[0]
        dev = w1.device
        self.c, self.dtype = c, dtype
        self.blob = torch.empty(L.lib().mgdt_cnx_mlp_packed_bytes(c, dtype_code(dtype)), dtype=torch.uint8, device=dev)
        f = lambda t: t.detach().float().contiguous()
        w1, b1, w2, b2 = f(w1), f(b1), f(w2), f(b2)
        L.check(L.lib().mgdt_cnx_mlp_pack(ptr(w1), ptr(b1), ptr(w2), ptr(b2), c, ptr(self.blob), dtype_code(dtype), stream()))


def cnx_mlp(t, res, pk, gamma, beta, out=None):
    """out = res + pwconv2(GRN(gelu(pwconv1(t)))) with the hidden map kept on chip (mgdt_cnx_mlp_fwd)."""
    b, c, h, w = t.shape
    out = like(t) if out is None else out
    ws = torch.empty(L.lib().mgdt_cnx_mlp_workspace_bytes(b, h, w, c), dtype=torch.uint8, device=t.device)
    if _PROF is not None:
        _META['cnx_mlp_fwd'] = dict(shape=(b, c, h, w, c, 1, 1), flops=2.0 * b * h * w * c * 4 * c * 3, bytes=float(4 * b * h * w * c * t.element_size()))
    _launch('cnx_mlp_fwd', 'mgdt_cnx_mlp_fwd', vp(t), vp(res), ptr(pk.blob), ptr(gamma), ptr(beta), ptr(ws), vp(out), dtype_code(t.dtype), stream())
    return out


FUSED_CNX_BLOCK = True  # tests flip this to compare against dwconv7_ln + the two-pass MLP
FUSED_CNX_TAIL = True   # ... and the IFM's closing 1x1 conv inside the last block's launch vs a launch of its own
_CNX_WS = {}            # (device, shape, lane) -> zero-initialised workspace of mgdt_cnx_block_fwd (its barrier words live across calls)
_LANE = [0]


class lane:
    """with ops.lane(i): forwards whose launches may run CONCURRENTLY with those of another lane (several captured graph instances replayed on
    different streams) - kernels that keep state in a cached workspace (the block barrier of mgdt_cnx_block_fwd) get one workspace per lane."""

    def __init__(self, i):
        self.i = int(i)

    def __enter__(self):
        self.prev, _LANE[0] = _LANE[0], self.i

    def __exit__(self, *exc):
        _LANE[0] = self.prev
        return False


def cnx_block_supported(x, dtype):
    b, c, h, w = x.shape
    return bool(FUSED_CNX_BLOCK and x.dtype == dtype and is_nhwc(x) and L.lib().mgdt_cnx_block_supported(b, h, w, c, dtype_code(dtype)))


def cnx_block(x, dw_w49c, dw_b, ln_w, ln_b, eps, pk, gamma, beta, out=None, tail=None, tail_act=0):
    """out = x + pwconv2(GRN(gelu(pwconv1(LayerNorm(dwconv7x7(x)))))) in one launch (mgdt_cnx_block_fwd); `tail`: PackedConv (input channels in
    `acc_order_index` order) of a 1x1 Conv+BN+act applied to that result inside the launch - `out` then has the conv's channels."""
    b, c, h, w = x.shape
    if out is None:
        out = like(x) if tail is None else new_act(b, tail.cout, h, w, x.dtype, x.device)
    if out.data_ptr() == x.data_ptr():
        raise RuntimeError('cnx_block: the output may not alias the input (tiles read their neighbours\' halo)')
    nbytes = L.lib().mgdt_cnx_block_workspace_bytes(b, h, w, c)
    key = (x.device, b, c, h, w, _LANE[0])
    ws = _CNX_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('cnx_block: run the model once before capturing it into a graph (the kernel\'s barrier words are allocated and zeroed on first use)')
        ws = _CNX_WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
    if _PROF is not None:
        _META['cnx_block_fwd'] = dict(shape=(b, c, h, w, c, 7, 1), flops=2.0 * b * h * w * c * (4 * c * 2 + 49), bytes=float(2 * b * h * w * c * x.element_size()))
    _launch('cnx_block_fwd', 'mgdt_cnx_block_fwd', vp(x), ptr(dw_w49c), ptr(dw_b), ptr(ln_w), ptr(ln_b), float(eps), ptr(pk.blob), ptr(gamma), ptr(beta),
            None if tail is None else ptr(tail.w), None if tail is None else ptr(tail.bias), int(tail_act), ptr(ws), nbytes, vp(out), dtype_code(x.dtype), stream())
    return out


FUSED_INJECT = True     # tests flip this to compare against conv + inject


def conv1x1_inject_supported(x, cout, ga, dtype):
    return bool(FUSED_INJECT and x.dtype == dtype and ga.dtype == dtype and is_nhwc(x) and is_nhwc(ga) and
                L.lib().mgdt_conv1x1_inject_supported(x.shape[1], cout, x.shape[2], x.shape[3], ga.shape[2], ga.shape[3], dtype_code(dtype)))


def acc_order_index(c, device):
    """Input-channel permutation that turns an ordinary packed 1x1 panel into one whose K runs in the MFMA ACCUMULATOR order of the conv in
    front of it: packed channel (j*4 + g)*8 + e <- channel (2*j + e//4)*16 + 4*g + e%4."""
    k = torch.arange(c, device=device)
    j, g, e = k // 32, (k // 8) % 4, k % 8
    return (2 * j + e // 4) * 16 + 4 * g + e % 4


def conv1x1_inject_conv_supported(x, cmid, cout2, ga, dtype):
    return bool(FUSED_INJECT and FUSED_INJECT_CONV and x.dtype == dtype and ga.dtype == dtype and is_nhwc(x) and is_nhwc(ga) and
                L.lib().mgdt_conv1x1_inject_conv_supported(x.shape[1], cmid, cout2, x.shape[2], x.shape[3], ga.shape[2], ga.shape[3], dtype_code(dtype)))


def conv1x1_inject_conv(x, pk, ga, gf, pk2, act2, out, gsrc=None, pkg=None):
    """out = act2(conv1x1_2(conv1x1(x) * bilinear(h_sigmoid(ga)) + bilinear(gf))) in one launch (mgdt_conv1x1_inject_conv_fwd); pk2: the second
    conv packed with its input channels in `acc_order_index` order.  With (gsrc, pkg) instead of (ga, gf): the two global 1x1 convs (merged
    panel pkg over the 32-channel gsrc) run inside the launch as well."""
    b, _, h, w = x.shape
    _same(x, ga, gf, gsrc, out)
    if _PROF is not None:
        _META['conv1x1_inject_conv_fwd'] = dict(shape=(b, pk.cin, h, w, pk2.cout, 1, 1), flops=2.0 * b * h * w * (pk.cout * pk.cin + pk2.cout * pk2.cin),
                                                bytes=float(b * h * w * (pk.cin + pk2.cout) * x.element_size() + (2 * ga.numel() * ga.element_size() if gsrc is None else gsrc.numel() * 2)))
    _launch('conv1x1_inject_conv_fwd', 'mgdt_conv1x1_inject_conv_fwd', vp(x), ptr(pk.w), ptr(pk.bias), vp(ga), vp(gf), vp(gsrc), None if pkg is None else ptr(pkg.w),
            None if pkg is None else ptr(pkg.bias), pk.cout, ptr(pk2.w), ptr(pk2.bias), int(act2), vp(out), dtype_code(pk.dtype), stream())
    return out


FUSED_INJECT_CONV = True   # tests flip this: injection + C2f.cv1 in one launch vs two
FUSED_INJECT_GCONV = True  # ... and the injection's two global 1x1 convs inside that launch vs a launch of their own


def conv1x1_inject(x, pk, ga, gf, out=None):
    """out = conv1x1(x) * bilinear(h_sigmoid(ga)) + bilinear(gf) in one launch (mgdt_conv1x1_inject_fwd)."""
    b, _, h, w = x.shape
    if out is None:
        out = new_act(b, pk.cout, h, w, pk.dtype, x.device)
    if _PROF is not None:
        _META['conv1x1_inject_fwd'] = dict(shape=(b, pk.cin, h, w, pk.cout, 1, 1), flops=2.0 * b * h * w * pk.cout * pk.cin,
                                           bytes=float(b * h * w * (pk.cin + pk.cout) * x.element_size() + 2 * ga.numel() * ga.element_size()))
    _launch('conv1x1_inject_fwd', 'mgdt_conv1x1_inject_fwd', vp(x), ptr(pk.w), ptr(pk.bias), vp(ga), vp(gf), vp(out), dtype_code(pk.dtype), stream())
    return out


# ------------------------------------------------------------------ TOODHead pieces
def groupnorm(x, gamma, beta, groups, eps, act, out=None):
    """out = act(GroupNorm(x)) on NHWC (mgdt_groupnorm_fwd)."""
    b, c = x.shape[:2]
    out = like(x) if out is None else out
    ws = torch.empty(L.lib().mgdt_groupnorm_workspace_bytes(b, c), dtype=torch.uint8, device=x.device)
    _launch('groupnorm_fwd', 'mgdt_groupnorm_fwd', vp(x), ptr(gamma), ptr(beta), groups, float(eps), act, ptr(ws), vp(out), dtype_code(x.dtype), stream())
    return out


def tood_layer_attn(feat, w1, b1, w2, b2, stacked, sums=None):
    """TaskDecomposition's layer attention as a per-(image, input channel) scale fp32 [B, C] for the reduction conv."""
    b, c, h, w = feat.shape
    sums = nc_reduce(feat) if sums is None else sums
    scale = torch.empty(b, c, dtype=torch.float32, device=feat.device)
    _launch('tood_layer_attn_fwd', 'mgdt_tood_layer_attn_fwd', ptr(sums), b, c, h * w, ptr(w1), ptr(b1), ptr(w2), ptr(b2), w1.shape[0], stacked, ptr(scale), stream())
    return scale


def dcnv2(x, offset_mask, w_gemm, bias, cout):
    b, _, h, w = x.shape
    out = new_act(b, cout, h, w, x.dtype, x.device)
    _launch('dcnv2_fwd', 'mgdt_dcnv2_fwd', vp(x), vp(offset_mask), ptr(w_gemm), ptr(bias), vp(out), dtype_code(x.dtype), stream())
    return out


def dcnv2_mfma(x, offset_mask, pk):
    """DCNv2 on the MFMA path (bf16): pk = PackedConv(weight, None, None, 3, bfloat16)."""
    b, _, h, w = x.shape
    out = new_act(b, pk.cout, h, w, x.dtype, x.device)
    _launch('dcnv2_mfma_fwd', 'mgdt_dcnv2_mfma_fwd', vp(x), vp(offset_mask), ptr(pk.w), vp(out), dtype_code(x.dtype), stream())
    return out


def pixel_gate(x, gate, out=None):
    out = like(x) if out is None else out
    _launch('pixel_gate_fwd', 'mgdt_pixel_gate_fwd', vp(x), vp(gate), vp(out), dtype_code(x.dtype), stream())
    return out


# ---- TOODHead training (tood_train.hip) ----
def gn_affine(y, gamma, beta, groups, eps):
    """GroupNorm of the conv output y as the per-(image, channel) affine u = y*A + B; returns (A, B, mean, rstd) fp32."""
    b, c, h, w = y.shape
    sy, syy = nc_reduce(y), nc_reduce(y, y)
    dev = y.device
    A, B = torch.empty(b, c, dtype=torch.float32, device=dev), torch.empty(b, c, dtype=torch.float32, device=dev)
    mean, rstd = torch.empty(b, groups, dtype=torch.float32, device=dev), torch.empty(b, groups, dtype=torch.float32, device=dev)
    _launch('gn_affine', 'mgdt_gn_affine', ptr(sy), ptr(syy), b, c, h * w, ptr(gamma), ptr(beta), groups, float(eps), ptr(mean), ptr(rstd), ptr(A), ptr(B), stream())
    return A, B, mean, rstd


def nc_affine_act_bwd(g, y, A, B, act):
    gu = like(y)
    _same(g, y)
    _launch('nc_affine_act_bwd', 'mgdt_nc_affine_act_bwd', vp(g), vp(y), ptr(A), ptr(B), act, vp(gu), dtype_code(y.dtype), stream())
    return gu


def relu_mask(g, out_act):
    """g * (out_act > 0): backward of a fused ReLU from its OUTPUT."""
    return nc_affine_act_bwd(g, out_act, None, None, ACT_RELU)


def nc_axpby(a, sa, b=None, sb=None, shift=None, out=None):
    out = like(a) if out is None else out
    _same(a, b, out)
    _launch('nc_axpby', 'mgdt_nc_axpby', vp(a), ptr(sa), vp(b), ptr(sb), ptr(shift), vp(out), dtype_code(a.dtype), stream())
    return out


def gn_act_bwd(g, y, gamma, beta, groups, eps, act, dgamma, dbeta, accumulate=False):
    """Backward of z = act(GroupNorm(y)): returns dy; dgamma / dbeta written or accumulated."""
    b, c, h, w = y.shape
    A, B, mean, rstd = gn_affine(y, gamma, beta, groups, eps)
    gu = nc_affine_act_bwd(g, y, A, B, act)
    s1, s2 = nc_reduce(gu), nc_reduce(gu, y)
    P, Q, R = (torch.empty(b, c, dtype=torch.float32, device=y.device) for _ in range(3))
    ws = torch.empty(L.lib().mgdt_gn_bwd_workspace_bytes(b, c), dtype=torch.uint8, device=y.device)
    _launch('gn_bwd_coef', 'mgdt_gn_bwd_coef', ptr(s1), ptr(s2), ptr(mean), ptr(rstd), ptr(gamma), b, c, h * w, groups, ptr(P), ptr(Q), ptr(R), ptr(dgamma), ptr(dbeta),
            int(accumulate), ptr(ws), stream())
    return nc_axpby(gu, P, y, Q, R, out=gu)


def pixel_gate_bwd(g, x, logit, glogit=None):
    """-> (gx, glogit); `glogit` may be a one-channel view of a wider zero-filled buffer (the padded gradient of a 1-output conv)."""
    gx = like(x)
    gl = like(logit) if glogit is None else glogit
    _same(g, x, logit, gl)
    _launch('pixel_gate_bwd', 'mgdt_pixel_gate_bwd', vp(g), vp(x), vp(logit), vp(gx), vp(gl), dtype_code(x.dtype), stream())
    return gx, gl


def tood_layer_attn_bwd(sums, dscale, hw, w1, b1, w2, b2, stacked, dw1, db1, dw2, db2, accumulate=False):
    n, c = sums.shape
    hid = w1.shape[0]
    dsums = torch.empty_like(sums)
    ws = torch.empty(L.lib().mgdt_tood_layer_attn_bwd_workspace_bytes(n, c, hid, stacked), dtype=torch.uint8, device=sums.device)
    _launch('tood_layer_attn_bwd', 'mgdt_tood_layer_attn_bwd', ptr(sums), ptr(dscale), n, c, hw, ptr(w1), ptr(b1), ptr(w2), ptr(b2), hid, stacked, ptr(dsums),
            ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), int(accumulate), ptr(ws), stream())
    return dsums


def dcn_im2col(x, om):
    b, c, h, w = x.shape
    col = new_act(b, 9 * c, h, w, x.dtype, x.device)
    _same(x, om)
    _launch('dcn_im2col', 'mgdt_dcn_im2col', vp(x), vp(om), vp(col), dtype_code(x.dtype), stream())
    return col


def dcn_col2im_bwd(gcol, x, om):
    """-> (gx in x's dtype, gom like om)."""
    b, c, h, w = x.shape
    gx32 = torch.zeros(b, h, w, c, dtype=torch.float32, device=x.device)
    gom = like(om)
    _same(gcol, x, om)
    _launch('dcn_col2im_bwd', 'mgdt_dcn_col2im_bwd', vp(gcol), vp(x), vp(om), ptr(gx32), vp(gom), dtype_code(x.dtype), stream())
    gx = gx32.permute(0, 3, 1, 2)                          # (B,C,H,W) view of the NHWC buffer = channels_last
    return (gx if x.dtype == torch.float32 else copy(gx, like(x))), gom


def ew_add_nc(a, b):
    """Sum of two small fp32 [n, c] coefficient arrays (torch element-wise on a few hundred values)."""
    return a + b


def inject(local, ga, gf, out=None):
    out = like(local) if out is None else out
    _same(local, ga, gf, out)
    _launch('inject_fwd', 'mgdt_inject_fwd', vp(local), vp(ga), vp(gf), vp(out), dtype_code(local.dtype), stream())
    return out


def detect_decode(feat, reg_max, nc, stride, a_off, y):
    """feat (B,no,H,W) NHWC -> y[B, 4+nc, A_total] fp32 at anchor offset a_off."""
    _launch('detect_decode_fwd', 'mgdt_detect_decode_fwd', vp(feat), reg_max, nc, float(stride), a_off, y.shape[2], ptr(y), dtype_code(feat.dtype),
                                           stream())


FUSED_DETECT_TAIL = True   # tests flip this to compare against conv + conv + decode


def detect_tail_supported(tb, tc, nc, reg_max, dtype):
    return bool(FUSED_DETECT_TAIL and dtype == torch.bfloat16 and tb.dtype == dtype and tc.dtype == dtype and is_nhwc(tb) and is_nhwc(tc) and
                L.lib().mgdt_detect_tail_supported(tb.shape[1], tc.shape[1], int(nc), int(reg_max), dtype_code(dtype)))


FUSED_DETECT_BOX3 = True   # tests flip this: the box branch's second 3x3 conv inside the Detect tail launch vs a launch of its own


def detect_tail(tb, tc, pkb, pkc, nc, stride, a_off, feat, y, best=None, pk3=None):
    """mgdt_detect_tail_fwd: final 1x1 convs of both branches + raw map `feat` + decode into y (+ per-anchor best-class NMS keys into `best`,
    int64 [B][A], when given).  pk3: PackedConv of the box branch's second 3x3 conv - `tb` is then that conv's 16-channel input and `pkb` the
    final 1x1 packed over 32 zero-padded input channels."""
    if _PROF is not None:
        b, _, h, w = tb.shape
        _META['detect_tail_fwd'] = dict(shape=(b, tb.shape[1] + tc.shape[1], h, w, 16 + nc, 1, 1), flops=2.0 * b * h * w * (tb.shape[1] * 16 + tc.shape[1] * nc),
                                        bytes=float((tb.numel() + tc.numel() + feat.numel()) * 2 + b * (4 + nc) * h * w * 4))
    _launch('detect_tail_fwd', 'mgdt_detect_tail_fwd', vp(tb), vp(tc), ptr(pkb.w), ptr(pkb.bias), ptr(pkc.w), ptr(pkc.bias), int(nc), float(stride), int(a_off),
            y.shape[2], vp(feat), ptr(y), ptr(best), None if pk3 is None else ptr(pk3.w), None if pk3 is None else ptr(pk3.bias), stream())


# ------------------------------------------------------------------ NMS
NMS_USE_BEST_KEYS = True      # tests flip this to compare against the scan of the score rows (nms_best_kernel)


def attach_best_keys(y, best):
    """The Detect tail kernel left the NMS key of every anchor's best class in `best`; remember it on the prediction tensor together with
    the tensor's version, so that `nms(y)` can skip its own scan over the nc score rows as long as nobody wrote into y since."""
    y._mgdt_best = (best, y._version)


def _best_keys_of(pred, b, a):
    bk = getattr(pred, '_mgdt_best', None)
    if not NMS_USE_BEST_KEYS or bk is None:
        return None
    best, ver = bk
    if ver != pred._version or best.shape != (b, a) or best.dtype != torch.int64 or best.device != pred.device or not best.is_contiguous():
        return None
    return best
def nms(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, max_nms, max_wh):
    """pred (B, 4+nc, A) fp32 cuda contiguous -> (out [B,max_det,6], kept_anchor [B,max_det] int32, counts [B] int32)."""
    _need_gpu(pred)
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        raise RuntimeError('nms: prediction must be a contiguous float32 tensor')
    b, ch, a = pred.shape
    nc = ch - 4
    lib = L.lib()
    ml = 1 if (multi_label and nc > 1) else 0
    ws_bytes = lib.mgdt_nms_workspace_bytes(b, nc, a, ml, max_nms)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=pred.device)
    # rows past counts[i] are never written (and never read: callers slice by counts); counts[i] is written for every image
    out = torch.empty(b, max_det, 6, dtype=torch.float32, device=pred.device)
    kept = torch.empty(b, max_det, dtype=torch.int32, device=pred.device)
    counts = torch.empty(b, dtype=torch.int32, device=pred.device)
    cls_t = None
    if classes is not None:
        cls_t = torch.as_tensor(list(classes), dtype=torch.int32).to(pred.device)
    best = None if ml else _best_keys_of(pred, b, a)
    _launch('nms_fwd', 'mgdt_nms_fwd', ptr(pred), b, nc, a, float(conf_thres), float(iou_thres), ptr(cls_t), 0 if cls_t is None else cls_t.numel(),
                             int(bool(agnostic)), ml, max_det, max_nms, float(max_wh), ptr(out), ptr(kept), ptr(counts), ptr(best), ptr(ws),
                             ws_bytes, stream())
    return out, kept, counts


def val_match(det, ndet, labels, nlab, iouv):
    """Validator matching for a batch (mgdt_val_match_fwd): det (B, max_det, 6) fp32 + ndet (B,) int32 as returned by `nms`, labels
    (B, max_lab, 5) fp32 [cls, x1, y1, x2, y2] + nlab (B,) int32, iouv (T,) fp32 -> correct (B, max_det, T) bool."""
    _need_gpu(det)
    b, md, _ = det.shape
    ml = labels.shape[1]
    correct = torch.empty(b, md, iouv.numel(), dtype=torch.uint8, device=det.device)
    _launch('val_match_fwd', 'mgdt_val_match_fwd', ptr(det), ptr(ndet), b, md, ptr(labels), ptr(nlab), ml, ptr(iouv), iouv.numel(), ptr(correct), stream())
    return correct.view(torch.bool)


# ------------------------------------------------------------------ detection loss (assigner + BCE/CIoU/DFL)
def _view_array(ts):
    views = [view(t) for t in ts]
    arr = (C.POINTER(View) * len(views))(*[C.pointer(v) for v in views])
    return arr, views


class DetectLossState:
    """Device buffers of one v8DetectionLoss evaluation (kept for the backward call)."""
    __slots__ = ('feats', 'strides', 'gt', 'n_gt', 'out5', 'ws', 'ws_bytes', 'reg_max', 'nc', 'gains', 'fg', 'gt_idx', 'tscore')


def detect_loss_fwd(feats, strides, reg_max, nc, gt, call_count, gains, want_assignment=False, call_count_dev=None):
    """feats: list of (B, 4R+nc, H, W) NHWC tensors; gt: (B, N, 5) fp32 cuda [cls, x1, y1, x2, y2] px.  Returns DetectLossState."""
    lib = L.lib()
    for f in feats:
        _need_gpu(f)
        if not is_nhwc(f):
            raise RuntimeError('detect_loss: head maps must be channels_last (NHWC) tensors')
    dev = feats[0].device
    b = feats[0].shape[0]
    a_total = sum(f.shape[2] * f.shape[3] for f in feats)
    n_gt = int(gt.shape[1])
    st = DetectLossState()
    st.feats, st.reg_max, st.nc, st.gains, st.n_gt = feats, reg_max, nc, gains, n_gt
    st.strides = torch.tensor([float(s) for s in strides], dtype=torch.float32)      # host array for the C call
    st.gt = gt.contiguous().float()
    st.out5 = torch.zeros(5, dtype=torch.float32, device=dev)
    st.ws_bytes = lib.mgdt_detect_loss_workspace_bytes(b, a_total, n_gt)
    st.ws = torch.empty(st.ws_bytes, dtype=torch.uint8, device=dev)
    st.fg = torch.empty(b, a_total, dtype=torch.uint8, device=dev) if want_assignment else None
    st.gt_idx = torch.empty(b, a_total, dtype=torch.int32, device=dev) if want_assignment else None
    st.tscore = torch.empty(b, a_total, dtype=torch.float32, device=dev) if want_assignment else None
    arr, keep = _view_array(feats)
    sarr = (C.c_float * len(feats))(*st.strides.tolist())
    if call_count_dev is not None:              # int32[1] on the device: the captured training step (hipGraph) replays with the host's counter
        _launch('detect_loss_fwd', 'mgdt_detect_loss_fwd_dev', arr, sarr, len(feats), reg_max, nc, ptr(st.gt) if n_gt else None, n_gt, ptr(call_count_dev),
                float(gains[0]), float(gains[1]), float(gains[2]), ptr(st.out5), ptr(st.fg), ptr(st.gt_idx), ptr(st.tscore), ptr(st.ws), st.ws_bytes,
                dtype_code(feats[0].dtype), stream())
        return st
    _launch('detect_loss_fwd', 'mgdt_detect_loss_fwd', arr, sarr, len(feats), reg_max, nc, ptr(st.gt) if n_gt else None, n_gt, int(call_count),
            float(gains[0]), float(gains[1]), float(gains[2]), ptr(st.out5), ptr(st.fg), ptr(st.gt_idx), ptr(st.tscore), ptr(st.ws), st.ws_bytes,
            dtype_code(feats[0].dtype), stream())
    return st


def detect_loss_bwd(st, gscale=1.0):
    """d(loss*B)/d feats for the state of a previous detect_loss_fwd; returns a list of NHWC grads."""
    grads = [like(f) for f in st.feats]
    arr, keep = _view_array(st.feats)
    garr, gkeep = _view_array(grads)
    sarr = (C.c_float * len(st.feats))(*st.strides.tolist())
    _launch('detect_loss_bwd', 'mgdt_detect_loss_bwd', arr, garr, sarr, len(st.feats), st.reg_max, st.nc, ptr(st.gt) if st.n_gt else None, st.n_gt,
            float(st.gains[0]), float(st.gains[1]), float(st.gains[2]), float(gscale), ptr(st.out5), ptr(st.ws), st.ws_bytes,
            dtype_code(st.feats[0].dtype), stream())
    return grads


# ------------------------------------------------------------------ training side (BN batch stats, backward kernels)
def _red_ws(c, dev):
    return torch.empty(L.lib().mgdt_reduce_workspace_bytes(c), dtype=torch.uint8, device=dev)


def bn_stats(y, eps, momentum, running_mean=None, running_var=None):
    """Batch mean / rstd (biased variance) of an NHWC map; updates the running stats in place when given."""
    c = y.shape[1]
    mean = torch.empty(c, dtype=torch.float32, device=y.device)
    rstd = torch.empty(c, dtype=torch.float32, device=y.device)
    ws = _red_ws(c, y.device)
    _launch('bn_stats_fwd', 'mgdt_bn_stats_fwd', vp(y), float(eps), float(momentum), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
            ptr(ws), dtype_code(y.dtype), stream())
    return mean, rstd


def bn_act(y, mean, rstd, gamma, beta, act, out=None, r1=None, r2=None):
    out = like(y) if out is None else out
    _same(y, out, r1, r2)
    _launch('bn_act_fwd', 'mgdt_bn_act_fwd', vp(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, vp(r1), vp(r2), vp(out), dtype_code(y.dtype), stream())
    return out


def bn_act_bwd(gz, y, mean, rstd, gamma, beta, act, dgamma=None, dbeta=None):
    """Returns dy (NHWC, same dtype); writes dgamma/dbeta (fp32) when given."""
    dy = like(y)
    _same(gz, y)
    ws = _red_ws(y.shape[1], y.device)
    _launch('bn_act_bwd', 'mgdt_bn_act_bwd', vp(gz), vp(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(dgamma), ptr(dbeta), vp(dy), ptr(ws),
            dtype_code(y.dtype), stream())
    return dy


class _PackedDgrad:
    """Weights of one conv packed for its stride-1 data gradient (mgdt_conv_pack_dgrad); same fields as PackedConv."""
    __slots__ = ('w', 'bias', 'k', 'cin', 'cout', 'dtype', 'direct', 'groups', 'key', 'owner', 'src', 'epoch', '__weakref__')

    def __init__(self, weight, k, dtype, key, phase=-1):
        import weakref
        self.owner = weakref.ref(weight)      # the address may be recycled by another parameter: the entry is valid for THIS tensor only
        cout, cin = weight.shape[0], weight.shape[1]
        code = dtype_code(dtype)
        self.k, self.cin, self.cout, self.dtype, self.direct, self.groups, self.key = k, cout, cin, dtype, False, 1, key   # a conv cout -> cin
        self.w = torch.empty(L.lib().mgdt_conv_packed_bytes(cout, cin, k, code), dtype=torch.uint8, device=weight.device)
        self.bias = torch.empty((cin + 15) // 16 * 16, dtype=torch.float32, device=weight.device)
        wf = weight.detach().float().contiguous()
        L.check(L.lib().mgdt_conv_pack_dgrad(ptr(wf), cin, cout, k, phase, code, ptr(self.w), ptr(self.bias), stream()), 'conv_pack_dgrad')
        if wf.data_ptr() == weight.data_ptr():                      # built from the live parameter storage: refreshable in place
            _register_pack(self, (wf, None, None, None, None, None, 0.0, cin, cout, k, code, 1 if phase < 0 else 2 + phase))


_DGRAD_PK = {}


def conv_dgrad(dy, weight, k, stride, dx, accumulate=False, r2=None):
    """dx (+)= d loss / d x of conv(x, weight).  Stride 1 on NHWC maps: the forward MFMA kernel on the flipped / transposed weights
    (re-packed once per optimizer epoch); otherwise the direct kernel.  `r2`: one more addend shaped like dx (a shortcut's gradient), fused
    into the convolution's epilogue where the MFMA path applies."""
    cout, cin = weight.shape[0], weight.shape[1]
    res = [t for t in (dx if accumulate else None, r2) if t is not None]
    r1_, r2_ = (res + [None, None])[:2]
    if (stride == 1 and (weight.dim() == 2 and k == 1 or weight.dim() == 4 and weight.shape[2] == k) and k in (1, 3) and dx.dtype == dy.dtype
            and is_nhwc(dx) and cin % 4 == 0
            and conv_can_mfma(dy, cout, cin, k, 1, 1, dy.dtype)):
        key = (PARAM_EPOCH[0], weight._version, dy.dtype)
        pk = _DGRAD_PK.get(weight.data_ptr())
        if pk is not None and pk.owner() is weight and pk.key[1:] == key[1:] and pack_is_current(pk):
            pk.key = key                                   # refreshed by repack_all() for this epoch
        if pk is None or pk.owner() is not weight or pk.key != key:
            pk = _DGRAD_PK[weight.data_ptr()] = _PackedDgrad(weight, k, dy.dtype, key)
        return conv2d(dy, pk, 1, ACT_NONE, out=dx, r1=r1_, r2=r2_)
    if (stride == 2 and k == 3 and weight.dim() == 4 and weight.shape[2] == 3 and dx.dtype == dy.dtype and is_nhwc(dx) and cin % 4 == 0
            and dx.shape[2] == 2 * dy.shape[2] and dx.shape[3] == 2 * dy.shape[3] and conv_can_mfma(dy, cout, cin, 3, 1, 1, dy.dtype)):
        # four phases (input-pixel parities), each a 3x3 convolution over dy written to a strided view of dx
        key = (PARAM_EPOCH[0], weight._version, dy.dtype)
        pks = _DGRAD_PK.get((weight.data_ptr(), 2))
        if pks is not None and pks[0].owner() is weight and pks[0].key[1:] == key[1:] and all(pack_is_current(q) for q in pks):
            for q in pks:
                q.key = key
        if pks is None or pks[0].owner() is not weight or pks[0].key != key:
            pks = _DGRAD_PK[(weight.data_ptr(), 2)] = [_PackedDgrad(weight, 3, dy.dtype, key, phase=ph) for ph in range(4)]
        for ph in range(4):
            sub = dx[:, :, ph >> 1::2, ph & 1::2]
            ra = None if r1_ is None else r1_[:, :, ph >> 1::2, ph & 1::2]
            rb = None if r2_ is None else r2_[:, :, ph >> 1::2, ph & 1::2]
            _same(dy, sub, ra, rb)
            if _PROF is not None:
                ntap = (1 + (ph >> 1)) * (1 + (ph & 1))
                _META['conv2d_fwd'] = dict(shape=(dy.shape[0], cout, dy.shape[2], dy.shape[3], cin, 3, 1), flops=2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cin * cout * ntap,
                                           bytes=float(dy.numel() * dy.element_size() + sub.numel() * sub.element_size() + pks[ph].w.numel()))
            _launch('conv2d_fwd', 'mgdt_conv2d_phase_fwd', vp(dy), ptr(pks[ph].w), ptr(pks[ph].bias), ph, vp(ra), vp(rb), vp(sub), dtype_code(dy.dtype), stream())
        return dx
    _same(dy, dx)
    _launch('conv_dgrad', 'mgdt_conv_dgrad', vp(dy), ptr(weight), k, stride, vp(dx), int(accumulate), dtype_code(dy.dtype), stream())
    return dx if r2 is None else add(dx, r2, out=dx)


def gconv_dgrad(dy, weight, k, stride, groups, dx, accumulate=False):
    """Grouped / depth-wise data gradient (DWConv, conv.py:82-86).  weight: [cout][cin/groups][k][k] fp32."""
    _same(dy, dx)
    _launch('conv_dgrad', 'mgdt_gconv_dgrad', vp(dy), ptr(weight), k, stride, groups, vp(dx), int(accumulate), dtype_code(dy.dtype), stream())
    return dx


def gconv_wgrad(x, dy, k, stride, groups, dw, accumulate=False):
    """Grouped / depth-wise weight gradient into dw ([cout][cin/groups][k][k] fp32)."""
    _same(x, dy)
    assert dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == dy.shape[1] * (x.shape[1] // groups) * k * k
    flush_wgrad()                                   # deferred final sums of earlier convolutions read their own workspaces: keep the order simple
    ws = torch.empty(L.lib().mgdt_gconv_wgrad_workspace_bytes(x.shape[1], dy.shape[1], k, groups), dtype=torch.uint8, device=x.device)
    _launch('conv_wgrad', 'mgdt_gconv_wgrad', vp(x), vp(dy), k, stride, groups, ptr(dw), int(accumulate), ptr(ws), dtype_code(dy.dtype), stream())
    return dw


def conv_wgrad(x, dy, k, stride, dw, dbias=None, x2=None, accumulate=False):
    lib = L.lib()
    if (not is_nhwc(x) or x.shape[1] % 4) and x.dtype in (torch.float32, torch.uint8, torch.bfloat16) and x2 is None and not accumulate and x.shape[1] < 4:
        # the 3-channel image (NCHW): one strided copy into a 4-channel NHWC buffer (4th channel zero) puts the stem on the tiled NHWC kernel;
        # the generic kernel would scan every pixel once per weight element
        b, c, h, w = x.shape
        x4 = new_act(b, 4, h, w, dy.dtype, x.device)                   # in dy's dtype; a uint8 image is divided by 255 on the way (detect/train.py:64)
        _launch('copy_fwd', 'mgdt_image_pad4_fwd', vp(x), U8 if x.dtype == torch.uint8 else dtype_code(x.dtype), vp(x4), dtype_code(x4.dtype), stream())
        dw4 = torch.empty((dw.shape[0], 4, k, k), dtype=torch.float32, device=x.device)
        conv_wgrad(x4, dy, k, stride, dw4, dbias=dbias)
        flush_wgrad()                                  # dw4 is read right below
        copy(dw4[:, :c], dw)
        return
    if is_nhwc(x) and x.shape[1] % 4 == 0 and dy.shape[1] % 4 == 0:
        _same(x, dy, x2)
    elif x.dtype != torch.float32 or dy.dtype not in (torch.float32, torch.bfloat16):
        # the generic kernel (any layout, any channel count) reads x as fp32 whatever dy is: any other dtype would be an out-of-range read
        raise RuntimeError(f'conv_wgrad: the generic (non-NHWC / channels % 4 != 0) path takes an fp32 input, got {x.dtype} (dy {dy.dtype})')
    ws = torch.empty(lib.mgdt_conv_wgrad_workspace_bytes(x.shape[1], dy.shape[1], k), dtype=torch.uint8, device=x.device)
    if _PROF is not None:
        b, ci, h, w = x.shape
        _META['conv_wgrad'] = dict(shape=(b, ci, h, w, dy.shape[1], k, stride), flops=2.0 * dy.numel() * ci * k * k,
                                   bytes=float(x.numel() * x.element_size() + dy.numel() * dy.element_size()))
    if _WGRAD_DEFER[0] > 0 and accumulate:
        flush_wgrad()                                  # an accumulating gradient must see the value the pending jobs will write
    defer = (_WGRAD_DEFER[0] > 0 and dbias is None and not accumulate and is_nhwc(x) and x.shape[1] % 4 == 0 and dy.shape[1] % 4 == 0 and dw.is_contiguous())
    _launch('conv_wgrad', 'mgdt_conv_wgrad', vp(x), vp(x2), vp(dy), k, stride, None if defer else ptr(dw), ptr(dbias), int(accumulate), ptr(ws), dtype_code(x.dtype), stream())
    if defer:
        _WGRAD_PENDING.append((ws, dw, dw.numel(), lib.mgdt_conv_wgrad_splits(x.shape[1], dy.shape[1], k)))


# Deferred final sums of the weight gradients: inside `with defer_wgrad():` (the model's reverse pass) a convolution leaves its per-split partial sums
# in its workspace and `flush_wgrad()` adds them up for all pending convolutions in one launch (bit-identical to the immediate form).
_WGRAD_DEFER = [0]
_WGRAD_PENDING = []


class defer_wgrad:
    def __enter__(self):
        _WGRAD_DEFER[0] += 1

    def __exit__(self, *exc):
        _WGRAD_DEFER[0] -= 1
        if _WGRAD_DEFER[0] == 0:
            flush_wgrad()
        return False


def flush_wgrad():
    if not _WGRAD_PENDING:
        return
    arr = (L.WgradFinalDesc * len(_WGRAD_PENDING))()
    for d, (ws, dw, n, nsplit) in zip(arr, _WGRAD_PENDING):
        d.partial, d.dw, d.n, d.nsplit, d.accumulate = ptr(ws), ptr(dw), n, nsplit, 0
    _launch('wgrad_final_batch', 'mgdt_wgrad_final_batch', arr, len(_WGRAD_PENDING), stream())
    _WGRAD_PENDING.clear()


def add(a, b, out=None):
    out = like(a) if out is None else out
    _same(a, b, out)
    _launch('add_fwd', 'mgdt_add_fwd', vp(a), vp(b), vp(out), dtype_code(a.dtype), stream())
    return out


def maxpool5_bwd(x, gy):
    """Adjoint of MaxPool2d(5,1,2): dense NHWC gradient (B,C,H,W channels_last) in the dtype of gy (computed in fp32)."""
    _same(x, gy)
    b, c, h, w = x.shape
    gx = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
    _launch('maxpool5_bwd', 'mgdt_maxpool5_bwd', vp(x), vp(gy), ptr(gx), dtype_code(x.dtype), stream())
    return gx if gy.dtype == torch.float32 else copy(gx, like(gy))


def nearest_bwd(gy, gx):
    _same(gy, gx)
    _launch('nearest_bwd', 'mgdt_nearest_bwd', vp(gy), vp(gx), dtype_code(gy.dtype), stream())
    return gx


# ------------------------------------------------------------------ adjoints of the MSPA / GD-neck ops
EW_MUL, EW_HSIG_GRAD, EW_MUL_HSIG, EW_HSIG = 0, 1, 2, 3


def ew(a, b, mode, out=None):
    out = like(a) if out is None else out
    _same(a, b, out)
    _launch('ew_binary', 'mgdt_ew_binary', vp(a), vp(b), vp(out), mode, dtype_code(a.dtype), stream())
    return out


def channel_affine(x, scale, shift, out=None):
    out = like(x) if out is None else out
    _same(x, out)
    _launch('channel_affine', 'mgdt_channel_affine', vp(x), ptr(scale), ptr(shift), vp(out), dtype_code(x.dtype), stream())
    return out


def nc_reduce(a, b=None):
    """sum over (h, w) of a*b (or a) per (image, channel) -> fp32 [B, C]."""
    _same(a, b)
    n, c = a.shape[:2]
    out = torch.empty(n, c, dtype=torch.float32, device=a.device)
    ws = torch.empty(L.lib().mgdt_nc_reduce_workspace_bytes(n, c), dtype=torch.uint8, device=a.device)
    _launch('nc_reduce', 'mgdt_nc_reduce', vp(a), vp(b), ptr(out), ptr(ws), dtype_code(a.dtype), stream())
    return out


def adaptive_avgpool_bwd(gy, gx, accumulate=False):
    _same(gy, gx)
    _launch('adaptive_avgpool_bwd', 'mgdt_adaptive_avgpool_bwd', vp(gy), vp(gx), int(accumulate), dtype_code(gy.dtype), stream())
    return gx


def bilinear_bwd(gy, gx, accumulate=False):
    _same(gy, gx)
    _launch('bilinear_bwd', 'mgdt_bilinear_bwd', vp(gy), vp(gx), int(accumulate), dtype_code(gy.dtype), stream())
    return gx


def spr_attention_train(x, fc1_w, fc1_b, fc2_w, fc2_b, groups):
    """Like spr_attention but also returns the pooled partial sums needed by the backward."""
    b, c, h, w = x.shape
    part = torch.empty(b * L.SPR_SPLITS * c * 5, dtype=torch.float32, device=x.device)
    _launch('spr_pool_fwd', 'mgdt_spr_pool_fwd', vp(x), ptr(part), dtype_code(x.dtype), stream())
    attn = torch.empty(b, c, dtype=torch.float32, device=x.device)
    _launch('spr_attn_fwd', 'mgdt_spr_attn_fwd', ptr(part), ptr(fc1_w), ptr(fc1_b), ptr(fc2_w), ptr(fc2_b), b, c, groups, h, w, 1, ptr(attn), stream())
    return attn, part


def spr_bwd(gy, part, attn, dattn, fc1_w, fc1_b, fc2_w, fc2_b, groups, out=None):
    """Returns (grad w.r.t. the pre-scale map, flat fp32 param grads [dW1 | db1 | dW2 | db2]); `out`: where to write the latter."""
    b, c = gy.shape[:2]
    lib = L.lib()
    cw = c // groups
    hid = cw // 4
    n_pg = hid * 5 * cw + hid + cw * hid + cw
    assert out is None or (out.numel() == n_pg and out.dtype == torch.float32 and out.is_contiguous())
    pg = torch.empty(n_pg, dtype=torch.float32, device=gy.device) if out is None else out
    ws = torch.empty(lib.mgdt_spr_bwd_workspace_bytes(b, c, groups), dtype=torch.uint8, device=gy.device)
    gx = like(gy)
    _launch('spr_bwd', 'mgdt_spr_bwd', vp(gy), ptr(part), L.SPR_SPLITS, ptr(attn), ptr(dattn), ptr(fc1_w), ptr(fc1_b), ptr(fc2_w), ptr(fc2_b), groups,
            vp(gx), ptr(pg), ptr(ws), dtype_code(gy.dtype), stream())
    return gx, pg


def dwconv7_ln_train(x, dw_w49c, dw_b, ln_w, ln_b, eps):
    y, u = like(x), like(x)
    _launch('dwconv7_ln_train_fwd', 'mgdt_dwconv7_ln_train_fwd', vp(x), ptr(dw_w49c), ptr(dw_b), ptr(ln_w), ptr(ln_b), eps, vp(y), vp(u),
            dtype_code(x.dtype), stream())
    return y, u


def dwconv7_ln_bwd(x, u, gy, dw_w49c, ln_w, eps, d_dw_w, d_dw_b, d_ln_w, d_ln_b):
    c = x.shape[1]
    dx, tmp = like(x), like(x)
    _same(x, u, gy)
    ws = torch.empty(L.lib().mgdt_dwconv7_ln_bwd_workspace_bytes(c), dtype=torch.uint8, device=x.device)
    _launch('dwconv7_ln_bwd', 'mgdt_dwconv7_ln_bwd', vp(x), vp(u), vp(gy), ptr(dw_w49c), ptr(ln_w), eps, vp(tmp), vp(dx), 0, ptr(d_dw_w), ptr(d_dw_b),
            ptr(d_ln_w), ptr(d_ln_b), ptr(ws), dtype_code(x.dtype), stream())
    return dx


def grn_bwd(g, t, S, A, B, gamma, dgamma, dbeta):
    n, c = t.shape[:2]
    dt = like(t)
    _same(g, t)
    ws = torch.empty(3 * n * c, dtype=torch.float32, device=t.device)
    _launch('grn_bwd', 'mgdt_grn_bwd', vp(g), vp(t), ptr(S), ptr(A), ptr(B), ptr(gamma), vp(dt), ptr(dgamma), ptr(dbeta), ptr(ws), dtype_code(t.dtype), stream())
    return dt


# ------------------------------------------------------------------ optimizer on the flat buffers
def grad_clip_coef(flat_grad, max_norm):
    """fp32[2] on device: {total gradient norm, clip coefficient} (no host sync)."""
    out = torch.empty(2, dtype=torch.float32, device=flat_grad.device)
    ws = torch.empty(L.lib().mgdt_grad_norm_workspace_bytes(), dtype=torch.uint8, device=flat_grad.device)
    _launch('grad_clip_coef', 'mgdt_grad_clip_coef', ptr(flat_grad), flat_grad.numel(), float(max_norm), ptr(out), ptr(ws), stream())
    return out


# Parameters updated by the HIP optimizer kernels change behind torch's back (no tensor version bump): every cache of packed / folded
# weights carries this epoch in its key and is rebuilt after an optimizer step.
PARAM_EPOCH = [0]


def sgd_step(p, g, buf, wd, lr, momentum, nesterov, first, clip=None, lr_bias=None):
    PARAM_EPOCH[0] += 1
    _launch('sgd_step', 'mgdt_sgd_step', ptr(p), ptr(g), ptr(buf), ptr(wd), p.numel(), float(lr), float(lr if lr_bias is None else lr_bias), float(momentum),
            int(nesterov), int(first), ptr(clip), stream())


def sgd_ema_step_dev(p, g, buf, wd, ema, data, hyper, nesterov, first, clip=None):
    """SGD over the parameters `p` (= data[:p.numel()]) and EMA over all of `data`, scalars {lr, lr_bias, momentum, ema_decay} from the device
    tensor `hyper` (fp32[4]): the form a captured training step uses."""
    PARAM_EPOCH[0] += 1
    _launch('sgd_ema_step_dev', 'mgdt_sgd_ema_step_dev', ptr(p), ptr(g), ptr(buf), ptr(wd), p.numel(), ptr(ema), data.numel(), ptr(hyper),
            int(nesterov), int(first), ptr(clip), stream())


def ema_update(ema, p, decay):
    _launch('ema_update', 'mgdt_ema_update', ptr(ema), ptr(p), p.numel(), float(decay), stream())
