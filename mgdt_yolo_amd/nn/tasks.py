"""Model graph / registry layer (reference: nn/tasks.py): `DetectionModel`, `parse_model`, `yaml_model_load`.

Same call surface as the reference's `tasks.DetectionModel` so the engine side (trainer / validator / predictor
counterparts) drives it as a drop-in:
  model(x: Tensor)  -> (y, feats) in eval, feats in train;  model(batch: dict) -> (loss*B, loss_items[3])
  .yaml .names .nc .args .stride .model (nn.Sequential) .save .inplace, .fuse() .info() .load() .init_criterion()
All per-layer compute runs in libmgdt_hip.so; there is no CPU or ATen fallback (CPU tensors raise).
"""
import ast
import contextlib
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from .. import ops
from ..yolo.utils.torch_utils import fuse_conv_and_bn, initialize_weights, intersect_dicts, make_divisible
from .modules import (C2f, IFM, MSPA_C2f, SPPF, Bottleneck, Concat, Conv, Detect, DWConv, InjectionMultiSum_Auto_pool,
                      SimFusion_3in, SimFusion_4in, TOODHead, Upsample)

# names a YAML row may use -> class (the reference resolves them with globals()[m] / getattr(torch.nn, ...), tasks.py:630)
REGISTRY = {c.__name__: c for c in (Conv, DWConv, Concat, Bottleneck, C2f, MSPA_C2f, SPPF, SimFusion_4in, SimFusion_3in, IFM,
                                    InjectionMultiSum_Auto_pool, Detect, TOODHead)}
REGISTRY['nn.Upsample'] = Upsample


class _TrainForwardFn(torch.autograd.Function):
    """Makes the explicit HIP reverse pass reachable from `loss.backward()` - the reference's training call sequence
    `self.loss, self.loss_items = self.model(batch)` -> `self.scaler.scale(self.loss).backward()` (yolo/engine/trainer.py:334-343).

    forward: the train-mode layer loop (`_predict_once`) with every module keeping its backward context; the raw head maps come out as
    autograd outputs.  backward: `BaseModel.backward(head_grads)` launches the adjoint kernels and writes each parameter's `.grad`
    directly (into the trainer's flat gradient buffer when there is one), so nothing is returned to autograd for the parameters - the
    anchor input only exists to make autograd call us.  Gradient w.r.t. the image is not produced (the reference never asks for it)."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        with ops.force_ctx():
            feats = model._predict_once(x)
        ctx.model = model
        ctx.n = len(feats)
        return tuple(feats)

    @staticmethod
    def backward(ctx, *gfeats):
        model = ctx.model
        gfeats = [g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last) for g in gfeats]
        acc = model._accumulate_snapshot() if getattr(model, 'grad_accumulate', False) else None
        model.backward(gfeats)
        if acc is not None:
            model._accumulate_restore(acc)
        return None, None, None


class BaseModel(nn.Module):
    """Reference: nn/tasks.py BaseModel (:28-216)."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):   # training / validating-while-training: batch dict -> loss (tasks.py:44-45)
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False):
        if augment or profile or visualize:
            raise RuntimeError('augment / profile / visualize are host-side tooling outside the hot path')
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.model[0].parameters()):
            # training call form of the reference: the outputs carry a grad_fn, `loss.backward()` runs the HIP reverse pass
            if not hasattr(self, '_anchor'):
                self._anchor = torch.zeros((), device=x.device, requires_grad=True)
            return list(_TrainForwardFn.apply(self, x, self._anchor))
        return self._predict_once(x)

    def _predict_once(self, x, profile=False, visualize=False):
        """Per-layer dispatch with the save-list (tasks.py:65-87)."""
        if not x.is_cuda:
            raise RuntimeError('mgdt_yolo_amd runs on MI355X (HIP) only: move the model and the input to cuda (no CPU fallback)')
        if self.training:       # contexts of a forward that was never followed by a backward (validation under train(), an aborted step) go now
            for m in self.modules():
                m.__dict__.pop('_ctx', None)
        y = []
        layers = self.model
        if not self.training and self._stem_fusable(x):
            # layers 0 and 1 (two stride-2 3x3 Convs) in one launch: layer 0's map never reaches HBM (mgdt_stem2_fwd)
            m0, m1 = self.model[0], self.model[1]
            pk0 = m0._cached(('stem2',), m0.affine_tensors(), lambda: ops.PackedStem2(m0.conv.weight, m0.folded_bn_like()))
            x = ops.stem2(x, pk0, m1.packed(torch.bfloat16, direct=False))
            y = [None, x if 1 in self.save else None]
            layers = list(self.model)[2:]
        plan = self._neck_plan(x) if not self.training else None
        bufs = {}                                           # consumer layer -> {'out': concat buffer, 'done': slots, 'pooled0': ...}
        head = None                                         # (injection module, its inputs) waiting for the C2f block behind it
        side = self._side_branch() if not self.training else None
        first = layers[0].i
        y += [None] * (len(self.model) - len(y))
        prev = {first - 1: x}                               # outputs waiting for the layer behind them (f = -1)

        def run(m):
            nonlocal head
            src = lambda j: prev[m.i - 1] if j == -1 else y[m.i + j if j < 0 else j]
            xin = src(m.f) if isinstance(m.f, int) else [src(j) for j in m.f]
            if head is not None:                            # the injection in front of this C2f runs inside this block's first launch
                out, head = m(None, head=head), None
            elif self._injection_feeds_next(m):
                head, out = (m, xin), None
            elif plan is not None and m.i in plan['producers']:
                out = m(xin, deliver=self._neck_deliveries(plan, m, xin, bufs))
            elif plan is not None and m.i in plan['consumers']:
                out = m(xin, pre=bufs.get(m.i))
            else:
                out = m(xin)
            prev.pop(m.i - 1, None)
            prev[m.i] = out
            y[m.i] = out if m.i in self.save else None

        if side is None or side[2] < first:
            for m in layers:
                run(m)
            return prev[len(self.model) - 1]
        s0, s1, dep = side
        main, other = torch.cuda.current_stream(), ops.side_stream(x.device)
        for m in layers:
            if s0 <= m.i <= s1:
                continue                                    # ran on the side stream, right behind layer `dep`
            if m.i == s1 + 1:
                main.wait_stream(other)
            run(m)
            if m.i == dep:
                other.wait_stream(main)
                with torch.cuda.stream(other):
                    for k in range(s0, s1 + 1):
                        run(self.model[k])
        if s1 + 1 >= len(self.model):
            main.wait_stream(other)
        return prev[len(self.model) - 1]

    def _side_branch(self):
        """(first, last, dep): layers first..last read nothing newer than layer `dep` < first - 1, i.e. they do not depend on layers dep+1 ..
        first-1 and can run beside them.  In the MSPA-GD graphs that is the high-level branch's local input (Conv on P4 + SimFusion_3in: four
        launches that fill a quarter of the chip each) next to the end of the backbone, the low-level SimFusion_4in and the IFM: they go to a second
        HIP stream right behind layer `dep` (a parallel branch of the captured graph) and are joined in front of layer last+1.  Static per
        model; None when the layer list has no such run, when hooks watch the layers, or when ops.SIDE_STREAM is off."""
        if not ops.SIDE_STREAM:
            return None
        if '_side_branch_cache' not in self.__dict__:
            absf = lambda m: [m.i + f if f < 0 else f for f in ([m.f] if isinstance(m.f, int) else m.f)]
            best = None
            n = len(self.model)
            for s0 in range(2, n - 1):
                deps, s1 = set(), s0 - 1
                for k in range(s0, n - 1):                   # the last layer (the head) stays on the main stream
                    ext = {j for j in absf(self.model[k]) if j < s0}
                    if ext and max(ext) >= s0 - 1:
                        break
                    deps |= ext
                    s1 = k
                if s1 < s0 or not deps:
                    continue
                dep = max(deps)
                if best is None or (s0 - 1 - dep, s1 - s0) > (best[0] - 1 - best[2], best[1] - best[0]):
                    best = (s0, s1, dep)
            self.__dict__['_side_branch_cache'] = best
        best = self.__dict__['_side_branch_cache']
        if best is None or any(self.model[k]._forward_hooks for k in range(best[2] + 1, best[1] + 1)):
            return None
        return best

    def _injection_feeds_next(self, m):
        """layer m is an InjectionMultiSum_Auto_pool whose only consumer is the C2f right behind it (bf16 inference, no hooks): the pair runs
        as injection + C2f.cv1 in one launch, then the block kernel (mgdt_conv1x1_inject_conv_fwd)."""
        if (self.training or not ops.FUSED_INJECT_CONV or getattr(self, 'compute_dtype', None) != torch.bfloat16 or not isinstance(m, InjectionMultiSum_Auto_pool)
                or m.i in self.save or m.i + 1 >= len(self.model)):
            return False
        nxt = self.model[m.i + 1]
        return isinstance(nxt, C2f) and nxt.f == -1 and not m._forward_hooks and not nxt._forward_hooks and len(nxt.m) >= 1

    # -- GD-neck data movement folded into the producers ------------------------------------
    def _neck_plan(self, x):
        """Which MSPA_C2f layers can hand their output to a SimFusion_4in / SimFusion_3in consumer without a launch of its own
        (nn/modules/block.py:289-329): an avg-pooled input becomes a pooled copy written by the block's attention-scaling launch, an identity
        input becomes that launch writing straight into the consumer's concat slot.  Static per model: {'producers': {layer: [(kind, consumer,
        slot)]}, 'consumers': {layers}}.  bf16 inference only, and nobody watching the layers involved (forward hooks)."""
        if not ops.FUSED_NECK or getattr(self, 'compute_dtype', None) != torch.bfloat16 or not hasattr(self, '_reductions'):
            return None
        plan = self.__dict__.get('_neck_plan_cache')
        if plan is None:
            prod, cons = {}, set()
            absf = lambda m: [m.i + f if f < 0 else f for f in ([m.f] if isinstance(m.f, int) else m.f)]
            for m in self.model:
                if isinstance(m, SimFusion_4in):
                    src = absf(m)
                    kinds = ('pool', 'pool', 'main')
                elif isinstance(m, SimFusion_3in):
                    src = absf(m)
                    kinds = ('pool', 'main' if isinstance(m.cv2, nn.Identity) else None)
                else:
                    continue
                for slot, kind in enumerate(kinds):
                    j = src[slot]
                    if kind is None or not isinstance(self.model[j], MSPA_C2f):
                        continue
                    if kind == 'main' and any(k == 'main' for k, _, _ in prod.get(j, [])):
                        continue                             # a block output can live in one concat buffer only
                    prod.setdefault(j, []).append((kind, m.i, slot))
                    cons.add(m.i)
            plan = self.__dict__['_neck_plan_cache'] = {'producers': prod, 'consumers': cons} if prod else False
        if not plan:
            return None
        if any(self.model[i]._forward_hooks for i in list(plan['producers']) + list(plan['consumers'])):
            return None
        return plan

    def _neck_deliveries(self, plan, m, x, bufs):
        """Views the MSPA block `m` (input x: its output has the same shape) writes for its SimFusion consumers; allocates their buffers."""
        b, c, h, w = x.shape
        out, pools = None, []
        red = self._reductions
        absf = lambda mm: [mm.i + f if f < 0 else f for f in ([mm.f] if isinstance(mm.f, int) else mm.f)]
        for kind, ci, slot in plan['producers'][m.i]:
            cm = self.model[ci]
            src = absf(cm)
            tgt = src[2] if isinstance(cm, SimFusion_4in) else src[1]         # the input whose size the module resamples to (block.py:294,316)
            if red[tgt] % red[m.i]:
                continue
            F = int(red[tgt] // red[m.i])
            if F < 1 or h % F or w % F or (kind == 'main' and F != 1) or (kind == 'pool' and F == 1):
                continue
            th, tw = h // F, w // F
            if isinstance(cm, SimFusion_4in):
                cs = [self.model[j].c2 for j in src]
                total, off = sum(cs), sum(cs[:slot])
            else:
                oc = cm.cv_fuse.conv.out_channels
                total, off = 3 * oc, slot * oc
            st = bufs.setdefault(ci, {'out': None, 'done': set()})
            if st['out'] is None:
                st['out'] = ops.new_act(b, total, th, tw, x.dtype, x.device)
            if tuple(st['out'].shape) != (b, total, th, tw):
                continue
            if isinstance(cm, SimFusion_3in) and slot == 0 and not isinstance(cm.cv1, nn.Identity):
                st['pooled0'] = ops.new_act(b, c, th, tw, x.dtype, x.device)      # the pooled map feeds cv1, not the concat buffer
                pools.append(st['pooled0'])
                continue
            view = st['out'][:, off:off + c]
            if kind == 'pool':
                pools.append(view)
            else:
                out = view
            st['done'].add(slot)
        return {'out': out, 'pools': pools[:2]} if (out is not None or pools) else None

    def _stem_fusable(self, x):
        """layers 0, 1 = Conv(3, 16, 3, 2) -> Conv(16, 32, 3, 2) with BN + SiLU, bf16 compute, layer 0's output used by layer 1 only, and nobody
        watching the individual layers (forward hooks)"""
        if not ops.FUSED_STEM or getattr(self, 'compute_dtype', None) != torch.bfloat16 or len(self.model) < 3 or 0 in self.save:
            return False
        if x.dtype not in (torch.bfloat16, torch.float32, torch.uint8) or x.dim() != 4 or x.shape[1] != 3:
            return False
        m0, m1 = self.model[0], self.model[1]
        for m, (ci, co) in ((m0, (3, 16)), (m1, (16, 32))):
            if not (isinstance(m, Conv) and not isinstance(m, DWConv) and m.plain_affine() and isinstance(m.act, nn.SiLU) and m.f == -1 and not m._forward_hooks
                    and m.conv.kernel_size == (3, 3) and m.conv.stride == (2, 2) and m.conv.padding == (1, 1) and m.conv.groups == 1
                    and m.conv.in_channels == ci and m.conv.out_channels == co):
                return False
        return True

    def backward(self, head_grads, layer_done=None):
        """Explicit reverse pass over the layer list (the counterpart of `_predict_once`; replaces torch.autograd on the
        hot path): `head_grads` = d loss / d raw head maps (list, one per level).  Every module's `backward` launches its HIP
        adjoint kernels and fills `.grad` of its parameters (overwrite semantics); gradients of tensors with several
        consumers (the save-list) are summed with the HIP add kernel.  `layer_done(i)` is called after layer i's kernels are queued (the
        trainer hangs its bucketed gradient all-reduce on it)."""
        with ops.defer_wgrad():          # final sums of all weight gradients in one launch at the end (or at the all-reduce bucket boundaries)
            self._backward_layers(head_grads, layer_done)

    def _backward_layers(self, head_grads, layer_done):
        n = len(self.model)
        pend = {n - 1: head_grads}
        for m in reversed(list(self.model)):
            g = pend.pop(m.i, None)
            if g is None:
                if layer_done is not None:
                    layer_done(m.i)
                continue
            if not hasattr(m, 'backward'):
                raise NotImplementedError(f'{type(m).__name__}.backward is not built yet (training path of this module: next)')
            first = m.i == 0
            gin = m.backward(g, need_dx=False) if (first and isinstance(m, Conv)) else m.backward(g)
            srcs = [m.f] if isinstance(m.f, int) else list(m.f)
            gins = [gin] if isinstance(m.f, int) else list(gin)
            for f, gi in zip(srcs, gins):
                if gi is None:
                    continue
                j = m.i - 1 if f == -1 else (f if f >= 0 else m.i + f)
                if j < 0:
                    continue                      # gradient w.r.t. the image: not needed
                if j in pend:
                    dst = pend[j] if (pend[j].is_contiguous(memory_format=torch.channels_last) and pend[j].shape == gi.shape) else None
                    pend[j] = ops.add(pend[j], gi, out=dst) if dst is not None else ops.add(pend[j], gi)
                else:
                    pend[j] = gi
            if layer_done is not None:
                layer_done(m.i)

    # gradient accumulation over micro-batches (trainer.py:250,345): the adjoint kernels overwrite `.grad`, so the running sum is kept aside
    def _accumulate_snapshot(self):
        fs = getattr(self, '_flat_state', None)
        if fs is not None:                                  # the trainer's flat gradient buffer: one copy
            return fs.grad.clone()
        return [(p, p.grad.clone()) for p in self.parameters() if p.requires_grad and p.grad is not None]

    def _accumulate_restore(self, snap):
        if torch.is_tensor(snap):
            self._flat_state.grad.add_(snap)
            return
        for p, g in snap:
            p.grad.add_(g)

    def fuse(self, verbose=True):
        """Fold BatchNorm into the conv parameters and drop the bn modules (tasks.py:121-146)."""
        if not self.is_fused():
            for m in self.model.modules():
                if isinstance(m, (Conv, DWConv)) and hasattr(m, 'bn'):
                    m.conv = fuse_conv_and_bn(m.conv, m.bn)
                    delattr(m, 'bn')
                    m.forward = m.forward_fuse
                    m.__dict__.pop('_pk', None)
        return self

    def is_fused(self, thresh=10):
        bn = tuple(v for k, v in nn.__dict__.items() if 'Norm' in k)
        return sum(isinstance(v, bn) for v in self.modules()) < thresh

    def info(self, detailed=False, verbose=True, imgsz=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_l = len(list(self.modules()))
        if verbose:
            print(f'{Path(self.yaml.get("yaml_file", "model")).stem} summary: {n_l} layers, {n_p} parameters')
        return n_l, n_p

    def _apply(self, fn):
        """Move the head's stride / anchor tensors with the module (tasks.py:171-188)."""
        self = super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            m.anchors = fn(m.anchors)
            m.strides = fn(m.strides)
        return self

    def load(self, weights, verbose=True):
        """Transfer name+shape matching entries of a checkpoint / module / state_dict (tasks.py:190-202)."""
        model = weights['model'] if isinstance(weights, dict) and 'model' in weights else weights
        csd = {k: v.float() for k, v in (model.state_dict() if isinstance(model, nn.Module) else model).items()}
        csd = intersect_dicts(csd, self.state_dict())
        self.load_state_dict(csd, strict=False)
        if verbose:
            print(f'Transferred {len(csd)}/{len(self.model.state_dict())} items from pretrained weights')

    def loss(self, batch, preds=None):
        if not hasattr(self, 'criterion'):
            self.criterion = self.init_criterion()
        preds = self.forward(batch['img']) if preds is None else preds
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError('compute_loss() needs to be implemented by task heads')

    # -- precision switch of the reference's call sites (autobackend.py:99, trainer.py:420): parameters stay fp32 masters, the COMPUTE dtype moves
    def half(self):
        """`model.half()`: the reduced-precision path of this hardware - bfloat16 operands on MFMA, fp32 accumulation."""
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        """`model.float()`: the exact fp32 path (the one the 1e-3 box contract is stated for)."""
        return self.set_compute_dtype(torch.float32)

    # -- MI355X-specific knobs (not in the reference) ------------------------------------------------------
    def quantize_fp8(self, calib, headroom=2.0, exclude=(), percentile=None):
        """fp8 inference (BASELINE configs[4]; the reference has no counterpart - trainer.py:223 is fp16 autocast only).  Every convolution
        that runs on the implicit-GEMM kernel (`Conv.run` on the bf16 MFMA path) switches to e4m3 operands: weights with per-output-channel
        scales (mgdt_conv_pack_fp8), activations with one power-of-two multiplier per convolution, chosen so that `headroom` x the largest
        |input| seen on the calibration images `calib` (one batch or a list of batches) lands at the top of the e4m3 range (448).  Activations
        stay bf16 in HBM; the block kernels (stem, CSP / MSPA blocks, ConvNeXt MLP, injection, detect tail) stay bf16.  `exclude`: substrings of
        module names (as in `named_modules()`, e.g. 'model.22.' = the Detect head) whose convolutions keep bf16 operands - the usual mixed-precision
        policy when the last layers decide the score ranking.  `percentile` (e.g. 99.99): the activation range is that percentile of |input| instead
        of its maximum (values beyond saturate at +-448) - with `headroom=1.0` this spends the e4m3 range on the bulk of the distribution.
        Returns the {module name: multiplier} table.  `dequantize_fp8()` restores the bf16 path."""
        import math
        self.set_compute_dtype(torch.bfloat16)
        was_training = self.training
        self.eval()
        self.dequantize_fp8()
        stats = {}
        ops.Q8_CALIB, ops.Q8_CALIB_PCT = stats, percentile
        try:
            with torch.no_grad():
                for xb in (calib if isinstance(calib, (list, tuple)) else [calib]):
                    self._predict_once(xb)
        finally:
            ops.Q8_CALIB, ops.Q8_CALIB_PCT = None, None
            self.train(was_training)
        names = {m: n for n, m in self.named_modules()}
        table = {}
        for (m, key), amax in stats.items():
            if any(e in names.get(m, '?') + '.' for e in exclude):
                continue
            q = 1.0 if not (amax > 0.0 and math.isfinite(amax)) else 2.0 ** math.floor(math.log2(448.0 / (headroom * amax)))
            m.__dict__.setdefault('_q8', {})[key] = q
            table[names.get(m, '?') + ('' if key is None else ':' + '/'.join(str(k) for k in key))] = q
        self.fp8_table = table
        return table

    def dequantize_fp8(self):
        for m in self.modules():
            m.__dict__.pop('_q8', None)
        self.fp8_table = {}
        return self

    def set_compute_dtype(self, dtype):
        """float32 (exact path, 1e-3 box parity) or bfloat16 (throughput path); parameters stay fp32 masters."""
        ops.dtype_code(dtype)
        for m in self.modules():
            if hasattr(m, 'out_dtype'):
                m._cdtype = dtype
        self.compute_dtype = dtype
        return self


class DetectionModel(BaseModel):
    """YOLOv8 detection model (reference tasks.py:222-294)."""

    def __init__(self, cfg='yolov8n.yaml', ch=3, nc=None, verbose=True):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if nc and nc != self.yaml['nc']:
            self.yaml['nc'] = nc
        self.model, self.save, reductions = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self._reductions = list(reductions)          # total down-sampling of every layer's output (the neck plan derives pooling factors from it)
        self.names = {i: f'{i}' for i in range(self.yaml['nc'])}
        self.inplace = self.yaml.get('inplace', True)
        self.compute_dtype = torch.float32
        self.args = None

        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            # The reference probes strides with a 640^2 zero image on the CPU (tasks.py:241-245).  There is no CPU
            # compute here, so the same numbers come from the graph: stride = total down-sampling of each head input.
            m.stride = torch.tensor([float(reductions[j]) for j in m.f])
            self.stride = m.stride
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)
        if verbose:
            self.info()

    def init_criterion(self):
        from ..yolo.utils.loss import v8DetectionLoss
        return v8DetectionLoss(self)


# ---------------------------------------------------------------------------------------------------------------
def parse_model(d, ch, verbose=True):
    """YAML dict -> nn.Sequential + save-list, with the reference's argument-rewriting rules (tasks.py:604-699).
    Also returns each layer's cumulative spatial reduction (for the head strides)."""
    max_channels = float('inf')
    nc, act, scales = (d.get(x) for x in ('nc', 'activation', 'scales'))
    depth, width = (d.get(x, 1.0) for x in ('depth_multiple', 'width_multiple'))
    if scales:
        scale = d.get('scale')
        if not scale:
            scale = tuple(scales.keys())[0]
            print(f"WARNING no model scale passed. Assuming scale='{scale}'.")
        depth, width, max_channels = scales[scale]
    if act:
        Conv.default_act = eval(act)   # noqa: S307 - same contract as the reference (tasks.py:620-621), e.g. 'nn.SiLU()'
    ch = [ch]
    red = [1]       # spatial reduction of each layer output relative to the input image
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, mname, args) in enumerate(d['backbone'] + d['head']):
        if mname not in REGISTRY:
            raise KeyError(mname)   # unknown module name -> KeyError like globals()[m] in the reference
        m = REGISTRY[mname]
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                with contextlib.suppress(ValueError):
                    args[j] = nc if a == 'nc' else ast.literal_eval(a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        r_in = red[f] if isinstance(f, int) else None
        if m in (Conv, DWConv, Bottleneck, SPPF, C2f, MSPA_C2f):
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m in (C2f, MSPA_C2f):
                args.insert(2, n)
                n = 1
            r_out = r_in * (args[3] if m in (Conv, DWConv) and len(args) > 3 else 1)
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
            r_out = red[f[0]]
        elif m in (Detect, TOODHead):      # TOODHead's hidc (args[1]) is passed unscaled, as in the reference (tasks.py:660-665)
            args.append([ch[x] for x in f])
            r_out = red[f[0]]
        elif m is SimFusion_4in:
            c2 = sum(ch[x] for x in f)
            r_out = red[f[2]]
        elif m is SimFusion_3in:
            c2 = args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [[ch[f_] for f_ in f], c2]
            r_out = red[f[1]]
        elif m is IFM:
            c1 = ch[f]
            c2 = sum(args[0])
            args = [c1, *args]
            r_out = r_in
        elif m is InjectionMultiSum_Auto_pool:
            c1 = ch[f[0]]
            c2 = args[0]
            args = [c1, *args]
            r_out = red[f[0]]
        elif m is Upsample:
            c2 = ch[f]
            r_out = r_in / args[1]
        else:
            c2 = ch[f]
            r_out = r_in
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = f'{m.__module__}.{m.__name__}'
        m.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, t
        m_.c2 = None if m in (Detect, TOODHead) else c2          # output channels (the neck plan sizes the SimFusion buffers from them)
        if verbose:
            print(f'{i:>3}{str(f):>20}{n_:>3}{m.np:10.0f}  {t:<45}{str(args):<30}')
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch, red = [], []
        ch.append(c2)
        red.append(r_out)
    return nn.Sequential(*layers), sorted(save), red


def torch_safe_load(weight):
    """Reference: nn/tasks.py:520-547 `torch_safe_load` -> (ckpt dict, path).  The reference unpickles Module objects (and pip-installs missing
    modules on failure); here the file is read WITHOUT importing or running anything it names (nn/checkpoint.py).  The returned dict has the
    reference's keys; 'model' / 'ema' are rebuilt DetectionModel objects (fp32 masters) carrying the stored weights."""
    from . import checkpoint as CK
    tree, stubbed = CK.read_checkpoint(weight)
    if not isinstance(tree, dict):
        raise RuntimeError(f'{weight}: expected the trainer\'s checkpoint dict (trainer.py:413-422)')
    ckpt = {}
    for k, v in tree.items():
        if k in ('model', 'ema') and isinstance(v, CK.Stub):
            st = v.state
            cfg = CK.plain(st.get('yaml'))
            if not isinstance(cfg, dict):
                raise RuntimeError(f'{weight}: the pickled {k} carries no model YAML dict')
            m = DetectionModel(deepcopy(cfg), ch=cfg.get('ch', 3), nc=cfg.get('nc'), verbose=False)
            sd = {kk: vv.float() for kk, vv in CK.module_state_dict(v).items()}
            own = m.state_dict()
            # the reference unpickles the module itself, so it can never load partially: a key of the rebuilt graph that the file lacks, or
            # stores with another shape, is an error here too (derived buffers that every build recomputes are exempt)
            exempt = lambda kk: kk.endswith(('num_batches_tracked', '.anchors', '.strides', 'dfl.conv.weight'))
            missing = [kk for kk in own if kk not in sd and not exempt(kk)]
            mismatched = [f'{kk}: file {tuple(sd[kk].shape)} vs graph {tuple(own[kk].shape)}' for kk in own if kk in sd and sd[kk].shape != own[kk].shape]
            if missing or mismatched:
                raise RuntimeError(f'{weight}: checkpoint does not match the graph its YAML builds - missing {missing[:8]}'
                                   f'{" ..." if len(missing) > 8 else ""}; shape mismatches {mismatched[:8]}')
            m.load_state_dict(intersect_dicts(sd, own), strict=False)
            names = st.get('names')
            if isinstance(names, dict):
                m.names = dict(names)
            args = CK.plain(st.get('args'))
            m.args = args if isinstance(args, dict) else None
            m.ckpt_missing_keys, m.ckpt_stubbed_globals = missing, stubbed
            ckpt[k] = m
        else:
            ckpt[k] = CK.plain(v)
    return ckpt, weight


def attempt_load_one_weight(weight, device=None, inplace=True, fuse=False):
    """Reference: nn/tasks.py:577-601 -> (model, ckpt): the EMA weights when present, eval mode, on `device`, optionally fused."""
    ckpt, weight = torch_safe_load(weight)
    model = ckpt.get('ema') or ckpt['model']
    if not isinstance(model, BaseModel):
        raise RuntimeError(f'{weight}: no model object in the checkpoint')
    if device is not None:
        model = model.to(device)
    model.pt_path = weight
    model.task = 'detect'
    model = model.fuse().eval() if fuse else model.eval()
    return model, ckpt


def attempt_load_weights(weights, device=None, inplace=True, fuse=False):
    """Reference: nn/tasks.py:550-574 (single model; ensembles of several checkpoints are host-side tooling outside the path)."""
    if isinstance(weights, (list, tuple)):
        if len(weights) != 1:
            raise RuntimeError('attempt_load_weights: model ensembles are outside the detection hot path')
        weights = weights[0]
    return attempt_load_one_weight(weights, device, inplace, fuse)[0]


def guess_model_scale(model_path):
    """Scale letter from a file stem such as 'yolov8n' / 'mspa_c2f_gd_yolov8s' (tasks.py:720-735)."""
    with contextlib.suppress(AttributeError):
        return re.search(r'yolov\d+([nslmx])', Path(model_path).stem).group(1)
    return ''


def yaml_model_load(path):
    """Load a model YAML in the reference's schema; '...yolov8n.yaml' resolves to '...yolov8.yaml' + scale n (tasks.py:702-717).
    Names of the built-in graphs (mgdt_yolo_amd.models.CONFIGS) resolve without a file."""
    import yaml
    from ..models import CONFIGS, get_config
    path = Path(path)
    scale = guess_model_scale(path)
    unified = Path(re.sub(r'(\d+)([nslmx])(.+)?$', r'\1\3', str(path)))
    for cand in (unified, path):
        if cand.is_file():
            with open(cand, errors='ignore', encoding='utf-8') as f:
                d = yaml.safe_load(f)
            d['scale'] = scale
            d['yaml_file'] = str(path)
            return d
    if unified.stem in CONFIGS:
        d = get_config(unified.stem, scale or 'n')
        d['scale'] = scale
        d['yaml_file'] = str(path)
        return d
    raise FileNotFoundError(f"'{path}' does not exist")
