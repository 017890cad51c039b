"""Training-step counterpart of the reference's DetectionTrainer hot loop (yolo/engine/trainer.py:314-362,462-470).

Only what touches the device per step lives here; the data pipeline, callbacks, checkpoints and logging of the reference's
BaseTrainer are out of scope.  One step = warm-up interpolation of lr / momentum (trainer.py:317-326) ->
preprocess (uint8 batch handed to the stem as is: /255 is fused into the first conv's loader) -> train-mode forward (HIP) ->
fused assigner+loss (HIP) -> reverse pass (HIP; reachable from `loss.backward()` too) with the flat gradient buffer
all-reduced in layer-ordered buckets while the earlier layers are still running backward (RCCL over xGMI) -> every
`accumulate` batches: clip(10) + SGD(Nesterov, the reference's three parameter groups) + EMA on the flat parameter buffer.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ... import ops, parallel
from ..utils.loss import loss_and_head_grads, v8DetectionLoss


def param_groups(model):
    """The reference's build_optimizer rule (trainer.py:641-649), name -> group: 'bias' anywhere in the full name -> 2 (no decay, bias
    warm-up lr); a parameter of an nn.*Norm* module -> 1 (no decay); everything else -> 0 (decay).  Note what this implies: the custom
    utils.LayerNorm / GRN of the ConvNeXt blocks are plain nn.Modules, so LayerNorm.weight and GRN.gamma / GRN.beta are decayed."""
    bn = tuple(v for k, v in nn.__dict__.items() if 'Norm' in k and isinstance(v, type))
    out = {}
    for module_name, module in model.named_modules():
        for param_name, _ in module.named_parameters(recurse=False):
            fullname = f'{module_name}.{param_name}' if module_name else param_name
            out[fullname] = 2 if 'bias' in fullname else 1 if isinstance(module, bn) else 0
    return out


class FlatState:
    """All trainable parameters (and float buffers) of a model as views into one flat fp32 buffer; grads likewise."""

    def __init__(self, model, weight_decay):
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        bufs = [(n, b) for n, b in model.named_buffers() if b.dtype.is_floating_point and b.numel() > 0 and 'anchors' not in n and 'strides' not in n]
        dev = params[0][1].device
        self.n_param = sum(p.numel() for _, p in params)
        self.n_total = self.n_param + sum(b.numel() for _, b in bufs)
        self.data = torch.empty(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        # per-element weight decay; -1 marks the bias group (no decay, its own warm-up learning rate: mgdt_sgd_step)
        self.wd = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        groups = param_groups(model)
        self.offsets = {}                  # parameter name -> (offset, numel)
        self.layer_end = {}                # top-level layer index i -> end offset of its parameters in the flat buffers
        off = 0
        for n, p in params:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            g = groups[n]
            self.wd[off:off + k] = weight_decay if g == 0 else (-1.0 if g == 2 else 0.0)
            self.offsets[n] = (off, k)
            parts = n.split('.')
            if len(parts) > 2 and parts[0] == 'model' and parts[1].isdigit():
                self.layer_end[int(parts[1])] = off + k
            off += k
        for n, b in bufs:
            k = b.numel()
            self.data[off:off + k].copy_(b.detach().reshape(-1))
            b.data = self.data[off:off + k].view(b.shape)
            off += k
        ops.LIVE_STORAGES.add(self.data.untyped_storage().data_ptr())       # packed panels built from these bytes may be refreshed in place (ops.repack_all)
        self.momentum_buf = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        self.ema = self.data.clone()
        self.steps = 0                     # optimizer steps taken (= ModelEMA.updates)


class BucketedAllReduce:
    """Data-parallel gradient exchange overlapped with the reverse pass.  The flat gradient buffer is laid out in layer order and the
    reverse pass finishes layers last-to-first, so the buffer completes from its tail: as soon as the layers of a bucket are done, that
    contiguous slice is all-reduced asynchronously (RCCL runs it on its own stream behind the kernels already queued) while the earlier
    layers are still running backward.  Messages are a few MB (the whole n model is 5.26 MB), i.e. latency-bound on xGMI, so buckets are few."""

    def __init__(self, state, n_layers, n_buckets=3):
        ends = [state.layer_end.get(i) for i in range(n_layers)]
        last = 0
        for i in range(n_layers):          # layers without parameters inherit the previous end
            ends[i] = last = ends[i] if ends[i] is not None else last
        total = state.n_param
        # cut so that the buckets hold ~equal numbers of elements, at layer boundaries; bucket b covers layers [lo_b, hi_b)
        cuts, target = [], [total * (k + 1) / n_buckets for k in range(n_buckets - 1)]
        for t in target:
            i = min(range(n_layers), key=lambda j: abs(ends[j] - t))
            if ends[i] not in (0, total) and (i + 1) not in cuts:
                cuts.append(i + 1)
        self.bounds = [0] + sorted(cuts) + [n_layers]                     # layer indices
        self.slices = []
        for lo, hi in zip(self.bounds[:-1], self.bounds[1:]):
            a = ends[lo - 1] if lo > 0 else 0
            self.slices.append((lo, a, ends[hi - 1] if hi < n_layers else total))
        self.state, self.work = state, []

    def layer_done(self, i):
        """Called by BaseModel.backward after layer i's adjoint kernels are queued."""
        import torch.distributed as dist
        for lo, a, b in self.slices:
            if lo == i and b > a:
                ops.flush_wgrad()                      # the slice must hold finished weight gradients before it is sent
                self.work.append(dist.all_reduce(self.state.grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        import torch.distributed as dist
        for w in self.work:
            w.wait()
        self.work = []
        self.state.grad.div_(dist.get_world_size())


class DetectionTrainer:
    """Per-step driver: `trainer.step(batch)` -> (loss*B, loss_items[3]).  Hyper-parameters follow yolo/cfg/default.yaml of the fork
    (lr0 0.001, lrf 0.01, momentum 0.937, weight_decay 5e-4, warmup 3 epochs / momentum 0.8 / bias lr 0.1, nbs 64, nesterov SGD; grad
    clip 10.0; EMA decay 0.9999, tau 2000).  `batch_size` is the GLOBAL batch (trainer.py:238,250), `nb` the batches per epoch."""

    def __init__(self, model, lr0=0.001, lrf=0.01, momentum=0.937, weight_decay=5e-4, world_size=1, ema_decay=0.9999, ema_tau=2000.0,
                 batch_size=None, nb=None, epochs=100, nbs=64, warmup_epochs=3.0, warmup_momentum=0.8, warmup_bias_lr=0.1, overlap=True, amp=False,
                 graph=False, graph_split=None):
        self.model = model.train()
        # amp: the reference trains under autocast (trainer.py:223,329: fp16 + GradScaler); on this hardware the reduced-precision training path
        # is bfloat16 activations / gradients with fp32 master weights, fp32 accumulation and fp32 weight gradients - no loss scaling needed
        model.set_compute_dtype(torch.bfloat16 if amp else torch.float32)
        self.amp = bool(amp)
        self.crit = v8DetectionLoss(model)
        self.world_size = world_size
        self.lr0, self.lrf, self.momentum, self.epochs = lr0, lrf, momentum, epochs
        self.warmup_momentum, self.warmup_bias_lr, self.nbs = warmup_momentum, warmup_bias_lr, nbs
        self.batch_size, self.nb = batch_size, nb
        if batch_size is None:                      # no schedule requested: constant lr0 / momentum, every batch is an optimizer step
            self.accumulate, self.nw, wd = 1, -1, weight_decay
        else:
            self.accumulate = max(round(nbs / batch_size), 1)                                    # trainer.py:250
            wd = weight_decay * batch_size * self.accumulate / nbs                               # trainer.py:251
            self.nw = max(round(warmup_epochs * nb), 100)                                        # trainer.py:284
        self.state = FlatState(model, wd)
        model._flat_state = self.state
        self.ema_decay, self.ema_tau = ema_decay, ema_tau
        self.ni, self.last_opt_step, self.lr, self.lr_bias, self.mom = 0, -1, lr0, lr0, momentum
        # graph=True: from the second optimizer step on, the whole step (forward, loss, reverse pass, clip + SGD + EMA: ~500 launches) is one
        # hipGraph replay; lr / momentum / EMA decay / the assigner's call counter live in device memory so that the warm-up schedule and the
        # EMA ramp keep advancing between replays.  accumulate == 1 only (otherwise the eager path runs).  With several ranks (or
        # graph_split=True) the step is captured as TWO graphs - forward/loss/reverse pass, then clip/SGD/EMA/re-pack - with the one
        # flat-gradient all-reduce issued eagerly between the two replays (RCCL is not captured).
        self.graph = bool(graph)
        self.graph_split = parallel.world() > 1 if graph_split is None else bool(graph_split)
        self._graphs, self._pool, self._static = {}, None, None
        self.exchange = None
        if parallel.world() > 1:
            parallel.broadcast_(self.state.data)            # what DDP's constructor does (trainer.py:225): rank 0's parameters AND buffers
            self.state.ema.copy_(self.state.data)
            if overlap:
                self.exchange = BucketedAllReduce(self.state, len(model.model))

    def lf(self, epoch):
        """linear lr schedule (trainer.py:262)"""
        return (1 - epoch / self.epochs) * (1.0 - self.lrf) + self.lrf

    def warmup(self, epoch=0):
        """lr / momentum / accumulate of iteration self.ni (trainer.py:317-326); after warm-up the epoch's scheduled values."""
        base = self.lr0 * self.lf(epoch) if self.batch_size is not None else self.lr0
        self.lr = self.lr_bias = base
        self.mom = self.momentum
        if self.ni <= self.nw:
            xi = [0, self.nw]
            self.accumulate = max(1, np.interp(self.ni, xi, [1, self.nbs / self.batch_size]).round())
            self.lr = float(np.interp(self.ni, xi, [0.0, base]))
            self.lr_bias = float(np.interp(self.ni, xi, [self.warmup_bias_lr, base]))
            self.mom = float(np.interp(self.ni, xi, [self.warmup_momentum, self.momentum]))

    def preprocess_batch(self, batch):
        """detect/train.py:62-65 moves the uint8 batch to the device and computes float()/255; here the uint8 tensor goes to the stem
        kernel, which divides by 255 exactly while loading (no separate pass)."""
        img = batch['img'].to(self.state.data.device, non_blocking=True)
        return img if img.dtype == torch.uint8 else (img.to(torch.bfloat16) if self.amp else img.float())

    def optimizer_step(self):
        """unscale (no loss scaling: fp32 / bf16) -> clip_grad_norm_(10) -> SGD -> zero_grad (implicit: overwrite) -> EMA (trainer.py:462-470)."""
        st = self.state
        clip = ops.grad_clip_coef(st.grad, 10.0)
        ops.sgd_step(st.data[:st.n_param], st.grad, st.momentum_buf, st.wd, self.lr, self.mom, True, st.steps == 0, clip, lr_bias=self.lr_bias)
        st.steps += 1
        d = self.ema_decay * (1 - math.exp(-st.steps / self.ema_tau))      # torch_utils.py:342
        ops.ema_update(st.ema, st.data, d)
        ops.repack_all()                            # the packed panels of this step, refreshed for the next one in a few launches

    # ---- captured step -------------------------------------------------------------------------------------------------------------
    def _graph_ok(self):
        return (self.graph and self.state.steps >= 1 and self.accumulate == 1
                and (self.batch_size is None or self.nbs / self.batch_size <= 1.0))

    def _graph_body(self, part=None):
        """part None: the whole step; 'grad': forward + loss + reverse pass; 'update': clip + SGD + EMA + re-pack."""
        st, S = self.state, self._static
        if part != 'update':
            feats = self.model._predict_once(S['img'])
            ls = ops.detect_loss_fwd(list(feats), self.crit.stride_list, self.crit.reg_max, self.crit.nc, S['gt'], 0,
                                     (self.crit.hyp.box, self.crit.hyp.cls, self.crit.hyp.dfl), call_count_dev=S['calls'])
            grads = ops.detect_loss_bwd(ls, float(self.world_size))
            self.model.backward(grads)
            S['out5'] = ls.out5
            if part == 'grad':
                return
        clip = ops.grad_clip_coef(st.grad, 10.0)
        ops.sgd_ema_step_dev(st.data[:st.n_param], st.grad, st.momentum_buf, st.wd, st.ema, st.data, S['hyper'], True, False, clip)
        S['panels'] = ops.repack_all()              # next step's packed weights, all convolutions in a few launches

    def _graph_step(self, batch):
        st = self.state
        img = self.preprocess_batch(batch)
        b = img.shape[0]
        gt = self.crit.preprocess(batch, b, (img.shape[2], img.shape[3]))
        nmax = max(16, -(-int(gt.shape[1]) // 16) * 16)             # label slots of the captured step (zero rows are padding, loss.py:177-181)
        S = self._static
        if S is None or S['img'].shape != img.shape or S['img'].dtype != img.dtype:
            S = self._static = dict(img=torch.empty_like(img), hyper=torch.empty(4, dtype=torch.float32, device=img.device),
                                    calls=torch.zeros(1, dtype=torch.int32, device=img.device), gts={})
            self._graphs, self._pool = {}, None
        # Every captured graph reads - and, in its trailing re-pack, rewrites - the packed weight panels that were current when it was
        # captured.  They must stay the ONE live panel set: each graph holds strong references to them (the caches only hold the newest
        # object), every replay stamps them with the new optimizer epoch (so the next capture / eager step reuses them instead of packing
        # new ones and freeing these), and if anything else moved the weights without refreshing them the graphs are dropped.
        if any(not ops.pack_is_current(o) for _, _, panels in self._graphs.values() for o in panels):
            self._graphs, self._pool = {}, None
        if nmax not in S['gts']:
            S['gts'][nmax] = torch.zeros(b, nmax, 5, dtype=torch.float32, device=img.device)
        S['gt'] = S['gts'][nmax]
        S['img'].copy_(img, non_blocking=True)
        S['gt'].zero_()
        if gt.shape[1]:
            S['gt'][:, :gt.shape[1]].copy_(gt)
        d = self.ema_decay * (1 - math.exp(-(st.steps + 1) / self.ema_tau))      # torch_utils.py:342 with updates = steps + 1
        S['hyper'].copy_(torch.tensor([self.lr, self.lr_bias, self.mom, d], dtype=torch.float32), non_blocking=True)
        S['calls'].fill_(int(self.crit.epoch))
        captured = nmax not in self._graphs
        if captured:
            gs = []
            torch.cuda.synchronize()
            epoch0 = ops.PARAM_EPOCH[0]
            for part in (('grad', 'update') if self.graph_split else (None,)):
                g = torch.cuda.CUDAGraph()
                # several ranks: the RCCL watchdog thread polls events while we capture, so only this thread's calls are checked
                with torch.cuda.graph(g, pool=self._pool, capture_error_mode='thread_local' if parallel.world() > 1 else 'global'):
                    self._graph_body(part)
                if self._pool is None:
                    self._pool = g.pool()
                gs.append(g)
            assert ops.PARAM_EPOCH[0] == epoch0 + 1        # the capture walked through exactly one optimizer step on the host side
            ops.PARAM_EPOCH[0] = epoch0                     # ... which has not run yet: the replay below is that step
            for o in S['panels']:
                o.epoch = epoch0
            self._graphs[nmax] = (gs, S['out5'], list(S['panels']))
        gs, out5, panels = self._graphs[nmax]
        gs[0].replay()
        if self.graph_split:
            parallel.all_reduce_mean_(st.grad)       # one flat message between the two replays (loss was scaled by world_size: trainer.py:337-338)
            gs[1].replay()
        ops.PARAM_EPOCH[0] += 1                     # the replay moved the weights: packed-weight caches of any eager forward are stale ...
        for _, _, ps in self._graphs.values():      # ... except the panels the replayed re-pack just refreshed in place
            for o in ps:
                o.epoch = ops.PARAM_EPOCH[0]
        st.steps += 1
        self.crit.epoch += 1
        self.last_opt_step = self.ni
        self.ni += 1
        return out5[0].clone(), out5[1:4].clone()

    def step(self, batch, epoch=0):
        st = self.state
        self.warmup(epoch)
        if self._graph_ok():
            return self._graph_step(batch)
        first_micro = self.ni == self.last_opt_step + 1            # first backward after an optimizer step (zero_grad)
        feats = self.model._predict_once(self.preprocess_batch(batch))
        # loss * world_size so that the mean all-reduce yields the global sum (trainer.py:337-338)
        total, items, head_grads = loss_and_head_grads(self.crit, feats, batch, gscale=float(self.world_size))
        keep = None if first_micro else st.grad.clone()              # gradient accumulation: the adjoint kernels overwrite
        will_step = self.ni - self.last_opt_step >= self.accumulate
        ex = self.exchange if (will_step and keep is None) else None
        self.model.backward(head_grads, layer_done=ex.layer_done if ex is not None else None)
        if keep is not None:
            st.grad.add_(keep)
        if will_step:
            if ex is not None:
                ex.finish()
            else:
                parallel.all_reduce_mean_(st.grad)                     # one flat message (accumulated micro-batches / no overlap)
            self.optimizer_step()
            self.last_opt_step = self.ni
        self.ni += 1
        return total, items
