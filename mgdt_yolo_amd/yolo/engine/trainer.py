"""Training-step counterpart of the reference's DetectionTrainer hot loop (yolo/engine/trainer.py:314-362,462-470).

Only what touches the device per step lives here; the data pipeline, callbacks, checkpoints and logging of the reference's
BaseTrainer are out of scope.  One step = preprocess (uint8/255) -> train-mode forward (HIP) -> fused
assigner+loss (HIP) -> explicit backward (HIP) -> [all-reduce of the flat gradient buffer, RCCL] -> clip(10) + SGD(Nesterov)
+ EMA on the flat parameter buffer (3 HIP launches).
"""
import math

import torch

from ... import ops, parallel
from ..utils.loss import loss_and_head_grads, v8DetectionLoss


class FlatState:
    """All trainable parameters (and float buffers) of a model as views into one flat fp32 buffer; grads likewise."""

    def __init__(self, model, weight_decay):
        import torch.nn as nn
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        bufs = [(n, b) for n, b in model.named_buffers() if b.dtype.is_floating_point and b.numel() > 0 and 'anchors' not in n and 'strides' not in n]
        dev = params[0][1].device
        self.n_param = sum(p.numel() for _, p in params)
        self.n_total = self.n_param + sum(b.numel() for _, b in bufs)
        self.data = torch.empty(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        self.wd = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        # weight-decay groups of trainer.py:645-660: decay on conv / linear weights only (not on norm weights, not on biases)
        norm_w = set()
        for mn, m in model.named_modules():
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm, nn.LayerNorm)) or type(m).__name__ in ('LayerNorm', 'GRN'):
                for pn, _ in m.named_parameters(recurse=False):
                    norm_w.add(f'{mn}.{pn}' if mn else pn)
        off = 0
        for n, p in params:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            if n.endswith('.weight') and n not in norm_w and p.ndim > 1:
                self.wd[off:off + k] = weight_decay
            off += k
        for n, b in bufs:
            k = b.numel()
            self.data[off:off + k].copy_(b.detach().reshape(-1))
            b.data = self.data[off:off + k].view(b.shape)
            off += k
        self.momentum_buf = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        self.ema = self.data.clone()
        self.steps = 0


class DetectionTrainer:
    """Minimal per-step driver: `trainer.step(batch)` -> (loss*B, loss_items[3]).  Hyper-parameters follow yolo/cfg/default.yaml
    of the fork (lr0 0.001, momentum 0.937, weight_decay 5e-4, nesterov SGD; grad clip 10.0; EMA decay 0.9999, tau 2000)."""

    def __init__(self, model, lr0=0.001, momentum=0.937, weight_decay=5e-4, world_size=1, ema_decay=0.9999, ema_tau=2000.0):
        self.model = model.train()
        self.crit = v8DetectionLoss(model)
        self.state = FlatState(model, weight_decay)
        self.lr, self.momentum, self.world_size = lr0, momentum, world_size
        self.ema_decay, self.ema_tau = ema_decay, ema_tau

    def preprocess_batch(self, batch):
        """uint8 -> float / 255 on the device (detect/train.py:62-65)."""
        img = batch['img'].to(self.state.data.device, non_blocking=True)
        return img.float() / 255 if img.dtype == torch.uint8 else img.float()

    def step(self, batch):
        st = self.state
        feats = self.model(self.preprocess_batch(batch))
        # loss * world_size so that the mean all-reduce yields the global sum (trainer.py:337-338)
        total, items, head_grads = loss_and_head_grads(self.crit, feats, batch, gscale=float(self.world_size))
        self.model.backward(head_grads)
        parallel.all_reduce_mean_(st.grad)                         # one flat RCCL message (5.26 MB for the n model)
        clip = ops.grad_clip_coef(st.grad, 10.0)                   # trainer.py:466
        ops.sgd_step(st.data[:st.n_param], st.grad, st.momentum_buf, st.wd, self.lr, self.momentum, True, st.steps == 0, clip)
        st.steps += 1
        d = self.ema_decay * (1 - math.exp(-st.steps / self.ema_tau))      # torch_utils.py:342
        ops.ema_update(st.ema, st.data, d)
        return total, items
