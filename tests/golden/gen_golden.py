"""Generate tests/golden/*.npz from the reference's own Python modules (BUILD CONTAINER ONLY).

    python tests/golden/gen_golden.py

Needs /root/reference (read-only) - see ref_import.py for the import recipe.  The outputs are data
(inputs are re-creatable from seeds via mgdt_yolo_amd.seeding; expected outputs come from running
the reference on CPU, fp32).  The GPU box never runs this file.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images, seeded_labels, seeded_tensor  # noqa: E402
from oracle import nms as onms  # noqa: E402  (only to stand in for the absent torchvision.ops.nms)
from inputs import (ASSIGNER_CASES, ASSIGNER_SHAPE, E2E_MODELS, E2E_SHAPES, IMG_SEED, LOSS_CASES, LOSS_SHAPE,  # noqa: E402
                    MODULE_CASES, MODULE_SEED, NMS_CASES, assigner_inputs, loss_inputs, module_inputs)

torch.set_num_threads(8)
ns = ref_import.load()
REFY = '/root/reference/models/v8/'


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def sample(t, n=2048):
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([f.sum().item(), f.abs().sum().item()], np.float64)


def build(yaml_name, nc=80, seed=0):
    m = ns.tasks.DetectionModel(REFY + yaml_name, nc=nc, verbose=False)
    seed_state_dict_(m, seed)
    m.args = types.SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    return m.eval()


# ------------------------------------------------------------------ end-to-end forward fixtures
def e2e(tag, yaml_name, shapes):
    m = build(yaml_name)
    arrs = {'stride': m.stride.numpy(), 'sd_keys': '\n'.join(m.state_dict().keys()),
            'sd_shapes': '\n'.join(','.join(str(d) for d in v.shape) for v in m.state_dict().values())}
    for (b, h, w) in shapes:
        x = seeded_images(b, h, w, seed=IMG_SEED)
        layers = []
        hooks = [l.register_forward_hook(lambda mod, i, o, L=layers: L.append(o)) for l in m.model]
        with torch.no_grad():
            y, feats = m(x)
        for hk in hooks:
            hk.remove()
        key = f'{b}x{h}x{w}'
        if h * w <= 160 * 160:
            arrs[f'y_{key}'] = y.numpy()
            for i, f in enumerate(feats):
                arrs[f'feat{i}_{key}'] = f.numpy()
        else:
            arrs[f'ysub_{key}'] = y[:, :, ::25].numpy()
        for i, o in enumerate(layers[:-1]):
            s, st = sample(o)
            arrs[f'L{i}_s_{key}'], arrs[f'L{i}_st_{key}'] = s, st
    # fused model (AutoBackend path, autobackend.py:94) on the first shape
    b, h, w = shapes[0]
    mf = build(yaml_name).fuse(verbose=False)
    with torch.no_grad():
        yf, _ = mf(seeded_images(b, h, w, seed=IMG_SEED))
    arrs[f'yfused_{b}x{h}x{w}'] = yf.numpy()
    save(f'e2e_{tag}', **arrs)
    return m


# ------------------------------------------------------------------ per-module fixtures
def modules():
    import ultralytics.nn.modules.block as RB
    arrs = {}
    for name, (cls, args, _) in MODULE_CASES.items():
        args = tuple(torch.nn.ReLU() if a == 'relu' else a for a in args)
        ctor = getattr(ns.modules, cls, None) or getattr(RB, cls)
        mod = seed_state_dict_(ctor(*args), MODULE_SEED).eval()
        for sub in mod.modules():
            if isinstance(sub, torch.nn.BatchNorm2d):
                sub.eps = 1e-3   # what initialize_weights does inside DetectionModel (torch_utils.py:254)
        xs = module_inputs(name)
        with torch.no_grad():
            arrs[name] = mod(xs[0] if len(xs) == 1 else xs).numpy()
    save('modules', **arrs)


# ------------------------------------------------------------------ CIoU / box_iou
def boxes():
    r = np.random.default_rng(5)
    c1, c2 = r.uniform(50, 500, (512, 2)), r.uniform(50, 500, (512, 2))
    c2[:256] = c1[:256] + r.normal(0, 12, (256, 2))
    w1, w2 = r.uniform(4, 200, (512, 2)), r.uniform(4, 200, (512, 2))
    b1 = np.concatenate([c1 - w1 / 2, c1 + w1 / 2], 1).astype(np.float32)
    b2 = np.concatenate([c2 - w2 / 2, c2 + w2 / 2], 1).astype(np.float32)
    t1, t2 = torch.from_numpy(b1), torch.from_numpy(b2)
    ciou = ns.metrics.bbox_iou(t1, t2, xywh=False, CIoU=True)
    iou = ns.metrics.box_iou(t1[:64], t2[:96])
    save('boxes', b1=b1, b2=b2, ciou=ciou.numpy(), box_iou=iou.numpy())


# ------------------------------------------------------------------ assigner + loss
def assigner():
    from oracle.loss import dense_targets
    arrs = {}
    B, nc, hw = ASSIGNER_SHAPE['B'], ASSIGNER_SHAPE['nc'], ASSIGNER_SHAPE['hw']
    for seed, calls in ASSIGNER_CASES:
        lab, pd_scores, pd_bboxes, anc = assigner_inputs(B, nc, hw, seed)
        imgsz = torch.tensor([hw[0] * 8, hw[1] * 8], dtype=torch.float32)
        tg = dense_targets(lab['batch_idx'], lab['cls'], lab['bboxes'], B, imgsz[[1, 0, 1, 0]])
        gl, gb = tg.split((1, 4), 2)
        mg = gb.sum(2, keepdim=True).gt_(0)
        a = ns.tal.HeuristicPositiveSampleAssigner_v1(num_classes=nc, alpha=0.5, beta=8.0, iou_threshold=0.4)
        tl, tb, ts, fg, gi = a(pd_scores, pd_bboxes, anc, gl, gb, mg, calls)
        k = f's{seed}'
        arrs[k + '_calls'] = calls
        arrs[k + '_fg'] = fg.numpy()
        arrs[k + '_gt_idx'] = gi.numpy().astype(np.int32)
        arrs[k + '_labels'] = tl.numpy().astype(np.int32)
        arrs[k + '_scores'] = ts.numpy()
        arrs[k + '_bboxes'] = tb.numpy()
        print('assigner', k, 'positives', int(fg.sum()))
    save('assigner', **arrs)


def loss():
    arrs = {}
    B, nc, R, hw = (LOSS_SHAPE[k] for k in ('B', 'nc', 'R', 'hw'))
    no = 4 * R + nc
    for seed, calls in LOSS_CASES:
        feats, lab = loss_inputs(seed, B, nc, R, hw)
        head = types.SimpleNamespace(stride=torch.tensor([8.0]), nc=nc, no=no, reg_max=R)
        model = types.SimpleNamespace(args=types.SimpleNamespace(box=7.5, cls=0.5, dfl=1.5), model=[head],
                                      parameters=lambda: iter([torch.zeros(1)]))
        crit = ns.loss.v8DetectionLoss(model)
        crit.epoch = calls
        f = feats.clone().requires_grad_(True)
        total, items = crit([f], lab)
        total.backward()
        k = f's{seed}'
        arrs[k + '_calls'] = calls
        arrs[k + '_total'] = total.detach().numpy()
        arrs[k + '_items'] = items.numpy()
        arrs[k + '_grad'] = f.grad.numpy()
        print('loss', k, float(total), items.tolist())
    save('loss', **arrs)


# ------------------------------------------------------------------ NMS (stages around the absent torchvision op)
def nms(models):
    import torchvision  # the stand-in module from ref_import

    def nms_standin(boxes, scores, thr):
        assert bool((scores[:-1] >= scores[1:]).all()), 'reference hands nms() descending scores'
        return torch.from_numpy(onms.greedy_nms(boxes.numpy(), thr))

    torchvision.ops.nms = nms_standin
    arrs = {}
    for tag, m in models.items():
        with torch.no_grad():
            y, _ = m(seeded_images(2, 160, 160, seed=IMG_SEED))
        for cname, kw in NMS_CASES:
            out = ns.ops.non_max_suppression(y.clone(), max_time_img=1e9, **kw)
            for i, o in enumerate(out):
                arrs[f'{tag}_{cname}_{i}'] = o.numpy()
            print('nms', tag, cname, [int(o.shape[0]) for o in out])
    save('nms', **arrs)


# ------------------------------------------------------------------ validator matching (yolo/v8/detect/val.py:152-175)
def val_match():
    """Runs the reference's DetectionValidator._process_batch.  Importing yolo/v8/detect/val.py pulls in the whole dataset / plotting stack
    (cv2 constants, torchvision.datasets), so the method is taken out of the class with `ast` and executed as is, bound to the reference's
    own `box_iou` - reference code running, nothing restated here."""
    import ast
    from inputs import VAL_MATCH_CASES, val_match_inputs
    src = open('/root/reference/yolo/v8/detect/val.py').read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == 'DetectionValidator')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '_process_batch')
    env = {'np': np, 'torch': torch, 'box_iou': ns.metrics.box_iou}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'ref:_process_batch', 'exec'), env)
    me = types.SimpleNamespace(iouv=torch.linspace(0.5, 0.95, 10))
    arrs = {'iouv': me.iouv.numpy()}
    for seed, nd, nl in VAL_MATCH_CASES:
        det, lab = val_match_inputs(seed, nd, nl)
        if nd == 0 or nl == 0:
            correct = np.zeros((nd, 10), bool)       # val.py:107-113: the caller never reaches _process_batch for empty sides
        else:
            correct = env['_process_batch'](me, torch.from_numpy(det), torch.from_numpy(lab)).numpy()
        arrs[f'c{seed}'] = correct
        print('val_match', seed, det.shape, lab.shape, 'true positives per IoU level', correct.sum(0).tolist())
    save('val_match', **arrs)


# ------------------------------------------------------------------ optimizer parameter groups (yolo/engine/trainer.py:615-664)
def optim_groups():
    """Runs the reference's BaseTrainer.build_optimizer on the reference's own model and records which group every parameter lands in.
    The method is taken out of the class with `ast` (importing engine/trainer.py needs the whole data / logging stack) and executed as is."""
    import ast
    from torch import nn, optim
    src = open('/root/reference/yolo/engine/trainer.py').read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == 'BaseTrainer')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'build_optimizer')
    env = {'nn': nn, 'optim': optim, 'LOGGER': types.SimpleNamespace(info=lambda *a, **k: None), 'colorstr': lambda *a: ''}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'ref:build_optimizer', 'exec'), env)
    arrs = {}
    for tag, yname in E2E_MODELS.items():
        m = build(yname + '.yaml')
        me = types.SimpleNamespace(args=types.SimpleNamespace(warmup_bias_lr=0.1))
        opt = env['build_optimizer'](me, m, name='SGD', lr=0.001, momentum=0.937, decay=5e-4, iterations=1e5)
        ident = {id(p): n for n, p in m.named_parameters()}
        # param_groups order of the reference: [0] biases, [1] decayed weights, [2] norm weights
        for gi, key in enumerate(('bias', 'decay', 'norm')):
            arrs[f'{tag}_{key}'] = '\n'.join(ident[id(p)] for p in opt.param_groups[gi]['params'])
            assert opt.param_groups[gi]['weight_decay'] == (5e-4 if key == 'decay' else 0.0)
        print('optim_groups', tag, [len(g['params']) for g in opt.param_groups])
    save('optim_groups', **arrs)


# ------------------------------------------------------------------ AP reduction, box helpers, LetterBox geometry
def metrics_ap():
    from inputs import AP_CASES, ap_inputs
    arrs = {}
    for seed, nd, nl, nc in AP_CASES:
        tp, conf, pcls, tcls = ap_inputs(seed, nd, nl, nc)
        out = ns.metrics.ap_per_class(tp, conf, pcls, tcls, plot=False, names={})
        for name, v in zip(('tp', 'fp', 'p', 'r', 'f1', 'ap', 'cls'), out):
            arrs[f's{seed}_{name}'] = np.asarray(v)
        print('ap', seed, 'mAP50', out[5][:, 0].mean(), 'mAP50-95', out[5].mean())
    save('metrics_ap', **arrs)


def boxes2():
    from inputs import SCALE_BOX_CASES, scale_box_inputs
    g = np.load(os.path.join(HERE, 'boxes.npz'))
    b1, b2 = torch.from_numpy(g['b1']), torch.from_numpy(g['b2'])
    arrs = {}
    for name, kw in (('iou', {}), ('giou', dict(GIoU=True)), ('diou', dict(DIoU=True)), ('ciou', dict(CIoU=True))):
        arrs[name + '_xyxy'] = ns.metrics.bbox_iou(b1, b2, xywh=False, **kw).numpy()
        arrs[name + '_xywh'] = ns.metrics.bbox_iou(ns.ops.xyxy2xywh(b1), ns.ops.xyxy2xywh(b2), xywh=True, **kw).numpy()
    arrs['one_vs_many'] = ns.metrics.bbox_iou(b1[:1], b2, xywh=False, CIoU=True).numpy()
    arrs['xyxy2xywh'] = ns.ops.xyxy2xywh(b1).numpy()
    arrs['xywh2xyxy'] = ns.ops.xywh2xyxy(ns.ops.xyxy2xywh(b1)).numpy()
    for k, (s1, s0, rp) in enumerate(SCALE_BOX_CASES):
        arrs[f'scale{k}'] = ns.ops.scale_boxes(s1, torch.from_numpy(scale_box_inputs(k)), s0, ratio_pad=rp).numpy()
    save('boxes2', **arrs)


def letterbox_geom():
    """The reference's LetterBox (yolo/data/augment.py:538-593) with recording stand-ins for the two cv2 calls: pins the geometry (resize
    target, border sizes), which is all of LetterBox that is not cv2's own arithmetic."""
    import ast
    import cv2      # the inert stand-in module of ref_import
    from inputs import LETTERBOX_CASES
    src = open('/root/reference/yolo/data/augment.py').read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == 'LetterBox')
    rec = {}
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0

    def resize(img, size, interpolation=None):
        rec['resize'] = tuple(size)
        return np.zeros((size[1], size[0], 3), np.uint8)

    def border(img, top, bottom, left, right, kind, value=None):
        rec['border'] = (top, bottom, left, right)
        return np.zeros((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    cv2.resize, cv2.copyMakeBorder = resize, border
    env = {'np': np, 'cv2': cv2}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), 'ref:LetterBox', 'exec'), env)
    arrs = {}
    for k, (shape, new_shape, auto) in enumerate(LETTERBOX_CASES):
        rec.clear()
        out = env['LetterBox'](new_shape, auto=auto, stride=32)(image=np.zeros((*shape, 3), np.uint8))
        rs = rec.get('resize', (shape[1], shape[0]))
        arrs[f'c{k}'] = np.array([out.shape[0], out.shape[1], rs[1], rs[0], rec['border'][0], rec['border'][2], int('resize' in rec)], np.int64)
        print('letterbox', shape, new_shape, auto, arrs[f'c{k}'].tolist())
    save('letterbox', **arrs)


# ------------------------------------------------------------------ one TRAINING step of the reference
def train():
    """The reference's own training call sequence on CPU (yolo/engine/trainer.py:334-343): `model.train()`; `loss, items = model(batch)`
    (nn/tasks.py:33-46,204-216 -> v8DetectionLoss) ; `loss.backward()`.  Stored per model: the train-mode head maps (batch-statistics
    BatchNorm, nn/modules/conv.py:36-38), loss and items, EVERY parameter gradient (whole when small, else a strided sample + l2 norm + sum)
    and every BatchNorm running_mean / running_var / num_batches_tracked after the forward (momentum 0.03, unbiased variance:
    yolo/utils/torch_utils.py:254-256)."""
    from inputs import TRAIN_CASE, grad_sample, train_inputs
    c = TRAIN_CASE
    x, lab = train_inputs()
    for tag, yname in E2E_MODELS.items():
        m = build(yname + '.yaml', nc=c['nc'], seed=c['weight_seed']).train()
        batch = dict(img=x.clone(), **{k: v.clone() for k, v in lab.items()})
        feats = []
        hk = m.model[-1].register_forward_hook(lambda mod, i, o: feats.extend(o))
        loss, items = m(batch)
        hk.remove()
        loss.backward()
        arrs = {'loss': loss.detach().numpy(), 'items': items.numpy(), 'stride': m.stride.numpy()}
        for i, f in enumerate(feats):
            arrs[f'feat{i}'] = f.detach().numpy()
        names, nograd = [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                nograd.append(k)
                continue
            names.append(k)
            arrs['g/' + k], arrs['gst/' + k] = grad_sample(p.grad)
        arrs['grad_names'] = '\n'.join(names)
        arrs['nograd_names'] = '\n'.join(nograd)
        for k, b in m.named_buffers():
            if 'running_' in k or 'num_batches_tracked' in k:
                arrs['bn/' + k] = b.detach().numpy()
        print('train', tag, float(loss), items.tolist(), len(names), 'gradients;', 'no grad:', nograd)
        save(f'train_{tag}', **arrs)


# ------------------------------------------------------------------ a checkpoint written by the reference's classes
def ref_ckpt():
    """`torch.save` of the dict BaseTrainer.save_model builds (yolo/engine/trainer.py:411-436): 'model' is the pickled reference
    DetectionModel in half precision (deepcopy(de_parallel(model)).half()), 'ema' likewise, plus epoch / best_fitness / updates /
    train_args / date / version.  Globals inside the pickle are the REFERENCE's class paths (ultralytics.nn.tasks.DetectionModel,
    ultralytics.nn.modules...), which is what nn/checkpoint.py must read without importing or executing anything.  The expected
    weights are the seeded state_dict (name-keyed: mgdt_yolo_amd.seeding), rounded to fp16."""
    from copy import deepcopy
    from inputs import CKPT_CASE
    try:                                     # trainer.py:424-428: dill if importable (it is, in this image), else pickle
        import dill as pickle_module
    except ImportError:
        import pickle as pickle_module
    m = build('mspa_c2f_gd_yolov8.yaml', nc=CKPT_CASE['nc'], seed=CKPT_CASE['weight_seed']).train()
    targs = dict(box=7.5, cls=0.5, dfl=1.5, imgsz=640, model='mspa_c2f_gd_yolov8.yaml', task='detect')
    m.args = targs                           # trainer.py:set_model_attributes attaches the hyper-parameters
    m.names = {i: f'class{i}' for i in range(CKPT_CASE['nc'])}
    half = deepcopy(m).half()
    for p_ in half.parameters():             # torch_utils.py:401-402
        p_.requires_grad = False
    # save_model's keys (trainer.py:413-422) after strip_optimizer (torch_utils.py:395-403): model <- EMA copy in fp16, the rest None
    ckpt = {'epoch': -1, 'best_fitness': None, 'model': half, 'ema': None, 'updates': None, 'optimizer': None, 'train_args': targs,
            'date': '2026-01-01T00:00:00', 'version': '8.0.120'}
    path = os.path.join(HERE, 'ref_last.pt')
    torch.save(ckpt, path, pickle_module=pickle_module)
    print(f'ref_last.pt: {os.path.getsize(path) / 1024:.1f} KiB, pickled with {pickle_module.__name__}')


if __name__ == '__main__':
    what = sys.argv[1:] or ['train', 'ref_ckpt', 'e2e', 'modules', 'boxes', 'assigner', 'loss', 'nms', 'val_match', 'optim_groups', 'metrics_ap', 'boxes2', 'letterbox']
    models = {}
    if 'e2e' in what or 'nms' in what:
        for tag, yname in E2E_MODELS.items():
            models[tag] = e2e(tag, yname + '.yaml', E2E_SHAPES) if 'e2e' in what else build(yname + '.yaml')
    if 'modules' in what:
        modules()
    if 'boxes' in what:
        boxes()
    if 'assigner' in what:
        assigner()
    if 'loss' in what:
        loss()
    if 'nms' in what:
        nms(models)
    if 'val_match' in what:
        val_match()
    if 'optim_groups' in what:
        optim_groups()
    if 'metrics_ap' in what:
        metrics_ap()
    if 'boxes2' in what:
        boxes2()
    if 'letterbox' in what:
        letterbox_geom()
    if 'train' in what:
        train()
    if 'ref_ckpt' in what:
        ref_ckpt()
