"""The CPU oracle against the fixtures generated from the reference (tests/golden/gen_golden.py).

Runs without a GPU.  This is what "pins" the oracle: every restated function reproduces what the
reference's own Python computed on the same seeded inputs.
"""
import numpy as np
import pytest
import torch

import inputs as GI
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.seeding import seeded_images, seeded_tensor
from oracle import boxes as OB
from oracle import layers as OL
from oracle import loss as OLoss
from oracle import nms as ON
from oracle import tal as OT

torch.set_num_threads(8)


def oracle_state_dict(shapes, seed):
    """state_dict with the reference's names, filled by the shared seeding rule."""
    sd = {}
    for k, shp in shapes.items():
        v = seeded_tensor(k, shp, seed)
        if v is None:
            v = torch.arange(shp[1], dtype=torch.float32).view(shp) if k.endswith('dfl.conv.weight') else torch.zeros(shp)
        sd[k] = v
    return sd


def model_shapes(name, scale='n', nc=80):
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel(get_config(name, scale, nc), verbose=False)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}, m.stride.tolist()


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_e2e_forward_matches_reference(golden, tag):
    g = golden('e2e_' + tag)
    name = GI.E2E_MODELS[tag]
    shapes, strides = model_shapes(name)
    assert strides == g['stride'].tolist()
    sd = oracle_state_dict(shapes, 0)
    cfg = get_config(name, 'n')
    for (b, h, w) in GI.E2E_SHAPES:
        key = f'{b}x{h}x{w}'
        x = seeded_images(b, h, w, seed=GI.IMG_SEED)
        with torch.no_grad():
            (y, feats), layers = OL.model_forward(cfg, sd, x, strides, return_layers=True)
        if f'y_{key}' in g:
            np.testing.assert_allclose(y.numpy()[:, :4], g[f'y_{key}'][:, :4], atol=5e-4, rtol=0)   # px
            np.testing.assert_allclose(y.numpy()[:, 4:], g[f'y_{key}'][:, 4:], atol=2e-5, rtol=0)   # conf
            for i, f in enumerate(feats):
                np.testing.assert_allclose(f.numpy(), g[f'feat{i}_{key}'], atol=1e-4, rtol=1e-5)
        else:
            np.testing.assert_allclose(y.numpy()[:, :4, ::25], g[f'ysub_{key}'][:, :4], atol=5e-4, rtol=0)
            np.testing.assert_allclose(y.numpy()[:, 4:, ::25], g[f"ysub_{key}"][:, 4:], atol=2e-5, rtol=0)
        for i, o in enumerate(layers[:-1]):
            f = o.reshape(-1).double()
            step = max(1, f.numel() // 2048)
            np.testing.assert_allclose(f[::step][:2048].float().numpy(), g[f'L{i}_s_{key}'], atol=2e-4, rtol=1e-4)
    # BN-folded path (fuse_conv_and_bn) agrees with the reference's fused model
    b, h, w = GI.E2E_SHAPES[0]
    with torch.no_grad():
        yf, _ = OL.model_forward(cfg, sd, seeded_images(b, h, w, seed=GI.IMG_SEED), strides, fused=True)
    np.testing.assert_allclose(yf.numpy()[:, :4], g[f'yfused_{b}x{h}x{w}'][:, :4], atol=2e-4, rtol=0)


def _oracle_module(name, cls, args, sd, xs):
    x = xs[0] if len(xs) == 1 else xs
    if cls == 'Conv':
        act = 'silu' if len(args) < 8 else {'relu': 'relu', False: 'none'}[args[7]]
        return OL.conv(x, sd, 'm', s=args[3], act=act)
    if cls == 'DWConv':     # conv.py:82-86: Conv with groups = gcd(c1, c2)
        import math
        return OL.conv(x, sd, 'm', s=args[3], g=math.gcd(args[0], args[1]))
    if cls == 'SPRModule':
        return OL.spr(x, sd, 'm')
    if cls == 'Bottleneck':
        return OL.bottleneck(x, sd, 'm', args[2])
    if cls == 'C2f':
        return OL.c2f(x, sd, 'm', args[2], args[3])
    if cls == 'MSPA_C2f':
        return OL.mspa_c2f(x, sd, 'm', args[2], args[3])
    if cls == 'SPPF':
        return OL.sppf(x, sd, 'm')
    if cls == 'SimFusion_4in':
        return OL.simfusion_4in(x)
    if cls == 'SimFusion_3in':
        return OL.simfusion_3in(x, sd, 'm')
    if cls == 'IFM':
        return OL.ifm(x, sd, 'm')
    if cls == 'InjectionMultiSum_Auto_pool':
        return OL.inject(x, sd, 'm', args[2], args[3])
    raise KeyError(cls)


@pytest.mark.parametrize('name', list(GI.MODULE_CASES))
def test_module_matches_reference(golden, name):
    import mgdt_yolo_amd.nn.modules as M
    g = golden('modules')
    cls, args, _ = GI.MODULE_CASES[name]
    ctor_args = tuple(torch.nn.ReLU() if a == 'relu' else a for a in args)
    shapes = {'m.' + k: tuple(v.shape) for k, v in getattr(M, cls)(*ctor_args).state_dict().items()}
    # seeding is keyed on the name the reference module used ('' prefix), so re-key
    sd = {}
    for k, shp in shapes.items():
        v = seeded_tensor(k[2:], shp, GI.MODULE_SEED)
        sd[k] = v if v is not None else torch.zeros(shp)
    with torch.no_grad():
        y = _oracle_module(name, cls, args, sd, GI.module_inputs(name))
    np.testing.assert_allclose(y.numpy(), g[name], atol=3e-5, rtol=1e-5)


def test_ciou_and_box_iou(golden):
    g = golden('boxes')
    b1, b2 = torch.from_numpy(g['b1']), torch.from_numpy(g['b2'])
    np.testing.assert_allclose(OB.ciou_xyxy(b1, b2).numpy(), g['ciou'], atol=1e-6, rtol=0)
    np.testing.assert_allclose(OB.box_iou(b1[:64], b2[:96]).numpy(), g['box_iou'], atol=1e-7, rtol=0)


@pytest.mark.parametrize('seed,calls', GI.ASSIGNER_CASES)
def test_assigner_matches_reference(golden, seed, calls):
    g = golden('assigner')
    B, nc, hw = (GI.ASSIGNER_SHAPE[k] for k in ('B', 'nc', 'hw'))
    lab, pd_scores, pd_bboxes, anc = GI.assigner_inputs(B, nc, hw, seed)
    imgsz = torch.tensor([hw[0] * 8, hw[1] * 8], dtype=torch.float32)
    tg = OLoss.dense_targets(lab['batch_idx'], lab['cls'], lab['bboxes'], B, imgsz[[1, 0, 1, 0]])
    gl, gb = tg.split((1, 4), 2)
    mg = (gb.sum(2, keepdim=True) > 0).float()
    tl, tb, ts, fg, gi = OT.assign(pd_scores, pd_bboxes, anc, gl, gb, mg, calls, nc)
    k = f's{seed}'
    assert int(g[k + '_calls']) == calls
    assert np.array_equal(fg.numpy(), g[k + '_fg'])                      # integer outputs: bit exact
    assert np.array_equal(gi.numpy(), g[k + '_gt_idx'])
    assert np.array_equal(tl.numpy(), g[k + '_labels'])
    np.testing.assert_allclose(tb.numpy(), g[k + '_bboxes'], atol=0, rtol=0)
    np.testing.assert_allclose(ts.numpy(), g[k + '_scores'], atol=1e-6, rtol=1e-5)


def test_assigner_empty_labels_is_all_background():
    """Reference crashes here (tal.py:102-108); the build defines upstream behaviour."""
    out = OT.assign(torch.rand(2, 30, 4), torch.rand(2, 30, 4), torch.rand(30, 2), torch.zeros(2, 0, 1),
                    torch.zeros(2, 0, 4), torch.zeros(2, 0, 1), 0, 4)
    assert out[3].sum() == 0 and (out[0] == 4).all() and out[2].sum() == 0


@pytest.mark.parametrize('seed,calls', GI.LOSS_CASES)
def test_detection_loss_matches_reference(golden, seed, calls):
    g = golden('loss')
    B, nc, R, hw = (GI.LOSS_SHAPE[k] for k in ('B', 'nc', 'R', 'hw'))
    feats, lab = GI.loss_inputs(seed, B, nc, R, hw)
    f = feats.clone().requires_grad_(True)
    total, items, _ = OLoss.detection_loss([f], lab, [8.0], R, nc, call_count=calls)
    total.backward()
    k = f's{seed}'
    np.testing.assert_allclose(total.item(), g[k + '_total'], rtol=2e-6)
    np.testing.assert_allclose(items.numpy(), g[k + '_items'], rtol=2e-6)
    np.testing.assert_allclose(f.grad.numpy(), g[k + '_grad'], atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_nms_stages_match_reference(golden, tag):
    """Everything in non_max_suppression around the (absent, unpinned) torchvision.ops.nms call."""
    g, y = golden('nms'), golden('e2e_' + tag)['y_2x160x160']
    for cname, kw in GI.NMS_CASES:
        out = ON.non_max_suppression(y, **kw)
        for i, o in enumerate(out):
            ref = g[f'{tag}_{cname}_{i}']
            assert o.shape == ref.shape, (cname, i, o.shape, ref.shape)
            assert np.array_equal(o, ref), (cname, i)


def test_greedy_nms_properties():
    """torchvision.ops.nms is unpinned: check the published semantics through its invariants."""
    r = np.random.default_rng(0)
    c = r.uniform(0, 100, (400, 2)); wh = r.uniform(5, 40, (400, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    keep = ON.greedy_nms(b, 0.5)
    assert keep[0] == 0 and np.all(np.diff(keep) > 0)
    iou = OB.box_iou(torch.from_numpy(b[keep]), torch.from_numpy(b[keep]), eps=0).numpy()
    assert (iou[np.triu_indices(len(keep), 1)] <= 0.5 + 1e-6).all()          # survivors do not overlap > thr
    dropped = np.setdiff1d(np.arange(400), keep)
    iou_d = OB.box_iou(torch.from_numpy(b[dropped]), torch.from_numpy(b[keep]), eps=0).numpy()
    for j, d in enumerate(dropped):                                             # each dropped box has an earlier keeper
        assert (iou_d[j][keep < d] > 0.5 - 1e-6).any()
    assert np.array_equal(ON.greedy_nms(b[keep], 0.5), np.arange(len(keep)))   # idempotent
    assert len(ON.greedy_nms(np.zeros((0, 4), np.float32), 0.5)) == 0


def test_compiled_greedy_nms_equals_the_numpy_restatement(golden):
    """oracle/csrc/greedy_nms.c (used by the timed CPU baseline) == oracle.nms.greedy_nms, kept positions bit for bit, incl. the early stop."""
    r = np.random.default_rng(3)
    for n in (0, 1, 7, 400, 3000):
        c = r.uniform(0, 300, (n, 2)); wh = r.uniform(5, 80, (n, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        for thr in (0.3, 0.7):
            full = ON.greedy_nms(b, thr)
            assert np.array_equal(ON.greedy_nms_c(b, thr), full)
            assert np.array_equal(ON.greedy_nms_c(b, thr, limit=50), full[:50])
    y = golden('e2e_mspa_c2f_gd_n')['y_2x160x160']
    for cname, kw in GI.NMS_CASES:
        a, ka = ON.non_max_suppression(y, return_index=True, **kw)
        b_, kb = ON.non_max_suppression(y, return_index=True, compiled=True, **kw)
        for u, v, (ia, ca), (ib, cb) in zip(a, b_, ka, kb):
            assert np.array_equal(u, v) and np.array_equal(ia, ib) and np.array_equal(ca, cb), cname


def test_validator_matching_matches_reference(golden):
    """oracle.val.process_batch vs the reference's own DetectionValidator._process_batch (fixture val_match.npz), all six cases bit-exact."""
    from oracle import val as OV
    g = golden('val_match')
    iouv = torch.from_numpy(g['iouv'])
    assert np.array_equal(g['iouv'], torch.linspace(0.5, 0.95, 10).numpy())
    for seed, nd, nl in GI.VAL_MATCH_CASES:
        det, lab = GI.val_match_inputs(seed, nd, nl)
        got = OV.process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv)
        assert got.shape == g[f'c{seed}'].shape and np.array_equal(got, g[f'c{seed}']), seed


# ------------------------------------------------------------------------------------------------ validator AP, box helpers, LetterBox
@pytest.mark.parametrize('seed,nd,nl,nc', GI.AP_CASES)
def test_ap_per_class_matches_reference(golden, seed, nd, nl, nc):
    from oracle import metrics as OM
    g = golden('metrics_ap')
    out = OM.ap_per_class(*GI.ap_inputs(seed, nd, nl, nc))
    for name, v in zip(('tp', 'fp', 'p', 'r', 'f1', 'ap', 'cls'), out):
        assert np.array_equal(np.asarray(v), g[f's{seed}_{name}']), name          # same numpy expressions: bit for bit


def test_box_helpers_match_reference(golden):
    from oracle import metrics as OM
    g, gb = golden('boxes2'), golden('boxes')
    for k, (s1, s0, rp) in enumerate(GI.SCALE_BOX_CASES):
        np.testing.assert_array_equal(OM.scale_boxes(s1, GI.scale_box_inputs(k), s0, rp), g[f'scale{k}'])
    b1, b2 = torch.from_numpy(gb['b1']), torch.from_numpy(gb['b2'])
    np.testing.assert_allclose(OB.ciou_xyxy(b1, b2).numpy(), g['ciou_xyxy'], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(OB.xywh2xyxy(torch.from_numpy(g['xyxy2xywh'])).numpy(), g['xywh2xyxy'])


def test_letterbox_geometry_matches_reference(golden):
    from oracle import metrics as OM
    g = golden('letterbox')
    for k, (shape, new_shape, auto) in enumerate(GI.LETTERBOX_CASES):
        assert list(OM.letterbox_geometry(shape, new_shape, auto)) == g[f'c{k}'].tolist(), (shape, new_shape, auto)
    # the resize restatement (cv2 absent: unpinned) at least reproduces what every bilinear resize must: identity at equal size, constants stay constant
    r = np.random.default_rng(0)
    im = r.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    assert np.array_equal(OM.resize_linear_u8(im, 17, 13), im)
    assert (OM.resize_linear_u8(np.full((9, 11, 3), 77, np.uint8), 23, 31) == 77).all()


def check_train_fixture(g, loss, items, feats, grads, running, tol):
    """Shared by the CPU (oracle) and GPU (HIP) tests: one training step against tests/golden/train_<tag>.npz, i.e. against what the
    REFERENCE's `model.train()(batch)` + `loss.backward()` produced on CPU (gen_golden.py:train).  `grads`: name -> tensor, `running`:
    BN prefix -> (mean, var).  tol = dict(feat=, loss=, grad=, grad_abs=, run=): grad is the per-tensor error in units of that tensor's l2 norm
    (measured on the stored sample), with tensors whose reference gradient is analytically ~0 held to grad_abs * the typical norm instead."""
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=tol['loss'])
    np.testing.assert_allclose(np.asarray(items, np.float64), g['items'], rtol=tol['loss'])
    for i, f in enumerate(feats):
        np.testing.assert_allclose(np.asarray(f), g[f'feat{i}'], atol=tol['feat'], rtol=tol['feat'])
    names = str(g['grad_names']).split('\n')
    norms = np.array([g['gst/' + k][0] / np.sqrt(max(np.prod(np.shape(grads[k])), 1)) for k in names if k in grads])
    typical = float(np.median(norms))
    worst = (0.0, None)
    for k in names:
        assert k in grads and grads[k] is not None, f'no gradient for {k}'
        got, st = GI.grad_sample(grads[k])
        ref, rst = g['g/' + k], g['gst/' + k]
        rms_ref = rst[0] / np.sqrt(max(np.prod(np.shape(grads[k])), 1))
        denom = max(rms_ref, tol['grad_abs'] * typical)
        err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) / denom)
        nerr = abs(st[0] - rst[0]) / (denom * np.sqrt(max(np.prod(np.shape(grads[k])), 1)))
        err = max(err, nerr)
        if err > worst[0]:
            worst = (err, k)
        assert err < tol['grad'], (k, err, rms_ref, typical)
    n_run = 0
    for key in g.files:
        if key.startswith('bn/') and key.endswith('running_mean'):
            pre = key[3:-len('.running_mean')]
            mu, var = running[pre]
            np.testing.assert_allclose(np.asarray(mu), g[key], atol=tol['run'], rtol=tol['run'], err_msg=pre)
            np.testing.assert_allclose(np.asarray(var), g[f'bn/{pre}.running_var'], atol=tol['run'], rtol=tol['run'], err_msg=pre)
            n_run += 1
    assert n_run > 40
    return worst


@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_training_step_matches_reference(golden, tag):
    """The oracle's TRAIN-mode branch (batch-statistics BatchNorm, oracle/layers.py BN_TRAIN) + loss + torch autograd against the reference's
    own CPU training step: head maps, loss, items, all ~200 parameter gradients and every BN running statistic.  This pins the branch every
    HIP gradient / batch-stat test is compared with."""
    g = golden('train_' + tag)
    c = GI.TRAIN_CASE
    name = GI.E2E_MODELS[tag]
    shapes, strides = model_shapes(name, nc=c['nc'])
    sd = oracle_state_dict(shapes, c['weight_seed'])
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k and 'dfl' not in k) for k, v in sd.items()}
    x, lab = GI.train_inputs()
    OL.BN_TRAIN, OL.BN_RUNNING_OUT = True, {}
    try:
        feats = OL.model_forward(get_config(name, 'n', c['nc']), sd, x, strides, decode=False)
        total, items, _ = OLoss.detection_loss(feats, lab, strides, 4, c['nc'], call_count=0)
        total.backward()
        running = {k: (m.numpy(), v.numpy()) for k, (m, v) in OL.BN_RUNNING_OUT.items()}
    finally:
        OL.BN_TRAIN, OL.BN_RUNNING_OUT = False, None
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
    assert str(g['nograd_names']).split('\n') == [k for k in shapes if k.endswith('dfl.conv.weight')]
    worst = check_train_fixture(g, total.detach(), items, [f.detach().numpy() for f in feats], grads, running,
                                dict(feat=2e-4, loss=2e-5, grad=2e-3, grad_abs=1e-2, run=1e-5))
    print('worst gradient error (in units of the tensor rms)', worst)


def test_reference_class_checkpoint_is_read_without_unpickling_code():
    """tests/golden/ref_last.pt was written by the REFERENCE's classes (gen_golden.py:ref_ckpt: save_model's dict, trainer.py:413-422, after
    strip_optimizer, pickled with dill as the reference does when dill is importable): its globals are ultralytics.nn.tasks.DetectionModel,
    ultralytics.nn.modules.block.MSPA_C2f, ... - none of which exists on this side.  nn/checkpoint.py must recover YAML, names, args and
    every weight (fp16-rounded seeded values) without importing or running anything the file names."""
    import os
    from mgdt_yolo_amd.nn.tasks import attempt_load_one_weight
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_last.pt')
    model, ckpt = attempt_load_one_weight(path)
    c = GI.CKPT_CASE
    assert ckpt['epoch'] == -1 and ckpt['ema'] is None and ckpt['optimizer'] is None and ckpt['version'] == '8.0.120'
    assert ckpt['train_args']['model'] == 'mspa_c2f_gd_yolov8.yaml'
    assert model.names == {i: f'class{i}' for i in range(c['nc'])} and model.args['box'] == 7.5 and model.yaml['nc'] == c['nc']
    assert any(s.startswith('ultralytics.nn.') for s in model.ckpt_stubbed_globals) and 'dill._dill._load_type' in model.ckpt_stubbed_globals
    assert not any(k.startswith('ultralytics') for k in __import__('sys').modules), 'reading the file must not import what it names'
    n = 0
    for k, v in model.state_dict().items():
        e = seeded_tensor(k, v.shape, c['weight_seed'])
        if e is not None:
            assert torch.equal(v, e.half().float()), k
            n += 1
    assert n > 300 and not model.training


def test_checkpoint_that_does_not_match_its_graph_is_refused(tmp_path):
    """a state_dict key that is missing or has another shape must raise (the reference cannot load partially)."""
    import os
    import zipfile
    from mgdt_yolo_amd.nn import checkpoint as CK
    from mgdt_yolo_amd.nn import tasks as T
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_last.pt')
    real = CK.module_state_dict

    def drop_one(mod, prefix=''):
        sd = real(mod, prefix)
        if not prefix:
            sd.pop('model.3.conv.weight')
            sd['model.0.bn.weight'] = sd['model.0.bn.weight'][:8]
        return sd
    CK.module_state_dict = drop_one
    try:
        with pytest.raises(RuntimeError, match='model.3.conv.weight'):
            T.torch_safe_load(path)
    finally:
        CK.module_state_dict = real
