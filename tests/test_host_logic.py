"""CPU-only checks of the host side: the C ABI loads and exports what include/mgdt.h declares, the YAML -> graph rules,
state_dict compatibility with the reference, argument validation, and the N>1 (gloo, world_size 2) bench path."""
import os
import re

import numpy as np
import pytest
import torch

import inputs as GI
from mgdt_yolo_amd import _lib
from mgdt_yolo_amd.models import CONFIGS, get_config
from mgdt_yolo_amd.nn import tasks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_YAML = '/root/reference/models/v8'


def test_c_abi_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'mgdt.h')).read()
    declared = set(re.findall(r'\b(mgdt_[a-z0-9_]+)\s*\(', hdr)) - {'mgdt_view', 'mgdt_stream'}
    assert declared, 'no declarations parsed'
    lib = _lib.lib()                       # dlopen; no GPU call
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/mgdt.h but not exported'
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert b'gfx950' in lib.mgdt_version()
    assert lib.mgdt_conv_packed_bytes(64, 64, 3, _lib.BF16) == 18 * 4 * 1024 + 64 * 4      # 576/32 chunks x 4 cout blocks (+ scale scratch)
    assert lib.mgdt_nms_workspace_bytes(2, 80, 6400, 0, 30000) == 2 * (8192 + 6400) * 8


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    lib = _lib.lib()
    v = _lib.View(None, 1, 8, 8, 8, 512, 64, 8, 1)
    assert lib.mgdt_conv2d_fwd(v, None, None, None, None, None, 3, 1, 1, None, None, v, 0, None) == -4    # null pointers
    assert b'null' in lib.mgdt_last_error()
    assert lib.mgdt_nms_fwd(None, 1, 1, 1, 0.5, 0.5, None, 0, 0, 0, 1, 1, 1.0, None, None, None, None, None, 0, None) == -4


@pytest.mark.parametrize('name', [n for n in CONFIGS if not n.endswith('_hidc128')])     # *_hidc128: our own s-scale variant, no reference file
def test_builtin_graphs_equal_reference_yaml_files(name):
    import yaml
    path = os.path.join(REF_YAML, name + '.yaml')
    if not os.path.exists(path):
        pytest.skip('reference tree not present on this box')
    ref = yaml.safe_load(open(path))
    mine = get_config(name)
    for k in ('nc', 'scales', 'backbone', 'head'):
        assert ref[k] == mine[k]


def test_parse_reference_yaml_files():
    """yaml_model_load + parse_model on the reference's own files (scale from the file stem, tasks.py:702-735)."""
    path = os.path.join(REF_YAML, 'mspa_c2f_gd_yolov8n.yaml')
    if not os.path.exists(os.path.join(REF_YAML, 'mspa_c2f_gd_yolov8.yaml')):
        pytest.skip('reference tree not present on this box')
    m = tasks.DetectionModel(path, verbose=False)
    assert m.yaml['scale'] == 'n' and sum(p.numel() for p in m.parameters()) == 1314298
    assert m.save == [2, 2, 4, 4, 6, 6, 9, 11, 15] and m.stride.tolist() == [8.0]
    s = tasks.DetectionModel(os.path.join(REF_YAML, 'mspa_c2f_gd_yolov8s.yaml'), verbose=False)
    assert sum(p.numel() for p in s.parameters()) == 4149424


def test_side_branch_plan_is_a_dependency_fact_of_the_layer_list():
    """BaseModel._side_branch: the run of layers launched on the second stream must read nothing newer than `dep`, and every layer between
    dep and the run must be independent of the run (they are earlier in the list).  MSPA-GD graphs: (12, 13, 6); stock yolov8: no such run;
    hooks on the layers involved or ops.SIDE_STREAM = False switch it off."""
    from mgdt_yolo_amd import ops
    for name in ('mspa_c2f_gd_yolov8', 'mspa_c2f_gd_tood_yolov8'):
        m = tasks.DetectionModel(get_config(name, 'n'), verbose=False)
        s0, s1, dep = m._side_branch()
        assert (s0, s1, dep) == (12, 13, 6)
        absf = lambda l: [l.i + f if f < 0 else f for f in ([l.f] if isinstance(l.f, int) else l.f)]
        ext = {j for k in range(s0, s1 + 1) for j in absf(m.model[k]) if j < s0}
        assert ext and max(ext) == dep and dep < s0 - 1
        assert any(j in range(s0, s1 + 1) for j in absf(m.model[s1 + 1])), 'the layer behind the run is its consumer (the join point)'
        h = m.model[12].register_forward_hook(lambda *a: None)
        assert m._side_branch() is None
        h.remove()
        assert m._side_branch() == (12, 13, 6)
        ops.SIDE_STREAM = False
        try:
            assert m._side_branch() is None
        finally:
            ops.SIDE_STREAM = True
    assert tasks.DetectionModel(get_config('yolov8', 'n'), verbose=False)._side_branch() is None


def test_model_structure_and_state_dict_names():
    m = tasks.DetectionModel(get_config('yolov8', 'n'), verbose=False)
    assert sum(p.numel() for p in m.parameters()) == 2847732 and m.stride.tolist() == [8.0, 16.0, 32.0]
    g = tasks.DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n'), verbose=False)
    keys = set(g.state_dict())
    for k in ('model.0.conv.weight', 'model.0.bn.running_var', 'model.2.convs.3.conv.weight', 'model.2.bottleneck.0.cv2.bn.bias',
              'model.2.attention.fc1.weight', 'model.2.attention.fc2.bias', 'model.9.cv2.conv.weight', 'model.11.conv.1.dwconv.weight',
              'model.11.conv.1.norm.weight', 'model.11.conv.2.grn.gamma', 'model.11.conv.3.pwconv2.bias', 'model.13.cv1.conv.weight',
              'model.13.cv_fuse.bn.weight', 'model.14.global_act.conv.weight', 'model.15.m.0.cv1.conv.weight', 'model.16.cv2.0.2.bias',
              'model.16.cv3.0.0.conv.weight', 'model.16.dfl.conv.weight'):
        assert k in keys, k
    assert 'model.13.cv2.conv.weight' not in keys            # Identity branches (block.py:312-314)
    det = g.model[-1]
    assert (det.nc, det.nl, det.reg_max, det.no) == (80, 1, 4, 96)
    assert abs(det.cv3[0][2].bias.data[0].item() - np.log(5 / 80 / (640 / 8) ** 2)) < 1e-6   # bias_init, head.py:186
    assert all(b.eps == 1e-3 and b.momentum == 0.03 for b in g.modules() if isinstance(b, torch.nn.BatchNorm2d))
    with pytest.raises(KeyError):
        bad = get_config('yolov8', 'n'); bad['head'][0][2] = 'NoSuchModule'; tasks.DetectionModel(bad, verbose=False)


def test_state_dict_matches_reference_keys_and_shapes(golden):
    for tag, name in GI.E2E_MODELS.items():
        g = golden('e2e_' + tag)
        if 'sd_keys' not in g:
            pytest.skip('fixture without key list')
        ref = dict(zip(str(g['sd_keys']).split('\n'), [tuple(int(v) for v in s.split(',') if v) for s in str(g['sd_shapes']).split('\n')]))
        mine = {k: tuple(v.shape) for k, v in tasks.DetectionModel(get_config(name, 'n'), verbose=False).state_dict().items()}
        assert mine == ref


def test_helpers():
    from mgdt_yolo_amd.yolo.utils.torch_utils import make_divisible
    assert tasks.guess_model_scale('x/mspa_c2f_gd_yolov8s.yaml') == 's' and tasks.guess_model_scale('foo.yaml') == ''
    assert make_divisible(33, 8) == 40 and make_divisible(64 * 0.25, 8) == 16
    assert tasks.yaml_model_load('yolov8n.yaml')['scale'] == 'n'            # built-in graph by name
    with pytest.raises(FileNotFoundError):
        tasks.yaml_model_load('nope.yaml')


def test_nms_argument_validation_and_cpu_refusal():
    from mgdt_yolo_amd.yolo.utils.ops import non_max_suppression
    with pytest.raises(AssertionError, match='Invalid Confidence'):
        non_max_suppression(torch.zeros(1, 84, 10), conf_thres=1.2)
    with pytest.raises(AssertionError, match='Invalid IoU'):
        non_max_suppression(torch.zeros(1, 84, 10), iou_thres=-0.1)
    with pytest.raises(RuntimeError, match='no CPU'):
        non_max_suppression(torch.zeros(1, 84, 10))
    m = tasks.DetectionModel(get_config('yolov8', 'n'), verbose=False).eval()
    with pytest.raises(RuntimeError, match='no CPU'):
        m(torch.zeros(1, 3, 64, 64))


def _rank_main(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from mgdt_yolo_amd import parallel
    from mgdt_yolo_amd.seeding import seeded_images
    dist = parallel.init('gloo')
    r, _, w = parallel.env_rank()
    x = seeded_images(2, 8, 8, seed=parallel.shard_seed(100, r))          # disjoint shards
    elapsed = 1.0 + 0.5 * r                                              # rank 1 is the slow one
    dist.barrier()
    t = parallel.max_over_ranks(elapsed)
    q.put((r, float(x.sum()), t, parallel.aggregate_throughput(32, 10, t, w)))
    dist.destroy_process_group()


def test_multi_rank_bench_path_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(30) for p in ps]
    assert res[0][1] != res[1][1]                                         # different data per rank
    assert res[0][2] == res[1][2] == 1.5                                  # MAX over ranks
    assert res[0][3] == pytest.approx(2 * 32 * 10 / 1.5)                  # whole-job aggregate


def _rank_allreduce(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from mgdt_yolo_amd import parallel
    dist = parallel.init('gloo')
    flat = torch.full((1000,), float(rank + 1))          # this rank's flat gradient buffer
    parallel.all_reduce_mean_(flat)
    q.put((rank, flat[0].item(), flat[-1].item()))
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_gloo_world2():
    """The training exchange step (one mean all-reduce over the flat gradient buffer) with 2 ranks on CPU/gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    ps = [ctx.Process(target=_rank_allreduce, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(30) for p in ps]
    assert res == [(0, 1.5, 1.5), (1, 1.5, 1.5)]


def test_tood_config_builds_and_oracle_dcn_reduces_to_conv():
    """a15 plumbing on CPU: the TOOD YAML rows parse (hidc passed unscaled like the reference), the state_dict carries the reference's
    key names, and the oracle's deformable conv with zero offsets / unit mask is the plain 3x3 conv (its only pin: mmcv is absent)."""
    import torch
    import torch.nn.functional as F
    from mgdt_yolo_amd.models import get_config
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from oracle import layers as OL, tood
    cfg = get_config('mspa_c2f_gd_tood_yolov8', 'n', 80)
    m = DetectionModel(cfg, verbose=False)
    head = m.model[-1]
    assert type(head).__name__ == 'TOODHead' and head.reg_max == 16 and head.no == 80 + 64 and float(m.stride[0]) == 8.0
    keys = set(m.state_dict())
    for k in ('share_conv.0.gn.weight', 'cls_decomp.reduction_conv.conv.bias', 'DyDCNV2.conv.weight', 'DyDCNV2.norm.bias', 'spatial_conv_offset.bias',
              'cls_prob_conv2.weight', 'cv2.bias', 'cv3.weight', 'scale.0.scale', 'dfl.conv.weight'):
        assert f'model.16.{k}' in keys, k
    rows, _ = OL.plan_from_yaml(cfg, 3)
    assert rows[-1]['type'] == 'TOODHead' and rows[-1]['args'] == [80, 64, [64]]
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(2, 8, 7, 9, generator=g), torch.randn(6, 8, 3, 3, generator=g)
    y = tood.modulated_deform_conv3x3(x, torch.zeros(2, 18, 7, 9), torch.ones(2, 9, 7, 9), w)
    assert (y - F.conv2d(x, w, None, 1, 1)).abs().max() < 1e-4
    # an integer shift of every kernel point by (+1, 0) samples the image one row lower, zero outside
    off = torch.zeros(2, 18, 7, 9)
    off[:, 0::2] = 1.0
    y1 = tood.modulated_deform_conv3x3(x, off, torch.ones(2, 9, 7, 9), w)
    xs = torch.zeros_like(x)
    xs[:, :, :-1] = x[:, :, 1:]
    ref = F.conv2d(F.pad(xs, (1, 1, 1, 1)), w)
    assert (y1[:, :, 1:-1] - ref[:, :, 1:-1]).abs().max() < 1e-4      # interior rows: the borders differ by what each form pads


# ------------------------------------------------------------------------------------------------ trainer host logic (no GPU call)
@pytest.mark.parametrize('tag', list(GI.E2E_MODELS))
def test_optimizer_groups_match_the_reference_build_optimizer(golden, tag):
    """param_groups() vs the membership the reference's own BaseTrainer.build_optimizer produced on the reference's model
    (tests/golden/optim_groups.npz, generated by gen_golden.py:optim_groups)."""
    from mgdt_yolo_amd.yolo.engine.trainer import param_groups
    g = golden('optim_groups')
    m = tasks.DetectionModel(get_config(GI.E2E_MODELS[tag], 'n'), verbose=False)
    mine = param_groups(m)
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    ref = {}
    for gi, key in ((2, 'bias'), (0, 'decay'), (1, 'norm')):
        for name in str(g[f'{tag}_{key}']).split('\n'):
            ref[name] = gi
    assert {k: v for k, v in mine.items() if k in trainable} == {k: v for k, v in ref.items() if k in trainable}
    assert set(ref) >= trainable
    if tag == 'mspa_c2f_gd_n':      # the reference rule decays the custom LayerNorm weight and GRN gamma/beta (plain nn.Modules)
        assert mine['model.11.conv.1.norm.weight'] == 0 and mine['model.11.conv.1.grn.gamma'] == 0 and mine['model.11.conv.1.grn.beta'] == 0
        assert mine['model.11.conv.1.norm.bias'] == 2 and mine['model.0.bn.weight'] == 1


def test_trainer_warmup_and_accumulate_schedule():
    """lr / bias-lr / momentum / accumulate per iteration as yolo/engine/trainer.py:250-251,284,317-326 computes them."""
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    m = tasks.DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 4), verbose=False)
    tr = DetectionTrainer(m, lr0=0.01, batch_size=16, nb=100, epochs=10)
    assert tr.accumulate == 4 and tr.nw == 300
    wd = tr.state.wd
    off, k = tr.state.offsets['model.0.conv.weight']
    assert torch.allclose(wd[off:off + k], torch.full((k,), 5e-4 * 16 * 4 / 64))            # decay scaled by batch*accumulate/nbs
    off, k = tr.state.offsets['model.0.bn.bias']
    assert (wd[off:off + k] == -1).all()                                                     # bias group marker
    off, k = tr.state.offsets['model.0.bn.weight']
    assert (wd[off:off + k] == 0).all()
    tr.ni = 0; tr.warmup(0)
    assert (tr.lr, tr.lr_bias, tr.mom, tr.accumulate) == (0.0, 0.1, 0.8, 1)
    tr.ni = 150; tr.warmup(0)
    assert tr.lr == pytest.approx(0.005) and tr.lr_bias == pytest.approx(0.055) and tr.mom == pytest.approx(0.8685) and tr.accumulate == 2
    tr.ni = 300; tr.warmup(2)
    lf2 = (1 - 2 / 10) * (1 - 0.01) + 0.01
    assert tr.lr == pytest.approx(0.01 * lf2) and tr.lr_bias == pytest.approx(0.01 * lf2) and tr.mom == pytest.approx(0.937) and tr.accumulate == 4
    tr.ni = 301; tr.warmup(3)
    assert tr.lr == pytest.approx(0.01 * ((1 - 3 / 10) * 0.99 + 0.01)) and tr.accumulate == 4


def _rank_trainer_exchange(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from mgdt_yolo_amd import parallel
    from mgdt_yolo_amd.seeding import seed_state_dict_
    from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
    dist = parallel.init('gloo')
    m = seed_state_dict_(tasks.DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 4), verbose=False), rank)     # ranks start DIFFERENT
    tr = DetectionTrainer(m, world_size=world)
    st, ex = tr.state, tr.exchange
    start = st.data.clone()
    n_layers = len(m.model)
    res = []
    for step in range(2):                       # two DetectionTrainer.step-shaped exchanges with unequal per-rank gradients
        g = torch.Generator().manual_seed(100 * step + rank)
        full = torch.randn(st.n_param, generator=g)
        st.grad.zero_()
        ends = [st.layer_end.get(i) for i in range(n_layers)]
        for i in reversed(range(n_layers)):     # the reverse pass: layer i's slice of the flat buffer is written, then the hook fires
            hi = ends[i]
            if hi is not None:
                lo = max([e for e in ends[:i] if e is not None], default=0)
                st.grad[lo:hi] = full[lo:hi]
            ex.layer_done(i)
        ex.finish()
        res.append(st.grad.clone())
    q.put((rank, start.numpy(), [r.numpy() for r in res], ex.slices))
    dist.destroy_process_group()


def test_trainer_broadcast_and_bucketed_all_reduce_gloo_world2():
    """DDP-style start (rank 0's parameters and buffers everywhere, trainer.py:225) and the layer-ordered bucketed all-reduce of the flat
    gradient buffer hung on BaseModel.backward's layer_done hook: two exchanges, unequal gradients, result = mean over ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 90
    ps = [ctx.Process(target=_rank_trainer_exchange, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=180) for _ in ps), key=lambda t: t[0])
    [p.join(30) for p in ps]
    (_, s0, g0, slices), (_, s1, g1, _) = res
    assert np.array_equal(s0, s1)                                          # rank 1 now holds rank 0's weights
    assert len(slices) >= 2 and slices[0][1] == 0 and all(a[2] == b[1] for a, b in zip(slices[:-1], slices[1:]))   # contiguous cover
    n = g0[0].shape[0]
    assert slices[-1][2] == n
    for step in range(2):
        want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(2)).numpy() / 2
        np.testing.assert_allclose(g0[step], want, rtol=1e-6, atol=1e-7)
        assert np.array_equal(g0[step], g1[step])


def test_bench_refuses_to_report_fewer_ranks_than_requested():
    """`python bench.py --gpus 2`: under a launcher with the wrong world size it exits non-zero before touching the GPU; without a launcher
    it starts the ranks itself as a child (torch.distributed.run) - on this GPU-less box they fail, so the parent must exit non-zero and
    print no JSON line (never a 1-GPU number under n_gpus 2)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, 'bench.py')
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'refusing to report' in r.stderr and '"value"' not in r.stdout
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"value"' not in r.stdout
    assert 'torch.distributed' in r.stderr or 'ChildFailedError' in r.stderr or 'Traceback' in r.stderr      # the child launcher really ran


def test_make_anchors_equals_the_pinned_oracle():
    """yolo/utils/tal.py:make_anchors (flat-index construction) == oracle.layers.make_anchors (pinned through the e2e fixtures), odd sizes."""
    import torch
    from mgdt_yolo_amd.yolo.utils.tal import make_anchors
    from oracle.layers import make_anchors as oracle_anchors
    shapes, strides = [(5, 7), (3, 4), (1, 1), (80, 80)], [8, 16, 32, 8]
    a, s = make_anchors([torch.zeros(2, 4, h, w) for h, w in shapes], strides)
    b, t = oracle_anchors(shapes, strides)
    assert torch.equal(a, b) and torch.equal(s, t) and a.shape == (sum(h * w for h, w in shapes), 2)
