"""Debug aid: single-stream forward vs the captured two-branch graph (BaseModel._side_branch); prints where the outputs differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops  # noqa: E402
from mgdt_yolo_amd.models import get_config  # noqa: E402
from mgdt_yolo_amd.nn.tasks import DetectionModel  # noqa: E402
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images  # noqa: E402

B, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 320, 256))]
dt = torch.bfloat16
m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().cuda().set_compute_dtype(dt)
xs = [seeded_images(B, H, W, seed=s).cuda().to(dt) for s in (3, 4)]
taps = {}


def tap(i):
    def hook(mod, inp, out):
        taps[i] = out
    return hook


with torch.no_grad():
    ops.SIDE_STREAM = False
    ref = []
    for x in xs:
        y, feats = m(x)
        ref.append([y.clone()] + [f.clone() for f in feats])
    ops.SIDE_STREAM = True
    for rep in range(3):
        y, feats = m(xs[0])
        print('eager two-stream equal:', all(torch.equal(a, b) for a, b in zip([y] + list(feats), ref[0])))
    xin = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(xin)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y, feats = m(xin)
    out = [y] + list(feats)
    nbad_rep = 0
    for rep in range(int(os.environ.get('REPS', 6))):
        k = rep % 2
        xin.copy_(xs[k])
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(out, ref[k]):
            bad = (a != b).nonzero()
            if len(bad):
                nbad_rep += 1
                print(f'replay {rep} input {k}: tensor {tuple(a.shape)} {len(bad)} of {a.numel()} differ; max |d| {(a.float() - b.float()).abs().max().item():.3g}; first {bad[:4].tolist()}')
    print('replays with a wrong tensor:', nbad_rep)

# ---- per-layer: which layer's output differs first in the replayed graph (clones recorded inside the capture)
print('per-layer comparison')


def flat(o):
    if o is None:
        return []
    if torch.is_tensor(o):
        return [o]
    return [t for e in o for t in flat(e)]


store = {}
for layer in m.model:
    def wrap(layer=layer, fwd=layer.forward):
        def f(*a, **k):
            o = fwd(*a, **k)
            store[layer.i] = [t.clone() for t in flat(o)]
            return o
        return f
    layer.forward = wrap()
with torch.no_grad():
    ops.SIDE_STREAM = False
    m(xs[0]); torch.cuda.synchronize()
    ref_l = {i: [t.clone() for t in v] for i, v in store.items()}
    ops.SIDE_STREAM = True
    store.clear()
    xin = xs[0].clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m(xin)
    gstore = dict(store)
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        line = []
        for i in sorted(store):
            nbad = sum(int((a != b).sum().item()) for a, b in zip(store[i], ref_l[i]))
            line.append(f'{i}:{nbad}')
        print('replay', rep, ' '.join(line))

# ---- are the wrong values of layer 13 the PREVIOUS replay's values (alternating inputs)?
print('stale-value check (layer 13 output, alternating inputs)')
with torch.no_grad():
    ops.SIDE_STREAM = False
    refs = []
    for x in xs:
        m(x); torch.cuda.synchronize()
        refs.append({i: [t.clone() for t in v] for i, v in store.items()})
    ops.SIDE_STREAM = True
    for rep in range(60):
        k = rep % 2
        xin.copy_(xs[k]); g.replay(); torch.cuda.synchronize()
        for i in (12, 13, 15, 16):
            a, b, o = gstore[i][0], refs[k][i][0], refs[1 - k][i][0]
            bad = a != b
            nb = int(bad.sum())
            if nb:
                idx = bad.nonzero()
                print(f'replay {rep} layer {i} {tuple(a.shape)}: {nb} wrong, of which equal to the other input\'s value: {int((a[bad] == o[bad]).sum())}; odd channels {int((idx[:, 1] % 2).sum())}; channels {sorted(set(idx[:, 1].tolist()))[:40]}; x%4 {sorted(set((idx[:, 3] % 4).tolist()))}; y {sorted(set(idx[:, 2].tolist()))[:12]}; n {sorted(set(idx[:, 0].tolist()))}')
