"""Debug aid: S captured instances of the inference step (forward + NMS) replayed concurrently on S streams vs their serial replays;
per-layer clones recorded inside the captures show which layer differs first.   python tools/inflight_dbg.py [B H W S reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops  # noqa: E402
from mgdt_yolo_amd.models import get_config  # noqa: E402
from mgdt_yolo_amd.nn.tasks import DetectionModel  # noqa: E402
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images  # noqa: E402

B, H, W, S, REPS = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 640, 640, 2, 20))]
dt = torch.bfloat16
m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().cuda().set_compute_dtype(dt)
xs = [seeded_images(B, H, W, seed=3 + s).cuda().to(dt) for s in range(S * int(os.environ.get('GPL', 1)))]


def flat(o):
    if o is None:
        return []
    if torch.is_tensor(o):
        return [o]
    return [t for e in o for t in flat(e)]


store = {}
GPL = int(os.environ.get('GPL', 1))             # graphs per lane (sharing the lane's pool, replayed one after the other)
NOSYNC = os.environ.get('NOSYNC', '0') == '1'   # queue all rounds without host synchronisation (compare after the last one)
for layer in (m.model if os.environ.get('NOCLONE', '0') != '1' else []):
    def wrap(layer=layer, fwd=layer.forward):
        def f(*a, **k):
            o = fwd(*a, **k)
            store[layer.i] = [t.clone() for t in flat(o)]
            return o
        return f
    layer.forward = wrap()


def step(x):
    y, _ = m(x)
    o = ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680)
    store['nms'] = [o[2].clone(), o[0].clone()]
    return o


with torch.no_grad():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for j in range(S):
            with ops.lane(j):
                step(xs[j]); step(xs[j])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs, stores = [], []
    pools = [torch.cuda.graph_pool_handle() for _ in range(S)]
    for r in range(S * GPL):
        g = torch.cuda.CUDAGraph()
        store.clear()
        with ops.lane(r % S), torch.cuda.graph(g, pool=pools[r % S]):
            step(xs[r])
        graphs.append(g)
        stores.append(dict(store))
    refs = []
    for r in range(S * GPL):
        graphs[r].replay(); torch.cuda.synchronize()
        refs.append({k: [t.clone() for t in v] for k, v in stores[r].items()})
    for r in range(S * GPL):
        graphs[r].replay(); torch.cuda.synchronize()
        assert all(torch.equal(a, b) for k in refs[r] for a, b in zip(stores[r][k], refs[r][k])), 'serial replay is not reproducible'
    lanes = [torch.cuda.Stream() for _ in range(S)]
    nbad = 0

    def check(rep):
        global nbad
        for r in range(S * GPL):
            line = []
            for k in stores[r]:
                bad = sum(int((a != b).sum().item()) for a, b in zip(stores[r][k], refs[r][k]))
                if bad:
                    line.append(f'{k}:{bad}')
            if line:
                nbad += 1
                print(f'rep {rep} graph {r} (lane {r % S}): ' + ' '.join(line))

    for rep in range(REPS):
        for r in range(S * GPL):
            with torch.cuda.stream(lanes[r % S]):
                graphs[r].replay()
        if not NOSYNC:
            torch.cuda.synchronize()
            check(rep)
    torch.cuda.synchronize()
    if NOSYNC:
        check(REPS - 1)
    print(f'{nbad} wrong graph replays; S {S} graphs/lane {GPL} reps {REPS} nosync {NOSYNC} SIDE_STREAM {ops.SIDE_STREAM}')
