"""Which layer first differs between concurrently replayed hipGraph instances?  python tools/graph_bisect.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
S = 6
dev = torch.device('cuda:0')
model = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().to(dev).set_compute_dtype(torch.bfloat16)
x = seeded_images(32, 640, 640, seed=100).to(dev).to(torch.bfloat16)
nl = len(model.model)

def upto(k):
    def f():
        y, t = [], x
        for m in model.model:
            if m.f != -1:
                t = y[m.f] if isinstance(m.f, int) else [t if j == -1 else y[j] for j in m.f]
            t = m(t)
            y.append(t if m.i in model.save else None)
            if m.i == k:
                break
        t = t[0] if isinstance(t, (tuple, list)) else t
        return t
    return f

with torch.no_grad():
    model(x)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(S)]
    for k in range(nl):
        step = upto(k)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ref = step().clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, outs = [], []
        for i in range(S):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs.append(step())
            graphs.append(g)
        bad = 0
        for rep in range(4):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
            torch.cuda.synchronize()
            bad += sum(not torch.equal(o, ref) for o in outs)
        print(f'layer {k:2d} {type(model.model[k]).__name__:28s} mismatching replays: {bad} / {4 * S}', flush=True)
        del graphs, outs
