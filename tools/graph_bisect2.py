import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
dev = torch.device('cuda:0')
model = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().to(dev).set_compute_dtype(torch.bfloat16)
x = seeded_images(32, 640, 640, seed=100).to(dev).to(torch.bfloat16)
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
        return t, [v for v in y if v is not None]
    return f
with torch.no_grad():
    model(x); torch.cuda.synchronize()
    step = upto(10)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ref, refsaved = step(); ref = ref.clone(); refsaved = [v.clone() for v in refsaved]
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, saved = step()
    S = 6
    graphs, outs = [g], [(out, saved)]
    for _ in range(S - 1):
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            outs.append(step())
        graphs.append(gg)
    for i, (o, sv) in enumerate(outs):
        print('inst', i, 'out ptr', hex(o.data_ptr()), 'saved ptrs', [hex(v.data_ptr()) for v in sv])
    streams = [torch.cuda.Stream() for _ in range(S)]
    for rep in range(6):
        for st, gg in zip(streams, graphs):
            with torch.cuda.stream(st):
                gg.replay()
        torch.cuda.synchronize()
        for i, (o, sv) in enumerate(outs):
            if not torch.equal(o, ref):
                d = (o.float() - ref.float()).abs().amax(dim=(0, 2, 3))
                nz = torch.nonzero(d).flatten().tolist()
                dn = (o.float() - ref.float()).abs().amax(dim=(1, 2, 3))
                print(f'rep {rep} inst {i}: mismatch channels {nz[:4]}..{nz[-2:]} n={len(nz)} images {torch.nonzero(dn).flatten().tolist()[:8]} saved equal:',
                      [bool(torch.equal(a_, b_)) for a_, b_ in zip(sv, refsaved)])
