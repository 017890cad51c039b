"""hipGraph instances of the bench step: do separately captured instances agree when replayed one at a time / concurrently?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda:0')
model = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().to(dev).set_compute_dtype(torch.bfloat16)
x = seeded_images(32, 640, 640, seed=100).to(dev).to(torch.bfloat16)
def step():
    y, _ = model(x)
    return (y,) + tuple(ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680))
def same(o, o0):
    cnt = o0[3]
    valid = torch.arange(o0[2].shape[1], device=dev)[None, :] < cnt[:, None]
    return (torch.equal(o[0], o0[0]), torch.equal(o[3], cnt), bool(torch.equal(o[2][valid], o0[2][valid])) if torch.equal(o[3], cnt) else False)
with torch.no_grad():
    o0 = [t.clone() for t in step()]
    torch.cuda.synchronize()
    junk = [torch.full((64 << 20,), float('nan'), device=dev) for _ in range(8)]   # poison whatever the allocator hands out next
    del junk
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs, outs = [], []
    for i in range(S):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs.append(step())
        graphs.append(g)
    for i, g in enumerate(graphs):
        g.replay()
        torch.cuda.synchronize()
        print('serial replay', i, '(y, counts, kept) equal to eager:', same(outs[i], o0))
    streams = [torch.cuda.Stream() for _ in range(S)]
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    bad = 0
    for rep in range(R):
        for k in range(3):                      # three rounds back to back, as the bench loop does
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
        res = [same(o, o0) for o in outs]
        bad += sum(not all(r) for r in res)
    print(f'concurrent replays on {S} streams: {bad} mismatching instance results in {R} x {S}')
