"""Determinism under concurrency: run the forward / NMS of the bench model on several streams at once and compare every result
with the serial one.  python tools/race_check.py [streams] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
model = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().to(dev).set_compute_dtype(torch.bfloat16)
x = seeded_images(32, 640, 640, seed=100).to(dev).to(torch.bfloat16)
nms = lambda y: ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680)
with torch.no_grad():
    y0, _ = model(x)
    o0 = nms(y0)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(S)]
    bad_y = bad_n = bad_n_same_y = 0
    for rep in range(R):
        ys, os_ = [], []
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                y, _ = model(x)
                ys.append(y)
                os_.append(nms(y))
        torch.cuda.synchronize()
        for y, o in zip(ys, os_):
            same_y = torch.equal(y, y0)
            bad_y += not same_y
            cnt = o0[2]
            valid = torch.arange(o0[1].shape[1], device=dev)[None, :] < cnt[:, None]
            same_n = torch.equal(o[2], cnt) and torch.equal(o[1][valid], o0[1][valid])
            bad_n += not same_n
            bad_n_same_y += (not same_n) and same_y
    print(f'streams {S} reps {R}: forward mismatches {bad_y}, nms mismatches {bad_n} (of which with identical forward output: {bad_n_same_y})')
    # NMS alone, concurrently, on the one serial forward output
    bad = 0
    for rep in range(R):
        os_ = []
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                os_.append(nms(y0))
        torch.cuda.synchronize()
        for o in os_:
            cnt = o0[2]
            valid = torch.arange(o0[1].shape[1], device=dev)[None, :] < cnt[:, None]
            bad += not (torch.equal(o[2], cnt) and torch.equal(o[1][valid], o0[1][valid]))
    print(f'nms alone, concurrent: mismatches {bad} of {R * S}')
