"""Per-launch table of one forward(+NMS) step, timed with HIP events on the launch stream (eager launches).
   python tools/profile_ops.py [--dtype bf16|f32] [--batch 32] [--imgsz 640] [--model mspa_c2f_gd_yolov8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops  # noqa: E402
from mgdt_yolo_amd.models import get_config  # noqa: E402
from mgdt_yolo_amd.nn.tasks import DetectionModel  # noqa: E402
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--imgsz', type=int, default=640)
ap.add_argument('--model', default='mspa_c2f_gd_yolov8')
ap.add_argument('--scale', default='n')
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
m = seed_state_dict_(DetectionModel(get_config(a.model, a.scale, 80), verbose=False), 0).eval().cuda().set_compute_dtype(dt)
x = seeded_images(a.batch, a.imgsz, a.imgsz, seed=100).cuda().to(dt)      # the image in the compute dtype, as bench.py feeds it
with torch.no_grad():
    for _ in range(2):
        y, _ = m(x)
        ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680)
    with ops.profile() as p:
        for _ in range(a.reps):
            y, _ = m(x)
            ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680)
n = len(p.rows) // a.reps
print(f'{n} launches per step; dtype {a.dtype} batch {a.batch} {a.imgsz}x{a.imgsz}')
tot = 0.0
print(f'{"#":>3} {"op":<26}{"shape (b,cin,h,w,cout,k,s)":<34}{"us":>9}{"GB/s":>9}{"TF/s":>8}')
for i in range(n):
    name, meta = p.rows[i][0], p.rows[i][1]
    us = sum(p.rows[i + r * n][2] for r in range(a.reps)) / a.reps * 1e3
    tot += us
    if meta:
        print(f'{i:>3} {name:<26}{str(meta["shape"]):<34}{us:9.1f}{meta["bytes"] / us / 1e3:9.0f}{meta["flops"] / us / 1e6:8.1f}')
    else:
        print(f'{i:>3} {name:<26}{"":<34}{us:9.1f}')
print(f'total {tot:.1f} us per step (sum of launches, eager)')
