"""Per-launch table of one training step (eager, HIP events): python tools/train_profile.py [--top 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images, seeded_labels
from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
ap = argparse.ArgumentParser(); ap.add_argument('--top', type=int, default=40); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--amp', action='store_true'); a = ap.parse_args()
m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).cuda()
tr = DetectionTrainer(m, amp=a.amp)
batch = dict(img=(seeded_images(a.batch, 640, 640, seed=1) * 255).to(torch.uint8).cuda(), **seeded_labels(a.batch, 80, seed=2))
for _ in range(2):
    tr.step(batch)
with ops.profile() as p:
    tr.step(batch)
agg = {}
for name, meta, ms in p.rows:
    key = (name, str(meta['shape']) if meta else '')
    v = agg.setdefault(key, [0.0, 0, meta]); v[0] += ms; v[1] += 1
tot = sum(v[0] for v in agg.values())
print(f'{len(p.rows)} launches, {tot:.2f} ms (sum of launches, eager)')
for (name, shape), (ms, n, meta) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    extra = f' {meta["flops"] * n / ms / 1e9:8.1f} TF/s' if meta else ''
    print(f'{name:<24}{shape:<36}{n:>4} x {ms / n * 1e3:9.1f} us = {ms:7.3f} ms{extra}')
