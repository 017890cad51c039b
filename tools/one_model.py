"""Run a model's eval forward a few times under rocprofv3: python tools/one_model.py <config> [scale] [batch] [imgsz] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
name = sys.argv[1]
scale = sys.argv[2] if len(sys.argv) > 2 else 'n'
b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
s = int(sys.argv[4]) if len(sys.argv) > 4 else 640
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
m = seed_state_dict_(DetectionModel(get_config(name, scale, 80), verbose=False), 0).eval().cuda().set_compute_dtype(torch.bfloat16)
x = seeded_images(b, s, s, seed=1).cuda().to(torch.bfloat16)
with torch.no_grad():
    for _ in range(reps):
        y, _ = m(x)
torch.cuda.synchronize()
print('done', tuple(y.shape))
