"""Training-step throughput of the HIP trainer (fp32, one GPU): python tools/train_bench.py [batch] [imgsz] [steps]
One step = uint8 batch -> forward (batch-stat BN) -> v8DetectionLoss + assigner -> explicit backward -> clip + SGD(nesterov) + EMA."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images, seeded_labels
from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda:0')
model = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).to(dev)
tr = DetectionTrainer(model)
batch = seeded_labels(B, 80, seed=1)
batch['img'] = (seeded_images(B, S, S, seed=2) * 255).round().to(torch.uint8).to(dev)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
losses = []
for i in range(3):
    losses.append(float(tr.step(batch)[0]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    l = tr.step(batch)[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'train: batch {B} @ {S}x{S} fp32, {K} steps: {dt / K * 1e3:.1f} ms/step = {B * K / dt:.0f} images/s; loss {losses[0]:.3f} -> {float(l):.3f}')
