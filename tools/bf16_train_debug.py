import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images, seeded_labels
from mgdt_yolo_amd.yolo.engine.trainer import DetectionTrainer
nc, B, S = 4, 4, 96
batch = dict(img=(seeded_images(B, S, S, seed=2) * 255).to(torch.uint8), **seeded_labels(B, nc, seed=6, max_boxes=4, min_boxes=2))
batch['bboxes'][:, 2:] = batch['bboxes'][:, 2:] * 0.5 + 0.1
res = {}
for amp in (False, True):
    m = seed_state_dict_(DetectionModel(get_config(sys.argv[1] if len(sys.argv) > 1 else 'mspa_c2f_gd_yolov8', 'n', nc), verbose=False), 0).cuda()
    tr = DetectionTrainer(m, lr0=0.0, amp=amp)
    tr.step(batch)
    res[amp] = (tr.state.grad.clone(), dict(tr.state.offsets))
g32, off = res[False]; g16, _ = res[True]
for k, (o, n) in off.items():
    a, b = g32[o:o + n], g16[o:o + n]
    na, nb = a.norm().item(), b.norm().item()
    cos = torch.nn.functional.cosine_similarity(a, b, 0).item()
    flag = '' if (cos > 0.95 or na < 1e-4) else '   <<<<'
    print(f'{k:<50}{na:12.5f}{nb:12.5f}  cos {cos:8.4f}{flag}')
