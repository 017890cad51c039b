import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MGDT_NMS_DBG'] = '1'
from mgdt_yolo_amd import ops, _lib as L
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
import ctypes as C
m = seed_state_dict_(DetectionModel(get_config('mspa_c2f_gd_yolov8', 'n', 80), verbose=False), 0).eval().cuda().set_compute_dtype(torch.bfloat16)
with torch.no_grad():
    y, _ = m(seeded_images(32, 640, 640, seed=100).cuda())
b, ch, a = y.shape
lib = L.lib()
wsb = lib.mgdt_nms_workspace_bytes(b, ch - 4, a, 0, 30000)
ws = torch.zeros(wsb // 8, dtype=torch.int64, device='cuda')
out = torch.zeros(b, 300, 6, device='cuda'); kept = torch.zeros(b, 300, dtype=torch.int32, device='cuda'); cnt = torch.zeros(b, dtype=torch.int32, device='cuda')
best = ops._best_keys_of(y, b, a) if os.environ.get('NMS_BEST', '1') == '1' else None
print('best keys from the detect tail:', best is not None)
for _ in range(2):
    L.check(lib.mgdt_nms_fwd(ops.ptr(y), b, ch - 4, a, 0.25, 0.7, None, 0, 0, 0, 300, 30000, 7680.0, ops.ptr(out), ops.ptr(kept), ops.ptr(cnt), ops.ptr(best), ops.ptr(ws), wsb, ops.stream()))
torch.cuda.synchronize()
per = wsb // 8 // b
for i in (0, 1, 31):
    print('img', i, 'ticks(10ns) total / select+compact / sort / greedy, K:', ws[i * per:i * per + 5].tolist(), 'kept', int(cnt[i]),
          '| greedy cycles kept-list / chunk matrix / barrier / walk / barrier, steps:', ws[i * per + 5:i * per + 11].tolist())
