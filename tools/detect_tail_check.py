"""Detect tail (fused kernel) vs the unfused head for several class counts / map sizes, and its best-class keys vs a scan of y."""
import sys, torch
sys.path.insert(0, '/root/repo')
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.models import get_config
from mgdt_yolo_amd.nn.tasks import DetectionModel
from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
for name in ('mspa_c2f_gd_yolov8', 'yolov8'):
    for nc in (4, 20, 36, 80):
        m = seed_state_dict_(DetectionModel(get_config(name, 'n', nc), verbose=False), 0).eval().cuda().set_compute_dtype(torch.bfloat16)
        for shape in ((2, 320, 256), (1, 224, 352)):
            x = seeded_images(*shape, seed=5).cuda().to(torch.bfloat16)
            with torch.no_grad():
                y1, f1 = m(x)
                k1 = ops._best_keys_of(y1, y1.shape[0], y1.shape[2])
                ops.FUSED_DETECT_TAIL = False
                try:
                    y0, f0 = m(x)
                finally:
                    ops.FUSED_DETECT_TAIL = True
            d = (y1.float() - y0.float()).abs()
            # best keys vs a scan of y1
            ok = True
            if k1 is not None:
                sc = y1[:, 4:, :]
                best, cls = sc.max(1)
                a = torch.arange(y1.shape[2], device=y1.device)[None, :]
                ref = ((0xFFFFFFFF - (best.contiguous().view(torch.int32).long() & 0xFFFFFFFF)) << 32) | (a * nc + cls)
                ok = bool((k1.view(torch.int64) == ref).all())
            print(name, nc, shape, 'max |dy| box %.3g conf %.3g' % (d[:, :4].max().item(), d[:, 4:].max().item()), 'feats equal', all(torch.equal(a_, b_) for a_, b_ in zip(f1, f0)), 'keys ok', ok)
