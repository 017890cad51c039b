"""Run one conv shape repeatedly (for rocprofv3 counter collection): python tools/one_conv.py cin cout k s h w [dtype] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
cin, cout, k, s, h, w = map(int, sys.argv[1:7])
dt = torch.bfloat16 if (len(sys.argv) < 8 or sys.argv[7] == 'bf16') else torch.float32
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
B = int(os.environ.get('B', 32))
wide = int(os.environ.get('SLICE', cin))   # > cin: x (and y) are channel slices of a wider NHWC tensor, as inside MSPA_C2f
xw = torch.randn(B, wide, h, w, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
x = xw[:, :cin]
wt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
pk = ops.PackedConv(wt, torch.zeros(cout, device='cuda'), None, k, dt)
yw = ops.new_act(B, max(wide, cout) if wide > cin else cout, h // s, w // s, dt, x.device)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
for i in range(reps):
    ev[i].record()
    y = ops.conv2d(x, pk, s, ops.ACT_SILU, out=yw[:, :cout])
ev[reps].record()
torch.cuda.synchronize()
print('done', tuple(y.shape), 'us/launch (last 3):', [round(ev[i].elapsed_time(ev[i + 1]) * 1e3, 1) for i in range(reps - 3, reps)])
