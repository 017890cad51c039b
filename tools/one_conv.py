"""Run one conv shape repeatedly (for rocprofv3 counter collection): python tools/one_conv.py cin cout k s h w [dtype] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
cin, cout, k, s, h, w = map(int, sys.argv[1:7])
dt = torch.bfloat16 if (len(sys.argv) < 8 or sys.argv[7] == 'bf16') else torch.float32
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
B = 32
x = torch.randn(B, cin, h, w, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
wt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
pk = ops.PackedConv(wt, torch.zeros(cout, device='cuda'), None, k, dt)
for _ in range(reps):
    y = ops.conv2d(x, pk, s, ops.ACT_SILU)
torch.cuda.synchronize()
print('done', tuple(y.shape))
