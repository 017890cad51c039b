"""Phase times of the LDS-staged 3x3 convolution at the Detect-branch shape: MGDT_C3_DBG=1 python tools/conv3_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
w = torch.randn(96, 64, 3, 3).cuda() / 24; b = torch.zeros(96).cuda()
pk = ops.PackedConv(w, b, None, 3, torch.bfloat16)
x = torch.randn(32, 64, 80, 80).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
y = torch.empty(32, 96, 80, 80, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(3): ops.conv2d(x, pk, 1, ops.ACT_SILU, out=y)
