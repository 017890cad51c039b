import torch, sys
sys.path.insert(0, '.')
from mgdt_yolo_amd import ops
x = torch.randn(32, 96, 40, 40, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(49, 96, device='cuda'); b = torch.randn(96, device='cuda'); lw = torch.ones(96, device='cuda'); lb = torch.zeros(96, device='cuda')
for _ in range(3):
    y = ops.dwconv7_ln(x, w, b, lw, lb, 1e-6)
torch.cuda.synchronize()
