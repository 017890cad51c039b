"""Run one ConvNeXtV2 block (bf16, fused MLP) repeatedly under rocprofv3: python tools/one_cnx.py [dim h w reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd.nn.modules import ConvNeXtV2_Block
from mgdt_yolo_amd.seeding import seed_state_dict_
dim, h, w, reps = (list(map(int, sys.argv[1:5])) + [96, 40, 40, 6][len(sys.argv) - 1:])[:4]
B = int(os.environ.get('B', 32))
m = seed_state_dict_(ConvNeXtV2_Block(dim), 0).eval().cuda()
x = torch.randn(B, dim, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(reps):
        y = m(x)
torch.cuda.synchronize()
print('done', tuple(y.shape))
