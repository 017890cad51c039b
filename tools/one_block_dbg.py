"""Phase stamps of the fused block kernel at the bench shapes: MGDT_CSP_DBG=1 python tools/one_block_dbg.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgdt_yolo_amd.nn.modules import MSPA_C2f, C2f
from mgdt_yolo_amd.seeding import seed_state_dict_
DEV = 'cuda:0'
for kind, c, n, hw in [('mspa', 32, 1, 160), ('mspa', 64, 2, 80), ('mspa', 128, 2, 40), ('mspa', 256, 1, 20), ('c2f', 256, 1, 80)]:
    m = seed_state_dict_(MSPA_C2f(c, c, n, True) if kind == 'mspa' else C2f(c, 64, n, False), 1).eval().to(DEV)
    for sub in m.modules():
        if hasattr(sub, 'out_dtype'):
            sub._cdtype = torch.bfloat16
    x = torch.randn(32, c, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            m(x)
    torch.cuda.synchronize()
