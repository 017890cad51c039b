"""conv3x3_lds_kernel, bf16 vs e4m3 form, on the two Detect-branch shapes of the bench: phase stamps (MGDT_C3_DBG=1) and event timing.
   MGDT_C3_DBG=1 python tools/c3_q8_dbg.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops

dev = 'cuda:0'
for cin, cout in ((64, 96), (80, 80)):
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    x = torch.randn(32, cin, 80, 80, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pk16 = ops.PackedConv(w, None, None, 3, torch.bfloat16)
    pk8 = ops.PackedConvFp8(w, None, None, 3, 16.0)
    for name, fn in (('bf16', lambda: ops.conv2d(x, pk16, 1, ops.ACT_SILU)), ('fp8', lambda: ops.conv2d_fp8(x, pk8, 1, ops.ACT_SILU))):
        print(f'--- {cin}->{cout} {name}', file=sys.stderr, flush=True)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f'{cin}->{cout} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch', file=sys.stderr, flush=True)
