"""Time one MSPA_C2f / C2f block (fused launch vs launch chain) at the bench shapes: python tools/one_block.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.nn.modules import MSPA_C2f, C2f
from mgdt_yolo_amd.seeding import seed_state_dict_

DEV = 'cuda:0'


def bench(m, x, reps=50):
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10):
                m(x)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 10 * 1e3


cases = [('mspa', 32, 1, 160), ('mspa', 64, 2, 80), ('mspa', 128, 2, 40), ('mspa', 256, 1, 20), ('c2f', 256, 1, 80)]
for kind, c, n, hw in cases:
    m = (MSPA_C2f(c, c, n, True) if kind == 'mspa' else C2f(c, 64, n, False))
    m = seed_state_dict_(m, 1).eval().to(DEV)
    m._cdtype = torch.bfloat16
    for sub in m.modules():
        if hasattr(sub, 'out_dtype'):
            sub._cdtype = torch.bfloat16
    x = torch.randn(32, c, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t1 = bench(m, x)
    ops.FUSED_CSP_BLOCK = False
    t0 = bench(m, x)
    ops.FUSED_CSP_BLOCK = True
    print(f'{kind} c={c} n={n} {hw}x{hw}: fused {t1:.1f} us, chain {t0:.1f} us  (tile {os.environ.get("MGDT_CSP_TILE", "auto")})', flush=True)
