"""Sweep the block kernel's tile (MGDT_CSP_TILE=th,tw, read at every launch) for each MSPA_C2f / C2f block of the bench model at B = 32 and print, per block,
the automatic choice's time and the forced tiles sorted by time:  python tools/csp_tile_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgdt_yolo_amd.nn.modules import MSPA_C2f, C2f
from mgdt_yolo_amd.seeding import seed_state_dict_

DEV = 'cuda:0'


def bench(m, x, reps=20):
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                m(x)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 10 * 1e3


cases = [('mspa', 32, 1, 160), ('mspa', 64, 2, 80), ('mspa', 128, 2, 40), ('mspa', 256, 1, 20), ('c2f', 256, 1, 80)]
for kind, c, n, hw in cases:
    m = (MSPA_C2f(c, c, n, True) if kind == 'mspa' else C2f(c, 64, n, False))
    m = seed_state_dict_(m, 1).eval().to(DEV)
    for sub in m.modules():
        if hasattr(sub, 'out_dtype'):
            sub._cdtype = torch.bfloat16
    x = torch.randn(32, c, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    os.environ.pop('MGDT_CSP_TILE', None)
    auto = bench(m, x)
    q = hw // 2 if kind == 'mspa' else hw
    divs = [d for d in range(2, 33) if q % d == 0]
    res = []
    for th in divs:
        for tw in divs:
            if th * tw < 32 or th * tw > 640:
                continue
            os.environ['MGDT_CSP_TILE'] = f'{th},{tw}'
            try:
                res.append((bench(m, x), th, tw))
            except Exception as e:      # the forced tile does not fit (LDS) or is not instantiated
                pass
    os.environ.pop('MGDT_CSP_TILE', None)
    res.sort()
    print(f'{kind} c={c} n={n} {hw}x{hw}: auto {auto:.1f} us; best forced: ' + ', '.join(f'{th}x{tw} {t:.1f}' for t, th, tw in res[:6]), flush=True)
