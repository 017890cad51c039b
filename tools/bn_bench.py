"""BatchNorm training kernels alone: python tools/bn_bench.py  -> us and GB/s per shape for stats / fwd / bwd (bf16, HIP events, 20 reps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgdt_yolo_amd import ops
shapes = [(32, 16, 320, 320), (32, 32, 160, 160), (32, 16, 160, 160), (32, 64, 80, 80), (32, 80, 80, 80), (32, 128, 40, 40), (32, 256, 20, 20)]
dt = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != 'f32' else torch.float32
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (n, c, h, w) in shapes:
    y = torch.randn(n, c, h, w, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    gz = torch.randn_like(y)
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    mean, rstd = ops.bn_stats(y, 1e-3, 0.03)
    z = torch.empty_like(y); dy = torch.empty_like(y)
    dg, db = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    nb = y.numel() * y.element_size()
    t0 = timeit(lambda: ops.bn_stats(y, 1e-3, 0.03))
    t1 = timeit(lambda: ops.bn_act(y, mean, rstd, gamma, beta, ops.ACT_SILU, out=z))
    t2 = timeit(lambda: ops.bn_act_bwd(gz, y, mean, rstd, gamma, beta, ops.ACT_SILU, dg, db))
    print(f'{(n, c, h, w)}: {nb / 1e6:6.1f} MB  stats {t0:6.1f} us ({nb / t0 / 1e3:5.0f} GB/s)  fwd {t1:6.1f} us ({2 * nb / t1 / 1e3:5.0f} GB/s)  bwd {t2:6.1f} us ({5 * nb / t2 / 1e3:5.0f} GB/s)')
