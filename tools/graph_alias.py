"""Concurrent hipGraph replay: do the captured instances' buffers alias?  Prints the address ranges of the intermediate `t` and of `out`
per instance, overlaps between instances, and the mismatch count for (a) t freed during capture, (b) t kept alive."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.nn.modules import Conv
from mgdt_yolo_amd.seeding import seed_state_dict_
dev = torch.device('cuda:0')
S, R = 5, 20
torch.manual_seed(0)
xin = torch.randn(32, 256, 20, 20, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
conv = seed_state_dict_(Conv(256, 128, 1, 1), 1).eval().to(dev); conv._cdtype = torch.bfloat16
ranges = []

def fn(keep):
    t = conv(xin)
    out = ops.bilinear(t, ops.new_act(32, 128, 40, 40, torch.bfloat16, dev))
    ranges.append((t.data_ptr(), t.numel() * 2, out.data_ptr(), out.numel() * 2))
    return (out, t) if keep else (out, None)

with torch.no_grad():
    for keep in (False, True):
        ranges.clear()
        ref = fn(keep)[0].clone(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(keep)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        ranges.clear()
        graphs, outs = [], []
        for _ in range(S):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs.append(fn(keep))
            graphs.append(g)
        segs = []
        for i, (tp, tn, op, on) in enumerate(ranges):
            segs += [(tp, tp + tn, f'g{i}.t'), (op, op + on, f'g{i}.out')]
        ov = [(a[2], b[2]) for i, a in enumerate(segs) for b in segs[i + 1:] if a[0] < b[1] and b[0] < a[1]]
        print('keep t alive:', keep, '| ranges:', [(hex(r[0]), hex(r[2])) for r in ranges], '| overlapping pairs:', ov)
        streams = [torch.cuda.Stream() for _ in range(S)]
        bad = 0
        for rep in range(R):
            for k in range(3):
                for st, g in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        g.replay()
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o[0], ref)) for o in outs)
        print(f'   concurrent replays: {bad} mismatching outputs of {R * S}', flush=True)
