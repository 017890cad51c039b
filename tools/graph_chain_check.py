"""Pure-ATen control experiment: S hipGraph instances of a dependent chain of element-wise kernels, replayed concurrently."""
import sys, torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
def chain(a):
    t = a
    for i in range(60):
        t = t * 1.0001 + 0.5 if i % 2 else torch.roll(t, 1, 0) - 0.25
    return t
srcs = [torch.randn(1 << 22, device=dev) for _ in range(S)]
with torch.no_grad():
    refs = [chain(a).clone() for a in srcs]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(srcs[0])
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graphs, outs = [], []
    for a in srcs:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs.append(chain(a))
        graphs.append(g)
    streams = [torch.cuda.Stream() for _ in range(S)]
    bad = 0
    for rep in range(30):
        for k in range(3):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, r) for o, r in zip(outs, refs))
    print(f'ATen chain graphs on {S} streams: {bad} mismatches in {30 * S}')
