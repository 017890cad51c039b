"""Minimal repro search: small graphs of a few of our kernels replayed concurrently."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops
from mgdt_yolo_amd.nn.modules import Conv
from mgdt_yolo_amd.seeding import seed_state_dict_
dev = torch.device('cuda:0')
S, R = 5, 30
torch.manual_seed(0)
mk = lambda c, h, w: torch.randn(32, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
conv = seed_state_dict_(Conv(256, 128, 1, 1), 1).eval().to(dev); conv._cdtype = torch.bfloat16
conv3 = seed_state_dict_(Conv(128, 128, 3, 1), 2).eval().to(dev); conv3._cdtype = torch.bfloat16
xin = mk(256, 20, 20)

def case_conv_bilinear():
    t = conv(xin)
    out = ops.new_act(32, 128, 40, 40, torch.bfloat16, dev)
    return ops.bilinear(t, out)
def case_bilinear_only():
    out = ops.new_act(32, 256, 40, 40, torch.bfloat16, dev)
    return ops.bilinear(xin, out)
def case_conv_chain():
    t = conv(xin)
    for _ in range(6):
        t = conv3(t)
    return t
def case_tmp_reuse():
    # allocate/free temporaries so that later tensors reuse their memory inside the graph's pool
    t = conv(xin)
    for _ in range(4):
        u = conv3(t)
        t = conv3(u)
        del u
    out = ops.new_act(32, 128, 40, 40, torch.bfloat16, dev)
    return ops.bilinear(t, out)

def case_conv_aten():
    return conv(xin) * 2
def case_aten_bilinear():
    t = xin * 1
    return ops.bilinear(t, ops.new_act(32, 256, 40, 40, torch.bfloat16, dev))
def case_conv_copy():
    t = conv(xin)
    return ops.copy(t, ops.new_act(32, 128, 20, 20, torch.bfloat16, dev))
def case_conv3_bilinear():
    t = conv3(mk128)
    return ops.bilinear(t, ops.new_act(32, 128, 40, 40, torch.bfloat16, dev))
def case_conv_avgpool():
    t = conv(xin)
    return ops.adaptive_avgpool(t, ops.new_act(32, 128, 10, 10, torch.bfloat16, dev))
mk128 = mk(128, 20, 20)
with torch.no_grad():
    def case_conv_only():
        return conv(xin)
    for name, fn in [('conv -> bilinear', case_conv_bilinear), ('bilinear only', case_bilinear_only), ('conv only', case_conv_only), ('conv -> copy', case_conv_copy),
                     ('aten -> bilinear', case_aten_bilinear), ('conv -> aten', case_conv_aten), ('conv -> avgpool', case_conv_avgpool), ('conv chain', case_conv_chain)]:
        ref = fn().clone(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        graphs, outs = [], []
        for _ in range(S):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs.append(fn())
            graphs.append(g)
        streams = [torch.cuda.Stream() for _ in range(S)]
        bad = 0
        for rep in range(R):
            for k in range(3):
                for st, g in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        g.replay()
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    if bad <= 3:
                        d = (o.float() - ref.float()).abs()
                        idx = torch.nonzero(d)
                        i0 = tuple(idx[0].tolist())
                        print('   diff count', idx.shape[0], 'of', d.numel(), 'max', d.max().item(), 'first at', i0, 'got', o[i0].item(), 'ref', ref[i0].item(),
                              'n set', sorted(set(idx[:, 0].tolist()))[:6], 'c set', sorted(set(idx[:, 1].tolist()))[:10], 'y set', sorted(set(idx[:, 2].tolist()))[:10], 'x set', sorted(set(idx[:, 3].tolist()))[:10])
        print(f'{name:45s}: {bad} mismatches in {R * S}', flush=True)
