"""Which neighbour makes a packed-fp32 kernel go wrong under concurrent graph replay?  Run with MGDT_LIB=.../libmgdt_hip_pkvictim.so (the library with
ONLY pointwise.o built with packed-fp32 instructions).  Each case = S graph instances replayed concurrently on S streams; the checked output is always a
bilinear up-sampling (the packed-fp32 victim), the other launches of the instance vary.      python tools/graph_bisect3.py [S rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgdt_yolo_amd import ops  # noqa: E402
from mgdt_yolo_amd.nn.modules import Conv  # noqa: E402
from mgdt_yolo_amd.seeding import seed_state_dict_  # noqa: E402

dev = torch.device('cuda:0')
S, R = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 40)
torch.manual_seed(0)
mk = lambda c, h, w: torch.randn(32, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
conv = seed_state_dict_(Conv(256, 128, 1, 1), 1).eval().to(dev); conv._cdtype = torch.bfloat16
conv3 = seed_state_dict_(Conv(128, 128, 3, 1), 2).eval().to(dev); conv3._cdtype = torch.bfloat16
conv_small = seed_state_dict_(Conv(32, 32, 1, 1), 3).eval().to(dev); conv_small._cdtype = torch.bfloat16
xin, x128, x32 = mk(256, 20, 20), mk(128, 20, 20), mk(32, 80, 80)
up = lambda t: ops.bilinear(t, ops.new_act(32, t.shape[1], 40, 40, torch.bfloat16, dev))


def conv_then_up_of_it():
    return up(conv(xin))


def conv_then_up_of_other():          # the bilinear input does not come from the convolution in front of it
    keep = conv(xin)
    return up(x128), keep


def up_then_conv():                   # the convolution FOLLOWS the bilinear in its own instance (it still overlaps other instances' bilinears)
    o = up(x128)
    return o, conv(xin)


def conv3_then_up():
    return up(conv3(x128))


def small_conv_then_up():
    keep = conv_small(x32)
    return up(x128), keep


def pool_then_up():                   # a non-MFMA neighbour
    keep = ops.adaptive_avgpool(xin, ops.new_act(32, 256, 10, 10, torch.bfloat16, dev))
    return up(x128), keep


def up_only():
    return up(x128)


# the synthetic MFMA neighbour (tools/repro/synth_aggressor.hip) in the convolution's place: mode bits = LDS panel | buffer loads | bpermute+SiLU epilogue | barrier
import ctypes as C  # noqa: E402
_sy = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'repro', 'libsynth.so'))
_sy.synth_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p]
_sbuf = torch.empty(1024 * 256 * 8, dtype=torch.uint8, device=dev)


def synth(mode, wgs, iters):
    def fn():
        rc = _sy.synth_launch(mode, _sbuf.data_ptr(), wgs, iters, xin.data_ptr(), xin.numel() * 2, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return up(x128)
    return fn


first = lambda o: o[0] if isinstance(o, tuple) else o
with torch.no_grad():
    for name, fn in [('conv1x1 256->128 -> bilinear(its output)', conv_then_up_of_it), ('conv1x1 -> bilinear(other tensor)', conv_then_up_of_other),
                     ('bilinear(other) -> conv1x1', up_then_conv), ('conv3x3 128->128 -> bilinear(its output)', conv3_then_up),
                     ('conv1x1 32->32 at 80x80 -> bilinear(other)', small_conv_then_up), ('avgpool -> bilinear(other)', pool_then_up), ('bilinear only', up_only)] + (
            [(f'synthetic MFMA mode {m:2d} ({w} wgs x {it} it) -> bilinear', synth(m, w, it)) for m, w, it in ((0, 256, 300), (15, 256, 300), (15, 512, 60), (7, 200, 100), (0, 1024, 40))] if os.environ.get('SYNTH') else []):
        ref = first(fn()).clone(); torch.cuda.synchronize()
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
        bad = nel = 0
        for rep in range(R):
            for k in range(3):
                for st, g in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        g.replay()
            torch.cuda.synchronize()
            for o in outs:
                d = first(o) != ref
                if d.any():
                    bad += 1; nel += int(d.sum())
        print(f'{name:<46}: {bad:3d} wrong outputs of {R * S} ({nel} elements)', flush=True)
