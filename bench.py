"""Headline benchmark: images/sec @640x640, batch 32 per GPU, MSPA-C2f + GD-neck YOLOv8n (BASELINE.json configs[1]).

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one resident batch: detection forward (all layers, HIP kernels through the
C ABI) + Detect decode + batched NMS (predict settings conf 0.25 / iou 0.7), replayed from a hipGraph.
Inputs (fp32 NCHW images) are already in HBM when the timed region starts.  Inference shards over the batch with
no collective: N ranks = N replicas with disjoint batches ("weak" scaling); the only collective is the MAX of
the per-rank times.  Prints ONE JSON line on rank 0, with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the CPU oracle - a port of the reference's path - timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak (spec)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--imgsz', type=int, default=640)
    ap.add_argument('--inflight', type=int, default=1, help='batches in flight: S hipGraphs replayed round-robin on S streams (each step is still one full batch)')
    ap.add_argument('--input', choices=['model', 'f32', 'u8'], default='model',
                    help="dtype of the resident image batch: 'model' = the compute dtype, what the reference's predictor hands its model "
                         "(img.half() / 255, engine/predictor.py:128-129); 'u8' = raw uint8, /255 fused into the stem kernel")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--model', default='mspa_c2f_gd_yolov8')
    ap.add_argument('--scale', default='n')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=16)
    return ap.parse_args()


def cpu_baseline(cfg, model, args):
    """The oracle (CPU port of the reference path: forward with folded BN + NMS) on a bounded sample."""
    from mgdt_yolo_amd.seeding import seeded_images
    from oracle import layers as OL
    from oracle import nms as ON
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # a 1-GPU box is granted a 16-core CPU share whatever the affinity mask says: more threads only thrash
    cores = int(os.environ.get('MGDT_CPU_THREADS', min(cores, 16)))
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    x = seeded_images(args.cpu_images, args.imgsz, args.imgsz, seed=0)
    strides = model.stride.tolist()

    def once():
        with torch.no_grad():
            y, _ = OL.model_forward(cfg, sd, x, strides, fused=True)
        ON.non_max_suppression(y.numpy(), conf_thres=0.25, iou_thres=0.7)
    once()                      # warm-up
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t0 < 10 and reps < 20):
        once()
        reps += 1
    dt = time.perf_counter() - t0
    return {'value': round(args.cpu_images * reps / dt, 2), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} x batch {args.cpu_images} @ {args.imgsz}x{args.imgsz}, torch-CPU fp32 oracle forward (BN folded) + numpy NMS'}


def main():
    args = parse()
    from mgdt_yolo_amd import parallel
    rank, local, world = parallel.env_rank()
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist = parallel.init('nccl', dev)

    from mgdt_yolo_amd import ops
    from mgdt_yolo_amd.models import get_config
    from mgdt_yolo_amd.nn.tasks import DetectionModel
    from mgdt_yolo_amd.seeding import seed_state_dict_, seeded_images
    from mgdt_yolo_amd.yolo.utils.ops import non_max_suppression  # noqa: F401  (the user-facing form; the graph uses ops.nms)

    cfg = get_config(args.model, args.scale, 80)
    tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = seed_state_dict_(DetectionModel(cfg, verbose=False), 0).eval().to(dev).set_compute_dtype(tdt)
    x = seeded_images(args.batch, args.imgsz, args.imgsz, seed=parallel.shard_seed(100, rank)).to(dev)     # resident in HBM, values in [0, 1]
    if args.input == 'u8':
        x = (x * 255).round().clamp_(0, 255).to(torch.uint8)
    elif args.input == 'model':
        x = x.to(tdt)

    def step():
        y, _ = model(x)
        return ops.nms(y, 0.25, 0.7, None, False, False, 300, 30000, 7680)

    with torch.no_grad():
        out = step()                       # packs the weights, allocates
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    out = step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
        run = graph.replay if graph is not None else step
        if graph is not None and args.inflight > 1:
            # S independent graph instances (own activation / NMS buffers, shared weights) on S streams: step i replays graph i % S, so the
            # single-workgroup-per-image NMS and the small-map layers of one batch overlap the wide layers of the next
            graphs, outs, streams = [graph], [out], [torch.cuda.Stream() for _ in range(args.inflight)]
            for _ in range(args.inflight - 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs.append(step())
                graphs.append(g)
            counter = [0]

            def run():
                i = counter[0] % args.inflight
                counter[0] += 1
                streams[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[i]):
                    graphs[i].replay()

        for _ in range(args.warmup):
            run()

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        elapsed = parallel.max_over_ranks(t1 - t0, dev)
        n_det = int(out[2].sum().item())
        if graph is not None and args.inflight > 1:      # every in-flight instance must have produced the same detections
            cnt = outs[0][2]
            valid = torch.arange(outs[0][1].shape[1], device=dev)[None, :] < cnt[:, None]        # rows past counts[b] are never written
            for o in outs[1:]:
                assert torch.equal(o[2], cnt) and torch.equal(o[1][valid], outs[0][1][valid]) and torch.equal(o[0][valid], outs[0][0][valid]), \
                    'in-flight graph instances disagree'

        # ---- roofline of the dominant kernel: eager launches bracketed by HIP events on the launch stream
        roof = None
        if rank == 0:
            with ops.profile() as prof:
                for _ in range(3):
                    step()
            agg = {}
            for name, meta, ms in prof.rows:
                key = (name, meta['shape'] if meta else None)
                a = agg.setdefault(key, [0.0, 0, meta])
                a[0] += ms
                a[1] += 1
            total_ms = sum(v[0] for v in agg.values()) / 3
            per_name = {}
            for (name, _), v in agg.items():
                per_name[name] = per_name.get(name, 0.0) + v[0] / 3
            # dominant kernel = the kernel (op) with the largest share of the step; `achieved` = its algorithmic bytes (or flops) per launch /
            # its average launch duration, both averaged over all its launches of one step
            fam = {}
            for (name, shape), (ms, cnt, meta) in agg.items():
                if meta is None:
                    continue
                f = fam.setdefault(name, {'ms': 0.0, 'n': 0, 'flops': 0.0, 'bytes': 0.0, 'shapes': []})
                f['ms'] += ms; f['n'] += cnt; f['flops'] += meta['flops'] * cnt; f['bytes'] += meta['bytes'] * cnt
                f['shapes'].append((ms / 3, cnt // 3, list(shape), meta))
            name, f = max(fam.items(), key=lambda kv: kv[1]['ms'])
            # eager launches start on an idle queue (Python issues slower than the GPU drains), which adds a ramp to every event pair; the graph
            # replay of the timed region has no such gaps.  Rescale the eager per-launch times so that they sum to the measured replay step.
            scale = (elapsed / args.steps * 1e3) / total_ms if graph is not None else 1.0
            avg_eager_s = f['ms'] / f['n'] * 1e-3
            avg_s = avg_eager_s * scale
            flops_l, bytes_l = f['flops'] / f['n'], f['bytes'] / f['n']
            ai = flops_l / bytes_l
            ridge = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
            if ai >= ridge:
                ach = flops_l / avg_s / 1e12
                roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                        'frac': round(ach / MFMA_PEAK_TFLOPS[args.dtype], 4)}
            else:
                ach = bytes_l / avg_s / 1e9
                roof = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4)}
            kern = {'conv2d_fwd': 'conv_igemm_kernel', 'conv2d_direct_fwd': 'conv_stem_kernel', 'cnx_mlp_fwd': 'cnx_mlp_kernel<STATS> + cnx_mlp_kernel<APPLY>',
                    'pw_chain3_fwd': 'pw_chain3_kernel', 'conv1x1_inject_fwd': 'conv1x1_inject_kernel'}.get(name, name)
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_pmc.sh), corrected
            if os.path.exists(tfile):
                ent = json.load(open(tfile)).get(f'{name}|{args.dtype}|b{args.batch}|{args.imgsz}')
                if ent:
                    traffic, tsrc = ent['hbm_bytes_per_launch'], ent.get('source')
            top = sorted(f['shapes'], key=lambda t: -t[0])[:6]
            roof.update({'traffic': traffic, 'traffic_source': tsrc, 'kernel': f'{kern} ({name})', 'launches_per_step': f['n'] // 3,
                         'avg_us': round(avg_s * 1e6, 2), 'avg_us_eager_events': round(avg_eager_s * 1e6, 2), 'eager_to_replay_scale': round(scale, 4), 'flops_per_launch': round(flops_l), 'bytes_per_launch': round(bytes_l), 'arith_intensity': round(ai, 1),
                         'share_of_eager_step': round(f['ms'] / 3 / total_ms, 3),
                         'largest_shapes_b_cin_h_w_cout_k_s': [{'shape': sh, 'launches': c, 'us_per_launch': round(ms_ / c * 1e3, 1),
                                                                'GBps': round(m['bytes'] / (ms_ / c * 1e-3) / 1e9), 'TFLOPs': round(m['flops'] / (ms_ / c * 1e-3) / 1e12, 1)}
                                                               for ms_, c, sh, m in top],
                         'eager_ms_per_step_by_op': {k: round(v, 3) for k, v in sorted(per_name.items(), key=lambda kv: -kv[1])},
                         'eager_ms_per_step': round(total_ms, 3)})

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = parallel.aggregate_throughput(args.batch, args.steps, elapsed, world)
        line = {'metric': 'images/sec @640x640 bs=32 per GPU, detection forward + NMS', 'value': round(value, 1), 'unit': 'images/sec',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                'config': {'workload': f'{args.model}-{args.scale} (MSPA-C2f + GD neck + Detect, nc=80) {args.imgsz}x{args.imgsz} inference, '
                                       f'batch {args.batch}/GPU: forward + decode + NMS(conf 0.25, iou 0.7), hipGraph replay' + (f', {args.inflight} batches in flight' if args.inflight > 1 else '') if graph is not None
                           else f'{args.model}-{args.scale} {args.imgsz}x{args.imgsz} eager',
                           'global_batch': world * args.batch, 'parallelism': f'replicas x{world} (batch-sharded, no collective)',
                           'input': f'{str(x.dtype).replace("torch.", "")} NCHW images resident in HBM', 'detections_last_step': n_det, 'weights': 'seeded random init (no checkpoints offline)'},
                'roofline': roof}
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(cfg, model, args)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
