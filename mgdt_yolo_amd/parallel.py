"""One process per GPU; inference is batch-sharded replicas (no data-path collective), so the only exchange is
the reduction of the per-rank wall time.  Backend 'nccl' (== RCCL over xGMI on ROCm) on GPUs, 'gloo' in CPU tests."""
import os

import torch


def env_rank():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    kw = {'device_id': device} if (device is not None and backend == 'nccl') else {}
    dist.init_process_group(backend, **kw)
    return dist


def world():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def broadcast_(flat, src=0):
    """Initial parameter / buffer broadcast of data-parallel training (what DDP's constructor does, trainer.py:225): one message."""
    import torch.distributed as dist
    if world() > 1:
        dist.broadcast(flat, src=src)
    return flat


def shard_seed(base, rank):
    """Each replica gets a disjoint synthetic batch."""
    return base + rank


def max_over_ranks(value, device='cpu'):
    """MAX all-reduce of a python float (the job's step time is its slowest rank's)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def aggregate_throughput(units_per_rank, steps, elapsed_max, world):
    """Whole-job units/s: every rank processed `units_per_rank` per step, the job took the slowest rank's time."""
    return world * units_per_rank * steps / elapsed_max


def all_reduce_mean_(flat):
    """Gradient exchange of data-parallel training: ONE all-reduce (mean) over the flat gradient buffer.
    No-op for a single process.  RCCL ('nccl') on GPUs; the CPU tests drive it with 'gloo'."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    return flat
