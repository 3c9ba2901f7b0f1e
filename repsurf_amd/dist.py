"""One-process-per-GPU data parallelism for the RepSurf-U step (RCCL over xGMI on MI355X).

The hot path shards by cloud with no data-path collective: every rank runs the whole model on its
own B clouds (per-GPU BatchNorm statistics, the reference's default — segmentation/tool/train.py:141-146)
and the only exchange is ONE gradient all-reduce per step.  The 1.48 M-parameter classifier is
5.9 MB of fp32 gradients: ONE RCCL ring all-reduce of one flat buffer, latency-bound (< 1 MB per xGMI link and hop).
Graphed steps (repsurf_amd.graph): the collective is recorded INSIDE the step's hipGraph between the gradient pack and the
Adam kernels -- one replay per rank-step, no host hop; it runs after the whole backward (it is not overlapped with it:
DESIGN.md 8 prices that).  Eager steps (`wrap`): DistributedDataParallel with a single 64 MB bucket, whose all-reduce starts
when the last gradient of the bucket is ready, i.e. at the end of backward too.
"""
import os

import torch
import torch.distributed as dist


def env():
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """Join the process group when WORLD_SIZE > 1.  backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU."""
    rank, world, _ = env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kwargs = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def wrap(model, device=None):
    """DistributedDataParallel with one flat bucket; identity when there is a single rank."""
    _, world, local = env()
    if world == 1:
        return model
    ids = [device.index if device.index is not None else local] if (device is not None and device.type == "cuda") else None
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, bucket_cap_mb=64, gradient_as_bucket_view=True, broadcast_buffers=False,
        find_unused_parameters=False)


def rank_seed(base, rank):
    """Distinct synthetic data / CPU-generator streams per rank (config 3: rank r uses seed base*8+r)."""
    return base * 8 + rank


def max_over_ranks(seconds, device=None):
    """The slowest rank's wall time (the job's time)."""
    _, world, _ = env()
    if world == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def time_allreduce(numel, device, reps=10):
    """Microseconds of ONE all-reduce (AVG) of a flat fp32 buffer of `numel` elements -- the step's only data-path
    collective -- averaged over `reps` back-to-back calls after a barrier; max over ranks.  None when world_size is 1."""
    _, world, _ = env()
    if world == 1 or not dist.is_initialized():
        return None
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    for _ in range(2):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    cuda = torch.device(device).type == "cuda"
    if cuda:
        torch.cuda.synchronize()
    dist.barrier()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if cuda:
        torch.cuda.synchronize()
    return max_over_ranks((time.perf_counter() - t0) / reps, device) * 1e6


def barrier():
    if dist.is_initialized():
        dist.barrier()


def finish():
    """Tear the process group down.  The caller has dropped its step objects first: a hipGraph with recorded collectives that outlives
    its communicator made destroy_process_group abort (seen once in three full test runs, with a 1-rank group); everything in
    flight is drained and every rank has arrived before any rank starts."""
    if dist.is_initialized():
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
