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


def all_reduce(buf, op=None, group=None, async_op=False):
    """The one way this package issues a collective on device memory.

    HIP keeps, per event, the stream it was last recorded in, and hipEventQuery on an event whose stream is CAPTURING -- now,
    not when the event was recorded -- fails with hipErrorCapturedEvent and invalidates the capture.  The RCCL process group's
    watchdog thread polls the end event of every eager collective until a poll (every ~100 ms) finds it complete, and c10d runs a
    synchronous collective on the caller's current stream: a warm-up all-reduce on the stream a step is captured on a few
    milliseconds later made the watchdog throw in about one process in four (profiles/r05/sharded_abort_root_cause.txt; rounds 3-4
    saw it as SIGABRT of the sharded-step tests and of destroy_process_group).  Hence:
      * eager (the stream is not capturing): async_op=True -- the collective and its events live on c10d's internal stream,
        which this package never captures -- and the caller's stream joins it through work.wait();
      * capturing: a synchronous collective on the capturing stream; c10d does not hand captured work to the watchdog.
    async_op=True returns something with wait(); eager sync calls have waited already and return None."""
    if op is None:
        op = dist.ReduceOp.SUM
    cuda = buf.is_cuda and dist.get_backend(group) == "nccl"
    if not cuda:                              # gloo: host-driven, nothing to capture and no device events
        work = dist.all_reduce(buf, op=op, group=group, async_op=async_op)
        return work if async_op else None
    if torch.cuda.is_current_stream_capturing():
        dist.all_reduce(buf, op=op, group=group, async_op=False)
        return _Done() if async_op else None
    work = dist.all_reduce(buf, op=op, group=group, async_op=True)
    if async_op:
        return work
    work.wait()                               # stream-side join: the host does not block
    return None


class _Done:
    def wait(self):
        return True


def broadcast(buf, src=0, group=None):
    """dist.broadcast under the same rule as `all_reduce`: eager -> async_op=True + work.wait(), so that the collective
    and its end event live on c10d's internal stream and never on a stream this package captures later; refused while capturing
    (replica synchronisation is set-up work, not part of a step)."""
    cuda = buf.is_cuda and dist.get_backend(group) == "nccl"
    if not cuda:
        dist.broadcast(buf, src=src, group=group)
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("repsurf_amd.dist.broadcast: called while the current stream is capturing")
    dist.broadcast(buf, src=src, group=group, async_op=True).wait()


def rank_seed(base, rank):
    """Distinct synthetic data / CPU-generator streams per rank (config 3: rank r uses seed base*8+r)."""
    return base * 8 + rank


def max_over_ranks(seconds, device=None):
    """The slowest rank's wall time (the job's time)."""
    _, world, _ = env()
    if world == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def time_allreduce(numel, device, reps=10):
    """Microseconds of ONE all-reduce (AVG) of a flat fp32 buffer of `numel` elements -- the step's only data-path
    collective -- averaged over `reps` back-to-back calls after a barrier; max over ranks.  None when world_size is 1."""
    _, world, _ = env()
    if world == 1 or not dist.is_initialized():
        return None
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    for _ in range(2):
        all_reduce(buf)
    cuda = torch.device(device).type == "cuda"
    if cuda:
        torch.cuda.synchronize()
    barrier()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        all_reduce(buf)
    if cuda:
        torch.cuda.synchronize()
    return max_over_ranks((time.perf_counter() - t0) / reps, device) * 1e6


def barrier(group=None):
    """Every rank has arrived AND this rank's device is idle.  Under RCCL: a 1-element all-reduce through `all_reduce` above (never
    dist.barrier(): its collective would sit on the caller's stream) followed by a device synchronize."""
    if not dist.is_initialized():
        return
    if dist.get_backend(group) == "nccl" and torch.cuda.is_available():
        all_reduce(torch.zeros(1, device="cuda"), group=group)
        torch.cuda.synchronize()
    else:
        dist.barrier(group)


def finish(*steps):
    """Tear the process group down: graphs -> synchronize -> barrier -> destroy_process_group, in that order, for every caller
    (tests and bench.py alike).  `steps`: the graphed steps still alive -- their close() releases the hipGraphs that hold recorded
    collectives of this communicator before it goes."""
    for s in steps:
        if s is not None and hasattr(s, "close"):
            s.close()
    if dist.is_initialized():
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        barrier()
        dist.destroy_process_group()
