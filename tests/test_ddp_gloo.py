"""The N>1 plumbing of bench.py / repsurf_amd.dist on CPU: world_size 2, gloo backend, 127.0.0.1.
(The model itself has no CPU path, so a small stand-in module carries the gradients; what is
under test is rank discovery, per-rank data, the single-bucket gradient averaging and the
max-over-ranks timing — the same calls bench.py makes with backend "nccl" on the GPUs.)"""
import os
import socket

import torch
import torch.multiprocessing as mp

from tests.util import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from repsurf_amd import dist as rdist
    r, w = rdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    net = rdist.wrap(model)
    g = torch.Generator().manual_seed(rdist.rank_seed(5, rank))   # rank-specific batch
    x = torch.randn(32, 8, generator=g)
    net(x).pow(2).mean().backward()
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    # local (un-averaged) gradient of this rank's batch, for the mean check
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    ref.load_state_dict(model.state_dict())
    ref(x).pow(2).mean().backward()
    local = torch.cat([p.grad.flatten() for p in ref.parameters()])
    slow = rdist.max_over_ranks(1.0 + rank)
    rdist.barrier()
    out[rank] = (grads, local, slow, float(x.sum()))
    rdist.finish()


def test_two_rank_gradient_average_and_timing():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        (g0, l0, s0, x0), (g1, l1, s1, x1) = out[0], out[1]
    assert torch.allclose(g0, g1)                           # all-reduced: same on both ranks
    assert torch.allclose(g0, (l0 + l1) / 2, atol=1e-6)     # = mean of the per-rank gradients
    assert s0 == s1 == 2.0                                  # job time = slowest rank
    assert x0 != x1                                         # ranks saw different data


def test_single_process_defaults():
    from repsurf_amd import dist as rdist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert rdist.env() == (0, 1, 0)
    m = torch.nn.Linear(2, 2)
    assert rdist.wrap(m) is m and rdist.max_over_ranks(3.5) == 3.5


def _flat_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from repsurf_amd import dist as rdist
    from repsurf_amd.graph import attach_flat_grads
    rdist.init(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    flat = attach_flat_grads(list(model.parameters()))
    g = torch.Generator().manual_seed(rdist.rank_seed(5, rank))
    x = torch.randn(32, 8, generator=g)
    for _ in range(2):                         # second pass checks the zero_() + in-place accumulation cycle
        flat.zero_()
        model(x).pow(2).mean().backward()
    local = flat.clone()
    assert all(p.grad.data_ptr() >= flat.data_ptr() for p in model.parameters())   # still views of the buffer
    dist.all_reduce(flat)
    flat.div_(world)
    out[rank] = (flat.clone(), local, torch.cat([p.grad.flatten() for p in model.parameters()]))
    rdist.finish()


def test_flat_gradient_buffer_allreduce():
    """the gradient path of ShardedGraphedStep (flat buffer + one all-reduce) on 2 gloo ranks"""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_flat_worker, args=(world, port, out), nprocs=world, join=True)
        (f0, l0, g0), (f1, l1, g1) = out[0], out[1]
    assert torch.allclose(f0, f1) and torch.allclose(f0, (l0 + l1) / 2, atol=1e-6)
    assert torch.equal(g0, f0)                 # parameters see the averaged gradient through their views


def _flatgrads_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from repsurf_amd import dist as rdist
    from repsurf_amd.graph import FlatGrads
    rdist.init(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    grads = FlatGrads(list(model.parameters()))
    g = torch.Generator().manual_seed(rdist.rank_seed(5, rank))
    x = torch.randn(32, 8, generator=g)
    for _ in range(2):                         # second pass: clear() -> fresh gradient tensors -> pack() again
        grads.clear()
        model(x).pow(2).mean().backward()
        assert all(p.grad.data_ptr() != v.data_ptr() for p, v in zip(grads.params, grads.views))   # handed over, not accumulated
        local = torch.cat([p.grad.flatten() for p in model.parameters()])
        grads.pack()
    assert torch.equal(grads.flat, local)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(grads.params, grads.views))
    grads.all_reduce_mean(dist)
    out[rank] = (grads.flat.clone(), local, torch.cat([p.grad.flatten() for p in model.parameters()]))
    rdist.finish()


def test_flatgrads_pack_and_allreduce():
    """the gradient path of the sharded steps (repsurf_amd.graph.FlatGrads: backward with p.grad = None, one
    multi-tensor pack, one all-reduce, optimizer reads the slices) on 2 gloo ranks"""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_flatgrads_worker, args=(world, port, out), nprocs=world, join=True)
        (f0, l0, g0), (f1, l1, g1) = out[0], out[1]
    assert torch.allclose(f0, f1) and torch.allclose(f0, (l0 + l1) / 2, atol=1e-6)
    assert torch.equal(g0, f0)


def _bucket_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from repsurf_amd import dist as rdist
    from repsurf_amd.graph import FlatGrads
    rdist.init(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    early = list(model[2].parameters())        # the layer closest to the loss: its gradients exist first
    grads = FlatGrads(list(model.parameters()), early=early)
    assert len(grads.buckets) == 2 and grads.buckets[0].numel() == 16 * 3 + 3 and grads.params[:2] == early
    g = torch.Generator().manual_seed(rdist.rank_seed(5, rank))
    x = torch.randn(32, 8, generator=g)
    import copy
    twin = copy.deepcopy(model)                # the rank's own gradient, computed the plain way, in the flat buffer's order
    twin(x).pow(2).mean().backward()
    plain = torch.cat([p.grad.flatten() for p in list(twin[2].parameters()) + list(twin[0].parameters())])
    h = model[1](model[0](x))
    seen = {}

    def early_bucket(grad):                    # what PipelinedStep._early_bucket does, from a tensor hook mid-backward
        assert model[0].weight.grad is None    # the rest of the backward has not run yet
        grads.pack(0)
        seen["work"] = grads.all_reduce_mean(dist, bucket=0, async_op=True)
    h.register_hook(early_bucket)
    grads.clear()
    model[2](h).pow(2).mean().backward()
    local = torch.cat([p.grad.flatten() for p in grads.params[:2]] + [p.grad.flatten() for p in grads.params[2:]])
    local_early = local[:grads.buckets[0].numel()].clone()    # (bucket 0 is already averaged in place when the wait returns)
    grads.pack(1)
    grads.all_reduce_mean(dist, bucket=1)
    if seen.get("work") is not None:
        seen["work"].wait()
    out[rank] = (grads.flat.clone(), torch.cat([p.grad.flatten() for p in grads.params]),
                 torch.cat([p.grad.flatten() for p in model.parameters()]), plain)
    rdist.finish()


def test_flatgrads_two_buckets_in_reverse_execution_order():
    """FlatGrads(early=...): bucket 0 packed and all-reduced from a tensor hook in the middle of backward (asynchronously),
    bucket 1 after it, on 2 gloo ranks: both ranks end with the same averaged gradient in every parameter's view."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
        (f0, v0, m0, p0), (f1, v1, m1, p1) = out[0], out[1]
    assert torch.allclose(f0, f1) and torch.equal(f0, v0) and torch.allclose(m0, m1)
    assert torch.allclose(f0, (p0 + p1) / 2, atol=1e-6) and f0.abs().sum() > 0


def _bucket_steps_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import copy
    import torch.distributed as dist
    from repsurf_amd import dist as rdist
    from repsurf_amd.graph import FlatGrads, sync_replicas
    rdist.init(backend="gloo")
    torch.manual_seed(100 + rank)              # replicas start DIFFERENT: sync_replicas (repsurf_amd.dist.broadcast) must make them rank 0's
    base = torch.nn.Sequential(torch.nn.Linear(6, 12), torch.nn.ReLU(), torch.nn.Linear(12, 12), torch.nn.ReLU(), torch.nn.Linear(12, 4))
    sync_replicas(base, dist)
    g = torch.Generator().manual_seed(rdist.rank_seed(9, rank))
    batches = [torch.randn(16, 6, generator=g) for _ in range(5)]
    finals = {}
    for form in ("one_bucket", "two_buckets"):
        model = copy.deepcopy(base)
        early = list(model[4].parameters()) + list(model[2].parameters()) if form == "two_buckets" else None
        grads = FlatGrads(list(model.parameters()), early=early)
        assert len(grads.buckets) == (2 if early else 1)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, foreach=False)
        for x in batches:
            grads.clear()
            h = model[1](model[0](x))
            seen = {}
            if early:
                def early_bucket(grad, seen=seen):      # PipelinedStep._early_bucket (graph.py): bucket 0 leaves from a hook mid-backward
                    grads.pack(0)
                    seen["work"] = grads.all_reduce_mean(dist, bucket=0, async_op=True)
                h.register_hook(early_bucket)
            model[4](model[3](model[2](h))).pow(2).mean().backward()
            if early:
                grads.pack(1)
                grads.all_reduce_mean(dist, bucket=1)
                seen["work"].wait()
            else:
                grads.pack()
                grads.all_reduce_mean(dist)
            opt.step()
        finals[form] = torch.cat([p.detach().flatten() for p in model.parameters()])
    out[rank] = (finals["one_bucket"], finals["two_buckets"])
    rdist.finish()


def test_early_bucket_path_ends_with_the_single_bucket_parameters():
    """five optimizer steps on 2 gloo ranks -- gradients reduced as ONE flat bucket after backward, and as two
    buckets with bucket 0 (the layers closest to the loss) packed and all-reduced asynchronously from a tensor hook in the middle
    of backward (what PipelinedStep does under REPSURF_GRAD_BUCKETS=2, repsurf_amd/graph.py `_early_bucket` / `_reduce`): the same
    parameters on both ranks and in both forms, to the last bit (the reductions add the same two numbers either way).  The
    replicas are seeded differently and meet through sync_replicas (repsurf_amd.dist.broadcast)."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bucket_steps_worker, args=(world, port, out), nprocs=world, join=True)
        (a0, b0), (a1, b1) = out[0], out[1]
    assert torch.equal(a0, a1) and torch.equal(b0, b1)          # replicas stay replicas
    assert torch.equal(a0, b0)                                  # the early-bucket form is the single-bucket form
    assert a0.abs().sum() > 0


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` PLAIN (no torch.distributed.run around it: the form a driver may use) re-executes itself under
    the launcher on 127.0.0.1 with a free port, the ranks join and reduce, rank 0 prints exactly one JSON line.  --dry-run stops
    before the model (no HIP device here); tests/test_graph_gpu.py runs the full plain form on the GPU box."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["REPSURF_DIST_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 2, "requested": 2, "rank_sum": 3.0, "backend": "gloo"}


def _sync_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repsurf_amd import mlp_hip
    plain, synced = torch.nn.BatchNorm1d(4), torch.nn.SyncBatchNorm(4)
    assert mlp_hip.sync_of(plain) is None and mlp_hip.sync_of(synced.eval()) is None
    sync = mlp_hip.sync_of(synced.train())
    assert sync is not None and sync[2] == world
    # what a producing kernel leaves on this rank: fp64 partial sums {sum y, sum y^2} of ITS rows, two workgroup rows
    g = torch.Generator().manual_seed(7)
    y = torch.randn(world * 10, 4, generator=g, dtype=torch.float64)
    mine = y[rank * 10:(rank + 1) * 10]
    part = torch.stack([torch.stack([mine[:5].sum(0), (mine[:5] ** 2).sum(0)]), torch.stack([mine[5:].sum(0), (mine[5:] ** 2).sum(0)])])
    factor = mlp_hip.sync_partials(part, sync, 10)
    rows = 10 * factor
    mean = part[:, 0].sum(0) / rows
    var = part[:, 1].sum(0) / rows - mean ** 2
    assert mlp_hip.sync_row_mismatch_count() == 0
    # ragged shards: rank r holds 3 + r partial rows of 4 + r data rows each -- the collective's shape must not depend
    # on the partial-row count (no hang), the sums are still the global sums, and the mismatch is counted
    ragged = torch.ones((3 + rank, 2, 4), dtype=torch.float64)
    mlp_hip.sync_partials(ragged, sync, (3 + rank) * (4 + rank))
    assert torch.equal(ragged[0], torch.full((2, 4), float(sum(3 + r for r in range(world))), dtype=torch.float64))
    assert ragged[1:].abs().sum() == 0
    assert mlp_hip.sync_row_mismatch_count() == 1
    out[rank] = (factor, mean, var, y.mean(0), y.var(0, unbiased=False))
    dist.destroy_process_group()


def test_sync_batchnorm_partials_span_the_ranks():
    """repsurf_amd.mlp_hip.sync_of / sync_partials (the --sync_bn option, segmentation/tool/train.py:141-142): an nn.SyncBatchNorm
    module in training mode asks for statistics over the process group; one all-reduce of the fp64 partial sums in front of the
    finalize and the global row count give the whole batch's mean and variance on every rank."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_sync_worker, args=(world, port, out), nprocs=world, join=True)
        for r in range(world):
            factor, mean, var, ref_mean, ref_var = out[r]
            assert factor == world
            assert torch.allclose(mean, ref_mean, atol=1e-12) and torch.allclose(var, ref_var, atol=1e-12)


def _helper_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from repsurf_amd import dist as rdist
    rdist.init(backend="gloo")
    a = torch.full((4,), float(rank + 1))
    assert rdist.all_reduce(a) is None                       # synchronous form: reduced on return
    b = torch.full((4,), float(rank + 1))
    work = rdist.all_reduce(b, op=dist.ReduceOp.MAX, async_op=True)      # asynchronous form: something with wait()
    work.wait()
    closed = []

    class Step:                                              # what finish() is handed: the graphed steps still alive
        def close(self):
            closed.append(dist.is_initialized())             # close() runs BEFORE the process group goes

    rdist.barrier()
    rdist.finish(Step(), None)
    out[rank] = (a.tolist(), b.tolist(), closed, dist.is_initialized())


def test_collective_helper_and_teardown_order():
    """repsurf_amd.dist.all_reduce (the one way the package issues a collective: sync / async forms), barrier and finish(*steps): the
    steps are closed first -- their graphs hold recorded collectives of the communicator -- then the group is destroyed."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_helper_worker, args=(world, port, out), nprocs=world, join=True)
        for r in range(world):
            a, b, closed, alive = out[r]
            assert a == [3.0] * 4 and b == [2.0] * 4
            assert closed == [True] and alive is False
