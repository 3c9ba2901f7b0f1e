"""Everything that brings up a process group on the GPU box, LAST in collection order (the file name sorts behind every other test
file): a communicator problem can then never hide a parity test from a `pytest -x` run.  The RCCL bodies run in a child process
each (`_isolated`): the child's full stderr is kept in a file, a non-zero exit FAILS the test and the first line that names a cause
leads the assertion message.  Mirrors what the reference does with mp.spawn + NCCL DDP (segmentation/tool/train.py:478-484,141-152)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import torch_executor
from tests.util import ROOT, cloud, disable_dropout, name_seeded_init, ref_args

pytestmark = pytest.mark.gpu

_CAUSE = re.compile(r"terminate called|what\(\)|HIP error|hipError|Memory access fault|Fatal Python error|Segmentation fault|Error:")


def _first_cause(text):
    for line in text.splitlines():
        if _CAUSE.search(line):
            return line.strip()[:500]
    return "(no line of the child's stderr names a cause)"


def _isolated(name, timeout=600):
    """Run a body of this file in a process of its own (one communicator per process, torn down by repsurf_amd.dist.finish()).
    The child must exit 0 AND print the marker."""
    logdir = os.environ.get("REPSURF_TEST_LOGDIR", os.path.join(ROOT, "gpurun_out", "dist_tests"))
    os.makedirs(logdir, exist_ok=True)
    code = ("import sys; sys.path[:0] = [%r, %r]; import tests.test_zz_dist_gpu as t; t.%s(); print('ISOLATED-BODY-OK', flush=True)"
            % (ROOT, os.path.join(ROOT, "repsurf_amd", "classification"), name))
    env = dict(os.environ, TORCH_SHOW_CPP_STACKTRACES="1", PYTHONFAULTHANDLER="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    err_file = os.path.join(logdir, name + ".stderr")
    with open(err_file, "w") as f:
        f.write(r.stderr)
    assert r.returncode == 0 and "ISOLATED-BODY-OK" in r.stdout, (
        f"{_first_cause(r.stderr)} | rc {r.returncode} | full stderr: {err_file} | stdout tail: {r.stdout[-500:]!r} | stderr tail: {r.stderr[-3000:]}")


def test_sharded_step_single_rank_process_group():
    _isolated("_sharded_step_single_rank_process_group")


def test_pipelined_sharded_step_single_rank_process_group():
    _isolated("_pipelined_sharded_step_single_rank_process_group")


def _sharded_step_single_rank_process_group():
    """ShardedGraphedStep (graph A -> all-reduce -> graph B) with a 1-rank RCCL process group."""
    import os
    import torch.distributed as dist
    from models.repsurf.repsurf_ssg_umb import Model
    from repsurf_amd import mlp
    from repsurf_amd.graph import ShardedGraphedStep
    from util.utils import SmoothClsLoss
    torch_executor.set_backend("hip")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        m = Model(ref_args())
        name_seeded_init(m)
        m = m.cuda().train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True, capturable=True)
        pts = torch.from_numpy(cloud(3, 8, 1024)).cuda().permute(0, 2, 1).contiguous()
        lab = torch.arange(8).cuda() % 15
        w0 = m.classfier[8].weight.detach().clone()
        step = ShardedGraphedStep(m, SmoothClsLoss(), opt, pts, lab, warmup=2)
        l1 = step().item()
        l2 = step().item()
        assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1 + 0.5
        assert not torch.equal(w0, m.classfier[8].weight)          # the optimizer graph ran
        assert step.flat.abs().sum() > 0
        print("ISOLATED-ASSERTS-OK", flush=True)
    finally:
        from repsurf_amd import dist as rdist
        held, step, m, opt = locals().get("step"), None, None, None
        rdist.finish(held)             # graphs -> synchronize -> barrier -> destroy_process_group


def _pipelined_sharded_step_single_rank_process_group():
    """PipelinedStep(sharded=True): graph[p] (geometry s+1 | forward/backward s into the flat gradient buffer) -> RCCL
    all-reduce -> Adam graph, with a 1-rank process group and this package's Adam; it must train like the unsharded
    pipelined step from the same initial state (same draws).  Both runs make 2 warm-up + 3 measured Adam updates with
    fp32 atomics in the scatter kernels, so the trajectories drift apart slowly: losses within 2e-2."""
    import os
    import torch.distributed as dist
    from models.repsurf.repsurf_ssg_umb import Model
    from repsurf_amd import mlp
    from repsurf_amd.graph import PipelinedStep
    from repsurf_amd.optim import Adam
    from util.utils import SmoothClsLoss
    torch_executor.set_backend("hip")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        pts = torch.from_numpy(cloud(3, 8, 1024)).cuda().permute(0, 2, 1).contiguous()
        lab = torch.arange(8).cuda() % 15
        losses = {}
        # round 3: the all-reduce (forced on this 1-rank group: REPSURF_FORCE_ALLREDUCE) and Adam are recorded INSIDE the network
        # graph -- "captured" -- against the round-2 form (network graph -> eager collective -> Adam graph) and the N = 1 step
        os.environ["REPSURF_FORCE_ALLREDUCE"] = "1"
        for kind in ("single", "captured", "between", "buckets"):
            os.environ["REPSURF_CAPTURE_ALLREDUCE"] = "0" if kind == "between" else "1"
            os.environ["REPSURF_GRAD_BUCKETS"] = "2" if kind == "buckets" else "1"    # bucket 0 (sa3 + head) all-reduced from a backward hook
            m = Model(ref_args())
            name_seeded_init(m)
            disable_dropout(m)
            m = m.cuda().train()
            opt = Adam(m.parameters(), lr=1e-3)
            torch.manual_seed(21)
            step = PipelinedStep(m, SmoothClsLoss(), opt, pts, lab, warmup=2, sharded=kind != "single")
            losses[kind] = [step().item() for _ in range(int(os.environ.get('REPSURF_SOAK_STEPS', '3')))]
            if kind != "single":
                assert step.flat.abs().sum() > 0
                assert step.collective_captured == (kind != "between"), kind
                assert len(step.grads.buckets) == (2 if kind == "buckets" else 1)
            step.close()               # one step's recorded collectives at a time on the communicator
        first = slice(0, 3)            # (a soak runs more steps: the trajectories are compared over the first three)
        assert np.allclose(losses["captured"][first], losses["single"][first], atol=2e-2), losses
        assert np.allclose(losses["buckets"][first], losses["single"][first], atol=2e-2), losses
        assert np.allclose(losses["captured"][first], losses["between"][first], atol=2e-2), losses
        assert losses["captured"][2] < losses["captured"][0] + 0.5
        assert all(np.isfinite(v).all() for v in losses.values())
        print("ISOLATED-ASSERTS-OK", flush=True)
    finally:
        os.environ.pop("REPSURF_FORCE_ALLREDUCE", None)
        os.environ.pop("REPSURF_CAPTURE_ALLREDUCE", None)
        os.environ.pop("REPSURF_GRAD_BUCKETS", None)
        from repsurf_amd import dist as rdist
        held, step, m, opt = locals().get("step"), None, None, None
        rdist.finish(held)             # close() the last step (its graphs hold recorded collectives), synchronize, barrier, destroy


@pytest.mark.parametrize("extra", [[], ["--no-pipeline", "--no-kernel-timing"], ["--no-graph", "--no-kernel-timing"], ["PLAIN", "--no-kernel-timing"]])
def test_bench_world_size_two_on_one_gpu(extra, tmp_path):
    """bench.py's N > 1 code (rank seeds, flat-gradient all-reduce between the captured graphs, Adam graph, barrier +
    max-over-ranks timing, rank-0 JSON line) launched the way the driver launches it, with two ranks sharing the one GPU of
    this box and gloo carrying the collective (RCCL refuses two ranks on one device).  Checks the contract fields and
    that both ranks leave the loop with the same parameters (they started equal and applied the same averaged gradient)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from tests.util import ROOT
    with socket.socket() as sk:      # a free rendezvous port (earlier tests of this process hold process groups of their own)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, REPSURF_DIST_BACKEND="gloo", REPSURF_BENCH_DEVICE="0", REPSURF_BENCH_DUMP=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline"] + extra        # extra = []: exactly the driver's flags (per-launch timing pass on rank 0)
    if extra and extra[0] == "PLAIN":          # `python bench.py --gpus 2` with no launcher around it: bench.py spawns its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + extra[1:]
        env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert "falling back" not in res.stderr, res.stderr[-3000:]
    a, b = (torch.load(os.path.join(str(tmp_path), f"params_rank{r}.pt")) for r in (0, 1))
    assert torch.isfinite(a).all() and torch.equal(a, b)
