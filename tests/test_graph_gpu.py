"""hipGraph replay of the training step gives the same numbers as eager launches (same CPU-generator draws)."""
import numpy as np
import pytest
import torch

from tests import torch_executor

from tests.util import cloud, disable_dropout, name_seeded_init, ref_args

pytestmark = pytest.mark.gpu


def test_graphed_step_matches_eager(monkeypatch):
    """GraphedStep: the replayed hipGraph gives the eager loss and gradients of the same batch with the same draws.  The CPU draws
    are replaced by a counter-indexed function (three per pass: normal flip, two FPS starts), so the draw set a replay consumes
    is known: warm-up passes take sets 0..w-1, the capture pass set w (recorded, not executed), replay r set w+1+r."""
    from models.repsurf.repsurf_ssg_umb import Model
    from repsurf_amd import mlp, rng
    from repsurf_amd.graph import GraphedStep
    from util.utils import SmoothClsLoss
    torch_executor.set_backend("hip")
    calls = {"i": 0}

    def fake_draw(kind, b, n):
        i = calls["i"]
        calls["i"] += 1
        j = torch.arange(b)
        if kind == "flip":
            return (((j * 5 + i * 3) % 2).float() * 2. - 1.)
        return ((j * 131 + i * 977) % n).to(torch.int32)

    monkeypatch.setattr(rng, "_cpu_draw", fake_draw)
    pts = torch.from_numpy(cloud(3, 8, 1024)).cuda().permute(0, 2, 1).contiguous()
    lab = torch.arange(8).cuda() % 15
    crit = SmoothClsLoss()

    def fresh():
        m = Model(ref_args())
        name_seeded_init(m)
        disable_dropout(m)
        return m.cuda().train()

    graphed = fresh()
    step = GraphedStep(graphed, crit, None, pts, lab, warmup=2)
    assert calls["i"] == 9                    # 2 warm-up passes + the capture pass
    got = [step().item(), step().item()]      # draw sets 3 and 4
    g_g = torch.cat([p.grad.flatten() for p in graphed.parameters()]).clone()
    eager = fresh()
    want = []
    for s_ in (3, 4):
        calls["i"] = 3 * s_
        for p in eager.parameters():
            p.grad = None
        loss = crit(eager(pts), lab)
        loss.backward()
        want.append(loss.item())
    g_e = torch.cat([p.grad.flatten() for p in eager.parameters()])
    assert np.allclose(got, want, atol=2e-5), (got, want)
    assert torch.isfinite(g_g).all()
    assert (g_g - g_e).norm() / g_e.norm() < 1e-3, ((g_g - g_e).norm() / g_e.norm()).item()


def test_pipelined_step_matches_eager_step_for_step(monkeypatch):
    """PipelinedStep computes the geometry of batch s+1 under the network of batch s (two alternating graphs); with the
    CPU draws replaced by a counter-indexed function the loss of every replay must be the eager loss of the same batch
    with the same draws.  Tolerance 2e-5 on the loss (fp32 sums through atomics in the scatter kernels)."""
    from models.repsurf.repsurf_ssg_umb import Model
    from repsurf_amd import mlp, rng
    from repsurf_amd.graph import PipelinedStep
    from util.utils import SmoothClsLoss
    torch_executor.set_backend("hip")
    calls = {"i": 0}

    def fake_draw(kind, b, n):
        i = calls["i"]
        calls["i"] += 1
        j = torch.arange(b)
        if kind == "flip":
            return (((j * 5 + i * 3) % 2).float() * 2. - 1.)
        return ((j * 131 + i * 977) % n).to(torch.int32)

    monkeypatch.setattr(rng, "_cpu_draw", fake_draw)
    batches = [torch.from_numpy(cloud(s, 8, 1024)).cuda().permute(0, 2, 1).contiguous() for s in (3, 4)]
    labels = [torch.arange(8).cuda() % 15, (torch.arange(8).cuda() * 7) % 15]
    crit = SmoothClsLoss()

    def fresh():
        m = Model(ref_args())
        name_seeded_init(m)
        disable_dropout(m)
        return m.cuda().train()

    piped = fresh()
    step = PipelinedStep(piped, crit, None, batches[0], labels[0], warmup=1)
    first_set = calls["i"] // 3 - 1                 # draws behind the geometry the first replay consumes
    got = []
    for s in range(4):                               # replay s trains batch s % 2 and prepares batch (s + 1) % 2
        got.append(step(batches[(s + 1) % 2], labels[(s + 1) % 2]).item())
    g_p = torch.cat([p.grad.flatten() for p in piped.parameters()]).clone()

    eager = fresh()
    want = []
    for s in range(4):
        calls["i"] = 3 * (first_set + s)
        for p in eager.parameters():
            p.grad = None
        loss = crit(eager(batches[s % 2]), labels[s % 2])
        loss.backward()
        want.append(loss.item())
    g_e = torch.cat([p.grad.flatten() for p in eager.parameters()])
    assert np.allclose(got, want, atol=2e-5), (got, want)
    assert (g_p - g_e).norm() / g_e.norm() < 1e-3
