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


def test_train_loop_step_is_the_eager_loop_batch_for_batch():
    """repsurf_amd.graph.TrainLoopStep: the five statements of the reference's loop body
    (classification/tool/train_cls_scanobjectnn.py:226-238) as one hipGraph replay.  Four batches -- three of 8 clouds and a last,
    shorter one of 5 (a second graph) -- through the adapter and through the plain eager loop on a twin model, both consuming the
    torch CPU generator from the same seed (FPS starts, normal flips): the predictions of every batch and the parameters, BatchNorm
    running statistics and Adam moments at the end agree -- the capture's warm-up passes leave no trace (snapshot / in-place restore)."""
    from models.repsurf.repsurf_ssg_umb import Model
    from repsurf_amd.graph import TrainLoopStep
    from repsurf_amd.optim import Adam
    from util.utils import SmoothClsLoss
    torch_executor.set_backend("hip")
    crit = SmoothClsLoss()
    batches = []
    for i, b in enumerate((8, 8, 8, 5)):
        pts = torch.from_numpy(cloud(20 + i, b, 1024)).cuda().permute(0, 2, 1).contiguous()
        batches.append((pts, (torch.arange(b).cuda() * (i + 1)) % 15))

    def fresh():
        m = Model(ref_args())
        name_seeded_init(m)
        disable_dropout(m)           # (dropout masks come from the device generator, whose stream differs between capture and eager)
        m = m.cuda().train()
        return m, Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)

    def state(m, opt):
        t = [p.detach().flatten() for p in m.parameters()] + [b.detach().flatten().float() for b in m.buffers()]
        t += [opt.state[p][k].flatten() for p in m.parameters() for k in ("exp_avg", "exp_avg_sq")]
        return torch.cat(t).cpu()

    eager, eopt = fresh()
    torch.manual_seed(77)
    want = []
    for pts, lab in batches:
        eopt.zero_grad()
        pred = eager(pts)
        loss = crit(pred, lab.long())
        loss.backward()
        eopt.step()
        want.append((pred.detach().cpu().clone(), loss.item()))
    graphed, gopt = fresh()
    step = TrainLoopStep(graphed, crit, gopt)
    torch.manual_seed(77)
    got = []
    for pts, lab in batches:
        pred, loss = step(pts, lab.long())
        got.append((pred.detach().cpu().clone(), loss.item()))
    assert len(step.steps) == 2                                   # one graph per batch shape
    for (pg, lg), (pw, lw) in zip(got, want):
        assert (pg - pw).abs().max() <= 1e-4 * max(pw.abs().max().item(), 1.0), (pg - pw).abs().max()
        assert abs(lg - lw) <= 1e-4
    sg, se = state(graphed, gopt), state(eager, eopt)
    assert torch.isfinite(sg).all()
    assert (sg - se).norm() / se.norm() < 1e-3, ((sg - se).norm() / se.norm()).item()
    assert int(gopt._dev[0]["step"].item()) == len(batches) == int(eopt._dev[0]["step"].item())
    step.close()
