"""repsurf_amd.optim.Adam (csrc/adam.hip) against torch.optim.Adam, the optimizer the reference's training tool builds
(classification/tool/train_cls_scanobjectnn.py:179-185).  Tolerance: the update is the same fp32 formula up to the
rounding of fused multiply-adds, so parameters agree to rtol 2e-6 / atol 1e-7 after 6 steps."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 10, 1, 1), (64,), (64,), (128, 64, 1, 1), (128,), (1, 7), (3,), (513, 129), (40, 1024), (15,)] * 5   # 50 tensors > one launch


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in SHAPES]


def _set_grads(ps, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).to(p.device) * 0.1


@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_adam_matches_torch(wd):
    from repsurf_amd.optim import Adam
    dev = torch.device("cuda:0")
    a, b = _params(dev), _params(dev)
    ours, ref = Adam(a, lr=1e-3, weight_decay=wd), torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    sched_o = torch.optim.lr_scheduler.StepLR(ours, step_size=2, gamma=0.7)
    sched_r = torch.optim.lr_scheduler.StepLR(ref, step_size=2, gamma=0.7)
    for it in range(6):
        _set_grads(a, it); _set_grads(b, it)
        ours.step(); ref.step()
        ours._dev_reset_host()       # eager use: the schedule is picked up at the next step
        sched_o.step(); sched_r.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)
    so, sr = ours.state_dict(), ref.state_dict()
    assert so["state"].keys() == sr["state"].keys()
    for k in sr["state"]:
        assert float(so["state"][k]["step"]) == float(sr["state"][k]["step"]) == 6.0
        torch.testing.assert_close(so["state"][k]["exp_avg"], sr["state"][k]["exp_avg"], rtol=2e-6, atol=1e-8)
        torch.testing.assert_close(so["state"][k]["exp_avg_sq"], sr["state"][k]["exp_avg_sq"], rtol=2e-6, atol=1e-10)


def test_adam_resumes_from_torch_state_dict():
    from repsurf_amd.optim import Adam
    dev = torch.device("cuda:0")
    a, b = _params(dev), _params(dev)
    ref = torch.optim.Adam(b, lr=2e-3, weight_decay=1e-4)
    for it in range(3):
        _set_grads(b, it)
        ref.step()
    with torch.no_grad():
        for x, y in zip(a, b):
            x.copy_(y)
    ours = Adam(a, lr=1e-3, weight_decay=0.0)
    ours.load_state_dict(copy.deepcopy(ref.state_dict()))           # brings lr / weight decay / moments / step along
    for it in range(3, 5):
        _set_grads(a, it); _set_grads(b, it)
        ours.step(); ref.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)


def test_adam_in_a_captured_graph_follows_the_schedule():
    from repsurf_amd.optim import Adam
    dev = torch.device("cuda:0")
    a, b = _params(dev), _params(dev)
    ours, ref = Adam(a, lr=1e-3), torch.optim.Adam(b, lr=1e-3)
    static = [torch.zeros_like(p) for p in a]
    for p, g in zip(a, static):
        p.grad = g
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ours.step()                      # eager warm-up with zero gradients and zero weight decay: a no-op on the values
    torch.cuda.current_stream().wait_stream(side)
    _set_grads(b, 99)
    for p in b:
        p.grad.zero_()
    ref.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ours.step()
    for it in range(4):
        _set_grads(b, it)
        for g, p in zip(static, b):
            g.copy_(p.grad)
        if it == 2:
            ours.param_groups[0]["lr"] = 5e-4
            ref.param_groups[0]["lr"] = 5e-4
        ours.sync_hyper()
        graph.replay()
        ref.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)


def test_adam_two_launches_per_step_and_changed_betas():
    """90 tensors are two launches per step (RS_ADAM_MAX = 80): only the last one advances the step counter and leaves the next
    step's bias corrections on the device; after a change of betas those cached corrections belong to other betas and the
    kernel has to compute its own (csrc/adam.hip)."""
    from repsurf_amd.optim import Adam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [(33, 7), (64,), (5, 5, 1, 1)] * 30
    a = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ours, ref = Adam(a, lr=1e-3, weight_decay=1e-4), torch.optim.Adam(b, lr=1e-3, weight_decay=1e-4)
    for it in range(6):
        if it == 3:
            for opt in (ours, ref):
                opt.param_groups[0]["betas"] = (0.8, 0.99)
            ours._dev_reset_host()
        gg = torch.Generator().manual_seed(100 + it)
        for x, y in zip(a, b):
            x.grad = torch.randn(x.shape, generator=gg).to(dev) * 0.1
            y.grad = x.grad.clone()
        ours.step(); ref.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)
    assert float(ours.state_dict()["state"][0]["step"]) == 6.0
