"""hipGraph capture of a whole training step (forward + loss + backward + optimizer).

At B=32 the step is ~170 kernel launches of 5–300 µs each; launched eagerly from Python the host
cannot keep ahead of the GPU (≈2 ms of a 9 ms step is launch gap).  The step has static shapes and
static memory (PyTorch's graph-private pool), every kernel of librepsurf_hip launches on the
stream it is handed, and the only host work inside a forward — the reference's CPU-generator
draws — is hoisted into `rng.StaticDraws` buffers that are refilled before each replay.  So the step
is captured once (`torch.cuda.CUDAGraph` = hipGraph on ROCm) and replayed.
"""
import torch

from . import rng


class GraphedStep:
    def __init__(self, net, criterion, optimizer, points, label, warmup=3):
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.points, self.label = points, label
        self.draws = rng.StaticDraws(points.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), self.draws:
            for _ in range(warmup):                       # eager warm-up on the capture stream
                self.draws.begin_pass()
                self.draws.refill()
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with self.draws:
            self.draws.begin_pass()
            self.draws.refill()
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
        torch.cuda.synchronize()

    def _body(self):
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for p in self.net.parameters():
                p.grad = None
        loss = self.criterion(self.net(self.points), self.label)
        loss.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def __call__(self):
        self.draws.refill()        # fresh FPS starts / normal flips, same CPU-generator order as eager
        self.graph.replay()
        return self.loss
