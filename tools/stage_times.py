#!/usr/bin/env python3
"""Profiler-free stage costs of the classification step: capture partial pipelines as hipGraphs and time replays.
(rocprofv3 inflates every kernel by a few microseconds and disturbs stream overlap; replay timing does not.)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "repsurf_amd", "classification")):
    sys.path.insert(0, p)
import torch
from repsurf_amd import rng, ops
from models.repsurf.repsurf_ssg_umb import Model
from util.utils import SmoothClsLoss

dev = torch.device("cuda")
args = argparse.Namespace(num_point=1024, return_dist=True, return_center=True, return_polar=True,
                          group_size=8, umb_pool="sum", cuda_ops=True, num_class=15)
torch.manual_seed(0)
model = Model(args).to(dev).train()
crit = SmoothClsLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
g = torch.Generator().manual_seed(1)
points = (torch.rand(32, 1024, 3, generator=g) * 2 - 1).permute(0, 2, 1).contiguous().to(dev)
label = torch.randint(0, 15, (32,), generator=g).to(dev)


def graph_time(body, reps=40, label_=""):
    draws = rng.StaticDraws(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), draws:
        for _ in range(2):
            draws.begin_pass(); draws.refill(); body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with draws:
        draws.begin_pass(); draws.refill()
        with torch.cuda.graph(gr):
            keep = body()
    torch.cuda.synchronize()
    for _ in range(5):
        draws.refill(); gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        draws.refill(); gr.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{label_:50s} {us:9.1f} us", flush=True)
    return us


def zero():
    for p in model.parameters():
        p.grad = None


def full(optim=True, backward=True):
    def body():
        zero()
        loss = crit(model(points), label)
        if backward:
            loss.backward()
        if optim:
            opt.step()
        return loss
    return body


def fwd_nograd():
    with torch.no_grad():
        return model(points)


center = points[:, :3, :]
xyz = center.permute(0, 2, 1).contiguous()


def constructor_only(backward):
    def body():
        zero()
        out = model.surface_constructor(center)
        if backward:
            out.sum().backward()
        return out
    return body


def geometry_only():
    from repsurf_amd.geometry import GeometryPlan
    plan = GeometryPlan(xyz, model._sampling)
    return [plan.stage(i) for i in range(len(model._sampling))]


def fps_only():
    return ops.furthestsampling(xyz, 512, rng.draw("fps", 32, 1024, dev))


def head_only():
    feat = torch.randn(32, 1024, device=dev, requires_grad=True)
    def body():
        zero()
        loss = crit(torch.log_softmax(model.classfier(feat), -1), label)
        loss.backward()
        return loss
    return body


def adam_only():
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    def body():
        opt.step()
    return body


graph_time(full(True, True), label_="forward + loss + backward + Adam")
graph_time(full(False, True), label_="forward + loss + backward")
graph_time(full(False, False), label_="forward + loss (autograd graph recorded)")
graph_time(fwd_nograd, label_="forward, no_grad")
graph_time(constructor_only(False), label_="umbrella constructor forward")
graph_time(constructor_only(True), label_="umbrella constructor forward + backward")
graph_time(geometry_only, label_="geometry plan (FPS1, ball1, FPS2, ball2)")
graph_time(fps_only, label_="FPS 1024 -> 512 alone")
graph_time(head_only(), label_="head + loss forward + backward (32 rows)")
graph_time(adam_only(), label_="Adam step alone")
model.overlap_geometry = False
graph_time(full(True, True), label_="full step, geometry on the main stream")
