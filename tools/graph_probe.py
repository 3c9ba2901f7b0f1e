#!/usr/bin/env python3
"""Isolate what breaks hipGraph capture of the step: run variants in subprocesses."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
ROOT = os.environ["RS_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "repsurf_amd", "classification"))
import argparse
from models.repsurf.repsurf_ssg_umb import Model
from util.utils import SmoothClsLoss
from repsurf_amd.graph import GraphedStep
variant = sys.argv[1]
a = argparse.Namespace(num_point=1024, return_dist=True, return_center=True, return_polar=True, group_size=8, umb_pool="sum", cuda_ops=True, num_class=15)
torch.manual_seed(0)
m = Model(a).cuda().train()
pts = (torch.rand(32, 3, 1024, device="cuda") * 2 - 1); lab = torch.randint(0, 15, (32,), device="cuda")
crit = SmoothClsLoss()
opt = None
if "adam" in variant:
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused="fused" in variant, capturable=True)
if "sgd" in variant:
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
def eager():
    for p in m.parameters(): p.grad = None
    l = crit(m(pts), lab); l.backward()
    if opt is not None: opt.step()
    return l.item()
if "pre" in variant:
    for _ in range(2): eager()
    torch.cuda.synchronize()
g = GraphedStep(m, crit, opt, pts, lab, warmup=3)
for _ in range(5): g()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): l = g()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("OK", variant, "ms/step %.3f" % (dt * 1e3), "loss", float(l))
'''
env = dict(os.environ, RS_ROOT=ROOT)
for v in ["noopt", "noopt_pre", "sgd", "adam", "adam_fused", "adam_fused_pre"]:
    r = subprocess.run([sys.executable, "-c", CHILD, v], env=env, capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or ["-"])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l][-2:]
    print(v, "rc", r.returncode, tail, err, flush=True)
