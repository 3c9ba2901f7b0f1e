#!/usr/bin/env python3
"""Stream-ordering stress of RaggedSegStep (GPU box): artificial delays (torch.cuda._sleep) on the side stream (the next batch's eager geometry)
or on the main stream (the network graph), every loss against the eager capacity-sized run.  A missing dependency shows as a wrong loss."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_seg_gpu import _seg_model, _ragged_batches
from tests.util import subproject
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
layouts, batches, labels = _ragged_batches()
crit = CrossEntropyLoss(ignore_index=255)
SLEEP = int(float(os.environ.get("SLEEP_CYCLES", "3e8")))
with subproject("segmentation"):
    base = _seg_model()
    base.surface_constructor.random_inv = False
    ref = RaggedSegStep(copy.deepcopy(base), crit, None, batches[0], labels[0], capacity=4096, capture=False)
    want = [ref(batches[(s + 1) % 4], labels[(s + 1) % 4]).item() for s in range(8)]
    ref.close()
    print("want", want[:4])
    for mode in ("none", "side", "main", "both", "host"):
        step = RaggedSegStep(copy.deepcopy(base), crit, None, batches[0], labels[0], capacity=4096)
        orig = step._prepare

        def delayed(q, batch, label, first=False, _o=orig, _m=mode):
            if _m in ("side", "both"):
                torch.cuda._sleep(SLEEP)              # (the current stream is the side stream here)
            return _o(q, batch, label, first)
        step._prepare = delayed
        got = []
        for s in range(8):
            if mode in ("main", "both"):
                with torch.cuda.stream(step.main):
                    torch.cuda._sleep(SLEEP)
            if mode == "host":
                import time; time.sleep(0.05)
            got.append(step(batches[(s + 1) % 4], labels[(s + 1) % 4], sync=(s % 2 == 0)))
        torch.cuda.synchronize()
        vals = [g.item() for g in got]
        # NOTE: with sync=False the returned static tensor is read later: it then holds the LAST replay of that parity
        step.close()
        bad = [i for i, (a, b) in enumerate(zip(vals[-2:], want[-2:])) if a != b]
        print(mode, "last two losses", vals[-2:], "expected", want[-2:], "MISMATCH" if bad else "ok")
