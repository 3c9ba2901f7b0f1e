"""Sampling/grouping geometry of all abstraction stages, issued ahead of time on a second HIP stream.

FPS and ball query of every stage depend only on the input coordinates (stage i samples the centres
stage i-1 picked), never on learned features, while the umbrella constructor (kNN + fan features +
its MLP) also depends only on the coordinates.  FPS is a 511-step latency chain that occupies one
workgroup per cloud — 32 of 256 CUs at B=32 — so it is launched on a side stream and runs underneath
the constructor instead of in front of the first abstraction stage.  Under hipGraph capture the
fork/join becomes two parallel branches of the graph.
"""
import torch

from . import _lib, rng

_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device)
    return _streams[key]


class StageGeometry:
    """index: the compacted groups' bookkeeping (ops.CompactIndex) when the plan was asked for it -- like everything here it
    depends on coordinates only, and the grouped MLP of the stage would otherwise build it on the critical path."""
    __slots__ = ("fps_idx", "new_center", "idx", "cnt", "index", "center")

    def __init__(self, fps_idx, new_center, idx, cnt=None, index=None, center=None):
        self.fps_idx, self.new_center, self.idx, self.cnt, self.index = fps_idx, new_center, idx, cnt, index
        self.center = center       # the (B, n, 3) contiguous coordinates this stage sampled from (the owner's tensor, not copied here)

    def tensors(self):
        out = [self.fps_idx, self.new_center, self.idx, self.cnt]
        if self.index is not None:
            # (grp, slot, src are rows of one (3, capacity) allocation: copied as that one tensor)
            out += [self.index.offsets, self.index.mult]
            out += [self.index.meta] if self.index.meta is not None else [self.index.grp, self.index.slot, self.index.src]
            if self.index.pts is not None:
                out += [self.index.pts]
        if any(t is None for t in out):
            raise RuntimeError("StageGeometry.tensors: a stage without ball-query counts cannot be part of a pipelined state set")
        return out

    def clone(self):
        from .ops import CompactIndex
        index = None
        if self.index is not None:
            i = self.index
            if i.meta is not None:
                index = CompactIndex.from_meta(i.offsets.clone(), i.mult.clone(), i.meta.clone(), None if i.pts is None else i.pts.clone())
                index.csr_ready = i.csr_ready
            else:
                index = CompactIndex(i.offsets.clone(), i.mult.clone(), i.grp.clone(), i.slot.clone(), i.src.clone())
        return StageGeometry(self.fps_idx.clone(), self.new_center.clone(), self.idx.clone(), self.cnt.clone(), index)


class GeometryPlan:
    """plan = GeometryPlan(xyz (B,N,3), [(npoint, radius, nsample), ...]); plan.stage(i) joins the side stream
    on first use and returns that stage's (fps_idx, new_center, ball idx)."""

    def __init__(self, xyz, stages, fork=True, compact=False):
        """fork=False: everything on the current stream (the caller already runs this off the critical path).
        compact=True: also build every stage's ops.CompactIndex (scan of the ball-query counts + row bookkeeping)."""
        from .ops import CompactIndex
        from . import ops as ops_mod
        self.stages = []
        self.events = []
        self.main = torch.cuda.current_stream()
        side = _side_stream(xyz.device) if fork else self.main
        b, dev = xyz.shape[0], xyz.device
        # FPS start indices are drawn here, in stage order — after the constructor's flip, which the
        # caller draws first — so the CPU-generator sequence is the reference's.
        sizes = self._sizes(xyz.shape[1], stages)
        starts = [rng.draw("fps", b, n, dev) for n in sizes]
        # Outputs are allocated on the MAIN stream (the allocator then orders their reuse against the
        # consumers); only the kernels run on the side stream, between a fork and a join event.
        for (npoint, radius, nsample), n_src in zip(stages, sizes):
            self.stages.append(StageGeometry(torch.empty((b, npoint), dtype=torch.int32, device=dev),
                                             torch.empty((b, npoint, 3), dtype=torch.float32, device=dev),
                                             torch.empty((b, npoint, nsample), dtype=torch.int32, device=dev),
                                             torch.empty((b, npoint), dtype=torch.int32, device=dev),
                                             CompactIndex.empty(b * npoint, nsample, dev, points=b * n_src) if compact else None))
        if fork:
            side.wait_stream(self.main)
        with torch.cuda.stream(side):
            st_ptr = side.cuda_stream
            center, n = xyz, xyz.shape[1]
            for (npoint, radius, nsample), start, g in zip(stages, starts, self.stages):
                r2 = torch.tensor(float(radius) ** 2, dtype=torch.float32).item()
                _lib.call("rs_furthestsampling", b, n, npoint, center.data_ptr(), start.data_ptr(), None,
                          g.fps_idx.data_ptr(), st_ptr)
                _lib.call("rs_gather_rows", b, n, npoint, 3, center.data_ptr(), g.fps_idx.data_ptr(),
                          g.new_center.data_ptr(), st_ptr)
                _lib.call("rs_ballquery", b, n, npoint, r2, nsample, g.new_center.data_ptr(), center.data_ptr(),
                          g.idx.data_ptr(), g.cnt.data_ptr(), st_ptr)
                if g.index is not None:
                    ci = g.index
                    _lib.call("rs_compact_index", b, n, npoint, nsample, g.idx.data_ptr(), g.cnt.data_ptr(), ci.offsets.data_ptr(),
                              ci.grp.data_ptr(), ci.slot.data_ptr(), ci.src.data_ptr(), ci.mult.data_ptr(), st_ptr)
                    if ci.pts is not None and n <= 16384 and ops_mod.COMPACT_CSR:      # the inverse of `src`: the grouping's backward becomes a gather (ops.COMPACT_CSR)
                        _lib.call("rs_compact_csr", b, n, npoint, ci.src.data_ptr(), ci.offsets.data_ptr(), g.fps_idx.data_ptr(),
                                  ci.pts.data_ptr(), ci.pts.data_ptr() + 4 * (b * n + 1), ci.csr_rows.data_ptr(), st_ptr)
                        ci.csr_ready = True
                g.center = center
                center, n = g.new_center, npoint
                if fork:
                    self.events.append(side.record_event())      # stage i is usable as soon as ITS kernels are done
        self.keep = (xyz, starts)
        self.joined = [not fork] * len(self.stages)

    @staticmethod
    def _sizes(n, stages):
        out = []
        for npoint, _, _ in stages:
            out.append(n)
            n = npoint
        return out

    def stage(self, i):
        if not self.joined[i]:
            self.main.wait_event(self.events[i])
            self.joined[i] = True
        return self.stages[i]
