"""hipGraph capture of a whole training step (forward + loss + backward + optimizer).

At B=32 the step is ~170 kernel launches of 5–300 µs each; launched eagerly from Python the host
cannot keep ahead of the GPU (≈2 ms of a 9 ms step is launch gap).  The step has static shapes and
static memory (PyTorch's graph-private pool), every kernel of librepsurf_hip launches on the
stream it is handed, and the only host work inside a forward — the reference's CPU-generator
draws — is hoisted into `rng.StaticDraws` buffers that are refilled before each replay.  So the step
is captured once (`torch.cuda.CUDAGraph` = hipGraph on ROCm) and replayed.
"""
import os

import torch

from . import dist as _rdist
from . import head as _head
from . import mlp_hip
from . import rng
from . import streams as _streams


class GraphedStep:
    def __init__(self, net, criterion, optimizer, points, label, warmup=3, record_calls=False):
        """record_calls=True: keep (name, arguments) of the GEMM-family ABI calls made while the step is CAPTURED in
        `self.recorded_calls` (repsurf_amd._lib.record_calls).  Their pointers lie in this graph's private memory pool, which lives
        as long as this object: replaying them on their own (bench.py times the matrix-pipe launches of a step that way) touches
        nothing but that pool."""
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.recorded_calls = None
        self.points, self.label = points, label
        self.draws = rng.StaticDraws(label.device)
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
            with torch.cuda.graph(self.graph, **_capture_mode()):
                if record_calls:
                    from . import _lib
                    _lib.record_calls(True)
                try:
                    self.loss = self._body()
                finally:
                    if record_calls:
                        self.recorded_calls = _lib.record_calls(False)
        torch.cuda.synchronize()

    def _body(self):
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for p in self.net.parameters():
                p.grad = None
        mlp_hip.owned_pass.check(self.net.parameters())
        with mlp_hip.owned_pass():        # gradients start as None and are read only after backward (mlp_hip.owned_pass)
            self.pred = self.net(self.points)
            loss = self.criterion(self.pred, self.label)
            loss.backward(_head.unit_gradient(loss.device)) if loss.dim() == 0 and loss.dtype == torch.float32 else loss.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def close(self):
        torch.cuda.synchronize()
        self.graph = self.loss = None
        torch.cuda.synchronize()

    def __call__(self):
        self.draws.refill()        # fresh FPS starts / normal flips, same CPU-generator order as eager
        if hasattr(self.optimizer, "sync_hyper"):
            self.optimizer.sync_hyper()    # learning-rate schedule -> device (repsurf_amd.optim.Adam)
        self.graph.replay()
        mlp_hip.weights_changed()      # the replay updated the parameters through raw pointers
        return self.loss


def snapshot_training_state(net, optimizer):
    """Parameters, buffers (BatchNorm running statistics) and the optimizer's device state, cloned: what the eager warm-up passes of a
    capture change and `restore_training_state` puts back IN PLACE (the captured graphs hold the tensors' addresses)."""
    tensors = list(net.parameters()) + list(net.buffers())
    had_state, counters = {}, []
    if optimizer is not None:
        had_state = {id(p): set(st.keys()) for p, st in optimizer.state.items()}
        for st in optimizer.state.values():
            tensors += [v for v in st.values() if torch.is_tensor(v) and v.is_cuda]
        counters = [c for gst in getattr(optimizer, "_dev", {}).values() for c in (gst["step"], gst["done"])]
    return [(t, t.detach().clone()) for t in tensors + counters], had_state


def restore_training_state(optimizer, snap):
    saved, had_state = snap
    with torch.no_grad():
        for t, v in saved:
            t.copy_(v)
        if optimizer is not None:
            for p, st in optimizer.state.items():        # moments that did not exist before the warm-up: back to zero, in place
                for k, v in st.items():
                    if torch.is_tensor(v) and v.is_cuda and k not in had_state.get(id(p), ()):
                        v.zero_()
            for gi, gst in getattr(optimizer, "_dev", {}).items():
                if not any(gst["step"] is t for t, _ in saved):
                    gst["step"].zero_()
                    gst["done"].zero_()
    mlp_hip.weights_changed()


class TrainLoopStep:
    """The body of the reference's training loop as ONE hipGraph replay, for a loop that is otherwise left as it is.

    classification/tool/train_cls_scanobjectnn.py:226-238 runs, per batch,
        optimizer.zero_grad(); pred = classifier(points); loss = criterion(pred, target.long()); loss.backward(); optimizer.step()
    as ~450 launches issued eagerly from Python: on this package's kernels that loop is HOST-bound (bench.py `eager_clouds_per_s`:
    ~1 000 clouds/s at B = 32 against ~21 000 for the same step replayed as a graph).  With this adapter the five statements become
        step = TrainLoopStep(classifier, criterion, optimizer)            # once, before the epoch loop
        pred, loss = step(points, target.long())                          # per batch
    and everything around them -- loader, sample(), augmentation, accuracy bookkeeping, scheduler.step(), checkpoints -- stays.

    * The step is captured the first time a batch shape is seen (the last, shorter batch of an epoch gets a graph of its own; at most
      `max_graphs` shapes are kept, further ones run eagerly).  Capturing needs a few eager warm-up passes; parameters, BatchNorm
      running statistics and the optimizer's moments / step count are snapshotted before and restored IN PLACE after them, so the
      sequence of updates is the eager loop's: one per batch, on that batch.
    * `optimizer` must be repsurf_amd.optim.Adam (torch.optim.Adam's rule, state dict and param_groups; its learning rate lives
      on the device, so LR schedulers keep working across replays).  `classifier.train()` must be in effect.
    * `pred` and `loss` are the graph's static output tensors: read them (`.max(1)`, `.item()`) before the next call.
    * The reference's CPU-generator draws (FPS start points, normal flips) are consumed in the eager order (rng.StaticDraws).
    * Classification only: a packed segmentation batch changes its cloud boundaries every step -- RaggedSegStep serves those."""

    def __init__(self, net, criterion, optimizer, warmup=2, max_graphs=3):
        from .optim import Adam
        if not isinstance(optimizer, Adam):
            raise TypeError("TrainLoopStep needs repsurf_amd.optim.Adam (torch.optim.Adam keeps its step count on the host and cannot be replayed)")
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.warmup, self.max_graphs = warmup, max_graphs
        self.steps = {}

    def _snapshot(self):
        return snapshot_training_state(self.net, self.optimizer)

    def _restore(self, snap):
        restore_training_state(self.optimizer, snap)

    def __call__(self, points, target):
        if not self.net.training:
            raise RuntimeError("TrainLoopStep: the network is in eval mode")
        key = (tuple(points.shape), points.dtype, tuple(target.shape), target.dtype)
        step = self.steps.get(key)
        if step is None and len(self.steps) >= self.max_graphs:
            self.optimizer.zero_grad()
            pred = self.net(points)
            loss = self.criterion(pred, target)
            loss.backward()
            self.optimizer.step()
            return pred, loss
        if step is None:
            torch.cuda.synchronize()
            cpu_rng, snap = torch.get_rng_state(), self._snapshot()
            cuda_rng = torch.cuda.get_rng_state()
            step = GraphedStep(self.net, self.criterion, self.optimizer, points.clone(), target.clone(), warmup=max(1, self.warmup))
            torch.cuda.synchronize()
            self._restore(snap)
            torch.set_rng_state(cpu_rng)                   # the warm-up passes drew FPS starts / flips / dropout masks: the loop's sequence starts here
            torch.cuda.set_rng_state(cuda_rng)
            self.steps[key] = step
        else:
            step.points.copy_(points, non_blocking=True)
            step.label.copy_(target, non_blocking=True)
        loss = step()
        return step.pred, loss

    def close(self):
        for st in self.steps.values():
            st.close()
        self.steps = {}


def _offset_slot(x):
    """Position of the packed batch's offsets in a list input -- the LAST element by the reference's collate convention
    ([coord, feat, offset], segmentation/util/data_util.py:15-23) -- or None for a plain tensor input.  Identified by
    position, never by dtype: any other int32 tensor of an input is data and is copied like the rest."""
    if torch.is_tensor(x):
        return None
    last = x[-1]
    if not (torch.is_tensor(last) and last.dtype == torch.int32 and last.dim() == 1):
        raise TypeError("PipelinedStep: a list input must end with the packed batch's (B,) int32 offsets")
    return len(x) - 1


def _clone_inputs(x):
    """A model input for one of the two buffer sets: a tensor, or a list of tensors (segmentation: [coord, feat, offset]).
    The offsets are SHAPES of the captured graphs (cloud boundaries decide grids, BatchNorm groups and the derived offsets
    of every stage): they stay shared and constant; `_check_offsets` refuses a batch whose offsets differ."""
    if torch.is_tensor(x):
        return x.clone()
    slot = _offset_slot(x)
    return [t if i == slot else t.clone() for i, t in enumerate(x)]


def _copy_inputs(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
        return
    slot = _offset_slot(dst)
    for i, (d, s_) in enumerate(zip(dst, src)):
        if i != slot:
            d.copy_(s_, non_blocking=True)


def _check_offsets(captured, batch):
    """The captured graphs bake the per-cloud row ranges in: a packed batch with the same total row count but other
    cloud boundaries would silently run kNN, FPS and BatchNorm grouping with STALE offsets.  The host copy of an offset
    tensor travels with it (ops.host_offsets: one device->host read per tensor object, none when the collate function built
    it through ops.offsets_tensor), so the comparison costs no synchronisation in a steady loop."""
    slot = _offset_slot(captured)
    if slot is None:
        return
    from . import ops
    have, want = ops.host_offsets(batch[slot]), ops.host_offsets(captured[slot])
    if have != want:
        raise ValueError(f"PipelinedStep was captured for packed batches with row ends {want[:4]}...{want[-1:]} "
                         f"({len(want)} clouds); this batch has {have[:4]}...{have[-1:]} ({len(have)} clouds).  Cloud boundaries "
                         "are shapes of the captured graphs: build a PipelinedStep per batch layout (or pad the clouds to a fixed size).")


class PipelinedStep:
    """Graphed training step with the NEXT batch's geometry computed while the CURRENT batch trains.

    FPS (a 511-pick latency chain on 32 of 256 CUs), ball query and the constructor's kNN / fan features read nothing
    but coordinates, and they head the step: ~0.26 ms during which the GEMM stack cannot start.  A training loop has
    its next batch in hand (the loader runs ahead), so one call replays TWO linear graphs on two streams:
        stream M :  forward(batch s, geometry from the previous call) -> loss -> backward -> optimizer
        stream S :  geometry(batch s+1)  ->  the other of two state sets
    ordered by two events per parity.  (Putting both into ONE graph as parallel branches costs 0.165 ms per replay on
    this runtime even when the side branch is a single 4-byte copy -- measured -- so each graph stays a plain chain.)
    Every call still performs one complete geometry pass and one complete network pass; results are those of
    GraphedStep step for step (the CPU-generator draws are requested in the same order, one step earlier).

        step = PipelinedStep(net, criterion, optimizer, points0, label0)     # geometry of batch 0 runs here
        loss0 = step(points1, label1)        # trains on batch 0, prepares batch 1
        loss1 = step(points2, label2)        # trains on batch 1, prepares batch 2 ...
    `net` must offer `geometry(points, fork=False)` and `forward(points, geo=...)`: the classifiers of this package
    (points = a (B,3,N) tensor) and the segmentation network (points = [coord, feat, offset]; the offsets are constants
    of the captured shapes).
    The returned loss is ready on the caller's current stream."""

    def __init__(self, net, criterion, optimizer, points, label, warmup=3, group=None, sharded=False):
        """sharded=True (data parallel, world_size > 1): gradients go into ONE flat buffer, the optimizer leaves the
        network graphs and every call is  network graph -> RCCL all-reduce -> Adam graph  on stream M, like
        ShardedGraphedStep."""
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.sharded, self.group = sharded, group
        if sharded:
            import torch.distributed as dist
            self.dist = dist
            sync_replicas(net, dist, group)
            # REPSURF_GRAD_BUCKETS=2 (opt-in, measured in DESIGN.md 8): the gradients of the layers closest to the loss
            # (net.early_gradient_modules(): the last SA stage + the head) form bucket 0, whose all-reduce is issued from a
            # tensor hook on that stage's input gradient -- i.e. as soon as they are complete -- and runs on the process
            # group's stream under the rest of the backward pass; bucket 1 follows the backward as before.
            early, self._early_work, self._hook = None, None, None
            if (os.environ.get("REPSURF_GRAD_BUCKETS", "1") == "2" and hasattr(net, "early_gradient_modules")
                    and os.environ.get("REPSURF_CAPTURE_ALLREDUCE", "1") != "0" and self._collective_capturable()):
                mods = net.early_gradient_modules()
                early = [p for m_ in mods for p in m_.parameters()]
                self._hook = mods[0].register_forward_pre_hook(self._arm_early_bucket)      # removed by close()
            self.grads = FlatGrads(list(net.parameters()), early=early)
            self.flat = self.grads.flat
        dev = label.device
        self.points = [_clone_inputs(points), _clone_inputs(points)]
        self.label = [label.clone(), label.clone()]
        self.draws = rng.StaticDraws(dev)
        # M and S on DISJOINT compute units (repsurf_amd.streams: the fan-feature arithmetic of the geometry is not reproducible while the
        # split-product GEMMs share its compute units)
        self.main, self.side = _streams.pair(dev)
        self.comm = torch.cuda.Stream() if sharded else None     # the early bucket's branch of the captured network graph
        self.main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.main), self.draws:
            self.draws.begin_pass()
            self.draws.refill()
            # state[q]: the geometry the network of parity q reads.  It is whatever the geometry pass that runs BESIDE the network of
            # parity 1 - q returns -- no staging copy (round 4 copied every geometry tensor into a second buffer set: two
            # multi-tensor torch launches, 54 us of each step): under capture those tensors live in the geometry graph's pool, stay
            # referenced from here, and every replay rewrites them in place.
            first = net.geometry(self.points[0], fork=False)
            self._copy_state = os.environ.get("REPSURF_PIPE_COPY_STATE", "0") != "0"      # (=1: the round-4 form, a staging copy per replay: A/B and debugging)
            self.state = [first.clone(), first.clone()] if self._copy_state else [first, first]
            for _ in range(warmup):                       # eager warm-up of the whole step on one stream
                self.draws.begin_pass()
                self.draws.refill()
                self._geometry(0)
                self._network(0)
                self._finish()
        self.side.wait_stream(self.main)
        torch.cuda.current_stream().wait_stream(self.main)
        torch.cuda.synchronize()
        # With a process group alive, its watchdog thread polls the end events of the eager collectives issued so far while this
        # thread captures.  Two rules keep that safe (root cause of the round-3/4 aborts: repsurf_amd.dist.all_reduce): no eager
        # collective of this package ever runs on a stream that is captured later (they live on c10d's internal stream), and
        # every capture made while a process group exists is thread-local, so what another thread does is not this capture's error.
        mode = _capture_mode()
        if sharded and self.dist.is_initialized():
            _rdist.barrier(group)      # every rank has finished its warm-up collectives before any rank starts capturing
        # Geometry graphs and network graphs run concurrently: separate memory pools.  The two geometry graphs have a pool EACH:
        # state[1 - p] lives in the pool of geometry graph p and is read by the network that runs beside geometry graph 1 - p --
        # in a shared pool the temporaries the first capture freed would be handed to the second capture's OUTPUTS, and a replay
        # of the first graph would scribble over the state the network beside it is reading (seen: memory faults in the
        # segmentation step; with the round-4 staging copy the aliasing was harmless, the two geometry graphs never overlap).
        self.g_geo, self.g_net, self.loss = [], [], []
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with self.draws:
                self.draws.begin_pass()
                with torch.cuda.graph(g, pool=(self.g_geo[0].pool() if (self.g_geo and self._copy_state) else None), stream=self.side, **mode):
                    self._geometry(p)
            self.g_geo.append(g)
        # Sharded: ONE replay per rank-step when RCCL accepts stream capture -- network + gradient pack + all-reduce + Adam in the
        # same graph, no host hop between them (round 2: network graph -> eager collective -> Adam graph, two launches and a
        # host-paced gap per step).  capture_error_mode="thread_local": the process group's watchdog thread polls events while
        # this thread captures.  If the capture throws (a build without capturable collectives, gloo in the CPU / one-device
        # tests) the round-2 form is built instead; `self.collective_captured` says which one runs.
        self.collective_captured = False
        if sharded and os.environ.get("REPSURF_CAPTURE_ALLREDUCE", "1") != "0" and self._collective_capturable():
            try:
                nets, losses = [], []
                for p in (0, 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=nets[0].pool() if nets else None, stream=self.main, **mode):
                        losses.append(self._network(p))
                        self._reduce()
                        if optimizer is not None:
                            optimizer.step()
                    nets.append(g)
                self.g_net, self.loss, self.collective_captured = nets, losses, True
            except Exception as e:  # noqa: BLE001 - any capture failure: fall back to the three-part form below
                import sys
                print(f"[repsurf_amd.graph] the collective could not be captured ({e!r}); using network graph -> eager all-reduce -> Adam graph", file=sys.stderr)
                torch.cuda.synchronize()
                self.g_net, self.loss = [], []
        self.graph_opt = None
        if not self.collective_captured:
            for p in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.g_net[0].pool() if self.g_net else None, stream=self.main, **mode):
                    self.loss.append(self._network(p))
                self.g_net.append(g)
            if sharded and optimizer is not None:
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, pool=self.g_net[0].pool(), stream=self.main, **mode):
                    optimizer.step()
        torch.cuda.synchronize()
        # the geometry the first call's network consumes (batch 0 sits in both input buffers): one replay of the graph that writes state[0]
        with torch.cuda.stream(self.side), self.draws:
            self.draws.begin_pass()
            self.draws.refill()
            if os.environ.get("REPSURF_PIPE_SKIP_GEO", "0") == "0":
                self.g_geo[1].replay()
        torch.cuda.synchronize()
        self.geo_done = [torch.cuda.Event(), torch.cuda.Event()]    # geo_done[q]: state[q] / points[q] are ready
        self.net_done = [torch.cuda.Event(), torch.cuda.Event()]    # net_done[q]: the network finished reading them
        for q in (0, 1):
            self.geo_done[q].record(self.side)
            self.net_done[q].record(self.main)
        self.parity = 0

    def close(self):
        """Detach from the model and release the captured graphs.  The forward pre-hook of the two-bucket mode is bound to this step
        (a copy.deepcopy of the model, or a second PipelinedStep on it, would carry / duplicate it); the graphs of a sharded step
        hold recorded collectives of the process group's communicator and must be gone, with the device idle, before another step
        is built on that communicator or the communicator is destroyed (repsurf_amd.dist.finish calls this)."""
        if getattr(self, "_hook", None) is not None:
            self._hook.remove()
            self._hook = None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.g_net, self.g_geo, self.graph_opt, self.loss = [], [], None, []
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _geometry(self, p):
        """geometry of the batch the NEXT call trains on (buffers 1 - p), as one serial chain"""
        if os.environ.get("REPSURF_PIPE_SKIP_GEO", "0") == "0":     # (=1: measurement only -- the network alone)
            fresh = self.net.geometry(self.points[1 - p], fork=False)
            if self._copy_state:
                self.state[1 - p].copy_(fresh)
            else:
                self.state[1 - p] = fresh

    def _network(self, p):
        if self.sharded:
            self.grads.clear()
        elif self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for q in self.net.parameters():
                q.grad = None
        mlp_hip.owned_pass.check(self.net.parameters())
        with mlp_hip.owned_pass():        # gradients start as None and are read only after backward (mlp_hip.owned_pass)
            loss = self.criterion(self.net(self.points[p], geo=self.state[p]), self.label[p])
            loss.backward(_head.unit_gradient(loss.device)) if loss.dim() == 0 and loss.dtype == torch.float32 else loss.backward()
        if self.sharded:
            self.grads.pack(1 if len(self.grads.buckets) == 2 else None)
        elif self.optimizer is not None:
            self.optimizer.step()
        return loss

    def _reduce(self):
        if self.sharded and self.dist.is_initialized():
            if len(self.grads.buckets) == 2:
                self.grads.all_reduce_mean(self.dist, self.group, bucket=1)
                if self._early_work is not None:
                    self._early_work.wait()          # the current stream joins bucket 0's collective
                    self._early_work = None
            else:
                self.grads.all_reduce_mean(self.dist, self.group)

    def _arm_early_bucket(self, module, args):
        """forward pre-hook of the first early-gradient module: its feature input's gradient marks the moment every gradient
        of bucket 0 exists (autograd runs AccumulateGrad nodes ahead of everything else that is ready)."""
        if not (torch.is_grad_enabled() and len(self.grads.buckets) == 2):
            return
        feats = [t for t in args if torch.is_tensor(t) and t.requires_grad]
        if feats:
            feats[-1].register_hook(self._early_bucket)

    def _early_bucket(self, grad):
        mlp_hip.flush_reduces()                      # weight-gradient sums still riding with a later launch (owned_pass)
        self.grads.pack(0)
        if self.dist.is_initialized():
            if torch.cuda.is_current_stream_capturing():
                # a forked branch of the capture on this step's own stream: c10d's internal stream stays out of every capture
                # (eager collectives keep their events there -- repsurf_amd.dist.all_reduce)
                cur = torch.cuda.current_stream()
                self.comm.wait_stream(cur)
                with torch.cuda.stream(self.comm):
                    self.grads.all_reduce_mean(self.dist, self.group, bucket=0)
                self._early_work = _StreamJoin(self.comm)
            else:
                self._early_work = self.grads.all_reduce_mean(self.dist, self.group, bucket=0, async_op=True)
        return None

    def _collective_capturable(self):
        """RCCL (backend "nccl") collectives can be recorded into a hipGraph; gloo's run on the host and cannot."""
        d = self.dist
        return d.is_available() and d.is_initialized() and d.get_backend(self.group) == "nccl"

    def _finish(self):
        """what follows the network graph in sharded mode (eagerly during warm-up, as a graph afterwards; nothing when the
        collective and the optimizer were captured into the network graph)"""
        if not self.sharded or getattr(self, "collective_captured", False):
            return
        self._reduce()
        if self.optimizer is not None:
            if getattr(self, "graph_opt", None) is not None:
                self.graph_opt.replay()
            else:
                self.optimizer.step()

    def __call__(self, next_points=None, next_label=None, sync=True):
        """Enqueue the network of the current batch, then (once the previous network has let go of the other buffer set)
        the geometry of the next one.  The two orderings between the streams are HOST waits on events: on this runtime a
        stream that waits for an event recorded behind a graph launch on another stream costs that other stream
        ~0.1 ms per call (measured: 1.75 ms network alone, 1.85 ms with the wait, 1.75 ms with the record only).  The
        host is always one network ahead of the GPU, so stream M never runs dry.
        sync=True also orders the caller's current stream after this step (the returned loss can be read right away);
        a loop that reads the loss only now and then passes sync=False and synchronises when it does."""
        p = self.parity
        if next_points is not None:
            _check_offsets(self.points[1 - p], next_points)      # before anything is enqueued: a refused batch leaves the step as it was
        caller = torch.cuda.current_stream()
        self.geo_done[p].synchronize()                     # state[p], points[p], label[p]: written by the previous call
        with torch.cuda.stream(self.main):
            if hasattr(self.optimizer, "sync_hyper"):
                self.optimizer.sync_hyper()
            self.g_net[p].replay()
            self._finish()
            self.net_done[p].record(self.main)
        self.net_done[1 - p].synchronize()                 # the previous network is done with buffers 1 - p
        with torch.cuda.stream(self.side):
            if next_points is not None or next_label is not None:
                _streams.after(self.side, caller)          # they were produced on the caller's stream
            if next_points is not None:
                _copy_inputs(self.points[1 - p], next_points)
            if next_label is not None:
                self.label[1 - p].copy_(next_label, non_blocking=True)
            self.draws.refill()        # the draws of the batch whose geometry this call computes
            self.g_geo[p].replay()
            self.geo_done[1 - p].record(self.side)
        if sync:
            caller.wait_event(self.net_done[p])
        self.parity = 1 - p
        mlp_hip.weights_changed()
        return self.loss[p]


def sync_replicas(net, dist, group=None):
    """Rank 0's parameters and buffers to every rank (what DistributedDataParallel does at construction): the sharded
    steps only ever exchange gradients, so replicas that were not built from the same seed would diverge silently."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return
    with torch.no_grad():
        for t in list(net.parameters()) + list(net.buffers()):
            _rdist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)      # (never a raw dist.broadcast: repsurf_amd.dist)
    mlp_hip.weights_changed()


class FlatGrads:
    """The all-reduce buffer of the sharded steps.  Backward runs with p.grad = None, so autograd hands every gradient
    tensor over as it is (no `grad += g` kernel per parameter: ~70 dependent graph nodes); `pack()` then copies them
    into ONE contiguous fp32 buffer with a multi-tensor launch and points every p.grad at its slice of that buffer --
    what the collective reduces and the optimizer reads.  Under graph capture the copy is part of the network graph and
    the slices' addresses are what the optimizer graph bakes in."""

    def __init__(self, params, early=None):
        """early: parameters whose gradients are complete FIRST in backward (the layers closest to the loss): they lead the flat
        buffer as bucket 0, the rest is bucket 1 -- reverse execution order, so bucket 0's collective can run under the rest of
        the backward pass (PipelinedStep, REPSURF_GRAD_BUCKETS=2)."""
        params = [p for p in params if p.requires_grad]
        early_ids = {id(p) for p in (early or [])}
        lead = [p for p in params if id(p) in early_ids]
        self.params = lead + [p for p in params if id(p) not in early_ids]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=self.params[0].dtype, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        n0 = sum(p.numel() for p in lead)
        self.split = len(lead)                                    # params [0, split) form bucket 0
        self.buckets = [self.flat[:n0], self.flat[n0:]] if 0 < n0 < total else [self.flat]

    def clear(self):
        for p in self.params:
            p.grad = None

    def pack(self, bucket=None):
        """bucket None: every parameter; 0 / 1: the parameters of that bucket only (bucket 0 mid-backward: a gradient that is
        not there yet is an ordering bug of the caller, not a zero)."""
        lo, hi = (0, len(self.params)) if bucket is None or len(self.buckets) == 1 else ((0, self.split) if bucket == 0 else (self.split, len(self.params)))
        src, dst = [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                if bucket == 0:
                    raise RuntimeError("FlatGrads.pack(0): an early-bucket gradient has not been produced yet")
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            p.grad = v

    def all_reduce_mean(self, dist, group=None, bucket=None, async_op=False):
        """Average the flat buffer (or one bucket of it) over the ranks; async_op=True returns the work handle (its wait()
        orders the current stream after the collective)."""
        world = dist.get_world_size(group)
        # (REPSURF_FORCE_ALLREDUCE=1: issue the collective on a 1-rank group too -- what lets a single-GPU box exercise the
        #  captured-collective path end to end; a 1-rank all-reduce is the identity)
        if world <= 1 and os.environ.get("REPSURF_FORCE_ALLREDUCE", "0") == "0":
            return None
        buf = self.flat if bucket is None or len(self.buckets) == 1 else self.buckets[bucket]
        if dist.get_backend(group) == "nccl" and getattr(self, "_avg_ok", True):    # RCCL averages in the collective
            try:
                return _rdist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
            except (RuntimeError, ValueError):           # a build without ncclAvg refuses at call time: sum and scale
                self._avg_ok = False
        work = _rdist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if not async_op:
            buf.div_(world)
            return None
        return _ScaledWork(work, buf, world)      # the mean is complete once the caller has waited


class _StreamJoin:
    """wait() orders the current stream after everything enqueued on `stream` so far (a captured branch rejoining its graph)"""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        return True


def _capture_mode():
    """keyword arguments of every torch.cuda.graph(...) of this module: thread-local error mode whenever a process group (and
    with it a watchdog thread that talks to the HIP runtime) exists"""
    import torch.distributed as dist
    return {"capture_error_mode": "thread_local"} if (dist.is_available() and dist.is_initialized()) else {}


class _ScaledWork:
    """async SUM all-reduce standing in for AVG: wait() joins the collective, then scales the buffer on the current stream"""

    def __init__(self, work, buf, world):
        self.work, self.buf, self.world = work, buf, world

    def wait(self):
        self.work.wait()
        self.buf.div_(self.world)
        return True


def attach_flat_grads(params):
    """One contiguous fp32 buffer holding every gradient; each p.grad becomes a view into it, so autograd
    accumulates in place and ONE collective (or one memset) covers the whole model — what DDP's
    gradient_as_bucket_view does, without the DDP module.  Returns the flat buffer."""
    params = [p for p in params if p.requires_grad]
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


class ShardedGraphedStep:
    """Data-parallel step for world_size > 1: two captured graphs around one RCCL all-reduce.

        graph A   grads.zero_()  ->  forward  ->  loss  ->  backward (accumulates into the flat buffer)
        eager     all_reduce(flat, AVG)            one 5.9 MB ring all-reduce over xGMI
        graph B   optimizer.step()

    Every rank works on its own clouds with its own BatchNorm statistics (the reference's default);
    nothing else is exchanged.  Capturing the compute keeps the host out of the ~200 launches per
    step; the collective stays outside the graphs so no RCCL capture support is assumed."""

    def __init__(self, net, criterion, optimizer, points, label, group=None, warmup=3):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.points, self.label = points, label
        sync_replicas(net, dist, group)
        self.grads = FlatGrads(list(net.parameters()))
        self.flat = self.grads.flat
        self.draws = rng.StaticDraws(label.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), self.draws:
            for _ in range(warmup):
                self.draws.begin_pass()
                self.draws.refill()
                self._fwd_bwd()
                self._reduce()
                if optimizer is not None:
                    optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        with self.draws:
            self.draws.begin_pass()
            self.draws.refill()
            with torch.cuda.graph(self.graph_a, **_capture_mode()):
                self.loss = self._fwd_bwd()
        self.graph_b = None
        if optimizer is not None:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), **_capture_mode()):
                optimizer.step()
        torch.cuda.synchronize()

    def _fwd_bwd(self):
        self.grads.clear()
        mlp_hip.owned_pass.check(self.net.parameters())
        with mlp_hip.owned_pass():        # gradients start as None and are read only after backward (mlp_hip.owned_pass)
            loss = self.criterion(self.net(self.points), self.label)
            loss.backward(_head.unit_gradient(loss.device)) if loss.dim() == 0 and loss.dtype == torch.float32 else loss.backward()
        self.grads.pack()
        return loss

    def _reduce(self):
        if self.dist.is_initialized():
            self.grads.all_reduce_mean(self.dist, self.group)

    def close(self):
        """release the captured graphs with the device idle (before another step is built, or the process group destroyed)"""
        torch.cuda.synchronize()
        self.graph_a = self.graph_b = self.loss = None
        torch.cuda.synchronize()

    def __call__(self):
        self.draws.refill()
        self.graph_a.replay()
        self._reduce()
        if self.graph_b is not None:
            if hasattr(self.optimizer, "sync_hyper"):
                self.optimizer.sync_hyper()
            self.graph_b.replay()
        mlp_hip.weights_changed()
        return self.loss


class RaggedSegStep:
    """ONE captured network graph for packed segmentation batches whose cloud boundaries change every step.

    The reference's loader concatenates clouds of whatever sizes it drew (segmentation/util/data_util.py:15-23, consumed at
    segmentation/tool/train.py:280-290): the total row count, every cloud's rows and therefore the row count of every level differ from
    batch to batch, while a hipGraph freezes grids and scalar arguments -- `PipelinedStep` refuses such batches, and the eager loop is
    host-bound (8.4 ms per 16-cloud step against 3.3 ms captured, profiles/r06).  Here
      * the NETWORK (forward, loss, backward, optimizer: ~400 launches) is captured once for a row CAPACITY: tensors are allocated for
        `capacity` rows at level 0 (capacity // stride at level 1, ...), every launch is sized for its level's capacity, and the
        kernels that reduce over rows read the batch's counts from a device table the host refills before each replay
        (repsurf_amd.ragged: `rows_dev` of include/repsurf_hip.h).  Labels beyond the batch's rows hold `ignore_index`;
      * the GEOMETRY (FPS, kNN, fan features, 3-NN: ~50 launches that read coordinates only) runs EAGERLY -- its launches are sized by
        the cloud boundaries the host knows from the collate function -- on a side stream for the NEXT batch, beside the current
        batch's network graph (overlap=False: behind it), and its results
        are copied into capacity-sized state buffers (indices beyond the batch's rows keep older, in-range values; the inverse
        index' offsets are padded with their last value).

        step = RaggedSegStep(net, criterion, optimizer, batch0, label0, capacity=16 * 4096)     # geometry of batch 0 runs here
        loss0 = step(batch1, label1)          # trains on batch 0, prepares batch 1
        loss1 = step(batch2, label2)          # trains on batch 1, prepares batch 2 ...
    Results: those of the eager loop up to the summation order of the row reductions (their slab boundaries follow the capacity,
    not the batch): tests/test_seg_gpu.py::test_ragged_seg_step_* compares three different ragged batches step for step.
    Limits: level-0 capacity a multiple of 256; clouds of at most `max_cloud_rows` rows (above 16 384 the grouping's backward of the
    first stage(s) is the atomic scatter instead of the gather over ops.inverse_index); fp32; per-GPU BatchNorm statistics; criterion = repsurf_amd.head.CrossEntropyLoss (ignored rows get exact
    zero gradients) with mean reduction."""

    def __init__(self, net, criterion, optimizer, batch, label, capacity=None, warmup=2, ignore_index=None, restore=True, capture=True,
                 max_cloud_rows=None, overlap=None):
        """overlap=True (default): the next batch's eager geometry runs BESIDE the current network graph, as in PipelinedStep -- 1.6x
        faster at BASELINE's sizes than overlap=False, where the side stream waits for the network graph first (step = network +
        geometry).  (Until the geometry kernels were rebuilt without compiler-vectorized packed-fp32 code -- Makefile, DESIGN.md section 6 --
        the overlapped form computed 16-point chunks of the fan features from other operands in 5-15 of 1 600 geometry passes; since: 0 of
        4 800, tools/ragged_flake3.py, profiles/r06/eager_beside_graph.txt.)
        max_cloud_rows: the largest cloud (rows at level 0) any batch will hold; default: the capacity (one cloud may fill it).  It
        fixes, per stage, the form of the grouping's backward in the captured graph: the gather over the inverse index
        (ops.inverse_index: deterministic, no atomics) where a cloud of the stage's source level has at most 16 384 rows, the
        atomic scatter (with the group count as device data) above that -- the reference's S3DIS clouds of up to 80 000 points take
        the scatter at the first stage(s), the gather below.
        restore=True: parameters, BatchNorm running statistics and optimizer state are put back after the eager warm-up passes (in
        place), so that the sequence of updates is the eager loop's: one per batch.
        capture=False: the same capacity-sized network launched EAGERLY under the device row counts (no hipGraph) -- the reference a
        replay must equal bit for bit (same kernels, same launch sizes, same summation order), and the fallback when capture fails."""
        from . import ops, ragged
        coord, feat, offset = batch
        dev = coord.device
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.overlap = True if overlap is None else bool(overlap)
        self.ignore = int(ignore_index if ignore_index is not None else getattr(criterion, "ignore_index", 255))
        sas = [net.sa1, net.sa2, net.sa3, net.sa4]
        self.strides = [sa.stride for sa in sas]
        nsample = {sa.nsample for sa in sas}
        if len(nsample) != 1:
            raise ValueError("RaggedSegStep: the abstraction stages must share one nsample")
        n0 = int(coord.shape[0])
        capacity = int(capacity if capacity is not None else n0)
        capacity = -(-max(capacity, n0) // 256) * 256
        self.levels = [capacity]
        for st in self.strides:
            self.levels.append(self.levels[-1] // st)
        self.fan = net.surface_constructor.k
        self.max_cloud_rows = int(max_cloud_rows if max_cloud_rows is not None else capacity)
        self.use_csr, rows = [], self.max_cloud_rows
        for st in self.strides:
            self.use_csr.append(rows <= 16384)          # the stage whose SOURCE level holds clouds of `rows` rows
            rows //= st
        ns = nsample.pop()
        self.caps = [ragged.Capacity(self.levels, ns, self.fan, dev) for _ in (0, 1)]
        self.coord = [torch.zeros((capacity, 3), dtype=torch.float32, device=dev) for _ in (0, 1)]
        self.feat = [torch.zeros((capacity, feat.shape[1]), dtype=torch.float32, device=dev) for _ in (0, 1)]
        self.label = [torch.full((capacity,), self.ignore, dtype=label.dtype, device=dev) for _ in (0, 1)]
        self.offset = ops.offsets_tensor([capacity], dev)      # (the network never reads offsets: a placeholder of the list input)
        self.main, self.side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)      # (streams.pair does not help EAGER launches: see streams.py)
        self.state = [None, None]
        self.counts = [None, None]
        caller = torch.cuda.current_stream()
        self.side.wait_stream(caller)
        with torch.cuda.stream(self.side):
            for q in (0, 1):
                self._prepare(q, batch, label, first=True)
        self.main.wait_stream(self.side)
        torch.cuda.synchronize()
        self.captured = bool(capture)
        snap = snapshot_training_state(net, optimizer) if restore else None
        cpu_rng, cuda_rng = torch.get_rng_state(), torch.cuda.get_rng_state()
        with torch.cuda.stream(self.main):      # eager warm-up of the network on the capture stream
            for _ in range(warmup if capture else 0):
                with self.caps[0]:
                    self._network(0)
        torch.cuda.synchronize()
        self.g_net, self.loss, self.grads = [], [None, None], []
        for p in ((0, 1) if capture else ()):
            g = torch.cuda.CUDAGraph()
            with self.caps[p]:
                with torch.cuda.graph(g, pool=self.g_net[0].pool() if self.g_net else None, stream=self.main, **_capture_mode()):
                    self.loss[p] = self._network(p)
            self.g_net.append(g)
            self.grads.append([q.grad for q in net.parameters()])      # graph p's gradient tensors (what its optimizer launch reads)
        torch.cuda.synchronize()
        if restore and capture:
            restore_training_state(optimizer, snap)
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(cuda_rng)
        torch.cuda.synchronize()
        self.geo_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.net_done = [torch.cuda.Event(), torch.cuda.Event()]
        for q in (0, 1):
            self.geo_done[q].record(self.side)
            self.net_done[q].record(self.main)
        self.parity = 0
        mlp_hip.weights_changed()

    # ---- the batch -> the capacity-sized buffers of parity q (side stream, eager)
    def _level_counts(self, offset):
        from . import ops
        counts, offs = [], [offset]
        for st in self.strides:
            offs.append(ops.strided_offset(offs[-1], st) if st > 1 else offs[-1])
        for o in offs:
            host = ops.host_offsets(o)
            counts.append(host[-1] if host else 0)
        host0 = ops.host_offsets(offset)
        largest = max((e - s for s, e in zip((0,) + host0[:-1], host0)), default=0)
        return counts, largest

    def _prepare(self, q, batch, label, first=False):
        coord, feat, offset = batch
        counts, largest = self._level_counts(offset)
        for li, (n, cap) in enumerate(zip(counts, self.levels)):
            if n > cap:
                raise ValueError(f"RaggedSegStep: this batch holds {n} rows at level {li}, the step was captured for at most {cap} (capacity {self.levels[0]})")
        if largest > self.max_cloud_rows:
            raise ValueError(f"RaggedSegStep: a cloud of {largest} rows, the step was built for clouds of at most {self.max_cloud_rows} (max_cloud_rows)")
        if feat.shape[1] != self.feat[q].shape[1]:
            raise ValueError("RaggedSegStep: feature width differs from the captured one")
        n0 = counts[0]
        self.coord[q][:n0].copy_(coord, non_blocking=True)
        self.feat[q][:n0].copy_(feat, non_blocking=True)
        self.label[q].fill_(self.ignore)
        self.label[q][:n0].copy_(label, non_blocking=True)
        fresh = self.net.geometry([self.coord[q][:n0], self.feat[q][:n0], offset])
        if first and self.state[q] is None:
            self.state[q] = self._capacity_state(fresh)
        self._scatter(self.state[q], fresh, counts)
        self.caps[q].fill(counts)
        self.counts[q] = counts

    def _capacity_state(self, fresh):
        """zero-filled capacity-sized twins of the geometry tensors (indices 0 = in range everywhere)"""
        from models.repsurf.repsurf_umb_ssg import SegGeoState
        from modules.repsurface_utils import StageGeometry
        lv, dev = self.levels, fresh.feat.device

        def z(shape, like):
            return torch.zeros(shape, dtype=like.dtype, device=dev)
        feat = z((lv[0],) + tuple(fresh.feat.shape[1:]), fresh.feat)
        stages = []
        for li, g in enumerate(fresh.stages):
            if g.fps_idx is None:
                raise ValueError("RaggedSegStep: every stage must sample (stride > 1)")
            if self.use_csr[li] and g.csr is None:
                raise ValueError("RaggedSegStep: the geometry carries no inverse grouping index (training mode, REPSURF_GATHER_BACKWARD=1 expected)")
            m = lv[li + 1]
            csr = (torch.zeros((lv[li] + 1,), dtype=torch.int32, device=dev), torch.zeros((m * g.group_idx.shape[1],), dtype=torch.int32, device=dev)) if self.use_csr[li] else None
            stages.append(StageGeometry(z((m,), g.fps_idx), z((m, 3), g.new_center), self.offset, z((m,) + tuple(g.group_idx.shape[1:]), g.group_idx), csr))
        fps = []
        for (fine, _coarse), f in zip(((3, 4), (2, 3), (1, 2), (0, 1)), fresh.fps):
            if len(f) > 2 and f[2] is not None:
                raise ValueError("RaggedSegStep: REPSURF_INTERP_GATHER=1 is not supported (the interpolation's backward scatters under a capacity)")
            fps.append((z((lv[fine], 3), f[0]), z((lv[fine], 3), f[1]), None))
        return SegGeoState(feat, stages, fps, None if fresh.moments is None else torch.zeros_like(fresh.moments))

    def _scatter(self, state, fresh, counts):
        pairs = [(state.feat, fresh.feat)]
        if fresh.moments is not None:
            pairs.append((state.moments, fresh.moments))
        for li, (s_, g) in enumerate(zip(state.stages, fresh.stages)):
            pairs += [(s_.fps_idx, g.fps_idx), (s_.new_center, g.new_center), (s_.group_idx, g.group_idx)]
            if s_.csr is not None:
                if g.csr is None:
                    raise ValueError(f"RaggedSegStep: stage {li} was captured with the gather form of the grouping's backward, this batch's geometry has no inverse index")
                pairs += [(s_.csr[1], g.csr[1]), (s_.csr[0], g.csr[0])]
        for s_, f in zip(state.fps, fresh.fps):
            pairs += [(s_[0], f[0]), (s_[1], f[1])]
        for dt in sorted({d.dtype for d, _ in pairs}, key=str):      # one multi-tensor launch per dtype
            sel = [(d[:s_.shape[0]], s_) for d, s_ in pairs if d.dtype == dt]
            torch._foreach_copy_([a for a, _ in sel], [b_ for _, b_ in sel])
        for li, (s_, g) in enumerate(zip(state.stages, fresh.stages)):
            if s_.csr is not None:      # source rows beyond the batch's count read zero edges: the offsets continue with the total
                s_.csr[0][g.csr[0].shape[0]:].fill_(int(g.csr[1].shape[0]))

    def _network(self, p):
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for q in self.net.parameters():
                q.grad = None
        mlp_hip.owned_pass.check(self.net.parameters())
        with mlp_hip.owned_pass():
            self.pred = self.net([self.coord[p], self.feat[p], self.offset], geo=self.state[p])
            loss = self.criterion(self.pred, self.label[p])
            loss.backward(_head.unit_gradient(loss.device)) if loss.dim() == 0 and loss.dtype == torch.float32 else loss.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def rows(self, parity=None):
        """level row counts of the batch the NEXT call trains on (or of the given parity)"""
        return self.counts[self.parity if parity is None else parity]

    def close(self):
        torch.cuda.synchronize()
        self.g_net, self.loss, self.grads = [], [None, None], []
        torch.cuda.synchronize()

    def __call__(self, next_batch=None, next_label=None, sync=True):
        """Replay the network on the batch prepared by the previous call (or the constructor), then prepare `next_batch` (its eager
        geometry runs on the side stream while the network graph runs).  The returned loss is ready on the caller's stream (sync=True)."""
        p = self.parity
        caller = torch.cuda.current_stream()
        self.geo_done[p].synchronize()
        with torch.cuda.stream(self.main):
            if hasattr(self.optimizer, "sync_hyper"):
                self.optimizer.sync_hyper()
            if self.captured:
                self.g_net[p].replay()
            else:
                with self.caps[p]:
                    self.loss[p] = self._network(p)
            self.net_done[p].record(self.main)
        self.net_done[1 - p].synchronize()
        if next_batch is not None:
            with torch.cuda.stream(self.side):
                self.side.wait_stream(caller)
                if not self.overlap:
                    self.side.wait_event(self.net_done[p])      # (overlap=False: the geometry starts when the network graph has finished)
                self._prepare(1 - p, next_batch, next_label)
                self.geo_done[1 - p].record(self.side)
        if sync:
            caller.wait_event(self.net_done[p])
        self.parity = 1 - p
        mlp_hip.weights_changed()
        return self.loss[p]
