"""HIP executor of the grouped shared-MLP stacks (product path of repsurf_amd.mlp).

Orchestrates the kernels of repsurf_amd/csrc/mlp.hip through the C ABI:
  forward  per layer: rs_mlp_gemm_rows (previous BN+ReLU fused into the operand load, BN sums in the
           epilogue) -> rs_bn_finalize (both BatchNorms of a two-branch first layer in one
           rs_bn_finalize_batch launch);  last layer -> rs_pool_max (BN+ReLU+max over nsample)
  backward per layer: rs_mlp_wgrad + rs_mlp_gemm_rows (BN-backward affine fused into the operand
           load, ReLU mask + BN-backward sums in the epilogue) -> rs_backward_tail (the BatchNorm-backward
           finalize(s) of the layer below plus the fixed-order sums of up to four pending weight-gradient
           partials in one launch)
Only the pre-BatchNorm conv outputs (bf16 tensors in bf16 mode, `stores_bf16`) and per-channel vectors are
saved for backward; the gradient through the max-pool is never materialised (RS_OP_POOLED operand).

Gradients of conv biases that feed a BatchNorm are returned as exact zeros: BatchNorm removes any
per-channel constant, so the analytic gradient is 0 (the reference's autograd produces rounding
noise of 1e-7..1e-3 there).
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib
from . import ragged as _ragged
from . import zeros as _zeros

c_int, c_ll, P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
OP_ID, OP_RELU1, OP_RELU2, OP_AFF2, OP_POOLED, OP_BCAST = range(6)
EPI_STORE, EPI_STATS, EPI_MASK = range(3)
FUSED_UMBRELLA = True     # 10-channel constructor MLP through csrc/umbrella_mlp.hip (False: generic row-GEMM path)
DEBUG = None              # set to a dict to capture backward intermediates (tools/mlp_debug.py)
PARTIAL_BLOCKS = int(os.environ.get("REPSURF_PARTIAL_BLOCKS", "512"))      # rows of the BatchNorm partial-sum buffers (>= persistent workgroups)


def partial_rows(rows, per_block=64):
    """Rows of a BatchNorm partial-sum buffer for a launch over `rows` rows: one per workgroup that can exist (a workgroup
    takes >= 64 rows; the pooled backward 4 groups), at most PARTIAL_BLOCKS.  The producing kernel zeroes the rows it does not
    own and the finalize kernel reads all of them: a 4 096-row launch with the full 512 rows wrote and re-read 7 MB of zeros
    per 1 024-channel layer."""
    return max(1, min(PARTIAL_BLOCKS, -(-int(rows) // per_block)))


WGRAD_CHUNKS = int(os.environ.get("REPSURF_WGRAD_CHUNKS", "512"))        # workgroups of one weight-gradient launch (row slabs x output blocks)
WGRAD_MIN_ROWS = int(os.environ.get("REPSURF_WGRAD_MIN_ROWS", "64"))    # rows per row-workgroup of the weight gradient, at least (sweep: 32..128, profiles/r02/wgrad_chunk_sweep.txt)


class RowOperand(ctypes.Structure):          # rs_row_operand
    _fields_ = [("a", P), ("lda", c_ll), ("b", P), ("ldb", c_ll), ("s1", P), ("t1", P), ("s2", P), ("t2", P),
                ("arg", P), ("ns", c_int), ("mode", c_int), ("mult", P), ("grp", P), ("slot", P),
                ("a_bf16", c_int), ("b_bf16", c_int)]


class Epilogue(ctypes.Structure):            # rs_mlp_epilogue
    _fields_ = [("bias", P), ("out", P), ("ldo", c_ll), ("mode", c_int),
                ("my1", P), ("ldm1", c_ll), ("ms1", P), ("mt1", P), ("mean1", P), ("invstd1", P),
                ("my2", P), ("ldm2", c_ll), ("ms2", P), ("mt2", P), ("mean2", P), ("invstd2", P),
                ("partial", P), ("partial_blocks", c_int),
                ("pool_ns", c_int), ("pool_max", P), ("pool_min", P), ("pool_amax", P), ("pool_amin", P),
                ("row_mult", P), ("out_bf16", c_int), ("my1_bf16", c_int), ("my2_bf16", c_int),
                ("w3", P), ("ldw3", c_int), ("w3_part", c_ll)]


class BnItem(ctypes.Structure):              # rs_bn_item
    _fields_ = [("c", c_int), ("nblk", c_int), ("rows", c_ll), ("partial", P), ("gamma", P), ("beta", P), ("eps", ctypes.c_float),
                ("momentum", ctypes.c_float), ("scale", P), ("shift", P), ("save_mean", P), ("save_invstd", P), ("running_mean", P),
                ("running_var", P), ("rows_dev", P)]


class BnBwdItem(ctypes.Structure):           # rs_bn_bwd_item
    _fields_ = [("c", c_int), ("nblk", c_int), ("nstat", c_int), ("which", c_int), ("rows", c_ll), ("partial", P), ("scale", P),
                ("mean", P), ("invstd", P), ("p", P), ("q", P), ("r", P), ("dgamma", P), ("dbeta", P), ("rows_dev", P)]


class ReduceItem(ctypes.Structure):          # rs_reduce_item
    _fields_ = [("chunks", c_int), ("n", c_ll), ("partial", P), ("out", P)]


TAIL_FIN_MAX, TAIL_RED_MAX, BN_BATCH_MAX = 2, 4, 4


class BackwardTail(ctypes.Structure):        # rs_backward_tail_work
    _fields_ = [("nfin", c_int), ("nred", c_int), ("fin", BnBwdItem * TAIL_FIN_MAX), ("red", ReduceItem * TAIL_RED_MAX)]


def _stream():
    return _lib.current_stream()


def _ptr(t, offset=0):
    return None if t is None else t.data_ptr() + t.element_size() * offset


def _bf(t):
    """1 for a bf16 tensor (bf16 activation storage), else 0"""
    return int(t is not None and t.dtype == torch.bfloat16)


# bf16 mode (mlp.PRECISION == "bf16", BASELINE configs[4]): an SA stack stores its pre-BatchNorm conv outputs -- the only large
# tensors it writes in forward and re-reads in backward -- as bf16 (torch.autocast stores them the same way); the BatchNorm
# sums, the pooled activations, every gradient and every parameter stay fp32.  The kernels fix which tensor of a launch is
# bf16 by its operand mode (csrc/mlp.hip, "storage roles"), so a stack stores either ALL its conv outputs as bf16 or none:
# all, when every layer width is a multiple of 4 (8-byte vector stores).  REPSURF_BF16_STORE=0: fp32 storage.
BF16_STORE = os.environ.get("REPSURF_BF16_STORE", "1") != "0"


def stores_bf16(widths):
    from . import mlp as _mlp
    return bool(BF16_STORE and _mlp.PRECISION == "bf16" and all(c % 4 == 0 for c in widths))


def operand(mode, a, lda, b=None, ldb=0, s1=None, t1=None, s2=None, t2=None, arg=None, ns=1, a_off=0, rs=None):
    """rs: RowSet — attaches the per-row multiplicity / ragged group maps of a compacted row set."""
    op = RowOperand(_ptr(a, a_off), lda, _ptr(b), ldb, _ptr(s1), _ptr(t1), _ptr(s2), _ptr(t2),
                    None if arg is None else arg.data_ptr(), ns, mode)
    op.a_bf16, op.b_bf16 = _bf(a), _bf(b)
    if rs is not None and rs.mult is not None and mode in (OP_AFF2, OP_POOLED):
        op.mult = _ptr(rs.mult)
        if mode == OP_POOLED:
            op.grp, op.slot = rs.grp.data_ptr(), rs.slot.data_ptr()
    return op


class RowSet:
    """The rows a stack works on: dense (cap rows, all valid) or compacted (valid count on the device)."""
    __slots__ = ("cap", "dev", "full", "mult", "grp", "slot", "offsets")

    def __init__(self, cap, compact=None):
        self.cap = cap
        if compact is None:
            self.dev, self.full, self.mult, self.grp, self.slot, self.offsets = None, cap, None, None, None, None
        else:
            self.dev, self.full = compact["rows_dev"], compact["rows_full"]
            self.mult, self.grp, self.slot, self.offsets = compact["mult"], compact["grp"], compact["slot"], compact["offsets"]


_pending_counters = []


_defer_counters = 0


def _flush_counters(force=False):
    """num_batches_tracked += 1 for every BatchNorm of the stack in ONE multi-tensor kernel (of the whole model when
    the caller wraps its forward in `deferred_counters()`)."""
    if _pending_counters and (force or not _defer_counters):
        torch._foreach_add_(_pending_counters, 1)
        _pending_counters.clear()


class deferred_counters:
    """with deferred_counters(): ...several stacks... -> one counter update for all of them at exit."""

    def __enter__(self):
        global _defer_counters
        _defer_counters += 1
        return self

    def __exit__(self, *exc):
        global _defer_counters
        _defer_counters -= 1
        if _defer_counters == 0:
            _flush_counters(force=True)


class BNVec:
    """Per-channel vectors of one BatchNorm for one forward pass (+ `sync`: how its statistics were reduced over the ranks)."""
    __slots__ = ("scale", "shift", "mean", "invstd", "sync")

    def __init__(self, c, device):
        buf = torch.empty((4, c), dtype=torch.float32, device=device)
        self.scale, self.shift, self.mean, self.invstd = buf[0], buf[1], buf[2], buf[3]
        self.sync = None


# ---- synchronized BatchNorm (the reference's --sync_bn: nn.SyncBatchNorm.convert_sync_batchnorm(model),
# segmentation/tool/train.py:47,141-142).  The stacks read BatchNorm modules as parameter containers; a module that
# convert_sync_batchnorm turned into nn.SyncBatchNorm asks for statistics over ALL ranks: the fp64 partial sums a producing
# kernel leaves ((workgroups, 2|3, C): sum y, sum y^2 forward; sum dz, sum dz*yhat backward) are summed over the process group
# by ONE all-reduce in front of the finalize launch that reads them, and the finalize divides by the global row count.
# Every rank runs the same batch layout (weak scaling; DistributedSampler + drop_last in the reference), so the global count
# is rows x world.  Under RCCL the collective is recorded into the step's hipGraph like the gradient all-reduce.
SYNC_BN_FORCE = os.environ.get("REPSURF_SYNC_BN_FORCE", "0") != "0"     # tests: also on a 1-rank group (the identity)


def sync_of(bn_mod):
    """(dist, group, world) when this BatchNorm's batch statistics span the process group, else None."""
    if not isinstance(bn_mod, torch.nn.SyncBatchNorm) or not bn_mod.training:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn_mod.process_group
    world = dist.get_world_size(group)
    if world <= 1 and not SYNC_BN_FORCE:
        return None
    return dist, group, world


_sync_mismatch = {}      # device -> int32 counter: SyncBatchNorm reductions whose ranks held DIFFERENT row counts


def sync_row_mismatch_count(device=None):
    """SyncBatchNorm reductions (since the process started) in which the ranks did not all hold the same number of rows.  The global
    row count handed to the finalize kernels is rows x world -- exact for the equal shards of weak scaling (DistributedSampler +
    drop_last), wrong for ragged ones -- so a non-zero count means the statistics of those layers were mis-normalised.  The check
    rides in the reduction itself (sum rows, sum rows^2) and costs no host synchronisation; read this where the loss is read."""
    if device is None:
        return sum(int(t.item()) for t in _sync_mismatch.values())
    t = _sync_mismatch.get(_lib.device_key(device))
    return 0 if t is None else int(t.item())


def sync_partials(part, sync, rows=None):
    """Sum a partial-sum tensor (nblk, nstat, C) over the ranks, in place: row 0 ends up holding the sum over every rank's rows
    and the other rows zero, so the finalize that adds the nblk rows sees the global sums.  Returns the factor the row count
    grows by (world).  The collective's shape, (nstat * C + 2,), does not depend on nblk: ranks with different row counts (ragged
    packed batches) have different nblk and an all-reduce of `part` itself would hang or corrupt; `rows` (this rank's
    count) travels with the sums as (rows, rows^2), and world * sum rows^2 != (sum rows)^2 counts a mismatch
    (sync_row_mismatch_count)."""
    if sync is None:
        return 1
    dist, group, world = sync
    from . import dist as _rdist
    flat = part.sum(0).reshape(-1)
    n = float(part.shape[0] if rows is None else rows)
    buf = torch.cat([flat, torch.full((1,), n, dtype=part.dtype, device=part.device),
                     torch.full((1,), n * n, dtype=part.dtype, device=part.device)])
    _rdist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)      # (eager: c10d's own stream; captured: this stream)
    part[1:].zero_()
    part[0].copy_(buf[:-2].view_as(part[0]))
    if rows is not None:
        key = _lib.device_key(part.device)
        if key not in _sync_mismatch:
            if part.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SyncBatchNorm: the first synchronized pass on a device must run eagerly (warm-up) before capture")
            _sync_mismatch[key] = torch.zeros((1,), dtype=torch.int32, device=part.device)
        bad = (buf[-1] * world - buf[-2] * buf[-2]).abs() > 0.5
        _sync_mismatch[key].add_(bad.to(torch.int32))
    return world


def _w2d(w):
    return w.detach().reshape(w.shape[0], -1).contiguous()


def _pad4(w):
    """zero-pad the last dimension to a multiple of 4 floats (the row GEMM reads weights as float4)"""
    pad = (-w.shape[1]) % 4
    return w if pad == 0 else torch.nn.functional.pad(w, (0, pad))


PACK_MAX = 32


class PackArgs(ctypes.Structure):            # rs_pack_weights_args
    _fields_ = [("src", P * PACK_MAX), ("dst", P * PACK_MAX), ("cout", c_int * PACK_MAX), ("cin", c_int * PACK_MAX),
                ("ld", c_int * PACK_MAX), ("transpose", c_int * PACK_MAX), ("n", c_int), ("dst3", P * PACK_MAX), ("ld3", c_int * PACK_MAX)]


# Round 5: weights are constant during a step, yet the split-product row GEMM (rs_mlp_gemm_split3) split its weight tile into
# three bf16 parts again in every row tile of every launch -- half of the loop's split arithmetic and all of its weight LDS
# stores.  The pack launch at the top of a step now also writes every weight operand as three bf16 parts (the `w3` operand of
# rs_mlp_epilogue), and the split-product instances run ONLY on such an image: `gemm_rows` finds it by the address of the fp32
# operand it is handed, or makes it on the spot (one small launch: operands nobody prepacked -- unit tests, eval forwards).
# Round 6: the image is TILE-ORDERED -- (3, ld3 / 8, outer, 8): 16-byte units of 8 consecutive k, contiguous over the rows of one k-octet --
# so that the row GEMM moves a plane of its weight tile global -> LDS with one global_load_lds_dwordx4 (include/repsurf_hip.h: rs_mlp_epilogue.w3).
_split3 = {}      # (address, shape) of the fp32 operand (n-major) -> (image (3, ld3 / 8, outer, 8) bf16, ld3, weights epoch, source version, source, fp32 operand)


def _split3_key(t):
    return (t.data_ptr(), tuple(t.shape))


def _presplit_on():
    from . import mlp as _mlp      # late: mlp imports this module on first use
    return _mlp.PRECISION != "bf16" and gemm_split3()      # (the bf16-operand kernels never read an image)


def _split3_of(wk):
    hit = _split3.get(_split3_key(wk))
    if hit is None or hit[2] != _weights_epoch or hit[3] != hit[4]._version:
        return None
    return hit


def _pack_items(items, device):
    """items = [(w2d (cout, cin), transpose)] -> zero-padded n-major copies, ONE launch per 32 of them:
    transpose=False -> (cout, pad4(cin)) for the forward GEMM (the weight itself when it needs no copy), transpose=True ->
    (cin, pad4(cout)) for dY . W.  With the split products on, every operand's three-part bf16 image is written by the same launch."""
    inner = [w.shape[0] if tr else w.shape[1] for w, tr in items]
    outer = [w.shape[1] if tr else w.shape[0] for w, tr in items]
    inplace = [(not tr) and not _needs_copy(w) for w, tr in items]      # forward weights used in place: only the split image
    lds = [(-(-n // 4)) * 4 for n in inner]
    sizes = [0 if ip else o * ld for o, ld, ip in zip(outer, lds, inplace)]
    flat = torch.empty((max(1, sum(-(-sz // 4) * 4 for sz in sizes)),), dtype=torch.float32, device=device)
    outs, off = [], 0
    for (w, _), o, ld, sz, ip in zip(items, outer, lds, sizes, inplace):
        outs.append(w if ip else flat[off:off + sz].view(o, ld))
        off += -(-sz // 4) * 4                      # keep every copy 16-byte aligned
    split = _presplit_on()
    ld3s = [(-(-n // 32)) * 32 for n in inner]
    if split:
        sz3 = [3 * o * l3 for o, l3 in zip(outer, ld3s)]
        flat3 = torch.empty((sum(-(-sz // 8) * 8 for sz in sz3),), dtype=torch.bfloat16, device=device)
        imgs, off = [], 0
        for o, l3, sz in zip(outer, ld3s, sz3):
            imgs.append(flat3[off:off + sz].view(3, l3 // 8, o, 8))
            off += -(-sz // 8) * 8                  # 16-byte aligned images
    for i in range(0, len(items), PACK_MAX):
        a = PackArgs()
        n = 0
        for j in range(i, min(i + PACK_MAX, len(items))):
            w, tr = items[j]
            if inplace[j] and not split:
                continue                            # nothing to write for this one
            a.src[n], a.dst[n] = w.data_ptr(), (None if inplace[j] else outs[j].data_ptr())
            a.cout[n], a.cin[n], a.ld[n], a.transpose[n] = w.shape[0], w.shape[1], lds[j], int(bool(tr))
            a.dst3[n], a.ld3[n] = (imgs[j].data_ptr(), ld3s[j]) if split else (None, 0)
            n += 1
        a.n = n
        if n:
            _lib.call("rs_pack_weights", ctypes.byref(a), _stream())
    if split:
        if len(_split3) > 512:                      # (eager loops that never prepack: entries pin their operands, so bound the table)
            _split3.clear()
        for (w, _), o, img, l3 in zip(items, outs, imgs, ld3s):
            _split3[_split3_key(o)] = (img, l3, _weights_epoch, w._version, w, o)
    return outs


# Copies made ahead of time for a whole model (prepack): (weight address, transposed) -> (weight version, copy, source).
# The six per-stack pack launches of a step (three forward, three backward) become one at the top of the forward.
# An entry keeps its SOURCE tensor alive: the key is an address, and the allocator would otherwise hand a dead model's
# block to another model's weight of the same version (two freshly initialised models), which then found the dead
# model's transposed copy -- wrong data gradients, seen when a segmentation model followed a classifier in one process.
# Every prepack() call starts from an empty table, so at most one model's weights are pinned.
# Validity: an entry is a copy of the weights AS THEY WERE when prepack() ran.  torch-side updates show in
# `tensor._version`; updates through raw pointers (repsurf_amd.optim.Adam, hipGraph replays) do not, so those call
# `weights_changed()`, which advances `_weights_epoch` and thereby retires every entry: an eval forward after a
# training step packs fresh copies instead of finding first-layer weights that are one optimizer step old.
_prepacked = {}
_weights_epoch = 0


def weights_changed():
    """Tell the copy cache that parameters were modified behind autograd's back (raw-pointer optimizer, graph replay)."""
    global _weights_epoch
    _weights_epoch += 1
    _prepacked.clear()
    _split3.clear()

PREPACK = os.environ.get("REPSURF_PREPACK", "1") != "0"


def _needs_copy(w):
    return bool(w.shape[1] % 4 or w.data_ptr() % 16)


def prepack(convs):
    """Pack, in one launch, what the SA stacks built on these 1x1 convolutions will ask for in this step: the padded
    forward copy of the weights whose cin is not a multiple of 4, and the transposed copy of every weight."""
    items = []
    _prepacked.clear()
    _split3.clear()
    if not PREPACK:
        return
    for conv in convs:
        w = _w2d(conv.weight)
        if _needs_copy(w) or _presplit_on():        # (a weight used in place still gets its split image)
            items.append((w, False))
        items.append((w, True))
    if not items:
        return
    for (w, tr), out in zip(items, _pack_items(items, items[0][0].device)):
        if out.data_ptr() != w.data_ptr():
            _prepacked[(w.data_ptr(), bool(tr))] = (w._version, out, w, _weights_epoch)


def pack_weights(w2ds, transpose, device):
    """Padded (transpose=False) / transposed (transpose=True) copies of several weights; copies `prepack` made of the
    same, unchanged weights are reused, the rest share one launch."""
    tr = bool(transpose)
    outs, miss = [None] * len(w2ds), []
    for i, w in enumerate(w2ds):
        hit = _prepacked.get((w.data_ptr(), tr))
        if (hit is not None and hit[0] == w._version and hit[3] == _weights_epoch and hit[2].shape == w.shape
                and hit[1].device == w.device):
            outs[i] = hit[1]
        else:
            miss.append(i)
    if miss:
        for i, c in zip(miss, _pack_items([(w2ds[i], tr) for i in miss], device)):
            outs[i] = c
    return outs


def fwd_weights(w2ds, device):
    """Forward operands: the conv weights themselves where cin % 4 == 0 (and the base is 16-byte aligned), one batched
    padded copy for the others (the first-layer branches with 6 / 10 / 138 / 266 input channels)."""
    need = [i for i, w in enumerate(w2ds) if _needs_copy(w)]
    outs = list(w2ds)
    if need:
        for i, c in zip(need, pack_weights([w2ds[i] for i in need], False, device)):
            outs[i] = c
    return outs


def gemm_split3():
    """True when the tiled MFMA kernels form fp32 products as six bf16 MFMAs over three-part operands (include/repsurf_hip.h:
    rs_mlp_gemm_split3; the library's default, RS_GEMM_SPLIT3=0 selects the fp32 MFMAs)."""
    return bool(_lib.load().rs_mlp_gemm_split3())


def w_fwd(w2d):
    return fwd_weights([w2d], w2d.device)[0]


def w_bwd(w2d):
    return pack_weights([w2d], True, w2d.device)[0]


def gemm_rows(rows, kdim, cols, x_op, wk, epi, rows_dev=None):
    """out[rows, cols] = E[rows, kdim] . wk[:cols, :kdim]^T   (wk n-major (cols, ld), ld % 4 == 0, zero beyond kdim)"""
    from . import mlp as _mlp      # late: mlp imports this module on first use
    if rows_dev is None:
        rows_dev = _ragged.dev(rows)      # a packed batch under a captured capacity (repsurf_amd.ragged): the count is device data
    epi.w3, epi.ldw3, epi.w3_part = None, 0, 0
    if _presplit_on():
        hit = _split3_of(wk)
        if hit is None:            # nobody packed this operand in this step: its image now (n-major fp32 -> three bf16 parts)
            _pack_items([(wk, False)], wk.device)
            hit = _split3_of(wk)
        epi.w3, epi.ldw3, epi.w3_part = hit[0].data_ptr(), hit[1], hit[0].shape[2] * hit[1]
    _lib.call("rs_mlp_gemm_rows_bf16" if _mlp.PRECISION == "bf16" else "rs_mlp_gemm_rows", rows, rows_dev, kdim, cols, ctypes.byref(x_op), _ptr(wk), wk.shape[1],
              ctypes.byref(epi), _stream())


def fused_pool_ok(cout, nsample):
    """the row GEMM can fold the max over nsample into its epilogue when whole groups sit in one thread's rows"""
    if nsample == 32 and os.environ.get("REPSURF_POOL32_DIRECT", "1") != "0":
        return True          # groups of 32 rows: pooled in the accumulators of a wave's 32-row tile (any width)
    rows_per_thread = 128 // (256 // (32 if cout <= 32 else (64 if cout <= 64 else 128)))
    return rows_per_thread % nsample == 0


def bn_finalize_batch(items):
    """rs_bn_finalize for several layers in one launch; items: what fwd_layer(finalize=False) returned (the item keeps its
    tensors alive)."""
    for i in range(0, len(items), BN_BATCH_MAX):
        chunk = items[i:i + BN_BATCH_MAX]
        arr = (BnItem * len(chunk))(*[it[0] for it in chunk])
        _lib.call("rs_bn_finalize_batch", ctypes.cast(arr, P), len(chunk), _stream())


def fwd_layer(rows, x_op, kdim, w2d, bias, bn_mod, training, device, pool_ns=0, rs=None, wk=None, store_bf16=False, finalize=True):
    """y = E . W^T + bias with BN statistics; returns (y, BNVec[, pooled (out, arg)]).
    rs: RowSet of a compacted operand (device row count, per-row weights of the statistics).
    store_bf16: y is written as bf16 (rounded first; statistics and pooling see the rounded values)."""
    cout = w2d.shape[0]
    rows_dev = rs.dev if rs is not None else None
    bn_rows = rs.full if rs is not None else rows
    y = torch.empty((rows, cout), dtype=torch.bfloat16 if store_bf16 else torch.float32, device=device)
    vec = BNVec(cout, device)
    pool = None
    if training:
        nblk = partial_rows(rows)
        part = torch.empty((nblk, 2, cout), dtype=torch.float64, device=device)
        epi = Epilogue(bias=_ptr(bias), out=_ptr(y), ldo=cout, mode=EPI_STATS, partial=part.data_ptr(),
                       partial_blocks=nblk, out_bf16=_bf(y))
        if rs is not None and rs.mult is not None:
            epi.row_mult = _ptr(rs.mult)
        if pool_ns:
            groups = rows // pool_ns
            ext = torch.empty((2, groups, cout), dtype=torch.float32, device=device)
            pos = torch.empty((2, groups, cout), dtype=torch.int32, device=device)
            epi.pool_ns = pool_ns
            epi.pool_max, epi.pool_min = _ptr(ext[0]), _ptr(ext[1])
            epi.pool_amax, epi.pool_amin = pos[0].data_ptr(), pos[1].data_ptr()
            pool = (ext, pos)
        gemm_rows(rows, kdim, cout, x_op, wk if wk is not None else w_fwd(w2d), epi, rows_dev)
        vec.sync = sync_of(bn_mod)
        bn_rows = bn_rows * sync_partials(part, vec.sync, bn_rows)      # SyncBatchNorm: statistics over every rank's rows
        track = bn_mod.track_running_stats and bn_mod.running_mean is not None
        if track:
            _pending_counters.append(bn_mod.num_batches_tracked)
        mom = bn_mod.momentum if bn_mod.momentum is not None else 0.1
        bn_dev = _ragged.dev(bn_rows)
        if not finalize or bn_dev is not None:
            item = BnItem(c=cout, nblk=nblk, rows=bn_rows, partial=part.data_ptr(), gamma=_ptr(bn_mod.weight), beta=_ptr(bn_mod.bias),
                          eps=float(bn_mod.eps), momentum=float(mom), scale=_ptr(vec.scale), shift=_ptr(vec.shift), save_mean=_ptr(vec.mean),
                          save_invstd=_ptr(vec.invstd), running_mean=_ptr(bn_mod.running_mean) if track else None,
                          running_var=_ptr(bn_mod.running_var) if track else None, rows_dev=bn_dev)
        if not finalize:       # the caller batches this layer's statistics with another layer's (bn_finalize_batch)
            assert pool is None
            return y, vec, (item, part, vec)
        if bn_dev is not None:      # the count lives on the device: the item form carries its address
            bn_finalize_batch([(item, part, vec)])
        else:
            _lib.call("rs_bn_finalize", cout, bn_rows, nblk, part.data_ptr(), _ptr(bn_mod.weight), _ptr(bn_mod.bias),
                      float(bn_mod.eps), float(mom), _ptr(vec.scale), _ptr(vec.shift), _ptr(vec.mean), _ptr(vec.invstd),
                      _ptr(bn_mod.running_mean) if track else None, _ptr(bn_mod.running_var) if track else None, _stream())
        if pool is not None:
            ext, pos = pool
            groups = rows // pool_ns
            out = torch.empty((groups, cout), dtype=torch.float32, device=device)
            arg = torch.empty((groups, cout), dtype=torch.int32, device=device)
            _lib.call("rs_pool_select", groups, cout, _ptr(ext[0]), _ptr(ext[1]), pos[0].data_ptr(), pos[1].data_ptr(),
                      _ptr(vec.scale), _ptr(vec.shift), _ptr(out), arg.data_ptr(), _stream())
            return y, vec, (out, arg)
    else:
        epi = Epilogue(bias=_ptr(bias), out=_ptr(y), ldo=cout, mode=EPI_STORE, out_bf16=_bf(y))
        gemm_rows(rows, kdim, cout, x_op, wk if wk is not None else w_fwd(w2d), epi, rows_dev)
        with torch.no_grad():
            invstd = torch.rsqrt(bn_mod.running_var + bn_mod.eps)
            vec.invstd.copy_(invstd)
            vec.mean.copy_(bn_mod.running_mean)
            vec.scale.copy_(bn_mod.weight * invstd)
            vec.shift.copy_(bn_mod.bias - bn_mod.running_mean * bn_mod.weight * invstd)
    return y, vec


def wgrad_chunks(rows, ncols, kcols):
    """row slabs of the weight-gradient reduction: >= 8 pipeline stages of 32 rows each, <= 2 workgroups
    per CU per output block, partial buffer <= 64 MB"""
    out_blocks = -(-ncols // 128) * (-(-kcols // 128) if kcols > 64 else 1)
    chunks = max(1, min(rows // WGRAD_MIN_ROWS, WGRAD_CHUNKS // out_blocks if out_blocks <= WGRAD_CHUNKS else 1))
    cap = max(1, (64 << 20) // (4 * ncols * kcols))
    return max(1, min(chunks, cap))


# Weight-gradient partials whose fixed-order sum rides along with the next BatchNorm-backward finalize launch of the same
# backward call (rs_bn_backward_finalize_reduce): (partial, chunks, elements, dw).  Same stream, consumed in order;
# `flush_reduces` sums what is left when the chain ends.
class _PendingReduces:
    """One queue per (device, stream): a weight gradient's partials are summed by a later launch on the SAME stream (that is what
    orders them), so work queued on one device / stream must never ride with, or be drained by, a launch on another (ADVICE r3:
    the single process-wide list was drained on whatever stream was current).  len() / bool(): over all queues."""

    def __init__(self):
        self.queues = {}

    def cur(self):
        key = (torch.cuda.current_device(), _stream())
        q = self.queues.get(key)
        if q is None:
            q = self.queues[key] = []
        return q

    def append(self, item):
        self.cur().append(item)

    def __len__(self):
        return sum(len(q) for q in self.queues.values())

    def __bool__(self):
        return any(self.queues.values())


_pending_reduce = _PendingReduces()


def wgrad(rows, ncols, kcols, p_op, q_op, device, rows_dev=None, defer=False):
    """defer=True: dw is complete only after the next bwd_coeffs() / flush_reduces() on this stream."""
    # a compacted row set fills a fraction of its capacity (the count is on the device): size the slab split
    # for a quarter of it so that slabs keep several pipeline stages and fewer partials need reducing
    chunks = wgrad_chunks(rows if rows_dev is None else max(rows // 4, 256), ncols, kcols)
    if rows_dev is None:
        rows_dev = _ragged.dev(rows)      # (a capacity that is nearly full: slabs sized for all of it)
    part = torch.empty((chunks, ncols * kcols), dtype=torch.float32, device=device)
    dw = torch.empty((ncols, kcols), dtype=torch.float32, device=device)
    from . import mlp as _mlp
    defer = defer and not SIDE_WGRAD and os.environ.get("REPSURF_DEFER_REDUCE", "1") != "0"
    _lib.call("rs_mlp_wgrad_bf16" if _mlp.PRECISION == "bf16" else "rs_mlp_wgrad", rows, rows_dev, ncols, kcols, ctypes.byref(p_op), ctypes.byref(q_op), _ptr(part), chunks,
              None if defer else _ptr(dw), _stream())
    if defer:
        _pending_reduce.append((part, chunks, ncols * kcols, dw))      # the queue of THIS device and stream
    return dw


def _tail(fin_items, max_red=TAIL_RED_MAX):
    """One rs_backward_tail launch: the given BatchNorm-backward finalizes plus up to `max_red` pending reductions."""
    work = BackwardTail()
    work.nfin = len(fin_items)
    for i, it in enumerate(fin_items):
        work.fin[i] = it
    queue = _pending_reduce.cur()
    nred = min(len(queue), max_red)
    keep = [queue.pop(0) for _ in range(nred)]
    work.nred = nred
    for j, (part, chunks, n, dw) in enumerate(keep):
        work.red[j] = ReduceItem(chunks=chunks, n=n, partial=_ptr(part), out=_ptr(dw))
    if work.nfin or work.nred:
        _lib.call("rs_backward_tail", ctypes.byref(work), _stream())


def flush_reduces(everywhere=False):
    """Sum what is pending on the current (device, stream); everywhere=True: on every queue, each under its own device and
    stream (the end-of-pass callback runs on the thread that called backward, whatever streams the pass used)."""
    while _pending_reduce.cur():
        _tail([])
    if everywhere:
        for (dev, stream), queue in list(_pending_reduce.queues.items()):
            if queue:
                with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    while queue:
                        _tail([])


# A stack's backward ends with weight-gradient partials whose sum is still pending (its first-layer weights).  Summing them
# there costs one more small launch per stack; left pending, they ride along with the NEXT stack's first finalize launch.
# That is only sound when nothing reads a gradient before the backward pass is over -- no `.grad` to accumulate into, no
# gradient hooks (DistributedDataParallel reads gradients as they appear) -- so it is opt-in: the graphed training steps of
# repsurf_amd.graph own their pass (gradients start as None, consumed by the optimizer / the flat-buffer pack after
# backward) and wrap it in `owned_pass()`; whatever is still pending when the autograd engine finishes is summed by an
# end-of-pass callback.
OWNED_PASS = 0
_flush_armed = False


class owned_pass:
    """Contract of the caller (the graphed steps of repsurf_amd.graph keep it): every parameter's `.grad` is None when
    backward starts, no parameter carries gradient hooks / post-accumulate hooks, no `create_graph`, and nothing reads a
    gradient before backward returns -- a weight gradient handed to autograd is COMPLETE only once the pass has ended (its
    fixed-order sum rides with a later launch on the same stream).  `check(params)` verifies the first two."""

    @staticmethod
    def check(params):
        for p in params:
            if p.grad is not None:
                raise RuntimeError("owned_pass: a parameter already holds a .grad (autograd would read the incomplete sum to accumulate)")
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
                raise RuntimeError("owned_pass: a parameter carries gradient hooks (they would see an incomplete weight gradient)")

    def __enter__(self):
        global OWNED_PASS
        OWNED_PASS += 1
        return self

    def __exit__(self, *exc):
        global OWNED_PASS
        OWNED_PASS -= 1
        flush_reduces()            # (nothing after a completed backward; an aborted one must not leak into the next pass)


def _end_of_pass_flush():
    global _flush_armed
    _flush_armed = False
    flush_reduces(everywhere=True)


def _stack_begins():
    if not OWNED_PASS:
        flush_reduces()      # (nothing, unless an earlier backward call was interrupted between a wgrad and its reduction)


def _stack_ends():
    global _flush_armed
    if not _pending_reduce.cur():
        return
    if OWNED_PASS:
        if not _flush_armed:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass_flush)
                _flush_armed = True
            except RuntimeError:       # not inside a backward pass
                flush_reduces()
        return
    flush_reduces()


def bwd_coeffs_multi(specs, device):
    """specs: [(c, rows, part, nstat, which, vec, nblk | None, frozen)] (at most TAIL_FIN_MAX) -> [(p, q, r, dgamma, dbeta)] from ONE
    launch, which also sums the pending weight-gradient partials (up to TAIL_RED_MAX of them).
    frozen (eval mode: the layer normalised with its running statistics, which are constants): dy = scale * dz, i.e.
    p = scale, q = r = 0; dgamma / dbeta are the same sums (vec.mean / vec.invstd hold the running statistics)."""
    items, outs, synced = [], [], []
    reduced = {}
    for c, rows, part, nstat, which, vec, nblk, frozen in specs:
        if vec.sync is not None and not frozen:          # SyncBatchNorm: the backward sums span the ranks too (once per tensor)
            if part.data_ptr() not in reduced:
                reduced[part.data_ptr()] = sync_partials(part if nblk is None else part[:nblk], vec.sync, rows)
            rows = rows * reduced[part.data_ptr()]
        buf = torch.empty((5, c), dtype=torch.float32, device=device)
        synced.append(reduced.get(part.data_ptr(), 1) if (vec.sync is not None and not frozen) else 1)
        items.append(BnBwdItem(c=c, nblk=part.shape[0] if nblk is None else nblk, nstat=nstat, which=which, rows=rows, partial=part.data_ptr(),
                               scale=_ptr(vec.scale), mean=_ptr(vec.mean), invstd=_ptr(vec.invstd), p=_ptr(buf[0]), q=_ptr(buf[1]), r=_ptr(buf[2]),
                               dgamma=_ptr(buf[3]), dbeta=_ptr(buf[4]), rows_dev=_ragged.dev(rows)))
        outs.append((buf, vec, frozen, c))
    _tail(items)
    for (buf, _, _, _), world in zip(outs, synced):
        if world > 1:
            # the finalize saw the sums over ALL ranks: right for p, q, r (the data gradient of synchronized statistics), but
            # dgamma / dbeta are then the global sums on every rank, and the gradient all-reduce AVERAGES over the ranks: scale
            # them so that the average is the whole batch's gradient (torch's SyncBatchNorm hands out the local sums instead;
            # after the averaging both give the same parameters)
            buf[3:5].mul_(1.0 / world)
    res = []
    for buf, vec, frozen, c in outs:
        if frozen:
            zero = torch.zeros((2, c), dtype=torch.float32, device=device)
            res.append((vec.scale, zero[0], zero[1], buf[3], buf[4]))
        else:
            res.append((buf[0], buf[1], buf[2], buf[3], buf[4]))
    return res


def bwd_coeffs(c, rows, part, nstat, which, vec, device, nblk=None, frozen=False):
    """BN backward sums (`nblk` partial rows, default: all rows of `part`) -> (p, q, r, dgamma, dbeta); see bwd_coeffs_multi."""
    return bwd_coeffs_multi([(c, rows, part, nstat, which, vec, nblk, frozen)], device)[0]


def dgrad_masked(rows, kdim, cols, p_op, w2d, y1, v1, y2=None, v2=None, device=None, rows_dev=None, wt=None):
    """dz_prev = (P . W) * relu'(z_prev) and the BN-backward sums of the previous layer(s)."""
    dz = torch.empty((rows, cols), dtype=torch.float32, device=device)
    nstat = 3 if y2 is not None else 2
    nblk = partial_rows(rows)
    part = torch.empty((nblk, nstat, cols), dtype=torch.float64, device=device)
    epi = Epilogue(bias=None, out=_ptr(dz), ldo=cols, mode=EPI_MASK,
                   my1=_ptr(y1), ldm1=cols, ms1=_ptr(v1.scale), mt1=_ptr(v1.shift), mean1=_ptr(v1.mean), invstd1=_ptr(v1.invstd),
                   partial=part.data_ptr(), partial_blocks=nblk, my1_bf16=_bf(y1), my2_bf16=_bf(y2))
    if y2 is not None:
        epi.my2, epi.ldm2 = _ptr(y2), cols
        epi.ms2, epi.mt2, epi.mean2, epi.invstd2 = _ptr(v2.scale), _ptr(v2.shift), _ptr(v2.mean), _ptr(v2.invstd)
    gemm_rows(rows, kdim, cols, p_op, wt if wt is not None else w_bwd(w2d), epi, rows_dev)      # dY . W: the transposed copy
    return dz, part, nstat


# ------------------------------------------------------------------------------------------- SA stacks
# Weight-gradient GEMMs hang off the backward chain (nothing downstream in the chain needs them), so they CAN go to a
# second HIP stream and overlap with the data-gradient GEMMs.  Measured (hipGraph replay, B=32): 2.67 ms/step with the
# fork against 2.64 ms without -- both kernel families are limited by the same per-CU load/store paths and matrix
# pipe, so running them side by side only makes each slower.  Off by default; REPSURF_WGRAD_STREAM=1 enables it.
SIDE_WGRAD = os.environ.get("REPSURF_WGRAD_STREAM", "0") != "0"
_side_streams = {}


class _Fork:
    """Fork/join of side-stream launches inside one backward call.  Temporaries read by the side stream are kept
    alive until the join (a freed block could otherwise be handed to a later main-stream allocation while the
    side stream still reads it)."""

    def __init__(self, device):
        self.on = SIDE_WGRAD
        self.keep = []
        self.used = False
        if self.on:
            self.main = torch.cuda.current_stream(device)
            key = (device.index, self.main.cuda_stream)
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device=device)
            self.side = _side_streams[key]

    def run(self, fn, *alive):
        self.keep.extend(alive)
        if not self.on:
            return fn()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            out = fn()
        self.used = True
        return out

    def join(self):
        if self.on and self.used:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)
        self.keep.clear()


class _SAStack(Function):
    """[two-branch | single] first layer -> [conv, BN, ReLU]* -> max over nsample.
    args: x (rows, cx), meta, then flat parameters (see sa_mlp_cd / sa_mlp_plain)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        dev = x.device
        x = x.contiguous()
        rows, cx = x.shape
        ns, pos, bns, training = meta["nsample"], meta["pos"], meta["bns"], meta["training"]
        rs = RowSet(rows, meta.get("compact"))
        groups = rs.full // ns
        saved = {"x": x, "rs": rs}
        pi = 0
        ys, vecs, w2ds = [], [], []
        all_w2d = [_w2d(params[i]) for i in range(0, len(params), 4)]
        wks = fwd_weights(all_w2d, dev)            # conv weights in place; one batched padded copy for odd cin
        sb = stores_bf16([w.shape[0] for w in all_w2d])
        foff, fk = meta.get("feat_off", pos), meta.get("feat_k", cx - pos)   # feature branch: columns [foff, foff + fk)
        if pos > 0:      # two-branch first layer (SurfaceAbstractionCD)
            wl, bl, wf, bf = params[0], params[1], params[4], params[5]
            wl2, wf2 = all_w2d[0], all_w2d[1]
            if training:       # both GEMMs, then the statistics of both BatchNorms in one launch
                yl, vl, fin_l = fwd_layer(rows, operand(OP_ID, x, cx), pos, wl2, bl, bns[0], training, dev, rs=rs, wk=wks[0], store_bf16=sb,
                                          finalize=False)
                yf, vf, fin_f = fwd_layer(rows, operand(OP_ID, x, cx, a_off=foff), fk, wf2, bf, bns[1], training, dev, rs=rs,
                                          wk=wks[1], store_bf16=sb, finalize=False)
                bn_finalize_batch([fin_l, fin_f])
            else:
                yl, vl = fwd_layer(rows, operand(OP_ID, x, cx), pos, wl2, bl, bns[0], training, dev, rs=rs, wk=wks[0], store_bf16=sb)
                yf, vf = fwd_layer(rows, operand(OP_ID, x, cx, a_off=foff), fk, wf2, bf, bns[1], training, dev, rs=rs,
                                   wk=wks[1], store_bf16=sb)
            saved.update(yl=yl, vl=vl, yf=yf, vf=vf, wl2=wl2, wf2=wf2)
            prev_op = operand(OP_RELU2, yl, yl.shape[1], yf, yf.shape[1], vl.scale, vl.shift, vf.scale, vf.shift)
            prev_c = wl2.shape[0]
            pi, bi = 8, 2
        else:
            lazy_in = meta.get("lazy_in")
            prev_op = operand(OP_ID, x, cx) if lazy_in is None else operand(OP_RELU1, x, cx, s1=lazy_in.vec.scale, t1=lazy_in.vec.shift)
            prev_c = cx
            pi, bi = 0, 0
        pooled = None
        while pi < len(params):
            w, b = params[pi], params[pi + 1]
            w2, wk = all_w2d[pi // 4], wks[pi // 4]
            last = pi + 4 >= len(params)
            if last and training and rs.dev is None and ns > 1 and meta.get("relu_last", True) and fused_pool_ok(w2.shape[0], ns):      # (ns = 1: a plain row stack, nothing to pool)
                y, vec, pooled = fwd_layer(rows, prev_op, prev_c, w2, b, bns[bi], training, dev, pool_ns=ns, wk=wk, store_bf16=sb)
            else:
                y, vec = fwd_layer(rows, prev_op, prev_c, w2, b, bns[bi], training, dev, rs=rs, wk=wk, store_bf16=sb)
            ys.append(y); vecs.append(vec); w2ds.append(w2)
            prev_op = operand(OP_RELU1, y, y.shape[1], s1=vec.scale, t1=vec.shift)
            prev_c = w2.shape[0]
            pi += 4; bi += 1
        if meta.get("lazy_out") is not None:      # the consumer applies this layer's BatchNorm + ReLU in its operand prologue (LazyRows)
            assert ys and ns == 1 and rs.dev is None
            out, arg = ys[-1], None
            meta["vec_last"] = vecs[-1]
        elif pooled is not None:
            out, arg = pooled
        elif ys:
            y_last, v_last = ys[-1], vecs[-1]
            out = torch.empty((groups, prev_c), dtype=torch.float32, device=dev)
            # groups of ONE row (row stacks of the segmentation decoder): nothing is selected, no index tensor is written or read
            arg = None if (ns == 1 and rs.dev is None) else torch.empty((groups, prev_c), dtype=torch.int32, device=dev)
            _lib.call("rs_pool_max", groups, ns, prev_c, int(meta.get("relu_last", True)), _ptr(rs.offsets), _ptr(y_last), _bf(y_last), _ptr(v_last.scale),
                      _ptr(v_last.shift), _ptr(out), None if arg is None else arg.data_ptr(), _stream())
        else:
            raise NotImplementedError("a stack needs at least one layer after the first")
        # `out` becomes an output of this node: a plain reference from ctx would close a cycle (node -> saved -> out -> grad_fn = node)
        # that only Python's generation-2 collector breaks -- an eager loop then keeps ~30 steps of activations alive (1.5 GB per
        # classification step).  Detached aliases share the storage and carry no edge.
        saved.update(ys=[y.detach() if y is out else y for y in ys], vecs=vecs, w2ds=w2ds, out=out.detach(), arg=arg)
        _flush_counters()
        ctx.saved = saved
        ctx.meta = meta
        ctx.nparams = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        frozen = not meta["training"]          # eval mode: BatchNorm is a constant affine (running statistics)
        x = s["x"]
        dev = x.device
        rows, cx = x.shape
        ns, pos = meta["nsample"], meta["pos"]
        rs = s["rs"]
        groups, full, rdev = rs.full // ns, rs.full, rs.dev
        ys, vecs, w2ds = s["ys"], s["vecs"], s["w2ds"]
        if dout.dim() != 2 or dout.stride(1) != 1 or dout.stride(0) < dout.shape[1]:      # (a column slice of a wider tensor is read in place)
            dout = dout.contiguous()
        grads = [None] * ctx.nparams
        nl = len(ys)
        _stack_begins()      # (nothing, unless an earlier backward call was interrupted between a wgrad and its reduction)
        first = 8 if pos > 0 else 0
        # ---- pooled layer: BN-backward sums from (groups, c) data only
        c_last = ys[-1].shape[1]
        lazy_out = meta.get("lazy_out")
        if lazy_out is not None:
            # the consumer's data-gradient GEMM already masked the gradient with this layer's ReLU and summed its BatchNorm-backward
            # moments (LazyRows): no pass over (rows, C) here
            if lazy_out.part is None:
                raise RuntimeError("LazyRows: the consumer's backward did not run before the producer's (a lazy activation has exactly one consumer)")
            v = dout.contiguous()
            part, nstat_last = lazy_out.part
            lazy_out.part = None
            p, q, r, dg, db = bwd_coeffs(c_last, full, part, nstat_last, 1, vecs[-1], dev, frozen=frozen)
        else:
            v = torch.empty(dout.shape, dtype=torch.float32, device=dev)
            pool_blk = partial_rows(groups, 4)
            part = torch.empty((pool_blk, 2, c_last), dtype=torch.float64, device=dev)
            _lib.call("rs_pool_max_backward", groups, ns, c_last, _ptr(rs.offsets), _ptr(dout), dout.stride(0), _ptr(s["out"]) if meta.get("relu_last", True) else None,
                      None if s["arg"] is None else s["arg"].data_ptr(), _ptr(ys[-1]), _bf(ys[-1]), _ptr(vecs[-1].mean), _ptr(vecs[-1].invstd), _ptr(v), part.data_ptr(),
                      pool_blk, _ragged.dev(groups), _stream())
            p, q, r, dg, db = bwd_coeffs(c_last, full, part, 2, 1, vecs[-1], dev, frozen=frozen)
        if s["arg"] is None:      # one-row groups: the pooled-gradient operand IS the two-tensor BatchNorm-backward affine (no index compare)
            p_op = operand(OP_AFF2, v, c_last, ys[-1], c_last, s1=p, t1=r, s2=q, rs=rs)
        else:
            p_op = operand(OP_POOLED, v, c_last, ys[-1], c_last, s1=p, t1=r, s2=q, arg=s["arg"], ns=ns, rs=rs)
        dx = None
        fork = _Fork(dev)
        fork.keep += [v, p, q, r]
        # transposed weight copies for the data-gradient GEMMs (dY . W), one launch: layers 1.. of the chain, plus
        # the feature branch of a two-branch first layer / the single first layer when the input needs a gradient
        need_t = {("l", li): w2ds[li] for li in range(1, nl)}
        if nl and (pos > 0 or ctx.needs_input_grad[0]):
            need_t[("l", 0)] = w2ds[0]
        if pos > 0 and ctx.needs_input_grad[0]:
            need_t[("f", 0)] = s["wf2"]
        wts = dict(zip(need_t.keys(), pack_weights(list(need_t.values()), True, dev))) if need_t else {}
        for li in range(nl - 1, -1, -1):
            pidx = first + 4 * li
            cout, cin = w2ds[li].shape
            grads[pidx + 2], grads[pidx + 3] = dg, db
            # bias before BN: exactly 0 with batch statistics; through frozen statistics it is scale * sum(dz)
            grads[pidx + 1] = vecs[li].scale * db if frozen else _zeros.take(cout, dev)
            # the activation that fed this layer, rebuilt on the fly from the stored conv outputs
            if li > 0:
                q_op = operand(OP_RELU1, ys[li - 1], cin, s1=vecs[li - 1].scale, t1=vecs[li - 1].shift)
            elif pos > 0:
                q_op = operand(OP_RELU2, s["yl"], cin, s["yf"], cin, s["vl"].scale, s["vl"].shift,
                               s["vf"].scale, s["vf"].shift)
            else:
                lazy_in = meta.get("lazy_in")
                q_op = operand(OP_ID, x, cx) if lazy_in is None else operand(OP_RELU1, x, cx, s1=lazy_in.vec.scale, t1=lazy_in.vec.shift)
            grads[pidx] = fork.run(lambda: wgrad(rows, cout, cin, p_op, q_op, dev, rdev, defer=True))
            if li > 0:      # data gradient, ReLU mask and BN-backward sums of layer li-1
                dz, part, nstat = dgrad_masked(rows, cout, cin, p_op, w2ds[li], ys[li - 1], vecs[li - 1], device=dev,
                                               rows_dev=rdev, wt=wts[("l", li)])
                p, q, r, dg, db = bwd_coeffs(cin, full, part, nstat, 1, vecs[li - 1], dev, frozen=frozen)
                if DEBUG is not None:
                    DEBUG["layer%d" % li] = dict(dz=dz, part=part, p=p, q=q, r=r, dg=dg, db=db, y=ys[li - 1], vec=vecs[li - 1])
                    if "log" in DEBUG:
                        DEBUG["log"].append(dict(li=li, rows=rows, full=full, dz=dz, part=part, p=p, q=q, r=r, dg=dg, db=db, y=ys[li - 1]))
                p_op = operand(OP_AFF2, dz, cin, ys[li - 1], cin, s1=p, t1=r, s2=q, rs=rs)
                fork.keep += [dz, p, q, r]
            elif pos > 0:   # two-branch first layer: one masked gradient, two BatchNorms
                dz, part, nstat = dgrad_masked(rows, cout, cin, p_op, w2ds[li], s["yl"], s["vl"], s["yf"], s["vf"],
                                               device=dev, rows_dev=rdev, wt=wts[("l", 0)])
                (pl, ql, rl, dgl, dbl), (pf, qf, rf, dgf, dbf) = bwd_coeffs_multi(       # both BatchNorms (+ pending reductions): one launch
                    [(cin, full, part, 3, 1, s["vl"], None, frozen), (cin, full, part, 3, 2, s["vf"], None, frozen)], dev)
                if DEBUG is not None:
                    DEBUG.update(dz0=dz, part0=part, pl=pl, ql=ql, rl=rl, pf=pf, qf=qf, rf=rf, dgl=dgl, dbl=dbl,
                                 dgf=dgf, dbf=dbf, yl=s["yl"], yf=s["yf"], vl=s["vl"], vf=s["vf"])
                opl = operand(OP_AFF2, dz, cin, s["yl"], cin, s1=pl, t1=rl, s2=ql, rs=rs)
                opf = operand(OP_AFF2, dz, cin, s["yf"], cin, s1=pf, t1=rf, s2=qf, rs=rs)
                fork.keep += [dz, pl, ql, rl, pf, qf, rf]
                foff, fk = meta.get("feat_off", pos), meta.get("feat_k", cx - pos)
                grads[0] = fork.run(lambda: wgrad(rows, cin, pos, opl, operand(OP_ID, x, cx), dev, rdev, defer=True))
                grads[4] = fork.run(lambda: wgrad(rows, cin, fk, opf, operand(OP_ID, x, cx, a_off=foff), dev, rdev, defer=True))
                grads[1] = s["vl"].scale * dbl if frozen else _zeros.take(cin, dev)
                grads[5] = s["vf"].scale * dbf if frozen else _zeros.take(cin, dev)
                grads[2], grads[3], grads[6], grads[7] = dgl, dbl, dgf, dbf
                if ctx.needs_input_grad[0]:
                    # Only the feature channels [pos:] carry a gradient (coordinates are inputs); the grouping
                    # backward reads nothing else, so the position columns are left unwritten (no 150 MB memset).
                    dx = torch.empty((rows, cx), dtype=torch.float32, device=dev)
                    epi = Epilogue(bias=None, out=_ptr(dx, foff), ldo=cx, mode=EPI_STORE)
                    gemm_rows(rows, cin, fk, opf, wts[("f", 0)], epi, rdev)
            elif ctx.needs_input_grad[0]:
                lazy_in = meta.get("lazy_in")
                if lazy_in is not None:      # the gradient of the producer's raw output: masked by its ReLU, its BatchNorm-backward moments beside it
                    dx, part_in, nstat_in = dgrad_masked(rows, cout, cx, p_op, w2ds[0], x, lazy_in.vec, device=dev, rows_dev=rdev, wt=wts[("l", 0)])
                    lazy_in.part = (part_in, nstat_in)
                else:
                    dx = torch.empty((rows, cx), dtype=torch.float32, device=dev)
                    epi = Epilogue(bias=None, out=_ptr(dx), ldo=cx, mode=EPI_STORE)
                    gemm_rows(rows, cout, cx, p_op, wts[("l", 0)], epi, rdev)
        _stack_ends()
        fork.join()
        out_grads = [None if g is None else g.reshape(shape) for g, shape in zip(grads, meta["shapes"])]
        return (dx, None) + tuple(out_grads)


def _flat_params(first, convs, bns):
    params, mods = [], []
    for conv, bn in first + list(zip(convs, bns)):
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
        mods.append(bn)
    return params, mods


def sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=None, feat_off=None, feat_k=None):
    """compact: ops.CompactGroups whose .x is `x` (duplicate ball-query slots removed) or None (dense rows).
    feat_off / feat_k: columns of the feature branch when x uses the aligned (padded) row layout."""
    params, mods = _flat_params([(mlp_l0, bn_l0), (mlp_f0, bn_f0)], convs, bns)
    meta = {"nsample": nsample, "pos": pos_channel, "bns": mods, "training": mods[0].training,
            "shapes": [p.shape for p in params]}
    if feat_off is not None:
        meta["feat_off"], meta["feat_k"] = feat_off, feat_k
    if compact is not None:
        meta["compact"] = {"rows_dev": compact.rows_dev_ptr, "rows_full": compact.rows_full, "mult": compact.mult,
                           "grp": compact.grp, "slot": compact.slot, "offsets": compact.offsets}
    return _SAStack.apply(x, meta, *params)


class LazyRows:
    """relu(BatchNorm(y)) of a row stack's last layer, NOT materialised (round 4): `y` is the layer's raw output (rows, C) -- the
    autograd edge --, `vec` its BatchNorm's (scale, shift, mean, invstd).  The consumer must be a node that applies the affine map
    and the ReLU in its operand prologue (_SAStack with lazy_in, _FPFront): forward, its first GEMM reads y through RS_OP_RELU1;
    backward, its data-gradient GEMM masks with relu'(.) and sums the BatchNorm-backward moments in its epilogue -- what the layers
    INSIDE a stack do for each other -- and leaves them here (`part`) for the producer's backward, which receives the masked
    gradient through autograd.  Saves the BatchNorm + ReLU pass over (rows, C) each way.  One consumer only."""
    __slots__ = ("y", "vec", "box")

    class Box:
        """The one field producer and consumer exchange in backward.  The producer's node keeps THIS, not the LazyRows: `y` is that
        node's own output, and holding it from the node would be a reference cycle (freed by the cycle collector only)."""
        __slots__ = ("part",)

        def __init__(self):
            self.part = None

    def __init__(self):
        self.y = self.vec = None
        self.box = LazyRows.Box()

    @property
    def part(self):
        return self.box.part

    @part.setter
    def part(self, v):
        self.box.part = v

    @property
    def shape(self):
        return self.y.shape

    def detach(self):
        """The activated rows, for inspection (forward hooks, tests): relu(scale * y + shift), detached.  Not the product path."""
        return torch.relu(torch.addcmul(self.vec.shift, self.y.detach(), self.vec.scale))


LAZY_ROWS = os.environ.get("REPSURF_LAZY_ROWS", "1") != "0"


def lazy_rows_usable(bn_mods):
    """row stacks can hand their last activation over unmaterialised: training mode, batch statistics, fp32 arithmetic"""
    from . import mlp as _mlp
    return LAZY_ROWS and _mlp.PRECISION == "fp32" and all(b.training and sync_of(b) is None for b in bn_mods)


def sa_mlp_plain(x, convs, bns, nsample, relu_last=True, lazy_out=False):
    """relu_last=False: the last layer ends at its BatchNorm (segmentation feature propagation, first layers).
    x: a tensor or a LazyRows (nsample = 1); lazy_out: return a LazyRows instead of the activated tensor (nsample = 1, training)."""
    params, mods = _flat_params([], convs, bns)
    meta = {"nsample": nsample, "pos": 0, "bns": mods, "training": mods[0].training,
            "shapes": [p.shape for p in params], "relu_last": relu_last}
    if isinstance(x, LazyRows):
        assert mods[0].training, "a LazyRows input needs training-mode consumers"
        meta["lazy_in"] = x
        x = x.y
    if lazy_out:
        assert nsample == 1 and relu_last and mods[0].training, "lazy_out: ungrouped rows ending in BatchNorm + ReLU, training mode"
        lazy = LazyRows()
        meta["lazy_out"] = lazy.box
        lazy.y = _SAStack.apply(x, meta, *params)
        lazy.vec = meta["vec_last"]
        return lazy
    return _SAStack.apply(x, meta, *params)


# ------------------------------------------------------------------------------------------- umbrella stack
class _UmbrellaStack(Function):
    """conv(no bias)-BN-ReLU-conv-BN-ReLU-conv, then sum | avg | max over the `group` fan triangles.
    The input (geometric features) never needs a gradient."""

    @staticmethod
    def forward(ctx, x, meta, w0, g0, b0, w1, c1, g1, b1, w2, c2):
        dev = x.device
        x = x.contiguous()
        rows, cx = x.shape
        group, aggr, training = meta["group"], meta["aggr"], meta["training"]
        bn0, bn1 = meta["bns"]
        w0_, w1_, w2_ = _w2d(w0), _w2d(w1), _w2d(w2)
        y0, v0 = fwd_layer(rows, operand(OP_ID, x, cx), cx, w0_, None, bn0, training, dev)
        y1, v1 = fwd_layer(rows, operand(OP_RELU1, y0, y0.shape[1], s1=v0.scale, t1=v0.shift), w0_.shape[0], w1_, c1,
                           bn1, training, dev)
        cout = w2_.shape[0]
        y2 = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        epi = Epilogue(bias=_ptr(c2), out=_ptr(y2), ldo=cout, mode=EPI_STORE)
        gemm_rows(rows, w1_.shape[0], cout, operand(OP_RELU1, y1, y1.shape[1], s1=v1.scale, t1=v1.shift), w_fwd(w2_), epi)
        points = rows // group
        out = torch.empty((points, cout), dtype=torch.float32, device=dev)
        arg = None
        if aggr == "max":
            arg = torch.empty((points, cout), dtype=torch.int32, device=dev)
            _lib.call("rs_pool_max", points, group, cout, 0, None, _ptr(y2), 0, None, None, _ptr(out), arg.data_ptr(), _stream())
        else:
            _lib.call("rs_pool_sum", points, group, cout, _ptr(y2), _ptr(out), _stream())
            if aggr == "avg":
                out.mul_(1.0 / group)
        ctx.saved = dict(x=x, y0=y0, v0=v0, y1=y1, v1=v1, y2=y2, w0=w0_, w1=w1_, w2=w2_, arg=arg)
        ctx.meta = meta
        _flush_counters()
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        frozen = not meta["training"]          # eval mode: BatchNorm is a constant affine (running statistics)
        x, y0, v0, y1, v1 = s["x"], s["y0"], s["v0"], s["y1"], s["v1"]
        dev = x.device
        rows, cx = x.shape
        group, aggr = meta["group"], meta["aggr"]
        dout = dout.contiguous()
        if aggr == "avg":
            dout = dout * (1.0 / group)
        c2n, c1n, c0n = s["w2"].shape[0], s["w1"].shape[0], s["w0"].shape[0]
        if aggr == "max":
            one = torch.ones(c2n, dtype=torch.float32, device=dev)
            zero = torch.zeros(c2n, dtype=torch.float32, device=dev)
            p2 = operand(OP_POOLED, dout, c2n, s["y2"], c2n, s1=one, t1=zero, s2=zero, arg=s["arg"], ns=group)
            g_c2 = dout.sum(0)
        else:
            p2 = operand(OP_BCAST, dout, c2n, ns=group)
            g_c2 = dout.sum(0) * group
        g_w2 = wgrad(rows, c2n, c1n, p2, operand(OP_RELU1, y1, c1n, s1=v1.scale, t1=v1.shift), dev)
        dz1, part, nstat = dgrad_masked(rows, c2n, c1n, p2, s["w2"], y1, v1, device=dev)
        pa, qa, ra, g_g1, g_b1 = bwd_coeffs(c1n, rows, part, nstat, 1, v1, dev, frozen=frozen)
        p1 = operand(OP_AFF2, dz1, c1n, y1, c1n, s1=pa, t1=ra, s2=qa)
        g_w1 = wgrad(rows, c1n, c0n, p1, operand(OP_RELU1, y0, c0n, s1=v0.scale, t1=v0.shift), dev)
        dz0, part0, nstat0 = dgrad_masked(rows, c1n, c0n, p1, s["w1"], y0, v0, device=dev)
        pb, qb, rb, g_g0, g_b0 = bwd_coeffs(c0n, rows, part0, nstat0, 1, v0, dev, frozen=frozen)
        p0 = operand(OP_AFF2, dz0, c0n, y0, c0n, s1=pb, t1=rb, s2=qb)
        g_w0 = wgrad(rows, c0n, cx, p0, operand(OP_ID, x, cx), dev)
        shp = meta["shapes"]
        # bias before BN: exactly 0 with batch statistics; scale * sum(dz) through frozen ones
        g_c1 = v1.scale * g_b1 if frozen else _zeros.take(c1n, dev)
        return (None, None, g_w0.reshape(shp[0]), g_g0, g_b0, g_w1.reshape(shp[1]), g_c1, g_g1, g_b1,
                g_w2.reshape(shp[2]), g_c2)


class UmbrellaMLPDesc(ctypes.Structure):      # rs_umbrella_mlp
    _fields_ = [("x", P), ("rows", c_ll), ("group", c_int), ("w0", P), ("w1", P), ("b1", P), ("w2", P), ("b2", P),
                ("bn0", P), ("bn1", P), ("c0", P), ("c1", P), ("dout", P), ("b0", P), ("layers", c_int)]


UMB_BLOCKS = 512
# backward passes reduce 110 weight-gradient sums per workgroup: fewer, longer workgroups amortise that reduction
UMB_BLOCKS_BWD = int(os.environ.get("REPSURF_UMB_BLOCKS_BWD", "256"))


class _UmbrellaFused(Function):
    """The 10-channel constructor MLP as six register-resident passes (csrc/umbrella_mlp.hip)."""

    @staticmethod
    def forward(ctx, x, meta, w0, g0, b0, w1, c1, g1, b1, w2, c2):
        dev = x.device
        x = x.contiguous()
        rows = x.shape[0]
        group, aggr = meta["group"], meta["aggr"]
        bn0, bn1 = meta["bns"]
        w0_, w1_, w2_ = _w2d(w0), _w2d(w1), _w2d(w2)
        c1_, c2_ = c1.detach().contiguous(), c2.detach().contiguous()
        v0, v1 = BNVec(10, dev), BNVec(10, dev)
        desc = UmbrellaMLPDesc(x=_ptr(x), rows=rows, group=group, w0=_ptr(w0_), w1=_ptr(w1_), b1=_ptr(c1_), w2=_ptr(w2_),
                               b2=_ptr(c2_), bn0=_ptr(v0.scale), bn1=_ptr(v1.scale))
        part = torch.empty((UMB_BLOCKS, 2, 10), dtype=torch.float64, device=dev)
        for pas, bn_mod, vec in ((0, bn0, v0), (1, bn1, v1)):
            _lib.call("rs_umbrella_mlp_pass", pas, ctypes.byref(desc), 1.0, None, part.data_ptr(), None, UMB_BLOCKS, _stream())
            track = bn_mod.track_running_stats and bn_mod.running_mean is not None
            if track:
                _pending_counters.append(bn_mod.num_batches_tracked)
            mom = bn_mod.momentum if bn_mod.momentum is not None else 0.1
            _lib.call("rs_bn_finalize", 10, rows, UMB_BLOCKS, part.data_ptr(), _ptr(bn_mod.weight), _ptr(bn_mod.bias),
                      float(bn_mod.eps), float(mom), _ptr(vec.scale), _ptr(vec.shift), _ptr(vec.mean), _ptr(vec.invstd),
                      _ptr(bn_mod.running_mean) if track else None, _ptr(bn_mod.running_var) if track else None, _stream())
        out = torch.empty((rows // group, 10), dtype=torch.float32, device=dev)
        scale = 1.0 / group if aggr == "avg" else 1.0
        _lib.call("rs_umbrella_mlp_pass", 2, ctypes.byref(desc), scale, _ptr(out), None, None, UMB_BLOCKS, _stream())
        ctx.saved = dict(x=x, w0=w0_, w1=w1_, w2=w2_, c1=c1_, c2=c2_, v0=v0, v1=v1)
        ctx.meta = meta
        _flush_counters()
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        x, v0, v1 = s["x"], s["v0"], s["v1"]
        dev = x.device
        rows, group = x.shape[0], meta["group"]
        dout = dout.contiguous()
        if meta["aggr"] == "avg":
            dout = dout * (1.0 / group)
        desc = UmbrellaMLPDesc(x=_ptr(x), rows=rows, group=group, w0=_ptr(s["w0"]), w1=_ptr(s["w1"]), b1=_ptr(s["c1"]),
                               w2=_ptr(s["w2"]), b2=_ptr(s["c2"]), bn0=_ptr(v0.scale), bn1=_ptr(v1.scale), dout=_ptr(dout))
        part = torch.empty((UMB_BLOCKS_BWD, 2, 10), dtype=torch.float64, device=dev)
        dwp = torch.empty((3, UMB_BLOCKS_BWD, 110), dtype=torch.float32, device=dev)
        res = torch.empty((3, 110), dtype=torch.float32, device=dev)

        def run(pas, slot):      # the fixed-order sum of the weight-gradient partials rides along with the next finalize launch
            _lib.call("rs_umbrella_mlp_pass", pas, ctypes.byref(desc), 1.0, None, part.data_ptr(), _ptr(dwp[slot]), UMB_BLOCKS_BWD, _stream())
            _pending_reduce.append((dwp[slot], UMB_BLOCKS_BWD, 110, res[slot]))

        run(3, 2)
        p1, q1, r1, g_g1, g_b1 = bwd_coeffs(10, rows, part, 2, 1, v1, dev, UMB_BLOCKS_BWD)
        desc.c1 = _ptr(p1)
        run(4, 1)
        p0, q0, r0, g_g0, g_b0 = bwd_coeffs(10, rows, part, 2, 1, v0, dev, UMB_BLOCKS_BWD)
        desc.c0 = _ptr(p0)
        run(5, 0)
        _stack_ends()
        shp = meta["shapes"]
        g_c1 = _zeros.take(10, dev)             # bias before BN: exactly 0
        return (None, None, res[0, :100].reshape(shp[0]), g_g0, g_b0, res[1, :100].reshape(shp[1]), g_c1, g_g1, g_b1,
                res[2, :100].reshape(shp[2]), res[2, 100:])


# ------------------------------------------------------------------------------------------- constructor MLP on the matrix pipe
class UmbrellaMFMADesc(ctypes.Structure):      # rs_umbrella_mfma
    _fields_ = [("x", P), ("rows", c_ll), ("group", c_int), ("layers", c_int),
                ("w0", P), ("b0", P), ("w1", P), ("b1", P), ("w2", P), ("b2", P),
                ("gamma0", P), ("beta0", P), ("gamma1", P), ("beta1", P),
                ("eps0", ctypes.c_float), ("eps1", ctypes.c_float), ("mom0", ctypes.c_float), ("mom1", ctypes.c_float),
                ("bn0", P), ("bn1", P), ("run_mean0", P), ("run_var0", P), ("run_mean1", P), ("run_var1", P),
                ("moments", P), ("dout", P), ("stat", P), ("nblk_f1", c_int), ("part_b1", P), ("nblk_b1", c_int),
                ("part_b2", P), ("nblk_b2", c_int), ("out_scale", ctypes.c_float), ("out", P), ("grads", P), ("rows_dev", P)]


UMB_F1, UMB_F2, UMB_B1, UMB_B2, UMB_FIN = 1, 2, 3, 4, 5
UMB_MOM_ROW, UMB_B1_ROW, UMB_B2_ROW, UMB_GRADS = 176, 544, 368, 360
UMB_MFMA = os.environ.get("REPSURF_UMB_MFMA", "1") != "0"         # 0: the register-resident VALU passes of csrc/umbrella_mlp.hip
# three-layer (classification) constructor: which direction runs on the matrix pipe (the BatchNorm vectors the forward publishes are
# the ones the VALU backward passes read, so the two mix freely); measured per direction inside the step, DESIGN.md 5
# Round 4, same box, three interleaved rounds of bench.py (ms per step): VALU passes 1.420 / 1.432 / 1.423, matrix-pipe forward + VALU
# backward 1.427 / 1.434 / 1.423, matrix pipe both ways 1.441 / 1.438 / 1.444 -- stand-alone the matrix-pipe passes are level
# (86 against 83 us) but their 512-thread workgroups share the CUs worse with the geometry stream's kernels at the head of the step;
# the two-layer (segmentation) variant wins both stand-alone (53 against 76 us) and in the step (4.31 against 4.35 ms).  So the
# three-layer constructor stays on the VALU passes by default.
UMB_MFMA_FWD3 = os.environ.get("REPSURF_UMB_MFMA_FWD3", "0") != "0"
UMB_MFMA_BWD3 = os.environ.get("REPSURF_UMB_MFMA_BWD3", "1") != "0"
UMB_MFMA_BLOCKS = int(os.environ.get("REPSURF_UMB_MFMA_BLOCKS", "256"))


def _umb_blocks(rows, group):
    """workgroups of a constructor pass: 8 waves each, a wave takes tiles of 16 points; at least one tile per wave"""
    tiles = -(-(rows // group) // 16)
    return max(1, min(UMB_MFMA_BLOCKS, -(-tiles // 8)))


def umbrella_moments(x):
    """(11, 16) fp64 moments of the (rows, 10) constructor features: S[m][n] = sum x_m x_n, index 10 = the constant 1.  They depend
    on the geometry only: the pipelined step computes them in the geometry stage (side stream), next to the features."""
    rows = x.shape[0]
    nblk = max(1, min(512, -(-rows // 512)))          # a wave takes 64 rows per trip, 8 waves per workgroup
    part = torch.empty((nblk, UMB_MOM_ROW), dtype=torch.float32, device=x.device)
    mom = torch.empty((11, 16), dtype=torch.float64, device=x.device)
    _lib.call("rs_umbrella_moments", _ptr(x), rows, _ptr(part), nblk, _ptr(mom), _stream())
    return mom


def _bn_track(bn_mod):
    track = bn_mod.track_running_stats and bn_mod.running_mean is not None
    if track:
        _pending_counters.append(bn_mod.num_batches_tracked)
    mom = bn_mod.momentum if bn_mod.momentum is not None else 0.1
    return track, float(mom)


class _UmbrellaMFMA(Function):
    """The constructor MLP on v_mfma_f32_16x16x4_f32 (csrc/umbrella_mfma.hip): three layers (classification): F1, F2 forward,
    B1, B2, FIN backward; two layers (segmentation): F2 | B2, FIN.  BatchNorm 0 and the linear part of dW0 come from the moments
    of x; the BatchNorm finalizes are prologues of the consuming passes.
    args: x, meta, moments | None, then the parameters in module order."""

    @staticmethod
    def forward(ctx, x, meta, moments, *params):
        dev = x.device
        x = x.contiguous()
        rows, group, layers = x.shape[0], meta["group"], meta["layers"]
        if moments is None:
            moments = umbrella_moments(x)
        if layers == 3:
            w0, g0, b0, w1, c1, g1, b1, w2, c2 = params
            bn0, bn1 = meta["bns"]
            cb0 = None
        else:
            w0, cb0, g0, b0, w1, c1 = params
            bn0, bn1 = meta["bns"][0], None
            w2 = c2 = g1 = b1 = None
        det = lambda t: None if t is None else t.detach().contiguous()      # noqa: E731
        w0_, w1_, w2_ = _w2d(w0), _w2d(w1), (None if w2 is None else _w2d(w2))
        v0 = BNVec(10, dev)
        v1 = BNVec(10, dev) if layers == 3 else None
        nblk = _umb_blocks(rows, group)
        tr0, mom0 = _bn_track(bn0)
        tr1, mom1 = _bn_track(bn1) if bn1 is not None else (False, 0.1)
        desc = UmbrellaMFMADesc(x=_ptr(x), rows=rows, group=group, layers=layers, w0=_ptr(w0_), b0=_ptr(det(cb0)), w1=_ptr(w1_), b1=_ptr(det(c1)),
                                w2=_ptr(w2_), b2=_ptr(det(c2)), gamma0=_ptr(det(g0)), beta0=_ptr(det(b0)), gamma1=_ptr(det(g1)), beta1=_ptr(det(b1)),
                                eps0=float(bn0.eps), eps1=float(bn1.eps) if bn1 is not None else 0.0, mom0=mom0, mom1=mom1,
                                bn0=_ptr(v0.scale), bn1=None if v1 is None else _ptr(v1.scale),
                                run_mean0=_ptr(bn0.running_mean) if tr0 else None, run_var0=_ptr(bn0.running_var) if tr0 else None,
                                run_mean1=_ptr(bn1.running_mean) if tr1 else None, run_var1=_ptr(bn1.running_var) if tr1 else None,
                                moments=_ptr(moments), rows_dev=_ragged.dev(rows))
        out = torch.empty((rows // group, 10), dtype=torch.float32, device=dev)
        desc.out, desc.out_scale = _ptr(out), (1.0 / group if meta.get("aggr") == "avg" else 1.0)
        if layers == 3:
            stat = torch.empty((nblk, 2, 16), dtype=torch.float64, device=dev)
            desc.stat, desc.nblk_f1 = stat.data_ptr(), nblk
            _lib.call("rs_umbrella_mfma_pass", UMB_F1, ctypes.byref(desc), nblk, _stream())
        _lib.call("rs_umbrella_mfma_pass", UMB_F2, ctypes.byref(desc), nblk, _stream())
        ctx.saved = dict(x=x, moments=moments, w0=w0_, w1=w1_, w2=w2_, cb0=det(cb0), c1=det(c1), c2=det(c2), v0=v0, v1=v1, nblk=nblk)
        ctx.meta = meta
        _flush_counters()
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        if meta["layers"] == 3 and not UMB_MFMA_BWD3:      # the register-resident VALU passes read the same saved tensors / vectors
            g = _UmbrellaFused.backward(ctx, dout)
            return g[:2] + (None,) + g[2:]
        x, v0, v1 = s["x"], s["v0"], s["v1"]
        dev = x.device
        rows, group, layers, nblk = x.shape[0], meta["group"], meta["layers"], s["nblk"]
        dout = dout.contiguous()
        if meta.get("aggr") == "avg":
            dout = dout * (1.0 / group)
        part_b2 = torch.empty((nblk, UMB_B2_ROW), dtype=torch.float32, device=dev)
        grads = torch.empty((UMB_GRADS,), dtype=torch.float32, device=dev)
        desc = UmbrellaMFMADesc(x=_ptr(x), rows=rows, group=group, layers=layers, w0=_ptr(s["w0"]), b0=_ptr(s["cb0"]), w1=_ptr(s["w1"]), b1=_ptr(s["c1"]),
                                w2=_ptr(s["w2"]), b2=_ptr(s["c2"]), bn0=_ptr(v0.scale), bn1=None if v1 is None else _ptr(v1.scale),
                                moments=_ptr(s["moments"]), dout=_ptr(dout), part_b2=_ptr(part_b2), nblk_b2=nblk, grads=_ptr(grads),
                                rows_dev=_ragged.dev(rows))
        if layers == 3:
            part_b1 = torch.empty((nblk, UMB_B1_ROW), dtype=torch.float32, device=dev)
            desc.part_b1, desc.nblk_b1 = _ptr(part_b1), nblk
        if layers == 3:
            _lib.call("rs_umbrella_mfma_pass", UMB_B1, ctypes.byref(desc), nblk, _stream())
        _lib.call("rs_umbrella_mfma_pass", UMB_B2, ctypes.byref(desc), nblk, _stream())
        _lib.call("rs_umbrella_mfma_pass", UMB_FIN, ctypes.byref(desc), nblk, _stream())
        shp = meta["shapes"]
        zero = _zeros.take(10, dev)             # the bias in front of a BatchNorm: exactly 0
        if layers == 3:
            return (None, None, None, grads[0:100].reshape(shp[0]), grads[100:110], grads[110:120], grads[120:220].reshape(shp[1]), zero,
                    grads[230:240], grads[240:250], grads[250:350].reshape(shp[2]), grads[350:360])
        return (None, None, None, grads[0:100].reshape(shp[0]), zero, grads[100:110], grads[110:120], grads[120:220].reshape(shp[1]), grads[220:230])


class _UmbrellaStack2(Function):
    """conv-BN-ReLU-conv, then the sum over the `group` fan triangles: the segmentation constructor's mlps
    (segmentation/modules/repsurface_utils.py:298-303,323-327).  The input (geometric features) never needs a
    gradient."""

    @staticmethod
    def forward(ctx, x, meta, w0, c0, g0, b0, w1, c1):
        dev = x.device
        x = x.contiguous()
        rows, cx = x.shape
        group, training, bn0 = meta["group"], meta["training"], meta["bn"]
        w0_, w1_ = _w2d(w0), _w2d(w1)
        y0, v0 = fwd_layer(rows, operand(OP_ID, x, cx), cx, w0_, c0, bn0, training, dev)
        cout = w1_.shape[0]
        y1 = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        epi = Epilogue(bias=_ptr(c1), out=_ptr(y1), ldo=cout, mode=EPI_STORE)
        gemm_rows(rows, w0_.shape[0], cout, operand(OP_RELU1, y0, y0.shape[1], s1=v0.scale, t1=v0.shift), w_fwd(w1_), epi)
        points = rows // group
        out = torch.empty((points, cout), dtype=torch.float32, device=dev)
        _lib.call("rs_pool_sum", points, group, cout, _ptr(y1), _ptr(out), _stream())
        ctx.saved = dict(x=x, y0=y0, v0=v0, w0=w0_, w1=w1_)
        ctx.meta = meta
        _flush_counters()
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        frozen = not meta["training"]
        x, y0, v0 = s["x"], s["y0"], s["v0"]
        dev = x.device
        rows, cx = x.shape
        group = meta["group"]
        dout = dout.contiguous()
        c1n, c0n = s["w1"].shape[0], s["w0"].shape[0]
        p1 = operand(OP_BCAST, dout, c1n, ns=group)
        from . import head as _head
        g_c1 = _head.col_sum(dout, scale=group)
        _stack_begins()
        g_w1 = wgrad(rows, c1n, c0n, p1, operand(OP_RELU1, y0, c0n, s1=v0.scale, t1=v0.shift), dev, defer=True)     # summed with the finalize below
        dz0, part0, nstat0 = dgrad_masked(rows, c1n, c0n, p1, s["w1"], y0, v0, device=dev)
        pb, qb, rb, g_g0, g_b0 = bwd_coeffs(c0n, rows, part0, nstat0, 1, v0, dev, frozen=frozen)
        p0 = operand(OP_AFF2, dz0, c0n, y0, c0n, s1=pb, t1=rb, s2=qb)
        g_w0 = wgrad(rows, c0n, cx, p0, operand(OP_ID, x, cx), dev, defer=True)
        _stack_ends()
        shp = meta["shapes"]
        g_c0 = v0.scale * g_b0 if frozen else _zeros.take(c0n, dev)   # bias before BN
        return None, None, g_w0.reshape(shp[0]), g_c0, g_g0, g_b0, g_w1.reshape(shp[1]), g_c1


class _UmbrellaFused2(Function):
    """The segmentation constructor's conv-BN-ReLU-conv + sum over the fan on the register-resident passes of
    csrc/umbrella_mlp.hip (two-layer variant): forward = statistics pass, finalize, output pass; backward = {dW1, db1} +
    BatchNorm-backward sums, finalize (+ the fixed-order sum of the dW1 partials), dW0.  The generic row-GEMM path
    (_UmbrellaStack2: 10 of 32 MFMA columns, every intermediate through HBM) took 0.31 ms of a 5.2 ms step for this."""

    @staticmethod
    def forward(ctx, x, meta, w0, c0, g0, b0, w1, c1):
        dev = x.device
        x = x.contiguous()
        rows = x.shape[0]
        group, bn0 = meta["group"], meta["bn"]
        w0_, w1_ = _w2d(w0), _w2d(w1)
        c0_, c1_ = c0.detach().contiguous(), c1.detach().contiguous()
        v0 = BNVec(10, dev)
        desc = UmbrellaMLPDesc(x=_ptr(x), rows=rows, group=group, w0=_ptr(w0_), w1=_ptr(w1_), b1=_ptr(c1_), bn0=_ptr(v0.scale),
                               b0=_ptr(c0_), layers=2)
        part = torch.empty((UMB_BLOCKS, 2, 10), dtype=torch.float64, device=dev)
        _lib.call("rs_umbrella_mlp_pass", 0, ctypes.byref(desc), 1.0, None, part.data_ptr(), None, UMB_BLOCKS, _stream())
        track = bn0.track_running_stats and bn0.running_mean is not None
        if track:
            _pending_counters.append(bn0.num_batches_tracked)
        mom = bn0.momentum if bn0.momentum is not None else 0.1
        _lib.call("rs_bn_finalize", 10, rows, UMB_BLOCKS, part.data_ptr(), _ptr(bn0.weight), _ptr(bn0.bias), float(bn0.eps), float(mom),
                  _ptr(v0.scale), _ptr(v0.shift), _ptr(v0.mean), _ptr(v0.invstd), _ptr(bn0.running_mean) if track else None,
                  _ptr(bn0.running_var) if track else None, _stream())
        out = torch.empty((rows // group, 10), dtype=torch.float32, device=dev)
        _lib.call("rs_umbrella_mlp_pass", 2, ctypes.byref(desc), 1.0, _ptr(out), None, None, UMB_BLOCKS, _stream())
        ctx.saved = dict(x=x, w0=w0_, w1=w1_, c0=c0_, c1=c1_, v0=v0)
        ctx.meta = meta
        _flush_counters()
        return out

    @staticmethod
    def backward(ctx, dout):
        s, meta = ctx.saved, ctx.meta
        x, v0 = s["x"], s["v0"]
        dev = x.device
        rows, group = x.shape[0], meta["group"]
        dout = dout.contiguous()
        desc = UmbrellaMLPDesc(x=_ptr(x), rows=rows, group=group, w0=_ptr(s["w0"]), w1=_ptr(s["w1"]), b1=_ptr(s["c1"]), bn0=_ptr(v0.scale),
                               b0=_ptr(s["c0"]), dout=_ptr(dout), layers=2)
        part = torch.empty((UMB_BLOCKS_BWD, 2, 10), dtype=torch.float64, device=dev)
        dwp = torch.empty((2, UMB_BLOCKS_BWD, 110), dtype=torch.float32, device=dev)
        res = torch.empty((2, 110), dtype=torch.float32, device=dev)
        _stack_begins()
        _lib.call("rs_umbrella_mlp_pass", 4, ctypes.byref(desc), 1.0, None, part.data_ptr(), _ptr(dwp[1]), UMB_BLOCKS_BWD, _stream())
        _pending_reduce.append((dwp[1], UMB_BLOCKS_BWD, 110, res[1]))
        p0, q0, r0, g_g0, g_b0 = bwd_coeffs(10, rows, part, 2, 1, v0, dev, UMB_BLOCKS_BWD)
        desc.c0 = _ptr(p0)
        _lib.call("rs_umbrella_mlp_pass", 5, ctypes.byref(desc), 1.0, None, None, _ptr(dwp[0]), UMB_BLOCKS_BWD, _stream())
        _pending_reduce.append((dwp[0], UMB_BLOCKS_BWD, 110, res[0]))
        _stack_ends()
        shp = meta["shapes"]
        g_c0 = _zeros.take(10, dev)             # bias before BN: exactly 0
        return (None, None, res[0, :100].reshape(shp[0]), g_c0, g_g0, g_b0, res[1, :100].reshape(shp[1]), res[1, 100:])


def umbrella_mlp2(x, mlps, group, moments=None):
    """moments: umbrella_moments(x) when the caller computed them ahead of time (geometry stage)."""
    conv0, bn0, _, conv1 = mlps
    meta = {"group": group, "bn": bn0, "training": bn0.training, "shapes": [conv0.weight.shape, conv1.weight.shape]}
    fused = (x.shape[1] == 10 and tuple(conv0.weight.shape[:2]) == (10, 10) and tuple(conv1.weight.shape[:2]) == (10, 10)
             and conv0.bias is not None and conv1.bias is not None and bn0.training and FUSED_UMBRELLA
             and sync_of(bn0) is None)      # (SyncBatchNorm: the generic stack, whose finalizes take the all-reduced sums)
    if fused and UMB_MFMA and bn0.weight is not None and group in (8, 9):      # (the tile's fan rows are register arrays: fans of 8 / 9)
        meta.update(layers=2, bns=(bn0,))
        return _UmbrellaMFMA.apply(x, meta, moments, conv0.weight, conv0.bias, bn0.weight, bn0.bias, conv1.weight, conv1.bias)
    fn = _UmbrellaFused2 if fused else _UmbrellaStack2
    return fn.apply(x, meta, conv0.weight, conv0.bias, bn0.weight, bn0.bias, conv1.weight, conv1.bias)


def umbrella_mlp(x, mlps, group, aggr, moments=None):
    """moments: umbrella_moments(x) when the caller computed them ahead of time (geometry stage)."""
    conv0, bn0, _, conv1, bn1, _, conv2 = mlps
    meta = {"group": group, "aggr": aggr, "bns": (bn0, bn1), "training": bn0.training,
            "shapes": [conv0.weight.shape, conv1.weight.shape, conv2.weight.shape]}
    fused = (x.shape[1] == 10 and conv0.weight.shape[:2] == (10, 10) and conv1.weight.shape[:2] == (10, 10)
             and conv2.weight.shape[:2] == (10, 10) and aggr in ("sum", "avg") and bn0.training and FUSED_UMBRELLA
             and sync_of(bn0) is None and sync_of(bn1) is None)
    if fused and UMB_MFMA and UMB_MFMA_FWD3 and conv0.bias is None and bn0.weight is not None and bn1.weight is not None and group in (8, 9):
        meta["layers"] = 3
        return _UmbrellaMFMA.apply(x, meta, moments, conv0.weight, bn0.weight, bn0.bias, conv1.weight, conv1.bias, bn1.weight,
                                   bn1.bias, conv2.weight, conv2.bias)
    fn = _UmbrellaFused if fused else _UmbrellaStack
    return fn.apply(x, meta, conv0.weight, bn0.weight, bn0.bias, conv1.weight, conv1.bias, bn1.weight,
                    bn1.bias, conv2.weight, conv2.bias)


# ------------------------------------------------------------------------------------------- plain row linear
class _FPFront(Function):
    """Feature propagation, everything in front of the [Linear, BN, ReLU]* chain (segmentation/modules/repsurface_utils.py:256-270):
        out = relu(interpolate(BN_f(Linear_f(points2)), idx, weight) + BN_s(Linear_s(points1)))
    as ONE autograd node (round 4): two row GEMMs with BatchNorm sums, one finalize launch for both BatchNorms, one interpolation
    launch that applies both affine maps on the fly -- no BatchNorm pass materialises the normalised tensors (two passes per
    stage and direction, 65 536 x 128 floats each way at the finest level); backward: one launch scatters the masked gradient to
    the coarse rows, writes it for the skip branch and leaves that branch's BatchNorm-backward sums, one pass over the 4 x smaller
    coarse gradient makes the other branch's sums, one launch turns both into coefficients, then the usual weight / data
    gradient GEMMs.  Training mode, batch statistics (eval / SyncBatchNorm take the layer-by-layer route)."""

    @staticmethod
    def forward(ctx, points2, points1, idx, weight, meta, wf, bf, gf, betaf, ws, bs, gs, betas):
        dev = points2.device
        lazy2 = meta.get("lazy_in")       # points2 = the raw last output of the previous stage's stack (LazyRows): BN + ReLU in the operand prologue
        points2, points1, weight = points2.contiguous(), points1.contiguous(), weight.contiguous()
        idx = (idx if idx.dtype == torch.int32 else idx.to(torch.int32)).contiguous()
        m, c2 = points2.shape
        n, c1 = points1.shape
        bn_f, bn_s = meta["bns"]
        wf2, ws2 = _w2d(wf), _w2d(ws)
        c = wf2.shape[0]
        op_in = operand(OP_ID, points2, c2) if lazy2 is None else operand(OP_RELU1, points2, c2, s1=lazy2.vec.scale, t1=lazy2.vec.shift)
        y2, v2, it2 = fwd_layer(m, op_in, c2, wf2, bf, bn_f, True, dev, wk=w_fwd(wf2), finalize=False)
        y1, v1, it1 = fwd_layer(n, operand(OP_ID, points1, c1), c1, ws2, bs, bn_s, True, dev, wk=w_fwd(ws2), finalize=False)
        bn_finalize_batch([it2, it1])
        out = torch.empty((n, c), dtype=torch.float32, device=dev)
        _lib.call("rs_three_interpolate_affine", 1, c, m, n, _ptr(y2), _ptr(v2.scale), _ptr(v2.shift), idx.data_ptr(), _ptr(weight),
                  _ptr(y1), _ptr(v1.scale), _ptr(v1.shift), 1, _ptr(out), _stream())
        _flush_counters()
        ctx.saved = dict(points2=points2, points1=points1, idx=idx, weight=weight, y2=y2, y1=y1, v2=v2, v1=v1, out=out.detach(), wf2=wf2, ws2=ws2, lazy2=lazy2,
                         csr=meta.get("csr") if (meta.get("csr") is not None and meta["csr"][0].numel() == m + 1 and m * c < 2 ** 31) else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        s = ctx.saved
        points2, points1, y2, y1, v2, v1, wf2, ws2 = s["points2"], s["points1"], s["y2"], s["y1"], s["v2"], s["v1"], s["wf2"], s["ws2"]
        dev = dout.device
        m, c2 = points2.shape
        n, c1 = points1.shape
        c = wf2.shape[0]
        dout = dout.contiguous()
        _stack_begins()
        g = torch.empty((n, c), dtype=torch.float32, device=dev)             # gradient at BN_s's output (= at the sum, masked)
        nb1 = max(1, min(2048, -(-(n * c) // 1024)))      # one partial row per workgroup of the interpolation backward (<= 2048: its usual grid)
        part1 = torch.empty((nb1, 2, c), dtype=torch.float64, device=dev)
        csr = s["csr"]
        if csr is not None:
            # gather form (ops.inverse_index of idx, built with the geometry): one pass masks the gradient and sums the skip branch's
            # moments, one pass gathers the coarse rows' gradient in ascending edge order and sums THEIR moments -- no atomics, no fill
            d2 = torch.empty((m, c), dtype=torch.float32, device=dev)
            _lib.call("rs_three_interpolate_affine_backward", 1, c, n, m, _ptr(dout), _ptr(s["out"]), s["idx"].data_ptr(), _ptr(s["weight"]),
                      None, _ptr(g), _ptr(y1), _ptr(v1.mean), _ptr(v1.invstd), part1.data_ptr(), nb1, _ragged.dev(n), _stream())
            nb2 = max(1, min(2048, -(-(m * c) // 256)))
            part2 = torch.empty((nb2, 2, c), dtype=torch.float64, device=dev)
            _lib.call("rs_three_interpolate_backward_csr", m, c, _ptr(g), None, None, _ptr(s["weight"]), _ptr(csr[0]), _ptr(csr[1]), _ptr(d2),
                      _ptr(y2), _ptr(v2.mean), _ptr(v2.invstd), part2.data_ptr(), nb2, _stream())
        else:
            d2 = torch.zeros((m, c), dtype=torch.float32, device=dev)            # gradient at BN_f's output (scatter target)
            _lib.call("rs_three_interpolate_affine_backward", 1, c, n, m, _ptr(dout), _ptr(s["out"]), s["idx"].data_ptr(), _ptr(s["weight"]),
                      _ptr(d2), _ptr(g), _ptr(y1), _ptr(v1.mean), _ptr(v1.invstd), part1.data_ptr(), nb1, _ragged.dev(n), _stream())
            nb2 = partial_rows(m, 4)
            part2 = torch.empty((nb2, 2, c), dtype=torch.float64, device=dev)
            _lib.call("rs_pool_max_backward", m, 1, c, None, _ptr(d2), c, None, None, _ptr(y2), 0, _ptr(v2.mean), _ptr(v2.invstd), None,
                      part2.data_ptr(), nb2, _ragged.dev(m), _stream())
        (p1, q1, r1, dg1, db1), (p2, q2, r2, dg2, db2) = bwd_coeffs_multi(
            [(c, n, part1, 2, 1, v1, None, False), (c, m, part2, 2, 1, v2, None, False)], dev)
        op1 = operand(OP_AFF2, g, c, y1, c, s1=p1, t1=r1, s2=q1)
        op2 = operand(OP_AFF2, d2, c, y2, c, s1=p2, t1=r2, s2=q2)
        dws = wgrad(n, c, c1, op1, operand(OP_ID, points1, c1), dev, None, defer=True)
        lazy2 = s["lazy2"]
        q2 = operand(OP_ID, points2, c2) if lazy2 is None else operand(OP_RELU1, points2, c2, s1=lazy2.vec.scale, t1=lazy2.vec.shift)
        dwf = wgrad(m, c, c2, op2, q2, dev, None, defer=True)
        dp1 = dp2 = None
        if ctx.needs_input_grad[1]:
            dp1 = torch.empty((n, c1), dtype=torch.float32, device=dev)
            gemm_rows(n, c, c1, op1, w_bwd(ws2), Epilogue(bias=None, out=_ptr(dp1), ldo=c1, mode=EPI_STORE))
        if ctx.needs_input_grad[0]:
            if lazy2 is not None:
                dp2, part_in, nstat_in = dgrad_masked(m, c, c2, op2, wf2, points2, lazy2.vec, device=dev, wt=w_bwd(wf2))
                lazy2.part = (part_in, nstat_in)
            else:
                dp2 = torch.empty((m, c2), dtype=torch.float32, device=dev)
                gemm_rows(m, c, c2, op2, w_bwd(wf2), Epilogue(bias=None, out=_ptr(dp2), ldo=c2, mode=EPI_STORE))
        _stack_ends()
        zb = _zeros.take(2 * c, dev)           # biases in front of a BatchNorm with batch statistics: exactly zero
        return (dp2, dp1, None, None, None, dwf.reshape(s["wf2"].shape), zb[:c], dg2, db2, dws.reshape(s["ws2"].shape), zb[c:], dg1, db1)


def fp_front_usable(lin_f, bn_f, lin_s, bn_s):
    """the fused node serves training-mode BatchNorm with batch statistics, fp32 arithmetic, at most 256 channels"""
    from . import mlp as _mlp
    return (bn_f.training and bn_s.training and _mlp.PRECISION == "fp32" and lin_f.out_features <= 256
            and lin_f.bias is not None and lin_s.bias is not None and sync_of(bn_f) is None and sync_of(bn_s) is None
            and os.environ.get("REPSURF_FP_FRONT", "1") != "0")


def fp_front(points2, points1, idx, weight, lin_f, bn_f, lin_s, bn_s, csr=None):
    """relu(interpolate(bn_f(lin_f(points2)), idx, weight) + bn_s(lin_s(points1))): points2 (M, C2) coarse rows, points1 (N, C1)
    fine rows, idx / weight (N, 3) -> (N, C).  csr: ops.inverse_index(idx, 3, ...) -- the backward then gathers."""
    meta = {"bns": (bn_f, bn_s), "csr": csr}
    if isinstance(points2, LazyRows):
        meta["lazy_in"] = points2
        points2 = points2.y
    return _FPFront.apply(points2, points1, idx, weight, meta, lin_f.weight, lin_f.bias, bn_f.weight, bn_f.bias,
                          lin_s.weight, lin_s.bias, bn_s.weight, bn_s.bias)


class _RowLinear(torch.autograd.Function):
    """y = x . W^T + b on ungrouped rows (the 13-class output layer of the segmentation classifier,
    segmentation/models/repsurf/repsurf_umb_ssg.py:36-41): the row GEMM with a plain store epilogue forward, the same
    kernel on the transposed weight copy for dx and the weight-gradient kernel for dW -- no library GEMM in the step."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        rows, k = x.shape
        n = w.shape[0]
        out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
        epi = Epilogue(bias=_ptr(b), out=_ptr(out), ldo=n, mode=EPI_STORE)
        gemm_rows(rows, k, n, operand(OP_ID, x, k), w_fwd(w), epi)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dout = dout.contiguous()
        rows, k = x.shape
        n = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((rows, k), dtype=torch.float32, device=x.device)
            epi = Epilogue(bias=None, out=_ptr(dx), ldo=k, mode=EPI_STORE)
            gemm_rows(rows, n, k, operand(OP_ID, dout, n), w_bwd(w), epi)
        if ctx.needs_input_grad[1]:
            if n <= 16 and k % 4 == 0:
                # few output classes: reduce as dW^T = x^T . dout, whose narrow side (<= 16 columns of dout) takes the
                # streaming weight-gradient kernel; the other orientation reads dout with scalar loads through the generic
                # MFMA instance (141 us against ~20 at 65 536 x 13 x 128)
                dw = wgrad(rows, k, n, operand(OP_ID, x, k), operand(OP_ID, dout, n), x.device).t().contiguous()
            else:
                dw = wgrad(rows, n, k, operand(OP_ID, dout, n), operand(OP_ID, x, k), x.device)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from . import head as _head
            db = _head.col_sum(dout)
        return dx, dw, db


def row_linear(x, linear):
    """nn.Linear on rows through the HIP row GEMM (x (rows, cin) -> (rows, cout))."""
    return _RowLinear.apply(x, linear.weight, linear.bias)
