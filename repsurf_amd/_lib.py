"""ctypes binding of librepsurf_hip.so (the C ABI declared in include/repsurf_hip.h).

There is NO fallback: if the shared library is missing or does not export a symbol this module
raises, and every operator in repsurf_amd.ops raises with it.  Build with `make` at the repo
root or `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REPSURF_HIP_LIB") or os.path.join(_HERE, "lib", "librepsurf_hip.so")   # override: experiment builds only
ABI_VERSION = 37
KNN_GRID_CELLS = 4096          # RS_KNN_GRID_CELLS of include/repsurf_hip.h

c_int, c_float, c_void_p, c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong
P = c_void_p  # device pointers and the stream travel as void*

# name -> argtypes (restype is always int unless listed in _SPECIAL)
SIGNATURES = {
    "rs_furthestsampling": [c_int, c_int, c_int, P, P, P, P, P],
    "rs_furthestsampling_offset": [c_int, c_int, P, P, P, P, P, P],
    "rs_gather_rows": [c_int, c_int, c_int, c_int, P, P, P, P],
    "rs_gather_rows_backward": [c_int, c_int, c_int, c_int, P, P, P, P],
    "rs_gather_rows_backward_dev": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "rs_ballquery": [c_int, c_int, c_int, c_float, c_int, P, P, P, P, P],
    "rs_knnquery": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "rs_knnquery_offset": [c_int, c_int, P, P, P, P, c_int, P, P, P],
    "rs_umbrella_fan_offset": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P],
    "rs_interp_weights": [c_ll, P, P, P],
    "rs_umbrella_features": [c_int, c_int, c_int, P, P, P, P, P],
    "rs_umbrella_features_grid": [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P],
    "rs_group_features": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, c_int, P],
    "rs_group_features_backward": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, c_int, P],
    "rs_group_features_backward_dev": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, c_int, P, P],
    "rs_group_all_features": [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P],
    "rs_exclusive_scan": [c_int, P, P, P],
    "rs_compact_index": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P],
    "rs_group_features_compact": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P, P],
    "rs_group_features_compact_backward": [c_ll, P, c_int, c_int, c_int, P, P, P, P, c_int, c_int, c_int, P, P, c_ll, P],
    "rs_compact_csr": [c_int, c_int, c_int, P, P, P, P, P, P, P],
    "rs_inverse_index": [c_int, c_int, c_int, P, P, P, P, P, P, P],
    "rs_group_features_backward_csr": [c_ll, c_int, c_int, c_int, c_int, P, P, P, P, P, P],
    "rs_group_features_compact_backward_csr": [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_ll, P, c_int, P],
    "rs_group_rows": [c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "rs_group_rows_backward": [c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "rs_three_nn": [c_int, c_int, c_int, P, P, P, P, P],
    "rs_three_interpolate": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "rs_three_interpolate_backward": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "rs_three_interpolate_fused": [c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P],
    "rs_three_interpolate_fused_backward": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P],
    "rs_three_interpolate_fused_backward_dev": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P],
    "rs_three_interpolate_affine": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P, P],
    "rs_three_interpolate_affine_backward": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, P],
    "rs_three_interpolate_backward_csr": [c_ll, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P],
    "rs_mlp_gemm_rows": [c_ll, P, c_int, c_int, P, P, c_int, P, P],
    "rs_mlp_gemm_rows_bf16": [c_ll, P, c_int, c_int, P, P, c_int, P, P],
    "rs_mlp_wgrad": [c_ll, P, c_int, c_int, P, P, P, c_int, P, P],
    "rs_mlp_wgrad_bf16": [c_ll, P, c_int, c_int, P, P, P, c_int, P, P],
    "rs_bn_finalize": [c_int, c_ll, c_int, P, P, P, c_float, c_float, P, P, P, P, P, P, P],
    "rs_bn_backward_finalize": [c_int, c_ll, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P],
    "rs_pool_max": [c_ll, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P],
    "rs_pool_max_backward": [c_ll, c_int, c_int, P, P, c_ll, P, P, P, c_int, P, P, P, P, c_int, P, P],
    "rs_pool_sum": [c_ll, c_int, c_int, P, P, P],
    "rs_pool_select": [c_ll, c_int, P, P, P, P, P, P, P, P, P],
    "rs_reduce_partials": [c_int, c_ll, P, P, P],
    "rs_backward_tail": [P, P],
    "rs_cross_entropy_forward": [c_ll, c_int, c_ll, P, P, P, P, P, P, P, P],
    "rs_scale_by_scalars": [c_ll, P, P, P, P, P],
    "rs_col_sum_partials": [c_ll, c_int, P, c_ll, ctypes.c_float, P, c_int, P],
    "rs_col_sum_partials_dev": [c_ll, c_int, P, c_ll, ctypes.c_float, P, c_int, P, P],
    "rs_bn_finalize_batch": [P, c_int, P],
    "rs_bn_backward_finalize_reduce": [c_int, c_ll, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, c_ll, P, P, P],
    "rs_pack_weights": [P, P],
    "rs_umbrella_mlp_pass": [c_int, P, c_float, P, P, P, c_int, P],
    "rs_umbrella_moments": [P, c_ll, P, c_int, P, P],
    "rs_umbrella_mfma_pass": [c_int, P, c_int, P],
    "rs_head_layer_forward": [P, P],
    "rs_head_output_forward": [c_int, c_int, c_int, P, P, P, P, P, P],
    "rs_head_output_backward": [c_int, c_int, c_int, P, P, P, P, P, P, P],
    "rs_head_layer_backward": [P, P],
    "rs_head_input_backward": [c_int, c_int, c_int, P, P, P, P],
    "rs_smooth_cls_loss": [c_int, c_int, c_float, P, P, P, P, P],
    "rs_adam_step": [P, P, P, P, c_int, P],
    "rs_knn_grid_build": [c_int, P, P, ctypes.c_float, P, P, P, P],
    "rs_knn_grid_query": [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P],
    "rs_scene_cells": [c_int, P, P, P, ctypes.c_float, P, P, P, P],
    "rs_scene_scatter": [c_int, P, P, P, P, P],
    "rs_scene_knn": [c_int, c_int, P, P, P, ctypes.c_float, P, P, P, P, P, P, P],
    "rs_sectorize": [c_int, P, P, P, P, c_int, c_int, P, P, P, P, P, P],
    "rs_furthestsampling_sectors": [c_int, c_int, P, P, P, P, P, P, P],
    "rs_take_int": [c_int, P, P, P, P],
    "rs_timestamp": [P, P],
    "rs_ballquery_grid_build": [c_int, c_int, c_float, P, P, P],
    "rs_ballquery_grid_query": [c_int, c_int, c_int, c_float, c_int, P, P, P, P, P],
}
_SPECIAL = {
    "rs_last_error": ([], ctypes.c_char_p),
    "rs_abi_version": ([], c_int),
    "rs_mlp_gemm_split3": ([], c_int),
    "rs_timestamp_khz": ([], c_int),
    "rs_ballquery_grid_bytes": ([c_int, c_int], c_ll),
    "rs_device_info": ([ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.c_char_p, c_int], c_int),
}

_lib = None


class RepSurfHipError(RuntimeError):
    pass


def load():
    """Load the library once, bind every declared symbol, check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RepSurfHipError(
            f"{LIB_PATH} not found: the HIP library is not built (run `make` in the repo root). "
            "repsurf_amd has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError -> missing export, propagate loudly
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (argtypes, restype) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.rs_abi_version()
    if got != ABI_VERSION:
        raise RepSurfHipError(f"librepsurf_hip ABI {got} != binding ABI {ABI_VERSION}: rebuild with `make`")
    _lib = lib
    return lib


_profile = None   # None = off; else list of (name, dims, start_event, end_event, row-count slot | None)
_rows_host = None  # pinned int32 arena the row counts of compacted launches are copied into (asynchronously)
_rows_used = 0


def profile_enable(on):
    """Per-launch timing of ABI calls with HIP events recorded on the launch stream (torch's
    current stream, the one every operator passes down).  Used by bench.py for the live
    `roofline` numbers; off by default (two event records per call)."""
    global _profile, _frozen, _rows_used
    if on:
        _profile = []
        _rows_used = 0
    elif _profile is not None:
        _frozen, _profile = _profile, None


_frozen = []


def _dims_of(entry):
    name, dims, _, _, slot = entry
    if slot is not None:
        dims = dims + (f"rows={int(_rows_host[slot])}",)      # dims[0] stays the (static) capacity
    return dims


def profile_collect():
    """-> {abi name: [(milliseconds, (int dims...)), ...]}; call after a device synchronise."""
    out = {}
    for entry in _frozen:
        out.setdefault(entry[0], []).append((entry[2].elapsed_time(entry[3]), _dims_of(entry)))
    return out


_hip = None


def _copy_device_int_async(ptr):
    """4-byte device->pinned-host copy on the launch stream, NOT waited for: a compacted launch's row count lives on the
    device, and reading it back synchronously (as round 1 did) drained the stream in front of every such launch -- the
    timed kernels then started on an idle, cooled-down GPU and measured 15-20 % slow.  Returns the arena slot."""
    global _hip, _rows_host, _rows_used
    import torch
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p]
        _hip.hipMemcpyAsync.restype = c_int
    if _rows_host is None:
        _rows_host = torch.zeros((1 << 16,), dtype=torch.int32).pin_memory()
    slot = _rows_used
    if slot >= _rows_host.numel():
        raise RepSurfHipError("profiling arena exhausted (more than 65536 compacted launches in one profiled region)")
    _rows_used += 1
    rc = _hip.hipMemcpyAsync(_rows_host.data_ptr() + 4 * slot, ptr, 4, 2, torch.cuda.current_stream().cuda_stream)   # 2 = DtoH
    if rc != 0:
        raise RepSurfHipError(f"hipMemcpyAsync of a device row count failed (hip error {rc})")
    return slot


def profile_sequence():
    """-> [(abi name, dims), ...] in launch order for the last profiled region (after a device synchronise)."""
    return [(entry[0], _dims_of(entry)) for entry in _frozen]


GEMM_FAMILY = ("rs_mlp_gemm_rows", "rs_mlp_gemm_rows_bf16", "rs_mlp_wgrad", "rs_mlp_wgrad_bf16")
_recorded = None   # None = off; else [(name, args)] of the GEMM-family calls made while recording


def record_calls(on):
    """Keep (name, arguments) of every GEMM-family ABI call made from now on (bench.py replays them as ONE hipGraph to time the
    matrix-pipe launches of a step the way a replayed step runs them: back to back, no host in between).  The argument tuples keep
    their ctypes structs alive; the caller keeps the TENSORS alive (it holds the step's loss, hence the autograd nodes' saved state).
    record_calls(False) -> the list."""
    global _recorded
    if on:
        _recorded = []
        return None
    out, _recorded = _recorded, None
    return out or []


def replay_calls(calls, stream):
    """Re-issue recorded calls on `stream` (a raw hipStream_t; the stream argument is the last one of every ABI function)."""
    lib = load()
    for name, args in calls:
        rc = getattr(lib, name)(*args[:-1], stream)
        if rc != 0:
            msg = lib.rs_last_error()
            raise RepSurfHipError(f"{name} (replayed) failed (code {rc}): {msg.decode() if msg else '?'}")


# Submit-per-launch (round 6, DIAGNOSIS).  The runtime batches the AQL packets of a stream and rings the doorbell when the batch is flushed.
# While the two-stream deviations of the fan-feature kernel (profiles/r06/eager_beside_graph.txt) still read as an ordering problem between
# eager launches, `submit_each_launch(True)` -- every ABI call ends with hipStreamQuery on its stream, which submits the batch -- was one of
# the switches tried (no effect: 5, 5 of 1 600).  The cause was elsewhere (compiler-vectorized packed-fp32 code beside MFMA waves: Makefile);
# nothing in the product path uses this.
_submit_each = 0
_hip_query = None
SUBMIT_ALWAYS = os.environ.get("REPSURF_SUBMIT_EACH_LAUNCH", "0") != "0"      # (diagnosis: every launch of the process)


class submit_each_launch:
    """with submit_each_launch(): ... every ABI launch is submitted to the hardware queue at once (hipStreamQuery after it)."""

    def __enter__(self):
        global _submit_each
        _submit_each += 1
        return self

    def __exit__(self, *exc):
        global _submit_each
        _submit_each -= 1


def submit(stream=None):
    """hipStreamQuery(stream): submits what the runtime has batched for this stream (result ignored: not-ready is the normal answer)."""
    global _hip_query
    if _hip_query is None:
        h = ctypes.CDLL("libamdhip64.so")
        h.hipStreamQuery.argtypes = [c_void_p]
        h.hipStreamQuery.restype = c_int
        _hip_query = h.hipStreamQuery
    _hip_query(current_stream() if stream is None else stream)


def call(name, *args):
    """Invoke an ABI function; non-zero return -> RepSurfHipError with the library's message."""
    lib = load()
    if _recorded is not None and name in GEMM_FAMILY:
        _recorded.append((name, args))
    if _profile is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        dims = tuple(a for a, t in zip(args, SIGNATURES[name]) if t is c_int or t is c_ll)
        slot = None
        # operand / epilogue modes: a forward GEMM and the data-gradient GEMM of the same sizes read different numbers of
        # tensors (1-2 against 3-4) -- separate launch classes with their own algorithmic bytes
        if name in ("rs_mlp_gemm_rows", "rs_mlp_gemm_rows_bf16"):
            op, ep = args[4]._obj, args[7]._obj
            dims = dims + (f"op={op.mode}", f"epi={ep.mode}{'+2' if ep.my2 else ''}")
            sb = f"{op.a_bf16}{op.b_bf16}{ep.out_bf16}{ep.my1_bf16}{ep.my2_bf16}"     # bf16-stored tensors: a, b, out, my1, my2
            if "1" in sb:
                dims = dims + (f"sb={sb}",)
        elif name in ("rs_mlp_wgrad", "rs_mlp_wgrad_bf16"):
            pp, qq = args[4]._obj, args[5]._obj
            dims = dims + (f"p={pp.mode}", f"q={qq.mode}")
            sb = f"{pp.a_bf16}{pp.b_bf16}{qq.a_bf16}{qq.b_bf16}"                     # P.a, P.b, Q.a, Q.b
            if "1" in sb:
                dims = dims + (f"sb={sb}",)
        if name in ("rs_mlp_gemm_rows", "rs_mlp_gemm_rows_bf16", "rs_mlp_wgrad", "rs_mlp_wgrad_bf16") and args[1] is not None:
            # compacted operand: args[0] is only the capacity, the launch's row count lives on the device
            slot = _copy_device_int_async(args[1])
        _profile.append((name, dims, e0, e1, slot))
    else:
        rc = getattr(lib, name)(*args)
    if _submit_each or SUBMIT_ALWAYS:
        import torch
        if not torch.cuda.is_current_stream_capturing():      # (a query would invalidate a capture; captured launches are graph nodes anyway)
            submit(args[-1])
    if rc != 0:
        msg = lib.rs_last_error()
        raise RepSurfHipError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")


_raw_stream = None


def current_stream():
    """Raw hipStream_t of torch's current stream on the current device -- what every operator hands to the ABI.  Through torch's raw
    accessor: `torch.cuda.current_stream().cuda_stream` builds a Stream object per call (~4 us; an eagerly launched step makes ~450
    of them, a tenth of its host time: tools/eager_host_profile.py)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get_raw, get_dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None and os.environ.get("REPSURF_STREAM_OBJECT", "0") == "0":
            _raw_stream = lambda: get_raw(get_dev())      # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
    return _raw_stream()


def device_key(device):
    """'cuda:<index>' (index resolved: torch.device('cuda') and 'cuda' name the CURRENT device) -- the key of the per-device counter tables
    (mlp_hip.sync_row_mismatch_count, ops.inverse_index_overflow_count, head.bad_label_count: ADVICE r5, a query with 'cuda' read 0)."""
    import torch
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return str(d)


def device_info():
    lib = load()
    cu, wave, lds = c_int(), c_int(), c_int()
    arch = ctypes.create_string_buffer(64)
    rc = lib.rs_device_info(ctypes.byref(cu), ctypes.byref(wave), ctypes.byref(lds), arch, 64)
    if rc != 0:
        raise RepSurfHipError(f"rs_device_info failed: {lib.rs_last_error().decode()}")
    return {"cu_count": cu.value, "wave_size": wave.value, "lds_bytes": lds.value, "arch": arch.value.decode()}
