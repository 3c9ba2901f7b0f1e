"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/repsurf_hip.h declares, and the Python binding covers them all (no compute)."""
import os
import re

from tests.util import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "repsurf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from repsurf_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in repsurf_hip.h but not exported"


def test_binding_covers_header():
    from repsurf_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib._SPECIAL)
    assert set(declared_symbols()) == bound


def test_header_argument_counts_match_binding():
    from repsurf_amd import _lib
    text = open(os.path.join(ROOT, "include", "repsurf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(argtypes), name


def test_argument_errors_are_reported_not_fatal():
    from repsurf_amd import _lib
    import pytest
    with pytest.raises(_lib.RepSurfHipError, match="negative size"):
        _lib.call("rs_ballquery", -1, 1, 1, 0.1, 1, None, None, None, None, None)
    with pytest.raises(_lib.RepSurfHipError, match="null pointer"):
        _lib.call("rs_furthestsampling", 1, 8, 2, None, None, None, None, None)
    with pytest.raises(_lib.RepSurfHipError, match="cannot supply"):        # (any nsample <= n is served since round 3: csrc/knn_wide.hip)
        _lib.call("rs_knnquery", 1, 10, 1, 65, 1, 1, 1, None, None)
    # zero-sized problems are valid no-ops
    _lib.call("rs_ballquery", 0, 0, 0, 0.1, 0, None, None, None, None, None)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every struct the binding passes by pointer has the size and field offsets the C compiler gives the header's
    definition (gcc on include/repsurf_hip.h): a field added on one side only would shift everything behind it silently."""
    import ctypes
    import subprocess
    from repsurf_amd import head, mlp_hip, optim
    pairs = {"rs_row_operand": mlp_hip.RowOperand, "rs_mlp_epilogue": mlp_hip.Epilogue, "rs_pack_weights_args": mlp_hip.PackArgs,
             "rs_umbrella_mlp": mlp_hip.UmbrellaMLPDesc, "rs_umbrella_mfma": mlp_hip.UmbrellaMFMADesc, "rs_bn_item": mlp_hip.BnItem, "rs_bn_bwd_item": mlp_hip.BnBwdItem,
             "rs_reduce_item": mlp_hip.ReduceItem, "rs_backward_tail_work": mlp_hip.BackwardTail, "rs_head_layer": head.HeadLayer,
             "rs_head_layer_bwd": head.HeadLayerBwd, "rs_adam_table": optim.AdamTable}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "repsurf_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, fname, value = line.split()
        cls = pairs[cname]
        expect = ctypes.sizeof(cls) if fname == "size" else getattr(cls, fname).offset
        assert int(value) == expect, (cname, fname, int(value), expect)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in pairs.values())
