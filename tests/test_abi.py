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
    with pytest.raises(_lib.RepSurfHipError, match="exceeds"):
        _lib.call("rs_knnquery", 1, 100, 1, 65, 1, 1, 1, None, None)
    # zero-sized problems are valid no-ops
    _lib.call("rs_ballquery", 0, 0, 0, 0.1, 0, None, None, None, None, None)
