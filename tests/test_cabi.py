"""The C-ABI library loads and exports every symbol include/ggnn.h declares; host-side
argument validation works without a GPU (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from graingraphnn_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ggnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ggnn_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in header_symbols():
        assert hasattr(lib, sym), sym


def test_version_and_host_only_queries():
    lib = _lib.load()
    assert lib.ggnn_version() == _lib.GGNN_ABI_VERSION == 25
    assert lib.ggnn_error_string(0) == b"ok"
    assert b"invalid" in lib.ggnn_error_string(-1)
    assert lib.ggnn_csr_workspace_bytes(60000, 20000) == (2 * 20000 + 2) * 4
    # decoder-sized scratch of one model forward at cfg3 (DESIGN.md "data layout")
    floats = 20000 * 2688 + 10000 * 1536 + 20000 * 784 + 10000 * 400 + 4 * 30000 * 96 + 20000 * 8 + 3 * 60000
    assert lib.ggnn_workspace_bytes(10000, 20000, 60000) == 4 * floats


def test_argument_validation_returns_einval_without_launching():
    lib = _lib.load()
    assert lib.ggnn_project(None, 8, 8, None, 0, 0, None, None, 10, 96, None, 96, None) == -1
    assert lib.ggnn_period_gat_aggregate(None, None) == -1
    assert lib.ggnn_lstm_epilogue(None, None) == -1
    a = _lib.AggregateArgs()
    assert lib.ggnn_period_gat_aggregate(ctypes.byref(a), None) == -1
    e = _lib.EpilogueArgs()
    assert lib.ggnn_lstm_epilogue(ctypes.byref(e), None) == -1
    assert lib.ggnn_period_gat_aggregate_batch(None, 1, None) == -1
    assert lib.ggnn_period_gat_aggregate_batch((_lib.AggregateArgs * 3)(), 4, None) == -1
    assert lib.ggnn_period_gat_aggregate_backward(None, None) == -1
    b = _lib.AggregateBwdArgs()
    assert lib.ggnn_period_gat_aggregate_backward(ctypes.byref(b), None) == -1
    assert lib.ggnn_aggregate_bwd_partials(20000) == 768 and lib.ggnn_aggregate_bwd_partials(5) == 2
    assert lib.ggnn_build_csr(None, 5, 3, 0, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.ggnn_csr_max_units(60000, 20000) == 40001
    assert lib.ggnn_edge_prepare(None, 1, None) == -1
    assert lib.ggnn_heads_regressor(None, 1, None, 1, None, 11, None, None, None, None, None, None) == -1
    assert lib.ggnn_heads_classifier(None, 1, None, 0, None, None, None, None, None, None, None) == -1
    assert lib.ggnn_step_update(None, 1, 8, None, 1, 11, 11, None, None, 0.0, 1.0, None, None) == -1
    assert lib.ggnn_step_refresh(None, 1, 8, None, 1, 11, 1.0, None, None, 0, None) == -1
    with pytest.raises(_lib.GGNNError):
        _lib.check(-1, "demo")


def test_struct_sizes_match_the_header():
    """ctypes mirrors of the POD argument blocks (natural alignment, no packing)."""
    assert ctypes.sizeof(_lib.AggregateArgs) == 8 * 8 + 7 * 8 + 8 * 4
    assert ctypes.sizeof(_lib.AggregateBwdArgs) == 18 * 8 + 8 * 8 + 8 * 4
    assert ctypes.sizeof(_lib.PrepareEdge) == 7 * 8 + 4 * 8 + 8     # (+ E_dev, ABI 25)
    assert ctypes.sizeof(_lib.EpilogueArgs) == 7 * 8 + 2 * 8 + 4 * 4 + 8 + 8 + 2 * 4
    assert ctypes.sizeof(_lib.RefreshEdge) == 4 * 8 + 5 * 8 + 8    # (+ E_dev, ABI 25)
