"""CPU tier: the C-ABI library loads and exports exactly what include/cmr_hip.h declares
(no compute calls without a GPU), and the product never routes through the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cmr_agent_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", ROOT, "-j4", "all"], check=True)
    return _lib


def test_header_symbols_exported(lib):
    protos = lib.parse_header()
    assert len(protos) >= 30
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    lib.load()


def test_library_exports_only_declared_entry_points(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (cmr_\w+)", out))
    assert exported == set(lib.parse_header()), exported ^ set(lib.parse_header())


def test_workspace_queries_run_on_cpu(lib):
    l = lib.load()
    assert l.cmr_la_reduce_workspace_bytes(2, 1000) == 2 * 2 * 576 * 4
    assert l.cmr_colreduce_workspace_bytes(2, 1000, 64) == 2 * 4 * 64 * 4


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cmr_agent_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(d, f)
                assert "cmr_oracle" not in src, os.path.join(d, f)
