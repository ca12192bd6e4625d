"""CPU suite: the C-ABI library loads and exports every symbol include/grb_hip.h
declares; host-only logic (Descriptor) behaves like the reference's.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "grb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grb_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from graphblast_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 60
    for n in names:
        assert hasattr(lib, n), "libgrb_hip.so does not export %s" % n
    # and the binding declares a signature for every one of them
    bound = set(_lib._SIGS) | {"grb_version", "grb_k_spmv_bytes"}
    assert set(names) <= bound, sorted(set(names) - bound)
    assert b"gfx950" in lib.grb_version()


def test_enum_order_matches_oracle_tables():
    """grb_semiring / grb_monoid numbering == graphblas/stddef.hpp order == oracle tables."""
    from graphblast_amd import api
    from oracle import semiring
    assert api.SEMIRINGS == list(semiring.SEMIRINGS)
    assert api.MONOIDS == list(semiring.MONOIDS)
    hdr = open(os.path.join(ROOT, "include", "grb_hip.h")).read()
    body = re.search(r"typedef enum \{([^{}]*?)\} grb_semiring;", hdr, re.S).group(1)
    names = [x.strip().split("=")[0].strip() for x in body.replace("\n", " ").split(",") if x.strip()]
    want = ["GRB_" + re.sub(r"(?<!^)(?=[A-Z])", "_", s).upper() for s in api.SEMIRINGS] + ["GRB_N_SEMIRINGS"]
    assert names == want


def test_descriptor_defaults_set_get_toggle():
    """test/gdescriptor.cu:90-112 behaviour + backend/cuda/descriptor.hpp:141-154 toggle."""
    import graphblast_amd as g
    d = g.Descriptor()
    assert [d.get(f) for f in range(4)] == [g.GrB_DEFAULT] * 4
    assert d.get(g.GrB_MXVMODE) == g.GrB_PUSHPULL and d.get(g.GrB_NT) == 128 and d.get(g.GrB_TOL) == 16
    assert d.set(g.GrB_MASK, g.GrB_SCMP) == 0 and d.get(g.GrB_MASK) == g.GrB_SCMP
    d.toggle(g.GrB_MASK)
    assert d.get(g.GrB_MASK) == g.GrB_DEFAULT
    d.toggle(g.GrB_MASK); d.toggle(g.GrB_OUTP); d.toggle(g.GrB_INP0); d.toggle(g.GrB_INP1)
    assert [d.get(f) for f in range(4)] == [g.GrB_SCMP, g.GrB_REPLACE, g.GrB_TRAN, g.GrB_TRAN]
    d.toggle(g.GrB_MODE)                                   # fields >= 4 are not toggled
    assert d.get(g.GrB_MODE) == 6
    assert d.set(99, 0) == g.GrB_INVALID_VALUE


def test_descriptor_loadargs_matches_oracle():
    import graphblast_amd as g
    from oracle import ops
    d = g.Descriptor()
    assert d.arg("earlyexit") == 0 and d.arg("switchpoint") == 0       # default-constructed
    assert d.loadArgs() == 0
    o = ops.Descriptor(); o.loadArgs()
    assert d.get(g.GrB_MXVMODE) == o.get(ops.GrB_MXVMODE) == g.GrB_PUSHONLY
    assert (d.arg("earlyexit"), d.arg("fusedmask"), d.arg("sort"), d.arg("max_niter")) == (1, 1, 1, 10000)
    assert d.arg("switchpoint") == pytest.approx(0.01)
    for mode, want in ((0, g.GrB_PUSHPULL), (1, g.GrB_PUSHONLY), (2, g.GrB_PULLONLY)):
        assert d.loadArgs(mxvmode=mode, struconly=1) == 0
        assert d.get(g.GrB_MXVMODE) == want and d.arg("struconly") == 1
    assert d.loadArgs(mxvmode=7) == g.GrB_INVALID_VALUE
    assert d.loadArgs(nthread=100) == g.GrB_INVALID_VALUE
    assert d.lastmxv_ == g.GrB_PUSHONLY


def test_no_cpu_fallback_in_product():
    """The product package must not import the oracle, and containers fail loudly
    without a device."""
    import subprocess, sys
    code = ("import sys; import graphblast_amd; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; "
            "assert not bad, bad; print('clean')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "clean" in out.stdout, out.stderr
    for f in os.listdir(os.path.join(ROOT, "graphblast_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "graphblast_amd", f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f
    # the oracle is reachable from tests/ (incl. tests/tools/), smoke() and bench.py's CPU-baseline leg only
    for d in ("tools", "examples", os.path.join("graphblast_amd", "csrc")):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".py", ".sh", ".hip", ".hpp", ".c")):
                src = open(os.path.join(ROOT, d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, (d, f)


def test_header_is_plain_c_and_the_example_links(tmp_path):
    """include/grb_hip.h is the boundary other hosts bind (cgo / JNI / ctypes): it must be valid C99,
    and examples/bfs_c_abi.c must compile against it and link to libgrb_hip.so with gcc alone."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "bfs_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "bfs_c_abi.c"), "-L" + os.path.join(root, "graphblast_amd"),
                           "-lgrb_hip", "-Wl,-rpath," + os.path.join(root, "graphblast_amd"), "-o", out])
    assert os.path.exists(out)
