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


def test_enumerations_equal_the_references_types_hpp(tmp_path):
    """Info / Storage / Desc_field / Desc_value numbering in all four places it is written down -- the C ABI
    header, the C++ frontend header, the Python mirror, the oracle -- against the values printed by the
    reference's own graphblas/types.hpp (oracle/_ref/types_ref -> tests/golden/types_ref.json)."""
    import json
    import subprocess
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "types_ref.json")))
    assert ref["sizeof_Index"] == 4 and ref["sizeof_T"] == 4
    names = [k for k in ref if k.startswith("GrB_")]
    # Python mirror and oracle
    import graphblast_amd as g
    from oracle import ops
    for k in names:
        if hasattr(g, k):
            assert getattr(g, k) == ref[k], ("api.py", k)
        if hasattr(ops, k):
            assert getattr(ops, k) == ref[k], ("oracle/ops.py", k)
    assert all(hasattr(g, k) for k in ("GrB_SUCCESS", "GrB_PANIC", "GrB_MASK", "GrB_MXVMODE", "GrB_SCMP", "GrB_PULLONLY"))
    # C ABI header: GRB_X <-> GrB_X (GRB_HIP takes the slot of GrB_CUDA)
    src = tmp_path / "enums.c"
    lines = ['#include <stdio.h>', '#include "grb_hip.h"', 'int main(void) {']
    cnames = {k: "GRB_" + k[4:] for k in names}
    cnames["GrB_CUDA"] = "GRB_HIP"
    hdr = open(os.path.join(ROOT, "include", "grb_hip.h")).read()
    checked = 0
    for k, c in cnames.items():
        if re.search(r"\b%s\b" % c, hdr):
            lines.append('  printf("%s %%d\\n", (int)%s);' % (k, c))
            checked += 1
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "enums")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    for ln in subprocess.check_output([exe], text=True).split("\n"):
        if ln:
            k, v = ln.split()
            assert int(v) == ref[k], ("grb_hip.h", k)
    assert checked >= 35
    # C++ frontend header
    cpp = tmp_path / "enums.cpp"
    body = "\n".join('  std::printf("%s %%d\\n", static_cast<int>(graphblas::%s));' % (k, k) for k in names)
    cpp.write_text('#include <cstdio>\n#include "graphblas/graphblas.hpp"\nint main() {\n%s\n  return 0;\n}\n' % body)
    exe2 = str(tmp_path / "enums_cpp")
    subprocess.check_call(["g++", "-std=c++11", "-w", "-fpermissive", "-I" + os.path.join(ROOT, "include"), str(cpp),
                           "-L" + os.path.join(ROOT, "graphblast_amd"), "-lgrb_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "graphblast_amd"), "-o", exe2])
    for ln in subprocess.check_output([exe2], text=True).split("\n"):
        if ln:
            k, v = ln.split()
            assert int(v) == ref[k], ("graphblas.hpp", k)


def test_every_entry_point_flushes_the_lazy_queue():
    """csrc/lazy.hip defers element-wise calls; that is only invisible if EVERY entry point of the C ABI starts by
    flushing the queue (GRB_API_ENTER / _NOINFO) or is one of the few that may append to it (GRB_API_ENTER_QUEUE, which
    flush themselves before they touch data; GRB_API_ENTER_HOST: descriptor fields only, nothing to see).  A new entry point without the macro would read stale vectors: this test
    is the guard.  Source-level check, no GPU."""
    import glob
    names = set(_declared_symbols())
    found = {}
    for path in glob.glob(os.path.join(ROOT, "graphblast_amd", "csrc", "*.hip")):
        src = open(path).read()
        for m in re.finditer(r'^(?:extern "C" )?([A-Za-z_][\w \*]*?)\s*\b(grb_[A-Za-z0-9_]+)\s*\(', src, re.M):
            ret, name = m.group(1).strip(), m.group(2)
            if name not in names or ret.startswith("static") or "return" in ret or "=" in ret:
                continue
            i, depth = m.end(), 1
            while depth and i < len(src):
                depth += (src[i] == "(") - (src[i] == ")")
                i += 1
            mm = re.match(r"\s*\{", src[i:i + 40])
            if not mm:
                continue                                   # a declaration or a call, not the definition
            body = src[i + mm.end(): i + mm.end() + 400]
            found[name] = bool(re.match(r"\s*(GRB_API_ENTER(_NOINFO|_QUEUE|_HOST|_BFSQ)?\(\)|grb::ApiScope api_scope__)", body)) \
                or name in ("grb_lazy_pending", "grb_lazy_fused_reductions")   # report the queue; must not flush it
    assert set(found) == names, sorted(names - set(found))
    assert all(found.values()), sorted(n for n, ok in found.items() if not ok)
