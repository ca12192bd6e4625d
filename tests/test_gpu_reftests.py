"""GPU suite: the reference's own unit tests (test/gvxm.cu, greduce.cu, gewiseadd.cu, gewisemult.cu,
gtrace.cu, gdensevector.cu, gsparsevector.cu, gdescriptor.cu, gbinaryop.cu) and its smoke /
micro-benchmark mains (gspmspv, gpush, gpull, g*bench), compiled UNCHANGED in the build container
against include/graphblas/graphblas.hpp (tools/build_reference_tests.sh) and run here.  Their
assertions (BOOST_ASSERT_LIST: exact element-wise comparison against the inline CPU loops and
literals of each test) abort the process on the first mismatch."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "refcheck", "tests")
DATA = os.path.join(ROOT, "tests", "golden", "data")


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    """The tests open data/small/<file>.mtx relative to the working directory."""
    d = tmp_path_factory.mktemp("refcwd")
    small = d / "data" / "small"
    small.mkdir(parents=True)
    for f in os.listdir(DATA):
        if f.endswith(".mtx") and not f.startswith("."):
            shutil.copy(os.path.join(DATA, f), small / f)
    return str(d)


def _run(exe, cwd, *args):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        pytest.skip("%s not built (needs the reference tree at build time)" % exe)
    out = subprocess.run([path] + list(args), capture_output=True, text=True, timeout=300, cwd=cwd)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    return out.stdout


# test file -> number of BOOST_*_TEST_CASEs it defines
UNIT = {"gvxm": 6, "greduce": 2, "gewiseadd": 7, "gewisemult": 9, "gdensevector": 3,
        "gsparsevector": 3, "gdescriptor": 2, "gbinaryop": 3}


@pytest.mark.parametrize("name", sorted(UNIT))
def test_reference_unit_test(name, workdir):
    out = _run(name + "_ref", workdir)
    assert "*** No errors detected" in out, out[-1500:]
    if UNIT[name] is not None:
        assert "(%d test cases)" % UNIT[name] in out, out[-300:]
    assert "INCORRECT" not in out


def test_reference_gtrace(workdir):
    """test/gtrace.cu: dup1 / dup2 (literal matrices, trace 91) pass.  dup3 expects 341 for
    chesapeake_trace.mtx = chesapeake's 170 symmetric off-diagonal entries plus all 39 diagonal ones:
    the loader (the reference's removeSelfloop, util.hpp:263-329, restated in the frontend header and
    in oracle/loader.py) drops every diagonal entry, leaving 340 stored entries of value 1, and
    trace(A A^T) is their number.  341 is not reachable from the reference's own sources as mounted
    (neither is 379 = self loops kept), so the case is recorded as a known stale expectation."""
    path = os.path.join(BIN, "gtrace_ref")
    if not os.path.exists(path):
        pytest.skip("gtrace_ref not built")
    out = subprocess.run([path], capture_output=True, text=True, timeout=300, cwd=workdir)
    assert "[  OK  ] dup1" in out.stdout and "[  OK  ] dup2" in out.stdout, out.stdout[-800:]
    assert "39 39 340" in out.stdout and "340 = 341" in out.stdout, out.stdout[-800:]


@pytest.mark.parametrize("name", ["gspmspv", "gpush", "gpull"])
@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_cc.mtx"])
def test_reference_smoke_main(name, graph, workdir):
    _run(name + "_ref", workdir, os.path.join(workdir, "data", "small", graph))     # the mains write a .bin cache beside it


@pytest.mark.parametrize("name", ["gpushbench", "gpullbench", "gspmvbench", "gspmspvbench"])
def test_reference_microbenchmark_main(name, workdir):
    """test/g{push,pull,spmv,spmspv}bench.cu: frontier-size sweeps printing `size, ms` lines."""
    out = _run(name + "_ref", workdir, "--niter", "1", os.path.join(workdir, "data", "small", "small.mtx"))
    assert "," in out
