"""GPU suite: the reference's own applications (example/gbfs.cu, gsssp.cu, gpr.cu -- their
main(), their algorithm/*.hpp loops and their CPU verification), compiled UNCHANGED in the
build container against include/graphblas/graphblas.hpp (tools/build_reference_examples.sh),
run here on the reference's data/small graphs.  Each prints CORRECT / INCORRECT itself
(test/test.hpp:60-114)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "refcheck")
DATA = os.path.join(ROOT, "tests", "golden", "data")


def _run(exe, *args):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        pytest.skip("build/refcheck/%s not built (needs the reference tree at build time)" % exe)
    out = subprocess.run([path] + list(args), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_cc.mtx", "test_bc.mtx", "test_mesh.mtx", "small.mtx"])
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_reference_gbfs_main(graph, mode):
    for extra in ([], ["--struconly", "1", "--opreuse", "1"]):
        out = _run("gbfs_ref", "--mxvmode", mode, "--niter", "2", "--timing", "0", *extra, os.path.join(DATA, graph))
        assert "INCORRECT" not in out, out[-1500:]
        assert out.count("CORRECT") >= 2, out[-1500:]


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_cc.mtx", "test_mesh.mtx"])
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_reference_gsssp_main(graph, mode):
    out = _run("gsssp_ref", "--mxvmode", mode, "--niter", "2", "--timing", "0", os.path.join(DATA, graph))
    assert "INCORRECT" not in out, out[-1500:]
    assert out.count("CORRECT") >= 2, out[-1500:]


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_pr.mtx"])
def test_reference_gpr_main(graph):
    out = _run("gpr_ref", "--mxvmode", "2", "--niter", "1", "--max_niter", "10", "--timing", "0",
               os.path.join(DATA, graph))
    assert "INCORRECT" not in out, out[-1500:]


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_mesh.mtx", "small.mtx"])
@pytest.mark.parametrize("mode", ["1", "2"])
def test_reference_gcc_main(graph, mode):
    out = _run("gcc_ref", "--mxvmode", mode, "--niter", "1", "--timing", "0", os.path.join(DATA, graph))
    assert "INCORRECT" not in out, out[-1500:]
    assert "CORRECT" in out, out[-1500:]


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_mesh.mtx", "small.mtx"])
def test_reference_gtc_main(graph):
    out = _run("gtc_ref", "--niter", "1", "--timing", "0", os.path.join(DATA, graph))
    assert "INCORRECT" not in out, out[-1500:]
    assert "CORRECT" in out, out[-1500:]
    if graph == "chesapeake.mtx":
        assert "194" in out
