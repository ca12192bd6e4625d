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
import shutil
import tempfile

# the mains write the reference's binary cache next to the graph they load (`.<file>.<ud|d>.<nosl|sl>.bin`,
# sparse_matrix.hpp:328-348): give them a scratch copy of the data files, not the fixtures themselves
_SRC = os.path.join(ROOT, "tests", "golden", "data")
DATA = tempfile.mkdtemp(prefix="grb_dropin_data_")
for _f in os.listdir(_SRC):
    if _f.endswith(".mtx") and not _f.startswith("."):
        shutil.copy(os.path.join(_SRC, _f), os.path.join(DATA, _f))


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


# ---- the other five mains of example/ (SURVEY.md 8(f)4).  --mxvmode 2: the mode in which the
# reference's mis / gc loops terminate (tests/test_oracle.py records why the others do not).
SYM = ["chesapeake.mtx", "test_mesh.mtx", "small.mtx", "test_bc.mtx", "test_mis.mtx"]


@pytest.mark.parametrize("graph", SYM)
def test_reference_gmis_main(graph):
    out = _run("gmis_ref", "--mxvmode", "2", "--niter", "1", "--timing", "0", "--directed", "2", "--source", "3",
               os.path.join(DATA, graph))
    assert "INCORRECT" not in out and "errors occurred" not in out, out[-1500:]
    assert out.count("CORRECT") >= 3, out[-1500:]              # CPU result, warm-up result, timed result


@pytest.mark.parametrize("graph", SYM)
@pytest.mark.parametrize("algo", ["0", "1", "2"])
def test_reference_ggc_main(graph, algo):
    out = _run("ggc_ref", "--mxvmode", "2", "--niter", "1", "--timing", "0", "--directed", "2", "--gcalgo", algo,
               "--maxcolors", "128", "--seed", "1", os.path.join(DATA, graph))
    assert "INCORRECT" not in out and "errors occurred" not in out, out[-1500:]
    assert out.count("CORRECT") >= 3, out[-1500:]


@pytest.mark.parametrize("graph", SYM)
def test_reference_ggc_cusparse_main(graph):
    out = _run("ggc_cusparse_ref", "--niter", "1", "--timing", "0", "--directed", "2", os.path.join(DATA, graph))
    assert "INCORRECT" not in out and "errors occurred" not in out, out[-1500:]
    assert out.count("CORRECT") >= 3, out[-1500:]


@pytest.mark.parametrize("graph", ["chesapeake.mtx", "test_mesh.mtx", "small.mtx"])
@pytest.mark.parametrize("max_niter", ["5", "30"])
def test_reference_glgc_main(graph, max_niter):
    """VERIFY_LIST_FLOAT against the reference's own SimpleReferenceLgc (example/glgc.cu:98-101)."""
    out = _run("glgc_ref", "--mxvmode", "2", "--niter", "1", "--max_niter", max_niter, "--timing", "0",
               "--directed", "2", os.path.join(DATA, graph))
    assert "INCORRECT" not in out, out[-1500:]
    assert out.count("CORRECT") >= 2, out[-1500:]


@pytest.mark.parametrize("graph,want", [("chesapeake.mtx", None), ("test_mesh.mtx", None)])
@pytest.mark.parametrize("mode", ["0", "2"])
def test_reference_gdiameter_main(graph, want, mode):
    import numpy as np
    from oracle import loader, simple_reference as sr
    r, c, v, nr, nc, nv = loader.read_mtx(os.path.join(DATA, graph), 2, np.float32)
    ptr, ind, _ = loader.coo2csr(r, c, v, nr, nc)
    out = _run("gdiameter_ref", "--mxvmode", mode, "--directed", "2", "--source_start", "0", "--source_end", "1",
               os.path.join(DATA, graph))
    depth, _, _ = sr.bfs(ptr, ind, 0)
    assert "diameter 0:1: %d from 0" % (int(depth.max()) - 1) in out, out[-800:]


# ---- the same mains on a graph large enough for the direction heuristic to matter: RMAT-15, written
# out as a MatrixMarket file (the reference's data/small graphs never leave the dense representation)
@pytest.fixture(scope="module")
def rmat_mtx(tmp_path_factory):
    import numpy as np
    from graphblast_amd.graphgen import rmat_edges
    s, d, n = rmat_edges(15, 12, seed=5)
    keep = s != d
    lo, hi = np.minimum(s[keep], d[keep]), np.maximum(s[keep], d[keep])
    key = np.unique(hi.astype(np.int64) * n + lo)
    path = tmp_path_factory.mktemp("rmat") / "rmat15.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n")
        f.write("%d %d %d\n" % (n, n, key.size))
        np.savetxt(f, np.stack([key // n + 1, key % n + 1], 1), fmt="%d")
    return str(path)


@pytest.mark.parametrize("exe,args,min_correct", [
    ("gbfs_ref", ["--mxvmode", "0", "--struconly", "1", "--opreuse", "1", "--source", "3"], 2),
    ("gbfs_ref", ["--mxvmode", "0", "--source", "17"], 2),
    ("gbfs_ref", ["--mxvmode", "1", "--source", "3"], 2),
    ("gbfs_ref", ["--mxvmode", "2", "--source", "3"], 2),
    ("gsssp_ref", ["--mxvmode", "0", "--source", "3"], 2),
    ("gsssp_ref", ["--mxvmode", "1", "--source", "3"], 2),
    ("gpr_ref", ["--mxvmode", "2", "--max_niter", "10"], 0),
    ("gcc_ref", ["--mxvmode", "0"], 1),
    ("gcc_ref", ["--mxvmode", "2"], 1),
    ("gtc_ref", [], 1),
    ("gmis_ref", ["--mxvmode", "2", "--source", "2"], 3),
    ("ggc_ref", ["--mxvmode", "2", "--gcalgo", "0", "--maxcolors", "4096", "--seed", "1"], 3),
    ("ggc_ref", ["--mxvmode", "2", "--gcalgo", "2", "--maxcolors", "4096", "--seed", "1"], 3),
    ("ggc_cusparse_ref", [], 3),
    ("glgc_ref", ["--mxvmode", "2", "--max_niter", "5"], 2),
])
def test_reference_mains_on_rmat15(rmat_mtx, exe, args, min_correct):
    out = _run(exe, *args, "--niter", "1", "--timing", "0", rmat_mtx)
    assert "INCORRECT" not in out and "errors occurred" not in out, out[-1500:]
    assert out.count("CORRECT") >= min_correct, out[-1500:]


def test_c_abi_example_runs(tmp_path):
    """examples/bfs_c_abi.c (plain C99 against include/grb_hip.h) on the GPU: both BFS entry points,
    labels checked by the program itself."""
    out = str(tmp_path / "bfs_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "bfs_c_abi.c"),
                           "-L" + os.path.join(ROOT, "graphblast_amd"), "-lgrb_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "graphblast_amd"), "-o", out])
    res = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "CORRECT" in res.stdout, res.stdout + res.stderr
    assert "gfx950" in res.stdout


# ---- the drop-in path is the fast path.  The mains above include
# "graphblas/algorithm/{bfs,sssp,pr}.hpp", which resolve to the shadows in include/graphblas/algorithm/:
# the library's one-launch drivers by default, the reference's own text with GRB_FRONTEND_FUSED=0.
def _run_env(exe, env, *args):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        pytest.skip("build/refcheck/%s not built" % exe)
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([path] + list(args), capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("exe,args,min_correct", [
    ("gbfs_ref", ["--mxvmode", "0", "--struconly", "1", "--opreuse", "1", "--source", "3"], 2),
    ("gbfs_ref", ["--mxvmode", "1", "--source", "17"], 2),
    ("gsssp_ref", ["--mxvmode", "0", "--source", "3"], 2),
    ("gpr_ref", ["--mxvmode", "2", "--max_niter", "10"], 0),
])
def test_reference_mains_both_driver_paths_and_timing_lines(rmat_mtx, exe, args, min_correct):
    """Same verdicts through the fused drivers and through the reference's own loop text, and under the
    reference's default --timing 1 the fused path prints the same per-iteration lines (same iteration numbers,
    frontier sizes / residuals and directions; the millisecond column differs, of course)."""
    import re
    outs = {}
    for fused in ("1", "0"):
        out = _run_env(exe, {"GRB_FRONTEND_FUSED": fused}, *args, "--niter", "1", "--timing", "1", rmat_mtx)
        assert "INCORRECT" not in out, out[-1500:]
        assert out.count("CORRECT") >= min_correct, out[-1500:]
        rows = []
        for line in out.splitlines():
            m = re.match(r"^(\d+), ([-+0-9.e]+)/(\d+), (?:(\d+), )?(push|pull), [-+0-9.e]+$", line.strip())
            if m:
                rows.append((int(m.group(1)), float(m.group(2)), m.group(4), m.group(5)))
        outs[fused] = rows
    assert len(outs["1"]) >= 2 and len(outs["1"]) == len(outs["0"]), (outs["1"][:12], outs["0"][:12])
    for a, b in zip(outs["1"], outs["0"]):
        assert a[0] == b[0] and a[2] == b[2], (a, b)
        assert abs(a[1] - b[1]) <= 1e-4 * max(1.0, abs(b[1])), (a, b)
        if exe != "gpr_ref":
            assert a[3] == b[3], (a, b)           # pr: the fused loop always pulls; the printed mode is lastmxv_


def test_reference_gtc_main_three_ways(rmat_mtx):
    """example/gtc.cu, unchanged, on RMAT-15: through the shadow of algorithm/tc.hpp with the library counting on the
    degree-ordered orientation (no product in the buffer matrix), with GRB_TC_PRODUCT=1 (the reference's two calls inside
    the library) and with GRB_FRONTEND_FUSED=0 (the reference's own text, op by op) -- CORRECT against the reference's CPU
    count each time, and the one line tc.hpp prints under --timing 1 each time."""
    import re
    for env in ({}, {"GRB_TC_PRODUCT": "1"}, {"GRB_FRONTEND_FUSED": "0"}):
        out = _run_env("gtc_ref", env, "--niter", "1", "--timing", "1", rmat_mtx)
        assert "INCORRECT" not in out, (env, out[-1500:])
        assert out.count("CORRECT") >= 1, (env, out[-1500:])
        rows = [ln for ln in out.splitlines() if re.match(r"^0, 1/\d+, (push|pull), [-+0-9.e]+$", ln.strip())]
        assert len(rows) >= 2, (env, out[-1500:])          # the warm-up call and the timed one


def test_reference_gbfs_main_at_scale_is_the_fast_path(tmp_path):
    """example/gbfs.cu, unchanged, on RMAT-20 (n = 1 Mi, ~31 M edges) handed over through the reference's own
    binary cache (a .mtx stub with banner + size line next to `.stub.mtx.ud.nosl.bin`, which readMtx finds
    and Matrix::build reads, util.hpp:398-409): prints CORRECT twice (against the reference's CPU BFS) and its
    `tight` time per traversal is within 1.3x of the C-ABI driver's on the same graph and source."""
    import re
    import numpy as np
    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    from oracle import loader
    s, d, n = rmat_edges(20, 16, seed=1)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    src = int(np.argmax(np.diff(ptr)))
    stub = tmp_path / "stub.mtx"
    stub.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n" + "%d %d %d\n" % (n, n, ind.size // 2))
    loader.write_cache(str(tmp_path / ".stub.mtx.ud.nosl.bin"), ptr, ind)
    out = _run_env("gbfs_ref", {}, "--mxvmode", "0", "--struconly", "1", "--opreuse", "1", "--earlyexit", "1",
                   "--source", str(src), "--niter", "20", "--timing", "0", str(stub))
    assert "Reading" in out and "INCORRECT" not in out and out.count("CORRECT") >= 2, out[-1500:]
    tight = float(re.search(r"^tight, ([0-9.e+-]+)", out, re.M).group(1))
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, np.ones(ind.size, np.float32)) == 0
    desc = g.Descriptor()
    assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
    v = g.Vector(n)
    for _ in range(3):
        g.bfs(v, A, src, desc, fused=True)
    mine = np.mean([g.bfs(v, A, src, desc, fused=True)[1]["tight_ms"] for _ in range(20)])
    assert tight <= 1.3 * mine + 0.01, (tight, mine)
    slow = _run_env("gbfs_ref", {"GRB_FRONTEND_FUSED": "0"}, "--mxvmode", "0", "--struconly", "1", "--opreuse", "1",
                    "--source", str(src), "--niter", "5", "--timing", "0", str(stub))
    assert "INCORRECT" not in slow
    t_slow = float(re.search(r"^tight, ([0-9.e+-]+)", slow, re.M).group(1))
    print("gbfs.cu tight per traversal: fused %.3f ms, call sequence %.3f ms, C ABI %.3f ms" % (tight, t_slow, mine))
