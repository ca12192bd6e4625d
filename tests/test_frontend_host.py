"""CPU suite: host-side pieces of the drop-in boundary that need no GPU -- the C++ frontend's readMtx
(include/graphblas/graphblas.hpp) against the reference's own readMtx (tests/golden/algo_ref.npz), and the
binary cache's name rule (grb_cache_name) against util.hpp:340-357."""
import os
import subprocess

import numpy as np
import pytest

from backends import GOLDEN
from test_oracle_pinned import cases_of, mtx_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "algo_ref.npz"))


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fe") / "readmtx_dump")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "tools", "readmtx_dump.cpp"),
                           "-L" + os.path.join(ROOT, "graphblast_amd"), "-lgrb_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "graphblast_amd"), "-o", out])
    return out


def run_dump(exe, path, directed, name=False):
    out = subprocess.check_output([exe, path, str(directed)] + (["name"] if name else [])).decode().splitlines()
    out = [l for l in out if l and l[0].isdigit() or l.startswith("/") or l.startswith(".")]
    head = [int(x) for x in out[0].split()]
    body = out[2:] if name else out[1:]
    arr = np.array([l.split() for l in body], dtype=np.float64).reshape(-1, 3)
    return head, (out[1] if name else None), arr[:, 0].astype(np.int32), arr[:, 1].astype(np.int32), arr[:, 2].astype(np.float32)


def test_frontend_readmtx_equals_the_references(fx, dump_exe, tmp_path):
    """Same coordinate lists and the same VALUE array as the reference's readMtx -- including the case where
    removed entries leave the values uncompacted (util.hpp:311-323; the weighted synthetic inputs)."""
    n_weighted = 0
    for case in cases_of(fx):
        path, directed = mtx_path(fx, case, tmp_path)
        head, _, r, c, v = run_dump(dump_exe, path, directed)
        ptr, ind, val = fx[case + "/csr_ptr"], fx[case + "/csr_ind"], fx[case + "/csr_val"]
        assert head == [int(fx[case + "/nrows"]), int(fx[case + "/ncols"]), int(fx[case + "/nvals"]), ind.size], case
        rows = np.repeat(np.arange(ptr.size - 1, dtype=np.int32), np.diff(ptr))
        assert np.array_equal(r, rows) and np.array_equal(c, ind), case
        assert np.array_equal(v, val), case                # coo2csr of a sorted list keeps the value order
        n_weighted += int(np.unique(val).size > 1)
    assert n_weighted >= 4


def test_cache_name_rule(dump_exe, tmp_path):
    import graphblast_amd as g
    assert g.cache_name("/data/graphs/soc-LiveJournal1.mtx", False) == "/data/graphs/.soc-LiveJournal1.mtx.d.nosl.bin"
    assert g.cache_name("/data/graphs/road_usa.mtx", True) == "/data/graphs/.road_usa.mtx.ud.nosl.bin"
    assert g.cache_name("x.mtx", True) == "./.x.mtx.ud.nosl.bin"
    os.environ["GRB_UTIL_REMOVE_SELFLOOP"] = "0"
    try:
        assert g.cache_name("/a/b.mtx", True) == "/a/.b.mtx.ud.sl.bin"
    finally:
        del os.environ["GRB_UTIL_REMOVE_SELFLOOP"]
    from oracle import ref_simple as rs, loader
    for p, u in (("/data/g/a.mtx", True), ("rel/dir/b.mtx", False), ("c.mtx", True)):
        assert g.cache_name(p, u) == loader.cache_name(p, u)
        if rs.available():
            assert g.cache_name(p, u) == rs.cache_name(p, u)
    # the frontend's readMtx hands the same name out through dat_name
    p = tmp_path / "t.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 2\n2 1\n3 2\n")
    head, name, r, c, v = run_dump(dump_exe, str(p), 0, name=True)
    assert name == str(tmp_path / ".t.mtx.ud.nosl.bin")
    assert head == [3, 3, 4, 4]
    # ... and returns empty lists once that file exists (util.hpp:398-409)
    (tmp_path / ".t.mtx.ud.nosl.bin").write_bytes(b"\0" * 8)
    head, name, r, c, v = run_dump(dump_exe, str(p), 0, name=True)
    assert head[3] == 0 and r.size == 0


def test_register_semiring_macros(tmp_path):
    """REGISTER_MONOID / REGISTER_SEMIRING of stddef.hpp:140-191 work against the drop-in header: the functors
    evaluate on the host like the reference's, and each composition resolves to a C-ABI id >= 64 (the same id on
    every use)."""
    out = str(tmp_path / "user_semiring")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "tools", "user_semiring.cpp"),
                           "-L" + os.path.join(ROOT, "graphblast_amd"), "-lgrb_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "graphblast_amd"), "-o", out])
    lines = subprocess.check_output([out]).decode().split("\n")
    ids = [int(x) for x in lines[0].split()]
    assert ids[0] == 1 and ids[1] >= 64 and ids[2] >= 64 and ids[3] >= 64 and len(set(ids[1:])) == 3
    assert [float(x) for x in lines[1].split()] == [0.0, 5.0, 7.0, -1000.0, 12.0]
    assert int(lines[2]) == ids[1]
