"""The SpMV's column-sorted band format (csrc/spmv_cband.hpp) against the textbook definition on the host: exactly
for integer-valued data and the idempotent monoids, within north_star's 1e-5 for float sums (a row is summed in
column-rank order by atomics).  Power-law matrices take the format by themselves; small, rectangular and
road-like ones are forced into it (grb_spmv_set_format(2)) so that one band / many column blocks / no hub band /
natural column order are all walked."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module", autouse=True)
def format_at_first_product():
    """`auto` normally waits for a few dozen CSR-kernel products before it prepares the format (the amortisation rule,
    grb_spmv_set_reuse_threshold); this module wants it at the first one"""
    import graphblast_amd as g
    before = g.spmv_set_reuse_threshold(0)
    yield
    g.spmv_set_reuse_threshold(before)


@pytest.fixture(scope="module")
def forced():
    import graphblast_amd as g
    before = g.spmv_set_format(-1)
    g.spmv_set_format(2)
    yield g
    g.spmv_set_format(before)


def _matrix(g, torch, dev, nrows, ncols, ptr, ind, vals):
    tp = torch.from_numpy(np.ascontiguousarray(ptr, dtype=np.int32)).to(dev)
    ti = torch.from_numpy(np.ascontiguousarray(ind, dtype=np.int32)).to(dev)
    tv = torch.from_numpy(np.ascontiguousarray(vals)).to(dev)
    A = g.Matrix(nrows, ncols, dtype=vals.dtype.type)
    assert A.build_device_csr(tp.data_ptr(), ti.data_ptr(), tv.data_ptr(), int(ind.size), keep=(tp, ti, tv)) == 0
    return A


def _run(g, torch, dev, A, nrows, op, u, mask=None, scmp=0, accum=0, w0=None):
    tu = torch.from_numpy(u).to(dev)
    tw = torch.from_numpy(w0.copy()).to(dev) if w0 is not None else torch.full((nrows,), -7, dtype=tu.dtype, device=dev)
    tm = torch.from_numpy(mask).to(dev) if mask is not None else None
    assert g.k_spmv(A, 0, op, tu.data_ptr(), tm.data_ptr() if tm is not None else None, scmp, accum, tw.data_ptr()) == 0
    torch.cuda.synchronize()
    return tw.cpu().numpy()


def _reference(ptr, ind, vals, u, op, nrows):
    rows = np.repeat(np.arange(nrows), np.diff(ptr))
    a, x = vals, u[ind]
    nonempty = np.diff(ptr) > 0
    starts = ptr[:-1][nonempty]
    if op == "PlusMultiplies":
        return np.bincount(rows, weights=a.astype(np.float64) * x.astype(np.float64), minlength=nrows)
    if op == "MinimumPlus":
        prod, red, ident = a + x, np.minimum, FMAX if vals.dtype != np.int32 else np.iinfo(np.int32).max
    elif op == "MaximumMultiplies":
        prod, red, ident = a * x, np.maximum, 0
    elif op == "LogicalOrAnd":
        prod, red, ident = ((a != 0) & (x != 0)).astype(vals.dtype), np.maximum, 0
    elif op == "MinimumSelectSecond":
        prod, red, ident = x.copy(), np.minimum, FMAX if vals.dtype != np.int32 else np.iinfo(np.int32).max
    else:
        raise AssertionError(op)
    want = np.full(nrows, ident, dtype=vals.dtype)
    want[nonempty] = red.reduceat(prod.astype(vals.dtype), starts)
    return want


@pytest.fixture(scope="module")
def powerlaw():
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    dev = torch.device("cuda", 0)
    assert g.spmv_set_format(-1) == 1                      # the default: auto
    src, dst, n = rmat_edges(19, 16, seed=5, device=dev)
    gr = finalize_edges(src, dst, n, symmetrize=True)
    ptr, ind = gr["csr"][0].cpu().numpy(), gr["csr"][1].cpu().numpy()
    rng = np.random.default_rng(11)
    out = {"g": g, "torch": torch, "dev": dev, "n": n, "ptr": ptr, "ind": ind}
    for name, vals in (("int", rng.integers(1, 4, ind.size).astype(np.float32)),
                       ("f", rng.random(ind.size, dtype=np.float32) + 0.25),
                       ("iso", np.ones(ind.size, dtype=np.float32)),
                       ("i32", rng.integers(1, 9, ind.size).astype(np.int32))):
        out["A_" + name] = _matrix(g, torch, dev, n, n, ptr, ind, vals)
        out["v_" + name] = vals
    return out


def test_power_law_matrix_takes_the_format_by_itself(powerlaw):
    b = powerlaw
    g = b["g"]
    u = np.random.default_rng(1).integers(0, 3, b["n"]).astype(np.float32)
    got = _run(g, b["torch"], b["dev"], b["A_int"], b["n"], "PlusMultiplies", u)
    assert np.array_equal(got, _reference(b["ptr"], b["ind"], b["v_int"], u, "PlusMultiplies", b["n"]).astype(np.float32))
    info = g.spmv_format_info(b["A_int"], 0)
    assert info["in_use"] == 1 and info["hub_rows"] > 0 and info["bands"] >= 2 and info["iso"] == 0, info
    assert info["groups"] * 64 >= b["ind"].size and info["groups"] * 64 < 1.05 * b["ind"].size     # padding stays small
    # all values equal: the value array is not stored at all
    got = _run(g, b["torch"], b["dev"], b["A_iso"], b["n"], "PlusMultiplies", u)
    assert np.array_equal(got, _reference(b["ptr"], b["ind"], b["v_iso"], u, "PlusMultiplies", b["n"]).astype(np.float32))
    iso = g.spmv_format_info(b["A_iso"], 0)
    assert iso["iso"] == 1 and iso["bytes_per_launch"] < 0.8 * info["bytes_per_launch"], (iso, info)


def test_float_sums_within_tolerance(powerlaw):
    b = powerlaw
    u = np.random.default_rng(2).random(b["n"], dtype=np.float32)
    got = _run(b["g"], b["torch"], b["dev"], b["A_f"], b["n"], "PlusMultiplies", u)
    want = _reference(b["ptr"], b["ind"], b["v_f"], u, "PlusMultiplies", b["n"])
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("op", ["MinimumPlus", "MaximumMultiplies", "LogicalOrAnd", "MinimumSelectSecond"])
def test_idempotent_monoids_exact(powerlaw, op):
    b = powerlaw
    rng = np.random.default_rng(3)
    u = (rng.random(b["n"], dtype=np.float32) * 8).astype(np.float32)
    u[rng.random(b["n"]) < 0.3] = 0.0
    got = _run(b["g"], b["torch"], b["dev"], b["A_f"], b["n"], op, u)
    assert np.array_equal(got, _reference(b["ptr"], b["ind"], b["v_f"], u, op, b["n"]))


@pytest.mark.parametrize("op", ["PlusMultiplies", "MinimumPlus", "MinimumSelectSecond"])
def test_int32_exact(powerlaw, op):
    b = powerlaw
    u = np.random.default_rng(4).integers(0, 50, b["n"]).astype(np.int32)
    got = _run(b["g"], b["torch"], b["dev"], b["A_i32"], b["n"], op, u)
    want = _reference(b["ptr"], b["ind"], b["v_i32"], u, op, b["n"])
    assert np.array_equal(got, want.astype(np.int32))


@pytest.mark.parametrize("scmp,accum", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_mask_and_accumulate(powerlaw, scmp, accum):
    b = powerlaw
    rng = np.random.default_rng(5)
    u = rng.integers(0, 3, b["n"]).astype(np.float32)
    mask = (rng.random(b["n"]) < 0.5).astype(np.float32)
    w0 = rng.integers(0, 5, b["n"]).astype(np.float32)
    got = _run(b["g"], b["torch"], b["dev"], b["A_int"], b["n"], "PlusMultiplies", u, mask=mask, scmp=scmp, accum=accum, w0=w0)
    full = _reference(b["ptr"], b["ind"], b["v_int"], u, "PlusMultiplies", b["n"]).astype(np.float32)
    want = np.where((mask != 0) != bool(scmp), full, np.float32(0))
    if accum:
        want = w0 + want
    assert np.array_equal(got, want)


def test_comparison_monoids_keep_the_csr_kernel(powerlaw):
    """the comparison "monoids" of stddef.hpp depend on the order of a row's products: they stay on the CSR kernel,
    whose results the 17-semiring test of test_gpu_ops.py pins; here: same answer with the format on and off"""
    b = powerlaw
    g = b["g"]
    u = np.random.default_rng(6).integers(0, 4, b["n"]).astype(np.float32)
    on = _run(g, b["torch"], b["dev"], b["A_int"], b["n"], "GreaterPlus", u)
    before = g.spmv_set_format(-1)
    g.spmv_set_format(0)
    try:
        off = _run(g, b["torch"], b["dev"], b["A_int"], b["n"], "GreaterPlus", u)
        plus_off = _run(g, b["torch"], b["dev"], b["A_int"], b["n"], "PlusMultiplies", u)
    finally:
        g.spmv_set_format(before)
    assert np.array_equal(on, off)
    assert np.array_equal(plus_off, _reference(b["ptr"], b["ind"], b["v_int"], u, "PlusMultiplies", b["n"]).astype(np.float32))


def _random_csr(rng, nrows, ncols, avg, empty_frac=0.2, heavy=()):
    deg = rng.poisson(avg, nrows)
    deg[rng.random(nrows) < empty_frac] = 0
    for r, d in heavy:
        deg[r] = d
    deg = np.minimum(deg, ncols)
    ptr = np.zeros(nrows + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(deg)
    ind = np.concatenate([np.sort(rng.choice(ncols, d, replace=False)) for d in deg]) if ptr[-1] else np.zeros(0, np.int64)
    return ptr.astype(np.int32), ind.astype(np.int32)


@pytest.mark.parametrize("shape", [(1, 1), (37, 5), (300, 70001), (5000, 140000), (40000, 3000), (70000, 70000)])
def test_forced_format_on_small_and_rectangular_matrices(forced, shape):
    """one band / several bands without hubs / a hub band by row count alone / more than one 65536-column block /
    empty rows at both ends / a row holding every column"""
    import torch
    g = forced
    dev = torch.device("cuda", 0)
    nrows, ncols = shape
    rng = np.random.default_rng(nrows * 7 + ncols)
    heavy = [(nrows // 2, ncols)] if nrows > 30 else []
    if nrows > 35000:
        heavy += [(int(r), 200) for r in rng.choice(nrows, 50, replace=False) if r != nrows // 2]
    ptr, ind = _random_csr(rng, nrows, ncols, 6, heavy=heavy)
    if ind.size == 0:
        ptr, ind = np.array([0] + [1] * nrows, dtype=np.int32), np.zeros(1, dtype=np.int32)
    vals = rng.integers(1, 5, ind.size).astype(np.float32)
    A = _matrix(g, torch, dev, nrows, ncols, ptr, ind, vals)
    u = rng.integers(0, 4, ncols).astype(np.float32)
    for op in ("PlusMultiplies", "MinimumPlus", "LogicalOrAnd"):
        got = _run(g, torch, dev, A, nrows, op, u)
        assert np.array_equal(got, _reference(ptr, ind, vals, u, op, nrows).astype(np.float32)), (shape, op)
    info = g.spmv_format_info(A, 0)
    assert info["in_use"] == 1, info
    if nrows > 35000:
        assert info["hub_rows"] >= 1


def test_forced_format_on_a_road_like_grid(forced):
    import torch
    g = forced
    dev = torch.device("cuda", 0)
    side = 300
    idx = np.arange(side * side).reshape(side, side)
    e = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]),
                        np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])], axis=1)
    from graphblast_amd.graphgen import finalize_edges
    gr = finalize_edges(e[0].astype(np.int64), e[1].astype(np.int64), side * side, symmetrize=True)
    ptr, ind = gr["csr"]
    n = side * side
    rng = np.random.default_rng(9)
    vals = (rng.integers(1, 65, ind.size)).astype(np.float32)
    A = _matrix(g, torch, dev, n, n, ptr, ind, vals)
    u = rng.integers(0, 100, n).astype(np.float32)
    for op in ("PlusMultiplies", "MinimumPlus"):
        assert np.array_equal(_run(g, torch, dev, A, n, op, u), _reference(ptr, ind, vals, u, op, n).astype(np.float32))
    assert g.spmv_format_info(A, 0)["hub_rows"] == 0


def test_values_rewritten_in_place_are_seen_by_the_next_product():
    """The band formats keep a private copy of the stored values (or one iso value) from their first product;
    grb_matrix_set_values and matrix (x) scalar / vector rewrite csr.val / csc.val in place (objects.hip, mxm.hip
    matrix_scale) and must drop those copies -- the PageRank set-up (pattern matrix, then 1 / outdegree scaling,
    gpr.cu / pr.hpp) is exactly mxv -> scale -> mxv.  Both orientations, an iso and a valued start."""
    import torch
    import graphblast_amd as g
    from graphblast_amd import _lib
    from graphblast_amd.api import _semiring_id
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    dev = torch.device("cuda", 0)
    before = g.spmv_set_format(-1)                      # the module's `forced` fixture may still be in force
    g.spmv_set_format(1)
    try:
        _values_rewritten(g, torch, dev, _lib, _semiring_id, rmat_edges, finalize_edges)
    finally:
        g.spmv_set_format(before)


def _values_rewritten(g, torch, dev, _lib, _semiring_id, rmat_edges, finalize_edges):
    src, dst, n = rmat_edges(18, 16, seed=9, device=dev)
    gr = finalize_edges(src, dst, n, symmetrize=True)
    ptr, ind = gr["csr"][0].cpu().numpy(), gr["csr"][1].cpu().numpy()
    rng = np.random.default_rng(21)
    u = rng.integers(0, 3, n).astype(np.float32)
    lib = _lib.load()
    d = g.Descriptor()
    d.loadArgs()
    for start in ("iso", "valued"):
        vals = np.ones(ind.size, dtype=np.float32) if start == "iso" else rng.integers(1, 4, ind.size).astype(np.float32)
        A = g.Matrix(n, n)
        assert A.build_csr(ptr, ind, vals) == 0
        rows = np.repeat(np.arange(n), np.diff(ptr))

        def transposed(v):
            tu = torch.from_numpy(u).to(dev)
            tw = torch.empty(n, dtype=torch.float32, device=dev)
            assert g.k_spmv(A, 1, "PlusMultiplies", tu.data_ptr(), None, 0, 0, tw.data_ptr()) == 0
            torch.cuda.synchronize()
            want = np.bincount(ind, weights=v.astype(np.float64) * u[rows].astype(np.float64), minlength=n)
            assert np.array_equal(tw.cpu().numpy(), want.astype(np.float32)), start

        # first products: both orientations prepare their plans (and their copies of the values)
        got = _run(g, torch, dev, A, n, "PlusMultiplies", u)
        assert np.array_equal(got, _reference(ptr, ind, vals, u, "PlusMultiplies", n).astype(np.float32)), start
        transposed(vals)
        assert g.spmv_format_info(A, 0)["in_use"] == 1
        # 1. new values through set_values
        vals2 = rng.integers(1, 6, ind.size).astype(np.float32)
        assert A.set_values(vals2) == 0
        got = _run(g, torch, dev, A, n, "PlusMultiplies", u)
        assert np.array_equal(got, _reference(ptr, ind, vals2, u, "PlusMultiplies", n).astype(np.float32)), start
        # 2. scaled by a scalar in place
        assert lib.grb_matrix_eWiseMult_scalar(A._h, _semiring_id("PlusMultiplies"), A._h, 2.0) == 0
        got = _run(g, torch, dev, A, n, "PlusMultiplies", u)
        assert np.array_equal(got, _reference(ptr, ind, vals2 * 2, u, "PlusMultiplies", n).astype(np.float32)), start
        # 3. scaled row-wise by a vector in place (the PageRank set-up's shape)
        scale = rng.integers(1, 4, n).astype(np.float32)
        B = g.Vector(n)
        assert B.build(scale) == 0
        assert lib.grb_matrix_eWiseMult_vector(A._h, _semiring_id("PlusMultiplies"), A._h, B._h, d._h) == 0
        vals3 = vals2 * 2 * scale[rows]
        got = _run(g, torch, dev, A, n, "PlusMultiplies", u)
        assert np.array_equal(got, _reference(ptr, ind, vals3, u, "PlusMultiplies", n).astype(np.float32)), start
        # the transposed orientation reads csc.val, rewritten by the same calls
        transposed(vals3)


def test_auto_waits_for_reuse_before_it_prepares_the_format(powerlaw):
    """The amortisation rule: under `auto` a fresh orientation runs the CSR kernel until it has been multiplied
    `threshold` times, then takes the column-sorted format; results are the same on both sides of the switch."""
    b = powerlaw
    g, torch, dev = b["g"], b["torch"], b["dev"]
    u = np.random.default_rng(31).integers(0, 3, b["n"]).astype(np.float32)
    want = _reference(b["ptr"], b["ind"], b["v_int"], u, "PlusMultiplies", b["n"]).astype(np.float32)
    fmt_before = g.spmv_set_format(-1)                  # the module's `forced` fixture may still be in force
    g.spmv_set_format(1)
    before = g.spmv_set_reuse_threshold(5)
    try:
        A = _matrix(g, torch, dev, b["n"], b["n"], b["ptr"], b["ind"], b["v_int"])
        for launch in range(8):
            got = _run(g, torch, dev, A, b["n"], "PlusMultiplies", u)
            assert np.array_equal(got, want), launch
            assert g.spmv_format_info(A, 0)["in_use"] == (1 if launch >= 5 else 0), launch
    finally:
        g.spmv_set_reuse_threshold(before)
        g.spmv_set_format(fmt_before)


_PREP_SCRIPT = r'''
import json, sys, os
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
g.spmv_set_reuse_threshold(int(os.environ.get("GRB_TEST_THRESHOLD", "0")))
out = []
for scale, ef, sym, forced in ((16, 16, True, False), (14, 8, False, True), (17, 4, True, True)):
    g.spmv_set_format(2 if forced else 1)
    s, d, n = rmat_edges(scale, ef, seed=3, device=dev)
    gr = finalize_edges(s, d, n, symmetrize=sym)
    tp, ti = gr["csr"]
    rng = np.random.default_rng(scale)
    vals = torch.from_numpy(rng.integers(1, 6, gr["nnz"]).astype(np.float32)).to(dev)
    A = g.Matrix(n, n)
    assert A.build_device_csr(tp.data_ptr(), ti.data_ptr(), vals.data_ptr(), gr["nnz"], keep=(tp, ti, vals)) == 0
    u = torch.from_numpy(rng.integers(0, 4, n).astype(np.float32)).to(dev)
    w = torch.zeros(n, dtype=torch.float32, device=dev)
    for _ in range(3):          # (with a threshold of 2 the first two go through the CSR kernel, which renames the columns)
        assert g.k_spmv(A, 0, "PlusMultiplies", u.data_ptr(), None, 0, 0, w.data_ptr()) == 0
    torch.cuda.synchronize()
    info = g.spmv_format_info(A, 0)
    out.append({"info": info, "sum": float(w.double().sum().item()), "w": w.cpu().numpy().astype(np.float64).tolist()[:2000]})
print("RESULT" + json.dumps(out))
'''


def test_preparation_on_the_device_equals_the_host_pass():
    """Hubs, bands and every row's place come from device kernels (degree histogram, scans, the band chase); the host
    pass of the earlier rounds stays behind GRB_CB_PREP_HOST=1.  Both must cut the same bands, deal the same items and
    give the same products (integer-valued data: exact) -- run in two fresh processes, the switch is read once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (host pass?, products through the CSR kernel first): the third run prepares the format from the CSR kernel's renamed
    # column ids (no rank gather in the key kernel) -- same codes, same format
    for host, thr in (("0", "0"), ("1", "0"), ("0", "2")):
        env = dict(os.environ, GRB_CB_PREP_HOST=host, GRB_TEST_THRESHOLD=thr,
                   PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        p = subprocess.run([sys.executable, "-c", _PREP_SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
        res[(host, thr)] = json.loads(line[len("RESULT"):])
    base = res[("0", "0")]
    assert len(base) == 3
    for key in (("1", "0"), ("0", "2")):
        for a, b in zip(base, res[key]):
            assert a["info"]["in_use"] == 1 and b["info"]["in_use"] == 1
            assert a["info"] == b["info"], (key, a["info"], b["info"])
            assert a["sum"] == b["sum"] and a["w"] == b["w"], key
