"""The column-banded SpMV (csrc/spmv_bands.hpp): a matrix large enough to be split into bands, checked
against the textbook definition on the host -- exactly for integer-valued data and idempotent monoids,
within rounding for float sums (the bands change the summation order of a row, nothing else)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max      # the identity of the reference's MinimumMonoid (tests/golden/semiring_ref.json)


@pytest.fixture(scope="module")
def banded():
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    dev = torch.device("cuda", 0)
    before = g.spmv_set_bands(0)
    fmt_before = g.spmv_set_format(-1)
    g.spmv_set_format(0)                                   # the CSR kernel (the column-sorted format would take these products)
    g.spmv_set_bands(8)                                    # off by default: on for this module's plans
    src, dst, n = rmat_edges(19, 16, seed=5, device=dev)
    gr = finalize_edges(src, dst, n, symmetrize=True)
    ptr, ind = gr["csr"]
    nnz = gr["nnz"]
    rng = np.random.default_rng(11)
    vals_int = rng.integers(1, 4, nnz).astype(np.float32)
    vals_f = rng.random(nnz, dtype=np.float32) + 0.25
    out = {"g": g, "torch": torch, "n": n, "nnz": nnz, "ptr": ptr.cpu().numpy(), "ind": ind.cpu().numpy(), "dev": dev}
    for name, vals in (("int", vals_int), ("f", vals_f)):
        tv = torch.from_numpy(vals).to(dev)
        A = g.Matrix(n, n)
        assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), tv.data_ptr(), nnz, keep=(ptr, ind, tv)) == 0
        assert g.spmv_plan_info(A, 0, warm=True)["bands"] >= 2     # the plan is prepared while the setting holds
        out["A_" + name] = A
        out["v_" + name] = vals
    g.spmv_set_bands(before)
    yield out
    g.spmv_set_format(fmt_before)


def _rows(b):
    return np.repeat(np.arange(b["n"]), np.diff(b["ptr"]))


def _run(b, A, op, u, mask=None, scmp=0, accum=0, w0=None):
    torch, g = b["torch"], b["g"]
    tu = torch.from_numpy(u).to(b["dev"])
    tw = torch.from_numpy(w0.copy()).to(b["dev"]) if w0 is not None else torch.empty(b["n"], dtype=torch.float32, device=b["dev"])
    tm = torch.from_numpy(mask).to(b["dev"]) if mask is not None else None
    torch.cuda.synchronize()
    assert g.k_spmv(A, 0, op, tu.data_ptr(), tm.data_ptr() if tm is not None else None, scmp, accum, tw.data_ptr()) == 0
    torch.cuda.synchronize()
    return tw.cpu().numpy()


def test_bands_are_in_use(banded):
    """the fixture really exercises the banded kernel: the library reports more than one LDS prefix"""
    info = banded["g"].spmv_plan_info(banded["A_int"], 0, warm=True)
    assert info["bands"] >= 2 and info["band_nnz"] > 0 and info["pieces"] > 0, info
    assert info["band_nnz"] < banded["nnz"]


def test_plus_multiplies_integer_data_exact(banded):
    b = banded
    rng = np.random.default_rng(1)
    u = rng.integers(0, 3, b["n"]).astype(np.float32)
    got = _run(b, b["A_int"], "PlusMultiplies", u)
    want = np.bincount(_rows(b), weights=b["v_int"].astype(np.float64) * u[b["ind"]], minlength=b["n"])
    assert np.array_equal(got, want.astype(np.float32))


def test_plus_multiplies_float_within_rounding(banded):
    b = banded
    rng = np.random.default_rng(2)
    u = rng.random(b["n"], dtype=np.float32)
    got = _run(b, b["A_f"], "PlusMultiplies", u)
    want = np.bincount(_rows(b), weights=b["v_f"].astype(np.float64) * u[b["ind"]].astype(np.float64), minlength=b["n"])
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)   # north_star's float tolerance
    # and run to run the order is fixed
    assert np.array_equal(got, _run(b, b["A_f"], "PlusMultiplies", u))


@pytest.mark.parametrize("op", ["MinimumPlus", "MaximumMultiplies", "LogicalOrAnd", "MinimumSelectSecond"])
def test_idempotent_monoids_exact(banded, op):
    b = banded
    rng = np.random.default_rng(3)
    u = (rng.random(b["n"], dtype=np.float32) * 8).astype(np.float32)
    u[rng.random(b["n"]) < 0.3] = 0.0
    a = b["v_f"]
    x = u[b["ind"]]
    nonempty = np.diff(b["ptr"]) > 0
    starts = b["ptr"][:-1][nonempty]
    if op == "MinimumPlus":
        prod, red, ident = a + x, np.minimum, FMAX
    elif op == "MaximumMultiplies":
        prod, red, ident = a * x, np.maximum, np.float32(0)
    elif op == "LogicalOrAnd":
        prod, red, ident = ((a != 0) & (x != 0)).astype(np.float32), np.maximum, np.float32(0)
    else:
        prod, red, ident = x.copy(), np.minimum, FMAX
    want = np.full(b["n"], ident, dtype=np.float32)
    want[nonempty] = red.reduceat(prod.astype(np.float32), starts)
    got = _run(b, b["A_f"], op, u)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scmp,accum", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_mask_and_accumulate(banded, scmp, accum):
    b = banded
    rng = np.random.default_rng(4)
    u = rng.integers(0, 3, b["n"]).astype(np.float32)
    mask = (rng.random(b["n"]) < 0.5).astype(np.float32)
    w0 = rng.integers(0, 5, b["n"]).astype(np.float32)
    got = _run(b, b["A_int"], "PlusMultiplies", u, mask=mask, scmp=scmp, accum=accum, w0=w0)
    full = np.bincount(_rows(b), weights=b["v_int"].astype(np.float64) * u[b["ind"]], minlength=b["n"]).astype(np.float32)
    passes = (mask != 0) != bool(scmp)
    want = np.where(passes, full, np.float32(0))
    if accum:
        want = w0 + want
    assert np.array_equal(got, want)
