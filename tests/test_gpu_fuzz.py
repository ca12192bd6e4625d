"""GPU suite: differential fuzzing of op SEQUENCES (tests/tools/ops_fuzz.py) -- random vxm / mxv / eWiseAdd /
eWiseMult / assign / reduce / build / dup / swap / clear calls on a shared pool of vectors, in lock-step
through the C ABI and through the oracle, every Info code and the whole observable state compared
after every call.  Campaigns: float vectors with all 17 semirings, int32 vectors, the traversal drivers' op mix with
struconly / opreuse drawn at random -- and three more with the SpMV's band format forced on."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags,seed,fmt", [((), 11, None), (("--int",), 12, None), (("--struconly",), 13, None),
                                            ((), 14, "cband"), (("--int",), 15, "cband"), (("--n", "300"), 16, "cband")])
def test_op_sequences_match_the_oracle(flags, seed, fmt):
    """fmt "cband": every pull product of a true monoid through the column-sorted band format, whatever the matrix
    looks like (the default takes it only where it pays)"""
    env = dict(os.environ)
    if fmt:
        env["GRB_SPMV_FORMAT"] = fmt
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "ops_fuzz.py"), "--seqs", "250", "--len", "40",
                          "--seed", str(seed)] + list(flags), capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "diverging sequences: 0" in out.stdout
