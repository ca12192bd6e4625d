#!/usr/bin/env python3
"""Regenerates the golden fixtures from the reference tree.  Run ONLY in the build
container (needs /root/reference); the outputs are committed, this script never runs
on the GPU box.

  tests/golden/data/*.mtx        the data files the reference's tests load (data/small)
  tests/golden/ref_tests.json    the literal input vectors of test/gvxm.cu, gewiseadd.cu,
                                 gewisemult.cu, greduce.cu (per BOOST test case) and the
                                 test function each case calls -- data only; expected
                                 values are the literals the tests assert (greduce) or are
                                 computed by the checkers in tests/ with the tests' inline
                                 formulas
  tests/golden/semiring_ref.json identity/add/mul tables printed by oracle/_ref/semiring_ref,
                                 i.e. by the reference's own graphblas/stddef.hpp
  tests/golden/types_ref.json    the enumerations of the reference's own graphblas/types.hpp (oracle/_ref/types_ref)
  tests/golden/mmio_ref.json     banner + size of every data file as parsed by the
                                 reference's own graphblas/mmio.hpp (oracle/_ref/libmmio_ref.so)
  tests/golden/known_answers.json  outputs of the reference CPU oracles recorded in
                                 SURVEY.md 8(c) (chesapeake BFS depth / TC, test_cc, test_bc)
"""
import ctypes
import json
import os
import re
import shutil
import subprocess
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def parse_cases(path):
    text = open(path).read()
    cases = {}
    for m in re.finditer(r"BOOST_FIXTURE_TEST_CASE\(\s*(\w+)\s*,\s*\w+\s*\)\s*\{(.*?)\n\}", text, re.S):
        name, body = m.group(1), m.group(2)
        vecs = {}
        for v in re.finditer(r"std::vector<([^>]+)>\s+(\w+)\s*\{([^}]*)\}\s*;", body, re.S):
            ctype, vname, items = v.group(1), v.group(2), v.group(3)
            vals = [x.strip().rstrip("f") for x in items.replace("\n", " ").split(",") if x.strip()]
            nums = [float(x) for x in vals]
            if "Index" in ctype or ctype.strip() == "int":
                nums = [int(x) for x in nums]
            vecs[vname] = nums
        for v in re.finditer(r"std::vector<([^>]+)>\s+(\w+)\s*\(\s*(\d+)\s*,\s*([-\d.]+)f?\s*\)\s*;", body):
            vecs[v.group(2)] = [float(v.group(4))] * int(v.group(3))
        calls = [" ".join(c.split()) for c in re.findall(r"\n\s*(test\w+\s*\([^;]*\))\s*;", body, re.S)]
        cases[name] = dict(vectors=vecs, calls=calls)
    return cases


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "data/small"))):
        if f.endswith(".mtx"):
            shutil.copy(os.path.join(REF, "data/small", f), os.path.join(HERE, "data", f))
    tests = {}
    for f in ("gvxm.cu", "gewiseadd.cu", "gewisemult.cu", "greduce.cu"):
        tests[f] = parse_cases(os.path.join(REF, "test", f))
    json.dump(tests, open(os.path.join(HERE, "ref_tests.json"), "w"), indent=1, sort_keys=True)

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle/_ref/semiring_ref")])
    json.dump(json.loads(out), open(os.path.join(HERE, "semiring_ref.json"), "w"), indent=1, sort_keys=True)

    out = subprocess.check_output([os.path.join(ROOT, "oracle/_ref/types_ref")])
    json.dump(json.loads(out), open(os.path.join(HERE, "types_ref.json"), "w"), indent=1, sort_keys=True)

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle/_ref/libmmio_ref.so"))
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    banner = getattr(lib, "_Z14mm_read_bannerP8_IO_FILEPA4_c")
    banner.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    size = getattr(lib, "_Z20mm_read_mtx_crd_sizeP8_IO_FILEPiS1_S1_")
    size.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    res = {}
    for f in sorted(os.listdir(os.path.join(HERE, "data"))):
        fp = libc.fopen(os.path.join(HERE, "data", f).encode(), b"r")
        code = ctypes.create_string_buffer(4)
        rc = banner(fp, code)
        m, n, nz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc2 = size(fp, ctypes.byref(m), ctypes.byref(n), ctypes.byref(nz))
        libc.fclose(fp)
        res[f] = dict(rc_banner=rc, typecode=code.raw.decode(), rc_size=rc2, nrows=m.value, ncols=n.value,
                      nnz=nz.value)
    json.dump(res, open(os.path.join(HERE, "mmio_ref.json"), "w"), indent=1, sort_keys=True)

    known = {
        "source": "SURVEY.md 8(c): outputs of the reference's SimpleReference* / readMtx / coo2csr "
                  "compiled from /root/reference during the survey",
        "chesapeake": {"n": 39, "nnz": 340, "bfs_source": 0, "bfs_search_depth": 3,
                       "bfs_depth": [int(x) for x in
                                     "1 3 3 3 3 3 2 2 3 3 2 2 2 3 3 3 3 3 3 3 3 2 2 3 3 3 3 3 3 3 3 3 3 2 2 3 2 3 2".split()],
                       "tc_tril": 194, "tc_full": 1164},
        "test_cc": {"bfs_source": 0, "bfs_depth": [1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                    "row_degrees": [1, 1, 3, 2, 2, 3, 3, 0, 1, 2, 2]},
        "test_bc": {"bfs_source": 0, "bfs_depth": [1, 2, 0, 0, 0, 0, 0],
                    "row_degrees": [1, 1, 3, 2, 2, 3, 3]},
    }
    json.dump(known, open(os.path.join(HERE, "known_answers.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
