"""CPU: the Bloom filter in front of the masked SpGEMM's LDS tables (csrc/mxm.hip), restated in numpy
with the kernel's own hash, word and bit selection and sizing rule: a key of the pivot always passes (no false
negative, whatever the pivot length), and the share of absent keys that pass is the "few percent" the design counts
on -- at 16 bits per key (the 64 KiB tables and the wave tables at capacity) and at 8 (the 128 KiB tables at
capacity).  The device kernels are checked against the oracle in tests/test_gpu_mxm.py / test_gpu_algorithms.py."""
import numpy as np


def tc_hash(c):
    return (c.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xffffffff)


def filter_words(length, first, limit):
    fwords = first                                         # block kernel: 64 .. 4096 words; wave kernel: 16 .. 256
    while 2 * fwords < length and fwords < limit:
        fwords <<= 1
    return fwords


def build(keys, fwords):
    h = tc_hash(keys)
    filt = np.zeros(fwords, dtype=np.uint64)
    word = ((h >> np.uint64(12)) & np.uint64(fwords - 1)).astype(np.int64)
    pat = (np.uint64(1) << (h & np.uint64(31))) | (np.uint64(1) << ((h >> np.uint64(5)) & np.uint64(31)))
    np.bitwise_or.at(filt, word, pat)
    return filt


def passes(filt, keys):
    h = tc_hash(keys)
    word = ((h >> np.uint64(12)) & np.uint64(filt.size - 1)).astype(np.int64)
    pat = (np.uint64(1) << (h & np.uint64(31))) | (np.uint64(1) << ((h >> np.uint64(5)) & np.uint64(31)))
    return (filt[word] & pat) == pat


def test_no_false_negative_and_few_false_positives():
    rng = np.random.default_rng(11)
    ncols = 1 << 22
    for length, first, limit, bound in ((37, 16, 256, 0.03), (512, 16, 256, 0.04),          # wave tables
                                         (700, 64, 4096, 0.04), (8192, 64, 4096, 0.04),       # 64 KiB tables
                                         (16384, 64, 4096, 0.12)):                            # 128 KiB tables at capacity
        # a hub's list: skewed towards low column ids, like the rows of a power-law lower triangle
        pool = np.unique((ncols * rng.random(4 * length) ** 3).astype(np.int64))
        keys = rng.choice(pool, size=min(length, pool.size), replace=False)
        fwords = filter_words(keys.size, first, limit)
        filt = build(keys, fwords)
        assert passes(filt, keys).all(), length
        probes = rng.integers(0, ncols, 200000)
        probes = probes[~np.isin(probes, keys)]
        fp = float(passes(filt, probes).mean())
        assert fp <= bound, (length, fwords, fp)
