"""Test-only pieces for the partitioned BFS: a numpy level engine (so the N > 1
orchestration of graphblast_amd/dist.py runs under gloo on CPU) and an in-process
thread communicator (so two simulated ranks can drive the real HIP engine on one GPU)."""
import threading

import numpy as np
import torch


def _bits(t):
    return t.numpy().view(np.uint32)


class NumpyEngine:
    def __init__(self, n, lo, lptr, lind, dev, in_lptr=None, in_lind=None):
        self.n, self.lo = n, lo
        self.ptr = lptr.cpu().numpy().astype(np.int64)
        self.ind = lind.cpu().numpy().astype(np.int64)
        self.n_local = self.ptr.size - 1
        self.rows = np.repeat(np.arange(self.n_local), np.diff(self.ptr))
        # in-edges of the owned vertices (pull, PageRank); a symmetric graph has one shard for both
        self.iptr = self.ptr if in_lptr is None else in_lptr.cpu().numpy().astype(np.int64)
        self.iind = self.ind if in_lind is None else in_lind.cpu().numpy().astype(np.int64)
        self.irows = np.repeat(np.arange(self.n_local), np.diff(self.iptr))

    @staticmethod
    def _test(bm, idx):
        return ((bm[idx >> 5] >> (idx & 31).astype(np.uint32)) & 1).astype(bool)

    @staticmethod
    def _set(bm, idx):
        np.bitwise_or.at(bm, idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))

    def pull(self, vis, new_local, label_local, new_label):
        v, nl = _bits(vis), _bits(new_local)
        own = np.arange(self.n_local) + self.lo
        unvisited = ~self._test(v, own)
        hit_edge = self._test(v, self.iind[:self.iptr[-1]])
        any_hit = np.zeros(self.n_local, dtype=bool)
        np.logical_or.at(any_hit, self.irows, hit_edge)
        found = np.nonzero(unvisited & any_hit)[0]
        self._set(nl, found + self.lo)
        label_local.numpy()[found] = new_label

    def push(self, frontier, vis, new_local):
        f, v, nl = _bits(frontier), _bits(vis), _bits(new_local)
        nl[:] = 0
        own = np.arange(self.n_local) + self.lo
        fr = np.nonzero(self._test(f, own))[0]
        if fr.size == 0:
            return
        sel = np.isin(self.rows, fr)
        dst = np.unique(self.ind[:self.ptr[-1]][sel])
        dst = dst[~self._test(v, dst)]
        self._set(nl, dst)

    def apply(self, new_global, vis, label_local, new_label):
        ng, v = _bits(new_global), _bits(vis)
        fresh = ng & ~v
        v |= fresh
        own = np.arange(self.n_local) + self.lo
        mine = np.nonzero(self._test(fresh, own))[0]
        label_local.numpy()[mine] = new_label
        return int(sum(bin(int(x)).count("1") for x in fresh[fresh != 0]))

    def pr_setup(self, vals, dev):
        self.pr_vals = vals.cpu().numpy().astype(np.float32)

    def pr_step(self, p_full, y_local, p_old_local, const):
        p = p_full.numpy()
        prod = self.pr_vals[:self.iptr[-1]] * p[self.iind[:self.iptr[-1]]]
        y = np.zeros(self.n_local, dtype=np.float32)
        np.add.at(y, self.irows, prod)
        y = (y + np.float32(const)).astype(np.float32)
        y_local.numpy()[:self.n_local] = y
        po = p_old_local.numpy()[:self.n_local]
        r = np.where((y == 0) | (po == 0), np.float32(0), y - po)      # eWiseMult identity short-circuit
        return float(np.sum((r * r).astype(np.float32), dtype=np.float32))

    def sssp_setup(self, vals, dev):
        self.sssp_vals = vals.cpu().numpy().astype(np.float32)

    def sssp_step(self, d_full, d_local):
        d = d_full.numpy()
        m = int(self.iptr[-1])
        cand = (d[self.iind[:m]] + self.sssp_vals[:m]).astype(np.float32)     # one rounding, like the kernel
        y = np.full(self.n_local, np.finfo(np.float32).max, dtype=np.float32)
        np.minimum.at(y, self.irows, cand)
        dl = d_local.numpy()[:self.n_local]
        better = y < dl
        dl[better] = y[better]
        return int(better.sum())

    def tally(self, label_local):
        lab = label_local.numpy()[:self.n_local]
        deg = np.diff(self.ptr)
        return int(deg[lab != 0].sum()), int(np.count_nonzero(lab))


class ThreadComm:
    """Lock-step communicator for `world` partitions living in threads of one process."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.lock = threading.RLock()         # serialises calls into the C library (re-entrant: engine methods call each other)

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def or_combine(self, new_local, new_global):
        self.s.slots[self.rank] = new_local
        self.s.barrier.wait()
        acc = self.s.slots[0].clone()
        for r in range(1, self.world):
            acc |= self.s.slots[r]
        new_global.copy_(acc)
        self.s.barrier.wait()

    def sum_(self, t):
        self.s.slots[self.rank] = t.clone()
        self.s.barrier.wait()
        tot = sum(self.s.slots[r] for r in range(self.world))
        self.s.barrier.wait()
        t.copy_(tot)
        return t

    def all_gather_padded(self, pad):
        self.s.slots[self.rank] = pad.clone()
        self.s.barrier.wait()
        out = torch.stack([self.s.slots[r] for r in range(self.world)])
        self.s.barrier.wait()
        return out


def locked_engine(engine_cls, lock):
    """Wrap an engine class so that EVERY library call of a simulated rank holds `lock` (the C library's context --
    scratch slots, the mailbox and its sequence number -- is per process, not per thread: two threads inside
    grb_bfs_part_apply2 at once take each other's sequence numbers and one of them waits for a record that was
    overwritten).  Every public method is wrapped, not a list of names: a method added to the engine later is
    covered too."""
    import functools

    class Locked(engine_cls):
        pass

    def wrap(fn):
        @functools.wraps(fn)
        def locked(self, *a, **k):
            with lock:
                r = fn(self, *a, **k)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                return r
        return locked
    for name in dir(engine_cls):
        if name.startswith("_"):
            continue
        fn = getattr(engine_cls, name)
        if callable(fn) and not isinstance(fn, (staticmethod, classmethod, type)):
            setattr(Locked, name, wrap(fn))
    return Locked
