"""1-D vertex-partitioned, multi-GPU direction-optimised BFS.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU
for the logic tests).  Rank r owns the vertex range [lo_r, hi_r) -- boundaries on
multiples of 64, chosen so every rank holds about nnz / P stored edges -- with the
out-edges (push) and in-edges (pull) of its vertices, a replica of the n-bit visited
bitmap and the labels of its own vertices.

Per level every rank expands locally (C ABI: grb_bfs_part_pull / grb_bfs_part_push), the
n/8-byte "new bits" bitmaps are OR-combined with ONE all-gather (each rank sends 512 KiB
at RMAT-22; xGMI is point-to-point, so an all-gather uses all links at once where a ring
all-reduce of the same bitmap would be per-link bound), and grb_bfs_part_apply folds the
result into the replicated state and returns the next frontier size -- the same number on
every rank, so the push/pull decision (the reference's `convert` rule,
backend/cuda/vector.hpp:291-323) is taken identically everywhere without another
collective.  The reference itself has no multi-GPU path.

The orchestration is engine-agnostic: `HipEngine` drives libgrb_hip.so; the CPU logic
tests plug in a numpy engine (tests/), which is how the N > 1 path is covered without GPUs.
"""
import ctypes as C

import time

import numpy as np
import torch
import torch.distributed as dist

GRB_PUSHPULL, GRB_PUSHONLY, GRB_PULLONLY = 10, 11, 12


def partition_bounds(ptr, world):
    """nnz-balanced vertex ranges, lower bounds on multiples of 64."""
    ptr = np.asarray(ptr, dtype=np.int64)
    n = ptr.size - 1
    nnz = int(ptr[-1])
    bounds = [0]
    for r in range(1, world):
        v = int(np.searchsorted(ptr, nnz * r // world, side="left"))
        v = min(n, max(bounds[-1], (v + 32) // 64 * 64))
        bounds.append(v)
    bounds.append(n)
    return bounds


def word_slices_usable(bounds):
    """Whether the ranks' vertex ranges are also disjoint ranges of 32-bit bitmap words -- the condition for the
    pull levels' in-place all-gather of word ranges.  partition_bounds clamps an interior bound to n: with
    n % 64 >= 32 and a heavy tail vertex a bound falls inside a word (n = 190, world 2: [0, 190, 190]); that word
    would then be broadcast by the empty rank from its stale copy and the owner's discoveries lost."""
    return all(b % 32 == 0 for b in bounds[1:-1])


def bitmap_words(n):
    return 2 * ((n + 63) // 64)


class HipEngine:
    """Level steps through the C ABI on device tensors."""

    def __init__(self, n, lo, lptr, lind, dev, in_lptr=None, in_lind=None):
        """lptr / lind: the OUT-edges of the owned vertices (rows lo.. of the CSR; walked by push, counted
        for TEPS); in_lptr / in_lind: their IN-edges (rows lo.. of the CSC; walked by pull and by the PageRank
        shard).  A symmetric graph passes one pair for both."""
        import graphblast_amd as g
        from . import _lib
        self._lib = _lib.load()
        self.n, self.lo, self.n_local = n, lo, lptr.numel() - 1
        self.keep = (lptr, lind)
        self.A = g.Matrix(self.n_local, n)
        info = self.A.build_device_csr(lptr.data_ptr(), lind.data_ptr(), None, int(lind.numel()), keep=self.keep)
        if info != 0:
            raise RuntimeError("grb_matrix_adopt_device_csr failed: %d" % info)
        if in_lptr is None:
            self.A_in, self.keep_in = self.A, self.keep
        else:
            self.keep_in = (in_lptr, in_lind)
            self.A_in = g.Matrix(self.n_local, n)
            info = self.A_in.build_device_csr(in_lptr.data_ptr(), in_lind.data_ptr(), None, int(in_lind.numel()),
                                              keep=self.keep_in)
            if info != 0:
                raise RuntimeError("grb_matrix_adopt_device_csr (in-edges) failed: %d" % info)
        self.work = torch.zeros(bitmap_words(n), dtype=torch.int32, device=dev)

    def pull(self, vis, new_local, label_local, new_label):
        info = self._lib.grb_bfs_part_pull(self.A_in._h, self.lo, self.n, vis.data_ptr(), new_local.data_ptr(),
                                           label_local.data_ptr(), float(new_label))
        assert info == 0, info

    def push(self, frontier, vis, new_local):
        info = self._lib.grb_bfs_part_push(self.A._h, self.lo, self.n, frontier.data_ptr(), vis.data_ptr(),
                                           self.work.data_ptr(), new_local.data_ptr(), None)
        assert info == 0, info

    def apply(self, new_global, vis, label_local, new_label):
        out = C.c_int32(0)
        info = self._lib.grb_bfs_part_apply(new_global.data_ptr(), vis.data_ptr(), self.lo, self.n_local, self.n,
                                            label_local.data_ptr(), float(new_label), C.byref(out))
        assert info == 0, info
        return out.value

    # ---- leaner steps (one launch and one host wake-up each); Partition1D uses them when present
    pull_zeroes = True                       # grb_bfs_part_pull clears the new-bits bitmap itself
    small_push_edges = 1 << 15               # below this many local frontier out-edges: the one-launch push

    def seed(self, vis, new_global, label_local, source):
        info = self._lib.grb_bfs_part_seed(vis.data_ptr(), new_global.data_ptr(), label_local.data_ptr(), self.lo,
                                           self.n_local, self.n, int(source))
        assert info == 0, info

    def apply2(self, new_global, vis, label_local, new_label, deg_full=None):
        """-> (vertices discovered by this level, out-degree sum of the OWNED ones among them, out-degree
        sum of all of them or -1 without `deg_full`, an int32 device tensor of every vertex's out-degree)"""
        out, edges, all_edges = C.c_int32(0), C.c_int64(0), C.c_int64(-1)
        info = self._lib.grb_bfs_part_apply2(new_global.data_ptr(), vis.data_ptr(), self.lo, self.n_local, self.n,
                                             self.A._h, None if deg_full is None else deg_full.data_ptr(),
                                             label_local.data_ptr(), float(new_label), C.byref(out), C.byref(edges),
                                             C.byref(all_edges))
        assert info == 0, info
        return out.value, edges.value, all_edges.value

    def push_small(self, frontier, vis, new_local):
        info = self._lib.grb_bfs_part_push_small(self.A._h, self.lo, self.n, frontier.data_ptr(), vis.data_ptr(),
                                                 new_local.data_ptr())
        assert info == 0, info

    def or_parts(self, gathered, world, nwords, out):
        info = self._lib.grb_bitmap_or_parts(gathered.data_ptr(), int(world), int(nwords), out.data_ptr())
        assert info == 0, info

    def unlabel(self, label_local, value):
        assert self._lib.grb_bfs_part_unlabel(label_local.data_ptr(), self.n_local, float(value)) == 0

    def tally(self, label_local):
        e, r = C.c_int64(0), C.c_int32(0)
        info = self._lib.grb_bfs_part_tally(self.A._h, label_local.data_ptr(), C.byref(e), C.byref(r))
        assert info == 0, info
        return e.value, r.value

    # ---- the whole traversal with the level loop on the device (csrc/bfs_part_run.hip)
    def part_context(self, rank, world, deg_full, nnz):
        """the rank context of grb_bfs_part_run: shards + replicated out-degrees (int32 device tensor, kept alive)"""
        if getattr(self, "_part", None) is None:
            h = C.c_void_p()
            a_in = None if self.A_in is self.A else self.A_in._h
            info = self._lib.grb_part_new(C.byref(h), int(rank), int(world), self.n, self.lo, self.A._h, a_in,
                                          deg_full.data_ptr(), int(nnz))
            if info != 0:
                raise RuntimeError("grb_part_new: Info %d" % info)
            self._part, self._part_keep = h, deg_full
        return self._part

    def run_bfs(self, part, source, mxvmode, switchpoint, edgeswitch, max_niter, label_local, levels_per_launch=1,
                want_trace=True):
        """-> (PartBfsResult, [(direction, frontier, discovered)] per level); nothing is read back per level"""
        from ._lib import PartBfsResult, BfsLevel
        if getattr(self, "_lv", None) is None:
            self._lv_cap = 4096
            self._lv = (BfsLevel * self._lv_cap)()
            self._res = PartBfsResult()
        res, lv = self._res, self._lv
        info = self._lib.grb_bfs_part_run(part, int(source), int(mxvmode), float(switchpoint), float(edgeswitch),
                                          int(max_niter), int(levels_per_launch), label_local.data_ptr(), C.byref(res),
                                          lv, self._lv_cap if want_trace else 0)
        if info != 0:
            raise RuntimeError("grb_bfs_part_run: Info %d" % info)
        trace = [("pull" if lv[i].direction else "push", int(lv[i].frontier), int(lv[i].discovered))
                 for i in range(min(res.levels, self._lv_cap))] if want_trace else None
        return res, trace

    # ---- algorithm::sssp in frontier form, round loop on the device (csrc/sssp_part_run.hip)
    def sssp_context(self, rank, world, out_weights_local, outbox_pairs=65536):
        """out_weights_local: the weights of the OWNED vertices' out-edges, in the order of this shard's arrays"""
        import graphblast_amd as g
        lptr, lind = self.keep
        w = out_weights_local.to(torch.float32).contiguous()
        if w.numel() == 0:
            w = torch.zeros(1, dtype=torch.float32, device=lind.device)
        old = getattr(self, "_sssp_part", None)
        if old is not None:
            self._lib.grb_part_sssp_free(old)
            self._sssp_part = None
        self.A_w = g.Matrix(self.n_local, self.n)
        info = self.A_w.build_device_csr(lptr.data_ptr(), lind.data_ptr(), w.data_ptr(), int(self.A.nvals()), keep=(lptr, lind, w))
        if info != 0:
            raise RuntimeError("grb_matrix_adopt_device_csr (weights) failed: %d" % info)
        h = C.c_void_p()
        info = self._lib.grb_part_sssp_new(C.byref(h), int(rank), int(world), self.n, self.lo, self.A_w._h, int(outbox_pairs))
        if info != 0:
            raise RuntimeError("grb_part_sssp_new: Info %d" % info)
        self._sssp_part = h
        return h

    def run_sssp(self, part, source, max_niter, dist_local, rounds_per_launch=1):
        from ._lib import PartSsspResult
        res = PartSsspResult()
        info = self._lib.grb_sssp_part_run(part, int(source), int(max_niter), int(rounds_per_launch), dist_local.data_ptr(),
                                           C.byref(res))
        if info != 0:
            raise RuntimeError("grb_sssp_part_run: Info %d" % info)
        return res

    def __del__(self):
        sp = getattr(self, "_sssp_part", None)
        if sp is not None:
            try:
                self._lib.grb_part_sssp_free(sp)
            except Exception:                                         # noqa: BLE001 -- interpreter shutdown
                pass
            self._sssp_part = None
        part = getattr(self, "_part", None)
        if part is not None:
            try:
                self._lib.grb_part_free(part)
            except Exception:                                         # noqa: BLE001 -- interpreter shutdown
                pass
            self._part = None

    # ---- PageRank shard: rows = in-edges of the owned vertices, values alpha / outdeg(source)
    def pr_setup(self, vals, dev):
        import graphblast_amd as g
        lptr, lind = self.keep_in
        self.pr_vals = vals
        self.Apr = g.Matrix(self.n_local, self.n)
        info = self.Apr.build_device_csr(lptr.data_ptr(), lind.data_ptr(), vals.data_ptr(), int(vals.numel()),
                                         keep=(lptr, lind, vals))
        assert info == 0, info
        self.g = g
        self.desc = g.Descriptor()
        assert self.desc.loadArgs(mxvmode=2) == 0
        n1 = max(self.n_local, 1)
        self._buf = {k: torch.zeros(n1, dtype=torch.float32, device=dev) for k in ("r", "r2")}
        self._vec = {}

    def pr_setup_chunks(self, vals, dev, nchunks=2):
        """The in-edge shard cut into `nchunks` row chunks (nnz-balanced, boundaries anywhere): chunk c's slice
        of the new vector is all-gathered on the communication stream while chunk c + 1 is multiplied."""
        import graphblast_amd as g
        lptr, lind = self.keep_in
        hp = lptr.cpu().numpy().astype(np.int64)
        nnz = int(hp[-1])
        cuts = [0]
        for c in range(1, nchunks):
            cuts.append(int(min(self.n_local, max(cuts[-1], np.searchsorted(hp, nnz * c // nchunks)))))
        cuts.append(self.n_local)
        self.pr_chunks = []
        for c in range(nchunks):
            a, b = cuts[c], cuts[c + 1]
            e0, e1 = int(hp[a]), int(hp[b])
            cptr = (lptr[a:b + 1] - e0).to(torch.int32).contiguous()
            cind = lind[e0:e1].contiguous() if e1 > e0 else torch.zeros(1, dtype=torch.int32, device=dev)
            cval = vals[e0:e1].contiguous() if e1 > e0 else torch.zeros(1, dtype=torch.float32, device=dev)
            M = g.Matrix(max(b - a, 0), self.n)
            if b > a:
                info = M.build_device_csr(cptr.data_ptr(), cind.data_ptr(), cval.data_ptr(), e1 - e0,
                                          keep=(cptr, cind, cval))
                assert info == 0, info
            self.pr_chunks.append((a, b, M))
        self.g = g
        return cuts

    def _adopt(self, key, tensor):
        v = self._vec.get(key)
        if v is None:
            v = self.g.Vector(max(self.n_local, 1))
            self._vec[key] = v
        assert v.build_device(tensor.data_ptr(), max(self.n_local, 1)) == 0
        return v

    def pr_step(self, p_full, y_local, p_old_local, const):
        """y = A_in p (local rows); y += const; returns sum((y - p_old)^2) over the owned slice.
        Exactly the op sequence of algorithm/pr.hpp:66-80 on the shard."""
        g = self.g
        if self.n_local == 0:
            return 0.0
        assert g.k_spmv(self.Apr, 0, "PlusMultiplies", p_full.data_ptr(), None, 0, 0, y_local.data_ptr()) == 0
        y = self._adopt("y", y_local)
        po = self._adopt("po", p_old_local)
        r = self._adopt("r", self._buf["r"])
        r2 = self._adopt("r2", self._buf["r2"])
        assert g.eWiseAdd(y, None, None, "PlusMultiplies", y, float(const), self.desc) == 0
        assert g.eWiseMult(r, None, None, "PlusMinus", y, po, self.desc) == 0
        assert g.eWiseAdd(r2, None, None, "MultipliesMultiplies", r, r, self.desc) == 0
        info, val = g.reduce(None, "Plus", r2, self.desc)
        assert info == 0
        return float(val)


    # ---- SSSP shard: rows = in-edges of the owned vertices with their weights
    def sssp_setup(self, vals, dev):
        import graphblast_amd as g
        lptr, lind = self.keep_in
        self.Asssp = g.Matrix(self.n_local, self.n)
        info = self.Asssp.build_device_csr(lptr.data_ptr(), lind.data_ptr(), vals.data_ptr(), int(vals.numel()),
                                           keep=(lptr, lind, vals))
        assert info == 0, info
        self.g = g
        self._sssp_y = torch.empty(max(self.n_local, 1), dtype=torch.float32, device=dev)

    def sssp_step(self, d_full, d_local):
        """d_local = min(d_local, A_in min.+ d_full) over the owned rows; -> how many of them improved.
        The dense (pull) form of a round as library calls -- the product, then the reference's own three
        element-wise steps (sssp.hpp:70-75, :83): m = y < d, d = min(d, y), succ = sum(m).  The HIP engine's
        own drivers use the frontier form (run_sssp); this is what the host-driven loop falls back to."""
        if self.n_local == 0:
            return 0
        g = self.g
        y = self._sssp_y
        assert g.k_spmv(self.Asssp, 0, "MinimumPlus", d_full.data_ptr(), None, 0, 0, y.data_ptr()) == 0
        if getattr(self, "_sssp_vec", None) is None:
            self._sssp_vec = {k: g.Vector(max(self.n_local, 1)) for k in ("y", "d", "m")}
            self._sssp_m = torch.empty(max(self.n_local, 1), dtype=torch.float32, device=y.device)
            self._sssp_desc = g.Descriptor()
            assert self._sssp_desc.loadArgs(mxvmode=2) == 0
        vy, vd, vm = self._sssp_vec["y"], self._sssp_vec["d"], self._sssp_vec["m"]
        n1 = max(self.n_local, 1)
        assert vy.build_device(y.data_ptr(), n1) == 0
        assert vd.build_device(d_local.data_ptr(), n1) == 0
        assert vm.build_device(self._sssp_m.data_ptr(), n1) == 0
        assert g.eWiseAdd(vm, None, None, "CustomLessPlus", vy, vd, self._sssp_desc) == 0
        assert g.eWiseAdd(vd, None, None, "MinimumPlus", vd, vy, self._sssp_desc) == 0
        info, val = g.reduce(None, "Plus", vm, self._sssp_desc)
        assert info == 0
        return int(round(float(val)))


class TorchComm:
    """The collectives of the partitioned BFS over torch.distributed (RCCL / gloo)."""

    def __init__(self, world, nwords, dev):
        self.world, self.nwords = world, nwords
        self.gathered = torch.zeros(world * nwords, dtype=torch.int32, device=dev) if world > 1 else None

    def or_combine(self, new_local, new_global, engine=None):
        """OR of every rank's new-bits bitmap: one all-gather + local OR (one kernel when the engine
        has it, world - 1 tensor ops otherwise)."""
        if self.world == 1:
            new_global.copy_(new_local)
            return
        dist.all_gather_into_tensor(self.gathered, new_local)
        if engine is not None and hasattr(engine, "or_parts") and new_global.is_cuda:
            engine.or_parts(self.gathered, self.world, self.nwords, new_global)
            return
        g = self.gathered.view(self.world, self.nwords)
        new_global.copy_(g[0])
        for r in range(1, self.world):
            new_global.bitwise_or_(g[r])

    def sum_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_gather_padded(self, pad):
        out = torch.zeros(self.world * pad.numel(), dtype=pad.dtype, device=pad.device)
        dist.all_gather_into_tensor(out, pad)
        return out.view(self.world, pad.numel())


class RcclComm:
    """The same collectives through the LIBRARY's communicator (csrc/comm.hip): RCCL calls enqueued from C++
    on a second HIP stream, fenced with events against the compute stream -- no torch.distributed in the data
    path, no host synchronisation per collective.  torch.distributed is only used once, to hand rank 0's
    ncclUniqueId to the other ranks."""

    def __init__(self, rank, world, nwords, dev):
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.nwords, self.dev = rank, world, nwords, dev
        have_r, have_w = C.c_int(0), C.c_int(0)
        self._lib.grb_comm_info(C.byref(have_r), C.byref(have_w))
        if have_w.value == 0:
            # byte 128 says whether rank 0 got an id: a failure there must reach every rank (they would otherwise
            # wait in the broadcast for a rank that has already fallen back to torch.distributed)
            ident = torch.zeros(129, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                if self._lib.grb_comm_unique_id(buf) == 0:
                    ident = torch.tensor(list(buf) + [1], dtype=torch.uint8)
            if world > 1:
                t = ident.to(dev) if dist.get_backend() == "nccl" else ident
                dist.broadcast(t, 0)
                ident = t.cpu()
            if int(ident[128]) != 1:
                raise RuntimeError("grb_comm_unique_id failed on rank 0 (no RCCL?)")
            raw = (C.c_ubyte * 128)(*ident[:128].tolist())
            info = self._lib.grb_comm_init(raw, rank, world)
            if info != 0:
                raise RuntimeError("grb_comm_init: Info %d" % info)
        elif (have_r.value, have_w.value) != (rank, world):
            raise RuntimeError("the library communicator is rank %d of %d" % (have_r.value, have_w.value))
        self.gathered = torch.zeros(world * nwords, dtype=torch.int32, device=dev) if world > 1 else None
        self._acc = torch.zeros(2, dtype=torch.float64, device=dev)
        if world > 1:
            self._self_test()

    def _self_test(self):
        """Every collective of this class once on known data; the ranks agree on the verdict through
        torch.distributed, so either all of them keep this communicator or all of them raise (and the caller
        falls back to TorchComm everywhere)."""
        rank, world, dev = self.rank, self.world, self.dev
        ok = True
        try:
            got = self.all_gather_padded(torch.full((16,), float(rank + 1), dtype=torch.float32, device=dev))
            want = torch.arange(1, world + 1, dtype=torch.float32, device=dev)[:, None].expand(world, 16)
            ok = ok and bool(torch.equal(got, want))
            # unequal slices in place: rank r owns r + 1 words
            offs = [r * (r + 1) // 2 for r in range(world + 1)]
            buf = torch.zeros(offs[world], dtype=torch.int32, device=dev)
            buf[offs[rank]:offs[rank + 1]] = rank + 1
            self.gather_slices_async(buf, [4 * offs[r] for r in range(world)], [4 * (r + 1) for r in range(world)])
            self.wait()
            want = torch.cat([torch.full((r + 1,), r + 1, dtype=torch.int32, device=dev) for r in range(world)])
            ok = ok and bool(torch.equal(buf, want))
            t = self.sum_(torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev))
            ok = ok and float(t.item()) == world * (world + 1) / 2
        except Exception:                                             # noqa: BLE001 -- reported through the verdict
            ok = False
        verdict = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if float(verdict.item()) != 1.0:
            raise RuntimeError("the library communicator failed its self-test on some rank")

    def or_combine(self, new_local, new_global, engine=None):
        if self.world == 1:
            new_global.copy_(new_local)
            return
        assert self._lib.grb_comm_allgather(new_local.data_ptr(), self.gathered.data_ptr(), 4 * self.nwords) == 0
        assert self._lib.grb_comm_wait() == 0                      # the compute stream waits, not the host
        engine.or_parts(self.gathered, self.world, self.nwords, new_global)

    def gather_word_slices(self, bitmap, word_bounds):
        """In-place all-gather of every rank's own word range of a replicated bitmap (pull levels: a rank only
        discovers vertices it owns, so the slices are disjoint -- no OR pass, 1/P of the bytes per rank)."""
        if self.world == 1:
            return
        off = (C.c_longlong * self.world)(*[4 * word_bounds[r] for r in range(self.world)])
        cnt = (C.c_longlong * self.world)(*[4 * (word_bounds[r + 1] - word_bounds[r]) for r in range(self.world)])
        assert self._lib.grb_comm_allgatherv_inplace(bitmap.data_ptr(), off, cnt) == 0
        assert self._lib.grb_comm_wait() == 0

    def gather_slices_async(self, buf, offsets_bytes, counts_bytes):
        """Enqueue an in-place all-gather of unequal slices; returns at once (grb_comm_wait fences it later)."""
        off = (C.c_longlong * self.world)(*offsets_bytes)
        cnt = (C.c_longlong * self.world)(*counts_bytes)
        assert self._lib.grb_comm_allgatherv_inplace(buf.data_ptr(), off, cnt) == 0

    def wait(self):
        assert self._lib.grb_comm_wait() == 0

    def sum_(self, t):
        if self.world > 1:
            acc = t.to(torch.float64).contiguous()
            assert self._lib.grb_comm_allreduce_sum_f64(acc.data_ptr(), acc.numel()) == 0
            assert self._lib.grb_comm_wait() == 0
            t.copy_(acc.to(t.dtype))
        return t

    def all_gather_padded(self, pad):
        out = torch.zeros(self.world * pad.numel(), dtype=pad.dtype, device=pad.device)
        if self.world == 1:
            out.copy_(pad)
        else:
            assert self._lib.grb_comm_allgather(pad.data_ptr(), out.data_ptr(), pad.numel() * pad.element_size()) == 0
            assert self._lib.grb_comm_wait() == 0
        return out.view(self.world, pad.numel())

    def timing(self, on):
        self._lib.grb_comm_timing(1 if on else 0)

    def stats(self, reset=True):
        us, calls = C.c_double(0), C.c_longlong(0)
        self._lib.grb_comm_stats(C.byref(us), C.byref(calls), 1 if reset else 0)
        return us.value, calls.value


class HostStagedComm(RcclComm):
    """The library communicator over HOST-STAGED collectives (grb_comm_set_host_transport): every grb_comm_* call of
    the library -- the ones this class makes and the ones the device-side level / round loops make themselves --
    becomes a torch.distributed collective on CPU tensors over pinned staging buffers.  For process groups RCCL cannot
    serve: gloo, or several ranks on ONE GPU (tests/test_gpu_part_run.py drives grb_bfs_part_run / grb_sssp_part_run
    from two processes this way).  Synchronous and slow on purpose."""

    def __init__(self, rank, world, nwords, dev, group=None):
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.nwords, self.dev = rank, world, nwords, dev
        self._group = group
        fn_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong,
                           C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))

        def view(ptr, nbytes, dtype=np.uint8):
            raw = (C.c_ubyte * int(nbytes)).from_address(ptr)
            return torch.from_numpy(np.frombuffer(raw, dtype=dtype))

        def transport(user, op, send, recv, nbytes, offsets, counts):
            try:
                if op == 0:
                    dist.all_gather_into_tensor(view(recv, nbytes * world), view(send, nbytes), group=group)
                elif op == 1:
                    for r in range(world):
                        if counts[r] > 0:
                            dist.broadcast(view(recv + offsets[r], counts[r]), src=dist.get_global_rank(group, r)
                                           if group is not None else r, group=group)
                elif op == 2:
                    dist.all_reduce(view(recv, nbytes, np.float64), op=dist.ReduceOp.SUM, group=group)
                else:
                    return 2
                return 0
            except Exception:                                            # noqa: BLE001 -- reported as GrB_PANIC
                import traceback
                traceback.print_exc()
                return 1
        self._cb = fn_t(transport)                                       # kept alive: the library holds the pointer
        info = self._lib.grb_comm_set_host_transport(int(rank), int(world), C.cast(self._cb, C.c_void_p), None)
        if info != 0:
            raise RuntimeError("grb_comm_set_host_transport: Info %d" % info)
        self.gathered = torch.zeros(world * nwords, dtype=torch.int32, device=dev) if world > 1 else None
        self._acc = torch.zeros(2, dtype=torch.float64, device=dev)
        if world > 1:
            self._self_test()

    def close(self):
        self._lib.grb_comm_set_host_transport(0, 1, None, None)


class LoopbackGroup:
    """Every rank of a world of `world` on ONE device, driven in lock-step by grb_bfs_part_run_group with device copies
    as the all-gather: what a multi-GPU run executes per rank (the same launches, the same apply over `world`
    gathered bitmaps) minus RCCL.  Used by the tests (N > 1 logic on a one-GPU box) and by tools/part_scaling.py
    (per-rank compute time as a function of the world size)."""

    def __init__(self, n, tptr, tind, world, dev, in_edges=None):
        from . import _lib
        self._lib = _lib.load()
        self.n, self.world, self.dev = n, world, dev
        ptr_host = tptr.cpu().numpy()
        self._ptr_host = ptr_host
        self.bounds = partition_bounds(ptr_host, world)
        self.nnz = int(ptr_host[-1])
        self.deg_full = (tptr[1:] - tptr[:-1]).to(torch.int32).contiguous()
        self.engines, self.labels, handles = [], [], []
        for r in range(world):
            lo, hi = self.bounds[r], self.bounds[r + 1]

            def shard(p, i):
                ph = p.cpu().numpy()
                e0, e1 = int(ph[lo]), int(ph[hi])
                lp = (p[lo:hi + 1] - e0).to(torch.int32).contiguous()
                li = i[e0:e1].to(torch.int32).contiguous()
                if li.numel() == 0:
                    li = torch.zeros(1, dtype=torch.int32, device=dev)
                return lp, li
            lptr, lind = shard(tptr, tind)
            if in_edges is None:
                eng = HipEngine(n, lo, lptr, lind, dev)
            else:
                ip, ii = shard(in_edges[0], in_edges[1])
                eng = HipEngine(n, lo, lptr, lind, dev, ip, ii)
            handles.append(eng.part_context(r, world, self.deg_full, self.nnz))
            self.engines.append(eng)
            self.labels.append(torch.zeros(max(hi - lo, 1), dtype=torch.float32, device=dev))
        self._handles = (C.c_void_p * world)(*[h.value for h in handles])
        self._label_ptrs = (C.c_void_p * world)(*[t.data_ptr() for t in self.labels])

    def bfs(self, source, mxvmode=GRB_PUSHPULL, switchpoint=0.01, edgeswitch=0.0, max_niter=10000):
        """-> (labels of the whole graph [n] as numpy, per-rank result dicts, trace of rank 0)"""
        from ._lib import PartBfsResult, BfsLevel
        res = (PartBfsResult * self.world)()
        cap = 1 << 15
        lv = (BfsLevel * cap)()
        info = self._lib.grb_bfs_part_run_group(self._handles, self.world, int(source), int(mxvmode),
                                                float(np.float32(switchpoint)), float(edgeswitch), int(max_niter),
                                                self._label_ptrs, res, lv, cap)
        if info != 0:
            raise RuntimeError("grb_bfs_part_run_group: Info %d" % info)
        labels = np.concatenate([self.labels[r][:self.bounds[r + 1] - self.bounds[r]].cpu().numpy()
                                 for r in range(self.world)])
        out = [dict(levels=int(x.levels), launches=int(x.launches), hit_cap=int(x.hit_cap),
                    edges_traversed=int(x.edges_traversed), reached=int(x.reached), device_ms=float(x.ms)) for x in res]
        trace = [("pull" if lv[i].direction else "push", int(lv[i].frontier), int(lv[i].discovered))
                 for i in range(min(out[0]["levels"], cap))]
        return labels, out, trace


    def sssp(self, weights, source, max_niter=10000, outbox_pairs=65536):
        """weights: of every stored out-edge, CSR order.  -> (distances [n] numpy, per-rank result dicts)"""
        from ._lib import PartSsspResult
        key = (weights.data_ptr(), int(weights.numel()), float(weights.double().sum().item()), int(outbox_pairs))
        if getattr(self, "_sssp_key", None) != key:
            ptr_host = self._ptr_host
            hs = []
            for r, eng in enumerate(self.engines):
                e0, e1 = int(ptr_host[self.bounds[r]]), int(ptr_host[self.bounds[r + 1]])
                hs.append(eng.sssp_context(r, self.world, weights[e0:e1], outbox_pairs))
            self._sssp_handles = (C.c_void_p * self.world)(*[h.value for h in hs])
            self._sssp_key = key
            self._dist = [torch.empty(max(self.bounds[r + 1] - self.bounds[r], 1), dtype=torch.float32, device=self.dev)
                          for r in range(self.world)]
            self._dist_ptrs = (C.c_void_p * self.world)(*[t.data_ptr() for t in self._dist])
        res = (PartSsspResult * self.world)()
        info = self._lib.grb_sssp_part_run_group(self._sssp_handles, self.world, int(source), int(max_niter),
                                                 self._dist_ptrs, res)
        if info != 0:
            raise RuntimeError("grb_sssp_part_run_group: Info %d" % info)
        d = np.concatenate([self._dist[r][:self.bounds[r + 1] - self.bounds[r]].cpu().numpy() for r in range(self.world)])
        return d, [dict(iterations=int(x.iterations), rounds=int(x.rounds), launches=int(x.launches), hit_cap=int(x.hit_cap),
                        device_ms=float(x.ms)) for x in res]


class Partition1D:
    def __init__(self, n, tptr, tind, rank, world, dev, engine_cls=HipEngine, mxvmode=GRB_PUSHPULL,
                 switchpoint=0.01, max_niter=10000, symmetric=True, comm=None, edgeswitch=0.0, in_edges=None,
                 device_loop=True, levels_per_launch=1):
        """tptr / tind: the whole graph's CSR (out-edges).  A directed graph also passes in_edges = (cptr, cind),
        its CSC: every rank then holds the out-edge rows (push) AND the in-edge rows (pull, PageRank) of the
        vertices it owns.  A symmetric graph needs one shard for both."""
        if not symmetric and in_edges is None:
            raise NotImplementedError("a directed graph needs its in-edges too: pass in_edges=(csc_ptr, csc_ind)")
        self.n, self.rank, self.world, self.dev = n, rank, world, dev
        ptr_host = tptr.cpu().numpy()
        self.bounds = partition_bounds(ptr_host, world)
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]

        def shard(p, i):
            ph = p.cpu().numpy()
            e0, e1 = int(ph[self.lo]), int(ph[self.hi])
            lp = (p[self.lo:self.hi + 1] - e0).to(torch.int32).contiguous()
            li = i[e0:e1].to(torch.int32).contiguous()
            if li.numel() == 0:
                li = torch.zeros(1, dtype=torch.int32, device=dev)
            return lp, li, (e0, e1)
        lptr, lind, out_range = shard(tptr, tind)
        self.lptr, self.lind, self.out_range = lptr, lind, out_range
        if in_edges is None:
            self.in_lptr, self.in_lind, self.in_range = lptr, lind, out_range
            self.engine = engine_cls(n, self.lo, lptr, lind, dev)
        else:
            self.in_lptr, self.in_lind, self.in_range = shard(in_edges[0], in_edges[1])
            self.engine = engine_cls(n, self.lo, lptr, lind, dev, self.in_lptr, self.in_lind)
        self.n_local = self.hi - self.lo
        self.nwords = bitmap_words(n)
        z = lambda: torch.zeros(self.nwords, dtype=torch.int32, device=dev)
        self.vis, self.new_local, self.new_global = z(), z(), z()
        self.comm = comm if comm is not None else TorchComm(world, self.nwords, dev)
        self.label = torch.zeros(max(self.n_local, 1), dtype=torch.float32, device=dev)
        self.deg_host = np.diff(ptr_host[self.lo:self.hi + 1])
        # graphblast_amd's edge-aware extension (0 = the reference's vertex-count rule only): a sparse frontier
        # whose out-edges exceed edgeswitch * nnz is pulled.  Needs the frontier's out-degree sum, which apply2
        # computes on every rank from the replicated new-bits bitmap (no collective).
        self.edgeswitch = float(edgeswitch)
        self.nnz = int(ptr_host[-1])
        self.deg_full = None
        if self.edgeswitch > 0 and hasattr(self.engine, "apply2"):
            self.deg_full = (tptr[1:] - tptr[:-1]).to(torch.int32).contiguous()
        self.deg_source = lambda s: int(ptr_host[s + 1] - ptr_host[s])
        # the engine's own level loop (HipEngine: one launch + one all-gather per level, nothing read back until
        # the end, csrc/bfs_part_run.hip).  It needs every vertex's out-degree on the device, and a communicator
        # the library can drive itself: its own RCCL one, or none at all for a world of one rank.
        self.device_loop = (hasattr(self.engine, "run_bfs") and device_loop
                            and (world == 1 or isinstance(self.comm, RcclComm)))
        self.levels_per_launch = int(levels_per_launch)
        if self.device_loop and self.deg_full is None:
            self.deg_full = (tptr[1:] - tptr[:-1]).to(torch.int32).contiguous()
        import inspect
        self._combine_takes_engine = "engine" in inspect.signature(self.comm.or_combine).parameters
        self.mxvmode, self.switchpoint, self.max_niter = mxvmode, float(np.float32(switchpoint)), max_niter

    def _combine(self):
        if self._combine_takes_engine:
            self.comm.or_combine(self.new_local, self.new_global, self.engine)
        else:                                              # communicators with the two-argument form
            self.comm.or_combine(self.new_local, self.new_global)

    def bfs(self, source, want_trace=True):
        n = self.n
        eng = self.engine
        if self.device_loop:
            part = eng.part_context(self.rank, self.world, self.deg_full, self.nnz)
            res, trace = eng.run_bfs(part, source, self.mxvmode, self.switchpoint, self.edgeswitch, self.max_niter,
                                     self.label, self.levels_per_launch, want_trace)
            return dict(levels=int(res.levels), edges_traversed=int(res.edges_traversed), reached=int(res.reached),
                        trace=trace, launches=int(res.launches), device_ms=float(res.ms))
        if hasattr(eng, "seed"):
            eng.seed(self.vis, self.new_global, self.label, source)
        else:
            self.vis.zero_()
            self.label.zero_()
            self.new_global.zero_()
            word, bit = source >> 5, source & 31
            seed = (1 << bit) if bit < 31 else -(1 << 31)
            self.new_global[word] = seed
            self.vis[word] = seed
            if self.lo <= source < self.hi:
                self.label[source - self.lo] = 1.0
        # out-edges of this rank's share of the frontier (decides locally which push kernel runs)
        local_edges = int(self.deg_host[source - self.lo]) if self.lo <= source < self.hi else 0
        all_edges = self.deg_source(source)
        f1_dense = self.mxvmode == GRB_PULLONLY
        ratio_f1 = ratio_f2 = np.float32(0)
        nf, levels, trace = 1, 0, []
        it = 1
        while it <= self.max_niter:
            if self.mxvmode == GRB_PUSHPULL:                       # vector.hpp:291-323
                ratio = np.float32(nf) / np.float32(n)
                if not f1_dense:
                    if ratio > np.float32(self.switchpoint) and ratio > ratio_f1:
                        f1_dense = True
                    else:
                        ratio_f1 = ratio
                else:
                    if ratio <= np.float32(self.switchpoint) and ratio < ratio_f1:
                        f1_dense = False
                    else:
                        ratio_f1 = ratio
            else:
                f1_dense = self.mxvmode == GRB_PULLONLY
            if (not f1_dense and self.mxvmode == GRB_PUSHPULL and self.deg_full is not None and self.edgeswitch > 0 and nf >= 32
                    and all_edges > self.edgeswitch * self.nnz):
                f1_dense = True                                    # the same rule as bfs_persist.hip:145-147
            if (f1_dense and hasattr(self.comm, "gather_word_slices") and getattr(eng, "pull_zeroes", False)
                    and word_slices_usable(self.bounds)):
                # pull discovers owned vertices only: written straight into the replicated bitmap, then ONE
                # in-place all-gather of the ranks' own word ranges (no OR pass, 1/P of the bytes)
                eng.pull(self.vis, self.new_global, self.label, it + 1)
                # (partition_bounds clamps an interior bound to n: with n % 64 >= 32 and a heavy tail vertex a
                # bound can fall inside a word -- then the word has two owners and the OR path below is taken)
                self.comm.gather_word_slices(self.new_global, [b // 32 for b in self.bounds[:-1]] + [self.nwords])
            else:
                if f1_dense:
                    if not getattr(eng, "pull_zeroes", False):
                        self.new_local.zero_()
                    eng.pull(self.vis, self.new_local, self.label, it + 1)
                elif hasattr(eng, "push_small") and local_edges <= eng.small_push_edges:
                    eng.push_small(self.new_global, self.vis, self.new_local)
                else:
                    eng.push(self.new_global, self.vis, self.new_local)
                self._combine()
            if hasattr(eng, "apply2"):
                found, local_edges, all_edges = eng.apply2(self.new_global, self.vis, self.label, it + 1, self.deg_full)
            else:
                found = eng.apply(self.new_global, self.vis, self.label, it + 1)
            trace.append(("pull" if f1_dense else "push", nf, found))
            levels += 1
            ratio_f1, ratio_f2 = ratio_f2, ratio_f1
            nf = found
            if nf == 0:
                break
            it += 1
        if it > self.max_niter and nf > 0:                          # bfs.hpp:48-66: never assigned
            if hasattr(eng, "unlabel"):
                eng.unlabel(self.label, float(self.max_niter + 1))
            else:
                self.label[self.label == float(self.max_niter + 1)] = 0.0
        e, r = self.engine.tally(self.label)
        if self.world > 1:                                           # one collective, one read-back
            t = torch.tensor([e, r], dtype=torch.int64, device=self.dev)
            self.comm.sum_(t)
            e, r = (int(x) for x in t.tolist())
        return dict(levels=levels, edges_traversed=int(e), reached=int(r), trace=trace)

    def pagerank(self, deg_full, alpha=0.85, eps=1e-8, max_niter=10):
        """algorithm::pr (graphblas/algorithm/pr.hpp:15-94) on the 1-D partition: every rank
        computes its owned slice of p . A from the replicated p (local SpMV over its in-edge
        shard, values alpha / outdeg), the slices are all-gathered into the next p and the
        squared residual is all-reduced.  deg_full: out-degree of every vertex (any float tensor
        of length n on this rank's device)."""
        n, dev = self.n, self.dev
        eng = self.engine
        lptr, lind = self.in_lptr, self.in_lind
        if isinstance(self.comm, RcclComm) and hasattr(eng, "pr_setup_chunks"):
            return self._pagerank_overlapped(deg_full, alpha, eps, max_niter)
        vals = (alpha / deg_full.to(torch.float32)[lind[:int(lptr[-1].item())].long()]).to(torch.float32).contiguous()
        if vals.numel() == 0:
            vals = torch.zeros(1, dtype=torch.float32, device=dev)
        eng.pr_setup(vals, dev)
        n1 = max(self.n_local, 1)
        p = torch.full((n,), 1.0 / n, dtype=torch.float32, device=dev)
        y = torch.zeros(n1, dtype=torch.float32, device=dev)
        p_old = torch.zeros(n1, dtype=torch.float32, device=dev)
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        m = max(max(sizes), 1)
        pad = torch.zeros(m, dtype=torch.float32, device=dev)
        const = np.float32((np.float32(1.0) - np.float32(alpha)) / np.float32(n))
        error, it, errs = 1.0, 0, []
        while error > eps and it < max_niter:
            p_old[:self.n_local] = p[self.lo:self.hi]
            res = eng.pr_step(p, y, p_old, const)
            t = self.comm.sum_(torch.tensor([res], dtype=torch.float64, device=dev))
            error = float(np.sqrt(np.float32(t[0].item())))
            errs.append(error)
            if self.world == 1:
                p[self.lo:self.hi] = y[:self.n_local]
            else:
                pad[:self.n_local] = y[:self.n_local]
                out = self.comm.all_gather_padded(pad)
                for r in range(self.world):
                    p[self.bounds[r]:self.bounds[r + 1]] = out[r, :sizes[r]]
            it += 1
        return p, dict(iterations=it, errors=errs)

    def _pagerank_overlapped(self, deg_full, alpha, eps, max_niter, nchunks=2):
        """The same iteration with the library's communicator: the owned rows are cut into row chunks; chunk
        c's slice of the next vector is all-gathered on the communication stream while chunk c + 1 is being
        multiplied on the compute stream (events, no host synchronisation); the squared residual is
        all-reduced behind the last gather and read once per iteration."""
        n, dev, eng, comm = self.n, self.dev, self.engine, self.comm
        lptr, lind = self.in_lptr, self.in_lind
        nnz_l = int(lptr[-1].item())
        vals = (alpha / deg_full.to(torch.float32)[lind[:nnz_l].long()]).to(torch.float32).contiguous()
        if vals.numel() == 0:
            vals = torch.zeros(1, dtype=torch.float32, device=dev)
        # the chunk matrices (and their SpMV plans: ~50 ms of preparation on RMAT-22) are kept between calls on the
        # same degrees / alpha -- every rank takes the same branch, so the set-up all-gather stays symmetric
        # (the sum guards against another tensor that happens to sit at a freed one's address)
        # only values that are the same on every rank go in the key (deg_full is replicated): an address is
        # rank-local, and one rank hitting while another misses would leave the set-up all-gather unmatched
        dsum = deg_full.double()
        key = (float(alpha), int(nchunks), int(deg_full.numel()), float(dsum.sum().item()),
               float((dsum * torch.arange(1, dsum.numel() + 1, device=dsum.device, dtype=torch.float64)).sum().item()))
        if getattr(self, "_pr_key", None) != key:
            cuts = eng.pr_setup_chunks(vals, dev, nchunks)
            # every rank's chunk boundaries (vertex ids), identical on all ranks: one small all-gather at set-up
            mine = torch.tensor([self.lo + c for c in cuts], dtype=torch.float64, device=dev)
            self._pr_allc = comm.all_gather_padded(mine).cpu().numpy().astype(np.int64)   # [world, nchunks + 1]
            self._pr_key = key
        allc = self._pr_allc
        g = eng.g
        lib = eng._lib
        p_cur = torch.full((n,), 1.0 / n, dtype=torch.float32, device=dev)
        p_next = torch.empty_like(p_cur)
        y = torch.zeros(max(self.n_local, 1), dtype=torch.float32, device=dev)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        const = float(np.float32((np.float32(1.0) - np.float32(alpha)) / np.float32(n)))
        # the loop itself runs in the library (grb_pr_part_run): SpMV + update per chunk, the chunk's slice gathered
        # behind it, one residual read per iteration
        nck = len(eng.pr_chunks)
        handles = (C.c_void_p * nck)(*[M._h.value for _, _, M in eng.pr_chunks])
        row_cut = (C.c_longlong * (nck + 1))(*([a for a, _, _ in eng.pr_chunks] + [eng.pr_chunks[-1][1]]))
        vcut = (C.c_longlong * (self.world * (nck + 1)))(*[int(x) for x in np.asarray(allc).reshape(-1)])
        n_it, in_next = C.c_int(0), C.c_int(0)
        err_buf = (C.c_double * max(int(max_niter), 1))()
        torch.cuda.synchronize()
        t_loop = time.perf_counter()
        info = lib.grb_pr_part_run(nck, handles, row_cut, vcut, int(self.lo), const, float(eps), int(max_niter),
                                   p_cur.data_ptr(), p_next.data_ptr(), y.data_ptr(), acc.data_ptr(),
                                   C.byref(n_it), err_buf, C.byref(in_next))
        assert info == 0, info
        it = n_it.value
        errs = [float(err_buf[i]) for i in range(it)]
        if in_next.value:
            p_cur, p_next = p_next, p_cur
        torch.cuda.synchronize()
        return p_cur, dict(iterations=it, errors=errs, overlapped_chunks=nchunks,
                           ms_iterations=(time.perf_counter() - t_loop) * 1e3)

    def sssp(self, in_weights, source, max_niter=None, out_weights=None, outbox_pairs=65536, rounds_per_launch=1):
        """algorithm::sssp (graphblas/algorithm/sssp.hpp:53-90) on the 1-D partition, as synchronous rounds:
        every rank relaxes the IN-edges of the vertices it owns against the replicated distance vector
        (MinimumPlus product over its in-edge shard -- round r+1 reads only round r's distances, so the distances
        after every round, and the number of rounds, are the reference's), keeps the minimum with its own slice,
        the slices are all-gathered into the next vector and the number of improved vertices is all-reduced;
        the loop ends with the first round that improves nothing.
        in_weights: the weight of every stored in-edge of the WHOLE graph, in the order of the in-edge arrays
        this partition was built from (the CSC's; for a symmetric graph the CSR's).  Non-negative f32.
        -> (distances of all vertices on every rank, FLT_MAX = unreached; {"iterations": rounds done})"""
        n, dev, eng = self.n, self.dev, self.engine
        fmax = float(np.finfo(np.float32).max)
        max_niter = self.max_niter if max_niter is None else max_niter
        if out_weights is None and self.in_lptr is self.lptr:
            out_weights = in_weights                                  # a symmetric graph: one shard, one weight array
        if self.device_loop and hasattr(eng, "run_sssp") and out_weights is not None:
            # the frontier form with the round loop on the device (csrc/sssp_part_run.hip): per round the
            # improved vertices' out-edges, an all-gather of (vertex, candidate) pairs, nothing read back
            o0, o1 = self.out_range
            key = (out_weights.data_ptr(), int(outbox_pairs), int(out_weights.numel()), float(out_weights.double().sum().item()))
            if getattr(self, "_sssp_key", None) != key:
                self._sssp_ctx = eng.sssp_context(self.rank, self.world, out_weights[o0:o1], outbox_pairs)
                self._sssp_key = key
            d_local = torch.empty(max(self.n_local, 1), dtype=torch.float32, device=dev)
            res = eng.run_sssp(self._sssp_ctx, source, max_niter, d_local, rounds_per_launch)
            if self.world == 1:
                d = d_local[:self.n_local].clone()
            else:
                sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
                pad = torch.zeros(max(max(sizes), 1), dtype=torch.float32, device=dev)
                pad[:self.n_local] = d_local[:self.n_local]
                out = self.comm.all_gather_padded(pad)
                d = torch.cat([out[r, :sizes[r]] for r in range(self.world)])
            return d, dict(iterations=int(res.iterations), rounds=int(res.rounds), launches=int(res.launches),
                           device_ms=float(res.ms), form="frontier, round loop on the device")
        e0, e1 = self.in_range
        w = in_weights[e0:e1].to(torch.float32).contiguous()
        if w.numel() == 0:
            w = torch.zeros(1, dtype=torch.float32, device=dev)
        eng.sssp_setup(w, dev)
        d = torch.full((n,), fmax, dtype=torch.float32, device=dev)
        d[source] = 0.0
        n1 = max(self.n_local, 1)
        d_local = torch.full((n1,), fmax, dtype=torch.float32, device=dev)
        d_local[:self.n_local] = d[self.lo:self.hi]
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        pad = torch.zeros(max(max(sizes), 1), dtype=torch.float32, device=dev)
        it, improved = 0, []
        for it in range(1, max_niter + 1):
            changed = eng.sssp_step(d, d_local)                      # d_local = min(d_local, A_in min.+ d)
            total = int(self.comm.sum_(torch.tensor([float(changed)], dtype=torch.float64, device=dev))[0].item())
            improved.append(total)
            if total == 0:
                break
            if self.world == 1:
                d[self.lo:self.hi] = d_local[:self.n_local]
            else:
                pad[:self.n_local] = d_local[:self.n_local]
                out = self.comm.all_gather_padded(pad)
                for r in range(self.world):
                    d[self.bounds[r]:self.bounds[r + 1]] = out[r, :sizes[r]]
        else:
            it = max_niter + 1                                       # the reference's loop counter after a cut-off
        return d, dict(iterations=it, improved=improved)

    def gather_labels(self):
        """Full label vector on every rank (tests / verification only)."""
        if self.world == 1:
            return self.label[:self.n_local].clone()
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        m = max(sizes)
        pad = torch.zeros(m, dtype=torch.float32, device=self.dev)
        pad[:self.n_local] = self.label[:self.n_local]
        out = self.comm.all_gather_padded(pad)
        return torch.cat([out[r, :sizes[r]] for r in range(self.world)])
