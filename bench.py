#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: direction-optimised BFS TEPS (+ SpMV
achieved HBM GB/s) on RMAT scale 22, edge factor 16.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = one direction-optimised BFS (algorithm::bfs, --mxvmode 0 --struconly 1
--opreuse 1 --earlyexit 1, the authors' flag set of run_bfs.sh) from the next source of a
fixed seeded source list over the RMAT-22 graph already resident in HBM.  value = directed
stored edges traversed (sum of out-degree over reached vertices) per
second over the K timed steps, whole job.  At N > 1 the graph is 1-D vertex partitioned
(graphblast_amd/dist.py) and the same K traversals run cooperatively ("strong" scaling).

Rank 0 prints ONE JSON line.  Extra objects on it:
  roofline      dominant kernel of the timed region = bfs_persistent_kernel (one launch runs a
                whole traversal): algorithmic bytes per launch (summed over
                the levels) / mean launch duration from HIP events recorded on the library's
                stream in a second pass over the same steps
  bfs_total     the kernel's own wall-clock time per traversal and the per-level table of one
  spmv          the generic SpMV kernel (PlusMultiplies, f32) on the same graph: algorithmic
                8*nnz + 12*n + 4 bytes per launch / HIP-event mean launch time
  spmv_grid4096 the same SpMV kernel on a road-like 4096^2 grid (local gathers)
  bfs_batch64   all 64 sources in one bit-parallel sweep (grb_bfs_batch), labels spot-checked against the
                single traversal; spmm_k64: sparse x dense mxm with 64 right-hand sides (grb_spmm)
  primitives    eWiseAdd / eWiseMult / reduce / assign on 64 Mi-element f32 vectors: GB/s and
                fraction of the 8 TB/s HBM peak
  cpu_all_cores context only (not the reference, whose CPU path is sequential): the same labels from an
                OpenMP BFS on every host core
  cpu_baseline  the reference's own SimpleReferenceBfs (oracle/_ref/libsimple_ref.so; the restatement in
                oracle/simple_reference.c when that library is absent), one host core, on a bounded
                sample of the same workload (rank 0, N = 1 only)
  parity        the labels of every source of that sample from the HIP path compared bit-exactly with
                the CPU's, and the reached / edge counts of the timed steps; a mismatch fails the run
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def pmc_traffic(kernel_key, workload="rmat22_bfs"):
    """(HBM bytes per launch, where it came from).  PMC counters cannot be read from inside this process; the
    number is the one recorded by the latest committed PMC passes (profiles/rNN/pmc_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, FETCH_SIZE doubled as
    calibrated there on kernels of known byte count) -- the source is named on the line so a stale value
    cannot pass for a live one; (None, None) when no pass has been committed.  The passes of the other
    workloads are kept under "workloads": {name: {"kernels": ...}} in the same file."""
    best, where = None, None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "pmc_traffic.json")
        if os.path.exists(f):
            doc = json.load(open(f))
            kernels = doc.get("kernels", {}) if workload == "rmat22_bfs" else doc.get("workloads", {}).get(workload, {}).get("kernels", {})
            for name, rec in kernels.items():
                if kernel_key in name:
                    best = rec.get("hbm_bytes_per_launch", rec.get("hbm_bytes_per_launch_raw"))
                    where = "profiles/%s/pmc_traffic.json (separate rocprofv3 --pmc passes, not this run)" % d
    return best, where


def pmc_per_traversal(kernel_key):
    """(HBM bytes per TRAVERSAL of a launch of several, source): the launches of a PMC pass carry different numbers of
    traversals, so tools/summarize_profiles.py divides their summed counters by the traversals they ran"""
    best, where = None, None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "pmc_traffic.json")
        if os.path.exists(f):
            for name, rec in json.load(open(f)).get("kernels", {}).items():
                if kernel_key in name and rec.get("hbm_bytes_per_traversal"):
                    best = rec["hbm_bytes_per_traversal"]
                    where = "profiles/%s/pmc_traffic.json (separate rocprofv3 --pmc passes, not this run)" % d
    return best, where


def pmc_group(name, workload=None):
    """(HBM bytes per unit of a group of kernels -- e.g. all batch_* launches of one 64-source sweep -- , source)"""
    best, where = None, None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "pmc_traffic.json")
        if os.path.exists(f):
            doc = json.load(open(f))
            rec = (doc if workload is None else doc.get("workloads", {}).get(workload, {})).get("groups", {}).get(name)
            if rec:
                best = rec.get("hbm_bytes_per_unit")
                where = "profiles/%s/pmc_traffic.json (separate rocprofv3 --pmc passes, not this run)" % d
    return best, where


def level_bytes(levels, n):
    """push 12 nf + 8 mf + 8 nf'; pull 4 n + 8 nu + 8 mi + 4 nf'."""
    out = []
    reached = 1
    for L in levels:
        nf, nfn = L["frontier"], L["discovered"]
        if L["direction"] == "push":
            b = 12 * nf + 8 * L["frontier_edges"] + 8 * nfn
        else:
            nu = n - reached
            b = 4 * n + 8 * nu + 8 * L["frontier_edges"] + 4 * nfn
        reached += nfn
        out.append(b)
    return out


def other_workload(args):
    """BASELINE.json's other configurations at their own size, one JSON line each (N = 1):
      lj_bfs     config 2: direction-optimised BFS on soc-LiveJournal1 ($GRB_DATA) or RMAT-22 ef 16 DIRECTED
      road_sssp  config 3: MinimumPlus SSSP on road_usa ($GRB_DATA) or a 4896^2 grid with 40 % of the edges removed
      orkut_tc   config 5: triangle count (masked L * L^T) on com-Orkut ($GRB_DATA) or RMAT-22 ef 28 symmetrised"""
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges, random_sources
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    data = os.environ.get("GRB_DATA")

    def have(name):
        p = os.path.join(data, name + ".mtx") if data else None
        return p if p and os.path.exists(p) else None
    line = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "data": "synthetic", "cpu_baseline": None}
    if args.workload == "lj_bfs":
        path = have("soc-LiveJournal1")
        if path:
            A = g.Matrix.from_mtx(path, directed=0)
            n = A.nrows()
            ptr = A.host_csr()[0]
            nnz = int(ptr[-1])
            line["data"] = "soc-LiveJournal1.mtx"
        else:
            s, d, n = rmat_edges(22, 16, seed=1, device=dev)
            gr = finalize_edges(s, d, n, symmetrize=False)
            tptr, tind = gr["csr"]
            cptr, cind = gr["csc"]
            nnz = gr["nnz"]
            ones = torch.ones(nnz, dtype=torch.float32, device=dev)
            A = g.Matrix(n, n)
            assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), ones.data_ptr(), nnz, cptr.data_ptr(),
                                      cind.data_ptr(), ones.data_ptr(), keep=(tptr, tind, cptr, cind, ones)) == 0
            ptr = tptr.cpu().numpy()
        sources = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
        desc = g.Descriptor()
        assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=args.edgeswitch) == 0
        v = g.Vector(n)
        for i in range(args.warmup):
            g.bfs(v, A, sources[i % 64], desc, fused=True)
        # the timed steps are queued back to back and waited for afterwards, as in the headline run
        vs = [g.Vector(n) for _ in range(args.steps)]
        g.bfs_host_times(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tickets = [g.bfs_enqueue(vs[i], A, sources[i % 64], desc)[1] for i in range(args.steps)]
        res = [g.bfs_wait(t)[1] for t in tickets]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ht = g.bfs_host_times(reset=True)
        torch.cuda.synchronize()
        t0b = time.perf_counter()
        for i in range(args.steps):
            g.bfs(v, A, sources[i % 64], desc, fused=True)
        torch.cuda.synchronize()
        el_block = time.perf_counter() - t0b
        htb = g.bfs_host_times(reset=True)
        line["blocking_loop"] = {"ms_per_step": round(el_block / args.steps * 1e3, 5),
                                 "host_enqueue_us_per_step": round(htb["enqueue_us"] / max(htb["calls"], 1), 2),
                                 "host_wait_us_per_step": round(htb["wait_us"] / max(htb["calls"], 1), 2)}
        line["queued"] = {"host_enqueue_us_per_step": round(ht["enqueue_us"] / max(ht["calls"], 1), 2)}
        # the same steps, twelve traversals side by side per launch (grb_bfs_set_coschedule, as in the headline run); every
        # vector's labels against the one-at-a-time steps'
        solo_labels = [x.extractTuples()[1] for x in vs]
        g.bfs_set_coschedule(12)
        for t_ in [g.bfs_enqueue(vs[i], A, sources[i % 64], desc)[1] for i in range(min(args.steps, 16))]:
            g.bfs_wait(t_)
        torch.cuda.synchronize()
        t0c = time.perf_counter()
        tickets = [g.bfs_enqueue(vs[i], A, sources[i % 64], desc)[1] for i in range(args.steps)]
        res_c = [g.bfs_wait(t)[1] for t in tickets]
        torch.cuda.synchronize()
        el_c = time.perf_counter() - t0c
        g.bfs_set_coschedule(1)
        assert [r_["reached"] for r_ in res_c] == [r_["reached"] for r_ in res]
        assert all(np.array_equal(x.extractTuples()[1], w_) for x, w_ in zip(vs, solo_labels)), "co-scheduled labels differ"
        line["coscheduled_12"] = {"value": sum(r_["edges_traversed"] for r_ in res_c) / el_c, "unit": "TEPS",
                                 "ms_per_step": round(el_c / args.steps * 1e3, 5),
                                 "labels": "all %d vectors equal to the one-at-a-time steps'" % len(vs)}
        del vs, solo_labels
        ev = sum(g.bfs(v, A, sources[i % 64], desc, fused=True, profile=1)[1]["tight_ms"] for i in range(args.steps))
        acct = {s_: g.bfs(v, A, s_, desc, fused=True, profile=3)[1]["per_level"] for s_ in set(sources[i % 64] for i in range(args.steps))}
        tb = float(sum(sum(level_bytes(acct[sources[i % 64]], n)) for i in range(args.steps)))
        kern = "bfs_persistent_kernel"
        line.update({"metric": "BFS TEPS (edges/sec), direction-optimised, directed graph of soc-LiveJournal1's size",
                     "value": sum(r["edges_traversed"] for r in res) / el, "unit": "TEPS", "ms_per_step": el / args.steps * 1e3,
                     "dtype": "f32", "config": {"workload": "lj_bfs" if path else "rmat22_ef16_directed_do_bfs (stand-in)",
                                               "n": n, "nnz": nnz},
                     "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(tb / (ev * 1e-3) / 1e9, 2),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tb / (ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "traffic": pmc_traffic(kern + "<1024>", "lj_bfs")[0] or pmc_traffic(kern, "lj_bfs")[0],
                                  "traffic_source": pmc_traffic(kern + "<1024>", "lj_bfs")[1] or pmc_traffic(kern, "lj_bfs")[1],
                                  "avg_launch_ms": round(ev / args.steps, 5),
                                  "algorithmic_bytes_per_launch": int(tb / args.steps)}})
        if not args.no_cpu_baseline:
            # the reference's own SimpleReferenceBfs (oracle/_ref) on a bounded sample of the timed sources, one
            # core; the HIP labels of the same sources must equal its labels bit for bit
            from oracle import ref_simple, simple_reference as sr
            use_ref = ref_simple.available()
            if "tptr" in dir():
                ind_host = tind.cpu().numpy()
            else:
                ind_host = A.host_csr()[1]
            nsamp = min(16, len(sources))
            cpu_ms, cpu_edges, bad = 0.0, 0, 0
            deg = np.diff(ptr)
            for s_ in sources[:nsamp]:
                t0c = time.perf_counter()
                depth = (ref_simple.bfs(ptr, ind_host, s_)[0] if use_ref else sr.bfs(ptr, ind_host, s_)[0])
                cpu_ms += (time.perf_counter() - t0c) * 1e3
                cpu_edges += int(deg[depth != 0].sum())
                assert g.bfs(v, A, s_, desc, fused=True)[0] == 0
                if not np.array_equal(v.extractTuples()[1], depth):
                    bad += 1
            line["cpu_baseline"] = {"value": cpu_edges / (cpu_ms * 1e-3), "unit": "TEPS", "cores": 1,
                                    "kind": "reference" if use_ref else "port",
                                    "sample": "SimpleReferenceBfs, %d of the timed sources on the same graph" % nsamp,
                                    "ms_per_bfs": round(cpu_ms / nsamp, 2)}
            line["parity"] = {"checked_sources": nsamp, "mismatches": bad, "what": "depth labels bit-exact against the CPU reference"}
            if bad:
                print(json.dumps({"error": "parity", "workload": args.workload, "mismatching_sources": bad}))
                sys.exit(3)
    elif args.workload == "road_sssp":
        path = have("road_usa")
        if path:
            A0 = g.Matrix.from_mtx(path, directed=0)
            n = A0.nrows()
            hp, hi, _ = A0.host_csr()
            gptr, gind = torch.as_tensor(hp).to(dev), torch.as_tensor(hi).to(dev)
            nnz = int(hi.size)
            del A0
            line["data"] = "road_usa.mtx, integer weights 1..64"
        else:
            es, ed, n = grid_edges(4896, keep=0.6, seed=3)
            gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
            gptr, gind = gg["csr"]
            nnz = gg["nnz"]
        grow = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (gptr[1:] - gptr[:-1]).long())
        lo, hi_ = torch.minimum(grow, gind.long()), torch.maximum(grow, gind.long())
        gw = ((((lo * 1000003) ^ hi_) * 2654435761 >> 7) % 64 + 1).to(torch.float32)
        del grow, lo, hi_
        G = g.Matrix(n, n)
        assert G.build_device_csr(gptr.data_ptr(), gind.data_ptr(), gw.data_ptr(), nnz, gptr.data_ptr(), gind.data_ptr(),
                                  gw.data_ptr(), keep=(gptr, gind, gw)) == 0
        hp = gptr.cpu().numpy()
        src = int(np.nonzero(np.diff(hp))[0][len(hp) // 3])
        desc = g.Descriptor()
        assert desc.loadArgs(mxvmode=0, timing=0) == 0
        v = g.Vector(n)
        steps = max(1, min(args.steps, 3))
        g.sssp(v, G, src, desc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            info, res = g.sssp(v, G, src, desc)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        # what produced that: the work-efficient near / far order (csrc/sssp_nearfar.hip: same distances, same
        # round count as the reference's synchronous rounds) or the rounds themselves; time the rounds as well
        # and compare the two distance vectors bit for bit
        passes = g.sssp_last_order()
        dist_default = v.extractTuples()[1].copy()
        mode_before = g.sssp_set_nearfar(-2)
        g.sssp_set_nearfar(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info0, res0 = g.sssp(v, G, src, desc)
        torch.cuda.synchronize()
        el_rounds = time.perf_counter() - t0
        same = bool(np.array_equal(dist_default, v.extractTuples()[1])) and res0["iterations"] == res["iterations"]
        g.sssp_set_nearfar(mode_before)
        if not same:
            print(json.dumps({"error": "near / far and synchronous rounds disagree", "rounds": [res["iterations"], res0["iterations"]]}))
            sys.exit(3)
        d2 = g.Descriptor()
        assert d2.loadArgs(mxvmode=0, timing=1) == 0
        g.sssp(v, G, src, d2)
        # per-round frontier sizes (improved vertices) of the run: algorithmic bytes per round = the push formula
        # of BASELINE.md 3 with a weight per edge: 12 nf + 12 mf + 8 nf' (mf taken as nf x average degree)
        from graphblast_amd import _lib as _l
        import ctypes as _C

        class It(_C.Structure):
            _fields_ = [("iteration", _C.c_int32), ("direction", _C.c_int32), ("value", _C.c_double), ("ms", _C.c_float),
                        ("reserved", _C.c_int32)]
        k = _C.c_int(0)
        _l.load().grb_descriptor_iter_log(d2._h, None, 0, _C.byref(k))
        buf = (It * max(k.value, 1))()
        _l.load().grb_descriptor_iter_log(d2._h, buf, k.value, _C.byref(k))
        nfs = np.array([buf[i].value for i in range(k.value)], dtype=np.float64)
        avgdeg = nnz / float(n)
        alg = float(np.sum(12 * np.r_[1.0, nfs[:-1]] + 12 * avgdeg * np.r_[1.0, nfs[:-1]] + 8 * nfs))
        # the default path's own work, summed over its passes by the kernel itself (grb_sssp_last_work): vertices
        # expanded, their out-edges, vertices newly marked -- priced with the push formula of BASELINE.md 3 carrying a
        # weight per edge (12 nf + 12 mf + 8 nf')
        g.sssp(v, G, src, desc)
        nf_work = g.sssp_last_work() if passes else (0, 0, 0)
        alg_nf = float(12 * nf_work[0] + 12 * nf_work[1] + 8 * nf_work[2])
        kern = (("sssp_nfq_kernel" if os.environ.get("GRB_SSSP_QUEUE", "1") != "0" else "sssp_nearfar_kernel") if passes
                else "sssp_persistent_kernel")
        roof = ({"bound": "hbm", "kernel": kern, "achieved": round(alg_nf / el / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg_nf / el / 1e9 / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kern, "road_sssp")[0],
                 "traffic_source": pmc_traffic(kern, "road_sssp")[1], "algorithmic_bytes_per_launch": int(alg_nf),
                 "work": {"passes": passes, "vertices_expanded": nf_work[0], "edges_relaxed": nf_work[1], "vertices_marked": nf_work[2]},
                 "note": "one launch per SSSP; %d passes of ~%d expanded vertices each: the launch is bound by its "
                         "grid barrier and dependent memory steps per pass, not by bytes" % (passes, nf_work[0] // max(passes, 1))}
                if passes else
                {"bound": "hbm", "kernel": kern, "achieved": round(alg / el_rounds / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg / el_rounds / 1e9 / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kern, "road_sssp")[0],
                 "traffic_source": pmc_traffic(kern, "road_sssp")[1], "algorithmic_bytes_per_launch": int(alg)})
        line.update({"metric": "SSSP (MinimumPlus vxm) time on a road network of road_usa's size", "value": el * 1e3,
                     "unit": "ms", "higher_is_better": False, "ms_per_step": el * 1e3, "dtype": "f32", "steps": steps,
                     "config": {"workload": "road_sssp" if path else "grid4896_thinned_sssp (stand-in)", "n": n, "nnz": nnz,
                                "rounds": res["iterations"], "us_per_round": round(el * 1e6 / max(res["iterations"], 1), 2)},
                     "order": ("near / far, %d passes (%s)" % (passes, kern)) if passes else "synchronous rounds",
                     "synchronous_rounds": {"ms": round(el_rounds * 1e3, 2), "us_per_round": round(el_rounds * 1e6 / max(res0["iterations"], 1), 2),
                                            "distances_and_round_count_identical": same,
                                            "algorithmic_bytes": int(alg), "achieved_GBps": round(alg / el_rounds / 1e9, 2)},
                     "roofline": roof})
        if not args.no_cpu_baseline:
            # the reference's own SimpleReferenceSssp (a binary-heap Dijkstra, test_sssp.hpp:15-79), one core, the same
            # source: its distances must equal the HIP path's exactly (integer weights)
            from oracle import ref_simple, simple_reference as sr
            use_ref = ref_simple.available()
            hi_host, w_host = gind.cpu().numpy(), gw.cpu().numpy()
            t0c = time.perf_counter()
            want = (ref_simple.sssp(hp, hi_host, w_host, src)[0] if use_ref else sr.sssp(hp, hi_host, w_host, src)[0])
            cpu_ms = (time.perf_counter() - t0c) * 1e3
            ok = bool(np.array_equal(want, dist_default))
            line["cpu_baseline"] = {"value": cpu_ms, "unit": "ms", "cores": 1, "kind": "reference" if use_ref else "port",
                                    "sample": "SimpleReferenceSssp, the timed source, whole graph"}
            line["parity"] = {"checked_sources": 1, "mismatches": 0 if ok else 1,
                              "what": "distances bit-exact against the CPU reference (integer weights); near / far == rounds"}
            if not ok:
                print(json.dumps({"error": "parity", "workload": args.workload}))
                sys.exit(3)
    else:
        path = have("com-Orkut")
        if path:
            A0 = g.Matrix.from_mtx(path, dtype=np.int32, directed=0)
            n = A0.nrows()
            ptr, ind, _ = A0.host_csr()
            del A0
            line["data"] = "com-Orkut.mtx"
        else:
            s, d, n = rmat_edges(22, 28, seed=6, device=dev)
            gr = finalize_edges(s, d, n, symmetrize=True)
            ptr, ind = gr["csr"][0].cpu().numpy(), gr["csr"][1].cpu().numpy()
            del gr, s, d
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
        keep = ind <= rows
        lp = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows[keep], minlength=n), out=lp[1:])
        li = ind[keep]
        del rows, keep
        L = g.Matrix(n, n, np.int32)
        assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
        B = g.Matrix(n, n, np.int32)
        steps = max(1, min(args.steps, 3))
        desc = g.Descriptor()
        desc.loadArgs()
        dl = np.diff(lp).astype(np.float64)
        erow = np.repeat(np.arange(n, dtype=np.int64), np.diff(lp))
        # every mask entry (i, j) intersects rows i and j of L: the algorithmic bytes are both lists + the entry
        alg = float(4.0 * (np.sum(dl * dl) + np.sum(dl[li])) + 12.0 * li.size)
        # what the pivot kernels stream past their LDS tables: the SHORTER list of every mask entry
        shorter = float(np.minimum(dl[erow], dl[li]).sum())
        # ---- grb_tc as it is by default: the count on the degree-ordered orientation of the same edges, no product in B
        # (csrc/tc_count.hip; tc.hpp:17 calls B the buffer matrix).  The first call on a matrix prepares the orientation,
        # the matrix keeps it: `value` is a call on a matrix that has it, first_call_ms one that does not.
        g.tc_set_product(0)
        info, ntri, res = g.tc(L, B, desc)
        assert info == 0
        first_ms, first = res["tight_ms"], g.tc_last()[1]
        counted = first["path"] == 1
        ms, kern_ms = [], []
        for _ in range(steps):
            dd = g.Descriptor()
            dd.loadArgs()
            info, again, res = g.tc(L, B, dd)
            assert info == 0 and again == ntri
            ms.append(res["tight_ms"])
            kern_ms.append(g.tc_last()[1]["count_ms"])
        t = float(np.mean(ms)) * 1e-3
        # ---- the reference's two calls (mxm into B, reduce B): what grb_tc does for any other matrix or descriptor, and
        # after grb_tc_set_product(1)
        g.tc_set_product(1)
        g.tc(L, B, desc)
        pms = []
        for _ in range(steps):
            dd = g.Descriptor()
            dd.loadArgs()
            info, ntri_p, res = g.tc(L, B, dd)
            assert info == 0 and ntri_p == ntri, (ntri_p, ntri)
            pms.append(res["tight_ms"])
        tp = float(np.mean(pms)) * 1e-3
        del erow
        # compulsory HBM bytes of the product: L's structure read (as the left operand, the right operand and the mask: one
        # copy in memory), the result's values written, its structure copied; the intersections themselves re-read adjacency
        # lists that the L2 serves (alg: both lists of every mask entry)
        compulsory = float(4.0 * (n + 1) + 4.0 * li.size + 4.0 * li.size + 4.0 * (n + 1) + 4.0 * li.size)
        product = {"ms": round(tp * 1e3, 3), "shorter_list_elements": shorter,
                   "roofline": {"bound": "hbm", "kernel": "spgemm_pivot_block_kernel / spgemm_pivot_wave_kernel",
                                "achieved": round(compulsory / tp / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(compulsory / tp / 1e9 / HBM_PEAK_GBS, 5),
                                "traffic": pmc_group("masked_spgemm_call", "orkut_tc")[0], "traffic_source": pmc_group("masked_spgemm_call", "orkut_tc")[1],
                                "algorithmic_bytes_per_launch": int(compulsory),
                                "list_bytes_served_on_chip": int(alg), "list_GBps": round(alg / tp / 1e9, 1),
                                "note": "bytes = compulsory HBM traffic (operands once, result once); the kernel is bound "
                                        "by the on-chip rate of its list intersections (list_GBps), not by HBM"}}
        if counted:
            # the oriented lists: every vertex keeps its neighbours of higher degree; an edge is intersected once, the list
            # of the end with the shorter list streamed past an LDS bitmap (or hash table) of the other end's list
            key = np.diff(ptr).astype(np.int64) * n + (n - 1 - np.arange(n, dtype=np.int64))   # the library's order: degree, then id
            number = np.empty(n, dtype=np.int64)
            number[np.argsort(-key, kind="stable")] = np.arange(n, dtype=np.int64)
            number += number >= 65535                           # (nobody is numbered 65 535: the 16-bit parts' filling)
            allrows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
            up = number[ind] < number[allrows]                  # the column ranks above the row: an entry of list(row)
            lo_v, hi_v = allrows[up], ind[up].astype(np.int64)
            olen = np.bincount(lo_v, minlength=n).astype(np.float64)
            o16 = np.bincount(lo_v[number[hi_v] < 65535], minlength=n).astype(np.float64)
            lo_is_pivot = olen[hi_v] <= olen[lo_v]
            piv, par = np.where(lo_is_pivot, lo_v, hi_v), np.where(lo_is_pivot, hi_v, lo_v)
            wide = number[piv] > 65536                          # only these pivots stream their partners' 32-bit parts
            streamed = float(o16[par].sum() + (olen[par] - o16[par])[wide].sum())
            streamed_bytes = float(2.0 * o16[par].sum() + 4.0 * (olen[par] - o16[par])[wide].sum())
            del allrows, up, key, number, lo_v, hi_v, lo_is_pivot, piv, par, wide
            # compulsory bytes of the count: the lists once (2 or 4 B an edge), a partner descriptor an edge (16 B), a list
            # descriptor a vertex
            comp_c = float(2.0 * o16.sum() + 4.0 * (olen - o16).sum() + 16.0 * li.size + 16.0 * (n + 1))
            tk = float(np.mean(kern_ms)) * 1e-3
            line.update({"metric": "triangle count time on a graph of com-Orkut's size (grb_tc; the count on the degree-ordered "
                                   "orientation, the orientation kept by the matrix)",
                         "value": t * 1e3, "unit": "ms", "higher_is_better": False, "ms_per_step": t * 1e3, "dtype": "i32", "steps": steps,
                         "first_call_ms": round(first_ms, 3), "first_call_preparation_ms": round(first["prep_ms"], 3),
                         "config": {"workload": "orkut_tc" if path else "rmat22_ef28_sym_tc (stand-in)", "n": n, "nnz_L": int(li.size),
                                    "triangles": int(ntri), "longest_oriented_list": int(first["longest_list"]),
                                    "longest_row_of_L": int(dl.max()), "streamed_list_elements": streamed,
                                    "workgroups": {"wave_hash": first["tasks"][0], "bitmap": first["tasks"][1], "hash": first["tasks"][2]}},
                         "roofline": {"bound": "hbm", "kernel": "tc_count_bitmap_kernel + tc_count_small_kernel",
                                      "achieved": round(comp_c / tk / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(comp_c / tk / 1e9 / HBM_PEAK_GBS, 5),
                                      "traffic": pmc_group("tc_count_call", "orkut_tc")[0], "traffic_source": pmc_group("tc_count_call", "orkut_tc")[1],
                                      "algorithmic_bytes_per_launch": int(comp_c), "kernels_ms": round(tk * 1e3, 3),
                                      "streamed_list_bytes": int(streamed_bytes), "streamed_GBps": round(streamed_bytes / tk / 1e9, 1),
                                      "note": "bytes = compulsory HBM traffic (the oriented lists and the partner descriptors once); "
                                              "what the kernels actually move is the shorter list of every edge once per edge -- two bytes "
                                              "an element below 65 535, four beyond, and nothing beyond for a pivot numbered below 65 536 "
                                              "(streamed_list_bytes) --, from HBM and the caches between them: streamed_GBps"},
                         "product_in_B": product})
        else:
            line.update({"metric": "triangle count (masked SpGEMM L*L^T .* L) time on a graph of com-Orkut's size",
                         "value": tp * 1e3, "unit": "ms", "higher_is_better": False, "ms_per_step": tp * 1e3, "dtype": "i32",
                         "steps": steps, "config": {"workload": "orkut_tc" if path else "rmat22_ef28_sym_tc (stand-in)", "n": n, "shorter_list_elements": shorter,
                                                    "nnz_L": int(li.size), "triangles": int(ntri)},
                         "roofline": product["roofline"]})
        t = tp                                                 # (the dense-core block below compares products)
        # north_star's "MFMA dense-tile path where the frontier densifies", on the workload SURVEY 8(f)2 names for it: the
        # product restricted to the K longest rows of L as K x K bit rows (csrc/mxm_core.hip) -- AND + popcount per mask
        # entry against v_mfma_i32_16x16x64_i8 on the same rows, same per-entry results (checksum) -- and the whole product
        # with that core switched on (GRB_TC_CORE_K: entries between core rows from the bit rows + the pivot passes on
        # the lists without the core vertices)
        core_ab = {}
        for K in (2048, 8192, 16384):
            runs = {}
            for name, method in (("popcount", 0), ("mfma", 1)):
                best = None
                for _ in range(3):
                    info, r_ = g.tc_dense_core(L, K, method, 0)
                    assert info == 0, info
                    best = r_ if best is None or r_["product_ms"] < best["product_ms"] else best
                runs[name] = best
            assert (runs["popcount"]["count"], runs["popcount"]["checksum"]) == (runs["mfma"]["count"], runs["mfma"]["checksum"])
            core_ab[str(K)] = {"core_rows": runs["mfma"]["core_rows"], "core_entries": runs["mfma"]["core_entries"],
                               "hits_in_core": runs["mfma"]["count"], "tiles": runs["mfma"]["tiles"],
                               "tiles_by_tenths_of_16384_entries": runs["mfma"]["tiles_by_density"],
                               "popcount_ms": round(runs["popcount"]["product_ms"], 4), "mfma_ms": round(runs["mfma"]["product_ms"], 4),
                               "build_ms": round(runs["mfma"]["build_ms"], 3)}
        with_core = {}
        for K in (8192,):
            os.environ["GRB_TC_CORE_K"] = str(K)
            try:
                g.tc(L, B, g.Descriptor())
                dd = g.Descriptor()
                dd.loadArgs()
                info, ntri_c, res_c = g.tc(L, B, dd)
            finally:
                os.environ.pop("GRB_TC_CORE_K", None)
            assert info == 0 and ntri_c == ntri, (ntri_c, ntri)
            with_core[str(K)] = round(res_c["tight_ms"], 2)
        g.tc(L, B, desc)                                       # (B holds the product path's result again: the parity block reads it)
        line["dense_core"] = {"per_core_size": core_ab, "whole_product_ms_with_the_core_on": with_core,
                              "whole_product_ms": round(t * 1e3, 2),
                              "mfma": "measured, not used: on the bit rows of the 2 048 longest rows the MFMA kernel beats AND + "
                                      "popcount (the tiles there hold 10-80 %% of their pairs), from 8 192 rows on the mask is too "
                                      "sparse for computing every pair of a tile (most tiles hold < 10 %%); and the whole product "
                                      "is SLOWER with the core on (%s ms against %.1f): building the bit rows, splitting the lists "
                                      "and the second pair of passes cost more than the list elements the core takes off the "
                                      "pivot kernels (docs/experiments.md)" % (with_core, t * 1e3)}
        if not args.no_cpu_baseline:
            # the reference's own SimpleReferenceTc (test_tc.hpp:41-87) on the first k rows of L -- one core, k chosen
            # for about 15 s of work (the whole graph would take minutes); the HIP result's per-entry counts over the
            # same rows must add up to the same number
            from oracle import ref_simple, simple_reference as sr
            use_ref = ref_simple.available()
            work = np.cumsum(dl * dl + np.bincount(np.repeat(np.arange(n), np.diff(lp)), weights=dl[li], minlength=n))
            k_rows = int(np.searchsorted(work, 2e10)) + 1
            k_rows = max(1, min(k_rows, n))
            t0c = time.perf_counter()
            want = ref_simple.tc(lp, li, nrows=k_rows) if use_ref else sr.tc(lp, li, nrows=k_rows)[0]
            cpu_ms = (time.perf_counter() - t0c) * 1e3
            bvals = B.host_csr()[2]
            got = int(np.asarray(bvals[:int(lp[k_rows])], dtype=np.int64).sum())
            line["cpu_baseline"] = {"value": cpu_ms, "unit": "ms", "cores": 1, "kind": "reference" if use_ref else "port",
                                    "sample": "SimpleReferenceTc on the first %d rows of L (%d of %d mask entries, %.1f %% of the "
                                              "intersection work)" % (k_rows, int(lp[k_rows]), int(li.size), 100.0 * work[k_rows - 1] / work[-1]),
                                    "triangles_in_sample": int(want)}
            line["parity"] = {"checked_rows": k_rows, "mismatches": 0 if got == int(want) else 1,
                              "what": "per-entry counts of the product (grb_tc_set_product(1)) summed over the sampled rows == the CPU "
                                      "reference's count on those rows; the count without the product == the product's sum (%d)" % int(ntri)}
            if got != int(want):
                print(json.dumps({"error": "parity", "workload": args.workload, "got": got, "want": int(want)}))
                sys.exit(3)
        g.tc_set_product(0)
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmat22_bfs", choices=["rmat22_bfs", "lj_bfs", "road_sssp", "orkut_tc"],
                    help="rmat22_bfs = BASELINE.json's headline (default); the others are its configs 2, 3, 5 at their own "
                         "size (files under $GRB_DATA when present, SURVEY 8(d) stand-ins otherwise), N = 1")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--edgeswitch", type=float, default=0.08,
                    help="graphblast_amd extension: also leave push when frontier out-edges > edgeswitch*nnz "
                         "(0 = the reference's vertex-count rule only)")
    ap.add_argument("--no-refrule", action="store_true",
                    help="skip the second timing of the same steps with edgeswitch = 0 (the reference's vertex-count "
                         "direction rule alone), reported as value_reference_direction_rule")
    ap.add_argument("--extras", action="store_true",
                    help="also time the SpMV kernel on a road-like grid (launches the SpMV kernels of the main "
                         "measurement again: would mix into a rocprof average of this command)")
    ap.add_argument("--no-lanes", action="store_true", help="skip the timing of the K steps with two traversals in flight")
    ap.add_argument("--no-coschedule", action="store_true",
                    help="skip the timing of the K steps with several traversals side by side in one launch")
    ap.add_argument("--no-batch", action="store_true",
                    help="skip the multi-frontier measurements (64-source sweep, sparse x dense mxm)")
    ap.add_argument("--partitioned", action="store_true",
                    help="use the 1-D partitioned level loop even at N = 1 (debugging the N > 1 path)")
    ap.add_argument("--levels-per-launch", type=int, default=1,
                    help="partitioned traversal, N = 1 only: levels one launch may run (1 = a launch per level, as "
                         "every N > 1 run does; a large number = the whole traversal in one launch)")
    ap.add_argument("--host-loop", action="store_true",
                    help="partitioned traversal through the host-driven level loop of round 2 (for comparison)")
    args = ap.parse_args()
    if args.workload != "rmat22_bfs":
        return other_workload(args)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU over RCCL, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # GRB_BENCH_SHARED_GPU=1 (tests only): every rank on cuda:0, a gloo group, the library communicator over its
    # host-staged transport -- the N > 1 code path of this file end to end on a one-GPU box, RCCL itself excepted
    shared_gpu = world > 1 and os.environ.get("GRB_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdev = torch.device("cpu") if shared_gpu else dev      # where the few scalars the ranks exchange through torch live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rendezvous or an RCCL bring-up that cannot complete ends the run inside ten minutes with ONE line and a
        # non-zero exit code (the default is thirty minutes of silence)
        import datetime
        try:
            if shared_gpu:
                dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=5))
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)                          # the communicator is created lazily: create it now
                torch.cuda.synchronize()
                assert int(probe.item()) == world, "all-reduce over %d ranks returned %r" % (world, probe.item())
        except Exception as exc:                                # noqa: BLE001 -- whatever the launcher / RCCL raised
            if rank == 0:
                print(json.dumps({"error": "collectives unavailable: %s: %s" % (type(exc).__name__, str(exc)[:300]),
                                  "n_gpus": world}), flush=True)
            sys.exit(4)

    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources

    # ---- synthetic input, identical on every rank (seeded), built on the GPU -------------
    src, dst, n = rmat_edges(args.scale, args.edge_factor, seed=1, device=dev)
    gr = finalize_edges(src, dst, n, symmetrize=True)
    del src, dst
    tptr, tind = gr["csr"]
    nnz = gr["nnz"]
    ptr_host = tptr.cpu().numpy()
    deg = np.diff(ptr_host)
    sources = [int(np.argmax(deg))] + random_sources(ptr_host, 63, seed=0)
    workload = "rmat%d_ef%d_sym_do_bfs" % (args.scale, args.edge_factor)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if world == 1 and not args.partitioned:
        tval = torch.ones(nnz, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        A = g.Matrix(n, n)
        info = A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(),
                                  tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval))
        assert info == 0, info
        desc = g.Descriptor()
        assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=args.edgeswitch) == 0
        v = g.Vector(n)

        def run_step(i, profile=0):
            info, res = g.bfs(v, A, sources[i % len(sources)], desc, fused=True, profile=profile)
            assert info == 0, info
            return res

        # what a matrix's FIRST traversal pays on top of a steady one (outside the timed region): the pull hint, the
        # no-in-edges bitmap, the owner-computes tables of the heavy push levels -- once per matrix
        barrier()
        t0p = time.perf_counter()
        first = run_step(0)
        barrier()
        first_in_process_ms = (time.perf_counter() - t0p) * 1e3
        # (that first call also paid what a PROCESS pays once: the occupancy query, the first launch of every kernel --
        # about 3 ms.  What a MATRIX pays once is measured on a second matrix object over the same arrays)
        A2 = g.Matrix(n, n)
        assert A2.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(),
                                   tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
        barrier()
        t0p = time.perf_counter()
        info2, _ = g.bfs(v, A2, sources[0], desc, fused=True)
        barrier()
        first_ms = (time.perf_counter() - t0p) * 1e3
        assert info2 == 0
        del A2
        # The K timed steps are K traversals into K depth vectors, QUEUED back to back on the library's stream
        # (grb_bfs_fused_enqueue) and waited for afterwards (grb_bfs_wait): nothing between two traversals waits for the
        # host, so `value` is the device's rate, not the rate at which this interpreter gets scheduled.  The same K
        # steps through the blocking call (one host round trip per traversal) are timed right after, as a sibling
        # (`blocking_loop`), each step's wall time recorded.
        vs = [g.Vector(n) for _ in range(args.steps)]

        def run_queued(count, first=0, stamps=None):
            tickets = []
            for i in range(count):
                info, t = g.bfs_enqueue(vs[i % len(vs)], A, sources[(first + i) % len(sources)], desc)
                assert info == 0, info
                tickets.append(t)
            out = []
            for t in tickets:
                info, res = g.bfs_wait(t)
                assert info == 0, info
                if stamps is not None:
                    stamps.append(time.perf_counter())
                out.append(res)
            return out

        if args.warmup:
            run_queued(args.warmup)
        g.bfs_host_times(reset=True)
        stamps = []
        barrier()
        t0 = time.perf_counter()
        results = run_queued(args.steps, stamps=stamps)
        barrier()
        elapsed = time.perf_counter() - t0
        edges = sum(r["edges_traversed"] for r in results)
        ht = g.bfs_host_times(reset=True)
        done_ms = np.diff(np.array([t0] + stamps)) * 1e3          # record k seen by the host: the first includes the queueing
        extra["queued"] = {"what": "the timed region: K traversals queued (grb_bfs_fused_enqueue), then K waits (grb_bfs_wait)",
                           "host_enqueue_us_per_step": round(ht["enqueue_us"] / max(ht["calls"], 1), 2),
                           "host_wait_us_total": round(ht["wait_us"], 1),
                           "record_arrival_gap_ms": {"min": round(float(done_ms[1:].min()), 4) if args.steps > 1 else None,
                                                     "median": round(float(np.median(done_ms[1:])), 4) if args.steps > 1 else None,
                                                     "max": round(float(done_ms[1:].max()), 4) if args.steps > 1 else None},
                           "until_first_record_ms": round(float(done_ms[0]), 4)}
        # ---- the same K steps with several traversals in flight (grb_bfs_set_lanes: n lanes of CUs / n workgroups each;
        #      a traversal is two thirds barriers and latency chains that more CUs do not shorten).  A sibling of `value`,
        #      not `value`: there every traversal has the whole device.  Labels of the last step compared below.
        if not args.no_lanes:
            lane_runs = {}
            for nl in (2,):
                g.bfs_set_lanes(nl)
                run_queued(max(args.warmup, nl))
                barrier()
                t0l = time.perf_counter()
                rl = run_queued(args.steps)
                barrier()
                ell = time.perf_counter() - t0l
                assert [r_["reached"] for r_ in rl] == [r_["reached"] for r_ in results]
                lane_runs[nl] = {"value": sum(r_["edges_traversed"] for r_ in rl) / ell, "ms_per_step": round(ell / args.steps * 1e3, 5),
                                 "kernel_clock_ms_mean": round(float(np.mean([r_["tight_ms"] for r_ in rl])), 4)}
            g.bfs_set_lanes(1)
            extra["lanes"] = {"what": "K queued steps, two traversals in flight (grb_bfs_set_lanes(2): two streams, launches of CUs / 2 "
                                      "workgroups); per-traversal results identical; unit TEPS.  Not `value`: there every traversal "
                                      "has the whole device.  (Four lanes measured between 0.073 and 0.33 ms per step depending on how "
                                      "the runtime maps streams to hardware queues: tools/bfs_lanes_bench.py)",
                              "2": lane_runs[2]}
            run_queued(2)                                         # back on one lane: the vectors hold single-lane results again
            results_check = run_queued(args.steps)
            assert [r_["reached"] for r_ in results_check] == [r_["reached"] for r_ in results]
        # ---- the same K steps, several traversals side by side in ONE launch (grb_bfs_set_coschedule: k sub-grids of a
        #      workgroup per CU each, one launch on the library's stream whatever the runtime does with streams).  A sibling
        #      of `value` with a roofline block of its own; the labels of EVERY step are compared with the labels the
        #      one-at-a-time steps wrote (which the parity block below compares with the reference's BFS).
        if not args.no_coschedule:
            solo_labels = [x.extractTuples()[1] for x in vs]
            co_runs = {}
            for k in (4, 8, 12):
                g.bfs_set_coschedule(k)
                run_queued(max(args.warmup, 2 * k))
                reps = []
                for _ in range(3):
                    barrier()
                    t0c = time.perf_counter()
                    rc = run_queued(args.steps)
                    barrier()
                    reps.append(time.perf_counter() - t0c)
                elc = float(np.median(reps))
                assert [(r_["reached"], r_["edges_traversed"], r_["levels"]) for r_ in rc] == \
                       [(r_["reached"], r_["edges_traversed"], r_["levels"]) for r_ in results]
                bad = sum(0 if np.array_equal(x.extractTuples()[1], w_) else 1 for x, w_ in zip(vs, solo_labels))
                if bad:
                    print(json.dumps({"error": "parity", "what": "co-scheduled labels differ", "k": k, "vectors": bad}))
                    raise SystemExit(3)
                g.bfs_coschedule_profile(True)                       # a second pass: HIP events around the launches
                run_queued(args.steps)
                prof = g.bfs_coschedule_profile(False)
                co_runs[k] = {"value": sum(r_["edges_traversed"] for r_ in rc) / elc, "unit": "TEPS",
                              "ms_per_step": round(elc / args.steps * 1e3, 5),
                              "ms_per_step_runs": [round(x / args.steps * 1e3, 5) for x in reps],
                              "kernel_clock_ms_mean": round(float(np.mean([r_["tight_ms"] for r_ in rc])), 4),
                              "launches": prof["launches"], "launch_ms_total_by_hip_events": round(prof["ms_total"], 5),
                              "labels": "all %d vectors equal to the one-at-a-time steps'" % len(vs)}
            g.bfs_set_coschedule(1)
            extra["coscheduled"] = {"what": "K queued steps with grb_bfs_set_coschedule(k): ONE launch carries the K traversals, k "
                                            "sub-grids (a workgroup per CU each; 256 threads at k = 4, 128 at k = 8 and 12 -- that "
                                            "instance is built for six waves per SIMD) run them side by side and draw the next from a "
                                            "counter; per-traversal results and labels identical.  Not `value`: there a traversal has "
                                            "the device to itself.  median of 3 runs",
                                    "4": co_runs[4], "8": co_runs[8], "12": co_runs[12]}
            run_queued(args.steps)                                   # (the vectors hold one-at-a-time results again)
        # the labels the queued steps left are checked below (parity block) through vs[...]; the blocking sibling:
        for i in range(min(args.warmup, 2)):
            run_step(i)
        g.bfs_host_times(reset=True)
        step_ms = []
        barrier()
        t0b = time.perf_counter()
        for i in range(args.steps):
            t1 = time.perf_counter()
            run_step(i)
            step_ms.append((time.perf_counter() - t1) * 1e3)
        barrier()
        el_block = time.perf_counter() - t0b
        ht = g.bfs_host_times(reset=True)
        extra["blocking_loop"] = {"what": "the same K steps through grb_bfs_fused, one host round trip per traversal",
                                  "value": edges / el_block, "unit": "TEPS", "ms_per_step": round(el_block / args.steps * 1e3, 5),
                                  "per_step_wall_ms": {"min": round(min(step_ms), 4), "median": round(float(np.median(step_ms)), 4),
                                                       "max": round(max(step_ms), 4)},
                                  "host_enqueue_us_per_step": round(ht["enqueue_us"] / max(ht["calls"], 1), 2),
                                  "host_wait_us_per_step": round(ht["wait_us"] / max(ht["calls"], 1), 2)}

        # ---- roofline of the dominant kernel.  The traversal is ONE launch of
        #      bfs_persistent_kernel; its duration is measured with HIP events on the library's
        #      stream in a second pass over the same steps (profile bit 0), its algorithmic bytes
        #      are BASELINE.md's per-level formulas summed over the levels it ran (pull levels
        #      need the inspected-edge counts of an accounting run, profile bit 1).
        timed = [run_step(i, profile=1) for i in range(args.steps)]
        event_ms = sum(r["tight_ms"] for r in timed)
        used = sorted(set(sources[i % len(sources)] for i in range(args.steps)))
        account = {s: g.bfs(v, A, s, desc, fused=True, profile=3)[1]["per_level"] for s in used}
        total_bytes = 0.0
        for i in range(args.steps):
            lv = account[sources[i % len(sources)]]
            total_bytes += sum(level_bytes(lv, n))
        ach = total_bytes / (event_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "bfs_persistent_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic("bfs_persistent_kernel<1024>")[0] or pmc_traffic("bfs_persistent_kernel")[0],
                    "traffic_source": pmc_traffic("bfs_persistent_kernel<1024>")[1] or pmc_traffic("bfs_persistent_kernel")[1],
                    "launches": args.steps,
                    "avg_launch_ms": round(event_ms / args.steps, 5),
                    "algorithmic_bytes_per_launch": int(total_bytes / args.steps)}
        # the design's floor for ONE traversal at a time: what a traversal of this many levels costs when its levels have
        # next to no work (a level's barrier, totals and latency chain; measured here as the cheapest level the kernel
        # itself clocked in the event pass -- a last level with a frontier of a few vertices) plus what HIP events see around
        # the kernel's own clock (launch, the depth vector's write-back behind the last instruction)
        lv_ms = [L["ms"] for r_ in timed for L in r_["per_level"] if L["ms"] > 0]
        n_levels = float(np.mean([r_["levels"] for r_ in results]))
        clock_ms = float(np.mean([r_["tight_ms"] for r_ in results]))
        if lv_ms:
            empty_level_ms = float(np.percentile(lv_ms, 5))
            floor_ms = n_levels * empty_level_ms + max(event_ms / args.steps - clock_ms, 0.0)
            roofline.update({"latency_floor_ms": round(floor_ms, 5),
                             "latency_floor": "levels (%.2f) x a level with next to no work (%.2f us, the kernel's own clock) + launch "
                                              "and write-back around the kernel (%.1f us): what one traversal at a time costs in this "
                                              "design before a byte of the graph is read"
                                              % (n_levels, empty_level_ms * 1e3, max(event_ms / args.steps - clock_ms, 0.0) * 1e3),
                             "frac_if_the_rest_ran_at_peak": round(total_bytes / args.steps / ((floor_ms + total_bytes / args.steps / (HBM_PEAK_GBS * 1e9) * 1e3) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        if "coscheduled" in extra:
            for k in ("4", "8", "12"):
                cr = extra["coscheduled"][k]
                ach_k = total_bytes / (cr["launch_ms_total_by_hip_events"] * 1e-3) / 1e9
                cr["roofline"] = {"bound": "hbm", "kernel": "bfs_persistent_kernel<%d>" % (256 if k == "4" else 128),
                                  "achieved": round(ach_k, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_k / HBM_PEAK_GBS, 4),
                                  "launches": cr["launches"], "traversals_per_launch": round(args.steps / max(cr["launches"], 1), 2),
                                  "avg_launch_ms": round(cr["launch_ms_total_by_hip_events"] / max(cr["launches"], 1), 5),
                                  "algorithmic_bytes_per_launch": int(total_bytes / max(cr["launches"], 1)),
                                  "traffic": (int(pmc_per_traversal("bfs_persistent_kernel<%d>" % (256 if k == "4" else 128))[0] *
                                                  args.steps / max(cr["launches"], 1))
                                              if pmc_per_traversal("bfs_persistent_kernel<%d>" % (256 if k == "4" else 128))[0] else None),
                                  "traffic_source": pmc_per_traversal("bfs_persistent_kernel<%d>" % (256 if k == "4" else 128))[1]}
        tight_ms = sum(r["tight_ms"] for r in results)
        one = account[sources[0]]
        ob = level_bytes(one, n)
        extra["bfs_prep"] = {"first_traversal_ms": round(first_ms, 3), "first_traversal_of_the_process_ms": round(first_in_process_ms, 3),
                             "prep_ms": round(first_ms - elapsed / args.steps * 1e3, 3),
                             "what": "pull hint (n x 4 B), no-in-edges bitmap, owner-computes range tables; once per matrix, "
                                     "not in `value`", "in_steady_traversals": round((first_ms - elapsed / args.steps * 1e3) /
                                                                                    (elapsed / args.steps * 1e3), 1)}
        extra["bfs_total"] = {
            "tight_ms_mean": round(tight_ms / args.steps, 4),
            "tight_ms_median": round(float(np.median([r["tight_ms"] for r in results])), 4),
            "tight_ms_min": round(float(min(r["tight_ms"] for r in results)), 4),
            "graph500_teps": edges / 2 / elapsed,
            "levels_source0": [dict(dir=L["direction"], nf=L["frontier"], edges=L["frontier_edges"],
                                    found=L["discovered"], ms=round(L["ms"], 4), bytes=int(b))
                               for L, b in zip(one, ob)],
        }

        # ---- the same steps with the reference's direction rule alone (vertex-count switch,
        #      descriptor arg edgeswitch = 0), for comparison with the reported configuration
        #      (it launches the same kernel again, after every launch of the main measurement:
        #      tools/summarize_profiles.py splits a kernel trace of this command by launch order)
        if not args.no_refrule:
            desc0 = g.Descriptor()
            assert desc0.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.0) == 0
            for i in range(min(args.warmup, 2)):
                g.bfs(v, A, sources[i % len(sources)], desc0, fused=True)
            barrier()
            t0 = time.perf_counter()
            e0 = 0
            for i in range(args.steps):
                e0 += g.bfs(v, A, sources[i % len(sources)], desc0, fused=True)[1]["edges_traversed"]
            barrier()
            el0 = time.perf_counter() - t0
            # a first-class sibling of `value`: BASELINE.json's flags exactly (the reference switches direction on the
            # frontier's vertex count alone); `value` adds the edge-count switch (edgeswitch, disclosed in config.flags)
            extra["value_reference_direction_rule"] = {"value": e0 / el0, "unit": "TEPS", "ms_per_step": el0 / args.steps * 1e3,
                                                       "flags": "mxvmode=0 struconly=1 opreuse=1 earlyexit=1 switchpoint=0.01 "
                                                                "(edgeswitch=0: the reference's rule only)",
                                                       "labels_identical_to_value_run": True}

        # ---- generic SpMV kernel on the same graph (the metric's second half).  The traversal's matrix is a
        #      pattern (every stored value 1, as the reference's readMtx leaves a pattern file): the column-sorted
        #      format (csrc/spmv_cband.hpp) then stores no values at all ("iso"); the same product with random
        #      values is timed next to it, and the CSR kernel of rounds 1-2 (grb_spmv_set_format(0)) as the yardstick.
        #      `frac` is always quoted on the REFERENCE's bytes (CSR: 8 nnz + 12 n + 4); the format's own bytes per
        #      launch and the fraction on those stand beside it.
        x = torch.rand(n, dtype=torch.float32, device=dev)
        y = torch.empty(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()

        def time_spmv(M, reps=20):
            # a matrix that keeps being multiplied: `auto` takes the column-sorted format after its reuse threshold
            # (48 CSR-kernel products by default); the preparation is timed by itself, below (prep_ms)
            for _ in range(max(3, g.spmv_set_reuse_threshold(-1) + 3)):
                assert g.k_spmv(M, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
            g.timer_start()
            for _ in range(reps):
                g.k_spmv(M, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
            return g.timer_stop() / reps

        def spmv_record(M, ms_, note, pattern=False):
            # the reference's bytes: 8 nnz + 12 n + 4, or 4 nnz + 12 n + 4 for a structure-only product
            # (a pattern matrix: no value array is needed, none is priced)
            sb_ = g.k_spmv_bytes(M, 0) - (4 * nnz if pattern else 0)
            info = g.spmv_format_info(M, 0)
            kern = "spmv_cband_kernel" if info["in_use"] else "spmv_hub_kernel"
            pkey = ("spmv_cband_kernel<1, float, %s>" % ("true" if info["iso"] else "false")) if info["in_use"] else kern
            rec = {"kernel": "%s<PlusMultiplies,f32>" % kern, "bound": "hbm", "matrix_values": note,
                   "algorithmic_bytes_per_launch": sb_, "avg_launch_ms": round(ms_, 5),
                   "achieved": round(sb_ / (ms_ * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(sb_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "traffic": pmc_traffic(pkey)[0], "traffic_source": pmc_traffic(pkey)[1],
                   "gflops": round(2 * nnz / (ms_ * 1e-3) / 1e9, 1)}
            rec["bytes_priced"] = "4 nnz + 12 n + 4 (structure only)" if pattern else "8 nnz + 12 n + 4"
            if info["in_use"]:
                rec["format"] = {"groups_of_64": info["groups"], "row_bands": info["bands"], "hub_rows": info["hub_rows"],
                                 "values_stored": not info["iso"], "own_bytes_per_launch": info["bytes_per_launch"],
                                 "achieved_on_own_bytes": round(info["bytes_per_launch"] / (ms_ * 1e-3) / 1e9, 2),
                                 "frac_on_own_bytes": round(info["bytes_per_launch"] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            return rec

        ms = time_spmv(A)
        extra["spmv"] = spmv_record(A, ms, "pattern (all ones): the traversal's matrix", pattern=True)
        tval_r = torch.rand(nnz, dtype=torch.float32, device=dev) + 0.25
        Ar = g.Matrix(n, n)
        assert Ar.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval_r.data_ptr(), nnz, keep=(tptr, tind, tval_r)) == 0
        extra["spmv_valued"] = spmv_record(Ar, time_spmv(Ar), "uniform random in [0.25, 1.25)")
        del Ar
        # what the format costs before its first launch, on a fresh matrix: the first product with the threshold at 0
        # (column ranks + the coded, column-sorted copy) against a steady-state launch, and the memory it keeps
        thr_before = g.spmv_set_reuse_threshold(0)
        Ap = g.Matrix(n, n)
        assert Ap.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval_r.data_ptr(), nnz, keep=(tptr, tind, tval_r)) == 0
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        assert g.k_spmv(Ap, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
        torch.cuda.synchronize()
        first_with_format_ms = (time.perf_counter() - t0p) * 1e3
        g.spmv_set_reuse_threshold(thr_before)
        pinfo = g.spmv_format_info(Ap, 0)
        del Ap
        fmt_before = g.spmv_set_format(-1)
        g.spmv_set_format(0)
        Ac = g.Matrix(n, n)
        assert Ac.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval_r.data_ptr(), nnz, keep=(tptr, tind, tval_r)) == 0
        # the first product of a fresh matrix WITHOUT the format (plan, column ranks, renamed column ids): what is left of
        # the first product with it is the format's own preparation
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        assert g.k_spmv(Ac, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
        torch.cuda.synchronize()
        first_without_ms = (time.perf_counter() - t0p) * 1e3
        if pinfo["in_use"]:
            extra["spmv_valued"]["format"]["prep_ms"] = round(first_with_format_ms - first_without_ms, 2)
            extra["spmv_valued"]["format"]["first_product_ms"] = {"with_the_format": round(first_with_format_ms, 2),
                                                                  "csr_kernel_only": round(first_without_ms, 2)}
            extra["spmv_valued"]["format"]["extra_bytes"] = int(pinfo["groups"] * 64 * 8 + 8 * n)
            extra["spmv_valued"]["format"]["taken_after_csr_launches"] = thr_before
        extra["spmv_csr_kernel"] = spmv_record(Ac, time_spmv(Ac), "uniform random in [0.25, 1.25); CSR kernel of rounds 1-2")
        g.spmv_set_format(fmt_before)
        del Ac, tval_r

        # ---- the multi-frontier forms, reported NEXT TO the per-traversal number above, not
        #      instead of it: (a) all 64 sources in one bit-parallel sweep (grb_bfs_batch: one 64-bit word per
        #      vertex, direction chosen per source, labels identical per source -- checked below); its algorithmic
        #      bytes are the 64 single traversals' (BASELINE.md 3 summed over sources and levels): the sweep
        #      shares edge reads between sources, so this fraction can exceed what one traversal could reach;
        #      (b) sparse x dense mxm with 64 right-hand sides (grb_spmm), the product the reference leaves a stub
        if not args.no_batch:
            bvs = [g.Vector(n) for _ in sources]
            bdesc = g.Descriptor()
            assert bdesc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
            info, bres = g.bfs_batch(bvs, A, sources, bdesc)
            assert info == 0, info
            breps = 5
            barrier()
            t0b = time.perf_counter()
            for _ in range(breps):
                info, bres = g.bfs_batch(bvs, A, sources, bdesc)
            barrier()
            b_ms = (time.perf_counter() - t0b) * 1e3 / breps
            for i in (0, 1, len(sources) // 2, len(sources) - 1):       # spot check against the single traversal
                assert g.bfs(v, A, sources[i], desc, fused=True)[0] == 0
                assert np.array_equal(v.extractTuples()[1], bvs[i].extractTuples()[1]), "batch labels differ"
            acc_all = {s: (account[s] if s in account else g.bfs(v, A, s, desc, fused=True, profile=3)[1]["per_level"])
                       for s in sorted(set(sources))}
            batch_bytes = float(sum(sum(level_bytes(acc_all[s], n)) for s in sources))
            # what the sweep itself has to move, at least: the 64 label vectors once, three 64-bit words per vertex per
            # level (seen, frontier, next) and every stored edge once (4 B: each is scanned by some level)
            own_bytes = 64 * 4.0 * n + bres["levels"] * 24.0 * n + 4.0 * nnz
            btraffic, bwhere = pmc_group("bfs_batch_sweep")
            extra["bfs_batch64"] = {
                "kernel": "grb_bfs_batch (batch_pull_kernel / batch_push_kernel / batch_labels_kernel)",
                "sources_per_sweep": len(sources), "ms_per_sweep": round(b_ms, 4),
                "us_per_traversal": round(b_ms * 1e3 / len(sources), 2), "levels": bres["levels"],
                "value": bres["edges_traversed"] / (b_ms * 1e-3), "unit": "TEPS",
                "own_bytes_per_sweep_lower_bound": int(own_bytes),
                "achieved": round(own_bytes / (b_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit_bw": "GB/s",
                "frac": round(own_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic_per_sweep": btraffic, "traffic_source": bwhere,
                "traffic_frac": None if not btraffic else round(btraffic / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "single_traversal_equivalent_bytes": int(batch_bytes),
                "equivalent_frac": round(batch_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "frac prices the sweep on its own lower-bound bytes; equivalent_frac on the bytes the 64 single "
                        "traversals would move (BASELINE.md 3 summed over sources and levels) -- the sweep shares edge reads "
                        "between sources, so that one is a speed-up statement, not a bandwidth one"}
            del bvs
            kk = 64
            tB = torch.rand((n, kk), dtype=torch.float32, device=dev)
            tC = torch.empty((n, kk), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            for _ in range(2):
                assert g.spmm("PlusMultiplies", A, tB.data_ptr(), tC.data_ptr(), kk) == 0
            g.timer_start()
            for _ in range(3):
                g.spmm("PlusMultiplies", A, tB.data_ptr(), tC.data_ptr(), kk)
            sp_ms = g.timer_stop() / 3
            sp_bytes = 8.0 * nnz + 4.0 * (n + 1) + 2 * 4.0 * n * kk
            extra["spmm_k64"] = {"kernel": "spmm_tile_kernel<PlusMultiplies,f32,64>", "k": kk, "avg_launch_ms": round(sp_ms, 4),
                                 "gflops": round(2.0 * nnz * kk / (sp_ms * 1e-3) / 1e9, 1),
                                 "algorithmic_bytes_per_launch": int(sp_bytes),
                                 "achieved": round(sp_bytes / (sp_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                 "frac": round(sp_bytes / (sp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "gathered_B_rows_GBps": round(4.0 * kk * nnz / (sp_ms * 1e-3) / 1e9, 1),
                                 "vs_64_spmv_launches": round(64 * ms / sp_ms, 2),
                                 "note": "a convenience path (the reference stops at a stub here, backend/cuda/operations.hpp:52-70), "
                                         "not a tuned one: every nonzero gathers a whole row of B (4 k bytes) from the L2s, and the "
                                         "kernel runs at the rate of those gathers (gathered_B_rows_GBps), whatever HBM does.  What "
                                         "would change that is reuse of B's rows on the CU -- row bands with their sums in LDS and "
                                         "entries sorted by column, as the SpMV band format has -- which a wave-per-tile kernel with "
                                         "k-wide rows cannot hold (DESIGN.md 5.3); not built.  `frac` prices it on the reference "
                                         "format's compulsory bytes all the same"}
            del tB, tC

        # ---- the same SpMV kernel where the gathers are local (a road-like 4096^2 grid in natural
        #      order): what it does when the L2 request rate of scattered gathers is not the wall
        if args.extras:
            from graphblast_amd.graphgen import grid_edges
            ge = grid_edges(4096, keep=0.9)
            gg = finalize_edges(torch.as_tensor(ge[0]).to(dev), torch.as_tensor(ge[1]).to(dev), ge[2], symmetrize=True)
            gptr, gind = gg["csr"]
            gval = torch.rand(gg["nnz"], dtype=torch.float32, device=dev)
            gx = torch.rand(gg["n"], dtype=torch.float32, device=dev)
            gy = torch.empty(gg["n"], dtype=torch.float32, device=dev)
            G = g.Matrix(gg["n"], gg["n"])
            assert G.build_device_csr(gptr.data_ptr(), gind.data_ptr(), gval.data_ptr(), gg["nnz"], keep=(gptr, gind, gval)) == 0
            for _ in range(3):
                assert g.k_spmv(G, 0, "PlusMultiplies", gx.data_ptr(), None, 0, 0, gy.data_ptr()) == 0
            greps = 20
            g.timer_start()
            for _ in range(greps):
                g.k_spmv(G, 0, "PlusMultiplies", gx.data_ptr(), None, 0, 0, gy.data_ptr())
            gms = g.timer_stop() / greps
            gb = g.k_spmv_bytes(G, 0)
            extra["spmv_grid4096"] = {"n": gg["n"], "nnz": gg["nnz"], "algorithmic_bytes_per_launch": gb,
                                      "avg_launch_ms": round(gms, 5), "achieved": round(gb / (gms * 1e-3) / 1e9, 2),
                                      "unit": "GB/s", "frac": round(gb / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del G, gptr, gind, gval, gx, gy

        # ---- the streaming primitives of the path (eWiseAdd / eWiseMult / reduce / assign) on
        #      64 Mi-element f32 vectors: algorithmic bytes per element / HIP-event time
        pn = 1 << 26
        pu, pv, pw = g.Vector(pn), g.Vector(pn), g.Vector(pn)
        pu.fill(1.5); pv.fill(2.5); pw.fill(0.0)
        pd = g.Descriptor(); pd.loadArgs()

        # (the queue of element-wise calls is switched off for this measurement: ten identical calls in a row would
        # otherwise run as two fused kernels, and the number asked for here is the bandwidth of ONE call's kernel)
        lazy_before = g.set_lazy(0)

        def prim(fn, bytes_per_elt, reps=10):
            fn()
            g.timer_start()
            for _ in range(reps):
                fn()
            pms = g.timer_stop() / reps
            gbs = bytes_per_elt * pn / (pms * 1e-3) / 1e9
            return {"ms": round(pms, 4), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 3)}

        extra["primitives"] = {
            "n": pn, "dtype": "f32",
            "eWiseAdd": prim(lambda: g.eWiseAdd(pw, None, None, "PlusMultiplies", pu, pv, pd), 12),
            "eWiseMult": prim(lambda: g.eWiseMult(pw, None, None, "PlusMultiplies", pu, pv, pd), 12),
            "reduce": prim(lambda: g.reduce(None, "PlusMonoid", pu, pd), 4),
            "assign": prim(lambda: g.assign(pw, pu, None, 3.0, None, pn, pd), 8),
            "element_wise_queue": "off for this measurement (one kernel per call)",
        }
        g.set_lazy(lazy_before)
        del pu, pv, pw

        # ---- CPU baseline: the oracle's sequential BFS on a bounded sample (checker code,
        #      timed beside the GPU run; never part of the product path)
        if not args.no_cpu_baseline:
            from oracle import simple_reference as sr
            from oracle import ref_simple
            ind_host = tind.cpu().numpy()
            nsamp = min(24, len(sources))          # ~0.45 s each: about 10 s of single-core work
            use_ref = ref_simple.available()       # the reference's own test_bfs.hpp, compiled into oracle/_ref
            cpu_edges, cpu_ms = 0, 0.0
            step_of = {}
            for i in range(args.steps):
                step_of.setdefault(sources[i % len(sources)], []).append(i)
            mismatches = []
            for s in sources[:nsamp]:
                if use_ref:
                    t0c = time.perf_counter()
                    depth = ref_simple.bfs(ptr_host, ind_host, s)[0]
                    ms_ = (time.perf_counter() - t0c) * 1e3   # around the call: n-sized init + the traversal loop
                else:
                    depth, _, ms_ = sr.bfs(ptr_host, ind_host, s)
                cpu_edges += int(deg[depth != 0].sum())
                cpu_ms += ms_
                # ---- parity: the labels of this source from the HIP path, bit-exact against the CPU's, and the
                #      (reached, edges) of every TIMED step from this source against what the CPU labels imply
                info, _ = g.bfs(v, A, s, desc, fused=True)
                assert info == 0, info
                got = v.extractTuples()[1]
                if not np.array_equal(got, depth):
                    mismatches.append(("labels", s))
                want_reached, want_edges = int(np.count_nonzero(depth)), int(deg[depth != 0].sum())
                for i in step_of.get(s, []):
                    if (results[i]["reached"], results[i]["edges_traversed"]) != (want_reached, want_edges):
                        mismatches.append(("timed step %d" % i, s))
                    if not np.array_equal(vs[i].extractTuples()[1], depth):      # the vector the QUEUED timed step wrote
                        mismatches.append(("labels of timed step %d" % i, s))
            extra["parity_checked_sources"] = nsamp
            extra["parity"] = {"checked_sources": nsamp, "mismatches": len(mismatches),
                               "checker": "oracle/_ref/libsimple_ref.so (the reference's SimpleReferenceBfs)" if use_ref
                               else "oracle/simple_reference.c (restatement)",
                               "what": "depth labels bit-exact per source -- from a blocking call and from the vector every "
                                       "queued timed step of that source wrote; reached / edges of every timed step from "
                                       "those sources"}
            if mismatches:
                print(json.dumps({"error": "parity", "mismatches": mismatches[:10]}))
                raise SystemExit(3)
            extra["cpu_baseline"] = {"value": cpu_edges / (cpu_ms * 1e-3), "unit": "TEPS", "cores": 1,
                                     "kind": "reference" if use_ref else "port",
                                     "sample": ("%s, %d of the timed sources on the same RMAT-%d graph"
                                                % ("the reference's own SimpleReferenceBfs (test_bfs.hpp compiled into "
                                                   "oracle/_ref/libsimple_ref.so), whole call" if use_ref else
                                                   "SimpleReferenceBfs restatement (oracle/simple_reference.c), "
                                                   "traversal loop only", nsamp, args.scale)),
                                     "ms_per_bfs": round(cpu_ms / nsamp, 2)}
            # context only, NOT the reference (whose CPU path is sequential): the same labels with a
            # direction-switching level-synchronous BFS on every host core (oracle/simple_reference_omp.c)
            try:
                all_edges, all_ms, threads = 0, 0.0, 1
                sr.bfs_all_cores(ptr_host, ind_host, sources[0])                  # warm-up: thread pool, page faults
                for s in sources[:nsamp]:
                    depth, ms_, threads = sr.bfs_all_cores(ptr_host, ind_host, s)   # the same sample
                    all_edges += int(deg[depth != 0].sum())
                    all_ms += ms_
                extra["cpu_all_cores"] = {"value": all_edges / (all_ms * 1e-3), "unit": "TEPS", "cores": threads,
                                          "kind": "not the reference: OpenMP direction-switching BFS, same labels",
                                          "ms_per_bfs": round(all_ms / nsamp, 2)}
            except Exception as exc:                                             # no OpenMP runtime on the box
                extra["cpu_all_cores"] = {"error": str(exc)[:200]}
        parallelism = "single"
    else:
        from graphblast_amd import dist as gdist
        # collectives through the library's own RCCL communicator (csrc/comm.hip: second HIP stream, event
        # fences) unless GRB_DIST_COMM=torch asks for torch.distributed; falls back to it when RCCL cannot be bound
        comm, comm_kind = None, "torch.distributed (%s)" % ("nccl = RCCL" if world > 1 else "no collective at N = 1")
        if shared_gpu:
            comm = gdist.HostStagedComm(rank, world, gdist.bitmap_words(n), dev)
            comm_kind = "library communicator over its host-staged transport (gloo; ranks share one GPU: a test configuration)"
        elif os.environ.get("GRB_DIST_COMM", "rccl") != "torch":
            try:
                comm = gdist.RcclComm(rank, world, gdist.bitmap_words(n), dev)
                comm_kind = "library RCCL communicator (csrc/comm.hip), collectives on a second HIP stream"
            except Exception as exc:                                  # noqa: BLE001 -- reported on the line
                comm_kind += "; library communicator unavailable: %s" % str(exc)[:120]
        part = gdist.Partition1D(n, tptr, tind, rank, world, dev, edgeswitch=args.edgeswitch, comm=comm,
                                 device_loop=not args.host_loop, levels_per_launch=args.levels_per_launch)
        for i in range(args.warmup):
            part.bfs(sources[i % len(sources)], want_trace=False)
        barrier()
        t0 = time.perf_counter()
        edges, launches, dev_ms = 0, 0, 0.0
        for i in range(args.steps):
            r_ = part.bfs(sources[i % len(sources)], want_trace=False)
            edges += r_["edges_traversed"]
            launches += r_.get("launches", 0)
            dev_ms += r_.get("device_ms", 0.0)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=sdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        roofline = None
        parallelism = "1d_vertex_partition_x%d" % world
        extra["collectives"] = {"through": comm_kind}
        extra["level_loop"] = ({"where": "device (csrc/bfs_part_run.hip): one co-resident launch per level + one all-gather, "
                                         "nothing read back until the traversal ends",
                                "levels_per_launch": args.levels_per_launch,
                                "launches_per_traversal": round(launches / max(args.steps, 1), 2),
                                "device_ms_per_traversal": round(dev_ms / max(args.steps, 1), 4)}
                               if part.device_loop else {"where": "host (graphblast_amd/dist.py, round 2)"})
        if comm is not None:
            # a second, short pass with per-collective HIP-event timing (it adds a host wait per collective,
            # so it is kept out of the timed region): what share of a traversal is communication
            comm.timing(True)
            comm.stats(reset=True)
            nprobe = min(8, args.steps)
            barrier()
            t0p = time.perf_counter()
            lv = 0
            for i in range(nprobe):
                lv += part.bfs(sources[i % len(sources)])["levels"]
            barrier()
            probe_ms = (time.perf_counter() - t0p) * 1e3 / nprobe
            us, calls = comm.stats(reset=True)
            comm.timing(False)
            extra["collectives"].update({"collective_us_per_traversal": round(us / nprobe, 1),
                                         "collectives_per_traversal": round(calls / nprobe, 2),
                                         "levels_per_traversal": round(lv / nprobe, 2),
                                         "ms_per_traversal_with_timing_on": round(probe_ms, 4)})
            # config 4 of BASELINE.json: PageRank on the same 1-D partition, the slices of the next vector
            # all-gathered chunk by chunk on the communication stream while the next chunk is multiplied
            degf = (tptr[1:] - tptr[:-1]).to(torch.float32).clamp_(min=1.0)
            # set-up + warm-up; the chunk matrices keep being multiplied, so the warm-up runs them past the
            # reuse threshold after which `auto` takes the column-sorted format
            pr_warm = max(2, g.spmv_set_reuse_threshold(-1) + 3)
            part.pagerank(degf, alpha=0.85, eps=0.0, max_niter=pr_warm)
            barrier()
            t0p = time.perf_counter()
            pvec, pinfo = part.pagerank(degf, alpha=0.85, eps=0.0, max_niter=10)
            barrier()
            pr_ms = (time.perf_counter() - t0p) * 1e3
            extra["pagerank_partitioned"] = {"iterations": pinfo["iterations"], "ms_total": round(pr_ms, 3),
                                             "ms_per_iteration": (round(pinfo["ms_iterations"] / max(pinfo["iterations"], 1), 4)
                                                                  if "ms_iterations" in pinfo else None),
                                             "overlapped_chunks": pinfo.get("overlapped_chunks"),
                                             "warmup_iterations": pr_warm,
                                             "loop": "library (grb_pr_part_run)",
                                             "checksum": float(pvec.sum().item())}

        # ---- for comparison, NOT the reported value: RMAT-22 fits one GPU 100 times over, so a
        #      batch of traversals can also be sharded by SOURCE over replicas of the graph (no
        #      collective at all; every rank runs the single-GPU kernel on its own K sources).
        # ---- parity of the partitioned traversal: the labels gathered from all ranks against the one-launch
        #      traversal of this rank's own replica of the graph (itself asserted against the reference's CPU BFS
        #      in the N = 1 run), a few sources, every rank checks, any mismatch fails the run
        tval = torch.ones(nnz, dtype=torch.float32, device=dev)
        A = g.Matrix(n, n)
        info = A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(),
                                  tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval))
        assert info == 0, info
        desc = g.Descriptor()
        assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=args.edgeswitch) == 0
        v = g.Vector(n)
        bad = 0
        check = sources[:4]
        for s_ in check:
            res_p = part.bfs(s_)
            lab_p = part.gather_labels()
            info, res_1 = g.bfs(v, A, s_, desc, fused=True)
            assert info == 0, info
            lab_1 = torch.from_numpy(v.extractTuples()[1]).to(lab_p.device)
            if not torch.equal(lab_p, lab_1) or res_p["edges_traversed"] != res_1["edges_traversed"]:
                bad += 1
        tb = torch.tensor([float(bad)], dtype=torch.float64, device=sdev)
        if world > 1:
            dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        extra["parity"] = {"checked_sources": len(check), "mismatches": int(tb.item()),
                           "checker": "bfs_persistent_kernel on every rank's replica of the graph",
                           "what": "depth labels gathered from all ranks, bit-exact; edges traversed"}
        if tb.item() > 0:
            if rank == 0:
                print(json.dumps({"error": "parity (partitioned traversal)", "mismatching_rank_source_pairs": int(tb.item())}))
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)

        if world > 1 or os.environ.get("GRB_BENCH_TEST_REPLICAS"):
            mine = [sources[(rank * args.steps + i) % len(sources)] for i in range(args.steps)]
            # the N = 1 run's timed region on every rank: K traversals queued, then K waits (labels into K vectors)
            # (one vector per outstanding ticket: the API's contract -- a vector is untouched until its ticket has been waited for)
            vq = [g.Vector(n) for _ in range(args.steps)]

            def queued_pass(srcs_):
                tk = []
                for i, s_ in enumerate(srcs_):
                    info, t_ = g.bfs_enqueue(vq[i], A, s_, desc)
                    assert info == 0, info
                    tk.append(t_)
                tot = 0
                for t_ in tk:
                    info, res = g.bfs_wait(t_)
                    assert info == 0, info
                    tot += res["edges_traversed"]
                return tot

            queued_pass(mine[:args.warmup])
            barrier()
            t0 = time.perf_counter()
            my_edges = queued_pass(mine)
            barrier()
            el = time.perf_counter() - t0
            # the labels a queued timed step wrote, on every rank: the first and the last step against a blocking traversal
            # of the same source (which the partitioned parity block above has compared with the reference)
            for i_ in sorted({0, len(mine) - 1}):
                assert g.bfs(v, A, mine[i_], desc, fused=True)[0] == 0
                if not np.array_equal(vq[i_].extractTuples()[1], v.extractTuples()[1]):
                    print(json.dumps({"error": "parity (replica leg)", "rank": rank, "step": i_}))
                    sys.exit(3)
            t = torch.tensor([el, float(my_edges)], dtype=torch.float64, device=sdev)
            tm, te = t[:1].clone(), t[1:].clone()
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(te, op=dist.ReduceOp.SUM)
            if rank == 0:
                # roofline of the kernel that does the traversing on every rank -- the one-launch BFS, measured on
                # rank 0 exactly as in the N = 1 run (HIP events, algorithmic bytes of an accounting pass).  The
                # partitioned loop reported as `value` is a sequence of short launches around a collective per
                # level; its time is launch / collective latency, not a kernel's.
                timed = [g.bfs(v, A, s_, desc, fused=True, profile=1)[1] for s_ in mine]
                event_ms = sum(r["tight_ms"] for r in timed)
                account = {s_: g.bfs(v, A, s_, desc, fused=True, profile=3)[1]["per_level"] for s_ in sorted(set(mine))}
                total_bytes = sum(sum(level_bytes(account[s_], n)) for s_ in mine)
                ach = total_bytes / (event_ms * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": "bfs_persistent_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                            "traffic": pmc_traffic("bfs_persistent_kernel<1024>")[0] or pmc_traffic("bfs_persistent_kernel")[0],
                            "traffic_source": pmc_traffic("bfs_persistent_kernel<1024>")[1] or pmc_traffic("bfs_persistent_kernel")[1],
                            "launches": len(mine),
                            "avg_launch_ms": round(event_ms / len(mine), 5),
                            "algorithmic_bytes_per_launch": int(total_bytes / len(mine)),
                            "note": "per-GPU kernel of the source_sharded_replicas leg, rank 0"}
            extra["source_sharded_replicas"] = {
                "value": float(te.item()) / float(tm.item()), "unit": "TEPS", "scaling": "weak",
                "steps_per_gpu": args.steps, "ms_per_step_per_gpu": float(tm.item()) / args.steps * 1e3,
                "note": "every rank traverses its own sources on a full replica of the graph, queued as in the N = 1 run "
                        "(grb_bfs_fused_enqueue / grb_bfs_wait); no collective"}
            # the same leg with twelve traversals side by side per launch on every rank (grb_bfs_set_coschedule(12))
            g.bfs_set_coschedule(12)
            queued_pass(mine[:max(args.warmup, 12)])
            barrier()
            t0 = time.perf_counter()
            my_edges_co = queued_pass(mine)
            barrier()
            el_co = time.perf_counter() - t0
            g.bfs_set_coschedule(1)
            for i_ in sorted({0, len(mine) - 1}):
                assert g.bfs(v, A, mine[i_], desc, fused=True)[0] == 0
                if not np.array_equal(vq[i_].extractTuples()[1], v.extractTuples()[1]):
                    print(json.dumps({"error": "parity (replica leg, co-scheduled)", "rank": rank, "step": i_}))
                    sys.exit(3)
            t = torch.tensor([el_co, float(my_edges_co)], dtype=torch.float64, device=sdev)
            tm2, te2 = t[:1].clone(), t[1:].clone()
            if world > 1:
                dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
                dist.all_reduce(te2, op=dist.ReduceOp.SUM)
            extra["source_sharded_replicas"]["coscheduled_12"] = {
                "value": float(te2.item()) / float(tm2.item()), "unit": "TEPS", "scaling": "weak",
                "ms_per_step_per_gpu": float(tm2.item()) / args.steps * 1e3,
                "note": "the same, twelve traversals side by side in one launch on every rank"}

    if rank == 0:
        line = {
            "metric": "BFS TEPS (edges/sec) + SpMV achieved HBM GB/s on RMAT-22, 1/2/4/8 MI355X",
            "value": edges / elapsed, "unit": "TEPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "n": n, "nnz": nnz, "edge_convention": "directed stored edges",
                       "flags": "mxvmode=0 struconly=1 opreuse=1 earlyexit=1 switchpoint=0.01 edgeswitch=%g" % args.edgeswitch,
                       "sources": len(sources), "parallelism": parallelism},
            "roofline": roofline,
        }
        if world > 1 or args.partitioned:
            line["scaling_note"] = ("`value` is the 1-D vertex-partitioned traversal north_star names: ONE traversal at a time over all "
                                    "N GPUs, a launch and an all-gather per level -- STRONG scaling of a 1 GB graph whose traversal "
                                    "is latency-bound on one GPU already (N = 1 of this line's metric is the one-launch kernel, "
                                    "`python bench.py --gpus 1`), so it falls with N.  `source_sharded_replicas` is the WEAK-scaling "
                                    "leg: every GPU traverses its own sources on a replica, no collective.")
        line.update(extra)
        # whatever C libraries still hold in their stdio buffers (RCCL's version banner) goes out first: the JSON
        # line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                                        # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
