"""Per-rank cost of the device-driven partitioned BFS as a function of the world size, measured on ONE GPU:
every rank of a world of P lives on this device (graphblast_amd.dist.LoopbackGroup -> grb_bfs_part_run_group; the
all-gather is a set of device copies), so the P ranks' launches run one after the other and
(time per traversal) / P is what one rank of a P-GPU run computes per traversal -- everything but RCCL.
    python tools/part_scaling.py [scale] [edge_factor] [worlds, comma separated]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources   # noqa: E402
from graphblast_amd.dist import LoopbackGroup, Partition1D                       # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    dev = torch.device("cuda", 0)
    s, d, n = rmat_edges(scale, ef, seed=1, device=dev)
    gr = finalize_edges(s, d, n, symmetrize=True)
    del s, d
    tptr, tind = gr["csr"]
    ptr_host = tptr.cpu().numpy()
    sources = [int(np.argmax(np.diff(ptr_host)))] + random_sources(ptr_host, 31, seed=0)
    out = {"graph": "rmat%d_ef%d_sym" % (scale, ef), "n": n, "nnz": gr["nnz"], "sources": len(sources)}
    # one rank, three ways: a launch per level (what N > 1 runs, minus the collective), two levels per launch,
    # every level in one launch
    for lpl in (1, 2, 1 << 20):
        part = Partition1D(n, tptr.long(), tind.long(), 0, 1, dev, edgeswitch=0.08, levels_per_launch=lpl)
        for src in sources[:4]:
            part.bfs(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dms, lv, la = 0.0, 0, 0
        for src in sources:
            r = part.bfs(src, want_trace=False)
            dms += r["device_ms"]; lv += r["levels"]; la += r["launches"]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / len(sources)
        out["one_rank_levels_per_launch_%s" % ("all" if lpl > 2 else lpl)] = {
            "ms_per_traversal_wall": round(wall, 4), "ms_per_traversal_device": round(dms / len(sources), 4),
            "levels": round(lv / len(sources), 2), "launches": round(la / len(sources), 2)}
        del part
    for world in worlds:
        grp = LoopbackGroup(n, tptr.long(), tind.long(), world, dev)
        for src in sources[:2]:
            grp.bfs(src, edgeswitch=0.08)
        torch.cuda.synchronize()
        dms = 0.0
        for src in sources[:16]:
            _, res, _ = grp.bfs(src, edgeswitch=0.08)
            dms += res[0]["device_ms"]
        out["world_%d" % world] = {"ms_per_traversal_all_ranks_on_one_gpu": round(dms / 16, 4),
                                   "ms_per_traversal_per_rank": round(dms / 16 / world, 4)}
        del grp
    print(json.dumps(out))


if __name__ == "__main__":
    main()
