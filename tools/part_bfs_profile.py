"""Where does a 1-D partitioned traversal spend its time (one rank, RMAT-22)?  GPU box."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
from graphblast_amd import dist as gdist

dev = torch.device("cuda", 0)
src, dst, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(src, dst, n, symmetrize=True)
tptr, tind = gr["csr"]
ptr_host = tptr.cpu().numpy()
sources = [int(np.argmax(np.diff(ptr_host)))] + random_sources(ptr_host, 15, seed=0)
part = gdist.Partition1D(n, tptr, tind, 0, 1, dev)
for s in sources[:4]:
    part.bfs(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in sources:
    r = part.bfs(s)
torch.cuda.synchronize()
print("plain: %.3f ms per traversal, %d levels" % ((time.perf_counter() - t0) / len(sources) * 1e3, r["levels"]))
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = f(*a, **k)
        torch.cuda.synchronize()
        acc.setdefault(name, [0.0, 0])
        acc[name][0] += time.perf_counter() - t; acc[name][1] += 1
        return out
    setattr(obj, name, g)
for nme in ("pull", "push", "apply", "tally"):
    wrap(part.engine, nme)
wrap(part, "_combine")
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in sources:
    part.bfs(s)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / len(sources) * 1e3
print("instrumented (sync around every step): %.3f ms per traversal" % tot)
for k, (t, c) in acc.items():
    print("  %-9s %7.3f ms per traversal  (%d calls, %.1f us each)" % (k, t / len(sources) * 1e3, c, t / c * 1e6))
print("  other     %7.3f ms" % (tot - sum(t for t, _ in acc.values()) / len(sources) * 1e3))
