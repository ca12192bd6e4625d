"""Streaming element-wise ops and CC on large vectors/graphs: achieved GB/s sanity check."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.set_lazy(0)                     # one kernel per call: this script measures kernels
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
d = g.Descriptor(); d.loadArgs()
u, v, w = g.Vector(n), g.Vector(n), g.Vector(n)
u.fill(1.5); v.fill(2.5); w.fill(0.0)
def t(label, fn, bytes_per_elt, reps=10):
    fn(); g.timer_start()
    for _ in range(reps): fn()
    ms = g.timer_stop() / reps
    print("%-28s %8.3f ms  %7.1f GB/s" % (label, ms, bytes_per_elt * n / ms / 1e6))
t("eWiseAdd dense,dense", lambda: g.eWiseAdd(w, None, None, "PlusMultiplies", u, v, d), 12)
t("eWiseMult dense,dense", lambda: g.eWiseMult(w, None, None, "PlusMultiplies", u, v, d), 12)
t("reduce plus", lambda: g.reduce(None, "PlusMonoid", u, d), 4)
t("assign dense mask", lambda: g.assign(w, u, None, 3.0, None, n, d), 8)
t("fill", lambda: w.fill(1.0), 4)
