import os, sys, tempfile
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np
import lesseq_amd as L
from bench import WORKLOADS
W = WORKLOADS["c3"]
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], W["types"] or L.EVENT_TYPES)
tmp = tempfile.mkdtemp()
L.synth_write(spec, tmp, "w", write_mrf=False)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
reads = L.Reads.synthetic(spec, ev)
ctx = L.Context(0); ctx.upload_events(ev); ctx.upload_reads(0, reads)
ctx.count(); ctx.solve()
theta, ll, iters, flags = ctx.solution()
import ctypes as C
n = len(ev)
d2o = np.zeros(n, np.int32)
L.lib.lsq_results_device_order(ctx.h, d2o.ctypes.data_as(C.POINTER(C.c_int)))
it_dev = iters[d2o]     # device order
print("events", n, "iters: mean %.1f median %d p90 %d p99 %d max %d" % (it_dev.mean(), np.median(it_dev), np.percentile(it_dev, 90), np.percentile(it_dev, 99), it_dev.max()))
pad = (-n) % 16
w = np.concatenate([it_dev, np.zeros(pad, it_dev.dtype)]).reshape(-1, 16).max(axis=1)
print("waves", len(w), "wave max iters: mean %.1f median %d p90 %d max %d  sum %d" % (w.mean(), np.median(w), np.percentile(w, 90), w.max(), w.sum()))
wg = np.concatenate([w, np.zeros((-len(w)) % 4, w.dtype)]).reshape(-1, 4)
print("sum over waves / 1024 SIMDs = %.1f iterations per SIMD" % (w.sum() / 1024.0))
Ks = np.array([ev.K(int(i)) for i in d2o])
for k in sorted(set(Ks.tolist())):
    sel = it_dev[Ks == k]
    print("K=%d events=%d iters mean %.1f p99 %d max %d; top5 %s" % (k, len(sel), sel.mean(), np.percentile(sel, 99), sel.max(), np.sort(sel)[-5:]))
# features for a predictor of the iteration count (class counts in device order)
cnt, bases = ctx.counts()
off = ev.class_offsets()
cc = np.zeros((n, 3), np.int64)
for d in range(n):
    o = int(d2o[d])
    k = ev.K(o)
    c = cnt[0, off[o]:off[o] + (1 << k) - 1]
    cc[d, :min(3, len(c))] = c[:3]
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/em_features.npz", iters=it_dev, cc=cc, K=Ks, theta0=np.array([theta[ev.iso_offsets()[int(o)]] if hasattr(ev, "iso_offsets") else 0.0 for o in d2o]))

