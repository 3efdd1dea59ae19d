#!/usr/bin/env python3
"""Developer experiment (GPU box): how well the asymptotic estimate of the EM's stopping iteration (lsq_em.hip, the closed form's
first guess) predicts the iteration counts from an event's counts alone, and what the EM kernel takes with the events placed by
that prediction (lsq_debug_set_em_order) against a placement by the true counts and against none."""
import os, sys, time, tempfile, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import lesseq_amd as L
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W = WORKLOADS[wl]
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], W["types"] or L.EVENT_TYPES, W.get("zipf", False))
tmp = tempfile.mkdtemp()
L.synth_write(spec, tmp, "w", write_mrf=False)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
ctx = L.Context(0); ctx.upload_events(ev); ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
ctx.set_option("em_regroup", 0)
ctx.count(); ctx.solve()
theta, ll, iters, flags = ctx.solution()
cnt, bases = ctx.counts()
off = ev.class_offsets()
n = len(ev)
d2o = ctx.device_order()
K = np.array([ev.K(int(o)) for o in d2o])
cc = np.zeros((n, 3)); G = np.ones((n, 2))
for d in range(n):
    o = int(d2o[d])
    c = cnt[0, off[o]:off[o + 1]]
    cc[d, :min(3, len(c))] = c[:3]
    G[d, 0] = ev.ars(0, o, 0)
    if K[d] > 1: G[d, 1] = ev.ars(0, o, 1)
it_dev = iters[d2o].astype(np.int64)
with np.errstate(all="ignore"):
    n1, n2, n3 = cc[:, 0], cc[:, 1], cc[:, 2]
    nt = n1 + n2 + n3
    a, b, g = n1 / nt, n3 / nt, G[:, 0] - G[:, 1]
    G0, G1 = G[:, 0], G[:, 1]
    al, be, de = a * g + b * G0, a * G1, G1
    qb = de - al; D = qb * qb + 4 * g * be; sq = np.sqrt(D); tq = -0.5 * (qb + np.copysign(sq, qb))
    r1, r2 = tq / g, -be / tq
    k1 = (g * r2 + de) / (g * r1 + de)
    first = (k1 >= 0) & (k1 < 1)
    p = np.where(first, r1, r2); q = np.where(first, r2, r1); kap = np.where(first, k1, 1 / k1)
    lin = g == 0
    p = np.where(lin, a / (1 - b), p); q = np.where(lin, p - 1, q); kap = np.where(lin, b, kap)
    p = np.where((n2 == 0) & (np.abs(p - 1) < 1e-9), 1.0, p); p = np.where((n1 == 0) & (np.abs(p) < 1e-9), 0.0, p)
    pq = np.where(lin, 1.0, p - q)
    x0 = 0.5
    w0 = np.where(lin, x0 - p, (x0 - p) / (x0 - q))
    sp = p * G0 + (1 - p) * G1
    lp = np.where(n1 > 0, n1 * np.log(p * G0), 0) + np.where(n2 > 0, n2 * np.log((1 - p) * G1), 0) + np.where(n3 > 0, n3 * np.log(sp), 0)
    d1 = np.where(n1 > 0, n1 / p, 0) - np.where(n2 > 0, n2 / (1 - p), 0) + np.where(n3 > 0, n3 * g / sp, 0)
    d2 = np.where(n1 > 0, n1 / p ** 2, 0) + np.where(n2 > 0, n2 / (1 - p) ** 2, 0) + np.where(n3 > 0, n3 * g * g / sp ** 2, 0)
    edge = (p == 1.0) | (p == 0.0)
    amp = np.where(edge, np.abs(d1 * pq * (1 - kap) / lp) * np.abs(w0), np.abs(0.5 * d2 * pq * pq * (1 - kap * kap) / lp) * w0 * w0)
    est = 2.0 + np.log(1e-6 / amp) / (np.where(edge, 1.0, 2.0) * np.log(kap))
pred = np.where(np.isfinite(est), est, 2.0)
pred = np.where((K != 2) | (nt <= 0), 1.0, pred)
pred = np.clip(pred, 1, 255)
ok = (K == 2) & (nt > 0)
print("events %d, two-isoform with reads %d; iterations: mean %.1f p99 %d max %d" % (n, ok.sum(), it_dev.mean(), np.percentile(it_dev, 99), it_dev.max()))
err = pred[ok] - np.minimum(it_dev[ok], 255)
print("predicted - actual: mean %.2f, |.| median %.1f p90 %.1f p99 %.1f max %.0f; rank correlation %.4f" % (
    err.mean(), np.median(np.abs(err)), np.percentile(np.abs(err), 90), np.percentile(np.abs(err), 99), np.abs(err).max(),
    np.corrcoef(np.argsort(np.argsort(pred[ok])), np.argsort(np.argsort(it_dev[ok])))[0, 1]))
slow = ok & (it_dev >= 32)
print("of %d events with >= 32 iterations the prediction says >= 32 for %d; of %d predicted >= 32, %d are" % (slow.sum(), (pred[slow] >= 32).sum(), (ok & (pred >= 32)).sum(), (it_dev[ok & (pred >= 32)] >= 32).sum()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "em_predict_%s.npz" % wl), iters=it_dev, cc=cc, G=G, K=K, pred=pred, p=p, q=q, kap=kap, w0=w0, edge=edge)
L.lib.lsq_debug_set_em_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
def em_ms(order):
    if order is not None:
        pad = (-len(order)) % 16
        o = np.concatenate([order.astype(np.uint32), np.full(pad, 0xFFFFFFFF, np.uint32)])
        assert L.lib.lsq_debug_set_em_order(ctx.h, o.ctypes.data, len(o), len(o)) == 0
    ctx.set_timing(True)
    ts = []
    for _ in range(8):
        ctx.count(); ctx.solve(); ctx.synchronize(); ts.append(ctx.timing()[1])
    ctx.set_timing(False)
    return float(np.median(ts[2:]))
print("EM kernel alone, four lanes an event: placement as uploaded %.4f ms" % em_ms(None))
print("   by the true iteration counts (slowest first) %.4f ms" % em_ms(np.argsort(-it_dev, kind="stable")))
print("   by the prediction from the counts            %.4f ms" % em_ms(np.argsort(-pred, kind="stable")))
ctx.close()
