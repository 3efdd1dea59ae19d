#!/usr/bin/env python3
"""Developer aid: what one rank of an N-rank event-sharded job does per step, measured on one GPU (no collective).
usage: python tools/shard_step_bench.py [workload=c3] [N=8]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lesseq_amd as L
from lesseq_amd import dist as ld
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = WORKLOADS[wl]
types = W["types"] or L.EVENT_TYPES
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
L.synth_write(spec, tmp, "w", write_mrf=True)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
ctx = L.Context(0)
for kv in filter(None, os.environ.get("SSB_OPTS", "").split(",")):          # SSB_OPTS="name=value,...": context options
    ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
text = ctx.stage_text(tmp + "/w.mrf")
ctx.upload_events(ev); ctx.upload_reads_text(0, text, free=False)
ctx.count(); ctx.solve()
cnt = ctx.counts()[0].copy()
iters = ctx.solution()[2].copy()
bounds = ev.shard_bounds(N, ld.event_weights(ev, cnt))
order = [int(x) for x in os.environ["SSB_ORDER"].split(",")] if os.environ.get("SSB_ORDER") else list(range(N))
for r in order:
    f, c = bounds[r]
    ev.set_shard(f, c)
    ctx.upload_events(ev); ctx.upload_reads_text(0, text, free=False)
    blk = torch.zeros(max(ev.record_words(f, c), 1), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    def step():
        ctx.count(); ctx.solve(); ctx.pack_results_device(blk.data_ptr())
    for _ in range(20): step()
    ctx.synchronize()
    t0 = time.perf_counter()
    K = 300
    for _ in range(K): step()
    t_submit = (time.perf_counter() - t0) / K * 1e3
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    ctx.set_timing(True)
    step(); ctx.synchronize()
    cm, sm = ctx.timing()
    ctx.set_timing(False)
    import ctypes as C
    B = ev.num_buckets
    so = (C.c_ulonglong * (B + 1))()
    L.lib.lsq_debug_slot_offsets(ctx.h, 0, so, B + 1)
    print("rank %d/%d events %6d max_iters %4d: %.4f ms per step (host submission %.4f; count stream alone %.4f, EM alone %.4f); buckets %d, pooled reads %d" % (r, N, c, int(iters[f:f + c].max()), dt, t_submit, cm, sm, B, so[B]), flush=True)
    sz = np.diff(np.array(so[:], dtype=np.int64))
    print("      reads per bucket: min %d p10 %d median %d mean %.0f p90 %d max %d; skew %.2f" % (sz.min(), np.percentile(sz, 10), np.median(sz), sz.mean(), np.percentile(sz, 90), sz.max(), sz.max() * len(sz) / max(sz.sum(), 1)), flush=True)
    if os.environ.get("LSQ_ABLATE"):
        buf = (C.c_ulonglong * 16)()
        step(); L.lib.lsq_debug_counters(ctx.h, buf)
        print("      parked1 %d parked2 %d walk_steps %d walk_lanes %d; not-in-cell %d one-owner %d shared %d" % (buf[0], buf[1], buf[2], buf[3], buf[5], buf[6], buf[7]), flush=True)
L.lib.lsq_text_free(text)
ctx.close()
