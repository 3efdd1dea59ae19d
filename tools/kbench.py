#!/usr/bin/env python3
"""Developer sweep: count-kernel time over LDS budgets / grid multipliers (not part of the product).
usage: python tools/kbench.py c2|c3 [budgets csv] [mults csv]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LSQ_LIB", os.path.join(ROOT, "lesseq_amd", "_build", "liblesseq_hip_dev.so"))      # the ablation switches live in the developer build
import numpy as np
import lesseq_amd as L
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
budgets = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4096").split(",")]
mults = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1").split(",")]
W = WORKLOADS[wl]
types = W["types"] or L.EVENT_TYPES
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
tmp = tempfile.mkdtemp()
L.synth_write(spec, tmp, "w", write_mrf=False)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ref = None
for bud in budgets:
    os.environ["LSQ_LDS_BUDGET"] = str(bud)
    ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
    reads = L.Reads.synthetic(spec, ev)
    ctx = L.Context(0)
    if os.environ.get("KB_SNAP"):
        ctx.set_option("snap_shares", int(os.environ["KB_SNAP"]))
    ctx.set_timing(True)
    ctx.upload_events(ev)
    t0 = time.time(); ctx.upload_reads(0, reads); ti = time.time() - t0
    del reads
    nb = ctx.retained_blocks(0)
    if os.environ.get("KB_DIST"):
        import ctypes as C
        B = L.lib.lsq_events_num_buckets(ev.h)
        buf = (C.c_ulonglong * (B + 1))()
        L.lib.lsq_debug_slot_offsets(ctx.h, 0, buf, B + 1)
        so = np.array(buf[:], dtype=np.int64)
        sz = np.diff(so)
        print("   buckets=%d reads/bucket: min=%d p10=%d median=%d mean=%.0f p90=%d max=%d empty=%d" % (B, sz.min(), np.percentile(sz, 10), np.median(sz), sz.mean(), np.percentile(sz, 90), sz.max(), int((sz == 0).sum())))
        G = 2560
        bounds = so[-1] * np.arange(G + 1) // G
        first = np.searchsorted(so, bounds[:-1], side="right") - 1
        last = np.searchsorted(so, bounds[1:] - 1, side="right") - 1
        span = last - first + 1
        print("   grid=%d buckets touched per workgroup: mean=%.2f p90=%d max=%d" % (G, span.mean(), np.percentile(span, 90), span.max()))
    for m, optset in [(m, o) for m in mults for o in os.environ.get("KB_OPTSETS", "").split(";")]:
      for kv in filter(None, optset.split(",")):          # KB_OPTSETS="name=value,name=value;name=value..." : context options per measurement
          ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
      if optset: print("options: " + optset)
      for abl in [int(x) for x in os.environ.get("KB_ABLATE", "0").split(",")]:
        os.environ["LSQ_ABLATE"] = str(abl)
        os.environ["LSQ_GRID_MULT"] = str(m)
        ts, es = [], []
        for it in range(12):
            ctx.count(); ctx.solve()
            c, s = ctx.timing()
            ts.append(c); es.append(s)
        cnt, bases = ctx.counts()
        chk = (int(cnt.sum()), int(bases.sum()))
        if ref is None: ref = chk
        t = float(np.median(ts[2:]))
        if abl & 256:
            import ctypes as C
            buf = (C.c_ulonglong * 16)()
            L.lib.lsq_debug_counters(ctx.h, buf)
            print("   dbg: two-block looks that ended in a park: block 1 in no one-owner cell %d, in one %d" % (buf[8], buf[9]))
            print("   dbg: one-block wave steps %d, of which with a parked read %d" % (buf[12], buf[11]))
            print("   dbg: parked1=%d parked2=%d walk_steps=%d walk_lanes=%d exceptions=%d  (retained %d)" % (buf[0], buf[1], buf[2], buf[3], buf[4], ctx.retained(0)))
            print("   dbg: parked one-block reads: not in the lane's cell %d, one-owner cell %d, two-owner cell %d" % (buf[5], buf[6], buf[7]))
            print("   dbg: parked two-block followers: first record crosses no junction %d, block 2 runs past the junction's segment %d" % (buf[13], buf[14]))
        if abl & 4194304:
            import ctypes as C
            if os.environ.get("KB_PIPE"):          # the trace of a launch inside the pipelined loop (steps submitted back to back), not of a launch alone
                ctx.set_timing(False)
                for it in range(8):
                    ctx.count(); ctx.solve()
                ctx.synchronize(); ctx.set_timing(True)
            cap = 1 << 16
            buf = (C.c_ulonglong * (4 * cap))(); n = C.c_ulonglong(0); nw = C.c_ulonglong(0)
            L.lib.lsq_debug_wg_trace(ctx.h, buf, cap, C.byref(n), C.byref(nw))
            tr = np.array(buf[:4 * n.value], dtype=np.int64).reshape(-1, 4)         # start, end, walk steps, reads walked
            live = tr[tr[:, 1] > 0]
            t0_, t1_ = live[:, 0].min(), live[:, 1].max()
            st = live[nw.value:] if nw.value < len(live) else live
            dur = (st[:, 1] - st[:, 0]) * 0.01          # us (100 MHz ticks)
            span = (t1_ - t0_) * 0.01
            # workgroups running over time
            ev_t = np.concatenate([live[:, 0], live[:, 1]]); ev_d = np.concatenate([np.ones(len(live)), -np.ones(len(live))])
            o = np.argsort(ev_t, kind="stable"); run = np.cumsum(ev_d[o]); tt = (ev_t[o] - t0_) * 0.01
            peak = run.max()
            area = float(np.sum(run[:-1] * np.diff(tt)))
            def last_above(f):
                idx = np.nonzero(run >= f * peak)[0]
                return tt[idx[-1]] if len(idx) else 0.0
            print("   trace: %d workgroups (%d workers), span %.1f us, peak running %d, slot use %.3f; running >= 90%% of peak until %.1f us, >= 50%% until %.1f us" % (
                len(live), nw.value, span, peak, area / (peak * span), last_above(0.9), last_above(0.5)))
            print("   trace: streaming workgroup durations us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; last start at %.1f us" % (
                dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), (st[:, 0].max() - t0_) * 0.01))
            if os.environ.get("KB_TRACE_OUT"):
                np.save(os.environ["KB_TRACE_OUT"], tr)
                for which, nm in ((0, "slots"), (1, "p1"), (2, "p2"), (3, "cuts"), (4, "park1"), (5, "park2")):
                    b2 = (C.c_ulonglong * (1 << 20))(); k = C.c_ulonglong(0)
                    L.lib.lsq_debug_offsets(ctx.h, 0, which, b2, 1 << 20, C.byref(k))
                    np.save(os.environ["KB_TRACE_OUT"].replace(".npy", "_%s.npy" % nm), np.array(b2[:k.value], dtype=np.int64))
        print("abl=%d " % abl, end="")
        print("lds=%d " % ev.lds_table_bytes, end="")
        print("%s budget=%6d buckets=%5d mult=%2d count_ms med=%.4f min=%.4f  %.0f GB/s (%.1f%% of 8TB/s)  em_ms=%.4f ingest_s=%.2f check=%s" % (
            wl, bud, ev.num_buckets, m, t, min(ts), 8.0 * nb / t / 1e6, 100 * 8.0 * nb / t / 1e6 / 8000, float(np.median(es)), ti, "ok" if chk == ref else "MISMATCH"), flush=True)
    ctx.close()
