#!/usr/bin/env python3
"""What an overflowed exception list costs on C3 (GPU box): steps of count + solve with the list as the ingest sizes it, then with
one entry -- every step then ends in a recount of every read -- on 16 workgroups (round 3's launch) and on the default quarter
of the compute units.  python tools/recount_timing.py [workload]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lesseq_amd as L  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W = WORKLOADS[wl]
d = tempfile.mkdtemp(prefix="lsq_rc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
types = W["types"] or L.EVENT_TYPES
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
L.synth_write(spec, d, "s", write_mrf=False)
ev = L.Events(L.Annotation(os.path.join(d, "s.interval"), os.path.join(d, "s.map")), ("SHORT_READ",), (W["R"],))
ctx = L.Context(0)
ctx.upload_events(ev)
reads = L.Reads.synthetic(spec, ev)
L.lib.lsq_set_log_level(0)


def steps(n):
    for _ in range(3):
        ctx.count(); ctx.solve()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.count(); ctx.solve()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {"workload": W["desc"]}
ctx.upload_reads(0, reads)
out["ms_per_step"] = steps(100)
for wgs in (16, 64, 128, 256):
    ctx.set_option("cleanup_workgroups", wgs)
    out["ms_per_step_exception_pass_on_%d_workgroups" % wgs] = min(steps(200), steps(200))
ctx.set_option("cleanup_workgroups", 0)
cnt0 = ctx.counts()[0].copy()
out["exception_pairs"] = ctx.count_status()[0][0]
ctx.set_option("exception_capacity", 1)
ctx.upload_reads(0, reads)
for wgs in (16, 64, 0):
    ctx.set_option("cleanup_workgroups", wgs)
    out["ms_per_step_recounting_on_%s_workgroups" % (wgs or "default_one_per_cu")] = steps(10)
    assert ctx.count_status()[1] == [1] and (ctx.counts()[0] == cnt0).all()
print(json.dumps(out, indent=1))
