#!/usr/bin/env python3
"""Loader timing on the GPU box: MRF text -> retained reads in HBM, host parser + upload against
the device parser.  python tools/load_bench.py [n_reads] [n_events]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lesseq_amd as L  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
n_events = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
d = tempfile.mkdtemp(prefix="lsq_load_", dir="/tmp")
spec = L.SynthSpec(7, n_events, n_reads, 100, 24, L.EVENT_TYPES, False, 0.10)
t0 = time.time()
L.synth_write(spec, d, "s")
t_write = time.time() - t0
mrf = os.path.join(d, "s.mrf")
size = os.path.getsize(mrf)
a = L.Annotation(os.path.join(d, "s.interval"), os.path.join(d, "s.map"), 0, 10 ** 9)
ev = L.Events(a, ("SHORT_READ",), (100,))
ctx = L.Context(0)
ctx.upload_events(ev)
out = {"n_reads": n_reads, "text_bytes": size, "write_s": round(t_write, 2)}
for rep in range(2):
    t0 = time.time()
    ctx.upload_reads_mrf(0, mrf)
    out["device_parse_s_%d" % rep] = round(time.time() - t0, 3)
    out["device_timing_%d" % rep] = ctx.mrf_timing()
kept = ctx.retained(0)
t0 = time.time()
r = L.Reads.from_mrf(mrf, ev)
t1 = time.time()
ctx.upload_reads(0, r)
t2 = time.time()
out["host_parse_s"] = round(t1 - t0, 3)
out["host_upload_s"] = round(t2 - t1, 3)
out["host_threads"] = os.cpu_count()
assert ctx.retained(0) == kept
out["retained"] = kept
print(json.dumps(out))
os.remove(mrf)
