#!/usr/bin/env python3
"""The loader chain on the GPU box: MRF text in the page cache -> pools in HBM, per pass.
    python tools/ingest_bench.py [workload c3|c2|c5s] [reps] [sorted 0|1]
Prints one JSON object: per pass the device milliseconds (HIP events on the library's stream), the bytes the pass has to
move at least, GB/s and the fraction of the 8 TB/s HBM peak; wall-clock of stage + upload; LSQ_CLI_TIMING=1 adds the
host-side phase clock on stderr."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lesseq_amd as L  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want_sorted = len(sys.argv) > 3 and sys.argv[3] == "1"
W = WORKLOADS[wl]
scale = float(os.environ.get("LSQ_INGEST_SCALE", "1"))
n_reads, n_events = int(W["n_reads"] * scale), max(10, int(W["n_events"] * scale))
d = tempfile.mkdtemp(prefix="lsq_ing_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
types = W["types"] or L.EVENT_TYPES
spec = L.SynthSpec(W["seed"], n_events, n_reads, W["R"], W["n_chrom"], types, W.get("zipf", False), sorted_reads=want_sorted) if want_sorted \
    else L.SynthSpec(W["seed"], n_events, n_reads, W["R"], W["n_chrom"], types, W.get("zipf", False))
t0 = time.time()
L.synth_write(spec, d, "s", write_mrf=True)
t_write = time.time() - t0
mrf = os.path.join(d, "s.mrf")
size = os.path.getsize(mrf)
a = L.Annotation(os.path.join(d, "s.interval"), os.path.join(d, "s.map"), 0, 10 ** 9)
ev = L.Events(a, ("SHORT_READ",), (W["R"],))
ctx = L.Context(0)
ctx.upload_events(ev)
out = {"workload": wl, "sorted": want_sorted, "n_reads": n_reads, "text_bytes": size, "write_s": round(t_write, 2), "runs": []}
for rep in range(reps):
    ctx.synchronize()
    t0 = time.perf_counter()
    text = ctx.stage_text(mrf)
    t1 = time.perf_counter()
    ctx.upload_reads_text(0, text, free=True)
    t2 = time.perf_counter()
    st = ctx.ingest_stages()
    for s in st:
        s["GBps"] = (s["bytes"] / (s["ms"] * 1e-3) / 1e9) if s["ms"] > 0 else None
        s["frac_of_8TBps"] = (s["GBps"] / 8000.0) if s["GBps"] else None
    out["runs"].append({"stage_text_s": round(t1 - t0, 4), "upload_text_s": round(t2 - t1, 4), "h2d": ctx.mrf_timing(),
                        "device_ms_total": round(sum(s["ms"] for s in st), 3), "stages": st})
out["retained"] = ctx.retained(0)
ctx.count(); ctx.solve()
cnt, bases = ctx.counts()
out["valid_assignments"] = int(cnt.sum())
print(json.dumps(out, indent=1))
os.remove(mrf)
