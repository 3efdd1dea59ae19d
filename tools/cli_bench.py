#!/usr/bin/env python3
"""Wall-clock of the drop-in executables from MRF text on the GPU box (SURVEY 8(d) item iii):
python tools/cli_bench.py [n_reads] [n_events].  Writes a synthetic set under /tmp, runs
lesseq_amd/bin/count and lesseq_amd/bin/solve as child processes, prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import lesseq_amd as L  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
n_events = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
d = tempfile.mkdtemp(prefix="lsq_cli_", dir="/tmp")
spec = L.SynthSpec(3, n_events, n_reads, 100, 24, L.EVENT_TYPES, False, 0.10)
t0 = time.time()
L.synth_write(spec, d, "s")
out = {"n_reads": n_reads, "n_events": n_events, "write_s": round(time.time() - t0, 2),
       "text_bytes": os.path.getsize(os.path.join(d, "s.mrf"))}
base = ["0", "s", "./", "LH_GENE_TXT", "s.interval", "UCSC_GENE2ISOFORM", "s.map", "0", "100000000"]
for tool, tail in (("count", ["MRF_SINGLE", "SHORT_READ", "100", "s.mrf"]), ("solve", ["MRF_SINGLE", "SHORT_READ", "100", "s.mrf", str(n_reads * 100)])):
    exe = os.path.join(ROOT, "lesseq_amd", "bin", tool)
    best = None
    for rep in range(2):
        t0 = time.time()
        p = subprocess.run([exe] + base + tail, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        assert p.returncode == 0, (tool, p.returncode)
        best = dt if best is None else min(best, dt)
    out[tool + "_wall_s"] = round(best, 3)
    out[tool + "_rows"] = p.stdout.count(b"\n")
out["mrf_reads_per_s_count"] = round(n_reads / out["count_wall_s"])
print(json.dumps(out))
os.remove(os.path.join(d, "s.mrf"))
