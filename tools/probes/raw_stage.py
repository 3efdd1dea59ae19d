import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lesseq_amd as L
from bench import WORKLOADS
W = WORKLOADS["c3"]
import tempfile
d = tempfile.mkdtemp(dir="/dev/shm")
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], L.EVENT_TYPES, False)
L.synth_write(spec, d, "s", write_mrf=False)
a = L.Annotation(os.path.join(d, "s.interval"), os.path.join(d, "s.map"), 0, 10 ** 9)
ev = L.Events(a, ("SHORT_READ",), (100,))
ctx = L.Context(0); ctx.upload_events(ev)
reads = L.Reads.synthetic(spec, ev)
for r in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    ctx.upload_reads(0, reads)
    t1 = time.perf_counter()
    print(round(t1 - t0, 4), [(s["stage"], round(s["ms"], 3)) for s in ctx.ingest_stages()])
print(ctx.retained(0))
