import sys, os, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lesseq_amd as L
from bench import WORKLOADS
W = WORKLOADS["c3"]
d = tempfile.mkdtemp(dir="/dev/shm")
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], L.EVENT_TYPES, False)
L.synth_write(spec, d, "s", write_mrf=True)
a = L.Annotation(os.path.join(d, "s.interval"), os.path.join(d, "s.map"), 0, 10 ** 9)
ev = L.Events(a, ("SHORT_READ",), (100,))
ctx = L.Context(0)
for rep in range(2):
    t = [time.perf_counter()]
    def mark(): t.append(time.perf_counter())
    text = ctx.stage_text(os.path.join(d, "s.mrf")); mark()
    ctx.upload_events(ev); mark()
    ctx.upload_reads_text(0, text, free=False); mark()
    ctx.count(); ctx.synchronize(); mark()
    ctx.solve(); ctx.synchronize(); mark()
    c, b = ctx.counts(); mark()
    th, ll, it, fl = ctx.solution(); mark()
    table = L.format_solve(ev, c, b, th, ll, [float(W["n_reads"] * 100)]); mark()
    L.lib.lsq_text_free(text); mark()
    names = ["stage_text", "upload_events", "upload_reads_text", "count(+sync)", "solve(+sync)", "counts()", "solution()", "format_solve", "text_free"]
    print(rep, " ".join("%s=%.1fms" % (n, (t[i + 1] - t[i]) * 1e3) for i, n in enumerate(names)), "total=%.1fms" % ((t[-1] - t[0]) * 1e3))
