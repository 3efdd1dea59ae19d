"""Developer A/B of the pipelined step (count + solve + pack) at C3: LSQ_LIB picks a variant library,
AB_OPTS="name=value,..." sets context options.  Run on the GPU box."""
import os, sys, time, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
import lesseq_amd as L
from bench import WORKLOADS
W = WORKLOADS[os.environ.get("AB_WL", "c3")]
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], W["types"] or L.EVENT_TYPES, W.get("zipf", False))
tmp = tempfile.mkdtemp()
L.synth_write(spec, tmp, "w", write_mrf=False)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ev = L.Events(ann, ("SHORT_READ",), (100,))
ctx = L.Context(0)
for kv in filter(None, os.environ.get("AB_OPTS", "").split(",")):
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.upload_events(ev); ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
print("pool format (compact, bytes, reads per pool):", ctx.pool_format(0), flush=True)
blk = torch.zeros(ev.record_words(0, len(ev)), dtype=torch.int64, device="cuda:0"); torch.cuda.synchronize()
def step():
    ctx.count(); ctx.solve(); ctx.pack_results_device(blk.data_ptr())
TOGGLE = os.environ.get("AB_TOGGLE")          # an option flipped 0/1 (or "name:a:b": between a and b) between repetitions of the same process
TVALS = (0.0, 1.0)
if TOGGLE and ":" in TOGGLE:
    TOGGLE, va, vb = TOGGLE.split(":")
    TVALS = (float(va), float(vb))
for rep in range(8 if TOGGLE else 3):
    if TOGGLE: ctx.set_option(TOGGLE, TVALS[rep & 1])
    for _ in range(40): step()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(300): step()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e3
    ctx.set_timing(True)
    for _ in range(3): step()
    ctx.synchronize(); fk = ctx.fast_kernel_ms(); ctx.set_timing(False)
    print("%s: %.4f ms per step, count kernel in the pipeline %.4f" % (os.environ.get("LSQ_LIB", "release").split("_")[-1] + " " + os.environ.get("AB_OPTS", "") + (" %s=%g" % (TOGGLE, TVALS[rep & 1]) if TOGGLE else ""), dt, fk), flush=True)
