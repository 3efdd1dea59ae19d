#!/usr/bin/env python3
"""Developer check of the step pipeline at bench size: N steps submitted back to back, each handing its tables to
buffers of its own; all of them must equal the tables of a synchronous step.  usage: python tools/pipeline_check.py [c3] [steps]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lesseq_amd as L
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = WORKLOADS[wl]
types = W["types"] or L.EVENT_TYPES
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
tmp = tempfile.mkdtemp()
L.synth_write(spec, tmp, "w", write_mrf=False)
ann = L.Annotation(tmp + "/w.interval", tmp + "/w.map")
ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
reads = L.Reads.synthetic(spec, ev)
ctx = L.Context(0)
ctx.upload_events(ev)
ctx.upload_reads(0, reads)
del reads
dev = torch.device("cuda:0")
n_cls = int(L.lib.lsq_results_num_classes(ctx.h))
def bufs():
    return (torch.full((n_cls,), -1, dtype=torch.int64, device=dev), torch.full((ev.total_isoforms,), -1.0, dtype=torch.float64, device=dev),
            torch.full((len(ev),), -1.0, dtype=torch.float64, device=dev))
ref = bufs()
ctx.count(); ctx.solve(); ctx.copy_results_device(ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr()); ctx.synchronize()
outs = [bufs() for _ in range(steps)]
torch.cuda.synchronize()
for b in outs:
    ctx.count(); ctx.solve(); ctx.copy_results_device(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
ctx.synchronize()
bad = 0
for k, b in enumerate(outs):
    ok = bool(torch.equal(b[0], ref[0])) and bool(torch.equal(b[1], ref[1])) and bool(torch.equal(b[2], ref[2]))
    if not ok:
        bad += 1
        print("step %d differs: counts %s theta %s logll %s" % (k, torch.equal(b[0], ref[0]), torch.equal(b[1], ref[1]), torch.equal(b[2], ref[2])))
print("%s: %d pipelined steps, %d differ from the synchronous step; sum of counts %d" % (wl, steps, bad, int(ref[0].sum().item())))
sys.exit(1 if bad else 0)
