#!/usr/bin/env python3
"""Developer check: wall-clock per count+solve+hand-off step of bench.py's workload, without reading any
event inside or after the loop (for A/B runs of launch-sequence changes).  usage: python tools/step_bench.py [c3] [steps]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LSQ_LIB", os.path.join(ROOT, "lesseq_amd", "_build", "liblesseq_hip_dev.so"))      # developer build: LSQ_EM_CAP etc.
import torch
import lesseq_amd as L
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
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
t_cnt = torch.zeros(int(L.lib.lsq_results_num_classes(ctx.h)), dtype=torch.int64, device=dev)
t_theta = torch.zeros(ev.total_isoforms, dtype=torch.float64, device=dev)
t_ll = torch.zeros(len(ev), dtype=torch.float64, device=dev)
mode = os.environ.get("SB_MODE", "full")        # full | nosolve | nocopy
def step():
    ctx.count()
    if mode != "nosolve": ctx.solve()
    if mode != "nocopy": ctx.copy_results_device(t_cnt.data_ptr(), t_theta.data_ptr() if mode != "nosolve" else None, t_ll.data_ptr() if mode != "nosolve" else None)
if os.environ.get("SB_SORT_EM"):
    # experiment: events placed in the EM grid by their iteration count of a first run (a perfect predictor)
    import ctypes as C, numpy as np
    ctx.count(); ctx.solve()
    theta, ll, iters, flags = ctx.solution()
    d2o = ctx.device_order()
    it_dev = iters[d2o]
    order = np.argsort(it_dev, kind="stable").astype(np.uint32)
    if os.environ["SB_SORT_EM"] == "rev": order = order[::-1].copy()
    pad = (-len(order)) % 16
    order = np.concatenate([order, np.full(pad, 0xFFFFFFFF, np.uint32)])
    L.lib.lsq_debug_set_em_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
    assert L.lib.lsq_debug_set_em_order(ctx.h, order.ctypes.data, len(order), len(order)) == 0
for rep in range(4):
    for _ in range(5): step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    t_sub = time.perf_counter() - t0
    ctx.synchronize()
    print("%s steps=%d ms_per_step=%.4f (host submission %.4f)  checksum=%d" % (wl, steps, (time.perf_counter() - t0) * 1e3 / steps, t_sub * 1e3 / steps, int(t_cnt.sum().item())))

ctx.set_timing(True)
piped, alone = [], []
for _ in range(10):
    for _ in range(3): step()
    ctx.synchronize(); piped.append(ctx.fast_kernel_ms())
for _ in range(10):
    step(); ctx.synchronize(); alone.append(ctx.fast_kernel_ms())
import numpy as np
print("mode=%s fast kernel ms: beside the previous step's tail %.4f, alone %.4f" % (mode, np.mean(piped), np.mean(alone)))
