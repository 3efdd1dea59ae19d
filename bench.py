#!/usr/bin/env python3
"""bench.py -- count+solve hot path on synthetic MRF reads, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1]

A step = one pass of the hot path over the reads resident in HBM: the count kernel over every
retained read, the batched EM kernel over every event, and the hand-off of the per-event
outputs to the caller's device buffers.
Parsing, the containment filter and the bucket/pool layout are ingest: done once before the
timed region (their wall time is reported under config.ingest_s, never in `value`).

Weak scaling: every rank holds its own shard -- `n_events` events and the reads over them
(the reference shards by gene index range the same way, count/count.cpp:204-215).  The shards
are independent: no collective runs inside the timed loop; one RCCL all-gather after it puts the
per-event tables of all ranks together.

The JSON line carries `roofline` for the count kernel (algorithmic bytes = 8 B per retained
read block + the event tables read once + the class tables written once, SURVEY.md 8(d);
duration from HIP events recorded on the library's stream around the kernel launches) and
`cpu_baseline`: the oracle (a port; never part of the product path) timed single-threaded on
a bounded prefix of the same read stream over the same events.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]/[3]: 100 M reads over 50 k mixed events, 24 chromosomes
    "c3": dict(n_events=50_000, n_reads=100_000_000, R=100, n_chrom=24, types=None, seed=3,
               desc="BASELINE configs[2]: 100M synthetic 100bp reads over 50k mixed events (SE/RI/A5SS/A3SS/MXE/AFE/ALE/T3), 24 chromosomes"),
    # configs[1]: 10 M reads over 5 k two-isoform SE/RI events, one chromosome
    "c2": dict(n_events=5_000, n_reads=10_000_000, R=100, n_chrom=1, types=("SE", "RI"), seed=2,
               desc="BASELINE configs[1]: 10M synthetic 100bp reads over 5k SE/RI events, one chromosome"),
    "c1": dict(n_events=100, n_reads=10_000, R=100, n_chrom=1, types=("SE",), seed=1,
               desc="BASELINE configs[0]: 10k reads over 100 SE events (plumbing)"),
    # one GPU's share of configs[4] (1 B reads / 200 k events over 8 GPUs), skewed read depth (hot genes)
    "c5s": dict(n_events=25_000, n_reads=125_000_000, R=100, n_chrom=24, types=None, seed=5, zipf=True,
                desc="one eighth of BASELINE configs[4]: 125M synthetic 100bp reads over 25k mixed events, Zipf read depth (hot genes), 24 chromosomes"),
    # all of configs[4] on ONE GPU (a size check: ~25 GB of parsed reads on the host, ~10 GB of pools in HBM)
    "c5": dict(n_events=200_000, n_reads=1_000_000_000, R=100, n_chrom=24, types=None, seed=5, zipf=True,
               desc="BASELINE configs[4] whole: 1B synthetic 100bp reads over 200k mixed events, Zipf read depth (hot genes), 24 chromosomes"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 0.25 ms each: the loop itself is ~50 ms
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=15_000_000, help="reads in the cpu_baseline sample (0 = skip)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import lesseq_amd as L

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU implementation")
    # rehearsal on a one-GPU box (developer aid): LSQ_BENCH_REHEARSE=1 puts every rank on cuda:0 and
    # moves the per-event outputs over gloo instead of RCCL; the driver never sets it
    rehearse = os.environ.get("LSQ_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    comm_dev = torch.device("cpu") if (rehearse and world > 1) else dev

    W = WORKLOADS[a.workload]
    types = W["types"] or L.EVENT_TYPES
    spec = L.SynthSpec(W["seed"] + 1000 * rank, W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
    tmp = tempfile.mkdtemp(prefix="lsq_bench_r%d_" % rank)

    # ---- ingest (untimed): annotation -> compiled events -> reads -> bucketed pools in HBM
    t0 = time.time()
    L.synth_write(spec, tmp, "w", write_mrf=False)
    ann = L.Annotation(os.path.join(tmp, "w.interval"), os.path.join(tmp, "w.map"))
    ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
    reads = L.Reads.synthetic(spec, ev)
    n_mrf_reads = len(reads)
    t_gen = time.time() - t0
    ctx = L.Context(local_rank)
    ctx.upload_events(ev)
    t0 = time.time()
    ctx.upload_reads(0, reads)
    t_ingest = time.time() - t0
    del reads
    retained, retained_blocks = ctx.retained(0), ctx.retained_blocks(0)
    n_ev = len(ev)
    n_cls = int(L.lib.lsq_results_num_classes(ctx.h))
    n_iso = ev.total_isoforms

    # per-event outputs as torch tensors so that RCCL can move them
    t_cnt = torch.zeros(max(n_cls, 1), dtype=torch.int64, device=dev)
    t_theta = torch.zeros(max(n_iso, 1), dtype=torch.float64, device=dev)
    t_ll = torch.zeros(max(n_ev, 1), dtype=torch.float64, device=dev)
    if world > 1:
        g_cnt = torch.zeros(world * t_cnt.numel(), dtype=torch.int64, device=comm_dev)
        g_theta = torch.zeros(world * t_theta.numel(), dtype=torch.float64, device=comm_dev)
        g_ll = torch.zeros(world * t_ll.numel(), dtype=torch.float64, device=comm_dev)

    count_ms, solve_ms, fast_ms = [], [], []

    def step(_=None):
        # one pass of the hot path over this rank's shard.  The shards are independent (LESSeq's own
        # scale-out unit is a gene range, each with its own output rows, count/count.cpp:204-215):
        # there is no exchange step, so no collective sits in the timed loop; the per-event tables of
        # all ranks are put together once, after it
        ctx.count()
        ctx.solve()
        ctx.copy_results_device(t_cnt.data_ptr(), t_theta.data_ptr(), t_ll.data_ptr())

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # Kernel durations, right after the timed loop, with the library's HIP events switched on (the
    # event records of a step cost ~25 us of queue time, so the timed loop runs without them).
    # (a) as in the timed loop: steps submitted back to back, so the count kernel runs beside the
    #     tail of the previous step's EM on the second stream; the events of the last step are read.
    #     That duration of lsq_count_fast_kernel is the roofline's.
    # (b) one step at a time (synchronised): the kernels on their own.
    ctx.set_timing(True)
    for _ in range(min(a.steps, 10)):
        for _ in range(3):
            step()
        ctx.synchronize()
        fast_ms.append(ctx.fast_kernel_ms())
    alone_fast_ms = []
    for _ in range(min(a.steps, 10)):
        step()
        ctx.synchronize()
        c, s = ctx.timing()
        count_ms.append(c)
        solve_ms.append(s)
        alone_fast_ms.append(ctx.fast_kernel_ms())
    ctx.set_timing(False)
    fence()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
    tot = torch.tensor([float(retained), float(retained_blocks), float(n_mrf_reads)], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_retained, total_blocks, total_mrf = [float(x) for x in tot.tolist()]

    if world > 1:
        # the whole job's tables on every rank (untimed hand-off; over RCCL on a real multi-GPU node)
        ctx.synchronize()          # the library's stream -> visible to torch's stream
        gather_ms = []
        for _ in range(3):         # first call sets the communicator up; the last is the one reported
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            dist.all_gather_into_tensor(g_cnt, t_cnt.to(comm_dev))
            dist.all_gather_into_tensor(g_theta, t_theta.to(comm_dev))
            dist.all_gather_into_tensor(g_ll, t_ll.to(comm_dev))
            torch.cuda.synchronize()
            gather_ms.append((time.perf_counter() - tg) * 1e3)
        assert int(g_cnt.view(world, -1)[rank].sum().item()) == int(t_cnt.sum().item())

    # sanity of the resident result (every step recomputes it from zeroed tables)
    cnt, bases = ctx.counts()
    theta, ll, iters, flags = ctx.solution()
    assert int(cnt.sum()) == int(t_cnt.sum().item()), "device copy and fetched counts disagree"
    assert 0 < int(cnt.sum()) <= 4 * retained
    assert np.isfinite(theta).all() and abs(float(theta.sum()) - n_ev) < 1e-6 * n_ev

    if rank == 0:
        ck = float(np.mean(count_ms))
        fk = float(np.mean(fast_ms))
        off = ev.class_offsets()
        ev_bytes = 0
        for i in range(0, n_ev):
            K, N = ev.K(i), ev.N(i)
            ev_bytes += 8 * N + 8 * K + 16 + 8 * ((1 << K) - 1) + 8
        alg_bytes = 8.0 * retained_blocks + ev_bytes
        achieved = alg_bytes / (fk * 1e-3) / 1e9
        # HBM bytes per launch of the same kernel from the committed rocprofv3 PMC passes (tools/bench_prof.sh)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % a.workload)
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("count_fast_kernel_hbm_bytes_per_launch")
            traffic_src = "profiles/r01_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2 x FETCH_SIZE + WRITE_SIZE)" % a.workload
        # what a plain streaming read of the same number of bytes reaches on this device (state both ceilings)
        import ctypes as C
        L.lib.lsq_debug_stream_read_rate.argtypes = [C.c_void_p, C.c_ulonglong, C.POINTER(C.c_double)]
        rate = C.c_double(0.0)
        plain_read = rate.value if L.lib.lsq_debug_stream_read_rate(ctx.h, int(max(alg_bytes, 1 << 26)), C.byref(rate)) == 0 else None
        out = {
            "metric": "MRF reads/sec through count+solve",
            "value": total_retained * a.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 coordinates / u64 counters (count), f64 (EM)",
            "data": "synthetic",
            "config": {
                "workload": W["desc"],
                "events_per_gpu": n_ev, "mrf_reads_per_gpu": n_mrf_reads, "retained_reads_per_gpu": retained,
                "retained_blocks_per_gpu": retained_blocks, "buckets": ev.num_buckets,
                "reads_counted": "retained reads (those that pass the load-time containment filter, count/count.cpp:319); off-target reads are dropped at ingest",
                "count_fast_kernel_ms": fk, "count_fast_kernel_ms_alone": float(np.mean(alone_fast_ms)),
                "count_stream_ms_alone": ck, "em_kernel_ms_alone": float(np.mean(solve_ms)),
                "pipeline": "lsq_count on one HIP stream; exception pass, EM, result hand-off and counter zeroing on a second one, beside the next step's count (two counter sets)",
                "valid_read_assignments": int(cnt.sum()), "em_flagged_events": int((flags & 1).sum()),
                "em_max_iters": int(iters.max()) if n_ev else 0,
                "generate_s": t_gen, "ingest_s": t_ingest,
                "gather_ms_after_loop": (gather_ms[-1] if world > 1 else None),
                "parallelism": "events and their reads sharded by rank, no collective in the timed loop; one RCCL all-gather of the per-event tables after it" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": "lsq_count_fast_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                # the same kernel on an otherwise idle device (steps synchronised one by one, no EM beside it)
                "frac_alone": alg_bytes / (float(np.mean(alone_fast_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "plain_read_GBps_this_device": plain_read, "frac_of_plain_read": (achieved / plain_read if plain_read else None),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        # ---- cpu_baseline: the oracle on a bounded prefix of the same stream (rank 0, N = 1 only)
        if world == 1 and a.cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            ns = min(a.cpu_sample, W["n_reads"])
            sspec = L.SynthSpec(W["seed"], W["n_events"], ns, W["R"], W["n_chrom"], types)
            L.synth_write(sspec, tmp, "s", write_mrf=True)
            argv = ["0", "s", "./", "LH_GENE_TXT", os.path.join(tmp, "w.interval"), "UCSC_GENE2ISOFORM", os.path.join(tmp, "w.map"),
                    "0", "1000000000", "MRF_SINGLE", "SHORT_READ", str(W["R"]), os.path.join(tmp, "s.mrf"), str(ns * W["R"])]
            t0 = time.perf_counter()
            rc, _, exact = ob.run("solve", argv)
            dt = time.perf_counter() - t0
            assert rc == 0
            out["cpu_baseline"] = {
                "value": ob.last_n_loaded[0] / dt, "unit": "reads/s",
                "cores": 1, "kind": "port",
                "sample": "first %d MRF reads of the same stream (%d retained) over the same %d events; oracle count+solve "
                          "from MRF text incl. parse, filter and index, single thread, %.1f s" % (ns, ob.last_n_loaded[0], W["n_events"], dt),
            }
            # the reference's own scale-out on the host's cores: one process per slice of the sorted gene list
            # (gene_begin_idx..gene_end_idx, count/count.cpp:204-215), each reading the whole file; wall-clock of the slowest
            import subprocess
            P = max(1, min(os.cpu_count() or 1, 16))
            child = ("import sys,time; sys.path.insert(0, %r); import oracle_binding as ob; a = sys.argv[1:]; t0 = time.perf_counter(); "
                     "rc, _, _ = ob.run('solve', a); print(rc, time.perf_counter() - t0)") % os.path.join(ROOT, "tests")
            t0 = time.perf_counter()
            procs = []
            for p_ in range(P):
                av = list(argv)
                av[7], av[8] = str(W["n_events"] * p_ // P), str(W["n_events"] * (p_ + 1) // P)
                procs.append(subprocess.Popen([sys.executable, "-c", child] + av, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            outs = [q.communicate()[0].split() for q in procs]
            dt_all = time.perf_counter() - t0
            if all(len(o) == 2 and o[0] == "0" for o in outs):
                out["cpu_baseline"]["all_cores"] = {
                    "value": ob.last_n_loaded[0] / dt_all, "unit": "reads/s", "cores": P,
                    "how": "%d oracle processes over disjoint gene_begin_idx..gene_end_idx slices of the same sample (each parses the whole "
                           "file, as the reference's own scale-out does), wall-clock %.1f s" % (P, dt_all),
                }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
