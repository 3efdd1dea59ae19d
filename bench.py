#!/usr/bin/env python3
"""bench.py -- count+solve hot path on synthetic MRF reads, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|c5s|c5] [--mode strong|weak]

The job: BASELINE.json configs[2] (default workload c3: 100 M synthetic 100-bp MRF reads over 50 k mixed local
events on 24 chromosomes).  Rank 0 writes the job's files once (annotation + MRF text, deterministic in the
seed); every rank copies the MRF text to its GPU and parses and ingests it there (lsq_text_stage /
lsq_reads_upload_text: the device parser, the load-time containment filter, the bucket / pool layout).

A step = one pass of the hot path over the reads resident in HBM: the count kernels over every retained read of
the rank's events, the batched EM over those events, the per-event records packed in output order on the device
-- and, with N > 1, the RCCL all-gather that puts the records of all ranks on every rank, inside the timed loop.

N > 1, --mode strong (default; BASELINE.json configs[3]): ONE job for all ranks.  The output-ordered events are
cut into N contiguous slices of equal read weight (reads per event from a first unsharded count,
lsq_shard_bounds; the reference's own scale-out unit is such a slice, count/count.cpp:204-215), every rank
ingests the reads of its slice, and `value` = the job's retained reads x steps / time: strong scaling.  After the
loop rank 0 checks that the gathered tables equal the unsharded run's.
With the default workload the line of an N > 1 run is the c3 job (BASELINE configs[3]) and carries, under
`config.c5_*`, the same measurement of BASELINE configs[4] (1 B reads, 200 k events, Zipf depth: a millisecond of
count work per step on one GPU, where the fixed terms of a step weigh less), and under `config.e2e_cli_gpus_*` the
wall-clock of `LSQ_GPUS=N lesseq_amd/bin/solve` from the MRF text as a child process, its table compared with the
one-GPU table.  Nothing on that path waits without a limit: the RCCL communicator of liblesseq_rccl is made with a time
limit, the first in-loop gather is waited for with one, and on expiry the gather falls back to torch.distributed's
all_gather_into_tensor; `config.gather_through` / `config.lsq_comm_size` say which ran.
N > 1, --mode weak: every rank runs its own job of the workload's size (seed + 1000 x rank), no collective in the
loop (the path has no exchange step); one all-gather after it.

The JSON line carries `roofline` for the count kernel (algorithmic bytes = 8 B per retained read block + the event
tables read once + the class tables written once; duration from HIP events recorded on the
library's stream around the kernel launches), the end-to-end figures of SURVEY 8(d) under `config` (kernel-resident
is `value`; device-resident from parsed arrays in host memory; from MRF text, in-process and as the `solve`
executable), and `cpu_baseline`: the oracle (a port; never part of the product path) on a bounded prefix of the
same read stream over the same events.
"""
import argparse
import datetime
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
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
    # all of configs[4]: on one GPU a size check (37 GB of MRF text); over 8 GPUs (--gpus 8) the config itself
    "c5": dict(n_events=200_000, n_reads=1_000_000_000, R=100, n_chrom=24, types=None, seed=5, zipf=True,
               desc="BASELINE configs[4] whole: 1B synthetic 100bp reads over 200k mixed events, Zipf read depth (hot genes), 24 chromosomes"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate
# the loader chain's passes (lsq_ingest_stage_name) and the kernels each launches, for the PMC rows of the traffic pass
INGEST_STAGE_KERNELS = {
    "newline_count": ("lsq_mrf_newline_count_kernel",),
    "route": ("lsq_mrf_route_fast_kernel", "lsq_mrf_route_kernel", "lsq_mrf_route_lines_kernel", "lsq_route_raw_kernel"),
    "partition_count": ("lsq_part_hist_kernel",),
    "partition_scatter": ("lsq_part_scatter_kernel", "lsq_piece_expand_kernel"),
    "group_classify": ("lsq_group_classify_kernel",),
    "group_offsets": ("lsq_ingest_offsets_kernel",),
    "group_place": ("lsq_group_place_kernel", "lsq_ingest_pad_kernel", "lsq_ingest_nblock_kernel"),
}


def shared_dir(tag):
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    # (LSQ_BENCH_TAG: a child run of this script -- measure_traffic_live -- keeps its files apart from its parent's)
    return os.path.join(base, "lsq_bench_%s_%s%s" % (os.environ.get("USER", "u"), tag, os.environ.get("LSQ_BENCH_TAG", "")))


def measure_traffic_live(wl_name, timeout=300):
    """HBM bytes per launch of the count kernel, measured now: two short child runs of this script under `rocprofv3 --pmc`
    (FETCH_SIZE, then WRITE_SIZE: separate passes, counters only, no trace domain), corrected as MI355X_MICROARCH.md's HBM section
    says (both in KiB; gfx950 counts a wide coalesced read at half its bytes: 2 x FETCH_SIZE + WRITE_SIZE).
    Returns (bytes or None, note)."""
    import csv, glob, shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found", None
    base = tempfile.mkdtemp(prefix="lsq_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", LSQ_BENCH_TAG="_traffic%d" % os.getpid())
    means = {}
    ingest = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, ctr)
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--workload", wl_name, "--steps", "5", "--warmup", "1", "--no-e2e", "--cpu-sample", "0", "--no-traffic"]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, "the %s pass did not finish in %d s" % (ctr, timeout), None
            if p.returncode != 0:
                return None, "the %s pass ended with %d: %s" % (ctr, p.returncode, p.stderr.decode(errors="replace")[-300:]), None
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    kn = row.get("Kernel_Name", "")
                    if "lsq_count_fast_kernel" in kn:
                        tot += float(row["Counter_Value"]); n += 1
                    for stage, pats in INGEST_STAGE_KERNELS.items():
                        if any(p_ in kn for p_ in pats):
                            # the first ingest of the child run is the text's (the only one under --no-e2e); a later one would be added to it
                            ingest.setdefault(stage, {}).setdefault(ctr, 0.0)
                            ingest[stage][ctr] += float(row["Counter_Value"])
            if n == 0:
                return None, "no %s rows for the count kernel" % ctr, None
            means[ctr] = (tot / n, n)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    b = (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]) * 1024.0
    ingest_bytes = {st: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for st, v in ingest.items()}
    return b, ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each over a 6-step child run of this command on this device "
               "(%d / %d launches); 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950 counts a wide coalesced read at half its bytes)" % (means["FETCH_SIZE"][1], means["WRITE_SIZE"][1])), ingest_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # ~0.25 ms each at N = 1: the loop itself is ~50 ms
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="strong", choices=("strong", "weak"), help="N > 1: one job sharded by events (strong), or one job per rank (weak)")
    ap.add_argument("--cpu-sample", type=int, default=15_000_000, help="reads in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs")
    ap.add_argument("--no-c5", action="store_true", help="N > 1: skip the configs[4] leg (config.c5_*)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live HBM-traffic measurement (two short child runs under rocprofv3 --pmc)")
    a = ap.parse_args()

    # stdout carries the one JSON line and nothing else: libraries that greet on stdout (RCCL's version banner, gloo's
    # connection notes) are sent to stderr for the duration
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    strong = world > 1 and a.mode == "strong"

    import numpy as np
    import torch
    import torch.distributed as dist
    import lesseq_amd as L
    from lesseq_amd import dist as ld

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU implementation")
    # rehearsal on a one-GPU box (developer aid): LSQ_BENCH_REHEARSE=1 puts every rank on cuda:0 and
    # moves the per-event records over gloo instead of RCCL; the driver never sets it
    rehearse = os.environ.get("LSQ_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            # (a collective that cannot complete raises after this long instead of waiting for ever)
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=float(os.environ.get("LSQ_BENCH_PG_TIMEOUT", "600"))))
    on_host = rehearse and world > 1          # gloo moves host tensors

    env = dict(rank=rank, world=world, local_rank=local_rank, strong=strong, rehearse=rehearse, on_host=on_host, dev=dev)
    out = run_workload(env, a, a.workload, True)
    # N > 1, default job: the same measurement on BASELINE configs[4] (1 B reads: a millisecond of count work per step on one
    # GPU) rides along under config.c5_* -- strong scaling of the 0.13 ms c3 job is bounded by the fixed terms of a step
    if strong and a.workload == "c3" and not a.no_c5:
        # The parsed line is c3's and must come out whatever the c5 leg does: a watchdog per rank ends the process after
        # LSQ_BENCH_C5_LIMIT seconds (default 300) -- rank 0 prints the c3 line with c5_error first -- and an exception on one
        # rank (which leaves the others inside a collective) is reported the same way.
        printed = threading.Lock()

        def give_up(why):
            if not printed.acquire(blocking=False):
                return
            if rank == 0:
                out["config"]["c5_error"] = why
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            os._exit(3)           # the c3 line is out; the run itself did not end well and says so
        limit = float(os.environ.get("LSQ_BENCH_C5_LIMIT", "300"))
        dog = threading.Timer(limit, give_up, args=("the configs[4] leg did not finish within %.0f s" % limit,))
        dog.daemon = True
        dog.start()
        try:
            c5 = run_workload(env, a, "c5", False)
        except BaseException as e:
            give_up("%s: %s" % (type(e).__name__, e))
        dog.cancel()
        if rank == 0:
            out["config"].update({
                "c5_workload": c5["config"]["workload"], "c5_value_reads_per_s": c5["value"], "c5_ms_per_step": c5["ms_per_step"],
                "c5_steps": c5["steps"], "c5_retained_reads": c5["config"]["retained_reads"], "c5_events": c5["config"]["events"],
                "c5_roofline_frac_rank0": c5["roofline"]["frac"], "c5_roofline_frac_alone_rank0": c5["roofline"]["frac_alone"],
                "c5_count_launch_rank0": c5["config"]["count_launch_rank0"],
                "c5_per_rank": c5["config"]["per_rank"], "c5_gather_ms_alone": c5["config"]["gather_ms_alone"],
                "c5_gather_through": c5["config"]["gather_through"], "c5_lsq_comm_size": c5["config"]["lsq_comm_size"],
                "c5_n1_ms_per_step_same_box": c5["config"]["n1_ms_per_step_same_box"], "c5_speedup_vs_n1": c5["config"]["speedup_vs_n1"],
                "c5_efficiency": c5["config"]["efficiency"],
                "c5_tables_equal_unsharded_run": c5["config"]["tables_equal_unsharded_run"],
                "c5_generate_s": c5["config"]["generate_s"], "c5_ingest_from_text_s": c5["config"]["ingest_from_text_s"],
                "c5_em_replayed_events": c5["config"]["em_replayed_events"],
                "c5_note": "BASELINE configs[4] measured by the same code in the same run (strong scaling: one job, events sharded by index, gather in the loop); "
                           "c5_value_reads_per_s = the job's retained reads x steps / max-over-ranks time",
            })
        if not printed.acquire(blocking=False):
            return                  # (the watchdog fired while the results were being folded in)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if env.get("comm") is not None:
        env["rccl"].lsq_comm_destroy(env["comm"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def setup_lsq_comm(env):
    """liblesseq_rccl's communicator for this run (env["rccl"], env["comm"]; None when it did not come up on EVERY rank):
    rank 0's RCCL id travels through torch.distributed, every rank joins with a time limit (lsq_comm_init_rank_for), and the
    ranks agree on the outcome -- one that failed or timed out sends everybody to the torch.distributed gather."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    import lesseq_amd as L
    rank, world, local_rank, dev = env["rank"], env["world"], env["local_rank"], env["dev"]
    limit = float(os.environ.get("LSQ_COLLECTIVE_TIMEOUT", "120"))
    env["rccl"], env["comm"], env["lsq_comm_error"] = None, None, None
    rccl, comm, ok, why = None, C.c_void_p(), 1, None
    try:
        rccl = C.CDLL(os.path.join(os.path.dirname(L._lib.LIB_PATH), "liblesseq_rccl.so"))
        rccl.lsq_comm_unique_id.argtypes = [C.c_void_p]
        rccl.lsq_comm_init_rank_for.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_void_p)]
        rccl.lsq_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        rccl.lsq_step_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        rccl.lsq_comm_destroy.argtypes = [C.c_void_p]
        rccl.lsq_comm_abort.argtypes = [C.c_void_p]
        rccl.lsq_comm_size.argtypes = [C.c_void_p]
        rccl.lsq_rccl_last_error.restype = C.c_char_p
        v = int(rccl.lsq_rccl_version())
        env["rccl_version"] = "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v else None
    except Exception as e:
        ok, rccl, why = 0, None, "liblesseq_rccl.so: %s" % e
    # every rank takes part in the exchanges below whatever happened above (a rank that skipped one would leave the
    # others waiting); the communicator is only set up when every rank has the library AND rank 0 has an id
    uid = torch.zeros(129, dtype=torch.uint8)
    if rank == 0 and ok:
        buf = (C.c_ubyte * 128)()
        ok = 1 if rccl.lsq_comm_unique_id(buf) == 0 else 0
        uid = torch.tensor(list(buf) + [ok], dtype=torch.uint8)
    have = torch.tensor([ok], dtype=torch.int32, device=dev)
    if world > 1:
        uid_d = uid.to(dev)
        dist.broadcast(uid_d, 0)
        uid = uid_d.cpu()
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
    ok = 1 if (int(have.item()) == 1 and int(uid[128]) == 1) or (world == 1 and ok) else 0
    if ok:
        idb = (C.c_ubyte * 128)(*uid[:128].tolist())
        st = rccl.lsq_comm_init_rank_for(world, rank, idb, local_rank, limit, C.byref(comm))
        if st != 0:
            ok, why = 0, rccl.lsq_rccl_last_error().decode("utf-8", "replace")
    elif why is None:
        why = "no RCCL id, or a rank without the library"
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if comm.value:
            rccl.lsq_comm_abort(comm)        # (abort, not destroy: some peer never joined)
        env["lsq_comm_error"] = why or "another rank failed to join"
        return
    env["rccl"], env["comm"] = rccl, comm


def run_workload(env, a, wl_name, primary):
    """One workload, measured as the module docstring says; returns the JSON line as a dict on rank 0 (None elsewhere).
    primary: the run's own workload (end-to-end legs, cpu_baseline); otherwise only the timed loop and what describes it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import lesseq_amd as L
    from lesseq_amd import dist as ld
    rank, world, local_rank, strong = env["rank"], env["world"], env["local_rank"], env["strong"]
    rehearse, on_host, dev = env["rehearse"], env["on_host"], env["dev"]
    W = dict(WORKLOADS[wl_name])
    scale = float(os.environ.get("LSQ_BENCH_SCALE_%s" % wl_name.upper(), "1"))      # rehearsals on one GPU (developer aid): a smaller job of the same shape
    if scale != 1:
        W["n_reads"], W["n_events"] = max(1000, int(W["n_reads"] * scale)), max(10, int(W["n_events"] * scale))
        W["desc"] += " -- SCALED by %g for a rehearsal" % scale
    no_e2e = a.no_e2e or not primary
    cpu_sample = a.cpu_sample if primary else 0
    types = W["types"] or L.EVENT_TYPES
    job_rank = 0 if (strong or world == 1) else rank            # weak mode: a job of its own per rank
    spec = L.SynthSpec(W["seed"] + 1000 * job_rank, W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False))
    tmp = shared_dir("%s_%s_j%d" % (os.environ.get("MASTER_PORT", "solo"), wl_name, job_rank))
    writer = rank == 0 or not strong

    # ---- the job's files (untimed): annotation + MRF text, written once per job
    t0 = time.time()
    if writer:
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        L.synth_write(spec, tmp, "w", write_mrf=True)
    t_gen = time.time() - t0
    if world > 1:
        dist.barrier()
    mrf = os.path.join(tmp, "w.mrf")
    argv_solve = ["0", "w", "./", "LH_GENE_TXT", os.path.join(tmp, "w.interval"), "UCSC_GENE2ISOFORM", os.path.join(tmp, "w.map"),
                  "0", "1000000000", "MRF_SINGLE", "SHORT_READ", str(W["R"]), mrf, str(W["n_reads"] * W["R"])]

    # ---- (iii) the `solve` executable on the text, as a child process, before this process holds a context
    e2e = {}

    def run_cli(tool, av, extra_env=None, timeout=600):
        """wall-clock, stdout and the phase breakdown (LSQ_CLI_TIMING lines on stderr) of one executable run"""
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "lesseq_amd", "bin", tool)] + av, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, LSQ_CLI_TIMING="1", **(extra_env or {})), timeout=timeout)
        dt = time.perf_counter() - t0
        phases = {}
        for ln in p.stderr.decode("utf-8", "replace").splitlines():
            if ln.startswith("[timing]"):
                name, _, sec = ln[len("[timing]"):].strip().rpartition(" s")[0].rpartition(" ")
                try:
                    phases[name.strip()] = phases.get(name.strip(), 0.0) + float(sec)
                except ValueError:
                    pass
        return p.returncode, dt, p.stdout, phases, p.stderr[-2000:].decode("utf-8", "replace")
    if world == 1 and not no_e2e:
        for tool, av in (("count", argv_solve[:-1]), ("solve", argv_solve)):
            rc, dt, stdout, phases, err = run_cli(tool, av)
            assert rc == 0, "lesseq_amd/bin/%s failed: %s" % (tool, err)
            e2e["e2e_cli_%s_s" % tool] = dt
            e2e["e2e_cli_%s_rows" % tool] = stdout.count(b"\n")
            if tool == "solve":
                e2e["e2e_cli_solve_table_sha256"] = hashlib.sha256(stdout).hexdigest()
                e2e["e2e_cli_phases"] = phases
        e2e["e2e_cli_solve_mrf_reads_per_s"] = W["n_reads"] / e2e["e2e_cli_solve_s"]
        e2e["e2e_cli_note"] = ("wall-clock of lesseq_amd/bin/{count,solve} as child processes on the %d-byte MRF text (page cache): process start, HIP "
                               "initialisation, annotation load, text copy, device parse, ingest, count, EM, formatting, every output row; e2e_cli_phases: the "
                               "executable's own phase clock (LSQ_CLI_TIMING), seconds, the solve run" % os.path.getsize(mrf))
    if strong and not no_e2e:
        # N > 1: the same from the text over N GPUs -- `LSQ_GPUS=N solve`, one process, a host thread per GPU, the RCCL gather of
        # liblesseq_rccl (ncclCommInitAll) -- beside the one-GPU run of the same executable, tables compared.  Rank 0 runs the two
        # children while the other ranks wait for a file (their GPUs idle: no collective is pending anywhere).
        done = os.path.join(tmp, "cli_done.json")
        if rank == 0:
            res = {}
            try:
                rc1, dt1, out1, ph1, err1 = run_cli("solve", argv_solve, timeout=300)
                res["e2e_cli_gpus_1_solve_s"] = dt1 if rc1 == 0 else None
                menv = {"LSQ_GPUS": str(world)}
                if rehearse:
                    menv.update(LSQ_DEVICES=",".join(["0"] * world), LSQ_GATHER="host")
                rcn, dtn, outn, phn, errn = run_cli("solve", argv_solve, menv, timeout=300)
                res["e2e_cli_gpus_n"] = world
                res["e2e_cli_gpus_n_solve_s"] = dtn if rcn == 0 else None
                res["e2e_cli_gpus_n_exit"] = rcn
                if rcn != 0:
                    res["e2e_cli_gpus_n_error"] = errn[-600:]
                res["e2e_cli_gpus_n_table_equals_one_gpu_table"] = bool(rc1 == 0 and rcn == 0 and out1 == outn)
                res["e2e_cli_gpus_n_table_sha256"] = hashlib.sha256(outn).hexdigest() if rcn == 0 else None
                res["e2e_cli_gpus_n_phases"] = phn
                res["e2e_cli_gpus_1_phases"] = ph1
                if rcn == 0:
                    res["e2e_cli_gpus_n_mrf_reads_per_s"] = W["n_reads"] / dtn
                    res["e2e_cli_gpus_n_speedup_over_one_gpu_executable"] = (dt1 / dtn) if rc1 == 0 else None
                # ... and the same job sharded by READS (LSQ_SHARD=reads: every GPU copies and parses a byte range of the text, one
                # all-reduce of the class counts, GPU 0 solves): the mode in which the loader scales with the GPUs
                rcr, dtr, outr, phr, errr = run_cli("solve", argv_solve, dict(menv, LSQ_SHARD="reads"), timeout=300)
                res["e2e_cli_gpus_n_by_reads_solve_s"] = dtr if rcr == 0 else None
                res["e2e_cli_gpus_n_by_reads_exit"] = rcr
                if rcr != 0:
                    res["e2e_cli_gpus_n_by_reads_error"] = errr[-600:]
                else:
                    res["e2e_cli_gpus_n_by_reads_mrf_reads_per_s"] = W["n_reads"] / dtr
                    res["e2e_cli_gpus_n_by_reads_speedup_over_one_gpu_executable"] = (dt1 / dtr) if rc1 == 0 else None
                res["e2e_cli_gpus_n_by_reads_table_equals_one_gpu_table"] = bool(rc1 == 0 and rcr == 0 and out1 == outr)
                res["e2e_cli_gpus_n_by_reads_phases"] = phr
                res["e2e_cli_gpus_n_note"] = ("LSQ_GPUS=%d lesseq_amd/bin/solve as a child process on the MRF text: pre-pass count on GPU 0, slices of equal read weight, every "
                                              "GPU's thread parses the text and ingests its slice, count + EM + pack, %s, one table printed" %
                                              (world, "blocks through host memory (rehearsal on one GPU)" if rehearse else "ncclAllGather"))
            except Exception as e:
                res["e2e_cli_gpus_n_error"] = "%s: %s" % (type(e).__name__, e)
            with open(done + ".tmp", "w") as f:
                json.dump(res, f)
            os.rename(done + ".tmp", done)
            e2e.update(res)
        else:
            t_wait = time.time()
            while not os.path.exists(done) and time.time() - t_wait < 700:
                time.sleep(0.05)
        dist.barrier()

    # ---- ingest (untimed): annotation -> compiled events -> MRF text -> HBM -> parsed, filtered, pooled
    ann = L.Annotation(argv_solve[4], argv_solve[6])
    ev = L.Events(ann, ("SHORT_READ",), (W["R"],))
    n_ev = len(ev)
    ctx = L.Context(local_rank)
    t0 = time.perf_counter()
    text = ctx.stage_text(mrf)
    ctx.upload_events(ev)
    ctx.upload_reads_text(0, text, free=False)
    t_ingest = time.perf_counter() - t0
    ingest_stages = ctx.ingest_stages()                # device time and minimum bytes of every pass of the loader chain, this ingest
    ingest_h2d_ms = ctx.mrf_timing()["h2d_ms"]
    ingest_paths = ctx.parse_paths()
    ctx.count()
    ctx.solve()
    cnt_full, bases_full = [x.copy() for x in ctx.counts()]
    theta_full, ll_full, iters_full, flags_full = [x.copy() for x in ctx.solution()]
    t_first = time.perf_counter() - t0
    job_retained, job_blocks = ctx.retained(0), ctx.retained_blocks(0)
    if world == 1 and not no_e2e:
        # (iii, in-process) text in the page cache -> the formatted solve table, given the compiled events
        table = L.format_solve(ev, cnt_full, bases_full, theta_full, ll_full, [float(W["n_reads"] * W["R"])])
        e2e["e2e_from_text_s"] = time.perf_counter() - t0
        e2e["e2e_from_text_reads_per_s"] = W["n_reads"] / e2e["e2e_from_text_s"]
        e2e["e2e_from_text_note"] = "in-process: lsq_text_stage (H2D of the text) + event upload + device parse + ingest + count + EM + fetch + lsq_format_solve"
        e2e["e2e_from_text_table_matches_cli"] = hashlib.sha256(table.encode()).hexdigest() == e2e.get("e2e_cli_solve_table_sha256")
    bounds = [(0, n_ev)]
    n1_same_box = None
    if strong:
        # N > 1: the whole job is resident on every GPU at this point (the pre-pass count that weighs the slices).  Twenty unsharded
        # steps of it -- count + EM + pack, no gather -- give the one-GPU step time of THIS box, so the line's speed-up does not
        # rest on another box's one-GPU run (boxes differ by ~5 %).  Every rank runs them (same work everywhere); rank 0's counts.
        blk1 = torch.zeros(max(ev.record_words(0, n_ev), 1), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for k in range(5):
            ctx.count(); ctx.solve(); ctx.pack_results_device(blk1.data_ptr())
        ctx.synchronize()
        t_n1 = time.perf_counter()
        for k in range(20):
            ctx.count(); ctx.solve(); ctx.pack_results_device(blk1.data_ptr())
        ctx.synchronize()
        n1_same_box = (time.perf_counter() - t_n1) / 20 * 1e3
        del blk1
    if strong:
        # slices of equal read weight; then this rank's slice only: event tables re-planned, reads re-ingested
        bounds = ev.shard_bounds(world, ld.event_weights(ev, cnt_full))
        ev.set_shard(*bounds[rank])
        ctx.upload_events(ev)
        ctx.upload_reads_text(0, text, free=False)
    L.lib.lsq_text_free(text)
    n_ev_mine = bounds[rank][1] if strong else n_ev
    my_weight = float(ld.event_weights(ev, cnt_full)[bounds[rank][0]:bounds[rank][0] + bounds[rank][1]].sum()) if strong else float(cnt_full.sum())

    # ---- (ii) device-resident end to end, from parsed arrays in host memory (N = 1)
    if world == 1 and not no_e2e:
        reads = L.Reads.synthetic(spec, ev)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.upload_reads(0, reads)
        ctx.count()
        ctx.solve()
        c2, b2 = ctx.counts()
        th2 = ctx.solution()[0]
        e2e["e2e_device_resident_s"] = time.perf_counter() - t0
        e2e["e2e_device_resident_reads_per_s"] = W["n_reads"] / e2e["e2e_device_resident_s"]
        e2e["e2e_device_resident_note"] = "parsed blocks in host memory -> H2D -> ingest kernels -> count -> EM -> D2H of the tables (PCIe-inclusive; never `value`)"
        assert np.array_equal(c2, cnt_full) and np.array_equal(b2, bases_full) and np.array_equal(th2, theta_full)
        e2e["ingest_stages_from_parsed_arrays"] = ctx.ingest_stages()
        del reads
        # the same reads in coordinate order (what an aligner's sorted output looks like): neighbouring lines fall into the same
        # bucket and the same cell.  The ingest of that file beside the shuffled one's (device time, per pass).
        # (an auxiliary leg: whatever goes wrong in it is reported under its own key and the job's own reads are put back)
        sdir = os.path.join(tmp, "sorted")
        os.makedirs(sdir, exist_ok=True)
        try:
            t_s = time.perf_counter()
            L.synth_write(L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False), sorted_reads=True), sdir, "w", write_mrf=True)
            e2e["sorted_generate_s"] = time.perf_counter() - t_s
            stext = ctx.stage_text(os.path.join(sdir, "w.mrf"))
            ctx.upload_reads_text(0, stext, free=True)
            sst = ctx.ingest_stages()
            if ctx.retained(0) != job_retained:
                raise RuntimeError("the sorted file holds other reads than the shuffled one")
            ctx.count()
            cs, bs_ = ctx.counts()
            # (read names are line numbers, so the handful of reads that tie with an event's span start may fall the other way: not compared bit for bit)
            e2e["ingest_sorted_input"] = {"device_ms": sum(x["ms"] for x in sst), "stages_ms": {x["stage"]: x["ms"] for x in sst},
                                          "device_ms_shuffled": sum(x["ms"] for x in ingest_stages),
                                          "valid_assignments": int(cs.sum()), "valid_assignments_shuffled": int(cnt_full.sum())}
            e2e["ingest_sorted_input"]["sorted_over_shuffled"] = e2e["ingest_sorted_input"]["device_ms"] / max(e2e["ingest_sorted_input"]["device_ms_shuffled"], 1e-9)
        except Exception as e_:
            e2e["ingest_sorted_input_error"] = "%s: %s" % (type(e_).__name__, e_)
        finally:
            shutil.rmtree(sdir, ignore_errors=True)
            ctx.upload_reads_text(0, ctx.stage_text(mrf), free=True)          # the job's own reads again, for the timed loop

    # ---- the records a step hands over: packed in output order; all-gathered with N > 1
    stride = max(max(ev.record_words(f, c) for f, c in bounds), 1)
    blocks = [torch.zeros(stride, dtype=torch.int64, device=dev) for _ in range(2)]
    # LSQ_BENCH_SELFTEST=1 (developer aid, one GPU): the stream / event ordering of the in-loop gather with a device copy in its place
    # (LSQ_BENCH_SELFTEST=2: the same through liblesseq_rccl's lsq_gather with a communicator of one rank)
    selftest = world == 1 and os.environ.get("LSQ_BENCH_SELFTEST") in ("1", "2")
    gathered = [torch.zeros(world * stride, dtype=torch.int64, device=("cpu" if on_host else dev)) for _ in range(2)] if (world > 1 or selftest) else None
    in_loop_gather = strong or selftest
    torch.cuda.synchronize()          # torch's zero fills run on torch's stream; the library packs into these buffers on its own
    # The gather itself: liblesseq_rccl's lsq_gather (include/lesseq_rccl.h) -- ncclAllGather on the step's result lane, right
    # behind the pack, one C call and no stream hand-over -- when its communicator comes up on every rank (rank 0's RCCL id
    # travels through torch.distributed); otherwise torch.distributed's all_gather_into_tensor on torch's stream, ordered
    # against the lane by events.  LSQ_BENCH_GATHER=torch forces the latter.
    import ctypes as C
    want_lsq = ((strong and not rehearse) or os.environ.get("LSQ_BENCH_SELFTEST") == "2") and os.environ.get("LSQ_BENCH_GATHER", "lsq") != "torch"
    if want_lsq and "rccl" not in env:
        setup_lsq_comm(env)
    rccl, comm = (env.get("rccl"), env.get("comm")) if want_lsq else (None, None)
    if comm is None:
        rccl, comm = None, C.c_void_p()
    if rccl is not None and comm.value:
        # the first gather of this job's blocks, waited for with a time limit: the collective's kernel spins until every peer has
        # joined it, so a peer that never does must not hang the run.  On expiry (or any error) on ANY rank every rank aborts its
        # communicator and the loop gathers through torch.distributed instead.
        limit = float(os.environ.get("LSQ_COLLECTIVE_TIMEOUT", "120"))
        ok = 1
        if rccl.lsq_step_gather(ctx.h, comm, blocks[0].data_ptr(), gathered[0].data_ptr(), stride) != 0:
            ok = 0
        elif L.lib.lsq_ctx_synchronize_for(ctx.h, limit) != 0:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            env["lsq_comm_error"] = "the first lsq_step_gather did not complete on every rank within %.0f s: %s" % (limit, L.lib.lsq_last_error().decode("utf-8", "replace") if not ok else "another rank")
            rccl.lsq_comm_abort(comm)
            env["rccl"], env["comm"] = None, None
            rccl, comm = None, C.c_void_p()
            ctx.synchronize()
    use_lsq_gather = rccl is not None and bool(comm.value)
    ext_streams = {}                          # the library's result streams (two lanes, taken in turn), for event ordering
    cur = torch.cuda.current_stream(dev)
    packed_ev = [torch.cuda.Event() for _ in range(2)]
    gathered_ev = [None, None]

    def step(k):
        b = k & 1
        if in_loop_gather and use_lsq_gather:
            # count + solve + pack + gather in one call.  Block b is always this lane's (lanes and blocks both alternate): the
            # gather that read it two steps ago sits ahead of this pack on the same stream
            if rccl.lsq_step_gather(ctx.h, comm, blocks[b].data_ptr(), gathered[b].data_ptr(), stride) != 0:
                raise RuntimeError("lsq_step_gather failed")
            return
        ctx.count()
        ctx.solve()
        if in_loop_gather and not on_host:
            ptr = ctx.result_stream               # this step's lane
            ext = ext_streams.get(ptr)
            if ext is None:
                ext = ext_streams[ptr] = torch.cuda.ExternalStream(ptr, device=dev)
            if gathered_ev[b] is not None:
                ext.wait_event(gathered_ev[b])        # the gather of two steps ago has read this block
        ctx.pack_results_device(blocks[b].data_ptr())
        if in_loop_gather:
            if on_host:
                ctx.synchronize()
                dist.all_gather_into_tensor(gathered[b], blocks[b].cpu())
            else:
                packed_ev[b].record(ext)
                cur.wait_event(packed_ev[b])
                if world > 1:
                    dist.all_gather_into_tensor(gathered[b], blocks[b])
                else:
                    gathered[b].copy_(blocks[b])
                if gathered_ev[b] is None:
                    gathered_ev[b] = torch.cuda.Event()
                gathered_ev[b].record(cur)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for k in range(a.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    last = (a.steps - 1) & 1

    # Kernel durations, right after the timed loop, with the library's HIP events switched on (the
    # event records of a step cost ~25 us of queue time, so the timed loop runs without them).
    # (a) as in the timed loop: steps submitted back to back, so the count kernel runs beside the
    #     tail of the previous step's EM on the second stream; the events of the last step are read.
    #     That duration of lsq_count_fast_kernel is the roofline's.
    # (b) one step at a time (synchronised): the kernels on their own.
    count_ms, solve_ms, fast_ms, alone_fast_ms = [], [], [], []
    saved = in_loop_gather
    in_loop_gather = False
    ctx.set_timing(True)
    for _ in range(min(a.steps, 10)):
        for k in range(3):
            step(k)
        ctx.synchronize()
        fast_ms.append(ctx.fast_kernel_ms())
    for _ in range(min(a.steps, 10)):
        step(0)
        ctx.synchronize()
        c, s = ctx.timing()
        count_ms.append(c)
        solve_ms.append(s)
        alone_fast_ms.append(ctx.fast_kernel_ms())
    ctx.set_timing(False)
    in_loop_gather = saved
    fence()
    # The same loop with the EM placement fixed at upload (option em_regroup off): the timed loop repeats one read set,
    # so the placement a lane learns from its earlier solve predicts the iteration counts exactly; a job whose read
    # sets change from step to step gets less out of it, at worst this number.
    ctx.set_option("em_regroup", 0)
    n_off = min(a.steps, 100)
    for k in range(10):
        step(k)
    fence()
    t_off = time.perf_counter()
    for k in range(n_off):
        step(k)
    fence()
    ms_regroup_off = (time.perf_counter() - t_off) / n_off * 1e3
    ctx.set_option("em_regroup", 1)
    # ... and what a job with changing read sets sees in practice: the placement learnt on ANOTHER batch of the same library
    # (the next n_reads reads of the same stream: same events, same expression, different reads), then the timed steps on
    # this batch before any lane refreshes its placement (every sixteenth solve)
    ms_other_batch = None
    if world == 1 and not no_e2e:
        reads_b = L.Reads.synthetic(L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], types, W.get("zipf", False), first_read=W["n_reads"]), ev)
        ctx.synchronize()
        ctx.upload_reads(0, reads_b)
        del reads_b
        for k in range(6):
            step(k)               # each lane learns from its first solve of the other batch
        fence()
        reads_a = L.Reads.synthetic(spec, ev)
        ctx.upload_reads(0, reads_a)
        del reads_a
        for k in range(2):
            step(k)               # (the first count of a read set plans the workgroups' shares: once, untimed)
        fence()
        n_ob = 22                 # a lane's placement is 4 solves old: 11 more per lane before the refresh
        t_ob = time.perf_counter()
        for k in range(n_ob):
            step(k)
        fence()
        ms_other_batch = (time.perf_counter() - t_ob) / n_ob * 1e3
    # ---- the wide-record kernel (lsq_count_fast_kernel<false, ..>: 8 bytes a block in HBM, so its algorithmic bytes ARE its resident
    # bytes): (a) this job with compact records switched off; (b) a long-read job -- reads of 1 500 bases, single blocks of 1-1.5 kb and
    # junction reads -- whose blocks do not fit compact records, so the ingest chooses wide records by itself
    wide = {}
    if world == 1 and not no_e2e:
        def kernel_and_step_ms():
            for k in range(10):
                step(k)
            fence()
            tw = time.perf_counter()
            for k in range(50):
                step(k)
            fence()
            ms = (time.perf_counter() - tw) / 50 * 1e3
            ks = []
            ctx.set_timing(True)
            for _ in range(5):
                for k in range(3):
                    step(k)
                ctx.synchronize()
                ks.append(ctx.fast_kernel_ms())
            ctx.set_timing(False)
            fence()
            return ms, float(np.mean(ks))
        saved_gather, in_loop_gather = in_loop_gather, False
        ev_keep, blocks_keep = ev, blocks
        ldir = os.path.join(tmp, "long")
        try:
            ctx.set_option("compact_pools", 0)
            ctx.upload_reads_text(0, ctx.stage_text(mrf), free=True)
            fmt_w = ctx.pool_format(0)
            ms_w, k_w = kernel_and_step_ms()
            assert not fmt_w[0]
            ev_b = sum(8 * ev.N(i) + 8 * ev.K(i) + 16 + 8 * ((1 << ev.K(i)) - 1) + 8 for i in range(n_ev))
            wide.update({"wide_ms_per_step": ms_w, "wide_count_fast_kernel_ms": k_w, "wide_resident_bytes": float(fmt_w[1] + ev_b),
                         "wide_frac": (fmt_w[1] + ev_b) / (k_w * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "wide_note": "this job with option compact_pools = 0: lsq_count_fast_kernel<false, ..> streams (start, end) pairs, 8 bytes a block -- algorithmic bytes = "
                                      "resident bytes, so wide_frac is that kernel's plain HBM fraction (resident bytes / HIP-event kernel time / 8 TB/s)"})
            ctx.set_option("compact_pools", 1)
            # (b)
            lr = dict(n_events=W["n_events"], n_reads=min(W["n_reads"], 20_000_000), R=1500)
            os.makedirs(ldir, exist_ok=True)
            lspec = L.SynthSpec(W["seed"] + 77, lr["n_events"], lr["n_reads"], lr["R"], W["n_chrom"], types, W.get("zipf", False))
            L.synth_write(lspec, ldir, "l", write_mrf=True)
            lev = L.Events(L.Annotation(os.path.join(ldir, "l.interval"), os.path.join(ldir, "l.map")), ("SHORT_READ",), (lr["R"],))
            ctx.upload_events(lev)
            ctx.upload_reads_text(0, ctx.stage_text(os.path.join(ldir, "l.mrf")), free=True)
            lst = ctx.ingest_stages()
            fmt_l = ctx.pool_format(0)
            ev = lev                     # (step() packs with the context's events)
            stride_l = max(lev.record_words(0, len(lev)), 1)
            blocks = [torch.zeros(stride_l, dtype=torch.int64, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            ms_l, k_l = kernel_and_step_ms()
            lev_b = sum(8 * lev.N(i) + 8 * lev.K(i) + 16 + 8 * ((1 << lev.K(i)) - 1) + 8 for i in range(len(lev)))
            wide.update({"long_reads_workload": "%d synthetic reads of %d bases over %d mixed events (flank exons %d bases), %d chromosomes" % (lr["n_reads"], lr["R"], lr["n_events"], lr["R"] + 1, W["n_chrom"]),
                         "long_reads_compact_records_chosen": bool(fmt_l[0]), "long_reads_pool_reads_one_two_many": list(fmt_l[2]),
                         "long_reads_retained": ctx.retained(0), "long_reads_ms_per_step": ms_l, "long_reads_count_fast_kernel_ms": k_l,
                         "long_reads_frac": (fmt_l[1] + lev_b) / (k_l * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "long_reads_ingest_device_ms": sum(x["ms"] for x in lst)})
        except Exception as e_:
            wide["wide_or_long_reads_error"] = "%s: %s" % (type(e_).__name__, e_)
        finally:
            ev, blocks = ev_keep, blocks_keep
            in_loop_gather = saved_gather
            shutil.rmtree(ldir, ignore_errors=True)
            # the job's own options, events and reads again: what follows (tables of the last step, counters) speaks of them
            ctx.set_option("compact_pools", 1)
            ctx.upload_events(ev)
            ctx.upload_reads_text(0, ctx.stage_text(mrf), free=True)
            for k in range(2):
                step(k)
            fence()
    # what the gather costs on its own (N > 1): submitted alone, timed on the host
    gather_ms = None
    if world > 1:
        ts = []
        for _ in range(5):
            fence()
            tg = time.perf_counter()
            if use_lsq_gather:
                rccl.lsq_gather(ctx.h, comm, blocks[last].data_ptr(), gathered[last].data_ptr(), stride)
                ctx.synchronize()
            else:
                dist.all_gather_into_tensor(gathered[last], blocks[last].cpu() if on_host else blocks[last])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - tg) * 1e3)
        gather_ms = min(ts[1:])

    # ---- the tables of the last step: every rank's records on every rank
    cmp_dev = "cpu" if on_host else dev
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cmp_dev)
    retained_mine, blocks_mine = ctx.retained(0), ctx.retained_blocks(0)
    pool_fmt = ctx.pool_format(0)
    launch_info = ctx.launch_info()
    per_rank = torch.tensor([float(np.mean(fast_ms)), float(np.mean(alone_fast_ms)), float(np.mean(solve_ms)), my_weight, float(n_ev_mine),
                             float(job_retained), float(job_blocks)], dtype=torch.float64, device=cmp_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        allr = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
        allr = [x.cpu().numpy() for x in allr]
    else:
        allr = [per_rank.cpu().numpy()]
    elapsed = float(tmax.item())
    if strong:
        total_retained, total_blocks, total_mrf = float(job_retained), float(job_blocks), float(W["n_reads"])
    else:
        total_retained, total_blocks, total_mrf = float(sum(x[5] for x in allr)), float(sum(x[6] for x in allr)), float(W["n_reads"] * world)

    tables_equal, max_theta_diff = None, None
    # events that took the exact-order replay in the first solve (ctx.solution() runs it; rare: none on C3, 3 of 200 000 on
    # the 1 B-read workload) carry the replayed theta in theta_full, the loop's records the kernel's: left out of the comparison
    off_iso = np.concatenate([[0], np.cumsum([ev.K(i) for i in range(n_ev)])]) if (flags_full & 4).any() else None
    keep = np.ones(len(theta_full), bool)
    if off_iso is not None:
        for i in np.nonzero(flags_full & 4)[0]:
            keep[off_iso[i]:off_iso[i + 1]] = False
    keep_ev = (flags_full & 4) == 0
    if strong:
        g = gathered[last].cpu().numpy().view(np.uint64)
        cnt, bases, theta, ll = ev.gathered_unpack(bounds, g, stride)
        tables_equal = bool(np.array_equal(cnt, cnt_full) and np.array_equal(bases, bases_full) and
                            np.array_equal(theta[keep], theta_full[keep]) and np.array_equal(ll[keep_ev], ll_full[keep_ev], equal_nan=True))
        max_theta_diff = float(np.max(np.abs(theta[keep] - theta_full[keep]))) if keep.any() else 0.0
        assert np.array_equal(cnt, cnt_full) and np.array_equal(bases, bases_full), "gathered count tables differ from the unsharded run"
        assert max_theta_diff <= 1e-12, "gathered theta differs from the unsharded run"
    else:
        blk = (gathered[last] if selftest else blocks[last]).cpu().numpy().view(np.uint64)
        cnt, bases, theta, ll = ev.gathered_unpack([(0, n_ev)], blk, stride)
        assert np.array_equal(cnt, cnt_full) and np.array_equal(bases, bases_full) and np.array_equal(theta[keep], theta_full[keep]), "a step's tables differ from the first count"
        if world > 1:         # weak mode: one all-gather of the per-event records after the loop
            dist.all_gather_into_tensor(gathered[last], blocks[last].cpu() if on_host else blocks[last])
    assert 0 < int(cnt_full.sum()) <= 4 * job_retained
    assert np.isfinite(theta_full).all() and abs(float(theta_full.sum()) - n_ev) < 1e-6 * n_ev
    exc, recounted = ctx.count_status()

    out = None
    if rank == 0:
        fk = float(np.mean(fast_ms))
        ev_bytes = 0
        lo_e, n_e = bounds[0] if strong else (0, n_ev)
        for i in range(lo_e, lo_e + n_e):
            K, N = ev.K(i), ev.N(i)
            ev_bytes += 8 * N + 8 * K + 16 + 8 * ((1 << K) - 1) + 8
        # of rank 0's launch: a rank of an event-sharded job streams the blocks of its slice's reads, not the job's
        alg_bytes = 8.0 * (ctx.pooled_blocks(0) if (world > 1 and strong) else blocks_mine) + ev_bytes
        achieved = alg_bytes / (fk * 1e-3) / 1e9
        # HBM bytes per launch of the same kernel from the committed rocprofv3 PMC passes (tools/bench_prof.sh): not measured by this run
        committed = None
        for rr in ("r03", "r02", "r01"):
            tpath = os.path.join(ROOT, "profiles", "%s_traffic_%s.json" % (rr, wl_name))
            if world == 1 and os.path.exists(tpath):
                committed = {"hbm_bytes_per_launch": json.load(open(tpath)).get("count_fast_kernel_hbm_bytes_per_launch"),
                             "source": "profiles/%s_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2 x FETCH_SIZE + WRITE_SIZE)" % (rr, wl_name)}
                break
        # ... and by this run, when it is the one-GPU default: two short child runs under the profiler's counters (after the timed loop)
        traffic_live, traffic_note, ingest_traffic = (None, "not measured by this run", None)
        if world == 1 and primary and not a.no_traffic:
            traffic_live, traffic_note, ingest_traffic = measure_traffic_live(wl_name)
        # what a plain streaming read of the same number of bytes reaches on this device (state both ceilings)
        import ctypes as C
        L.lib.lsq_debug_stream_read_rate.argtypes = [C.c_void_p, C.c_ulonglong, C.POINTER(C.c_double)]
        rate = C.c_double(0.0)
        plain_read = rate.value if L.lib.lsq_debug_stream_read_rate(ctx.h, int(max(alg_bytes, 1 << 26)), C.byref(rate)) == 0 else None
        if world == 1:
            par = "single GPU"
        elif strong:
            par = ("one job over %d GPUs: output-ordered events cut into contiguous slices of equal read weight (pre-pass count), each rank parses the whole MRF text "
                   "on its GPU and keeps the reads of its slice; per step count + EM + pack on the library's two streams and one RCCL all-gather of the packed "
                   "per-event records (%d bytes per rank) inside the timed loop, overlapped with the next step's count" % (world, stride * 8))
        else:
            par = "one job per rank (weak): events and their reads per rank, no collective in the timed loop; one RCCL all-gather of the per-event records after it"
        out = {
            "metric": "MRF reads/sec through count+solve",
            "value": total_retained * a.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "int32 coordinates / u64 counters (count), f64 (EM)",
            "data": "synthetic",
            "config": {
                "workload": W["desc"] + ("; BASELINE configs[3]: the same job sharded by event index over %d GPUs with an RCCL gather" % world if strong and wl_name == "c3" else ""),
                "events": n_ev * (1 if strong or world == 1 else world), "mrf_reads": total_mrf, "retained_reads": total_retained,
                "retained_blocks": total_blocks, "buckets_rank0": ev.num_buckets,
                "count_launch_rank0": {"one_block_reads_per_lane_and_look": launch_info[0], "workgroups_per_cu": launch_info[1]},
                "pool_records_rank0": {"compact": pool_fmt[0], "resident_bytes_of_block_coordinates": pool_fmt[1],
                                       "reads_as_one_block_two_block_many_block_records": list(pool_fmt[2]),
                                       "note": "compact: 4 bytes a block in HBM (22-bit offset from the bucket, 10-bit length); the roofline's algorithmic bytes stay "
                                               "SURVEY 8(d)'s 8 bytes a block (start, end), so `achieved` may exceed what the HBM moved -- see `traffic`"},
                "reads_counted": "`value` counts retained reads (those that pass the load-time containment filter, count/count.cpp:319) per pass over the pools resident in HBM; "
                                 "off-target reads are dropped at ingest.  The rate from MRF text on disk is e2e_cli_solve_mrf_reads_per_s / e2e_from_text_reads_per_s",
                "count_fast_kernel_ms": fk, "count_fast_kernel_ms_alone": float(np.mean(alone_fast_ms)),
                "count_stream_ms_alone": float(np.mean(count_ms)), "em_kernel_ms_alone": float(np.mean(solve_ms)),
                "pipeline": "lsq_count on one HIP stream; exception pass (which turns into a recount of every read where the exception list overflowed), EM, record "
                            "packing and counter zeroing on a second one, beside the next step's count (two counter sets)",
                "em_placement": "each step lane sorts the EM grid by the iteration counts of its own earlier solve (refreshed every 16th solve; the sort kernel "
                                "runs inside the timed loop); the loop repeats one read set, so the prediction is exact here",
                "ms_per_step_em_regroup_off_rank0": ms_regroup_off,
                "ms_per_step_em_placement_from_another_batch": ms_other_batch,
                "em_placement_from_another_batch_note": "EM placement learnt on the next n_reads reads of the same synthetic library (same events and expression, other reads), "
                                                        "then 22 timed steps on this batch before any refresh: what a loop over changing batches of one library sees",
                "valid_read_assignments": int(cnt_full.sum()), "exception_pairs": int(sum(exc)), "recounted": int(sum(recounted)),
                "em_guard_band_events": int((flags_full & 1).sum()), "em_replayed_events": int(((flags_full >> 2) & 1).sum()),
                "em_max_iters": int(iters_full.max()) if n_ev else 0,
                "generate_s": t_gen, "ingest_from_text_s": t_ingest, "first_count_solve_fetch_s": t_first - t_ingest,
                "gather_ms_alone": gather_ms,
                "n1_ms_per_step_same_box": n1_same_box,
                "speedup_vs_n1": ((n1_same_box / (1e3 * elapsed / a.steps)) if (n1_same_box and strong) else None),
                "efficiency": ((n1_same_box / (1e3 * elapsed / a.steps) / world) if (n1_same_box and strong) else None),
                "n1_same_box_note": ("20 unsharded steps (count + EM + pack, no gather) of the same job on rank 0's GPU before the events were sharded; speedup_vs_n1 = that / "
                                     "ms_per_step, efficiency = speedup / N -- informational, from this run's own clock on one box; the driver computes its own from the per-N lines" if strong else None),
                "rccl_version": env.get("rccl_version"),
                "gather_through": ("liblesseq_rccl lsq_gather (ncclAllGather on the step's result lane)" if use_lsq_gather else
                                   ("torch.distributed all_gather_into_tensor (%s)" % ("gloo, host tensors: rehearsal" if on_host else "RCCL through torch") if world > 1 else None)),
                "lsq_comm_size": (int(rccl.lsq_comm_size(comm)) if use_lsq_gather else 0),
                "lsq_comm_error": env.get("lsq_comm_error"),
                "tables_equal_unsharded_run": tables_equal, "max_abs_theta_diff_vs_unsharded": max_theta_diff,
                "count_table_sha256": hashlib.sha256(cnt_full.tobytes()).hexdigest(),
                "per_rank": [{"rank": r, "events": int(x[4]), "valid_read_assignments": x[3], "count_fast_kernel_ms": x[0],
                              "count_fast_kernel_ms_alone": x[1], "em_kernel_ms_alone": x[2]} for r, x in enumerate(allr)] if world > 1 else None,
                "parallelism": par,
            },
            "roofline": {
                "bound": "hbm", "kernel": "lsq_count_fast_kernel<%s, %d>" % ("true" if pool_fmt[0] else "false", launch_info[0] // 2) + (" (rank 0's launch)" if world > 1 else ""),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_algorithmic": achieved / HBM_PEAK_GBS,
                # what the HBM actually moved over the same kernel time: the PMC traffic when this run measured it, else the pools' resident bytes
                "frac_on_bytes_moved": ((traffic_live if traffic_live else float(pool_fmt[1] + ev_bytes)) / (fk * 1e-3) / 1e9 / HBM_PEAK_GBS),
                "frac_note": "`achieved` / `frac` count SURVEY 8(d)'s ALGORITHMIC 8 bytes per read block, as the bench contract prescribes; the compact pools hold and stream "
                             "4 bytes per block, so the HBM moves about half of that: `frac_on_bytes_moved` is the device's real bandwidth use by this kernel",
                # the same kernel on an otherwise idle device (steps synchronised one by one, no EM beside it)
                "frac_alone": alg_bytes / (float(np.mean(alone_fast_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # consecutive counts run on two streams and overlap at their ends (the next one fills the tail of this one), so a
                # launch's own duration is longer than the time the loop spends per launch: the same bytes over the step time
                "frac_at_step_rate": (alg_bytes / (1e-3 * 1e3 * elapsed / a.steps) / 1e9 / HBM_PEAK_GBS) if world == 1 else None,
                "plain_read_GBps_this_device": plain_read, "frac_of_plain_read": (achieved / plain_read if plain_read else None),
                # the same kernel time against the bytes the pools actually hold (compact records: 4 B per block) -- what the
                # HBM has to deliver at least; `achieved` above counts SURVEY 8(d)'s 8 B per block
                "resident_bytes_per_launch": float(pool_fmt[1] + ev_bytes),
                "achieved_on_resident_bytes": (pool_fmt[1] + ev_bytes) / (fk * 1e-3) / 1e9,
                "frac_on_resident_bytes": (pool_fmt[1] + ev_bytes) / (fk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic_live, "traffic_note": traffic_note,
                "traffic_over_algorithmic": (traffic_live / alg_bytes if traffic_live else None),
                "traffic_from_committed_profile": committed,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        out["config"].update(e2e)
        out["config"].update(wide)
        # ---- the loader chain on the same roofline: what a job spends once per read file (count/count.cpp:279-364 on the device)
        ing_ms = sum(x["ms"] for x in ingest_stages)
        stages = []
        for x in ingest_stages:
            tr = (ingest_traffic or {}).get(x["stage"])
            gb = x["bytes"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else None
            stages.append({"stage": x["stage"], "ms": x["ms"], "algorithmic_bytes": x["bytes"], "achieved": gb, "frac": (gb / HBM_PEAK_GBS if gb else None),
                           "traffic": tr, "traffic_over_algorithmic": (tr / x["bytes"] if tr and x["bytes"] else None)})
        out["roofline_ingest"] = {
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "what": "the passes of the loader chain over this job's MRF text in HBM (lsq_last_ingest_stages): ms = HIP events on the library's stream around the pass's "
                    "launches; algorithmic_bytes = the pass's input read once + its output written once; traffic = 2 x FETCH_SIZE + WRITE_SIZE of the pass's kernels "
                    "in the same rocprofv3 --pmc child runs that measure the count kernel (null when those did not run)",
            "text_bytes": os.path.getsize(mrf) if os.path.exists(mrf) else None, "lines": W["n_reads"], "h2d_ms": ingest_h2d_ms,
            "device_ms": ing_ms, "stages": stages, "parse_paths": ingest_paths,
            "frac_whole_chain": (sum(x["bytes"] for x in ingest_stages) / (ing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ing_ms > 0 else None,
        }
        out["config"]["one_shot_device_ms"] = ing_ms + float(np.mean(count_ms)) + float(np.mean(solve_ms))
        out["config"]["one_shot_device_note"] = ("device time of ONE fresh read set: the loader chain (parse + filter + partition + groups + pools) + one count + one EM, "
                                                 "each on an idle device; %.0f reads/s of MRF text" % (W["n_reads"] / max((ing_ms + float(np.mean(count_ms)) + float(np.mean(solve_ms))) * 1e-3, 1e-9)))
        # ---- cpu_baseline: the oracle on a bounded prefix of the same stream (rank 0, N = 1 only)
        if world == 1 and cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            ns = min(cpu_sample, W["n_reads"])
            sspec = L.SynthSpec(W["seed"], W["n_events"], ns, W["R"], W["n_chrom"], types, W.get("zipf", False))
            sdir = os.path.join(tmp, "sample")
            os.makedirs(sdir, exist_ok=True)
            L.synth_write(sspec, sdir, "s", write_mrf=True)
            argv = list(argv_solve)
            argv[12], argv[13] = os.path.join(sdir, "s.mrf"), str(ns * W["R"])
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
    ctx.synchronize()
    ctx.close()
    if world > 1:
        dist.barrier()
    if writer:
        shutil.rmtree(tmp, ignore_errors=True)
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
