"""Multi-GPU drivers for the count + solve path: one process per GPU.  run_sharded shards the events by
index (below); run_read_sharded, at the end of the file, shards the reads of one job.

Events are independent (the reference already scales out by running disjoint
gene_begin_idx..gene_end_idx slices, count/count.cpp:204-215).  Here every rank loads the whole
selected range -- so the covered regions, hence the load-time read filter, are those of the
unsharded run -- restricts its device plan to a contiguous slice of the output-ordered events
(lsq_events_set_shard: slices of equal read weight from a first unsharded count, lsq_shard_bounds),
counts and solves its slice on its GPU, packs its per-event records (class counts, class bases, theta,
log-likelihood) in output order on the device (lsq_results_pack_device), and one all-gather over the
process group -- RCCL over xGMI with backend "nccl", gloo on host copies in the tests -- gives every rank
the blocks of all; lsq_gathered_unpack lays them out as the whole job's tables.
"""
import numpy as np

from . import api


def shard_bounds(n_events, world, weights=None):
    """Contiguous slices [(first, count)] of the output-ordered events, balanced by `weights`
    (estimated reads per event) when given, else by event count."""
    if weights is None:
        weights = np.ones(n_events, np.float64)
    w = np.asarray(weights, np.float64)
    assert len(w) == n_events
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(cum, target, side="left")))
    cuts.append(n_events)
    cuts = [min(max(c, 0), n_events) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def combine(arrays, group=None, device=None):
    """all-reduce(SUM) of per-rank arrays (numpy, or torch tensors already on the communication device, which stay
    there); uint64 travels as int64 (counts < 2^63).  Used by the read-sharded run, whose exchange is a sum."""
    import torch
    import torch.distributed as dist
    out = []
    for a in arrays:
        if isinstance(a, torch.Tensor):
            dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
            out.append(a)
            continue
        a = np.ascontiguousarray(a)
        view = a.view(np.int64) if a.dtype == np.uint64 else a
        t = torch.from_numpy(view.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        r = t.cpu().numpy()
        out.append(r.view(np.uint64) if a.dtype == np.uint64 else r)
    return out


def parse_cli(tool, argv):
    per = 5 if tool == "solve" else 4
    groups = [argv[9 + i * per: 9 + (i + 1) * per] for i in range((len(argv) - 9) // per)]
    return dict(isoform_format=argv[3], isoforms=argv[4], g2i_format=argv[5], g2i=argv[6], begin=int(argv[7]), end=int(argv[8]),
                read_formats=[g[0] for g in groups], read_types=tuple(g[1] for g in groups), read_lengths=tuple(int(g[2]) for g in groups),
                read_paths=[g[3] for g in groups], total_read_bases=[float(g[4]) for g in groups] if tool == "solve" else None)


class StagedReads:
    """The read files of a job, ready to be ingested more than once (the weight pre-pass, then the shard): MRF_SINGLE text is
    copied to HBM once and parsed there by every upload; the name-keyed formats are parsed on the host once."""

    def __init__(self, ctx, ev, a):
        self.ctx, self.items = ctx, []
        try:
            for fmt, path in zip(a["read_formats"], a["read_paths"]):
                if fmt == "MRF_SINGLE":
                    self.items.append(("text", ctx.stage_text(path)))
                else:
                    self.items.append(("host", api.Reads.from_mrf(path, ev, read_format=fmt)))
        except Exception:
            self.free()
            raise

    def upload(self):
        for m, (kind, x) in enumerate(self.items):
            if kind == "text":
                self.ctx.upload_reads_text(m, x, free=False)
            else:
                self.ctx.upload_reads(m, x)

    def free(self):
        for kind, x in self.items:
            if kind == "text":
                api.lib.lsq_text_free(x)
        self.items = []


def event_weights(ev, cnt):
    """reads per output-ordered event from a count table [method][class] (a read valid for an event is in one class)"""
    off = np.asarray(ev.class_offsets(), np.int64)
    tot = np.concatenate([[0], np.cumsum(np.asarray(cnt, np.float64).sum(axis=0))])
    return tot[off[1:]] - tot[off[:-1]]


def gather_blocks(block, stride, world, group=None, comm_device=None):
    """all-gather of every rank's packed record block (a device tensor of `stride` int64 words, zero-padded behind its
    records): RCCL's ncclAllGather with backend "nccl", gloo on a host copy in the CPU tests.  Returns uint64 [world * stride]."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return block.cpu().numpy().view(np.uint64)
    if comm_device is None and dist.get_backend(group) == "gloo":
        comm_device = torch.device("cpu")
    mine = block if comm_device is None or block.device == comm_device else block.to(comm_device)
    out = torch.zeros(world * stride, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().view(np.uint64)


def run_sharded(tool, argv, rank, world, device_index=0, group=None, comm_device=None, weights="prepass"):
    """count / solve of ONE job over `world` ranks, events sharded (BASELINE configs[3]).  Every rank compiles the whole
    selected range, so the covered regions -- hence the load-time read filter -- are those of the unsharded run.  The slices
    are balanced by reads per event from a first, unsharded count (weights="prepass"; None = by event count; or an array).
    Each rank then ingests the reads against its slice only (reads that start in no event of the slice are dropped at
    ingest), counts and solves, packs its per-event records in output order on the device, and one all-gather puts the
    blocks of all ranks on every rank.  Returns the table text: byte-identical to the single-process run."""
    import torch
    a = parse_cli(tool, argv)
    ann = api.Annotation(a["isoforms"], a["g2i"], a["begin"], a["end"], a["isoform_format"], a["g2i_format"])
    ev = api.Events(ann, a["read_types"], a["read_lengths"])
    ctx = api.Context(device_index)
    staged = StagedReads(ctx, ev, a)
    try:
        if world > 1 and isinstance(weights, str) and weights == "prepass":
            ctx.upload_events(ev)
            staged.upload()
            ctx.count()
            weights = event_weights(ev, ctx.counts()[0])
        elif isinstance(weights, str):
            weights = None
        bounds = ev.shard_bounds(world, weights)
        first, count = bounds[rank]
        ev.set_shard(first, count)
        ctx.upload_events(ev)
        staged.upload()
    finally:
        staged.free()
    ctx.count()
    ctx.solve()
    ctx.solve_finalize()          # guard-band events in the reference's summation order (each rank holds its events' reads)
    stride = max(max(ev.record_words(f, c) for f, c in bounds), 1)
    dev = torch.device("cuda", device_index)
    block = torch.zeros(stride, dtype=torch.int64, device=dev)
    torch.cuda.current_stream(dev).synchronize()      # torch's fill runs on torch's stream, the pack on the library's: the fill must be done first
    ctx.pack_results_device(block.data_ptr())
    ctx.synchronize()
    blocks = gather_blocks(block, stride, world, group, comm_device)
    ctx.close()
    cnt, bases, theta, ll = ev.gathered_unpack(bounds, blocks, stride)
    if tool == "count":
        return api.format_count(ev, cnt)
    return api.format_solve(ev, cnt, bases, theta, ll, a["total_read_bases"])


# ----------------------------------------------------------------------------------------------
# Read-sharded run of ONE job: every rank takes a slice of every MRF file against all events.
# The class histograms of the slices add up (integer sums, exact), so one all-reduce of the counts
# and matched bases gives every rank the whole job's counts; the EM then runs on them.  Unlike the
# event-sharded run each rank parses and filters only its share of the text.  Read names are
# "read-<line number>" (count/count.cpp:293-295) and decide span-start ties, so every rank needs the
# file-wide number of its first line: the newline counts of the slices are exchanged first.

def slice_bounds(path, world, window=1 << 16):
    """byte cuts [b_0 = 0, ..., b_world = size]: the nominal cut size*r/world moved forward to the
    start of the next line (every rank computes the same cuts from the file alone)"""
    import os
    size = os.path.getsize(path)
    cuts = [0]
    with open(path, "rb") as f:
        for r in range(1, world):
            pos = max(size * r // world, cuts[-1])
            cut = size
            f.seek(pos)
            while pos < size:
                buf = f.read(window)
                if not buf:
                    break
                k = buf.find(b"\n")
                if k >= 0:
                    cut = pos + k + 1
                    break
                pos += len(buf)
            cuts.append(min(cut, size))
    cuts.append(size)
    return cuts


def run_read_sharded(tool, argv, rank, world, device_index=0, group=None, comm_device=None):
    """count / solve of one job over `world` ranks, reads sharded.  Returns the table text (the same
    on every rank, byte-identical to the single-process run).  MRF_SINGLE read files only: the name-keyed
    formats of `solve` group lines by read name across the whole file, which a byte slice cannot do."""
    import torch
    import torch.distributed as dist
    a = parse_cli(tool, argv)
    for fmt in a["read_formats"]:
        if fmt != "MRF_SINGLE":
            raise ValueError("the read-sharded run takes MRF_SINGLE read files only (got %s); use --shard events" % fmt)
    ann = api.Annotation(a["isoforms"], a["g2i"], a["begin"], a["end"], a["isoform_format"], a["g2i_format"])
    ev = api.Events(ann, a["read_types"], a["read_lengths"])
    ctx = api.Context(device_index)
    ctx.upload_events(ev)
    texts = []
    try:
        for path in a["read_paths"]:
            cuts = slice_bounds(path, world)
            texts.append(ctx.stage_text(path, cuts[rank], cuts[rank + 1]))
        # file-wide line numbers: a data line's number is the count of newlines before it
        mine = torch.tensor([ctx.text_lines(t) for t in texts], dtype=torch.int64)
        if comm_device is not None:
            mine = mine.to(comm_device)
        if world > 1:
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
        else:
            every = [mine]
        for m, t in enumerate(texts):
            before = sum(int(every[r][m].item()) for r in range(rank))
            ctx.upload_reads_text(m, t, has_header=(rank == 0), first_line=(1 if rank == 0 else before), free=False)
    finally:
        for t in texts:
            api.lib.lsq_text_free(t)
    ctx.count()
    if world > 1:
        # the sum over the ranks on the device: the class counts and matched bases as they lie in HBM (every rank has the same
        # events, hence the same order), one all-reduce, and the sums become the counts the EM and the getters work on
        # (gloo, i.e. the tests on one GPU: the same words through a host tensor and the library's own copies; the tensors of
        # a gloo group are host tensors, so torch.cuda is not needed there)
        n_words = max(ctx.counts_device_words(), 1)
        if dist.get_backend(group) == "gloo" or (comm_device is not None and torch.device(comm_device).type == "cpu"):
            import ctypes as C
            d_buf = C.c_void_p()
            api.check(api.lib.lsq_device_alloc(ctx.h, 8 * n_words, C.byref(d_buf)))
            try:
                ctx.export_counts_device(d_buf.value)
                host = np.zeros(n_words, dtype=np.int64)
                api.check(api.lib.lsq_device_read(ctx.h, host.ctypes.data_as(C.c_void_p), d_buf, 8 * n_words))
                h = torch.from_numpy(host)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                api.check(api.lib.lsq_device_write(ctx.h, d_buf, host.ctypes.data_as(C.c_void_p), 8 * n_words))
                ctx.import_counts_device(d_buf.value)
                ctx.synchronize()
            finally:
                api.lib.lsq_device_free(ctx.h, d_buf)
        else:
            dev = torch.device("cuda", device_index)
            t = torch.empty(n_words, dtype=torch.int64, device=dev)
            ctx.export_counts_device(t.data_ptr())
            ctx.synchronize()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize(dev)
            ctx.import_counts_device(t.data_ptr())
            ctx.synchronize()
    cnt, bases = ctx.counts()
    if tool == "count":
        text = api.format_count(ev, cnt)
    else:
        ctx.solve()
        theta, ll, iters, flags = ctx.solution()
        text = api.format_solve(ev, cnt, bases, theta, ll, a["total_read_bases"])
    ctx.close()
    return text


def main(args=None):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m lesseq_amd.dist
           [--shard reads|events] count|solve <the reference's argv>
    One process per GPU (LOCAL_RANK picks it), RCCL for the exchange; rank 0 prints the table."""
    import argparse
    import os
    import sys
    import torch
    import torch.distributed as dist
    ap = argparse.ArgumentParser(prog="lesseq_amd.dist")
    ap.add_argument("--shard", choices=("reads", "events"), default="reads")
    ap.add_argument("tool", choices=("count", "solve"))
    ap.add_argument("argv", nargs=argparse.REMAINDER)
    a = ap.parse_args(args)
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    argv = a.argv                     # the reference's arguments, without the program name
    dev = None
    if world > 1:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    try:
        run = run_read_sharded if a.shard == "reads" else run_sharded
        text = run(a.tool, argv, rank, world, device_index=local, comm_device=dev)
        if rank == 0:
            sys.stdout.write(text)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
