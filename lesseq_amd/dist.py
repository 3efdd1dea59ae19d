"""Multi-GPU drivers for the count + solve path: one process per GPU.  run_sharded shards the events by
index (below); run_read_sharded, at the end of the file, shards the reads of one job.

Events are independent (the reference already scales out by running disjoint
gene_begin_idx..gene_end_idx slices, count/count.cpp:204-215).  Here every rank loads the whole
selected range -- so the covered regions, hence the load-time read filter, are those of the
unsharded run -- restricts its device plan to a contiguous slice of the output-ordered events
(lsq_events_set_shard), counts and solves its slice on its GPU, and the fixed-stride per-event
outputs (class counts, class bases, theta, log-likelihood) are combined over the process group:
RCCL over xGMI with backend "nccl", gloo on CPU tensors in the tests.  Slices are disjoint, so the
sum over ranks of zero-padded arrays IS the concatenation; integer sums keep it exact.
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
    """all-reduce(SUM) of zero-padded per-rank arrays; uint64 travels as int64 (counts < 2^63)."""
    import torch
    import torch.distributed as dist
    out = []
    for a in arrays:
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
    return dict(isoforms=argv[4], g2i=argv[6], begin=int(argv[7]), end=int(argv[8]),
                read_types=tuple(g[1] for g in groups), read_lengths=tuple(int(g[2]) for g in groups),
                read_paths=[g[3] for g in groups], total_read_bases=[float(g[4]) for g in groups] if tool == "solve" else None)


def run_sharded(tool, argv, rank, world, device_index=0, group=None, comm_device=None, weights=None):
    """count / solve over `world` ranks.  Returns the table text on every rank (they all hold the
    combined arrays after the all-reduce); byte-identical to the single-process run."""
    a = parse_cli(tool, argv)
    ann = api.Annotation(a["isoforms"], a["g2i"], a["begin"], a["end"])
    ev = api.Events(ann, a["read_types"], a["read_lengths"])
    first, count = shard_bounds(len(ev), world, weights)[rank]
    ev.set_shard(first, count)
    ctx = api.Context(device_index)
    ctx.upload_events(ev)
    for m, path in enumerate(a["read_paths"]):
        ctx.upload_reads(m, api.Reads.from_mrf(path, ev))
    ctx.count()
    if tool == "solve":
        ctx.solve()
    cnt, bases = ctx.counts()
    parts = [cnt, bases]
    if tool == "solve":
        theta, ll, iters, flags = ctx.solution()
        parts += [theta, ll]          # events outside the slice hold zeros
    ctx.close()
    if world > 1:
        parts = combine(parts, group, comm_device)
    if tool == "count":
        return api.format_count(ev, parts[0])
    return api.format_solve(ev, parts[0], parts[1], parts[2], parts[3], a["total_read_bases"])


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
    on every rank, byte-identical to the single-process run)."""
    import torch
    import torch.distributed as dist
    a = parse_cli(tool, argv)
    ann = api.Annotation(a["isoforms"], a["g2i"], a["begin"], a["end"])
    ev = api.Events(ann, a["read_types"], a["read_lengths"])
    ctx = api.Context(device_index)
    ctx.upload_events(ev)
    texts = []
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
        ctx.upload_reads_text(m, t, has_header=(rank == 0), first_line=(1 if rank == 0 else before))
    ctx.count()
    cnt, bases = ctx.counts()
    if world > 1:
        cnt, bases = combine([cnt, bases], group, comm_device)
        ctx.set_counts(cnt, bases)
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
