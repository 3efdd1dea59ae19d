"""The N > 1 path on CPU: slicing of the output-ordered events and the combination of per-rank,
zero-padded per-event outputs over a gloo process group of two ranks (the same code runs over
RCCL with backend "nccl" on GPUs).  The device work itself is covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lesseq_amd as L
from lesseq_amd import dist as ld
from test_oracle_golden import GOLD


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_every_event_once():
    for n, w in ((10, 3), (1, 4), (0, 2), (50000, 8), (7, 7)):
        b = ld.shard_bounds(n, w)
        assert len(b) == w and b[0][0] == 0 and sum(c for _, c in b) == n
        for (f0, c0), (f1, _) in zip(b, b[1:]):
            assert f0 + c0 == f1
    # balanced by weight: one heavy event gets a slice of its own
    wts = np.ones(100)
    wts[10] = 1000.0
    b = ld.shard_bounds(100, 4, wts)
    sums = [wts[f:f + c].sum() for f, c in b]
    assert max(sums) <= 1000.0 + 50


def _worker(rank, world, port, iv, mp_path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ann = L.Annotation(iv, mp_path)
        ev = L.Events(ann, ("SHORT_READ",), (100,))
        n_ev = len(ev)
        off = ev.class_offsets()
        n_cls, n_iso = off[-1], ev.total_isoforms
        # what a rank's GPU would hand back: values only inside its slice of the events
        first, count = ld.shard_bounds(n_ev, world)[rank]
        full_cnt = (np.arange(n_cls, dtype=np.uint64) * 2654435761 % 1000003).reshape(1, -1)
        full_bases = full_cnt * np.uint64(97)
        full_theta = (np.arange(n_iso) % 7 + 1) / 8.0
        full_ll = -np.arange(n_ev, dtype=np.float64) - 1.5
        io = np.concatenate([[0], np.cumsum([ev.K(i) for i in range(n_ev)])])
        cnt, bases = np.zeros_like(full_cnt), np.zeros_like(full_bases)
        theta, ll = np.zeros_like(full_theta), np.zeros_like(full_ll)
        lo_c, hi_c = off[first], off[first + count]
        cnt[:, lo_c:hi_c] = full_cnt[:, lo_c:hi_c]
        bases[:, lo_c:hi_c] = full_bases[:, lo_c:hi_c]
        theta[io[first]:io[first + count]] = full_theta[io[first]:io[first + count]]
        ll[first:first + count] = full_ll[first:first + count]
        g_cnt, g_bases, g_theta, g_ll = ld.combine([cnt, bases, theta, ll])
        ok = (np.array_equal(g_cnt, full_cnt) and np.array_equal(g_bases, full_bases)
              and np.array_equal(g_theta, full_theta) and np.array_equal(g_ll, full_ll))
        # the event-sharded run's exchange: every rank's packed record block (what lsq_results_pack_device writes:
        # counts, bases, theta, log-likelihood of its slice in output order), all-gathered, then lsq_gathered_unpack;
        # slices weighted as a read-depth pre-pass would weight them (a few heavy events)
        wts = np.ones(n_ev)
        wts[::17] = 500.0
        bounds = ev.shard_bounds(world, wts)
        assert bounds == ld.shard_bounds(n_ev, world, wts)
        f2, c2 = bounds[rank]
        stride = max(ev.record_words(f, c) for f, c in bounds)
        blk = np.concatenate([full_cnt[0, off[f2]:off[f2 + c2]], full_bases[0, off[f2]:off[f2 + c2]],
                              full_theta[io[f2]:io[f2 + c2]].view(np.uint64), full_ll[f2:f2 + c2].view(np.uint64)])
        assert len(blk) == ev.record_words(f2, c2)
        block = torch.zeros(stride, dtype=torch.int64)
        block[:len(blk)] = torch.from_numpy(blk.view(np.int64).copy())
        blocks = ld.gather_blocks(block, stride, world)
        u_cnt, u_bases, u_theta, u_ll = ev.gathered_unpack(bounds, blocks, stride)
        ok = ok and (np.array_equal(u_cnt, full_cnt) and np.array_equal(u_bases, full_bases)
                     and np.array_equal(u_theta, full_theta) and np.array_equal(u_ll, full_ll))
        sums = [wts[f:f + c].sum() for f, c in bounds]
        ok = ok and max(sums) <= wts.sum() / world + 500.0
        text = L.format_solve(ev, g_cnt, g_bases, g_theta, g_ll, [1e6])
        ref = L.format_solve(ev, full_cnt, full_bases, full_theta, full_ll, [1e6])
        q.put((rank, ok and text == ref and L.format_count(ev, g_cnt) == L.format_count(ev, full_cnt)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_combine_and_format():
    d = os.path.join(GOLD, "events_s1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, os.path.join(d, "ev.interval"), os.path.join(d, "ev.map"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _gpu_worker(rank, world, port, argv_count, argv_solve, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = ld.run_sharded("count", argv_count, rank, world, device_index=0)
        s = ld.run_sharded("solve", argv_solve, rank, world, device_index=0)
        q.put((rank, c, s))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("zipf", [False, True], ids=["even_depth", "zipf_depth"])
def test_two_ranks_sharded_run_equals_single_process(zipf, tmp_path):
    """two ranks (sharing the one GPU of the test box, gloo for the exchange) give the byte-identical
    count table and the identical solve table of the unsharded run: MRF text parsed on the device by each
    rank, slices weighted by reads per event from a first unsharded count (with Zipf depth the cut is far
    from the middle of the event list), records packed on the device, all-gathered, unpacked"""
    spec = L.SynthSpec(31, 300, 60000, 100, 3, L.EVENT_TYPES, zipf)
    L.synth_write(spec, str(tmp_path), "d")
    base = ["0", "d", "./", "LH_GENE_TXT", str(tmp_path / "d.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "d.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", "100", str(tmp_path / "d.mrf")]
    rc, single_count = L.cli_run("count", base)
    rc2, single_solve = L.cli_run("solve", base + ["6000000"])
    assert rc == rc2 == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, base, base + ["6000000"], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, c, s in res:
        assert c == single_count
        assert s == single_solve


def test_slice_bounds_cut_on_line_starts(tmp_path):
    p = tmp_path / "t.mrf"
    body = b"AlignmentBlocks\n" + b"".join(b"chr1:+:%d:%d:1:50\n" % (i * 7 + 1, i * 7 + 50) for i in range(5000)) + b"tail without newline"
    p.write_bytes(body)
    for world in (1, 2, 3, 8, 64):
        cuts = ld.slice_bounds(str(p), world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == len(body)
        assert cuts == sorted(cuts)
        assert all(c == len(body) or body[c - 1:c] == b"\n" for c in cuts[1:-1])
    tiny = tmp_path / "tiny.mrf"
    tiny.write_bytes(b"x\n")
    assert ld.slice_bounds(str(tiny), 4) == [0, 2, 2, 2, 2]


def _gpu_read_worker(rank, world, port, argv_count, argv_solve, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = ld.run_read_sharded("count", argv_count, rank, world, device_index=0)
        s = ld.run_read_sharded("solve", argv_solve, rank, world, device_index=0)
        q.put((rank, c, s))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_three_ranks_read_sharded_run_equals_single_process(tmp_path):
    """three ranks each parse a third of the MRF text (file-wide line numbers from the exchanged newline
    counts: the adversarial reads of the generator include span-start ties decided by the read name),
    count it against all events, and all-reduce the class histograms: byte-identical tables"""
    spec = L.SynthSpec(37, 300, 90000, 100, 3, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "d")
    base = ["0", "d", "./", "LH_GENE_TXT", str(tmp_path / "d.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "d.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", "100", str(tmp_path / "d.mrf")]
    rc, single_count = L.cli_run("count", base)
    rc2, single_solve = L.cli_run("solve", base + ["9000000"])
    assert rc == rc2 == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gpu_read_worker, args=(r, 3, port, base, base + ["9000000"], q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, c, s in res:
        assert c == single_count
        assert s == single_solve


@pytest.mark.gpu
def test_a_shard_pools_only_the_reads_of_its_events(tmp_path):
    """lsq_events_set_shard: the load-time filter stays that of the whole range (same retained count), but only reads that
    start in the span of one of the slice's events are kept in the pools -- a half-job shard holds about half of them, the two
    halves together every read once (plus the few that start in overlapping events of both halves), and the buckets of a
    sparse slice stay short on the chromosome"""
    spec = L.SynthSpec(41, 2000, 400000, 100, 4, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "d", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "d.interval"), str(tmp_path / "d.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    reads = L.Reads.synthetic(spec, ev)
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, reads)
    full_retained, full_pooled, full_buckets = ctx.retained(0), ctx.pooled(0), ev.num_buckets
    assert full_pooled == full_retained > 300000
    ctx.count()
    cnt_full = ctx.counts()[0].copy()
    bounds = ev.shard_bounds(2, ld.event_weights(ev, cnt_full))
    pooled, total = [], np.zeros_like(cnt_full)
    for f, c in bounds:
        ev.set_shard(f, c)
        ctx.upload_events(ev)
        ctx.upload_reads(0, reads)
        assert ctx.retained(0) == full_retained
        pooled.append(ctx.pooled(0))
        assert ev.num_buckets < 0.75 * full_buckets
        ctx.count()
        total += ctx.counts()[0]
    ctx.close()
    assert all(0.4 * full_pooled < p < 0.6 * full_pooled for p in pooled), (pooled, full_pooled)
    assert full_pooled <= sum(pooled) < 1.1 * full_pooled
    assert np.array_equal(total, cnt_full)
