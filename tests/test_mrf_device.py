"""The device MRF parser (lsq_reads_upload_mrf / lsq_mrf_parse_device) against the host parser
(lsq_mrf_parse), which the CPU suite pins to the oracle and so to the reference's loader
(count/count.cpp:279-336)."""
import os

import numpy as np
import pytest

import lesseq_amd as L
from test_oracle_golden import CASES, load_case

pytestmark = pytest.mark.gpu


def strand_names(ev, ids):
    return [L.lib.lsq_events_strand_name(ev.h, int(i)) for i in ids]


def parsed_equal(ev, host, dev):
    h, d = host.arrays(), dev.arrays()
    assert len(host) == len(dev) and host.num_blocks == dev.num_blocks
    for k in range(5):
        assert np.array_equal(h[k], d[k]), ("blk_off", "line_no", "blk_start", "blk_end", "blk_chrom")[k]
    # strand ids are handed out in order of first sight, which differs between the parsers: compare the strings
    hu, du = np.unique(h[5]), np.unique(d[5])
    hmap = {int(i): n for i, n in zip(hu, strand_names(ev, hu))}
    dmap = {int(i): n for i, n in zip(du, strand_names(ev, du))}
    assert None not in hmap.values() and None not in dmap.values()
    assert [hmap[int(i)] for i in h[5]] == [dmap[int(i)] for i in d[5]]


def setup(interval, gmap, R=50, rtype="SHORT_READ"):
    a = L.Annotation(interval, gmap, 0, 10 ** 9)
    ev = L.Events(a, (rtype,), (R,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    return ev, ctx


@pytest.mark.parametrize("name", [c for c in CASES if c != "errors"])
def test_device_parser_equals_host_parser_on_golden_inputs(name, tmp_path, monkeypatch):
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    seen = set()
    for r in c["count"]:
        argv = r["argv"]
        key = (argv[4], argv[6], argv[12])
        if key in seen or r["exit"] != 0:
            continue
        seen.add(key)
        ev, ctx = setup(argv[4], argv[6], int(argv[11]), argv[10])
        parsed_equal(ev, L.Reads.from_mrf(argv[12], ev), ctx.parse_mrf_device(argv[12]))
        ctx.close()


ANNOT = ("i1\tc1\t+\t100\t400\t2\t100,300,\t200,400,\n"
         "i2\tc1\t+\t100\t400\t1\t100,\t400,\n"
         "j1\tc2\t-\t1000\t1500\t1\t1000,\t1500,\n")
GMAP = "g\ti1\ng\ti2\nh\tj1\n"


def write_annot(tmp_path):
    (tmp_path / "a.interval").write_text(ANNOT)
    (tmp_path / "a.map").write_text(GMAP)
    return str(tmp_path / "a.interval"), str(tmp_path / "a.map")


ODD_LINES = [
    "c1:+:101:150:1:50",
    "c1:+:101:150",                                   # no query fields: the end field runs to the line's end
    "c1:+:101:150:1:50,c1:+:301:350:51:100",
    "c1:+:301:350:51:100,c1:+:101:150:1:50",          # blocks out of order
    "# a comment line takes a line number",
    "AlignmentBlocks",
    "AlignmentBlocks ",                               # not the literal: a cast failure?  no -- see below
    "zz:+:5:9:1:5",                                   # chromosome no event knows
    "c1:+:5000000000:5000000010:1:10",                # beyond int32: can never be contained
    "c1:+:-5:10:1:10",
    "c1:+:+120:+130:1:10",
    "c1::120:130:1:10",                               # empty strand
    "c1:strand7:120:130:1:10",                        # 7-byte strand
    "c2:-:1001:1050:1:50\tACGT\tIIII",
    "c1:.:101:150:1:50,c2:*:1001:1050:1:50",          # chromosome / strand of the last kept block
    "c1:+:101:150:1:50,",                             # trailing comma: a second, field-less block?  see below
]


def good_odd_lines():
    # lines 7 and 16 of ODD_LINES fail the cast in the reference; they belong to the error test
    return [l for i, l in enumerate(ODD_LINES) if i not in (6, 15)]


def test_odd_lines_and_unterminated_tail(tmp_path):
    iv, mp = write_annot(tmp_path)
    p = tmp_path / "odd.mrf"
    p.write_text("AlignmentBlocks\n" + "\n".join(good_odd_lines()) + "\n" + "c1:+:101:150:1:50")   # last line has no newline
    ev, ctx = setup(iv, mp)
    host, dev = L.Reads.from_mrf(str(p), ev), ctx.parse_mrf_device(str(p))
    assert len(host) == len(good_odd_lines()) - 2      # comment + AlignmentBlocks skipped, tail never seen
    parsed_equal(ev, host, dev)
    # and through the whole path
    ctx.upload_reads(0, host)
    ctx.count()
    a = ctx.counts()
    kept = ctx.retained(0)
    ctx.upload_reads_mrf(0, str(p))
    ctx.count()
    b = ctx.counts()
    assert ctx.retained(0) == kept > 0
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ctx.close()


@pytest.mark.parametrize("bad", [ODD_LINES[6], ODD_LINES[15], "", "c1:+:10x:20:1:10", "c1:+:10:20\r", "c1:+:99999999999999999999:5:1:1", "c1:+"])
def test_first_failing_line_is_reported_like_the_host_parser(bad, tmp_path):
    iv, mp = write_annot(tmp_path)
    lines = ["c1:+:101:150:1:50"] * 700
    lines[300] = bad
    lines[650] = "c1:+:x:150:1:50"
    p = tmp_path / "bad.mrf"
    p.write_text("AlignmentBlocks\n" + "\n".join(lines) + "\n")
    ev, ctx = setup(iv, mp)
    with pytest.raises(L.LsqError) as eh:
        L.Reads.from_mrf(str(p), ev)
    with pytest.raises(L.LsqError) as ed:
        ctx.parse_mrf_device(str(p))
    assert eh.value.status == ed.value.status == -4      # LSQ_E_PARSE
    assert str(eh.value) == str(ed.value)
    assert "#301:" in str(ed.value)
    with pytest.raises(L.LsqError):
        ctx.upload_reads_mrf(0, str(p))
    ctx.close()


def test_format_io_and_long_strand(tmp_path):
    iv, mp = write_annot(tmp_path)
    ev, ctx = setup(iv, mp)
    p = tmp_path / "x.mrf"
    p.write_text("AlignmentBlocks\nc1:+:101:150:1:50\n")
    with pytest.raises(L.LsqError, match="Unknown file format"):
        ctx.upload_reads_mrf(0, str(p), read_format="MRF_PAIRED")
    with pytest.raises(L.LsqError, match="cannot open"):
        ctx.upload_reads_mrf(0, str(tmp_path / "absent.mrf"))
    p.write_text("AlignmentBlocks\nc1:eightchr:101:150:1:50\n")
    with pytest.raises(L.LsqError, match="7 bytes"):
        ctx.upload_reads_mrf(0, str(p))
    # the command line takes such a file through the host parser and still prints the reference's rows
    argv = ["0", "x", "./", "LH_GENE_TXT", iv, "UCSC_GENE2ISOFORM", mp, "0", "10", "MRF_SINGLE", "SHORT_READ", "50", str(p)]
    rc, text = L.cli_run("count", argv)
    assert rc == 0 and text.startswith("g\t")
    ctx.close()


@pytest.mark.parametrize("text", ["", "AlignmentBlocks", "AlignmentBlocks\n", "\n", "AlignmentBlocks\nc1:+:101:150:1:50"])
def test_files_without_data_lines(text, tmp_path):
    iv, mp = write_annot(tmp_path)
    ev, ctx = setup(iv, mp)
    p = tmp_path / "e.mrf"
    p.write_text(text)
    dev = ctx.parse_mrf_device(str(p))
    assert len(dev) == 0 and dev.num_blocks == 0 and len(L.Reads.from_mrf(str(p), ev)) == 0
    ctx.upload_reads_mrf(0, str(p))
    ctx.count()
    assert ctx.retained(0) == 0 and int(ctx.counts()[0].sum()) == 0
    ctx.close()


def test_two_million_lines_tile_boundaries_and_results(tmp_path, monkeypatch):
    spec = L.SynthSpec(21, 3000, 2_000_000, 100, 6, L.EVENT_TYPES, False, 0.10)
    L.synth_write(spec, str(tmp_path), "s")
    ev, ctx = setup(str(tmp_path / "s.interval"), str(tmp_path / "s.map"), 100)
    mrf = str(tmp_path / "s.mrf")
    host = L.Reads.from_mrf(mrf, ev)
    dev = ctx.parse_mrf_device(mrf)
    assert len(host) == 2_000_000
    parsed_equal(ev, host, dev)
    ctx.upload_reads(0, host)
    ctx.count(); ctx.solve()
    a, s1 = ctx.counts(), ctx.solution()
    ctx.upload_reads_mrf(0, mrf)
    assert parse_paths(ctx) == (0, 0, 0)               # a file of reads: every line settled by the fast kernel
    ctx.count(); ctx.solve()
    b, s2 = ctx.counts(), ctx.solution()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1], equal_nan=True)
    t = ctx.mrf_timing()
    assert t["parse_ms"] > 0
    # the copy path of gigabyte files (two pinned buffers filled by host threads), forced on this 65 MB one
    monkeypatch.setenv("LSQ_PINNED_COPY_MIN", "1")
    parsed_equal(ev, host, ctx.parse_mrf_device(mrf))
    monkeypatch.delenv("LSQ_PINNED_COPY_MIN")
    print("device parse of %d MB: h2d %.1f ms, kernels %.1f ms" % (os.path.getsize(mrf) >> 20, t["h2d_ms"], t["parse_ms"]))
    ctx.close()


# ---- the three ways a line gets parsed on the way into the pools (round 4): the fast kernel's delimiter tables, the line list
# for lines of another shape than a read's, the byte-walking kernel for tiles (or files) the fast kernel does not take

def parse_paths(ctx):
    import ctypes as C
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    L.lib.lsq_debug_last_parse_paths.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    assert L.lib.lsq_debug_last_parse_paths(ctx.h, C.byref(a), C.byref(b), C.byref(c)) == 0
    return a.value, b.value, c.value


def pools_and_counts(ctx, path=None, reads=None):
    if reads is not None:
        ctx.upload_reads(0, reads)
    else:
        ctx.upload_reads_mrf(0, path)
    ctx.count()
    c = ctx.counts()
    return ctx.retained(0), ctx.retained_blocks(0), ctx.pooled(0), c[0].copy(), c[1].copy()


def same(a, b):
    return a[:3] == b[:3] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


def random_valid_lines(rng, n):
    """lines the reference accepts, in every shape its find / substr walk makes sense of: coordinates of 1-9 digits with leading
    zeros and signs, ten and more digits, blocks without query fields, trailing text, unknown and long chromosome names,
    strands of 0-7 bytes, comments; blocks in and around the exons of ANNOT"""
    out = []
    for _ in range(n):
        r = rng.random()
        if r < 0.03:
            out.append("#" + ":" * int(rng.integers(0, 9)) + "," * int(rng.integers(0, 3)) + "x")
            continue
        blocks = []
        for _b in range(int(rng.choice([1, 1, 1, 2, 2, 3, 5]))):
            chrom = rng.choice(["c1", "c1", "c1", "c2", "c2", "zz", "", "chromosome_with_a_long_name", "c1x4567", "c12345678"])
            strand = rng.choice(["+", "+", "-", "-", ".", "*", "", "ab", "strand7"])
            if chrom == "c2":
                a = int(rng.integers(990, 1500)); b = a + int(rng.integers(0, 60))
            else:
                a = int(rng.integers(90, 400)); b = a + int(rng.integers(0, 60))
            if rng.random() < 0.05:
                a = int(rng.choice([0, 1, 9, 99999999, 100000000, 999999999, 1000000000, 2147483647, 2147483648, 5000000000]))
                b = a + int(rng.integers(0, 50))

            def fmt(v):
                q = rng.random()
                if q < 0.70:
                    return str(v)
                if q < 0.80:
                    return "+" + str(v)
                if q < 0.95:
                    return str(v).zfill(int(rng.integers(1, 12)))
                return "-" + str(v) if v and rng.random() < 0.5 else "-0"
            blk = "%s:%s:%s:%s" % (chrom, strand, fmt(a), fmt(b))
            blocks.append(blk)
        # query fields: on every block but possibly the last (a block without them swallows what follows into its end field)
        line = ",".join(b + ":1:%d" % int(rng.integers(1, 100)) for b in blocks[:-1])
        last = blocks[-1]
        q = rng.random()
        if q < 0.75:
            last += ":1:50"
        elif q < 0.85:
            last += ":1:50\tACGT\tIIII"
        elif q < 0.90:
            last += ":"
        line = (line + "," if line else "") + last
        out.append(line)
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fast_kernel_line_list_and_byte_walking_kernel_agree_with_the_host_parser(seed, tmp_path, monkeypatch):
    iv, mp = write_annot(tmp_path)
    rng = np.random.default_rng(seed)
    lines = random_valid_lines(rng, 30000) + good_odd_lines()
    p = tmp_path / "r.mrf"
    p.write_text("AlignmentBlocks\n" + "\n".join(lines) + "\n")
    ev, ctx = setup(iv, mp)
    want = pools_and_counts(ctx, reads=L.Reads.from_mrf(str(p), ev))
    assert want[0] > 5000 and int(want[3].sum()) > 1000
    fast = pools_and_counts(ctx, path=str(p))
    assert same(want, fast)
    tiles, listed, slow = parse_paths(ctx)
    assert slow == 0 and 1000 < listed < 25000          # the odd lines went to the shared splitter, the others did not
    monkeypatch.setenv("LSQ_MRF_LINE_LIST", "50")          # the list runs over: the file goes through the byte-walking kernel
    assert same(want, pools_and_counts(ctx, path=str(p)))
    assert parse_paths(ctx)[2] == 1
    monkeypatch.delenv("LSQ_MRF_LINE_LIST")
    monkeypatch.setenv("LSQ_MRF_SLOW", "1")                # ... which this asks for outright
    assert same(want, pools_and_counts(ctx, path=str(p)))
    ctx.close()


def test_windows_with_more_delimiters_than_the_tables_hold(tmp_path):
    """comment lines of thousands of colons and commas (longer than the 512 bytes a tile sees ahead of itself, too) between
    reads: their tiles go to the byte-walking kernel, the reads around them keep their line numbers"""
    iv, mp = write_annot(tmp_path)
    rng = np.random.default_rng(5)
    lines = []
    for k in range(4000):
        if k % 97 == 13:
            lines.append("#" + "".join(rng.choice([":", ",", "x"], size=int(rng.integers(3000, 9000)))))
        else:
            a = int(rng.integers(100, 350))
            lines.append("c1:+:%d:%d:1:50" % (a + 1, a + 50))
    p = tmp_path / "d.mrf"
    p.write_text("AlignmentBlocks\n" + "\n".join(lines) + "\n")
    ev, ctx = setup(iv, mp)
    want = pools_and_counts(ctx, reads=L.Reads.from_mrf(str(p), ev))
    assert want[0] > 3000
    assert same(want, pools_and_counts(ctx, path=str(p)))
    tiles, listed, slow = parse_paths(ctx)
    assert tiles > 10 and slow == 0                    # the comment lines' tiles went to the byte-walking kernel
    parsed_equal(ev, L.Reads.from_mrf(str(p), ev), ctx.parse_mrf_device(str(p)))
    ctx.close()


@pytest.mark.parametrize("bad", ["c1:+:10x:20:1:10", "c1:+:99999999999999999999:5:1:1", "c1:+:1:5,c1:+:7:9", "", "c1:+"])
def test_first_failing_line_through_the_ingest_path(bad, tmp_path):
    """the same verdict and the same line number whether the fast kernel meets the line or the list does"""
    iv, mp = write_annot(tmp_path)
    lines = ["c1:+:101:150:1:50"] * 900
    lines[300] = bad
    lines[650] = "c1:+:x:150:1:50"
    p = tmp_path / "bad.mrf"
    p.write_text("AlignmentBlocks\n" + "\n".join(lines) + "\n")
    ev, ctx = setup(iv, mp)
    with pytest.raises(L.LsqError) as eh:
        L.Reads.from_mrf(str(p), ev)
    with pytest.raises(L.LsqError) as ed:
        ctx.upload_reads_mrf(0, str(p))
    assert eh.value.status == ed.value.status == -4 and str(eh.value) == str(ed.value) and "#301:" in str(ed.value)
    ctx.close()


def test_more_chromosomes_than_the_fast_kernel_takes_and_long_names(tmp_path, monkeypatch):
    """80 chromosomes (the fast kernel's tables hold 64: the whole file goes through the byte-walking kernel), and an annotation whose
    chromosome names are longer than the seven bytes the fast kernel keys on (every line on them goes to the line list; with a short
    list the file is routed again by the byte-walking kernel): the tables of the host-parsed path every time"""
    spec = L.SynthSpec(33, 1200, 150000, 100, 80, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "m")
    ev, ctx = setup(str(tmp_path / "m.interval"), str(tmp_path / "m.map"), 100)
    want = pools_and_counts(ctx, reads=L.Reads.from_mrf(str(tmp_path / "m.mrf"), ev))
    assert want[0] > 100000 and int(want[3].sum()) > 50000
    assert same(want, pools_and_counts(ctx, path=str(tmp_path / "m.mrf")))
    ctx.close()
    # long names: chr<N> -> chromosome_number_<N> in all three files
    spec = L.SynthSpec(34, 600, 80000, 100, 5, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "n")
    import re
    for f, pat in (("n.interval", r"\tchr(\d+)\t"), ("n.mrf", r"(?m)(?:^|,)chr(\d+):")):
        text = open(tmp_path / f).read()
        if f == "n.interval":
            text = re.sub(pat, lambda m: "\tchromosome_number_%s\t" % m.group(1), text)
        else:
            text = re.sub(r"chr(\d+):", lambda m: "chromosome_number_%s:" % m.group(1), text)
        open(tmp_path / f, "w").write(text)
    ev, ctx = setup(str(tmp_path / "n.interval"), str(tmp_path / "n.map"), 100)
    want = pools_and_counts(ctx, reads=L.Reads.from_mrf(str(tmp_path / "n.mrf"), ev))
    assert want[0] > 50000 and int(want[3].sum()) > 30000
    assert same(want, pools_and_counts(ctx, path=str(tmp_path / "n.mrf")))          # every line through the list
    monkeypatch.setenv("LSQ_MRF_LINE_LIST", "1000")
    assert same(want, pools_and_counts(ctx, path=str(tmp_path / "n.mrf")))          # the list runs over
    ctx.close()


def test_partition_without_lds_counters_and_a_many_block_list_that_runs_over(tmp_path, monkeypatch):
    """the partition's form for very many buckets (counters in global memory), and a list of many-block reads that is sized again
    after it ran over, on inputs the other forms have settled: the same tables"""
    spec = L.SynthSpec(35, 2000, 300000, 100, 4, L.EVENT_TYPES, True)
    L.synth_write(spec, str(tmp_path), "p")
    ev, ctx = setup(str(tmp_path / "p.interval"), str(tmp_path / "p.map"), 100)
    want = pools_and_counts(ctx, path=str(tmp_path / "p.mrf"))
    monkeypatch.setenv("LSQ_PART_NO_LDS", "1")
    assert same(want, pools_and_counts(ctx, path=str(tmp_path / "p.mrf")))
    assert same(want, pools_and_counts(ctx, reads=L.Reads.from_mrf(str(tmp_path / "p.mrf"), ev)))
    monkeypatch.delenv("LSQ_PART_NO_LDS")
    ctx.close()
    # long reads: many-block reads and compact misfits in numbers, the list sized for sixteen of them
    spec = L.SynthSpec(36, 300, 60000, 1500, 3, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "w")
    iv, mp, mrf, R = str(tmp_path / "w.interval"), str(tmp_path / "w.map"), str(tmp_path / "w.mrf"), 1500
    ev, ctx = setup(iv, mp, R)
    ctx.set_option("compact_pools", 1)
    want = pools_and_counts(ctx, path=mrf)
    monkeypatch.setenv("LSQ_NB_LIST", "16")
    assert same(want, pools_and_counts(ctx, path=mrf))
    assert same(want, pools_and_counts(ctx, reads=L.Reads.from_mrf(mrf, ev)))
    ctx.close()


def test_ingest_stage_report(tmp_path):
    """lsq_last_ingest_stages: the loader's seven passes by name, each with device time and the bytes it has to move, for the text path
    and (without the newline count) for parsed arrays"""
    spec = L.SynthSpec(41, 800, 400000, 100, 3, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "s")
    ev, ctx = setup(str(tmp_path / "s.interval"), str(tmp_path / "s.map"), 100)
    ctx.upload_reads_mrf(0, str(tmp_path / "s.mrf"))
    st = ctx.ingest_stages()
    assert [s["stage"] for s in st] == ["newline_count", "route", "partition_count", "partition_scatter", "group_classify", "group_offsets", "group_place"]
    size = os.path.getsize(tmp_path / "s.mrf")
    assert all(s["ms"] > 0 for s in st) and st[0]["bytes"] >= size and st[1]["bytes"] >= size + 20 * 400000
    assert sum(s["ms"] for s in st) < 50
    ctx.upload_reads(0, L.Reads.from_mrf(str(tmp_path / "s.mrf"), ev))
    st = ctx.ingest_stages()
    assert st[0]["ms"] == 0 and st[0]["bytes"] == 0 and all(s["ms"] > 0 for s in st[1:])
    ctx.close()
