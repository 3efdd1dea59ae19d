"""Host-side logic of the product library on CPU: the C ABI loads and exports what the header
declares; annotation loading, event compilation (segments, masks, ARS), MRF parsing, classify
and the exit statuses that are decided before a GPU is touched.  No device calls here."""
import json
import os
import re

import pytest

import numpy as np
import lesseq_amd as L
import oracle_binding as ob
from test_oracle_golden import GOLD, CASES, load_case



def test_library_exports_every_declared_symbol():
    """every entry point include/*.h declares is exported: lesseq_hip.h / lesseq_hip_dev.h by liblesseq_hip.so,
    lesseq_rccl.h by liblesseq_rccl.so (the one library that links librccl); no compute call is made"""
    import ctypes
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(L._lib.LIB_PATH), "liblesseq_rccl.so"))
    libs = {"lesseq_hip.h": L.lib, "lesseq_hip_dev.h": L.lib, "lesseq_rccl.h": rccl}
    total = 0
    for h in sorted(os.listdir(inc)):
        if not h.endswith(".h"):
            continue
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, h)).read(), flags=re.S)        # prose in comments names functions of other libraries
        names = set(re.findall(r"\b(lsq_[a-z0-9_]+)\s*\(", text))
        assert names, h
        for n in sorted(names):
            assert hasattr(libs[h], n), "%s: missing export %s" % (h, n)
        total += len(names)
    assert total > 60
    assert L.lib.lsq_abi_version() == 2


def test_device_calls_fail_loudly_without_a_gpu():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(L.LsqError) as ei:
        L.Context(0)
    assert ei.value.status == -7


def parse_interval(path):
    iso = {}
    for line in open(path).read().split("\n")[:-1]:
        t = line.split()
        n = int(t[5])
        s = [int(x) for x in t[6].split(",") if x][:n]
        e = [int(x) for x in t[7].split(",") if x][:n]
        iso[t[0]] = list(zip(s, e))      # last duplicate wins
    return iso


def parse_map(path):
    genes = {}
    for line in open(path).read().split("\n")[:-1]:
        g, i = line.split()
        genes.setdefault(g, []).append(i)
    return genes


@pytest.mark.parametrize("name", [c for c in CASES if c not in ("errors", "fmt1m", "formats")])
def test_event_compilation_matches_oracle(name, tmp_path):
    c, d = load_case(name, tmp_path)
    argv = c["count"][0]["argv"]
    iv, mp = os.path.join(d, argv[4]), os.path.join(d, argv[6])
    R = int(argv[11])
    rtype = argv[10]
    a = L.Annotation(iv, mp, int(argv[7]), int(argv[8]))
    ev = L.Events(a, (rtype,), (R,))
    iso, genes = parse_interval(iv), parse_map(mp)
    gsel = sorted(genes, key=lambda s: s.encode())[int(argv[7]):int(argv[8])]
    assert [ev.gene_name(i) for i in range(len(ev))] == gsel
    for i, g in enumerate(gsel):
        exons = [x for nm in genes[g] for x in iso[nm]]
        segs = ob.segments(exons)
        assert ev.segments(i) == segs, (name, g)
        merged = ob.merge_intervals(exons)
        assert ev.span(i) == (merged[0][0], merged[-1][1])
        seg_len = [e - s for s, e in segs]
        for j, nm in enumerate(genes[g]):
            assert ev.isoform_name(i, j) == nm
            mask = ev.isoform_mask(i, j)
            idx = [n for n in range(len(segs)) if mask >> n & 1]
            assert ev.ars(0, i, j) == ob.ars_total(seg_len, idx, R, rtype == "SHORT_READ"), (name, g, nm)
            assert ev.isoform_length(i, j) == sum(seg_len[n] for n in idx)


def _event_tables(ev):
    out = []
    for i in range(len(ev)):
        K = ev.K(i)
        out.append((ev.gene_name(i), ev.chrom(i), ev.strand(i), ev.segments(i), ev.span(i),
                    [(ev.isoform_name(i, j), ev.isoform_mask(i, j), ev.ars(0, i, j)) for j in range(K)]))
    return out


def test_solve_annotation_formats_compile_alike(tmp_path):
    """the formats only `solve` reads (solve/solve.cpp:158-329): the column formats give the events of
    LH_GENE_TXT; the exon-per-line formats give each other's, with every isoform's exons merged by
    interval_list::add_interval in file order"""
    c, d = load_case("formats", tmp_path)
    p = lambda f: os.path.join(d, f)
    def tables(ifmt, ipath, gfmt, gpath):
        a = L.Annotation(p(ipath), p(gpath), 0, 1000, ifmt, gfmt)
        return _event_tables(L.Events(a, ("SHORT_READ",), (60,)))
    lh = tables("LH_GENE_TXT", "f.interval", "UCSC_GENE2ISOFORM", "f.map")
    assert len(lh) == 30
    assert tables("UCSC_GENE_TXT", "f.ucsc.txt", "UCSC_GENE2ISOFORM", "f.map") == lh
    assert tables("LH_GENE_TXT", "f.interval", "WORMBASE_GENE2ISOFORMS", "f.worm.map") == lh
    gff = tables("UCSC_GFF", "f.gff", "UCSC_GENE2ISOFORM", "f.map")
    assert tables("WORMBASE_GFF2", "f.worm.gff2", "WORMBASE_GENE2ISOFORMS", "f.worm.map") == gff
    assert tables("GENELETS_GFF3", "f.genelets.gff3", "UCSC_GENE2ISOFORM", "f.map") == gff
    # expected from the file itself: exons of a name in file order through the oracle's interval list
    per = {}
    for line in open(p("f.gff")).read().split("\n")[2:-1]:
        t = line.split("\t")
        per.setdefault(t[8].strip('"'), []).append((int(t[3]) - 1, int(t[4])))
    genes = parse_map(p("f.map"))
    for (gname, chrom, strand, segs, span, isos) in gff:
        exons = [x for nm in genes[gname] for x in ob.merge_intervals(per[nm])]
        assert segs == ob.segments(exons), gname
    assert gff != lh          # touching exons of one isoform are one exon in these formats
    with pytest.raises(L.LsqError, match="Unknown file format"):
        L.Annotation(p("f.gff"), p("f.map"), 0, 10, "UCSC_BED", "UCSC_GENE2ISOFORM")
    with pytest.raises(L.LsqError, match="Unknown file format"):
        L.Annotation(p("f.gff"), p("f.map"), 0, 10, "UCSC_GFF", "NO_SUCH_MAP")


def test_count_and_classify_refuse_solve_only_formats(tmp_path, monkeypatch):
    c, d = load_case("formats", tmp_path)
    monkeypatch.chdir(d)
    for r in c["count"]:
        rc, text = L.cli_run("count", r["argv"])
        assert rc == r["exit"] == 1 and text == ""
    rc, _ = L.cli_run("classify", ["0", "f", str(tmp_path) + "/", "UCSC_GFF", "f.gff", "UCSC_GENE2ISOFORM", "f.map", "0", "10"])
    assert rc == 1
    for idx in (6, 7):          # unknown literals in `solve`
        r = c["solve"][idx]
        rc, text = L.cli_run("solve", r["argv"])
        assert rc == r["exit"] == 1 and text == ""


def test_interval_merge_is_order_dependent():
    # the reference's add_interval merges a touching interval only when it is already stored
    # to the LEFT of the new one
    assert ob.merge_intervals([(10, 20), (20, 30)]) == [(10, 30)]
    assert ob.merge_intervals([(20, 30), (10, 20)]) == [(10, 20), (20, 30)]
    assert ob.merge_intervals([(10, 20), (15, 40), (50, 60), (5, 55)]) == [(5, 60)]


def test_mrf_parse_counts(tmp_path):
    c, d = load_case("edge", tmp_path)
    a = L.Annotation(os.path.join(d, "e.interval"), os.path.join(d, "e.map"))
    ev = L.Events(a, ("SHORT_READ",), (100,))
    r = L.Reads.from_mrf(os.path.join(d, "e.mrf"), ev)
    # 15 lines after the header, one comment, the unterminated last line is never seen
    assert len(r) == 13
    assert r.num_blocks == 16
    with pytest.raises(L.LsqError) as ei:
        L.Reads.from_mrf(os.path.join(d, "e.mrf"), ev, read_format="MRF_PAIRED")
    assert ei.value.status == -3
    with pytest.raises(L.LsqError) as ei:
        L.Reads.from_mrf(os.path.join(GOLD, "errors", "bad_number.mrf"), ev)
    assert ei.value.status == -4


def test_mrf_parse_threads_agree(tmp_path):
    spec = L.SynthSpec(seed=5, n_events=50, n_reads=120000, read_length=75, n_chrom=3)
    L.synth_write(spec, str(tmp_path), "t", write_mrf=True)
    a = L.Annotation(str(tmp_path / "t.interval"), str(tmp_path / "t.map"))
    ev = L.Events(a, ("SHORT_READ",), (75,))
    r1 = L.Reads.from_mrf(str(tmp_path / "t.mrf"), ev, n_threads=1)
    r4 = L.Reads.from_mrf(str(tmp_path / "t.mrf"), ev, n_threads=4)
    rs = L.Reads.synthetic(spec, ev)
    assert len(r1) == len(r4) == len(rs) == 120000
    assert r1.num_blocks == r4.num_blocks == rs.num_blocks


CLASSIFY_CASES = ["toy", "classify_mix", "events_s1", "events_s2", "events_s3"]


@pytest.mark.parametrize("name", CLASSIFY_CASES)
def test_classify_matches_reference(name, tmp_path, monkeypatch):
    """one .matrix per selected gene with two or more isoforms (classify/classify.cpp:159,199-228), byte for byte what
    the reference's classify wrote: single-isoform genes (classify_mix: `solo`, and `cut` whose second map line has no
    newline) leave no file; events_s2 selects a gene range"""
    c, d = load_case(name, tmp_path)
    out = tmp_path / "classify"
    out.mkdir()
    monkeypatch.chdir(d)
    argv = list(c["classify"]["argv"])
    argv[2] = str(out) + "/"
    rc, _ = L.cli_run("classify", argv)
    assert rc == c["classify"]["exit"] == 0
    assert sorted(os.listdir(out)) == c["classify"]["files"] and c["classify"]["files"]
    for fn in c["classify"]["files"]:
        assert open(out / fn).read() == open(os.path.join(d, "classify", fn)).read()


BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lesseq_amd", "bin")


def test_classify_executable_as_a_process(tmp_path):
    """lesseq_amd/bin/classify spawned like the reference's: files as above, nothing on stdout, log lines with the
    reference's prefix (jsc/util/log.hpp:64-70) on stderr at log level 2, exit 0; needs no GPU"""
    import re
    import subprocess
    c, d = load_case("classify_mix", tmp_path)
    out = tmp_path / "cls"
    out.mkdir()
    argv = list(c["classify"]["argv"])
    argv[0], argv[2] = "2", str(out) + "/"
    p = subprocess.run([os.path.join(BIN, "classify")] + argv, cwd=d, capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout == ""
    assert sorted(os.listdir(out)) == c["classify"]["files"]
    for fn in c["classify"]["files"]:
        assert open(out / fn).read() == open(os.path.join(d, "classify", fn)).read()
    lines = [ln for ln in p.stderr.split("\n") if ln]
    assert lines and all(re.match(r"^\[LOG \d{4}-\d{2}-\d{2} \d{2}:\d{2}:\d{2} [A-Z]+\d?\] ", ln) for ln in lines), p.stderr
    # too few arguments: usage error, exit 1 (classify/classify.cpp:20-23,58-60)
    p = subprocess.run([os.path.join(BIN, "classify")] + argv[:5], cwd=d, capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout == ""


def test_executables_exit_statuses_decided_before_the_gpu_as_processes(tmp_path):
    """count / solve spawned with the `errors` golden argvs whose status the reference decides before any read is
    counted: usage (too few arguments), bad number, unknown format -> exit 1 with an error line; an isoform file that
    does not open -> the reference asserts (SIGABRT, shell status 134; count/count.cpp:139)"""
    import subprocess
    c, d = load_case("errors", tmp_path)
    n = 0
    for tool in ("count", "solve"):
        for r in c[tool]:
            fmt_err = r["argv"][3] != "LH_GENE_TXT" or "MRF_PAIRED" in r["argv"] or len(r["argv"]) < 13 or "fifty" in r["argv"] or "abc" in r["argv"]
            if not fmt_err:
                continue
            p = subprocess.run([os.path.join(BIN, tool)] + r["argv"], cwd=d, capture_output=True, text=True)
            if p.returncode == 3:
                continue          # a status the reference reaches only after loading reads: needs the device (GPU suite)
            assert p.returncode == r["exit"] == 1 and p.stdout == "", (tool, r["argv"], p.returncode)
            assert "ERROR" in p.stderr
            n += 1
    assert n >= 3
    base = ["0", "x", "./", "LH_GENE_TXT", "no_such_file.interval", "UCSC_GENE2ISOFORM", "toy.map", "0", "10", "MRF_SINGLE", "SHORT_READ", "50", "toy.mrf"]
    p = subprocess.run([os.path.join(BIN, "count")] + base, cwd=d, capture_output=True, text=True)
    # the reference's assert(ifs) aborts: a shell sees status 134 (128 + SIGABRT); this executable exits with that status
    assert p.returncode in (134, -6) and p.stdout == "" and "cannot open" in p.stderr, p.returncode


def test_exit_statuses_decided_before_the_gpu(tmp_path, monkeypatch):
    c, d = load_case("errors", tmp_path)
    monkeypatch.chdir(d)
    # count #1 (unknown read type) is only detected after the reads were uploaded, #0 and #6 (fields
    # that fail the cast) by the device parser: GPU tests (test_cli_matches_reference_golden)
    for idx in (2, 3, 4, 5):
        r = c["count"][idx]
        rc, text = L.cli_run("count", r["argv"])
        assert rc == r["exit"] == 1, r["argv"]
        assert text == ""
    r = c["solve"][0]
    rc, text = L.cli_run("solve", r["argv"])
    assert rc == r["exit"] == 1 and text == ""
    # unopenable file: the reference asserts (SIGABRT); here exit status 134
    rc, _ = L.cli_run("count", ["0", "x", "./", "LH_GENE_TXT", "nope.interval", "UCSC_GENE2ISOFORM", "toy.map", "0", "10", "MRF_SINGLE", "SHORT_READ", "50", "toy.mrf"])
    assert rc == 134


def test_synthetic_generator_is_deterministic(tmp_path):
    spec = L.SynthSpec(seed=3, n_events=30, n_reads=2000, read_length=100, n_chrom=2)
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    L.synth_write(spec, str(tmp_path / "a"), "s")
    L.synth_write(spec, str(tmp_path / "b"), "s")
    for ext in ("interval", "map", "mrf"):
        assert open(tmp_path / "a" / ("s." + ext)).read() == open(tmp_path / "b" / ("s." + ext)).read()
    assert len(open(tmp_path / "a" / "s.mrf").read().splitlines()) == 2001


def test_named_read_formats_group_lines_by_name(tmp_path, monkeypatch):
    """solve's UCSC_GFF / UCSC_BED / WORMBASE_GFF3 readers: the host parser keeps exactly the reads the
    oracle loads (lines of one name make one read; BED and GFF3 lines stand or fall by their span)"""
    c, d = load_case("readfmts", tmp_path)
    monkeypatch.chdir(d)
    a = L.Annotation("rf.interval", "rf.map", 0, 1000)
    ev = L.Events(a, ("SHORT_READ",), (50,))
    for idx, (fmt, path) in enumerate((("UCSC_GFF", "rf.gff"), ("UCSC_BED", "rf.bed"), ("WORMBASE_GFF3", "rf.gff3"))):
        r = c["solve"][idx]
        assert r["argv"][9] == fmt
        rc, _, _ = ob.run("solve", r["argv"])
        assert rc == 0
        reads = L.Reads.from_mrf(path, ev, read_format=fmt)
        assert len(reads) == ob.last_n_loaded[0] > 500
        blk_off = reads.arrays()[0]
        assert int(blk_off[-1]) == reads.num_blocks >= len(reads)
    with pytest.raises(L.LsqError, match="Unknown file format"):
        L.Reads.from_mrf("rf.bed", ev, read_format="UCSC_PSL")
    # `count` knows MRF_SINGLE only
    for r in c["count"]:
        rc, text = L.cli_run("count", r["argv"])
        assert rc == r["exit"] == 1 and text == ""


def test_oracle_fisher_information_against_a_direct_derivation(tmp_path, monkeypatch):
    """fim.h restated in the oracle (parity unpinned) checked against the textbook form on the toy inputs: for K = 2,
    I = sum over isoforms k and their accessible starts of theta_k G_k (dG_1 - dG_2)^2 / (sum_j theta_j dG_j)^2 with the
    reads enumerated here position by position from the segment lengths (SHORT reads: a segment offers its length + 1
    starts, the one past its end being the next segment's first)"""
    monkeypatch.setenv("LSQO_FIM", "1")
    c, d = load_case("toy", tmp_path)
    monkeypatch.chdir(d)
    r = [x for x in c["runs"] if x["tool"] == "solve" and x["exit"] == 0][0] if "runs" in c else None
    if r is None:
        from test_parity_gpu import runs
        r = [x for t, x in runs(c) if t == "solve" and x["exit"] == 0][0]
    argv = r["argv"]
    rc, _, exact = ob.run("solve", argv)
    assert rc == 0 and all("fim" in g for g in exact)
    R = int(argv[11])
    iso = parse_interval(argv[4])
    g2i = {}
    for line in open(argv[6]).read().split("\n")[:-1]:
        g, i = line.split()
        g2i.setdefault(g, []).append(i)
    for g in exact:
        if g["K"] != 2:
            continue
        forms = [iso[n] for n in g2i[g["gname"]]]
        segs = ob.segments([x for f in forms for x in f])
        member = [[any(s >= a and e <= b for a, b in f) for (s, e) in segs] for f in forms]
        lens = [e - s for s, e in segs]
        th = g["theta"]
        reads = []           # (isoform, tuple of segment indices)
        G = []
        for k in range(2):
            mine = [n for n in range(len(segs)) if member[k][n]]
            L = sum(lens[n] for n in mine)
            starts = []
            cum = 0
            for i, n in enumerate(mine):
                cum += lens[n]
                if cum + R > L:
                    starts += [cum - lens[n] + o for o in range(max(lens[n] + 1 - (cum + R - L), 0))]
                    break
                starts += [cum - lens[n] + o for o in range(lens[n] + 1)]
            G.append(1.0 / len(starts) if starts else 0.0)
            ends = np.cumsum([lens[n] for n in mine])
            for s0 in starts:
                se = int(np.searchsorted(ends, s0, side="right"))
                ee = int(np.searchsorted(ends, s0 + R, side="left"))
                reads.append((k, tuple(mine[se:ee + 1])))
        info = 0.0
        for k, run in reads:
            comp = []
            for j in range(2):
                mine = [n for n in range(len(segs)) if member[j][n]]
                comp.append(any(tuple(mine[a:a + len(run)]) == run for a in range(len(mine))))
            dG = [G[j] if comp[j] else 0.0 for j in range(2)]
            s_ = sum(th[j] * dG[j] for j in range(2) if dG[j] > 0 and th[j] > 0)
            if th[k] != 0:
                info += th[k] * G[k] * (dG[0] - dG[1]) ** 2 / s_ ** 2
        assert abs(info - g["fim"][0][0][0]) <= 1e-9 * abs(info), (g["gname"], info, g["fim"])
        assert abs(g["fim_var"][0][0] - 1.0 / info) <= 1e-9 / info
        assert abs(g["fim_var"][0][1] - 2.0 / info) <= 1e-9 / info       # fim.h:74-93: all entries of the inverse plus its trace


def test_no_exception_crosses_the_boundary(tmp_path):
    """include/lesseq_hip.h: "no exceptions across the boundary" (the reference's contract for a failure is a logged
    message and a status, count/count.cpp:20-38).  A C++ exception below an entry point -- out of memory in a loader, a
    length_error, something thrown inside a helper thread -- comes back as LSQ_E_INTERNAL (-9) with its text; before
    round 3 it was std::terminate, i.e. SIGABRT in the host process.  A child process makes the calls, so that a
    regression shows up as that child's death and not as the death of the test run."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import ctypes as C
        import lesseq_amd as L
        lib = L.lib
        for kind, want in ((0, b"out of memory"), (1, b"vector"), (2, b"unknown exception"), (3, b"thrown inside a helper thread")):
            st = lib.lsq_debug_throw(kind)
            assert st == -9, (kind, st)
            assert want in lib.lsq_last_error(), (kind, lib.lsq_last_error())
        assert lib.lsq_debug_throw(9) == 0
        # a real path: with the address space capped, the arrays of 2^31 reads cannot be allocated (std::bad_alloc inside lsq_synth_reads)
        d = %r
        spec = L.SynthSpec(seed=1, n_events=4, n_reads=100, read_length=50, n_chrom=1)
        L.synth_write(spec, d, "s", write_mrf=False)
        a = L.Annotation(d + "/s.interval", d + "/s.map")
        ev = L.Events(a, ("SHORT_READ",), (50,))
        big = L.SynthSpec(seed=1, n_events=4, n_reads=2 ** 31, read_length=50, n_chrom=1)
        import resource
        used = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
        resource.setrlimit(resource.RLIMIT_AS, (used + (1 << 30), used + (1 << 30)))
        h = C.c_void_p()
        st = lib.lsq_synth_reads(C.byref(big.c), ev.h, 1, C.byref(h))
        assert st == -9 and lib.lsq_last_error().startswith(b"lsq_synth_reads"), (st, lib.lsq_last_error())
        print("contained")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LSQ_NO_TORCH="1"))
    assert p.returncode == 0 and "contained" in p.stdout, (p.returncode, p.stdout, p.stderr)


def test_in_process_cli_leaves_no_thread_behind(tmp_path, monkeypatch):
    """lsq_cli_run inside a host process makes its device context on the calling thread (an executable overlaps it with
    the annotation load on a second thread, lsq_cli_main); whatever the outcome -- here: no GPU, or a GPU -- the call
    returns with every helper thread joined"""
    import threading
    c, d = load_case("toy", tmp_path)
    monkeypatch.chdir(d)

    def native_threads():
        return len(os.listdir("/proc/self/task"))
    L.cli_run("count", c["count"][0]["argv"])          # first call: the runtime may start service threads of its own
    before = native_threads()
    for _ in range(3):
        rc, _ = L.cli_run("count", c["count"][0]["argv"])
        assert rc in (0, 3)
    assert native_threads() <= before
    assert threading.active_count() >= 1


def test_share_plan_of_a_count_launch():
    """run_count's share plan on made-up buckets (host only, lsq_debug_plan_shares): the bounds rise from 0 to the last slot,
    no share is longer than 2^21 slots, weighted shares are equal in cost where no cut snaps (a two-block record, a walk
    look and a visit weigh what the options say), a taper makes them fall off along the grid, cuts near a bucket boundary
    land on it, and without weights the shares are equal in slots."""
    import ctypes as C
    rng = np.random.default_rng(5)
    B = 400
    n1 = rng.integers(0, 40000, B).astype(np.uint64)
    n2 = rng.integers(0, 12000, B).astype(np.uint64)
    rest = rng.integers(0, 50, B).astype(np.uint64)
    n1[7] = n2[7] = rest[7] = 0                       # an empty bucket
    so = np.concatenate([[0], np.cumsum(n1 + n2 + rest)]).astype(np.uint64)
    look1 = (n1 * rng.random(B) * 0.05).astype(np.uint32)
    look2 = (n2 * rng.random(B) * 0.5).astype(np.uint32)
    look1[11] = 30 * int(n1[11])                      # a bucket the walk is busy with
    packed = np.ones(B, dtype=np.uint8)

    def ptr(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    def plan(grid, costs, weighted=1, snap=0, with_counts=True):
        cuts = np.zeros(grid + 1, dtype=np.uint64)
        c4 = (C.c_double * 4)(*costs)
        f = L.lib.lsq_debug_plan_shares
        f.restype = C.c_int
        rc = f(ptr(so, C.c_ulonglong), ptr(packed, C.c_ubyte),
               ptr(n1, C.c_ulonglong) if with_counts else None, ptr(n2, C.c_ulonglong) if with_counts else None,
               ptr(look1, C.c_uint) if with_counts else None, ptr(look2, C.c_uint) if with_counts else None,
               C.c_ulonglong(B), C.c_ulonglong(grid), c4, weighted, snap, ptr(cuts, C.c_ulonglong))
        assert rc == 0
        return cuts.astype(np.int64)

    def cost_of(cuts, c2, cw, cv):
        """cost of every share by the plan's own model, record by record"""
        dens = np.zeros(int(so[-1]))
        for b in range(B):
            a = int(so[b])
            if n1[b]:
                dens[a:a + int(n1[b])] = 1.0 + cw * look1[b] / n1[b]
            if n2[b]:
                dens[a + int(n1[b]):a + int(n1[b] + n2[b])] = c2 + cw * look2[b] / n2[b]
            dens[a + int(n1[b] + n2[b]):int(so[b + 1])] = 1e-3
        cum = np.concatenate([[0.0], np.cumsum(dens)])
        per = cum[cuts[1:]] - cum[cuts[:-1]]
        starts = np.searchsorted(so.astype(np.int64), cuts[:-1], side="right") - 1
        ends = np.searchsorted(so.astype(np.int64), np.maximum(cuts[1:] - 1, cuts[:-1]), side="right") - 1
        return per, per + cv * (ends - starts + 1)

    grid = 1500
    for costs in ((4.3, 9.0, 7000.0, 1.0), (2.0, 0.0, 0.0, 1.0), (4.3, 9.0, 7000.0, 0.25)):
        cuts = plan(grid, costs)
        assert cuts[0] == 0 and cuts[-1] == int(so[-1]) and np.all(np.diff(cuts) >= 0) and np.diff(cuts).max() <= 1 << 21
        per, with_visits = cost_of(cuts, costs[0], costs[1], costs[2])
        if costs[3] == 1.0:
            # equal in cost: a share may differ from the mean by the visits the model charges per bucket start, not per cut
            assert abs(with_visits - with_visits.mean()).max() < 0.02 * with_visits.mean() + 2 * costs[2] + 50
            assert with_visits.std() < 0.5 * (np.diff(cuts) * 1.0).std() or costs[1] == 0.0
        else:
            k = grid // 10
            first, last = with_visits[:k].mean(), with_visits[-k:].mean()
            assert 0.2 < last / first < 0.45              # the last tenth against the first, taper 0.25 in a line
            assert np.all(np.diff(np.convolve(with_visits, np.ones(100) / 100, mode="valid")) < 0.02 * first)
    # the hot bucket gets more shares than its slots alone would
    cuts = plan(grid, (4.3, 9.0, 7000.0, 1.0))
    inside = np.sum((cuts[1:-1] > int(so[11])) & (cuts[1:-1] < int(so[12])))
    assert inside > 5 * (int(so[12]) - int(so[11])) * grid / int(so[-1])
    # snapping: with about one share per bucket most cuts land on bucket boundaries
    cuts = plan(B, (4.3, 9.0, 7000.0, 1.0), snap=1)
    on_boundary = np.isin(cuts[1:-1], so.astype(np.int64)).mean()
    # (without snapping a cut lands on a boundary only where its target falls into the staging a bucket begins with)
    assert on_boundary > 0.5 and np.isin(plan(B, (4.3, 9.0, 7000.0, 1.0), snap=0)[1:-1], so.astype(np.int64)).mean() < 0.5 * on_boundary
    # a bucket the streaming kernel does not visit (a generic or host bucket) weighs nothing: no share is spent inside it
    packed[20] = 0
    cuts = plan(grid, (4.3, 9.0, 7000.0, 1.0))
    assert np.sum((cuts[1:-1] > int(so[20])) & (cuts[1:-1] < int(so[21]))) <= 1
    packed[20] = 1
    # no weights: equal slots
    for kw in (dict(weighted=0), dict(with_counts=False)):
        cuts = plan(grid, (4.3, 9.0, 7000.0, 0.25), **kw)
        assert np.abs(np.diff(cuts) - int(so[-1]) / grid).max() <= 1
    # a share longer than 2^21 slots is never planned: the weights give way to equal shares
    big = plan(3, (1000.0, 0.0, 0.0, 0.05))
    assert np.diff(big).max() <= max(1 << 21, int(so[-1]) // 3 + 1)


def test_sorted_generator_writes_the_same_reads_in_coordinate_order(tmp_path):
    """lsq_synth_spec.sorted: the stream's reads by chromosome, first base and number; the same lines as the shuffled file"""
    for zipf in (False, True):
        a = L.SynthSpec(seed=7, n_events=300, n_reads=150000, read_length=100, n_chrom=4, zipf=zipf, sorted_reads=True)
        b = L.SynthSpec(seed=7, n_events=300, n_reads=150000, read_length=100, n_chrom=4, zipf=zipf)
        L.synth_write(a, str(tmp_path), "s")
        L.synth_write(b, str(tmp_path), "u")
        sl = open(tmp_path / "s.mrf").read().split("\n")
        ul = open(tmp_path / "u.mrf").read().split("\n")
        assert sl[0] == ul[0] == "AlignmentBlocks" and sl[-1] == ul[-1] == ""
        keys = [(int(l.split(":")[0][3:]), int(l.split(":")[2])) for l in sl[1:-1]]
        assert keys == sorted(keys) and len(keys) == 150000
        assert sorted(sl[1:-1]) == sorted(ul[1:-1])
        assert open(tmp_path / "s.interval").read() == open(tmp_path / "u.interval").read()
