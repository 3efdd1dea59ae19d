"""Parity tests proper: the HIP path through the C ABI against the oracle and the golden
vectors.  Need an MI355X: python -m pytest tests -m gpu."""
import json
import os

import numpy as np
import pytest

import lesseq_amd as L
import oracle_binding as ob
from test_oracle_golden import GOLD, CASES, load_case, runs

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6      # north_star: solve expression levels within 1e-6 relative


@pytest.mark.parametrize("name", CASES)
def test_cli_matches_reference_golden(name, tmp_path, monkeypatch):
    """count tables byte-identical to the reference's stdout; solve tables identical as printed
    (six significant digits), allowing one unit in the last printed digit"""
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    for tool, r in runs(c):
        rc, text = L.cli_run(tool, r["argv"])
        exp = open(os.path.join(d, r["stdout"])).read()
        assert rc == r["exit"], (name, tool, r["argv"])
        if tool == "count":
            assert text == exp, (name, r["argv"])
        else:
            assert ob.solve_text_close(text, exp), (name, r["argv"], text, exp)


def gpu_exact(argv, tool="solve", want_fim=False, band=None, repeat=1, options=None):
    """runs the library pipeline step by step; returns per-gene dicts like the oracle's exact output.
    band: lsq_set_em_guard_band (None = the library's 1e-11); repeat: count + solve that many times (from the third
    time on a lane's solve uses the placement, and the one-lane form, it learned from its first)"""
    per = 5 if tool == "solve" else 4
    groups = [argv[9 + i * per: 9 + (i + 1) * per] for i in range((len(argv) - 9) // per)]
    a = L.Annotation(argv[4], argv[6], int(argv[7]), int(argv[8]), argv[3], argv[5])
    ev = L.Events(a, tuple(g[1] for g in groups), tuple(int(g[2]) for g in groups))
    ctx = L.Context(0)
    if band is not None:
        ctx.set_em_guard_band(band)
    for k, v in (options or {}).items():
        ctx.set_option(k, v)
    ctx.upload_events(ev)
    for m, g in enumerate(groups):
        ctx.upload_reads(m, L.Reads.from_mrf(g[3], ev, read_format=g[0]))
    for _ in range(repeat):
        ctx.count()
        ctx.solve()
    cnt, bases = ctx.counts()
    theta, ll, iters, flags = ctx.solution()
    off = ev.class_offsets()
    out = []
    io = 0
    for i in range(len(ev)):
        K = ev.K(i)
        cl = cnt[:, off[i]:off[i + 1]]
        bl = bases[:, off[i]:off[i + 1]]
        out.append({
            "gname": ev.gene_name(i), "K": K,
            "supports": [int(cl[m].sum()) for m in range(len(groups))],
            "bases": [int(bl[m].sum()) for m in range(len(groups))],
            "iso_count": [int(sum(cl[:, c - 1].sum() for c in range(1, 1 << K) if c >> j & 1)) for j in range(K)],
            "theta": [float(theta[io + j]) for j in range(K)], "logll": float(ll[i]),
            "iters": int(iters[i]), "flags": int(flags[i]),
        })
        io += K
    if want_fim:
        foff, fim, vd, vi = ctx.fim()
        for i, g in enumerate(out):
            D = g["K"] - 1
            g["fim"] = [fim[m, int(foff[i]):int(foff[i + 1])].reshape(D, D).tolist() for m in range(len(groups))]
            g["fim_var"] = [(float(vd[m, i]), float(vi[m, i])) for m in range(len(groups))]
    ctx.close()
    return out


def close(a, b):
    """equal (also both -inf / both nan, the ARS = 0 corner) or within REL_TOL relative"""
    if a == b or (a != a and b != b):
        return True
    return abs(a - b) <= REL_TOL * max(abs(a), abs(b))


def same_bits(a, b):
    return a == b or (a != a and b != b)


def compare_exact(got, exp, what):
    """exact integers; theta, log-likelihood and iteration count of EVERY event: within REL_TOL where the kernel's
    class-wise sums stand, bit for bit where the event was replayed in the reference's per-read order (flag bit 2).
    No event may be left with the guard-band flag and no replay.  Returns the number of replayed events."""
    assert len(got) == len(exp), what
    n_replayed = 0
    for g, e in zip(got, exp):
        assert g["gname"] == e["gname"] and g["K"] == e["K"], what
        assert g["supports"] == e["supports"], (what, g["gname"])
        assert g["bases"] == e["bases"], (what, g["gname"])
        assert g["iso_count"] == e["iso_count"], (what, g["gname"])
        if e["theta"] is not None:
            assert not (g["flags"] & 1) or (g["flags"] & 4), (what, g["gname"], "flagged but not replayed")
            assert g["iters"] == e["iters"], (what, g["gname"], g["iters"], e["iters"], g["flags"])
            if g["flags"] & 4:
                n_replayed += 1
                for a, b in zip(g["theta"], e["theta"]):
                    assert same_bits(a, b), (what, g["gname"], a, b)
                assert same_bits(g["logll"], e["logll"]), (what, g["gname"], g["logll"], e["logll"])
            else:
                for a, b in zip(g["theta"], e["theta"]):
                    assert close(a, b), (what, g["gname"], a, b)
                assert close(g["logll"], e["logll"]), (what, g["gname"], g["logll"], e["logll"])
    return n_replayed


@pytest.mark.parametrize("name", [c for c in CASES if c not in ("errors",)])
def test_exact_integers_and_theta_on_golden_inputs(name, tmp_path, monkeypatch):
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    for r in c["solve"]:
        if r["exit"] != 0:
            continue
        rc, _, exact = ob.run("solve", r["argv"])
        assert rc == 0
        compare_exact(gpu_exact(r["argv"]), exact, (name, r["argv"][3:9]))


SYNTH = [
    # BASELINE.json configs[0]: 10k reads / 100 SE events (plumbing)
    dict(id="c1_10k_100se", seed=1, n_events=100, n_reads=10000, R=100, n_chrom=1, types=("SE",)),
    dict(id="mixed_200k_2k", seed=2, n_events=2000, n_reads=200000, R=100, n_chrom=5, types=L.EVENT_TYPES),
    dict(id="zipf_300k_3k", seed=5, n_events=3000, n_reads=300000, R=75, n_chrom=4, types=L.EVENT_TYPES, zipf=True),
    dict(id="dense_overlap", seed=9, n_events=400, n_reads=150000, R=50, n_chrom=1, types=L.EVENT_TYPES, overlap=0.6),
    dict(id="se_ri_1chrom", seed=2, n_events=1500, n_reads=400000, R=100, n_chrom=1, types=("SE", "RI")),
]


@pytest.mark.parametrize("cfg", SYNTH, ids=[c["id"] for c in SYNTH])
def test_synthetic_parity_vs_oracle(cfg, tmp_path):
    spec = L.SynthSpec(cfg["seed"], cfg["n_events"], cfg["n_reads"], cfg["R"], cfg["n_chrom"], cfg["types"],
                       cfg.get("zipf", False), cfg.get("overlap", 0.10))
    L.synth_write(spec, str(tmp_path), "s")
    argv = ["0", "s", "./", "LH_GENE_TXT", str(tmp_path / "s.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "s.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", str(cfg["R"]), str(tmp_path / "s.mrf"), str(cfg["n_reads"] * cfg["R"])]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    got = gpu_exact(argv)
    compare_exact(got, exact, cfg["id"])
    # and the printed tables
    rc, text = L.cli_run("count", argv[:-1])
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext
    rc, text = L.cli_run("solve", argv)
    assert rc == 0 and ob.solve_text_close(text, otext)
    assert sum(sum(g["supports"]) for g in got) > cfg["n_reads"] // 3


def test_gene_range_and_unknown_read_type(tmp_path, monkeypatch):
    c, d = load_case("errors", tmp_path)
    monkeypatch.chdir(d)
    r = c["count"][1]          # LONG_READ: noticed in the per-gene loop, after the reads were loaded
    rc, text = L.cli_run("count", r["argv"])
    assert rc == r["exit"] == 1 and text == ""
    # with an empty gene range the reference never reaches that check
    argv = list(r["argv"])
    argv[7], argv[8] = "5", "5"
    rc, text = L.cli_run("count", argv)
    assert rc == 0 and text == ""


def test_results_do_not_depend_on_launch_geometry(tmp_path, monkeypatch):
    """bucket size (LDS budget), grid size and resident workgroups change the work split, never the integers"""
    spec = L.SynthSpec(11, 3000, 500000, 100, 3, L.EVENT_TYPES)
    a_dir = str(tmp_path)
    L.synth_write(spec, a_dir, "g", write_mrf=False)
    results = []
    for budget, mult, per_cu, per_look in (("8192", "2", -1, 0), ("1024", "1", 0, 8), ("65536", "7", 5, 4), ("90000", "16", 3, 8), ("8192", "2", 5, 0)):
        monkeypatch.setenv("LSQ_LDS_BUDGET", budget)
        a = L.Annotation(os.path.join(a_dir, "g.interval"), os.path.join(a_dir, "g.map"))
        ev = L.Events(a, ("SHORT_READ",), (100,))
        ctx = L.Context(0)
        ctx.set_option("grid_multiplier", int(mult))
        ctx.set_option("workgroups_per_cu", per_cu)
        ctx.set_option("reads_per_look", per_look)
        with pytest.raises(L.LsqError):
            ctx.set_option("workgroups_per_cu", 1000)
        ctx.upload_events(ev)
        ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
        ctx.count()
        cnt, bases = ctx.counts()
        results.append((ev.num_buckets, cnt.copy(), bases.copy()))
        assert ctx.launch_info()[0] == (per_look if per_look else (8 if per_cu == 5 else 4))
        ctx.close()
    assert len({r[0] for r in results}) > 1
    for r in results[1:]:
        assert np.array_equal(r[1], results[0][1]) and np.array_equal(r[2], results[0][2])
    # ... nor does the share plan (round 3: shares equal in estimated cost, tapering towards the end of the grid): one context,
    # replanned between counts -- equal reads, extreme weights, the steepest taper, every cut snapped or none
    monkeypatch.setenv("LSQ_LDS_BUDGET", "8192")
    a = L.Annotation(os.path.join(a_dir, "g.interval"), os.path.join(a_dir, "g.map"))
    ev = L.Events(a, ("SHORT_READ",), (100,))
    zipf = L.SynthSpec(12, 3000, 500000, 100, 3, L.EVENT_TYPES, True)
    for sp in (spec, zipf):
        ctx = L.Context(0)
        ctx.upload_events(ev)
        ctx.upload_reads(0, L.Reads.synthetic(sp, ev))
        ref = None
        for opts in ({"share_weighted": 0}, {}, {"share_weighted": 1, "share_taper": 0.05, "grid_multiplier": 9}, {"share_taper": 1, "share_cost_parked": 1e6, "grid_multiplier": 0.3},
                     {"share_cost_visit": 1e6, "share_cost_two_block": 0}, {"share_cost_visit": 0, "share_cost_two_block": 1000, "snap_shares": 0, "share_taper": 0.2},
                     {"share_cost_hot": 50, "share_cost_parked": 0}):
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.count()
            got = [x.copy() for x in ctx.counts()]
            if ref is None:
                ref = got
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), opts
        with pytest.raises(L.LsqError):
            ctx.set_option("share_taper", -1)
        ctx.close()


def _full_size_linearity_and_sample(tmp_path, seed, n_ev, n_reads, n_chrom, types, zipf, chunks, sample_reads, R=100):
    """count(all) == sum of count(chunk) over disjoint chunks of the same read stream (integer sums are additive),
    retained reads add up likewise; then the first `sample_reads` reads through the MRF text path: exact integers,
    theta, iteration counts against the oracle, and the printed count table byte for byte."""
    spec_all = L.SynthSpec(seed, n_ev, n_reads, R, n_chrom, types, zipf)
    L.synth_write(spec_all, str(tmp_path), "w", write_mrf=False)
    a = L.Annotation(str(tmp_path / "w.interval"), str(tmp_path / "w.map"))
    ev = L.Events(a, ("SHORT_READ",), (R,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec_all, ev))
    ctx.count()
    ctx.solve()
    cnt_all, bases_all = [x.copy() for x in ctx.counts()]
    theta_all, ll_all, it_all, fl_all = [x.copy() for x in ctx.solution()]
    retained_all = ctx.retained(0)
    assert int(cnt_all.sum()) > n_reads // 2      # a read may be valid for two overlapping events
    assert not np.any((fl_all & 1) & ~((fl_all >> 2) & 1)), "an event kept the guard-band flag without a replay"
    assert np.isfinite(theta_all).all()
    cs_sum, bs_sum, retained = np.zeros_like(cnt_all), np.zeros_like(bases_all), 0
    per = n_reads // chunks
    assert per * chunks == n_reads
    for k in range(chunks):
        ctx.upload_reads(0, L.Reads.synthetic(L.SynthSpec(seed, n_ev, per, R, n_chrom, types, zipf, first_read=k * per), ev))
        ctx.count()
        c, b = ctx.counts()
        cs_sum += c
        bs_sum += b
        retained += ctx.retained(0)
    ctx.close()
    assert retained == retained_all
    assert np.array_equal(cs_sum, cnt_all) and np.array_equal(bs_sum, bases_all)
    # the first reads of the stream through the text path, against the oracle
    L.synth_write(L.SynthSpec(seed, n_ev, sample_reads, R, n_chrom, types, zipf), str(tmp_path), "wm", write_mrf=True)
    argv = ["0", "w", "./", "LH_GENE_TXT", str(tmp_path / "w.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "w.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", str(R), str(tmp_path / "wm.mrf"), str(sample_reads * R)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    compare_exact(gpu_exact(argv), exact, "first %d reads" % sample_reads)
    rc, text = L.cli_run("count", argv[:-1])
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext
    rc, text = L.cli_run("solve", argv)
    assert rc == 0 and ob.solve_text_close(text, otext)


def test_full_size_config2_linearity_and_sample(tmp_path):
    """BASELINE.json configs[1] at full size (10 M reads, 5 k SE/RI events, one chromosome): ten 1 M-read chunks,
    the first of them against the oracle."""
    _full_size_linearity_and_sample(tmp_path, 2, 5000, 10_000_000, 1, ("SE", "RI"), False, 10, 1_000_000)


def test_full_size_config3_linearity_and_sample(tmp_path):
    """BASELINE.json configs[2] at full size -- bench.py's workload: 100 M reads, 50 k mixed events, 24 chromosomes --
    as ten 10 M-read chunks; the first 2 M reads against the oracle (from MRF text, count table byte-identical)."""
    _full_size_linearity_and_sample(tmp_path, 3, 50_000, 100_000_000, 24, L.EVENT_TYPES, False, 10, 2_000_000)


def test_full_size_config5_share_linearity_and_sample(tmp_path):
    """One GPU's eighth of BASELINE.json configs[4] (bench.py --workload c5s): 125 M reads over 25 k events with Zipf
    read depth (hot genes), ten chunks; the first 2 M reads against the oracle."""
    _full_size_linearity_and_sample(tmp_path, 5, 25_000, 125_000_000, 24, L.EVENT_TYPES, True, 10, 2_000_000)


@pytest.mark.parametrize("cfg", SYNTH[:3], ids=[c["id"] for c in SYNTH[:3]])
def test_guard_band_events_are_replayed_in_the_reference_order(cfg, tmp_path):
    """lsq_set_em_guard_band(1.0) puts every event with an EM loop inside the band: each is then solved again over its
    valid reads in index order with the reference's per-read sums (common/read.h:592-660), and theta, log-likelihood
    and iteration count must equal the oracle's bit for bit -- the path the library takes by itself for the rare
    event whose stop test lands within 1e-11 of the threshold."""
    spec = L.SynthSpec(cfg["seed"], cfg["n_events"], cfg["n_reads"], cfg["R"], cfg["n_chrom"], cfg["types"],
                       cfg.get("zipf", False), cfg.get("overlap", 0.10))
    L.synth_write(spec, str(tmp_path), "s")
    argv = ["0", "s", "./", "LH_GENE_TXT", str(tmp_path / "s.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "s.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", str(cfg["R"]), str(tmp_path / "s.mrf"), str(cfg["n_reads"] * cfg["R"])]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    n = compare_exact(gpu_exact(argv, band=1.0), exact, cfg["id"])
    assert n > cfg["n_events"] // 2          # events with reads and two isoforms all took the replay


@pytest.mark.parametrize("name", ["toy", "edge", "multi_method", "wild_s11", "wild_s13", "readfmts", "events_s2"])
def test_replay_on_golden_inputs(name, tmp_path, monkeypatch):
    """the same on golden inputs: several read files (per-file rows, file-major sums), named reads (UCSC_GFF / BED /
    GFF3: names against names in the index order), span-start ties, odd strands, K up to 5"""
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    n = 0
    for r in c["solve"]:
        if r["exit"] != 0:
            continue
        rc, _, exact = ob.run("solve", r["argv"])
        assert rc == 0
        n += compare_exact(gpu_exact(r["argv"], band=1.0), exact, (name, r["argv"][3:9]))
    assert n > 0


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)


def test_empty_and_degenerate_inputs(tmp_path, monkeypatch):
    """empty gene range, header-only read file, reads that all miss the events, a read file whose
    last line is unterminated: same rows and exit status as the oracle"""
    import golden_inputs as gi
    gi.write_toy(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    _write("empty.mrf", "AlignmentBlocks\n")
    _write("offtarget.mrf", "AlignmentBlocks\n" + "".join("chr1:+:%d:%d:1:50\n" % (20000 + 7 * i, 20049 + 7 * i) for i in range(300))
           + "chr7:-:5:54:1:50\nchr1:+:3001:3050:1:50")
    base = ["0", "t", "./", "LH_GENE_TXT", "toy.interval", "UCSC_GENE2ISOFORM", "toy.map"]
    for lo, hi, mrf in (("0", "10", "empty.mrf"), ("0", "10", "offtarget.mrf"), ("3", "3", "toy.mrf"), ("1", "2", "toy.mrf"), ("0", "1", "toy.mrf")):
        argv = base + [lo, hi, "MRF_SINGLE", "SHORT_READ", "50", mrf]
        rc, text = L.cli_run("count", argv)
        orc, otext, _ = ob.run("count", argv)
        assert (rc, text) == (orc, otext), (lo, hi, mrf)
        rc, text = L.cli_run("solve", argv + ["1000"])
        orc, otext, _ = ob.run("solve", argv + ["1000"])
        assert rc == orc and ob.solve_text_close(text, otext), (lo, hi, mrf)
        # the --fim extension on the same degenerate inputs: the table, then one finite-or-inf line per event and read file
        rc2, text2 = L.cli_run("solve", argv + ["1000", "--fim"])
        assert rc2 == rc and text2.startswith(text)
        n_rows = len([x for x in text.split("\n") if x])
        extra = [x for x in text2[len(text):].split("\n") if x]
        assert all(x.startswith("#fim\t") for x in extra) and len(extra) * 2 == n_rows      # the toy events have two isoforms each


def _random_gene_set(rng, n_genes, max_exons, max_iso, chroms=("c1", "c2"), spacing=4000, min_iso=1):
    """genes with up to `max_iso` isoforms over up to `max_exons` shared exon slots (so events get
    many segments and isoforms: the generic kernel's territory)"""
    iv, mp, genes = [], [], []
    pos = {c: 500 for c in chroms}
    for g in range(n_genes):
        c = chroms[rng.randrange(len(chroms))]
        n_slots = rng.randint(2, max_exons)
        slots = []
        p = pos[c]
        for _ in range(n_slots):
            ln = rng.randint(30, 160)
            slots.append((p, p + ln))
            p += ln + (0 if rng.random() < 0.15 else rng.randint(40, 400))
        pos[c] = p + rng.randint(200, spacing)
        K = rng.randint(min_iso, max_iso)
        forms = []
        for k in range(K):
            pick = sorted(rng.sample(range(n_slots), rng.randint(1, n_slots)))
            ex = []
            for i in pick:
                s, e = slots[i]
                if rng.random() < 0.2:      # alternative boundary inside the slot -> more segments
                    cut = rng.randint(5, e - s - 5)
                    s, e = (s + cut, e) if rng.random() < 0.5 else (s, s + cut)
                ex.append((s, e))
            forms.append(ex)
            name = "G%d.%d" % (g, k)
            iv.append(gi_line(name, c, "+" if g % 2 else "-", ex))
            mp.append("G%d\t%s\n" % (g, name))
        genes.append((c, "+" if g % 2 else "-", forms))
    return "".join(iv), "".join(mp), genes


def test_a_thousand_genes_beyond_the_kernel_limits_vs_oracle(tmp_path):
    """1 500 genes of 7 to 10 isoforms each (LSQ_MAX_ISOFORMS = 6): every one of them is evaluated on the host inside lsq_count /
    lsq_solve -- their reads pooled on the device, events dealt to LSQ_THREADS host threads, the reference's per-read EM --
    and equals the oracle bit for bit; the executable says at log level 1 that, and how much, the host evaluated"""
    import random
    import ctypes as C
    import golden_inputs as gi
    rng = random.Random(4711)
    R = 60
    iv, mp, genes = _random_gene_set(rng, 1500, 9, 10, chroms=("c1", "c2", "c3"), spacing=1500, min_iso=7)
    reads = []
    for _ in range(60000):
        c, strand, forms = genes[rng.randrange(len(genes))]
        f = sorted(forms[rng.randrange(len(forms))])
        tlen = sum(e - s for s, e in f)
        ln = min(tlen, rng.choice([R, R, R // 2, R + 30]))
        reads.append(gi.mrf_line(c, strand, gi.transcript_blocks(f, rng.randint(0, tlen - ln), ln)))
    _write(tmp_path / "h.interval", iv)
    _write(tmp_path / "h.map", mp)
    _write(tmp_path / "h.mrf", "AlignmentBlocks\n" + "".join(reads))
    argv = ["1", "h", "./", "LH_GENE_TXT", str(tmp_path / "h.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "h.map"),
            "0", "100000", "MRF_SINGLE", "SHORT_READ", str(R), str(tmp_path / "h.mrf"), str(60000 * R)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    got = gpu_exact(argv)
    n_exact = compare_exact(got, exact, "1500 genes beyond the limits")
    assert min(g["K"] for g in got) >= 7 and n_exact >= 1000
    assert sum(sum(g["supports"]) for g in got) > 30000
    # the context's own account of what the host did
    a = L.Annotation(argv[4], argv[6])
    ev = L.Events(a, ("SHORT_READ",), (R,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.from_mrf(argv[12], ev))
    ctx.count()
    hg, hr = C.c_uint64(0), C.c_uint64(0)
    assert L.lib.lsq_host_evaluated(ctx.h, C.byref(hg), C.byref(hr)) == 0
    assert hg.value == 1500 and 30000 < hr.value <= 60000
    ctx.close()
    import subprocess
    p = subprocess.run([os.path.join(BIN, "solve")] + argv, capture_output=True, text=True)
    assert p.returncode == 0 and ob.solve_text_close(p.stdout, otext)
    assert "WARNING" in p.stderr and "1500 gene(s) beyond the device kernels' limits" in p.stderr, p.stderr[-400:]


def gi_line(name, chrom, strand, exons):
    import golden_inputs as gi
    return gi.interval_line(name, chrom, strand, exons)


@pytest.mark.parametrize("seed,max_exons,max_iso,R", [(101, 6, 6, 40), (102, 12, 5, 75), (103, 20, 3, 120), (104, 4, 2, 150),
                                                      (105, 8, 11, 50), (106, 45, 4, 90), (107, 30, 9, 60)])
def test_many_segments_isoforms_and_blocks_vs_oracle(seed, max_exons, max_iso, R, tmp_path):
    """events with up to 6 isoforms / dozens of segments (generic kernel), short exons under long
    reads (three and more blocks per read: cleanup kernel), mixed with packed buckets; seeds 105-107: genes beyond
    the kernels' limits (up to 11 isoforms, more than 32 segments) among normal ones -- those clusters are evaluated
    on the host inside lsq_count / lsq_solve (per-read EM, bit-equal to the oracle's), the rest on the device"""
    import random
    import golden_inputs as gi
    rng = random.Random(seed)
    iv, mp, genes = _random_gene_set(rng, 60, max_exons, max_iso)
    reads = []
    for _ in range(6000):
        c, strand, forms = genes[rng.randrange(len(genes))]
        f = sorted(forms[rng.randrange(len(forms))])
        tlen = sum(e - s for s, e in f)
        u = rng.random()
        if u < 0.8 and tlen > 10:
            ln = min(tlen, rng.choice([R, R, R // 2, R + 30]))
            t0 = rng.randint(0, tlen - ln)
            bl = gi.transcript_blocks(f, t0, ln)
        elif u < 0.9:
            s0 = f[0][0] + rng.randint(-3, 3)
            bl = [(max(s0, 0), max(s0, 0) + rng.choice([10, R]))]
        else:
            s0 = rng.randint(f[0][0], f[-1][1])
            bl = [(s0, s0 + R // 2), (s0 + R // 2 + rng.randint(0, 90), s0 + R + rng.randint(91, 150))]
        reads.append(gi.mrf_line(c, strand if rng.random() < 0.85 else "+", bl))
    _write(tmp_path / "m.interval", iv)
    _write(tmp_path / "m.map", mp)
    _write(tmp_path / "m.mrf", "AlignmentBlocks\n" + "".join(reads))
    argv = ["0", "m", "./", "LH_GENE_TXT", str(tmp_path / "m.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "m.map"),
            "0", "100000", "MRF_SINGLE", "SHORT_READ", str(R), str(tmp_path / "m.mrf"), str(6000 * R)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    got = gpu_exact(argv)
    n_exact = compare_exact(got, exact, "seed %d" % seed)
    if max_iso > 6 or max_exons > 32:
        assert max(g["K"] for g in got) > 6 or max_exons > 32
        assert n_exact > 0          # host-evaluated genes carry flag bit 2 and must equal the oracle bit for bit
    assert sum(sum(g["supports"]) for g in got) > 1500
    rc, text = L.cli_run("count", argv[:-1])
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext
    rc, text = L.cli_run("solve", argv)
    assert rc == 0 and ob.solve_text_close(text, otext)


def test_many_block_reads_and_the_ingest_block_limit(tmp_path, monkeypatch):
    """reads with up to 16 kept blocks go through the device ingest and the cleanup kernel like the
    oracle's; a read that keeps more separate blocks is refused loudly (LSQ_E_RANGE), not dropped"""
    import golden_inputs as gi
    exons = [(1000 + 40 * i, 1000 + 40 * i + 25) for i in range(24)]       # 24 exons of 25 bp, 15 bp introns
    iv = gi.interval_line("M.a", "c1", "+", exons) + gi.interval_line("M.b", "c1", "+", exons[:3] + exons[5:])
    _write(tmp_path / "m.interval", iv)
    _write(tmp_path / "m.map", "M\tM.a\nM\tM.b\n")
    lines = ["AlignmentBlocks"]
    for nb in (3, 5, 9, 12, 16):
        for start in (0, 2, 4):
            bl = [(s + 1, e) for s, e in exons[start:start + nb]]
            bl[0] = (bl[0][0] + 3, bl[0][1])
            lines.append(gi.mrf_line("c1", "+", bl).rstrip("\n"))
    # the same blocks listed right-to-left: touching ones stay separate, order does not matter otherwise
    lines.append(gi.mrf_line("c1", "+", list(reversed(exons[2:9]))).rstrip("\n"))
    _write(tmp_path / "m.mrf", "\n".join(lines) + "\n")
    monkeypatch.chdir(tmp_path)
    argv = ["0", "m", "./", "LH_GENE_TXT", "m.interval", "UCSC_GENE2ISOFORM", "m.map", "0", "10", "MRF_SINGLE", "SHORT_READ", "100", "m.mrf"]
    # 24 segments + 2 isoforms: a generic bucket
    rc, text = L.cli_run("count", argv)
    orc, otext, _ = ob.run("count", argv)
    assert (rc, text) == (orc, otext) and rc == 0 and text.split("\t")[1] != "0"
    rc, text = L.cli_run("solve", argv + ["5000"])
    orc, otext, _ = ob.run("solve", argv + ["5000"])
    assert rc == orc == 0 and ob.solve_text_close(text, otext)
    # 17 separate kept blocks
    _write(tmp_path / "big.mrf", "AlignmentBlocks\n" + gi.mrf_line("c1", "+", exons[:17]))
    a = L.Annotation("m.interval", "m.map")
    ev = L.Events(a, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    with pytest.raises(L.LsqError) as ei:
        ctx.upload_reads(0, L.Reads.from_mrf("big.mrf", ev))
    assert ei.value.status == -5
    ctx.close()


def test_long_introns_wide_bins_and_a_hot_spot(tmp_path):
    """spans of megabases: the bucket's coordinate bins are far wider than the ingest's per-bin sort
    handles (the reads of a bin stay in arrival order, so neighbouring lanes see different cells),
    plus one exon that takes most of the reads"""
    import random
    import golden_inputs as gi
    rng = random.Random(77)
    R = 60
    iv, mp, genes = [], [], []
    p = 1000
    for g in range(8):
        ex = [(p, p + 220), (p + 1_500_000, p + 1_500_180), (p + 3_000_000, p + 3_000_260)]
        forms = [ex, [ex[0], ex[2]]]
        for k, f in enumerate(forms):
            iv.append(gi_line("L%d.%d" % (g, k), "c1", "+", f))
            mp.append("L%d\tL%d.%d\n" % (g, g, k))
        genes.append(forms)
        p += 400_000 if g % 2 else 3_400_000       # every other gene starts inside the previous one's span
    reads = []
    for n in range(40000):
        forms = genes[0] if rng.random() < 0.6 else genes[rng.randrange(len(genes))]
        f = forms[rng.randrange(2)]
        tlen = sum(e - s for s, e in f)
        ln = rng.choice([R, R, R, 25])
        t0 = rng.randint(0, 150) if rng.random() < 0.5 else rng.randint(0, tlen - ln)
        reads.append(gi.mrf_line("c1", "+", gi.transcript_blocks(f, min(t0, tlen - ln), ln)))
    _write(tmp_path / "l.interval", "".join(iv))
    _write(tmp_path / "l.map", "".join(mp))
    _write(tmp_path / "l.mrf", "AlignmentBlocks\n" + "".join(reads))
    argv = ["0", "l", "./", "LH_GENE_TXT", str(tmp_path / "l.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "l.map"),
            "0", "100000", "MRF_SINGLE", "SHORT_READ", str(R), str(tmp_path / "l.mrf"), str(40000 * R)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    got = gpu_exact(argv)
    compare_exact(got, exact, "long introns")
    assert sum(sum(g["supports"]) for g in got) > 30000
    rc, text = L.cli_run("count", argv[:-1])
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext


@pytest.mark.parametrize("block", range(4))
def test_adversarial_inputs_many_seeds_vs_oracle(block, tmp_path):
    """the generator behind the wild_* golden sets (overlapping isoform structures on a tiny range,
    odd names and strands, shuffled blocks, ARS = 0) on seeds the reference never saw: the GPU path
    against the oracle, exact integers, theta to 1e-6, and the printed count table"""
    import golden_inputs as gi
    for seed in range(200 + 12 * block, 200 + 12 * (block + 1)):
        d = tmp_path / ("s%d" % seed)
        d.mkdir()
        info = gi.write_wild_case(str(d), "w", seed)
        argv = ["0", "w", "./", "LH_GENE_TXT", str(d / "w.interval"), "UCSC_GENE2ISOFORM", str(d / "w.map"), "0", "1000",
                "MRF_SINGLE", "SHORT_READ" if seed % 3 else "MEDIUM_READ", str(info["R"]), str(d / "w.mrf"), str(info["total_read_bases"])]
        rc, otext, exact = ob.run("solve", argv)
        assert rc == 0, seed
        compare_exact(gpu_exact(argv), exact, "wild seed %d" % seed)
        rc, text = L.cli_run("count", argv[:-1])
        rc2, ctext, _ = ob.run("count", argv[:-1])
        assert rc == rc2 == 0 and text == ctext, seed


@pytest.mark.parametrize("seed", range(300, 308))
def test_event_shaped_inputs_more_seeds_vs_oracle(seed, tmp_path):
    """the generator behind the events_s* golden sets (all eight event types, overlaps, 15 % off-target
    reads, Zipf depth on even seeds) on seeds and read lengths the golden sets do not have"""
    import golden_inputs as gi
    R = [40, 60, 90, 120][seed % 4]
    info = gi.write_events_case(str(tmp_path), "ev", seed=seed, n_events=60, n_reads=4000, R=R, n_chrom=2, zipf=(seed % 2 == 0))
    argv = ["0", "ev", "./", "LH_GENE_TXT", str(tmp_path / "ev.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "ev.map"), "0", "1000",
            "MRF_SINGLE", "SHORT_READ" if seed % 3 else "MEDIUM_READ", str(R), str(tmp_path / "ev.mrf"), str(info["total_read_bases"])]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    compare_exact(gpu_exact(argv), exact, "events seed %d" % seed)
    rc, text = L.cli_run("solve", argv)
    assert rc == 0 and ob.solve_text_close(text, otext)


def _tables(ev, reads, options=()):
    ctx = L.Context(0)
    for k, v in options:
        ctx.set_option(k, v)
    ctx.upload_events(ev)
    ctx.upload_reads(0, reads)
    ctx.count(); ctx.solve()
    out = [x.copy() for x in ctx.counts()] + [x.copy() for x in ctx.solution()], ctx.pool_format(0), ctx.count_status()
    ctx.close()
    return out


@pytest.mark.parametrize("zipf", [False, True], ids=["even_depth", "zipf_depth"])
def test_compact_and_wide_pool_records_give_the_same_tables(zipf, tmp_path):
    """lsq_reads_pool_format: the same reads as 4-byte compact block records and as 8-byte wide ones -- every count,
    base sum, theta, log-likelihood, iteration count and flag equal; also with the recount over every read (the
    one-lane-per-read kernel unpacks the records on its own) and with a tiny exception list (overflow -> recount)"""
    spec = L.SynthSpec(61, 3000, 700000, 100, 3, L.EVENT_TYPES, zipf)
    L.synth_write(spec, str(tmp_path), "f", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "f.interval"), str(tmp_path / "f.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    reads = L.Reads.synthetic(spec, ev)
    wide, fmt_w, _ = _tables(ev, reads, [("compact_pools", 0)])
    comp, fmt_c, _ = _tables(ev, reads)
    assert fmt_w[0] is False and fmt_c[0] is True
    assert fmt_c[2] == fmt_w[2] and fmt_c[2][0] > 100000 and fmt_c[2][1] > 10000          # the same reads in the same pools
    assert fmt_c[1] < 0.52 * fmt_w[1]                                                     # in half the bytes
    for a, b, what in zip(wide, comp, ("counts", "bases", "theta", "logll", "iters", "flags")):
        assert np.array_equal(a, b, equal_nan=True), what
    assert int(wide[0].sum()) > 500000
    rec, _, st = _tables(ev, reads, [("recount_every_read", 1)])
    assert st[1][0] == 1
    ovf, _, st = _tables(ev, reads, [("exception_capacity", 2)])
    for other, what in ((rec, "recount"), (ovf, "overflow")):
        for a, b in zip(wide[:2], other[:2]):
            assert np.array_equal(a, b), what


def test_reads_that_do_not_fit_compact_records(tmp_path):
    """Compact records hold blocks shorter than 1 024 bases that start within 2 Mi bases of their bucket's first base
    (and second blocks within 4 Mi of the first).  Reads beyond that -- long blocks, a 5 Mb intron, a gene 3 Mb long --
    are kept with the many-block reads and counted all the same; when more than 1 in 16 reads is like that the whole
    read file falls back to wide records.  Both cases against the oracle."""
    import random
    import golden_inputs as gi
    rng = random.Random(9)
    iv, mp = [], []
    # a gene with a 5 Mb intron, a gene with one 3 Mb exon and a far second one, a gene of long exons, and ordinary ones
    genes = {"FAR": [[(10_000, 10_400), (5_010_400, 5_010_900)], [(10_000, 10_400), (12_000, 12_300), (5_010_400, 5_010_900)]],
             "WIDE": [[(6_000_000, 9_000_000), (9_100_000, 9_100_500)], [(6_000_000, 6_000_600), (9_100_000, 9_100_500)]],
             "LONG": [[(12_000_000, 12_003_000), (12_004_000, 12_006_500)], [(12_000_000, 12_003_000)]]}
    for g in range(30):
        s = 13_000_000 + 9_000 * g
        genes["N%d" % g] = [[(s, s + 200), (s + 500, s + 650), (s + 1200, s + 1500)], [(s, s + 200), (s + 1200, s + 1500)]]
    for name, forms in genes.items():
        for k, ex in enumerate(forms):
            iv.append(gi.interval_line("%s.%d" % (name, k), "c1", "+", ex))
            mp.append("%s\t%s.%d\n" % (name, name, k))
    _write(tmp_path / "g.interval", "".join(iv))
    _write(tmp_path / "g.map", "".join(mp))

    def read_set(path, n, long_share, special_share):
        lines = ["AlignmentBlocks\n"]
        names = list(genes)
        for _ in range(n):
            name = names[rng.randrange(3)] if rng.random() < special_share else names[3 + rng.randrange(30)]
            f = genes[name][rng.randrange(len(genes[name]))]
            tlen = sum(e - s for s, e in f)
            ln = rng.choice([1024, 1500, 2200]) if rng.random() < long_share else rng.choice([100, 100, 100, 76, 1023])
            ln = min(ln, tlen)
            # near an exon junction half of the time, so that two-block reads are common
            t0 = rng.randint(0, tlen - ln)
            if rng.random() < 0.5 and len(f) > 1:
                t0 = max(0, min(tlen - ln, (f[0][1] - f[0][0]) - rng.randint(1, ln - 1)))
            lines.append(gi.mrf_line("c1", "+", gi.transcript_blocks(f, t0, ln)))
        _write(path, "".join(lines))

    for tag, long_share, special_share, want_compact in (("few", 0.02, 0.1, True), ("many", 0.6, 0.6, False)):
        read_set(tmp_path / ("%s.mrf" % tag), 30000, long_share, special_share)
        argv = ["0", "g", "./", "LH_GENE_TXT", str(tmp_path / "g.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "g.map"), "0", "1000",
                "MRF_SINGLE", "MEDIUM_READ", "100", str(tmp_path / ("%s.mrf" % tag)), "3000000"]
        rc, otext, exact = ob.run("solve", argv)
        assert rc == 0
        compare_exact(gpu_exact(argv), exact, tag)
        a = L.Annotation(argv[4], argv[6])
        ev = L.Events(a, ("MEDIUM_READ",), (100,))
        reads = L.Reads.from_mrf(argv[12], ev)
        tabs, fmt, _ = _tables(ev, reads)
        assert fmt[0] is want_compact, (tag, fmt)
        if want_compact:
            assert fmt[2][2] > 300              # misfits (and the few many-block reads) sit in the third pool
        wide, fmt_w, _ = _tables(ev, reads, [("compact_pools", 0)])
        assert fmt_w[0] is False and sum(fmt_w[2]) == sum(fmt[2])
        for x, y, what in zip(wide, tabs, ("counts", "bases", "theta", "logll", "iters", "flags")):
            assert np.array_equal(x, y, equal_nan=True), (tag, what)
        rc, text = L.cli_run("count", argv[:-1])
        rc2, ctext, _ = ob.run("count", argv[:-1])
        assert rc == rc2 == 0 and text == ctext, tag


@pytest.mark.parametrize("compact", [1, 0], ids=["compact_records", "wide_records"])
def test_pools_are_laid_out_by_cell_and_by_junction(compact, tmp_path):
    """the ingest's layout, which the count kernel's speed rests on (its results do not): every aligned group of eight
    one-block records starts in one cell of the annotation, every aligned pair of two-block records of a junction group
    crosses one junction, padding is empty records only, and the records that are not padding are the reads counted"""
    import ctypes as C
    spec = L.SynthSpec(71, 2500, 500000, 100, 3, L.EVENT_TYPES, True)
    L.synth_write(spec, str(tmp_path), "q", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "q.interval"), str(tmp_path / "q.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.set_option("compact_pools", compact)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
    out = (C.c_ulonglong * 6)()
    L.lib.lsq_debug_check_pool_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    L.lib.lsq_debug_check_pool_layout.restype = C.c_int
    assert L.lib.lsq_debug_check_pool_layout(ctx.h, 0, out) == 0, L.lib.lsq_last_error()
    n1, pad1, mixed1, n2, pad2, mixed2 = [int(x) for x in out]
    fmt = ctx.pool_format(0)
    assert fmt[0] is bool(compact)
    assert (n1 - pad1, n2 - pad2) == fmt[2][:2] and n1 - pad1 > 200000 and n2 - pad2 > 30000
    assert mixed1 == 0 and mixed2 == 0
    assert 0 < pad1 < 0.2 * n1 and 0 < pad2 < 0.6 * n2           # a few records per cell / junction
    assert ctx.pooled(0) == sum(fmt[2]) == ctx.retained(0)
    assert fmt[2][0] + 2 * fmt[2][1] + 3 * fmt[2][2] <= ctx.pooled_blocks(0) <= ctx.retained_blocks(0)
    ctx.close()


def test_counts_leave_and_enter_the_context_on_the_device(tmp_path):
    """lsq_counts_export_device / lsq_counts_import_device: what a read-sharded job sums over its ranks without the host.
    A buffer exported, doubled and imported gives twice the tables, and the EM of proportional counts the same theta
    (read.h:592-618 is homogeneous in the counts)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # the runtime the library itself is linked to
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    spec = L.SynthSpec(23, 400, 60000, 100, 2, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "d", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "d.interval"), str(tmp_path / "d.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
    ctx.count(); ctx.solve()
    cnt, bases = [x.copy() for x in ctx.counts()]
    theta = ctx.solution()[0].copy()
    n = ctx.counts_device_words()
    assert n > 0
    d_buf = C.c_void_p()
    assert L.lib.lsq_device_alloc(ctx.h, 8 * n, C.byref(d_buf)) == 0
    ctx.export_counts_device(d_buf.value)
    host = np.zeros(n, dtype=np.uint64)
    assert L.lib.lsq_device_read(ctx.h, host.ctypes.data_as(C.c_void_p), d_buf, 8 * n) == 0      # waits for the context's streams
    assert int(host.sum()) == int(cnt.sum()) + int(bases.sum())
    host *= 2
    assert hip.hipMemcpy(d_buf, host.ctypes.data_as(C.c_void_p), 8 * n, 1) == 0                    # hipMemcpyHostToDevice
    ctx.import_counts_device(d_buf.value)
    cnt2, bases2 = ctx.counts()
    assert np.array_equal(cnt2, 2 * cnt) and np.array_equal(bases2, 2 * bases)
    ctx.solve()
    assert np.allclose(ctx.solution()[0], theta, rtol=0, atol=1e-6)
    L.lib.lsq_device_free(ctx.h, d_buf)
    ctx.close()


def test_steps_submitted_back_to_back_with_changing_reads(tmp_path):
    """The step pipeline (two streams, two counter sets): steps are only submitted, with another
    read set uploaded in between and the hand-off going to a buffer per step; every step's tables must be the ones
    the same reads give in a synchronous run, and the host getters must return the last step's."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # the runtime the library itself is linked to
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]

    class DevArr:
        def __init__(self, n, dtype, fill=0):
            self.n, self.dtype = n, np.dtype(dtype)
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), max(n, 1) * self.dtype.itemsize) == 0
            self.p = p.value
            assert hip.hipMemset(self.p, fill, max(n, 1) * self.dtype.itemsize) == 0     # synchronous

        def get(self):
            out = np.empty(self.n, self.dtype)
            assert hip.hipMemcpy(out.ctypes.data, self.p, self.n * self.dtype.itemsize, 2) == 0
            return out

        def free(self):
            hip.hipFree(self.p)
    specs = [L.SynthSpec(40 + i, 2500, 300000 + 50000 * i, 100, 3, L.EVENT_TYPES) for i in range(3)]
    L.synth_write(specs[0], str(tmp_path), "p", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "p.interval"), str(tmp_path / "p.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    # the annotation is that of specs[0]; the other read sets lie over other synthetic annotations of the same
    # chromosomes (whatever falls on these events is counted)
    reads = [L.Reads.synthetic(sp, ev) for sp in specs]
    ctx = L.Context(0)
    ctx.upload_events(ev)
    n_cls = int(L.lib.lsq_results_num_classes(ctx.h))
    # synchronous reference per read set
    ref = []
    for r in reads[:1]:
        ctx.upload_reads(0, r)
        ctx.count(); ctx.solve()
        cnt, bases = ctx.counts()
        theta, ll, iters, flags = ctx.solution()
        t_c, t_t, t_l = DevArr(n_cls, np.int64), DevArr(ev.total_isoforms, np.float64), DevArr(len(ev), np.float64)
        ctx.copy_results_device(t_c.p, t_t.p, t_l.p); ctx.synchronize()
        ref.append((t_c.get(), t_t.get(), t_l.get(), cnt.copy(), theta.copy()))
    assert int(ref[0][0].sum()) == int(ref[0][3].sum()) > 0
    # nine steps in a row, nothing waited for; count-only steps and repeated solves mixed in
    outs = []
    for k in range(9):
        ctx.count()
        if k % 4 != 3:
            ctx.solve()
        if k % 5 == 4:
            ctx.solve()
        t_c, t_t, t_l = DevArr(n_cls, np.int64, 0xFF), DevArr(ev.total_isoforms, np.float64, 0xFF), DevArr(len(ev), np.float64, 0xFF)
        ctx.copy_results_device(t_c.p, t_t.p if k % 4 != 3 else None, t_l.p if k % 4 != 3 else None)
        outs.append((k, t_c, t_t, t_l))
    ctx.synchronize()
    for k, t_c, t_t, t_l in outs:
        assert np.array_equal(t_c.get(), ref[0][0]), "step %d" % k
        if k % 4 != 3:
            assert np.array_equal(t_t.get(), ref[0][1]) and np.array_equal(t_l.get(), ref[0][2]), "step %d" % k
    cnt, _ = ctx.counts()
    assert np.array_equal(cnt, ref[0][3])
    # timing getters: refused for runs without events, fine after set_timing
    with pytest.raises(L.LsqError):
        ctx.timing()
    ctx.set_timing(True)
    ctx.count(); ctx.solve()
    c_ms, s_ms = ctx.timing()
    assert c_ms > 0 and s_ms > 0 and ctx.fast_kernel_ms() > 0
    ctx.set_timing(False)
    # other read sets through the same context, again without waiting in between
    seen = []
    for i in (1, 2, 1):
        ctx.upload_reads(0, reads[i])
        ctx.count(); ctx.solve()
        t_c = DevArr(n_cls, np.int64)
        ctx.copy_results_device(t_c.p, None, None)
        ctx.count()        # a second count of the same reads while the first one's solve may still run
        seen.append((i, t_c))
    ctx.synchronize()
    sync = {}
    for i in (1, 2):
        ctx.upload_reads(0, reads[i]); ctx.count(); ctx.synchronize()
        sync[i] = ctx.counts()[0].copy()
    for i, t_c in seen:
        assert int(t_c.get().sum()) == int(sync[i].sum()), "read set %d" % i
    assert int(sync[1].sum()) != int(sync[2].sum())
    ctx.close()


def test_em_numbers_do_not_depend_on_which_events_share_a_wave(tmp_path):
    """theta, log-likelihood and iteration counts of an event are functions of its own counts: any placement of the
    events in the EM grid (16 per wave) gives bit-identical numbers"""
    import ctypes as C
    spec = L.SynthSpec(77, 4000, 600000, 100, 4, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "w", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "w.interval"), str(tmp_path / "w.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
    ctx.count(); ctx.solve()
    ref = [x.copy() for x in ctx.solution()]
    assert ref[2].max() > 20              # some slow events among fast ones
    n = len(ev)
    L.lib.lsq_debug_set_em_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
    rng = np.random.default_rng(5)
    for trial in range(3):
        order = rng.permutation(n).astype(np.uint32)
        if trial == 2:                    # and with holes in the grid
            holes = np.full(n // 3, 0xFFFFFFFF, np.uint32)
            order = rng.permutation(np.concatenate([order, holes]))
        order = np.concatenate([order, np.full((-len(order)) % 16, 0xFFFFFFFF, np.uint32)])
        assert L.lib.lsq_debug_set_em_order(ctx.h, order.ctypes.data, len(order), len(order)) == 0
        ctx.solve()
        got = ctx.solution()
        for a, b, what in zip(ref, got, ("theta", "logll", "iters", "flags")):
            assert np.array_equal(a, b, equal_nan=True), "%s differs with placement %d" % (what, trial)
    ctx.close()


def test_em_regrouping_by_earlier_iteration_counts_keeps_every_number(tmp_path):
    """option em_regroup (on by default): a lane's solves place the events by the iteration counts of one of its
    earlier solves, and (em_flat_min_events) solve the fast ones one lane per event, the slow ones four lanes per event.  The placement learned on one read set is then used for another read set; every event must
    still be solved (none dropped from the grid, none twice) with the numbers a fresh context without the option
    gives, and the reference-order replay of flagged events still applies."""
    specs = [L.SynthSpec(90 + i, 3000, 400000 + 150000 * i, 100, 3, L.EVENT_TYPES, bool(i)) for i in range(2)]
    L.synth_write(specs[0], str(tmp_path), "g", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "g.interval"), str(tmp_path / "g.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    reads = [L.Reads.synthetic(sp, ev) for sp in specs]
    plain = L.Context(0)
    plain.set_option("em_regroup", 0)
    plain.upload_events(ev)
    want = []
    for r in reads:
        plain.upload_reads(0, r); plain.count(); plain.solve()
        want.append([x.copy() for x in plain.solution()])
    plain.close()
    assert not np.array_equal(want[0][2], want[1][2])          # the two read sets converge differently
    ctx = L.Context(0)
    ctx.set_option("em_flat_min_events", 0)        # the one-lane-per-event form for the fast events, on this small job too
    ctx.upload_events(ev)
    ctx.upload_reads(0, reads[0])
    for k in range(40):                    # both lanes learn a placement, and refresh it at least once
        ctx.count(); ctx.solve()
        if k in (0, 1, 2, 3, 17, 18, 39):
            for a, b, what in zip(want[0], ctx.solution(), ("theta", "logll", "iters", "flags")):
                assert np.array_equal(a, b, equal_nan=True), "%s differs at step %d" % (what, k)
    ctx.upload_reads(0, reads[1])          # the learned placement stays; the reads are others
    for k in range(3):
        ctx.count(); ctx.solve()
        for a, b, what in zip(want[1], ctx.solution(), ("theta", "logll", "iters", "flags")):
            assert np.array_equal(a, b, equal_nan=True), "%s differs on the second read set, step %d" % (what, k)
    ctx.close()


@pytest.mark.parametrize("name", ["toy", "multi_method", "events_s1", "readfmts", "formats"])
def test_example_host_prints_the_reference_solve_table(name, tmp_path, monkeypatch):
    """examples/solve_host.c (plain C against include/lesseq_hip.h, the binding INTEGRATION.md describes) through the
    fine-grained entry points: its table is the reference's golden solve table"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lesseq_amd", "bin", "solve_host")
    assert os.path.exists(exe), "lesseq_amd/bin/solve_host is built by lesseq_amd/csrc/Makefile"
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    n = 0
    for tool, r in runs(c):
        if tool != "solve" or r["exit"] != 0:
            continue
        a = r["argv"]                      # [log, proj, prefix, iso_fmt, iso, g2i_fmt, g2i, begin, end, groups...]
        p = subprocess.run([exe] + a[3:], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, (name, a, p.stderr)
        assert ob.solve_text_close(p.stdout, open(os.path.join(d, r["stdout"])).read()), (name, a)
        n += 1
    assert n >= 1


@pytest.mark.parametrize("name", ["toy", "events_s1", "events_s2", "multi_method", "wild_s11", "wild_s12", "wild_s13", "quirks"])
def test_fisher_information_and_variances_vs_oracle(name, tmp_path, monkeypatch):
    """lsq_fim (fim.h / linalg.h -- dead code in the reference, parity unpinned): the HIP path's class-grouped sum
    against the oracle's restatement of bruteforce_fim / ofim / the two variance estimates, on the golden inputs'
    solve runs: matrices to 1e-9 relative, variances likewise where the matrix is well conditioned"""
    monkeypatch.setenv("LSQO_FIM", "1")
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    checked = 0
    for tool, r in runs(c):
        if tool != "solve" or r["exit"] != 0:
            continue
        rc, _, exact = ob.run("solve", r["argv"])
        assert rc == 0
        got = gpu_exact(r["argv"], want_fim=True)
        assert len(got) == len(exact)
        for g, e in zip(got, exact):
            assert g["gname"] == e["gname"] and g["K"] == e["K"]
            if g["K"] > 6 or "fim" not in e:
                continue
            if any(abs(a - b) > 1e-9 * max(abs(a), abs(b)) for a, b in zip(g["theta"], e["theta"])):
                continue                  # (an event whose EM was flagged: another theta, another matrix)
            for m in range(len(e["fim"])):
                A, B = np.array(g["fim"][m], float).reshape(-1), np.array(e["fim"][m], float).reshape(-1)
                scale = max(np.abs(B).max(), 1e-300) if B.size else 1.0
                assert np.all(np.abs(A - B) <= 1e-9 * scale), (name, g["gname"], m, A, B)
                for w in (0, 1):
                    a, b = g["fim_var"][m][w], e["fim_var"][m][w]
                    if np.isfinite(b) and abs(b) < 1e12:
                        assert abs(a - b) <= 1e-7 * max(abs(b), 1e-300), (name, g["gname"], m, w, a, b)
                    else:
                        assert not np.isfinite(a) or abs(a) >= 1e11 or (not np.isfinite(b)), (name, g["gname"], m, w, a, b)
                checked += 1
    assert checked >= 1


def test_solve_fim_extension_keeps_the_table_and_appends_the_matrices(tmp_path, monkeypatch):
    """`solve ... --fim` (an extension: the reference's fim.h is dead code): the table is the one without the flag, the
    `#fim` lines behind it carry what lsq_results_fim returns; `count` does not know the flag"""
    monkeypatch.setenv("LSQO_FIM", "1")
    c, d = load_case("multi_method", tmp_path)
    monkeypatch.chdir(d)
    r = [x for t, x in runs(c) if t == "solve" and x["exit"] == 0][0]
    rc0, plain = L.cli_run("solve", r["argv"])
    rc1, with_fim = L.cli_run("solve", r["argv"] + ["--fim"])
    assert rc0 == rc1 == 0 and with_fim.startswith(plain)
    extra = with_fim[len(plain):].split("\n")[:-1]
    rc, _, exact = ob.run("solve", r["argv"])
    M = len(exact[0]["supports"])
    assert len(extra) == len(exact) * M and all(x.startswith("#fim\t") for x in extra)
    by_gene = {g["gname"]: g for g in exact}
    for line in extra:
        t = line.split("\t")
        g, m = by_gene[t[1]], int(t[2])
        D = g["K"] - 1
        vals = [float(x) for x in t[3:]]
        assert len(vals) == 2 + D * D
        ref = [g["fim_var"][m][0], g["fim_var"][m][1]] + [x for row in g["fim"][m] for x in row]
        for a, b in zip(vals, ref):
            if np.isfinite(b) and abs(b) < 1e12:
                assert abs(a - b) <= 1e-7 * max(abs(b), 1e-300), (line, ref)
    rc, text = L.cli_run("count", [x for i, x in enumerate(r["argv"]) if i < 9 or (i - 9) % 5 != 4] + ["--fim"])
    assert rc == 1                      # the reference's usage error: an incomplete read group


def test_recount_over_every_read_gives_the_same_tables(tmp_path):
    """when the exception list overflows, two kernels behind the exception pass zero the method's tables and count
    every read again, one lane per read; the option "recount_every_read" takes that path without an overflow: same
    integers, same theta, also when further steps were submitted behind the one that is fetched"""
    spec = L.SynthSpec(91, 3000, 400000, 100, 3, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "r", write_mrf=False)
    ann = L.Annotation(str(tmp_path / "r.interval"), str(tmp_path / "r.map"))
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
    ctx.count(); ctx.solve()
    cnt0, bases0 = [x.copy() for x in ctx.counts()]
    sol0 = [x.copy() for x in ctx.solution()]
    assert ctx.count_status()[1] == [0]
    ctx.set_option("recount_every_read", 1)
    for steps in (1, 3):
        for _ in range(steps):
            ctx.count(); ctx.solve()
        cnt1, bases1 = ctx.counts()
        sol1 = ctx.solution()
        assert ctx.count_status()[1] == [1]
        assert np.array_equal(cnt0, cnt1) and np.array_equal(bases0, bases1)
        assert np.array_equal(sol0[0], sol1[0]) and np.array_equal(sol0[1], sol1[1], equal_nan=True) and np.array_equal(sol0[2], sol1[2])
    ctx.close()


def test_exception_list_overflow_is_settled_on_the_device(tmp_path, monkeypatch):
    """an exception list that is too small (option "exception_capacity") overflows on an input full of reads that cover
    an event's span exactly (the strand/name order decides, count/count.cpp:64-85).  The recount runs on the result
    stream without the host looking: the oracle's tables come out of the executables, and the tables a pipelined loop
    takes through lsq_results_copy_device -- no fetch, no check in between -- are the complete ones at every step."""
    import torch
    import golden_inputs as gi
    info = gi.write_events_case(str(tmp_path), "x", seed=77, n_events=40, n_reads=3000, R=60, n_chrom=2)
    lines = open(tmp_path / "x.mrf").read().split("\n")
    extra = []
    for e in info["events"]:
        gs = min(f[0][0] for f in e["forms"]); ge = max(f[-1][1] for f in e["forms"])
        for strand in ("+", "-", e["strand"]):
            extra.append("%s:%s:%d:%d:1:%d" % (e["chrom"], strand, gs + 1, ge, ge - gs))
    with open(tmp_path / "x.mrf", "w") as f:
        f.write("\n".join(lines[:-1] + extra) + "\n")
    argv = ["0", "x", "./", "LH_GENE_TXT", str(tmp_path / "x.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "x.map"), "0", "1000",
            "MRF_SINGLE", "SHORT_READ", "60", str(tmp_path / "x.mrf")]
    orc, otext, _ = ob.run("count", argv)
    orc2, otext2, _ = ob.run("solve", argv + ["200000"])
    assert orc == 0 and orc2 == 0
    for cap in ("1", "7", None):
        if cap is None:
            monkeypatch.delenv("LSQ_OPTIONS")
        else:
            monkeypatch.setenv("LSQ_OPTIONS", "exception_capacity=" + cap)
        rc, text = L.cli_run("count", argv)
        assert rc == 0 and text == otext, cap
        rc, text = L.cli_run("solve", argv + ["200000"])
        assert rc == 0 and ob.solve_text_close(text, otext2), cap
    ann = L.Annotation(argv[4], argv[6], 0, 1000)
    ev = L.Events(ann, ("SHORT_READ",), (60,))
    reads = L.Reads.from_mrf(argv[12], ev)
    ref = L.Context(0)
    ref.upload_events(ev)
    ref.upload_reads(0, reads)
    ref.count(); ref.solve()
    ref_cnt = ref.counts()[0].copy()
    assert ref.count_status()[1] == [0] and ref.count_status()[0][0] > 7
    n_cls = int(L.lib.lsq_results_num_classes(ref.h))
    order = ref.device_order()
    ref.close()
    ctx = L.Context(0)
    ctx.set_option("exception_capacity", 7)
    ctx.upload_events(ev)
    ctx.upload_reads(0, reads)
    bufs = [torch.zeros(n_cls, dtype=torch.int64, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()             # torch's fills run on torch's stream, the hand-offs on the library's
    for b in bufs:                       # submitted back to back, each step hands its table to its own buffer
        ctx.count(); ctx.solve()
        ctx.copy_results_device(b.data_ptr(), None, None)
    ctx.synchronize()
    exc, rec = ctx.count_status()
    assert rec == [1] and exc[0] > 7
    off = ev.class_offsets()
    want = np.zeros(n_cls, np.int64)     # the fetched (output-order) counts in device order
    pos = 0
    for d in order:
        nc = off[d + 1] - off[d]
        want[pos:pos + nc] = ref_cnt[0, off[d]:off[d + 1]]
        pos += nc
    for b in bufs:
        assert np.array_equal(b.cpu().numpy(), want)
    ctx.close()


BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lesseq_amd", "bin")


@pytest.mark.parametrize("name", ["toy", "edge", "events_s1", "multi_method", "classify_mix", "errors"])
def test_executables_as_processes_match_reference_golden(name, tmp_path):
    """lesseq_amd/bin/{count,solve} spawned the way the reference's binaries are: stdout carries the table only (count byte
    for byte, solve as printed), the exit status is the reference's (0, 1 -- also for the statuses it reaches only after
    the reads were loaded: unknown read type, bad number inside an MRF line), stderr carries log lines with the
    reference's `[LOG date time LEVEL] ` prefix (jsc/util/log.hpp:64-70) and nothing else"""
    import re
    import subprocess
    c, d = load_case(name, tmp_path)
    n = 0
    for tool, r in runs(c):
        argv = list(r["argv"])
        argv[0] = "2"                       # log level 2: progress lines on stderr, never on stdout
        p = subprocess.run([os.path.join(BIN, tool)] + argv, cwd=d, capture_output=True, text=True)
        exp = open(os.path.join(d, r["stdout"])).read()
        assert p.returncode == r["exit"], (name, tool, argv, p.returncode, p.stderr)
        if tool == "count":
            assert p.stdout == exp, (name, argv)
        else:
            assert ob.solve_text_close(p.stdout, exp), (name, argv)
        lines = [ln for ln in p.stderr.split("\n") if ln]
        # (a usage error is one log entry of several lines, as in the reference: count/count.cpp:20-23)
        assert lines and all(re.match(r"^\[LOG \d{4}-\d{2}-\d{2} \d{2}:\d{2}:\d{2} [A-Z]+\d?\] ", ln) for ln in lines if ln.startswith("[") or ln is lines[0]), p.stderr
        if r["exit"] == 0:
            assert all(ln.startswith("[LOG ") for ln in lines) and any("oaded" in ln for ln in lines)
        n += 1
    assert n


def test_executable_device_selection_and_missing_device(tmp_path):
    """LSQ_DEVICE picks the GPU; a device that does not exist is a start-up failure of this build (exit 3, an error line),
    not a silent fallback: there is no CPU implementation behind the executables"""
    import subprocess
    c, d = load_case("toy", tmp_path)
    r = c["count"][0]
    env = dict(os.environ, LSQ_DEVICE="0")
    p = subprocess.run([os.path.join(BIN, "count")] + r["argv"], cwd=d, capture_output=True, text=True, env=env)
    assert p.returncode == 0 and p.stdout == open(os.path.join(d, r["stdout"])).read()
    env["LSQ_DEVICE"] = "63"
    p = subprocess.run([os.path.join(BIN, "count")] + r["argv"], cwd=d, capture_output=True, text=True, env=env)
    assert p.returncode == 3 and p.stdout == "" and "ERROR" in p.stderr


@pytest.mark.parametrize("name,env", [("events_s1", dict(LSQ_GPUS="2", LSQ_DEVICES="0,0", LSQ_GATHER="host")),
                                      ("multi_method", dict(LSQ_GPUS="3", LSQ_DEVICES="0,0,0", LSQ_GATHER="host")),
                                      ("readfmts", dict(LSQ_GPUS="2", LSQ_DEVICES="0,0", LSQ_GATHER="host")),
                                      ("events_s2", dict(LSQ_GPUS="1", LSQ_GATHER="rccl"))])
def test_executables_run_one_job_over_several_gpu_slices(name, env, tmp_path):
    """LSQ_GPUS=N: the executables cut the sorted gene list into N slices of equal read weight (first count = pre-pass), a host
    thread per slice compiles the whole range, ingests, counts, solves and packs its slice, the blocks are gathered and the
    table printed is the single-GPU one -- i.e. the reference's golden table.  The test box has one GPU: every "GPU" is
    device 0 and the blocks go through host memory (LSQ_GATHER=host); the RCCL all-gather itself (liblesseq_rccl.so,
    ncclCommInitAll + ncclAllGather on the result stream) runs here with a single slice."""
    import subprocess
    c, d = load_case(name, tmp_path)
    n = 0
    for tool, r in runs(c):
        if r["exit"] != 0:
            continue
        p = subprocess.run([os.path.join(BIN, tool)] + r["argv"], cwd=d, capture_output=True, text=True, env=dict(os.environ, **env))
        exp = open(os.path.join(d, r["stdout"])).read()
        assert p.returncode == 0, (name, tool, p.stderr)
        if tool == "count":
            assert p.stdout == exp, (name, r["argv"])
        else:
            assert ob.solve_text_close(p.stdout, exp), (name, r["argv"])
        n += 1
    assert n


def test_reads_on_every_threshold_of_the_streaming_loops_vs_oracle(tmp_path):
    """The streaming loops of the count kernel settle reads by where they END (lsq_count.hip, round 3): inside the owner's
    segment, in the segment abutting it, past an end nothing abuts (two-owner cells: valid only if 50 x the overhang < the
    read), two-block reads through junctions of abutting runs, and two-block reads that cross no junction (nothing when
    50 |block 1| <= 49 total).  Hand-made events -- an A5SS-like one with abutting segments, an SE-like one, and a third that
    overlaps both (two-owner cells with and without abutting neighbours) -- and reads enumerated on, one base before and one
    base after every such boundary, in every block-2 variant, with lengths that put the 98 % rule on and beside equality.
    Exact integers against the oracle; compact and wide records."""
    import golden_inputs as gi
    c = "chr1"
    #   event A (long form splits its first exon in two abutting segments): segments [1000,1100) [1100,1160) [1400,1500)
    #   event B (skipped exon):                                             segments [2000,2100) [2300,2360) [2600,2700)
    #   event C overlaps A's last segment and B's first (its own segments abut once): [1450,1520) [1520,1600) [2050,2130)
    iv = (gi.interval_line("A.l", c, "+", [(1000, 1160), (1400, 1500)]) + gi.interval_line("A.s", c, "+", [(1000, 1100), (1400, 1500)])
          + gi.interval_line("B.i", c, "+", [(2000, 2100), (2300, 2360), (2600, 2700)]) + gi.interval_line("B.k", c, "+", [(2000, 2100), (2600, 2700)])
          + gi.interval_line("C.l", c, "-", [(1450, 1600), (2050, 2130)]) + gi.interval_line("C.s", c, "-", [(1450, 1520), (2050, 2130)]))
    mp = "A\tA.l\nA\tA.s\nB\tB.i\nB\tB.k\nC\tC.l\nC\tC.s\n"
    ends = sorted({1100, 1160, 1500, 1520, 1600, 2100, 2130, 2360, 2700})
    starts = sorted({1000, 1100, 1400, 1450, 1520, 2000, 2050, 2300, 2600})
    reads = []
    # one-block reads: every start region x every end -1 / 0 / +1 / +2 / +30, several lengths
    for s0 in (1001, 1040, 1099, 1101, 1130, 1401, 1449, 1451, 1470, 1499, 1501, 1530, 2001, 2049, 2051, 2080, 2099, 2101, 2120, 2301, 2601):
        for e in ends:
            for d in (-1, 0, 1, 2, 30):
                if e + d > s0 and e + d - s0 < 900:
                    reads.append(gi.mrf_line(c, "+" if (s0 + e) % 3 else "-", [(s0, e + d)]))
    # two-block reads: block 1 ends on / beside every segment end, block 2 starts on / beside every later segment start or
    # touches block 1; block 2 lengths put 50 |block 1| against 49 total on and around equality
    for s0 in (1005, 1030, 1060, 1099, 1105, 1140, 1405, 1455, 1470, 1490, 1525, 1560, 2005, 2030, 2055, 2070, 2090, 2105, 2305, 2340):
        for e in ends:
            for d1 in (-1, 0, 1):
                y = e + d1
                if not (0 < y - s0 < 300):
                    continue
                l1 = y - s0
                for z0 in starts:
                    for dz in (0, 1):
                        z = z0 + dz
                        if z < y:
                            continue
                        for l2 in sorted({1, 2, max(1, l1 // 49), l1 // 49 + 1, 20, 60, 100, 131}):
                            reads.append(gi.mrf_line(c, "+" if (s0 + z + l2) % 5 else "-", [(s0, y), (z, z + l2)]))
                reads.append(gi.mrf_line(c, "+", [(s0, y), (y, y + 7)]))          # touching blocks
    _write(tmp_path / "t.interval", iv)
    _write(tmp_path / "t.map", mp)
    _write(tmp_path / "t.mrf", "AlignmentBlocks\n" + "".join(reads))
    assert len(reads) > 6000
    for R in (40, 100):
        argv = ["0", "t", "./", "LH_GENE_TXT", str(tmp_path / "t.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "t.map"), "0", "100",
                "MRF_SINGLE", "SHORT_READ", str(R), str(tmp_path / "t.mrf"), str(len(reads) * R)]
        rc, otext, exact = ob.run("solve", argv)
        assert rc == 0
        for opts in ({}, {"compact_pools": 0}, {"reads_per_look": 4, "workgroups_per_cu": 6}, {"recount_every_read": 1}):
            compare_exact(gpu_exact(argv, options=opts), exact, "thresholds R=%d %s" % (R, opts))
        assert sum(sum(g["supports"]) for g in gpu_exact(argv)) > 3000
        rc1, text = L.cli_run("count", argv[:-1])
        rc2, ctext, _ = ob.run("count", argv[:-1])
        assert rc1 == rc2 == 0 and text == ctext


@pytest.mark.parametrize("seed", [501, 502, 503, 504, 505, 506, 507, 508])
def test_em_closed_form_stops_where_the_reference_stops(seed, tmp_path):
    """option em_closed_form: a two-isoform event with one read file runs six ordinary EM iterations and finishes in the
    closed form of its EM map (a Moebius map of theta_0; the stopping iteration by search, lsq_em.hip) -- iteration count equal
    to the oracle's for EVERY event, theta and log-likelihood within 1e-6 (guard-band events replayed as ever).  Event-shaped
    inputs with skewed depth (few reads per event: fixed points on the boundary, slow events) and the golden toy."""
    import golden_inputs as gi
    R = [40, 60, 90, 120][seed % 4]
    info = gi.write_events_case(str(tmp_path), "ev", seed=seed, n_events=400, n_reads=30000 if seed % 2 else 3000, R=R, n_chrom=2, zipf=(seed % 2 == 0))
    argv = ["0", "ev", "./", "LH_GENE_TXT", str(tmp_path / "ev.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "ev.map"), "0", "1000",
            "MRF_SINGLE", "SHORT_READ" if seed % 3 else "MEDIUM_READ", str(R), str(tmp_path / "ev.mrf"), str(info["total_read_bases"])]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    got = gpu_exact(argv, options={"em_closed_form": 1})
    compare_exact(got, exact, "closed form, seed %d" % seed)
    assert max(g["iters"] for g in got) > 6          # some event went past the head
    plain = gpu_exact(argv)
    for a, b in zip(got, plain):
        assert a["iters"] == b["iters"] or (a["flags"] & 4) or (b["flags"] & 4), a["gname"]


@pytest.mark.parametrize("name,env", [("events_s1", dict(LSQ_GPUS="3", LSQ_DEVICES="0,0,0", LSQ_GATHER="host", LSQ_SHARD="reads")),
                                      ("multi_method", dict(LSQ_GPUS="2", LSQ_DEVICES="0,0", LSQ_GATHER="host", LSQ_SHARD="reads")),
                                      ("edge", dict(LSQ_GPUS="2", LSQ_DEVICES="0,0", LSQ_GATHER="host", LSQ_SHARD="reads")),
                                      ("wild_s12", dict(LSQ_GPUS="3", LSQ_DEVICES="0,0,0", LSQ_GATHER="host", LSQ_SHARD="reads")),
                                      ("errors", dict(LSQ_GPUS="2", LSQ_DEVICES="0,0", LSQ_GATHER="host", LSQ_SHARD="reads")),
                                      ("events_s3", dict(LSQ_GPUS="1", LSQ_GATHER="rccl", LSQ_SHARD="reads"))])
def test_executables_run_one_job_over_several_gpus_by_reads(name, env, tmp_path):
    """LSQ_GPUS=N LSQ_SHARD=reads: every GPU copies and parses a byte range of each MRF file (cut at line starts; the slices'
    newline counts give every slice its file-wide first line number, which names the reads in span-start ties), counts it
    against ALL events, the class counts are summed over the GPUs (lsq_allreduce_counts; here through host memory, every
    "GPU" being device 0) and GPU 0 goes on as a single-GPU run: the reference's golden tables, exit statuses included
    (`errors`: a field that fails the cast is reported from whichever slice holds it).  With LSQ_GPUS=1 nothing is
    sharded (one slice needs no exchange): the plain run."""
    import subprocess
    c, d = load_case(name, tmp_path)
    n = 0
    for tool, r in runs(c):
        p = subprocess.run([os.path.join(BIN, tool)] + r["argv"], cwd=d, capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
        exp = open(os.path.join(d, r["stdout"])).read()
        assert p.returncode in ((r["exit"],) if r["exit"] != 134 else (134, -6)), (name, tool, r["argv"], p.returncode, p.stderr[-500:])
        if tool == "count":
            assert p.stdout == exp, (name, r["argv"])
        else:
            assert ob.solve_text_close(p.stdout, exp), (name, r["argv"])
        n += 1
    assert n


@pytest.mark.parametrize("env", [
    {"LSQ_GPUS": "3", "LSQ_DEVICES": "0,0,0", "LSQ_GATHER": "host", "LSQ_FAIL_RANK": "1"},      # a slice fails before the hand-over
    {"LSQ_GPUS": "3", "LSQ_DEVICES": "0,0,0", "LSQ_GATHER": "host", "LSQ_FAIL_RANK": "2", "LSQ_SHARD": "reads"},      # ... of a read-sharded job
    {"LSQ_GPUS": "2", "LSQ_DEVICES": "0,99", "LSQ_GATHER": "host"},                               # a slice whose device does not exist
    {"LSQ_GPUS": "1", "LSQ_GATHER": "rccl", "LSQ_FAIL_RANK": "0"},                                # ... with the RCCL communicator made
])
def test_a_failing_slice_ends_the_job_with_its_message(env, tmp_path):
    """The slices' host threads agree before anyone enters the collective: one slice that failed on the way (no context,
    an unreadable file, out of memory) makes the job exit 3 with that slice's message -- the others neither gather nor
    wait for it (they used to enqueue ncclAllGather and block for the missing peer)."""
    import subprocess
    c, d = load_case("toy", tmp_path)
    r = c["count"][0]
    p = subprocess.run([os.path.join(BIN, "count")] + r["argv"], cwd=d, capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
    assert p.returncode == 3 and p.stdout == "", (p.returncode, p.stdout, p.stderr)
    assert "ERROR" in p.stderr and ("LSQ_FAIL_RANK" in p.stderr or "out of range" in p.stderr), p.stderr


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_whole_reference_runs_of_config0_and_config1(name, tmp_path):
    """tests/golden_full: stdout of the reference's own binaries on BASELINE.json configs[0] and configs[1] at full size
    (10 M reads; 158 s / 166 s of the reference).  The executables' count table is that output byte for byte, the solve
    table equals it as printed."""
    import subprocess
    from test_oracle_golden import full_case
    argv, cexp, sexp = full_case(name, tmp_path)
    p = subprocess.run([os.path.join(BIN, "count")] + argv[:-1], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout == cexp
    p = subprocess.run([os.path.join(BIN, "solve")] + argv, capture_output=True, text=True)
    assert p.returncode == 0 and ob.solve_text_close(p.stdout, sexp)


# ---- round 4 ----------------------------------------------------------------------------------------------------------------

def test_long_reads_choose_wide_records_and_match_the_oracle(tmp_path):
    """reads of 1 500 bases (single blocks of 1-1.5 kb, junction reads): no block fits a compact record's ten length bits, the
    ingest routes the file again for wide records by itself (common/read.h:204-274 on multi-kb blocks) -- tables as the oracle's"""
    spec = L.SynthSpec(77, 300, 60000, 1500, 3, L.EVENT_TYPES)
    L.synth_write(spec, str(tmp_path), "l")
    argv = ["0", "l", "./", "LH_GENE_TXT", str(tmp_path / "l.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "l.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", "1500", str(tmp_path / "l.mrf"), str(60000 * 1500)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    compare_exact(gpu_exact(argv), exact, "long reads")
    rc, text = L.cli_run("count", argv[:-1])
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext
    rc, text = L.cli_run("solve", argv)
    assert rc == 0 and ob.solve_text_close(text, otext)
    # ... and it was the wide-record kernel that counted, chosen by the ingest, from the text as from parsed arrays
    ann = L.Annotation(argv[4], argv[6])
    ev = L.Events(ann, ("SHORT_READ",), (1500,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads_mrf(0, argv[12])
    fmt = ctx.pool_format(0)
    assert fmt[0] is False and fmt[2][0] > 20000 and fmt[2][2] < fmt[2][0] // 4
    ctx.count()
    a = [x.copy() for x in ctx.counts()]
    ctx.upload_reads(0, L.Reads.from_mrf(argv[12], ev))
    assert ctx.pool_format(0)[0] is False
    ctx.count()
    b = ctx.counts()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and int(a[0].sum()) > 20000
    ctx.close()


@pytest.mark.parametrize("zipf", [False, True], ids=["even_depth", "zipf_depth"])
def test_coordinate_sorted_file_vs_oracle(zipf, tmp_path):
    """the same reads in coordinate order, as an aligner's sorted output has them: neighbouring lines share a bucket, a cell and
    -- with hot genes -- a counter; line numbers (read names) follow the file, so the oracle reads the sorted file too"""
    spec = L.SynthSpec(83, 1500, 300000, 100, 4, L.EVENT_TYPES, zipf, sorted_reads=True)
    L.synth_write(spec, str(tmp_path), "s")
    lines = open(tmp_path / "s.mrf").read().split("\n")[1:-1]
    keys = [(int(l.split(":")[0][3:]), int(l.split(":")[2])) for l in lines]
    assert keys == sorted(keys) and len(lines) == 300000
    argv = ["0", "s", "./", "LH_GENE_TXT", str(tmp_path / "s.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "s.map"),
            "0", "100000000", "MRF_SINGLE", "SHORT_READ", "100", str(tmp_path / "s.mrf"), str(300000 * 100)]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    compare_exact(gpu_exact(argv), exact, "sorted")
    rc, text = L.cli_run("count", argv[:-1])          # (the executable's loader: the device parser and the chain behind it)
    rc2, ctext, _ = ob.run("count", argv[:-1])
    assert rc == rc2 == 0 and text == ctext
    # the generator's direct read set is the file's
    ann = L.Annotation(argv[4], argv[6])
    ev = L.Events(ann, ("SHORT_READ",), (100,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    ctx.upload_reads(0, L.Reads.synthetic(spec, ev))
    ctx.count()
    a = [x.copy() for x in ctx.counts()]
    ctx.upload_reads_mrf(0, argv[12])
    ctx.count()
    b = ctx.counts()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ctx.close()


def test_overflow_recount_on_any_number_of_workgroups_and_its_warning(tmp_path):
    """the recount an overflowed exception list ends in is bucket by bucket, each bucket cleared and counted by one wave: the same
    tables from 1, 16 or 300 workgroups of the exception pass, with a second count in flight behind the first -- and the host says
    so, once per count, the first time it looks (log format of the reference, jsc/util/log.hpp)"""
    import golden_inputs as gi
    info = gi.write_events_case(str(tmp_path), "x", seed=79, n_events=300, n_reads=60000, R=60, n_chrom=3)
    lines = open(tmp_path / "x.mrf").read().split("\n")
    extra = []
    for e in info["events"]:               # reads that cover an event's span exactly: the strand / name order decides (count/count.cpp:64-85)
        gs = min(f[0][0] for f in e["forms"]); ge = max(f[-1][1] for f in e["forms"])
        for strand in ("+", "-", e["strand"]):
            extra.append("%s:%s:%d:%d:1:%d" % (e["chrom"], strand, gs + 1, ge, ge - gs))
    with open(tmp_path / "x.mrf", "w") as f:
        f.write("\n".join(lines[:-1] + extra) + "\n")
    ann = L.Annotation(str(tmp_path / "x.interval"), str(tmp_path / "x.map"), 0, 1000)
    ev = L.Events(ann, ("SHORT_READ",), (60,))
    ctx = L.Context(0)
    ctx.upload_events(ev)
    reads = L.Reads.from_mrf(str(tmp_path / "x.mrf"), ev)
    ctx.upload_reads(0, reads)
    ctx.count(); ctx.solve()
    cnt0, bases0 = [x.copy() for x in ctx.counts()]
    assert ctx.count_status()[1] == [0]
    L.lib.lsq_set_log_level(2)             # (an in-process run of the executables leaves its own log_level argument behind)
    log = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "native_stderr.log")     # where conftest.py points fd 2 during a GPU test
    seen = lambda: open(log, errors="replace").read().count("WARNING] read file 0: the exception list overflowed") if os.path.exists(log) else None
    before = seen()
    ctx.set_option("exception_capacity", 1)
    ctx.upload_reads(0, reads)
    for wgs in (1, 16, 300, 0):
        ctx.set_option("cleanup_workgroups", wgs)
        for _ in range(3):                 # (steps submitted back to back: the recount of one runs beside the count of the next)
            ctx.count(); ctx.solve()
        cnt1, bases1 = ctx.counts()
        assert np.array_equal(cnt0, cnt1) and np.array_equal(bases0, bases1), wgs
        exc, rec = ctx.count_status()
        assert rec == [1] and exc[0] > 1
        if before is not None:
            assert seen() == before + 1, wgs          # once per count the host looked at, however often it looked
            before += 1
    L.lib.lsq_set_log_level(0)
    ctx.count()
    ctx.counts()
    if before is not None:
        assert seen() == before
    L.lib.lsq_set_log_level(2)
    ctx.close()


@pytest.mark.parametrize("seed,cap", [(511, 4), (512, 8), (513, 16), (514, 48), (515, 1), (516, 8)])
def test_em_capped_four_lane_kernel_and_tail_stop_where_the_reference_stops(seed, cap, tmp_path):
    """option em_quad_cap: the four-lane lean kernel hands events that still run after `cap` accepted iterations to the tail kernel
    (closed form of the EM map where it applies, ordinary iterations otherwise): iteration count equal to the oracle's for every
    event, theta and log-likelihood within 1e-6, also over repeated steps (the lane's list is cleared by the tail) and with the
    placement by earlier iteration counts switched off (the path the cap is for)"""
    import golden_inputs as gi
    R = [40, 60, 90, 120][seed % 4]
    info = gi.write_events_case(str(tmp_path), "ev", seed=seed, n_events=400, n_reads=30000 if seed % 2 else 3000, R=R, n_chrom=2, zipf=(seed % 2 == 0))
    argv = ["0", "ev", "./", "LH_GENE_TXT", str(tmp_path / "ev.interval"), "UCSC_GENE2ISOFORM", str(tmp_path / "ev.map"), "0", "1000",
            "MRF_SINGLE", "SHORT_READ" if seed % 3 else "MEDIUM_READ", str(R), str(tmp_path / "ev.mrf"), str(info["total_read_bases"])]
    rc, otext, exact = ob.run("solve", argv)
    assert rc == 0
    plain = gpu_exact(argv)
    for repeat in (1, 4):
        got = gpu_exact(argv, options={"em_quad_cap": cap, "em_regroup": 0}, repeat=repeat)
        compare_exact(got, exact, "capped at %d, seed %d" % (cap, seed))
        assert max(g["iters"] for g in got) > cap          # some event went past the cap
        for a, b in zip(got, plain):
            assert a["iters"] == b["iters"] or (a["flags"] & 4) or (b["flags"] & 4), a["gname"]
