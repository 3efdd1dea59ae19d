"""The oracle (oracle/lsq_oracle.c) against stdout captured from the reference's own binaries.
CPU only.  This is what pins the oracle; the GPU parity tests then compare against it."""
import json
import os
import shutil

import pytest

import golden_inputs as gi
import oracle_binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isfile(os.path.join(GOLD, d, "case.json")))


def load_case(name, tmp_path):
    """returns (case dict, directory holding the inputs) -- regenerates inputs that are not stored"""
    d = os.path.join(GOLD, name)
    c = json.load(open(os.path.join(d, "case.json")))
    if c.get("gen") == "fmt1m":
        w = str(tmp_path / "fmt1m")
        shutil.copytree(d, w)
        with open(os.path.join(w, "fmt1m.mrf"), "w") as f:
            for ln in gi.fmt1m_lines():
                f.write(ln)
        d = w
    return c, d


def runs(case):
    for tool in ("count", "solve"):
        for r in case.get(tool, []):
            yield tool, r


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_stdout(name, tmp_path, monkeypatch):
    c, d = load_case(name, tmp_path)
    monkeypatch.chdir(d)
    for tool, r in runs(c):
        rc, text, _ = ob.run(tool, r["argv"])
        exp = open(os.path.join(d, r["stdout"])).read()
        assert rc == r["exit"], (name, tool, r["argv"])
        assert text == exp, (name, tool, r["argv"])


def test_golden_set_covers_the_edge_cases():
    # the inputs exercise: empty events, ragged reads, unterminated last line, comment lines,
    # string-sorted gene indices, range selection, touching exons, name ties, two read files
    assert {"toy", "edge", "quirks", "multi_method", "errors", "fmt1m"} <= set(CASES)
    assert sum(1 for c in CASES if c.startswith("wild_")) >= 6
    assert "1.2e+06" in open(os.path.join(GOLD, "fmt1m", "count.out")).read()


FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_full")


def full_case(name, tmp_path):
    """(argv for solve, expected count stdout, expected solve stdout) of a tests/golden_full case; the inputs are re-created"""
    import gzip
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lesseq_amd as L
    c = json.load(open(os.path.join(FULL, "meta.json")))["cases"][name]
    sp = c["spec"]
    L.synth_write(L.SynthSpec(sp["seed"], sp["n_events"], sp["n_reads"], sp["read_length"], sp["n_chrom"], tuple(sp["event_types"])), str(tmp_path), name, write_mrf=True)
    argv = ["0", name, "./", "LH_GENE_TXT", str(tmp_path / (name + ".interval")), "UCSC_GENE2ISOFORM", str(tmp_path / (name + ".map")), "0", "100000000",
            "MRF_SINGLE", "SHORT_READ", str(sp["read_length"]), str(tmp_path / (name + ".mrf")), c["total_read_bases"]]
    exp = [gzip.open(os.path.join(FULL, name, t + ".out.gz")).read().decode() for t in ("count", "solve")]
    return argv, exp[0], exp[1]


def test_oracle_matches_reference_on_whole_config0(tmp_path):
    """BASELINE.json configs[0] (10 k reads, 100 SE events), the reference's own CPU-runnable case: whole stdout of the
    reference's count and solve"""
    argv, cexp, sexp = full_case("c1", tmp_path)
    rc, text, _ = ob.run("count", argv[:-1])
    assert rc == 0 and text == cexp
    rc, text, _ = ob.run("solve", argv)
    assert rc == 0 and text == sexp


def test_oracle_matches_reference_on_whole_config1(tmp_path):
    """BASELINE.json configs[1] at full size (10 M reads, 5 k SE/RI events): the 10 000 rows the reference's count printed
    in its 158 s run and the 10 000 rows its solve printed in 166 s (solve/solve.cpp:829-847: theta, RPKM, mean
    log-likelihood at six significant digits), both byte for byte"""
    argv, cexp, sexp = full_case("c2", tmp_path)
    rc, text, _ = ob.run("count", argv[:-1])
    assert rc == 0 and text == cexp
    rc, text, _ = ob.run("solve", argv)
    assert rc == 0 and text == sexp
