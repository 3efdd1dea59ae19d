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
