#!/usr/bin/env python3
"""One-off large parity check on the GPU box: the executables' tables against the oracle's on a
C3-shaped input (mixed events, 24 chromosomes) too large for the test suite.
python tests/big_parity.py [n_reads] [n_events] [zipf]"""
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lesseq_amd as L  # noqa: E402
import oracle_binding as ob  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
n_events = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
zipf = len(sys.argv) > 3 and sys.argv[3] == "zipf"
d = tempfile.mkdtemp(prefix="lsq_big_", dir="/tmp")
L.synth_write(L.SynthSpec(17, n_events, n_reads, 100, 24, L.EVENT_TYPES, zipf, 0.10), d, "s")
os.chdir(d)
base = ["0", "s", "./", "LH_GENE_TXT", "s.interval", "UCSC_GENE2ISOFORM", "s.map", "0", "100000000", "MRF_SINGLE", "SHORT_READ", "100", "s.mrf"]
t0 = time.time()
rc, ctext, _ = ob.run("count", base)
rc2, stext, _ = ob.run("solve", base + [str(n_reads * 100)])
t1 = time.time()
assert rc == 0 and rc2 == 0
rc, text = L.cli_run("count", base)
assert rc == 0
rc2, text2 = L.cli_run("solve", base + [str(n_reads * 100)])
t2 = time.time()
assert rc2 == 0
print("oracle %.1f s, library %.1f s; count tables identical: %s (%d rows); solve tables within 1e-6: %s" %
      (t1 - t0, t2 - t1, text == ctext, text.count("\n"), ob.solve_text_close(text2, stext)))
assert text == ctext and ob.solve_text_close(text2, stext)
os.remove("s.mrf")
