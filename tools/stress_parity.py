#!/usr/bin/env python3
"""Developer aid (GPU box): the generators behind the golden sets on seeds no test has, GPU path against the oracle.
usage: python tools/stress_parity.py [first_seed=1000] [n=200]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lesseq_amd as L
import oracle_binding as ob
import golden_inputs as gi
from test_parity_gpu import gpu_exact, compare_exact
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + n):
    d = tempfile.mkdtemp()
    try:
        if seed % 3:
            info = gi.write_wild_case(d, "w", seed)
            R = info["R"]
        else:
            R = [40, 60, 90, 120, 150][seed % 5]
            info = gi.write_events_case(d, "w", seed=seed, n_events=40 + seed % 50, n_reads=3000 + 37 * (seed % 100), R=R, n_chrom=1 + seed % 3, zipf=(seed % 2 == 0))
        argv = ["0", "w", "./", "LH_GENE_TXT", d + "/w.interval", "UCSC_GENE2ISOFORM", d + "/w.map", "0", "100000",
                "MRF_SINGLE", "SHORT_READ" if seed % 4 else "MEDIUM_READ", str(R), d + "/w.mrf", str(info["total_read_bases"])]
        rc, otext, exact = ob.run("solve", argv)
        if rc != 0:                     # the generator made an input the reference itself refuses (it asserts): the tools must too
            rc1, _ = L.cli_run("solve", argv)
            assert rc1 != 0, (seed, rc1, rc)      # (in a host process the library reports the error; the executables abort like the reference)
            continue
        for opts in ("em_flat_min_events=0", "compact_pools=0", "reads_per_look=8,workgroups_per_cu=5", "reads_per_look=4,workgroups_per_cu=6,em_closed_form=1",
                     "share_taper=0.05,grid_multiplier=7,share_cost_hot=3", "share_weighted=0,grid_multiplier=0.2"):
            os.environ["LSQ_OPTIONS"] = opts
            compare_exact(gpu_exact(argv, repeat=4 if opts.startswith("em_flat") else 1), exact, "seed %d %s" % (seed, opts))
            rc1, text = L.cli_run("count", argv[:-1])
            rc2, ctext, _ = ob.run("count", argv[:-1])
            assert rc1 == rc2 == 0 and text == ctext, (seed, opts)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if (seed - first) % 50 == 49:
        print("done", seed - first + 1, "bad", bad, flush=True)
print("total", n, "bad", bad)
sys.exit(1 if bad else 0)
