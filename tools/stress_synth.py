#!/usr/bin/env python3
"""Developer aid (GPU box): random configurations of the synthetic generator (events, reads, read length, chromosomes, event
types, depth skew, overlap), GPU path against the oracle with compact records, wide records and the recount kernel."""
import os, sys, tempfile, shutil, random
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lesseq_amd as L
import oracle_binding as ob
from test_parity_gpu import gpu_exact, compare_exact
bad = 0
rng = random.Random(12345)
for k in range(60):
    seed = 5000 + k
    n_ev = rng.choice([50, 300, 1200, 2500]); n_reads = rng.choice([20000, 90000, 250000]); R = rng.choice([36, 50, 75, 100, 150])
    n_chrom = rng.choice([1, 2, 5]); zipf = rng.random() < 0.5; overlap = rng.choice([0.0, 0.1, 0.4, 0.7])
    types = L.EVENT_TYPES if rng.random() < 0.6 else tuple(rng.sample(list(L.EVENT_TYPES), rng.randint(1, 3)))
    d = tempfile.mkdtemp()
    try:
        spec = L.SynthSpec(seed, n_ev, n_reads, R, n_chrom, types, zipf, overlap)
        L.synth_write(spec, d, "s")
        argv = ["0", "s", "./", "LH_GENE_TXT", d + "/s.interval", "UCSC_GENE2ISOFORM", d + "/s.map", "0", "100000000", "MRF_SINGLE", "SHORT_READ", str(R), d + "/s.mrf", str(n_reads * R)]
        rc, otext, exact = ob.run("solve", argv)
        assert rc == 0
        for opts in ("em_flat_min_events=0", "compact_pools=0", "recount_every_read=1", "reads_per_look=8,workgroups_per_cu=5", "reads_per_look=4,workgroups_per_cu=6,em_closed_form=1"):
            os.environ["LSQ_OPTIONS"] = opts
            compare_exact(gpu_exact(argv, repeat=4 if opts.startswith("em_flat") else 1), exact, "synth %d %s" % (seed, opts))
        os.environ["LSQ_OPTIONS"] = ""
        rc1, text = L.cli_run("count", argv[:-1]); rc2, ctext, _ = ob.run("count", argv[:-1])
        assert rc1 == rc2 == 0 and text == ctext
    except AssertionError as e:
        bad += 1; print("MISMATCH", seed, n_ev, n_reads, R, n_chrom, zipf, overlap, types, str(e)[:300], flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if k % 10 == 9: print("done", k + 1, "bad", bad, flush=True)
print("total bad", bad)
