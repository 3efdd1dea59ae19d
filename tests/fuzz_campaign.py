#!/usr/bin/env python3
"""One-off randomized parity campaign on the GPU box (not collected by pytest): the generators behind the
wild_* and events_s* golden sets on seed ranges of any length, HIP path against the oracle -- exact integers,
theta to 1e-6, printed count tables byte for byte.
python tests/fuzz_campaign.py [first_seed] [n_seeds]"""
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import lesseq_amd as L  # noqa: E402
import oracle_binding as ob  # noqa: E402
import golden_inputs as gi  # noqa: E402
from test_parity_gpu import gpu_exact, compare_exact  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time()
bad, flagged, refused = [], 0, 0
FIM = os.environ.get("LSQO_FIM") is not None      # also lsq_fim against the oracle's fim.h restatement
fim_checked = 0
for seed in range(first, first + n):
    d = tempfile.mkdtemp(prefix="lsq_fuzz_%d_" % seed, dir="/tmp")
    try:
        if seed % 2:
            info = gi.write_wild_case(d, "w", seed)
            R = info["R"]
        else:
            R = [36, 50, 75, 100, 150][seed % 5]
            info = gi.write_events_case(d, "w", seed=seed, n_events=40 + seed % 50, n_reads=3000 + 37 * (seed % 100), R=R, n_chrom=1 + seed % 3,
                                        zipf=(seed % 4 == 0))
        argv = ["0", "w", "./", "LH_GENE_TXT", d + "/w.interval", "UCSC_GENE2ISOFORM", d + "/w.map", "0", "100000",
                "MRF_SINGLE", "SHORT_READ" if seed % 3 else "MEDIUM_READ", str(R), d + "/w.mrf", str(info["total_read_bases"])]
        if seed % 2 == 0 and seed % 10 < 4:       # a second (and third) read file over the same events: several sampling methods
            for extra in range(1 + (seed % 10) // 2):
                R2 = [40, 60, 80][(seed + extra) % 3]
                info2 = gi.write_reads_only(d, "w", "w%d.mrf" % extra, seed + 7777 + extra, 1500 + 11 * (seed % 50), R2)
                argv += ["MRF_SINGLE", "MEDIUM_READ" if (seed + extra) % 4 == 0 else "SHORT_READ", str(R2), d + "/w%d.mrf" % extra, str(info2["total_read_bases"])]
        rc, otext, exact = ob.run("solve", argv)
        if rc != 0:                      # an input the reference itself refuses (assert / exit 1): the library must refuse it too
            rc_lib, _ = L.cli_run("solve", argv)
            assert rc_lib != 0, "oracle rc %d but the library ran" % rc
            refused += 1
            continue
        got = gpu_exact(argv, want_fim=FIM)
        flagged += compare_exact(got, exact, "seed %d" % seed)
        if FIM:
            import numpy as np
            for g, e in zip(got, exact):
                if "fim" not in e or any(abs(a - b) > 1e-9 * max(abs(a), abs(b)) for a, b in zip(g["theta"], e["theta"])):
                    continue
                for m in range(len(e["fim"])):
                    A, B = np.array(g["fim"][m], float).reshape(-1), np.array(e["fim"][m], float).reshape(-1)
                    scale = max(np.abs(B).max(), 1e-300) if B.size else 1.0
                    assert np.all(np.abs(A - B) <= 1e-9 * scale), "fim %s method %d" % (g["gname"], m)
                    fim_checked += 1
        cargv = argv[:9] + [x for g in range((len(argv) - 9) // 5) for x in argv[9 + 5 * g:9 + 5 * g + 4]]     # count takes no total_read_bases
        rc, text = L.cli_run("count", cargv)
        rc2, ctext, _ = ob.run("count", cargv)
        assert rc == rc2 == 0 and text == ctext, "count table differs"
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
    finally:
        for f in os.listdir(d):
            os.unlink(os.path.join(d, f))
        os.rmdir(d)
    if (seed - first) % 50 == 49:
        print("  ... %d seeds, %d failures, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
print("seeds %d..%d: %d failures, %d inputs refused by both, %d events flagged by the EM guard band, %.0f s" % (first, first + n - 1, len(bad), refused, flagged, time.time() - t0))
if FIM:
    print("fisher information matrices compared: %d" % fim_checked)
for b in bad[:20]:
    print("  FAIL seed %d: %s" % b)
sys.exit(1 if bad else 0)
