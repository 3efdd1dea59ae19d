#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own prebuilt
executables (/root/reference/bin/{count,solve,classify}).  Test infrastructure only.

Runs only in the build container (the reference tree does not exist on the GPU box).  The
reference binaries are executed where they lie; nothing of the reference is copied into the
repo -- only the inputs this script invents and the stdout the reference printed for them.

    make -C oracle ref && python oracle/make_golden.py [--only NAME]

Every case directory holds: the three input files, `case.json` (argv for count / solve with
paths relative to the directory, expected exit codes) and the captured `count.out` /
`solve.out` (+ `classify/` matrices where asked).  Large inputs are not stored: a case may
name a generator (`gen` in case.json) that tests/golden_inputs.py re-creates on the fly.
"""
import argparse
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("LSQ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi  # noqa: E402  (deterministic input writers shared with the tests)


def run_ref(tool, argv, cwd):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(HERE, "_ref", "lib")
    p = subprocess.run([os.path.join(REF, "bin", tool)] + argv, cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr


def finish_case(name, d, spec):
    """spec: dict(count=[argv...], solve=[argv...], classify=[argv...] optional)"""
    out = {"name": name}
    for k in ("gen", "note"):
        if k in spec:
            out[k] = spec[k]
    for tool in ("count", "solve"):
        for i, argv in enumerate(spec.get(tool, [])):
            rc, so, se = run_ref(tool, argv, d)
            fn = "%s%s.out" % (tool, "" if i == 0 else str(i))
            with open(os.path.join(d, fn), "wb") as f:
                f.write(so)
            out.setdefault(tool, []).append({"argv": argv, "exit": rc, "stdout": fn})
            print("  %-28s %s #%d exit=%d %d bytes" % (name, tool, i, rc, len(so)))
    if "classify" in spec:
        cdir = os.path.join(d, "classify")
        shutil.rmtree(cdir, ignore_errors=True)
        os.makedirs(cdir)
        rc, so, se = run_ref("classify", spec["classify"], d)
        out["classify"] = {"argv": spec["classify"], "exit": rc, "files": sorted(os.listdir(cdir))}
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


def std_args(stem, lo, hi, reads):
    """reads: list of (read_type, R, mrf, total_read_bases)"""
    base = ["0", stem, "./", "LH_GENE_TXT", stem + ".interval", "UCSC_GENE2ISOFORM", stem + ".map", str(lo), str(hi)]
    c = list(base)
    s = list(base)
    for (rt, R, mrf, trb) in reads:
        c += ["MRF_SINGLE", rt, str(R), mrf]
        s += ["MRF_SINGLE", rt, str(R), mrf, str(trb)]
    return c, s


def make_full():
    """tests/golden_full: the reference's whole runs on BASELINE configs[0] and configs[1]: inputs from
    lsq_synth_write (not stored), stdout gzip'd, wall-clock / CPU time / peak RSS of each run into meta.json.  The 10 M-read
    runs take ~2.7 minutes and ~4.8 GB each."""
    import gzip
    import hashlib
    import resource
    import tempfile
    import time
    sys.path.insert(0, ROOT)
    import lesseq_amd as L
    full = os.path.join(ROOT, "tests", "golden_full")
    meta = json.load(open(os.path.join(full, "meta.json")))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "lib"))
    with tempfile.TemporaryDirectory() as d:
        for name, c in sorted(meta["cases"].items()):
            sp = c["spec"]
            L.synth_write(L.SynthSpec(sp["seed"], sp["n_events"], sp["n_reads"], sp["read_length"], sp["n_chrom"], tuple(sp["event_types"])), d, name, write_mrf=True)
            os.makedirs(os.path.join(full, name), exist_ok=True)
            for tool in ("count", "solve"):
                argv = [os.path.join(REF, "bin", tool), "0", name, "./", "LH_GENE_TXT", name + ".interval", "UCSC_GENE2ISOFORM", name + ".map", "0", "100000000",
                        "MRF_SINGLE", "SHORT_READ", str(sp["read_length"]), name + ".mrf"] + ([c["total_read_bases"]] if tool == "solve" else [])
                r0, t0 = resource.getrusage(resource.RUSAGE_CHILDREN), time.perf_counter()
                p = subprocess.run(argv, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                dt, r1 = time.perf_counter() - t0, resource.getrusage(resource.RUSAGE_CHILDREN)
                assert p.returncode == 0
                with open(os.path.join(full, name, tool + ".out.gz"), "wb") as f:
                    f.write(gzip.compress(p.stdout, 9, mtime=0))
                c["reference"][tool] = {"wall_s": round(dt, 2), "user_s": round(r1.ru_utime - r0.ru_utime, 2), "sys_s": round(r1.ru_stime - r0.ru_stime, 2),
                                        "maxrss_MB": int(r1.ru_maxrss / 1024), "rows": p.stdout.count(b"\n"), "stdout_sha256": hashlib.sha256(p.stdout).hexdigest()}
                print("  %s %s: %.1f s, %d rows" % (name, tool, dt, p.stdout.count(b"\n")))
    with open(os.path.join(full, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--full", action="store_true", help="regenerate tests/golden_full (whole reference runs on configs[0] and [1]; minutes)")
    a = ap.parse_args()
    if a.full:
        if not os.path.isdir(REF):
            sys.exit("reference tree not present")
        subprocess.check_call(["make", "-C", HERE, "ref"])
        return make_full()
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: golden vectors can only be regenerated in the build container")
    if not os.path.exists(os.path.join(HERE, "_ref", "lib", "libgsl.so.0")):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)

    cases = []

    def case(name, writer, spec_fn, store_inputs=True, **kw):
        if a.only and a.only != name:
            return
        d = os.path.join(GOLD, name)
        os.makedirs(d, exist_ok=True)
        info = writer(d)           # writes inputs, returns dict for spec_fn
        spec = spec_fn(info)
        spec.update(kw)
        finish_case(name, d, spec)
        if not store_inputs:
            for fn in info.get("big_files", []):
                os.remove(os.path.join(d, fn))
        cases.append(name)

    # A.1 toy, A.2 edge
    def toy_spec(info):
        c, s = std_args("toy", 0, 10, [("SHORT_READ", 50, "toy.mrf", 650)])
        c[0] = "2"
        return {"count": [c], "solve": [s], "classify": ["0", "toy", "classify/", "LH_GENE_TXT", "toy.interval", "UCSC_GENE2ISOFORM", "toy.map", "0", "10"]}
    case("toy", gi.write_toy, toy_spec)

    def edge_spec(info):
        c, s = std_args("e", 0, 10, [("SHORT_READ", 100, "e.mrf", 1400)])
        c1, s1 = std_args("e", 0, 1, [("SHORT_READ", 100, "e.mrf", 1400)])
        c2, s2 = std_args("e", 1, 3, [("MEDIUM_READ", 100, "e.mrf", 1400)])
        return {"count": [c, c1, c2], "solve": [s, s1, s2]}
    case("edge", gi.write_edge, edge_spec)

    # touching / ordering quirks of interval_list::add_interval and ExonSet::insert
    def quirk_spec(info):
        c, s = std_args("q", 0, 100, [("SHORT_READ", 40, "q.mrf", 12345)])
        return {"count": [c], "solve": [s]}
    case("quirks", gi.write_quirks, quirk_spec)

    # LESSeq-shaped events (all 8 types), synthetic reads per SURVEY 8(d)
    for seed, R in ((1, 50), (2, 75), (3, 100)):
        def w(d, seed=seed, R=R):
            return gi.write_events_case(d, "ev", seed=seed, n_events=40, n_reads=1500, R=R, n_chrom=3)
        def sp(info, R=R):
            c, s = std_args("ev", 0, 1000, [("SHORT_READ", R, "ev.mrf", info["total_read_bases"])])
            c1, s1 = std_args("ev", 5, 17, [("SHORT_READ", R, "ev.mrf", info["total_read_bases"])])
            # classify over a gene range (seed 2) or everything: one .matrix per selected gene with two or more isoforms
            lo, hi = ("3", "20") if seed == 2 else ("0", "1000")
            return {"count": [c, c1], "solve": [s, s1], "classify": ["0", "ev", "classify/", "LH_GENE_TXT", "ev.interval", "UCSC_GENE2ISOFORM", "ev.map", lo, hi]}
        case("events_s%d" % seed, w, sp)

    # classify's own filter: single-isoform genes leave no file (classify/classify.cpp:159)
    def cm_spec(info):
        c, s = std_args("cm", 0, 100, [("SHORT_READ", 10, "cm.mrf", 110)])
        return {"count": [c], "solve": [s], "classify": ["0", "cm", "classify/", "LH_GENE_TXT", "cm.interval", "UCSC_GENE2ISOFORM", "cm.map", "0", "100"]}
    case("classify_mix", gi.write_classify_mix, cm_spec)

    # arbitrary isoform structures, tiny coordinate range, adversarial names / strands / blocks
    for seed in (11, 12, 13, 14, 15, 16):
        def w(d, seed=seed):
            return gi.write_wild_case(d, "w", seed=seed)
        def sp(info):
            c, s = std_args("w", 0, 1000, [("SHORT_READ", info["R"], "w.mrf", info["total_read_bases"])])
            return {"count": [c], "solve": [s]}
        case("wild_s%d" % seed, w, sp)

    # two sampling methods with different read types / lengths
    def w(d):
        i1 = gi.write_events_case(d, "mm", seed=21, n_events=25, n_reads=900, R=50, n_chrom=2)
        i2 = gi.write_reads_only(d, "mm", "mm2.mrf", seed=22, n_reads=700, R=90)
        i1["trb2"] = i2["total_read_bases"]
        return i1
    def sp(info):
        c, s = std_args("mm", 0, 1000, [("SHORT_READ", 50, "mm.mrf", info["total_read_bases"]),
                                         ("MEDIUM_READ", 90, "mm2.mrf", info["trb2"])])
        return {"count": [c], "solve": [s]}
    case("multi_method", w, sp)

    # error paths: exit codes and (empty) stdout
    def w(d):
        return gi.write_errors(d)
    def sp(info):
        base = ["0", "x", "./", "LH_GENE_TXT", "toy.interval", "UCSC_GENE2ISOFORM", "toy.map", "0", "10"]
        return {"count": [base + ["MRF_SINGLE", "SHORT_READ", "50", "bad_number.mrf"],
                          base + ["MRF_SINGLE", "LONG_READ", "50", "toy.mrf"],
                          base + ["MRF_PAIRED", "SHORT_READ", "50", "toy.mrf"],
                          base + ["MRF_SINGLE", "SHORT_READ", "fifty", "toy.mrf"],
                          base[:7] + ["0", "10", "MRF_SINGLE", "SHORT_READ", "50"],
                          ["0", "x", "./", "UCSC_GENE_TXT", "toy.interval", "UCSC_GENE2ISOFORM", "toy.map", "0", "10", "MRF_SINGLE", "SHORT_READ", "50", "toy.mrf"],
                          base + ["MRF_SINGLE", "SHORT_READ", "50", "no_qfields.mrf"]],
                "solve": [base + ["MRF_SINGLE", "SHORT_READ", "50", "toy.mrf", "abc"]]}
    case("errors", w, sp)

    # the annotation formats only `solve` reads (solve/solve.cpp:158-329); `count` refuses them
    def w(d):
        return gi.write_formats(d)
    def sp(info):
        tail = ["0", "1000", "MRF_SINGLE", "SHORT_READ", str(info["R"]), "f.mrf"]
        trb = [str(info["total_read_bases"])]
        combos = [("LH_GENE_TXT", "f.interval", "UCSC_GENE2ISOFORM", "f.map"),
                  ("UCSC_GENE_TXT", "f.ucsc.txt", "UCSC_GENE2ISOFORM", "f.map"),
                  ("UCSC_GFF", "f.gff", "UCSC_GENE2ISOFORM", "f.map"),
                  ("WORMBASE_GFF2", "f.worm.gff2", "WORMBASE_GENE2ISOFORMS", "f.worm.map"),
                  ("GENELETS_GFF3", "f.genelets.gff3", "UCSC_GENE2ISOFORM", "f.map"),
                  ("LH_GENE_TXT", "f.interval", "WORMBASE_GENE2ISOFORMS", "f.worm.map"),
                  ("UCSC_GFF", "f.gff", "NO_SUCH_MAP", "f.map"),
                  ("NO_SUCH_FORMAT", "f.gff", "UCSC_GENE2ISOFORM", "f.map")]
        solve = [["0", "f", "./", a, b, c, dd] + tail + trb for (a, b, c, dd) in combos]
        solve.append(["0", "f", "./", "UCSC_GFF", "f.gff", "UCSC_GENE2ISOFORM", "f.map", "3", "11", "MRF_SINGLE", "MEDIUM_READ", str(info["R"]), "f.mrf"] + trb)
        count = [["0", "f", "./", a, b, c, dd] + tail for (a, b, c, dd) in combos[1:6]]
        return {"count": count, "solve": solve}
    case("formats", w, sp)

    # the read formats only `solve` reads (solve/solve.cpp:413-428,487-634): reads keyed by name
    def w(d):
        return gi.write_readfmts(d)
    def sp(info):
        base = ["0", "rf", "./", "LH_GENE_TXT", "rf.interval", "UCSC_GENE2ISOFORM", "rf.map", "0", "1000"]
        trb = str(info["total_read_bases"])
        R = str(info["R"])
        solve = [base + [fmt, "SHORT_READ", R, path, trb] for fmt, path in (("UCSC_GFF", "rf.gff"), ("UCSC_BED", "rf.bed"), ("WORMBASE_GFF3", "rf.gff3"), ("MRF_SINGLE", "rf.mrf"))]
        solve.append(base + ["UCSC_GFF", "SHORT_READ", R, "rf.gff", trb, "UCSC_BED", "MEDIUM_READ", R, "rf.bed", trb])
        solve.append(base[:7] + ["2", "9", "WORMBASE_GFF3", "MEDIUM_READ", R, "rf.gff3", trb])
        count = [base + [fmt, "SHORT_READ", R, path] for fmt, path in (("UCSC_GFF", "rf.gff"), ("UCSC_BED", "rf.bed"), ("WORMBASE_GFF3", "rf.gff3"))]
        return {"count": count, "solve": solve}
    case("readfmts", w, sp)

    # 6-significant-digit formatting of counts >= 1e6 (input regenerated by the tests)
    def w(d):
        return gi.write_fmt1m(d)
    def sp(info):
        c, s = std_args("toy", 0, 10, [("SHORT_READ", 50, "fmt1m.mrf", 75000000)])
        return {"count": [c], "solve": [s], "gen": "fmt1m"}
    case("fmt1m", w, sp, store_inputs=False)

    print("wrote:", ", ".join(cases))


if __name__ == "__main__":
    main()
