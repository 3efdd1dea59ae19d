"""Deterministic writers for the small input sets behind tests/golden/.

Used by oracle/make_golden.py (to create the inputs the reference is run on) and by the
tests (to re-create inputs that are too large to commit).  Pure Python; no reference code.
Event shapes follow SURVEY.md 8(d) (derived there from the reference's bin/Events.r).
"""
import os
import random


# ----------------------------------------------------------------------------- helpers

def _w(path, text, newline_at_end=True):
    with open(path, "w") as f:
        f.write(text)


def interval_line(name, chrom, strand, exons, exon_count=None):
    starts = ",".join(str(s) for s, _ in exons)
    ends = ",".join(str(e) for _, e in exons)
    tx_s = min(s for s, _ in exons)
    tx_e = max(e for _, e in exons)
    n = len(exons) if exon_count is None else exon_count
    return "%s\t%s\t%s\t%d\t%d\t%d\t%s\t%s\n" % (name, chrom, strand, tx_s, tx_e, n, starts, ends)


def mrf_line(chrom, strand, blocks):
    """blocks: list of 0-based half-open (s,e) -> MRF 1-based inclusive with query coords"""
    out = []
    q = 1
    for (s, e) in blocks:
        ln = e - s
        out.append("%s:%s:%d:%d:%d:%d" % (chrom, strand, s + 1, e, q, q + ln - 1))
        q += ln
    return ",".join(out) + "\n"


def transcript_blocks(exons, tstart, length):
    """map [tstart, tstart+length) in transcript coordinates to genomic blocks"""
    blocks = []
    pos = 0
    remaining = length
    for (s, e) in exons:
        ln = e - s
        if tstart < pos + ln and remaining > 0:
            off = max(tstart - pos, 0)
            take = min(ln - off, remaining)
            blocks.append((s + off, s + off + take))
            remaining -= take
            tstart += take
        pos += ln
    return blocks


# ----------------------------------------------------------------------------- fixed cases

TOY_INTERVAL = (
    interval_line("SE1.inc", "chr1", "+", [(1000, 1100), (1200, 1300), (1500, 1600)])
    + interval_line("SE1.skp", "chr1", "+", [(1000, 1100), (1500, 1600)])
    + interval_line("RI1.ret", "chr1", "+", [(5000, 5500)])
    + interval_line("RI1.spl", "chr1", "+", [(5000, 5200), (5300, 5500)])
)
TOY_MAP = "SE1\tSE1.inc\nSE1\tSE1.skp\nRI1\tRI1.ret\nRI1\tRI1.spl\n"
TOY_MRF = """AlignmentBlocks
chr1:+:1001:1050:1:50
chr1:+:1011:1060:1:50
chr1:+:1081:1100:1:20,chr1:+:1201:1230:21:50
chr1:+:1081:1100:1:20,chr1:+:1501:1530:21:50
chr1:+:1221:1270:1:50
chr1:+:1271:1300:1:30,chr1:+:1501:1520:31:50
chr1:+:1541:1590:1:50
chr1:+:1052:1101:1:50
chr1:+:5101:5150:1:50
chr1:+:5181:5230:1:50
chr1:+:5181:5200:1:20,chr1:+:5301:5330:21:50
chr1:+:5401:5450:1:50
chr2:+:1001:1050:1:50
"""


def write_toy(d):
    _w(os.path.join(d, "toy.interval"), TOY_INTERVAL)
    _w(os.path.join(d, "toy.map"), TOY_MAP)
    _w(os.path.join(d, "toy.mrf"), TOY_MRF)
    return {}


def write_classify_mix(d):
    """classify keeps genes with two or more isoforms only (classify/classify.cpp:159): `solo` must leave no file; `tri`
    has three isoforms with overlapping, nested and abutting exons; `far` sits on another chromosome and strand; the last
    line of the map has no newline (dropped by the reference's reader: `cut` then has one isoform left and vanishes too)."""
    iv = (
        interval_line("solo.1", "chr1", "+", [(100, 300), (500, 800)])
        + interval_line("tri.a", "chr1", "-", [(2000, 2100), (2300, 2500), (2900, 3000)])
        + interval_line("tri.b", "chr1", "-", [(2000, 2150), (2300, 2400), (2400, 2500), (2900, 3000)])
        + interval_line("tri.c", "chr1", "-", [(2050, 2100), (2900, 3000)])
        + interval_line("far.x", "chrX", "-", [(10, 20)])
        + interval_line("far.y", "chrX", "-", [(10, 20), (30, 40)])
        + interval_line("cut.1", "chr2", "+", [(5, 50)])
        + interval_line("cut.2", "chr2", "+", [(5, 50), (70, 90)])
    )
    mp = "solo\tsolo.1\ntri\ttri.a\ntri\ttri.b\ntri\ttri.c\nfar\tfar.x\nfar\tfar.y\ncut\tcut.1\ncut\tcut.2"
    _w(os.path.join(d, "cm.interval"), iv)
    _w(os.path.join(d, "cm.map"), mp)
    _w(os.path.join(d, "cm.mrf"), "AlignmentBlocks\nchr1:+:111:160:1:50\nchr1:-:2011:2060:1:50\nchr1:-:2301:2350:1:50\nchr1:-:2091:2100:1:10,chr1:-:2301:2340:11:50\nchrX:-:12:20:1:9\nchr2:+:11:40:1:30\n")
    return {}


def write_edge(d):
    iv = (
        interval_line("10.a", "chr1", "+", [(1000, 1100), (1200, 1300), (1500, 1600)])
        + interval_line("10.b", "chr1", "+", [(1000, 1100), (1500, 1600)])
        + interval_line("9.a", "chr1", "+", [(1000, 1100), (1250, 1300), (1500, 1600)])
        + interval_line("9.b", "chr1", "+", [(1000, 1100), (1200, 1300), (1500, 1600)])
        + interval_line("2.a", "chr1", "-", [(3000, 3400)])
        + interval_line("2.b", "chr1", "-", [(3000, 3200)])
    )
    mp = "10\t10.a\n10\t10.b\n9\t9.a\n9\t9.b\n2\t2.a\n2\t2.b\n"
    mrf = """AlignmentBlocks
chr1:+:1001:1100:1:100
chr1:+:1002:1101:1:100
# comment line
chr1:+:1201:1300:1:100
chr1:+:1251:1300:1:50
chr1:+:1081:1100:1:20,chr1:+:1251:1280:21:50
chr1:+:1081:1100:1:20,chr1:+:1211:1240:21:50
chr1:+:1081:1100:1:20,chr1:+:2001:2030:21:50
chr1:-:3001:3100:1:100
chr1:-:3002:3101:1:100
chr1:-:3151:3250:1:100
chr1:-:3301:3400:1:100
chr1:+:1501:1600:1:100
chr1:+:1502:1600:1:99
chr1:+:1021:1060:1:40"""          # no trailing newline: the reference drops this line
    _w(os.path.join(d, "e.interval"), iv)
    _w(os.path.join(d, "e.map"), mp)
    _w(os.path.join(d, "e.mrf"), mrf)
    return {}


def write_quirks(d):
    """Insertion-order effects: touching exons merge in covered regions only when inserted
    left-to-right; zero-length exons split segments; duplicate isoform names (last wins);
    a single-isoform gene; a gene whose name ties with read names; last map line unterminated."""
    iv = (
        # gene A: isoform exons listed so that [200,300) is inserted before [100,200)
        interval_line("A.1", "c1", "+", [(200, 300), (400, 500)])
        + interval_line("A.2", "c1", "+", [(100, 200), (400, 500)])
        # gene B: touching exons in ascending order (merge) + retained intron shape
        + interval_line("B.1", "c1", "-", [(1000, 1100), (1100, 1180), (1300, 1400)])
        + interval_line("B.2", "c1", "-", [(1000, 1400)])
        # gene C: zero-length exon inside another exon
        + interval_line("C.1", "c1", "+", [(2000, 2200)])
        + interval_line("C.2", "c1", "+", [(2100, 2100), (2150, 2200)])
        # duplicate isoform name: the second record wins
        + interval_line("D.1", "c2", "+", [(10, 60)])
        + interval_line("D.1", "c2", "+", [(100, 190), (300, 380)])
        + interval_line("D.2", "c2", "+", [(100, 190), (240, 270), (300, 380)])
        # single-isoform gene
        + interval_line("E.only", "c2", ".", [(1000, 1200)])
        # gene named like a read
        + interval_line("R.1", "c2", "+", [(5000, 5040), (5100, 5140)])
        + interval_line("R.2", "c2", "+", [(5000, 5040), (5060, 5080), (5100, 5140)])
        + interval_line("Z.unused", "c9", "+", [(1, 2)])
        + interval_line("Y.1", "c3", "+", [(100, 200)])
        + interval_line("Y.2", "c3", "+", [(100, 150)])
    )
    mp = ("A\tA.1\nA\tA.2\nB\tB.1\nB\tB.2\nC\tC.1\nC\tC.2\nD\tD.1\nD\tD.2\nE\tE.only\n"
          "read-25\tR.1\nread-25\tR.2\nY\tY.1\nY\tY.2")     # last line unterminated -> Y has one isoform
    lines = ["AlignmentBlocks"]
    rd = []
    # gene A region: blocks across the 200 boundary (kept only if [100,300) merged -- it is not)
    rd += [("c1", "+", [(150, 190)]), ("c1", "+", [(180, 220)]), ("c1", "+", [(210, 250)]),
           ("c1", "+", [(260, 300), (400, 440)]), ("c1", "+", [(160, 200), (400, 440)]),
           ("c1", "+", [(100, 140)]), ("c1", "-", [(100, 140)]), ("c1", "+", [(200, 240)])]
    # gene B: touching blocks ascending (merge into one) and descending (stay two)
    rd += [("c1", "-", [(1080, 1100), (1100, 1120)]), ("c1", "-", [(1100, 1120), (1080, 1100)]),
           ("c1", "-", [(1150, 1180), (1300, 1340)]), ("c1", "-", [(1150, 1190)]),
           ("c1", "-", [(1040, 1080)]), ("c1", "-", [(1360, 1400)]), ("c1", "-", [(1170, 1210)]),
           ("c1", "-", [(1000, 1040)]), ("c1", "-", [(1000, 1400)])]
    # gene C
    rd += [("c1", "+", [(2080, 2120)]), ("c1", "+", [(2130, 2170)]), ("c1", "+", [(2000, 2040)]),
           ("c1", "+", [(2150, 2190)]), ("c1", "+", [(2090, 2100), (2150, 2180)])]
    # gene D
    rd += [("c2", "+", [(20, 60)]), ("c2", "+", [(150, 190), (300, 340)]), ("c2", "+", [(170, 190), (240, 270), (300, 310)]),
           ("c2", "+", [(120, 160)]), ("c2", "+", [(250, 270), (300, 320)]), ("c2", "+", [(240, 270)])]
    # gene E (K = 1)
    rd += [("c2", ".", [(1000, 1040)]), ("c2", "+", [(1100, 1140)]), ("c2", ".", [(1160, 1200)]), ("c2", "-", [(1001, 1041)])]
    for (c, s, b) in rd:
        lines.append(mrf_line(c, s, b).rstrip("\n"))
    # gene "read-25": reads spanning exactly [5000,5140) on '+' tie with the gene key on
    # (start,end,strand) and are ordered by name: "read-N" < "read-25" decides (bytewise).
    while len(lines) < 60:
        n = len(lines)          # this line will be read-<n>
        if n % 3 == 0:
            lines.append(mrf_line("c2", "+", [(5000, 5040), (5100, 5140)]).rstrip("\n"))
        elif n % 3 == 1:
            lines.append(mrf_line("c2", "+", [(5000, 5040), (5060, 5080), (5100, 5140)]).rstrip("\n"))
        else:
            lines.append(mrf_line("c2", "-" if n % 2 else "+", [(5000, 5040), (5100, 5140)]).rstrip("\n"))
    lines.insert(30, "# a comment consumes a read number")
    lines.insert(45, "AlignmentBlocks")
    _w(os.path.join(d, "q.interval"), iv)
    _w(os.path.join(d, "q.map"), mp)
    _w(os.path.join(d, "q.mrf"), "\n".join(lines) + "\n")
    return {}


# ----------------------------------------------------------------------------- LESSeq-shaped events

EVENT_TYPES = ("SE", "RI", "A5SS", "A3SS", "MXE", "AFE", "ALE", "T3")


def make_event(rng, etype, pos, R):
    """returns (formA_exons, formB_exons, next_free_pos); exon lengths U[60,400], introns U[100,5000]"""
    def L():
        return rng.randint(max(60, R + 1), 400)

    def I():
        return rng.randint(100, 5000)
    a = pos
    if etype == "SE":
        e1 = (a, a + L()); e2s = e1[1] + I(); e2 = (e2s, e2s + rng.randint(60, 400)); e3s = e2[1] + I(); e3 = (e3s, e3s + L())
        return [e1, e2, e3], [e1, e3], e3[1]
    if etype == "RI":
        e1 = (a, a + L()); e2s = e1[1] + rng.randint(80, 900); e2 = (e2s, e2s + L())
        return [(e1[0], e2[1])], [e1, e2], e2[1]
    if etype in ("A5SS", "A3SS"):
        e1 = (a, a + L()); ext = rng.randint(20, 200); gap = I(); e3s = e1[1] + ext + gap; e3 = (e3s, e3s + L())
        if etype == "A5SS":   # alternative donor: first exon extended
            return [e1, (e1[1], e1[1] + ext), e3], [e1, e3], e3[1]
        # alternative acceptor: last exon extended upstream
        return [e1, (e3[0] - ext, e3[0]), e3], [e1, e3], e3[1]
    if etype == "MXE":
        e1 = (a, a + L()); b1s = e1[1] + I(); b1 = (b1s, b1s + rng.randint(60, 300)); b2s = b1[1] + I(); b2 = (b2s, b2s + rng.randint(60, 300))
        e4s = b2[1] + I(); e4 = (e4s, e4s + L())
        return [e1, b1, e4], [e1, b2, e4], e4[1]
    if etype == "AFE":
        e1 = (a, a + L()); e2s = e1[1] + I(); e2 = (e2s, e2s + L()); e3s = e2[1] + I(); e3 = (e3s, e3s + L())
        return [e2, e3], [e1, e3], e3[1]
    if etype == "ALE":
        e1 = (a, a + L()); e2s = e1[1] + I(); e2 = (e2s, e2s + L()); e3s = e2[1] + I(); e3 = (e3s, e3s + L())
        return [e1, e2], [e1, e3], e3[1]
    # T3: long single exon vs its prefix
    ln = rng.randint(2 * R + 50, 1200)
    cut = rng.randint(R + 5, ln - 20)
    return [(a, a + ln)], [(a, a + cut)], a + ln


def gen_events(rng, n_events, R, chroms, overlap_frac=0.10, types=EVENT_TYPES):
    """events left-to-right per chromosome; ids are decimal counters like Events.r"""
    events = []
    pos = {c: rng.randint(2000, 20000) for c in chroms}
    for i in range(n_events):
        c = chroms[rng.randrange(len(chroms))] if len(chroms) > 1 else chroms[0]
        et = types[rng.randrange(len(types))]
        strand = "+" if rng.random() < 0.5 else "-"
        start = pos[c]
        if events and rng.random() < overlap_frac:
            # overlap the previous event on this chromosome: start inside it
            prev = [e for e in events if e["chrom"] == c]
            if prev:
                p = prev[-1]
                start = rng.randint(p["span"][0] + 1, max(p["span"][0] + 2, p["span"][1] - 1))
        fa, fb, end = make_event(rng, et, start, R)
        gname = str(i + 1)
        events.append({"name": gname, "type": et, "chrom": c, "strand": strand,
                       "forms": [fa, fb], "span": (min(fa[0][0], fb[0][0]), max(fa[-1][1], fb[-1][1]))})
        pos[c] = max(pos[c], end) + rng.randint(2000, 20000)
    return events


def gen_reads(rng, events, n_reads, R, chroms, zipf=False):
    """85 % on events (70 % exonic, 25 % junction, 5 % adversarial), 15 % off-target"""
    reads = []
    chrom_end = {c: 0 for c in chroms}
    for e in events:
        chrom_end[e["chrom"]] = max(chrom_end[e["chrom"]], e["span"][1])
    weights = None
    if zipf:
        weights = [1.0 / (r + 1) ** 1.1 for r in range(len(events))]
    for _ in range(n_reads):
        u = rng.random()
        if u >= 0.85 or not events:
            c = chroms[rng.randrange(len(chroms))]
            s = rng.randint(0, chrom_end[c] + 20000)
            reads.append((c, "+" if rng.random() < 0.5 else "-", [(s, s + R)]))
            continue
        e = rng.choices(events, weights)[0] if weights else events[rng.randrange(len(events))]
        form = e["forms"][rng.randrange(2)]
        tlen = sum(b - a for a, b in form)
        strand = e["strand"] if rng.random() < 0.9 else ("+" if e["strand"] == "-" else "-")
        v = rng.random()
        if v < 0.70:
            ex = [x for x in form if x[1] - x[0] >= R]
            if ex:
                a, b = ex[rng.randrange(len(ex))]
                s = rng.randint(a, b - R)
                reads.append((e["chrom"], strand, [(s, s + R)]))
            else:
                t0 = rng.randint(0, max(tlen - R, 0))
                reads.append((e["chrom"], strand, transcript_blocks(form, t0, min(R, tlen))))
        elif v < 0.95:
            if len(form) >= 2 and tlen > R:
                j = rng.randrange(len(form) - 1)
                before = sum(b - a for a, b in form[: j + 1])
                o = rng.randint(1, R - 1)
                t0 = min(max(before - o, 0), tlen - R)
                reads.append((e["chrom"], strand, transcript_blocks(form, t0, R)))
            else:
                t0 = rng.randint(0, max(tlen - R, 0))
                reads.append((e["chrom"], strand, transcript_blocks(form, t0, min(R, tlen))))
        else:
            k = rng.randrange(5)
            gs, ge = e["span"]
            if k == 0:      # starts exactly at the event's first base
                reads.append((e["chrom"], strand, [(gs, gs + R)]))
            elif k == 1:    # 1-2 bp overhang past an exon end
                a, b = form[rng.randrange(len(form))]
                o = rng.randint(1, 2)
                reads.append((e["chrom"], strand, [(b - R + o, b + o)]))
            elif k == 2:    # spliced read whose second block starts mid-exon
                a, b = form[0]
                c2, d2 = form[-1]
                m = rng.randint(1, 20)
                reads.append((e["chrom"], strand, [(b - 20, b), (c2 + m, c2 + m + R - 20)]))
            elif k == 3:    # second block outside every covered region
                a, b = form[0]
                reads.append((e["chrom"], strand, [(b - 20, b), (ge + 1000, ge + 1000 + R - 20)]))
            else:           # ends exactly at the event's last base / spans to it
                reads.append((e["chrom"], strand, [(ge - R, ge)]))
    return reads


def write_events_case(d, stem, seed, n_events, n_reads, R, n_chrom, zipf=False, types=EVENT_TYPES):
    rng = random.Random(seed)
    chroms = ["chr%d" % (i + 1) for i in range(n_chrom)]
    events = gen_events(rng, n_events, R, chroms, types=types)
    iv, mp = [], []
    for e in events:
        for k, form in enumerate(e["forms"]):
            iname = "%s.%s" % (e["name"], "ab"[k])
            iv.append(interval_line(iname, e["chrom"], e["strand"], form))
            mp.append("%s\t%s\n" % (e["name"], iname))
    reads = gen_reads(rng, events, n_reads, R, chroms, zipf=zipf)
    _w(os.path.join(d, stem + ".interval"), "".join(iv))
    _w(os.path.join(d, stem + ".map"), "".join(mp))
    _w(os.path.join(d, stem + ".mrf"), "AlignmentBlocks\n" + "".join(mrf_line(*r) for r in reads))
    return {"total_read_bases": n_reads * R, "events": events}


def write_reads_only(d, stem, mrf_name, seed, n_reads, R):
    """a second read file over the events of an existing case (re-generated from its seed)"""
    # events are re-read from the interval file to stay independent of generator state
    events = {}
    with open(os.path.join(d, stem + ".interval")) as f:
        for line in f:
            t = line.split()
            g = t[0].split(".")[0]
            ex = list(zip([int(x) for x in t[6].split(",") if x], [int(x) for x in t[7].split(",") if x]))
            ev = events.setdefault(g, {"name": g, "chrom": t[1], "strand": t[2], "forms": []})
            ev["forms"].append(ex)
    evl = []
    for g, ev in events.items():
        ev["span"] = (min(f[0][0] for f in ev["forms"]), max(f[-1][1] for f in ev["forms"]))
        evl.append(ev)
    rng = random.Random(seed)
    chroms = sorted(set(e["chrom"] for e in evl))
    reads = gen_reads(rng, evl, n_reads, R, chroms)
    _w(os.path.join(d, mrf_name), "AlignmentBlocks\n" + "".join(mrf_line(*r) for r in reads))
    return {"total_read_bases": n_reads * R}


# ----------------------------------------------------------------------------- wild structures

def write_wild_case(d, stem, seed):
    """Arbitrary overlapping isoform structures on a tiny coordinate range so that every
    ExonSet::insert branch, interval merge rule and Read::build branch is hit; K from 1 to 5;
    gene names that sort oddly and tie with read names; odd strands; multi-block reads in any
    block order; isoforms shorter than the read length (ARS = 0)."""
    rng = random.Random(seed)
    R = rng.choice([20, 30, 45])
    chroms = ["c1", "c2"]
    names = ["10", "9", "2", "g7", "read-3", "read-12", "zz", "G", "g", "read-", "read-1x", "11", "1", "a.b"]
    rng.shuffle(names)
    n_genes = rng.randint(5, 9)
    genes = []
    iv, mp = [], []
    for gi_ in range(n_genes):
        gname = names[gi_]
        chrom = chroms[rng.randrange(2)]
        strand = rng.choice(["+", "-", "+", "-", "."])
        base = rng.randrange(0, 12) * 150 + 100
        # a pool of breakpoints so that starts/ends collide across isoforms
        bps = sorted(set(base + rng.randrange(0, 60) * 10 for _ in range(rng.randint(6, 14))))
        K = rng.choice([1, 2, 2, 2, 3, 4, 5])
        forms = []
        for k in range(K):
            m = rng.randint(1, min(5, len(bps) // 2))
            pts = sorted(rng.sample(bps, 2 * m))
            ex = [(pts[2 * i], pts[2 * i + 1]) for i in range(m)]
            if rng.random() < 0.25 and len(ex) >= 2:
                # make two consecutive exons touch
                i = rng.randrange(len(ex) - 1)
                ex[i] = (ex[i][0], ex[i + 1][0])
            if rng.random() < 0.08:
                p = rng.choice(bps)
                ex.append((p, p))         # zero-length exon, listed last (unsorted)
            if rng.random() < 0.06 and len(ex) >= 2:
                ex[0], ex[1] = ex[1], ex[0]   # unsorted exon list
            forms.append(ex)
            iname = "%s_i%d" % (gname, k)
            cnt = None
            if rng.random() < 0.05 and len(ex) > 1:
                cnt = len(ex) - 1         # exonCount smaller than the lists
            iv.append(interval_line(iname, chrom, strand, ex, exon_count=cnt))
            mp.append("%s\t%s\n" % (gname, iname))
        genes.append({"name": gname, "chrom": chrom, "strand": strand, "forms": forms})
    rng.shuffle(iv)
    reads = []
    n_reads = rng.randint(250, 400)
    strands = ["+", "-", "+", "-", ".", "*"]
    for _ in range(n_reads):
        g = genes[rng.randrange(len(genes))]
        form = [x for x in g["forms"][rng.randrange(len(g["forms"]))] if x[1] > x[0]]
        form.sort()
        tlen = sum(b - a for a, b in form)
        u = rng.random()
        strand = g["strand"] if rng.random() < 0.8 else rng.choice(strands)
        if u < 0.55 and tlen > 0:
            ln = min(tlen, rng.choice([R, R, R, R - 3, 7]))
            t0 = rng.randint(0, tlen - ln)
            bl = transcript_blocks(form, t0, ln)
            if rng.random() < 0.15:
                rng.shuffle(bl)
            reads.append((g["chrom"], strand, bl))
        elif u < 0.70 and form:
            # read anchored exactly at the gene start / end
            gs = min(a for f in g["forms"] for a, b in f if b > a)
            ge = max(b for f in g["forms"] for a, b in f if b > a)
            if rng.random() < 0.5:
                reads.append((g["chrom"], strand, [(gs, gs + rng.choice([5, 10, R]))]))
            else:
                reads.append((g["chrom"], strand, [(gs, ge)]))
        else:
            nb = rng.choice([1, 1, 2, 2, 3])
            lo = min(a for f in g["forms"] for a, b in f) - 20
            hi = max(b for f in g["forms"] for a, b in f) + 20
            bl = []
            for _b in range(nb):
                s = rng.randrange(max(lo, 0), hi) // 5 * 5 + rng.choice([0, 0, 0, 1])
                bl.append((s, s + rng.choice([5, 10, 10, 20, R])))
            if rng.random() < 0.5:
                bl.sort()
            reads.append((rng.choice([g["chrom"], g["chrom"], "c1", "c3"]), strand, bl))
    lines = ["AlignmentBlocks"] + [mrf_line(*r).rstrip("\n") for r in reads]
    for _ in range(3):
        lines.insert(rng.randint(1, len(lines)), "# note")
    _w(os.path.join(d, stem + ".interval"), "".join(iv))
    _w(os.path.join(d, stem + ".map"), "".join(mp))
    _w(os.path.join(d, stem + ".mrf"), "\n".join(lines) + "\n")
    return {"R": R, "total_read_bases": n_reads * R}


# ----------------------------------------------------------------------------- errors / formatting

def write_errors(d):
    write_toy(d)
    _w(os.path.join(d, "bad_number.mrf"), "AlignmentBlocks\nchr1:+:1001:1050:1:50\nchr1:+:10x1:1050:1:50\n")
    # MRF block without the query-coordinate fields: the `end` field swallows the next block
    _w(os.path.join(d, "no_qfields.mrf"), "AlignmentBlocks\nchr1:+:1081:1100,chr1:+:1201:1230\n")
    return {}


def fmt1m_lines():
    """1.2 M on-target reads over the toy SE event (6 patterns cycling) + 0.1 M off-target"""
    pats = [
        [(1010, 1060)], [(1220, 1270)], [(1540, 1590)],
        [(1080, 1100), (1200, 1230)], [(1080, 1100), (1500, 1530)], [(1270, 1300), (1500, 1520)],
    ]
    pl = [mrf_line("chr1", "+", p) for p in pats]
    off = mrf_line("chr1", "+", [(9000, 9050)])
    yield "AlignmentBlocks\n"
    for i in range(1200000):
        yield pl[(i * 7 + i // 5) % 6]
    for i in range(100000):
        yield off


def write_fmt1m(d):
    write_toy(d)
    with open(os.path.join(d, "fmt1m.mrf"), "w") as f:
        for ln in fmt1m_lines():
            f.write(ln)
    return {"big_files": ["fmt1m.mrf"]}


# ----------------------------------------------------------------------------- solve's other annotation formats

def write_formats(d):
    """One set of events written in every annotation format `solve` reads (solve/solve.cpp:158-329):
    LH_GENE_TXT, UCSC_GENE_TXT, UCSC_GFF, WORMBASE_GFF2, GENELETS_GFF3 for the isoforms;
    UCSC_GENE2ISOFORM and WORMBASE_GENE2ISOFORMS for the gene map.  The exon-per-line formats list
    the exons out of order, some of them cut in two touching or overlapping pieces (the loader
    merges them), and carry lines the loader must skip."""
    rng = random.Random(41)
    R = 60
    chroms = ["chr1", "chr2", "chrX"]
    events = gen_events(rng, 30, R, chroms)
    lh, ucsc, g2i, worm_g2i = [], [], [], []
    gff = ["track name=demo\n", "browser position chr1\n"]
    gen3 = ["##gff-version 3\n", "##source demo\n"]
    worm = []
    exon_lines = []      # (iname, chrom, strand, start, end)
    for e in events:
        names = []
        for k, form in enumerate(e["forms"]):
            iname = "%s.%s" % (e["name"], "ab"[k])
            names.append(iname)
            lh.append(interval_line(iname, e["chrom"], e["strand"], form))
            starts = ",".join(str(s) for s, _ in form) + ","
            ends = ",".join(str(x) for _, x in form) + ","
            ucsc.append("%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n" % (iname, e["chrom"], e["strand"], form[0][0], form[-1][1],
                                                                       form[0][0] + 3, form[-1][1] - 3, len(form), starts, ends))
            g2i.append("%s\t%s\n" % (e["name"], iname))
            for (s, x) in form:
                pieces = [(s, x)]
                u = rng.random()
                if u < 0.2 and x - s > 20:                       # two touching pieces
                    m = rng.randint(s + 5, x - 5)
                    pieces = [(s, m), (m, x)]
                elif u < 0.35 and x - s > 30:                    # two overlapping pieces
                    m = rng.randint(s + 10, x - 10)
                    pieces = [(s, m + 4), (m - 4, x)]
                for (a, b) in pieces:
                    exon_lines.append((iname, e["chrom"], e["strand"], a, b))
        worm_g2i.append("%s\t%s\n" % (e["name"], ";".join(names)))
    rng.shuffle(exon_lines)
    for (iname, c, strand, a, b) in exon_lines:
        gff.append("%s\tdemo\texon\t%d\t%d\t.\t%s\t.\t\"%s\"\n" % (c, a + 1, b, strand, iname))
        worm.append("%s\tdemo\texon\t%d\t%d\t.\t%s\t.\tTranscript\t\"%s\"\n" % (c[3:], a + 1, b, strand, iname))
    # GENELETS: exons shared by both isoforms of an event are written once with Parent=a,b
    seen = {}
    for (iname, c, strand, a, b) in exon_lines:
        seen.setdefault((c, strand, a, b, iname.split(".")[0]), []).append(iname)
    n = 0
    for (c, strand, a, b, g), inames in seen.items():
        n += 1
        if n % 7 == 0:
            gen3.append("%s\tdemo\tmRNA\t%d\t%d\t.\t%s\t.\tID=%s\n" % (c[3:], a + 1, b, strand, inames[0]))
        gen3.append("%s\tdemo\texon\t%d\t%d\t.\t%s\t.\tID=ex%d;Parent=%s;Note=x\n" % (c[3:], a + 1, b, strand, n, ",".join(inames)))
    reads = gen_reads(rng, events, 1200, R, chroms)
    _w(os.path.join(d, "f.interval"), "".join(lh))
    _w(os.path.join(d, "f.ucsc.txt"), "".join(ucsc))
    _w(os.path.join(d, "f.gff"), "".join(gff))
    _w(os.path.join(d, "f.worm.gff2"), "".join(worm))
    _w(os.path.join(d, "f.genelets.gff3"), "".join(gen3))
    _w(os.path.join(d, "f.map"), "".join(g2i))
    _w(os.path.join(d, "f.worm.map"), "".join(worm_g2i))
    _w(os.path.join(d, "f.mrf"), "AlignmentBlocks\n" + "".join(mrf_line(*r) for r in reads))
    return {"total_read_bases": 1200 * R, "R": R}


# ----------------------------------------------------------------------------- solve's other read formats

def write_readfmts(d):
    """One set of named reads in the three read formats only `solve` takes (solve/solve.cpp:413-428,
    487-634): UCSC_GFF (a line per block, lines of a name make one read), UCSC_BED (a line per read,
    kept or dropped as a whole by its span), WORMBASE_GFF3 (a line per read, an optional intron in
    the Parent attribute).  Names decide span-start ties against gene names, so some reads cover an
    event's span exactly and carry names below, equal to and above the gene's."""
    rng = random.Random(43)
    R = 50
    chroms = ["chr1", "chr2"]
    events = gen_events(rng, 25, R, chroms)
    iv, mp = [], []
    for e in events:
        for k, form in enumerate(e["forms"]):
            iname = "%s.%s" % (e["name"], "ab"[k])
            iv.append(interval_line(iname, e["chrom"], e["strand"], form))
            mp.append("%s\t%s\n" % (e["name"], iname))
    raw = [r for r in gen_reads(rng, events, 900, R, chroms) if len(r[2]) <= 2]
    reads = []           # (name, chrom, strand, blocks)
    for i, (c, strand, blocks) in enumerate(raw):
        reads.append(("r%04d" % i, c, strand, blocks))
    for j, e in enumerate(events):          # span-start ties: exactly the event's span, one block
        gs, ge = e["span"]
        for nm, st in ((e["name"], e["strand"]), (e["name"] + "0", e["strand"]), ("!" + e["name"], e["strand"]), ("zz%d" % j, e["strand"]),
                       ("!s%d" % j, "+" if e["strand"] == "-" else "-"), ("zs%d" % j, "+" if e["strand"] == "-" else "-")):
            if rng.random() < 0.6:
                reads.append((nm, e["chrom"], st, [(gs, ge)]))
    for k in range(6):                      # two entries under one name: one read with the union of their blocks
        a = reads[rng.randrange(200)]
        reads.append((a[0], a[1], a[2], [(a[3][0][0] + 7, a[3][0][0] + 7 + R)]))
    rng.shuffle(reads)
    gff = ["track name=reads\n", "browser position chr1\n"]
    bed = ["track name=reads\n"]
    gff3 = []
    for (nm, c, strand, blocks) in reads:
        for (s, x) in blocks:
            gff.append("%s\tdemo\tread\t%d\t%d\t.\t%s\t.\t%s\n" % (c, s + 1, x, strand, nm))
        s0, e0 = blocks[0][0], blocks[-1][1]
        sizes = ",".join(str(x - s) for s, x in blocks) + ","
        starts = ",".join(str(s - s0) for s, _ in blocks) + ","
        bed.append("%s\t%d\t%d\t%s\t0\t%s\t%d\t%d\t0\t%d\t%s\t%s\n" % (c, s0, e0, nm, strand, s0, e0, len(blocks), sizes, starts))
        attr = "ID=m%s;Target=%s 1 %d +;" % (nm, nm, sum(x - s for s, x in blocks))
        if len(blocks) == 2:
            attr += "Parent=intron_X_%d_%d;" % (blocks[0][1] + 1, blocks[1][0])
        gff3.append("%s\tdemo\tmatch\t%d\t%d\t.\t%s\t.\t%s\n" % (c[3:], s0 + 1, e0, strand, attr))
    _w(os.path.join(d, "rf.interval"), "".join(iv))
    _w(os.path.join(d, "rf.map"), "".join(mp))
    _w(os.path.join(d, "rf.gff"), "".join(gff))
    _w(os.path.join(d, "rf.bed"), "".join(bed))
    _w(os.path.join(d, "rf.gff3"), "".join(gff3))
    _w(os.path.join(d, "rf.mrf"), "AlignmentBlocks\n" + "".join(mrf_line(c, st, bl) for (_, c, st, bl) in reads))
    return {"total_read_bases": len(reads) * R, "R": R}
