/*
 * lsq_oracle.c -- CPU restatement of LESSeq's count + solve path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C restatement of what the reference's `count`
 * and `solve` executables compute, structured the way the reference computes it (per-read
 * interval lists, a totally ordered read index, a per-event lower_bound + scan, a per-read
 * likelihood matrix and a per-read EM).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or run it.  The product (lesseq_amd/) never
 * links or calls it.
 *
 * Pinning: tests/test_oracle_golden.py checks this restatement against stdout captured
 * from the reference's own prebuilt binaries (/root/reference/bin/{count,solve}) on the
 * inputs under tests/golden/ (made by oracle/make_golden.py, which only runs where
 * /root/reference exists).
 *
 * Reference lines followed (all under /root/reference):
 *   count/count.cpp:88-501           CLI, loaders, read filter, index, per-gene loop, output
 *   solve/solve.cpp:102-146,429-486,665-847   same + delta-G, EM call, RPKM, output
 *   common/read.h:44-79              contiguous-run compatibility
 *   common/read.h:204-274            Read_single::build state machine
 *   common/read.h:331-340            G_j = 1/ARS_j
 *   common/read.h:592-660            EM step, log-likelihood, stop rule
 *   common/accessible_read_starts.h:48-89,221-274   ARS (MEDIUM / SHORT)
 *   common/splicing_graph.h:88-169   ExonSet::insert
 *   common/splicing_graph.h:318-361  isoform x segment array, lengths
 *   jdu_source_collection/jsc/util/interval_list.hpp:348-357,396-422,462-503
 *
 *   common/fim.h:58-93,115-158,320-367, common/linalg.h:28-71 (with LSQO_FIM in the
 *       environment only)  expected Fisher information and the variance estimates -- PARITY
 *       UNPINNED: the reference never includes these headers, no binary prints these numbers,
 *       and linalg.h's inverse is GSL 1.12's LU (un-vendored blob), restated from its
 *       published algorithm
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; x86-64 SSE2 doubles as the
 * reference's g++ -O2 build).
 */
#include <ctype.h>
#include <errno.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NPOS ((size_t)-1)

/* ------------------------------------------------------------------ small utilities */

static void *xmalloc(size_t n) {
	void *p = malloc(n ? n : 1);
	if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); }
	return p;
}
static void *xrealloc(void *q, size_t n) {
	void *p = realloc(q, n ? n : 1);
	if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); }
	return p;
}
static char *xstrndup(const char *s, size_t n) {
	char *p = (char *)xmalloc(n + 1);
	memcpy(p, s, n);
	p[n] = 0;
	return p;
}

/* output sink: either a FILE* or a growing memory buffer (for the shared-library entry) */
typedef struct { FILE *fp; char *buf; size_t len, cap; } sink;
static void sink_write(sink *o, const char *s, size_t n) {
	if (o->fp) { fwrite(s, 1, n, o->fp); return; }
	if (o->len + n + 1 > o->cap) {
		while (o->len + n + 1 > o->cap) o->cap = o->cap ? o->cap * 2 : 4096;
		o->buf = (char *)xrealloc(o->buf, o->cap);
	}
	memcpy(o->buf + o->len, s, n);
	o->len += n;
	o->buf[o->len] = 0;
}
static void sink_str(sink *o, const char *s) { sink_write(o, s, strlen(s)); }
/* C++ `ostream << double` with default flags == printf("%g") (precision 6). */
static void sink_dbl(sink *o, double v) {
	char t[64];
	int n = snprintf(t, sizeof t, "%g", v);
	sink_write(o, t, (size_t)n);
}

/* ------------------------------------------------------------------ interval_list<long> */

typedef struct { long *s, *e; int n, cap; } ilist;

static void il_free(ilist *il) { free(il->s); free(il->e); il->s = il->e = NULL; il->n = il->cap = 0; }

static int lb_long(const long *a, int n, long v) { /* std::lower_bound */
	int lo = 0, hi = n;
	while (lo < hi) { int mid = lo + (hi - lo) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
	return lo;
}

/* interval_list.hpp:462-503.  Overlapping intervals merge; touching intervals merge only
 * when the earlier-stored one ends exactly where the new one starts. */
void lsqo_il_add(ilist *il, long start, long end) {
	if (!(start < end)) return;
	int s_starts = lb_long(il->s, il->n, start);
	int s_ends = lb_long(il->e, il->n, start);
	int e_starts = lb_long(il->s, il->n, end);
	int e_ends = lb_long(il->e, il->n, end);
	int s_on = (s_starts - s_ends == 1);
	int e_on = (e_starts - e_ends == 1);
	/* the two vectors are edited independently, exactly as in the reference */
	int ns = il->n - (e_starts - s_starts);
	int ne = il->n - (e_ends - s_ends);
	int ins_s = !s_on;            /* (!s&&!e) or (!s&&e) insert a start */
	int ins_e = !e_on;            /* (!s&&!e) or (s&&!e) insert an end  */
	int new_n = ns + ins_s;
	if (new_n != ne + ins_e) { fprintf(stderr, "oracle: interval_list invariant broken\n"); exit(2); }
	if (new_n + 1 > il->cap) {
		il->cap = (new_n + 1) * 2;
		il->s = (long *)xrealloc(il->s, sizeof(long) * (size_t)il->cap);
		il->e = (long *)xrealloc(il->e, sizeof(long) * (size_t)il->cap);
	}
	/* starts: erase [s_starts, e_starts), optionally insert `start` at s_starts */
	memmove(il->s + s_starts + ins_s, il->s + e_starts, sizeof(long) * (size_t)(il->n - e_starts));
	if (ins_s) il->s[s_starts] = start;
	/* ends: erase [s_ends, e_ends), optionally insert `end` at s_ends */
	memmove(il->e + s_ends + ins_e, il->e + e_ends, sizeof(long) * (size_t)(il->n - e_ends));
	if (ins_e) il->e[s_ends] = end;
	il->n = new_n;
}

/* interval_list.hpp:396-422 */
int lsqo_il_contains(const ilist *il, long start, long end) {
	if (!(start < end)) return 1;
	long idx = lb_long(il->s, il->n, start);
	if (idx >= 0 && idx < il->n && !(start < il->s[idx]) && !(il->e[idx] < end)) return 1;
	if (idx - 1 >= 0 && idx - 1 < il->n && !(start < il->s[idx - 1]) && !(il->e[idx - 1] < end)) return 1;
	return 0;
}

/* interval_list.hpp:348-357 (accumulates in double) */
static double il_total_length(const ilist *il) {
	double t = 0;
	for (int i = 0; i < il->n; i++) t += (double)(il->e[i] - il->s[i]);
	return t;
}

/* ------------------------------------------------------------------ ExonSet (splicing_graph.h:82-170) */

typedef struct { long start, end; } seg_t;
typedef struct { seg_t *v; int n, cap; } exonset;   /* ordered by start, starts unique */

static int es_lower_bound(const exonset *es, long start) {
	int lo = 0, hi = es->n;
	while (lo < hi) { int mid = lo + (hi - lo) / 2; if (es->v[mid].start < start) lo = mid + 1; else hi = mid; }
	return lo;
}
/* std::set::insert with a start-only comparator: an element with an equal start wins */
static void es_insert_unique(exonset *es, long start, long end) {
	int p = es_lower_bound(es, start);
	if (p < es->n && es->v[p].start == start) return;
	if (es->n + 1 > es->cap) { es->cap = es->cap ? es->cap * 2 : 8; es->v = (seg_t *)xrealloc(es->v, sizeof(seg_t) * (size_t)es->cap); }
	memmove(es->v + p + 1, es->v + p, sizeof(seg_t) * (size_t)(es->n - p));
	es->v[p].start = start; es->v[p].end = end; es->n++;
}

/* splicing_graph.h:88-169 */
void lsqo_exonset_insert(exonset *es, long start, long end) {
	seg_t nw = { start, end };
	seg_t ins[2 * 64]; int nins = 0;
	seg_t *insv = ins; int inscap = 128;
	int lower = es_lower_bound(es, nw.start);
	if (es->n > 0 && lower != 0) lower--;
	for (int i = lower; i < es->n && es->v[i].start < nw.end; i++) {
		seg_t *it = &es->v[i];
		if (nins + 2 > inscap) {
			inscap *= 2;
			seg_t *t = (seg_t *)xmalloc(sizeof(seg_t) * (size_t)inscap);
			memcpy(t, insv, sizeof(seg_t) * (size_t)nins);
			if (insv != ins) free(insv);
			insv = t;
		}
		if (it->start == nw.start) {
			if (it->end > nw.end) {
				insv[nins++] = (seg_t){ nw.end, it->end };
				it->end = nw.end;
				nw.start = nw.end;
			} else {
				nw.start = it->end;
			}
		} else if (it->start < nw.start) {
			if (it->end > nw.start) {
				if (it->end > nw.end) {
					insv[nins++] = (seg_t){ nw.start, nw.end };
					insv[nins++] = (seg_t){ nw.end, it->end };
					it->end = nw.start;
					nw.start = nw.end;
				} else {
					insv[nins++] = (seg_t){ nw.start, it->end };
					long old_end = it->end;
					it->end = nw.start;
					nw.start = old_end;
				}
			}
		} else {
			if (it->end > nw.end) {
				insv[nins++] = (seg_t){ nw.end, it->end };
				it->end = nw.end;
				nw.end = it->start;
			} else {
				insv[nins++] = (seg_t){ nw.start, it->start };
				nw.start = it->end;
			}
		}
	}
	if (nw.start < nw.end) es_insert_unique(es, nw.start, nw.end);
	for (int k = 0; k < nins; k++)
		if (insv[k].start < insv[k].end) es_insert_unique(es, insv[k].start, insv[k].end);
	if (insv != ins) free(insv);
}

/* ------------------------------------------------------------------ annotation records */

typedef struct {
	char *name, *chrom, *strand;
	long txStart, txEnd;
	unsigned long exonCount;
	long *exonStarts, *exonEnds;
	int nStarts, nEnds;
} ga_entry;

typedef struct {
	char *gname;
	ga_entry **isos; int niso, isocap;       /* order of lines in the g2i file */
	/* built for selected genes */
	exonset exons;
	ilist gene_il;
	const char *chrom, *strand;
	int K, N;
	int **iso_idx; int *iso_n;               /* known_iso_exon_indices */
	unsigned long *exon_len;                 /* exon_lengths[N] */
	unsigned long *iso_total_len;            /* known_iso_exon_total_lengths[i].back() */
} gene_t;

/* ------------------------------------------------------------------ strict numeric casts */

/* boost::lexical_cast<long>(std::string): whole string, optional sign, decimal digits */
static int lexical_cast_long(const char *s, size_t n, long *out) {
	if (n == 0 || n > 40) return 0;
	char t[48]; memcpy(t, s, n); t[n] = 0;
	size_t i = 0;
	if (t[i] == '+' || t[i] == '-') i++;
	if (i == n) return 0;
	for (size_t j = i; j < n; j++) if (!isdigit((unsigned char)t[j])) return 0;
	errno = 0;
	char *end;
	long v = strtol(t, &end, 10);
	if (errno == ERANGE || *end) return 0;
	*out = v;
	return 1;
}
static int lexical_cast_ulong(const char *s, unsigned long *out) {
	size_t n = strlen(s), i = 0;
	if (n == 0) return 0;
	if (s[i] == '+' || s[i] == '-') i++;
	if (i == n) return 0;
	for (size_t j = i; j < n; j++) if (!isdigit((unsigned char)s[j])) return 0;
	errno = 0;
	char *end;
	unsigned long v = strtoul(s, &end, 10);
	if (errno == ERANGE || *end) return 0;
	*out = v;
	return 1;
}
static int lexical_cast_ulong_n(const char *s, size_t n, unsigned long *out) {
	char t[64]; if (n == 0 || n > 62) return 0;
	memcpy(t, s, n); t[n] = 0;
	return lexical_cast_ulong(t, out);
}
static int lexical_cast_double(const char *s, double *out) {
	if (!*s || isspace((unsigned char)*s)) return 0;
	char *end;
	double v = strtod(s, &end);
	if (end == s || *end) return 0;
	*out = v;
	return 1;
}

/* `iss >> long` on one whitespace-delimited token of a well-formed file: optional sign and digits */
static int stream_long(const char *s, size_t n, long *out) {
	char t[64]; if (n == 0 || n > 62) return 0;
	memcpy(t, s, n); t[n] = 0;
	char *end; errno = 0;
	long v = strtol(t, &end, 10);
	if (end == t || errno == ERANGE) return 0;
	*out = v;
	return 1;
}

/* ------------------------------------------------------------------ text loaders */

typedef struct { char *data; size_t len; } text_t;

static int load_text(const char *path, text_t *t) {
	FILE *f = fopen(path, "rb");
	if (!f) return 0;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	t->data = (char *)xmalloc((size_t)n + 1);
	t->len = fread(t->data, 1, (size_t)n, f);
	t->data[t->len] = 0;
	fclose(f);
	return 1;
}

/* `while (getline(ifs, line) && !ifs.eof())`: yields only '\n'-terminated lines
 * (count.cpp:142,189,285).  Returns 0 when exhausted. */
static int next_line(const text_t *t, size_t *pos, const char **line, size_t *n) {
	if (*pos >= t->len) return 0;
	const char *nl = (const char *)memchr(t->data + *pos, '\n', t->len - *pos);
	if (!nl) return 0;
	*line = t->data + *pos;
	*n = (size_t)(nl - *line);
	*pos = (size_t)(nl - t->data) + 1;
	return 1;
}

/* whitespace tokenizer for `iss >> a >> b ...` */
static int next_tok(const char *line, size_t n, size_t *p, const char **tok, size_t *tn) {
	while (*p < n && isspace((unsigned char)line[*p])) (*p)++;
	if (*p >= n) return 0;
	*tok = line + *p;
	size_t q = *p;
	while (q < n && !isspace((unsigned char)line[q])) q++;
	*tn = q - *p;
	*p = q;
	return 1;
}

/* tokenizer<char_separator<char>>(",") + atol on every non-empty token (count.cpp:154-168) */
static int split_atol(const char *s, size_t n, long **out) {
	int cap = 8, cnt = 0;
	long *v = (long *)xmalloc(sizeof(long) * (size_t)cap);
	size_t i = 0;
	while (i < n) {
		while (i < n && s[i] == ',') i++;
		if (i >= n) break;
		size_t j = i;
		while (j < n && s[j] != ',') j++;
		char t[64]; size_t m = j - i < 63 ? j - i : 63;
		memcpy(t, s + i, m); t[m] = 0;
		if (cnt == cap) { cap *= 2; v = (long *)xrealloc(v, sizeof(long) * (size_t)cap); }
		v[cnt++] = atol(t);
		i = j;
	}
	*out = v;
	return cnt;
}

/* ------------------------------------------------------------------ reads */

typedef struct {
	unsigned long line_num;    /* name == "read-<line_num>" when `name` is NULL */
	char *name;                /* the formats only solve reads carry their own read names */
	int chrom, strand;         /* ids into string tables (compared as strings) */
	ilist il;
	long start, end;
} read_t;

typedef struct { char **v; int n, cap; } strtab;
static int strtab_id(strtab *t, const char *s, size_t n) {
	for (int i = 0; i < t->n; i++) if (strlen(t->v[i]) == n && memcmp(t->v[i], s, n) == 0) return i;
	if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 16; t->v = (char **)xrealloc(t->v, sizeof(char *) * (size_t)t->cap); }
	t->v[t->n] = xstrndup(s, n);
	return t->n++;
}

typedef struct { char **name; ilist *il; int n, cap; } chrom_regions;   /* covered_regions */
static ilist *regions_get(chrom_regions *cr, const char *chrom, size_t n) {
	for (int i = 0; i < cr->n; i++) if (strlen(cr->name[i]) == n && memcmp(cr->name[i], chrom, n) == 0) return &cr->il[i];
	if (cr->n == cr->cap) {
		cr->cap = cr->cap ? cr->cap * 2 : 32;
		cr->name = (char **)xrealloc(cr->name, sizeof(char *) * (size_t)cr->cap);
		cr->il = (ilist *)xrealloc(cr->il, sizeof(ilist) * (size_t)cr->cap);
	}
	cr->name[cr->n] = xstrndup(chrom, n);
	memset(&cr->il[cr->n], 0, sizeof(ilist));
	return &cr->il[cr->n++];
}

static size_t find_ch(const char *s, size_t n, char c, size_t pos) { /* std::string::find */
	if (pos >= n) return NPOS;
	const char *p = (const char *)memchr(s + pos, c, n - pos);
	return p ? (size_t)(p - s) : NPOS;
}
/* std::string::substr(pos, cnt) as a (ptr,len) view; pos <= n always holds at the call sites */
static void substr_view(const char *s, size_t n, size_t pos, size_t cnt, const char **o, size_t *on) {
	if (pos > n) pos = n;
	size_t avail = n - pos;
	*o = s + pos;
	*on = cnt < avail ? cnt : avail;
}

typedef struct { read_t *v; size_t n, cap; } readvec;

/* count.cpp:279-336 == solve.cpp:429-486.  Returns 0 ok, 1 on lexical_cast error. */
static int load_mrf(const text_t *t, chrom_regions *covered, strtab *chroms, strtab *strands, readvec *out) {
	size_t pos = 0; const char *line; size_t n;
	if (!next_line(t, &pos, &line, &n)) {
		/* getline of the header consumed whatever there was */
		return 0;
	}
	unsigned long line_num = 0;
	while (next_line(t, &pos, &line, &n)) {
		line_num++;
		if ((n >= 1 && line[0] == '#') || (n == 15 && memcmp(line, "AlignmentBlocks", 15) == 0)) continue;
		read_t rd; memset(&rd, 0, sizeof rd);
		int kept = 0;
		size_t last_comma = 0;
		while (last_comma != NPOS) {
			size_t colon = find_ch(line, n, ':', last_comma);
			size_t cpos = last_comma == 0 ? 0 : last_comma + 1;
			const char *chr; size_t chrn;
			substr_view(line, n, cpos, colon - cpos, &chr, &chrn);
			size_t old_colon = colon;
			colon = find_ch(line, n, ':', colon + 1);
			const char *strand; size_t strandn;
			substr_view(line, n, old_colon + 1, colon - old_colon - 1, &strand, &strandn);
			old_colon = colon;
			colon = find_ch(line, n, ':', colon + 1);
			const char *f; size_t fn;
			long start, end;
			substr_view(line, n, old_colon + 1, colon - old_colon - 1, &f, &fn);
			if (!lexical_cast_long(f, fn, &start)) { il_free(&rd.il); return 1; }
			old_colon = colon;
			colon = find_ch(line, n, ':', colon + 1);
			substr_view(line, n, old_colon + 1, colon - old_colon - 1, &f, &fn);
			if (!lexical_cast_long(f, fn, &end)) { il_free(&rd.il); return 1; }
			ilist *cov = regions_get(covered, chr, chrn);
			if (lsqo_il_contains(cov, start - 1, end)) {
				kept = 1;
				rd.chrom = strtab_id(chroms, chr, chrn);
				rd.strand = strtab_id(strands, strand, strandn);
				lsqo_il_add(&rd.il, start - 1, end);
			}
			last_comma = find_ch(line, n, ',', colon);
		}
		if (kept && rd.il.n > 0) {
			rd.line_num = line_num;
			rd.start = rd.il.s[0];
			rd.end = rd.il.e[rd.il.n - 1];
			if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 1024; out->v = (read_t *)xrealloc(out->v, sizeof(read_t) * out->cap); }
			out->v[out->n++] = rd;
		} else {
			/* kept with an empty interval list (start-1 >= end) is undefined behaviour in the
			 * reference (reads starts[0] of an empty vector); such reads are dropped here */
			il_free(&rd.il);
		}
	}
	return 0;
}

/* ------------------------------------------------------------------ solve's name-keyed read formats
 * (solve.cpp:413-428 UCSC_GFF, :487-551 UCSC_BED, :552-634 WORMBASE_GFF3).  Every accepted line adds
 * its intervals to the read of its name, in file order; chromosome and strand are those of the last
 * accepted line.  Accepted lines are collected first, then grouped by name with a stable sort. */

typedef struct { char *name; int chrom, strand; long s[2], e[2]; int n; size_t ord; } named_line;
typedef struct { named_line *v; size_t n, cap; } named_vec;

static void named_push(named_vec *nv, const char *name, size_t nn, int chrom, int strand, long s0, long e0, int two, long s1, long e1) {
	if (nv->n == nv->cap) { nv->cap = nv->cap ? nv->cap * 2 : 1024; nv->v = (named_line *)xrealloc(nv->v, sizeof(named_line) * nv->cap); }
	named_line *l = &nv->v[nv->n];
	l->name = xstrndup(name, nn); l->chrom = chrom; l->strand = strand;
	l->s[0] = s0; l->e[0] = e0; l->s[1] = s1; l->e[1] = e1; l->n = two ? 2 : 1; l->ord = nv->n;
	nv->n++;
}
static int cmp_named(const void *pa, const void *pb) {
	const named_line *a = (const named_line *)pa, *b = (const named_line *)pb;
	int c = strcmp(a->name, b->name);
	if (c) return c;
	return a->ord < b->ord ? -1 : (a->ord > b->ord ? 1 : 0);
}
static void named_finish(named_vec *nv, readvec *out) {
	qsort(nv->v, nv->n, sizeof(named_line), cmp_named);
	size_t i = 0;
	while (i < nv->n) {
		size_t j = i;
		read_t rd; memset(&rd, 0, sizeof rd);
		rd.name = nv->v[i].name;
		while (j < nv->n && strcmp(nv->v[j].name, rd.name) == 0) {
			rd.chrom = nv->v[j].chrom; rd.strand = nv->v[j].strand;
			for (int k = 0; k < nv->v[j].n; k++) lsqo_il_add(&rd.il, nv->v[j].s[k], nv->v[j].e[k]);
			if (j > i) free(nv->v[j].name);
			j++;
		}
		if (rd.il.n > 0) {
			rd.start = rd.il.s[0]; rd.end = rd.il.e[rd.il.n - 1];
			if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 1024; out->v = (read_t *)xrealloc(out->v, sizeof(read_t) * out->cap); }
			out->v[out->n++] = rd;
		} else il_free(&rd.il);        /* a read with no interval at all: undefined behaviour in the reference */
		i = j;
	}
	free(nv->v);
}

/* tab fields: field k (0-based) of a line as the reference finds it with find('\t') chains */
static int tab_field(const char *line, size_t n, int k, const char **f, size_t *fn) {
	size_t pos = 0;
	for (int i = 0; i < k; i++) {
		const char *t = (const char *)memchr(line + pos, '\t', n - pos);
		if (!t) return 0;
		pos = (size_t)(t - line) + 1;
	}
	const char *t = (const char *)memchr(line + pos, '\t', n - pos);
	*f = line + pos; *fn = t ? (size_t)(t - (line + pos)) : n - pos;
	return 1;
}

/* returns 0 ok, 1 lexical_cast error */
static int load_named_reads(const char *fmt, const text_t *t, chrom_regions *covered, strtab *chroms, strtab *strands, readvec *out) {
	named_vec nv; memset(&nv, 0, sizeof nv);
	size_t pos = 0; const char *line; size_t n;
	if (strcmp(fmt, "UCSC_GFF") == 0) {
		next_line(t, &pos, &line, &n); next_line(t, &pos, &line, &n);            /* two header lines */
		while (next_line(t, &pos, &line, &n)) {
			const char *tok[9]; size_t tn[9]; int nt = 0; size_t p = 0;
			while (nt < 9 && next_tok(line, n, &p, &tok[nt], &tn[nt])) nt++;
			long start = 0, end = 0;
			if (nt < 9 || !stream_long(tok[3], tn[3], &start) || !stream_long(tok[4], tn[4], &end)) continue;   /* malformed: the reference reads garbage */
			if (lsqo_il_contains(regions_get(covered, tok[0], tn[0]), start - 1, end))
				named_push(&nv, tok[8], tn[8], strtab_id(chroms, tok[0], tn[0]), strtab_id(strands, tok[6], tn[6]), start - 1, end, 0, 0, 0);
		}
	} else if (strcmp(fmt, "UCSC_BED") == 0) {
		next_line(t, &pos, &line, &n);                                            /* one header line */
		while (next_line(t, &pos, &line, &n)) {
			const char *f[12]; size_t fn[12];
			for (int k = 0; k < 12; k++) if (!tab_field(line, n, k, &f[k], &fn[k])) { f[k] = ""; fn[k] = 0; }
			long start, end, nb;
			if (!lexical_cast_long(f[1], fn[1], &start) || !lexical_cast_long(f[2], fn[2], &end)) return 1;
			char cbuf[512]; snprintf(cbuf, sizeof cbuf, "%.*s", (int)fn[0], f[0]);
			if (!lsqo_il_contains(regions_get(covered, f[0], fn[0]), start, end)) continue;
			if (!lexical_cast_long(f[9], fn[9], &nb)) return 1;
			/* blockSizes (col 11) and blockStarts (col 12), comma separated, the first blockCount of them */
			size_t ps = 0, pz = 0;
			int chrom = strtab_id(chroms, f[0], fn[0]), strand = strtab_id(strands, f[5], fn[5]);
			if (nb <= 0) named_push(&nv, f[3], fn[3], chrom, strand, 0, 0, 0, 0, 0);   /* a name with no interval */
			for (long i = 0; i < nb; i++) {
				size_t qs = ps; while (qs < fn[11] && f[11][qs] != ',') qs++;
				size_t qz = pz; while (qz < fn[10] && f[10][qz] != ',') qz++;
				long istart, isize;
				if (!lexical_cast_long(f[11] + ps, qs - ps, &istart) || !lexical_cast_long(f[10] + pz, qz - pz, &isize)) return 1;
				named_push(&nv, f[3], fn[3], chrom, strand, start + istart, start + istart + isize, 0, 0, 0);
				ps = qs + 1; pz = qz + 1;
			}
		}
	} else {   /* WORMBASE_GFF3 */
		while (next_line(t, &pos, &line, &n)) {
			const char *f[9]; size_t fn[9];
			for (int k = 0; k < 9; k++) if (!tab_field(line, n, k, &f[k], &fn[k])) { f[k] = ""; fn[k] = 0; }
			long start, end;
			if (!lexical_cast_long(f[3], fn[3], &start) || !lexical_cast_long(f[4], fn[4], &end)) return 1;
			char chr[512]; int cn = snprintf(chr, sizeof chr, "chr%.*s", (int)fn[0], f[0]);
			/* attributes: only those closed by ';' are looked at */
			const char *ri = f[8]; size_t rn = fn[8];
			const char *rname = ""; size_t rnn = 0;
			int found_parent = 0; unsigned long start2 = 0, end2 = 0;
			size_t a0 = 0;
			for (;;) {
				size_t a1 = a0; while (a1 < rn && ri[a1] != ';') a1++;
				if (a1 >= rn) break;
				const char *at = ri + a0; size_t an = a1 - a0;
				if (an >= 7 && memcmp(at, "Target=", 7) == 0) {
					const char *v = at + 7; size_t vn = an - 7, sp = 0;
					while (sp < vn && v[sp] != ' ') sp++;
					rname = v; rnn = sp;
				} else if (an >= 7 && memcmp(at, "Parent=", 7) == 0) {
					const char *v = at + 7; size_t vn = an - 7;
					if (vn >= 7 && memcmp(v, "intron_", 7) == 0) {
						found_parent = 1;
						size_t u0 = 7; while (u0 < vn && v[u0] != '_') u0++;          /* end of the intron's own name */
						size_t u1 = u0 + 1; while (u1 < vn && v[u1] != '_') u1++;
						size_t u2 = u1 + 1; while (u2 < vn && v[u2] != '_') u2++;
						if (u0 >= vn || !lexical_cast_ulong_n(v + u0 + 1, u1 - u0 - 1, &start2)) return 1;
						if (u1 >= vn || !lexical_cast_ulong_n(v + u1 + 1, (u2 < vn ? u2 : vn) - u1 - 1, &end2)) return 1;
					}
				}
				a0 = a1 + 1;
			}
			if (lsqo_il_contains(regions_get(covered, chr, (size_t)cn), start - 1, end)) {
				int chrom = strtab_id(chroms, chr, (size_t)cn), strand = strtab_id(strands, f[6], fn[6]);
				if (!found_parent) named_push(&nv, rname, rnn, chrom, strand, start - 1, end, 0, 0, 0);
				else named_push(&nv, rname, rnn, chrom, strand, start - 1, (long)start2 - 1, 1, (long)end2, end);
			}
		}
	}
	named_finish(&nv, out);
	return 0;
}

/* ------------------------------------------------------------------ read index order (count.cpp:64-85) */

static strtab *g_chroms, *g_strands;

static const char *read_name(const read_t *r, char *buf, size_t cap) {
	if (r->name) return r->name;
	snprintf(buf, cap, "read-%lu", r->line_num);
	return buf;
}
static int cmp_read(const void *pa, const void *pb) {
	const read_t *a = (const read_t *)pa, *b = (const read_t *)pb;
	if (a->chrom != b->chrom) { int c = strcmp(g_chroms->v[a->chrom], g_chroms->v[b->chrom]); if (c) return c; }
	if (a->start != b->start) return a->start < b->start ? -1 : 1;
	if (a->end != b->end) return a->end < b->end ? -1 : 1;
	if (a->strand != b->strand) { int c = strcmp(g_strands->v[a->strand], g_strands->v[b->strand]); if (c) return c; }
	char x[40], y[40];
	return strcmp(read_name(a, x, sizeof x), read_name(b, y, sizeof y));
}
/* RinfopComp(read, gp): is read < (chrom, start, end, strand, name) ? */
static int read_less_than_key(const read_t *a, const char *chrom, long start, long end, const char *strand, const char *name) {
	int c = strcmp(g_chroms->v[a->chrom], chrom);
	if (c) return c < 0;
	if (a->start != start) return a->start < start;
	if (a->end != end) return a->end < end;
	c = strcmp(g_strands->v[a->strand], strand);
	if (c) return c < 0;
	char x[40];
	return strcmp(read_name(a, x, sizeof x), name) < 0;
}

/* ------------------------------------------------------------------ Read_single::build (read.h:204-274) */

typedef struct { int idx[64]; int n; unsigned long read_length; } built_t;
/* exon_indices kept as an ascending list; the reference collects them in a std::set */
void lsqo_read_build(const ilist *rd, const seg_t *segs, int nseg, int *out_idx, int *out_n, unsigned long *out_len, int idx_cap) {
	unsigned long matching = 0;
	int n_idx = 0;
	int found_start = 0;
	int il_idx = 0;
	int it = 0;
	long cur_start = 0, cur_end = 0, cur_exon_start = 0;
	while (il_idx < rd->n) {
		cur_start = rd->s[il_idx];
		cur_end = rd->e[il_idx];
		while (it != nseg && segs[it].start < cur_end) {
			if (segs[it].start > cur_exon_start) cur_exon_start = segs[it].start;
			if (cur_start >= cur_exon_start && cur_start < segs[it].end) {
				if (!found_start) {
					found_start = 1;
				} else if (cur_start > cur_exon_start) {
					break;
				}
				/* index_set.insert(distance(begin, itr)) */
				int dup = 0;
				for (int q = 0; q < n_idx; q++) if (out_idx[q] == it) dup = 1;
				if (!dup && n_idx < idx_cap) out_idx[n_idx++] = it;
				cur_exon_start = segs[it].end < cur_end ? segs[it].end : cur_end;
				matching += (unsigned long)(cur_exon_start - cur_start);
				if (cur_end < segs[it].end) {
					cur_start = cur_end;
					++il_idx;
					break;
				} else if (cur_end == segs[it].end) {
					cur_start = cur_end;
					++il_idx;
				} else {
					cur_start = segs[it].end;
				}
			} else if (cur_exon_start > segs[it].start && cur_exon_start < segs[it].end) {
				break;
			}
			it++;
		}
		if (cur_start == cur_end) continue; else break;
	}
	/* ascending order (std::set iteration) */
	for (int a = 1; a < n_idx; a++) { int v = out_idx[a], b = a - 1; while (b >= 0 && out_idx[b] > v) { out_idx[b + 1] = out_idx[b]; b--; } out_idx[b + 1] = v; }
	*out_n = n_idx;
	*out_len = matching;
}

/* read.h:44-79 */
long lsqo_connected_compat(const int *exon_indices, int n, const int *iso_exon_indices, int m) {
	int i = 0, j = 0;
	int is_compatible = 0, found_first = 0;
	long first_match_idx = 0;
	while (i != n) {
		if (j == m) { is_compatible = 0; break; }
		if (exon_indices[i] != iso_exon_indices[j]) {
			if (!found_first) { j++; first_match_idx++; }
			else { is_compatible = 0; break; }
		} else {
			if (!found_first) { found_first = 1; is_compatible = 1; }
			i++; j++;
		}
	}
	return is_compatible ? first_match_idx : -1;
}

/* ------------------------------------------------------------------ ARS (accessible_read_starts.h) */

/* total ARS length of one isoform; short_read != 0 -> AccessibleShortReadStarts with
 * min_partial_exon_size = 0 (:221-274), else AccessibleReadStarts (:48-89). */
unsigned long lsqo_ars_total(const unsigned long *exon_len, const int *iso_idx, int n, unsigned long read_length, int short_read) {
	unsigned long total_length = 0, iso_length = 0, iso_total_length = 0;
	const unsigned long min_partial = 0;
	for (int i = 0; i < n; i++) iso_total_length += exon_len[iso_idx[i]];
	for (int i = 0; i < n; i++) {
		unsigned long l = exon_len[iso_idx[i]];
		iso_length += l;
		if (iso_length + read_length > iso_total_length) {
			long v = (long)l + 1 - (long)(iso_length + read_length - iso_total_length);
			unsigned long last2 = (unsigned long)(v > 0 ? v : 0);
			total_length += last2;            /* [0,last2) or nothing */
			break;
		}
		if (short_read) {
			ilist ars; memset(&ars, 0, sizeof ars);
			long last = (long)l - (long)read_length + 1; if (last < 0) last = 0;
			lsqo_il_add(&ars, 0, last);
			long pstart = (long)l - (long)read_length + (long)min_partial; if (pstart < 0) pstart = 0;
			if (l > min_partial) lsqo_il_add(&ars, pstart, (long)(l - min_partial + 1));
			total_length += (unsigned long)il_total_length(&ars);
			il_free(&ars);
		} else {
			total_length += l;                /* [0,l) */
		}
	}
	return total_length;
}

/* ------------------------------------------------------------------ EM (read.h:592-660) */

typedef struct { double *g; unsigned long n; } dmat;   /* n x K, row-major */

static double reads_log_likelihood(int K, const dmat *ms, int nm, const double *theta) {
	double ll = 0;
	for (int m = 0; m < nm; m++)
		for (unsigned long i = 0; i < ms[m].n; i++) {
			double s = 0;
			for (int k = 0; k < K; k++) s += theta[k] * ms[m].g[i * (unsigned long)K + (unsigned long)k];
			ll += log(s);
		}
	return ll;
}
static void em_step(int K, const dmat *ms, int nm, const double *old_theta, double *new_theta) {
	for (int k = 0; k < K; k++) {
		double sum_zeta = 0, num_total_reads = 0;
		for (int m = 0; m < nm; m++) {
			num_total_reads += (double)ms[m].n;
			for (unsigned long i = 0; i < ms[m].n; i++) {
				const double *row = ms[m].g + i * (unsigned long)K;
				double s = 0;
				for (int k2 = 0; k2 < K; k2++) s += old_theta[k2] * row[k2];
				if (s > 0) {
					double local = old_theta[k] * row[k];
					if (local > 0) sum_zeta += local / s;
				}
			}
		}
		new_theta[k] = sum_zeta / num_total_reads;
	}
}
/* returns the number of iterations */
unsigned long lsqo_em(int K, const dmat *ms, int nm, double *theta) {
	double *old_theta = (double *)xmalloc(sizeof(double) * (size_t)K);
	for (int k = 0; k < K; k++) theta[k] = 1.0 / (double)K;
	double ll, old_ll;
	unsigned long iters = 0;
	do {
		memcpy(old_theta, theta, sizeof(double) * (size_t)K);
		old_ll = reads_log_likelihood(K, ms, nm, old_theta);
		em_step(K, ms, nm, old_theta, theta);
		ll = reads_log_likelihood(K, ms, nm, theta);
		iters++;
	} while (fabs(1.0 - old_ll / ll) > 1E-6);
	free(old_theta);
	return iters;
}

typedef struct { char *g, *i; int ord; } pair_t;
static int cmp_pair(const void *a, const void *b) {
	const pair_t *x = (const pair_t *)a, *y = (const pair_t *)b;
	int c = strcmp(x->g, y->g);
	if (c) return c;
	return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static ga_entry **g_sort_gas;
static int cmp_ga_ord(const void *a, const void *b) {
	int x = *(const int *)a, y = *(const int *)b;
	int c = strcmp(g_sort_gas[x]->name, g_sort_gas[y]->name);
	if (c) return c;
	return x < y ? -1 : (x > y);
}

/* ------------------------------------------------------------------ the two executables */

typedef struct {
	int is_solve;
	const char *isoform_format, *isoforms_path, *g2i_format, *g2i_path;
	unsigned long gene_begin, gene_end;
	int M;
	const char **read_formats, **read_types, **reads_paths;
	unsigned long *exp_len;
	double *total_read_bases;
} params_t;


/* ------------------------------------------------------------------ FIM (fim.h, linalg.h)
 * PARITY UNPINNED: no translation unit of the reference includes fim.h, no binary produces
 * these numbers; what follows restates the header's arithmetic for the HIP path to be checked
 * against (the expected Fisher information of theta_1..theta_{K-1} per read, and the two
 * variance estimates made from it). */

/* per-exon ARS lengths of one isoform (accessible_read_starts.h:48-89 MEDIUM, :221-274 SHORT with
 * min_partial_exon_size = 0): in every exon the accessible starts are one interval [0, len) --
 * SHORT: [0,max(l-R+1,0)) and [max(l-R,0),l+1) merge to [0,l+1) -- so only the lengths are kept;
 * cum[] = iso_ARS_total_lengths */
static void fim_ars_lengths(const gene_t *g, int k, unsigned long R, int short_read, unsigned long *cum) {
	const int n = g->iso_n[k];
	unsigned long total = 0, iso_length = 0, iso_total = 0;
	for (int i = 0; i < n; i++) iso_total += g->exon_len[g->iso_idx[k][i]];
	int i = 0;
	for (; i < n; i++) {
		const unsigned long l = g->exon_len[g->iso_idx[k][i]];
		iso_length += l;
		if (iso_length + R > iso_total) {
			long v = (long)l + 1 - (long)(iso_length + R - iso_total);
			total += (unsigned long)(v > 0 ? v : 0);
			cum[i] = total;
			for (int j = i + 1; j < n; j++) cum[j] = total;
			return;
		}
		total += short_read ? l + 1 : l;          /* (an exon of length 0 cannot exist: segments are non-empty) */
		cum[i] = total;
	}
}

/* accessible_read_starts.h:131-182 with one interval [0, len) per exon */
static unsigned long fim_ars_to_iso_start(const gene_t *g, int k, const unsigned long *ars_cum, unsigned long a) {
	const int n = g->iso_n[k];
	int e = 0;
	while (e < n && !(ars_cum[e] > a)) e++;                    /* upper_bound */
	unsigned long ars_before = e > 0 ? ars_cum[e - 1] : 0, exon_before = 0;
	for (int i = 0; i < e; i++) exon_before += g->exon_len[g->iso_idx[k][i]];
	return exon_before + (a - ars_before);
}

/* read.h:276-329 generate_read with a fixed start: the isoform's exons from the one that holds
 * the start to the one that holds the last base */
static void fim_generate_read(const gene_t *g, int k, unsigned long read_start, unsigned long R, int *idx, int *nidx) {
	const int n = g->iso_n[k];
	unsigned long cum[64]; unsigned long t = 0;
	for (int i = 0; i < n; i++) { t += g->exon_len[g->iso_idx[k][i]]; cum[i] = t; }
	const unsigned long read_end = read_start + R;
	int se = 0; while (se < n && !(cum[se] > read_start)) se++;     /* upper_bound(read_start) */
	int ee = 0; while (ee < n && cum[ee] < read_end) ee++;         /* lower_bound(read_end) */
	*nidx = 0;
	if (se >= n || ee >= n) return;                               /* the reference asserts */
	for (int i = se; i <= ee; i++) idx[(*nidx)++] = g->iso_idx[k][i];
}

/* fim.h:320-367 ofim + :115-158 bruteforce_fim.  I is (K-1) x (K-1), row-major. */
static void fim_bruteforce(const gene_t *g, const double *theta, const double *G, unsigned long R, int short_read, double *I) {
	const int K = g->K, D = K - 1;
	for (int i = 0; i < D * D; i++) I[i] = 0;
	for (int k = 0; k < K; k++) {
		if (theta[k] == 0) continue;
		unsigned long ars_cum[64];
		fim_ars_lengths(g, k, R, short_read, ars_cum);
		const unsigned long ars_total = ars_cum[g->iso_n[k] - 1];
		for (unsigned long a = 0; a < ars_total; a++) {
			int idx[64], nidx;
			fim_generate_read(g, k, fim_ars_to_iso_start(g, k, ars_cum, a), R, idx, &nidx);
			if (nidx == 0) continue;
			const double log_scaler = log(G[k]) + log(theta[k]);
			double v[16];
			for (int j = 0; j < K; j++) v[j] = lsqo_connected_compat(idx, nidx, g->iso_idx[j], g->iso_n[j]) >= 0 ? G[j] : 0.0;
			double sum = 0;
			for (int j = 0; j < K; j++) if (v[j] > 0 && theta[j] > 0) sum += theta[j] * v[j];
			const double log_prod = log(sum) + log(sum);
			for (int p = 0; p < D; p++)
				for (int q = 0; q < D; q++) {
					double m = (v[p] - v[K - 1]) * (v[q] - v[K - 1]);
					if (m != 0) {
						const int sign = m > 0 ? 1 : -1;
						if (m < 0) m = -m;
						m = log(m) - log_prod;
						I[p * D + q] += sign > 0 ? exp(m + log_scaler) : -exp(m + log_scaler);
					}
				}
		}
	}
}

/* fim.h:64-72 */
static double fim_var_by_diag(const double *I, int D) {
	double sum = 0;
	for (int p = 0; p < D; p++) sum += 1.0 / I[p * D + p];
	return sum;
}

/* linalg.h:28-71: gsl_linalg_LU_decomp + gsl_linalg_LU_invert (GSL 1.12: Gaussian elimination with
 * partial pivoting, then the columns of the identity solved one by one), and fim.h:74-93 */
static double fim_var_by_inv(const double *I, int D) {
	double A[25], inv[25]; int perm[5];
	if (D == 0) return 0;
	for (int i = 0; i < D * D; i++) A[i] = I[i];
	for (int i = 0; i < D; i++) perm[i] = i;
	for (int j = 0; j < D - 1; j++) {
		double amax = fabs(A[j * D + j]); int ip = j;
		for (int i = j + 1; i < D; i++) { const double a = fabs(A[i * D + j]); if (a > amax) { amax = a; ip = i; } }
		if (ip != j) { for (int c = 0; c < D; c++) { const double t = A[j * D + c]; A[j * D + c] = A[ip * D + c]; A[ip * D + c] = t; } const int t = perm[j]; perm[j] = perm[ip]; perm[ip] = t; }
		const double ajj = A[j * D + j];
		if (ajj != 0.0)
			for (int i = j + 1; i < D; i++) {
				const double aij = A[i * D + j] / ajj;
				A[i * D + j] = aij;
				for (int c = j + 1; c < D; c++) A[i * D + c] -= aij * A[j * D + c];
			}
	}
	for (int col = 0; col < D; col++) {
		double x[5];
		for (int i = 0; i < D; i++) x[i] = perm[i] == col ? 1.0 : 0.0;        /* P e_col */
		for (int i = 1; i < D; i++) for (int c = 0; c < i; c++) x[i] -= A[i * D + c] * x[c];          /* L y = b (unit diagonal) */
		for (int i = D - 1; i >= 0; i--) { for (int c = i + 1; c < D; c++) x[i] -= A[i * D + c] * x[c]; x[i] /= A[i * D + i]; }   /* U x = y */
		for (int i = 0; i < D; i++) inv[i * D + col] = x[i];
	}
	double sum = 0;
	for (int p = 0; p < D; p++) { for (int q = 0; q < D; q++) sum += inv[p * D + q]; sum += inv[p * D + p]; }
	return sum;
}

/* optional exact side-output for tests: per (gene, method) supports and bases, per
 * (gene, iso) counts, theta, EM iterations -- filled when non-NULL */
typedef struct {
	int n_genes;
	char **gname;
	int *K;
	unsigned long **supports;     /* [gene][M] */
	unsigned long **bases;        /* [gene][M] */
	unsigned long **iso_count;    /* [gene][K] */
	double **theta;               /* [gene][K] */
	double *logll;
	unsigned long *iters;
	unsigned long n_loaded[16];
	double **fim;                 /* [gene][M * (K-1)^2], filled when the environment has LSQO_FIM (parity unpinned) */
	double **fim_var;             /* [gene][M * 2]: by_diag, by_inv */
} exact_t;

static void err_line(const char *msg) { fprintf(stderr, "[oracle ERROR] %s\n", msg); }

static int run(const params_t *P, sink *out, exact_t *ex) {
	text_t txt;
	/* ---- isoforms: LH_GENE_TXT (count.cpp:141-171); `solve` also takes UCSC_GENE_TXT
	 * (jsc/bioinfo/gene_anno.hpp:59-117), GENELETS_GFF3, UCSC_GFF and WORMBASE_GFF2
	 * (solve.cpp:158-296).  The reference opens the file before it looks at the literal. */
	if (!load_text(P->isoforms_path, &txt)) { err_line("cannot open isoforms file"); return 134; }
	const int f_lh = strcmp(P->isoform_format, "LH_GENE_TXT") == 0;
	const int f_ucsc = P->is_solve && strcmp(P->isoform_format, "UCSC_GENE_TXT") == 0;
	const int f_gff = P->is_solve && strcmp(P->isoform_format, "UCSC_GFF") == 0;
	const int f_worm = P->is_solve && strcmp(P->isoform_format, "WORMBASE_GFF2") == 0;
	const int f_gen = P->is_solve && strcmp(P->isoform_format, "GENELETS_GFF3") == 0;
	if (!(f_lh || f_ucsc || f_gff || f_worm || f_gen)) { err_line("Unknown file format error"); return 1; }
	ga_entry **gas = NULL; int nga = 0, gacap = 0;
	if (f_lh || f_ucsc) {
		size_t pos = 0; const char *line; size_t n;
		const int o = f_ucsc ? 2 : 0;          /* cdsStart cdsEnd sit between txEnd and exonCount */
		while (next_line(&txt, &pos, &line, &n)) {
			ga_entry *g = (ga_entry *)calloc(1, sizeof(ga_entry));
			const char *tok[10]; size_t tn[10]; int nt = 0; size_t p = 0;
			while (nt < 8 + o && next_tok(line, n, &p, &tok[nt], &tn[nt])) nt++;
			g->name = xstrndup(nt > 0 ? tok[0] : "", nt > 0 ? tn[0] : 0);
			g->chrom = xstrndup(nt > 1 ? tok[1] : "", nt > 1 ? tn[1] : 0);
			g->strand = xstrndup(nt > 2 ? tok[2] : "", nt > 2 ? tn[2] : 0);
			int ok = nt > 3 && lexical_cast_long(tok[3], tn[3], &g->txStart);
			ok = ok && nt > 4 && lexical_cast_long(tok[4], tn[4], &g->txEnd);
			long ec = 0, cds = 0;
			if (f_ucsc) ok = ok && nt > 6 && lexical_cast_long(tok[5], tn[5], &cds) && lexical_cast_long(tok[6], tn[6], &cds);
			ok = ok && nt > 5 + o && lexical_cast_long(tok[5 + o], tn[5 + o], &ec) && ec >= 0;
			if (ok) {
				g->exonCount = (unsigned long)ec;
				if (nt > 6 + o) g->nStarts = split_atol(tok[6 + o], tn[6 + o], &g->exonStarts);
				if (nt > 7 + o) g->nEnds = split_atol(tok[7 + o], tn[7 + o], &g->exonEnds);
			}
			if (nga == gacap) { gacap = gacap ? gacap * 2 : 256; gas = (ga_entry **)xrealloc(gas, sizeof(ga_entry *) * (size_t)gacap); }
			gas[nga++] = g;
		}
		free(txt.data);
	} else {
		/* one line per exon: the exons of a name are merged with add_interval in file order; chromosome
		 * and strand are those of the name's last line; names are walked in std::set order */
		typedef struct { char *name, *chrom, *strand; ilist il; int ord; } grp_t;
		grp_t *grp = NULL; int ngrp = 0, grpcap = 0;
		size_t pos = 0; const char *line; size_t n; int li = 0;
		while (next_line(&txt, &pos, &line, &n)) {
			if (!f_worm && li++ < 2) continue;           /* two header lines (solve.cpp:164-165,240-241) */
			const char *tok[12]; size_t tn[12]; int nt = 0; size_t p = 0;
			while (nt < 12 && next_tok(line, n, &p, &tok[nt], &tn[nt])) nt++;
			long start = 0, end = 0;
			char chrom[512];
			if (f_gen) {
				if (nt < 3 || tn[2] != 4 || memcmp(tok[2], "exon", 4) != 0) continue;
				if (nt < 9 || !stream_long(tok[3], tn[3], &start) || !stream_long(tok[4], tn[4], &end)) { err_line("malformed GENELETS_GFF3 exon line (reference reads uninitialised values)"); return 139; }
				snprintf(chrom, sizeof chrom, "chr%.*s", (int)tn[0], tok[0]);
				/* attributes ';'-separated; Parent=a,b names the isoforms */
				const char *inf = tok[8]; size_t in = tn[8], i = 0;
				while (i <= in) {
					size_t j = i; while (j < in && inf[j] != ';') j++;
					if (j - i > 7 && memcmp(inf + i, "Parent=", 7) == 0) {
						size_t u = i + 7;
						while (u <= j) {
							size_t v = u; while (v < j && inf[v] != ',') v++;
							if (v > u) {
								int k; for (k = 0; k < ngrp; k++) if (strlen(grp[k].name) == v - u && memcmp(grp[k].name, inf + u, v - u) == 0) break;
								if (k == ngrp) { if (ngrp == grpcap) { grpcap = grpcap ? grpcap * 2 : 64; grp = (grp_t *)xrealloc(grp, sizeof(grp_t) * (size_t)grpcap); } memset(&grp[k], 0, sizeof(grp_t)); grp[k].name = xstrndup(inf + u, v - u); ngrp++; }
								free(grp[k].chrom); free(grp[k].strand);
								grp[k].chrom = xstrndup(chrom, strlen(chrom)); grp[k].strand = xstrndup(tok[6], tn[6]);
								lsqo_il_add(&grp[k].il, start - 1, end);
							}
							u = v + 1;
						}
					}
					i = j + 1;
				}
			} else {
				const int ni = f_worm ? 9 : 8;             /* WORMBASE_GFF2 has one more column before the name */
				if (nt <= ni || !stream_long(tok[3], tn[3], &start) || !stream_long(tok[4], tn[4], &end)) { err_line("malformed GFF line (reference reads uninitialised values)"); return 139; }
				if (f_worm) snprintf(chrom, sizeof chrom, "chr%.*s", (int)tn[0], tok[0]); else snprintf(chrom, sizeof chrom, "%.*s", (int)tn[0], tok[0]);
				const char *nm = tok[ni]; size_t nn = tn[ni];
				while (nn > 0 && nm[0] == '"') { nm++; nn--; }       /* trim_if(iname, is_any_of("\"")) */
				while (nn > 0 && nm[nn - 1] == '"') nn--;
				int k; for (k = 0; k < ngrp; k++) if (strlen(grp[k].name) == nn && memcmp(grp[k].name, nm, nn) == 0) break;
				if (k == ngrp) { if (ngrp == grpcap) { grpcap = grpcap ? grpcap * 2 : 64; grp = (grp_t *)xrealloc(grp, sizeof(grp_t) * (size_t)grpcap); } memset(&grp[k], 0, sizeof(grp_t)); grp[k].name = xstrndup(nm, nn); ngrp++; }
				free(grp[k].chrom); free(grp[k].strand);
				grp[k].chrom = xstrndup(chrom, strlen(chrom)); grp[k].strand = xstrndup(tok[6], tn[6]);
				lsqo_il_add(&grp[k].il, start - 1, end);
			}
		}
		free(txt.data);
		for (int k = 0; k < ngrp; k++) {
			ga_entry *g = (ga_entry *)calloc(1, sizeof(ga_entry));
			g->name = grp[k].name; g->chrom = grp[k].chrom; g->strand = grp[k].strand;
			g->exonCount = (unsigned long)grp[k].il.n;
			g->nStarts = g->nEnds = grp[k].il.n;
			g->exonStarts = (long *)xmalloc(sizeof(long) * (size_t)(grp[k].il.n ? grp[k].il.n : 1));
			g->exonEnds = (long *)xmalloc(sizeof(long) * (size_t)(grp[k].il.n ? grp[k].il.n : 1));
			for (int q = 0; q < grp[k].il.n; q++) { g->exonStarts[q] = grp[k].il.s[q]; g->exonEnds[q] = grp[k].il.e[q]; }
			if (nga == gacap) { gacap = gacap ? gacap * 2 : 256; gas = (ga_entry **)xrealloc(gas, sizeof(ga_entry *) * (size_t)gacap); }
			gas[nga++] = g;
		}
		free(grp);
	}
	/* iname2gap: last duplicate wins (count.cpp:176-179); sorted (name, position) view */
	g_sort_gas = gas;
	int *ord = (int *)xmalloc(sizeof(int) * (size_t)(nga ? nga : 1));
	for (int i = 0; i < nga; i++) ord[i] = i;
	qsort(ord, (size_t)nga, sizeof(int), cmp_ga_ord);
	/* ---- gene -> isoform map: UCSC_GENE2ISOFORM (count.cpp:188-195); `solve` also takes
	 * WORMBASE_GENE2ISOFORMS: gene iso1;iso2;... (solve.cpp:318-329) */
	if (!load_text(P->g2i_path, &txt)) { err_line("cannot open g2i file"); return 134; }
	const int g_worm = P->is_solve && strcmp(P->g2i_format, "WORMBASE_GENE2ISOFORMS") == 0;
	if (strcmp(P->g2i_format, "UCSC_GENE2ISOFORM") != 0 && !g_worm) { err_line("Unknown file format error"); return 1; }
	gene_t *genes = NULL; int ngenes = 0, gcap = 0;
	{
		/* collect (gene, isoform) pairs, then group by gene name keeping file order inside a
		 * gene: std::set<string> order for genes, push_back order for isoforms */
		pair_t *pairs = NULL; int npairs = 0, pcap = 0;
		size_t pos = 0; const char *line; size_t n;
		while (next_line(&txt, &pos, &line, &n)) {
			const char *a = "", *b = ""; size_t an = 0, bn = 0; size_t p = 0;
			next_tok(line, n, &p, &a, &an);
			next_tok(line, n, &p, &b, &bn);
			size_t u = 0;
			do {
				size_t v = u;
				if (g_worm) { while (v < bn && b[v] != ';') v++; } else v = bn;
				if (!g_worm || v > u) {
					if (npairs == pcap) { pcap = pcap ? pcap * 2 : 256; pairs = (pair_t *)xrealloc(pairs, sizeof(pair_t) * (size_t)pcap); }
					pairs[npairs].g = xstrndup(a, an); pairs[npairs].i = xstrndup(b + u, v - u); pairs[npairs].ord = npairs; npairs++;
				}
				u = v + 1;
			} while (g_worm && u < bn);
		}
		qsort(pairs, (size_t)npairs, sizeof(pair_t), cmp_pair);
		for (int q = 0; q < npairs; q++) {
			if (ngenes == 0 || strcmp(genes[ngenes - 1].gname, pairs[q].g) != 0) {
				if (ngenes == gcap) { gcap = gcap ? gcap * 2 : 256; genes = (gene_t *)xrealloc(genes, sizeof(gene_t) * (size_t)gcap); }
				memset(&genes[ngenes], 0, sizeof(gene_t));
				genes[ngenes].gname = pairs[q].g;
				ngenes++;
			}
			gene_t *g = &genes[ngenes - 1];
			/* last isoform record with this name */
			int lo = 0, hi = nga;
			while (lo < hi) { int mid = lo + (hi - lo) / 2; if (strcmp(gas[ord[mid]]->name, pairs[q].i) <= 0) lo = mid + 1; else hi = mid; }
			ga_entry *ga = (lo > 0 && strcmp(gas[ord[lo - 1]]->name, pairs[q].i) == 0) ? gas[ord[lo - 1]] : NULL;
			if (!ga) { err_line("g2i names an isoform that is not in the isoforms file (reference dereferences NULL)"); return 139; }
			if (g->niso == g->isocap) { g->isocap = g->isocap ? g->isocap * 2 : 4; g->isos = (ga_entry **)xrealloc(g->isos, sizeof(ga_entry *) * (size_t)g->isocap); }
			g->isos[g->niso++] = ga;
		}
		free(pairs);
		free(txt.data);
	}
	free(ord);
	/* ---- gene selection by index into the bytewise-sorted name set (count.cpp:204-215) */
	gene_t **sel = (gene_t **)xmalloc(sizeof(gene_t *) * (size_t)(ngenes ? ngenes : 1));
	int nsel = 0;
	for (int i = 0; i < ngenes; i++)
		if ((unsigned long)i >= P->gene_begin && (unsigned long)i < P->gene_end) sel[nsel++] = &genes[i];
	/* ---- per selected gene: segments, covered regions, span, isoform arrays (count.cpp:235-258) */
	chrom_regions covered; memset(&covered, 0, sizeof covered);
	for (int gi = 0; gi < nsel; gi++) {
		gene_t *g = sel[gi];
		for (int k = 0; k < g->niso; k++) {
			ga_entry *ga = g->isos[k];
			for (unsigned long i = 0; i < ga->exonCount; i++) {
				if ((int)i >= ga->nStarts || (int)i >= ga->nEnds) { err_line("exonCount exceeds listed exons (reference reads past the vector)"); return 139; }
				lsqo_exonset_insert(&g->exons, ga->exonStarts[i], ga->exonEnds[i]);
				lsqo_il_add(regions_get(&covered, ga->chrom, strlen(ga->chrom)), ga->exonStarts[i], ga->exonEnds[i]);
				lsqo_il_add(&g->gene_il, ga->exonStarts[i], ga->exonEnds[i]);
			}
		}
		/* Isoforms::build -> build_isoform_array (splicing_graph.h:235-252,318-361) */
		g->K = g->niso;
		g->N = g->exons.n;
		g->chrom = g->K > 0 ? g->isos[0]->chrom : "";
		g->strand = g->K > 0 ? g->isos[0]->strand : "";
		g->iso_idx = (int **)xmalloc(sizeof(int *) * (size_t)(g->K ? g->K : 1));
		g->iso_n = (int *)xmalloc(sizeof(int) * (size_t)(g->K ? g->K : 1));
		g->iso_total_len = (unsigned long *)xmalloc(sizeof(unsigned long) * (size_t)(g->K ? g->K : 1));
		g->exon_len = (unsigned long *)xmalloc(sizeof(unsigned long) * (size_t)(g->N ? g->N : 1));
		for (int n = 0; n < g->N; n++) g->exon_len[n] = (unsigned long)(g->exons.v[n].end - g->exons.v[n].start);
		for (int k = 0; k < g->K; k++) {
			ga_entry *ga = g->isos[k];
			g->iso_idx[k] = (int *)xmalloc(sizeof(int) * (size_t)(g->N ? g->N : 1));
			g->iso_n[k] = 0;
			unsigned long iso_exon_idx = 0, total = 0;
			for (int n = 0; n < g->N; n++) {
				for (unsigned long i = iso_exon_idx; i < ga->exonCount; i++) {
					/* Exon::is_inside(a,b): a <= start && end <= b */
					if (ga->exonStarts[i] <= g->exons.v[n].start && g->exons.v[n].end <= ga->exonEnds[i]) {
						g->iso_idx[k][g->iso_n[k]++] = n;
						total += g->exon_len[n];
						iso_exon_idx = i;
						break;
					}
				}
			}
			g->iso_total_len[k] = total;
		}
	}
	/* ---- reads per sampling method (count.cpp:267-344) and index (count.cpp:348-364) */
	strtab chroms, strands; memset(&chroms, 0, sizeof chroms); memset(&strands, 0, sizeof strands);
	g_chroms = &chroms; g_strands = &strands;
	readvec *rv = (readvec *)calloc((size_t)P->M, sizeof(readvec));
	for (int m = 0; m < P->M; m++) {
		if (!load_text(P->reads_paths[m], &txt)) { err_line("cannot open reads file"); return 134; }
		const char *rf = P->read_formats[m];
		const int named = P->is_solve && (strcmp(rf, "UCSC_GFF") == 0 || strcmp(rf, "UCSC_BED") == 0 || strcmp(rf, "WORMBASE_GFF3") == 0);
		if (strcmp(rf, "MRF_SINGLE") != 0 && !named) { err_line("Unknown file format error"); return 1; }
		int rc = named ? load_named_reads(rf, &txt, &covered, &chroms, &strands, &rv[m]) : load_mrf(&txt, &covered, &chroms, &strands, &rv[m]);
		free(txt.data);
		if (rc) { err_line("Lexical_cast error when converting arguments to numeric values"); return 1; }
		qsort(rv[m].v, rv[m].n, sizeof(read_t), cmp_read);
		if (ex && m < 16) ex->n_loaded[m] = rv[m].n;
	}
	if (ex) {
		ex->n_genes = nsel;
		ex->gname = (char **)calloc((size_t)nsel + 1, sizeof(char *));
		ex->K = (int *)calloc((size_t)nsel + 1, sizeof(int));
		ex->supports = (unsigned long **)calloc((size_t)nsel + 1, sizeof(void *));
		ex->bases = (unsigned long **)calloc((size_t)nsel + 1, sizeof(void *));
		ex->iso_count = (unsigned long **)calloc((size_t)nsel + 1, sizeof(void *));
		ex->theta = (double **)calloc((size_t)nsel + 1, sizeof(void *));
		ex->logll = (double *)calloc((size_t)nsel + 1, sizeof(double));
		ex->iters = (unsigned long *)calloc((size_t)nsel + 1, sizeof(unsigned long));
		ex->fim = (double **)calloc((size_t)nsel + 1, sizeof(void *));
		ex->fim_var = (double **)calloc((size_t)nsel + 1, sizeof(void *));
	}
	/* ---- per gene (count.cpp:369-497 / solve.cpp:668-852) */
	for (int gi = 0; gi < nsel; gi++) {
		gene_t *g = sel[gi];
		int K = g->K;
		if (g->gene_il.n == 0) { err_line("gene without exons (reference reads starts[0] of an empty vector)"); return 139; }
		long gene_start = g->gene_il.s[0];
		long gene_end = g->gene_il.e[g->gene_il.n - 1];
		double *supports = (double *)xmalloc(sizeof(double) * (size_t)P->M);
		double *support_bases = (double *)xmalloc(sizeof(double) * (size_t)P->M);
		double *iso_count = (double *)calloc((size_t)(K ? K : 1), sizeof(double));
		dmat *mats = (dmat *)calloc((size_t)P->M, sizeof(dmat)); int nmat = 0;
		if (ex) {
			ex->gname[gi] = g->gname; ex->K[gi] = K;
			ex->supports[gi] = (unsigned long *)calloc((size_t)P->M, sizeof(unsigned long));
			ex->bases[gi] = (unsigned long *)calloc((size_t)P->M, sizeof(unsigned long));
			ex->iso_count[gi] = (unsigned long *)calloc((size_t)(K ? K : 1), sizeof(unsigned long));
			ex->theta[gi] = (double *)calloc((size_t)(K ? K : 1), sizeof(double));
		}
		for (int m = 0; m < P->M; m++) {
			int short_read;
			if (strcmp(P->read_types[m], "MEDIUM_READ") == 0) short_read = 0;
			else if (strcmp(P->read_types[m], "SHORT_READ") == 0) short_read = 1;
			else { err_line("Unknown read type error"); return 1; }
			double *G = (double *)xmalloc(sizeof(double) * (size_t)(K ? K : 1));
			for (int k = 0; k < K; k++) {
				if (g->iso_n[k] <= 0) { err_line("isoform without segments (reference asserts)"); return 134; }
				unsigned long ars = lsqo_ars_total(g->exon_len, g->iso_idx[k], g->iso_n[k], P->exp_len[m], short_read);
				double nd = (double)ars;                 /* read.h:333-339 */
				G[k] = nd <= 0 ? 0.0 : (double)1.0 / nd;
			}
			readvec *R = &rv[m];
			/* lower_bound over the index (count.cpp:423-430) */
			size_t lo = 0, hi = R->n;
			while (lo < hi) {
				size_t mid = lo + (hi - lo) / 2;
				if (read_less_than_key(&R->v[mid], g->chrom, gene_start, gene_end, g->strand, g->gname)) lo = mid + 1; else hi = mid;
			}
			size_t *valid = NULL; size_t nvalid = 0, vcap = 0;
			double num_valid_read_bases = 0;
			int idx[64]; int nidx; unsigned long rl;
			for (size_t it = lo; it < R->n && strcmp(chroms.v[R->v[it].chrom], g->chrom) == 0 && R->v[it].start <= gene_end; ++it) {
				read_t *rd = &R->v[it];
				if (rd->start >= gene_start) {
					lsqo_read_build(&rd->il, g->exons.v, g->N, idx, &nidx, &rl, 64);
					for (int j = 0; j < K; j++) {
						if (lsqo_connected_compat(idx, nidx, g->iso_idx[j], g->iso_n[j]) >= 0) {
							if ((double)rl / il_total_length(&rd->il) > 0.98) {
								if (nvalid == vcap) { vcap = vcap ? vcap * 2 : 256; valid = (size_t *)xrealloc(valid, sizeof(size_t) * vcap); }
								valid[nvalid++] = it;
								num_valid_read_bases += (double)rl;
								if (ex) ex->bases[gi][m] += rl;
								break;
							}
						}
					}
				}
			}
			/* second pass (count.cpp:467-481 / solve.cpp:767-790) */
			dmat dm; dm.n = nvalid; dm.g = (double *)calloc(nvalid * (size_t)(K ? K : 1) + 1, sizeof(double));
			for (size_t i = 0; i < nvalid; i++) {
				read_t *rd = &R->v[valid[i]];
				lsqo_read_build(&rd->il, g->exons.v, g->N, idx, &nidx, &rl, 64);
				for (int j = 0; j < K; j++) {
					if (lsqo_connected_compat(idx, nidx, g->iso_idx[j], g->iso_n[j]) >= 0) {
						iso_count[j] += 1.0;
						if (ex) ex->iso_count[gi][j]++;
						if ((double)rl / il_total_length(&rd->il) > 0.98) dm.g[i * (size_t)K + (size_t)j] = G[j];
					}
				}
			}
			if (nvalid > 0 && P->is_solve) mats[nmat++] = dm; else free(dm.g);
			supports[m] = (double)nvalid;
			support_bases[m] = num_valid_read_bases;
			if (ex) ex->supports[gi][m] = nvalid;
			free(valid);
			free(G);
		}
		if (!P->is_solve) {
			/* count.cpp:486-492 */
			for (int i = 0; i < K; i++) {
				sink_str(out, g->gname); sink_str(out, "\t");
				for (int m = 0; m < P->M; m++) { sink_dbl(out, supports[m]); sink_str(out, "\t"); }
				sink_str(out, g->isos[i]->name); sink_str(out, "\t"); sink_dbl(out, iso_count[i]); sink_str(out, "\n");
			}
		} else {
			/* solve.cpp:797-847 */
			double *theta = (double *)calloc((size_t)(K ? K : 1), sizeof(double));
			unsigned long iters = 0;
			if (nmat == 0) { for (int k = 0; k < K; k++) theta[k] = 1.0 / (double)K; }
			else if (K == 1) theta[0] = 1;
			else iters = lsqo_em(K, mats, nmat, theta);
			double *rpkm = (double *)calloc((size_t)(K ? K : 1), sizeof(double));
			double total_read_mbases = 0.0;
			for (int m = 0; m < P->M; m++) {
				total_read_mbases += P->total_read_bases[m] / 1.0E6;
				for (int i = 0; i < K; i++) rpkm[i] += (double)(support_bases[m]) * theta[i];
			}
			for (int i = 0; i < K; i++) {
				rpkm[i] /= ((double)(g->iso_total_len[i]) / 1.0E3);
				rpkm[i] /= total_read_mbases;
			}
			double logll = reads_log_likelihood(K, mats, nmat, theta);
			double sum_supports = 0.0;
			for (int m = 0; m < P->M; m++) sum_supports += supports[m];
			for (int i = 0; i < K; i++) {
				sink_str(out, g->gname); sink_str(out, "\t");
				for (int m = 0; m < P->M; m++) { sink_dbl(out, supports[m]); sink_str(out, "\t"); }
				sink_str(out, g->isos[i]->name); sink_str(out, "\t"); sink_dbl(out, theta[i]);
				sink_str(out, "\t"); sink_dbl(out, rpkm[i]);
				if (sum_supports > 1E-5) { sink_str(out, "\t"); sink_dbl(out, logll / sum_supports); sink_str(out, "\n"); }
				else sink_str(out, "\t0\n");
			}
			if (ex) { for (int k = 0; k < K; k++) ex->theta[gi][k] = theta[k]; ex->logll[gi] = logll; ex->iters[gi] = iters; }
			if (ex && getenv("LSQO_FIM") && K >= 1 && K <= 6) {
				const int D = K - 1;
				ex->fim[gi] = (double *)calloc((size_t)P->M * (size_t)(D * D) + 1, sizeof(double));
				ex->fim_var[gi] = (double *)calloc((size_t)P->M * 2 + 1, sizeof(double));
				for (int m = 0; m < P->M; m++) {
					const int short_read = strcmp(P->read_types[m], "MEDIUM_READ") != 0;
					double G[8];
					for (int k = 0; k < K; k++) {
						const unsigned long ars = lsqo_ars_total(g->exon_len, g->iso_idx[k], g->iso_n[k], P->exp_len[m], short_read);
						G[k] = ars == 0 ? 0.0 : 1.0 / (double)ars;
					}
					fim_bruteforce(g, theta, G, P->exp_len[m], short_read, ex->fim[gi] + (size_t)m * (size_t)(D * D));
					ex->fim_var[gi][2 * m] = fim_var_by_diag(ex->fim[gi] + (size_t)m * (size_t)(D * D), D);
					ex->fim_var[gi][2 * m + 1] = fim_var_by_inv(ex->fim[gi] + (size_t)m * (size_t)(D * D), D);
				}
			}
			free(theta); free(rpkm);
		}
		for (int m = 0; m < nmat; m++) free(mats[m].g);
		free(mats); free(supports); free(support_bases); free(iso_count);
	}
	/* the process-lifetime tables are left to exit() in CLI mode; library callers leak a few
	 * KB per call, which test sizes tolerate */
	for (int m = 0; m < P->M; m++) { for (size_t i = 0; i < rv[m].n; i++) il_free(&rv[m].v[i].il); free(rv[m].v); }
	free(rv);
	free(sel);
	return 0;
}

/* argv layout of the reference (count.cpp:95-129, solve.cpp:109-146); argv[0] ignored */
static int parse_args(int is_solve, int argc, char **argv, params_t *P) {
	memset(P, 0, sizeof *P);
	P->is_solve = is_solve;
	int per = is_solve ? 5 : 4;
	if (argc < (is_solve ? 15 : 14)) { err_line("Usage: (reference syntax)"); return 1; }
	int argi = 1;
	long lvl;
	if (!lexical_cast_long(argv[argi], strlen(argv[argi]), &lvl)) return 134; /* uncaught exception */
	argi++;
	argi += 2; /* proj_name out_prefix */
	P->isoform_format = argv[argi++]; P->isoforms_path = argv[argi++];
	P->g2i_format = argv[argi++]; P->g2i_path = argv[argi++];
	if (!lexical_cast_ulong(argv[argi++], &P->gene_begin)) { err_line("Lexical_cast error"); return 1; }
	if (!lexical_cast_ulong(argv[argi++], &P->gene_end)) { err_line("Lexical_cast error"); return 1; }
	int maxm = (argc - argi) / per + 1;
	P->read_formats = (const char **)xmalloc(sizeof(char *) * (size_t)maxm);
	P->read_types = (const char **)xmalloc(sizeof(char *) * (size_t)maxm);
	P->reads_paths = (const char **)xmalloc(sizeof(char *) * (size_t)maxm);
	P->exp_len = (unsigned long *)xmalloc(sizeof(unsigned long) * (size_t)maxm);
	P->total_read_bases = (double *)xmalloc(sizeof(double) * (size_t)maxm);
	while (argi < argc) {
		if (argc - argi < per) { err_line("Usage: (reference syntax)"); return 1; }
		P->read_formats[P->M] = argv[argi++];
		P->read_types[P->M] = argv[argi++];
		if (!lexical_cast_ulong(argv[argi++], &P->exp_len[P->M])) { err_line("Lexical_cast error"); return 1; }
		P->reads_paths[P->M] = argv[argi++];
		if (is_solve) { if (!lexical_cast_double(argv[argi++], &P->total_read_bases[P->M])) { err_line("Lexical_cast error"); return 1; } }
		P->M++;
	}
	return 0;
}

/* ------------------------------------------------------------------ shared-library entry points */

/* Runs `count` (is_solve=0) or `solve` (is_solve=1) with the reference's argv (argv[0] is a
 * placeholder).  *out_text receives a malloc'd NUL-terminated copy of stdout. Returns the
 * exit status. */
int lsqo_run(int is_solve, int argc, char **argv, char **out_text, exact_t **out_exact) {
	params_t P;
	int rc = parse_args(is_solve, argc, argv, &P);
	sink o; memset(&o, 0, sizeof o);
	exact_t *ex = NULL;
	if (out_exact) { ex = (exact_t *)calloc(1, sizeof(exact_t)); *out_exact = ex; }
	if (rc == 0) rc = run(&P, &o, ex);
	if (out_text) { if (!o.buf) { o.buf = (char *)xmalloc(1); o.buf[0] = 0; } *out_text = o.buf; } else free(o.buf);
	return rc;
}
void lsqo_free(void *p) { free(p); }

/* accessors for ctypes (exact side output) */
int lsqo_exact_n_genes(const exact_t *e) { return e->n_genes; }
const char *lsqo_exact_gname(const exact_t *e, int g) { return e->gname[g]; }
int lsqo_exact_K(const exact_t *e, int g) { return e->K[g]; }
unsigned long lsqo_exact_support(const exact_t *e, int g, int m) { return e->supports[g][m]; }
unsigned long lsqo_exact_bases(const exact_t *e, int g, int m) { return e->bases[g][m]; }
unsigned long lsqo_exact_iso_count(const exact_t *e, int g, int k) { return e->iso_count[g][k]; }
double lsqo_exact_theta(const exact_t *e, int g, int k) { return e->theta[g][k]; }
double lsqo_exact_logll(const exact_t *e, int g) { return e->logll[g]; }
unsigned long lsqo_exact_iters(const exact_t *e, int g) { return e->iters[g]; }
unsigned long lsqo_exact_n_loaded(const exact_t *e, int m) { return e->n_loaded[m]; }
int lsqo_exact_has_fim(const exact_t *e, int g) { return e->fim && e->fim[g] != NULL; }
double lsqo_exact_fim(const exact_t *e, int g, int m, int p, int q) { const int D = e->K[g] - 1; return e->fim[g][(size_t)m * (size_t)(D * D) + (size_t)(p * D + q)]; }
double lsqo_exact_fim_var(const exact_t *e, int g, int m, int which) { return e->fim_var[g][2 * m + which]; }

/* fine-grained helpers for unit parity tests */
int lsqo_segments(const long *starts, const long *ends, int n, long *out_s, long *out_e, int cap) {
	exonset es; memset(&es, 0, sizeof es);
	for (int i = 0; i < n; i++) lsqo_exonset_insert(&es, starts[i], ends[i]);
	int m = es.n < cap ? es.n : cap;
	for (int i = 0; i < m; i++) { out_s[i] = es.v[i].start; out_e[i] = es.v[i].end; }
	int r = es.n; free(es.v); return r;
}
int lsqo_merge_intervals(const long *starts, const long *ends, int n, long *out_s, long *out_e, int cap) {
	ilist il; memset(&il, 0, sizeof il);
	for (int i = 0; i < n; i++) lsqo_il_add(&il, starts[i], ends[i]);
	int m = il.n < cap ? il.n : cap;
	for (int i = 0; i < m; i++) { out_s[i] = il.s[i]; out_e[i] = il.e[i]; }
	int r = il.n; il_free(&il); return r;
}
/* build one read (already merged block list) against segments: returns segment bitmask */
unsigned long long lsqo_build_mask(const long *bs, const long *be, int nb, const long *ss, const long *se, int ns, unsigned long *matched) {
	ilist il; il.s = (long *)bs; il.e = (long *)be; il.n = nb; il.cap = nb;
	seg_t segs[64]; if (ns > 64) ns = 64;
	for (int i = 0; i < ns; i++) { segs[i].start = ss[i]; segs[i].end = se[i]; }
	int idx[64], nidx; unsigned long rl;
	lsqo_read_build(&il, segs, ns, idx, &nidx, &rl, 64);
	unsigned long long mask = 0;
	for (int i = 0; i < nidx; i++) mask |= 1ull << idx[i];
	*matched = rl;
	return mask;
}
/* EM on explicit per-read rows: g is n x K row-major (one method) */
unsigned long lsqo_em_rows(int K, const double *g, unsigned long n, double *theta, double *logll) {
	dmat m; m.g = (double *)g; m.n = n;
	unsigned long it = lsqo_em(K, &m, 1, theta);
	*logll = reads_log_likelihood(K, &m, 1, theta);
	return it;
}

#ifdef LSQ_ORACLE_MAIN
/* oracle_count / oracle_solve: same argv as the reference executables; which one is chosen
 * by the program name (ends in "solve") or a leading "--solve"/"--count". */
int main(int argc, char **argv) {
	int is_solve = 0;
	const char *base = strrchr(argv[0], '/'); base = base ? base + 1 : argv[0];
	if (strstr(base, "solve")) is_solve = 1;
	if (argc > 1 && strcmp(argv[1], "--solve") == 0) { is_solve = 1; argv++; argc--; }
	else if (argc > 1 && strcmp(argv[1], "--count") == 0) { is_solve = 0; argv++; argc--; }
	params_t P;
	int rc = parse_args(is_solve, argc, argv, &P);
	if (rc) return rc;
	sink o; memset(&o, 0, sizeof o); o.fp = stdout;
	rc = run(&P, &o, NULL);
	fflush(stdout);
	return rc;
}
#endif
