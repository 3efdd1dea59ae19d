"""Thin object wrappers over the C ABI (include/lesseq_hip.h).  No compute lives here."""
import os
import ctypes as C

import numpy as np

from ._lib import lib, check, check_runtime_once, SynthSpecStruct, u64, u32, i32, u16, u8, i64, vp, cs, P

EVENT_TYPES = ("SE", "RI", "A5SS", "A3SS", "MXE", "AFE", "ALE", "T3")


def _b(s):
    return s.encode() if isinstance(s, str) else s


def _take_text(ptr):
    if not ptr:
        return ""
    s = C.string_at(ptr).decode()
    lib.lsq_free(ptr)
    return s


def _ptr(a, ct):
    return a.ctypes.data_as(P(ct))


class Annotation:
    """lsq_annotation_load: LH_GENE_TXT + UCSC_GENE2ISOFORM, genes [begin, end) of the sorted name set"""

    def __init__(self, isoforms_path, g2i_path, begin=0, end=2 ** 62, isoform_format="LH_GENE_TXT", g2i_format="UCSC_GENE2ISOFORM"):
        h = vp()
        check(lib.lsq_annotation_load(_b(isoform_format), _b(isoforms_path), _b(g2i_format), _b(g2i_path), begin, end, C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            lib.lsq_annotation_free(self.h)
            self.h = None

    @property
    def num_genes(self):
        return lib.lsq_annotation_num_genes(self.h)


class Events:
    """lsq_events_compile: segments, isoform masks, ARS per read file, covered regions, device plan"""

    def __init__(self, annotation, read_types=("SHORT_READ",), read_lengths=(100,)):
        M = len(read_types)
        rt = (cs * max(M, 1))(*[_b(t) for t in read_types])
        rl = (u64 * max(M, 1))(*read_lengths)
        h = vp()
        check(lib.lsq_events_compile(annotation.h, M, rt, rl, C.byref(h)))
        self.h = h
        self.n_methods = M

    def __del__(self):
        if getattr(self, "h", None):
            lib.lsq_events_free(self.h)
            self.h = None

    def __len__(self):
        return lib.lsq_events_count(self.h)

    @property
    def total_isoforms(self):
        return lib.lsq_events_total_isoforms(self.h)

    @property
    def num_buckets(self):
        return lib.lsq_events_num_buckets(self.h)

    @property
    def lds_table_bytes(self):
        return lib.lsq_events_lds_table_bytes(self.h)

    def gene_name(self, ev):
        return lib.lsq_events_gene_name(self.h, ev).decode()

    def chrom(self, ev):
        return lib.lsq_events_chrom(self.h, ev).decode()

    def strand(self, ev):
        return lib.lsq_events_strand(self.h, ev).decode()

    def K(self, ev):
        return lib.lsq_events_num_isoforms(self.h, ev)

    def N(self, ev):
        return lib.lsq_events_num_segments(self.h, ev)

    def isoform_name(self, ev, j):
        return lib.lsq_events_isoform_name(self.h, ev, j).decode()

    def segments(self, ev):
        out = []
        s, e = i64(), i64()
        for n in range(self.N(ev)):
            check(lib.lsq_events_segment(self.h, ev, n, C.byref(s), C.byref(e)))
            out.append((s.value, e.value))
        return out

    def isoform_mask(self, ev, j):
        return lib.lsq_events_isoform_mask(self.h, ev, j)

    def isoform_length(self, ev, j):
        return lib.lsq_events_isoform_length(self.h, ev, j)

    def ars(self, method, ev, j):
        return lib.lsq_events_ars(self.h, method, ev, j)

    def span(self, ev):
        s, e = i64(), i64()
        check(lib.lsq_events_span(self.h, ev, C.byref(s), C.byref(e)))
        return s.value, e.value

    def set_shard(self, first_event, n_events):
        check(lib.lsq_events_set_shard(self.h, first_event, n_events))

    def shard_bounds(self, world, weights=None):
        """lsq_shard_bounds: [(first, count)] per process, contiguous slices of equal weight"""
        first, count = (u64 * world)(), (u64 * world)()
        w = None
        if weights is not None:
            w = np.ascontiguousarray(weights, np.float64)
            assert len(w) == len(self)
        check(lib.lsq_shard_bounds(self.h, world, _ptr(w, C.c_double) if w is not None else None, first, count))
        return [(int(first[r]), int(count[r])) for r in range(world)]

    def record_words(self, first, count):
        return int(lib.lsq_record_words(self.h, first, count))

    def gathered_unpack(self, bounds, blocks, stride_words):
        """blocks: uint64 array [world * stride_words] of packed records -> (cnt, bases, theta, logll) of the whole job"""
        world = len(bounds)
        first = (u64 * world)(*[b[0] for b in bounds])
        count = (u64 * world)(*[b[1] for b in bounds])
        off = self.class_offsets()
        n_cls, n_ev, M = off[-1], len(self), self.n_methods
        cnt = np.zeros((max(M, 1), max(n_cls, 1)), np.uint64)
        bases = np.zeros((max(M, 1), max(n_cls, 1)), np.uint64)
        theta = np.zeros(max(self.total_isoforms, 1), np.float64)
        ll = np.zeros(max(n_ev, 1), np.float64)
        blocks = np.ascontiguousarray(blocks).view(np.uint64)
        # the C side indexes [method][n_classes] rows of exactly n_cls entries
        c2 = np.zeros(max(M * n_cls, 1), np.uint64)
        b2 = np.zeros(max(M * n_cls, 1), np.uint64)
        check(lib.lsq_gathered_unpack(self.h, world, first, count, _ptr(blocks, u64), stride_words, _ptr(c2, u64), _ptr(b2, u64),
                                      _ptr(theta, C.c_double), _ptr(ll, C.c_double)))
        cnt = c2[:M * n_cls].reshape(M, n_cls) if M * n_cls else cnt[:M, :n_cls]
        bases = b2[:M * n_cls].reshape(M, n_cls) if M * n_cls else bases[:M, :n_cls]
        return cnt, bases, theta[:self.total_isoforms], ll[:n_ev]

    def chrom_id(self, name):
        return lib.lsq_events_chrom_id(self.h, _b(name))

    def strand_id(self, name):
        return lib.lsq_events_strand_id(self.h, _b(name))

    def class_offsets(self):
        off = [0]
        for ev in range(len(self)):
            off.append(off[-1] + (1 << self.K(ev)) - 1)
        return off


class Reads:
    """A parsed read set in file order: from an MRF file, caller arrays, or the synthetic generator"""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def from_mrf(cls, path, events, n_threads=0, read_format="MRF_SINGLE"):
        h = vp()
        check(lib.lsq_reads_parse(_b(read_format), _b(path), events.h, n_threads, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, blk_off, line_no, blk_start, blk_end, blk_chrom, blk_strand):
        arrs = (np.ascontiguousarray(blk_off, np.uint64), np.ascontiguousarray(line_no, np.uint32),
                np.ascontiguousarray(blk_start, np.int32), np.ascontiguousarray(blk_end, np.int32),
                np.ascontiguousarray(blk_chrom, np.uint16), np.ascontiguousarray(blk_strand, np.uint8))
        h = vp()
        check(lib.lsq_reads_wrap(len(arrs[1]), _ptr(arrs[0], u64), _ptr(arrs[1], u32), _ptr(arrs[2], i32),
                                 _ptr(arrs[3], i32), _ptr(arrs[4], u16), _ptr(arrs[5], u8), C.byref(h)))
        return cls(h, keep=arrs)

    @classmethod
    def synthetic(cls, spec, events, n_threads=0):
        h = vp()
        check(lib.lsq_synth_reads(C.byref(spec.c), events.h, n_threads, C.byref(h)))
        return cls(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.lsq_reads_free(self.h)
            self.h = None

    def __len__(self):
        return lib.lsq_reads_count(self.h)

    @property
    def num_blocks(self):
        return lib.lsq_reads_num_blocks(self.h)

    def arrays(self):
        """copies of (blk_off, line_no, blk_start, blk_end, blk_chrom_id, blk_strand_id)"""
        ptrs = [vp() for _ in range(6)]
        check(lib.lsq_reads_arrays(self.h, *[C.byref(p) for p in ptrs]))
        n, nb = len(self), self.num_blocks
        spec = ((np.uint64, n + 1), (np.uint32, n), (np.int32, nb), (np.int32, nb), (np.uint16, nb), (np.uint8, nb))
        out = []
        for p, (dt, cnt) in zip(ptrs, spec):
            if cnt == 0 or not p.value:
                out.append(np.zeros(cnt, dt))
            else:
                buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(p.value)
                out.append(np.frombuffer(buf, dtype=dt, count=cnt).copy())
        return tuple(out)


class Context:
    """One GPU: uploads, the count kernel, the EM kernel, result fetch"""

    def __init__(self, device=0):
        h = vp()
        check(lib.lsq_ctx_create(device, C.byref(h)))
        self.h = h
        self.events = None
        check_runtime_once()              # (the HIP runtime is up now: compiled-against and running versions are compared once)
        # LSQ_OPTIONS="name=value,...": the executables pass these to lsq_ctx_set_option (lsq_cli.cpp), so does this class
        for item in filter(None, os.environ.get("LSQ_OPTIONS", "").split(",")):
            if "=" in item:
                name, value = item.split("=", 1)
                self.set_option(name.strip(), float(value))

    def close(self):
        if getattr(self, "h", None):
            lib.lsq_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload_events(self, events):
        check(lib.lsq_events_upload(self.h, events.h))
        self.events = events

    def upload_reads(self, method, reads):
        check(lib.lsq_reads_upload(self.h, method, reads.h))

    def upload_reads_mrf(self, method, path, read_format="MRF_SINGLE"):
        """MRF text -> HBM -> parsed and ingested on the device (lsq_reads_upload_mrf)"""
        check(lib.lsq_reads_upload_mrf(self.h, method, _b(read_format), _b(path)))

    def parse_mrf_device(self, path, read_format="MRF_SINGLE"):
        """the device parser's blocks, copied back as a Reads (tests, tools)"""
        h = vp()
        check(lib.lsq_mrf_parse_device(self.h, _b(read_format), _b(path), C.byref(h)))
        return Reads(h)

    def mrf_timing(self):
        a, b = C.c_float(), C.c_float()
        check(lib.lsq_last_mrf_timing(self.h, C.byref(a), C.byref(b)))
        return {"h2d_ms": a.value, "parse_ms": b.value}

    def parse_paths(self):
        """(tiles handed to the byte-walking kernel, lines handed to the shared splitter, whole file through the byte-walking kernel)
        of the latest device parse (lsq_debug_last_parse_paths)"""
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        lib.lsq_debug_last_parse_paths.argtypes = [vp, P(C.c_uint), P(C.c_uint), P(C.c_uint)]
        lib.lsq_debug_last_parse_paths.restype = C.c_int
        check(lib.lsq_debug_last_parse_paths(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"tiles_to_byte_walking_kernel": a.value, "lines_to_shared_splitter": b.value, "whole_file_byte_walking": bool(c.value)}

    def ingest_stages(self):
        """device milliseconds and minimum bytes of every pass of the latest ingest (lsq_last_ingest_stages), in order"""
        n = lib.lsq_ingest_stage_count()
        ms, by = (C.c_float * n)(), (C.c_uint64 * n)()
        check(lib.lsq_last_ingest_stages(self.h, ms, by, n))
        return [{"stage": lib.lsq_ingest_stage_name(i).decode(), "ms": float(ms[i]), "bytes": int(by[i])} for i in range(n)]

    def retained(self, method):
        return lib.lsq_reads_retained(self.h, method)

    def pooled(self, method):
        """retained reads kept in the pools: those that start in the span of an event planned on this context"""
        return lib.lsq_reads_pooled(self.h, method)

    def pooled_blocks(self, method):
        """blocks of the pooled reads: what one count() streams"""
        return lib.lsq_reads_pooled_blocks(self.h, method)

    def pool_format(self, method):
        """lsq_reads_pool_format: (compact records?, bytes of block coordinates in HBM,
        reads kept as one-block / two-block records / with the many-block reads)"""
        a, n = C.c_int(0), C.c_uint64(0)
        k = (C.c_uint64 * 3)()
        check(lib.lsq_reads_pool_format(self.h, method, C.byref(a), C.byref(n), k))
        return bool(a.value), n.value, tuple(int(x) for x in k)

    def retained_blocks(self, method):
        return lib.lsq_reads_retained_blocks(self.h, method)

    def count(self):
        check(lib.lsq_count(self.h))

    def solve(self):
        check(lib.lsq_solve(self.h))

    def set_option(self, name, value):
        """lsq_ctx_set_option: grid_multiplier, exception_capacity, recount_every_read, em_guard_band,
        snap_shares, em_regroup, em_flat_min_events, compact_pools"""
        check(lib.lsq_ctx_set_option(self.h, _b(name), float(value)))

    def count_status(self):
        """per read file: (pairs handed to the exception pass, recount ran) of the latest count()"""
        M = max(self.events.n_methods, 1)
        e, r = (u32 * M)(), (u32 * M)()
        check(lib.lsq_count_status(self.h, e, r))
        return list(e)[:self.events.n_methods], list(r)[:self.events.n_methods]

    def counts_device_words(self):
        return int(lib.lsq_counts_device_words(self.h))

    def export_counts_device(self, d_ptr):
        """the latest count's class counts and matched bases, device order, into a device buffer (result stream)"""
        check(lib.lsq_counts_export_device(self.h, vp(d_ptr)))

    def import_counts_device(self, d_ptr):
        """such a buffer (e.g. summed over the ranks of a read-sharded job) becomes the counts solve() works on"""
        check(lib.lsq_counts_import_device(self.h, vp(d_ptr)))

    def launch_info(self):
        """(one-block reads a lane settles per table look, resident workgroups per compute unit) of the latest count()"""
        a, b = u32(), u32()
        check(lib.lsq_count_launch_info(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_em_guard_band(self, band):
        """events whose EM stop test comes within `band` of its threshold are replayed in the reference's per-read order"""
        check(lib.lsq_set_em_guard_band(self.h, float(band)))

    def solve_finalize(self):
        """the exact-order replay of flagged events on its own; returns how many were redone"""
        n = u32()
        check(lib.lsq_solve_finalize(self.h, C.byref(n)))
        return n.value

    def synchronize(self):
        check(lib.lsq_ctx_synchronize(self.h))

    @property
    def stream(self):
        return lib.lsq_ctx_stream(self.h)

    def set_timing(self, on=True):
        """HIP events around the kernels of the following count()/solve() calls (off by default: they cost queue time)"""
        check(lib.lsq_set_timing(self.h, 1 if on else 0))

    def timing(self):
        a, b = C.c_float(), C.c_float()
        check(lib.lsq_last_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def copy_results_device(self, d_class_count=None, d_theta=None, d_logll=None):
        """raw device-order results into caller device buffers (integer addresses), async"""
        check(lib.lsq_results_copy_device(self.h, vp(d_class_count), vp(d_theta), vp(d_logll)))

    def pack_results_device(self, d_block):
        """the shard's per-event records in output order into a caller device buffer (integer address), async on the result stream"""
        check(lib.lsq_results_pack_device(self.h, vp(d_block)))

    @property
    def result_stream(self):
        return lib.lsq_ctx_result_stream(self.h)

    def device_order(self):
        o = np.zeros(max(len(self.events), 1), np.int32)
        check(lib.lsq_results_device_order(self.h, _ptr(o, i32)))
        return o[:len(self.events)]

    def fast_kernel_ms(self):
        a = C.c_float()
        check(lib.lsq_last_fast_kernel_ms(self.h, C.byref(a)))
        return a.value

    def counts(self):
        """(class_count, class_bases) as uint64 arrays of shape [n_methods, n_classes], output order"""
        n = lib.lsq_results_num_classes(self.h)
        M = self.events.n_methods
        cnt = np.zeros((max(M, 1), max(n, 1)), np.uint64)
        bases = np.zeros((max(M, 1), max(n, 1)), np.uint64)
        check(lib.lsq_results_counts(self.h, _ptr(cnt, u64), _ptr(bases, u64)))
        return cnt[:M, :n], bases[:M, :n]

    def set_counts(self, cnt, bases):
        """the inverse of counts(): arrays of shape [n_methods, n_classes] become the context's counts"""
        cnt = np.ascontiguousarray(cnt, np.uint64)
        bases = np.ascontiguousarray(bases, np.uint64)
        check(lib.lsq_results_set_counts(self.h, _ptr(cnt, u64), _ptr(bases, u64)))

    def stage_text(self, path, byte_begin=0, byte_end=2 ** 64 - 1):
        """bytes [byte_begin, byte_end) of an MRF file copied to HBM; returns an opaque handle"""
        h = vp()
        check(lib.lsq_text_stage_range(self.h, _b(path), byte_begin, byte_end, C.byref(h)))
        return h

    def text_lines(self, text):
        n = u64()
        check(lib.lsq_text_lines(self.h, text, C.byref(n)))
        return n.value

    def upload_reads_text(self, method, text, has_header=True, first_line=1, read_format="MRF_SINGLE", free=True):
        try:
            check(lib.lsq_reads_upload_text_at(self.h, method, _b(read_format), text, 1 if has_header else 0, first_line))
        finally:
            if free:
                lib.lsq_text_free(text)

    def solution(self):
        """(theta[n_isoforms], logll[n_events], iters, flags), output order"""
        ne, ni = len(self.events), self.events.total_isoforms
        theta = np.zeros(max(ni, 1), np.float64)
        ll = np.zeros(max(ne, 1), np.float64)
        it = np.zeros(max(ne, 1), np.uint32)
        fl = np.zeros(max(ne, 1), np.uint8)
        check(lib.lsq_results_solve(self.h, _ptr(theta, C.c_double), _ptr(ll, C.c_double), _ptr(it, u32), _ptr(fl, u8)))
        return theta[:ni], ll[:ne], it[:ne], fl[:ne]

    def fim(self):
        """after solve(): (offsets[n_events+1], fim[n_methods, size], var_by_diag[n_methods, n_events], var_by_inverse[...]) --
        the expected Fisher information per read of theta_1..theta_{K-1} at the solved theta and fim.h's two variance
        estimates (parity unpinned: dead code in the reference)"""
        check(lib.lsq_fim(self.h))
        ne, M = len(self.events), self.events.n_methods
        off = np.zeros(ne + 1, np.uint64)
        check(lib.lsq_results_fim_offsets(self.h, _ptr(off, C.c_uint64)))
        size = int(off[ne])
        f = np.zeros(max(M * size, 1), np.float64)
        vd = np.zeros(max(M * ne, 1), np.float64)
        vi = np.zeros(max(M * ne, 1), np.float64)
        check(lib.lsq_results_fim(self.h, _ptr(f, C.c_double), _ptr(vd, C.c_double), _ptr(vi, C.c_double)))
        return off, f[:M * size].reshape(M, size), vd[:M * ne].reshape(M, ne), vi[:M * ne].reshape(M, ne)


class SynthSpec:
    def __init__(self, seed, n_events, n_reads, read_length=100, n_chrom=1, event_types=EVENT_TYPES, zipf=False, overlap_frac=0.10, first_read=0, sorted_reads=False):
        mask = 0
        for t in event_types:
            mask |= 1 << EVENT_TYPES.index(t)
        self.c = SynthSpecStruct(seed, n_events, n_reads, read_length, n_chrom, mask, 1 if zipf else 0, overlap_frac, first_read, 1 if sorted_reads else 0, 0)


def synth_write(spec, directory, stem, write_mrf=True):
    check(lib.lsq_synth_write(C.byref(spec.c), _b(directory), _b(stem), 1 if write_mrf else 0))


def format_count(events, cnt):
    out = vp()
    a = np.ascontiguousarray(cnt, np.uint64)
    check(lib.lsq_format_count(events.h, events.n_methods, _ptr(a, u64), C.byref(out)))
    return _take_text(out)


def format_solve(events, cnt, bases, theta, logll, total_read_bases):
    out = vp()
    a = np.ascontiguousarray(cnt, np.uint64)
    b = np.ascontiguousarray(bases, np.uint64)
    t = np.ascontiguousarray(theta, np.float64)
    l = np.ascontiguousarray(logll, np.float64)
    r = np.ascontiguousarray(total_read_bases, np.float64)
    check(lib.lsq_format_solve(events.h, events.n_methods, _ptr(a, u64), _ptr(b, u64), _ptr(t, C.c_double),
                               _ptr(l, C.c_double), _ptr(r, C.c_double), C.byref(out)))
    return _take_text(out)


def cli_run(tool, argv):
    """Runs count / solve / classify in-process with the reference's argv (without argv[0]).
    Returns (exit_status, stdout_text).  The log goes to this process's stderr."""
    full = [b"lsq"] + [_b(a) for a in argv]
    arr = (cs * len(full))(*full)
    out = vp()
    rc = lib.lsq_cli_run(_b(tool), len(full), arr, C.byref(out))
    return rc, _take_text(out)
