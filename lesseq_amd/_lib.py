"""ctypes loader for liblesseq_hip.so (built in-tree by lesseq_amd/csrc/Makefile)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSQ_LIB") or os.path.join(_HERE, "_build", "liblesseq_hip.so")   # LSQ_LIB: developer override for build variants


class LsqError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("lsq status %d: %s" % (status, msg))
        self.status = status


if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with `make -C lesseq_amd/csrc` (or __graft_entry__.build()); "
                      "there is no fallback implementation" % LIB_PATH)

# One HIP runtime per process.  PyTorch-ROCm brings libamdhip64 of its own; if this library came first it would bind the
# system's copy, torch would then load its own beside it, and whichever of the two opens the device second finds none
# (measured: torch.cuda.is_available() False after a Context, or LSQ_E_DEVICE after torch.cuda's start-up).  With torch
# loaded first the library's libamdhip64 dependency resolves to the copy already in the process.  bench.py, the multi-GPU
# drivers (dist.py under torchrun) and the distributed tests all hold both; a host without torch skips this.
# (LSQ_NO_TORCH=1: leave torch out, for a process that will never use it.)
if os.environ.get("LSQ_NO_TORCH") != "1":
    try:
        import torch  # noqa: F401
    except Exception:      # not installed, or an install that fails to load: the library then binds the system's runtime
        pass

lib = C.CDLL(LIB_PATH)

u64, i64, u32, i32, u16, u8 = C.c_uint64, C.c_int64, C.c_uint32, C.c_int32, C.c_uint16, C.c_uint8
P = C.POINTER
vp = C.c_void_p
cs = C.c_char_p


class SynthSpecStruct(C.Structure):
    _fields_ = [("seed", u64), ("n_events", u64), ("n_reads", u64), ("read_length", u32), ("n_chrom", u32),
                ("event_types", u32), ("zipf", u32), ("overlap_frac", C.c_double), ("first_read", u64),
                ("sorted", u32), ("reserved", u32)]


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("lsq_last_error", cs)
_sig("lsq_abi_version", C.c_int)
_sig("lsq_free", None, vp)
_sig("lsq_annotation_load", C.c_int, cs, cs, cs, cs, u64, u64, P(vp))
_sig("lsq_annotation_free", None, vp)
_sig("lsq_annotation_num_genes", i64, vp)
_sig("lsq_annotation_num_isoforms_loaded", i64, vp)
_sig("lsq_annotation_num_genes_loaded", i64, vp)
_sig("lsq_events_compile", C.c_int, vp, C.c_int, P(cs), P(u64), P(vp))
_sig("lsq_events_free", None, vp)
_sig("lsq_events_count", i64, vp)
_sig("lsq_events_total_isoforms", i64, vp)
_sig("lsq_events_gene_name", cs, vp, i64)
_sig("lsq_events_chrom", cs, vp, i64)
_sig("lsq_events_strand", cs, vp, i64)
_sig("lsq_events_num_isoforms", C.c_int, vp, i64)
_sig("lsq_events_num_segments", C.c_int, vp, i64)
_sig("lsq_events_isoform_name", cs, vp, i64, C.c_int)
_sig("lsq_events_segment", C.c_int, vp, i64, C.c_int, P(i64), P(i64))
_sig("lsq_events_isoform_mask", u64, vp, i64, C.c_int)
_sig("lsq_events_isoform_length", u64, vp, i64, C.c_int)
_sig("lsq_events_ars", u64, vp, C.c_int, i64, C.c_int)
_sig("lsq_events_span", C.c_int, vp, i64, P(i64), P(i64))
_sig("lsq_events_num_buckets", i64, vp)
_sig("lsq_events_lds_table_bytes", i64, vp)
_sig("lsq_events_host_genes", i64, vp)
_sig("lsq_events_set_shard", C.c_int, vp, u64, u64)
_sig("lsq_mrf_parse", C.c_int, cs, cs, vp, C.c_int, P(vp))
_sig("lsq_reads_parse", C.c_int, cs, cs, vp, C.c_int, P(vp))
_sig("lsq_reads_wrap", C.c_int, u64, P(u64), P(u32), P(i32), P(i32), P(u16), P(u8), P(vp))
_sig("lsq_reads_free", None, vp)
_sig("lsq_reads_count", u64, vp)
_sig("lsq_reads_num_blocks", u64, vp)
_sig("lsq_events_chrom_id", C.c_int, vp, cs)
_sig("lsq_events_strand_id", C.c_int, vp, cs)
_sig("lsq_ctx_create", C.c_int, C.c_int, P(vp))
_sig("lsq_ctx_destroy", None, vp)
_sig("lsq_ctx_stream", vp, vp)
_sig("lsq_ctx_synchronize", C.c_int, vp)
_sig("lsq_ctx_synchronize_for", C.c_int, vp, C.c_double)
_sig("lsq_debug_throw", C.c_int, C.c_int)
_sig("lsq_debug_hip_versions", C.c_int, P(C.c_int), P(C.c_int))
_sig("lsq_events_upload", C.c_int, vp, vp)
_sig("lsq_reads_upload", C.c_int, vp, C.c_int, vp)
_sig("lsq_reads_arrays", C.c_int, vp, P(vp), P(vp), P(vp), P(vp), P(vp), P(vp))
_sig("lsq_events_strand_name", cs, vp, C.c_int)
_sig("lsq_reads_upload_mrf", C.c_int, vp, C.c_int, cs, cs)
_sig("lsq_text_stage", C.c_int, vp, cs, P(vp))
_sig("lsq_text_stage_range", C.c_int, vp, cs, u64, u64, P(vp))
_sig("lsq_text_lines", C.c_int, vp, vp, P(u64))
_sig("lsq_reads_upload_text_at", C.c_int, vp, C.c_int, cs, vp, C.c_int, u64)
_sig("lsq_results_set_counts", C.c_int, vp, P(u64), P(u64))
_sig("lsq_text_free", None, vp)
_sig("lsq_reads_upload_text", C.c_int, vp, C.c_int, cs, vp)
_sig("lsq_mrf_parse_device", C.c_int, vp, cs, cs, P(vp))
_sig("lsq_last_mrf_timing", C.c_int, vp, P(C.c_float), P(C.c_float))
_sig("lsq_set_log_level", None, C.c_int)
_sig("lsq_ingest_stage_count", C.c_int)
_sig("lsq_ingest_stage_name", cs, C.c_int)
_sig("lsq_last_ingest_stages", C.c_int, vp, P(C.c_float), P(u64), C.c_int)
_sig("lsq_reads_retained", u64, vp, C.c_int)
_sig("lsq_reads_retained_blocks", u64, vp, C.c_int)
_sig("lsq_reads_pooled", u64, vp, C.c_int)
_sig("lsq_reads_pooled_blocks", u64, vp, C.c_int)
_sig("lsq_reads_pool_format", C.c_int, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("lsq_count", C.c_int, vp)
_sig("lsq_solve", C.c_int, vp)
_sig("lsq_results_num_classes", i64, vp)
_sig("lsq_results_class_offsets", C.c_int, vp, P(u64))
_sig("lsq_results_counts", C.c_int, vp, P(u64), P(u64))
_sig("lsq_results_solve", C.c_int, vp, P(C.c_double), P(C.c_double), P(u32), P(u8))
_sig("lsq_count_status", C.c_int, vp, P(u32), P(u32))
_sig("lsq_count_launch_info", C.c_int, vp, P(u32), P(u32))
_sig("lsq_host_evaluated", C.c_int, vp, P(u64), P(u64))
_sig("lsq_counts_device_words", u64, vp)
_sig("lsq_counts_export_device", C.c_int, vp, vp)
_sig("lsq_counts_import_device", C.c_int, vp, vp)
_sig("lsq_ctx_set_option", C.c_int, vp, cs, C.c_double)
_sig("lsq_solve_finalize", C.c_int, vp, P(u32))
_sig("lsq_set_em_guard_band", C.c_int, vp, C.c_double)
_sig("lsq_results_copy_device", C.c_int, vp, vp, vp, vp)
_sig("lsq_ctx_result_stream", vp, vp)
_sig("lsq_shard_bounds", C.c_int, vp, C.c_int, P(C.c_double), P(u64), P(u64))
_sig("lsq_record_words", u64, vp, u64, u64)
_sig("lsq_results_pack_device", C.c_int, vp, vp)
_sig("lsq_gathered_unpack", C.c_int, vp, C.c_int, P(u64), P(u64), P(u64), u64, P(u64), P(u64), P(C.c_double), P(C.c_double))
_sig("lsq_device_alloc", C.c_int, vp, u64, P(vp))
_sig("lsq_device_free", None, vp, vp)
_sig("lsq_device_read", C.c_int, vp, vp, vp, u64)
_sig("lsq_device_write", C.c_int, vp, vp, vp, u64)
_sig("lsq_results_device_order", C.c_int, vp, P(i32))
_sig("lsq_fim", C.c_int, vp)
_sig("lsq_results_fim_size", C.c_int64, vp)
_sig("lsq_results_fim_offsets", C.c_int, vp, P(C.c_uint64))
_sig("lsq_results_fim", C.c_int, vp, P(C.c_double), P(C.c_double), P(C.c_double))
_sig("lsq_set_timing", C.c_int, vp, C.c_int)
_sig("lsq_last_fast_kernel_ms", C.c_int, vp, P(C.c_float))
_sig("lsq_last_timing", C.c_int, vp, P(C.c_float), P(C.c_float))
_sig("lsq_format_count", C.c_int, vp, C.c_int, P(u64), P(vp))
_sig("lsq_format_solve", C.c_int, vp, C.c_int, P(u64), P(u64), P(C.c_double), P(C.c_double), P(C.c_double), P(vp))
_sig("lsq_cli_run", C.c_int, cs, C.c_int, P(cs), P(vp))
_sig("lsq_synth_write", C.c_int, P(SynthSpecStruct), cs, cs, C.c_int)
_sig("lsq_synth_reads", C.c_int, P(SynthSpecStruct), vp, C.c_int, P(vp))


def _warn_on_runtime_mismatch():
    """The library was compiled against one HIP (hipcc's) and runs on whichever libamdhip64 the process holds (torch's,
    when torch came first -- on this image 7.0 under a library built with 7.2, which works): another MAJOR version would
    show up far from its cause (LSQ_E_DEVICE, missing code objects), so that is said at import."""
    comp, run = hip_versions()
    if run and comp // 10000000 != run // 10000000:
        import warnings
        warnings.warn("liblesseq_hip.so was compiled against HIP %d and runs on HIP runtime %d" % (comp, run))


def hip_versions():
    """(HIP_VERSION the library was compiled against, hipRuntimeGetVersion() of the runtime in this process; 0 = unknown)"""
    comp, run = C.c_int(0), C.c_int(0)
    lib.lsq_debug_hip_versions(C.byref(comp), C.byref(run))
    return comp.value, run.value


# (not at import: hipRuntimeGetVersion may start the HIP runtime, and importing the package -- build(), the CPU test run, a launcher
# that goes on to spawn workers -- must make no HIP call.  lesseq_amd.api.Context makes the check when the first context exists.)
_runtime_checked = False


def check_runtime_once():
    global _runtime_checked
    if not _runtime_checked:
        _runtime_checked = True
        _warn_on_runtime_mismatch()


def check(status):
    if status != 0:
        raise LsqError(status, lib.lsq_last_error().decode("utf-8", "replace"))
