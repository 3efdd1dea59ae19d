"""lesseq_amd -- MI355X-native count + solve path of LESSeq.

The product is the C-ABI library `liblesseq_hip.so` (include/lesseq_hip.h: host logic in C++,
hand-written HIP kernels for gfx950) and the `count` / `solve` / `classify` executables built
from it.  This package is only the ctypes binding used by the tests and bench.py; it holds
no compute of its own and refuses to import without the built library.
"""
from ._lib import lib, LsqError, check  # noqa: F401
from .api import (Annotation, Events, Reads, Context, SynthSpec, cli_run, synth_write,  # noqa: F401
                  format_count, format_solve, EVENT_TYPES)

__all__ = ["lib", "LsqError", "check", "Annotation", "Events", "Reads", "Context", "SynthSpec",
           "cli_run", "synth_write", "format_count", "format_solve", "EVENT_TYPES"]
