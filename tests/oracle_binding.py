"""ctypes binding of the parity oracle (oracle/lsq_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "liblsq_oracle.so")
if not os.path.exists(_SO):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"])
_o = C.CDLL(_SO)
_o.lsqo_run.restype = C.c_int
_o.lsqo_run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
_o.lsqo_free.argtypes = [C.c_void_p]
for name, res, args in [
    ("lsqo_exact_n_genes", C.c_int, [C.c_void_p]),
    ("lsqo_exact_gname", C.c_char_p, [C.c_void_p, C.c_int]),
    ("lsqo_exact_K", C.c_int, [C.c_void_p, C.c_int]),
    ("lsqo_exact_support", C.c_ulong, [C.c_void_p, C.c_int, C.c_int]),
    ("lsqo_exact_bases", C.c_ulong, [C.c_void_p, C.c_int, C.c_int]),
    ("lsqo_exact_iso_count", C.c_ulong, [C.c_void_p, C.c_int, C.c_int]),
    ("lsqo_exact_theta", C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    ("lsqo_exact_logll", C.c_double, [C.c_void_p, C.c_int]),
    ("lsqo_exact_iters", C.c_ulong, [C.c_void_p, C.c_int]),
    ("lsqo_exact_n_loaded", C.c_ulong, [C.c_void_p, C.c_int]),
    ("lsqo_exact_has_fim", C.c_int, [C.c_void_p, C.c_int]),
    ("lsqo_exact_fim", C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("lsqo_exact_fim_var", C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
]:
    f = getattr(_o, name)
    f.restype = res
    f.argtypes = args
_o.lsqo_segments.restype = C.c_int
_o.lsqo_merge_intervals.restype = C.c_int
_o.lsqo_build_mask.restype = C.c_ulonglong
_o.lsqo_ars_total.restype = C.c_ulong
_o.lsqo_em_rows.restype = C.c_ulong
_o.lsqo_connected_compat.restype = C.c_long


last_n_loaded = []      # retained reads per read file of the most recent run()


def run(tool, argv, n_methods=None):
    """Runs the oracle's count/solve with the reference's argv (without argv[0]).
    Returns (exit_status, stdout_text, exact) where exact is a list of per-gene dicts."""
    is_solve = 1 if tool == "solve" else 0
    full = [b"oracle"] + [a.encode() if isinstance(a, str) else a for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    text, ex = C.c_void_p(), C.c_void_p()
    rc = _o.lsqo_run(is_solve, len(full), arr, C.byref(text), C.byref(ex))
    out = C.string_at(text).decode() if text else ""
    if text:
        _o.lsqo_free(text)
    per = 5 if is_solve else 4
    M = n_methods if n_methods is not None else max((len(argv) - 9) // per, 0)
    exact = []
    del last_n_loaded[:]
    if rc == 0 and ex:
        last_n_loaded.extend(_o.lsqo_exact_n_loaded(ex, m) for m in range(M))
        for g in range(_o.lsqo_exact_n_genes(ex)):
            K = _o.lsqo_exact_K(ex, g)
            exact.append({
                "gname": _o.lsqo_exact_gname(ex, g).decode(),
                "K": K,
                "supports": [_o.lsqo_exact_support(ex, g, m) for m in range(M)],
                "bases": [_o.lsqo_exact_bases(ex, g, m) for m in range(M)],
                "iso_count": [_o.lsqo_exact_iso_count(ex, g, k) for k in range(K)],
                "theta": [_o.lsqo_exact_theta(ex, g, k) for k in range(K)] if is_solve else None,
                "logll": _o.lsqo_exact_logll(ex, g) if is_solve else None,
                "iters": _o.lsqo_exact_iters(ex, g) if is_solve else None,
            })
            if is_solve and _o.lsqo_exact_has_fim(ex, g):      # LSQO_FIM in the environment (parity unpinned, fim.h)
                D = K - 1
                exact[-1]["fim"] = [[[_o.lsqo_exact_fim(ex, g, m, p, q) for q in range(D)] for p in range(D)] for m in range(M)]
                exact[-1]["fim_var"] = [(_o.lsqo_exact_fim_var(ex, g, m, 0), _o.lsqo_exact_fim_var(ex, g, m, 1)) for m in range(M)]
    return rc, out, exact


def _arr(vals, ct=C.c_long):
    return (ct * max(len(vals), 1))(*vals)


def segments(exons):
    """ExonSet::insert over (start,end) pairs in order -> atomic segments"""
    s, e = _arr([x[0] for x in exons]), _arr([x[1] for x in exons])
    os_, oe = (C.c_long * 256)(), (C.c_long * 256)()
    n = _o.lsqo_segments(s, e, len(exons), os_, oe, 256)
    return [(os_[i], oe[i]) for i in range(n)]


def merge_intervals(ivs):
    s, e = _arr([x[0] for x in ivs]), _arr([x[1] for x in ivs])
    os_, oe = (C.c_long * 256)(), (C.c_long * 256)()
    n = _o.lsqo_merge_intervals(s, e, len(ivs), os_, oe, 256)
    return [(os_[i], oe[i]) for i in range(n)]


def build_mask(blocks, segs):
    bs, be = _arr([x[0] for x in blocks]), _arr([x[1] for x in blocks])
    ss, se = _arr([x[0] for x in segs]), _arr([x[1] for x in segs])
    m = C.c_ulong()
    mask = _o.lsqo_build_mask(bs, be, len(blocks), ss, se, len(segs), C.byref(m))
    return mask, m.value


def ars_total(seg_len, iso_idx, R, short_read=True):
    return _o.lsqo_ars_total(_arr(seg_len, C.c_ulong), _arr(iso_idx, C.c_int), len(iso_idx), C.c_ulong(R), 1 if short_read else 0)


def em_rows(K, rows):
    flat = [v for r in rows for v in r]
    g = (C.c_double * max(len(flat), 1))(*flat)
    theta = (C.c_double * K)()
    ll = C.c_double()
    it = _o.lsqo_em_rows(K, g, C.c_ulong(len(rows)), theta, C.byref(ll))
    return list(theta), ll.value, it


def _num(tok):
    try:
        return float(tok)
    except ValueError:
        return None


def solve_text_close(a, b, rel=2e-5):
    """Two solve tables agree: same rows and names; numeric columns equal as printed or within
    one unit of the sixth significant digit (the tables carry six digits)."""
    la, lb = a.splitlines(), b.splitlines()
    if len(la) != len(lb):
        return False
    for x, y in zip(la, lb):
        if x == y:
            continue
        fx, fy = x.split("\t"), y.split("\t")
        if len(fx) != len(fy):
            return False
        for p, q in zip(fx, fy):
            if p == q:
                continue
            u, v = _num(p), _num(q)
            if u is None or v is None:
                return False
            if u != u and v != v:
                continue
            if abs(u - v) > rel * max(abs(u), abs(v)):
                return False
    return True
