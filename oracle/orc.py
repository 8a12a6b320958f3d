"""ctypes front end of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/basetype_oracle.h).  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from basevarc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


class OrcResult(C.Structure):
    _fields_ = [
        ("called", C.c_int32), ("n_alt", C.c_int32), ("alt_base", C.c_int8 * 4),
        ("af", C.c_double * 4), ("var_qual", C.c_double), ("chi", C.c_double),
        ("depth_total", C.c_double), ("depth", C.c_int32 * 4), ("n_kept", C.c_int32),
        ("kept", C.c_int8 * 4), ("base_frq", C.c_double * 4), ("lr_alt", C.c_double),
        ("n_fits", C.c_int32), ("n_passes", C.c_int32), ("status", C.c_int32), ("tie_gap", C.c_double),
        ("n_fits_pruned", C.c_int32), ("n_passes_pruned", C.c_int32), ("prune_edge", C.c_double),
        ("max_quals", C.c_int32), ("min_qual", C.c_int32), ("dup_candidate", C.c_int32),
    ]

    def as_dict(self):
        return dict(
            called=int(self.called), n_alt=int(self.n_alt),
            alt_base=[int(self.alt_base[i]) for i in range(self.n_alt)],
            af=[float(self.af[i]) for i in range(self.n_alt)],
            var_qual=float(self.var_qual), chi=float(self.chi),
            depth_total=float(self.depth_total), depth=[int(x) for x in self.depth],
            kept=[int(self.kept[i]) for i in range(self.n_kept)],
            base_frq=[float(x) for x in self.base_frq], lr_alt=float(self.lr_alt),
            n_fits=int(self.n_fits), n_passes=int(self.n_passes), status=int(self.status),
            tie_gap=float(self.tie_gap), n_fits_pruned=int(self.n_fits_pruned),
            n_passes_pruned=int(self.n_passes_pruned), prune_edge=float(self.prune_edge),
            max_quals=int(self.max_quals), min_qual=int(self.min_qual), dup_candidate=int(self.dup_candidate))


def build(force=False):
    """Compile oracle/liborc.so with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("basetype_oracle.c", "basetype_oracle.h", "synth_tables.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i8p, u8p, u32p, i32p, f64p = (C.POINTER(t) for t in (C.c_int8, C.c_uint8, C.c_uint32, C.c_int32, C.c_double))
        rp = C.POINTER(OrcResult)
        L.orc_kf_lgamma.restype = C.c_double; L.orc_kf_lgamma.argtypes = [C.c_double]
        L.orc_kf_gammaq.restype = C.c_double; L.orc_kf_gammaq.argtypes = [C.c_double, C.c_double]
        L.orc_chisf.restype = C.c_double; L.orc_chisf.argtypes = [C.c_double, C.c_double]
        L.orc_basetype_lrt.restype = C.c_int
        L.orc_basetype_lrt.argtypes = [C.c_int32, i8p, i8p, C.c_int8, C.c_double, i8p, C.c_int32, rp]
        L.orc_basetype_lrt_mode.restype = C.c_int
        L.orc_basetype_lrt_mode.argtypes = [C.c_int32, i8p, i8p, C.c_int8, C.c_double, i8p, C.c_int32, C.c_int, rp]
        L.orc_hist_lrt.restype = C.c_int
        L.orc_hist_lrt.argtypes = [u32p, C.c_int8, C.c_double, i8p, C.c_int32, rp]
        L.orc_hist_lrt_mode.restype = C.c_int
        L.orc_hist_lrt_mode.argtypes = [u32p, C.c_int8, C.c_double, i8p, C.c_int32, C.c_int, rp]
        L.orc_dense_site.restype = C.c_int
        L.orc_dense_site.argtypes = [C.c_int64, i8p, i8p, C.c_int8, C.c_double, rp]
        L.orc_dense_hist.restype = None
        L.orc_dense_hist.argtypes = [C.c_int64, i8p, i8p, u8p, C.c_int32, u32p]
        L.orc_dense_batch.restype = C.c_int
        L.orc_dense_batch.argtypes = [C.c_int64, C.c_int64, C.c_int64, i8p, i8p, i8p, C.c_double,
                                      C.c_int, C.c_int, rp]
        L.orc_dense_site_groups.restype = C.c_int
        L.orc_dense_site_groups.argtypes = [C.c_int64, i8p, i8p, C.c_int8, C.c_double, u8p, C.c_int32,
                                            C.c_int, rp, i32p, f64p, i32p, i32p]
        L.orc_synth_site.restype = None
        L.orc_synth_site.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_uint32, i8p, i8p, i8p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _comb(base_comb):
    if base_comb is None:
        return None, 0
    a = np.ascontiguousarray(base_comb, dtype=np.int8)
    return a, len(a)


def chisf(x, k=1.0):
    return lib().orc_chisf(float(x), float(k))


MODE_HIST = 1          # ORC_MODE_HIST
MODE_COMPENSATED = 2   # ORC_MODE_COMPENSATED: long-double sums over samples -- NOT the reference's arithmetic


def _mode(use_hist, compensated):
    return (MODE_HIST if use_hist else 0) | (MODE_COMPENSATED if compensated else 0)


def basetype_lrt(bases, quals, ref_base, min_af, base_comb=None, compensated=False):
    """BaseType(bases, quals, ref, min_af) [+ SetBase(base_comb)] + LRT() -- faithful per-sample path.
    compensated=True keeps the loops but accumulates the sums over samples in long double (oracle header)."""
    b = np.ascontiguousarray(bases, dtype=np.int8)
    q = np.ascontiguousarray(quals, dtype=np.int8)
    cb, nc = _comb(base_comb)
    r = OrcResult()
    lib().orc_basetype_lrt_mode(len(b), _p(b, C.c_int8), _p(q, C.c_int8), int(ref_base), float(min_af),
                                _p(cb, C.c_int8) if cb is not None else None, nc, _mode(False, compensated),
                                C.byref(r))
    return r.as_dict()


def hist_lrt(counts512, ref_base, min_af, base_comb=None, compensated=False):
    c = np.ascontiguousarray(counts512, dtype=np.uint32).reshape(512)
    cb, nc = _comb(base_comb)
    r = OrcResult()
    lib().orc_hist_lrt_mode(_p(c, C.c_uint32), int(ref_base), float(min_af),
                            _p(cb, C.c_int8) if cb is not None else None, nc, _mode(True, compensated), C.byref(r))
    return r.as_dict()


def dense_hist(bases_row, quals_row, group_of_sample=None, group=-1):
    b = np.ascontiguousarray(bases_row, dtype=np.int8)
    q = np.ascontiguousarray(quals_row, dtype=np.int8)
    out = np.zeros(512, dtype=np.uint32)
    g = None if group_of_sample is None else np.ascontiguousarray(group_of_sample, dtype=np.uint8)
    lib().orc_dense_hist(len(b), _p(b, C.c_int8), _p(q, C.c_int8),
                         _p(g, C.c_uint8) if g is not None else None, int(group), _p(out, C.c_uint32))
    return out


def dense_batch(bases, quals, ref_base, min_af, use_hist=False, threads=0, compensated=False):
    """bases/quals: int8 [n_sites, n_samples] (C-contiguous).  Returns (list of dicts, threads used)."""
    b = np.ascontiguousarray(bases, dtype=np.int8)
    q = np.ascontiguousarray(quals, dtype=np.int8)
    r = np.ascontiguousarray(ref_base, dtype=np.int8)
    ns, n = b.shape
    res = (OrcResult * ns)()
    used = lib().orc_dense_batch(ns, n, n, _p(b, C.c_int8), _p(q, C.c_int8), _p(r, C.c_int8),
                                 float(min_af), _mode(use_hist, compensated), int(threads), res)
    return [x.as_dict() for x in res], used


def dense_site_groups(bases_row, quals_row, ref_base, min_af, group_of_sample, n_groups, use_hist=False,
                      compensated=False):
    b = np.ascontiguousarray(bases_row, dtype=np.int8)
    q = np.ascontiguousarray(quals_row, dtype=np.int8)
    g = np.ascontiguousarray(group_of_sample, dtype=np.uint8)
    r = OrcResult()
    gd = np.zeros((n_groups, 4), dtype=np.int32)
    ga = np.zeros((n_groups, 3), dtype=np.float64)
    gr = np.zeros(n_groups, dtype=np.int32)
    gp = np.zeros(n_groups, dtype=np.int32)
    lib().orc_dense_site_groups(len(b), _p(b, C.c_int8), _p(q, C.c_int8), int(ref_base), float(min_af),
                                _p(g, C.c_uint8), int(n_groups), _mode(use_hist, compensated), C.byref(r),
                                _p(gd, C.c_int32), _p(ga, C.c_double), _p(gr, C.c_int32), _p(gp, C.c_int32))
    return r.as_dict(), gd, ga, gr, gp


def synth_tile(seed, site0, n_sites, n_samples, cov_thr16=65536):
    """CPU synthetic pileup rows for sites site0..site0+n_sites-1 (bit-identical to the device generator)."""
    b = np.empty((n_sites, n_samples), dtype=np.int8)
    q = np.empty((n_sites, n_samples), dtype=np.int8)
    r = np.empty(n_sites, dtype=np.int8)
    L = lib()
    for s in range(n_sites):
        L.orc_synth_site(int(seed), int(site0 + s), int(n_samples), int(cov_thr16),
                         _p(b[s], C.c_int8), _p(q[s], C.c_int8),
                         C.cast(r.ctypes.data + s, C.POINTER(C.c_int8)))
    return b, q, r
