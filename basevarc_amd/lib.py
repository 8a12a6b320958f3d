"""ctypes binding of libbvc.so (include/bvc.h).  Device memory, streams and multi-process plumbing come
from torch; the arithmetic is all inside the library's HIP kernels."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("BVC_LIBBVC") or os.path.join(_HERE, "libbvc.so")      # (BVC_LIBBVC: a variant build for A/B runs, tools/)

BVC_PTR_HOST = 0
BVC_PTR_DEVICE = 1
NCLASS = 512


class BvcError(RuntimeError):
    pass


class SiteResult(C.Structure):          # bvc_site_result, 120 bytes
    _fields_ = [
        ("var_qual", C.c_double), ("chi", C.c_double), ("depth_total", C.c_double),
        ("af", C.c_double * 3), ("lr_alt", C.c_double), ("base_frq", C.c_double * 4),
        ("depth", C.c_int32 * 4), ("n_passes", C.c_int32), ("alt_base", C.c_int8 * 3),
        ("n_alt", C.c_uint8), ("called", C.c_uint8), ("n_kept", C.c_uint8), ("kept", C.c_int8 * 4),
        ("status", C.c_uint8), ("n_fits", C.c_uint8),
    ]


class GroupResult(C.Structure):         # bvc_group_result, 48 bytes
    _fields_ = [("af", C.c_double * 3), ("depth", C.c_int32 * 4), ("ran", C.c_uint8), ("present", C.c_uint8),
                ("pad", C.c_uint8 * 6)]


class Profile(C.Structure):
    _fields_ = [("hist_ms", C.c_double), ("em_ms", C.c_double), ("hist_launches", C.c_int64),
                ("em_launches", C.c_int64), ("sites", C.c_int64)]


SITE_DTYPE = np.dtype([
    ("var_qual", "<f8"), ("chi", "<f8"), ("depth_total", "<f8"), ("af", "<f8", (3,)), ("lr_alt", "<f8"),
    ("base_frq", "<f8", (4,)), ("depth", "<i4", (4,)), ("n_passes", "<i4"), ("alt_base", "i1", (3,)),
    ("n_alt", "u1"), ("called", "u1"), ("n_kept", "u1"), ("kept", "i1", (4,)), ("status", "u1"), ("n_fits", "u1"),
])
GROUP_DTYPE = np.dtype([("af", "<f8", (3,)), ("depth", "<i4", (4,)), ("ran", "u1"), ("present", "u1"), ("pad", "u1", (6,))])
assert SITE_DTYPE.itemsize == C.sizeof(SiteResult) == 120
assert GROUP_DTYPE.itemsize == C.sizeof(GroupResult) == 48

EXPORTS = [
    "bvc_version", "bvc_device_count", "bvc_create", "bvc_destroy", "bvc_last_error", "bvc_set_stream",
    "bvc_synchronize", "bvc_set_overlap", "bvc_join", "bvc_set_profiling", "bvc_get_profile", "bvc_lrt_dense", "bvc_lrt_dense_groups",
    "bvc_lrt_csr", "bvc_lrt_csr_comb", "bvc_hist_dense", "bvc_lrt_hist", "bvc_synth_dense", "bvc_stream_read_ms", "bvc_set_tuning",
    "bvc_lrt_dense_packed", "bvc_pack_dense", "bvc_hist_dense_packed", "bvc_lrt_dense_groups_packed",
    "bvc_lrt_csr_packed", "bvc_lrt_csr_groups", "bvc_pileup_begin", "bvc_pileup_finish", "bvc_pileup_finish_called", "bvc_inflate_blocks", "bvc_pileup_begin_bgzf", "bvc_pileup_text",
    "bvc_host_alloc", "bvc_host_free",
]

_lib = None


def library_path():
    return _LIB


def load_library():
    """Loads libbvc.so.  Raises if it has not been built (there is no fallback implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch is this package's plumbing layer (device memory, streams, torch.distributed) and ships its own
    # HIP runtime.  It must be in the process BEFORE libbvc.so so that libbvc's libamdhip64.so.7 dependency
    # resolves to that same runtime: two HIP runtimes in one process cannot both open the device.
    import torch  # noqa: F401
    if not os.path.exists(_LIB):
        raise BvcError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); basevarc_amd has no CPU fallback")
    L = C.CDLL(_LIB)
    vp, i64, u32, i32, dbl = C.c_void_p, C.c_int64, C.c_uint32, C.c_int32, C.c_double
    L.bvc_version.restype = C.c_char_p
    L.bvc_device_count.restype = C.c_int
    L.bvc_create.restype = C.c_int; L.bvc_create.argtypes = [C.POINTER(vp), C.c_int]
    L.bvc_destroy.restype = None; L.bvc_destroy.argtypes = [vp]
    L.bvc_last_error.restype = C.c_char_p; L.bvc_last_error.argtypes = [vp]
    L.bvc_set_stream.restype = C.c_int; L.bvc_set_stream.argtypes = [vp, vp]
    L.bvc_synchronize.restype = C.c_int; L.bvc_synchronize.argtypes = [vp]
    L.bvc_set_overlap.restype = C.c_int; L.bvc_set_overlap.argtypes = [vp, C.c_int]
    L.bvc_join.restype = C.c_int; L.bvc_join.argtypes = [vp]
    L.bvc_set_profiling.restype = C.c_int; L.bvc_set_profiling.argtypes = [vp, C.c_int]
    L.bvc_get_profile.restype = C.c_int; L.bvc_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.bvc_lrt_dense.restype = C.c_int
    L.bvc_lrt_dense.argtypes = [vp, i64, i64, i64, vp, vp, vp, dbl, vp, u32]
    L.bvc_lrt_dense_groups.restype = C.c_int
    L.bvc_lrt_dense_groups.argtypes = [vp, i64, i64, i64, vp, vp, vp, dbl, vp, i32, vp, vp, u32]
    L.bvc_lrt_csr.restype = C.c_int
    L.bvc_lrt_csr.argtypes = [vp, i64, vp, vp, vp, vp, dbl, vp, u32]
    L.bvc_hist_dense.restype = C.c_int
    L.bvc_hist_dense.argtypes = [vp, i64, i64, i64, vp, vp, vp, u32]
    L.bvc_lrt_hist.restype = C.c_int
    L.bvc_lrt_hist.argtypes = [vp, i64, vp, vp, dbl, vp, vp, vp, u32]
    L.bvc_synth_dense.restype = C.c_int
    L.bvc_synth_dense.argtypes = [vp, C.c_uint64, i64, i64, i64, i64, u32, vp, vp, vp]
    L.bvc_lrt_csr_comb.restype = C.c_int
    L.bvc_lrt_csr_comb.argtypes = [vp, i64, vp, vp, vp, vp, dbl, vp, vp, vp, u32]
    L.bvc_set_tuning.restype = C.c_int; L.bvc_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
    L.bvc_stream_read_ms.restype = C.c_int
    L.bvc_stream_read_ms.argtypes = [vp, vp, i64, C.c_int, C.POINTER(C.c_double)]
    L.bvc_lrt_dense_packed.restype = C.c_int
    L.bvc_lrt_dense_packed.argtypes = [vp, i64, i64, i64, vp, vp, dbl, vp, u32]
    L.bvc_pack_dense.restype = C.c_int
    L.bvc_pack_dense.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp, C.POINTER(i64), u32]
    L.bvc_lrt_dense_groups_packed.restype = C.c_int
    L.bvc_lrt_dense_groups_packed.argtypes = [vp, i64, i64, i64, vp, vp, dbl, vp, i32, vp, vp, u32]
    L.bvc_hist_dense_packed.restype = C.c_int
    L.bvc_hist_dense_packed.argtypes = [vp, i64, i64, i64, vp, vp, u32]
    L.bvc_lrt_csr_packed.restype = C.c_int
    L.bvc_lrt_csr_packed.argtypes = [vp, i64, vp, vp, vp, dbl, vp, u32]
    L.bvc_lrt_csr_groups.restype = C.c_int
    L.bvc_lrt_csr_groups.argtypes = [vp, i64, vp, vp, vp, vp, vp, dbl, vp, i64, i32, vp, vp, u32]
    L.bvc_pileup_begin.restype = C.c_int
    L.bvc_pileup_begin.argtypes = [vp, vp, i64, vp, vp, vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.bvc_pileup_finish.restype = C.c_int
    L.bvc_pileup_finish.argtypes = [vp, vp, dbl, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bvc_pileup_finish_called.restype = C.c_int
    L.bvc_pileup_finish_called.argtypes = [vp, vp, dbl, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.bvc_pileup_begin_bgzf.restype = C.c_int
    L.bvc_pileup_begin_bgzf.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(i32), vp, C.POINTER(i64), C.POINTER(i64),
                                        C.POINTER(i64)]
    L.bvc_pileup_text.restype = C.c_int
    L.bvc_pileup_text.argtypes = [vp, vp, i64, C.POINTER(i64), vp]
    L.bvc_inflate_blocks.restype = C.c_int
    L.bvc_inflate_blocks.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, u32]
    _lib = L
    return L


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr())


class Context:
    """One bvc_ctx: one gfx950 device + one HIP stream.  Not shared between threads."""

    def __init__(self, device=0, stream=None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.bvc_create(C.byref(h), int(device))
        if rc != 0:
            raise BvcError(f"bvc_create(device={device}) failed with code {rc}: no usable gfx950 device "
                           "(libbvc has no CPU fallback)")
        self._h = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self._L.bvc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise BvcError(f"libbvc error {rc}: {self._L.bvc_last_error(self._h).decode()}")

    # ---- plumbing
    def set_stream(self, stream):
        """stream: a torch.cuda.Stream, a raw hipStream_t integer, or None for the default stream."""
        raw = 0 if stream is None else int(getattr(stream, "cuda_stream", stream))
        self._check(self._L.bvc_set_stream(self._h, C.c_void_p(raw)))

    def synchronize(self):
        self._check(self._L.bvc_synchronize(self._h))

    def set_overlap(self, on=True):
        """Run stage 2 of each device-pointer call under stage 1 of the next; results need join()/synchronize()."""
        self._check(self._L.bvc_set_overlap(self._h, int(bool(on))))

    def join(self):
        self._check(self._L.bvc_join(self._h))

    def set_profiling(self, on=True):
        self._check(self._L.bvc_set_profiling(self._h, int(bool(on))))

    def profile(self, reset=False):
        p = Profile()
        self._check(self._L.bvc_get_profile(self._h, C.byref(p), int(bool(reset))))
        return dict(hist_ms=p.hist_ms, em_ms=p.em_ms, hist_launches=p.hist_launches,
                    em_launches=p.em_launches, sites=p.sites)

    def debug_report(self, reset=False):
        """Diagnostic builds of the library only (-DBVC_CHECK_LDS, csrc/bvc_device.h): the recorded LDS bound violations as
        {translation unit: [count, check id, value, limit, blockIdx.x, threadIdx.x]}; None with the product library."""
        if not hasattr(self._L, "bvc_debug_report"):
            return None
        out = (C.c_uint32 * 24)()
        self._L.bvc_debug_report.restype = C.c_int
        self._L.bvc_debug_report.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        self._check(self._L.bvc_debug_report(self._h, out, int(bool(reset))))
        return {tu: [int(x) for x in out[8 * i:8 * i + 6]] for i, tu in enumerate(("hist_kernel", "em_kernel", "em_items"))}

    # ---- host-pointer calls (numpy in, numpy structured array out)
    def lrt_dense(self, bases, quals, ref_base, min_af):
        b = np.ascontiguousarray(bases, dtype=np.int8)
        q = np.ascontiguousarray(quals, dtype=np.int8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        if b.ndim != 2 or b.shape != q.shape or r.shape != (b.shape[0],):
            raise ValueError("bases/quals must be [n_sites, n_samples] and ref_base [n_sites]")
        out = np.zeros(b.shape[0], dtype=SITE_DTYPE)
        self._check(self._L.bvc_lrt_dense(self._h, b.shape[0], b.shape[1], b.shape[1], _np_ptr(b), _np_ptr(q),
                                          _np_ptr(r), float(min_af), _np_ptr(out), BVC_PTR_HOST))
        return out

    def lrt_dense_groups(self, bases, quals, ref_base, min_af, group_of_sample, n_groups):
        b = np.ascontiguousarray(bases, dtype=np.int8)
        q = np.ascontiguousarray(quals, dtype=np.int8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        g = np.ascontiguousarray(group_of_sample, dtype=np.uint8)
        if b.ndim != 2 or b.shape != q.shape or r.shape != (b.shape[0],) or g.shape != (b.shape[1],):
            raise ValueError("shape mismatch")
        out = np.zeros(b.shape[0], dtype=SITE_DTYPE)
        gout = np.zeros((b.shape[0], n_groups), dtype=GROUP_DTYPE)
        self._check(self._L.bvc_lrt_dense_groups(self._h, b.shape[0], b.shape[1], b.shape[1], _np_ptr(b),
                                                 _np_ptr(q), _np_ptr(r), float(min_af), _np_ptr(g), int(n_groups),
                                                 _np_ptr(out), _np_ptr(gout), BVC_PTR_HOST))
        return out, gout

    def lrt_csr(self, offsets, bases, quals, ref_base, min_af, base_comb=None, n_comb=None):
        """Ragged sites (the vectors bt_f builds); base_comb/n_comb: optional per-site SetBase lists."""
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        b = np.ascontiguousarray(bases, dtype=np.int8)
        q = np.ascontiguousarray(quals, dtype=np.int8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        n = len(o) - 1
        out = np.zeros(n, dtype=SITE_DTYPE)
        cb = nc = None
        if base_comb is not None:
            cb = np.ascontiguousarray(base_comb, dtype=np.int8).reshape(-1, 4)
            nc = np.ascontiguousarray(n_comb, dtype=np.uint8)
        self._check(self._L.bvc_lrt_csr_comb(self._h, n, _np_ptr(o), _np_ptr(b), _np_ptr(q), _np_ptr(r), float(min_af),
                                             _np_ptr(cb) if cb is not None else None,
                                             _np_ptr(nc) if nc is not None else None, _np_ptr(out), BVC_PTR_HOST))
        return out

    def lrt_csr_groups(self, offsets, bases, quals, sample_of_obs, ref_base, min_af, group_of_sample, n_groups):
        """The --group loop on ragged sites: per observation its sample index; group_of_sample[n_samples] (>= n_groups: no group)."""
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        b = np.ascontiguousarray(bases, dtype=np.int8)
        q = np.ascontiguousarray(quals, dtype=np.int8)
        sm = np.ascontiguousarray(sample_of_obs, dtype=np.int32)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        g = np.ascontiguousarray(group_of_sample, dtype=np.uint8)
        n = len(o) - 1
        out = np.zeros(n, dtype=SITE_DTYPE)
        gout = np.zeros((n, n_groups), dtype=GROUP_DTYPE)
        self._check(self._L.bvc_lrt_csr_groups(self._h, n, _np_ptr(o), _np_ptr(b), _np_ptr(q), _np_ptr(sm), _np_ptr(r), float(min_af),
                                               _np_ptr(g), len(g), int(n_groups), _np_ptr(out), _np_ptr(gout), BVC_PTR_HOST))
        return out, gout

    def lrt_csr_groups_device(self, offsets_t, bases_t, quals_t, samples_t, ref_t, min_af, group_t, n_groups):
        import torch
        ns = offsets_t.numel() - 1
        res = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        gres = torch.empty(ns * n_groups * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        self._check(self._L.bvc_lrt_csr_groups(self._h, ns, _dev_ptr(offsets_t), _dev_ptr(bases_t), _dev_ptr(quals_t), _dev_ptr(samples_t),
                                               _dev_ptr(ref_t), float(min_af), _dev_ptr(group_t), group_t.numel(), int(n_groups),
                                               _dev_ptr(res), _dev_ptr(gres), BVC_PTR_DEVICE))
        return res, gres

    def pileup_tile(self, text, line_start, sample0, n_in_batch, ref_base, min_af, carry_in=(0, 0, 0, 0, 0), group_of_sample=None,
                    n_groups=0, called_only=False):
        """bvc_pileup_begin + bvc_pileup_finish on one tile of temp-batch pileup text (include/bvc.h).  text: bytes;
        line_start: uint32 [n_batches, n_positions + 1].  Returns None when a line is not regular (BVC_PILEUP_IRREGULAR), else a
        dict: entry_off, tally [T, 32], entries (structured), samples, indels (sorted by entry), results, grp_results, carry_out."""
        ls = np.ascontiguousarray(line_start, dtype=np.uint32)
        nb, T = ls.shape[0], ls.shape[1] - 1
        s0 = np.ascontiguousarray(sample0, dtype=np.int32)
        nib = np.ascontiguousarray(n_in_batch, dtype=np.int32)
        buf = np.frombuffer(bytes(text), dtype=np.uint8)
        ne, ni = C.c_int64(0), C.c_int64(0)
        rc = self._L.bvc_pileup_begin(self._h, _np_ptr(buf) if len(buf) else None, len(buf), _np_ptr(ls), _np_ptr(s0), _np_ptr(nib), nb, T,
                                      C.byref(ne), C.byref(ni))
        if rc == 1:
            return None
        self._check(rc)
        return self._pileup_finish(T, ne.value, ni.value, 0, ref_base, min_af, carry_in, group_of_sample, n_groups, called_only)

    def _pileup_finish(self, T, n_entries, n_indels, indel_text_bytes, ref_base, min_af, carry_in, group_of_sample, n_groups,
                       called_only=False, called_cap=None):
        """called_only: bvc_pileup_finish_called -- `entries` / `samples` hold the called positions' entries only, position t's at
        called_off[t] .. called_off[t + 1] (key "called_off")."""
        ENTRY = np.dtype([("base", "u1"), ("mapq", "u1"), ("qual", "u1"), ("rpr", "u1"), ("strand", "u1"), ("is_indel", "u1"), ("pad", "<u2")])
        INDEL = np.dtype([("entry", "<i8"), ("text_off", "<i8"), ("len", "<i4"), ("pad", "<i4")])
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        assert r.shape == (T,)
        entry_off = np.zeros(T + 1, dtype=np.int64)
        tally = np.zeros((T, 32), dtype=np.int32)
        entries = np.zeros(max(1, n_entries), dtype=ENTRY)
        samples = np.zeros(max(1, n_entries), dtype=np.int32)
        indels = np.zeros(max(1, n_indels), dtype=INDEL)
        itext = np.zeros(max(1, indel_text_bytes), dtype=np.uint8)
        res = np.zeros(T, dtype=SITE_DTYPE)
        gres = np.zeros((T, max(1, n_groups)), dtype=GROUP_DTYPE)
        g = np.ascontiguousarray(group_of_sample, dtype=np.uint8) if n_groups else np.zeros(0, dtype=np.uint8)
        cin = np.asarray(carry_in, dtype=np.uint8)
        cout = np.zeros(5, dtype=np.uint8)
        if called_only:
            called_off = np.zeros(T + 1, dtype=np.int64)
            cap = n_entries if called_cap is None else int(called_cap)
            self._check(self._L.bvc_pileup_finish_called(self._h, _np_ptr(r), float(min_af), _np_ptr(cin), _np_ptr(cout),
                                                         _np_ptr(g) if n_groups else None, len(g), int(n_groups), _np_ptr(entry_off), _np_ptr(tally),
                                                         _np_ptr(called_off), cap, _np_ptr(entries), _np_ptr(samples), _np_ptr(indels),
                                                         _np_ptr(itext) if indel_text_bytes else None, _np_ptr(res),
                                                         _np_ptr(gres) if n_groups else None))
            n_c = int(called_off[T])
            ind = indels[:n_indels]
            ind = ind[np.argsort(ind["entry"], kind="stable")]
            return dict(entry_off=entry_off, called_off=called_off, tally=tally, entries=entries[:n_c], samples=samples[:n_c], indels=ind,
                        results=res, grp_results=gres if n_groups else None, carry_out=[int(x) for x in cout],
                        indel_text=itext[:indel_text_bytes].tobytes())
        self._check(self._L.bvc_pileup_finish(self._h, _np_ptr(r), float(min_af), _np_ptr(cin), _np_ptr(cout), _np_ptr(g) if n_groups else None,
                                              len(g), int(n_groups), _np_ptr(entry_off), _np_ptr(tally), _np_ptr(entries), _np_ptr(samples),
                                              _np_ptr(indels), _np_ptr(itext) if indel_text_bytes else None, _np_ptr(res),
                                              _np_ptr(gres) if n_groups else None))
        ind = indels[:n_indels]
        ind = ind[np.argsort(ind["entry"], kind="stable")]
        return dict(entry_off=entry_off, tally=tally, entries=entries[:n_entries], samples=samples[:n_entries], indels=ind, results=res,
                    grp_results=gres if n_groups else None, carry_out=[int(x) for x in cout], indel_text=itext[:indel_text_bytes].tobytes())

    def pileup_begin_bgzf(self, comp, blocks, blocks_of_batch, skip_bytes, sample0, n_in_batch, max_positions, reset):
        """bvc_pileup_begin_bgzf.  comp: bytes; blocks: [(comp_off, comp_len, isize)] batch after batch.  Returns a dict with rc
        (0, 1 = irregular, negative = error), T, lines (per batch) and the sizes bvc_pileup_finish needs."""
        BLOCK = np.dtype([("comp_off", "<i8"), ("out_off", "<i8"), ("comp_len", "<i4"), ("isize", "<i4"), ("crc32", "<u4"), ("check_crc", "<u4")])
        tab = np.zeros(max(1, len(blocks)), dtype=BLOCK)
        for i, blk in enumerate(blocks):
            co, cl, isz = blk[:3]
            tab[i] = (co, 0, cl, isz, blk[3] if len(blk) > 3 else 0, 1 if len(blk) > 3 else 0)
        if isinstance(comp, np.ndarray):                         # e.g. a view of page-locked memory (host_alloc): used where it lies
            buf = comp
            comp_len = len(buf)
        else:
            buf = np.frombuffer(bytes(comp) + b"\0" * 8, dtype=np.uint8)
            comp_len = len(buf) - 8
        bob = np.ascontiguousarray(blocks_of_batch, dtype=np.int32)
        nb = len(bob)
        sk = np.ascontiguousarray(skip_bytes, dtype=np.int32) if skip_bytes is not None else None
        s0 = np.ascontiguousarray(sample0, dtype=np.int32)
        nib = np.ascontiguousarray(n_in_batch, dtype=np.int32)
        lines = np.zeros(max(1, nb), dtype=np.int32)
        T, ne, ni, nt = C.c_int32(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = self._L.bvc_pileup_begin_bgzf(self._h, _np_ptr(buf), comp_len, _np_ptr(tab), _np_ptr(bob), _np_ptr(sk) if sk is not None else None,
                                           _np_ptr(s0), _np_ptr(nib), nb, int(max_positions), int(bool(reset)), C.byref(T), _np_ptr(lines),
                                           C.byref(ne), C.byref(ni), C.byref(nt))
        return dict(rc=rc, T=T.value, lines=lines[:nb].copy(), n_entries=ne.value, n_indels=ni.value, indel_text_bytes=nt.value,
                    error=self._L.bvc_last_error(self._h).decode() if rc < 0 else "")

    def host_alloc(self, nbytes):
        """bvc_host_alloc: (address, uint8 view of the page-locked bytes); free with host_free(address)."""
        self._L.bvc_host_alloc.restype = C.c_void_p
        self._L.bvc_host_alloc.argtypes = [C.c_size_t]
        addr = self._L.bvc_host_alloc(int(nbytes))
        if not addr:
            raise BvcError("bvc_host_alloc failed")
        return addr, np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(addr))

    def host_free(self, addr):
        self._L.bvc_host_free.restype = None
        self._L.bvc_host_free.argtypes = [C.c_void_p]
        self._L.bvc_host_free(addr)

    def pileup_text(self, n_batches, T):
        need = C.c_int64(0)
        self._check(self._L.bvc_pileup_text(self._h, None, 0, C.byref(need), None))
        text = np.zeros(max(1, need.value), dtype=np.uint8)
        ls = np.zeros((n_batches, T + 1), dtype=np.uint32)
        self._check(self._L.bvc_pileup_text(self._h, _np_ptr(text), need.value, C.byref(need), _np_ptr(ls)))
        return text[:need.value].tobytes(), ls

    def inflate_blocks(self, comp, blocks):
        """Raw-deflate streams inflated on the device.  comp: bytes; blocks: [(comp_off, comp_len, isize)] -- outputs are laid out one
        after the other.  Returns (list of bytes, status array)."""
        BLOCK = np.dtype([("comp_off", "<i8"), ("out_off", "<i8"), ("comp_len", "<i4"), ("isize", "<i4"), ("crc32", "<u4"), ("check_crc", "<u4")])
        tab = np.zeros(len(blocks), dtype=BLOCK)
        at = 0
        for i, blk in enumerate(blocks):                        # (comp_off, comp_len, isize[, crc32]): with a CRC it is compared
            co, cl, isz = blk[:3]
            tab[i] = (co, at, cl, isz, blk[3] if len(blk) > 3 else 0, 1 if len(blk) > 3 else 0)
            at += isz
        buf = np.frombuffer(bytes(comp) + b"\0" * 8, dtype=np.uint8)
        out = np.zeros(max(1, at), dtype=np.uint8)
        status = np.zeros(max(1, len(blocks)), dtype=np.uint32)
        self._check(self._L.bvc_inflate_blocks(self._h, _np_ptr(buf), len(buf), _np_ptr(tab), len(blocks), _np_ptr(out), at, _np_ptr(status),
                                               BVC_PTR_HOST))
        return [out[int(t["out_off"]):int(t["out_off"]) + int(t["isize"])].tobytes() for t in tab], status[:len(blocks)]

    def lrt_csr_packed(self, offsets, packed, ref_base, min_af):
        """Ragged sites at one byte per observation (base << 6 | qual); host arrays, synchronous."""
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        pk = np.ascontiguousarray(packed, dtype=np.uint8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        out = np.zeros(len(o) - 1, dtype=SITE_DTYPE)
        self._check(self._L.bvc_lrt_csr_packed(self._h, len(o) - 1, _np_ptr(o), _np_ptr(pk), _np_ptr(r), float(min_af),
                                               _np_ptr(out), BVC_PTR_HOST))
        return out

    def lrt_csr_packed_device(self, offsets_t, packed_t, ref_t, min_af, results_t=None):
        import torch
        ns = offsets_t.numel() - 1
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=packed_t.device)
        self._check(self._L.bvc_lrt_csr_packed(self._h, ns, _dev_ptr(offsets_t), _dev_ptr(packed_t), _dev_ptr(ref_t),
                                               float(min_af), _dev_ptr(results_t), BVC_PTR_DEVICE))
        return results_t

    def lrt_csr_device(self, offsets_t, bases_t, quals_t, ref_t, min_af, results_t=None):
        """offsets_t: int64 [n_sites + 1]; bases_t/quals_t: int8 [total] CUDA tensors (asynchronous on the stream)."""
        import torch
        ns = offsets_t.numel() - 1
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        self._check(self._L.bvc_lrt_csr(self._h, ns, _dev_ptr(offsets_t), _dev_ptr(bases_t), _dev_ptr(quals_t),
                                        _dev_ptr(ref_t), float(min_af), _dev_ptr(results_t), BVC_PTR_DEVICE))
        return results_t

    def set_tuning(self, key, value):
        """Launch policy of this context (include/bvc.h); results never depend on it."""
        self._check(self._L.bvc_set_tuning(self._h, key.encode(), int(value)))

    def hist_dense(self, bases, quals):
        b = np.ascontiguousarray(bases, dtype=np.int8)
        q = np.ascontiguousarray(quals, dtype=np.int8)
        out = np.zeros((b.shape[0], NCLASS), dtype=np.uint32)
        self._check(self._L.bvc_hist_dense(self._h, b.shape[0], b.shape[1], b.shape[1], _np_ptr(b), _np_ptr(q),
                                           _np_ptr(out), BVC_PTR_HOST))
        return out

    def lrt_hist(self, counts, ref_base, min_af, base_comb=None, n_comb=None):
        c = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1, NCLASS)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        out = np.zeros(c.shape[0], dtype=SITE_DTYPE)
        cb = nc = None
        if base_comb is not None:
            cb = np.ascontiguousarray(base_comb, dtype=np.int8).reshape(-1, 4)
            nc = np.ascontiguousarray(n_comb, dtype=np.uint8)
        self._check(self._L.bvc_lrt_hist(self._h, c.shape[0], _np_ptr(c), _np_ptr(r), float(min_af),
                                         _np_ptr(cb) if cb is not None else None,
                                         _np_ptr(nc) if nc is not None else None, _np_ptr(out), BVC_PTR_HOST))
        return out

    # ---- device-pointer calls (torch tensors on this context's device; asynchronous on the stream)
    def lrt_dense_device(self, bases_t, quals_t, ref_t, min_af, results_t=None):
        """bases_t/quals_t: int8 [n_sites, row_stride]-strided CUDA tensors; results_t: uint8 [n_sites*120]."""
        import torch
        ns, n = bases_t.shape
        stride = bases_t.stride(0)
        assert bases_t.stride(1) == 1 and quals_t.stride(1) == 1 and quals_t.stride(0) == stride
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        self._check(self._L.bvc_lrt_dense(self._h, ns, n, stride, _dev_ptr(bases_t), _dev_ptr(quals_t),
                                          _dev_ptr(ref_t), float(min_af), _dev_ptr(results_t), BVC_PTR_DEVICE))
        return results_t

    def lrt_dense_groups_device(self, bases_t, quals_t, ref_t, min_af, group_t, n_groups, results_t=None,
                                grp_results_t=None):
        """Group mode on device tensors; group_t: uint8 [n_samples].  Returns (results_t, grp_results_t)."""
        import torch
        ns, n = bases_t.shape
        stride = bases_t.stride(0)
        assert bases_t.stride(1) == 1 and quals_t.stride(1) == 1 and quals_t.stride(0) == stride
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        if grp_results_t is None:
            grp_results_t = torch.empty(ns * n_groups * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=bases_t.device)
        self._check(self._L.bvc_lrt_dense_groups(self._h, ns, n, stride, _dev_ptr(bases_t), _dev_ptr(quals_t),
                                                 _dev_ptr(ref_t), float(min_af), _dev_ptr(group_t), int(n_groups),
                                                 _dev_ptr(results_t), _dev_ptr(grp_results_t), BVC_PTR_DEVICE))
        return results_t, grp_results_t

    def lrt_hist_device(self, counts_t, ref_t, min_af, results_t=None):
        """Stage 2 alone on device tensors: counts_t int32/uint32 [n_sites, 512]; asynchronous on the stream."""
        import torch
        ns = counts_t.shape[0]
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=counts_t.device)
        self._check(self._L.bvc_lrt_hist(self._h, ns, _dev_ptr(counts_t), _dev_ptr(ref_t), float(min_af), None, None,
                                         _dev_ptr(results_t), BVC_PTR_DEVICE))
        return results_t

    def hist_dense_device(self, bases_t, quals_t, counts_t=None):
        import torch
        ns, n = bases_t.shape
        if counts_t is None:
            counts_t = torch.empty((ns, NCLASS), dtype=torch.int32, device=bases_t.device)
        self._check(self._L.bvc_hist_dense(self._h, ns, n, bases_t.stride(0), _dev_ptr(bases_t), _dev_ptr(quals_t),
                                           _dev_ptr(counts_t), BVC_PTR_DEVICE))
        return counts_t

    # ---- packed tiles: one byte per sample (base << 6 | qual, qual <= 62; 0xFF = no observation) ----
    def pack_dense_device(self, bases_t, quals_t, packed_t=None):
        """Two-byte device tile -> packed device tile.  Returns (packed_t, n_unrepresentable)."""
        import torch
        ns, n = bases_t.shape
        assert bases_t.stride(1) == 1 and quals_t.stride(1) == 1 and quals_t.stride(0) == bases_t.stride(0)
        if packed_t is None:
            stride = (n + 127) // 128 * 128
            packed_t = torch.empty((ns, stride), dtype=torch.uint8, device=bases_t.device)[:, :n]
        bad = C.c_int64(0)
        self._check(self._L.bvc_pack_dense(self._h, ns, n, bases_t.stride(0), _dev_ptr(bases_t), _dev_ptr(quals_t),
                                           packed_t.stride(0), _dev_ptr(packed_t), C.byref(bad), BVC_PTR_DEVICE))
        return packed_t, int(bad.value)

    def lrt_dense_packed_device(self, packed_t, ref_t, min_af, results_t=None):
        import torch
        ns, n = packed_t.shape
        assert packed_t.stride(1) == 1
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=packed_t.device)
        self._check(self._L.bvc_lrt_dense_packed(self._h, ns, n, packed_t.stride(0), _dev_ptr(packed_t), _dev_ptr(ref_t),
                                                 float(min_af), _dev_ptr(results_t), BVC_PTR_DEVICE))
        return results_t

    def lrt_dense_groups_packed_device(self, packed_t, ref_t, min_af, group_t, n_groups, results_t=None, grp_results_t=None):
        import torch
        ns, n = packed_t.shape
        assert packed_t.stride(1) == 1
        if results_t is None:
            results_t = torch.empty(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device=packed_t.device)
        if grp_results_t is None:
            grp_results_t = torch.empty(ns * n_groups * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=packed_t.device)
        self._check(self._L.bvc_lrt_dense_groups_packed(self._h, ns, n, packed_t.stride(0), _dev_ptr(packed_t), _dev_ptr(ref_t),
                                                        float(min_af), _dev_ptr(group_t), int(n_groups),
                                                        _dev_ptr(results_t), _dev_ptr(grp_results_t), BVC_PTR_DEVICE))
        return results_t, grp_results_t

    def lrt_dense_groups_packed(self, packed, ref_base, min_af, group_of_sample, n_groups):
        p = np.ascontiguousarray(packed, dtype=np.uint8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        g = np.ascontiguousarray(group_of_sample, dtype=np.uint8)
        out = np.zeros(p.shape[0], dtype=SITE_DTYPE)
        gout = np.zeros((p.shape[0], n_groups), dtype=GROUP_DTYPE)
        self._check(self._L.bvc_lrt_dense_groups_packed(self._h, p.shape[0], p.shape[1], p.shape[1], _np_ptr(p), _np_ptr(r),
                                                        float(min_af), _np_ptr(g), int(n_groups), _np_ptr(out), _np_ptr(gout),
                                                        BVC_PTR_HOST))
        return out, gout

    def lrt_dense_packed(self, packed, ref_base, min_af):
        """Host (numpy) packed tile [n_sites, n_samples] uint8."""
        p = np.ascontiguousarray(packed, dtype=np.uint8)
        r = np.ascontiguousarray(ref_base, dtype=np.int8)
        ns, n = p.shape
        out = np.zeros(ns, dtype=SITE_DTYPE)
        self._check(self._L.bvc_lrt_dense_packed(self._h, ns, n, n, _np_ptr(p), _np_ptr(r), float(min_af), _np_ptr(out),
                                                 BVC_PTR_HOST))
        return out

    def hist_dense_packed_device(self, packed_t, counts_t=None):
        import torch
        ns, n = packed_t.shape
        if counts_t is None:
            counts_t = torch.empty((ns, NCLASS), dtype=torch.int32, device=packed_t.device)
        self._check(self._L.bvc_hist_dense_packed(self._h, ns, n, packed_t.stride(0), _dev_ptr(packed_t),
                                                  _dev_ptr(counts_t), BVC_PTR_DEVICE))
        return counts_t

    def synth_dense_device(self, seed, site0, bases_t, quals_t, ref_t, cov_thr16=65536):
        ns, n = bases_t.shape
        self._check(self._L.bvc_synth_dense(self._h, int(seed), int(site0), ns, n, bases_t.stride(0), int(cov_thr16),
                                            _dev_ptr(bases_t), _dev_ptr(quals_t), _dev_ptr(ref_t)))


def _stream_read_gbs(self, tensor, repeats=5):
    """Empirical HBM read bandwidth (GB/s): a plain 16 B/lane streaming read of `tensor` (device, contiguous)."""
    ms = C.c_double()
    nbytes = tensor.numel() * tensor.element_size()
    self._check(self._L.bvc_stream_read_ms(self._h, _dev_ptr(tensor), nbytes, int(repeats), C.byref(ms)))
    return nbytes / (ms.value * 1e-3) / 1e9


Context.stream_read_gbs = _stream_read_gbs


def results_from_tensor(results_t):
    """uint8 CUDA/CPU tensor of packed bvc_site_result -> numpy structured array (synchronises)."""
    return results_t.cpu().numpy().view(SITE_DTYPE)
