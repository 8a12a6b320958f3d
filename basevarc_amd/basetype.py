"""Host-side mirror of the reference's BaseType interface (src/BaseType.h:59-74) over libbvc.

`BaseType(bases, quals, ref, min_af)`, `SetBase(v)`, `LRT()` and the public fields `var_qual`,
`depth_total`, `alt_bases`, `depth`, `af_lrt` have the reference's names and meaning, so a test written
against the reference class reads the same here.  One object = one site = one library call, which is the
slow way to use a GPU: a `basevarc_amd.Context` takes a whole tile of sites per call (`lrt_dense`, `lrt_csr`,
`lrt_dense_groups`), which is the form bt_f's successor uses.
The C++ counterpart for the reference's own callers is include/bvc_basetype.hpp.
"""
import numpy as np

from .lib import Context, NCLASS

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def caller_min_af(n_samples_total, maf=0.001):
    """min_af as bt_f derives it from the total sample count N and --maf (src/BaseVarC.cpp:541-543)."""
    m = 100.0 / n_samples_total
    if m > 0.001:
        m = 0.001
    if maf < m:
        m = maf
    return m


class BaseType:
    def __init__(self, bases, quals, ref, min_af, ctx=None):
        self._bases = np.ascontiguousarray(bases, dtype=np.int8)
        self._quals = np.ascontiguousarray(quals, dtype=np.int8)
        if self._bases.shape != self._quals.shape or self._bases.ndim != 1:
            raise ValueError("bases and quals must be 1-D and of equal length")
        self._ref = int(ref)
        self._min_af = float(min_af)
        self._comb = [0, 1, 2, 3]                 # base_comb default, src/BaseType.h:79
        self._ctx = ctx or default_context()
        self._done = False
        self.var_qual = 0.0
        self.depth_total = 0.0
        self.alt_bases = []
        self.depth = {0: 0, 1: 0, 2: 0, 3: 0}
        self.af_lrt = {}
        self.record = None

    def SetBase(self, v):
        v = [int(b) for b in v]
        if len(v) > 4 or any(b < 0 or b > 3 for b in v):
            raise ValueError("SetBase takes at most four bases in 0..3")
        self._comb = v

    def LRT(self):
        if self._done:
            raise RuntimeError("LRT() is single-shot, as in the reference (it consumes the per-sample vectors)")
        self._done = True
        comb = np.zeros((1, 4), dtype=np.int8)
        comb[0, :len(self._comb)] = self._comb
        rec = self._ctx.lrt_csr([0, len(self._bases)], self._bases, self._quals, [self._ref], self._min_af,
                                base_comb=comb, n_comb=[len(self._comb)])[0]
        self.record = rec
        self.var_qual = float(rec["var_qual"])
        self.depth_total = float(rec["depth_total"])
        self.depth = {j: int(rec["depth"][j]) for j in range(4)}
        self.alt_bases = [int(rec["alt_base"][i]) for i in range(rec["n_alt"])]
        self.af_lrt = {int(rec["alt_base"][i]): float(rec["af"][i]) for i in range(rec["n_alt"])}
        return bool(rec["called"])


__all__ = ["BaseType", "caller_min_af", "default_context", "NCLASS"]
