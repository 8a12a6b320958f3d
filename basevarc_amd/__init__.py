"""basevarc_amd -- MI355X (gfx950) implementation of BaseVarC's per-site basetype hot path.

The product is libbvc.so (HIP kernels behind the C ABI of include/bvc.h).  This package is the thin
Python host side: a ctypes binding (`lib`) and a mirror of the reference's BaseType interface
(`basetype`).  There is no CPU implementation in here: without the built library and a gfx950 device
every compute call raises.
"""
from .lib import BvcError, Context, SiteResult, GroupResult, load_library, library_path  # noqa: F401
from .basetype import BaseType, caller_min_af  # noqa: F401

__all__ = ["BvcError", "Context", "SiteResult", "GroupResult", "load_library", "library_path",
           "BaseType", "caller_min_af"]
