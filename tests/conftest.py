import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _lds_violations_fail_the_test(request):
    """Diagnostic builds of libbvc (-DBVC_CHECK_LDS, tools/poison_run.sh) record every data-derived LDS address or index that
    falls outside its bounds instead of faulting; a GPU test that leaves such a record fails, with the record in the message.
    The product library does not export bvc_debug_report and this fixture does nothing."""
    yield
    if "gpu" not in request.keywords or not _has_gpu():
        return
    import ctypes as C
    from basevarc_amd import lib as bl
    L = bl.load_library()
    if not hasattr(L, "bvc_debug_report"):
        return
    out = (C.c_uint32 * 24)()
    L.bvc_debug_report.restype = C.c_int
    L.bvc_debug_report.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    rc = L.bvc_debug_report(None, out, 1)
    assert rc == 0, f"bvc_debug_report failed: {rc}"
    rec = {tu: list(out[8 * i:8 * i + 6]) for i, tu in enumerate(("hist_kernel", "em_kernel", "em_items"))}
    bad = {tu: r for tu, r in rec.items() if r[0]}
    assert not bad, f"LDS bound violations [count, check id, value, limit, blockIdx.x, threadIdx.x]: {bad}"
