"""What the compiler emitted for the kernels of libbvc (CPU container: hipcc cross-compiles gfx950 without a GPU).

The hot path must not address memory through the FLAT aperture or through scratch: every LDS access is a ds_*
instruction on an LDS-typed address, every global access a global_* instruction, no kernel has a private segment and
no VGPR is spilled.  (Round 3 shipped a `volatile uint32_t *` into LDS that compiled to flat_store_dword /
flat_load_dword in hist_dense_groups_slots_kernel: this test is what would have caught it.)
SGPR spills go to VGPR lanes, not to memory; their list is recorded in profiles/r04_isa_resources.txt.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_report  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    return isa_report.report()


def test_every_kernel_source_yields_kernels(kernels):
    by_src = {}
    for k in kernels:
        by_src.setdefault(k["source"], []).append(k)
    assert set(by_src) == set(isa_report.DEVICE_SOURCES)
    # the parse found the code of every kernel it found metadata for
    assert all(k["instructions"] > 10 for k in kernels), [k["pretty"] for k in kernels if k["instructions"] <= 10]
    names = {k["pretty"].split("<")[0] for k in kernels}
    for must in ("hist_dense_kernel", "hist_packed_kernel", "hist_dense_groups_slots_kernel", "hist_packed_groups_kernel",
                 "hist_csr_block_kernel", "hist_wave_kernel", "region_kernel",
                 "lrt_kernel", "var_qual_kernel", "synth_dense_kernel"):
        assert must in names, must


def test_no_flat_and_no_scratch_addressing(kernels):
    bad = [(k["pretty"], k["flat"], k["scratch"]) for k in kernels if k["flat"] or k["scratch"]]
    assert not bad, f"FLAT / scratch instructions on the hot path: {bad}"


def test_no_private_segment_and_no_vgpr_spill(kernels):
    bad = [(k["pretty"], k["private"], k["vgpr_spill"]) for k in kernels if k["private"] or k["vgpr_spill"]]
    assert not bad, f"private segment / VGPR spills: {bad}"


def test_no_matrix_instructions(kernels):
    # nothing on this path is a dense contraction (FP64 MFMA as an adder was measured and lost)
    assert sum(k["mfma"] for k in kernels) == 0


def test_diagnostic_builds_keep_the_same_addressing():
    # the poison / check build (tools/poison_run.sh) must exercise the same instruction classes as the product
    rows = isa_report.report(extra_flags=("-DBVC_POISON", "-DBVC_CHECK_LDS"))
    bad = [(k["pretty"], k["flat"], k["scratch"], k["private"]) for k in rows if k["flat"] or k["scratch"] or k["private"]]
    assert not bad, bad
