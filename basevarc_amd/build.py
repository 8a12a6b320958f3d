"""Builds basevarc_amd/libbvc.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbvc.so")
SOURCES = ["bvc_api.hip", "hist_kernel.hip", "em_kernel.hip", "synth_kernel.hip"]
DEPS = ["bvc_device.h", "bvc_internal.h", "synth_tables.inc", os.path.join("..", "..", "include", "bvc.h")]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the path of the library."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wextra", "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    cmd[1:1] = os.environ.get("BVC_EXTRA_FLAGS", "").split()      # experiments: -DBVC_HIST_UNROLL=8 ...
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
