"""Builds basevarc_amd/libbvc.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbvc.so")
SOURCES = ["bvc_api.hip", "hist_kernel.hip", "em_kernel.hip", "em_items.hip", "synth_kernel.hip"]
DEPS = ["bvc_device.h", "bvc_internal.h", "synth_tables.inc", os.path.join("..", "..", "include", "bvc.h")]


def code_sha16(path=None):
    """sha256 (16 hex digits) of a kernel source with its // comments and blank space removed: the stamp that ties a
    committed counter pass (profiles/pmc_traffic.json) to the code it was taken on, indifferent to comment edits."""
    import hashlib
    import re
    path = path or os.path.join(CSRC, "hist_kernel.hip")
    text = open(path, encoding="utf-8").read()
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"\s+", " ", text)
    return hashlib.sha256(text.encode()).hexdigest()[:16]


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


HOST = os.path.join(HERE, "host")
HOST_SOURCES = ["stats.cpp", "pileup.cpp", "bgzf.cpp", "inflate.cpp", "bam.cpp"]
HOST_LIB = os.path.join(HERE, "libbvchost.so")
HOST_EXE = os.path.join(HERE, "BaseVarC")


def build_host(force=False):
    """g++ build of the host side (the reference's `BaseVarC basetype` command line and its text formats):
    basevarc_amd/BaseVarC (links libbvc.so) and basevarc_amd/libbvchost.so (test hooks, no GPU code)."""
    srcs = [os.path.join(HOST, f) for f in HOST_SOURCES + ["main.cpp", "capi.cpp"]]
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    newest = max(os.path.getmtime(f) for f in srcs + hdrs)
    inc = ["-I", os.path.join(HERE, "..", "include")]
    flags = ["-std=c++11", "-O2", "-Wall", "-Wextra", "-fPIC", "-pthread"]
    if force or not os.path.exists(HOST_LIB) or os.path.getmtime(HOST_LIB) < newest:
        subprocess.check_call(["g++"] + flags + inc + ["-shared", "-o", HOST_LIB] +
                              [os.path.join(HOST, f) for f in HOST_SOURCES + ["capi.cpp"]] + ["-lz"])
    build()
    if force or not os.path.exists(HOST_EXE) or os.path.getmtime(HOST_EXE) < max(newest, os.path.getmtime(LIB)):
        subprocess.check_call(["g++"] + flags + inc + ["-o", HOST_EXE] +
                              [os.path.join(HOST, f) for f in HOST_SOURCES + ["main.cpp"]] +
                              ["-L", HERE, "-lbvc", "-lz", "-Wl,-rpath,$ORIGIN"])
    return HOST_EXE, HOST_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True))
