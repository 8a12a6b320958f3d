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


# Flags of single translation units.  em_items.hip: the narrow region kernel sits at the 168 VGPRs of three wavefronts per
# SIMD, and the machine-level loop-invariant code motion of this compiler keeps hoisting lane-constant addresses, zero vectors
# and the like out of the region's level loop, where they stay live across the fits and end in scratch (or in 30-70 SGPR
# spills); without that pass the kernel has no spill of either kind (tools/isa_report.py, tests/test_isa.py).
PER_SOURCE_FLAGS = {"em_items.hip": ["-mllvm", "-disable-machine-licm"],
                    # the one-wavefront-per-site kernels: 46-64 SGPR spills (to VGPR lanes) with the pass, 15-38 without, fewer VGPRs
                    "em_kernel.hip": ["-mllvm", "-disable-machine-licm"]}


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the path of the library."""
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("BVC_EXTRA_FLAGS", "").split()         # experiments: -DBVC_HIST_UNROLL=8 ...
    common = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"] + extra
    if verbose:
        common.append("-Rpass-analysis=kernel-resource-usage")
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for f in SOURCES:                                            # one compile per source, side by side
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        procs.append((obj, subprocess.Popen(common + PER_SOURCE_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", obj])))
    objs = []
    for obj, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, "hipcc -c " + obj)
        objs.append(obj)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
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
