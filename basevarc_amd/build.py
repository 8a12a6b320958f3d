"""Builds basevarc_amd/libbvc.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbvc.so")
SOURCES = ["bvc_api.hip", "hist_kernel.hip", "em_kernel.hip", "em_items.hip", "synth_kernel.hip", "pileup_kernel.hip", "inflate_kernel.hip"]
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


STAMP = LIB + ".flags"


def _flag_stamp():
    """What the library on disk must have been compiled with to count as the product: every flag that reaches hipcc, including the
    experiment / diagnostic ones of BVC_EXTRA_FLAGS (-DBVC_POISON, -DBVC_CHECK_LDS ...).  Kept beside the library."""
    extra = os.environ.get("BVC_EXTRA_FLAGS", "").split()
    return repr((COMMON_FLAGS, extra, sorted(PER_SOURCE_FLAGS.items())))


def needs_build():
    """True when the library is missing, older than a source, or was built with other flags than this process would use (so a
    diagnostic build left behind by a failed or killed tools/poison_run.sh is never taken for the product)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    if open(STAMP, encoding="utf-8").read() != _flag_stamp():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]
# Flags of single translation units.  em_items.hip: the narrow region kernel sits at the 168 VGPRs of three wavefronts per
# SIMD, and the machine-level loop-invariant code motion of this compiler keeps hoisting lane-constant addresses, zero vectors
# and the like out of the region's level loop, where they stay live across the fits and end in scratch (or in 30-70 SGPR
# spills); without that pass the kernel has no spill of either kind (tools/isa_report.py, tests/test_isa.py).
PER_SOURCE_FLAGS = {"em_items.hip": ["-mllvm", "-disable-machine-licm"],
                    # the one-wavefront-per-site kernels: 46-64 SGPR spills (to VGPR lanes) with the pass, 15-38 without, fewer VGPRs
                    "em_kernel.hip": ["-mllvm", "-disable-machine-licm"]}


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the path of the library.

    Objects and the linked library are made in a directory of this build's own and the library is moved into place in one
    rename, so two builds at once (two ranks, two test processes) cannot mix their objects, and a build that fails or is killed
    leaves the previous library untouched.  Every compiler started is waited for (or killed) before an error is raised."""
    if not force and not needs_build():
        return LIB
    import tempfile
    extra = os.environ.get("BVC_EXTRA_FLAGS", "").split()         # experiments: -DBVC_HIST_UNROLL=8 ...
    common = [_hipcc()] + COMMON_FLAGS + extra
    if verbose:
        common.append("-Rpass-analysis=kernel-resource-usage")
    os.makedirs(os.path.join(CSRC, "_obj"), exist_ok=True)
    objdir = tempfile.mkdtemp(prefix="build.", dir=os.path.join(CSRC, "_obj"))
    try:
        procs = []
        for f in SOURCES:                                            # one compile per source, side by side
            obj = os.path.join(objdir, f.replace(".hip", ".o"))
            # a session of its own: hipcc is a driver that starts compilers of its own, and a failed build ends them all
            procs.append((obj, subprocess.Popen(common + PER_SOURCE_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", obj],
                                                start_new_session=True)))
        failed = None
        for obj, pr in procs:
            if failed is not None and pr.poll() is None:
                try:
                    os.killpg(pr.pid, 9)
                except ProcessLookupError:
                    pass
            if pr.wait() != 0 and failed is None:
                failed = (pr.returncode, obj)
        if failed is not None:
            raise subprocess.CalledProcessError(failed[0], "hipcc -c " + os.path.basename(failed[1]))
        tmp_lib = os.path.join(objdir, "libbvc.so")
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + [o for o, _ in procs])
        if os.path.exists(STAMP):
            os.remove(STAMP)                                         # never a new library under an old stamp
        os.replace(tmp_lib, LIB)
        with open(STAMP + ".tmp", "w", encoding="utf-8") as f:
            f.write(_flag_stamp())
        os.replace(STAMP + ".tmp", STAMP)
        for o, _ in procs:                                           # the objects of the library in place (tools/build_variants.sh links against them)
            os.replace(o, os.path.join(CSRC, "_obj", os.path.basename(o)))
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
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
