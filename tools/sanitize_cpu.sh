#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code (the oracle's C restatement and the host library):
# builds both instrumented in a scratch clone and runs their CPU tests under them.  GPU sanitizers are not
# available on this pool; this is the CPU build only.  usage: bash tools/sanitize_cpu.sh
set -e
SRC=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/bvc_sanitize
rm -rf "$W" && mkdir -p "$W" && git clone -q "$SRC" "$W/repo" && cd "$W/repo"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
gcc $SAN -std=c99 -fPIC -fopenmp -ffp-contract=off -shared -o oracle/liborc.so oracle/basetype_oracle.c -lm
g++ $SAN -std=c++11 -fPIC -pthread -Iinclude -shared -o basevarc_amd/libbvchost.so \
    basevarc_amd/host/stats.cpp basevarc_amd/host/pileup.cpp basevarc_amd/host/bgzf.cpp basevarc_amd/host/inflate.cpp basevarc_amd/host/bam.cpp \
    basevarc_amd/host/capi.cpp -lz
# the GPU library is not under test here: reuse the tree's; the host executable is instrumented too (its --load
# phase -- BGZF, BAM records, index seek, pileup, temp-batch writer -- runs without a GPU)
cp "$SRC/basevarc_amd/libbvc.so" basevarc_amd/
g++ $SAN -std=c++11 -pthread -Iinclude -o basevarc_amd/BaseVarC basevarc_amd/host/stats.cpp basevarc_amd/host/pileup.cpp \
    basevarc_amd/host/bgzf.cpp basevarc_amd/host/inflate.cpp basevarc_amd/host/bam.cpp basevarc_amd/host/capi.cpp basevarc_amd/host/main.cpp \
    -L basevarc_amd -lbvc -lz -Wl,-rpath,'$ORIGIN'
touch basevarc_amd/libbvc.so basevarc_amd/libbvchost.so basevarc_amd/BaseVarC oracle/liborc.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_oracle.py tests/test_prune_bound.py tests/test_host.py -x -q -m "not gpu"
