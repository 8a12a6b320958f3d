#!/bin/bash
# libbvc variants that differ in inflate_kernel.hip's ring size: basevarc_amd/_variants/libbvc_win<bytes>.so   (experiments; needs a built libbvc.so)
set -e
cd "$(dirname "$0")/.."
V=basevarc_amd/_variants; mkdir -p $V
O=basevarc_amd/csrc/_obj
for w in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DBVC_INFLATE_WINDOW=$w -c basevarc_amd/csrc/inflate_kernel.hip -o $V/inflate_$w.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libbvc_win$w.so $(ls $O/*.o | grep -v inflate_kernel) $V/inflate_$w.o \
    && rm $V/inflate_$w.o && echo built win$w ) &
done
wait
