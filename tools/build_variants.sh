#!/bin/bash
# builds libbvc variants that differ in em_items.hip's macros: basevarc_amd/_variants/libbvc_<name>.so   (experiments)
set -e
cd "$(dirname "$0")/.."
V=basevarc_amd/_variants; mkdir -p $V
O=basevarc_amd/csrc/_obj
while [ $# -gt 0 ]; do
  name=${1%%:*}; flags=${1#*:}; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm $flags -c basevarc_amd/csrc/em_items.hip -o $V/em_items_$name.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libbvc_$name.so $O/bvc_api.o $O/hist_kernel.o $O/em_kernel.o $O/synth_kernel.o $V/em_items_$name.o \
    && rm $V/em_items_$name.o && echo built $name ) &
done
wait
