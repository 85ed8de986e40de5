#!/bin/bash
# lab: build variants of libtessphot_hip.so with extra -D flags (scratch, not product)
set -e
cd "$(dirname "$0")/../../photometry_amd/csrc"
rm -f ../../tools/lab/lib_*.so
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $flags -x hip -c background.hip -o /tmp/bkg_$name.o
  objs=$(ls build/*.o | grep -v background)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/lib_$name.so $objs /tmp/bkg_$name.o -L/opt/rocm/lib -lrccl
done
ls ../../tools/lab/
