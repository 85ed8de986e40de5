#!/bin/bash
# lab: build variants of libtessphot_hip.so with extra -D flags (scratch, not product); SRC=<file>.hip picks the translation unit
set -e
cd "$(dirname "$0")/../../photometry_amd/csrc"
rm -f ../../tools/lab/lib_*.so
SRC=${SRC:-background.hip}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $flags -x hip -c $SRC -o /tmp/var_$name.o
  objs=$(ls build/*.o | grep -v "build/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/lib_$name.so $objs /tmp/var_$name.o -L/opt/rocm/lib -lrccl
done
ls ../../tools/lab/
