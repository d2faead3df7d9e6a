#!/bin/bash
# A differently compiled libamgcore_hip.so for same-box A/Bs (AMGCORE_HIP_LIB=tools/_bin/libamg_NAME.so):
#   tools/build_variant.sh NAME "file1.hip file2.hip" "-DFOO=1 -DBAR"
# recompiles the named sources with the extra flags and links them with the regular objects of the others.
set -e
cd "$(dirname "$0")/../pyamg_amd/csrc"
NAME=$1; FILES=$2; EXTRA=$3
make -s
mkdir -p ../../tools/_bin/_obj_$NAME
OBJS=""
for f in kernels hier capi ne devapi schwarz comm spgemm sell gsflow; do
  if [[ " $FILES " == *" $f.hip "* ]]; then
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value $EXTRA -c $f.hip -o ../../tools/_bin/_obj_$NAME/$f.o
    OBJS="$OBJS ../../tools/_bin/_obj_$NAME/$f.o"
  else
    OBJS="$OBJS _obj/$f.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o ../../tools/_bin/libamg_$NAME.so
echo tools/_bin/libamg_$NAME.so
