#!/bin/bash
# Build a VARIANT of the library for a same-box A/B (scripts/ab_libs.sh): a copy of csrc with a patched chain_device.h, only the
# translation unit of the headline instantiations (chain_kernels_r2w8_m42.hip) recompiled, linked against the tree's other objects.
#   bash scripts/mk_variant.sh <name> <patched chain_device.h>   ->  deepctr_amd/lib/libdctr_hip_<name>.so
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; PATCHED=$2
V=/tmp/variant_$NAME
rm -rf $V && mkdir -p $V && cp $ROOT/deepctr_amd/csrc/*.h $ROOT/deepctr_amd/csrc/*.inc $ROOT/deepctr_amd/csrc/*.hip $ROOT/deepctr_amd/csrc/*.cpp $V/
cp $PATCHED $V/chain_device.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I $ROOT/include -I $V -x hip -c $V/chain_kernels_r2w8_m42.hip -o $V/m42.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|ScratchSize|VGPRs:" | sed -e 's/.*remark: [^ ]* *//' | paste -d' ' - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | sort | uniq -c
OBJS=$(ls $ROOT/deepctr_amd/csrc/_obj/*.o | grep -v "/chain_kernels_r2w8_m42.o")
g++ -shared -fPIC -o $ROOT/deepctr_amd/lib/libdctr_hip_$NAME.so $OBJS $V/m42.o -L/opt/rocm/lib -lamdhip64 -pthread
echo built $ROOT/deepctr_amd/lib/libdctr_hip_$NAME.so
