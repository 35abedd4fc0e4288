#!/bin/bash
# Developer helper: an A/B variant of libarmnet_hip.so that differs from the product build in ONE translation unit.
#   tools/scratch/variant_lib.sh NAME "-DFLAG ..." file.hip   ->  arm-net_amd/lib/exp/libarmnet_NAME.so
# (compiles only that file with the extra flags and links it against the product objects of arm-net_amd/lib/obj)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2; SRC=$3
OBJ=$ROOT/arm-net_amd/lib/obj; OUT=$ROOT/arm-net_amd/lib/exp
mkdir -p $OUT/obj_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wall -Wno-unused-function $FLAGS \
    -c $ROOT/arm-net_amd/csrc/$SRC -o $OUT/obj_$NAME/${SRC%.hip}.o
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/obj_$NAME/${SRC%.hip}.o -o $OUT/libarmnet_$NAME.so
echo $OUT/libarmnet_$NAME.so
