#!/bin/bash
# Build a VARIANT of librssf.so for an A/B of one translation unit:  tools/ab_lib.sh <name> <file.hip> <extra hipcc flags...>
# -> representationlearning_amd/lib/ab/librssf_<name>.so (select it with RSSF_LIB_OVERRIDE=<path>; the shipping library is untouched)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../representationlearning_amd/csrc"
mkdir -p build/ab ../lib/ab
obj=build/ab/${src%.hip}_$name.o
extra=""; [ "$src" = "gate.hip" ] && extra="-fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast $extra "$@" -c $src -o $obj
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/librssf_$name.so $objs $obj -ldl
echo ../lib/ab/librssf_$name.so
