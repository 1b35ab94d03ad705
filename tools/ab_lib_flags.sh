#!/bin/bash
# like tools/ab_lib.sh, but REPLACING the optimisation flags of the one translation unit:  tools/ab_lib_flags.sh <name> <file.hip> <all hipcc flags after the arch...>
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../representationlearning_amd/csrc"
mkdir -p build/ab ../lib/ab
obj=build/ab/${src%.hip}_$name.o
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result "$@" -c $src -o $obj
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/librssf_$name.so $objs $obj -ldl
echo ../lib/ab/librssf_$name.so
