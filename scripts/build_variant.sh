#!/bin/bash
# Builds a variant of the library that differs in ONE translation unit:  scripts/build_variant.sh <name> <tu.hip> [-D...]
# -> scripts/_bin/libbbh_<name>.so (objects of the other units are taken from baybe_amd/csrc as they are).
set -e
NAME=$1; TU=$2; shift 2
cd "$(dirname "$0")/../baybe_amd/csrc"
mkdir -p ../../scripts/_bin /tmp/bbh_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -c "$TU" -o /tmp/bbh_variants/$NAME.o
OBJS=$(ls *.o | grep -v "^${TU%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/bbh_variants/$NAME.o -ldl -o ../../scripts/_bin/libbbh_$NAME.so
echo built scripts/_bin/libbbh_$NAME.so
