#!/bin/bash
# Builds a variant of libbmq.so with extra -D switches into build/variants/libbmq_<name>.so (experiments only; select it with BMQ_LIB).
#   tools/build_variant.sh mw6 -DBMQ_WALK_MIN_WAVES=6 -DBMQ_FAST_LEVELS=8
set -e
name=$1; shift
cd "$(dirname "$0")/../bifromq_amd/csrc"
mkdir -p ../../build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c -o ../../build/variants/engine_$name.o bmq_engine.hip 2> ../../build/variants/ru_$name.txt
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libbmq_$name.so ../../build/variants/engine_$name.o bmq_codec.o bmq_retain.o bmq_router.o bmq_cache.o -ldl
grep -A10 "Function Name: _ZN3bmq6k_walkILi512ELi176ELi152ELb0" ../../build/variants/ru_$name.txt | grep -E "SGPRs|VGPRs|Scratch|Occupancy" | sed 's/.*remark: //; s/ \[-R.*//' | tr '\n' ' '; echo
