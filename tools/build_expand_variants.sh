#!/bin/bash
# Builds libbmq.so once per k_expand variant (defaults of bmq_expand_kernel.h rewritten, so that the source that ships is the source that ran):
#   build/variants/<name>/bmq_expand_kernel.h + libbmq.so     usage: tools/build_expand_variants.sh name:K:LONG:MIN_WAVES ...
set -e
cd "$(dirname "$0")/.."
H=bifromq_amd/csrc/bmq_expand_kernel.h
cp $H /tmp/bmq_expand_kernel.h.orig
trap 'cp /tmp/bmq_expand_kernel.h.orig '$H EXIT
for spec in "$@"; do
  IFS=: read name K L W <<< "$spec"
  mkdir -p build/variants/$name
  python - "$K" "$L" "$W" <<'PY'
import re,sys
K,L,W=sys.argv[1:4]
p='bifromq_amd/csrc/bmq_expand_kernel.h'
s=open('/tmp/bmq_expand_kernel.h.orig').read()
for macro,val in (('BMQ_EXP_K',K),('BMQ_EXP_LONG',L),('BMQ_EXP_MIN_WAVES',W)):
    s,n=re.subn(r'(#define %s )\d+'%macro, r'\g<1>%s'%val, s, count=1)
    assert n==1, macro
open(p,'w').write(s)
PY
  cp $H build/variants/$name/bmq_expand_kernel.h
  ( cd bifromq_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c -o ../../build/variants/$name/engine.o bmq_engine.hip 2> ../../build/variants/$name/ru.txt \
    && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$name/libbmq.so ../../build/variants/$name/engine.o bmq_codec.o bmq_retain.o bmq_router.o bmq_cache.o -ldl && rm ../../build/variants/$name/engine.o )
  echo "$name: $(grep -A12 'Function Name: _ZN3bmq8k_expandE' build/variants/$name/ru.txt | sed 's/.*remark: //; s/ \[-R.*//' | grep -E 'VGPRs:|Scratch|Occupancy|LDS' | tr -s ' ' | tr '\n' ';')"
done
