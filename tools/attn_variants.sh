#!/bin/bash
# Build librssf variants that differ only in the window-attention forward's tuning macros:
#   tools/attn_variants.sh "OCC=2 STATS=1" "OCC=3 STATS=0" ...   ->  representationlearning_amd/lib/variants/librssf_<i>.so
set -e
cd "$(dirname "$0")/../representationlearning_amd/csrc"
make -j16 >/dev/null
mkdir -p ../lib/variants build/variants
i=0
for v in "$@"; do
  eval "$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast -DRSSF_FWD_OCC=$OCC -DRSSF_FWD_PREFETCH_STATS=$STATS \
        -c win_attn_fwd.hip -o build/variants/win_attn_fwd_$i.o
  objs=$(ls build/*.o | grep -v win_attn_fwd.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/librssf_$i.so $objs build/variants/win_attn_fwd_$i.o -ldl
  echo "variant $i: $v"
  i=$((i+1))
done
