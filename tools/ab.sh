#!/bin/bash
# In-call A/B of two library builds (boxes differ by +-1.5 %): tools/bin/lib_base.so against emip_amd/libemip_hip.so.
#   tools/ab.sh <tag> [bench args]     -> gpurun_out/ab_<tag>.log
tag=$1; shift
out=gpurun_out/ab_$tag.log
: > $out
for rep in 1 2; do
  for lib in tools/bin/lib_base.so emip_amd/libemip_hip.so; do
    EMIP_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-sub "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'rep$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms')" >> $out
  done
done
cat $out
