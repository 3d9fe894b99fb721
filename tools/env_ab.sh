#!/bin/bash
# In-call comparison of environment settings:  tools/env_ab.sh <tag> "VAR=a VAR2=b" "VAR=c" ...  (each run twice, interleaved)
tag=$1; shift
out=gpurun_out/envab_$tag.log
: > $out
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-sub 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', 'rep$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms')" >> $out
  done
done
cat $out
