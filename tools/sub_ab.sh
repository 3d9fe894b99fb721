#!/bin/bash
# full default bench (sub-records included) under environment settings: the sub-record values
out=gpurun_out/sub_ab.log; : > $out
for cfg in "$@"; do
env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', d['value'], 'train', d['train'].get('ms_per_step'), 'long', d['long'].get('value'), d['long'].get('ms_per_step'))" >> $out
done; cat $out
