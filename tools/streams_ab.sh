#!/bin/bash
out=gpurun_out/streams_ab.log; : > $out
for rep in 1 2; do for s in 1 2 4; do
python bench.py --no-cpu-baseline --no-sub --streams $s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $s rep$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms')" >> $out
done; done; cat $out
