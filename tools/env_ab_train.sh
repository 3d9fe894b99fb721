#!/bin/bash
out=gpurun_out/envab_train.log; : > $out
for rep in 1 2; do for cfg in "$@"; do
env $cfg python bench.py --workload train --steps 12 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', 'rep$rep', d['value'], d['unit'], d['ms_per_step'], 'ms')" >> $out
done; done; cat $out
