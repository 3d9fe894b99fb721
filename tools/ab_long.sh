#!/bin/bash
# In-call A/B of two library builds on EMIP-long: tools/bin/lib_base.so against emip_amd/libemip_hip.so
for rep in 1 2; do for lib in tools/bin/lib_base.so emip_amd/libemip_hip.so; do
EMIP_HIP_LIB=$PWD/$lib python bench.py --workload long 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], d['ms_per_step'], 'ms')"
done; done
