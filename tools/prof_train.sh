#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the training step alone -> gpurun_out/<tag>_train_kernel_stats.csv
tag=${1:-tmp}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof/train -- python3 $root/bench.py --workload train --steps 17 --warmup 3 > $out/prof_train.log 2>&1
f=$(find $out/prof/train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_train_kernel_stats.csv
echo "train stats: $f"
rm -rf $out/prof
