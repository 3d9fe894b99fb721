#!/bin/bash
# Run ON THE GPU BOX (gpurun -- tools/collect_profiles.sh r03): the rocprofv3 evidence bench.py and DESIGN.md cite.
#   gpurun_out/<tag>_bench_kernel_stats.csv   --kernel-trace --stats of the default inference workload
#   gpurun_out/<tag>_train_kernel_stats.csv   ... of the training step
#   gpurun_out/pmc_traffic.json               two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains)
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof/bench -- python3 $root/bench.py --no-cpu-baseline --no-sub > $out/prof_bench.log 2>&1
f=$(find $out/prof/bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_bench_kernel_stats.csv
echo "bench stats: $f"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof/train -- python3 $root/bench.py --workload train --train-eager --steps 17 --warmup 3 > $out/prof_train.log 2>&1
f=$(find $out/prof/train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_train_kernel_stats.csv
echo "train stats: $f"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/prof/fetch -- python3 $root/bench.py --no-graph --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-sub > $out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/prof/write -- python3 $root/bench.py --no-graph --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-sub > $out/prof_write.log 2>&1
ff=$(find $out/prof/fetch -name "*counter_collection.csv" | head -1)
fw=$(find $out/prof/write -name "*counter_collection.csv" | head -1)
echo "pmc: $ff $fw"
# forwards profiled: 3 steps (2 timed + 1 warm-up) + 1 eager measurement pass of bench.py = 4
[ -n "$ff" ] && [ -n "$fw" ] && python3 $root/tools/pmc_traffic.py $ff $fw $out/pmc_traffic.json 4
rm -rf $out/prof
ls -la $out | tail -8
