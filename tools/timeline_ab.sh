#!/bin/bash
# on the GPU box: one-step-at-a-time kernel trace of the default configuration -> gpurun_out/r4_timeline.log
cd /tmp && export TMPDIR=/tmp
for cfg in "${@:-none:X=0}"; do
  tag=$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  c=$cfg; [ "$cfg" = "none:X=0" ] && c=""
  EMIP_DBG="$c" timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $GRAFT_REPO_ROOT/tools/pipe_trace.py 1 > /tmp/tl_$tag.log 2>&1 || exit 1
  f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
  echo "== $cfg" >> $GRAFT_REPO_ROOT/gpurun_out/r4_timeline.log
  python3 $GRAFT_REPO_ROOT/tools/pipe_trace_timeline.py $f 60 >> $GRAFT_REPO_ROOT/gpurun_out/r4_timeline.log 2>&1
done
