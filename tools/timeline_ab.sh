cd /tmp && export TMPDIR=/tmp
for cfg in "emip_amd.lib.pvt_v2:MLP_BAND=False" "emip_amd.ops:MLP_BAND_BANDS=8" "emip_amd.ops:MLP_BAND_BANDS=4"; do
  tag=$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  EMIP_DBG="$cfg" timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $GRAFT_REPO_ROOT/tools/pipe_trace.py 1 > /tmp/tl_$tag.log 2>&1 || exit 1
  f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
  echo "== $cfg" >> $GRAFT_REPO_ROOT/gpurun_out/r4_timeline.log
  python3 $GRAFT_REPO_ROOT/tools/pipe_trace_timeline.py $f >> $GRAFT_REPO_ROOT/gpurun_out/r4_timeline.log 2>&1
done
