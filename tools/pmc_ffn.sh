cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_ffn -- python3 $GRAFT_REPO_ROOT/tools/ffn_block_bench.py > /tmp/pmc_ffn.log 2>&1
f=$(find /tmp/pmc_ffn -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == "WRITE_SIZE":
        a = agg[r["Kernel_Name"][:60]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    if "ffn" in k: print(k, n, "launches, WRITE_SIZE %.1f MB per launch" % (v / n * 1024 / 1e6))
PY
