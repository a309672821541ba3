#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/fc_prof
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fc_256 -- python3 $GRAFT_REPO_ROOT/tools/forecaster_bench.py 256 256 4 4 > $out/fc_256.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/fc_256/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"][:100].replace("(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
tail -1 $out/fc_256.log | cut -c1-600
