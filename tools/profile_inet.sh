#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the InteractionNet block bench (K6), outputs under gpurun_out/$1
tag=${1:-inet}
F=${2:-64}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/inet_bench.py --channels $F > $out/bench.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    with open("$out/kernel_stats_short.csv", "w", newline="") as o:
        w = csv.writer(o)
        for r in rows:
            if r[0] == "Name" or "k_mlp2" in r[0] or "k_linear" in r[0] or "k_split_w" in r[0]:
                name = r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                w.writerow([name] + r[1:])
PY
cat $out/kernel_stats_short.csv; cat $out/bench.log
