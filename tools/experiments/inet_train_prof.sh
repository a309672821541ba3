#!/bin/bash
# rocprofv3 kernel stats of the InteractionNet forecaster's training step:  tools/experiments/inet_train_prof.sh [hidden]
H=${1:-64}
out=$GRAFT_REPO_ROOT/gpurun_out/inet_train_prof_$H
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
export INET_SKIP_OLD=1       # only the product's step: no torch-recompute comparison, no graph capture in the trace
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $GRAFT_REPO_ROOT/tools/inet_train_bench.py $H > $out/t.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/t/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:40]:
        print(r["Name"][:110].replace("(anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
tail -2 $out/t.log
