#!/bin/bash
# rocprofv3 kernel stats of a GCN training step:  tools/experiments/train_prof.sh [mesh|ref] [more train_bench.py args]
which=${1:-mesh}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/train_prof_$which
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py $which "$@" > $out/t.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/t/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:28]:
        print(r["Name"][:95].replace("(anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
grep "training step" $out/t.log
