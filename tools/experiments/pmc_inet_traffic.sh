#!/bin/bash
# usage (on the GPU box): bash tools/experiments/pmc_inet_traffic.sh <F>   -- HBM-side bytes of the K6 edge kernel
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_inet_traffic_$1
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 $GRAFT_REPO_ROOT/tools/experiments/inet_one.py $1 edge > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 $GRAFT_REPO_ROOT/tools/experiments/inet_one.py $1 edge > $out/w.log 2>&1
python3 - <<PY
import csv, glob
res = {}
for sub, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    vals = []
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp2" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    res[name] = sum(vals) / max(len(vals), 1)
hbm = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
print(f"F=$1 k_mlp2 edge: FETCH_SIZE {res['FETCH_SIZE']:.0f} KiB (raw), WRITE_SIZE {res['WRITE_SIZE']:.0f} KiB -> HBM-side bytes per launch (2*FETCH+WRITE)*1024 = {hbm/1e6:.1f} MB")
PY
