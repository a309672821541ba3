#!/bin/bash
# usage (on the GPU box): bash tools/experiments/pmc_inet.sh <F> <mode>
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_inet_$1_$2
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $out/a -- python3 $GRAFT_REPO_ROOT/tools/experiments/inet_one.py $1 $2 > $out/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python3 $GRAFT_REPO_ROOT/tools/experiments/inet_one.py $1 $2 > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_mlp2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
