#!/bin/bash
# usage (on the GPU box): bash tools/experiments/pmc_wide.sh <tag> [nu F M]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wide_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R="python3 $GRAFT_REPO_ROOT/tools/experiments/wide_one.py $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM --output-format csv -d $out/a -- $R > $out/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/b -- $R > $out/b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $out/c -- $R > $out/c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- $R > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/w -- $R > $out/w.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- $R > $out/t.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b", "c", "f", "w"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_wide" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} {sum(v)/len(v):18.0f}  (n={len(v)})")
for f in glob.glob("$out/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_wide" in r["Name"]:
            print("kernel avg ns", r["AverageNs"], "calls", r["Calls"])
PY
tail -2 $out/a.log $out/c.log | cut -c1-300
