#!/bin/bash
# usage (on the GPU box): bash tools/experiments/pmc_hops.sh <tag> [nu members]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_hops_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R="python3 $GRAFT_REPO_ROOT/tools/experiments/hops_time.py $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $out/a -- $R > $out/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/b -- $R > $out/b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/c -- $R > $out/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b", "c"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_narrow_chain" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v3 = v[5:55]            # the 3-stage launches (after 5 warm-ups)
            print(f"{k:32s} {sum(v3)/max(1,len(v3)):18.0f}  (n={len(v)})")
PY
tail -2 $out/a.log $out/c.log | cut -c1-300
