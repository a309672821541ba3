#!/bin/bash
# SQ counter passes over bench.py's kernels (on the GPU box): bash tools/experiments/pmc_bench.sh
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-edge-mlp --event-stride 1000"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $out/a -- $B > $out/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- $B > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_(?:layer|chain|gather)<[^>]*>)", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v)/len(v):14.0f}")
PY
