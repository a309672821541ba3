cd /tmp; export TMPDIR=/tmp
i=0
for cfg in "4 1" "8 1"; do
  set -- $cfg; export GWEN_K4_WAVES=$1 GWEN_K4_MINW=$2
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
    i=$((i+1)); d=$GRAFT_REPO_ROOT/gpurun_out/pmc6_$i
    echo "pass $i: W=$1 M=$2 :: $c" >> $GRAFT_REPO_ROOT/gpurun_out/exp6.log
    timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/kbench.py all 64 > $d.log 2>&1 || tail -3 $d.log >> $GRAFT_REPO_ROOT/gpurun_out/exp6.log
  done
done
