set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/kbench.py all 16 32 64 128
GWEN_K2_REMAP=1 python tools/kbench.py k2 16 32 64 128
GWEN_K4_WAVES=8 python tools/kbench.py k4 16 32 64 128
GWEN_K4_WAVES=16 python tools/kbench.py k4 16 32 64
GWEN_K4_VIDX=1 python tools/kbench.py k4 16 32 64 128
GWEN_K4_WAVES=8 GWEN_K4_VIDX=1 python tools/kbench.py k4 16 32 64 128
KB_REORDER=none python tools/kbench.py all 64
KB_REORDER=none GWEN_K2_REMAP=1 GWEN_K4_WAVES=8 python tools/kbench.py all 64
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/kbench.py all 64 > $d.log 2>&1 || tail -5 $d.log
done
ls $GRAFT_REPO_ROOT/gpurun_out/
