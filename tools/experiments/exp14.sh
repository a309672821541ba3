cd $GRAFT_REPO_ROOT
GWEN_K4_BLK=1 GWEN_K4_SPLIT=1 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -k "fused or stack or model or golden" 2>&1 | tail -8
echo "== K4b fp32"; GWEN_K4_BLK=1 python tools/kbench.py k4 32 64 2>&1 | grep K4
echo "== K4b split"; GWEN_K4_BLK=1 GWEN_K4_SPLIT=1 python tools/kbench.py k4 32 64 2>&1 | grep K4
GWEN_K4_BLK=1 GWEN_K4_SPLIT=1 python bench.py --steps 400 --warmup 20 --event-stride 100000 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['cpu_baseline']['gpu_vs_oracle_rel_err'])"
