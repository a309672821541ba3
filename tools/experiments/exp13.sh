cd $GRAFT_REPO_ROOT
GWEN_K4_BLK=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or stack or model or golden" 2>&1 | tail -3
echo "== K4 v2"; python tools/kbench.py k4 16 32 64 2>&1 | grep K4
echo "== K4b"; GWEN_K4_BLK=1 python tools/kbench.py k4 16 32 64 2>&1 | grep K4
GWEN_K4_BLK=1 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --event-stride 100000 | cut -c1-200
GWEN_K4_BLK=1 python bench.py --steps 400 --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['all_kernels_us'])"
