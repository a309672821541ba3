cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== K4 split"; python tools/kbench.py all 16 32 64 128 2>&1 | grep -E "K2|K4"
echo "== K4 exact"; KB_EXACT=1 python tools/kbench.py k4 16 32 64 128 2>&1 | grep K4
python bench.py --steps 400 --warmup 20 > gpurun_out/bench4.json; cat gpurun_out/bench4.json
