cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 4 5 8 9 11; do echo "== ABL=$abl (W4 M1)"; GWEN_K4_ABL=$abl python tools/kbench.py k4 64 2>&1 | grep K4; done
for abl in 0 1 3 11; do echo "== ABL=$abl (W8 M1)"; GWEN_K4_WAVES=8 GWEN_K4_ABL=$abl python tools/kbench.py k4 64 2>&1 | grep K4; done
