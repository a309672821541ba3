cd $GRAFT_REPO_ROOT
for br in 32 64 128; do echo "== BR=$br"; GWEN_K4_BR=$br python tools/kbench.py k4 16 32 64 128 2>&1 | grep K4; done
