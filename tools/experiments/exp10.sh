cd $GRAFT_REPO_ROOT
for abl in 11 27 59 123 251 187 155; do echo "== ABL=$abl (W4 M1)"; GWEN_K4_ABL=$abl python tools/kbench.py k4 64 2>&1 | grep K4; done
