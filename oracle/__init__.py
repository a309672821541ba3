"""CPU oracle (test infrastructure).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; nothing under gwen_amd/ does."""
