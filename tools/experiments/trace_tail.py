"""Print the last N kernel records of a rocprofv3 --kernel-trace csv: python trace_tail.py <dir> [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = list(csv.DictReader(open(f)))
prev_end = None
for r in rows[-n:]:
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f} us"
    print(f"{name:34s} {(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {gap}")
    prev_end = e
