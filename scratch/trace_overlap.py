"""Per-stream overlap of the kernels of the last N ms of a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_fa" in r["Kernel_Name"] or "scan" in r["Kernel_Name"]]
t_end = max(int(r["End_Timestamp"]) for r in rows)
win = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 400000]
t0 = min(int(r["Start_Timestamp"]) for r in win)
for r in sorted(win, key=lambda r: int(r["Start_Timestamp"])):
    print("%8.1f %8.1f  q%-4s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
