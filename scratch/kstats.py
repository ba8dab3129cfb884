"""Print a rocprofv3 kernel_stats.csv compactly: short name, calls, avg us, total us, %."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("ibh::", "")
    print("%-44s %5d  avg %9.1f us  total/build %9.1f us  %5.1f %%" % (name[:44], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / div, float(r["Percentage"])))
