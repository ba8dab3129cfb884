"""Timeline of the LAST build in a rocprofv3 kernel trace (dev tool): per launch start offset, duration, gap to the previous one."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ibh::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_first2")]
i0 = idx[-1]
# include the memsets/fills that precede k_first2 of the last build (after the previous build's k_scale/k_weights)
j = i0
while j > 0 and not names[j - 1].startswith(("k_scale", "k_weights", "k_col_sums_long")):
    j -= 1
t0 = int(rows[j]["Start_Timestamp"]); prev_end = t0
tot = 0
for r, n in zip(rows[j:], names[j:]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:50]))
    prev_end = e; tot += e - s
print("launches %d, span %.1f us, kernel time %.1f us" % (len(rows) - j, (prev_end - t0) / 1e3, tot / 1e3))
