"""print a rocprofv3 kernel_stats.csv per train step:  python scripts/show_stats.py gpurun_out/qs_x_2d_kernel_stats.csv [steps incl. warm-up = 4] [rows = 40]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = 0.0
for r in rows:
    tot += int(r["TotalDurationNs"]) / nst / 1e6
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / nst:6.1f}/step {int(r['TotalDurationNs']) / nst / 1e6:7.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
print(f"sum of kernel durations: {tot:.2f} ms/step")
